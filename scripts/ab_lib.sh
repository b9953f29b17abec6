# same-box A/B of the whole step against another build of the library: bash scripts/ab_lib.sh <tag> [pairs] [kernel substring]
cd $GRAFT_REPO_ROOT
L=vae_gan_amd/libvg_hip.so
cp $L /tmp/new.so
K=${3:-igemm4}
run() { timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), [(k,round(v['avg_ms'],4)) for k,v in d['kernels'].items() if '$K' in k])"; }
for r in $(seq 1 ${2:-2}); do
  cp vae_gan_amd/libvg_hip_$1.so $L && run $1 || exit 1
  cp /tmp/new.so $L && run new || exit 1
done
cp /tmp/new.so $L
