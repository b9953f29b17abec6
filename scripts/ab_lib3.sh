# three-way same-box A/B: bash scripts/ab_lib3.sh <tagA> <tagB> [rounds]
cd $GRAFT_REPO_ROOT
L=vae_gan_amd/libvg_hip.so
cp $L /tmp/new.so
run() { timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), [(k,round(v['avg_ms'],4)) for k,v in d['kernels'].items() if 'igemm4' in k])"; }
for r in $(seq 1 ${3:-2}); do
  cp vae_gan_amd/libvg_hip_$1.so $L && run $1 || exit 1
  cp vae_gan_amd/libvg_hip_$2.so $L && run $2 || exit 1
  cp /tmp/new.so $L && run new || exit 1
done
cp /tmp/new.so $L
