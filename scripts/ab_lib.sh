# same-box A/B of the whole step against another build of the library: bash scripts/ab_lib.sh <tag> [pairs]   (vae_gan_amd/libvg_hip_<tag>.so)
cd $GRAFT_REPO_ROOT
L=vae_gan_amd/libvg_hip.so
cp $L /tmp/new.so
for r in $(seq 1 ${2:-2}); do
  cp vae_gan_amd/libvg_hip_$1.so $L && timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], [(k,round(v['avg_ms'],4)) for k,v in d['kernels'].items() if 'igemm4' in k])" || exit 1
  cp /tmp/new.so $L && timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['ms_per_step'], [(k,round(v['avg_ms'],4)) for k,v in d['kernels'].items() if 'igemm4' in k])" || exit 1
done
cp /tmp/new.so $L
