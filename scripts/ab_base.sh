cd $GRAFT_REPO_ROOT
L=vae_gan_amd/libvg_hip.so
cp $L /tmp/new.so
for r in 1 2; do
  cp vae_gan_amd/libvg_hip_base.so $L && timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['ms_per_step'])" || exit 1
  cp /tmp/new.so $L && timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['ms_per_step'], [ (k,round(v['avg_ms'],4)) for k,v in d['kernels'].items() if 'igemm3s2' in k])" || exit 1
done
cp /tmp/new.so $L
