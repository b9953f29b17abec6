# same-box A/B of the BatchNorm apply passes' grid sizes on the whole step: round-2 values through the environment against the defaults
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  VG_BN_APPLY_CAP=2048 VG_BN_APPLY_RPT=16 VG_BN_FWD_CAP=4096 VG_BN_FWD_RPT=16 timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old grids', d['ms_per_step'])" || exit 1
  timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new grids', d['ms_per_step'])" || exit 1
done
