# same-box A/B of the whole step with / without an environment switch: bash scripts/env_step_ab.sh VAR=value [pairs]
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${2:-3}); do
  env $1 timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])" || exit 1
  timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'])" || exit 1
done
