# same-box A/B of wgrad3 builds: bash scripts/wgrad_late_ab.sh <tag> [<tag> ...]
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/tests VG_WGP_FAST=1
for r in 1 2; do
  echo "== current"; timeout -k 10 120 python -u tests/gpu_wgrad_perf.py || exit 1
  for t in "$@"; do echo "== $t"; VG_LIB_EXP=$t timeout -k 10 120 python -u tests/gpu_wgrad_perf.py || exit 1; done
done
