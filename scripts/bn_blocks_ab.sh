cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/tests
for cfg in "2048 16 4096 16 2048 16" "8192 8 8192 8 4096 16" "16384 4 16384 4 4096 8" "32768 4 32768 4 8192 8" "8192 8 8192 8 2048 16"; do
  set -- $cfg
  echo "== apply cap $1 rpt $2 | fwd cap $3 rpt $4 | reduce cap $5 rpt $6"
  VG_BN_APPLY_CAP=$1 VG_BN_APPLY_RPT=$2 VG_BN_FWD_CAP=$3 VG_BN_FWD_RPT=$4 VG_BN_RED_CAP=$5 VG_BN_RED_RPT=$6 timeout -k 10 100 python -u tests/gpu_bn_perf.py 2>/dev/null | grep -E "bn_act_fwd |bn_act_fwd\+res|bn_bwd_apply  |bn_bwd_reduce" || exit 1
done
