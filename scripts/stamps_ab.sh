# igemm4 stage stamps of several -DVG_ABLATE=1 builds on one box: bash scripts/stamps_ab.sh <tag> ...
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/tests
for r in 1 2; do for t in "$@"; do echo "== $t"; VG_LIB_EXP=$t timeout -k 10 120 python -u tests/gpu_igemm4_stamps.py 2>&1 | grep -v amdgpu.ids || exit 1; done; done
