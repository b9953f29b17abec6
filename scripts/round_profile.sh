# usage: bash scripts/round_profile.sh <tag>   (on the GPU box, from the repo root): the round's bench line + kernel stats
set -e
TAG=$1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cut -c1-260 gpurun_out/${TAG}_bench.json
rm -rf gpurun_out/${TAG}_prof
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 -u bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_prof.log 2>&1
ls gpurun_out/${TAG}_prof/*/ | head
