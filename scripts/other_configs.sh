# the other BASELINE configurations and the two-rank rehearsal of bench.py on the current tree (GPU box):
#   bash scripts/other_configs.sh <tag>   ->  gpurun_out/<tag>_bench_{beta,hier,cfg5}.json, <tag>_rehearsal_2ranks_shared_gpu_vq.json
set -e
TAG=$1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
timeout -k 10 200 python3 bench.py --arch beta --batch 32 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_beta.json 2> gpurun_out/${TAG}_bench_beta.err
timeout -k 10 200 python3 bench.py --arch hier --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_hier.json 2> gpurun_out/${TAG}_bench_hier.err
timeout -k 10 200 python3 bench.py --batch 16 --size 512 --num-embeddings 8192 --embedding-dim 256 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg5.json 2> gpurun_out/${TAG}_bench_cfg5.err
VG_BENCH_SHARE_GPU=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_rehearsal_2ranks_shared_gpu_vq.json 2> gpurun_out/${TAG}_rehearsal.err
for f in beta hier cfg5; do cut -c1-160 gpurun_out/${TAG}_bench_$f.json; done
tail -c 400 gpurun_out/${TAG}_rehearsal_2ranks_shared_gpu_vq.json | cut -c1-300
