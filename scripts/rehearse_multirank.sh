# Rehearsal of bench.py's multi-rank path on a ONE-GPU box: two ranks share cuda:0 (VG_BENCH_SHARE_GPU=1: gloo, gradients
# staged through the host), so the NUMBERS MEAN NOTHING; what it shows is that the torchrun launch, the rendezvous, the
# broadcast, the bucketed exchange from the backward boundaries and rank 0's JSON line all work for the three
# architectures.  The real runs (one GPU per rank, RCCL) are the driver's.
set -e
cd $GRAFT_REPO_ROOT
export VG_BENCH_SHARE_GPU=1
port=29621
for arch in vq beta hier; do
  b=64; [ $arch = beta ] && b=32; [ $arch = hier ] && b=16
  timeout -k 10 280 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --arch $arch --batch $b > gpurun_out/r3_rehearsal_$arch.json 2> gpurun_out/r3_rehearsal_$arch.err
  echo "$arch done: $(cut -c1-200 gpurun_out/r3_rehearsal_$arch.json)"
  port=$((port + 1))
done
