#!/usr/bin/env python3
"""Throughput bench of the VQ-VAE-GAN training step (BASELINE.json configs[1]:
num_embeddings=512, embedding_dim=64, commitment_cost=0.25, batch 64 per GPU, bf16, 256x256
synthetic RGB) on N MI355X of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line: whole-job train images/s (weak scaling: 64 images per GPU), plus
 * "roofline": algorithmic FLOPs of the dominant kernel (the gather-GEMM conv variant with the largest total
   time in the timed region; the library reports which variant each launch used) divided by its
   HIP-event-timed launch durations, against the 2.5 PFLOP/s dense bf16 MFMA peak; "traffic" = its
   HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/r*_pmc_traffic.json);
 * "cpu_baseline": the CPU oracle's train step (oracle/vaegan_oracle.py, a port of the reference
   step) timed on this host on a bounded sample (N=1, rank 0 only): one same-size warm-up, median of 3;
   "cpu_baseline_configs0": the same protocol on BASELINE.json configs[0] (beta-VAE-GAN, batch 16).
"""
import argparse
import glob
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
GF_PER_IMG_LITERAL = 446.9         # reference schedule, fwd F + bwd 2F
GF_PER_IMG_NECESSARY = 391.7       # same minus D-wgrad in the G phase and the repeated encoder fwd


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(arch, batch, size, threads, repeats=3):
    """the CPU oracle's train step (oracle/vaegan_oracle.py, a port of the reference step: the
    reference's Python cannot travel to the GPU box), fp32, ``threads`` torch threads: ONE untimed warm-up step of
    the same size, then the MEDIAN of ``repeats`` timed steps.  arch 'vq' = the workload of the metric;
    'beta' = BASELINE.json configs[0] (batch 16, 256x256, one train step on the CPU path; NON-REFERENCE shim D1:
    the reference's own beta step raises)."""
    from oracle import vaegan_oracle as O
    from vae_gan_amd.testing import fill_state_dict, synthetic_images, synthetic_eps
    torch.set_num_threads(threads)
    imgs = synthetic_images(batch, size, seed=7)
    opt = {"g": {}, "d": {}}
    if arch == "vq":
        sd = fill_state_dict(O.vq_template())
        step = lambda: O.vq_train_step(sd, imgs, O.VQ_DEFAULT_CFG, opt)
        what = "VQ-VAE-GAN"
    else:
        sd = fill_state_dict(O.beta_template())
        eps = synthetic_eps((batch, 128), 9)
        step = lambda: O.beta_train_step(sd, imgs, eps, O.BETA_DEFAULT_CFG, opt)
        what = "beta-VAE-GAN (configs[0]; shim D1: area-downsampled target)"
    step()                                           # warm-up, same size
    times = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    return {"value": batch / dt, "unit": "images/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"{what} oracle train step (D+G, fp32, torch CPU), batch {batch} at {size}x{size}: 1 warm-up step, "
                      f"median of {repeats} timed steps = {dt:.2f} s (all: {', '.join(f'{t:.2f}' for t in times)})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--arch", default="vq", choices=["vq", "beta", "hier"],
                    help="vq = BASELINE.json configs[1] (the driver's line); beta / hier = configs[2] / configs[3] per GPU, "
                         "with the documented non-reference shims D1 / D2 (the reference step raises without them)")
    ap.add_argument("--num-embeddings", type=int, default=512)
    ap.add_argument("--embedding-dim", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--kernel-shapes", action="store_true", help="key the live kernel timings by layer shape too")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs (the HIP path has no CPU fallback)")
    # VG_BENCH_SHARE_GPU=1: a rehearsal of the multi-rank path on a one-GPU box (every rank on cuda:0, gloo, gradients
    # staged through the host); its numbers mean nothing.  The driver's runs use one GPU per rank over RCCL.
    share = os.environ.get("VG_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from vae_gan_amd import ops
    from vae_gan_amd import train as T
    from vae_gan_amd.train import DataParallel, build_optimizers

    ops.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    torch.manual_seed(42)
    if args.arch == "vq":
        from vae_gan_amd.vq_vae_gan.config import Config
        from vae_gan_amd.vq_vae_gan.model import VQVAEGAN
        cfg = Config()
        cfg.num_embeddings, cfg.embedding_dim = args.num_embeddings, args.embedding_dim
        model_cls, step_fn = VQVAEGAN, T.vq_train_step
        workload = (f"VQ-VAE-GAN train step (D step + G step), num_embeddings={cfg.num_embeddings} "
                    f"embedding_dim={cfg.embedding_dim} commitment_cost=0.25")
    elif args.arch == "beta":
        from vae_gan_amd.beta_vae_gans.config import Config
        from vae_gan_amd.beta_vae_gans.model import BetaVAEGAN
        cfg = Config()
        cfg.recon_target_policy = "area"
        model_cls, step_fn = BetaVAEGAN, T.beta_train_step
        workload = "beta-VAE-GAN train step (latent 128; NON-REFERENCE shim D1: area-downsampled target)"
    else:
        from vae_gan_amd.hierarchical_vae_gan.config import Config
        from vae_gan_amd.hierarchical_vae_gan.model import HierarchicalVAEGAN
        cfg = Config()
        cfg.recon_target_policy, cfg.hier_decoder_policy = "area", "nearest"
        model_cls, step_fn = HierarchicalVAEGAN, T.hier_train_step
        workload = "hierarchical VAE-GAN train step (3 levels, self-attention; NON-REFERENCE shims D2 + D1)"
    cfg.image_size, cfg.batch_size = args.size, args.batch
    model = model_cls(cfg).cuda().train()
    if args.arch == "hier":     # the reference initialises every attention gate to 0 (identity block): open them
        for n_, p_ in model.named_parameters():
            if n_.endswith("gamma"):
                p_.data.fill_(0.1)

    def vq_train_step(model, images, opt_g, opt_d, dp):
        out = step_fn(model, images, opt_g, opt_d, dp)
        return out[0], out[1], None
    dp = DataParallel() if world > 1 else None
    if dp:
        dp.broadcast_model(model)
    opt_g, opt_d, _, _ = build_optimizers(model, cfg)
    g = torch.Generator(device="cuda")
    g.manual_seed(42 + rank)
    images = torch.rand(args.batch, 3, args.size, args.size, generator=g, device="cuda")   # resident in HBM

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        vq_train_step(model, images, opt_g, opt_d, dp)
    # Kernel timing.  One extra UNTIMED step with a HIP event pair around every conv launch gives the per-kernel
    # table and names the dominant gather-GEMM variant; in the timed region only that variant's launches carry
    # events (a pair around all ~100 conv launches of a step costs ~3 % of the step), so the roofline numbers are
    # measured live over the timed region without distorting the throughput they sit next to.
    prof = survey = None
    if not args.no_kernel_timing:
        survey = ops.KernelProfile(shapes=args.kernel_shapes)
        ops.PROFILE[0] = survey
        vq_train_step(model, images, opt_g, opt_d, dp)
        torch.cuda.synchronize()
        ops.PROFILE[0] = None
        per = {}
        for name, _, e0, e1 in survey.records:
            base = name.split(" ")[0]
            if base.startswith("igemm"):
                per[base] = per.get(base, 0.0) + e0.elapsed_time(e1)
        dom_kernel = max(per, key=per.get)
        prof = ops.KernelProfile(only_tags={t for t, k in survey.tag_kernel.items() if k == dom_kernel})
        ops.PROFILE[0] = prof
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        d_loss, g_loss, _ = vq_train_step(model, images, opt_g, opt_d, dp)
    host_dt = time.perf_counter() - t0      # all launches of the timed steps enqueued (the GPU is still working)
    barrier()
    dt = time.perf_counter() - t0
    ops.PROFILE[0] = None
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    finite = bool(torch.isfinite(g_loss["total"]).item() and torch.isfinite(d_loss["total"]).item())
    # Host cost of a step by itself: one more (untimed) step enqueued into an EMPTY queue.  host_dt above is mostly
    # back-pressure: with ten steps in flight the launch calls block until the GPU has retired older dispatches (cProfile
    # with the autograd engine on the calling thread, tests/gpu_host_profile.py: 70 % of it sits inside the launch calls of
    # the backward pass), so it tracks the GPU's time, not Python's.
    barrier()
    t1 = time.perf_counter()
    vq_train_step(model, images, opt_g, opt_d, dp)
    host_one = time.perf_counter() - t1
    barrier()

    if rank == 0:
        imgs_per_s = world * args.batch * args.steps / dt
        # SURVEY.md §8(a) necessary work per image and train step (known for the 256x256 default shapes only)
        gf_img = {"vq": GF_PER_IMG_NECESSARY, "beta": 209.2}.get(args.arch) if (
            args.size == 256 and args.num_embeddings == 512 and args.embedding_dim == 64) else None
        roof = None
        kernels = {}
        if prof is not None:
            for name, d in survey.summary().items():     # the untimed survey step (all conv launches)
                kernels[name] = {"launches_per_step": d["launches"], "avg_ms": d["ms"] / d["launches"],
                                 "tflops": d["flops"] / d["ms"] / 1e9}
            summ = prof.summary()                         # the dominant variant, over the timed region
            dom = max((n for n in summ if n.startswith("igemm")), key=lambda n: summ[n]["ms"])
            ach = summ[dom]["flops"] / summ[dom]["ms"] / 1e9
            # HBM-side traffic of that kernel per launch: measured in separate rocprofv3 --pmc passes over this same
            # command (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE; profiles/make_pmc_summary.py)
            traffic, tsrc = None, None
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
                ent = json.load(open(f)).get(dom)
                if ent:
                    traffic, tsrc = ent["traffic_bytes_per_launch"], os.path.basename(f)
                    break
            roof = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": tsrc,
                    "launches_per_step": summ[dom]["launches"] / args.steps,
                    "avg_launch_ms": summ[dom]["ms"] / summ[dom]["launches"],
                    "avg_launch_gflop": summ[dom]["flops"] / summ[dom]["launches"] / 1e9}
        out = {
            "metric": "train images/sec (256x256, whole node)", "value": imgs_per_s, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{workload}, {args.size}x{args.size} RGB, random-init weights",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}"},
            "step_tflops_necessary": imgs_per_s * gf_img / 1e3 / world if gf_img else None,
            "step_frac_of_bf16_peak": imgs_per_s * gf_img / 1e3 / world / PEAK_BF16_TFLOPS if gf_img else None,
            "losses_finite": finite,
            # host time to enqueue a step (rank 0): the margin before the step becomes launch-bound on a slower host
            "host_enqueue_ms_per_step": 1000.0 * host_dt / args.steps,
            # ... and the same for one step enqueued into an empty queue (no back-pressure): what Python + ctypes cost
            "host_ms_per_step_empty_queue": 1000.0 * host_one,
            "roofline": roof, "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline and args.arch == "vq":
            threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
            out["cpu_baseline"] = cpu_baseline("vq", 16, args.size, threads)
            if args.size == 256:      # BASELINE.json configs[0]: the reference's own CPU-runnable case
                out["cpu_baseline_configs0"] = cpu_baseline("beta", 16, 256, threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
