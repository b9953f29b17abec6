"""Throughput of the device input transform (beta_vae_gans/data.py:106-112) on one MI355X.

Workload: a batch of decoded Kodak-sized uint8 images (512x768x3) resident in HBM -> flips, RandomResizedCrop
(scale 0.8-1), Pillow-bilinear resize to 256x256, ToTensor: float32 [B,3,256,256].  One step = one batch.
`roofline`: HBM; algorithmic bytes per image = crop bytes read + float32 output written (the uint8 horizontal-pass
intermediate, rows x 256 x 3 written and read once, is the implementation's own traffic).  cpu_baseline: the numpy
oracle of the same arithmetic on a few images."""
import argparse
import json
import time

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    import vae_gan_amd.lib as L
    from vae_gan_amd.data import DeviceTransform, draw_train_params
    L.load()
    rng = np.random.default_rng(0)
    base = [torch.from_numpy(rng.integers(0, 256, (args.height, args.width, 3)).astype(np.uint8)) for _ in range(8)]
    imgs = [base[b % 8].cuda().clone() for b in range(args.batch)]
    g = torch.Generator().manual_seed(42)
    params = [draw_train_params(args.height, args.width, generator=g) for _ in range(args.batch)]
    tf = DeviceTransform(args.size)
    times = []
    for it in range(args.warmup + args.steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = tf(imgs, params=params)
        e1.record()
        torch.cuda.synchronize()
        if it >= args.warmup:
            times.append(e0.elapsed_time(e1))
    ms = float(np.mean(times))
    alg = sum(p[4] * p[5] * 3 for p in params) + args.batch * 3 * args.size * args.size * 4
    line = {"metric": "transformed training images/sec (flip+crop+resize+ToTensor)", "value": args.batch / (ms * 1e-3),
            "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.batch} x {args.height}x{args.width} uint8 RGB -> {args.size}x{args.size} float32"},
            "roofline": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": alg / (ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                         "algorithmic_bytes_per_step": alg}}
    if not args.no_cpu_baseline:
        from oracle import augment_oracle as A
        m = 4
        t0 = time.perf_counter()
        for b in range(m):
            ref = A.train_transform(base[b % 8].numpy(), args.size, params[b])
            assert torch.equal(out[b].cpu(), ref)
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": m / dt, "unit": "images/s", "cores": 1, "kind": "port",
                                "sample": f"numpy oracle of the same arithmetic on {m} images (checked equal): {dt:.2f} s"}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
