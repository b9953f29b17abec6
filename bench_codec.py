"""Throughput of the VQ index entropy coder (vq_vae_gan/model.py:337-576) on one MI355X.

Workload: the index maps of a batch of 256x256 images (32x32 = 1024 indices per image, K = 512 codes, a skewed usage
histogram), one stream per image, indices resident in HBM.  One "step" = histogram + table + encode of the batch,
then table + decode.  Prints one JSON line (symbols/s); `cpu_baseline` is the oracle (a port of the reference's
pure-Python coder, the thing compress_image runs today) on a bounded sample.  The coder is a serial dependency chain
per stream, so the bound is latency, not HBM: `roofline` is reported against HBM for completeness and is tiny by
construction; the number that matters is symbols/s per stream (chain speed) x streams in flight.
"""
import argparse
import json
import time

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4096, help="images per batch (one stream each)")
    ap.add_argument("--symbols", type=int, default=1024)
    ap.add_argument("--num-embeddings", type=int, default=512)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    import vae_gan_amd.lib as L
    from vae_gan_amd.vq_vae_gan.codec import ArithmeticEncoder, calculate_codebook_usage
    lib = L.load()
    K, S, n = args.num_embeddings, args.streams, args.symbols
    rng = np.random.default_rng(42)
    w = 1.0 / np.arange(1, K + 1) ** 1.2
    idx_cpu = torch.from_numpy(rng.choice(K, size=(S, n), p=w / w.sum()))
    idx = idx_cpu.cuda()
    enc = ArithmeticEncoder(K)

    def ev():
        return torch.cuda.Event(enable_timing=True)

    t_enc, t_dec = [], []
    out = nbytes = None
    for it in range(args.warmup + args.steps):
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        probs = calculate_codebook_usage(idx, K)
        out, nbytes = enc.encode_streams(idx, probs)
        e1.record()
        back = enc.decode_streams(out, nbytes, n, probs)
        e2.record()
        torch.cuda.synchronize()
        if it >= args.warmup:
            t_enc.append(e0.elapsed_time(e1))
            t_dec.append(e1.elapsed_time(e2))
    assert torch.equal(back, idx)
    total_bytes = int(nbytes.sum())
    ms_e, ms_d = float(np.mean(t_enc)), float(np.mean(t_dec))
    line = {"metric": "entropy-coded VQ indices/sec (encode incl. histogram+table)", "value": S * n / (ms_e * 1e-3),
            "unit": "symbols/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_e,
            "higher_is_better": True, "dtype": "u32/f64", "data": "synthetic",
            "config": {"workload": f"{S} streams x {n} indices, num_embeddings={K}, zipf(1.2) usage, one wave per stream"},
            "decode_symbols_per_s": S * n / (ms_d * 1e-3), "decode_ms": ms_d,
            "bits_per_symbol": total_bytes * 8 / (S * n),
            "per_stream_symbols_per_s": {"encode": n / (ms_e * 1e-3), "decode": n / (ms_d * 1e-3)},
            "roofline": {"bound": "hbm", "achieved": (S * n * 8 + total_bytes) / (ms_e * 1e-3) / 1e9, "peak": 8000.0,
                         "unit": "GB/s", "frac": (S * n * 8 + total_bytes) / (ms_e * 1e-3) / 1e9 / 8000.0, "traffic": None,
                         "note": "serial chain per stream: latency-bound by construction"}}
    if not args.no_cpu_baseline:
        from oracle import vq_codec_oracle as O
        m = min(S, 64)
        t0 = time.perf_counter()
        p = O.codebook_usage(idx_cpu[:m], K)
        cum = O.cum_table(p, K)
        data = [O.encode_with_table(idx_cpu[i].numpy(), cum) for i in range(m)]
        t1 = time.perf_counter()
        for i in range(min(m, 16)):
            assert np.array_equal(O.decode_with_table(bytes(data[i]), n, cum), idx_cpu[i].numpy())
        t2 = time.perf_counter()
        line["cpu_baseline"] = {"value": m * n / (t1 - t0), "unit": "symbols/s", "cores": 1, "kind": "port",
                                "sample": f"oracle (pure-Python coder, as the reference) encode of {m} streams x {n}: {t1 - t0:.2f} s; "
                                          f"decode of {min(m, 16)} streams: {t2 - t1:.2f} s"}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
