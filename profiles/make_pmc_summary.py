#!/usr/bin/env python3
"""Reduce two rocprofv3 PMC passes over bench.py (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, each in
its own run, with --kernel-trace only) to per-kernel HBM traffic per launch.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- \
        python3 -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing
    rocprofv3 --pmc WRITE_SIZE ... -d gpurun_out/pmc_write -- (same command)
    python3 profiles/make_pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

Both counters are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide streaming reads at 64 bytes
(MI355X_MICROARCH.md, HBM section), so reads are doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.
Infinity-Cache hits are counted, so this is memory-side traffic of the L2, an upper bound of HBM bytes.
"""
import collections
import csv
import glob
import json
import re
import sys


def label(mangled: str) -> str:
    """igemm kernels -> the label vg_last_conv_kernel() reports; everything else: the name as is."""
    m = re.match(r"_Z\d+(igemm3?2?)_kernelI(DF16b|f)((?:Li\d+E)+)E", mangled)
    if not m:
        return mangled.split("(")[0][:80]
    v = [int(x) for x in re.findall(r"Li(\d+)E", m.group(3))]
    t = "bf16" if m.group(2) == "DF16b" else "f32"
    kind = m.group(1)
    if kind == "igemm3":
        return f"igemm3<{t},{v[0]},{v[1]},{v[2]},{v[3]},{v[4]}x{v[5]}>"
    return f"{kind}<{t}," + ",".join(map(str, v)) + ">"


def load(d):
    out = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            out[label(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
    return out


def main():
    fetch, write, dst = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    res = {}
    for k in fetch:
        f, w = fetch[k], write.get(k, [0.0])
        res[k] = {"launches": len(f), "read_bytes_per_launch": 2.0 * sum(f) / len(f), "write_bytes_per_launch": sum(w) / len(w),
                  "traffic_bytes_per_launch": 2.0 * sum(f) / len(f) + sum(w) / len(w),
                  "max_read_bytes": 2.0 * max(f), "max_write_bytes": max(w)}
    json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
    for k in sorted(res, key=lambda k: -res[k]["traffic_bytes_per_launch"] * res[k]["launches"])[:12]:
        print(f"{k:60s} {res[k]['launches']:5d} launches  {res[k]['traffic_bytes_per_launch'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
