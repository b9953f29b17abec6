#!/usr/bin/env python3
"""Reduce rocprofv3 PMC passes over bench.py to per-kernel figures: HBM-side traffic per launch, achieved GB/s and
MFMA utilisation.  Each pass is a run of its own with --kernel-trace only (counters are never combined with other
trace domains, and FETCH_SIZE / WRITE_SIZE do not fit one pass):

    B="python3 -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing"
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- $B
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- $B
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
              -d gpurun_out/pmc_mfma -- $B
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- $B      (un-profiled durations)
    python3 profiles/make_pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma \
            gpurun_out/prof profiles/r01_pmc.json

Units and gfx950 corrections (MI355X_MICROARCH.md, HBM / rocprofv3 sections): FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE tallies the 128-byte requests of wide streaming reads at 64 bytes, so reads are DOUBLED; WRITE_SIZE is
exact for 16-byte-per-lane stores and float atomics.  Infinity-Cache hits are counted: this is the memory-side
traffic of the L2, an upper bound of HBM bytes.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD summed over the
chip (32 per v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the kernel's cycle count is
GRBM_GUI_ACTIVE / 8 and MFMA utilisation = MFMA_BUSY / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).  GB/s uses the average
duration of the kernel in the UN-profiled --stats run (counter runs serialise dispatches and clock differently).
"""
import collections
import csv
import glob
import json
import re
import sys

N_SIMD = 256 * 4


def label(mangled: str) -> str:
    """igemm kernels -> the label vg_last_conv_kernel() reports; everything else: a readable short name."""
    m = re.match(r"_Z\d+(igemm3?2?)_kernelI(DF16b|f)((?:Li\d+E)+)E", mangled)
    if m:
        v = [int(x) for x in re.findall(r"Li(\d+)E", m.group(3))]
        t = "bf16" if m.group(2) == "DF16b" else "f32"
        kind = m.group(1)
        if kind == "igemm3":
            return f"igemm3<{t},{v[0]},{v[1]},{v[2]},{v[3]},{v[4]}x{v[5]}>"
        return f"{kind}<{t}," + ",".join(map(str, v)) + ">"
    m = re.match(r"_Z\d+([a-z0-9_]+?)I(DF16b|f)(.*?)E[vP]", mangled)
    if m:
        flags = "".join("1" if b == "1" else "0" for b in re.findall(r"Lb([01])E", m.group(3)))
        return m.group(1) + "<" + ("bf16" if m.group(2) == "DF16b" else "f32") + ("," + flags if flags else "") + ">"
    return re.sub(r"^void ", "", mangled).split("(")[0][:80]


def load_counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            out[label(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def load_durations(d):
    out = {}
    for f in glob.glob(d + "/*/*_kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            k = label(r["Name"])
            calls, tot = out.get(k, (0, 0.0))
            out[k] = (calls + int(r["Calls"]), tot + float(r["TotalDurationNs"]))
    return {k: tot / calls for k, (calls, tot) in out.items()}


def main():
    fetch, write, mfma = (load_counters(a) for a in sys.argv[1:4])
    dur, dst = load_durations(sys.argv[4]), sys.argv[5]
    res = {}
    for k in fetch:
        f = [v * 1024.0 for v in fetch[k]["FETCH_SIZE"]]
        w = [v * 1024.0 for v in write.get(k, {}).get("WRITE_SIZE", [0.0])]
        e = {"launches": len(f), "read_bytes_per_launch": 2.0 * sum(f) / len(f), "write_bytes_per_launch": sum(w) / len(w)}
        e["traffic_bytes_per_launch"] = e["read_bytes_per_launch"] + e["write_bytes_per_launch"]
        if k in dur:
            e["avg_ns_unprofiled"] = dur[k]
            e["hbm_side_GBps"] = e["traffic_bytes_per_launch"] / dur[k]
            e["frac_of_8TBps"] = e["hbm_side_GBps"] / 8000.0
        m = mfma.get(k)
        if m and sum(m["GRBM_GUI_ACTIVE"]) > 0:
            e["mfma_util"] = sum(m["SQ_VALU_MFMA_BUSY_CYCLES"]) / (N_SIMD * sum(m["GRBM_GUI_ACTIVE"]) / 8.0)
        res[k] = e
    json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
    print(f"{'kernel':58s} {'n':>4s} {'MB/launch':>10s} {'GB/s':>7s} {'MFMA util':>9s}")
    for k in sorted(res, key=lambda k: -res[k].get("avg_ns_unprofiled", 0) * res[k]["launches"])[:30]:
        e = res[k]
        print(f"{k:58s} {e['launches']:4d} {e['traffic_bytes_per_launch'] / 1e6:10.1f} {e.get('hbm_side_GBps', 0):7.0f} "
              f"{100 * e.get('mfma_util', 0):8.1f}%")


if __name__ == "__main__":
    main()
