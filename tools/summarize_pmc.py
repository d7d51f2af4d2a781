#!/usr/bin/env python3
"""Summarise rocprofv3 output for the bench: per-kernel average duration (kernel_stats.csv) joined with per-launch
FETCH_SIZE / WRITE_SIZE from two separate --pmc passes (counter_collection.csv).

Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half the bytes of a wide coalesced streaming read, so the corrected read traffic is 2 x FETCH_SIZE
(upper bound for narrow accesses); WRITE_SIZE is taken as is (uncalibrated).
usage: summarize_pmc.py <dir> [<out.md>]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[a-z0-9_]+|__amd_rocclr_[A-Za-z]+)", name)
    base = m.group(1) if m else name[:40]
    t = re.search(r"<([^>]*)>", name)
    return base + ("<" + t.group(1).replace("unsigned char", "u8").replace("short", "i16").replace("(anonymous namespace)::", "") + ">" if t else "")


def counters(path):
    acc = defaultdict(list)
    try:
        with open(path) as f:
            for r in csv.DictReader(f):
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    except FileNotFoundError:
        pass
    return acc


def main():
    d = sys.argv[1]
    stats = {}
    with open(d + "/kt_kernel_stats.csv") as f:
        for r in csv.DictReader(f):
            stats[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]), float(r["Percentage"]))
    fe, wr = counters(d + "/fetch_counter_collection.csv"), counters(d + "/write_counter_collection.csv")
    lines = ["| kernel | calls | avg us | % time | FETCH_SIZE KiB/launch | read MB/launch (2x corrected) | WRITE_SIZE KiB/launch | write MB/launch |",
             "|---|---|---|---|---|---|---|---|"]
    for k, (calls, avg, pct) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
        f_ = sum(fe[k]) / len(fe[k]) if fe.get(k) else float("nan")
        w_ = sum(wr[k]) / len(wr[k]) if wr.get(k) else float("nan")
        lines.append("| %s | %d | %.1f | %.2f | %.0f | %.1f | %.0f | %.1f |" % (k, calls, avg / 1e3, pct, f_, 2 * f_ * 1024 / 1e6, w_, w_ * 1024 / 1e6))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    if len(sys.argv) > 4:  # <traffic.json> <steps of the kernel-trace run incl. warmup>
        import json
        # steps of the kernel-trace run (warm-up + timed + the bench's instrumented extra steps); "auto" = the call count
        # of a kernel that runs exactly once per step
        once = [k for k in stats if k.startswith("k_threshold16")]  # (k_flood_clear no longer runs in the fused start)
        nsteps = stats[once[0]][0] if sys.argv[4] == "auto" else int(sys.argv[4])
        stage_of = lambda k: ("threshold" if k.startswith("k_threshold") else "marching_cubes" if k.startswith("k_mc_")
                              else "region_grow" if k.startswith(("k_flood_", "k_ccl_", "k_scan_")) and not k.startswith("k_flood_count")
                              else None)
        traffic = {}
        for k, (calls, avg, pct) in stats.items():
            st = stage_of(k)
            if st is None or not fe.get(k) or not wr.get(k):
                continue
            per_launch = (2 * sum(fe[k]) / len(fe[k]) + sum(wr[k]) / len(wr[k])) * 1024
            # launches per STEP: a kernel that runs once per step shows nsteps calls plus the few of bench.py's extra
            # calls outside the steps (the end-to-end download runs marching cubes once more): those must not be spread
            # over the steps (14 launches / 13 steps made the figure 8 % high); only kernels that run several
            # times per step (the flood's rounds) are averaged
            lps = calls / nsteps if (calls >= 2 * nsteps or calls < nsteps) else 1.0
            traffic[st] = traffic.get(st, 0.0) + per_launch * lps
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import src_sha16
        json.dump({"unit": "bytes per bench step (2 x FETCH_SIZE + WRITE_SIZE, KiB -> B), kernels of the stage only",
                   "config": "grow_mc", "src_sha16": src_sha16(),
                   "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)",
                   "traffic_bytes_per_step": {k: round(v) for k, v in traffic.items()}}, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
