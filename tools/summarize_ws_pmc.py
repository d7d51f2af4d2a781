#!/usr/bin/env python3
"""HBM traffic of one watershed flood from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`bench.py --config watershed|watershed_sk --size 512`: every kernel between the cost image and the merge, summed, divided by
the number of floods in the run (= launches of the labels kernel, which runs once per flood).  Corrections as in
tools/summarize_pmc.py (MI355X_MICROARCH.md: KiB units, FETCH_SIZE x 2 for wide streaming reads -- an UPPER bound for the
narrow gathers that dominate here).
usage: summarize_ws_pmc.py <config> <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [<out.md>] [<size>]
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[a-z0-9_]+|__amd_rocclr_[A-Za-z]+|rocprim[A-Za-z_:0-9]*)", name)
    return m.group(1)[:48] if m else name[:48]


def load(path):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            a = acc[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    config, fe, wr = sys.argv[1], load(sys.argv[2]), load(sys.argv[3])
    size = int(sys.argv[6]) if len(sys.argv) > 6 else 512
    # (k_wsa_* / k_ska_* / k_flood_*: the cost map's level floods on bit planes run inside the flood too)
    flood_kernels = [k for k in fe if k.startswith(("k_ws_", "k_wsa_", "k_sk_", "k_ska_", "k_flood_", "rocprim", "k_mscan", "k_mailbox"))]
    lab = [k for k in fe if k in ("k_ws_labels", "k_sk_labels")]
    nfloods = fe[lab[0]][0]
    rows, total = [], 0.0
    for k in sorted(flood_kernels, key=lambda k: -(2 * fe[k][1] + wr.get(k, [0, 0.0])[1])):
        rd, wt = 2 * fe[k][1] * 1024 / nfloods, wr.get(k, [0, 0.0])[1] * 1024 / nfloods
        total += rd + wt
        rows.append((k, fe[k][0] / nfloods, rd / 1e6, wt / 1e6))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import src_sha16
    json.dump({"unit": "bytes per flood (2 x FETCH_SIZE + WRITE_SIZE, KiB -> B), every kernel of the flood", "config": config,
               "size": size, "src_sha16": src_sha16(), "floods_in_run": nfloods,
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --config %s --size %d" % (config, size),
               "traffic_bytes_per_step": {"flood": round(total)}}, open(sys.argv[4], "w"), indent=1)
    md = ["| kernel | launches / flood | read MB / flood (2 x FETCH_SIZE) | written MB / flood |", "|---|---|---|---|"]
    md += ["| %s | %.1f | %.1f | %.1f |" % r for r in rows]
    md.append("| **total** | | **%.1f MB** (algorithmic: 7 B/voxel = %.1f MB at %d^3) | |" % (total / 1e6, 7 * size ** 3 / 1e6, size))
    print("\n".join(md))
    if len(sys.argv) > 5:
        open(sys.argv[5], "w").write("\n".join(md) + "\n")


if __name__ == "__main__":
    main()
