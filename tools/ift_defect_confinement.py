#!/usr/bin/env python3
"""Where does live scipy.ndimage.watershed_ift leave the algorithm it documents?  

The reference's IFT branch (invesalius/data/watershed_process.py:44-46,54-57) IS scipy's NI_WatershedIFT, whose bucket
queue has a linked-list defect.  This tool runs the defect-faithful restatement (oracle/ivx_oracle_ws.c,
== live scipy on every test) with per-voxel / per-level event tracing next to the defect-free statement
(oracle/ivx_oracle_wsz.c, == the HIP flood) on bench.py's watershed volume, and measures

  * how many voxels differ (the number bench.py --config watershed reports as `differs_from_reference`),
  * whether the differing voxels are CONFINED to the neighbourhood of the defect's late / lost pops
    (connected components of the difference set that contain or touch such a voxel),
  * which bucket levels see the events, and how many pops a serial replay "of the affected levels only" would have to walk.

Test infrastructure (imports oracle/): never part of the product path.
    python tools/ift_defect_confinement.py [--size 512] [--live-scipy] > profiles/r04_ift_defect_confinement.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--live-scipy", action="store_true", help="also run scipy itself and check the restatement against it")
    args = ap.parse_args()
    from scipy import ndimage

    import bench
    from oracle import oracle as orc

    orc.build()
    n = args.size
    img = bench.synth_v512((n, n, n), seed=bench.SEED)
    mk = bench.ws_markers(img)
    cost = (img - img.min()).astype(np.uint16)
    s6 = ndimage.generate_binary_structure(3, 1)
    t = time.perf_counter()
    defect, ev, flags, lvl = orc.watershed_ift_trace(cost, mk, s6)
    t_defect = time.perf_counter() - t
    t = time.perf_counter()
    clean, ccost = orc.watershed_ift_clean(cost, mk, s6, want_cost=True)
    t_clean = time.perf_counter() - t
    res = {"volume": "bench.py --config watershed --size %d (synth_v512 seed %d, ws_markers, 6 neighbours)" % (n, bench.SEED),
           "voxels": int(img.size), "seconds": {"defect_faithful": round(t_defect, 1), "defect_free": round(t_clean, 1)}}
    if args.live_scipy:
        t = time.perf_counter()
        sci = ndimage.watershed_ift(cost, mk, s6)
        res["seconds"]["live_scipy"] = round(time.perf_counter() - t, 1)
        res["restatement_equals_live_scipy"] = bool(np.array_equal(sci, defect))
    diff = defect != clean
    nd = int(diff.sum())
    res["differs_from_reference"] = nd
    res["events"] = {"requeued_unlinked": ev[0], "popped_late": ev[1], "popped_twice": ev[2], "never_popped": ev[3]}
    late = (flags & 1) != 0
    lost = (flags & 2) != 0
    trig = (flags & 4) != 0
    res["voxels_flagged"] = {"late": int(late.sum()), "never_popped": int(lost.sum()), "trigger": int(trig.sum())}
    # confinement: components of the difference set (6-connected, like the flood) that contain or touch a late / lost voxel
    src = ndimage.binary_dilation(late | lost, structure=s6)
    lab, ncomp = ndimage.label(diff, structure=s6)
    sizes = np.bincount(lab.ravel(), minlength=ncomp + 1)
    touched = np.zeros(ncomp + 1, bool)
    touched[np.unique(lab[src & diff])] = True
    touched[0] = False
    res["difference_components"] = {
        "count": int(ncomp), "largest": int(sizes[1:].max()) if ncomp else 0,
        "touching_a_late_or_lost_voxel": int(touched.sum()),
        "voxels_in_touching_components": int(sizes[touched].sum()),
        "voxels_in_other_components": int(nd - sizes[touched].sum()),
    }
    # the other direction: late / lost voxels whose label still equals the defect-free one
    res["late_or_lost_voxels_with_the_clean_label"] = int(((late | lost) & ~diff).sum())
    # a difference is a voxel whose bucket-queue history changed: how far from the nearest late / lost voxel does it lie?
    if nd:
        dist = ndimage.distance_transform_cdt(~(late | lost), metric="taxicab")
        dd = dist[diff]
        res["distance_of_differing_voxels_to_nearest_late_or_lost_voxel"] = {
            "max": int(dd.max()), "mean": round(float(dd.mean()), 2), "p50": int(np.percentile(dd, 50)), "p99": int(np.percentile(dd, 99))}
    # levels
    pops, latepops, trigs = lvl
    lv_late = np.flatnonzero(latepops)
    lv_trig = np.flatnonzero(trigs)
    used = np.flatnonzero(pops)
    res["levels"] = {
        "non_empty": int(len(used)), "highest": int(used.max()),
        "with_triggers": int(len(lv_trig)), "first_trigger_level": int(lv_trig.min()) if len(lv_trig) else None,
        "with_late_pops": int(len(lv_late)), "first_late_level": int(lv_late.min()) if len(lv_late) else None,
        "pops_total": int(pops.sum()),
        "pops_in_levels_with_late_pops": int(pops[lv_late].sum()),
        "pops_from_first_trigger_level_on": int(pops[lv_trig.min():].sum()) if len(lv_trig) else 0,
        "pops_in_levels_where_a_trigger_happened": int(pops[lv_trig].sum()),
    }
    # cost of the late voxels in the clean statement (the bucket they were spliced OUT of) -- the bucket whose exact stack order a
    # replay needs: everything pushed before the trigger voxel and still queued is what gets deferred
    if late.any():
        c = ccost[late]
        res["clean_cost_of_late_voxels"] = {"min": int(c.min()), "max": int(c.max()), "distinct": int(len(np.unique(c)))}
        res["pops_in_the_buckets_late_voxels_were_spliced_out_of"] = int(pops[np.unique(c)].sum())
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
