#!/usr/bin/env python3
"""PCIe-inclusive rates of the HOST-level entry points (numpy in, numpy out, strided (d+1,h+1,w+1) mask): what a
drop-in caller that does not keep the volume resident pays.  512^3, wall-clock around the Python call."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, "/root/repo")
sys.path.insert(0, ".")
from scipy.ndimage import generate_binary_structure  # noqa: E402

from bench import BONE, synth_v512  # noqa: E402
from invesalius3_amd import invesalius_rs as rs, slice_, surface_process as sp  # noqa: E402


def timeit(fn, reps=5, before=None):
    """wall times of fn() over `reps` calls (after one warm-up): {"s": the MEDIAN, "min_s", "max_s", "calls"}; `before()` runs
    untimed ahead of every call.  (Rounds 1 - 5 kept the minimum of three: README / DESIGN then quoted best cases the tracked
    file did not reproduce -- VERDICT r5 weak #7.)"""
    t = []
    for i in range(reps + 1):
        if before is not None:
            before()
        t0 = time.perf_counter()
        fn()
        if i:
            t.append(time.perf_counter() - t0)
    return {"s": float(np.median(t)), "min_s": min(t), "max_s": max(t), "calls": len(t)}


def main():
    n = 512
    img = synth_v512((n, n, n))
    nvox = img.size
    mask = np.zeros((n + 1,) * 3, np.uint8)
    res = {}

    def thr_reset():
        mask[1:, 0, 0] = 0

    res["do_threshold_to_all_slices (strided mask, preserve rule)"] = timeit(
        lambda: slice_.do_threshold_to_all_slices(mask, img, BONE), before=thr_reset)
    z, y, x = np.unravel_index(int(np.argmax(img)), img.shape)
    out = np.zeros(img.shape, np.uint8)
    s26 = generate_binary_structure(3, 3)

    def out_reset():
        out[:] = 0

    res["floodfill_threshold (26-conn, dense out)"] = timeit(
        lambda: rs.floodfill_threshold(img, [(int(x), int(y), int(z))], BONE[0], BONE[1], 1, s26, out), before=out_reset)
    view = mask[1:, 1:, 1:]

    def view_reset():
        view[view == 254] = 255

    res["floodfill_threshold_inplace (mask view [1:,1:,1:])"] = timeit(
        lambda: rs.floodfill_threshold_inplace(view, [(int(x), int(y), int(z))], 253, 255, 254, s26), reps=5, before=view_reset)
    tri = [None]

    def mc():
        tri[0] = sp.surface_piece(None, mask, slice(0, n), (1.0, 1.0, 1.0), 0, 0, True)

    def drop():
        # the previous call's 228 MB soup released OUTSIDE the timed call (its munmap alone costs ~9 ms)
        tri[0] = None
    # both ways, side by side: rounds 1 - 4 timed the call WITH the previous result's release inside it, round 5 without
    res["create_surface_piece (whole volume, from_binary)"] = timeit(mc, before=drop)
    res["create_surface_piece (whole volume, from_binary), previous soup released inside the timed call"] = timeit(mc)
    o2 = np.zeros((n, n), np.int16)
    res["mida axis 0"] = timeit(lambda: rs.mida(img, 0, 300, 600, o2))
    res["project MaxIP axis 0"] = timeit(lambda: slice_.project(img, 0, slice_.PROJECTION_MaxIP))
    # do_watershed as the reference calls it (memmap out, queue), the GUI's default settings and the IFT alternative
    import os
    import tempfile

    from invesalius3_amd import watershed_process as wp
    from tools.bench_wsift import markers_for
    mk = markers_for(img).astype(np.int16)
    fd, tfile = tempfile.mkstemp(suffix=".dat")
    os.close(fd)
    np.memmap(tfile, shape=img.shape, dtype="uint8", mode="w+").flush()
    s6 = generate_binary_structure(3, 1)
    import warnings
    warnings.simplefilter("ignore")
    flush = {}

    def dows(alg):
        # wall time of the whole hook and, inside it, of the reference's own `mask.flush()` (msync of 134 MB to the file system:
        # 16 - 49 ms on this box's overlay file system, the spread of the call) -- medians over five calls, msync apart
        ts, fl = [], []
        wp.do_watershed(img, mk, tfile, img.shape, s6, alg, (3, 3, 3), True, 300, 400, None)
        for _ in range(5):
            t0 = time.perf_counter()
            wp.do_watershed(img, mk, tfile, img.shape, s6, alg, (3, 3, 3), True, 300, 400, None)
            ts.append(time.perf_counter() - t0)
            fl.append(wp.do_watershed.last_flush_ms / 1e3)
        net = [a - b for a, b in zip(ts, fl)]
        return {"s": float(np.median(ts)), "min_s": min(ts), "max_s": max(ts), "calls": len(ts), "of_which_msync_s": float(np.median(fl)),
                "net_of_msync_s": float(np.median(net))}
    for alg, name in (("Watershed", "do_watershed (Watershed, ww/wl, 6 neighbours)"), ("Watershed IFT", "do_watershed (Watershed IFT, ww/wl, 6 neighbours)")):
        res[name] = dows(alg)
    # ... and the download into the memmap through the page-locked lanes (every chunk's page faults on its own thread) or as one
    # hipMemcpy (IVX_D2H_LANES is read per call; unset = lanes where the destination's pages are mostly not resident)
    for mode in ("1", "0"):
        os.environ["IVX_D2H_LANES"] = mode
        res["do_watershed (Watershed IFT, ww/wl) IVX_D2H_LANES=%s" % mode] = dows("Watershed IFT")
    os.environ.pop("IVX_D2H_LANES")
    os.remove(tfile)
    if tri[0] is None:
        mc()
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for d in ("invesalius3_amd/csrc", "invesalius3_amd"):
        for f in sorted(os.listdir(os.path.join(root, d))):
            if f.endswith((".hip", ".h", ".py")):
                h.update(open(os.path.join(root, d, f), "rb").read())
    print(json.dumps({"size": "512^3", "triangles": int(len(tri[0])), "protocol": "median of 5 calls after one warm-up (min / max beside it)",
                      "sources_sha16": h.hexdigest()[:16],
                      "results": {k: dict({kk: round(vv, 4) if isinstance(vv, float) else vv for kk, vv in v.items()},
                                          **{"Mvoxel/s": round(nvox / v["s"] / 1e6, 1)}) for k, v in res.items()}}, indent=1))


if __name__ == "__main__":
    main()
