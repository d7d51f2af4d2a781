#!/usr/bin/env python3
"""ws{512,1024}_full.npz: the watershed floods of bench.py's configs[2] volumes, WHOLE volume, from the serial CPU side --
so that bench.py and the GPU tests can gate the flood they time at its stated size (1024^3: ten minutes of serial scipy and
43 GB of its queue elements per run otherwise).

Volume: bench.synth_v512((n, n, n), seed 20260924), markers bench.ws_markers, 6 neighbours.
  IFT branch (watershed_process.py:54-57): cost = (image - image.min()).astype(uint16), int8 markers.
    scipy_crc32      CRC-32 of live scipy.ndimage.watershed_ift's labels as uint8 -- THE REFERENCE's bits
    clean_crc32      CRC-32 of the defect-free statement's labels (oracle/ivx_oracle_wsz.c) -- what the HIP flood equals
    differs_at       sorted linear indices where the two differ (delta-encoded: cumsum restores them).  Labels are 1 / 2
                     everywhere, so scipy's volume IS the defect-free one with `3 - label` at these places: a holder of the
                     defect-free labels can rebuild the reference's and check scipy_crc32.
  "Watershed" branch with the GUI's defaults (watershed_process.py:33-39; bench.py: ww 400, wl 300, 3x3x3 gradient):
    sk_heap_crc32    CRC-32 of the serial (value, age) heap flood (oracle/ivx_oracle_wssk.c, tie_mode 0: pinned move for move to
                     scikit-image 0.18.3's compiled kernel by tests/golden/watershed_sk.npz)
    sk_raster_crc32  the same with equal-valued markers in raster order (tie_mode 1: the statement the HIP flood implements)
    sk_differs       voxels where the two differ
  image_crc32        CRC-32 of the int16 volume: a run whose synthetic volume differs (another numpy's float32 exp / sin) must
                     not quote this file.

    python tests/golden/make_golden_ws_full.py --size 512      # ~4 min;  1024: ~25 min and ~50 GB of RAM
(this container: numpy 2.2.6, scipy 1.15.3; the cost image of the "Watershed" branch needs the HIP library only for the
GPU, so it is restated here with numpy / scipy exactly as tests/test_gpu_wssk.py does)
"""
import argparse
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    args = ap.parse_args()
    import scipy
    from scipy import ndimage

    import bench
    from oracle import oracle as orc

    orc.build()
    n = args.size
    t0 = time.time()
    img = bench.synth_v512((n, n, n), seed=bench.SEED)
    mk = bench.ws_markers(img)
    s6 = ndimage.generate_binary_structure(3, 1)
    rec = {"image_crc32": np.uint32(zlib.crc32(img)), "shape": np.array(img.shape),
           "versions": np.array(["numpy " + np.__version__, "scipy " + scipy.__version__])}
    print("volume %.0fs" % (time.time() - t0), flush=True)
    cost = (img - img.min()).astype(np.uint16)
    sci = ndimage.watershed_ift(cost, mk, s6).astype(np.uint8)
    print("scipy %.0fs" % (time.time() - t0), flush=True)
    rec["scipy_crc32"] = np.uint32(zlib.crc32(sci))
    clean = orc.watershed_ift_clean(cost, mk, s6).astype(np.uint8)
    print("clean %.0fs" % (time.time() - t0), flush=True)
    del cost
    assert set(np.unique(clean[::7, ::7, ::7]).tolist()) <= {1, 2}
    rec["clean_crc32"] = np.uint32(zlib.crc32(clean))
    at = np.flatnonzero((sci != clean).ravel()).astype(np.int64)
    assert ((sci.ravel()[at] + clean.ravel()[at]) == 3).all()  # 1 <-> 2 only
    rec["differs_at"] = np.diff(at, prepend=0).astype(np.uint32)
    rec["differs"] = np.int64(len(at))
    del sci, clean
    # the GUI's default branch: window/level LUT -> 3x3x3 morphological gradient -> heap flood
    lut = orc.get_LUT_value(img, 400, 300).astype(np.uint16)
    grad = ndimage.morphological_gradient(lut, size=(3, 3, 3))
    del lut
    mk16 = mk.astype(np.int16)
    heap = orc.watershed_sk(grad, mk16, s6, 0).astype(np.uint8)
    print("heap %.0fs" % (time.time() - t0), flush=True)
    rec["sk_heap_crc32"] = np.uint32(zlib.crc32(heap))
    raster = orc.watershed_sk(grad, mk16, s6, 1).astype(np.uint8)
    rec["sk_raster_crc32"] = np.uint32(zlib.crc32(raster))
    rec["sk_differs"] = np.int64((heap != raster).sum())
    rec["grad_crc32"] = np.uint32(zlib.crc32(grad))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ws%d_full.npz" % n)
    np.savez_compressed(out, **rec)
    print(out, os.path.getsize(out), "bytes; IFT: scipy != defect-free statement in", int(rec["differs"]),
          "voxels; heap vs raster marker ties:", int(rec["sk_differs"]), "; %.0fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
