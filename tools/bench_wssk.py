"""The scikit-image branch of do_watershed on the GPU (gradient image + csrc/k_wssk.hip) at BASELINE configs[2]-like sizes:
stage times (HIP events inside the library); with --oracle the serial heap flood of oracle/ on the same input (1 Mvoxel/s:
keep the size small) and the voxel counts that differ from it with raster / with scikit-image's heap marker ties.
python tools/bench_wssk.py 256 512 [conn=1] [mode=lut|minshift] [--oracle]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from bench import synth_v512  # noqa: E402
from invesalius3_amd import _lib as L, watershed_process as wp  # noqa: E402
from scipy.ndimage import generate_binary_structure  # noqa: E402
from tools.bench_wsift import markers_for  # noqa: E402


def main():
    L.require_device()
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [256]
    conns = [int(a[5:]) for a in sys.argv[1:] if a.startswith("conn=")] or [1, 3]
    modes = [a[5:] for a in sys.argv[1:] if a.startswith("mode=")] or ["lut", "minshift"]
    for n in sizes:
        img = synth_v512((n, n, n))
        mk = markers_for(img).astype(np.int16)
        for mode in modes:
            t = time.perf_counter()
            grad = wp.cost_image(img, mode == "lut", 300, 400, (3, 3, 3))
            t_grad = time.perf_counter() - t
            for conn in conns:
                s = generate_binary_structure(3, conn)
                wp.watershed(grad[:16], mk[:16], s)  # warm up
                t = time.perf_counter()
                got, st = wp.watershed(grad, mk, s, want_stats=True)
                wall = time.perf_counter() - t
                rec = dict(n=n, mode=mode, conn=conn, wall_s=round(wall, 4), grad_host_s=round(t_grad, 4), label1=int((got == 1).sum()),
                           image_levels=int(len(np.unique(grad))), **st)
                if "--oracle" in sys.argv:
                    from oracle import oracle as O
                    t = time.perf_counter()
                    o1 = O.watershed_sk(grad, mk, s, 1)
                    rec["oracle_s"] = round(time.perf_counter() - t, 2)
                    rec["mismatch_vs_serial_raster_ties"] = int((o1 != got).sum())
                    rec["mismatch_vs_serial_heap_ties"] = int((O.watershed_sk(grad, mk, s, 0) != got).sum())
                print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
