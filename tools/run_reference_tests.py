"""Run the REFERENCE's own test files for this path (from /root/reference/tests, unmodified) in the build container, with
  * stand-in modules for the GUI-side imports (wx, VTK, pubsub ...),
  * `invesalius_rs._native` (the Rust extension, not buildable here) bound to oracle/'s C restatement,
  * `skimage.segmentation.watershed` bound to a proxy that runs the real scikit-image 0.18.3 under /opt/conda.
What passes shows that the restatement under the reference's own Python satisfies the reference's own assertions.

    python3 tools/run_reference_tests.py [pytest args ...]        # default: test_segmentation_tools.py test_bone_thresholding.py
    python3 tools/run_reference_tests.py --record tests/golden/ref_suite_calls.npz
        also writes down every call the reference's tests make across the boundary this repository replaces -- the native
        functions of invesalius_rs, skimage.segmentation.watershed and scipy.ndimage.watershed_ift as watershed_process.py
        calls them -- with the arguments before the call, the arrays after it and the value returned.  The file is DATA
        (inputs and the outputs the reference's own assertions accepted: the suite passed while they were recorded);
        tests/test_gpu_reference_calls.py replays every call through the product's functions of the same names on the GPU
        and asks for the same bits.  (The reference's sources cannot travel to the GPU box; its calls can.)
"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import make_golden_ref_dowatershed as M  # noqa: E402
from oracle import oracle as O  # noqa: E402


CALLS = []


def _plain(a):
    import numpy as np
    if isinstance(a, np.ndarray):
        return np.array(a, copy=True)
    if isinstance(a, (np.integer,)):
        return int(a)
    if isinstance(a, (np.floating,)):
        return float(a)
    if isinstance(a, (np.bool_,)):
        return bool(a)
    if isinstance(a, (list, tuple)):
        return [_plain(x) for x in a]
    return a


def recorded(name, fn):
    def w(*args):
        before = [_plain(a) for a in args]
        r = fn(*args)
        CALLS.append((name, before, [_plain(a) for a in args], _plain(r)))
        return r
    return w


def save_calls(path):
    import json

    import numpy as np
    arrays, manifest = {}, []

    def put(v, key):
        if isinstance(v, np.ndarray):
            arrays[key] = v
            return {"a": key}
        return {"v": v}

    for i, (name, before, after, ret) in enumerate(CALLS):
        manifest.append({"name": name,
                         "args": [put(v, "c%d_in%d" % (i, k)) for k, v in enumerate(before)],
                         "after": [put(v, "c%d_out%d" % (i, k)) if isinstance(v, np.ndarray) else None for k, v in enumerate(after)],
                         "ret": put(ret, "c%d_ret" % i)})
    np.savez_compressed(path, manifest=np.array(json.dumps(manifest)), **arrays)
    print("recorded %d calls (%s) -> %s" % (len(CALLS), ", ".join(sorted({c[0] for c in CALLS})), path))


def main():
    record = None
    if "--record" in sys.argv:
        k = sys.argv.index("--record")
        record = os.path.abspath(sys.argv[k + 1])
        del sys.argv[k:k + 2]
    tmp_root = os.path.join(ROOT, "gpurun_out", "ref_tmp")
    os.makedirs(tmp_root, exist_ok=True)
    tempfile.tempdir = tmp_root
    os.environ["HOME"] = tmp_root
    M._Finder.ROOTS = tuple(r for r in M._Finder.ROOTS if r != "invesalius_rs")
    native = M._Fake("invesalius_rs._native")
    native.floodfill = lambda data, i, j, k, v, fill, out: O.floodfill(data, i, j, k, v, fill, out)
    native.floodfill_threshold = lambda data, seeds, t0, t1, fill, strct, out: O.floodfill_threshold(data, seeds, t0, t1, fill, strct, out)
    native.floodfill_threshold_inplace = lambda data, seeds, t0, t1, fill, strct: O.floodfill_threshold_inplace(data, seeds, t0, t1, fill, strct)
    native.floodfill_auto_threshold = lambda data, seeds, p, fill, out: O.floodfill_auto_threshold(data, seeds, p, fill, out)
    native.fill_holes_automatically = lambda mask, labels, nlabels, size: O.fill_holes_automatically(mask, labels, nlabels, size)
    if record:
        for nm in ("floodfill", "floodfill_threshold", "floodfill_threshold_inplace", "floodfill_auto_threshold", "fill_holes_automatically"):
            setattr(native, nm, recorded(nm, getattr(native, nm)))
        import scipy.ndimage
        scipy.ndimage.watershed_ift = recorded("watershed_ift", scipy.ndimage.watershed_ift)
    sys.modules["invesalius_rs._native"] = native
    import types
    seg = types.ModuleType("skimage.segmentation")  # (the rest of skimage stays a stand-in)
    seg.watershed = recorded("watershed", M.skimage_watershed_proxy) if record else M.skimage_watershed_proxy
    sys.modules["skimage.segmentation"] = seg
    sys.meta_path.insert(0, M._Finder())
    import pubsub.pub
    pubsub.pub.subscribe = lambda *a, **k: (None, True)
    pubsub.pub.sendMessage = lambda *a, **k: None
    sys.path.insert(0, "/root/reference")
    import pytest
    from unittest import mock

    class MockerPlugin:  # (pytest-mock is not installed: the two calls the reference's tests make of it)
        @pytest.fixture
        def mocker(self):
            started = []

            class Mocker:
                def patch(self, target, *a, **k):
                    p = mock.patch(target, *a, **k)
                    started.append(p)
                    return p.start()

            yield Mocker()
            for p in started:
                p.stop()

    args = sys.argv[1:] or ["/root/reference/tests/test_segmentation_tools.py", "/root/reference/tests/test_bone_thresholding.py",
                            "/root/reference/tests/test_mask.py"]
    os.chdir(tmp_root)
    rc = pytest.main(["-q", "-p", "no:cacheprovider", "--rootdir", tmp_root] + args, plugins=[MockerPlugin()])
    if record and rc == 0:
        save_calls(record)
    return rc


if __name__ == "__main__":
    sys.exit(main())
