"""Run the REFERENCE's own test files for this path (from /root/reference/tests, unmodified) in the build container, with
  * stand-in modules for the GUI-side imports (wx, VTK, pubsub ...),
  * `invesalius_rs._native` (the Rust extension, not buildable here) bound to oracle/'s C restatement,
  * `skimage.segmentation.watershed` bound to a proxy that runs the real scikit-image 0.18.3 under /opt/conda.
What passes shows that the restatement under the reference's own Python satisfies the reference's own assertions.

    python3 tools/run_reference_tests.py [pytest args ...]        # default: test_segmentation_tools.py test_bone_thresholding.py
"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import make_golden_ref_dowatershed as M  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
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
    sys.modules["invesalius_rs._native"] = native
    import types
    seg = types.ModuleType("skimage.segmentation")  # (the rest of skimage stays a stand-in)
    seg.watershed = M.skimage_watershed_proxy
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
    return pytest.main(["-q", "-p", "no:cacheprovider", "--rootdir", tmp_root] + args, plugins=[MockerPlugin()])


if __name__ == "__main__":
    sys.exit(main())
