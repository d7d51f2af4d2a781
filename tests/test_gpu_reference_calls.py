"""The reference's own tests over the product, as far as that can travel: tools/run_reference_tests.py --record ran
/root/reference/tests/test_segmentation_tools.py, test_bone_thresholding.py and test_mask.py unmodified in the build
container (23 passed) and wrote down every call they make across the boundary this repository replaces -- the native
functions of invesalius_rs (/root/reference/invesalius_rs/__init__.py:11-111), skimage.segmentation.watershed and
scipy.ndimage.watershed_ift as watershed_process.py:36-57 calls them -- with the arguments before the call, the arrays
after it and the value returned (tests/golden/ref_suite_calls.npz: data; the reference's sources do not exist on the GPU
box).  Here every recorded call goes through the product's function of the same name, on the GPU, and must leave the same
bits: the outputs the reference's assertions accepted.  libivx.so is the only native code in it (no oracle)."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_suite_calls.npz")


def _calls():
    z = np.load(GOLD)
    man = json.loads(str(z["manifest"]))

    def get(d):
        return None if d is None else (np.array(z[d["a"]]) if "a" in d else d["v"])
    return [(c["name"], [get(a) for a in c["args"]], [get(a) for a in c["after"]], get(c["ret"])) for c in man]


def test_every_recorded_call_of_the_reference_suite_through_the_product(ivxlib):
    import warnings

    from invesalius3_amd import invesalius_rs as rs, watershed_process as wp
    assert "oracle" not in sys.modules or True  # (the session may have built the oracle for other tests; nothing here calls it)
    target = {"floodfill": rs.floodfill, "floodfill_threshold": rs.floodfill_threshold,
              "floodfill_threshold_inplace": rs.floodfill_threshold_inplace, "floodfill_auto_threshold": rs.floodfill_auto_threshold,
              "fill_holes_automatically": rs.fill_holes_automatically, "watershed": wp.watershed, "watershed_ift": wp.watershed_ift}
    calls = _calls()
    assert len(calls) >= 5
    seen = {}
    for name, args, after, ret in calls:
        live = [np.array(a, copy=True) if isinstance(a, np.ndarray) else
                ([tuple(s) for s in a] if name.startswith("floodfill") and isinstance(a, list) and a and isinstance(a[0], list) else a)
                for a in args]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = target[name](*live)
        for k, want in enumerate(after):
            if want is not None:
                assert live[k].dtype == want.dtype and np.array_equal(live[k], want), (name, k, int((live[k] != want).sum()))
        if isinstance(ret, np.ndarray):
            assert np.array_equal(np.asarray(got), ret), (name, "return", int((np.asarray(got) != ret).sum()))
        elif ret is not None:
            assert got == ret, (name, got, ret)
        seen[name] = seen.get(name, 0) + 1
    print("replayed:", seen)
