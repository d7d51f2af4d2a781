"""The reference's OWN test files for this path, unmodified, run from its checkout when that exists (the build container):
tools/run_reference_tests.py imports the reference with stand-in modules for wx / VTK / pubsub, binds the Rust extension
(`invesalius_rs._native`, not buildable here) to oracle/'s C restatement and `skimage.segmentation.watershed` to the real
scikit-image under /opt/conda.  All of test_segmentation_tools.py (region growing, fill holes, statistics, do_watershed,
brush edits), test_bone_thresholding.py and test_mask.py pass: the restatement satisfies the reference's own assertions under the
reference's own Python."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/tests"


@pytest.mark.skipif(not os.path.isdir(REF) or not os.path.exists("/opt/conda/bin/python3.9"),
                    reason="needs the reference checkout and the scikit-image interpreter of the build container")
def test_reference_test_files_pass_over_the_restated_native_code():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_reference_tests.py"),
                        os.path.join(REF, "test_segmentation_tools.py"), os.path.join(REF, "test_bone_thresholding.py"),
                        os.path.join(REF, "test_mask.py")],
                       capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert "23 passed" in r.stdout, tail
