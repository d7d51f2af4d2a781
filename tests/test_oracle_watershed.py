"""Pin the C restatement of scipy.ndimage.watershed_ift (oracle/ivx_oracle_ws.c) against the live scipy function --
the same third-party call the reference makes (invesalius/data/watershed_process.py:41-46,54-57).  CPU only."""
import numpy as np
import pytest
from scipy import ndimage
from scipy.ndimage import generate_binary_structure


def test_reference_fixture_counts(oracle):
    """tests/test_segmentation_tools.py:170-213 fixture through the IFT branch: 27 voxels label 1, 98 label 2."""
    image = np.zeros((5, 5, 5), dtype=np.int16)
    image[1:4, 1:4, 1:4] = 100
    markers = np.zeros((5, 5, 5), dtype=np.int16)
    markers[2, 2, 2] = 1
    markers[0, 0, 0] = 2
    cost = (image - image.min()).astype("uint16")
    s = generate_binary_structure(3, 1)
    exp = ndimage.watershed_ift(cost, markers, s)
    got = oracle.watershed_ift(cost, markers, s)
    assert np.array_equal(got, exp)
    assert (got == 1).sum() == 27 and (got == 2).sum() == 98


def test_black_box_facts(oracle):
    """max-arc path cost, LIFO inside a cost bucket, negative markers after positive ones."""
    line = np.array([[0, 0, 0], [1, 1, 1], [0, 0, 0]], np.uint8)  # 1-D connectivity inside a (1, n) image
    for img, mk in (([0, 6, 5, 8], [1, 0, 0, 2]), ([0, 0, 0, 0, 0], [1, 0, 0, 0, 2]), ([0, 0, 0, 0, 0], [2, 0, 0, 0, 1]),
                    ([3, 3, 9, 1, 1, 1], [-1, 0, 0, 0, 0, 2])):
        a = np.array(img, np.uint16).reshape(1, -1)
        m = np.array(mk, np.int16).reshape(1, -1)
        assert np.array_equal(oracle.watershed_ift(a, m, line), ndimage.watershed_ift(a, m, line))
    got = oracle.watershed_ift(np.array([[0, 6, 5, 8]], np.uint16), np.array([[1, 0, 0, 2]], np.int16), line)
    assert list(got.ravel()) == [1, 2, 2, 2]       # arc weights |dI|, path cost = max arc
    flat = oracle.watershed_ift(np.zeros((1, 5), np.uint16), np.array([[1, 0, 0, 0, 2]], np.int16), line)
    assert list(flat.ravel()) == [1, 2, 2, 2, 2]   # LIFO: the later marker floods the plateau


@pytest.mark.parametrize("conn", [1, 2, 3])
@pytest.mark.parametrize("dtype,mdtype", [(np.uint16, np.int16), (np.uint8, np.int8)])
def test_random_volumes_match_scipy(oracle, conn, dtype, mdtype):
    rng = np.random.default_rng(100 + conn)
    for shape in ((6, 9, 11), (3, 17, 5), (1, 12, 12)):
        hi = 40 if dtype == np.uint8 else 3000
        img = rng.integers(0, hi, shape).astype(dtype)
        img[rng.random(shape) < 0.3] = 0  # plateaus: ties are the common case on LUT-windowed images
        mk = np.zeros(shape, mdtype)
        idx = rng.integers(0, img.size, 6)
        mk.ravel()[idx] = rng.choice(np.array([1, 2, -1], mdtype), 6)
        s = generate_binary_structure(3, conn)
        assert np.array_equal(oracle.watershed_ift(img, mk, s), ndimage.watershed_ift(img, mk, s))


def test_2d_variant(oracle):
    """styles.py:1926-2000 runs the same four branches on one slice"""
    rng = np.random.default_rng(7)
    img = rng.integers(0, 500, (20, 30)).astype(np.uint16)
    mk = np.zeros((20, 30), np.int16)
    mk[3, 4] = 1
    mk[15, 22] = 2
    s = generate_binary_structure(2, 2)
    assert np.array_equal(oracle.watershed_ift(img, mk, s), ndimage.watershed_ift(img, mk, s))
