"""GPU parity: MaxIP / MinIP / MeanIP vs numpy (bit-exact, mean in float64)."""
import numpy as np
import pytest

from conftest import synth_volume

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(20, 24, 40), (7, 13, 29), (64, 64, 64), (1, 5, 9), (3, 300, 8)])
@pytest.mark.parametrize("axis", [0, 1, 2])
def test_projections_match_numpy(ivxlib, shape, axis):
    from invesalius3_amd import slice_
    a = synth_volume(shape, seed=31)
    for proj, f in ((slice_.PROJECTION_MaxIP, np.max), (slice_.PROJECTION_MinIP, np.min),
                    (slice_.PROJECTION_MeanIP, np.mean)):
        got = slice_.project(a, axis, proj)
        exp = f(np.array(a), axis=axis)
        assert got.dtype == exp.dtype
        assert np.array_equal(got, exp)


def test_strided_slab_and_u8(ivxlib):
    from invesalius3_amd import slice_
    a = synth_volume((30, 40, 50), seed=32)
    slab = a[5:15]  # matrix[n:n+number_slices], slice_.py:863
    assert np.array_equal(slice_.project(slab, 0, slice_.PROJECTION_MaxIP), slab.max(0))
    cor = a[:, 3:9, :]
    assert np.array_equal(slice_.project(cor, 1, slice_.PROJECTION_MinIP), cor.min(1))
    u = (a & 0xFF).astype(np.uint8)
    assert np.array_equal(slice_.project(u, 2, slice_.PROJECTION_MeanIP), u.mean(2))


def test_full_size_512_three_axis_sweep(ivxlib):
    from invesalius3_amd import slice_
    a = np.random.default_rng(9).integers(-1024, 3072, (512, 512, 512), dtype=np.int16)
    for axis in range(3):
        assert np.array_equal(slice_.project(a, axis, slice_.PROJECTION_MaxIP), a.max(axis))
    assert np.array_equal(slice_.project(a, 0, slice_.PROJECTION_MeanIP), a.mean(0))
