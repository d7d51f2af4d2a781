"""GPU parity: MIDA / LMIP / fast contour MIP vs the C oracle (restatement of invesalius_rs/src/mips.rs).
Integer outputs must be bit-exact; float64 contour outputs within 1e-5 relative (f32 pow: see k_rays.hip)."""
import numpy as np
import pytest

from conftest import synth_volume

pytestmark = pytest.mark.gpu


def _out(img, axis, dtype=None):
    shp = tuple(d for i, d in enumerate(img.shape) if i != axis)
    return np.zeros(shp, dtype or img.dtype)


@pytest.mark.parametrize("shape", [(20, 24, 40), (7, 13, 29), (66, 70, 130), (3, 64, 64)])
@pytest.mark.parametrize("axis", [0, 1, 2])
def test_mida_i16(ivxlib, oracle, shape, axis):
    from invesalius3_amd import invesalius_rs as mips
    img = synth_volume(shape, seed=51)
    for wl, ww in ((300, 300), (40, 400), (-600, 1500)):  # slice_.py:898-900 passes window_level twice (quirk Q1)
        g, r = _out(img, axis), _out(img, axis)
        mips.mida(img, axis, wl, ww, g)
        oracle.mida(img, axis, wl, ww, r)
        assert np.array_equal(g, r)


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_mida_u8_and_f64(ivxlib, oracle, axis):
    from invesalius3_amd import invesalius_rs as mips
    img = synth_volume((18, 30, 70), seed=52)
    u = ((img.astype(np.int32) + 1024) // 17).clip(0, 255).astype(np.uint8)
    g, r = _out(u, axis), _out(u, axis)
    mips.mida(u, axis, 60, 80, g)
    oracle.mida(u, axis, 60, 80, r)
    assert np.array_equal(g, r)
    f = (u.astype(np.float64) * 0.9 + 3.0)  # f64 image -> u8 output (mips_py.rs:184-194)
    g, r = _out(f, axis, np.uint8), _out(f, axis, np.uint8)
    mips.mida(f, axis, 60, 80, g)
    oracle.mida(f, axis, 60, 80, r)
    assert np.array_equal(g, r)


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_mida_windows_at_their_edges(ivxlib, oracle, axis):
    """get_opacity's ramp at its corners: windows of width 0 and 1, odd widths (half-integer edges), negative widths, wide windows,
    a window at the top and at the bottom of int16, samples at both ends of the type -- same bits as the oracle's, all three axes."""
    from invesalius3_amd import invesalius_rs as mips
    img = synth_volume((24, 40, 72), seed=53)
    img[::3, ::5, ::7] = 32767
    img[1::3, 1::5, 1::7] = -32768
    for wl, ww in ((0, 0), (100, 1), (101, 7), (5, 20000), (40, -50), (-1000, 8190), (-1000, 8191), (-1000, 8192), (300, 4094), (300, 4095),
                   (300, 4096), (32000, 3), (-32768, 2), (32767, 32767)):
        _same_mida(mips, oracle, img, axis, wl, ww)
    u = ((img.astype(np.int32) + 1024) // 9).clip(0, 255).astype(np.uint8)
    for wl, ww in ((128, 255), (3, 1), (0, 0), (255, 2), (60, 81)):
        _same_mida(mips, oracle, u, axis, wl, ww)


def _same_mida(mips, oracle, img, axis, wl, ww):
    """Same bits -- or, where the reference panics (a width of 0 divides 0 by 0 and NumCast refuses the NaN), the same refusal."""
    g, r = _out(img, axis), _out(img, axis)
    try:
        oracle.mida(img, axis, wl, ww, r)
    except Exception as e:  # noqa: BLE001 (whatever the restatement raises for the reference's panic)
        with pytest.raises(type(e)):
            mips.mida(img, axis, wl, ww, g)
        return
    mips.mida(img, axis, wl, ww, g)
    assert np.array_equal(g, r), (wl, ww)


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_lmip(ivxlib, oracle, axis):
    from invesalius3_amd import invesalius_rs as mips
    img = synth_volume((33, 45, 150), seed=53)
    for tmin, tmax in ((-200, 500), (700, 3033), (5000, 6000)):
        g, r = _out(img, axis), _out(img, axis)
        mips.lmip(img, axis, tmin, tmax, g)
        oracle.lmip(img, axis, tmin, tmax, r)
        assert np.array_equal(g, r)
    sl = img[4:20, 3:40, ::2]  # strided slab
    g, r = _out(sl, axis), _out(sl, axis)
    mips.lmip(sl, axis, -300, 300, g)
    oracle.lmip(np.ascontiguousarray(sl), axis, -300, 300, r)
    assert np.array_equal(g, r)


@pytest.mark.parametrize("tmip", [0, 1, 2])
@pytest.mark.parametrize("axis", [0, 1, 2])
def test_fast_countour_mip_i16(ivxlib, oracle, axis, tmip):
    from invesalius3_amd import invesalius_rs as mips
    img = synth_volume((24, 40, 72), seed=54)
    g, r = _out(img, axis), _out(img, axis)
    mips.fast_countour_mip(img, 1.0, axis, 300, 300, tmip, g)  # border size 1.0: pow exact
    oracle.fast_countour_mip(img, 1.0, axis, 300, 300, tmip, r)
    assert np.array_equal(g, r)
    assert g.any()


def test_fcm_volume_other_exponents(ivxlib, oracle):
    """n != 1: `base ** n` is the platform libm's powf in the reference (Rust's f32::powf, mips.rs:211) and in the oracle (glibc's
    powf, called from C).  The GPU restates glibc's algorithm (csrc/glibc_powf.h: table + polynomial in double, in the build
    this machine's glibc selects), so every exponent is an exact parity case -- including 2.0, where glibc's result is NOT the
    correctly rounded product.  All three projections of the contour volume, and the volume itself through tmip 1 / 2."""
    from invesalius3_amd import invesalius_rs as mips
    img = synth_volume((20, 36, 64), seed=55)
    for n in (0.5, 2.0, 3.3, 0.25, 7.0, 1.5):
        for axis in range(3):
            for tmip in (0, 1, 2):
                g, r = _out(img, axis), _out(img, axis)
                mips.fast_countour_mip(img, n, axis, 300, 300, tmip, g)
                oracle.fast_countour_mip(img, n, axis, 300, 300, tmip, r)
                assert np.array_equal(g, r), (n, axis, tmip)
    # the non-vectorised kernels (rows that are not whole 16-byte chunks; uint8 images)
    odd = np.ascontiguousarray(img[:, :, :61])
    u = ((img.astype(np.int32) + 1024) // 17).clip(0, 255).astype(np.uint8)
    for a in (odd, u):
        for n in (0.7, 2.0):
            g, r = _out(a, 1), _out(a, 1)
            mips.fast_countour_mip(a, n, 1, 100, 100, 0, g)
            oracle.fast_countour_mip(a, n, 1, 100, 100, 0, r)
            assert np.array_equal(g, r)


def test_fcm_u8_f64_and_errors(ivxlib, oracle):
    from invesalius3_amd import invesalius_rs as mips
    img = synth_volume((12, 20, 66), seed=56)
    u = ((img.astype(np.int32) + 1024) // 17).clip(0, 255).astype(np.uint8)
    for tmip in (0, 2):
        g, r = _out(u, 1), _out(u, 1)
        mips.fast_countour_mip(u, 1.0, 1, 100, 100, tmip, g)
        oracle.fast_countour_mip(u, 1.0, 1, 100, 100, tmip, r)
        assert np.array_equal(g, r)
    with pytest.raises(ValueError):  # NumCast::from(700) into u8 panics in the reference
        mips.fast_countour_mip(u, 1.0, 1, 100, 100, 1, _out(u, 1))
    f = img.astype(np.float64) * 0.37
    g, r = _out(f, 2), _out(f, 2)
    mips.fast_countour_mip(f, 1.0, 2, 100.0, 50.0, 0, g)
    oracle.fast_countour_mip(f, 1.0, 2, 100.0, 50.0, 0, r)
    np.testing.assert_allclose(g, r, rtol=1e-6)
    with pytest.raises(TypeError):
        mips.mida(img, 0, 1, 1, np.zeros((20, 66), np.uint8))
    with pytest.raises(OverflowError):
        mips.mida(img, 0, 40000, 1, _out(img, 0))
    # (the wrapped T-subtraction of finite_difference bounds |g| <= 2^15/2 per axis, so the contour intensity
    #  always fits the image dtype: the NumCast panic of mips.rs:241 is unreachable for integer images)


@pytest.mark.parametrize("shape", [(70, 48, 64), (40, 96, 64), (9, 10, 520), (33, 5, 8), (64, 64, 128), (5, 7, 30)])
def test_fused_contour_maxip_equals_the_materialised_contour_volume(ivxlib, oracle, shape):
    """ivx_dev_fcm_maxip (contour value folded into the running maximum, ray split into segments, no temp volume) ==
    ivx_dev_fcm_volume + MaxIP == the C restatement of mips.rs:237-247, every axis, n = 1 / 2 (exact products) and 3.3;
    the last shape has ragged rows and takes the materialised route inside the same entry point"""
    import ctypes

    from invesalius3_amd import _lib as L
    from invesalius3_amd.device import DeviceBuffer, c64
    img = synth_volume(shape, seed=57)
    img[0, 0, :4] = [-32768, 32767, -32768, 32767]  # the T-subtraction of finite_difference wraps (quirk Q3)
    dz, dy, dx = shape
    lib = L.lib()
    d_img, d_tmp, d_st = DeviceBuffer(img.nbytes), DeviceBuffer(img.nbytes), DeviceBuffer(64)
    d_img.upload(img)
    d_st.zero()
    for n in (1.0, 2.0, 3.3):
        for axis in range(3):
            oshp = tuple(d for i, d in enumerate(shape) if i != axis)
            d_a, d_b = DeviceBuffer(int(np.prod(oshp)) * 2 + 64), DeviceBuffer(int(np.prod(oshp)) * 2 + 64)
            L.check(lib.ivx_dev_fcm_maxip(L.I16, d_img.ptr, c64(dz), c64(dy), c64(dx), ctypes.c_float(n), axis, d_a.ptr, d_st.ptr, None))
            L.check(lib.ivx_dev_fcm_volume(L.I16, d_img.ptr, c64(dz), c64(dy), c64(dx), ctypes.c_float(n), axis, d_tmp.ptr, d_st.ptr, None))
            L.check(lib.ivx_dev_mip_reduce(L.I16, d_tmp.ptr, c64(dz), c64(dy), c64(dx), axis, L.MIP_MAX, d_b.ptr, None))
            L.synchronize()
            a, b = d_a.download(oshp, np.int16), d_b.download(oshp, np.int16)
            assert np.array_equal(a, b), (n, axis)
            if True:  # (every exponent is exact since glibc's powf is restated on the device: test_fcm_volume_other_exponents)
                r = np.zeros(oshp, np.int16)
                oracle.fast_countour_mip(img, n, axis, 300, 300, 0, r)
                assert np.array_equal(a, r), (n, axis)
            d_a.close()
            d_b.close()
    assert int(d_st.download((1,), np.int32)[0]) == 0


def test_resident_mida_keeps_the_image_range_until_the_image_changes(ivxlib, oracle):
    """DeviceVolume.mida takes the volume's min / max from DeviceVolume.image_range(): computed once, kept while the image's
    bytes are the pipeline's own, dropped by any outside access to `image` (upload, .ptr) and by forget_image_range()."""
    from invesalius3_amd.device import DeviceBuffer, DeviceVolume
    img = synth_volume((24, 40, 64), seed=58)
    vol = DeviceVolume(img)
    status = DeviceBuffer(64)
    status.zero(vol.stream)
    for round_ in range(2):
        for axis in range(3):
            shp = tuple(d for i, d in enumerate(img.shape) if i != axis)
            out = DeviceBuffer(int(np.prod(shp)) * 2 + 64)
            vol.mida(axis, 300, 900, out, status)
            vol.sync()
            r = np.zeros(shp, np.int16)
            oracle.mida(img, axis, 300, 900, r)
            assert np.array_equal(out.download(shp, np.int16), r), (round_, axis)
            assert vol._range_valid
            out.close()
        if round_ == 0:
            img = (img // 2 + 100).astype(np.int16)   # another range
            vol.image.upload(img)
            assert not vol._range_valid              # an upload from outside drops the note
    mm = vol.image_range().download((2,), np.float32)
    assert mm[0] == img.min() and mm[1] == img.max()
    vol.forget_image_range()
    assert not vol._range_valid
    assert int(status.download((1,), np.int32)[0]) == 0
    vol.close()


def _powf_inputs(seed, n):
    """the contour MIP's own domain (base = 1 - |d / gm|, exponents a user types) + arbitrary bit patterns (NaN, inf,
    subnormals, negative bases with integer / non-integer exponents, overflow and underflow)"""
    rng = np.random.default_rng(seed)
    d = rng.integers(0, 65536, n).astype(np.float32)
    g = d + rng.integers(0, 65536, n).astype(np.float32) + np.float32(1)
    x = np.float32(1) - np.abs(d / g)
    y = (rng.integers(1, 2000, n) / np.float32(64)).astype(np.float32)
    k = n // 4
    x[:k] = rng.integers(0, 2 ** 32, k, dtype=np.uint64).astype(np.uint32).view(np.float32)
    y[:k] = rng.integers(0, 2 ** 32, k, dtype=np.uint64).astype(np.uint32).view(np.float32)
    x[k:2 * k] = -(rng.integers(0, 65536, k) / np.float32(256)).astype(np.float32)
    y[k:2 * k] = rng.integers(-128, 128, k).astype(np.float32)
    return x, y


def test_device_powf_is_the_host_libms_powf_bit_for_bit(ivxlib, oracle):
    """ivx_dev_powf (the function the contour-MIP kernels call) against libm's powf on this machine -- what Rust's f32::powf
    resolves to -- on 4 M inputs, in the build glibc selects here; the other build may differ in a last bit, rarely."""
    import ctypes

    from invesalius3_amd import _lib as L
    from invesalius3_amd.device import DeviceBuffer, c64
    n = 1 << 22
    x, y = _powf_inputs(5, n)
    want = oracle.powf_array(x, y)
    dx_, dy_, do_ = DeviceBuffer(n * 4), DeviceBuffer(n * 4), DeviceBuffer(n * 4)
    dx_.upload(x)
    dy_.upload(y)
    res = {}
    for variant in (-1, 0, 1):
        L.check(L.lib().ivx_dev_powf(dx_.ptr, dy_.ptr, do_.ptr, c64(n), variant, None))
        L.synchronize()
        res[variant] = do_.download((n,), np.float32)
    got = res[-1]
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan])
    assert np.array_equal(got.view(np.uint32), res[L.lib().ivx_powf_variant()].view(np.uint32))
    other = res[1 - L.lib().ivx_powf_variant()]
    assert (other.view(np.uint32)[~nan] != want.view(np.uint32)[~nan]).sum() <= 4  # (~3 in 10^9 differ between the builds)


def test_fast_power_of_the_contour_maxip_stays_inside_its_bound(ivxlib, oracle):
    """The contour MaxIP folds bounds from the transcendental unit (v_log_f32 / v_exp_f32) and takes glibc's powf only for the
    pixels whose bounds leave an integer open (k_fcm_max_decide / k_fcm_fix): the bound must hold -- |fast - libm's powf| <=
    rel * fast (+ a flush-to-zero allowance far below one count) for bases in [0, 1] and exponents in (0, 64], 8 M inputs
    with the edges (0, 1, the float below 1, 2^-24, exponents 2, 0.5, 64)."""
    from invesalius3_amd import _lib as L
    from invesalius3_amd.device import DeviceBuffer, c64
    n = 1 << 23
    rng = np.random.default_rng(17)
    x = rng.random(n, dtype=np.float32)
    x[: n // 4] = (1.0 - rng.random(n // 4) ** 4).astype(np.float32)          # crowded towards 1 (flat regions of the ray)
    x[n // 4: n // 2] = (rng.random(n // 4) ** 6).astype(np.float32)           # and towards 0
    y = rng.uniform(0.01, 64.0, n).astype(np.float32)
    y[::3] = rng.choice(np.array([0.5, 1.5, 2.0, 3.0, 3.3, 8.0, 64.0], np.float32), len(y[::3]))
    x[(x > 0) & (x < 2.0 ** -24)] = 2.0 ** -24  # (1 - |d / gm| is 0 or at least one ulp of 1: the kernel never sees less)
    x[:8] = [0.0, 1.0, np.nextafter(np.float32(1), np.float32(0)), 2.0 ** -24, 0.5, 2.0 ** -20, 0.999, 0.25]
    want = oracle.powf_array(x, y).astype(np.float64)
    dx_, dy_, do_ = DeviceBuffer(n * 4), DeviceBuffer(n * 4), DeviceBuffer(n * 4)
    dx_.upload(x)
    dy_.upload(y)
    out = {}
    for variant in (2, 3):
        L.check(L.lib().ivx_dev_powf(dx_.ptr, dy_.ptr, do_.ptr, c64(n), variant, None))
        L.synchronize()
        out[variant] = do_.download((n,), np.float32).astype(np.float64)
    fast, rel = out[2], out[3]
    err = np.abs(fast - want)
    assert np.isfinite(fast).all() and np.isfinite(rel).all()
    assert (err <= rel * fast + 1e-37).all(), (int((err > rel * fast + 1e-37).sum()), float((err / np.maximum(fast, 1e-300)).max()))
    # and the margin the bound keeps (a factor of four was the design): the worst observed error against the bound
    ok = (fast > 1e-30) & (rel > 0)
    used = err[ok] / (rel[ok] * fast[ok])
    print("fast power: worst error / bound = %.3f" % float(used.max()))
    assert used.max() < 0.6


@pytest.mark.parametrize("n", [1.0, 2.0, 1.5, 3.3, 8.0, 64.0])
def test_contour_bounds_from_the_unit_hold_the_reference_value(ivxlib, oracle, n):
    """Exponents >= 1 fold bounds built on the wrapped int16 differences D: S = D0^2 + D1^2 + D2^2 exact in 32 bits, then
    v_sqrt_f32(S) - |Dray| (exponent 1, the GUI's default) or v_rsq_f32(S) in front of the fast power (fcm_pow_fold), and take the
    reference's float sequence (correctly rounded root and quotient, glibc's powf) only for the pixels the bounds leave open.
    The bounds must hold that sequence's value (numpy's float32 sqrt and division are correctly rounded; the power is the host
    libm's) for every difference an int16 volume can produce: 4 M draws with the edges (one axis only, equal axes, the largest)."""
    import ctypes
    from invesalius3_amd import _lib as L
    from invesalius3_amd.device import DeviceBuffer, c64
    cnt = 1 << 22
    rng = np.random.default_rng(23)
    scale = 2.0 ** rng.uniform(0, 15, (3, cnt))
    D = np.clip(np.round(rng.uniform(-1, 1, (3, cnt)) * scale), -32768, 32767).astype(np.int32)
    D[:, :6] = np.array([[1, 0, 0], [32767, 32767, 32767], [-32768, -32768, -32768], [6, 8, 0], [0, 0, 15], [1, 32767, 0]], np.int32).T
    D[1:, 6:cnt // 16] = 0            # the ray along the only gradient: base 0
    D[0, cnt // 16:cnt // 8] = 0      # ... and across it: base 1
    D[:, cnt // 8:cnt // 4] //= 256   # small gradients (flat tissue)
    g = (D.astype(np.float32) / np.float32(2))
    d = g[0]
    s = ((d * d + g[1] * g[1]).astype(np.float32) + g[2] * g[2]).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        gm = np.sqrt(s)
        base = np.where(s != 0, np.float32(1) - np.abs(d / gm), np.float32(0)).astype(np.float32)
    power = base if n == 1.0 else oracle.powf_array(base, np.full(cnt, n, np.float32))
    want = np.where(s != 0, gm * power, np.float32(0)).astype(np.float64)
    other = ((D[1].astype(np.uint32) & 0xFFFF) | ((D[2].astype(np.uint32) & 0xFFFF) << 16)).astype(np.uint32)
    dd, do, dlo, dhi = DeviceBuffer(cnt * 4), DeviceBuffer(cnt * 4), DeviceBuffer(cnt * 4), DeviceBuffer(cnt * 4)
    dd.upload(np.ascontiguousarray(D[0]))
    do.upload(other)
    L.check(L.lib().ivx_dev_fcm_bounds(dd.ptr, do.ptr, ctypes.c_float(n), dlo.ptr, dhi.ptr, c64(cnt), None))
    L.synchronize()
    lo, hi = dlo.download((cnt,), np.float32).astype(np.float64), dhi.download((cnt,), np.float32).astype(np.float64)
    assert np.isfinite(lo).all() and np.isfinite(hi).all()
    bad = (want < lo) | (want > hi)
    assert not bad.any(), (int(bad.sum()), D[:, bad][:, :4], want[bad][:4], lo[bad][:4], hi[bad][:4])
    half = (hi - lo) / 2
    ok = half > 0
    used = np.abs(want - (hi + lo) / 2)[ok] / half[ok]
    print("contour bounds, exponent %g: worst error / bound = %.3f, median half-width %.2e" % (n, float(used.max()), float(np.median(half[ok]))))
    assert used.max() < 0.75  # (the analysis in k_rays.hip leaves a quarter in hand)
