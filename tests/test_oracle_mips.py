"""The projections of invesalius_rs/src/mips.rs have no tests upstream and no Rust toolchain exists here, so the C restatement
in oracle/ivx_oracle.c is what the HIP kernels are compared with ("parity unpinned upstream").  This file holds a SECOND,
independent restatement -- plain Python loops over numpy float32 scalars, written from the Rust source's behaviour, not from
the C -- and requires the two to agree bit for bit on small random volumes: a transcription error would have to be made
twice, the same way, to go unnoticed.

  lmip                          mips.rs:7-86      first local maximum behind the threshold window
  get_opacity / mida_internal   mips.rs:88-168    f32 state, `(1.0 / range) * (vl - img_min)`, break at alpha >= 1
  finite_difference / calc_fcm_intensity / fast_countour_mip_internal   mips.rs:171-279   (the subtraction happens in T and
                                wraps in a release build: quirk Q3)"""
import numpy as np
import pytest

F = np.float32


def _rays(img, axis):
    """(output index, the ray's samples in walking order) for every output pixel"""
    sz, sy, sx = img.shape
    if axis == 0:
        for y in range(sy):
            for x in range(sx):
                yield (y, x), img[:, y, x]
    elif axis == 1:
        for z in range(sz):
            for x in range(sx):
                yield (z, x), img[z, :, x]
    else:
        for z in range(sz):
            for y in range(sy):
                yield (z, y), img[z, y, :]


def lmip_py(img, axis, tmin, tmax):
    out = np.zeros([s for a, s in enumerate(img.shape) if a != axis], img.dtype)
    for idx, ray in _rays(img, axis):
        max_val = ray[0]
        start = tmin <= max_val <= tmax
        for val in ray:
            if val > max_val:
                max_val = val
            elif val < max_val and start:
                break
            if tmin <= val <= tmax:
                start = True
        out[idx] = max_val
    return out


def _opacity(vl, wl, ww):
    lo, hi = F(wl - F(ww / F(2.0))), F(wl + F(ww / F(2.0)))
    if vl < lo:
        return F(0.0)
    if vl > hi:
        return F(1.0)
    return F(F(vl - lo) / F(hi - lo))


def _numcast_i16(v):
    """num-traits NumCast f32 -> i16: None (the Rust code unwraps: a panic) unless -32769 < v < 32768, then truncation"""
    assert F(-32769.0) < v < F(32768.0), v
    return np.int16(int(v))


def mida_py(img, axis, wl, ww):
    f = img.astype(F)
    img_min, img_max = F(f.min()), F(f.max())
    rng = F(img_max - img_min)
    wl, ww = F(wl), F(ww)
    out = np.zeros([s for a, s in enumerate(img.shape) if a != axis], np.int16)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = F(F(1.0) / rng)
        for idx, ray in _rays(f, axis):
            fmax = alpha_p = colour_p = final = F(0.0)
            for vl in ray:
                fpi = F(inv * F(vl - img_min))
                if fpi > fmax:
                    dl = F(fpi - fmax)
                    fmax = fpi
                else:
                    dl = F(0.0)
                bt = F(F(1.0) - dl)
                alpha = _opacity(vl, wl, ww)
                colour = F(F(bt * colour_p) + F(F(F(F(1.0) - F(bt * alpha_p)) * fpi) * alpha))
                cur = F(F(bt * alpha_p) + F(F(F(1.0) - F(bt * alpha_p)) * alpha))
                colour_p, alpha_p, final = colour, cur, colour
                if cur >= F(1.0):
                    break
            out[idx] = _numcast_i16(F(F(rng * final) + img_min))
    return out


def fcm_volume_py(img, n, axis):
    sz, sy, sx = img.shape
    tmp = np.zeros(img.shape, np.int16)
    d_of = {0: 2, 1: 1, 2: 0}[axis]  # dir = unit vector: axis 0 -> gz, 1 -> gy, 2 -> gx

    def sub(a, b):  # i16 - i16 in i16, wrapping (release build)
        return F(np.int16((int(a) - int(b) + 32768) % 65536 - 32768))

    for z in range(sz):
        for y in range(sy):
            for x in range(sx):
                px, fx = max(x - 1, 0), min(x + 1, sx - 1)
                py, fy = max(y - 1, 0), min(y + 1, sy - 1)
                pz, fz = max(z - 1, 0), min(z + 1, sz - 1)
                g = (F(sub(img[z, y, fx], img[z, y, px]) / F(2.0)), F(sub(img[z, fy, x], img[z, py, x]) / F(2.0)),
                     F(sub(img[fz, y, x], img[pz, y, x]) / F(2.0)))
                gm = F(np.sqrt(F(F(F(g[0] * g[0]) + F(g[1] * g[1])) + F(g[2] * g[2]))))
                if gm == 0:
                    continue
                base = F(F(1.0) - abs(F(g[d_of] / gm)))
                sf = base if n == 1.0 else F(np.power(base, F(n)))
                tmp[z, y, x] = _numcast_i16(F(gm * sf))
    return tmp


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_lmip_and_mida_second_restatement(oracle, axis):
    rng = np.random.default_rng(100 + axis)
    for trial in range(6):
        shape = tuple(int(v) for v in rng.integers(3, 9, 3))
        img = rng.integers(-1200, 3200, shape).astype(np.int16)
        if trial == 0:
            img[:] = np.sort(img, axis=axis)  # monotone rays: LMIP never breaks
        oshape = [s for a, s in enumerate(shape) if a != axis]
        got = np.zeros(oshape, np.int16)
        oracle.lmip(img, axis, 700, 3033, got)
        assert np.array_equal(got, lmip_py(img, axis, 700, 3033))
        for wl, ww in ((300, 1500), (0, 2), (2500, 400)):
            got = np.zeros(oshape, np.int16)
            oracle.mida(img, axis, wl, ww, got)
            assert np.array_equal(got, mida_py(img, axis, wl, ww)), (shape, wl, ww)


@pytest.mark.parametrize("n", [0.3, 1.0, 2.0])
def test_contour_volume_and_its_three_projections_second_restatement(oracle, n):
    rng = np.random.default_rng(7)
    img = rng.integers(-1000, 3000, (5, 6, 7)).astype(np.int16)
    img[2, 3, :] = 500  # a flat row: zero gradients along x
    for axis in (0, 1, 2):
        tmp = fcm_volume_py(img, n, axis)
        assert np.array_equal(oracle.fcm_volume(img, n, axis), tmp)
        oshape = [s for a, s in enumerate(img.shape) if a != axis]
        for tmip in (0, 1, 2):
            got = np.zeros(oshape, np.int16)
            oracle.fast_countour_mip(img, n, axis, 300, 1500, tmip, got)
            want = tmp.max(axis) if tmip == 0 else lmip_py(tmp, axis, 700, 3033) if tmip == 1 else mida_py(tmp, axis, 300, 1500)
            assert np.array_equal(got, want), (axis, tmip)


def test_the_restated_glibc_powf_is_this_machines_libm_powf_bit_for_bit(tmp_path):
    """invesalius3_amd/csrc/glibc_powf.h (the power function of the contour MIP's kernels: glibc's algorithm and tables
    restated, in the plain and the FMA build) compiled for the HOST and run against libm's powf -- Rust's f32::powf
    (mips.rs:211) on this machine -- on 2 * 10^7 inputs: the contour MIP's own domain, arbitrary bit patterns (NaN, inf,
    subnormal), negative bases, the overflow / underflow range.  tools/check_powf.c exits 0 iff the build that glibc's
    selector picks on this CPU (FMA + AVX2 -> the FMA build) agrees on every input."""
    import os
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        import pytest
        pytest.skip("no host C++ compiler")
    exe = str(tmp_path / "check_powf")
    subprocess.run([cxx, "-O2", "-ffp-contract=off", "-o", exe, "-x", "c++", os.path.join(root, "tools", "check_powf.c"), "-lm"], check=True)
    r = subprocess.run([exe, "20"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "differs from libm" in r.stdout
