"""Resident-volume pipeline: upload the int16 volume once, then threshold -> region growing -> marching cubes ->
projections all run on buffers that stay in HBM (at one GPU the PCIe staging of the host-level
entry points dominates; the GUI data layer keeps `Slice.matrix` for the whole session, so does this class).

Everything here is a thin ctypes veneer over the `ivx_dev_*` C ABI (include/ivx.h); kernels are launched on one
HIP stream owned by the object, and `Timer` records HIP events on that same stream.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import _lib as L


class DeviceBuffer:
    """A hipMalloc'ed block (freed on close / GC)."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        L.check(L.lib().ivx_malloc(ctypes.byref(p), ctypes.c_size_t(max(self.nbytes, 16))), "ivx_malloc")
        self.ptr = p

    def upload(self, a: np.ndarray):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        L.check(L.lib().ivx_memcpy_h2d(self.ptr, L.ptr(a), ctypes.c_size_t(a.nbytes)))

    def download(self, shape, dtype, out: np.ndarray | None = None) -> np.ndarray:
        """`out`: a C-contiguous array with room for the result (e.g. `_lib.pinned_empty`); a view of it is returned"""
        if out is None:
            out = np.empty(shape, dtype)
        else:
            n = int(np.prod(shape)) * np.dtype(dtype).itemsize
            assert out.flags["C_CONTIGUOUS"] and out.nbytes >= n
            out = out.reshape(-1).view(np.uint8)[:n].view(dtype).reshape(shape)
        assert out.nbytes <= self.nbytes
        L.check(L.lib().ivx_memcpy_d2h(L.ptr(out), self.ptr, ctypes.c_size_t(out.nbytes)))
        return out

    def zero(self, stream=None, nbytes=None):
        L.check(L.lib().ivx_memset(self.ptr, 0, ctypes.c_size_t(self.nbytes if nbytes is None else nbytes), stream))

    def at(self, offset_bytes: int) -> ctypes.c_void_p:
        return ctypes.c_void_p(self.ptr.value + int(offset_bytes))

    def close(self):
        if self.ptr is not None and self.ptr.value:
            L.lib().ivx_free(self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TrackedBuffer(DeviceBuffer):
    """A DeviceBuffer that reports every access from outside its owner (`ptr`, and through it upload / zero / at):
    the owner then stops trusting what it derived from the contents (DeviceVolume's inside-bit plane of the mask,
    the "out_mask is all zero" note).  The owner's own kernels use `raw`."""

    def __init__(self, nbytes: int, on_touch):
        self._on_touch = None
        super().__init__(nbytes)
        self._on_touch = on_touch

    @property
    def ptr(self):
        if self._on_touch is not None:
            self._on_touch()
        return self._p

    @ptr.setter
    def ptr(self, v):
        self._p = v

    @property
    def raw(self):
        return self._p

    def download(self, shape, dtype, out=None) -> np.ndarray:  # reading changes nothing
        if out is None:
            out = np.empty(shape, dtype)
        else:
            n = int(np.prod(shape)) * np.dtype(dtype).itemsize
            assert out.flags["C_CONTIGUOUS"] and out.nbytes >= n
            out = out.reshape(-1).view(np.uint8)[:n].view(dtype).reshape(shape)
        assert out.nbytes <= self.nbytes
        L.check(L.lib().ivx_memcpy_d2h(L.ptr(out), self._p, ctypes.c_size_t(out.nbytes)))
        return out

    def raw_at(self, offset_bytes: int) -> ctypes.c_void_p:
        return ctypes.c_void_p(self._p.value + int(offset_bytes))


class Timer:
    """HIP events on the pipeline's stream: `with t.span('name'):` accumulates milliseconds per name."""

    def __init__(self, stream):
        self.stream = stream
        self.spans = []  # (name, ev0, ev1)
        self._pool = []
        # Names to record (None = all).  A recorded event is a barrier packet in the stream: ~4 us of idle GPU each, so
        # a caller timing whole steps by the wall clock brackets only what it needs.
        self.only = None

    class _Off:
        def __enter__(self):
            return None

        def __exit__(self, *a):
            return False

    _OFF = _Off()

    def _ev(self):
        if self._pool:
            return self._pool.pop()
        e = ctypes.c_void_p()
        L.check(L.lib().ivx_event_create(ctypes.byref(e)))
        return e

    class _Span:
        def __init__(self, t, name):
            self.t, self.name = t, name

        def __enter__(self):
            self.e0 = self.t._ev()
            L.check(L.lib().ivx_event_record(self.e0, self.t.stream))

        def __exit__(self, *a):
            e1 = self.t._ev()
            L.check(L.lib().ivx_event_record(e1, self.t.stream))
            self.t.spans.append((self.name, self.e0, e1))

    def span(self, name):
        if self.only is not None and name not in self.only:
            return Timer._OFF
        return Timer._Span(self, name)

    def collect(self) -> dict:
        """-> {name: [ms, ...]} and recycles the events."""
        out = {}
        for name, e0, e1 in self.spans:
            ms = ctypes.c_float(0)
            L.check(L.lib().ivx_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
            out.setdefault(name, []).append(ms.value)
            self._pool += [e0, e1]
        self.spans = []
        return out


class DeviceVolume:
    """An int16 (dz, dy, dx) volume resident in HBM with its dense uint8 mask and the work buffers of the path."""

    def __init__(self, image: np.ndarray | None = None, shape=None, spacing=(1.0, 1.0, 1.0), device: int | None = None):
        L.require_device()
        if device is not None:
            L.set_device(device)
        if image is not None:
            if image.dtype != np.int16 or image.ndim != 3:
                raise TypeError("image must be a 3-D int16 array")
            shape = image.shape
        self.shape = tuple(int(s) for s in shape)
        self.dz, self.dy, self.dx = self.shape
        self.n = self.dz * self.dy * self.dx
        self.spacing = tuple(float(s) for s in spacing)
        s = ctypes.c_void_p()
        L.check(L.lib().ivx_stream_create(ctypes.byref(s)))
        self.stream = s
        self.timer = Timer(self.stream)
        # Derived state the pipeline keeps about its buffers (see `threshold`): any access to these three buffers from
        # outside this class's own kernels (their .ptr / upload / zero) drops the corresponding note.
        self._mbits_valid = False   # self._mbits == (mask >= 127), the inside plane marching cubes needs at iso 127
        self._mbits_range = None    # (lo, hi) while additionally _mbits == (lo <= image <= hi)
        # (v_in, v_sel | None) while the mask's BYTES are known: 0 outside _mbits, v_sel where self.reached has a bit (if
        # given), v_in elsewhere inside -- marching cubes then needs no voxel at all (ivx_dev_mc_emit_levels)
        self._mask_levels = None
        self._out_bytes_zero = False  # the BYTES of out_mask are all zero
        self._mc_params_cache = {}
        # surface_prefetch: marching cubes' count + list queued on a second stream under the region growing
        self._stream2 = None
        self._sync_events = []
        self._prefetch = None     # (params, z0, capacity, plane version) of the outstanding prefetch
        self._gate = None         # device word the flood opens once its busy rounds are over
        self._gate_epoch = 0
        self._gate_armed = False  # armed and no flood has run since (so nobody will open it)
        self._mbits_version = 0   # bumped whenever the inside plane is rewritten
        self._out_pending = None      # fill value of a deferred `out_mask[reached] = fill` (self.reached still holds it)
        self._fuse = os.environ.get("IVX_NO_FUSE", "") == ""
        self.image = TrackedBuffer(self.n * 2, self._image_touched)
        self.mask = TrackedBuffer(self.n, self._mask_touched)       # dense interior of mask.matrix[1:,1:,1:]
        self.out_mask = TrackedBuffer(self.n, self._out_touched)    # region-growing `out` (styles.py:3190)
        if image is not None:
            self.image.upload(image)
        self.mask.zero(self.stream)
        self.zero_out_mask()
        # region growing bit planes + tile work-list
        self.plan = L.FloodPlan(self.dz, self.dy, self.dx, (self.dx + 63) // 64, 0)
        nb, ns = ctypes.c_size_t(0), ctypes.c_size_t(0)
        L.check(L.lib().ivx_flood_bits_bytes(ctypes.byref(self.plan), ctypes.byref(nb)))
        L.check(L.lib().ivx_flood_scratch_bytes(ctypes.byref(self.plan), ctypes.byref(ns)))
        self.cand = DeviceBuffer(nb.value)
        self.reached = DeviceBuffer(nb.value)
        self._mbits = DeviceBuffer(nb.value)
        self._plane_words = nb.value // 8
        self.flood_scratch = DeviceBuffer(ns.value)
        self._mc_scratch = None
        self._tris = None
        self._verts = None
        self._faces = None
        self._range_buf = None
        self._range_valid = False
        self.sync()

    # -- plumbing -------------------------------------------------------------------------------------
    def _image_touched(self):
        self._mbits_range = None
        self._range_valid = False

    def _mask_touched(self):
        self._mbits_valid = False
        self._mbits_range = None
        self._mask_levels = None
        self._mbits_version += 1

    # out_mask is the reference's throw-away `np.zeros_like` of styles.py:3190: the flood writes `fill` into it and the
    # caller turns it into `mask[out_mask.astype(bool)] = 254`.  The pipeline keeps it as what the flood really produces
    # -- the reached bit plane -- and writes the bytes only when somebody looks at them: `_out_pending` remembers a
    # deferred `out_mask[reached] = fill` over bytes that are known to be zero.  Zeroing it again is then free, and the
    # next flood does not have to read it.  Every path that reads or exposes the bytes calls _materialize_out() first.
    def _out_logically_zero(self) -> bool:
        return self._out_bytes_zero and self._out_pending is None

    def _materialize_out(self):
        if self._out_pending is not None:
            L.check(L.lib().ivx_dev_flood_apply(ctypes.byref(self.plan), self.reached.ptr, L.U8, self.out_mask.raw,
                                                ctypes.c_double(self._out_pending), self.stream))
            self._out_pending = None
            self._out_bytes_zero = False

    def _out_touched(self):
        self._materialize_out()
        self._out_bytes_zero = False

    def zero_out_mask(self):
        """out_mask = zeros (the np.zeros of styles.py:3190)"""
        if not self._out_bytes_zero:
            L.check(L.lib().ivx_memset(self.out_mask.raw, 0, ctypes.c_size_t(self.n), self.stream))
        self._out_bytes_zero, self._out_pending = True, None

    def sync(self):
        L.check(L.lib().ivx_stream_synchronize(self.stream))

    def close(self):
        if getattr(self, "stream", None) is None:
            return  # already closed (or never fully built)
        self._out_pending = None
        self._prefetch = None
        if getattr(self, "flood_scratch", None) is not None and self.flood_scratch.ptr is not None:
            L.lib().ivx_dev_flood_disarm_gate(self.flood_scratch.ptr)  # an unconsumed arm must not outlive the buffers
        self._gate_armed = False
        if self._stream2 is not None:
            L.lib().ivx_stream_destroy(self._stream2)  # synchronises it first
            self._stream2 = None
        for e in self._sync_events:
            L.lib().ivx_event_destroy(e)
        self._sync_events = []
        for b in (self.image, self.mask, self.out_mask):
            b._on_touch = None  # freeing is not "somebody looked at the contents"
        for b in (self.image, self.mask, self.out_mask, self.cand, self.reached, self._mbits, self.flood_scratch,
                  self._mc_scratch, self._tris, self._verts, self._faces, self._gate, getattr(self, "_range_buf", None)):
            if b is not None:
                b.close()
        if self.stream is not None:
            L.lib().ivx_stream_destroy(self.stream)
            self.stream = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def download_mask(self, out: np.ndarray | None = None) -> np.ndarray:
        self.sync()
        return self.mask.download(self.shape, np.uint8, out)

    def download_out_mask(self) -> np.ndarray:
        self._materialize_out()
        self.sync()
        return self.out_mask.download(self.shape, np.uint8)

    # -- threshold (slice_.py:1240-1247 / 1722-1769) -------------------------------------------------------
    def threshold(self, lo: int, hi: int, preserve: bool = False):
        """mask = 255 where lo <= image <= hi else 0 (with the preserve rule of do_threshold_to_a_slice when asked).
        When the rows are whole 64-voxel words the same pass also writes the mask's inside-bit plane (1/8 B/voxel
        more): it IS the candidate plane of a region growing with the same thresholds into a zero out_mask, and the
        inside plane of marching cubes at iso 127 -- both then skip their own pass over the volume."""
        lib = L.lib()
        self._join_prefetch()  # a count still reading the plane must not see it change
        if self._fuse and not preserve and self.dx % 64 == 0:
            self._mbits_version += 1
            L.check(lib.ivx_dev_threshold_i16_bits(self.image.raw, c64(self.dz), c64(self.dy), c64(self.dx), int(lo), int(hi),
                                                   self.mask.raw, self._mbits.ptr, self.stream), "threshold")
            self._mbits_valid, self._mbits_range = True, (int(lo), int(hi))
            self._mask_levels = (255, None)  # mask == 255 * plane, byte for byte
            return
        L.check(lib.ivx_dev_threshold_i16(self.image.raw, c64(self.dz), c64(self.dy), c64(self.dx), int(lo), int(hi),
                                          int(bool(preserve)), None, self.mask.raw, self.stream), "threshold")
        self._mask_touched()

    # -- region growing (floodfill.rs:96-166 on the image; styles.py:3151-3216) ----------------------------
    def lut_image_255(self, ww, wl) -> DeviceBuffer:
        """get_LUT_value_255(image, ww, wl) (imagedata_utils.py:540-552) as a resident int16 volume: the image the
        "dynamic" and "confidence" region-growing modes flood when use_ww_wl is set (styles.py:3166-3171, 3222-3225)."""
        buf = DeviceBuffer(self.n * 2)
        L.check(L.lib().ivx_dev_lut_i16(self.image.raw, c64(self.n), ctypes.c_double(float(ww)), ctypes.c_double(float(wl)),
                                        1, buf.ptr, self.stream), "lut_image_255")
        return buf

    def region_grow(self, seeds_xyz, t0, t1, strct, fill: int = 1, select_value: int | None = 254,
                    image: DeviceBuffer | None = None) -> int:
        """floodfill_threshold(image, seeds, t0, t1, fill, strct, out_mask) followed (when select_value is not None)
        by `mask[out_mask.astype(bool)] = select_value` (styles.py:3214).  `image` = an alternative resident int16
        volume (e.g. lut_image_255).  Returns the number of global rounds."""
        lib = L.lib()
        img_ptr = image.ptr if image is not None else self.image.raw
        s3 = np.ascontiguousarray(strct, dtype=np.uint8)
        bits = ctypes.c_uint32(0)
        L.check(lib.ivx_flood_strct_bits(L.ptr(s3), L.i64(s3.shape), ctypes.byref(bits)))
        self.plan.strct_bits = bits.value
        seeds = np.ascontiguousarray(np.array([tuple(s) for s in seeds_xyz], dtype=np.int64).reshape(-1, 3))
        p = ctypes.byref(self.plan)
        st = self.stream
        t0, t1 = float(int(t0)), float(int(t1))  # wrapper int() truncation for integer images
        self._before_flood()
        cand, shared = self._candidate_plane(image, t0, t1, fill)
        rounds, pending = ctypes.c_int(0), ctypes.c_int(0)
        # clear + seed + run (an in-range seed is already a candidate here: the kernel's OR is a no-op on a shared plane).
        # The call may return right behind the flood's resident launch: the writes that depend on the reached plane are
        # queued at once, so they start the moment the flood ends, and the round count is fetched afterwards.
        L.check(lib.ivx_dev_flood_grow_async(p, L.I16, img_ptr, ctypes.c_double(t0), ctypes.c_double(t1), L.ptr(seeds),
                                             c64(len(seeds)), cand.ptr, self.reached.ptr, self.flood_scratch.ptr,
                                             ctypes.byref(rounds), ctypes.byref(pending), st), "region_grow")
        self._gate_armed = False  # an armed gate has been opened by this flood
        self._apply_reached(fill, select_value, shared)
        if pending.value:
            late = ctypes.c_int(0)
            L.check(lib.ivx_dev_flood_wait(p, cand.ptr, self.reached.ptr, self.flood_scratch.ptr, ctypes.byref(rounds),
                                           ctypes.byref(late), st), "region_grow")
            if late.value:  # the launch ended early and the flood was completed only now: the dependent writes once more
                self._apply_reached(fill, select_value, shared)
        return rounds.value

    def _before_flood(self):
        """the reached plane is about to be cleared: a deferred out_mask write that is still wanted must land first, and a
        note that describes the mask's bytes through that plane dies with it"""
        self._materialize_out()
        if self._mask_levels is not None and self._mask_levels[1] is not None:
            self._mask_levels = None

    def _candidate_plane(self, image, t0, t1, fill):
        """The flood's candidate plane: in range AND out_mask != fill.  With out_mask known to be zero (fill != 0) and
        the mask's plane known to be "image in [t0, t1]", the plane the threshold pass left behind is exactly that and
        no pass over the volume is needed.  Returns (buffer, shared?)."""
        shared = (image is None and self._mbits_valid and self._out_logically_zero() and int(fill) != 0
                  and self._mbits_range == (int(t0), int(t1)))
        if shared:
            return self._mbits, True
        img_ptr = image.ptr if image is not None else self.image.raw
        L.check(L.lib().ivx_dev_flood_candidates(ctypes.byref(self.plan), L.I16, img_ptr, ctypes.c_double(t0),
                                                 ctypes.c_double(t1), self.out_mask.raw, 1, ctypes.c_double(fill),
                                                 self.cand.ptr, self.stream))
        return self.cand, False

    def _apply_reached(self, fill, select_value, shared):
        """out_mask[reached] = fill and, when asked, mask[reached] = select_value; keeps the notes in step."""
        lib, p, st = L.lib(), ctypes.byref(self.plan), self.stream
        defer = self._out_logically_zero() and int(fill) != 0
        if defer:
            self._out_pending = int(fill)  # bytes stay zero; self.reached carries the result until it is needed
            if select_value is not None:
                L.check(lib.ivx_dev_flood_apply(p, self.reached.ptr, L.U8, self.mask.raw, ctypes.c_double(int(select_value)), st))
        elif select_value is not None:
            L.check(lib.ivx_dev_flood_apply2(p, self.reached.ptr, self.out_mask.raw, int(fill), self.mask.raw,
                                             int(select_value), st))
            self._out_bytes_zero = False
        else:
            L.check(lib.ivx_dev_flood_apply(p, self.reached.ptr, L.U8, self.out_mask.raw, ctypes.c_double(fill), st))
            self._out_bytes_zero = False
        if select_value is not None:
            # the mask's bytes stay known when it was 255 * plane and the selected voxels become another inside value
            lv = self._mask_levels
            self._mask_levels = (lv[0], int(select_value)) if (lv is not None and lv[1] is None and int(select_value) >= 127
                                                                and self._mbits_valid) else None
        if select_value is not None and self._mbits_valid and not (shared and int(select_value) >= 127):
            # mask[reached] = select_value: keep the inside plane in step (reached is a subset of a shared plane)
            self._join_prefetch()
            self._mbits_version += 1
            L.check(lib.ivx_dev_bits_combine(self._mbits.ptr, self.reached.ptr, c64(self._plane_words),
                                             0 if int(select_value) >= 127 else 1, st))
            self._mbits_range = None

    def region_grow_confidence(self, seed_xyz, strct, confid_mult=2.5, confid_iters=3, select_value=254,
                               image: DeviceBuffer | None = None):
        """do_rg_confidence (invesalius/data/styles.py:3220-3251): `confid_iters` rounds of
        mean +- confid_mult * std over the grown region -> floodfill_threshold into the SAME out_mask (never cleared:
        SURVEY quirk Q4).  Statistics are exact integer sums on the GPU; mean/std are formed in float64 on the host
        (std = sqrt(E[x^2] - mean^2): differs from numpy's two-pass value by rounding only, far below the int()
        truncation the wrapper applies to t0/t1)."""
        lib = L.lib()
        x, y, z = (int(v) for v in seed_xyz)
        sel = np.zeros(self.shape, np.uint8)
        sel[max(z - 1, 0):z + 2, max(y - 1, 0):y + 2, max(x - 1, 0):x + 2] = 1  # styles.py:3226-3235
        d_sel = DeviceBuffer(self.n)
        d_sel.upload(sel)
        rounds = 0
        for _ in range(int(confid_iters)):
            acc = (ctypes.c_int64 * 3)()
            L.check(lib.ivx_dev_masked_stats_i16(image.ptr if image is not None else self.image.raw, d_sel.ptr, c64(self.n),
                                                 acc, self.stream))
            cnt, s1, s2 = int(acc[0]), int(acc[1]), int(acc[2])
            mean = s1 / cnt
            var = max(s2 / cnt - mean * mean, 0.0)
            std = float(np.sqrt(var))
            t0, t1 = mean - std * confid_mult, mean + std * confid_mult
            rounds += self.region_grow([(x, y, z)], t0, t1, strct, fill=1, select_value=None, image=image)
            self._materialize_out()
            L.check(lib.ivx_dev_or_equal_u8(d_sel.ptr, self.out_mask.raw, c64(self.n), 1, self.stream))
        if select_value is not None:
            self._materialize_out()
            L.check(lib.ivx_dev_flood_apply_where(self.mask.ptr, self.out_mask.raw, c64(self.n), 1, int(select_value),
                                                  self.stream))  # mask.ptr: drops the inside-plane note
        self.sync()
        d_sel.close()
        return rounds

    def reached_count(self) -> int:
        n = ctypes.c_int64(0)
        L.check(L.lib().ivx_dev_flood_count(ctypes.byref(self.plan), self.reached.ptr, ctypes.byref(n), self.stream))
        return n.value

    # -- marching cubes over the whole resident volume (surface_process.py:100-186) ------------------------
    def _mc_params(self, from_binary, min_value, max_value, fill_border_holes=True, z0=0, z1=None, roi_start=None,
                   pad_bottom=None, pad_top=None) -> L.McParams:
        z1 = self.dz if z1 is None else z1
        key = (bool(from_binary), min_value, max_value, bool(fill_border_holes), z0, z1, roi_start, pad_bottom, pad_top,
               tuple(self.spacing))
        hit = self._mc_params_cache.get(key)
        if hit is not None:
            return hit
        p = L.McParams()
        pb = (z0 == 0) if pad_bottom is None else pad_bottom
        pt = (z1 >= self.dz) if pad_top is None else pad_top
        if fill_border_holes:
            p.pad_xy, p.pad_bottom, p.pad_top, p.vtk_pz = 1, int(pb), int(pt), int(pb)
        else:
            p.pad_xy = p.pad_bottom = p.pad_top = p.vtk_pz = 0
        p.nz, p.ny, p.nx = z1 - z0, self.dy, self.dx
        p.roi_start = z0 if roi_start is None else roi_start
        p.spacing[:] = self.spacing
        if from_binary:
            p.dtype, p.niso, p.pad_value = L.U8, 1, 0.0
            p.iso[:] = [127.0, 0.0]
        else:
            p.dtype, p.niso, p.pad_value = L.I16, 2, float(np.iinfo(np.int16).min)
            p.iso[:] = [float(min_value), float(max_value)]
        if len(self._mc_params_cache) < 64:
            self._mc_params_cache[key] = p  # callers treat the struct as read-only
        return p

    # -- marching cubes -----------------------------------------------------------------------------------------------
    def _surface_params(self, from_binary, min_value, max_value, fill_border_holes):
        """(piece parameters, first slice) of this volume's surface call; SlabVolume answers with its slab's piece"""
        return self._mc_params(from_binary, min_value, max_value, fill_border_holes), 0

    def _mc_setup(self, p, z0):
        """scratch of the right size, the voxel source and -- when the pipeline still holds it -- the inside plane"""
        nb = ctypes.c_size_t(0)
        L.check(L.lib().ivx_dev_mc_scratch_bytes(ctypes.byref(p), ctypes.byref(nb)))
        if self._mc_scratch is None or self._mc_scratch.nbytes < nb.value:
            self._join_prefetch()
            if self._mc_scratch is not None:
                self._mc_scratch.close()
            self._mc_scratch = DeviceBuffer(nb.value)
        isz = 1 if p.dtype == L.U8 else 2
        src = (self.mask if p.dtype == L.U8 else self.image).raw_at(z0 * self.dy * self.dx * isz)
        plane = None
        if (p.dtype == L.U8 and p.niso == 1 and p.iso[0] == 127.0 and self._mbits_valid and p.ny == self.dy
                and p.nx == self.dx):
            plane = self._mbits.at(z0 * self.dy * ((self.dx + 63) // 64) * 8)
        return src, plane

    def _emit(self, p, src, plane, z0, cap, stream):
        """the triangle emit of a counted piece: from the known byte levels of the mask when the pipeline can prove them
        (no voxel is read), else from the voxels"""
        lib, lv = L.lib(), self._mask_levels
        # (the same predicate as marching_cubes_indexed: a caller's own `params` may carry another padding value or levels
        # that do not straddle the iso-value the way the constants assume)
        if (plane is not None and lv is not None and self._fuse and os.environ.get("IVX_MC_LEVELS", "1") != "0"
                and float(p.pad_value) == 0.0 and lv[0] > 127 and (lv[1] is None or lv[1] > 127)):
            if lv[1] is None:
                sel, v_sel = plane, float(lv[0])  # (no second level: any plane will do, both values are the same)
            else:
                sel, v_sel = self.reached.at(z0 * self.dy * ((self.dx + 63) // 64) * 8), float(lv[1])
            L.check(lib.ivx_dev_mc_emit_levels(ctypes.byref(p), self._mc_scratch.ptr, sel, ctypes.c_double(0.0),
                                               ctypes.c_double(float(lv[0])), ctypes.c_double(v_sel), self._tris.ptr, c64(cap), stream),
                    "mc_emit")
        else:
            L.check(lib.ivx_dev_mc_emit(ctypes.byref(p), src, self._mc_scratch.ptr, self._tris.ptr, c64(cap), stream), "mc_emit")

    def _surface_one_launch(self, p, src, plane, z0, cap, stream):
        """count + offsets + emit of a one-iso piece in ONE kernel (ivx_dev_mc_surface / _levels: no per-word counts, no scan
        launch, no triangle list); `ivx_dev_mc_total` reads the count afterwards.  Same predicate as `_emit` for the levels form."""
        lib, lv = L.lib(), self._mask_levels
        if (plane is not None and lv is not None and self._fuse and os.environ.get("IVX_MC_LEVELS", "1") != "0"
                and float(p.pad_value) == 0.0 and lv[0] > 127 and (lv[1] is None or lv[1] > 127)):
            if lv[1] is None:
                sel, v_sel = plane, float(lv[0])
            else:
                sel, v_sel = self.reached.at(z0 * self.dy * ((self.dx + 63) // 64) * 8), float(lv[1])
            L.check(lib.ivx_dev_mc_surface_levels(ctypes.byref(p), plane, sel, ctypes.c_double(0.0), ctypes.c_double(float(lv[0])),
                                                  ctypes.c_double(v_sel), self._mc_scratch.ptr, self._tris.ptr, c64(cap), stream), "mc_surface")
        else:
            L.check(lib.ivx_dev_mc_surface(ctypes.byref(p), src, plane, self._mc_scratch.ptr, self._tris.ptr, c64(cap), stream), "mc_surface")

    def _second_stream(self):
        if self._stream2 is None:
            s = ctypes.c_void_p()
            if os.environ.get("IVX_PREFETCH_PRIORITY", "low") == "low":
                L.check(L.lib().ivx_stream_create_low_priority(ctypes.byref(s)))  # fills idle CUs, yields to the flood
            else:
                L.check(L.lib().ivx_stream_create(ctypes.byref(s)))
            self._stream2 = s
            for _ in range(2):
                e = ctypes.c_void_p()
                L.check(L.lib().ivx_event_create_sync(ctypes.byref(e)))
                self._sync_events.append(e)
        return self._stream2

    def _join_prefetch(self):
        """an outstanding prefetch is about to lose its inputs (or its scratch): let it finish, then forget it"""
        if self._prefetch is not None:
            if self._gate_armed:
                L.check(L.lib().ivx_dev_gate_open(self._gate.ptr, ctypes.c_uint32(self._gate_epoch), self.stream))
                L.check(L.lib().ivx_dev_flood_disarm_gate(self.flood_scratch.ptr))
                self._gate_armed = False
            L.check(L.lib().ivx_stream_synchronize(self._stream2))
            self._prefetch = None

    def surface_prefetch(self, from_binary=True, min_value=0, max_value=0, fill_border_holes=True,
                         behind_flood: bool = True) -> bool:
        """Queue the part of the coming `marching_cubes(...)` call (same arguments) that depends on the mask's inside
        plane only -- count, scan and triangle list -- on a second stream, NOW: it then runs under whatever follows on
        the main stream (typically the region growing, whose `mask[reached] = 254` changes values the emit
        interpolates but not which side of iso 127 a voxel is on).  Needs the plane the threshold pass left behind
        and a triangle buffer from an earlier call (its size is the capacity); returns False, doing nothing, otherwise.
        Any change to the plane in between simply voids the prefetch.  `behind_flood`: hold the prefetched passes back
        until the next region growing has left its throughput-bound first rounds (they would slow each other down;
        its latency-bound tail leaves most of the GPU idle) -- or for 300 us at most, if no region growing follows."""
        if not from_binary or not self._fuse or self._tris is None or not self._mbits_valid:
            return False
        self._join_prefetch()
        p, z0 = self._surface_params(from_binary, min_value, max_value, fill_border_holes)
        src, plane = self._mc_setup(p, z0)
        if plane is None:
            return False
        lib, s2 = L.lib(), self._second_stream()
        cap = self._tris.nbytes // 36
        L.check(lib.ivx_event_record(self._sync_events[0], self.stream))   # the plane is complete at this point
        L.check(lib.ivx_stream_wait_event(s2, self._sync_events[0]))
        if behind_flood:
            if self._gate is None:
                self._gate = DeviceBuffer(64)
                self._gate.zero(self.stream)
                self.sync()
            self._gate_epoch = (self._gate_epoch % 0x7fffffff) + 1
            L.check(lib.ivx_dev_flood_arm_gate(self.flood_scratch.ptr, self._gate.ptr, ctypes.c_uint32(self._gate_epoch),
                                               ctypes.c_uint32(512)))
            L.check(lib.ivx_dev_gate_wait(self._gate.ptr, ctypes.c_uint32(self._gate_epoch), ctypes.c_uint32(300), s2))
            self._gate_armed = True
        L.check(lib.ivx_dev_mc_count_bits_async(ctypes.byref(p), plane, self._mc_scratch.ptr, s2), "mc_count")
        L.check(lib.ivx_dev_mc_list(ctypes.byref(p), self._mc_scratch.ptr, c64(cap), s2), "mc_list")
        self._prefetch = (p, z0, cap, self._mbits_version)
        return True

    def marching_cubes(self, from_binary=True, min_value=0, max_value=0, fill_border_holes=True, download=False,
                       params: L.McParams | None = None, z0: int = 0, out: np.ndarray | None = None):
        """count + emit on the resident mask (from_binary, iso 127) or image (two iso-values).  Returns the triangle
        count, or the (T,3,3) float32 soup when download=True (written into `out` -- e.g. a `_lib.pinned_empty` array with
        room for it -- when given)."""
        lib = L.lib()
        if params is not None:
            p = params
        else:
            p, z0 = self._surface_params(from_binary, min_value, max_value, fill_border_holes)
        pf, n = self._prefetch, ctypes.c_int64(0)
        if pf is not None:
            ok = (pf[0] is p and pf[1] == z0 and self._tris is not None and pf[2] == self._tris.nbytes // 36
                  and pf[3] == self._mbits_version and self._mbits_valid)
            if not ok:
                self._join_prefetch()
            else:
                # count and list are done or running on the second stream; the emit needs the mask's final bytes
                src, plane = self._mc_setup(p, z0)
                s2, cap = self._stream2, pf[2]
                if self._gate_armed:  # no flood came in between: open the gate by hand instead of waiting it out
                    L.check(lib.ivx_dev_gate_open(self._gate.ptr, ctypes.c_uint32(self._gate_epoch), self.stream))
                    L.check(lib.ivx_dev_flood_disarm_gate(self.flood_scratch.ptr))
                    self._gate_armed = False
                L.check(lib.ivx_event_record(self._sync_events[1], self.stream))
                L.check(lib.ivx_stream_wait_event(s2, self._sync_events[1]))
                L.check(lib.ivx_dev_mc_emit(ctypes.byref(p), src, self._mc_scratch.ptr, self._tris.ptr, c64(cap), s2), "mc_emit")
                L.check(lib.ivx_dev_mc_total(ctypes.byref(p), self._mc_scratch.ptr, ctypes.byref(n), s2), "mc_total")
                self._prefetch = None  # the host has seen the total: everything on the second stream has finished
                nt = n.value
                if nt <= cap:
                    if download:
                        return self._tris.download((nt, 3, 3), np.float32, out if out is not None and out.nbytes >= nt * 36 else None)
                    return nt
                # the surface outgrew the buffer: take the ordinary path below (it re-counts and re-emits)
        src, plane = self._mc_setup(p, z0)
        n = ctypes.c_int64(0)
        if self._tris is not None and p.niso == 1 and os.environ.get("IVX_MC_ONE_LAUNCH", "0") == "1":
            # opt-in (measured slower at 512^3, equal at 1024^3: csrc/k_mc.hip, k_mc_fused): the whole surface in ONE launch
            # (count, output offsets by a look-back across the workgroups, emit); the count is read afterwards and only a
            # surface that outgrew the buffer is emitted again
            cap = self._tris.nbytes // 36
            with self.timer.span("mc_emit"):
                self._surface_one_launch(p, src, plane, z0, cap, self.stream)
            L.check(lib.ivx_dev_mc_total(ctypes.byref(p), self._mc_scratch.ptr, ctypes.byref(n), self.stream), "mc_total")
            nt = n.value
            if nt > cap:
                self._tris.close()
                self._tris = DeviceBuffer(int(nt * 36 * 1.25) + 4096)
                self._surface_one_launch(p, src, plane, z0, nt, self.stream)
        elif self._tris is not None:
            # steady state: the triangle buffer of the previous call gives a capacity, so count, list and emit are queued
            # back to back and the count is read afterwards (no host round trip between the two halves)
            cap = self._tris.nbytes // 36
            with self.timer.span("mc_count"):
                if plane is not None:
                    L.check(lib.ivx_dev_mc_count_bits_async(ctypes.byref(p), plane, self._mc_scratch.ptr, self.stream), "mc_count")
                else:
                    L.check(lib.ivx_dev_mc_count_async(ctypes.byref(p), src, self._mc_scratch.ptr, self.stream), "mc_count")
            with self.timer.span("mc_emit"):
                self._emit(p, src, plane, z0, cap, self.stream)
            L.check(lib.ivx_dev_mc_total(ctypes.byref(p), self._mc_scratch.ptr, ctypes.byref(n), self.stream), "mc_total")
            nt = n.value
            if nt > cap:  # the surface outgrew the buffer: emit again into a larger one
                self._tris.close()
                self._tris = DeviceBuffer(int(nt * 36 * 1.25) + 4096)
                self._emit(p, src, plane, z0, nt, self.stream)
        else:
            with self.timer.span("mc_count"):
                if plane is not None:
                    L.check(lib.ivx_dev_mc_count_bits(ctypes.byref(p), plane, self._mc_scratch.ptr, ctypes.byref(n), self.stream),
                            "mc_count")
                else:
                    L.check(lib.ivx_dev_mc_count(ctypes.byref(p), src, self._mc_scratch.ptr, ctypes.byref(n), self.stream), "mc_count")
            nt = n.value
            self._tris = DeviceBuffer(int(nt * 36 * 1.25) + 4096)
            with self.timer.span("mc_emit"):
                self._emit(p, src, plane, z0, nt, self.stream)
        if download:
            self.sync()
            return self._tris.download((nt, 3, 3), np.float32, out if out is not None and out.nbytes >= nt * 36 else None)
        return nt

    def marching_cubes_indexed(self, from_binary=True, min_value=0, max_value=0, fill_border_holes=True, download=False,
                               params: L.McParams | None = None, z0: int = 0):
        """Same surface with coincident points merged (ivx.h "indexed surface").  Returns (n_verts, n_tris), or the
        (V,3) float32 / (T,3) int32 arrays when download=True; the device buffers stay in self._verts / self._faces."""
        lib = L.lib()
        if params is not None:
            p = params
        else:
            p, z0 = self._surface_params(from_binary, min_value, max_value, fill_border_holes)
        self._join_prefetch()  # same scratch
        src, plane = self._mc_setup(p, z0)
        nt, nv = ctypes.c_int64(0), ctypes.c_int64(0)
        with self.timer.span("mc_count"):
            if plane is not None:
                L.check(lib.ivx_dev_mc_count_bits(ctypes.byref(p), plane, self._mc_scratch.ptr, ctypes.byref(nt), self.stream),
                        "mc_count")
            else:
                L.check(lib.ivx_dev_mc_count(ctypes.byref(p), src, self._mc_scratch.ptr, ctypes.byref(nt), self.stream), "mc_count")
        # the mask's bytes known through the pipeline's planes (see _emit): neither the strictly-inside plane nor a vertex
        # needs a voxel
        lv = self._mask_levels
        levels = (plane is not None and lv is not None and self._fuse and os.environ.get("IVX_MC_LEVELS", "1") != "0"
                  and float(p.pad_value) == 0.0 and lv[0] > 127 and (lv[1] is None or lv[1] > 127))
        with self.timer.span("mci_count"):
            if levels:
                L.check(lib.ivx_dev_mc_indexed_count_levels(ctypes.byref(p), self._mc_scratch.ptr, ctypes.byref(nv), self.stream),
                        "mc_indexed_count")
            else:
                L.check(lib.ivx_dev_mc_indexed_count(ctypes.byref(p), src, self._mc_scratch.ptr, ctypes.byref(nv), self.stream),
                        "mc_indexed_count")
        for name, need in (("_verts", nv.value * 12), ("_faces", nt.value * 12)):
            buf = getattr(self, name)
            if buf is None or buf.nbytes < need:
                if buf is not None:
                    buf.close()
                setattr(self, name, DeviceBuffer(int(need * 1.25) + 4096))
        with self.timer.span("mci_emit"):
            if levels:
                if lv[1] is None:
                    sel, v_sel = None, float(lv[0])
                else:
                    sel, v_sel = self.reached.at(z0 * self.dy * ((self.dx + 63) // 64) * 8), float(lv[1])
                L.check(lib.ivx_dev_mc_indexed_emit_levels(ctypes.byref(p), self._mc_scratch.ptr, sel, ctypes.c_double(0.0),
                                                           ctypes.c_double(float(lv[0])), ctypes.c_double(v_sel), self._verts.ptr,
                                                           c64(nv.value), self._faces.ptr, c64(nt.value), self.stream),
                        "mc_indexed_emit")
            else:
                L.check(lib.ivx_dev_mc_indexed_emit(ctypes.byref(p), src, self._mc_scratch.ptr, self._verts.ptr, c64(nv.value),
                                                    self._faces.ptr, c64(nt.value), self.stream), "mc_indexed_emit")
        if download:
            self.sync()
            return self._verts.download((nv.value, 3), np.float32), self._faces.download((nt.value, 3), np.int32)
        return nv.value, nt.value

    # -- watershed (watershed_process.py:19-60 + the merge of styles.py:2147-2152) on the resident volume -------------------
    def watershed(self, markers: np.ndarray, strct, use_ww_wl: bool = False, wl=0, ww=0, overwrite: bool = False,
                  algorithm: str = "Watershed IFT", mg_size=(3, 3, 3)):
        """do_watershed followed by the caller's merge rule, all in HBM: cost image (LUT or ``image - image.min()``) ->
        [`algorithm == "Watershed"`: morphological gradient of `mg_size`] -> marker flood (`ivx_dev_watershed_ift`, or
        `ivx_dev_watershed_sk` for "Watershed", the GUI's default) -> ``mask`` gets 253 where the flood says 1 and 2 where
        it says 2 (only over cells that hold 0 / 2 / 253, or over everything after zeroing with `overwrite`).
        `markers`: int8 / int16 (0, 1, 2) array of the volume's shape, uploaded for the call.  Returns the flood's stats."""
        mk = np.ascontiguousarray(markers)
        if mk.shape != self.shape or mk.dtype.type not in (np.int8, np.int16):
            raise TypeError("markers must be an int8 / int16 array of the volume's shape")
        s3 = np.zeros((3, 3, 3), np.uint8)
        s3[:] = np.asarray(strct).astype(bool)
        lib, n = L.lib(), self.n
        d_mk, d_cost, d_lab = DeviceBuffer(mk.nbytes), DeviceBuffer(n * 2), DeviceBuffer(n)
        d_mk.upload(mk)
        try:
            if use_ww_wl:
                L.check(lib.ivx_dev_lut_u16(self.image.raw, c64(n), ctypes.c_double(float(ww)), ctypes.c_double(float(wl)), 0,
                                            d_cost.ptr, self.stream), "lut")
            else:
                mm = DeviceBuffer(64)
                L.check(lib.ivx_dev_minmax_f32(L.I16, self.image.raw, c64(n), mm.ptr, self.stream))
                self.sync()
                imin = int(mm.download((2,), np.float32)[0])
                mm.close()
                L.check(lib.ivx_dev_shift_min_u16(self.image.raw, c64(n), imin, d_cost.ptr, self.stream), "min shift")
            stats = (ctypes.c_int64 * 16)()
            mdt = L.I16 if mk.dtype == np.int16 else L.I8
            if algorithm == "Watershed":  # watershed_process.py:33-39,47-52
                d_grad = DeviceBuffer(n * 2)
                try:
                    gsz = (ctypes.c_int * 3)(*[int(v) for v in mg_size])
                    L.check(lib.ivx_dev_morph_gradient_u16(d_cost.ptr, c64(self.dz), c64(self.dy), c64(self.dx), gsz, d_grad.ptr,
                                                           self.stream), "gradient")
                    with self.timer.span("watershed_flood"):
                        L.check(lib.ivx_dev_watershed_sk(d_grad.ptr, mdt, d_mk.ptr, c64(self.dz), c64(self.dy), c64(self.dx), L.ptr(s3),
                                                         None, None, d_lab.ptr, None, stats, self.stream), "watershed_sk")
                    self.sync()
                finally:
                    d_grad.close()
            else:
                with self.timer.span("watershed_flood"):
                    L.check(lib.ivx_dev_watershed_ift(d_cost.ptr, mdt, d_mk.ptr, c64(self.dz), c64(self.dy), c64(self.dx), L.ptr(s3),
                                                      None, d_lab.ptr, None, stats, self.stream), "watershed_ift")
            L.check(lib.ivx_dev_watershed_merge(self.mask.ptr, d_lab.ptr, c64(n), int(bool(overwrite)), self.stream), "merge")
            self.sync()
        finally:
            for b in (d_mk, d_cost, d_lab):
                b.close()
        names = (("rounds", "tile_visits", "levels", "generations", "markers", "generation0", "tied_markers_of_different_labels")
                 if algorithm == "Watershed" else ("rounds", "tile_visits", "levels", "time_stamps", "markers", "entries", "tiles"))
        res = {k: int(v) for k, v in zip(names, stats)}
        if algorithm == "Watershed":
            from .watershed_process import _warn_ties
            _warn_ties(res["tied_markers_of_different_labels"], "DeviceVolume.watershed")
        return res

    # -- projections ---------------------------------------------------------------------------------
    def project(self, axis: int, op: int, out: DeviceBuffer):
        L.check(L.lib().ivx_dev_mip_reduce(L.I16, self.image.raw, c64(self.dz), c64(self.dy), c64(self.dx), int(axis),
                                           int(op), out.ptr, self.stream), "project")

    def image_range(self) -> DeviceBuffer:
        """float32[2] on the device: min and max of the resident image -- mida_internal's own pre-pass over the volume
        (mips.rs:113-121).  The reference runs it inside every call; a resident volume keeps the result until the image's
        bytes change (any access to `image` from outside the pipeline's kernels drops it, like the bit-plane notes), the way
        `Slice` keeps the image's histogram (slice_.py:192-194; SURVEY 8d counts the cached range as allowed).
        `forget_image_range()` drops it by hand -- and every writer inside this package that goes to `image.raw` /
        `image.raw_at()` directly (past the tracked accessor) must call it."""
        if getattr(self, "_range_buf", None) is None:
            self._range_buf = DeviceBuffer(64)
            self._range_valid = False
        if not self._range_valid:
            L.check(L.lib().ivx_dev_minmax_f32(L.I16, self.image.raw, c64(self.n), self._range_buf.ptr, self.stream), "minmax")
            self._range_valid = True
        return self._range_buf

    def forget_image_range(self):
        self._range_valid = False

    def mida(self, axis: int, wl, ww, out: DeviceBuffer, status: DeviceBuffer):
        """mida (mips.rs:102-168) of the resident image along `axis` into `out` (int16 image of the projection's shape); the
        volume's range comes from `image_range()`.  `status` (int32, zeroed by the caller) receives IVX_EDOM where the
        reference's NumCast would panic.  Window level / width are truncated like the reference's wrapper does
        (invesalius_rs/__init__.py:91-95: ``_native.mida(image, axis, int(wl), int(ww), out)``)."""
        L.check(L.lib().ivx_dev_mida(L.I16, self.image.raw, c64(self.dz), c64(self.dy), c64(self.dx), int(axis), ctypes.c_float(float(int(wl))),
                                     ctypes.c_float(float(int(ww))), self.image_range().ptr, L.I16, out.ptr, status.ptr, self.stream), "mida")


def c64(v):
    return ctypes.c_int64(int(v))
