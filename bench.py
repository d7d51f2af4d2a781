#!/usr/bin/env python3
"""bench.py -- BASELINE.json configs[1] (default): 512x512x512 synthetic int16 volume,
threshold + 26-neighbour region growing + marching cubes, on N x MI355X (one process per GPU, RCCL through libivx's
ivx_comm_* -- no PyTorch).  `--config watershed` = configs[2] (1024^3 IFT watershed), `--config sharded2048` = configs[3]
(2048^3 split over the ranks, threshold + marching cubes + stitch), `--config mip` = configs[4]
(3-axis MaxIP sweep of 512^3 into a 2048^2 viewport); each prints its own line with `roofline` and `cpu_baseline`.

A "step" is one pass of the hot path over the resident volume:
    1. out_mask = zeros                       (np.zeros of styles.py:3190)
    2. threshold image -> mask (255/0)         (slice_.py:1240-1247)
    3. floodfill_threshold(image, seed, lo, hi, 1, 26-conn, out_mask); mask[out_mask==1] = 254   (styles.py:3200-3214)
    4. marching cubes of the mask at iso 127 (from_binary), whole volume   (surface_process.py:100-186)
Which line answers which BASELINE.json config:
    configs[1] (the metric's)  the default line (512^3; `roofline.per_stage_frac` per stage) + `other_configs.grow_mc_1024` (the same step
                               past the 256 MiB Infinity Cache, volume made in HBM) + `other_configs.region_grow_generic_512` (a click
                               whose thresholds are not the mask's: the candidate pass is paid); N > 1: weak scaling on the line,
                               `strong_scaling` beside it (`--scaling strong`: ONE 512^3 volume split into N slabs)
    configs[2] (1024^3)        `other_configs.watershed_ift_1024` / `watershed_gui_default_1024` (and `_512`); full lines: `--config watershed[_sk]`
    configs[3] (2048^3 / 8)    `other_configs.sharded2048_on_one_gpu`; `--config sharded2048 --gpus 8` on a node
    configs[4] (MIP sweep)     `other_configs.mip_sweep_512`; full line `--config mip`
    configs[0] (Cranium, CPU)  tests/test_gpu_cranium.py (phantom of the sample's geometry); not a bench line
The volume is uploaded once before the timed region (inputs resident in HBM).  N > 1: weak scaling, every rank
owns one 512^3 Z-slab of a (512*N) x 512 x 512 volume; slab boundaries exchange one reached-bit plane per
region-growing round (ncclSend/Recv + a 4-byte all-reduce, one enqueue-only call) and the image's halo slice once.
Launch: under torchrun (RANK / WORLD_SIZE / LOCAL_RANK in the environment) or plainly `python bench.py --gpus N`,
which then starts the N ranks itself; fewer than N visible GPUs is an error, never a silent N=1.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel family of the step, measured live with HIP
events on the stream the kernels run on; `cpu_baseline` is the CPU oracle (a C/numpy restatement of the reference:
"port") timed on the SAME volume on this box's host cores; its outputs double as the full-size parity gate
(`parity`: region voxels, triangle count and a CRC of the whole mask must equal the oracle's).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 20260924
BONE = (226, 3071)  # invesalius/presets.py:37
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)


_VOLUMES = {}


def synth_v512(shape=(512, 512, 512), seed=SEED, z_offset=0, z_total=None):
    """(cached per process: the default run times several configs on the same 512^3 volume)"""
    key = (tuple(shape), seed, z_offset, z_total or shape[0])
    if key not in _VOLUMES:
        if len(_VOLUMES) >= 1 and int(np.prod(shape)) >= 2 ** 29:
            _VOLUMES.clear()
        _VOLUMES[key] = _synth_v512(shape, seed, z_offset, z_total)
        _VOLUMES[key].setflags(write=False)
    return _VOLUMES[key]


def _synth_v512(shape=(512, 512, 512), seed=SEED, z_offset=0, z_total=None):
    """V512 of SURVEY.md 8(d): 6 Gaussian 'bone' blobs (peak 1800) + low-frequency sinusoid + N(0,25) noise,
    offset -1000, clipped to [-1024, 3071].  Built separably, slab by slab, in float32.

    Weak scaling (z_total > shape[0]): the taller volume is V512 repeated along Z -- every rank's slab holds the same six
    blobs at the same places of ITS slab, plus what the blobs of the slabs directly below and above contribute across
    the shared faces (so regions do join their neighbours' and the halo exchange has real work), and its own noise.
    Per-GPU work is therefore the single-GPU work, which is what "weak" means; at one GPU this is V512 itself."""
    dz, dy, dx = shape
    z_total = z_total or dz
    rng = np.random.default_rng(seed)
    cz = rng.uniform(0.15, 0.85, 6)
    cy = rng.uniform(0.15, 0.85, 6)
    cx = rng.uniform(0.15, 0.85, 6)
    sg = rng.uniform(0.12, 0.28, 6)
    zz = (np.arange(dz) / max(dz - 1, 1)).astype(np.float32)  # this slab's own normalised z
    yy = (np.arange(dy) / max(dy - 1, 1)).astype(np.float32)
    xx = (np.arange(dx) / max(dx - 1, 1)).astype(np.float32)
    rank, world = z_offset // dz, max(z_total // dz, 1)
    pitch = dz / max(dz - 1, 1)  # one slab further, in normalised z
    shifts = [0.0] + ([-pitch] if rank > 0 else []) + ([pitch] if rank + 1 < world else [])
    out = np.empty(shape, np.int16)
    nrng = np.random.default_rng(seed + 1 + z_offset)
    step = 32

    def work(z0, z1, noise):
        f = np.zeros((z1 - z0, dy, dx), np.float32)
        for b in range(6):
            gz = sum(np.exp(-((zz[z0:z1] - cz[b] - sh) ** 2) / (2 * sg[b] ** 2)) for sh in shifts)
            gy = np.exp(-((yy - cy[b]) ** 2) / (2 * sg[b] ** 2))
            gx = np.exp(-((xx - cx[b]) ** 2) / (2 * sg[b] ** 2))
            f += 1800.0 * gz[:, None, None] * gy[None, :, None] * gx[None, None, :]
        f += 150.0 * np.sin(6.0 * xx)[None, None, :] * np.cos(5.0 * zz[z0:z1])[:, None, None] * np.cos(4.0 * yy)[None, :, None]
        noise *= 25.0
        f += noise
        np.clip(f - 1000.0, -1024, 3071, out=f)
        out[z0:z1] = f.astype(np.int16)

    # The noise comes from ONE generator in slab order (that fixes the volume's bits); everything else of a 32-slice slab
    # is independent of the other slabs and runs on a few threads (numpy drops the GIL): 1024^3 in ~25 s instead of minutes,
    # bit-identical to the one-thread loop (tests/golden/ws{512,1024}_full.npz are keyed by this volume's CRC-32).
    from concurrent.futures import ThreadPoolExecutor
    nthreads = max(1, min(8, (os.cpu_count() or 2) - 1))
    with ThreadPoolExecutor(nthreads) as ex:
        pending = []
        for z0 in range(0, dz, step):
            z1 = min(dz, z0 + step)
            noise = nrng.standard_normal((z1 - z0, dy, dx), dtype=np.float32)
            pending.append(ex.submit(work, z0, z1, noise))
            del noise
            while len(pending) > nthreads + 1:  # bounded: at most a few slabs of float32 in flight
                pending.pop(0).result()
        for f_ in pending:
            f_.result()
    return out




def src_sha16():
    """fingerprint of the kernel sources + this file: a committed PMC summary is only quoted when it was measured on them"""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(ROOT, "bench.py")]
    for d in (os.path.join(ROOT, "invesalius3_amd", "csrc"), os.path.join(ROOT, "include")):
        files += sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hip", ".h")))
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(config, stage, size=512):
    """HBM bytes per step of `stage` from the rocprofv3 --pmc passes of tools/profile_round.sh, if (and only if) the
    committed summary was measured on exactly these sources (and this size); None otherwise (rocprofv3 cannot run inside
    the bench)."""
    try:
        name = "pmc_traffic.json" if config == "grow_mc" else "pmc_traffic_%s%s.json" % (config, "" if size == 512 else "_%d" % size)
        with open(os.path.join(ROOT, "profiles", name)) as f:
            j = json.load(f)
        if j.get("src_sha16") != src_sha16() or j.get("config", "grow_mc") != config or j.get("size", 512) != size:
            return None
        return j["traffic_bytes_per_step"].get(stage)
    except (OSError, ValueError, KeyError):
        return None


def cpu_grow_mc(img, seed_xyz):
    """CPU oracle on the WHOLE bench volume, with the parallelism the REFERENCE has on each stage: numpy threshold (one
    thread, slice loop), serial C flood fill (the Rust one is serial too), and marching cubes over the reference's
    20+1-slice pieces on a pool of min(pieces, host cores) workers (surface.py:1362-1380 uses multiprocessing.Pool the
    same way; ctypes releases the GIL, so threads do here).  Returns the baseline record and the oracle's outputs."""
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    from scipy.ndimage import generate_binary_structure

    from oracle import oracle as orc

    orc.build()
    dz = img.shape[0]
    t0 = time.perf_counter()
    mask = np.zeros(tuple(s + 1 for s in img.shape), np.uint8)
    orc.set_mask_threshold_volume(mask, img, BONE)
    t1 = time.perf_counter()
    out_mask = np.zeros(img.shape, np.uint8)
    orc.floodfill_threshold(img, [seed_xyz], BONE[0], BONE[1], 1, generate_binary_structure(3, 3), out_mask)
    mask[1:, 1:, 1:][out_mask.astype(bool)] = 254
    t2 = time.perf_counter()
    n_pieces = int(round(dz / 20 + 0.5, 0))
    rois = [slice(i * 20, i * 20 + 21) for i in range(n_pieces) if i * 20 < dz]
    piece = lambda roi: len(orc.create_surface_piece(None, mask, roi, (1.0, 1.0, 1.0), 0, 0, True))
    cores = max(1, min(len(rois), os.cpu_count() or 1))
    with ThreadPoolExecutor(cores) as pool:
        ntri = sum(pool.map(piece, rois))
    t3 = time.perf_counter()
    nvox = img.size
    total = t3 - t0
    rec = {
        "value": round(nvox / total / 1e6, 3), "unit": "Mvoxel/s", "cores": cores, "kind": "port",
        "sample": "the whole bench volume, %d slices (%d voxels): numpy threshold %.2fs + serial C floodfill %.2fs + C marching "
                  "cubes %.2fs on %d threads over %d pieces of 20+1 slices (%d triangles, %.2f Mtri/s)"
                  % (dz, nvox, t1 - t0, t2 - t1, t3 - t2, cores, len(rois), ntri, ntri / max(t3 - t2, 1e-9) / 1e6),
    }
    interior = np.ascontiguousarray(mask[1:, 1:, 1:])
    return rec, {"region_voxels": int(out_mask.sum()), "triangles": int(ntri), "mask_crc32": zlib.crc32(interior)}


def rank_envs(world, idfile):
    """the environments of the N ranks of one launch: ONE nonce for all of them (comm._nonce() hashes it into the id file's
    header, and a rank only accepts an id file that carries its own launch's nonce)"""
    nonce = "%d-%.6f" % (os.getpid(), time.time())
    return [dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), IVX_COMM_FILE=idfile, IVX_COMM_NONCE=nonce,
                 HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")) for r in range(world)]


def launch_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks (fresh processes, one GPU each), relay rank 0's line."""
    import subprocess
    import tempfile

    from invesalius3_amd import _lib as L

    L.require_device()
    have = L.device_count()
    if have < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible -- one GPU per rank is required" % (args.gpus, have))
    idfile = os.path.join(tempfile.mkdtemp(prefix="ivx_bench_"), "comm.id")
    procs = []
    for env in rank_envs(args.gpus, idfile):
        r = int(env["RANK"])
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out = procs[0].communicate()[0].decode()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out)
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit("bench.py: rank exit codes %s" % rcs)


class CStdoutToStderr:
    """RCCL announces itself (version / host / library path) on the C-level stdout, block-buffered, so it would land
    after rank 0's JSON line when the process exits.  While this is active file descriptor 1 IS stderr; leaving it
    flushes the C buffers first.  `stay()` makes the switch permanent (used right before the ranks exit)."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import ctypes
        ctypes.CDLL(None).fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False

    @staticmethod
    def stay():
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        os.dup2(2, 1)


def dry_comm(world):
    """--dry-comm: communicator up, ivx_comm_selftest on every rank, one JSON line from rank 0"""
    from invesalius3_amd import _lib as L
    from invesalius3_amd.comm import RcclComm, init_from_env

    rank = int(os.environ.get("RANK", "0"))
    t0 = time.perf_counter()
    with CStdoutToStderr():
        if world > 1:
            comm = init_from_env()
        else:
            L.set_device(0)
            comm = RcclComm(0, 1, RcclComm.unique_id())
        t1 = time.perf_counter()
        comm.selftest()
        comm.barrier()
        t2 = time.perf_counter()
    if rank == 0:
        print(json.dumps({"dry_comm": "ok", "n_gpus": world, "init_s": round(t1 - t0, 3), "selftest_s": round(t2 - t1, 3),
                          "checked": "ivx_comm_exchange, _exchange_vote, _allreduce (sum/max/min), _allgather, _bcast, _send/_recv"
                                     + (" + RCCL driven directly on the one-rank communicator" if world == 1 else ""),
                          "device": L.device_name()}), flush=True)
    CStdoutToStderr.stay()
    comm.close()


class Ranks:
    """what the timing contract needs from the job: barrier, max over ranks, sums -- over RCCL, or trivially at N = 1"""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self.comm = None
        if self.world > 1:
            from invesalius3_amd.comm import init_from_env
            with CStdoutToStderr():
                self.comm = init_from_env()
                # every ivx_comm_* entry point once on 4 KB buffers, checked against the analytic answer: a broken link or
                # a wrong rank order fails HERE, in the first second, with the name of the collective -- not as a hang in
                # the first region-growing round
                self.comm.selftest()

    def barrier(self):
        from invesalius3_amd import _lib as L
        L.synchronize()
        if self.comm is not None:
            self.comm.barrier()

    def max(self, v: float) -> float:
        return float(self.comm.allreduce_array(np.array([v], np.float64), "max")[0]) if self.comm is not None else v

    def sum(self, v: int) -> int:
        return int(self.comm.allreduce_array(np.array([int(v)], np.int64), "sum")[0]) if self.comm is not None else int(v)


def copy_bandwidth(vol, nvox):
    """achievable streaming bandwidth on this box, measured the same way (HIP events, same stream): device-to-device copy
    of the int16 volume, read + written bytes over the time of the copy"""
    import ctypes

    from invesalius3_amd import _lib as L
    from invesalius3_amd.device import DeviceBuffer
    tmp = DeviceBuffer(nvox * 2)
    for _ in range(2):
        L.check(L.lib().ivx_memcpy_d2d(tmp.ptr, vol.image.raw, ctypes.c_size_t(nvox * 2), vol.stream))
    for _ in range(5):
        with vol.timer.span("copy"):
            L.check(L.lib().ivx_memcpy_d2d(tmp.ptr, vol.image.raw, ctypes.c_size_t(nvox * 2), vol.stream))
    vol.sync()
    copy_ms = float(np.median(vol.timer.collect()["copy"]))
    tmp.close()
    return 4.0 * nvox / (copy_ms * 1e-3) / 1e9


def roofline(kernel, nbytes, ms, traffic, copy_gbs, extra=None):
    achieved = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    r = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
         "measured_copy_gbs": round(copy_gbs, 1) if copy_gbs else None,
         "frac_of_copy": round(achieved / copy_gbs, 4) if copy_gbs else None,
         "algorithmic_bytes": nbytes, "ms": round(ms, 4)}
    if extra:
        r.update(extra)
    return r


# ----------------------------------------------------------------------------------------------------------------
# configs[1]: threshold + 26-neighbour region growing + marching cubes
# ----------------------------------------------------------------------------------------------------------------
def run_grow_mc(args, job):
    import zlib

    from scipy.ndimage import generate_binary_structure

    from invesalius3_amd import _lib as L
    from invesalius3_amd.device import DeviceVolume

    rank, world = job.rank, job.world
    n = args.size or 512
    strong = getattr(args, "scaling", "weak") == "strong"
    hbm = bool(getattr(args, "hbm_synth", False))  # the volume is made in HBM (k_synth) and fetched for the oracle: 1024^3 without minutes of numpy
    vol = None
    if strong:
        # configs[1]'s ONE n^3 volume split into `world` Z-slabs of n / world slices (8 GPUs: 64 slices = 4 flood-tile layers each);
        # one seed, the whole volume's brightest voxel; at one GPU this is the default line, bit for bit
        if n % world:
            raise SystemExit("bench.py --scaling strong: %d slices do not split over %d ranks" % (n, world))
        nz = n // world
        full = synth_v512((n, n, n))
        z, y, x = np.unravel_index(int(np.argmax(full)), full.shape)
        seed = (int(x), int(y), int(z))
        img = full[rank * nz:(rank + 1) * nz]
    elif hbm:
        if world != 1:
            raise SystemExit("bench.py: the HBM-synthesised volume is a one-GPU configuration")
        import ctypes

        from invesalius3_amd.device import c64
        rng = np.random.default_rng(SEED)
        blobs = np.stack([rng.uniform(0.15, 0.85, 6), rng.uniform(0.15, 0.85, 6), rng.uniform(0.15, 0.85, 6),
                          rng.uniform(0.12, 0.28, 6)], axis=1).astype(np.float32)  # (cz, cy, cx, sigma) x 6, as synth_v512 draws them
        vol = DeviceVolume(None, shape=(n, n, n))
        L.check(L.lib().ivx_dev_synth_volume(vol.image.raw, c64(n), c64(n), c64(n), c64(0), c64(n), ctypes.c_uint32(SEED), L.ptr(blobs),
                                             vol.stream), "synth_volume")
        vol._image_touched()
        vol.sync()
        img = vol.image.download((n, n, n), np.int16)
        z, y, x = np.unravel_index(int(np.argmax(img)), img.shape)
        seed = (int(x), int(y), int(z))
    else:
        img = synth_v512((n, n, n), z_offset=rank * n, z_total=world * n)
        z, y, x = np.unravel_index(int(np.argmax(img)), img.shape)
        seed = (int(x), int(y), int(z) + rank * n)  # global (x, y, z): every rank seeds the brightest voxel of its slab
    shape = tuple(img.shape)
    strct = generate_binary_structure(3, 3)
    force_slab = os.environ.get("IVX_FORCE_SLAB") == "1"  # exercise the sharded path at world 1
    t_up = time.perf_counter()
    if vol is not None:
        pass
    elif world > 1 or force_slab:
        from invesalius3_amd.parallel import SlabVolume

        vol = SlabVolume(img, rank, world, comm=job.comm, device=job.local_rank)
    else:
        vol = DeviceVolume(img)
    vol.sync()
    upload_ms = (time.perf_counter() - t_up) * 1e3
    nvox = img.size

    # Marching cubes' count, scan and triangle list depend on WHICH voxels are >= 127 only, and `mask[reached] = 254`
    # does not change that: with IVX_PREFETCH=1 the timed steps queue them on a second, low-priority stream right after
    # the threshold pass, held back until the region growing's busy rounds are over; the emit waits for the mask's final
    # bytes (DeviceVolume.surface_prefetch).  The default is strictly one stage after the other: the per-stage roofline
    # figures then describe kernels that had the GPU to themselves, and every GPU count runs the same schedule.
    overlap = os.environ.get("IVX_PREFETCH", "") == "1"

    def step(prefetch=False):
        with vol.timer.span("zero_out_mask"):
            vol.zero_out_mask()
        with vol.timer.span("threshold"):
            vol.threshold(BONE[0], BONE[1], preserve=False)
        if prefetch:
            vol.surface_prefetch(from_binary=True)
        with vol.timer.span("region_grow"):
            rounds = vol.region_grow([seed], BONE[0], BONE[1], strct, fill=1, select_value=254)
        ntri = vol.marching_cubes(from_binary=True)
        return rounds, ntri

    def barrier():
        vol.sync()
        job.barrier()

    # A recorded HIP event is a barrier packet in the stream (~4 us of idle GPU each; ten per step with every stage
    # bracketed).  The timed steps bracket only the dominant stage -- the roofline figure is measured live in the timed
    # region -- and a few extra steps AFTER the timed region, with every stage bracketed, give the per-stage table.
    for _ in range(args.warmup):
        step(overlap)
    barrier()
    vol.timer.collect()
    vol.timer.only = {"region_grow"}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rounds, ntri = step(overlap)
    barrier()
    dt = time.perf_counter() - t0
    spans_timed = vol.timer.collect()
    vol.timer.only = None
    step()  # the sequential path's own one-time work (its triangle-list workspace) stays out of the table
    barrier()
    vol.timer.collect()
    n_extra = max(1, min(args.steps, 5))
    for _ in range(n_extra):
        step()
    barrier()
    spans = vol.timer.collect()
    # N > 1: device time of the collectives per step (HIP events around every ivx_comm_* call of the bracketed steps), so that
    # the 1 -> 8 curve arrives with its communication share attached
    comm_ms = {k[5:]: {"ms_per_step": round(float(np.sum(v)) / n_extra, 4), "calls_per_step": round(len(v) / n_extra, 2)}
               for k, v in spans.items() if k.startswith("comm_")}
    spans["region_grow"] = spans_timed.get("region_grow", spans.get("region_grow", []))
    reached = vol.reached_count()
    copy_gbs = copy_bandwidth(vol, nvox) if rank == 0 else None
    # one step the way a caller without a resident volume pays for it: host -> HBM, the step, mask + triangles -> host
    e2e_ms = e2e_first_ms = None
    e2e = world == 1 and not force_slab and not hbm
    mask_crc = ntri_dl = None
    if e2e:
        for k in range(2):  # the first pass also pays the process's first pageable copies (page locking set-up, lane buffers)
            t = time.perf_counter()
            vol.image.upload(img)
            step()
            mask_host = vol.download_mask()  # (fresh np.empty arrays every pass, like a caller's)
            tris_host = vol.marching_cubes(from_binary=True, download=True)
            e2e_ms = (time.perf_counter() - t) * 1e3
            if k == 0:
                e2e_first_ms = e2e_ms
            mask_crc, ntri_dl = zlib.crc32(mask_host), len(tris_host)
            del tris_host, mask_host
    # ... and the same with the caller's arrays in page-locked host memory (invesalius3_amd._lib.pinned_empty): the DMA
    # engines then reach them directly instead of through the runtime's bounce buffers
    e2e_pinned_ms = None
    if e2e:
        p_img = L.pinned_empty(img.shape, np.int16)
        p_img[:] = img
        p_mask = L.pinned_empty(img.shape, np.uint8)
        p_tris = L.pinned_empty((int(ntri * 1.05) + 16, 3, 3), np.float32)
        for _ in range(2):  # (the first pass touches the fresh pages)
            t = time.perf_counter()
            vol.image.upload(p_img)
            step()
            m2 = vol.download_mask(out=p_mask)
            t2 = vol.marching_cubes(from_binary=True, download=True, out=p_tris)
            e2e_pinned_ms = (time.perf_counter() - t) * 1e3
        if zlib.crc32(m2) != mask_crc or len(t2) != ntri_dl:
            raise SystemExit("bench.py: the pinned end-to-end pass differs from the pageable one")
        del p_img, p_mask, p_tris, m2, t2
    dt = job.max(dt)
    ntri_all, reached_all = job.sum(ntri), job.sum(reached)
    if world == 1 and not force_slab and not e2e:  # (the HBM-made volume: the parity gate's mask CRC without the end-to-end passes)
        mask_crc, ntri_dl = zlib.crc32(vol.download_mask()), ntri
    if rank != 0:
        return None
    ms_per_step = dt / args.steps * 1e3
    stage_ms = {k: float(np.mean(v)) for k, v in spans.items()}
    mc_ms = stage_ms.get("mc_count", 0.0) + stage_ms.get("mc_emit", 0.0)
    stage_bytes = {"threshold": 3.0 * nvox, "region_grow": 3.0 * nvox, "marching_cubes": 1.0 * nvox + 36.0 * ntri}
    stage_time = {"threshold": stage_ms.get("threshold", 0.0), "region_grow": stage_ms.get("region_grow", 0.0),
                  "marching_cubes": mc_ms}
    dom = max(stage_time, key=lambda k: stage_time[k])
    traffic = pmc_traffic("grow_mc", dom) if n == 512 and world == 1 and not hbm else None
    res = {
        "metric": "Mvoxel/s segmentation + Mtriangles/s marching-cubes, 512^3 int16, 1/2/4/8 GPU",
        "value": round(world * nvox / (dt / args.steps) / 1e6, 2),
        "unit": "Mvoxel/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "i16", "data": "synthetic (made in HBM by k_synth)" if hbm else "synthetic",
        "config": {"workload": "configs[1]: %dx%dx%d int16 per GPU, threshold(226..3071) + 26-neighbour region-grow + marching-cubes(mask@127)" % shape
                               + (" -- ONE %d^3 volume split over the GPUs (strong scaling; no multi-GPU box has run this yet: the curve is "
                                  "the driver's to measure)" % n if strong else ""),
                   "global_voxels": world * nvox, "parallelism": "z-slab x%d" % world,
                   "collectives": "RCCL via libivx ivx_comm_* (no PyTorch)" if world > 1 else "none",
                   "overlap": "marching-cubes count+scan+list on a second stream under region growing" if overlap else "none"},
        "mtriangles_per_s": round(ntri / (mc_ms * 1e-3) / 1e6, 2) if mc_ms > 0 else None,
        "triangles": ntri_all, "region_voxels": reached_all, "region_grow_rounds": rounds,
        "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
        "comm_ms_per_step": comm_ms or None,
        "region_grow_ms_min_med_max": [round(float(f(spans["region_grow"])), 4) for f in (np.min, np.median, np.max)]
        if len(spans.get("region_grow", [])) else None,
        "timed_region_s": round(dt, 6),
        "stage_ms_source": "region_grow: HIP events inside the timed steps; other stages: HIP events in up to 5 extra steps "
                           "after the timed region, one stage after the other (every recorded event idles the stream for ~4 us)",
        "stage_mvoxel_per_s": {k: round(nvox / (v * 1e-3) / 1e6, 1) for k, v in stage_time.items() if v > 0},
        "end_to_end_ms": round(e2e_ms, 2) if e2e_ms else None,
        "end_to_end_first_call_ms": round(e2e_first_ms, 2) if e2e_first_ms else None,
        "end_to_end_pinned_ms": round(e2e_pinned_ms, 2) if e2e_pinned_ms else None,
        "end_to_end_note": "host int16 volume -> HBM, one step, dense uint8 mask and float32 triangle soup back to the host: "
                           "`end_to_end_ms` with pageable numpy arrays (results into fresh np.empty arrays; second pass -- "
                           "`end_to_end_first_call_ms` is the first, which also sets up the process's page-locking and lane buffers), "
                           "`end_to_end_pinned_ms` with the caller's three arrays in page-locked memory (_lib.pinned_empty); first upload "
                           "at start-up took %.1f ms" % upload_ms,
        "roofline": roofline(dom, stage_bytes[dom], stage_time[dom], traffic, copy_gbs,
                             {"per_stage_frac": {k: round(stage_bytes[k] / (stage_time[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                                 for k in stage_time if stage_time[k] > 0}}),
        "device": L.device_name(),
    }
    if args.cpu and world == 1 and not force_slab:
        rec, orc_out = cpu_grow_mc(np.ascontiguousarray(img), seed)
        res["cpu_baseline"] = rec
        got = {"region_voxels": reached, "triangles": ntri, "mask_crc32": mask_crc}
        ok = got == orc_out and ntri_dl == ntri
        res["parity"] = {"ok": bool(ok), "checked": "region voxels, triangle count, CRC-32 of the whole uint8 mask vs the CPU oracle "
                         "on the same volume (oracle pinned upstream for threshold / flood fill; marching cubes: the reference's vtkContourFilter "
                         "delegates to vtkSynchronizedTemplates3D for image data -- own templates, point-merged output -- so only the vertex "
                         "set and closedness are pinned, the case table is the builder's own)",
                         "gpu": got, "oracle": orc_out}
        if not ok:
            print(json.dumps(res), flush=True)
            raise SystemExit("bench.py: GPU result differs from the CPU oracle: %s vs %s" % (got, orc_out))
    else:
        res["cpu_baseline"] = None
    return res


def run_grow_generic(args, job):
    """configs[1]'s region growing when the click's thresholds are NOT the mask's: the headline step
    floods with the thresholds the mask was made with, so the threshold pass's bit plane IS the candidate plane
    (DeviceVolume._candidate_plane); any other click pays its own pass over the volume for the candidates
    (k_flood_candidates16).  Step = out_mask zeros + threshold(226..3071) + floodfill_threshold(image, seed, 300, 3071) + mask[out] = 254
    on the 512^3 volume; the flood stage is timed with HIP events inside the timed steps."""
    import zlib

    from scipy.ndimage import generate_binary_structure

    from invesalius3_amd import _lib as L
    from invesalius3_amd.device import DeviceVolume

    n = args.size or 512
    img = synth_v512((n, n, n))
    z, y, x = np.unravel_index(int(np.argmax(img)), img.shape)
    seed = (int(x), int(y), int(z))
    strct = generate_binary_structure(3, 3)
    T0, T1 = 300, BONE[1]
    vol = DeviceVolume(img, device=job.local_rank)
    nvox = img.size

    def step():
        vol.zero_out_mask()
        vol.threshold(BONE[0], BONE[1], preserve=False)
        with vol.timer.span("region_grow"):
            return vol.region_grow([seed], T0, T1, strct, fill=1, select_value=254)

    for _ in range(args.warmup):
        step()
    vol.sync()
    vol.timer.collect()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rounds = step()
    vol.sync()
    dt = time.perf_counter() - t0
    grow_ms = float(np.mean(vol.timer.collect()["region_grow"]))
    reached = vol.reached_count()
    out = vol.download_out_mask()
    vol.close()
    res = {"metric": "Mvoxel/s segmentation + Mtriangles/s marching-cubes, 512^3 int16, 1/2/4/8 GPU",
           "value": round(nvox / (grow_ms * 1e-3) / 1e6, 2), "unit": "Mvoxel/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i16",
           "data": "synthetic",
           "config": {"workload": "configs[1]'s region growing with thresholds that differ from the mask's: %d^3 int16, mask = threshold(226..3071), "
                                  "26-neighbour flood of image in [%d, %d] from the brightest voxel (candidate pass k_flood_candidates16 + flood + "
                                  "apply); `value` and `frac` are the flood stage's, `ms` the whole step's" % (n, T0, T1)},
           "stage_ms": {"region_grow": round(grow_ms, 4)}, "region_voxels": reached, "region_grow_rounds": rounds,
           "roofline": roofline("region_grow (generic thresholds)", 3.0 * nvox, grow_ms, None, None), "device": L.device_name()}
    if args.cpu:
        from oracle import oracle as orc
        orc.build()
        ref = np.zeros(img.shape, np.uint8)
        t = time.perf_counter()
        orc.floodfill_threshold(img, [seed], T0, T1, 1, strct, ref)
        ts = time.perf_counter() - t
        ok = zlib.crc32(out) == zlib.crc32(ref) and reached == int(ref.sum())
        res["cpu_baseline"] = {"value": round(nvox / ts / 1e6, 2), "unit": "Mvoxel/s", "cores": 1, "kind": "port",
                               "sample": "the serial flood of floodfill.rs:96-166 restated in C on the whole volume, %.2f s" % ts}
        res["parity"] = {"ok": bool(ok), "checked": "CRC-32 of the whole out_mask and the region's voxel count vs the CPU oracle (pinned by the "
                                                    "reference's golden vectors)"}
        if not ok:
            print(json.dumps(res), flush=True)
            raise SystemExit("bench.py: generic region growing differs from the CPU oracle")
    else:
        res["cpu_baseline"] = None
    return res


# ----------------------------------------------------------------------------------------------------------------
# configs[2]: IFT watershed at 1024^3 (replicas only: SURVEY.md 8e)
# ----------------------------------------------------------------------------------------------------------------
def ws_markers(img):
    """label 1 = 5^3 cube at the brightest voxel, label 2 = 5^3 cubes at the 8 corners; int8 like
    watershed_process.py:57"""
    mk = np.zeros(img.shape, np.int8)
    d, h, w = img.shape
    z, y, x = np.unravel_index(int(np.argmax(img)), img.shape)
    z, y, x = min(max(z, 2), d - 3), min(max(y, 2), h - 3), min(max(x, 2), w - 3)
    for cz in (0, d - 5):
        for cy in (0, h - 5):
            for cx in (0, w - 5):
                mk[cz:cz + 5, cy:cy + 5, cx:cx + 5] = 2
    mk[z - 2:z + 3, y - 2:y + 3, x - 2:x + 3] = 1
    return mk


def full_volume_fixture(n, img):
    """tests/golden/ws{n}_full.npz (made by tests/golden/make_golden_ws_full.py from live scipy and the serial oracles on THIS
    volume), or None when there is none for this size or the synthetic volume of this box differs from the one it was made on"""
    import zlib
    path = os.path.join(ROOT, "tests", "golden", "ws%d_full.npz" % n)
    if not os.path.exists(path):
        return None
    z = np.load(path)
    if tuple(z["shape"]) != tuple(img.shape) or int(z["image_crc32"]) != zlib.crc32(img):
        return None
    return z


def ift_parity(args, img, mk, strct, lab, n):
    """`differs_from_reference` for the IFT flood, on the volume and markers the bench TIMED (`lab` = its labels).

    The reference is live scipy.ndimage.watershed_ift (watershed_process.py:57).  512^3 and 1024^3: scipy's labels on exactly
    this volume are on file (tests/golden/ws{n}_full.npz: CRC-32 of scipy's labels, CRC-32 of the defect-free statement's, and
    the voxels where the two differ) -- the timed labels are compared with both, whole volume, in a second.  Other sizes, a box
    whose synthetic volume differs, or --live-reference: scipy and the serial oracle run here on the whole volume (minutes).
    `cpu_baseline` = live scipy timed on a bounded slab of the same cost image (that slab's flood is not the timed one and
    is not used for parity)."""
    import zlib

    from scipy import ndimage

    from oracle import oracle as orc
    orc.build()
    nvox = img.size
    cost = (img - img.min()).astype(np.uint16)
    # (a) the reference's speed: live scipy, one core like the reference's worker process, bounded sample
    sl = min(n, max(16, int(2.5e7 // (n * n))))
    sub, smk = np.ascontiguousarray(cost[:sl]), np.ascontiguousarray(mk[:sl])
    t = time.perf_counter()
    ndimage.watershed_ift(sub, smk, strct)
    ts = time.perf_counter() - t
    out = {"cpu_baseline": {"value": round(sub.size / ts / 1e6, 3), "unit": "Mvoxel/s", "cores": 1, "kind": "reference",
                            "sample": "live scipy.ndimage.watershed_ift (the call of watershed_process.py:57) on the first %d slices of "
                                      "the timed cost image with the timed markers (%d voxels), %.2f s" % (sl, sub.size, ts)}}
    fx = None if args.live_reference else full_volume_fixture(n, img)
    lab = np.ascontiguousarray(lab, dtype=np.uint8)
    if fx is not None:
        at = np.cumsum(fx["differs_at"].astype(np.int64))
        n_clean = 0 if zlib.crc32(lab) == int(fx["clean_crc32"]) else None
        if n_clean == 0:
            ref = lab.copy().reshape(-1)
            ref[at] = 3 - ref[at]  # labels are 1 / 2: scipy's volume is the defect-free one flipped at these places
            if zlib.crc32(ref) != int(fx["scipy_crc32"]):
                raise SystemExit("bench.py: tests/golden/ws%d_full.npz is inconsistent (rebuilt reference labels miss their CRC)" % n)
            n_ref = len(at)
            how = ("tests/golden/ws%d_full.npz: live scipy (%s) and the defect-free serial statement on this very volume; the timed "
                   "labels' CRC-32 equals the statement's, and flipping them at the %d recorded voxels gives scipy's CRC-32"
                   % (n, ", ".join(str(v) for v in fx["versions"]), n_ref))
            ev = None
    if (fx is None or n_clean is None) and getattr(args, "bounded", False):
        # inside the default run's `other_configs` the whole-volume serial floods (minutes) are not affordable: say so
        out["differs_from_reference"] = None
        out["parity"] = {"ok": None, "reference": "live scipy.ndimage.watershed_ift (the reference's call)", "differs_from_reference": None,
                         "compared": "nothing: tests/golden/ws%d_full.npz does not describe this box's synthetic volume%s; run "
                                     "`bench.py --config watershed --size %d --live-reference`"
                                     % (n, "" if fx is None else " or the flood differs from the defect-free statement", n)}
        return out
    if fx is None or n_clean is None:
        # live, whole volume: the reference itself, the defect-free statement and the defect's event counts
        sci = ndimage.watershed_ift(cost, mk, strct)
        clean = orc.watershed_ift_clean(cost, mk, strct)
        _, ev = orc.watershed_ift_events(cost, mk, strct)
        n_ref, n_clean = int((lab != sci.astype(np.uint8)).sum()), int((lab != clean.astype(np.uint8)).sum())
        how = "live scipy.ndimage.watershed_ift and oracle/ivx_oracle_wsz.c run on the whole timed volume in this process"
    out["differs_from_reference"] = n_ref
    out["parity"] = {"ok": n_ref == 0, "reference": "live scipy.ndimage.watershed_ift (the reference's call)",
                     "differs_from_reference": n_ref, "compared_voxels": int(nvox), "compared": "the whole timed volume, the timed flood's own labels",
                     "how": how, "equals_defect_free_statement": n_clean == 0, "mismatch_vs_defect_free_oracle": n_clean,
                     "scipy_defect_events": None if ev is None else {"requeued_unlinked": ev[0], "popped_late": ev[1], "popped_twice": ev[2],
                                                                     "never_popped": ev[3]},
                     "note": "bit-exact integer masks are the contract: %d of %d voxels (%.4f %%) differ from the reference.  The GPU flood "
                             "equals the defect-free statement of NI_WatershedIFT bit for bit; live scipy leaves that statement only downstream "
                             "of its linked-list defect (ni_measure.c: `if (p->next || p->prev)`); profiles/r04_ift_defect_confinement.json "
                             "has where and why a parallel replay is not possible" % (n_ref, nvox, 100.0 * n_ref / nvox)}
    if n_clean:
        print(json.dumps(dict(out, error="watershed differs from the defect-free oracle")), flush=True)
        raise SystemExit("bench.py: watershed differs from the defect-free oracle")
    return out


def run_watershed(args, job):
    import ctypes

    from scipy import ndimage
    from scipy.ndimage import generate_binary_structure

    from invesalius3_amd import _lib as L
    from invesalius3_amd import watershed_process as wp
    from invesalius3_amd.device import DeviceBuffer, Timer, c64

    L.require_device()
    L.set_device(job.local_rank)
    n = args.size or 1024
    shape = (n, n, n)
    nvox = n ** 3
    img = synth_v512(shape, seed=SEED + job.rank)
    mk = ws_markers(img)
    strct = generate_binary_structure(3, 1)  # con_3d = 6 is the reference's default (styles.py:1632)
    s3 = np.ascontiguousarray(strct, dtype=np.uint8)
    lib = L.lib()
    st = ctypes.c_void_p()
    L.check(lib.ivx_stream_create(ctypes.byref(st)))
    timer = Timer(st)
    d_img, d_mk = DeviceBuffer(nvox * 2), DeviceBuffer(nvox)
    d_cost, d_lab, d_mask = DeviceBuffer(nvox * 2), DeviceBuffer(nvox), DeviceBuffer(nvox)
    d_mm = DeviceBuffer(64)
    d_img.upload(img)
    d_mk.upload(mk)
    d_mask.zero(st)
    stats = (ctypes.c_int64 * 16)()

    def step():
        # watershed_process.py:55-59 + styles.py:2147-2152: (image - image.min()).astype(uint16) -> watershed_ift -> merge
        L.check(lib.ivx_dev_minmax_f32(L.I16, d_img.ptr, c64(nvox), d_mm.ptr, st))
        L.check(lib.ivx_stream_synchronize(st))
        imin = int(d_mm.download((2,), np.float32)[0])
        with timer.span("cost_image"):
            L.check(lib.ivx_dev_shift_min_u16(d_img.ptr, c64(nvox), imin, d_cost.ptr, st))
        with timer.span("flood"):
            L.check(lib.ivx_dev_watershed_ift(d_cost.ptr, L.I8, d_mk.ptr, c64(n), c64(n), c64(n), L.ptr(s3), None, d_lab.ptr,
                                              None, stats, st), "watershed_ift")
        with timer.span("merge"):
            L.check(lib.ivx_dev_watershed_merge(d_mask.ptr, d_lab.ptr, c64(nvox), 1, st))

    def barrier():
        L.check(lib.ivx_stream_synchronize(st))
        job.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    timer.collect()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = job.max(time.perf_counter() - t0)
    spans = {k: float(np.mean(v)) for k, v in timer.collect().items()}
    lab = d_lab.download(shape, np.uint8)
    obj = job.sum(int((lab == 1).sum()))
    if job.rank != 0:
        return
    names = ("rounds", "tile_visits", "levels", "time_stamps", "markers", "entries", "tiles", "tile_sweeps", "us_costs",
             "us_zones", "us_bucket", "us_levels", "us_labels", "cost_levels", "cost_level_rounds", "cost_level_voxels")
    flood_ms = spans.get("flood", 0.0)
    res = {
        "metric": "Mvoxel/s segmentation + Mtriangles/s marching-cubes, 512^3 int16, 1/2/4/8 GPU",
        "value": round(job.world * nvox / (dt / args.steps) / 1e6, 2), "unit": "Mvoxel/s",
        "n_gpus": job.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
        "config": {"workload": "configs[2]: %dx%dx%d int16, watershed segmentation, IFT branch of do_watershed (min-shift cost image, "
                               "6-neighbour marker flood, merge), markers: 5^3 cube at the maximum (1) + 8 corner cubes (2)" % shape,
                   "parallelism": "replicas only (global priority order: SURVEY.md 8e)" if job.world > 1 else "single GPU"},
        "stage_ms": {k: round(v, 3) for k, v in spans.items()},
        "flood": {k: int(v) for k, v in zip(names, stats)},
        "object_voxels": obj,
        "roofline": roofline("watershed flood (k_ws_*)", 7.0 * nvox, flood_ms, pmc_traffic("watershed", "flood", n), None,
                             {"note": "7 B/voxel = cost 2 + markers 2 read, labels 2 + mask 1 written (SURVEY.md 8d); the flood is a "
                                      "multi-pass algorithm (relaxation rounds + zones + level chain), so the fraction is small by design"}),
        "device": L.device_name(),
    }
    if args.cpu:
        res.update(ift_parity(args, img, mk, strct, lab, n))
    else:
        res["cpu_baseline"] = None
    return res


# ----------------------------------------------------------------------------------------------------------------
# configs[4]: MIP raycasting, 512^3 volume to a 2048^2 viewport, 3-axis sweep
# ----------------------------------------------------------------------------------------------------------------

def run_watershed_sk(args, job):
    """configs[2] with the GUI's DEFAULT settings (styles.py:1628-1634: algorithm "Watershed", 6 neighbours, gradient size 3,
    use_ww_wl): window/level LUT -> 3x3x3 morphological gradient -> skimage.segmentation.watershed's marker flood
    (csrc/k_wssk.hip) -> merge (watershed_process.py:33-39, styles.py:2147-2152).  --ws-raw takes the other branch
    (watershed_process.py:47-52: image - image.min(), the harder input: a noise-dominated gradient)."""
    import ctypes

    from scipy.ndimage import generate_binary_structure

    from invesalius3_amd import _lib as L
    from invesalius3_amd import watershed_process as wp
    from invesalius3_amd.device import DeviceBuffer, Timer, c64

    L.require_device()
    L.set_device(job.local_rank)
    n = args.size or 1024
    shape = (n, n, n)
    nvox = n ** 3
    img = synth_v512(shape, seed=SEED + job.rank)
    mk = ws_markers(img).astype(np.int16)  # watershed_process.py:52 casts to int16
    strct = generate_binary_structure(3, 1)
    s3 = np.ascontiguousarray(strct, dtype=np.uint8)
    lib = L.lib()
    st = ctypes.c_void_p()
    L.check(lib.ivx_stream_create(ctypes.byref(st)))
    timer = Timer(st)
    d_img, d_mk = DeviceBuffer(nvox * 2), DeviceBuffer(nvox * 2)
    d_cost, d_grad, d_lab, d_mask = DeviceBuffer(nvox * 2), DeviceBuffer(nvox * 2), DeviceBuffer(nvox), DeviceBuffer(nvox)
    d_mm = DeviceBuffer(64)
    d_img.upload(img)
    d_mk.upload(mk)
    d_mask.zero(st)
    stats = (ctypes.c_int64 * 16)()
    gsz = (ctypes.c_int * 3)(3, 3, 3)

    use_ww_wl = not args.ws_raw
    WL, WW = 300, 400  # a bone window on the synthetic volume (values -1024 .. 3071)

    def step():
        if use_ww_wl:
            with timer.span("cost_image"):
                L.check(lib.ivx_dev_lut_u16(d_img.ptr, c64(nvox), ctypes.c_double(WW), ctypes.c_double(WL), 0, d_cost.ptr, st), "lut")
                L.check(lib.ivx_dev_morph_gradient_u16(d_cost.ptr, c64(n), c64(n), c64(n), gsz, d_grad.ptr, st), "gradient")
        else:
            L.check(lib.ivx_dev_minmax_f32(L.I16, d_img.ptr, c64(nvox), d_mm.ptr, st))
            L.check(lib.ivx_stream_synchronize(st))
            imin = int(d_mm.download((2,), np.float32)[0])
            with timer.span("cost_image"):
                L.check(lib.ivx_dev_shift_min_u16(d_img.ptr, c64(nvox), imin, d_cost.ptr, st))
                L.check(lib.ivx_dev_morph_gradient_u16(d_cost.ptr, c64(n), c64(n), c64(n), gsz, d_grad.ptr, st), "gradient")
        with timer.span("flood"):
            L.check(lib.ivx_dev_watershed_sk(d_grad.ptr, L.I16, d_mk.ptr, c64(n), c64(n), c64(n), L.ptr(s3), None, None, d_lab.ptr,
                                             None, stats, st), "watershed_sk")
        with timer.span("merge"):
            L.check(lib.ivx_dev_watershed_merge(d_mask.ptr, d_lab.ptr, c64(nvox), 1, st))

    def barrier():
        L.check(lib.ivx_stream_synchronize(st))
        job.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    timer.collect()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = job.max(time.perf_counter() - t0)
    spans = {k: float(np.mean(v)) for k, v in timer.collect().items()}
    lab = d_lab.download(shape, np.uint8)
    obj = job.sum(int((lab == 1).sum()))
    if job.rank != 0:
        return
    names = ("rounds", "tile_visits", "levels", "generations", "markers", "generation0", "tied_markers_of_different_labels",
             "frontier_launches", "us_costs", "us_generation0", "us_levels", "us_labels", "basin_rounds", "generation_steps", "small_level_runs", "tile_rounds")
    flood_ms = spans.get("flood", 0.0)
    res = {
        "metric": "Mvoxel/s segmentation + Mtriangles/s marching-cubes, 512^3 int16, 1/2/4/8 GPU",
        "value": round(job.world * nvox / (dt / args.steps) / 1e6, 2), "unit": "Mvoxel/s",
        "n_gpus": job.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
        "config": {"workload": "configs[2]: %dx%dx%d int16, watershed segmentation, 'Watershed' branch of do_watershed with %s, 3x3x3 "
                               "morphological gradient, scikit-image's 6-neighbour marker flood, merge; markers: 5^3 cube at the "
                               "maximum (1) + 8 corner cubes (2)"
                               % (shape + ("the GUI's default settings (window/level LUT, ww 400 wl 300)" if use_ww_wl
                                           else "image - image.min() (--ws-raw)",)),
                   "parallelism": "replicas only (global priority order: SURVEY.md 8e)" if job.world > 1 else "single GPU"},
        "stage_ms": {k: round(v, 3) for k, v in spans.items()},
        "flood": {k: int(v) for k, v in zip(names, stats) if k != "_"},
        "object_voxels": obj,
        "roofline": roofline("watershed flood (k_ws_relax + k_sk_*)", 7.0 * nvox, flood_ms,
                             pmc_traffic("watershed_sk", "flood", n) if use_ww_wl else None, None,
                             {"note": "7 B/voxel = image 2 + markers 2 read, labels 2 + mask 1 written (SURVEY.md 8d; + 4 B/voxel for "
                                      "the gradient pass, timed apart); the flood is a level-ordered breadth-first search whose serial "
                                      "depth (generations) bounds it, not the bytes"}),
        "device": L.device_name(),
    }
    if args.cpu:
        # the serial heap flood (oracle/ivx_oracle_wssk.c, pinned move for move to scikit-image's compiled kernel) on a bounded
        # sample of the same volume: its first slices with the same marker rule, one core like the reference's worker process
        from oracle import oracle as orc
        orc.build()
        sl = min(n, max(8, int(1.6e7 // (n * n))))
        sub = np.ascontiguousarray(img[:sl])
        smk = ws_markers(sub).astype(np.int16)
        grad = wp.cost_image(sub, use_ww_wl, WL, WW, (3, 3, 3))
        t = time.perf_counter()
        heap = orc.watershed_sk(grad, smk, strct, 0)
        ts = time.perf_counter() - t
        raster = orc.watershed_sk(grad, smk, strct, 1)
        got, gst = wp.watershed(grad, smk, strct, want_stats=True)
        res["cpu_baseline"] = {"value": round(sub.size / ts / 1e6, 3), "unit": "Mvoxel/s", "cores": 1, "kind": "port",
                               "sample": "the (value, age) binary-heap flood of skimage.segmentation.watershed restated in C and pinned to "
                                         "scikit-image 0.18.3's compiled kernel (tests/golden/watershed_sk.npz), on the first %d slices "
                                         "(%d voxels), %.2f s" % (sl, sub.size, ts)}
        # the reference is scikit-image's heap flood (pinned move for move by `heap`): `ok` is the comparison with IT
        n_ref = int((got != heap).sum())
        sample = {"sample_voxels": int(sub.size), "sample": "first %d slices, markers re-derived on the sample (a different flood from the timed one)" % sl,
                  "equals_raster_tie_statement": bool(np.array_equal(got, raster)),
                  "mismatch_vs_serial_flood_raster_marker_ties": int((got != raster).sum()),
                  "mismatch_vs_serial_flood_heap_marker_ties": n_ref,
                  "tied_markers_of_different_labels": gst["tied_markers_of_different_labels"]}
        # ... and the TIMED flood, whole volume, against the serial floods run on this very volume when their CRCs are on file
        # (tests/golden/ws{n}_full.npz, GUI-default settings only): cost image, raster-tie statement, heap-ordered reference
        import zlib
        fx = full_volume_fixture(n, img) if use_ww_wl else None
        whole = None
        if fx is not None:
            grad_ok = zlib.crc32(d_grad.download(shape, np.uint16)) == int(fx["grad_crc32"])
            lab_ok = zlib.crc32(np.ascontiguousarray(lab)) == int(fx["sk_raster_crc32"])
            whole = {"compared_voxels": int(nvox), "how": "CRC-32 of the timed cost image and labels vs tests/golden/ws%d_full.npz (numpy LUT + "
                     "scipy morphological_gradient + oracle/ivx_oracle_wssk.c on this very volume)" % n,
                     "cost_image_equals_numpy_scipy": bool(grad_ok), "equals_raster_tie_statement": bool(lab_ok),
                     "differs_from_reference": int(fx["sk_differs"]) if lab_ok else None}
            if lab_ok and int(fx["sk_differs"]) == 0 and zlib.crc32(np.ascontiguousarray(lab)) != int(fx["sk_heap_crc32"]):
                raise SystemExit("bench.py: tests/golden/ws%d_full.npz is inconsistent" % n)
        res["differs_from_reference"] = whole["differs_from_reference"] if whole is not None else None
        res["parity"] = {"ok": (whole["differs_from_reference"] == 0 and whole["cost_image_equals_numpy_scipy"]) if whole is not None else n_ref == 0,
                         "reference": "scikit-image's (value, age) heap flood, heap-ordered marker ties (C restatement pinned to the "
                                      "compiled 0.18.3 kernel)",
                         "differs_from_reference": res["differs_from_reference"],
                         "compared": "the whole timed volume" if whole is not None else "a sample only (no full-volume record for this size / these settings)",
                         "whole_volume": whole, "sample": sample, "equals_raster_tie_statement": sample["equals_raster_tie_statement"] and (whole is None or whole["equals_raster_tie_statement"]),
                         "note": "the GPU flood equals the serial flood bit for bit when equal-valued marker voxels are taken in raster "
                                 "order; scikit-image's heap takes them in an order that depends on its array layout, which matters "
                                 "only where tied markers of different labels compete"}
        if not res["parity"]["equals_raster_tie_statement"]:
            print(json.dumps(res), flush=True)
            raise SystemExit("bench.py: watershed_sk differs from the serial flood")
    else:
        res["cpu_baseline"] = None
    return res

def run_mip(args, job):
    import ctypes

    from invesalius3_amd import _lib as L
    from invesalius3_amd.device import DeviceBuffer, DeviceVolume, c64

    n = args.size or 512
    shape = (n, n, n)
    nvox = n ** 3
    img = synth_v512(shape, seed=SEED + job.rank)
    L.require_device()
    vol = DeviceVolume(img, device=job.local_rank)
    lib = L.lib()
    f = 2048 // n if n <= 2048 and 2048 % n == 0 else 1
    proj = DeviceBuffer(n * n * 2 + 64)
    view = {(k, a): DeviceBuffer(n * f * n * f * 2) for k in ("maxip", "mida") for a in range(3)}
    view[("contour", 0)] = DeviceBuffer(n * f * n * f * 2)
    status = DeviceBuffer(64)
    status.zero(vol.stream)
    WL, WW = 300.0, 300.0  # get_image_slice hands the window LEVEL in for level and width alike (slice_.py:898-900, quirk Q1)
    # fast_countour_mip's exponent is the GUI's border size (constants.py:814: PROJECTION_BORDER_SIZE = 1.0): the sweep times that
    # default (base ** 1 is the identity in glibc's powf, checked exhaustively); any other exponent goes through the restated glibc
    # powf on the device (csrc/glibc_powf.h) -- timed apart as `contour_mip_axis0_exponent_2`, same parity gate
    FCM_N, FCM_N2 = 1.0, 2.0

    def step():
        # MaxIP (int16-exact) + MIDA (f32, the reference's operation order) along each of the three axes,
        # plus one contour MIP (fast_countour_mip, tmip 0) -- every image blown up to the 2048^2 viewport.
        # mida_internal's min / max pre-pass over the volume (mips.rs:113-121) is taken ONCE PER SWEEP, inside the step: the
        # three MIDA images of a sweep see the same resident volume (DeviceVolume.image_range; nothing is carried over from
        # the previous step -- the range is dropped first)
        vol.forget_image_range()
        with vol.timer.span("minmax_prepass"):
            vol.image_range()
        for axis in range(3):
            with vol.timer.span("maxip_axis%d" % axis):
                L.check(lib.ivx_dev_mip_reduce(L.I16, vol.image.raw, c64(n), c64(n), c64(n), axis, L.MIP_MAX, proj.ptr, vol.stream))
            with vol.timer.span("viewport"):
                L.check(lib.ivx_dev_replicate_i16(proj.ptr, c64(n), c64(n), f, view[("maxip", axis)].ptr, vol.stream))
            with vol.timer.span("mida_axis%d" % axis):
                vol.mida(axis, WL, WW, proj, status)
            with vol.timer.span("viewport"):
                L.check(lib.ivx_dev_replicate_i16(proj.ptr, c64(n), c64(n), f, view[("mida", axis)].ptr, vol.stream))
        with vol.timer.span("contour_mip_axis0"):
            # fast_countour_mip(tmip 0) = max fold of the contour volume (mips.rs:237-247): folded as it is computed, no temp
            L.check(lib.ivx_dev_fcm_maxip(L.I16, vol.image.raw, c64(n), c64(n), c64(n), ctypes.c_float(FCM_N), 0, proj.ptr, status.ptr,
                                          vol.stream), "fcm_maxip")
        with vol.timer.span("viewport"):
            L.check(lib.ivx_dev_replicate_i16(proj.ptr, c64(n), c64(n), f, view[("contour", 0)].ptr, vol.stream))

    def barrier():
        vol.sync()
        job.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    vol.timer.collect()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = job.max(time.perf_counter() - t0)
    raw = vol.timer.collect()
    spans = {k: float(np.mean(v)) * (7 if k == "viewport" else 1) for k, v in raw.items()}  # ("viewport": seven per step)
    # the contour MIP with an exponent that is not 1 (outside the sweep: four passes, the last one kept for the parity gate)
    proj2 = DeviceBuffer(n * n * 2 + 64)
    for _ in range(4):
        with vol.timer.span("fcm_n2"):
            L.check(lib.ivx_dev_fcm_maxip(L.I16, vol.image.raw, c64(n), c64(n), c64(n), ctypes.c_float(FCM_N2), 0, proj2.ptr, status.ptr,
                                          vol.stream), "fcm_maxip")
    vol.sync()
    fcm_n2_ms = float(np.min(vol.timer.collect()["fcm_n2"]))
    got_n2 = proj2.download((n, n), np.int16)
    copy_gbs = copy_bandwidth(vol, nvox) if job.rank == 0 else None
    up = lambda a2: np.repeat(np.repeat(a2, f, axis=0), f, axis=1)
    got = {k: view[k].download((n * f, n * f), np.int16) for k in view}
    ok_max = all(np.array_equal(got[("maxip", a)], up(img.max(axis=a))) for a in range(3))
    if job.rank != 0:
        return
    proj_ms = sum(v for k, v in spans.items() if k != "viewport")
    # algorithmic bytes: 2 B/voxel per projection, + 2 B/voxel for MIDA's min/max pre-pass ONCE per sweep (the
    # cached range SURVEY 8d allows), the contour MIP 2 B/voxel (fused: no contour volume), and the viewports
    sweep_bytes = 3 * 2.0 * nvox + (3 * 2.0 + 2.0) * nvox + 2.0 * nvox + 7 * (2.0 * n * n + 2.0 * (n * f) ** 2) + 7 * 2.0 * n * n
    worst = max((k for k in spans if k != "viewport"), key=lambda k: spans[k])
    per_kernel_bytes = {"maxip": 2.0 * nvox, "mida": 2.0 * nvox, "contour": 2.0 * nvox, "minmax": 2.0 * nvox}
    res = {
        "metric": "Mvoxel/s segmentation + Mtriangles/s marching-cubes, 512^3 int16, 1/2/4/8 GPU",
        "value": round(job.world * 7 * nvox / (dt / args.steps) / 1e6, 2), "unit": "Mvoxel/s",
        "n_gpus": job.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i16 (MaxIP) / f32 (MIDA, contour MIP)", "data": "synthetic",
        "config": {"workload": "configs[4]: MaxIP (int16-exact) + MIDA (f32) of a %d^3 int16 volume along each of the 3 axes and one contour "
                               "MIP (axis 0), each image written to a %dx%d int16 viewport (%dx%d rays per voxel column, volume.py:678); MIDA's min / max "
                               "pre-pass over the volume once per sweep, inside the step; "
                               "7 projections per step; fp16 cannot hold int16 data (SURVEY H3), VTK's ray caster (volume.py:519-526,641) "
                               "is not installed here and is not the comparator" % (n, n * f, n * f, f, f),
                   "parallelism": "replicas x%d" % job.world},
        "stage_ms": {k: round(v, 4) for k, v in spans.items()},
        "contour_mip_axis0_exponent_2": {"ms": round(fcm_n2_ms, 4), "frac": round(2.0 * nvox / (fcm_n2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         "powf": "glibc's algorithm restated on the device, %s build (ivx_powf_variant: what this host's "
                                                 "libm runs)" % ("FMA" if lib.ivx_powf_variant() else "plain")},
        "roofline": roofline("3-axis MaxIP + MIDA sweep, contour MIP, viewports", sweep_bytes, sum(spans.values()), None, copy_gbs,
                             {"slowest": worst, "projection_ms": round(proj_ms, 4),
                              "per_kernel_frac": {k: round(per_kernel_bytes[k.split("_")[0]] / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                                  for k, v in spans.items() if k != "viewport" and v > 0}}),
        "device": L.device_name(),
    }
    ok_rays = None
    if args.cpu:
        # the C restatement of mips.rs on ALL host cores (OpenMP over the rays, as rayon does in the reference),
        # the whole volume; its images double as the parity gate of the MIDA / contour viewports
        from oracle import oracle as orc
        orc.build()
        t = time.perf_counter()
        ref_max = [np.array(img).max(axis=a) for a in range(3)]
        t1 = time.perf_counter()
        ref_mida = []
        for a in range(3):
            o = np.zeros(tuple(s for i, s in enumerate(shape) if i != a), np.int16)
            orc.mida(img, a, int(WL), int(WW), o)
            ref_mida.append(o)
        t2 = time.perf_counter()
        ref_fcm = np.zeros(shape[1:], np.int16)
        orc.fast_countour_mip(img, FCM_N, 0, int(WL), int(WW), 0, ref_fcm)
        t3 = time.perf_counter()
        ref_fcm2 = np.zeros(shape[1:], np.int16)
        orc.fast_countour_mip(img, FCM_N2, 0, int(WL), int(WW), 0, ref_fcm2)
        ok_rays = all(np.array_equal(got[("mida", a)], up(ref_mida[a])) for a in range(3)) and \
            np.array_equal(got[("contour", 0)], up(ref_fcm)) and np.array_equal(got_n2, ref_fcm2)
        res["cpu_baseline"] = {"value": round(7 * nvox / (t3 - t) / 1e6, 2), "unit": "Mvoxel/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": "the whole volume: numpy .max(axis) x3 %.2fs (one thread: that IS the reference, slice_.py:885-889) + "
                                         "C MIDA x3 %.2fs + C contour MIP %.2fs with OpenMP over the rays on %d cores (rayon in the "
                                         "reference; the Rust originals cannot be built here, VTK's ray caster is not installed)"
                                         % (t1 - t, t2 - t1, t3 - t2, os.cpu_count())}
    else:
        res["cpu_baseline"] = None
    res["parity"] = {"ok": bool(ok_max and ok_rays is not False),
                     "checked": "MaxIP viewports == numpy max(axis) repeated %dx%d bit for bit%s" % (
                         f, f, "; MIDA x3 and contour-MIP viewports (exponents 1 and 2: the host libm's powf) == the C restatement of mips.rs "
                               "bit for bit (unpinned upstream: no Rust toolchain, no reference test)" if ok_rays is not None else " (--no-cpu: MIDA / contour not compared)")}
    if not res["parity"]["ok"]:
        print(json.dumps(res), flush=True)
        raise SystemExit("bench.py: viewport differs from the reference")
    return res


# ----------------------------------------------------------------------------------------------------------------
# what join_process_surface does to the soup (surface_process.py:204-472), on the bench surface
# ----------------------------------------------------------------------------------------------------------------
def run_surface_tail(args, job):
    """The stages behind marching cubes on configs[1]'s surface (6.3 M triangles at 512^3): indexed mesh (the point merge of
    vtkAppendPolyData + vtkCleanPolyData, :229-268), keep-largest (:376-391), mass properties (:452-458), face normals, ten
    context-aware smoothing steps (invesalius_rs/src/mesh.rs:27-395, call :313-317) -- device-resident, wall time around
    queued calls on the volume's stream -- and the two host-level calls that close the pipeline, fill holes (:396-416) and
    point normals (:420-435), arrays in / arrays out (PCIe inclusive, said so).  `ms_per_step` = the device-resident stages."""
    import ctypes

    from invesalius3_amd import _lib as L, surface_process as sp
    from invesalius3_amd.device import DeviceBuffer, DeviceVolume

    n = args.size or 512
    nvox = n ** 3
    img = synth_v512((n, n, n), seed=SEED + job.rank)
    L.require_device()
    vol = DeviceVolume(img, spacing=(0.5, 0.5, 0.5), device=job.local_rank)
    vol.threshold(*BONE)
    lib = L.lib()
    c64 = ctypes.c_int64
    reps = max(1, args.steps or 5)

    def timed(fn, k=reps):
        fn()
        vol.sync()
        t0 = time.perf_counter()
        for _ in range(k):
            r = fn()
        vol.sync()
        return (time.perf_counter() - t0) * 1e3 / k, r

    stage = {}
    stage["soup"], ntri_soup = timed(vol.marching_cubes)
    stage["indexed_mesh"], (nv, nt) = timed(vol.marching_cubes_indexed)
    ov, of, mass = DeviceBuffer(nv * 12 + 16), DeviceBuffer(nt * 12 + 16), DeviceBuffer(64)
    nrm = DeviceBuffer(nt * 24 + 16)
    n1, n2, nr = c64(0), c64(0), c64(0)

    def keep():
        L.check(lib.ivx_dev_mesh_keep_largest(vol._verts.ptr, c64(nv), vol._faces.ptr, c64(nt), ov.ptr, c64(nv), of.ptr, c64(nt),
                                              ctypes.byref(n1), ctypes.byref(n2), ctypes.byref(nr), vol.stream))
        return n1.value, n2.value, nr.value

    stage["keep_largest"], largest = timed(keep)

    def massp():
        L.check(lib.ivx_dev_mesh_mass_properties(vol._verts.ptr, vol._faces.ptr, c64(nt), mass.ptr, vol.stream))

    stage["mass_properties"], _ = timed(massp)
    vol.sync()
    vol_area = [float(x) for x in mass.download((8,), np.float64)[:2]]

    def normals():
        L.check(lib.ivx_dev_mesh_face_normals(vol._verts.ptr, L.F32, vol._faces.ptr, c64(nt), nrm.ptr, vol.stream))

    stage["face_normals"], _ = timed(normals)
    work = DeviceBuffer(nv * 12 + 16)

    def smooth(iters):
        def run():
            L.check(lib.ivx_memcpy_d2d(work.ptr, vol._verts.ptr, ctypes.c_size_t(nv * 12), vol.stream))
            L.check(lib.ivx_dev_context_aware_smoothing(work.ptr, L.F32, c64(nv), vol._faces.ptr, c64(nt), nrm.ptr,
                                                        ctypes.c_double(0.7), ctypes.c_double(3.0), ctypes.c_double(0.5),
                                                        ctypes.c_int(iters), None, None, vol.stream))
        return run

    t0, _ = timed(smooth(0), 3)
    t10, _ = timed(smooth(10), 3)
    stage["smoothing_setup"] = t0
    stage["smoothing_10_steps"] = t10 - t0
    verts = vol._verts.download((nv, 3), np.float32)
    faces = vol._faces.download((nt, 3), np.int32)
    soup = vol.marching_cubes(download=True)
    # host-level calls (arrays in, arrays out: upload + kernels + download), on the largest region like the reference's pipeline
    kv, kf = ov.download((largest[0], 3), np.float32), of.download((largest[1], 3), np.int32)
    t_fill = t_pn = None
    for _ in range(2):  # (the second call: the first one sizes the library's workspace blocks -- hipMalloc of a few hundred MB)
        t = time.perf_counter()
        fv, ff, nholes = sp.fill_holes(kv, kf, 300.0)
        t_fill = (time.perf_counter() - t) * 1e3
        t = time.perf_counter()
        pv, pf, pn, cn = sp.point_normals(fv, ff, 80.0, True, True)
        t_pn = (time.perf_counter() - t) * 1e3
    # parity: what the indexed mesh must be (the soup, bit for bit) and what the numbers must add up to (float64 numpy)
    ok_soup = None
    if soup is not None:
        ok_soup = bool(soup.shape[0] == nt and np.array_equal(verts[faces].reshape(-1, 9), soup.reshape(-1, 9)))
    a, b, c = (verts[faces[:, k]].astype(np.float64) for k in range(3))
    cr = np.cross(b - a, c - a)
    area_np = 0.5 * np.sqrt((cr * cr).sum(1)).sum()
    vol_np = abs((a * np.cross(b, c)).sum() / 6.0)
    ok_mass = bool(abs(vol_area[0] - vol_np) <= 1e-9 * max(1.0, vol_np) and abs(vol_area[1] - area_np) <= 1e-9 * max(1.0, area_np))
    unit = np.abs(np.sqrt((pn.astype(np.float64) ** 2).sum(1)) - 1.0).max() if len(pn) else 0.0
    ok_tail = bool(len(pf) == len(ff) and unit < 1e-5)
    V, F = float(nv), float(nt)
    alg = {"soup": nvox * 1.0 + 36.0 * F, "indexed_mesh": nvox * 1.0 + 12.0 * V + 12.0 * F, "keep_largest": 2.0 * (12.0 * V + 12.0 * F),
           "mass_properties": 12.0 * V + 12.0 * F, "face_normals": 12.0 * V + 36.0 * F, "smoothing_10_steps": 20.0 * (24.0 * V + 12.0 * F)}
    frac = {k: round(alg[k] / (stage[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for k in alg}
    dev_ms = sum(stage[k] for k in ("indexed_mesh", "keep_largest", "mass_properties", "face_normals", "smoothing_setup", "smoothing_10_steps"))
    res = {"metric": "Mvoxel/s segmentation + Mtriangles/s marching-cubes, 512^3 int16, 1/2/4/8 GPU", "value": round(nt / (dev_ms * 1e-3) / 1e6, 2), "unit": "Mtriangles/s", "n_gpus": 1, "steps": reps, "warmup": 1,
           "ms_per_step": round(dev_ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "SURVEY 8(f) on configs[1]'s surface (%d^3, %d triangles, %d points): indexed mesh, keep-largest, mass "
                                  "properties, face normals, 10 context-aware smoothing steps (device-resident); fill holes + point "
                                  "normals as host calls on the largest region" % (n, nt, nv), "parallelism": "single GPU"},
           "stage_ms": {k: round(v, 4) for k, v in stage.items()},
           "host_call_ms": {"fill_holes": round(t_fill, 2), "point_normals": round(t_pn, 2),
                            "note": "arrays in, arrays out: upload + kernels + download of %d triangles (PCIe inclusive), second call" % len(kf)},
           "triangles": int(nt), "points": int(nv), "regions": int(largest[2]), "largest_region_triangles": int(largest[1]),
           "holes_filled": int(nholes), "points_after_splitting": int(len(pv)), "volume_area": vol_area,
           "roofline": roofline("indexed mesh (k_mc_count + k_mci_count + k_mci_vertices_levels + k_mci_faces)", alg["indexed_mesh"],
                                stage["indexed_mesh"], None, None, {"per_stage_frac": frac, "algorithmic_bytes_per_stage": alg,
                                "note": "mask 1 B/voxel + 12 B per point + 12 B per face written (SURVEY 8d's per-unit figures); every "
                                        "stage's fraction of 8 TB/s from wall time around queued calls"}),
           "cpu_baseline": None,
           "parity": {"ok": bool(ok_mass and ok_tail and ok_soup is not False),
                      "compared": "verts[faces] == the soup bit for bit, same order (%s); volume / area == float64 numpy on the "
                                  "downloaded mesh to 1e-9 relative; point normals unit length, faces kept; VTK's own filters "
                                  "(vtkCleanPolyData, vtkPolyDataConnectivityFilter, vtkFillHolesFilter, vtkPolyDataNormals) are "
                                  "not installed: their outputs are unpinned, tests/test_mesh_tail.py pins the rules" % ok_soup}}
    if not res["parity"]["ok"]:
        print(json.dumps(res), flush=True)
        raise SystemExit("bench.py: the surface tail's checks failed")
    return res


# ----------------------------------------------------------------------------------------------------------------
# configs[3]: 2048^3 Z-sharded over the ranks (STRONG scaling), threshold + marching cubes + cross-slab stitch
# ----------------------------------------------------------------------------------------------------------------
def run_sharded2048(args, job):
    import ctypes
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    from invesalius3_amd import _lib as L
    from invesalius3_amd.device import DeviceVolume, c64
    from invesalius3_amd.parallel import SlabVolume

    rank, world = job.rank, job.world
    n = args.size or 2048
    if n % world:
        raise SystemExit("bench.py --config sharded2048: %d slices do not split over %d ranks" % (n, world))
    nz = n // world
    rng = np.random.default_rng(SEED)
    blobs = np.stack([rng.uniform(0.15, 0.85, 6), rng.uniform(0.15, 0.85, 6), rng.uniform(0.15, 0.85, 6),
                      rng.uniform(0.12, 0.28, 6)], axis=1).astype(np.float32)  # (cz, cy, cx, sigma) x 6, as synth_v512 draws them
    lib = L.lib()

    def fill(ptr, stream, z0=rank * nz, dz=nz):
        L.check(lib.ivx_dev_synth_volume(ptr, c64(dz), c64(n), c64(n), c64(z0), c64(n), ctypes.c_uint32(SEED), L.ptr(blobs), stream),
                "synth_volume")

    L.set_device(job.local_rank)
    vol = SlabVolume(None, rank, world, comm=job.comm, device=job.local_rank, shape=(nz, n, n), fill=fill)
    nvox_local = nz * n * n

    def step():
        # threshold -> marching cubes as an indexed piece -> cross-slab stitch on the device (edge identity; one neighbour
        # exchange of 32 bytes per point word of the shared plane + one 8-byte all-gather), all inside the timed step
        with vol.timer.span("threshold"):
            vol.threshold(BONE[0], BONE[1], preserve=False)
        return vol.marching_cubes_stitched(from_binary=True)[2]

    def barrier():
        vol.sync()
        job.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    vol.timer.collect()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ntri = step()
    barrier()
    dt = job.max(time.perf_counter() - t0)
    raw_spans = vol.timer.collect()
    spans = {k: float(np.mean(v)) for k, v in raw_spans.items()}
    comm_ms = {k[5:]: {"ms_per_step": round(float(np.sum(v)) / max(args.steps, 1), 4), "calls_per_step": round(len(v) / max(args.steps, 1), 2)}
               for k, v in raw_spans.items() if k.startswith("comm_")}
    sc = vol.stitch_counts
    nverts, merged = sc["vertices"] - sc["dropped_copies"], sc["dropped_copies"]
    ntri_all, nverts_all, merged_all = job.sum(ntri), job.sum(nverts), job.sum(merged)
    copy_gbs = copy_bandwidth(vol, nvox_local) if rank == 0 else None
    if rank != 0:
        return
    mc_ms = spans.get("mc_count", 0.0) + spans.get("mci_count", 0.0) + spans.get("mci_emit", 0.0)
    stage_time = {"threshold": spans.get("threshold", 0.0), "marching_cubes": mc_ms}
    stage_bytes = {"threshold": 3.0 * nvox_local, "marching_cubes": 1.0 * nvox_local + 36.0 * ntri}
    dom = max(stage_time, key=lambda k: stage_time[k])
    res = {
        "metric": "Mvoxel/s segmentation + Mtriangles/s marching-cubes, 512^3 int16, 1/2/4/8 GPU",
        "value": round(n ** 3 / (dt / args.steps) / 1e6, 2), "unit": "Mvoxel/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "i16", "data": "synthetic",
        "config": {"workload": "configs[3]: %d^3 int16 (synthesised in HBM, %d slices per GPU), threshold(226..3071) + marching-cubes(mask@127) "
                               "+ cross-slab stitch" % (n, nz), "parallelism": "z-slab x%d" % world,
                   "collectives": "RCCL via libivx ivx_comm_* (one image-halo slice per neighbour, once)" if world > 1 else "none"},
        "triangles": ntri_all, "mtriangles_per_s": round(ntri / (mc_ms * 1e-3) / 1e6, 2) if mc_ms > 0 else None,
        "stage_ms": {k: round(v, 4) for k, v in spans.items()},
        "comm_ms_per_step": comm_ms or None,
        "stitch": {"ms_per_step_inside_the_timed_steps": round(spans.get("stitch", 0.0), 4), "stitched_vertices": nverts_all,
                   "merged_on_shared_planes": merged_all,
                   "note": "device kernels (k_mci_sig / _match / _gid0 / _stitch_*): the shared planes' vertices are matched by edge "
                           "identity, faces renumbered to global ids, kept vertices compacted; marching cubes runs as the indexed "
                           "mesh (one interpolation per unique vertex) so that there is something to stitch"},
        "roofline": roofline(dom, stage_bytes[dom], stage_time[dom], None, copy_gbs,
                             {"per_stage_frac": {k: round(stage_bytes[k] / (stage_time[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                                 for k in stage_time if stage_time[k] > 0}}),
        "device": L.device_name(),
    }
    if args.cpu:
        # bounded sample: the first slices of rank 0's slab, through the oracle with the reference's own parallelism (numpy threshold on
        # one thread, marching cubes pooled over 20+1-slice pieces) and, for parity, through a fresh resident volume on the GPU
        from oracle import oracle as orc
        orc.build()
        # a multiple of the reference's 20-slice piece: with dz % 20 == 1 its piece loop (surface.py:1366-1380) contours the top cell
        # layer twice, and the sample could not be compared with one whole-volume pass
        # (six pieces of the reference's Pool decomposition, like configs[1]'s baseline: ~5e8 voxels at 2048^2, a few seconds)
        sl = max(20, min(nz, 120) // 20 * 20)
        from invesalius3_amd.device import DeviceBuffer
        buf = DeviceBuffer(sl * n * n * 2)
        fill(buf.ptr, None, 0, sl)
        L.synchronize()
        sub = buf.download((sl, n, n), np.int16)
        buf.close()
        t = time.perf_counter()
        mask = np.zeros((sl + 1, n + 1, n + 1), np.uint8)
        orc.set_mask_threshold_volume(mask, sub, BONE)
        t1 = time.perf_counter()
        rois = [slice(i * 20, i * 20 + 21) for i in range(int(round(sl / 20 + 0.5, 0))) if i * 20 < sl]
        cores = max(1, min(len(rois), os.cpu_count() or 1))
        with ThreadPoolExecutor(cores) as pool:
            parts = list(pool.map(lambda r: orc.create_surface_piece(None, mask, r, (1.0, 1.0, 1.0), 0, 0, True), rois))
        t2 = time.perf_counter()
        want = np.concatenate(parts)
        small = DeviceVolume(sub)
        small.threshold(BONE[0], BONE[1])
        got = small.marching_cubes(from_binary=True, download=True)
        gmask = small.download_mask()
        small.close()
        ok = got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)) and \
            zlib.crc32(gmask) == zlib.crc32(np.ascontiguousarray(mask[1:, 1:, 1:]))
        res["cpu_baseline"] = {"value": round(sub.size / (t2 - t) / 1e6, 3), "unit": "Mvoxel/s", "cores": cores, "kind": "port",
                               "sample": "first %d slices of the volume (%d voxels): numpy threshold %.2fs + C marching cubes %.2fs on %d "
                                         "threads over %d pieces (%d triangles)" % (sl, sub.size, t1 - t, t2 - t1, cores, len(rois), len(want))}
        res["parity"] = {"ok": bool(ok), "checked": "mask CRC and float32 triangle soup of the sample, bit for bit, GPU vs CPU oracle"}
        if not ok:
            print(json.dumps(res), flush=True)
            raise SystemExit("bench.py: sharded2048 sample differs from the CPU oracle")
    else:
        res["cpu_baseline"] = None
    return res


def other_configs(args, job, runners):
    """After the headline's timed region (default run, one GPU): BASELINE configs[2] at 512^3 (both branches of do_watershed),
    configs[4] and configs[3] on this one GPU, once each with a short timed region, so that the driver's own run carries
    their numbers and parity -- {name: {ms, frac, parity_ok, differs_from_reference, ...}}.  Full lines: `--config <name>`."""
    import contextlib
    import copy
    plan = (("watershed_ift_512", "watershed", 512, 2, 1, False), ("watershed_gui_default_512", "watershed_sk", 512, 2, 1, False),
            ("mip_sweep_512", "mip", None, 10, 2, False), ("sharded2048_on_one_gpu", "sharded2048", None, 3, 1, False),
            # past the 256 MiB Infinity Cache, and configs[2] at its stated size
            ("region_grow_generic_512", "grow_generic", 512, 10, 2, False), ("grow_mc_1024", "grow_mc", 1024, 5, 2, True),
            ("watershed_ift_1024", "watershed", 1024, 2, 1, False), ("watershed_gui_default_1024", "watershed_sk", 1024, 2, 1, False),
            # SURVEY 8(f)'s stages on the bench surface
            ("surface_tail_512", "surface_tail", 512, 5, 1, False))
    skip = set(filter(None, os.environ.get("IVX_BENCH_SKIP", "").split(",")))  # (names to leave out: short smoke runs)
    out = {}
    for name, cfg, size, steps, warmup, hbm in plan:
        if name in skip:
            continue
        a = copy.copy(args)
        a.config, a.size, a.steps, a.warmup, a.ws_raw, a.bounded, a.hbm_synth, a.scaling = cfg, size, steps, warmup, False, True, hbm, "weak"
        t = time.perf_counter()
        try:
            with contextlib.redirect_stdout(sys.stderr):  # (a failing gate prints its own record: keep stdout to ONE line)
                r = runners[cfg](a, job)
            par = r.get("parity") or {}
            out[name] = {"workload": r["config"]["workload"], "ms": r["ms_per_step"], "steps": steps, "mvoxel_per_s": r["value"],
                         "kernel": r["roofline"]["kernel"], "frac": r["roofline"]["frac"],
                         "per_stage_frac": r["roofline"].get("per_stage_frac"), "stage_ms": r.get("stage_ms"),
                         "parity_ok": par.get("ok"), "differs_from_reference": r.get("differs_from_reference", 0 if par.get("ok") else None),
                         "parity_compared": par.get("compared") or par.get("checked"),
                         "cpu_baseline": {k: (r.get("cpu_baseline") or {}).get(k) for k in ("value", "unit", "cores", "kind")},
                         "wall_s": round(time.perf_counter() - t, 1)}
        except BaseException as e:  # (SystemExit of a failed parity gate included: the headline line still goes out, with the failure in it)
            if isinstance(e, KeyboardInterrupt):
                raise
            out[name] = {"error": "%s: %s" % (type(e).__name__, e), "parity_ok": False, "wall_s": round(time.perf_counter() - t, 1)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=("grow_mc", "grow_generic", "watershed", "watershed_sk", "mip", "sharded2048", "surface_tail"), default="grow_mc",
                    help="grow_mc = BASELINE configs[1] (default, the metric's config); watershed = configs[2] (IFT branch), watershed_sk = configs[2] with the GUI's default scikit-image branch; sharded2048 = "
                         "configs[3] (strong scaling: the whole volume split over --gpus); mip = configs[4]")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="grow_mc with --gpus N: weak = every GPU owns its own 512^3 slab of a (512 N) x 512 x 512 volume (default); strong = "
                         "configs[1]'s ONE 512^3 volume split into N slabs of 512 / N slices")
    ap.add_argument("--hbm-synth", action="store_true", help="grow_mc, one GPU: make the volume in HBM (k_synth) instead of with numpy "
                    "(--size 1024 in seconds); the CPU oracle then runs on the downloaded volume")
    ap.add_argument("--size", type=int, default=None, help="edge of the volume (defaults: 512 / 1024 / 512 per GPU; 2048 in total for sharded2048)")
    ap.add_argument("--ws-raw", action="store_true", help="watershed_sk: the image - image.min() branch instead of the GUI's default window/level")
    ap.add_argument("--no-cpu", dest="cpu", action="store_false", help="skip the CPU baseline + full-size parity check")
    ap.add_argument("--no-others", dest="others", action="store_false",
                    help="default config only: do not run configs[2] (both branches, 512^3), configs[4] and configs[3] (one GPU) once each "
                         "after the headline's timed region (`other_configs` of the JSON line)")
    ap.add_argument("--live-reference", action="store_true",
                    help="watershed: run live scipy and the serial oracle on the WHOLE timed volume instead of quoting "
                         "tests/golden/ws{512,1024}_full.npz (512^3: ~3 min of one core; 1024^3: ~25 min and ~50 GB of RAM)")
    ap.add_argument("--cpu-slices", type=int, default=None, help="(kept for old command lines; 0 = --no-cpu)")
    ap.add_argument("--dry-comm", action="store_true", help="bring the communicator up, run ivx_comm_selftest on every rank "
                    "(at --gpus 1: on a one-rank RCCL communicator), print one JSON line and exit")
    args = ap.parse_args()
    if args.cpu_slices == 0:
        args.cpu = False
    dflt = {"grow_mc": (20, 3), "grow_generic": (20, 3), "watershed": (3, 1), "watershed_sk": (2, 1), "mip": (20, 3), "sharded2048": (5, 2), "surface_tail": (5, 1)}[args.config]
    args.steps = dflt[0] if args.steps is None else args.steps
    args.warmup = dflt[1] if args.warmup is None else args.warmup

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    from invesalius3_amd import _lib as L
    L.require_device()
    if args.dry_comm:
        return dry_comm(world)
    job = Ranks()
    runners = {"grow_mc": run_grow_mc, "grow_generic": run_grow_generic, "watershed": run_watershed, "watershed_sk": run_watershed_sk, "mip": run_mip, "sharded2048": run_sharded2048,
               "surface_tail": run_surface_tail}
    res = runners[args.config](args, job)
    if res is not None and args.config == "grow_mc" and args.others and args.cpu and world == 1 and args.size is None \
            and not args.hbm_synth and os.environ.get("IVX_FORCE_SLAB") != "1":
        res["other_configs"] = other_configs(args, job, runners)
    if args.config == "grow_mc" and world > 1 and args.scaling == "weak" and args.size is None and not args.hbm_synth \
            and os.environ.get("IVX_BENCH_NO_STRONG") != "1":
        # The metric reads "512^3 ... 1/2/4/8 GPU": the first multi-GPU run yields BOTH curves -- the line's own numbers are weak
        # scaling (512^3 per GPU), `strong_scaling` is configs[1]'s one 512^3 volume split over the same ranks, same steps,
        # same keys.  Every rank takes part (the slabs exchange planes); a failure is recorded, the weak line still goes out.
        import copy
        a = copy.copy(args)
        a.scaling, a.cpu = "strong", False
        try:
            r2 = run_grow_mc(a, job)
            if res is not None and r2 is not None:
                res["strong_scaling"] = {k: r2.get(k) for k in ("value", "unit", "ms_per_step", "steps", "warmup", "scaling", "stage_ms",
                                                                "comm_ms_per_step", "triangles", "region_voxels", "region_grow_rounds",
                                                                "timed_region_s", "config")}
        except Exception as e:  # noqa: BLE001
            if res is not None:
                res["strong_scaling"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if res is not None:
        print(json.dumps(res), flush=True)
    if job.comm is not None:
        CStdoutToStderr.stay()  # the JSON line is out; whatever RCCL still has to say goes to stderr
        job.comm.barrier()
        job.comm.close()


if __name__ == "__main__":
    main()
