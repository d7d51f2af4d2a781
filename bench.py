#!/usr/bin/env python3
"""bench.py -- BASELINE.json configs[1]: 512x512x512 synthetic int16 volume,
threshold + 26-neighbour region growing + marching cubes, on N x MI355X (one process per GPU).

A "step" is one pass of the hot path over the resident volume:
    1. out_mask = zeros                       (np.zeros of styles.py:3190)
    2. threshold image -> mask (255/0)         (slice_.py:1240-1247)
    3. floodfill_threshold(image, seed, lo, hi, 1, 26-conn, out_mask); mask[out_mask==1] = 254   (styles.py:3200-3214)
    4. marching cubes of the mask at iso 127 (from_binary), whole volume   (surface_process.py:100-186)
The volume is uploaded once before the timed region (inputs resident in HBM).  N > 1: weak scaling, every rank
owns one 512^3 Z-slab of a (512*N) x 512 x 512 volume; slab boundaries exchange one mask plane per region-growing
round and one mask slice for marching cubes over RCCL (torch.distributed, backend "nccl").

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel family of the step, measured live with HIP
events on the stream the kernels run on; `cpu_baseline` is the CPU oracle (a C/numpy restatement of the reference:
"port") timed on a bounded sample of the same volume on this box's host cores.
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


def synth_v512(shape=(512, 512, 512), seed=SEED, z_offset=0, z_total=None):
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
    for z0 in range(0, dz, step):
        z1 = min(dz, z0 + step)
        f = np.zeros((z1 - z0, dy, dx), np.float32)
        for b in range(6):
            gz = sum(np.exp(-((zz[z0:z1] - cz[b] - sh) ** 2) / (2 * sg[b] ** 2)) for sh in shifts)
            gy = np.exp(-((yy - cy[b]) ** 2) / (2 * sg[b] ** 2))
            gx = np.exp(-((xx - cx[b]) ** 2) / (2 * sg[b] ** 2))
            f += 1800.0 * gz[:, None, None] * gy[None, :, None] * gx[None, None, :]
        f += 150.0 * np.sin(6.0 * xx)[None, None, :] * np.cos(5.0 * zz[z0:z1])[:, None, None] * np.cos(4.0 * yy)[None, :, None]
        f += nrng.standard_normal(f.shape, dtype=np.float32) * 25.0
        np.clip(f - 1000.0, -1024, 3071, out=f)
        out[z0:z1] = f.astype(np.int16)
    return out


def cpu_baseline(img, seed_xyz, sample_slices):
    """CPU oracle on the first `sample_slices` slices, with the parallelism the REFERENCE has on each stage: numpy
    threshold (one thread, slice loop), serial C flood fill (the Rust one is serial too), and marching cubes over the
    reference's 20+1-slice pieces on a pool of min(pieces, host cores) workers (surface.py:1362-1380 uses
    multiprocessing.Pool the same way; ctypes releases the GIL, so threads do here).  `value` uses the pooled
    marching-cubes time; the one-thread time is in `sample`."""
    from concurrent.futures import ThreadPoolExecutor

    from scipy.ndimage import generate_binary_structure

    from oracle import oracle as orc

    orc.build()
    sub = np.ascontiguousarray(img[:sample_slices])
    dz = sub.shape[0]
    z, y, x = np.unravel_index(int(np.argmax(sub)), sub.shape)
    t0 = time.perf_counter()
    mask = np.zeros(tuple(s + 1 for s in sub.shape), np.uint8)
    orc.set_mask_threshold_volume(mask, sub, BONE)
    t1 = time.perf_counter()
    out_mask = np.zeros(sub.shape, np.uint8)
    orc.floodfill_threshold(sub, [(int(x), int(y), int(z))], BONE[0], BONE[1], 1, generate_binary_structure(3, 3), out_mask)
    mask[1:, 1:, 1:][out_mask.astype(bool)] = 254
    t2 = time.perf_counter()
    n_pieces = int(round(dz / 20 + 0.5, 0))
    rois = [slice(i * 20, i * 20 + 21) for i in range(n_pieces) if i * 20 < dz]
    piece = lambda roi: len(orc.create_surface_piece(None, mask, roi, (1.0, 1.0, 1.0), 0, 0, True))
    ntri = sum(piece(r) for r in rois)
    t3 = time.perf_counter()
    cores = max(1, min(len(rois), os.cpu_count() or 1))
    with ThreadPoolExecutor(cores) as pool:
        ntri_pool = sum(pool.map(piece, rois))
    t4 = time.perf_counter()
    assert ntri_pool == ntri
    nvox = sub.size
    total = (t2 - t0) + (t4 - t3)
    return {
        "value": round(nvox / total / 1e6, 3), "unit": "Mvoxel/s", "cores": cores, "kind": "port",
        "sample": "first %d of 512 slices (%d voxels): numpy threshold %.2fs + serial C floodfill %.2fs + C marching cubes "
                  "%.2fs on %d threads over %d pieces (%.2fs on one thread; %d triangles, %.2f Mtri/s pooled); all on one "
                  "thread: %.3f Mvoxel/s"
                  % (dz, nvox, t1 - t0, t2 - t1, t4 - t3, cores, len(rois), t3 - t2, ntri, ntri / max(t4 - t3, 1e-9) / 1e6,
                     nvox / (t3 - t0) / 1e6),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=512, help="edge of the per-GPU volume (BASELINE: 512)")
    ap.add_argument("--cpu-slices", type=int, default=192, help="slices of the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    force_slab = os.environ.get("IVX_FORCE_SLAB") == "1"  # exercise the sharded path (torch + RCCL) at world 1
    if world > 1 or force_slab:
        import torch
        import torch.distributed as dist  # RCCL

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from scipy.ndimage import generate_binary_structure

    from invesalius3_amd import _lib as L
    from invesalius3_amd.device import DeviceVolume

    L.require_device()
    L.set_device(local_rank if dist is not None else 0)
    n = args.size
    shape = (n, n, n)
    img = synth_v512(shape, z_offset=rank * n, z_total=world * n)
    z, y, x = np.unravel_index(int(np.argmax(img)), img.shape)
    seed = (int(x), int(y), int(z) + rank * n)  # global (x, y, z): every rank seeds the brightest voxel of its slab
    strct = generate_binary_structure(3, 3)

    if dist is not None:
        from invesalius3_amd.parallel import SlabVolume

        vol = SlabVolume(img, rank, world, dist, device=local_rank)
    else:
        vol = DeviceVolume(img)
    nvox = img.size

    # Marching cubes' count, scan and triangle list depend on WHICH voxels are >= 127 only, and `mask[reached] = 254`
    # does not change that: with IVX_PREFETCH=1 the timed steps queue them on a second, low-priority stream right after
    # the threshold pass, held back until the region growing's busy rounds are over; the emit waits for the mask's final
    # bytes (DeviceVolume.surface_prefetch; measured 0.498 -> 0.466 ms per step, profiles/r01_bench_v11_overlap*).
    # The default is strictly one stage after the other: the per-stage roofline figures then describe kernels that
    # had the GPU to themselves, and every GPU count runs the same schedule.
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
        L.synchronize()
        if dist is not None:
            import torch

            torch.cuda.synchronize()
            dist.barrier()

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
    for _ in range(max(1, min(args.steps, 5))):
        step()
    barrier()
    spans = vol.timer.collect()
    spans["region_grow"] = spans_timed.get("region_grow", spans.get("region_grow", []))
    reached = vol.reached_count()
    # achievable streaming bandwidth on this box, measured the same way (HIP events, same stream): device-to-device
    # copy of the int16 volume, read + written bytes over the time of the copy
    copy_gbs = None
    if rank == 0:
        import ctypes

        from invesalius3_amd.device import DeviceBuffer
        tmp = DeviceBuffer(nvox * 2)
        for _ in range(2):
            L.check(L.lib().ivx_memcpy_d2d(tmp.ptr, vol.image.raw, ctypes.c_size_t(nvox * 2), vol.stream))
        for _ in range(5):
            with vol.timer.span("copy"):
                L.check(L.lib().ivx_memcpy_d2d(tmp.ptr, vol.image.raw, ctypes.c_size_t(nvox * 2), vol.stream))
        vol.sync()
        copy_ms = float(np.median(vol.timer.collect()["copy"]))
        copy_gbs = 4.0 * nvox / (copy_ms * 1e-3) / 1e9
        tmp.close()

    if dist is not None:
        import torch

        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([float(ntri), float(reached)], dtype=torch.float64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        ntri, reached = int(c[0].item()), int(c[1].item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        stage_ms = {k: float(np.mean(v)) for k, v in spans.items()}
        mc_ms = stage_ms.get("mc_count", 0.0) + stage_ms.get("mc_emit", 0.0)
        ntri_local = ntri // world
        # algorithmic bytes per launch family (SURVEY.md 8d / DESIGN.md)
        stage_bytes = {
            "threshold": 3.0 * nvox,
            "region_grow": 3.0 * nvox,
            "marching_cubes": 1.0 * nvox + 36.0 * ntri_local,
        }
        stage_time = {"threshold": stage_ms.get("threshold", 0.0), "region_grow": stage_ms.get("region_grow", 0.0),
                      "marching_cubes": mc_ms}
        dom = max(stage_time, key=lambda k: stage_time[k])
        achieved = stage_bytes[dom] / (stage_time[dom] * 1e-3) / 1e9 if stage_time[dom] > 0 else 0.0
        # HBM traffic of the dominant stage from the committed PMC passes (rocprofv3 cannot run inside the bench):
        # profiles/pmc_traffic.json = 2 x FETCH_SIZE + WRITE_SIZE per step, produced by tools/summarize_pmc.py
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                traffic = json.load(f)["traffic_bytes_per_step"].get(dom) if n == 512 and world == 1 else None
        except (OSError, ValueError, KeyError):
            traffic = None
        res = {
            "metric": "Mvoxel/s segmentation + Mtriangles/s marching-cubes, 512^3 int16, 1/2/4/8 GPU",
            "value": round(world * nvox / (dt / args.steps) / 1e6, 2),
            "unit": "Mvoxel/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i16", "data": "synthetic",
            "config": {"workload": "configs[1]: %dx%dx%d int16 per GPU, threshold(226..3071) + 26-neighbour region-grow + marching-cubes(mask@127)" % shape,
                       "global_voxels": world * nvox, "parallelism": "z-slab x%d" % world,
                       "overlap": "marching-cubes count+scan+list on a second stream under region growing" if overlap else "none"},
            "mtriangles_per_s": round(ntri / (mc_ms * 1e-3) / 1e6, 2) if mc_ms > 0 else None,
            "triangles": ntri, "region_voxels": reached, "region_grow_rounds": rounds,
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_ms_source": "region_grow: HIP events inside the timed steps (with marching cubes' count + list running "
                               "beside it when config.overlap says so); other stages: HIP events in up to 5 extra steps "
                               "after the timed region, one stage after the other (every recorded event idles the stream "
                               "for ~4 us)",
            "stage_mvoxel_per_s": {k: round(nvox / (v * 1e-3) / 1e6, 1) for k, v in stage_time.items() if v > 0},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "measured_copy_gbs": round(copy_gbs, 1) if copy_gbs else None,
                         "frac_of_copy": round(achieved / copy_gbs, 4) if copy_gbs else None,
                         "algorithmic_bytes": stage_bytes[dom], "ms": round(stage_time[dom], 4),
                         "per_stage_frac": {k: round(stage_bytes[k] / (stage_time[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                            for k in stage_time if stage_time[k] > 0}},
            "device": L.device_name(),
        }
        if args.cpu_slices > 0 and world == 1:
            res["cpu_baseline"] = cpu_baseline(img, seed, min(args.cpu_slices, n))
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
