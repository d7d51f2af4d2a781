"""CPU model of the region-growing engine's ROUND STRUCTURE (k_flood.hip): coarse pass on all-candidate blocks, then rounds of
dirty-tile visits -- a visit = at most `itcap` local iterations of "OR of the 3 x 3 neighbour rows, each dilated by one voxel
in x, AND candidates, then close the x-runs" on a tile staged with a one-voxel halo, the halo constant during the visit --
with the wake-up rule of the kernel (a changed face wakes the neighbours that can see it, an exhausted tile wakes itself).
It counts what the GPU timeline is made of: productive rounds, tile visits, local iterations; the result is checked against
scipy.ndimage.label.  Use: explore tile shapes / iteration caps / coarse-block sizes offline before spending GPU time.

    python tools/sim_flood.py [n=512] [--ty 16 --tz 16 --itcap 16 --block 16]      (~1-3 min per configuration at 512^3)"""
import argparse
import sys
import time

import numpy as np
from scipy import ndimage

sys.path.insert(0, ".")
from bench import BONE, synth_v512  # noqa: E402

TX = 64


def run_fill(R, C):
    """close the x-runs: every candidate run (last axis) that holds a reached voxel becomes reached"""
    n, W = R.shape[:-1], R.shape[-1]
    Cf, Rf = C.reshape(-1, W), R.reshape(-1, W)
    start = Cf.copy()
    start[:, 1:] &= ~Cf[:, :-1]
    ids = np.cumsum(start, axis=1) + (np.arange(Cf.shape[0])[:, None] * (W + 1))
    hit = np.zeros(Cf.shape[0] * (W + 1) + W + 2, bool)
    hit[ids[Rf & Cf]] = True
    return (Cf & hit[ids]).reshape(*n, W)


def visit(Rw, Cw, itcap):
    """Rw, Cw: (T, tz+2, ty+2, TX+2) windows of T tiles; returns (new interior R, iterations per tile, exhausted flags)"""
    T = Rw.shape[0]
    R = Rw.copy()
    Ci = Cw[:, 1:-1, 1:-1, 1:-1]
    its = np.zeros(T, np.int32)
    live = np.ones(T, bool)
    exhausted = np.zeros(T, bool)
    for it in range(itcap):
        idx = np.nonzero(live)[0]
        if not len(idx):
            break
        Rl = R[idx]
        xd = Rl.copy()
        xd[..., 1:] |= Rl[..., :-1]
        xd[..., :-1] |= Rl[..., 1:]
        nb = np.zeros_like(Rl[:, 1:-1, 1:-1, 1:-1])
        for dz in range(3):
            for dy in range(3):
                nb |= xd[:, dz:dz + nb.shape[1], dy:dy + nb.shape[2], 1:-1]
        cur = Rl[:, 1:-1, 1:-1, 1:-1]
        new = run_fill(cur | (nb & Ci[idx]), Ci[idx])
        ch = (new != cur).reshape(len(idx), -1).any(1)
        Rl[:, 1:-1, 1:-1, 1:-1] = new
        R[idx] = Rl
        its[idx] += 1
        live[idx] = ch
        if it == itcap - 1:
            exhausted[idx] = ch
    return R[:, 1:-1, 1:-1, 1:-1], its, exhausted


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", nargs="?", type=int, default=512)
    ap.add_argument("--ty", type=int, default=16)
    ap.add_argument("--tz", type=int, default=16)
    ap.add_argument("--itcap", type=int, default=16)
    ap.add_argument("--block", type=int, default=16, help="coarse blocks of block^3 voxels (0: no coarse pass)")
    ap.add_argument("--recoarse", type=int, default=0, help="after every round from this one on, every all-candidate block that "
                    "holds a reached voxel floods its component of the block graph again (0: the coarse pass runs once)")
    a = ap.parse_args()
    n, ty, tz = a.n, a.ty, a.tz
    img = synth_v512((n, n, n))
    C = (img >= BONE[0]) & (img <= BONE[1])
    z, y, x = np.unravel_index(int(np.argmax(img)), img.shape)
    assert C[z, y, x]
    t0 = time.time()
    want = ndimage.label(C, structure=np.ones((3, 3, 3), bool))[0]
    want = want == want[z, y, x]
    print("component %d voxels of %d candidates (scipy.ndimage.label %.1f s)" % (want.sum(), C.sum(), time.time() - t0), flush=True)
    R = np.zeros_like(C)
    R[z, y, x] = True
    # ---- coarse pass: all-candidate blocks reach each other across faces, edges and corners (26-neighbourhood) -------
    lab = None
    if a.block:
        b = a.block
        nb = n // b
        full = C.reshape(nb, b, nb, b, nb, b).all(axis=(1, 3, 5))
        lab = ndimage.label(full, structure=np.ones((3, 3, 3), bool))[0]
        done_labels = set()
        sb = lab[z // b, y // b, x // b]
        done_labels.add(int(sb))
        if sb:
            whole = lab == sb
            R |= np.repeat(np.repeat(np.repeat(whole, b, 0), b, 1), b, 2) & C
            print("coarse pass: %d of %d blocks of %d^3 wholly reached (%d voxels, %.1f %% of the component)"
                  % (whole.sum(), full.size, b, R.sum(), 100.0 * R.sum() / want.sum()), flush=True)
    ntz, nty, ntx = n // tz, n // ty, n // TX
    Rp = np.pad(R, 1)
    Cp = np.pad(C, 1)

    def tiles_near(Rcur):
        """tiles that hold an unreached candidate next to a reached voxel (what the coarse apply / the seed enlists)"""
        front = ndimage.binary_dilation(Rcur, structure=np.ones((3, 3, 3), bool)) & C & ~Rcur
        return set(map(tuple, np.argwhere(front.reshape(ntz, tz, nty, ty, ntx, TX).any(axis=(1, 3, 5)))))

    dirty = tiles_near(R)
    rounds = visits = iters = 0
    hist = []
    while dirty:
        tl = sorted(dirty)
        T = len(tl)
        Rw = np.empty((T, tz + 2, ty + 2, TX + 2), bool)
        Cw = np.empty_like(Rw)
        for i, (a_, b_, c_) in enumerate(tl):
            Rw[i] = Rp[a_ * tz:a_ * tz + tz + 2, b_ * ty:b_ * ty + ty + 2, c_ * TX:c_ * TX + TX + 2]
            Cw[i] = Cp[a_ * tz:a_ * tz + tz + 2, b_ * ty:b_ * ty + ty + 2, c_ * TX:c_ * TX + TX + 2]
        newR, its, exhausted = visit(Rw, Cw, a.itcap)
        nxt = set()
        gained = 0
        for i, (a_, b_, c_) in enumerate(tl):
            old = Rw[i, 1:-1, 1:-1, 1:-1]
            chg = newR[i] & ~old
            if not chg.any():
                continue
            gained += int(chg.sum())
            Rp[a_ * tz + 1:a_ * tz + tz + 1, b_ * ty + 1:b_ * ty + ty + 1, c_ * TX + 1:c_ * TX + TX + 1] = newR[i]
            zl, zh = chg[0].any(), chg[-1].any()
            yl, yh = chg[:, 0].any(), chg[:, -1].any()
            xl, xh = chg[:, :, 0].any(), chg[:, :, -1].any()
            for dz in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        if not (dz or dy or dx):
                            continue
                        if (dz == 0 or (zl if dz < 0 else zh)) and (dy == 0 or (yl if dy < 0 else yh)) and (dx == 0 or (xl if dx < 0 else xh)):
                            t = (a_ + dz, b_ + dy, c_ + dx)
                            if 0 <= t[0] < ntz and 0 <= t[1] < nty and 0 <= t[2] < ntx:
                                nxt.add(t)
            if exhausted[i]:
                nxt.add((a_, b_, c_))
        rounds += 1
        visits += T
        iters += int(its.sum())
        hist.append((T, int(its.max()), gained))
        print("round %2d: %5d tiles, max %2d / mean %.1f iterations, %8d voxels gained" % (rounds, T, its.max(), its.mean(), gained), flush=True)
        if not gained:
            rounds -= 1  # (the round that finds nothing is the GPU's first empty round only if its list is empty: here it is a visit round)
        dirty = nxt
        if a.recoarse and lab is not None and rounds >= a.recoarse and gained:
            b = a.block
            nb = n // b
            Rc = Rp[1:-1, 1:-1, 1:-1]
            has = Rc.reshape(nb, b, nb, b, nb, b).any(axis=(1, 3, 5))
            new_labels = set(np.unique(lab[has & (lab > 0)]).tolist()) - done_labels
            if new_labels:
                done_labels |= new_labels
                whole = np.isin(lab, list(new_labels))
                add = np.repeat(np.repeat(np.repeat(whole, b, 0), b, 1), b, 2) & C & ~Rc
                Rc |= add
                near = tiles_near(Rc)
                dirty |= near
                print("          coarse pass again: %d more blocks wholly reached, %d voxels, %d tiles enlisted" % (whole.sum(), add.sum(), len(near)), flush=True)
    got = Rp[1:-1, 1:-1, 1:-1]
    ok = bool(np.array_equal(got, want))
    print("tile %dx%dx%d itcap %d block %d: %d productive rounds, %d visits, %d tile-iterations, == scipy component: %s (%.0f s)"
          % (TX, ty, tz, a.itcap, a.block, rounds, visits, iters, ok, time.time() - t0))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
