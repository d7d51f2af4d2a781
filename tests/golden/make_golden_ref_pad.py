"""Golden vectors from the REFERENCE's own pad_image (invesalius/data/surface_process.py:52-68), imported from /root/reference:
what create_surface_piece feeds to the contour filter when fill_border_holes is on (image: iinfo.min border, mask: 0 border;
one extra slice at the volume's bottom / top).

    python3 tests/golden/make_golden_ref_pad.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_ref_dowatershed as M  # noqa: E402


def main(path):
    sys.meta_path.insert(0, M._Finder())
    import pubsub.pub
    pubsub.pub.subscribe = lambda *a, **k: (None, True)
    sys.path.insert(0, "/root/reference")
    from invesalius.data import surface_process as rsp
    rng = np.random.default_rng(20261002)
    img = rng.integers(-1000, 2000, size=(6, 7, 8)).astype(np.int16)
    msk = rng.choice(np.array([0, 1, 2, 253, 254, 255], np.uint8), size=(6, 7, 8))
    d = {"img": img, "msk": msk}
    for pb in (0, 1):
        for pt in (0, 1):
            d["img_%d%d" % (pb, pt)] = rsp.pad_image(img, np.iinfo(img.dtype).min, bool(pb), bool(pt))
            d["msk_%d%d" % (pb, pt)] = rsp.pad_image(msk, 0, bool(pb), bool(pt))
    np.savez_compressed(path, **d)
    print(len(d), "arrays")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_pad.npz"))
