"""Golden vectors from the REFERENCE ITSELF: invesalius.data.watershed_process.do_watershed imported from /root/reference
and run in the build container, all four branches.

    python3 tests/golden/make_golden_ref_dowatershed.py          # Python 3.10 (the reference needs >= 3.10)

The module's GUI-side imports (wx, vtk, pubsub, gdcm ...) are satisfied by empty stand-in modules: do_watershed touches none
of them.  Its numpy / scipy calls run under this interpreter's numpy 2.2 / scipy 1.15 (np126.npz shows that the pinned
numpy 1.26.4 / scipy 1.7.1 return the same bits); `skimage.segmentation.watershed` is not importable here, so the name the
reference imports is bound to a proxy that runs the real scikit-image 0.18.3 under /opt/conda/bin/python3.9 on the very
arrays the reference passes.  Nothing here is read at test time except the .npz it writes.
"""
import importlib.abc
import importlib.machinery
import os
import queue
import subprocess
import sys
import tempfile
import types
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CONDA_PY = "/opt/conda/bin/python3.9"


class _Fake(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        # VTK classes are subclassed by the reference (interactor styles): they have to be real classes
        m = (type(name, (), {"__getattr__": lambda self_, n: (lambda *a, **k: "9.3.0" if n == "GetVTKVersion" else mock.MagicMock())}) if name.startswith("vtk")
             else mock.MagicMock(name=self.__name__ + "." + name))
        setattr(self, name, m)
        return m


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("skimage", "vtkmodules", "vtk", "wx", "gdcm", "pubsub", "nibabel", "h5py", "imageio", "PIL", "psutil", "invesalius_rs",
             "invesalius_cy", "torch", "onnxruntime")

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Fake(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def skimage_watershed_proxy(image, markers, connectivity):
    """skimage.segmentation.watershed(image, markers, connectivity) by the real scikit-image in the other interpreter"""
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "i.npy"), np.asarray(image))
        np.save(os.path.join(d, "m.npy"), np.asarray(markers))
        np.save(os.path.join(d, "c.npy"), np.asarray(connectivity))
        code = ("import numpy as np, warnings; warnings.simplefilter('ignore'); from skimage.segmentation import watershed; "
                "np.save(%r, watershed(np.load(%r), np.load(%r), np.load(%r)))"
                % (os.path.join(d, "o.npy"), os.path.join(d, "i.npy"), os.path.join(d, "m.npy"), os.path.join(d, "c.npy")))
        subprocess.run([CONDA_PY, "-c", code], check=True, cwd=d, stderr=subprocess.DEVNULL)
        return np.load(os.path.join(d, "o.npy"))


def ct_like(shape, seed):
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    f = np.zeros(shape)
    for _ in range(4):
        c = rng.uniform(0.2, 0.8, 3) * np.array(shape)
        s = rng.uniform(3, 7)
        f += 1800 * np.exp(-(((z - c[0]) * 1.5) ** 2 + (y - c[1]) ** 2 + (x - c[2]) ** 2) / (2 * s * s))
    f += rng.normal(0, 25, shape) - 1000
    return np.clip(f, -1024, 3071).astype(np.int16), np.unravel_index(int(np.argmax(f)), shape)


def main(path):
    sys.meta_path.insert(0, _Finder())
    import pubsub.pub
    pubsub.pub.subscribe = lambda *a, **k: (None, True)
    sys.path.insert(0, "/root/reference")
    os.environ.setdefault("HOME", tempfile.mkdtemp())
    from scipy.ndimage import generate_binary_structure

    from invesalius.data import watershed_process as ref
    ref.watershed = skimage_watershed_proxy
    data, names = {}, []

    def run(name, image, markers, bstruct, algorithm, mg_size, use_ww_wl, wl, ww):
        with tempfile.TemporaryDirectory() as d:
            tfile = os.path.join(d, "mask.dat")
            mm = np.memmap(tfile, shape=image.shape, dtype="uint8", mode="w+")
            mm.flush()
            q = queue.Queue()
            ref.do_watershed(image, markers, tfile, image.shape, bstruct, algorithm, mg_size, use_ww_wl, wl, ww, q)
            assert q.get(timeout=2) == 1
            out = np.array(np.memmap(tfile, shape=image.shape, dtype="uint8", mode="r"))
        names.append(name)
        data["img_" + name], data["mk_" + name], data["st_" + name] = image, markers, bstruct.astype(np.uint8)
        data["par_" + name] = np.array([algorithm, str(int(use_ww_wl)), str(wl), str(ww), "x".join(str(v) for v in mg_size)])
        data["out_" + name] = out

    # the reference's own fixture (tests/test_segmentation_tools.py:170-213), both algorithms
    image = np.zeros((5, 5, 5), dtype=np.int16)
    image[1:4, 1:4, 1:4] = 100
    markers = np.zeros((5, 5, 5), dtype=np.int16)
    markers[2, 2, 2] = 1
    markers[0, 0, 0] = 2
    for alg in ("Watershed", "Watershed IFT"):
        run("fixture_" + alg.replace(" ", ""), image, markers, generate_binary_structure(3, 1), alg, (3, 3, 3), False, 0, 0)
    # CT-like volumes with brush markers, the four branches, 6 / 26 neighbours, and one slice
    k = 0
    for shape, conn in (((14, 30, 34), 1), ((12, 26, 28), 3)):
        img, am = ct_like(shape, 31 + k)
        mk = np.zeros(shape, np.int16)
        mk[max(am[0] - 1, 0):am[0] + 2, am[1] - 2:am[1] + 3, am[2] - 2:am[2] + 3] = 1
        mk[:2, :4, :4] = 2
        mk[-2:, -4:, -4:] = 2
        for alg in ("Watershed", "Watershed IFT"):
            for use_ww_wl in (True, False):
                run("vol%d_%s_%d" % (k, alg.replace(" ", ""), use_ww_wl), img, mk, generate_binary_structure(3, conn), alg, (3, 3, 3),
                    use_ww_wl, 300, 400)
        k += 1
    img, am = ct_like((1, 48, 52), 40)
    sl, mk = img[0], np.zeros((48, 52), np.int16)
    mk[am[1] - 2:am[1] + 3, am[2] - 2:am[2] + 3] = 1
    mk[:4, :4] = 2
    for alg in ("Watershed", "Watershed IFT"):
        run("slice_" + alg.replace(" ", ""), sl, mk, generate_binary_structure(2, 1), alg, (3, 3), True, 300, 400)
    data["names"] = np.array(names)
    np.savez_compressed(path, **data)
    print(len(names), "reference runs:", ", ".join(names))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_do_watershed.npz"))
