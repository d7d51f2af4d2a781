"""The .inv3 project container, read and written without wx / VTK -- the data format either side of the hot path
(SURVEY.md 8(f) rank 4).

Layout, as the reference writes it (invesalius/project.py:219-345 SavePlistProject, 652-690 Compress; masks:
invesalius/data/mask.py:315-366): a tar (optionally gzip) archive holding ONE directory with

    main.plist        {format_version, name, modality, orientation, window_width, window_level, scalar_range,
                       spacing, matrix: {filename: "matrix.dat", shape, dtype}, masks: {idx: "mask_N.plist"},
                       surfaces: {...}, measurements: "measurements.plist", ...}
    matrix.dat        raw C-order dump of the (dz, dy, dx) image
    mask_N.plist      {index, name, colour, opacity, threshold_range, edition_threshold_range, visible,
                       mask_file: "mask_N.dat", mask_shape: (dz+1, dy+1, dx+1), edited, derived_from}
    mask_N.dat        raw C-order dump of the padded uint8 mask matrix

Reading mirrors Project.OpenPlistProject / load_from_folder (project.py:346-536) and Mask.OpenPList
(mask.py:347-366): arrays come back as np.memmap views of the extracted files, exactly what the reference hands to
its slice / surface code.  Surfaces are carried as their plist dictionaries (their .vtp payload is VTK's format and
stays untouched)."""
from __future__ import annotations

import datetime
import os
import plistlib
import shutil
import tarfile
import tempfile
import warnings
from dataclasses import dataclass, field

import numpy as np

FORMAT_VERSION = 1.1  # invesalius/constants.py:32


@dataclass
class MaskRecord:
    """The persistent part of invesalius.data.mask.Mask."""
    index: int
    name: str
    matrix: np.ndarray  # (dz+1, dy+1, dx+1) uint8, border row/column/slice = the "modified" flags
    threshold_range: tuple = (0, 0)
    edition_threshold_range: tuple = (0, 0)
    colour: tuple = (0.0, 1.0, 0.0)
    opacity: float = 0.4
    visible: bool = True
    edited: bool = False
    derived_from: str = "original"

    @property
    def interior(self) -> np.ndarray:
        """matrix[1:, 1:, 1:]: the voxels (slice_.py works on this view)."""
        return self.matrix[1:, 1:, 1:]


@dataclass
class Project:
    name: str = ""
    modality: str = "CT"
    orientation: int = 0
    window: float = 0.0
    level: float = 0.0
    threshold_range: tuple = (0, 0)
    spacing: tuple = (1.0, 1.0, 1.0)
    matrix: np.ndarray | None = None
    matrix_shape: tuple = ()
    matrix_dtype: str = "int16"
    matrix_filename: str = ""
    affine: list | None = None
    compress: bool = False
    format_version: float = FORMAT_VERSION
    masks: dict = field(default_factory=dict)
    surfaces: dict = field(default_factory=dict)
    measurements: dict = field(default_factory=dict)
    image_versions: list = field(default_factory=list)
    dirpath: str = ""
    _owned_tmp: str = ""  # the directory open_inv3 made with mkdtemp (removed by close()); never a caller's workdir

    def close(self):
        """drop the memmaps and the extraction directory"""
        self.matrix = None
        self.masks.clear()
        self.image_versions.clear()
        if self._owned_tmp and os.path.isdir(self._owned_tmp):
            shutil.rmtree(self._owned_tmp, ignore_errors=True)
        self._owned_tmp = ""
        self.dirpath = ""


def _safe_member(member: tarfile.TarInfo, folder: str):
    """project.py:693-705 custom_tar_filter: refuse members that would land outside `folder`, links and devices."""
    if not (member.isfile() or member.isdir()):
        return None
    target = os.path.abspath(os.path.join(folder, member.name))
    if os.path.commonpath([os.path.abspath(folder), target]) != os.path.abspath(folder):
        return None
    return member


def extract(filename, folder) -> list:
    """project.py:708-730 Extract: unpack every safe member, return the extracted paths in archive order."""
    out = []
    with tarfile.open(filename, "r") as tar:
        for t in tar.getmembers():
            m = _safe_member(t, folder)
            if m is None:
                warnings.warn("skipping unsafe archive member %r" % t.name, stacklevel=2)
                continue
            try:
                tar.extract(m, path=folder, filter="data")
            except TypeError:  # Python without extraction filters
                tar.extract(m, path=folder)
            if m.isfile():
                out.append(os.path.join(folder, m.name))
    return out


def compress(folder_name: str, filename, filelist: dict, gz: bool = False):
    """project.py:652-690 Compress: `filelist` maps source paths to archive names inside the directory `folder_name`."""
    fd, tmp = tempfile.mkstemp()
    os.close(fd)
    with tarfile.open(tmp, "w:gz" if gz else "w") as tar:
        for src, arc in filelist.items():
            arc = os.path.normpath(arc)
            if ".." in arc or os.path.isabs(arc):
                continue
            if not os.path.exists(src):
                if arc.startswith("matrix"):
                    raise FileNotFoundError("Critical project file missing during save: %s (target: %s)" % (src, arc))
                continue
            tar.add(src, arcname=os.path.join(folder_name, arc))
    shutil.move(tmp, filename)


def _inside(dirpath: str, name: str) -> str:
    """`name` (a file name taken from a plist of the project in `dirpath`) as a path, provided it stays inside that
    folder once symlinks and `..` are resolved.  Plists come out of archives people exchange: a name such as
    `../../home/x/.ssh/id_rsa`, or an absolute one, must never be opened -- let alone copied into the next archive."""
    root = os.path.realpath(dirpath)
    path = os.path.realpath(os.path.join(root, str(name)))
    if os.path.isabs(str(name)) or os.path.commonpath([root, path]) != root:
        raise ValueError("project file name %r leaves the project folder" % (name,))
    return path


def load_from_folder(dirpath: str) -> Project:
    """project.py:378-536 load_from_folder, minus the GUI objects."""
    with open(os.path.join(dirpath, "main.plist"), "rb") as f:
        main = plistlib.load(f, fmt=plistlib.FMT_XML)
    p = Project(dirpath=dirpath)
    p.format_version = main["format_version"]
    if p.format_version > FORMAT_VERSION:
        warnings.warn("project written by a newer format version (%s)" % p.format_version, stacklevel=2)
    p.name = main["name"]
    p.modality = main["modality"]
    p.orientation = main["orientation"]
    p.window = main["window_width"]
    p.level = main["window_level"]
    p.threshold_range = tuple(main["scalar_range"])
    p.spacing = tuple(main["spacing"])
    p.compress = main.get("compress", True)
    p.matrix_filename = _inside(dirpath, main["matrix"]["filename"])
    p.matrix_shape = tuple(int(v) for v in main["matrix"]["shape"])
    p.matrix_dtype = main["matrix"]["dtype"]
    if main.get("affine", ""):
        p.affine = main["affine"]
    need = int(np.prod(p.matrix_shape)) * np.dtype(p.matrix_dtype).itemsize
    have = os.path.getsize(p.matrix_filename)
    if have < need:
        raise ValueError("matrix.dat holds %d bytes, shape %s of %s needs %d" % (have, p.matrix_shape, p.matrix_dtype, need))
    p.matrix = np.memmap(p.matrix_filename, shape=p.matrix_shape, dtype=p.matrix_dtype, mode="r+")
    for version in main.get("image_versions", []):
        vpath = _inside(dirpath, version["filename"])
        if os.path.exists(vpath):
            p.image_versions.append((version["label"], np.memmap(vpath, shape=p.matrix_shape, dtype=p.matrix_dtype, mode="r+")))
    masks = main.get("masks", {})
    for key in sorted(masks, key=lambda k: int(k)):
        try:
            rec = open_mask_plist(_inside(dirpath, masks[key]))
        except FileNotFoundError as e:  # project.py:478-487: a missing mask file is skipped with a warning
            warnings.warn("Skipping mask %r: %s" % (masks[key], e), stacklevel=2)
            continue
        rec.index = len(p.masks)
        p.masks[rec.index] = rec
    surfaces = main.get("surfaces", {})
    for key in sorted(surfaces, key=lambda k: int(k)):
        spath = _inside(dirpath, surfaces[key])
        if os.path.exists(spath):
            with open(spath, "rb") as f:
                sd = plistlib.load(f, fmt=plistlib.FMT_XML)
            payload = sd.get("polydata")
            if payload:  # the reference parses this file as VTP; here it is carried along byte for byte, so it must be ours
                try:
                    _inside(dirpath, payload)
                except ValueError as e:
                    warnings.warn("surface %s: %s -- payload dropped" % (key, e), stacklevel=2)
                    sd["polydata"] = ""
            p.surfaces[int(key)] = sd
    mpath = _inside(dirpath, main.get("measurements", "measurements.plist"))
    if os.path.exists(mpath):
        with open(mpath, "rb") as f:
            p.measurements = plistlib.load(f, fmt=plistlib.FMT_XML)
    return p


def open_mask_plist(filename: str) -> MaskRecord:
    """mask.py:347-366 Mask.OpenPList + _open_mask (392-401)."""
    with open(filename, "rb") as f:
        m = plistlib.load(f, fmt=plistlib.FMT_XML)
    path = _inside(os.path.dirname(os.path.abspath(filename)), m["mask_file"])
    if not os.path.exists(path):
        raise FileNotFoundError("Mask data file not found: %r" % path)
    shape = tuple(int(v) for v in m["mask_shape"])
    if os.path.getsize(path) < int(np.prod(shape)):
        raise ValueError("%s is smaller than its mask_shape %s" % (m["mask_file"], shape))
    return MaskRecord(index=m["index"], name=m["name"], matrix=np.memmap(path, shape=shape, dtype="uint8", mode="r+"),
                      threshold_range=tuple(m["threshold_range"]), edition_threshold_range=tuple(m["edition_threshold_range"]),
                      colour=tuple(m["colour"]), opacity=m["opacity"], visible=m["visible"], edited=m.get("edited", False),
                      derived_from=m.get("derived_from", "original"))


def open_inv3(filename, workdir: str | None = None) -> Project:
    """Project.OpenPlistProject (project.py:346-376): extract, then load the folder the archive holds."""
    owned = "" if workdir else tempfile.mkdtemp(prefix="ivx3_")
    base = workdir or owned
    try:
        files = extract(filename, base)
        if not files:
            raise ValueError("%s holds no files" % filename)
        p = load_from_folder(os.path.abspath(os.path.dirname(files[0])))
    except BaseException:
        if owned:
            shutil.rmtree(owned, ignore_errors=True)  # nothing is returned that could clean it up later
        raise
    p._owned_tmp = owned
    return p


def save_inv3(filename, project: Project, gz: bool | None = None):
    """Project.SavePlistProject (project.py:219-345): image, filtered image versions, masks, surfaces (plist + polydata
    file) and measurements.  A project opened with open_inv3 and written back keeps all of them (not carried: the
    per-version filter parameters and `active_image_version` of project.py:253-293)."""
    gz = project.compress if gz is None else gz
    tmp = tempfile.mkdtemp(prefix="ivx3_save_")
    try:
        filelist = {}
        image = np.ascontiguousarray(project.matrix)
        mpath = os.path.join(tmp, "matrix.dat")
        image.tofile(mpath)
        filelist[mpath] = "matrix.dat"
        main = {
            "format_version": FORMAT_VERSION,
            "invesalius_version": "invesalius3_amd",
            "date": datetime.datetime.now().isoformat(),
            "compress": bool(gz),
            "name": project.name,
            "modality": project.modality,
            "orientation": project.orientation,
            "window_width": project.window,
            "window_level": project.level,
            "scalar_range": list(project.threshold_range),
            "spacing": list(project.spacing),
            "image_fiducials": [],
            "matrix": {"filename": "matrix.dat", "shape": list(image.shape), "dtype": str(image.dtype)},
            "annotations": {},
        }
        # filtered image versions (project.py:267-295): matrix_vN.dat + label
        versions = []
        for i, (label, mat) in enumerate(project.image_versions):
            vname = "matrix_v%d.dat" % i
            vpath = os.path.join(tmp, vname)
            np.ascontiguousarray(mat).tofile(vpath)
            filelist[vpath] = vname
            versions.append({"label": label, "filename": vname})
        main["image_versions"] = versions
        if project.affine is not None:
            main["affine"] = project.affine
        masks = {}
        for index, rec in project.masks.items():
            stem = "mask_%d" % index
            dpath = os.path.join(tmp, stem + ".dat")
            np.ascontiguousarray(rec.matrix, dtype=np.uint8).tofile(dpath)
            filelist[dpath] = stem + ".dat"
            plist = {"index": int(index), "name": rec.name, "colour": list(rec.colour[:3]), "opacity": rec.opacity,
                     "threshold_range": list(rec.threshold_range),
                     "edition_threshold_range": list(rec.edition_threshold_range), "visible": rec.visible,
                     "mask_file": stem + ".dat", "mask_shape": list(rec.matrix.shape), "edited": rec.edited,
                     "derived_from": rec.derived_from}
            ppath = os.path.join(tmp, stem + ".plist")
            with open(ppath, "wb") as f:
                plistlib.dump(plist, f)
            filelist[ppath] = stem + ".plist"
            masks[str(index)] = stem + ".plist"
        main["masks"] = masks
        # surfaces (surface.py:121-156 SavePlist): the dictionary as loaded / given + its polydata file, found next to
        # the project it was loaded from (or at an absolute path the caller put there)
        surfaces = {}
        for index, sdict in project.surfaces.items():
            stem = "surface_%d" % int(index)
            sd = dict(sdict)
            payload = sd.get("polydata")
            if payload:
                # a relative name belongs to the project the surface was loaded from and must stay inside its folder;
                # an absolute one is a file the CALLER put there (load_from_folder never lets one through)
                src = payload if os.path.isabs(payload) else _inside(project.dirpath or os.getcwd(), payload)
                if not os.path.exists(src):
                    raise FileNotFoundError("surface %s: polydata file %r not found (looked in %r)" % (index, payload, project.dirpath))
                ext = os.path.splitext(payload)[1] or ".vtp"
                filelist[src] = stem + ext
                sd["polydata"] = stem + ext
            ppath = os.path.join(tmp, stem + ".plist")
            with open(ppath, "wb") as f:
                plistlib.dump(sd, f)
            filelist[ppath] = stem + ".plist"
            surfaces[str(index)] = stem + ".plist"
        main["surfaces"] = surfaces
        mp = os.path.join(tmp, "measurements.plist")
        with open(mp, "wb") as f:
            plistlib.dump(project.measurements or {}, f)
        filelist[mp] = "measurements.plist"
        main["measurements"] = "measurements.plist"
        pp = os.path.join(tmp, "main.plist")
        with open(pp, "wb") as f:
            plistlib.dump(main, f)
        filelist[pp] = "main.plist"
        compress(os.path.basename(tmp), filename, filelist, gz)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def new_mask(project: Project, name: str, threshold_range, index: int | None = None) -> MaskRecord:
    """An empty padded mask matrix for `project` (mask.py create_mask: shape + 1 on every axis, zero filled)."""
    shape = tuple(s + 1 for s in project.matrix.shape)
    index = len(project.masks) if index is None else index
    rec = MaskRecord(index=index, name=name, matrix=np.zeros(shape, np.uint8), threshold_range=tuple(threshold_range),
                     edition_threshold_range=tuple(threshold_range))
    project.masks[index] = rec
    return rec
