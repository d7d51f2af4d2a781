"""The piece file format of create_surface_piece (VTK XML PolyData, inline binary): writer -> reader round trip and the
layout vtkXMLPolyDataReader expects (surface_process.py:193-196, 237-243).  CPU only: no kernel is involved."""
import base64
import re

import numpy as np


def test_vtp_round_trip_and_layout(tmp_path):
    from invesalius3_amd import surface_process as sp
    rng = np.random.default_rng(0)
    verts = rng.normal(size=(17, 3)).astype(np.float32)
    faces = rng.integers(0, 17, (9, 3)).astype(np.int32)
    path = str(tmp_path / "piece_0_21.vtp")
    sp.write_vtp(path, verts, faces)
    v, f = sp.read_vtp(path)
    assert v.dtype == np.float32 and f.dtype == np.int32 and np.array_equal(v, verts) and np.array_equal(f, faces)
    txt = open(path).read()
    assert '<VTKFile type="PolyData"' in txt and 'byte_order="LittleEndian"' in txt and 'header_type="UInt32"' in txt
    assert 'NumberOfPoints="17"' in txt and 'NumberOfPolys="9"' in txt
    m = re.search(r'Name="offsets"[^>]*>([^<]*)<', txt)
    raw = base64.b64decode(m.group(1))
    assert np.frombuffer(raw[:4], "<u4")[0] == 9 * 4 and np.array_equal(np.frombuffer(raw[4:], "<i4"), np.arange(1, 10) * 3)
    sp.write_vtp(path, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32))  # an empty piece is a valid file
    v, f = sp.read_vtp(path)
    assert v.shape == (0, 3) and f.shape == (0, 3)
