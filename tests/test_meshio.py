"""The OBJ/PLY reader of librt_amd.so (include/rt_host.h, rt_mesh_load).  The reference has no loader --
its meshes are literals (src/main.cpp:25-121) -- so there is nothing to pin against: the tests check the
reader against files written here, and that a loaded mesh goes through the same preprocessing and tree
build as a literal one."""
import struct

import numpy as np
import pytest

from raytracing_amd import scene as S
from raytracing_amd import scenes


CUBE_OBJ = """# unit cube, quads, per-face normals, negative and positive indices mixed
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1
v 1 0 1
v 1 1 1
v 0 1 1
vn 0 0 -1
vn 0 0 1
vn 0 -1 0
vn 0 1 0
vn -1 0 0
vn 1 0 0
f 1//1 4//1 3//1 2//1
f 5//2 6//2 7//2 8//2
f 1/9/3 2/9/3 6/9/3 5/9/3
f -5//4 -1//4 -2//4 -6//4
f 1//5 5//5 8//5 4//5
f 2//6 3//6 7//6 6//6
"""


def test_obj_quads_with_normals(tmp_path):
    path = tmp_path / "cube.obj"
    path.write_text(CUBE_OBJ)
    pos, nrm, tris = S.load_mesh(path)
    assert pos.shape == (24, 3) and nrm.shape == (24, 3) and tris.shape == (12, 3)     # 6 faces x 4 merged pairs
    assert tris.max() == 23
    # every triangle's geometric normal agrees with the normals stored at its vertices
    e1, e2 = pos[tris[:, 1]] - pos[tris[:, 0]], pos[tris[:, 2]] - pos[tris[:, 0]]
    g = np.cross(e1, e2)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    for k in range(3):
        assert np.allclose(nrm[tris[:, k]], g)
    # face "f -5//4 -1//4 -2//4 -6//4" is the top face: vertices 4,8,7,3 (1-based)
    top = pos[tris[6:8].ravel()]
    assert (top[:, 1] == 1).all()


def test_obj_without_normals_gets_area_weighted_normals(tmp_path):
    path = tmp_path / "tetra.obj"
    path.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1 3 2\nf 1 2 4\nf 1 4 3\nf 2 3 4\n")
    pos, nrm, tris = S.load_mesh(path)
    assert pos.shape == (4, 3) and tris.shape == (4, 3)
    assert np.allclose(np.linalg.norm(nrm, axis=1), 1.0, atol=1e-6)
    # the normal at the origin vertex points away from the opposite face
    assert np.allclose(nrm[0], -np.ones(3) / np.sqrt(3), atol=1e-6)


def _grid_mesh(n=5):
    pos, nrm, tris = scenes.grid_plane(n)
    return np.asarray(pos, np.float32).reshape(-1, 3), np.asarray(nrm, np.float32).reshape(-1, 3), np.asarray(tris, np.uint32).reshape(-1, 3)


def _write_ply(path, pos, nrm, tris, binary):
    header = ["ply", "format %s 1.0" % ("binary_little_endian" if binary else "ascii"), "comment test file",
              "element vertex %d" % len(pos), "property float x", "property float y", "property float z"]
    if nrm is not None:
        header += ["property float nx", "property float ny", "property float nz"]
    header += ["property uchar confidence", "element face %d" % len(tris), "property list uchar int vertex_indices", "end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode())
        for i in range(len(pos)):
            row = list(pos[i]) + (list(nrm[i]) if nrm is not None else [])
            if binary:
                f.write(struct.pack("<%df" % len(row), *row) + struct.pack("<B", 7))
            else:
                f.write((" ".join(repr(float(v)) for v in row) + " 7\n").encode())
        for t in tris:
            if binary:
                f.write(struct.pack("<B3i", 3, *[int(v) for v in t]))
            else:
                f.write(("3 %d %d %d\n" % tuple(int(v) for v in t)).encode())


@pytest.mark.parametrize("binary", [False, True])
@pytest.mark.parametrize("with_normals", [False, True])
def test_ply_round_trip(tmp_path, binary, with_normals):
    pos, nrm, tris = _grid_mesh()
    path = tmp_path / "grid.ply"
    _write_ply(path, pos, nrm if with_normals else None, tris, binary)
    p2, n2, t2 = S.load_mesh(path)
    assert np.array_equal(p2, pos) and np.array_equal(t2, tris)
    if with_normals:
        assert np.array_equal(n2, nrm)
    else:
        assert np.allclose(np.abs(n2), np.abs(nrm), atol=1e-6)          # a plane: all normals along one axis


def test_loaded_mesh_builds_the_same_scene_as_the_literal_one(tmp_path):
    pos, nrm, tris = _grid_mesh(6)
    path = tmp_path / "grid.ply"
    _write_ply(path, pos, nrm, tris, True)
    def make(mesh):
        m = S.Models()
        m.add(S.MeshTransform((0, 0, 0), S.IDENTITY_QUAT, (4, 1, 4)), 0, 0)
        m.add(S.MeshTransform((0, 2, 0), S.IDENTITY_QUAT, (1, 1, 1)), 0, 0)
        return S.Scene(S.Camera((0, 5, 5), (0.8, 3.14159, 0), 0.8), S.Meshes([mesh]), [S.Material((0.5, 0.5, 0.5), 0.5, 0.0)], m,
                       inflate_mode=1).preprocess()
    a, b = make((pos, nrm, tris)), make(S.load_mesh(path))
    assert np.array_equal(a.preprocessed_tris.tris, b.preprocessed_tris.tris)
    assert np.array_equal(a.bsp_tree.leaf_refs, b.bsp_tree.leaf_refs) and a.bsp_tree.inode_count == b.bsp_tree.inode_count


def test_bad_files_are_reported(tmp_path):
    with pytest.raises(ValueError, match="cannot read"):
        S.load_mesh(tmp_path / "nope.obj")
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nf 1 2 9\n")
    with pytest.raises(ValueError, match="out of range"):
        S.load_mesh(bad)
    trunc = tmp_path / "trunc.ply"
    trunc.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\nend_header\n\x00\x00")
    with pytest.raises(ValueError, match="truncated"):
        S.load_mesh(trunc)
    empty = tmp_path / "empty.obj"
    empty.write_text("# nothing\n")
    pos, nrm, tris = S.load_mesh(empty)
    assert pos.shape == (0, 3) and tris.shape == (0, 3)


# ---- files in the dialects common exporters write ---------------------------------------------------------------------
# (hand-written here after the formats' published descriptions: Wavefront's OBJ text format and the Stanford PLY header
# grammar; parity unpinned by construction, since the reference reads no files)

BLENDER_STYLE_OBJ = (
    "# Blender 3.6.0\r\n"
    "# www.blender.org\r\n"
    "mtllib tetra.mtl\r\n"
    "o Tetra\r\n"
    "v 0.000000 0.000000 0.000000\r\n"
    "v 1.000000 0.000000 0.000000 \r\n"          # trailing blank
    "v\t0.000000\t1.000000\t0.000000\r\n"        # tabs
    "v 0.000000 0.000000 1.000000 1.0\r\n"       # optional w
    "vt 0.0 0.0\r\n"
    "vt 1.0 0.0\r\n"
    "vt 0.0 1.0\r\n"
    "vn 0.0 0.0 -1.0\r\n"
    "vn 0.0 -1.0 0.0\r\n"
    "vn -1.0 0.0 0.0\r\n"
    "vn 0.5774 0.5774 0.5774\r\n"
    "s off\r\n"
    "g side\r\n"
    "usemtl Material\r\n"
    "f 1/1/1 3/3/1 2/2/1\r\n"
    "f 1/1/2 2/2/2 4/3/2\r\n"
    "f 1/1/3 4/2/3 3/3/3\r\n"
    "l 1 2\r\n"                                  # a polyline element: not geometry this reader keeps
    "f 2/1/4 3/2/4 4/3/4\r\n")


def test_obj_as_a_modelling_program_writes_it(tmp_path):
    path = tmp_path / "tetra.obj"
    path.write_bytes(BLENDER_STYLE_OBJ.encode())
    pos, nrm, tris = S.load_mesh(path)
    assert tris.shape == (4, 3)
    assert pos.shape == nrm.shape == (12, 3)          # 4 faces x 3 distinct (position, normal) pairs
    corners = {tuple(p) for p in pos.tolist()}
    assert corners == {(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)}
    assert np.allclose(nrm[tris[0]], [0, 0, -1]) and np.allclose(nrm[tris[3]], 0.5774, atol=1e-4)


def test_obj_polygon_is_split_into_a_fan_and_texture_only_indices_work(tmp_path):
    path = tmp_path / "pentagon.obj"
    path.write_text("v 0 0 0\nv 2 0 0\nv 3 1 0\nv 1 2 0\nv -1 1 0\nvt 0 0\nf 1/1 2/1 3/1 4/1 5/1\n")
    pos, nrm, tris = S.load_mesh(path)
    assert tris.tolist() == [[0, 1, 2], [0, 2, 3], [0, 3, 4]]
    assert np.allclose(nrm, [0, 0, 1])


def _ply(header_lines, body, binary=False):
    head = "ply\nformat %s 1.0\n" % ("binary_little_endian" if binary else "ascii") + "\n".join(header_lines) + "\nend_header\n"
    return head.encode() + (body if binary else body.encode())


def test_ply_with_colours_doubles_and_alternative_property_names(tmp_path):
    """What scanning and mesh-processing tools write: comments, per-vertex colours and a confidence value between the
    coordinates and after them, double coordinates, `vertex_index` with uint indices and a per-face flags byte."""
    header = ["comment VCGLIB generated", "obj_info scanned", "element vertex 4",
              "property double x", "property double y", "property double z",
              "property uchar red", "property uchar green", "property uchar blue", "property float confidence",
              "element face 2", "property uchar flags", "property list uchar uint vertex_index"]
    verts = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)]
    ascii_body = "".join(f"{x} {y} {z} 255 128 0 0.5\n" for x, y, z in verts) + "0 3 0 1 2\n0 3 0 2 3\n"
    path = tmp_path / "a.ply"
    path.write_bytes(_ply(header, ascii_body))
    pos, nrm, tris = S.load_mesh(path)
    assert pos.tolist() == [list(map(float, v)) for v in verts] and tris.tolist() == [[0, 1, 2], [0, 2, 3]]
    assert np.allclose(nrm, [0, 0, 1])
    binary_body = b"".join(struct.pack("<dddBBBf", x, y, z, 255, 128, 0, 0.5) for x, y, z in verts)
    binary_body += struct.pack("<BBIII", 0, 3, 0, 1, 2) + struct.pack("<BBIII", 0, 3, 0, 2, 3)
    path = tmp_path / "b.ply"
    path.write_bytes(_ply(header, binary_body, binary=True))
    pos2, nrm2, tris2 = S.load_mesh(path)
    assert np.array_equal(pos, pos2) and np.array_equal(tris, tris2) and np.allclose(nrm, nrm2)


def test_ply_quads_strips_of_other_elements_and_crlf_header(tmp_path):
    header = ["element vertex 4", "property float x", "property float y", "property float z",
              "property float nx", "property float ny", "property float nz",
              "element face 1", "property list uchar int vertex_indices",
              "element edge 1", "property int vertex1", "property int vertex2"]
    body = "0 0 0 0 0 1\n1 0 0 0 0 1\n1 1 0 0 0 1\n0 1 0 0 0 1\n4 0 1 2 3\n0 1\n"
    text = ("ply\nformat ascii 1.0\n" + "\n".join(header) + "\nend_header\n" + body).replace("\n", "\r\n")
    path = tmp_path / "quad.ply"
    path.write_bytes(text.encode())
    pos, nrm, tris = S.load_mesh(path)
    assert tris.tolist() == [[0, 1, 2], [0, 2, 3]] and np.allclose(nrm, [0, 0, 1])


def test_ply_big_endian_is_refused_with_a_message(tmp_path):
    path = tmp_path / "be.ply"
    path.write_bytes(b"ply\nformat binary_big_endian 1.0\nelement vertex 0\nproperty float x\nproperty float y\nproperty float z\n"
                     b"element face 0\nproperty list uchar int vertex_indices\nend_header\n")
    with pytest.raises(Exception) as e:
        S.load_mesh(path)
    assert "big" in str(e.value).lower() or "endian" in str(e.value).lower()
