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
