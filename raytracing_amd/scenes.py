"""Scene data: the reference's built-in scene (src/main.cpp:25-187) and the procedural scenes used by
the benchmark configurations (the reference has no asset loader and there is no network here, so the
"Cornell-box-style" and "dragon-scale" scenes are generated)."""
import numpy as np

from . import _lib
from .scene import Camera, Material, Meshes, MeshTransform, Models, Scene, IDENTITY_QUAT

F = np.float32


def quad_mesh_tris(quad_count):
    """Two triangles per quad with the winding of quadMeshTris (src/main.cpp:25-37)."""
    i = 4 * np.arange(quad_count, dtype=np.uint32)[:, None]
    return (i + np.array([[0, 2, 1, 1, 2, 3]], np.uint32)).reshape(-1, 3)


def plane():
    """Unit plane in y=0, normal +y (src/main.cpp:40-56)."""
    pos = [(-.5, 0, -.5), (.5, 0, -.5), (-.5, 0, .5), (.5, 0, .5)]
    return np.array(pos, F), np.tile(np.array([[0, 1, 0]], F), (4, 1)), quad_mesh_tris(1)


def cube():
    """Unit cube, four vertices per face in the order front, rear, top, bottom, left, right and the
    per-face corner order of src/main.cpp:59-121."""
    h = .5
    faces = [
        ((0, 0, 1), [(-h, h, h), (h, h, h), (-h, -h, h), (h, -h, h)]),
        ((0, 0, -1), [(h, h, -h), (-h, h, -h), (h, -h, -h), (-h, -h, -h)]),
        ((0, 1, 0), [(-h, h, -h), (h, h, -h), (-h, h, h), (h, h, h)]),
        ((0, -1, 0), [(-h, -h, h), (h, -h, h), (-h, -h, -h), (h, -h, -h)]),
        ((-1, 0, 0), [(-h, h, -h), (-h, h, h), (-h, -h, -h), (-h, -h, h)]),
        ((1, 0, 0), [(h, h, h), (h, h, -h), (h, -h, h), (h, -h, -h)]),
    ]
    pos = np.array([v for _, vs in faces for v in vs], F)
    nrm = np.array([n for n, vs in faces for _ in vs], F)
    return pos, nrm, quad_mesh_tris(6)


def builtin_scene(divisor=3):
    """The scene hard-coded in src/main.cpp:134-187: floor, two mirrors, divisor^3 emissive cubes."""
    L = _lib.lib()
    half_pi = F(1.57079632679489661923132169163975144)
    fov = F(45.0) * F(0.01745329251994329576923690768489)          # glm::radians(45.0f)
    camera = Camera((9.0, 8.0, 16.0), (0.3, -2.6, 0.0), fov)
    meshes = Meshes([plane(), cube()])
    materials = [Material((0.25, 0.25, 0.25), 0.9, 0.0), Material((1.0, 1.0, 1.0), 0.04, 1.0)]
    models = Models()
    models.add(MeshTransform((2.0, 0.0, 2.0), IDENTITY_QUAT, (16.0, 1.0, 16.0)), 0, 0)
    models.add(MeshTransform((0.0, 5.0, -6.0), (half_pi, 0.0, 0.0), (20.0, 1.0, 10.0)), 0, 1)
    models.add(MeshTransform((-6.0, 5.0, 0.0), (0.0, 0.0, -half_pi), (10.0, 1.0, 20.0)), 0, 1)
    scale, subscale = F(4.0), F(0.75)
    span = scale - scale / F(divisor)
    cube_scale = subscale * scale / F(divisor)
    for x in range(divisor):
        xf = F(x) / F(divisor - 1)
        for y in range(divisor):
            yf = F(y) / F(divisor - 1)
            for z in range(divisor):
                zf = F(z) / F(divisor - 1)
                colour = [L.rt_srgb_to_linear(float(c)) for c in (xf, yf, zf)]
                materials.append(Material(colour, 0.5, 0.5, colour))
                position = np.array([(xf - F(0.5)) * span, (yf - F(0.5)) * span, (zf - F(0.5)) * span], F) \
                    + np.array([0.0, 2.5, 0.0], F)
                models.add(MeshTransform(position, IDENTITY_QUAT, (cube_scale,) * 3), 1, len(materials) - 1)
    return Scene(camera, meshes, materials, models, inflate_mode=0)


# ---- procedural meshes ----------------------------------------------------------------------------

def grid_plane(n):
    """Unit plane in y=0 (normal +y) cut into n x n quads; triangle normals (v2-v1)x(v3-v1) point to +y."""
    g = np.linspace(-0.5, 0.5, n + 1, dtype=F)
    zz, xx = np.meshgrid(g, g, indexing="ij")
    pos = np.stack([xx.ravel(), np.zeros(xx.size, F), zz.ravel()], 1).astype(F)
    nrm = np.tile(np.array([[0, 1, 0]], F), (len(pos), 1))
    j, i = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    v00 = (j * (n + 1) + i).ravel()
    v10, v01, v11 = v00 + 1, v00 + (n + 1), v00 + (n + 2)
    # +y normal needs (z-step) x (x-step): first edge along z.
    tris = np.concatenate([np.stack([v00, v01, v10], 1), np.stack([v10, v01, v11], 1)]).astype(np.uint32)
    return pos, nrm, tris


def grid_cube(n):
    """Unit cube whose six faces are n x n grids with outward normals."""
    base_p, base_n, base_t = grid_plane(n)
    base_p = base_p + np.array([0, 0.5, 0], F)
    rots = [np.eye(3),                                     # +y
            np.array([[1, 0, 0], [0, -1, 0], [0, 0, -1]]),  # -y (rotate pi about x)
            np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]]),   # +y -> +z
            np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0]]),   # +y -> -z
            np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1]]),   # +y -> +x
            np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]])]   # +y -> -x
    pos, nrm, tris = [], [], []
    for k, r in enumerate(rots):
        r = r.astype(F)
        pos.append(base_p @ r.T)
        nrm.append(base_n @ r.T)
        tris.append(base_t + k * len(base_p))
    return np.concatenate(pos).astype(F), np.concatenate(nrm).astype(F), np.concatenate(tris).astype(np.uint32)


def displaced_sphere(n, seed=1234, octaves=5, amplitude=0.18):
    """Closed surface: a cube whose six faces are n x n grids, projected onto the unit sphere and
    displaced radially by a multi-octave product of sines.  12*n*n triangles, outward winding, smooth
    vertex normals, vertex valence at most 6 (a fan of more than Leaf::MAX_TRIS triangles around one
    vertex can never be separated by the reference's builder, src/bsp.hpp:206-208)."""
    rng = np.random.default_rng(seed)
    pos, _, tris = grid_cube(n)
    dirs = pos.astype(np.float64)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    radius = np.ones(len(dirs))
    for o in range(octaves):
        k = rng.normal(size=(3, 3)) * (2.0 ** o) * 1.5
        phase = rng.uniform(0, 2 * np.pi, 3)
        radius += amplitude / (2.0 ** o) * np.prod(np.sin(dirs @ k.T + phase), axis=1)
    p = dirs * radius[:, None]
    fn = np.cross(p[tris[:, 1]] - p[tris[:, 0]], p[tris[:, 2]] - p[tris[:, 0]])
    # weld the duplicated cube-edge vertices when averaging normals so shading is smooth across faces
    _, weld = np.unique(np.round(dirs * 1e6).astype(np.int64), axis=0, return_inverse=True)
    weld = weld.ravel()
    acc = np.zeros((weld.max() + 1, 3))
    for c in range(3):
        np.add.at(acc, weld[tris[:, c]], fn)
    vn = acc[weld]
    vn /= np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-30)
    return p.astype(F), vn.astype(F), tris


# ---- benchmark scenes -----------------------------------------------------------------------------

def cornell_scene(wall_grid=16, box_grid=6):
    """Closed Cornell-box-style room (about 4 k triangles, mixed GGX roughness 0.04-0.9): six tessellated
    walls, two tessellated boxes, one ceiling emitter; camera inside looking down -z."""
    pi = F(np.pi)
    hp = F(np.pi / 2)
    S = 10.0
    meshes = Meshes([grid_plane(wall_grid), grid_cube(box_grid), grid_plane(2)])
    white, red, green = (0.73, 0.73, 0.73), (0.65, 0.05, 0.05), (0.12, 0.45, 0.15)
    materials = [Material(white, 0.9, 0.0), Material(white, 0.5, 0.0), Material(red, 0.9, 0.0),
                 Material(green, 0.9, 0.0), Material((0.95, 0.95, 0.95), 0.04, 1.0),
                 Material((0.8, 0.6, 0.2), 0.2, 0.5), Material((0, 0, 0), 0.9, 0.0, (17.0, 14.0, 10.0))]
    m = Models()
    m.add(MeshTransform((0, 0, 0), IDENTITY_QUAT, (S, 1, S)), 0, 0)                     # floor, normal +y
    m.add(MeshTransform((0, S, 0), (pi, 0, 0), (S, 1, S)), 0, 0)                        # ceiling, normal -y
    m.add(MeshTransform((0, S / 2, -S / 2), (hp, 0, 0), (S, 1, S)), 0, 1)               # back, normal +z
    m.add(MeshTransform((0, S / 2, S / 2), (-hp, 0, 0), (S, 1, S)), 0, 0)               # front, normal -z
    m.add(MeshTransform((-S / 2, S / 2, 0), (0, 0, -hp), (S, 1, S)), 0, 2)              # left, normal +x
    m.add(MeshTransform((S / 2, S / 2, 0), (0, 0, hp), (S, 1, S)), 0, 3)                # right, normal -x
    m.add(MeshTransform((-1.8, 3.0, -1.8), (0, 0.3, 0), (3, 6, 3)), 1, 4)               # tall mirror box
    m.add(MeshTransform((1.9, 1.5, 1.0), (0, -0.3, 0), (3, 3, 3)), 1, 5)                # short glossy box
    m.add(MeshTransform((0, S - 0.01, 0), (pi, 0, 0), (3.5, 1, 3.5)), 2, 6)             # emitter, facing down
    camera = Camera((0.0, 5.0, 4.6), (0.0, float(pi), 0.0), F(np.radians(60.0)))
    return Scene(camera, meshes, materials, m, inflate_mode=1)


def dragon_scale_scene(target_tris=900_000, seed=1234, copies=1):
    """One displaced closed surface of about target_tris triangles (stand-in for a scanned mesh such as
    the Stanford dragon, which is not available offline) over a floor, lit by two emitter quads.
    copies > 1 instantiates the surface several times (multi-mesh, about copies*target_tris triangles)."""
    n = max(2, int(round(np.sqrt(target_tris / 12.0))))
    surface = displaced_sphere(n, seed)
    meshes = Meshes([surface, grid_plane(8), grid_plane(1)])
    materials = [Material((0.25, 0.25, 0.25), 0.9, 0.0), Material((0.9, 0.75, 0.4), 0.3, 0.6),
                 Material((0, 0, 0), 0.9, 0.0, (20.0, 18.0, 15.0)), Material((0, 0, 0), 0.9, 0.0, (6.0, 8.0, 12.0))]
    m = Models()
    side = int(np.ceil(np.sqrt(copies)))
    extent = 3.0 * side
    m.add(MeshTransform((0, 0, 0), IDENTITY_QUAT, (4 * extent, 1, 4 * extent)), 1, 0)   # floor
    rng = np.random.default_rng(seed + 1)
    for c in range(copies):
        cx = (c % side - (side - 1) / 2.0) * 3.0
        cz = (c // side - (side - 1) / 2.0) * 3.0
        rough = 0.3 if copies == 1 else float(rng.uniform(0.04, 0.1))
        if copies > 1:
            materials.append(Material(tuple(rng.uniform(0.4, 0.95, 3)), rough, float(rng.uniform(0.3, 1.0))))
            mat = len(materials) - 1
        else:
            mat = 1
        m.add(MeshTransform((cx, 1.45, cz), (0, float(rng.uniform(0, 6.28)) if copies > 1 else 0.4, 0),
                            (1.2, 1.2, 1.2)), 0, mat)
    hp = F(np.pi / 2)
    m.add(MeshTransform((0, 3.2 + extent, 0), (float(np.pi), 0, 0), (2 * extent, 1, 2 * extent)), 2, 2)  # top light
    m.add(MeshTransform((-2.5 * extent, 2.0, 0), (0, 0, -hp), (3, 1, 2 * extent)), 2, 3)                  # side light
    dist = 4.2 * side
    camera = Camera((0.9 * dist, 0.75 * dist + 1.0, dist), (0.55, float(np.pi) + 0.72, 0.0), F(np.radians(45.0)))
    return Scene(camera, meshes, materials, m, inflate_mode=1)


def multi_mesh_scene():
    """The multi-mesh benchmark scene: about 10 M triangles in eleven instances of the displaced surface, low-roughness GGX."""
    return dragon_scale_scene(900_000, copies=11)
