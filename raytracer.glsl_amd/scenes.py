"""Synthetic, seeded scene generators for the BASELINE.json configurations.

Everything here is plain integer / IEEE + - * / arithmetic (no libm calls), so the byte blobs are
identical wherever they are generated: the oracle, the llvmpipe run of the reference shader and the
HIP renderer all see exactly the same bytes.

Buffer layouts are the reference's GPU layouts (reference
shaders/raytracer.glsl:11-60, src/renderer.h:22-96, src/kdtree.h:63-69):

    spheres   (n, 8)  float32   center.xyz, radius, material(int32 bits), pad x3      stride 32
    materials (n, 8)  float32   albedo.rgb, smoothness, emission.rgb, type(uint32)    stride 32
    meshes    (n, 4)  uint32    start, size, material, pad                            stride 16
    vertices  (nv, 4) float32   xyz, w = material id as float (3 consecutive = 1 tri) stride 16
    nodes     (n, 12) float32   min.xyzw, max.xyzw, left, right, offset, count (u32)  stride 48
    env       (6, H, W, C) uint8 faces +X,-X,+Y,-Y,+Z,-Z, row 0 = first image row
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
import numpy as np

INVALID = 0xFFFFFFFF
DIFFUSE, SPECULAR, TRANSMISSIVE = 0, 1, 2


# --------------------------------------------------------------------------- frame parameters

def radians_f32(deg: float) -> np.float32:
    """glm::radians on a float (reference src/renderer.cpp:116)."""
    return np.float32(deg) * np.float32(0.01745329251994329576923690768489)


@dataclass
class FrameParams:
    """The 17 uniforms of reference shaders/raytracer.glsl:62-81 (host side src/renderer.cpp:96-123).
    Defaults are the reference's (src/renderer.h:107-125,167-186, src/renderer.cpp:35)."""
    frames: int = 1
    samples: int = 1
    max_bounce: int = 5
    time: float = 0.0
    background: tuple = (0.52, 0.80, 0.92)
    reset_flag: int = 0
    use_envmap: int = 1
    use_dof: int = 1
    random: int = 0
    camera_position: tuple = (0.0, 0.0, -35.0)
    camera_fov: float = float(radians_f32(33.0))
    camera_aperture: float = 0.001
    camera_focal_length: float = 10.0
    camera_forward: tuple = (0.0, 0.0, 1.0)
    camera_up: tuple = (0.0, 1.0, 0.0)
    camera_right: tuple = (-1.0, 0.0, 0.0)

    def replace(self, **kw) -> "FrameParams":
        return replace(self, **kw)


@dataclass
class Scene:
    spheres: np.ndarray = field(default_factory=lambda: np.zeros((0, 8), np.float32))
    materials: np.ndarray = field(default_factory=lambda: np.zeros((0, 8), np.float32))
    meshes: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.uint32))
    vertices: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.float32))
    nodes: np.ndarray = field(default_factory=lambda: np.zeros((0, 12), np.float32))
    env: np.ndarray | None = None  # (6, H, W, C) uint8

    @property
    def n_triangles(self) -> int:
        return int(self.vertices.shape[0] // 3)


# --------------------------------------------------------------------------- glibc rand()

class GlibcRand:
    """glibc's rand() (TYPE_3 additive feedback generator), the source of u_random in the
    reference (src/main.cpp:207 srand(0); src/renderer.cpp:102 rand() once per frame).
    Restated from the published algorithm so frame sequences do not depend on the host libc."""

    def __init__(self, seed: int = 0):
        seed = seed & 0xFFFFFFFF
        if seed == 0:
            seed = 1
        r = [0] * 34
        r[0] = seed
        for i in range(1, 31):
            hi, lo = divmod(r[i - 1] if r[i - 1] < 0x80000000 else r[i - 1] - (1 << 32), 127773)
            word = 16807 * lo - 2836 * hi
            if word < 0:
                word += 2147483647
            r[i] = word
        for i in range(31, 34):
            r[i] = r[i - 31]
        self._r = [x & 0xFFFFFFFF for x in r]
        for _ in range(34, 344):
            self._next_raw()

    def _next_raw(self) -> int:
        r = self._r
        v = (r[-31] + r[-3]) & 0xFFFFFFFF
        r.append(v)
        del r[0]
        return v

    def rand(self) -> int:
        return self._next_raw() >> 1


# --------------------------------------------------------------------------- builders

def make_spheres(items) -> np.ndarray:
    """items: iterable of (cx, cy, cz, radius, material)."""
    items = list(items)
    out = np.zeros((len(items), 8), np.float32)
    for i, (x, y, z, r, m) in enumerate(items):
        out[i, 0:4] = (x, y, z, r)
        out[i, 4:5].view(np.int32)[0] = int(m)
    return out


def rgb(hex_: int):
    """gfx::rgb(hex) (reference src/gfx/util.h:11-21): [0,255] -> [0,1] by a float division."""
    c = np.array([(hex_ >> 16) & 0xFF, (hex_ >> 8) & 0xFF, hex_ & 0xFF], np.float32)
    return c / np.float32(255.0)


def make_materials(items) -> np.ndarray:
    """items: iterable of (albedo rgb, emission rgb, smoothness, type)."""
    items = list(items)
    out = np.zeros((len(items), 8), np.float32)
    for i, (alb, emi, smooth, typ) in enumerate(items):
        out[i, 0:3] = alb
        out[i, 3] = smooth
        out[i, 4:7] = emi
        out[i, 7:8].view(np.uint32)[0] = int(typ)
    return out


def make_meshes(items) -> np.ndarray:
    items = list(items)
    out = np.zeros((len(items), 4), np.uint32)
    for i, (start, size, mat) in enumerate(items):
        out[i, 0], out[i, 1] = start, size
        out[i, 2:3].view(np.int32)[0] = int(mat)
    return out


def make_nodes(items) -> np.ndarray:
    """items: iterable of (min xyz, max xyz, left, right, offset, count)."""
    items = list(items)
    out = np.zeros((len(items), 12), np.float32)
    u = out.view(np.uint32)
    for i, (mn, mx, left, right, offset, count) in enumerate(items):
        out[i, 0:3] = mn
        out[i, 4:7] = mx
        u[i, 8], u[i, 9], u[i, 10], u[i, 11] = left, right, offset, count
    return out


def single_leaf(n_spheres: int) -> np.ndarray:
    """One leaf node covering every sphere: the minimal node buffer that makes spheres visible
    (SURVEY.md A.9 item 14)."""
    return make_nodes([((-1e5,) * 3, (1e5,) * 3, INVALID, INVALID, 0, n_spheres)])


def demo_materials() -> np.ndarray:
    """The eight materials of the reference's first demo scene (src/main.cpp:66-75)."""
    z = (0.0, 0.0, 0.0)
    return make_materials([
        (rgb(0xAAAAAA), z, 0.0, DIFFUSE),
        (rgb(0xFFFFFF), rgb(0xFFFEFA) * np.float32(30.0), 0.0, DIFFUSE),
        (rgb(0xBC0000), z, 0.0, DIFFUSE),
        (rgb(0x00BC00), z, 0.0, DIFFUSE),
        (rgb(0xAAAAAA), z, 1.0, SPECULAR),
        (rgb(0xFFFFFF), z, 0.0, TRANSMISSIVE),
        (rgb(0xFF5733), z, 0.0, TRANSMISSIVE),
        (rgb(0xAAAAAA), z, 0.5, SPECULAR),
    ])


def demo_spheres(with_light: bool = True) -> np.ndarray:
    """Sphere set of the reference's first demo scene, non-Cornell variant (src/main.cpp:46-60)."""
    r, room, sr = 10000.0, 16.0, 4.0
    items = [(0.0, -(room + r), 0.0, r, 0)]
    if with_light:
        items.append((3.0, room + 10.0, 0.0, 3.0, 1))
    items += [(-18.0, -room + sr, 0.0, sr, 4), (-6.0, -room + sr, 0.0, sr, 5), (18.0, -room + sr, 0.0, sr, 7)]
    return make_spheres(items)


# --- libm-free smooth functions (Bhaskara-style rational approximations) used only to shape inputs

def _wave(x: np.ndarray) -> np.ndarray:
    """A smooth 2*pi-ish periodic bump built from + - * / only (period 6.0, range [-1, 1])."""
    x = np.asarray(x, np.float64)
    p = x - 6.0 * np.floor(x / 6.0)          # [0, 6)
    h = np.where(p < 3.0, p, p - 3.0)         # [0, 3)
    v = 16.0 * h * (3.0 - h) / (45.0 - 4.0 * h * (3.0 - h))
    return np.where(p < 3.0, v, -v)


def grid_mesh(nx: int, ny: int, materials=(0, 7, 5), x0=-20.0, x1=20.0, y0=-14.0, y1=6.0,
              z0=5.0, amp=2.0, facing_camera: bool = True) -> np.ndarray:
    """Height-field of nx*ny quads = 2*nx*ny triangles.  Triangles are wound so
    that their geometric normal points to -z (towards the default camera at z=-35); the reference's
    test accepts front faces only (shaders/raytracer.glsl:243-245).  w carries the material id,
    cycling per quad."""
    i = np.arange(nx + 1, dtype=np.float64)
    j = np.arange(ny + 1, dtype=np.float64)
    X = x0 + (x1 - x0) * i / nx
    Y = y0 + (y1 - y0) * j / ny
    Z = z0 + amp * _wave(0.3 * i)[:, None] * _wave(0.2 * j + 1.5)[None, :]
    P = np.zeros((nx + 1, ny + 1, 3), np.float64)
    P[..., 0] = X[:, None]
    P[..., 1] = Y[None, :]
    P[..., 2] = Z
    a = P[:-1, :-1]; b = P[1:, :-1]; c = P[1:, 1:]; d = P[:-1, 1:]
    # normal of (a, d, c) = (d-a) x (c-a) ~ (0,1,0) x (1,1,0) = (0,0,-1): faces -z
    t1 = np.stack([a, d, c], axis=2)
    t2 = np.stack([a, c, b], axis=2)
    if not facing_camera:
        t1 = t1[:, :, ::-1]
        t2 = t2[:, :, ::-1]
    tris = np.stack([t1, t2], axis=2).reshape(-1, 3, 3)      # (nx*ny*2, 3 verts, xyz)
    quad = np.repeat(np.arange(nx * ny), 2)
    mats = np.asarray(materials, np.float64)[quad % len(materials)]
    v = np.zeros((tris.shape[0], 3, 4), np.float32)
    v[..., :3] = tris.astype(np.float32)
    v[..., 3] = mats[:, None].astype(np.float32)
    return v.reshape(-1, 4)


def _pcg32_stream(n: int, seed: int) -> np.ndarray:
    """n 32-bit outputs of a PCG-XSH-RR generator (integer only)."""
    out = np.empty(n, np.uint32)
    state = np.uint64(seed * 2 + 1442695040888963407 & 0xFFFFFFFFFFFFFFFF)
    mult = np.uint64(6364136223846793005)
    inc = np.uint64(1442695040888963407)
    with np.errstate(over="ignore"):
        for k in range(n):
            old = state
            state = old * mult + inc
            xs = np.uint32(((old >> np.uint64(18)) ^ old) >> np.uint64(27))
            rot = np.uint32(old >> np.uint64(59))
            out[k] = (xs >> rot) | (xs << ((np.uint32(32) - rot) & np.uint32(31)))
    return out


def sky_cubemap(size: int = 256, channels: int = 4, seed: int = 1) -> np.ndarray:
    """Procedural environment: vertical sky gradient + a sun disc + low-amplitude hashed grain
    (so bilinear filtering is exercised), 6 faces of size x size, 8-bit."""
    faces = np.zeros((6, size, size, channels), np.uint8)
    k = (np.arange(size, dtype=np.float64) + 0.5) / size * 2.0 - 1.0
    sc, tc = np.meshgrid(k, k)            # sc varies along a row, tc down the rows
    one = np.ones_like(sc)
    # inverse of the GL face table (s,t) -> direction, for each face
    dirs = [(one, -tc, -sc), (-one, -tc, sc), (sc, one, tc), (sc, -one, -tc), (sc, -tc, one), (-sc, -tc, -one)]
    sun = np.array([0.35, 0.80, -0.45]); sun = sun / (sun @ sun) ** 0.5
    # coarse hashed grain: one value per 4x4 block, seeded
    g = max(size // 4, 1)
    grain = (_pcg32_stream(6 * g * g, seed) >> np.uint32(24)).astype(np.float64).reshape(6, g, g) / 255.0
    for f, (dx, dy, dz) in enumerate(dirs):
        inv = 1.0 / (dx * dx + dy * dy + dz * dz) ** 0.5
        ux, uy, uz = dx * inv, dy * inv, dz * inv
        h = 0.5 * (uy + 1.0)
        r = 0.85 - 0.55 * h; gr = 0.90 - 0.30 * h; b = 0.98 - 0.08 * h
        ground = uy < -0.02
        r = np.where(ground, 0.32 + 0.1 * ux, r); gr = np.where(ground, 0.30 + 0.1 * uz, gr); b = np.where(ground, 0.26, b)
        cs = ux * sun[0] + uy * sun[1] + uz * sun[2]
        disc = np.clip((cs - 0.985) / 0.01, 0.0, 1.0)
        r = r + disc * (1.0 - r); gr = gr + disc * (0.97 - gr); b = b + disc * (0.85 - b)
        gn = np.kron(grain[f], np.ones((size // g, size // g)))[:size, :size] - 0.5
        rgbf = np.stack([r + 0.06 * gn, gr + 0.06 * gn, b + 0.04 * gn], axis=-1)
        faces[f, :, :, :3] = np.clip(np.floor(rgbf * 255.0 + 0.5), 0, 255).astype(np.uint8)
        if channels == 4:
            faces[f, :, :, 3] = 255
    return faces


def noise_cubemap(size: int = 8, channels: int = 4, seed: int = 7) -> np.ndarray:
    """Tiny high-contrast random cube map: a hard test for face selection, edge clamping and the
    8-bit filter."""
    n = 6 * size * size * channels
    return (_pcg32_stream(n, seed) >> np.uint32(24)).astype(np.uint8).reshape(6, size, size, channels)


# --------------------------------------------------------------------------- named configurations

def scene_c1(with_light: bool = False) -> Scene:
    """BASELINE.json configs[0]: 4 spheres (ground, mirror, glass, half-rough; the emissive light
    of the reference demo scene is dropped, with_light=True restores it), lit by the background
    colour, one leaf node."""
    sp = demo_spheres(with_light)
    return Scene(spheres=sp, materials=demo_materials(), nodes=single_leaf(len(sp)))


def scene_mesh(nx: int, ny: int, env_size: int = 256, with_spheres: bool = True) -> Scene:
    """BASELINE.json configs[1..4]: the C1 spheres (with the light) + an nx*ny*2-triangle
    height-field + procedural cube map.  nx,ny = 100,50 -> 10,000 triangles; 250,200 -> 100,000."""
    sp = demo_spheres(True) if with_spheres else np.zeros((0, 8), np.float32)
    v = grid_mesh(nx, ny)
    return Scene(spheres=sp, materials=demo_materials(), meshes=make_meshes([(0, v.shape[0] // 3, 0)]),
                 vertices=v, nodes=single_leaf(len(sp)) if len(sp) else np.zeros((0, 12), np.float32),
                 env=sky_cubemap(env_size))


def params_c1() -> FrameParams:
    return FrameParams(max_bounce=5, use_envmap=0, use_dof=1)


def params_c2() -> FrameParams:
    return FrameParams(max_bounce=8, use_envmap=1, use_dof=1)


def params_c5() -> FrameParams:
    return FrameParams(max_bounce=16, use_envmap=1, use_dof=1, camera_aperture=0.5, camera_focal_length=40.0)


CONFIGS = {
    "C1": dict(width=256, height=256, scene=lambda: scene_c1(), params=params_c1),
    "C2": dict(width=1920, height=1080, scene=lambda: scene_mesh(100, 50), params=params_c2),
    "C4": dict(width=1920, height=1080, scene=lambda: scene_mesh(250, 200), params=params_c2),
    "C5": dict(width=3840, height=2160, scene=lambda: scene_mesh(100, 50), params=params_c5),
}
