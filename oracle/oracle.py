"""ctypes front-ends of the TEST-ONLY checkers (see oracle/pathtrace_oracle.c, oracle/llvmpipe/lpgl.c).

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  Never the product.

  CpuOracle          the scalar/threaded C restatement of the reference path tracer
  LlvmpipeReference  runs the reference's OWN GLSL (read from a path at run time, never stored in
                     this repo) on Mesa llvmpipe; only usable where Mesa and /root/reference exist
                     (the build container), used to make tests/golden/ and to time the reference.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_ORACLE = os.path.join(HERE, "liboracle.so")
LIB_LPGL = os.path.join(HERE, "_ref", "liblpgl.so")
REFERENCE_SHADER = "/root/reference/shaders/raytracer.glsl"


def build(force: bool = False) -> None:
    """Compile the checkers with oracle/Makefile (gcc only)."""
    src = os.path.join(HERE, "pathtrace_oracle.c")
    if force or not os.path.exists(LIB_ORACLE) or os.path.getmtime(LIB_ORACLE) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, LIB_ORACLE])
    lsrc = os.path.join(HERE, "llvmpipe", "lpgl.c")
    if os.path.exists("/usr/include/GL/internal/dri_interface.h") and (
            force or not os.path.exists(LIB_LPGL) or os.path.getmtime(LIB_LPGL) < os.path.getmtime(lsrc)):
        subprocess.check_call(["make", "-s", "-C", HERE, "lpgl"])


class _Params(C.Structure):
    _fields_ = [("frames", C.c_int32), ("samples", C.c_uint32), ("max_bounce", C.c_uint32), ("time", C.c_float),
                ("background", C.c_float * 3), ("reset_flag", C.c_int32), ("use_envmap", C.c_int32),
                ("use_dof", C.c_int32), ("random", C.c_int32), ("camera_position", C.c_float * 3),
                ("camera_fov", C.c_float), ("camera_aperture", C.c_float), ("camera_focal_length", C.c_float),
                ("camera_forward", C.c_float * 3), ("camera_up", C.c_float * 3), ("camera_right", C.c_float * 3)]


class _Scene(C.Structure):
    _fields_ = [("spheres", C.c_void_p), ("n_spheres", C.c_uint32), ("materials", C.c_void_p), ("n_materials", C.c_uint32),
                ("meshes", C.c_void_p), ("n_meshes", C.c_uint32), ("vertices", C.c_void_p), ("n_vertices", C.c_uint32),
                ("nodes", C.c_void_p), ("n_nodes", C.c_uint32), ("env", C.c_void_p),
                ("env_w", C.c_int32), ("env_h", C.c_int32), ("env_channels", C.c_int32), ("env_faces", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("paths", "segments", "sphere_tests", "triangle_tests", "rand_calls", "env_lookups")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def _fill_params(p) -> _Params:
    cp = _Params()
    cp.frames, cp.samples, cp.max_bounce, cp.time = int(p.frames), int(p.samples), int(p.max_bounce), float(p.time)
    cp.background[:] = [float(x) for x in p.background]
    cp.reset_flag, cp.use_envmap, cp.use_dof, cp.random = int(p.reset_flag), int(p.use_envmap), int(p.use_dof), int(p.random)
    cp.camera_position[:] = [float(x) for x in p.camera_position]
    cp.camera_fov, cp.camera_aperture, cp.camera_focal_length = float(p.camera_fov), float(p.camera_aperture), float(p.camera_focal_length)
    cp.camera_forward[:] = [float(x) for x in p.camera_forward]
    cp.camera_up[:] = [float(x) for x in p.camera_up]
    cp.camera_right[:] = [float(x) for x in p.camera_right]
    return cp


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class CpuOracle:
    def __init__(self):
        build()
        self.lib = C.CDLL(LIB_ORACLE)
        self.lib.oracle_render.restype = C.c_int
        self.lib.oracle_render.argtypes = [C.POINTER(_Scene), C.POINTER(_Params), C.c_int, C.c_int, C.c_void_p,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(Counters)]

    def _scene(self, scene):
        keep = [np.ascontiguousarray(scene.spheres, np.float32), np.ascontiguousarray(scene.materials, np.float32),
                np.ascontiguousarray(scene.meshes, np.uint32), np.ascontiguousarray(scene.vertices, np.float32),
                np.ascontiguousarray(scene.nodes, np.float32),
                None if scene.env is None else np.ascontiguousarray(scene.env, np.uint8)]
        cs = _Scene()
        cs.spheres, cs.n_spheres = _ptr(keep[0]), keep[0].shape[0]
        cs.materials, cs.n_materials = _ptr(keep[1]), keep[1].shape[0]
        cs.meshes, cs.n_meshes = _ptr(keep[2]), keep[2].shape[0]
        cs.vertices, cs.n_vertices = _ptr(keep[3]), keep[3].shape[0]
        cs.nodes, cs.n_nodes = _ptr(keep[4]), keep[4].shape[0]
        if keep[5] is not None:
            cs.env = _ptr(keep[5])
            cs.env_faces, cs.env_h, cs.env_w, cs.env_channels = keep[5].shape
        return cs, keep

    def render(self, scene, params, image: np.ndarray, rect=None, threads: int = 1, want_seeds: bool = False):
        """One frame, in place, on `image` (H, W, 4) float32.  rect = (x0, y0, x1, y1); default is
        the reference's dispatch footprint (W//8*8, H//8*8).  Returns (counters dict, seeds|None)."""
        assert image.dtype == np.float32 and image.flags.c_contiguous and image.shape[2] == 4
        H, W = image.shape[:2]
        if rect is None:
            rect = (0, 0, W // 8 * 8, H // 8 * 8)
        cs, keep = self._scene(scene)
        cp = _fill_params(params)
        cnt = Counters()
        seeds = np.zeros((H, W, 4), np.uint32) if want_seeds else None
        rc = self.lib.oracle_render(C.byref(cs), C.byref(cp), W, H, _ptr(image), rect[0], rect[1], rect[2], rect[3],
                                    int(threads), _ptr(seeds) if want_seeds else None, C.byref(cnt))
        assert rc == 0
        del keep
        return cnt.as_dict(), seeds

    def sincos(self, a: np.ndarray):
        a = np.ascontiguousarray(a, np.float32)
        s = np.empty_like(a); c = np.empty_like(a)
        self.lib.oracle_sincos_array(_ptr(a), _ptr(s), _ptr(c), a.size)
        return s, c

    def env_lookup(self, scene, dirs: np.ndarray):
        dirs = np.ascontiguousarray(dirs, np.float32)
        cs, keep = self._scene(scene)
        out = np.zeros_like(dirs)
        self.lib.oracle_env_lookup_array(C.byref(cs), _ptr(dirs), _ptr(out), dirs.shape[0])
        return out

    def pcg4d(self, v4, rounds=1):
        v = np.array(v4, np.uint32)
        self.lib.oracle_pcg4d(_ptr(v), int(rounds))
        return v


class LlvmpipeReference:
    """The reference GLSL on Mesa llvmpipe.  Mirrors what src/renderer.cpp does around the
    dispatch: 5 SSBOs at bindings 1..5 (:89-93), 17 uniforms (:96-123), cube map on unit 0,
    RGBA32F image on image unit 0 (:129), glDispatchCompute(W/8, H/8, 1) (:131-134)."""

    @staticmethod
    def available() -> bool:
        return os.path.exists(LIB_LPGL) and os.path.exists(REFERENCE_SHADER)

    def __init__(self, shader_path: str = REFERENCE_SHADER):
        build()
        self.lib = L = C.CDLL(LIB_LPGL)
        L.lpgl_last_error.restype = C.c_char_p
        L.lpgl_info.restype = C.c_char_p
        L.lpgl_ssbo.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_long]
        L.lpgl_image_rgba32f.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.lpgl_read_image.argtypes = [C.c_int, C.c_void_p]
        L.lpgl_cubemap.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.lpgl_uniform1f.argtypes = [C.c_int, C.c_char_p, C.c_float]
        L.lpgl_uniform3f.argtypes = [C.c_int, C.c_char_p, C.c_float, C.c_float, C.c_float]
        L.lpgl_uniform1i.argtypes = [C.c_int, C.c_char_p, C.c_int]
        L.lpgl_uniform1ui.argtypes = [C.c_int, C.c_char_p, C.c_uint]
        if L.lpgl_init() != 0:
            raise RuntimeError("llvmpipe context: " + L.lpgl_last_error().decode())
        with open(shader_path, "rb") as f:
            src = f.read()
        log = C.create_string_buffer(16384)
        self.prog = L.lpgl_compute_program(src, log, len(log))
        if self.prog <= 0:
            raise RuntimeError("reference shader did not compile on llvmpipe: " + log.value.decode())
        self.version = L.lpgl_info(0).decode() + " / " + L.lpgl_info(1).decode()
        self._bufs = {}
        self._tex = None
        self._cube = None
        self._size = None

    def set_scene(self, scene):
        L = self.lib
        for binding, arr in ((1, scene.spheres), (2, scene.materials), (3, scene.meshes), (4, scene.vertices), (5, scene.nodes)):
            a = np.ascontiguousarray(arr)
            if binding in self._bufs:
                L.lpgl_delete_buffer(self._bufs[binding])
            self._bufs[binding] = L.lpgl_ssbo(0, binding, _ptr(a), a.nbytes)
        if self._cube is not None:
            L.lpgl_delete_texture(self._cube)
            self._cube = None
        if scene.env is not None:
            e = np.ascontiguousarray(scene.env, np.uint8)
            self._cube = L.lpgl_cubemap(_ptr(e), e.shape[0], e.shape[2], e.shape[1], e.shape[3])

    def set_image(self, image: np.ndarray):
        H, W = image.shape[:2]
        if self._tex is not None:
            self.lib.lpgl_delete_texture(self._tex)
        img = np.ascontiguousarray(image, np.float32)
        self._tex = self.lib.lpgl_image_rgba32f(W, H, _ptr(img))
        self._size = (W, H)

    def render(self, p):
        L, g = self.lib, self.prog
        L.lpgl_use(g)
        L.lpgl_uniform1f(g, b"u_time", p.time)
        L.lpgl_uniform1i(g, b"u_frames", int(p.frames))
        L.lpgl_uniform1ui(g, b"u_samples", int(p.samples))
        L.lpgl_uniform1ui(g, b"u_max_bounce", int(p.max_bounce))
        L.lpgl_uniform3f(g, b"u_background", *[float(x) for x in p.background])
        L.lpgl_uniform1i(g, b"u_random", int(p.random))
        L.lpgl_uniform1i(g, b"u_use_envmap", int(p.use_envmap))
        L.lpgl_uniform1i(g, b"u_use_dof", int(p.use_dof))
        L.lpgl_uniform3f(g, b"u_camera_position", *[float(x) for x in p.camera_position])
        L.lpgl_uniform1f(g, b"u_camera_fov", float(p.camera_fov))
        L.lpgl_uniform1f(g, b"u_camera_aperture", float(p.camera_aperture))
        L.lpgl_uniform1f(g, b"u_camera_focal_length", float(p.camera_focal_length))
        L.lpgl_uniform3f(g, b"u_camera_forward", *[float(x) for x in p.camera_forward])
        L.lpgl_uniform3f(g, b"u_camera_right", *[float(x) for x in p.camera_right])
        L.lpgl_uniform3f(g, b"u_camera_up", *[float(x) for x in p.camera_up])
        L.lpgl_uniform1i(g, b"u_reset_flag", int(p.reset_flag))
        L.lpgl_bind_image(self._tex)
        W, H = self._size
        L.lpgl_dispatch(W // 8, H // 8, 1)

    def read_image(self) -> np.ndarray:
        W, H = self._size
        out = np.zeros((H, W, 4), np.float32)
        self.lib.lpgl_read_image(self._tex, _ptr(out))
        return out
