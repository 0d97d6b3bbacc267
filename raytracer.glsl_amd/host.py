"""Host-side Python binding of the C ABI (include/rtgl_amd.h) plus a headless mirror of the
reference's Renderer/Window frame loop.  Used by tests/, bench.py and __graft_entry__.py.

There is no CPU fallback: if librtgl_amd.so is missing or no HIP device is present, everything here
raises.  (The C++ facade with the reference's class names lives in include/rtgl/.)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

from .scenes import FrameParams, GlibcRand, Scene

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTGL_AMD_LIB") or os.path.join(PKG_DIR, "librtgl_amd.so")   # override: A/B of experimental builds
CSRC_DIR = os.path.join(PKG_DIR, "csrc")

KERNEL_MEGA, KERNEL_WAVEFRONT = 0, 1

# every symbol include/rtgl_amd.h declares
ABI_SYMBOLS = [
    "rtgl_create", "rtgl_create_tiled", "rtgl_destroy", "rtgl_last_error",
    "rtgl_upload_spheres", "rtgl_upload_materials", "rtgl_upload_meshes", "rtgl_upload_vertices",
    "rtgl_upload_nodes", "rtgl_upload_envmap", "rtgl_set_frame_params", "rtgl_render_frame",
    "rtgl_synchronize", "rtgl_read_image_f32", "rtgl_read_image_u8", "rtgl_write_image_f32",
    "rtgl_clear_image", "rtgl_local_rows", "rtgl_local_row_to_global", "rtgl_device_image",
    "rtgl_bind_device_image", "rtgl_set_stream", "rtgl_get_counters", "rtgl_read_rng_state",
    "rtgl_set_option", "rtgl_get_option", "rtgl_last_frame_ms", "rtgl_last_frame_timing",
    "rtgl_accumulated_timing", "rtgl_timing_reset", "rtgl_create_multi", "rtgl_device_count", "rtgl_gather_tiles",
]


class RtglError(RuntimeError):
    pass


class CFrameParams(C.Structure):
    """rtgl_frame_params"""
    _fields_ = [("frames", C.c_int32), ("samples", C.c_uint32), ("max_bounce", C.c_uint32), ("time", C.c_float),
                ("background", C.c_float * 3), ("reset_flag", C.c_int32), ("use_envmap", C.c_int32),
                ("use_dof", C.c_int32), ("random", C.c_int32), ("camera_position", C.c_float * 3),
                ("camera_fov", C.c_float), ("camera_aperture", C.c_float), ("camera_focal_length", C.c_float),
                ("camera_forward", C.c_float * 3), ("camera_up", C.c_float * 3), ("camera_right", C.c_float * 3)]


class CCounters(C.Structure):
    """rtgl_counters"""
    _fields_ = [("paths", C.c_uint64), ("segments", C.c_uint64), ("triangle_tests", C.c_uint64),
                ("candidates", C.c_uint64), ("env_lookups", C.c_uint64), ("culled_tests", C.c_uint64), ("reserved", C.c_uint64 * 2)]


class CFrameTiming(C.Structure):
    """rtgl_frame_timing"""
    _fields_ = [("frame_ms", C.c_float), ("intersect_ms", C.c_float), ("intersect_launches", C.c_uint32), ("reserved", C.c_uint32)]


def build_library(force: bool = False) -> str:
    """hipcc-compile the HIP kernels + C ABI for gfx950 into librtgl_amd.so (in-tree)."""
    srcs = [os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith((".hip", ".hpp", ".h"))]
    srcs.append(os.path.join(PKG_DIR, "..", "include", "rtgl_amd.h"))
    stale = not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        # -B: `force` really recompiles (a library newer than its sources would otherwise make `make all` a no-op and a
        # stale binary could travel to the GPU box)
        subprocess.check_call(["make", "-s", "-C", CSRC_DIR] + (["-B"] if force else []) + ["all"])
    return LIB_PATH


_lib = None


def load_library() -> C.CDLL:
    """dlopen the product library; raises if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtglError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc) first; there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, u32, i = C.c_void_p, C.c_uint32, C.c_int
    L.rtgl_create.argtypes = [C.POINTER(vp), i, i, i]
    L.rtgl_create_tiled.argtypes = [C.POINTER(vp), i, i, i, i, i, i]
    L.rtgl_create_multi.argtypes = [C.POINTER(vp), i, i, C.POINTER(i), i, i]
    L.rtgl_device_count.argtypes = [vp]
    L.rtgl_gather_tiles.argtypes = [vp]
    L.rtgl_destroy.argtypes = [vp]; L.rtgl_destroy.restype = None
    L.rtgl_last_error.argtypes = [vp]; L.rtgl_last_error.restype = C.c_char_p
    for name in ("rtgl_upload_spheres", "rtgl_upload_materials", "rtgl_upload_meshes", "rtgl_upload_vertices", "rtgl_upload_nodes"):
        getattr(L, name).argtypes = [vp, vp, u32]
    L.rtgl_upload_envmap.argtypes = [vp, vp, i, i, i, i]
    L.rtgl_set_frame_params.argtypes = [vp, C.POINTER(CFrameParams)]
    for name in ("rtgl_render_frame", "rtgl_synchronize", "rtgl_clear_image", "rtgl_local_rows"):
        getattr(L, name).argtypes = [vp]
    L.rtgl_read_image_f32.argtypes = [vp, vp]
    L.rtgl_read_image_u8.argtypes = [vp, vp, i]
    L.rtgl_write_image_f32.argtypes = [vp, vp]
    L.rtgl_local_row_to_global.argtypes = [vp, i]
    L.rtgl_device_image.argtypes = [vp]; L.rtgl_device_image.restype = vp
    L.rtgl_bind_device_image.argtypes = [vp, vp]
    L.rtgl_set_stream.argtypes = [vp, vp]
    L.rtgl_get_counters.argtypes = [vp, C.POINTER(CCounters)]
    L.rtgl_read_rng_state.argtypes = [vp, vp]
    L.rtgl_set_option.argtypes = [vp, C.c_char_p, i]
    L.rtgl_get_option.argtypes = [vp, C.c_char_p, C.POINTER(i)]
    L.rtgl_last_frame_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.rtgl_last_frame_timing.argtypes = [vp, C.POINTER(CFrameTiming)]
    L.rtgl_accumulated_timing.argtypes = [vp, C.POINTER(CFrameTiming), C.POINTER(C.c_uint32)]
    L.rtgl_timing_reset.argtypes = [vp]
    _lib = L
    return L


def to_c_params(p: FrameParams) -> CFrameParams:
    cp = CFrameParams()
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


class Context:
    """One rtgl_context.  rank/world/strip_rows select row-strip tiling for multi-GPU runs."""

    def __init__(self, width: int, height: int, device: int = 0, rank: int = 0, world: int = 1, strip_rows: int = 16, devices=None):
        """devices=[ordinals]: one single-process multi-device context (rtgl_create_multi) instead of a (rank, world) tile."""
        self.lib = load_library()
        self.width, self.height = int(width), int(height)
        self.rank, self.world, self.strip_rows = rank, world, strip_rows
        h = C.c_void_p()
        if devices is not None:
            self.rank, self.world = 0, 1
            arr = (C.c_int * len(devices))(*devices)
            rc = self.lib.rtgl_create_multi(C.byref(h), width, height, arr, len(devices), strip_rows)
        else:
            rc = self.lib.rtgl_create_tiled(C.byref(h), width, height, device, rank, world, strip_rows)
        if rc != 0:
            raise RtglError(f"rtgl_create failed ({rc}): {self.lib.rtgl_last_error(None).decode()}")
        self.h = h
        self.local_rows = self.lib.rtgl_local_rows(self.h)

    def _chk(self, rc):
        if rc != 0:
            raise RtglError(f"rtgl call failed ({rc}): {self.lib.rtgl_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.rtgl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- scene
    def upload_spheres(self, a):
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 8); self._chk(self.lib.rtgl_upload_spheres(self.h, _ptr(a), a.shape[0]))

    def upload_materials(self, a):
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 8); self._chk(self.lib.rtgl_upload_materials(self.h, _ptr(a), a.shape[0]))

    def upload_meshes(self, a):
        a = np.ascontiguousarray(a, np.uint32).reshape(-1, 4); self._chk(self.lib.rtgl_upload_meshes(self.h, _ptr(a), a.shape[0]))

    def upload_vertices(self, a):
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 4); self._chk(self.lib.rtgl_upload_vertices(self.h, _ptr(a), a.shape[0]))

    def upload_nodes(self, a):
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 12); self._chk(self.lib.rtgl_upload_nodes(self.h, _ptr(a), a.shape[0]))

    def upload_envmap(self, env):
        if env is None:
            return
        e = np.ascontiguousarray(env, np.uint8)
        self._chk(self.lib.rtgl_upload_envmap(self.h, _ptr(e), e.shape[0], e.shape[2], e.shape[1], e.shape[3]))

    def upload_scene(self, s: Scene):
        self.upload_spheres(s.spheres); self.upload_materials(s.materials); self.upload_meshes(s.meshes)
        self.upload_vertices(s.vertices); self.upload_nodes(s.nodes); self.upload_envmap(s.env)

    # --- frames
    def set_params(self, p: FrameParams):
        cp = to_c_params(p)
        self._chk(self.lib.rtgl_set_frame_params(self.h, C.byref(cp)))

    def render(self, p: FrameParams | None = None, sync: bool = True):
        if p is not None:
            self.set_params(p)
        self._chk(self.lib.rtgl_render_frame(self.h))
        if sync:
            self._chk(self.lib.rtgl_synchronize(self.h))

    def synchronize(self):
        self._chk(self.lib.rtgl_synchronize(self.h))

    def last_frame_ms(self) -> float:
        ms = C.c_float()
        self._chk(self.lib.rtgl_last_frame_ms(self.h, C.byref(ms)))
        return float(ms.value)

    def last_frame_timing(self) -> dict:
        t = CFrameTiming()
        self._chk(self.lib.rtgl_last_frame_timing(self.h, C.byref(t)))
        return dict(frame_ms=float(t.frame_ms), intersect_ms=float(t.intersect_ms), intersect_launches=int(t.intersect_launches))

    def accumulated_timing(self) -> dict:
        t, n = CFrameTiming(), C.c_uint32()
        self._chk(self.lib.rtgl_accumulated_timing(self.h, C.byref(t), C.byref(n)))
        return dict(frames=int(n.value), frame_ms=float(t.frame_ms), intersect_ms=float(t.intersect_ms), intersect_launches=int(t.intersect_launches))

    def timing_reset(self):
        self._chk(self.lib.rtgl_timing_reset(self.h))

    # --- image
    def read_image(self) -> np.ndarray:
        out = np.zeros((self.local_rows, self.width, 4), np.float32)
        self._chk(self.lib.rtgl_read_image_f32(self.h, _ptr(out)))
        return out

    def read_image_u8(self, flip: bool = False) -> np.ndarray:
        out = np.zeros((self.local_rows, self.width, 4), np.uint8)
        self._chk(self.lib.rtgl_read_image_u8(self.h, _ptr(out), int(flip)))
        return out

    def write_image(self, img: np.ndarray):
        img = np.ascontiguousarray(img, np.float32)
        assert img.shape == (self.local_rows, self.width, 4)
        self._chk(self.lib.rtgl_write_image_f32(self.h, _ptr(img)))

    def clear_image(self):
        self._chk(self.lib.rtgl_clear_image(self.h))

    def global_rows(self) -> np.ndarray:
        return np.array([self.lib.rtgl_local_row_to_global(self.h, r) for r in range(self.local_rows)], np.int64)

    def device_image_ptr(self) -> int:
        return int(self.lib.rtgl_device_image(self.h) or 0)

    def bind_device_image(self, ptr: int):
        self._chk(self.lib.rtgl_bind_device_image(self.h, C.c_void_p(ptr)))

    def set_stream(self, stream_handle: int):
        self._chk(self.lib.rtgl_set_stream(self.h, C.c_void_p(stream_handle)))

    # --- diagnostics
    def set_option(self, key: str, value: int):
        self._chk(self.lib.rtgl_set_option(self.h, key.encode(), int(value)))

    def get_option(self, key: str) -> int:
        v = C.c_int()
        self._chk(self.lib.rtgl_get_option(self.h, key.encode(), C.byref(v)))
        return int(v.value)

    def counters(self) -> dict:
        c = CCounters()
        self._chk(self.lib.rtgl_get_counters(self.h, C.byref(c)))
        return {k: int(getattr(c, k)) for k in ("paths", "segments", "triangle_tests", "candidates", "env_lookups", "culled_tests")}

    def read_rng_state(self) -> np.ndarray:
        out = np.zeros((self.local_rows, self.width, 4), np.uint32)
        self._chk(self.lib.rtgl_read_rng_state(self.h, _ptr(out)))
        return out


class FrameLoop:
    """Pure host logic of the reference's Window::run + Renderer::render frame bookkeeping (no GPU):
    m_frames is incremented BEFORE render (src/window.cpp:42), u_random = rand() once per frame after
    srand(0) (src/main.cpp:207, src/renderer.cpp:102), a reset uploads the stale frame count with
    u_reset_flag = 1 and then zeroes the count (src/renderer.cpp:98,123-127)."""

    def __init__(self, params: FrameParams | None = None, seed: int = 0):
        self.params = params or FrameParams()
        self.m_frames = 0
        self.m_reset = False
        self._rand = GlibcRand(seed)

    def reset_buffer(self):
        self.m_reset = True

    def next_frame(self) -> FrameParams:
        self.m_frames += 1
        p = self.params.replace(frames=self.m_frames, random=self._rand.rand(), reset_flag=int(self.m_reset))
        if self.m_reset:
            self.m_reset = False
            self.m_frames = 0
        return p


class HeadlessRenderer(FrameLoop):
    """FrameLoop driving a Context: the Python twin of include/rtgl/renderer.h's Renderer."""

    def __init__(self, width: int, height: int, device: int = 0, seed: int = 0, **tiling):
        super().__init__(seed=seed)
        self.ctx = Context(width, height, device, **tiling)

    def set_scene(self, scene: Scene):
        self.ctx.upload_scene(scene)

    def render_frame(self, sync: bool = True) -> FrameParams:
        p = self.next_frame()
        self.ctx.render(p, sync=sync)
        return p

    def run(self, frames: int):
        for _ in range(frames):
            self.render_frame(sync=False)
        self.ctx.synchronize()
