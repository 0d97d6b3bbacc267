"""The C-ABI library: loads, exports every symbol include/rtgl_amd.h declares, and fails loudly
(no CPU fallback) when no HIP device is present.  No compute calls are made here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rtgl_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtgl_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(rt):
    assert declared_symbols() == sorted(rt.host.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(rt):
    rt.host.build_library()
    lib = rt.host.load_library()
    for name in declared_symbols():
        assert hasattr(lib, name), f"librtgl_amd.so does not export {name}"


def test_struct_layouts_match_header(rt):
    # rtgl_frame_params: 17 uniforms = 26 four-byte fields; rtgl_counters: 8 x u64
    assert C.sizeof(rt.host.CFrameParams) == 26 * 4
    assert C.sizeof(rt.host.CCounters) == 64


def test_create_fails_loudly_without_a_device(rt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    lib = rt.host.load_library()
    h = C.c_void_p()
    rc = lib.rtgl_create(C.byref(h), 64, 64, 0)
    assert rc < 0 and not h.value
    assert b"no HIP device" in lib.rtgl_last_error(None) or b"hip" in lib.rtgl_last_error(None).lower()
    with pytest.raises(rt.host.RtglError):
        rt.host.Context(64, 64)
    rc = lib.rtgl_create_multi(C.byref(h), 64, 64, (C.c_int * 2)(0, 1), 2, 8)           # the multi-device form fails the same way
    assert rc < 0 and not h.value
    with pytest.raises(rt.host.RtglError):
        rt.host.Context(64, 64, devices=[0, 0])


def test_multi_device_arguments_are_checked(rt):
    lib = rt.host.load_library()
    h = C.c_void_p()
    assert lib.rtgl_create_multi(C.byref(h), 64, 64, None, 2, 8) < 0 and not h.value
    assert lib.rtgl_create_multi(C.byref(h), 64, 64, (C.c_int * 1)(0), 0, 8) < 0 and not h.value
    assert lib.rtgl_device_count(None) < 0


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under raytracer.glsl_amd/ or include/ may refer to it."""
    for base in ("raytracer.glsl_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "oracle/" not in text and "import oracle" not in text and "from oracle" not in text, f"{f} refers to oracle/"
