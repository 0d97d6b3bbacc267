"""MI355X-native progressive path tracer: HIP kernels + C ABI (csrc/), the host-side mirror of the
reference's Renderer/Window interface (include/, host.py), framebuffer tiling (tiling.py) and the
synthetic scene generators (scenes.py)."""
from . import scenes  # noqa: F401
from . import host  # noqa: F401
from . import tiling  # noqa: F401

__all__ = ["scenes", "host", "tiling"]
