"""raytracing-rust_amd: MI355X (gfx950) back end for the path-tracing hot path of
nonl4331/raytracing-rust -- BVH traversal, ray/sphere + ray/triangle intersection and the
per-bounce shade/sample step -- behind a C ABI (include/rt_hip.h, csrc/ -> librt_hip.so).

Python here is host-side plumbing only (ctypes bindings, the `.ssml` scene reader, the
multi-GPU shard/gather driver).  The directory name carries a hyphen, so load it with
`importlib.import_module("raytracing-rust_amd")`.
"""
from . import abi  # noqa: F401
from .scene import SceneDescription  # noqa: F401
from . import ssml  # noqa: F401

__all__ = ["abi", "SceneDescription", "ssml"]
