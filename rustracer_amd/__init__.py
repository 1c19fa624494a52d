"""rustracer_amd — MI355X-native wavefront path tracer behind rustracer's `renderer::render` boundary.

Layout: `csrc/` HIP kernels + C-ABI (`librtx_hip.so`, include/rtx_hip.h) and the C++
host layer (`librtx_host.so`, include/rtx_host.h: SAH BVH build, flattening, camera/film set-up);
`host.py` ctypes mirror of the reference's render interface; `scenes/` synthetic scene generators.
"""
from .scene_desc import SceneDesc  # noqa: F401
