"""raytracer_amd -- MI355X (gfx950) implementation of the RayTracer2 propagation loop of cericdahl/RayTracer.

Public surface mirrors the reference's Python modules for this path:
    raytracer_amd.RayTracer2.RayTracer2(...)   python/RayTracer2.py:185
    raytracer_amd.surface.surface              python/surface.py:51
    raytracer_amd.rayInterfaces.rayInterfaces  python/rayInterfaces.py:4
The arithmetic runs in hand-written HIP kernels behind the C ABI of include/rt_mi355x.h (librt_mi355x.so, loaded with
ctypes on first use).  There is no CPU path: tracing raises if the library or a GPU is missing.
"""
__all__ = ["RayTracer2", "surface", "rayInterfaces", "bounds", "table"]
