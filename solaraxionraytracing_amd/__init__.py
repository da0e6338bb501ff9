"""solaraxionraytracing_amd — MI355X-native Monte-Carlo ray tracer for the per-ray hot path of
jovoy/SolarAxionRayTracing (``traceAxion``), behind the C-ABI of include/sart.h.

Python here is plumbing (ctypes + numpy); the product is libsart.so (hand-written HIP for gfx950) and
libsart_host.so (C++ host mirror of the reference's setup/driver layer).
"""
from . import _lib, tables  # noqa: F401
from .raytracer import (FullRaytraceSetup, RayTracer, accumulator_len, angular_scan_len, calculateFluxFractions,  # noqa: F401
                        initFullSetup, mass_scan_len, newFullSetup, performAngularScan, performAxionMassScan,
                        performAxionMassScanHostLoop, split_angular_scan, split_mass_scan)

__all__ = ["FullRaytraceSetup", "RayTracer", "accumulator_len", "angular_scan_len", "calculateFluxFractions", "initFullSetup",
           "mass_scan_len", "newFullSetup", "performAngularScan", "performAxionMassScan", "performAxionMassScanHostLoop",
           "split_angular_scan", "split_mass_scan", "tables"]
