"""visualslam_amd -- MI355X-native feature-tracking front-end (Gaussian pyramid, Harris/FAST track,
pyramidal LK) behind the OpenVX / nvxcu C API the VisionWorks samples call.

The product is libvisualslam_amd.so (HIP, gfx950); this package holds its sources (csrc/), the
ctypes binding of the C ABI (nvxcu.py) and device-memory/synthetic-data plumbing for tests and bench.
"""
__version__ = "0.1.0"
