"""3d_sln_amd - MI355X (gfx950) native implementation of the 3D_SLN hot path.

The package name is not a Python identifier; import it with
``importlib.import_module("3d_sln_amd")`` (tests/conftest.py::pkg does that).

Layout
  csrc/   hand-written HIP kernels + the C ABI (include/sln_hip.h) -> libsln_hip.so
  host/   Python mirror of the reference's call surfaces (same class / method / state_dict names)
  _lib.py ctypes loader (no torch types cross the boundary)
"""
__version__ = "0.1.0"
