"""cuda-go-icp_amd: host-side mirror of the reference's operator interface over libgoicp_mi355.so.

The product is the HIP/C++ shared library (csrc/ -> libgoicp_mi355.so, C ABI in
include/goicp_mi355.h).  This package is plumbing: a ctypes binding (`binding`), Python mirrors of
the reference classes for tests/bench (`fgoicp`: Config, load_cloud, FastGoICP, Registration,
IterativeClosestPoint3D) and the one-process-per-GPU driver over torch.distributed
(`sharded`).  Nothing here computes on the CPU: every operator raises if the library or a GPU is
missing.
"""
from . import binding  # noqa: F401
from .binding import GoicpError, build_library, library_path, load_library  # noqa: F401
from .fgoicp import (Config, FastGoICP, IterativeClosestPoint3D, Registration, RotNode, TransNode,  # noqa: F401
                     load_cloud)
