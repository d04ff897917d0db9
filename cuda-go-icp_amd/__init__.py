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


def kernel_source_hash():
    """The hash goicp_kernel_source_hash() reports, recomputed from the sources in the tree (csrc/Makefile: sha256 over device.hip,
    bnbqueue.hip, kdbuild.hip, device.hpp, first 16 hex digits): a library that is out of date with its sources shows here."""
    import hashlib
    import os
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    for n in ("device.hip", "bnbqueue.hip", "kdbuild.hip", "device.hpp"):
        with open(os.path.join(d, n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]
