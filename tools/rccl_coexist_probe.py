#!/usr/bin/env python3
"""Which RCCL does the library talk to inside a torch process (bench.py at N > 1)?  torch.distributed (backend nccl) and the library's own
communicator (csrc/rccl_comm.cpp) in ONE process at world 1: an all-reduce through each, then the librccl objects mapped into the process.
On this image: ONE copy -- torch's bundled librccl.so (soname librccl.so.1) satisfies the library's DT_NEEDED, both communicators live in it."""
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(4, device="cuda"); dist.all_reduce(x); torch.cuda.synchronize()
from __graft_entry__ import _pkg
pkg = _pkg(); lib = pkg.load_library()
from cuda_go_icp_amd import binding as B
ident = C.create_string_buffer(128)
B.check(lib.goicp_rccl_unique_id(ident))
comm = B.CCommOps()
B.check(lib.goicp_rccl_comm_create(ident, 0, 1, 0, C.byref(comm)))
w = (C.c_uint64 * 6)(5, 4, 3, 2, 1, 0)
print("allreduce rc", comm.allreduce_min_u64(comm.ctx, w, 6), list(w))
print([l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l and "r-xp" in l])
print("torch nccl version", torch.cuda.nccl.version())
B.check(lib.goicp_rccl_comm_destroy(C.byref(comm)))
dist.destroy_process_group()
print("ok")
