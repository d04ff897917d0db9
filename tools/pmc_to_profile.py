#!/usr/bin/env python3
"""Turn the raw per-dispatch counter averages of tools/pmc_collect.sh into the summary committed under profiles/.
usage: pmc_to_profile.py bounds <pmc outdir> <workload text> > profiles/rNN_pmc_bounds_<w>.json
       pmc_to_profile.py icp    <pmc outdir> <workload text> > profiles/rNN_pmc_icp.json
Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes (HBM / rocprofv3 section): separate --pmc passes; FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-byte request, so fabric-side read bytes = 2 x FETCH_SIZE x 1024."""
import json
import os
import subprocess
import sys

mode, out, workload = sys.argv[1], sys.argv[2], sys.argv[3]
here = os.path.dirname(os.path.abspath(__file__))
want = "bounds_kernel" if mode == "bounds" else "icp_"
raw = json.loads(subprocess.check_output([sys.executable, os.path.join(here, "pmc_summary.py"), out, want, "--largest"]))


def traffic(e):
    f, w = e.get("FETCH_SIZE", 0.0) * 1024.0, e.get("WRITE_SIZE", 0.0) * 1024.0
    return f + w, 2.0 * f + w


if mode == "bounds":
    k = [n for n in raw if "bounds_kernel" in n][0]
    e = raw[k]
    rawb, corr = traffic(e)
    hit, miss = e.get("TCC_HIT_sum", 0.0), e.get("TCC_MISS_sum", 0.0)
    res = {"kernel": k, "workload": workload, "dispatches_averaged": e["dispatches"], "avg_duration_ns_under_pmc": e["avg_duration_ns_under_pmc"],
           "FETCH_SIZE_KB_per_launch": e.get("FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": e.get("WRITE_SIZE"), "TCC_HIT_sum": hit, "TCC_MISS_sum": miss,
           "l2_hit_rate": hit / max(hit + miss, 1.0), "hbm_bytes_per_launch_raw": rawb, "hbm_bytes_per_launch_corrected": corr,
           "l2_miss_bytes_per_launch_128B_lines": miss * 128.0}
    for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
              "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TA_TA_BUSY_sum", "GRBM_GUI_ACTIVE"):
        res[c] = e.get(c)
    dur_s = e["avg_duration_ns_under_pmc"] * 1e-9
    res["derived"] = {"fabric_GBs": corr / dur_s / 1e9, "hbm_frac_of_8TBs": corr / dur_s / 8e12,
                      "TA_busy_frac": (e.get("TA_TA_BUSY_sum") or 0.0) / 256.0 / max((e.get("GRBM_GUI_ACTIVE") or 8.0) / 8.0, 1.0),   # 256 texture-address units; GRBM_GUI_ACTIVE is summed over the 8 XCDs
                      "tag_accesses_per_gather": (e.get("TCP_TOTAL_CACHE_ACCESSES_sum") or 0.0) / max(e.get("SQ_INSTS_VMEM_RD") or 1.0, 1.0),
                      "valu_per_wave": (e.get("SQ_INSTS_VALU") or 0.0) / max(e.get("SQ_WAVES") or 1.0, 1.0)}
    res["commands"] = ["tools/pmc_collect.sh <out> bounds_kernel <bench.py args>   (five rocprofv3 --pmc passes of bench.py, counters listed in the script)",
                       "python tools/pmc_to_profile.py bounds <out> '<workload>'"]
    res["note"] = ("separate --pmc passes (FETCH_SIZE needs 3 of the 4 TCC slots). gfx950: FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM section), "
                   "so the corrected figure doubles it; TCC_MISS_sum x 128 B is the independent cross-check.  Fabric side of L2: Infinity-Cache hits included, an upper bound on HBM bytes.")
else:
    res = {"workload": workload, "commands": ["tools/pmc_collect.sh <out> icp_ --steps 1 --warmup 0 --prewarm 0 --sustain-s 0 --s2-steps 0 --no-cpu --no-e2e --no-probe",
                                              "python tools/pmc_to_profile.py icp <out> '<workload>'"]}
    tot_corr = tot_miss = 0.0
    for k, e in raw.items():
        if "icp_pass_kernel" in k or "icp_finalize" in k:
            res[k] = e
            _, corr = traffic(e)
            tot_corr += corr
            tot_miss += e.get("TCC_MISS_sum", 0.0) * 128.0
            if "icp_pass_kernel" in k:
                w = max(e.get("SQ_WAVES") or 1.0, 1.0)
                res["pass_per_wave"] = {c: (e.get(c) or 0.0) / w for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")}
    res["hbm_bytes_per_pass_corrected"] = tot_corr
    res["l2_miss_bytes_per_pass_128B_lines"] = tot_miss
# provenance: the kernels these counters were collected on.  kernel_source_hash = what the loaded library reports (csrc/Makefile: sha256 over
# the kernel sources); git_head = GOICP_GIT_HEAD of the collecting command (the GPU box has no .git; tools/profile_round.sh passes it on).
# bench.py quotes `traffic` from a profile only when its hash equals the library's.
sys.path.insert(0, os.path.dirname(here))
from __graft_entry__ import _pkg  # noqa: E402
_p = _pkg()
res["kernel_source_hash"] = _p.load_library().goicp_kernel_source_hash().decode()
res["kernel_source_hash_of_tree"] = _p.kernel_source_hash()
res["git_head"] = os.environ.get("GOICP_GIT_HEAD", "unknown")
print(json.dumps(res, indent=1))
