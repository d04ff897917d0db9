"""ctypes binding of libgoicp_mi355.so (include/goicp_mi355.h)."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK = 0
STATUS = {-1: "GOICP_ERR_INVALID", -2: "GOICP_ERR_IO", -3: "GOICP_ERR_CONFIG", -4: "GOICP_ERR_NO_DEVICE",
          -5: "GOICP_ERR_DEVICE", -6: "GOICP_ERR_INTERNAL", -7: "GOICP_ERR_TIMEOUT", -8: "GOICP_ERR_PEER"}
PATH_MAX = 1024


class GoicpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (STATUS.get(code, str(code)), msg))
        self.code = code


class CConfig(C.Structure):
    _fields_ = [("mode", C.c_int32), ("trim", C.c_int32), ("subsample", C.c_float), ("mse_threshold", C.c_float),
                ("resize", C.c_float), ("target", C.c_char * PATH_MAX), ("source", C.c_char * PATH_MAX),
                ("output", C.c_char * PATH_MAX), ("visualization", C.c_char * PATH_MAX),
                ("viz_theta", C.c_float), ("viz_phi", C.c_float), ("viz_spin_after_finish", C.c_int32),
                ("rot_min", C.c_float * 3), ("rot_max", C.c_float * 3), ("rot_search_depth", C.c_int32),
                ("trans_min", C.c_float * 3), ("trans_max", C.c_float * 3), ("trans_search_depth", C.c_int32),
                ("description", C.c_char * PATH_MAX), ("has_rotation_range", C.c_int32), ("has_translation_range", C.c_int32)]


class CParams(C.Structure):
    _fields_ = [("dt_size", C.c_int32), ("dt_expand", C.c_double), ("mse_threshold", C.c_float),
                ("dt_layout", C.c_int32), ("device", C.c_int32), ("trans_batch", C.c_int32),
                ("wide_children", C.c_int32), ("icp_max_iter", C.c_int32), ("verbose", C.c_int32),
                ("morton_sort", C.c_int32), ("rot_batch", C.c_int32), ("kd_gpu_build", C.c_int32), ("trim_fraction", C.c_float),
                ("use_rot_range", C.c_int32), ("use_trans_range", C.c_int32), ("rot_min", C.c_float * 3), ("rot_max", C.c_float * 3),
                ("trans_min", C.c_float * 3), ("trans_max", C.c_float * 3), ("rot_search_depth", C.c_int32), ("trans_search_depth", C.c_int32),
                ("icp_fused", C.c_int32), ("bounds_fp16", C.c_int32), ("icp_nn_cache", C.c_int32), ("flow", C.c_int32), ("adaptive_k", C.c_int32), ("queue_cap", C.c_int32), ("device_queues", C.c_int32),
                ("lds_tiles", C.c_int32), ("tile_spread_vox", C.c_float), ("tile_min", C.c_int32), ("stale_widen", C.c_int32), ("ub_tiebreak", C.c_int32), ("icp_point_seed", C.c_int32), ("ub_share", C.c_float), ("twin_fusion", C.c_int32), ("sort_items", C.c_int32), ("stale_compact", C.c_int32), ("stream_priority", C.c_int32), ("lanes", C.c_int32), ("lane_min_searches", C.c_int32)]


class CCube(C.Structure):
    _fields_ = [("tx", C.c_float), ("ty", C.c_float), ("tz", C.c_float), ("delta", C.c_float),
                ("coeff", C.c_float), ("rot", C.c_int32)]


class CCounters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("rot_pops", "trans_pops", "cubes", "inner_calls", "icp_runs", "icp_iters",
                                         "bounds_launches", "queue_fallbacks", "tile_expansions", "lane_batches")]


class CResult(C.Structure):
    _fields_ = [("optR", C.c_float * 9), ("optT", C.c_float * 3), ("curR", C.c_float * 9), ("curT", C.c_float * 3),
                ("best_sse", C.c_float), ("finished", C.c_int32), ("counters", CCounters),
                ("dt_build_ms", C.c_double), ("register_ms", C.c_double)]


class CStepStatus(C.Structure):
    _fields_ = [("finished", C.c_int32), ("early_exit", C.c_int32), ("best_sse", C.c_float),
                ("frontier_lb", C.c_float), ("rot_pops", C.c_int64)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t)
BCAST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32)


class CCommOps(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int32), ("world", C.c_int32), ("allreduce_min_u64", ALLREDUCE_FN), ("bcast", BCAST_FN)]


class CShardStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("steps", "exchanges", "broadcasts", "donations", "donated_cubes", "steps_idle")] + \
               [("wait_ms", C.c_double), ("step_ms", C.c_double), ("best_sse", C.c_float), ("failed_rank", C.c_int32)]


class CShardOptions(C.Structure):
    _fields_ = [("rot_pops_per_step", C.c_int32), ("rebalance", C.c_int32), ("stale_exchange", C.c_int32), ("ramp_to", C.c_int32)]


_fpp = C.POINTER(C.c_float)
ENG_BEGIN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32)
ENG_STEP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(CStepStatus))
ENG_POSE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _fpp, _fpp, _fpp)
ENG_OFFER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_float, _fpp, _fpp)
ENG_QSIZE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32))
ENG_DONATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, _fpp, C.POINTER(C.c_int32))
ENG_RECEIVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _fpp, C.c_int32)
ENG_END_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class CShardEngineOps(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("sse_threshold", C.c_float), ("begin", ENG_BEGIN_FN), ("step", ENG_STEP_FN), ("pose", ENG_POSE_FN),
                ("offer", ENG_OFFER_FN), ("queue_size", ENG_QSIZE_FN), ("donate", ENG_DONATE_FN), ("receive", ENG_RECEIVE_FN), ("end", ENG_END_FN)]


# every symbol include/goicp_mi355.h declares: name -> (restype, argtypes)
_fp, _vp = C.POINTER(C.c_float), C.c_void_p
SYMBOLS = {
    "goicp_last_error": (C.c_char_p, []),
    "goicp_abi_version": (C.c_int, []),
    "goicp_kernel_source_hash": (C.c_char_p, []),
    "goicp_config_load": (C.c_int, [C.c_char_p, C.POINTER(CConfig)]),
    "goicp_cloud_load": (C.c_int, [C.c_char_p, C.c_float, C.c_float, C.c_uint64, C.POINTER(_fp), C.POINTER(C.c_size_t)]),
    "goicp_cloud_free": (None, [_fp]),
    "goicp_params_default": (None, [C.POINTER(CParams)]),
    "goicp_params_from_config": (None, [C.POINTER(CConfig), C.POINTER(CParams)]),
    "goicp_thresholds": (C.c_int, [_vp, _fp, C.POINTER(C.c_int32)]),
    "goicp_device": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "goicp_set_progress_callback": (C.c_int, [_vp, C.c_void_p, C.c_void_p]),
    "goicp_probe_gather": (C.c_int, [_vp, C.c_int32, C.c_size_t, C.POINTER(C.c_double)]),
    "goicp_debug_kabsch": (C.c_int, [_fp, _fp]),
    "goicp_debug_queue_expand": (C.c_int, [C.c_void_p, _fp, C.c_int32, _fp, C.c_int32, _fp, _fp, _fp, _fp, C.POINTER(C.c_int32)]),
    "goicp_debug_bounds_tile": (C.c_int, [C.c_void_p, _fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _fp, _fp, _fp, _fp, _fp, C.POINTER(C.c_uint32)]),
    "goicp_debug_cache_hits": (C.c_int, [_vp, _fp, _fp, C.POINTER(C.c_int64)]),
    "goicp_create": (C.c_int, [C.POINTER(CParams), _fp, C.c_size_t, _fp, C.c_size_t, C.POINTER(_vp)]),
    "goicp_destroy": (C.c_int, [_vp]),
    "goicp_dt_info": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "goicp_dt_download": (C.c_int, [_vp, _fp]),
    "goicp_eval_bounds": (C.c_int, [_vp, _fp, _fp, C.c_size_t, C.c_int32, _fp, _fp]),
    "goicp_eval_bounds_batch": (C.c_int, [_vp, _fp, C.c_size_t, C.POINTER(CCube), C.c_size_t, _fp, _fp]),
    "goicp_eval_bounds_device": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "goicp_eval_bounds_device_grouped": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, _vp]),
    "goicp_time_bounds_device_grouped": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, C.c_int32, _fp]),
    "goicp_reduce_min_device": (C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "goicp_time_bounds_device": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp, C.c_int32, _fp]),
    "goicp_rot_coeff": (C.c_float, [_vp, C.c_int32]),
    "goicp_trans_delta": (C.c_float, [C.c_float]),
    "goicp_rodrigues": (None, [_fp, _fp]),
    "goicp_eval_sse": (C.c_int, [_vp, _fp, _fp, _fp]),
    "goicp_inner_bnb": (C.c_int, [_vp, _fp, C.c_int32, C.c_float, _fp, _fp, C.POINTER(CCounters)]),
    "goicp_icp_run": (C.c_int, [_vp, _fp, _fp, C.c_int32, C.c_float, _fp, C.POINTER(C.c_int32)]),
    "goicp_time_icp_pass": (C.c_int, [_vp, _fp, _fp, C.c_int32, _fp]),
    "goicp_time_icp_pass_cached": (C.c_int, [_vp, _fp, _fp, C.c_int32, _fp]),
    "goicp_nn_query": (C.c_int, [_vp, _fp, C.c_size_t, C.POINTER(C.c_int32), _fp]),
    "goicp_icp_step": (C.c_int, [_vp]),
    "goicp_register": (C.c_int, [_vp]),
    "goicp_cancel": (C.c_int, [_vp]),
    "goicp_poll": (C.c_int, [_vp, C.POINTER(CResult)]),
    "goicp_result_write_toml": (C.c_int, [_vp, C.c_char_p]),
    "goicp_result_write_ply": (C.c_int, [_vp, C.c_char_p]),
    "goicp_transform_source": (C.c_int, [_vp, _fp, _fp, _fp]),
    "goicp_set_shard": (C.c_int, [_vp, C.c_int32, C.c_int32]),
    "goicp_register_begin": (C.c_int, [_vp]),
    "goicp_register_step": (C.c_int, [_vp, C.c_int32, C.POINTER(CStepStatus)]),
    "goicp_offer_best": (C.c_int, [_vp, C.c_float, _fp, _fp]),
    "goicp_register_end": (C.c_int, [_vp]),
    "goicp_run_sharded": (C.c_int, [C.POINTER(CShardEngineOps), C.POINTER(CCommOps), C.c_int32, C.c_int32, C.POINTER(CShardStats)]),
    "goicp_register_sharded": (C.c_int, [_vp, C.POINTER(CCommOps), C.c_int32, C.c_int32, C.POINTER(CShardStats)]),
    "goicp_shard_options_default": (None, [C.POINTER(CShardOptions)]),
    "goicp_run_sharded_opt": (C.c_int, [C.POINTER(CShardEngineOps), C.POINTER(CCommOps), C.POINTER(CShardOptions), C.POINTER(CShardStats)]),
    "goicp_register_sharded_opt": (C.c_int, [_vp, C.POINTER(CCommOps), C.POINTER(CShardOptions), C.POINTER(CShardStats)]),
    "goicp_comm_set_timeout_ms": (C.c_int, [C.POINTER(CCommOps), C.c_int32]),
    "goicp_thread_comm_create": (C.c_int, [C.c_int32, C.POINTER(CCommOps)]),
    "goicp_thread_comm_destroy": (C.c_int, [C.POINTER(CCommOps)]),
    "goicp_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "goicp_rccl_comm_create": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(CCommOps)]),
    "goicp_rccl_comm_wrap": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(CCommOps)]),
    "goicp_rccl_comm_destroy": (C.c_int, [C.POINTER(CCommOps)]),
    "goicp_register_multi_gpu": (C.c_int, [C.POINTER(CParams), _fp, C.c_size_t, _fp, C.c_size_t, C.c_int32, C.c_int32, C.POINTER(_vp), C.POINTER(CShardStats)]),
}


def library_path():
    # GOICP_LIBRARY: experiments load a variant build (tools/); the product and the tests load the in-tree library
    return os.environ.get("GOICP_LIBRARY") or os.path.join(HERE, "libgoicp_mi355.so")


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of csrc/ (cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", os.path.join(HERE, "csrc"), "all"]
    if force:
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "csrc"), "clean"])
    subprocess.check_call(args)
    return library_path()


def load_library():
    """Load the HIP library.  No fallback: a missing library is an error."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise GoicpError(-6, "libgoicp_mi355.so is not built (run __graft_entry__.build() or make -C cuda-go-icp_amd/csrc)")
        lib = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)      # AttributeError if the ABI and the header disagree
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def check(code):
    if code != OK:
        raise GoicpError(code, load_library().goicp_last_error().decode("utf-8", "replace"))
