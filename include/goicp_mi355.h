/* goicp_mi355.h -- C ABI of libgoicp_mi355.so, the MI355X-native (gfx950, HIP) Go-ICP engine.
 *
 * Drop-in boundary for the hot path of zjsun1017/CUDA-Go-ICP: BnB cube-bound evaluation, the
 * inner ICP loop and the rotation-cube search.  Every entry point names the reference interface
 * it replaces (paths relative to the reference checkout).  Plain pointers and sizes only; all
 * matrices are row-major float[9]; clouds are packed float xyz triples.
 *
 * Conventions
 *   - every function returns GOICP_OK (0) or a negative goicp_status; the message of the last
 *     failure on the calling thread is available from goicp_last_error().
 *   - handles are opaque; one handle per host thread, except goicp_poll()/goicp_cancel(), which
 *     are safe concurrently with goicp_register() (the reference's viewer polls the worker,
 *     src/goicp_kernel.cu:161-177).
 *   - output buffers are caller-owned unless a matching *_free() exists.
 *   - there is NO CPU fallback: goicp_create() fails with GOICP_ERR_NO_DEVICE without a GPU.
 *   - numerical contract: per-point arithmetic is bit-identical to the reference CPU path
 *     (src/goicp/); sums are tree reductions, so bounds agree to ~1e-5 relative.
 */
#ifndef GOICP_MI355_H
#define GOICP_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GOICP_ABI_VERSION 4

typedef enum goicp_status {
	GOICP_OK = 0,
	GOICP_ERR_INVALID = -1,    /* bad argument */
	GOICP_ERR_IO = -2,         /* file missing / unreadable / malformed  (reference: std::runtime_error, src/common.cpp:139-142,196-199,210,226) */
	GOICP_ERR_CONFIG = -3,     /* TOML parse error / missing info.description (src/common.cpp:28-40) */
	GOICP_ERR_NO_DEVICE = -4,  /* no HIP device: the product never computes on the CPU */
	GOICP_ERR_DEVICE = -5,     /* HIP runtime failure (reference: exit(EXIT_FAILURE), src/kernel.cu:29-38) */
	GOICP_ERR_INTERNAL = -6,
	GOICP_ERR_TIMEOUT = -7,    /* a collective of the sharded search did not complete within the communicator's deadline: the
	                              process must exit non-zero (a rank is lost); never re-execute */
	GOICP_ERR_PEER = -8        /* another rank of the sharded search reported a failure (goicp_shard_stats.failed_rank) */
} goicp_status;

const char* goicp_last_error(void);
int goicp_abi_version(void);
/* first 16 hex digits of the SHA-256 over the kernel sources (csrc/device.hip, bnbqueue.hip, kdbuild.hip, device.hpp, in that order)
 * this library was built from -- measurement hygiene: counter profiles under profiles/ carry the hash they were collected on */
const char* goicp_kernel_source_hash(void);

/* ---------------------------------------------------------------------------------------------
 * Config  -- replaces class Config (src/common.h:133-180, src/common.cpp:12-77).
 * Same keys, defaults and clamps.  Keys the reference declares but never parses
 * ([params.rotation], [params.translation], search_depth: src/common.h:157-169) are read here too.
 * ------------------------------------------------------------------------------------------- */
#define GOICP_PATH_MAX 1024
typedef struct goicp_config {
	int32_t mode;                 /* 0 ICP CPU, 1 ICP GPU, 2 ICP k-d tree GPU, 3 Go-ICP CPU, 4 Go-ICP GPU (src/common.h:7-11) */
	int32_t trim;
	float subsample;              /* clamped to [0,1]      (src/common.cpp:63) */
	float mse_threshold;          /* clamped to >= 1e-10   (src/common.cpp:64) */
	float resize;
	char target[GOICP_PATH_MAX];
	char source[GOICP_PATH_MAX];
	char output[GOICP_PATH_MAX];
	char visualization[GOICP_PATH_MAX];
	float viz_theta, viz_phi;
	int32_t viz_spin_after_finish;
	float rot_min[3], rot_max[3];     /* degrees */
	int32_t rot_search_depth;
	float trans_min[3], trans_max[3];
	int32_t trans_search_depth;
	char description[GOICP_PATH_MAX];
	int32_t has_rotation_range;       /* the TOML carries a [params.rotation] table (test/skull_goicp.toml:22-30) */
	int32_t has_translation_range;    /* ... a [params.translation] table (test/skull_goicp.toml:32-41) */
} goicp_config;

int goicp_config_load(const char* toml_path, goicp_config* out);

/* ---------------------------------------------------------------------------------------------
 * Cloud loader -- replaces load_cloud / load_cloud_ply / load_cloud_txt (src/common.cpp:79-228):
 * ".ply" (ascii or binary_little_endian, element vertex with float x,y,z, other properties and
 * elements skipped) or ".txt" ("N" then N lines "x y z").  Points are multiplied by `resize`;
 * `subsample` keeps each point with that probability, capped at floor(N*subsample).
 * seed = 0 reproduces the reference's non-deterministic std::random_device seeding.
 * ------------------------------------------------------------------------------------------- */
int goicp_cloud_load(const char* path, float subsample, float resize, uint64_t seed, float** xyz, size_t* n);
void goicp_cloud_free(float* xyz);

/* ---------------------------------------------------------------------------------------------
 * Engine -- replaces icp::FastGoICP + icp::Registration + icp::IterativeClosestPoint3D
 * (src/fgoicp/fgoicp.hpp:11-69, registration.hpp:44-98, icp3d.hpp:9-41) with the CPU path's
 * semantics (GoICP, src/goicp/jly_goicp.h:84-142).
 * ------------------------------------------------------------------------------------------- */
typedef struct goicp_engine* goicp_handle;

typedef struct goicp_params {
	int32_t dt_size;         /* DT grid side, reference 300 (src/goicp/jly_goicp.cpp:56) */
	double dt_expand;        /* bbox expand factor, reference 2.0 (src/goicp/jly_goicp.cpp:57) */
	float mse_threshold;     /* SSE threshold = mse_threshold * N (src/goicp/jly_goicp.cpp:208, src/fgoicp/fgoicp.hpp:23) */
	int32_t dt_layout;       /* 0 linear [z][y][x]; 1 bricked 4x4x4 (default) */
	int32_t device;          /* HIP device ordinal, -1 = current */
	int32_t trans_batch;     /* translation nodes expanded per search per launch; 1 = reference visit order */
	int32_t wide_children;   /* 1: the rotation children's ub and lb inner searches run in lock-step; 0 = reference order */
	int32_t icp_max_iter;    /* reference 10000 (src/goicp/jly_icp3d.hpp:114) */
	int32_t verbose;
	int32_t morton_sort;     /* source order on the device: 0 input order, 1 Morton curve, 2 k-d order (default) */
	int32_t rot_batch;       /* most rotation nodes expanded per round when wide_children (default 64; rounds ramp 8, 16, 32 ...) */
	int32_t kd_gpu_build;    /* k-d tree (box hierarchy) built on the device (Morton order, looser boxes): 1 yes, 0 or -1 host median splits (default) */
	float trim_fraction;     /* GoICP::trimFraction (src/goicp/jly_goicp.h:116): fraction of the largest residuals ignored; reference 0 */
	/* Search domain: the [params.rotation] / [params.translation] tables of the reference's configs
	 * (test/skull_goicp.toml:22-41; declared in src/common.h:157-169 but never parsed or applied there).
	 * use_* = 0 (default): the CPU path's fixed domain (src/goicp/jly_goicp.cpp:44-53), rotation cube [-pi,pi]^3 in
	 * angle-axis space, translation cube [-0.5,0.5]^3, unlimited depth.  use_* = 1: the root cube is the smallest cube
	 * that contains the box, children outside the box are never evaluated as candidates or expanded; a rotation range
	 * of +-180 degrees on every axis is the reference root itself.  *_search_depth > 0: nodes at that depth are
	 * evaluated but not split further (the guarantee is then limited to that resolution). */
	int32_t use_rot_range, use_trans_range;
	float rot_min[3], rot_max[3];       /* degrees; components of the angle-axis vector */
	float trans_min[3], trans_max[3];   /* in cloud units after `resize` */
	int32_t rot_search_depth, trans_search_depth;
	int32_t icp_fused;       /* 1: one launch per ICP iteration, the last workgroup to arrive runs the update;
	                          * 0: correspondence pass + update as two launches (same arithmetic, bit-identical states) */
	int32_t bounds_fp16;     /* 1: the BnB cube bounds read a half-precision copy of the bricked DT (one 128-byte line per 4x4x4 brick
	                          * instead of two), rounded toward zero so that lower bounds stay valid; upper bounds come out low by
	                          * <= 2^-10 relative.  ICP, the DT re-score of a pose and trimmed bounds keep the fp32 grid.  NOT the
	                          * bit-parity path: opt-in, default 0 */
	int32_t icp_nn_cache;    /* opt-in (default 0; measured slower on real ICP trajectories).  1: the ICP pass keeps, per source point, its last neighbour m, the position q_ref it was found at
	                          * and a lower bound s on the distance from q_ref to every OTHER target point (a 2-nearest walk); at a later
	                          * position q the walk is skipped whenever |q - m| + |q - q_ref| < s, which proves m is still THE nearest
	                          * neighbour -- exact, not approximate; 0: every query walks the tree in every pass (bit-identical results) */
	int32_t flow;            /* opt-in (default 0 = lock-step batches: every rotation batch runs until its slowest inner search has stopped).
	                          * L > 0: continuous flow of the outer search over the device queues -- a rotation child is handled as soon as
	                          * both its inner searches have stopped, and the next rotation parents are admitted when at most L searches
	                          * still run.  Same bounds and prune rules, same optimum; more speculative work -- faster or slower depending
	                          * on which candidate happens to be refined first */
	int32_t adaptive_k;      /* 1 (default): when few inner searches are still running (the stragglers of a batch) each may expand up to 512
	                          * nodes per round instead of trans_batch: fewer latency-bound rounds; 0: always trans_batch */
	int32_t queue_cap;       /* test hook: nodes a device-resident queue may hold before its batch is re-run through the host queues
	                          * (0 = the full 8 192-node slab) */
	int32_t device_queues;   /* 1 (default): the inner-BnB translation queues live on the device -- a round of all active inner
	                          * searches is two launches, no host round trip; 0: host-side queues (always used with trans_batch == 1,
	                          * the reference visit order) */
	int32_t lds_tiles;       /* LDS-staged distance-transform tiles (north_star (a)) for the device-queue search: an inner search whose
	                          * selected nodes lie within tile_spread_vox voxels of each other (the deep rounds of a search) has them
	                          * evaluated by bounds_tile_kernel -- the DT box a 64-point patch of the cloud can reach under ALL of them is
	                          * staged in LDS once, a lookup is a ds_read -- instead of by the gathering kernel; same per-point float
	                          * expressions (bounds agree to the order of the sums).  2 (default): that evaluation is launched only while the
	                          * previous rounds had searches that qualify (shallow registrations never pay for it); 1: every round; 0: off */
	float tile_spread_vox;   /* default 10 */
	int32_t tile_min;        /* fewest expansions of a search for the tile list (default and minimum 8) */
	int32_t stale_widen;     /* adaptive_k refinement (default 1).  An inner search whose incumbent did not improve in its last round is PROVING, not finding: every
	                          * queued node whose lower bound is more than SSEThresh below the incumbent has to be expanded whatever the order (the per-node
	                          * stop rule, jly_goicp.cpp:257, admits nothing else), so a wider round wastes no cube bound and saves rounds.  1: its round
	                          * doubles after one such round and again after three; 2: x4 after one; 0: off */
	int32_t ub_tiebreak;     /* opt-in (default 0: measured, no registration got faster -- ties are rare beyond level 2).  1 (widened search only): rotation cubes of EQUAL lower bound and width are
	                          * expanded in the order of the smallest upper bound their own inner search saw (the basin most likely to refine the
	                          * optimum first).  The reference leaves that order to its heap (src/goicp/jly_goicp.h:44-56); any order of a best-first
	                          * BnB keeps the bounds valid.  0: the reference's key only.  Ignored in the reference-order mode */
	int32_t icp_point_seed;  /* 1 (default): every neighbour walk of the ICP pass / NN operator starts from a REAL candidate -- per DT voxel the engine keeps the
	                          * leaf slot of a target point whose seed voxel is nearest (the exact EDT passes once more, carrying their arg-min; V^3 x 4
	                          * bytes, built with the k-d tree) -- instead of from the distance-transform bound |q - c(v)| + DT[v] + 0.9 voxel, whose slack
	                          * leaves far-from-surface queries a shell of dozens of candidate leaves.  Exactness does not depend on the table (any
	                          * target point is a valid candidate; ties still go to the lowest index).  0: the DT bound */
	float ub_share;          /* widened search: on top of the rot_batch parents drawn by smallest lower bound, this fraction more are drawn per batch by the
	                          * smallest upper bound seen inside them (needs ub_tiebreak = 1; default 0 = none: measured slower) */
	int32_t twin_fusion;     /* 1 (default): the two inner searches of a rotation child -- GoICP::InnerBnB without and with the rotation uncertainty
	                          * (jly_goicp.cpp:494 and :532), same rotated cloud -- run in lock-step; when both list the SAME translation node in a
	                          * round its 8 children are gathered from the distance transform ONCE and the two passes' sums are formed from the
	                          * same fetched values (bit-identical to separate evaluations).  0: every listed node is evaluated on its own */
	int32_t sort_items;      /* 1 (default): a round of >= 2 048 expansions (the first rounds of a rotation batch: hundreds of rotations x whole levels of
	                          * the translation tree) walks its (expansion, 4 096-point chunk) work items in the order of the distance-transform cell
	                          * their gathers land in, so that an XCD's L2 serves neighbouring items instead of streaming the grid from the Infinity
	                          * Cache.  Changes no bound's terms, only the chunking of its sum.  0: search order */
	int32_t stale_compact;   /* default 2048.  A proving inner search (stale_widen) whose queue holds at least this many nodes selects its next nodes by
	                          * the Morton order of their corners instead of by lower bound: every queued node that passes the stop rule
	                          * (jly_goicp.cpp:257) has to be expanded whatever the order, a run of Morton-neighbours is spatially compact (LDS-tile
	                          * material) and is explored depth-first-like, so the queue slab stops overflowing into the host fall-back.  0: always
	                          * by lower bound (the reference's order, jly_goicp.h:64-71) */
	/* (icp_nn_cache, above: 2 = the exact neighbour cache switched on in the tail of a run only -- round 4, measured +0..3 %, opt-in) */
	int32_t stream_priority; /* 0 (default): the engine's HIP stream has the default priority; 1: the highest the device offers (hipStreamCreateWithPriority)
	                          * -- for a latency-bound engine (an ICP loop) that shares the GPU with a throughput engine; measured: tools/overlap_probe.py */
	int32_t lanes;           /* lanes of the device-queue search (round 4): a batch of at least lane_min_searches inner searches (GoICP::InnerBnB calls,
	                          * jly_goicp.cpp:227-340 -- independent of each other) is cut into lanes by rotation child and the lanes run their lock-step
	                          * rounds side by side on their own HIP streams, each with its own lists and control block, so one lane's dependent launches
	                          * drain beside the others'.  0 (default): three lanes when the previous batch's mean round was throughput-bound (>= 64 M
	                          * point-expansions: a count, so the choice is deterministic); 1: never; 2..4: that many, for every batch of at least
	                          * lane_min_searches.  Same searches, same bounds, same result per search; measured (tools/lanes_probe.py): prove-the-optimum
	                          * bunny 6.74 -> 5.7 s, synthetic 40 k 721 -> 649 ms, default (early-exit) registrations never qualify and are unchanged */
	int32_t lane_min_searches; /* default 64 */
} goicp_params;

void goicp_params_default(goicp_params* p);
/* defaults + what a parsed config carries for the engine: mse_threshold and, when the TOML has them, the search
 * ranges and depths (main.cpp:33-42 reads mse_threshold the same way) */
void goicp_params_from_config(const goicp_config* c, goicp_params* p);

/* FastGoICP::FastGoICP(pct, pcs, mse_threshold, mtx) (src/fgoicp/fgoicp.hpp:14-28) + Registration ctor
 * (registration.hpp:66-80) + GoICP::BuildDT/Initialize (src/goicp/jly_goicp.cpp:75-90,134-209):
 * uploads both clouds, builds the distance transform (exact EDT, on the GPU) and the k-d tree. */
int goicp_create(const goicp_params* params, const float* target_xyz, size_t n_target,
                 const float* source_xyz, size_t n_source, goicp_handle* out);
int goicp_destroy(goicp_handle h);
/* SSEThresh = mse_threshold * inlierNum and inlierNum = (int)(N * (1 - trim_fraction))
 * (src/goicp/jly_goicp.cpp:198-208; FastGoICP::sse_threshold, src/fgoicp/fgoicp.hpp:23) as the engine uses them */
int goicp_thresholds(goicp_handle h, float* sse_threshold, int32_t* inliers);
/* HIP device ordinal the engine lives on.  Every entry point makes it current for the calling thread
 * (HIP's current device is per thread) and restores the caller's afterwards. */
int goicp_device(goicp_handle h, int32_t* ordinal);

/* DT3D geometry (src/goicp/jly_3ddt.h:100-111) and the grid itself ([z][y][x], V^3 floats) */
int goicp_dt_info(goicp_handle h, int32_t* V, double* scale, double origin_xyz[3]);
int goicp_dt_download(goicp_handle h, float* grid);

/* ---- bounds operator ------------------------------------------------------------------------
 * Registration::compute_sse_error(RotNode&, std::vector<TransNode>&, bool fix_rot, StreamPool&)
 * (src/fgoicp/registration.hpp:97, registration.cu:88-151) == the inner body of GoICP::InnerBnB
 * (src/goicp/jly_goicp.cpp:262-315).  cubes: B x {centre x,y,z, child width w}.
 * level < 0  <=> fix_rot (no rotation uncertainty radius); otherwise the radii of rotation level
 * `level` (maxRotDis[level], src/goicp/jly_goicp.cpp:148-160) are subtracted. */
int goicp_eval_bounds(goicp_handle h, const float R[9], const float* cubes, size_t B, int32_t level,
                      float* ub, float* lb);

/* batched form: K rotations, B cube records (24 bytes each), each naming its rotation */
typedef struct goicp_cube {
	float tx, ty, tz;  /* cube centre */
	float delta;       /* translation uncertainty radius sqrt(3)/2*w (src/goicp/jly_goicp.cpp:263) */
	float coeff;       /* rotation uncertainty coefficient (goicp_rot_coeff(level)), 0 for fix_rot */
	int32_t rot;       /* index into rots */
} goicp_cube;
int goicp_eval_bounds_batch(goicp_handle h, const float* rots /* K x 9 */, size_t K,
                            const goicp_cube* cubes, size_t B, float* ub, float* lb);
/* device-resident form (no host copies): all four pointers are device pointers; stream = a
 * hipStream_t (NULL = the engine's own stream); asynchronous. */
int goicp_eval_bounds_device(goicp_handle h, const void* d_rots, const void* d_cubes, size_t B,
                             void* d_ub, void* d_lb, void* stream);
/* the same for a batch of UNRELATED cubes (arbitrary rotations and translations in any order -- what the four-argument
 * Registration::compute_sse_error may be handed, and SURVEY's microbench): the cubes are first bucketed on the device by (rotation,
 * pass, translation cell) with a counting sort, evaluated in that order and written back in the caller's order.  Bounds are per
 * cube, so not a bit changes; 65 536 unrelated cubes on the bunny take 2.3 ms instead of 3.75.  n_rots = entries of the rotation
 * table (1..16).  The search itself never needs this: its expansions arrive grouped by search, eight siblings at a time. */
int goicp_eval_bounds_device_grouped(goicp_handle h, const void* d_rots, size_t n_rots, const void* d_cubes, size_t B,
                                     void* d_ub, void* d_lb, void* stream);
int goicp_time_bounds_device_grouped(goicp_handle h, const void* d_rots, size_t n_rots, const void* d_cubes, size_t B,
                                     void* d_ub, void* d_lb, int32_t iters, float* ms_per_call);
/* min of n device floats and (d_argmin != NULL) the first index attaining it -- what a search does with a batch's upper
 * bounds (incumbent = smallest ub, first child on ties: jly_goicp.cpp:319-324); one workgroup, asynchronous on `stream`
 * (NULL = the engine's own).  d_values must be 16-byte aligned. */
int goicp_reduce_min_device(goicp_handle h, const void* d_values, size_t n, void* d_min, void* d_argmin, void* stream);
/* average duration (ms, HIP events on the launch stream) of `iters` back-to-back launches */
int goicp_time_bounds_device(goicp_handle h, const void* d_rots, const void* d_cubes, size_t B,
                             void* d_ub, void* d_lb, int32_t iters, float* ms_per_launch);
float goicp_rot_coeff(goicp_handle h, int32_t level);
float goicp_trans_delta(float child_width);
/* angle-axis vector (a rotation-cube centre) -> rotation matrix, float arithmetic in the order of
 * GoICP::OuterBnB (src/goicp/jly_goicp.cpp:449-467); host-side helper */
void goicp_rodrigues(const float v[3], float R[9]);

/* Registration::compute_sse_error(glm::mat3 R, glm::vec3 t) (src/fgoicp/registration.hpp:96) scored
 * with the DT as GoICP::ICP does (src/goicp/jly_goicp.cpp:100-129). */
int goicp_eval_sse(goicp_handle h, const float R[9], const float t[3], float* sse);

/* FastGoICP::branch_and_bound_R3(rnode, fix_rot) (src/fgoicp/fgoicp.cpp:107-181) == GoICP::InnerBnB
 * (src/goicp/jly_goicp.cpp:227-340).  best_node = {corner x,y,z, width}, written only on improvement. */
typedef struct goicp_counters {
	int64_t rot_pops, trans_pops, cubes, inner_calls, icp_runs, icp_iters, bounds_launches;
	int64_t queue_fallbacks;     /* batches of inner searches in which a device queue outgrew its slab: the searches concerned (only they, since round 4)
	                              * were re-run through the host queues */
	int64_t tile_expansions;     /* BnB expansions (8 cube bounds each, counted in `cubes` too) evaluated from LDS-staged DT tiles */
	int64_t lane_batches;        /* batches of inner searches that ran as two lanes (goicp_params::lanes) */
} goicp_counters;
int goicp_inner_bnb(goicp_handle h, const float R[9], int32_t level, float incumbent, float* value,
                    float best_node[4], goicp_counters* counters);

/* IterativeClosestPoint3D::run (src/fgoicp/icp3d.hpp:30-35, icp3d.cu:83-108) == ICP3D<float>::Run
 * (src/goicp/jly_icp3d.hpp:181-295).  R,t in/out; err = sum of squared NN distances of the last pass. */
int goicp_icp_run(goicp_handle h, float R[9], float t[3], int32_t max_iter, float err_diff,
                  float* err, int32_t* iters);
/* average duration (ms) of one ICP correspondence pass (NN + sums) at the given pose */
int goicp_time_icp_pass(goicp_handle h, const float R[9], const float t[3], int32_t iters, float* ms_per_pass);
/* the same with the neighbour cache in play at a repeated pose: every query hits (the steady-state floor of a pass);
 * goicp_time_icp_pass itself bypasses the cache: every query walks the tree (a pass at a new pose) */
int goicp_time_icp_pass_cached(goicp_handle h, const float R[9], const float t[3], int32_t iters, float* ms_per_pass);

/* kernKDSearchNearest (src/icp_kernel.cu:146-157) / kernFindNearestNeighbor (src/fgoicp/icp3d.cu:13-30):
 * exact 1-NN of n query points in the target; ties -> lowest target index. */
int goicp_nn_query(goicp_handle h, const float* query_xyz, size_t n, int32_t* index, float* dist_sq);

/* ICP::kdTreeGPUStep / ICP::naiveGPUStep (src/icp_kernel.h:9-13, icp_kernel.cu:176-279): ONE ICP
 * iteration from the engine's current step pose (identity after create); the accumulated pose is
 * visible through goicp_poll().curR/curT. */
int goicp_icp_step(goicp_handle h);

/* FastGoICP::run() (src/fgoicp/fgoicp.cpp:9-30) == GoICP::Register() (src/goicp/jly_goicp.cpp:569-585).
 * Blocking; intended for a worker thread. */
int goicp_register(goicp_handle h);
int goicp_cancel(goicp_handle h);      /* the reference's global `goicp_finished` flag (src/goicp/jly_goicp.cpp:400) */

/* Result API: FastGoICP::{get_best_error(), optR, optT, curR, curT, finished} (src/fgoicp/fgoicp.hpp:34,67-69),
 * read by ICP::goicpGPUStep (src/goicp_kernel.cu:161-177).  A consistent snapshot; thread-safe. */
typedef struct goicp_result {
	float optR[9], optT[3];
	float curR[9], curT[3];
	float best_sse;
	int32_t finished;
	goicp_counters counters;
	double dt_build_ms, register_ms;
} goicp_result;
int goicp_poll(goicp_handle h, goicp_result* out);
/* Progress callback: invoked on the registering thread after every published snapshot (new best pose, end of a
 * rotation round, finish).  This is how the C++ shim keeps FastGoICP::{optR,optT,curR,curT,finished} current while
 * run() is executing on a worker thread (the reference's worker writes those members itself,
 * src/fgoicp/fgoicp.cpp:68-69,85-86).  cb = NULL clears it.  Not to be changed while goicp_register() runs. */
typedef void (*goicp_progress_fn)(const goicp_result* snapshot, void* user);
int goicp_set_progress_callback(goicp_handle h, goicp_progress_fn cb, void* user);
/* the output.toml the reference's configs promise (test/bunny_goicp.toml:12) but never write */
int goicp_result_write_toml(goicp_handle h, const char* path);
/* the viz.ply the reference's configs promise (test/bunny_goicp.toml:13) but never write: binary
 * little-endian PLY, target points (grey) followed by the source under optR|optT (red) */
int goicp_result_write_ply(goicp_handle h, const char* path);
/* source cloud under (R,t), original point order (what src/goicp_kernel.cu:181-193 draws) */
int goicp_transform_source(goicp_handle h, const float R[9], const float t[3], float* out_xyz);

/* ---- multi-GPU: rotation cubes sharded over ranks (new; the reference is single-GPU) -----------
 * Each rank owns every world-th cube of the 64 level-2 rotation cubes and runs its own best-first
 * search; between steps the caller min-all-reduces best_sse (and broadcasts the winner's R|t) and
 * feeds the result back with goicp_offer_best(). */
typedef struct goicp_step_status {
	int32_t finished, early_exit;
	float best_sse, frontier_lb;
	int64_t rot_pops;
} goicp_step_status;
int goicp_set_shard(goicp_handle h, int32_t rank, int32_t world);
int goicp_register_begin(goicp_handle h);
int goicp_register_step(goicp_handle h, int32_t max_rot_pops, goicp_step_status* out);
int goicp_offer_best(goicp_handle h, float sse, const float R[9], const float t[3]);
int goicp_register_end(goicp_handle h);

/* ---- the sharded registration inside the library (csrc/shard.cpp, csrc/rccl_comm.cpp) ------------------------------
 * One call per rank drives the whole protocol: step the local search, ONE all-reduce(MIN) of six packed 64-bit words
 * per step ({best SSE, rank}, frontier lower bound, early-exit / active / idle flags, failure word), a 48-byte broadcast
 * of the winner's R|t only when the global best moved, global termination, and -- when a rank runs dry -- rebalancing
 * (queue sizes gathered, every second cube of the largest queue broadcast to the idle rank).
 * Failure is a collective decision: a rank whose engine callback fails keeps taking part in the exchange, every rank
 * leaves the loop in the same iteration, ends its registration and returns an error (its own status, or GOICP_ERR_PEER
 * with goicp_shard_stats.failed_rank set).  A collective that misses the communicator's deadline returns
 * GOICP_ERR_TIMEOUT on the rank that waited.
 * The communicator is a callback table, so the protocol also runs (and is tested) without RCCL. */
typedef struct goicp_comm_ops {
	void* ctx;
	int32_t rank, world;
	/* element-wise MIN over the ranks of n unsigned 64-bit words, in place; blocking */
	int (*allreduce_min_u64)(void* ctx, uint64_t* words, size_t n);
	/* broadcast `bytes` bytes from rank `root`; blocking */
	int (*bcast)(void* ctx, void* buf, size_t bytes, int32_t root);
} goicp_comm_ops;
typedef struct goicp_shard_stats {
	int64_t steps, exchanges, broadcasts, donations, donated_cubes;
	int64_t steps_idle;   /* steps in which this rank had nothing left to expand while the search went on */
	double wait_ms;       /* host time this rank spent blocked in collectives (waiting for the slowest rank) */
	double step_ms;       /* host time inside the engine's step() */
	float best_sse;
	int32_t failed_rank;  /* -1, or the rank whose failure ended the run */
} goicp_shard_stats;
typedef struct goicp_shard_options {
	int32_t rot_pops_per_step;  /* rotation parents per step (>= 1); with ramp_to: of the FIRST step */
	int32_t rebalance;          /* 1: idle ranks receive cubes from the largest queue */
	int32_t stale_exchange;     /* 0: the exchange of a step is consumed before the next step (bulk-synchronous);
	                               1: it runs on a helper thread while the next step is evaluated and is consumed after it --
	                               a rank waits only for ranks more than one step behind */
	int32_t ramp_to;            /* 0: every step expands rot_pops_per_step parents.  R > rot_pops_per_step: the step width doubles
	                               from step to step up to R -- the single-GPU driver's own ramp (8, 16, 32, 64 rotation parents per
	                               batch: few, large launches once a registration has proved to be long), one exchange per batch.
	                               (This field was `reserved`, always 0, up to ABI 3.) */
} goicp_shard_options;
/* the engine side of the protocol as a callback table (goicp_register_sharded fills it for a real engine; tests
 * supply a CPU stand-in).  nodes7: 7 floats per rotation cube {corner x,y,z, width, ub, lb, level}. */
typedef struct goicp_shard_engine_ops {
	void* ctx;
	float sse_threshold;
	int (*begin)(void* ctx, int32_t rank, int32_t world);
	int (*step)(void* ctx, int32_t max_rot_pops, goicp_step_status* out);
	int (*pose)(void* ctx, float* sse, float R[9], float t[3]);
	int (*offer)(void* ctx, float sse, const float R[9], const float t[3]);
	int (*queue_size)(void* ctx, int32_t* n);
	int (*donate)(void* ctx, int32_t max_nodes, float* nodes7, int32_t* n);
	int (*receive)(void* ctx, const float* nodes7, int32_t n);
	int (*end)(void* ctx);
} goicp_shard_engine_ops;
int goicp_run_sharded(const goicp_shard_engine_ops* engine, const goicp_comm_ops* comm, int32_t rot_pops_per_step,
                      int32_t rebalance, goicp_shard_stats* stats);
/* the same for an engine handle (blocking; one call per rank, each rank with its own engine on its own GPU) */
int goicp_register_sharded(goicp_handle h, const goicp_comm_ops* comm, int32_t rot_pops_per_step, int32_t rebalance,
                           goicp_shard_stats* stats);
/* both with the full option set */
void goicp_shard_options_default(goicp_shard_options* out);   /* 8 parents in the first step, ramp_to 32, rebalancing on, bulk-synchronous */
int goicp_run_sharded_opt(const goicp_shard_engine_ops* engine, const goicp_comm_ops* comm, const goicp_shard_options* opt,
                          goicp_shard_stats* stats);
int goicp_register_sharded_opt(goicp_handle h, const goicp_comm_ops* comm, const goicp_shard_options* opt, goicp_shard_stats* stats);
/* deadline of every single collective of a communicator made by this library (thread or RCCL); default: the
 * environment's GOICP_COMM_TIMEOUT_MS, else 60 000 ms.  A collective that misses it returns GOICP_ERR_TIMEOUT and leaves
 * the communicator unusable (destroy it; an RCCL communicator the library owns is aborted with ncclCommAbort). */
int goicp_comm_set_timeout_ms(goicp_comm_ops* comm, int32_t timeout_ms);
/* in-process communicator: `world` host threads of ONE process, rank r calling with out[r] (tests; rehearsing the
 * N-rank path with N engines on one GPU).  Destroy every element. */
int goicp_thread_comm_create(int32_t world, goicp_comm_ops* out /* [world] */);
int goicp_thread_comm_destroy(goicp_comm_ops* comm);
/* RCCL communicator (ncclAllReduce / ncclBroadcast over xGMI on a stream of its own, separate from the engine's compute
 * stream).  id128: an ncclUniqueId (128 bytes) made by ONE rank with goicp_rccl_unique_id() and handed to the others
 * by whatever launched them (MPI, torch.distributed, a file); goicp_rccl_comm_create() is collective.  `device` = the
 * HIP device of this rank. */
#define GOICP_RCCL_ID_BYTES 128
int goicp_rccl_unique_id(char id128[GOICP_RCCL_ID_BYTES]);
int goicp_rccl_comm_create(const char id128[GOICP_RCCL_ID_BYTES], int32_t rank, int32_t world, int32_t device, goicp_comm_ops* out);
/* wrap an existing ncclComm_t (e.g. one of ncclCommInitAll's); the communicator stays the caller's */
int goicp_rccl_comm_wrap(void* nccl_comm, int32_t rank, int32_t world, int32_t device, goicp_comm_ops* out);
int goicp_rccl_comm_destroy(goicp_comm_ops* comm);
/* single-process form: `world` engines (one per GPU, device r for rank r) created from the same clouds and driven by
 * `world` host threads over ncclCommInitAll -- what goicp_cli --ranks N runs.  Results: rank 0's engine handle is
 * returned in *out (poll / write it like any other); stats per rank optional ([world]). */
int goicp_register_multi_gpu(const goicp_params* params, const float* target_xyz, size_t n_target, const float* source_xyz,
                             size_t n_source, int32_t world, int32_t rot_pops_per_step, goicp_handle* out, goicp_shard_stats* stats);

/* ---- measurement / test helpers ----------------------------------------------------------------
 * goicp_probe_gather: measured ceiling of the path that bounds the cube-bound kernel -- independent 4-byte loads
 * into the engine's resident distance transform, nothing else.  mode 0: the 64 lanes of a wave-instruction read 64
 * consecutive floats; mode 1: 64 different 128-byte lines; mode k in {4, 8, 16, 32}: k distinct lines per instruction, the
 * lanes in k runs of 64/k consecutive floats (the cost curve between the two extremes); mode 2: the lookups served from a
 * 64 KiB tile staged in LDS (what an LDS-staged DT tile would deliver once it is loaded; window_bytes ignored).  window_bytes = footprint each workgroup draws its
 * addresses from (rounded up to a power of two, at least 16 KiB, at most the grid).  Result: lookups per second.
 * goicp_debug_kabsch: the device-side 3x3 SVD / Kabsch routine of the ICP update (Matrix::svd use in
 * src/goicp/jly_icp3d.hpp:266-285) on a caller-supplied H (row-major), on the current device; test-only. */
int goicp_probe_gather(goicp_handle h, int32_t mode, size_t window_bytes, double* lookups_per_s);
int goicp_debug_kabsch(const float H[9], float R[9]);
/*
 * goicp_debug_bounds_tile (measurement / test): the cube bounds of the 8 children of nseg x n translation nodes (n <= 64; segment i
 * uses rotation rots9[9*i..], its nodes parents4[(i*n + e)*4..] = corner xyz + width, rotation level `level`) evaluated twice:
 * by the LDS-staged-tile kernel (north_star (a): the DT box a 64-point patch can reach under all the segment's translations is copied
 * to LDS once) and by the direct gather kernel of the search (registration.cu:27-60's role).  ub/lb arrays: 8*nseg*n floats each;
 * ms[0] / ms[1]: milliseconds per launch, tile / direct; stats[0] / stats[1]: 64-point patches staged / too large to stage.
 */
int goicp_debug_bounds_tile(goicp_handle h, const float* rots9, const float* parents4, int32_t nseg, int32_t n, int32_t level, int32_t chunks,
                            float* ub_tile, float* lb_tile, float* ub_direct, float* lb_direct, float ms[2], uint32_t stats[2]);
/*
 * goicp_debug_queue_expand (test): the n (<= 128) given translation nodes expanded by ONE round of the device-resident inner-BnB queues --
 * an upper-bound search (maxRotDisL == NULL, jly_goicp.cpp:492) and a lower-bound search (rotation level `level`, :551) of the same
 * rotation, both listing all n nodes: the lock-step round the outer search runs (bnb_queue_kernel selection + bounds_queue_kernel, the
 * twin-fused evaluation when the engine has it on).  Outputs: the 8 children's (ub, lb) of every node for both passes, in the order of
 * parents4 (inner body of GoICP::InnerBnB, jly_goicp.cpp:262-335).  info[0] = point chunks of the evaluation, info[1] = twin lists in use.
 */
int goicp_debug_queue_expand(goicp_handle h, const float R[9], int32_t level, const float* parents4, int32_t n, float* ub_ubpass, float* lb_ubpass,
                             float* ub_lbpass, float* lb_lbpass, int32_t info[2]);
/* diagnostics of the ICP pass's neighbour cache: two scoring passes at (R, t); *hits = queries of the second pass that
 * skipped the tree walk (-1 when the cache is off) */
int goicp_debug_cache_hits(goicp_handle h, const float R[9], const float t[3], int64_t* hits);

#ifdef __cplusplus
}
#endif
#endif /* GOICP_MI355_H */
