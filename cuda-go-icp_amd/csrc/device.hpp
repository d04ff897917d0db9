// Device layer of the MI355X Go-ICP engine: plain structs + launch wrappers (implemented in
// device.hip, compiled by hipcc for gfx950).  Host C++ (engine.cpp, goicp_api.cpp) only sees this.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstddef>
#include <cstdint>

namespace goicp {

// Search-range cull shared by the host queues (engine.hpp) and the device queues (bnbqueue.hip): the half-open cube
// [x, x+w)^3 against the closed range [lo, hi] -- low face of the range exclusive for a cube that only ends there,
// high face inclusive for the cube that starts there ([params.rotation] / [params.translation] of the reference's
// configs, src/common.h:157-169).
#if defined(__HIPCC__)
__host__ __device__
#endif
inline bool cube_in_range(float x, float y, float z, float w, const float lo[3], const float hi[3])
{
	return x + w > lo[0] && x <= hi[0] && y + w > lo[1] && y <= hi[1] && z + w > lo[2] && z <= hi[2];
}

// Distance transform of the target cloud, resident in HBM.
//   layout 0: linear  [z][y][x], x fastest                       (reference order, jly_3ddt.h:53-79)
//   layout 1: bricked 4x4x4 voxels per 256-B brick, bricks [bz][by][bx]; a surface patch touched
//             by one wavefront then spans ~4x fewer cache lines than in the linear layout
//   layout 2: the bricked grid in half precision (128-B bricks), rounded toward zero; bounds evaluation only, opt-in
struct DtDesc {
	const float* grid;
	int V;          // voxels per side
	int VB;         // bricks per side = ceil(V/4)         (layout 1)
	int layout;
	double scale;   // voxels per unit                      (jly_3ddt.cpp:923)
	double xmin, ymin, zmin;
	// float images of the geometry + the error model of the float index fast path (device.hip voxel_fast)
	float scale_f, xmin_f, ymin_f, zmin_f;
	float c1, c2;   // |F_float - F_exact| <= c1 + c2*|F|
	const double* overshoot;   // [n_overshoot]: (double)sqrtf(s)/scale for the out-of-grid extension (jly_3ddt.cpp:1025)
	int n_overshoot;
	// nearest-target-point table (ICP only; null = absent): per voxel, in the grid's layout, the leaf slot (KdDesc::pts) of a
	// target point whose seed voxel is nearest to this voxel -- a REAL candidate for the neighbour search of any query that
	// falls into the voxel, i.e. an upper bound that is almost always the answer itself (launch_nn_seed_build)
	const int32_t* nn_ids;
};

// One translation sub-cube to bound (the inner body of GoICP::InnerBnB, jly_goicp.cpp:262-315).
struct CubeRec {
	float tx, ty, tz;   // cube centre  (jly_goicp.cpp:271-273)
	float delta;        // translation uncertainty radius sqrt(3)/2*w (jly_goicp.cpp:263)
	float coeff;        // rotation uncertainty coefficient 2*sin(min(sqrt3*sigma_l,pi)/2), 0 for the ub pass
	int32_t rot;        // index into the rotation table
};
static_assert(sizeof(CubeRec) == 24, "CubeRec layout");

// One BnB expansion: the kernels derive the 8 children themselves (same float operations as the host,
// jly_goicp.cpp:262-273), so the search driver uploads 24 B per expansion instead of 192.
struct ParentRec {
	float x, y, z;      // parent cube corner
	float w;            // parent cube width
	float coeff;        // rotation uncertainty coefficient of the children, 0 for the ub pass
	int32_t rot;        // index into the rotation table
};
static_assert(sizeof(ParentRec) == 24, "ParentRec layout");

struct Rot9 { float r[9]; };   // row-major

// Flattened k-d tree over the target cloud.  The host builds a balanced binary k-d
// tree by median splits and flattens it into a 64-ary hierarchy of tight bounding boxes with K
// levels (branching factor = wavefront width): group g of level l holds its 64 children's boxes as
// six runs of 64 floats {lo_x, lo_y, lo_z, hi_x, hi_y, hi_z} (384 floats); level l has 64^l groups;
// the children of group g are groups 64g .. 64g+63 of level l+1, or, on the last level, leaves.
// Leaf f owns the kLeafSlots consecutive float4 slots pts[kLeafSlots*f ...] (one aligned 256-B block);
// unused slots hold +inf coordinates with index INT_MAX, empty children have inverted infinite boxes.
constexpr int kLeafSlots = 16;
constexpr int kMaxLevels = 3;
struct KdDesc {
	const float* boxes[kMaxLevels];  // boxes[l]: 64^l groups x 384 floats
	const float4* pts;               // [kLeafSlots * 64^K]  leaf order; .w = original index (int bits)
	int K;                           // levels: 64^K leaves
	int M;
};

struct Pose { float R[9]; float t[3]; };

constexpr int kGroup = 8;          // cubes per workgroup pass (the 8 siblings of one BnB expansion)
constexpr int kBoundsThreads = 256;
constexpr int kIcpAcc = 16;        // sum(q-cq)[3] sum(m-cm)[3] sum((q-cq)(m-cm)^T)[9] sum(d^2)

// ---- bounds ---------------------------------------------------------------------------------
// scratch must hold groups*chunks*2*kGroup floats.  ub/lb: [B].
size_t bounds_scratch_floats(int B, int N, int* groups_out, int* chunks_out);
// `parents` != nullptr: B/8 expansions, `cubes` ignored
hipError_t launch_bounds(const float4* src, int N, const DtDesc& dt, const Rot9* rots, const CubeRec* cubes, const ParentRec* parents,
                         int B, float* scratch, float* ub, float* lb, hipStream_t stream);

// an UNRELATED cube batch, evaluated grouped by (rotation, pass, translation cell) and written back in the caller's order (same bits)
size_t bounds_grouped_scratch_bytes(int B, int nrots);
hipError_t launch_bounds_grouped(const float4* src, int N, const DtDesc& dt, const Rot9* rots, int nrots, const CubeRec* cubes, int B, void* group_scratch,
                                 float* scratch, float* ub, float* lb, hipStream_t stream);

// trimmed form: only the `inliers` smallest residuals of each cube are summed (jly_goicp.cpp:293-315)
hipError_t launch_bounds_trim(const float4* src, int N, const DtDesc& dt, const Rot9* rots, const CubeRec* cubes, const ParentRec* parents, int B,
                              int inliers, float* ub, float* lb, hipStream_t stream);

// ---- device-resident inner BnB (bnbqueue.hip): one queue per inner search, rounds without the host --------------
constexpr int kQueueCap = 8192;     // nodes per search queue (a queue that would overflow sends the batch back to the host driver)
constexpr int kQueueMaxPop = 512;   // most expansions per search and round: what a LONE search may list (a round costs ~150 us of launches and barriers whatever it
                                    // lists, so the last searches of a batch take big steps)
constexpr int kQueueRoundPop = 128;  // ... and what the round's lists are sized for per search slot (QParams::kmax keeps the total inside: many searches, small steps)
struct QNode { float x, y, z, w, ub, lb; };                    // corner + width (TRANSNODE, jly_goicp.h:59-72)
struct QSearch {                    // one GoICP::InnerBnB call (jly_goicp.cpp:227-340)
	float best;                     // optErrorT (in: the incumbent; out: the search's value)
	float coeff;                    // rotation uncertainty coefficient of this pass, 0 for the upper-bound pass
	int32_t rot;                    // rotation slot
	int32_t count;                  // queued nodes
	int32_t done, improved;         // done: 1 = finished, 2 = stopped because its queue outgrew the slab (QParams::soft_overflow)
	int32_t n_parents, parent_off;  // this round's expansions in the round's list
	int32_t pops, cubes;            // tNodeCount of this search / children evaluated
	float bx, by, bz, bw;           // best child (corner, width), valid when improved
	int32_t tile;                   // this round's expansions are in the tile list (parent_off counts in that list)
	float min_ub;                   // smallest upper bound of any child this search evaluated (whether or not it beat the incumbent)
	int32_t deep;                   // diagnostics: the last selection lay within the tile spread
	int32_t stale;                  // rounds since this search's incumbent last improved (QParams::stale_widen)
	int32_t twin;                   // the other search of the same rotation child (upper-bound pass <-> lower-bound pass), -1: none.  When both list
	                                // the SAME translation node in a round, the lower-bound pass's work items evaluate it for both (one gather, two
	                                // subtractions): bounds_work, device.hip
};
struct TileSeg { int32_t off, n, rot; };   // tile list: a search's expansions parents[off .. off+n), n <= 64, one rotation
struct QCtl {
	int32_t n_groups[2];            // expansions listed per round parity
	int32_t overflow;
	int32_t chunks;                 // point chunks per expansion of the last bound evaluation: > 1 -> its sums are chunk partials in the scratch block,
	                                // added up by the next round's digest (bnb_queue_kernel) -- no separate finalize launch
	int32_t work[2][8];             // per round parity and XCD slot: next work item of the bound evaluation (dynamic distribution)
	// second expansion list of a round (LDS-staged DT tiles, device.hip bounds_tile_kernel): the searches whose selected nodes lie
	// close together -- a few voxels -- are listed here instead and evaluated from DT boxes staged in LDS
	int32_t n_tile_groups[2];       // expansions in the tile list per round parity
	int32_t n_tile_segs[2];         // segments (<= 64 expansions of one search) per round parity
	int32_t tile_chunks;            // point chunks per segment of the last tile evaluation (its sums are chunk partials when > 1)
	int32_t tile_hint;              // running count of (search, round) pairs that qualified for the tile list, whether it was on or not
	int32_t tile_total;             // running count of expansions listed in the tile list
	int32_t n_active[2];            // searches that listed expansions, per round parity (exact: the host sizes the next rounds' steps by it)
	int32_t sel_hist[4][4];         // diagnostics (verbose): expansions selected, by [selection size < 16, < 32, < 64, >= 64][spread <= 5, <= 10, <= 20, > 20 voxels]
};
struct QTile {                      // buffers of the tile list (all null / zero: tiles off)
	ParentRec* parents[2];
	TileSeg* segs[2];
	float* ub; float* lb; float* scratch;
};
struct QParams {
	float thr;                      // SSEThresh
	int32_t K;                      // expansions per search and round (<= kQueueMaxPop)
	int32_t kmax;                   // cap on K after the kernel's own widening: searches still running x kmax fits the round's lists
	int32_t list_cap;               // expansions the round's lists hold (a round that would exceed it flags overflow: host fallback)
	int32_t seg_cap;                // segments the tile list holds (list_cap / 64 + search slots: every search adds at most one partial segment)
	float root_x, root_y, root_z, root_w;
	int32_t boxed, depth;           // translation range culling / depth limit (0 = none)
	int32_t cap;                    // nodes a queue may hold (<= kQueueCap; smaller values only to exercise the overflow path)
	int32_t soft_overflow;          // 1: a search whose queue outgrows its slab stops with QSearch::done = 2 and the others go on (the host re-runs that search
	                                // alone through its own queues); 0: it raises QCtl::overflow and the whole batch goes back
	float lo[3], hi[3];
	int32_t tile_on;                // 1: searches that qualify are listed in the tile list this round (its evaluation is launched)
	int32_t tile_min;               // fewest expansions for the tile list (a lane group of the tile kernel = one expansion)
	float tile_spread;              // largest extent, per axis, of the selected nodes' translations (world units) for the tile list
	int32_t stale_widen;            // 1: a search whose incumbent did not improve in its last round expands up to 2 K nodes, after three such rounds 4 K (<= kQueueMaxPop and QParams::kmax)
	int32_t stale_compact;          // > 0: such a search takes its nodes in Morton order (compact, depth-first-like) once its queue holds this many (bnbqueue.hip)
	int32_t tile_stats;             // 1: fill QCtl::sel_hist (verbose runs)
	float tile_stats_scale;         // voxels per world unit
};
hipError_t launch_bnb_init(QSearch* searches, QNode* q, int nsearch, const QParams& qp, QCtl* ctl, hipStream_t stream);
// footprint-ordered work items of the bound evaluation (device.hip): buffers and shape of the sorted launch
struct QSort {
	const float4* cen;              // centroid of every chunk of chunk_pts points
	unsigned* keys;                 // per item: Morton cell of the item's footprint centre
	unsigned* hist;                 // 32 768 bins
	unsigned* order;                // the items in bucket order; nullptr = feature off
	int32_t chunk_pts, chunks;      // the sorted launch shape
	int32_t min_groups;             // smaller rounds keep the unsorted shape
	int32_t shift;                  // voxel -> cell
};
struct QInit { int32_t slot; float best; float coeff; int32_t rot; int32_t twin; };
hipError_t launch_bnb_init_list(QSearch* searches, QNode* q, const QInit* d_list, int n, const QParams& qp, hipStream_t stream);
// digest the previous round (prev_parents + ubs/lbs), select this round's expansions into `parents`, count them in ctl->n_groups[parity]
hipError_t launch_bnb_queue(QSearch* searches, QNode* q, int nsearch, const QParams& qp, const ParentRec* prev_parents, ParentRec* parents,
                            const float* ubs, const float* lbs, const float* scratch, QCtl* ctl, int parity, hipStream_t stream, const QTile* tile = nullptr,
                            int* parent_search = nullptr,        // parent_search[g] = the search that listed expansion g of the direct list (for the twin test)
                            bool deep = true);                   // deep: the 64-VGPR build (two searches per CU; batches of a prove-the-optimum run), else the 128-VGPR one (no spills)
// bounds of the tile list of round `parity` (segment count known to the device only); fixed grid
hipError_t launch_bounds_tile_queue(const float4* src, int N, const DtDesc& dt, const Rot9* rots, const QTile& tile, QCtl* ctl, int parity, hipStream_t stream);
size_t bounds_tile_queue_scratch_floats(int max_groups);
// bounds of the 8 children of the *d_groups expansions in `parents` (count known to the device only); max_groups sizes the grids
// (d_chunks: QCtl::chunks -- the evaluation leaves chunk partials in `scratch` when it splits the cloud, and says so there)
hipError_t launch_bounds_queue(const float4* src, int N, const DtDesc& dt, const Rot9* rots, const ParentRec* parents, const int* d_groups,
                               int* d_work8, int* d_chunks, int max_groups, int inliers, float* scratch, float* ub, float* lb, hipStream_t stream,
                               const QSearch* searches = nullptr, const int* parent_search = nullptr,    // both given: twin expansions are evaluated once
                               const QSort* qsort = nullptr);                                           // given: a round of >= min_groups expansions walks its items in the order launch_queue_sort left
size_t bounds_queue_scratch_floats(int max_groups, int sorted_chunks = 0);
hipError_t launch_chunk_centroids(const float4* src, int N, int chunk_pts, float4* cen, hipStream_t stream);
hipError_t launch_queue_sort(const ParentRec* parents, const Rot9* rots, const int* d_groups, int max_groups, const QSort& qs, const DtDesc& dt, hipStream_t stream);
int qsort_shift(int V);
bool bounds_uses_lean(const DtDesc& dt);      // the launch picks the lean sibling path (and with it twin fusion and footprint-ordered items) for this grid
size_t qsort_hist_bytes();
// LDS-staged DT tiles for the deep expansions of a search (device.hip bounds_tile_kernel): nseg segments of n <= 64 expansions,
// segs = nseg x {int off, int n, int rot}; stats (may be null): [0] sub-patches staged, [1] sub-patches whose box did not fit
hipError_t launch_bounds_tile(const float4* src, int N, const DtDesc& dt, const Rot9* rots, const ParentRec* parents, const void* segs, int nseg, int n,
                              int chunks, float* scratch, float* ub, float* lb, unsigned* stats, hipStream_t stream);

// ---- ICP ------------------------------------------------------------------------------------
// Device-resident state of the ICP loop (ICP3D<float>::Run, jly_icp3d.hpp:181-295).  One
// iteration = icp_pass_kernel (transform, exact 1-NN, pivoted sums) + icp_finalize_update
// (double-precision reduction, convergence test, SVD, pose update); iterations can be queued
// back-to-back, a converged state turns the remaining launches into no-ops.
struct IcpState {
	float R[9], t[3];            // current pose
	float mu_m[3], mu_d[3];      // means carried across iterations (jly_icp3d.hpp:205-206,244-263)
	float cq[3], cm[3];          // pivots of the next pass (current centroids)
	float src_centroid[3];
	float err, err_new;          // previous / last pass: sum of squared NN distances
	float err_diff_n;            // err_diff * num (jly_icp3d.hpp:255)
	float n;                     // number of source points
	int32_t converged, iters, passes;
	int32_t carry_means;         // 1: the reference's carried means; 0: fresh means (single-step API)
	int32_t frozen;              // 1: passes only score, the pose is not updated
	float acc_scale, acc_inv;    // small clouds: the pass adds its sums as 64-bit fixed point, value * acc_scale (a power of two chosen from the
	                             // clouds' extent so that N terms cannot overflow), into kIcpAccReplicas x 16 accumulators
};
constexpr int kIcpAccReplicas = 32;   // workgroup b adds to replica b % 32: ~60 adds per address and pass (a device-scope atomic takes ~12 ns)
constexpr int kIcpStridedMaxN = 40000;   // up to this many source points: strangers per wavefront + fixed-point sums (device.hip icp_pass_kernel)
int icp_blocks(int N);           // workgroups per pass
size_t icp_partials_floats(int N);   // floats the `partials` buffer of launch_icp_iteration must hold
// ticket: a device int that is zero between launches -> ONE fused launch (the last workgroup to arrive runs the
// finalize); nullptr -> two launches (pass, finalize).  Same arithmetic and summation order: bit-identical states.
// nn_cache: 2 float4 per source point, zero-initialised once (never invalidated: its entries are statements about the
// static target cloud) -> a pass skips, exactly, the tree walk of every query whose cached neighbour is provably still the
// nearest; nullptr -> every query walks.
// acc: kIcpAccReplicas x 16 zeroed 64-bit words (kept zero between iterations by the finalize) -> clouds of up to kIcpStridedMaxN
// points sum there instead of writing a row of partial sums per workgroup; nullptr -> rows for every size
hipError_t launch_icp_iteration(const float4* src, int N, IcpState* d_state, const KdDesc& kd, const DtDesc& dt,
                                float* partials, int* ticket, float4* nn_cache, int* hit_counter, hipStream_t stream, unsigned long long* acc = nullptr);
// trimmed iteration: only the `num` nearest correspondences enter the sums (IcpState.n must be num)
int icp_trim_blocks(int N);
hipError_t launch_icp_iteration_trim(const float4* src, int N, int num, IcpState* d_state, const KdDesc& kd, const DtDesc& dt,
                                     float* nn_d2, int* nn_slot, unsigned char* include, float* partials, hipStream_t stream);
// In-place p <- R p + t, norm recomputed (ICP::kdTreeGPUStep's kernTransform, icp_kernel.cu:138-144)
hipError_t launch_transform(float4* src, int N, const Pose& pose, hipStream_t stream);
// NN operator on arbitrary queries (kernKDSearchNearest, icp_kernel.cu:146-157)
hipError_t launch_nn_query(const float* q_xyz, int n, const KdDesc& kd, const DtDesc& dt, int32_t* idx, float* d2,
                           hipStream_t stream);

// min of n floats (+ first index attaining it, may be null): one workgroup; v must be 16-byte aligned
hipError_t launch_reduce_min(const float* v, int n, float* out_min, int* out_idx, hipStream_t stream);

// ---- test / measurement helpers ---------------------------------------------------------------
hipError_t launch_kabsch_debug(const float* d_H9, float* d_R9, hipStream_t stream);
// window: floats per workgroup window (power of two >= 4096, <= grid size); 8*iters loads per lane
hipError_t launch_probe_gather(const DtDesc& dt, int mode, unsigned window, int blocks, int iters, float* sink, hipStream_t stream);

// ---- device-side build of the box hierarchy (kdbuild.hip): Morton sort + bottom-up boxes ----------
hipError_t launch_kd_build(const float* d_xyz, int M, int K, const float mn[3], float ext, float* const boxes[kMaxLevels],
                           float4* pts, hipStream_t stream);

// ---- distance transform build (DT3D::Build, jly_3ddt.cpp:889-979; exact EDT) -------------------
// work: V^3 int32 (linear).  out: V^3 floats in dt.layout (may alias work only for layout 0).
hipError_t launch_dt_build(const float* model_xyz, int M, const DtDesc& dt, int32_t* work, float* out,
                           hipStream_t stream);
hipError_t launch_dt_to_half(const float* bricked, void* out_half, size_t n, hipStream_t stream);
// nearest-target-point table: the same exact EDT passes, carrying the arg-min.  pts: the k-d tree's leaf slots (padding slots
// have infinite coordinates); work_d / work_id: V^3 int32 each (linear); out: ids in dt.layout (V^3, or VB^3 x 64 bricked)
hipError_t launch_nn_seed_build(const float4* pts, int nslots, const DtDesc& dt, int32_t* work_d, int32_t* work_id, int32_t* out, hipStream_t stream);

}  // namespace goicp
