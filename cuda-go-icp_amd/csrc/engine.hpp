// Host side of the MI355X Go-ICP engine: owns the HBM-resident clouds / distance transform /
// k-d tree, drives the HIP kernels, and runs the branch-and-bound search and the ICP loop.
// Semantics follow the reference CPU Go-ICP path (src/goicp/jly_goicp.cpp) behind the surface of
// the reference GPU classes (src/fgoicp/fgoicp.hpp, registration.hpp, icp3d.hpp).
#pragma once
#include <atomic>
#include <cstdint>
#include <functional>
#include <mutex>
#include <queue>
#include <string>
#include <vector>

#include "device.hpp"

namespace goicp {

struct Params {
	int dt_size = 300;            // jly_goicp.cpp:56
	double dt_expand = 2.0;       // jly_goicp.cpp:57
	float mse_threshold = 1e-3f;  // Config::mse_threshold (common.cpp:62)
	int dt_layout = 1;            // 0 linear, 1 bricked 4x4x4
	int device = -1;              // -1: current HIP device
	int trans_batch = 32;         // translation nodes expanded per inner search per launch (1 = reference order)
	int wide_children = 1;        // run the rotation children's inner searches concurrently (0 = reference order)
	int rot_batch = 64;           // most rotation nodes expanded per round when wide_children (their 8 children x {ub,lb} searches share launches); ramps up from 8
	int icp_max_iter = 10000;     // jly_icp3d.hpp:114
	int verbose = 0;
	int morton_sort = 2;          // source order on the device: 0 input order, 1 Morton curve, 2 k-d order (locality of the DT gathers)
	int icp_chunk = 16;           // ICP iterations queued per host round trip
	int kd_gpu_build = -1;        // box hierarchy built on the device (Morton sort, looser boxes): 1 yes, 0 / -1 host median splits (threaded)
	int bounds_fp16 = 0;          // 1: BnB cube bounds read a half-precision copy of the bricked DT (rounded toward zero: lower bounds stay valid, upper bounds low by <= 2^-10 relative); ICP, the DT re-score and trimmed bounds keep the fp32 grid.  Not bit-parity: opt-in
	int icp_nn_cache = 0;         // 1: an ICP pass skips (exactly) the tree walk of every query whose cached neighbour is provably still the nearest (measured slower on real trajectories, EXPERIMENTS 3.6: opt-in);
	                              // 2: the same, switched on inside a run only once the error falls by < 2 % per 16 iterations (the tail; measured +0..3 %, EXPERIMENTS R4.11: opt-in);
	                              // 0: every query walks every pass; bit-identical states in every mode
	int flow = 0;                 // opt-in; L > 0: continuous flow over the device queues -- rotation children are harvested one by one and the next batch of parents is admitted when at most this many inner searches still run; 0: lock-step batches
	int adaptive_k = 1;           // 1: when few inner searches still run, each may expand up to 512 nodes per round instead of trans_batch
	int queue_cap = 0;            // test hook: nodes a device queue may hold before the batch falls back to the host queues (0 = the full slab)
	int device_queues = 1;        // 1: inner-BnB queues live on the device, a round is two launches and no host work (bnbqueue.hip); 0: host queues (always used when trans_batch == 1 = the reference visit order)
	int icp_point_seed = 1;       // 1: the ICP neighbour search starts every walk from a real candidate read from a per-voxel nearest-target-point table
	                              // (built once with the k-d tree: the EDT passes carrying their arg-min; V^3 x 4 bytes); 0: from the distance-transform bound
	int ub_tiebreak = 0;          // opt-in, widened search only: rotation cubes with EQUAL lower bounds and equal width are expanded in the order of the smallest
	                              // upper bound their own inner search saw -- the reference leaves that order to its heap; 0: as the reference.  Measured (round 3,
	                              // DESIGN 4): ties are rare beyond level 2, no registration got faster -- default off
	float ub_share = 0.f;         // opt-in, widened search: on top of a batch's parents by smallest lower bound, this fraction more are drawn by the smallest upper
	                              // bound seen inside them (needs ub_tiebreak = 1 for the key).  Measured slower everywhere: default 0
	int sort_items = 1;           // rounds of >= 2 048 expansions walk their (expansion, chunk) items in the order of where their gathers land (device.hip, launch_queue_sort)
	int twin_fusion = 1;          // the same translation node listed by both searches of a rotation child in a round is gathered once (device.hip lean_points<.., 2>)
	int lds_tiles = 2;            // LDS-staged DT tiles for inner searches whose selected nodes lie within a few voxels of each other (deep rounds): 0 off, 1 the
	                              // tile evaluation is launched every round, 2 only while the previous rounds had searches that qualify (default)
	float tile_spread_vox = 10.f; // ... "a few": largest extent of a search's selected translations, in DT voxels (measured: the tile kernel is 1.7x the gathering one at 3 voxels, 1.3x at 5, even at 10)
	int stale_widen = 1;          // adaptive_k: an inner search whose incumbent did not improve in its last round(s) is PROVING, not finding: every queued node whose lower
	                              // bound is more than SSEThresh below the incumbent has to be expanded whatever the order, so a wider round wastes nothing -- its width
	                              // doubles after one such round and again after three (lower-bound searches almost never improve; upper-bound searches until they settle)
	int stale_compact = 2048;     // a proving inner search whose queue holds at least this many nodes selects by Morton order of the cubes' corners instead of by lower
	                              // bound (spatially compact, depth-first-like: LDS-tile material, and the slab stops overflowing); 0: always by lower bound
	int tile_min = 8;             // ... and at least this many expansions (a lane group of the tile kernel is one expansion)
	int lanes = 0;                // n = 2..4 (always n) / 0 (auto: three, when the previous batch's rounds were throughput-bound; default) / 1 (never): a batch of at least
	                              // lane_min_searches inner searches is cut into lanes by rotation slot and the lanes run their lock-step rounds side by side on their own
	                              // streams (own lists, own control block): one lane's dependent launches drain beside the others' (run_inner_device)
	int lane_min_searches = 64;
	int stream_priority = 0;      // 1: the engine's stream gets the highest priority of the device (an ICP engine beside a bounds engine on one GPU: tools/overlap_probe.py)
	int icp_fused = 0;            // 1: one launch per ICP iteration (last workgroup finalizes); 0: pass + finalize launches (A/B, bit-identical)
	float trim_fraction = 0.f;    // GoICP::trimFraction (jly_goicp.h:116; the reference hard-wires 0, jly_goicp.cpp:55)
	// Search domain ([params.rotation] / [params.translation] of the reference's configs, test/skull_goicp.toml:22-41;
	// declared in src/common.h:157-169, never parsed there).  Unset = the CPU path's fixed domain
	// (jly_goicp.cpp:44-53): rotation cube [-pi,pi]^3, translation cube [-0.5,0.5]^3, no depth limit.
	int use_rot_range = 0, use_trans_range = 0;
	float rot_min[3] = {-180.f, -180.f, -180.f}, rot_max[3] = {180.f, 180.f, 180.f};   // degrees, angle-axis components
	float trans_min[3] = {-0.5f, -0.5f, -0.5f}, trans_max[3] = {0.5f, 0.5f, 0.5f};
	int rot_search_depth = 0, trans_search_depth = 0;   // 0 = unlimited; d: nodes of depth d are evaluated, not expanded
};

struct Counters {
	long long rot_pops = 0, trans_pops = 0, cubes = 0, inner_calls = 0, icp_runs = 0, icp_iters = 0;
	long long bounds_launches = 0;
	long long queue_fallbacks = 0;
	long long tile_expansions = 0;   // BnB expansions evaluated from LDS-staged DT tiles (8 cube bounds each; counted in `cubes` too)
	long long lane_batches = 0;      // batches of inner searches that ran as two lanes
};

// what the viewer polls (fgoicp.hpp:34,67-69; goicp_kernel.cu:161-177)
struct Result {
	float optR[9], optT[3], curR[9], curT[3];
	float best_sse;
	int finished;
	Counters counters;
	double dt_build_ms, register_ms;
};

// corner + width node, ordered like jly_goicp.h:44-72 (smaller lb first, then the wider cube)
struct Node {
	float x, y, z, w, ub, lb;
	int l;
	float tie = 0.f;     // third key of the ROTATION queue in the widened search: the smallest upper bound seen inside the cube (0 = unused)
	friend bool operator<(const Node& a, const Node& b)
	{
		if (a.lb != b.lb) return a.lb > b.lb;      // smaller lower bound first, then the wider cube (jly_goicp.h:44-72)
		if (a.w != b.w) return a.w < b.w;
		return a.tie > b.tie;                      // equal in the reference's order (there the heap decides): the more promising cube first
	}
};

struct StepStatus {
	int finished;        // this rank has nothing left (queue empty, converged, or early exit)
	int early_exit;      // best_sse < sse_threshold (jly_goicp.cpp:527): every rank may stop
	float best_sse;
	float frontier_lb;   // min lb over this rank's queue (+inf when empty)
	long long rot_pops;
};

class Engine {
public:
	Engine(const Params& p, const float* target_xyz, size_t M, const float* source_xyz, size_t N);
	~Engine();
	Engine(const Engine&) = delete;

	// ---- operators (all synchronous) ----
	// Registration::compute_sse_error(RotNode&, vector<TransNode>&, fix_rot, StreamPool&) with the CPU
	// path's semantics: cubes = B x {centre xyz, child width}; level < 0 => no rotation radius.
	void eval_bounds(const float R[9], const float* cubes4, size_t B, int level, float* ub, float* lb);
	void eval_bounds_batch(const float* rots9, size_t K, const CubeRec* cubes, size_t B, float* ub, float* lb);
	void eval_bounds_dev(const Rot9* d_rots, const CubeRec* d_cubes, int B, float* d_ub, float* d_lb, hipStream_t s, const ParentRec* d_parents = nullptr);
	void reduce_min_dev(const float* d_v, int n, float* d_min, int* d_idx, hipStream_t s);
	float time_bounds_dev(const Rot9* d_rots, const CubeRec* d_cubes, int B, float* d_ub, float* d_lb, int iters, int grouped_nrots = 0);
	void eval_bounds_dev_grouped(const Rot9* d_rots, int nrots, const CubeRec* d_cubes, int B, float* d_ub, float* d_lb, hipStream_t s);
	float eval_sse(const float R[9], const float t[3]);
	float inner_bnb(const float R[9], int level, float incumbent, float best_node[4], Counters* c);
	float icp_run(float R[9], float t[3], int max_iter, float err_diff, int* iters);
	float time_icp_pass(const float R[9], const float t[3], int iters, bool cached = false);   // cached: every query hits the neighbour cache (steady state); else every query walks
	void nn_query(const float* q_xyz, size_t n, int32_t* idx, float* d2);
	void icp_step();   // one ICP iteration on the engine's current pose (ICP::kdTreeGPUStep)
	// measured ceiling of the gather path (4-byte loads into the resident DT): lookups/s; mode 0 coalesced, 1 divergent
	double probe_gather(int mode, size_t window_bytes);
	long long debug_cache_hits(const float R[9], const float t[3]);
	// test: the n (<= 128) given translation nodes (parents4: corner xyz + width) expanded by ONE round of the device-resident queues -- an
	// upper-bound search (coeff 0) and a lower-bound search (rotation level `level`) of the same rotation, twins of each other, both listing
	// all n nodes: the round the outer search's lock-step batches run (selection by bnb_queue_kernel, evaluation by bounds_queue_kernel with
	// twin fusion when the engine has it on).  out[pass][8 n]: the children's bounds of pass 0 (upper-bound search) and pass 1
	// (lower-bound search), in the order of parents4.  info[0] = point chunks the evaluation split the cloud into, info[1] = 1 when the
	// twin lists were in use.
	void debug_queue_expand(const float R[9], int level, const float* parents4, int n, float* ub0, float* lb0, float* ub1, float* lb1, int info[2]);
	// measurement / test: the 8 children of nseg x n expansions (segment i: rotation i, parents4[(i*n + e)*4 ..] = corner xyz + width) through
	// the LDS-tile kernel and through the direct kernel; out arrays hold 8*nseg*n floats each; ms[0] tile, ms[1] direct (per launch)
	void debug_bounds_tile(const float* rots9, const float* parents4, int nseg, int n, int level, int chunks, float* ub_tile, float* lb_tile,
	                       float* ub_direct, float* lb_direct, float ms[2], unsigned stats[2]);   // queries of a repeated pass that skipped the tree walk (-1: cache off)

	// ---- registration ----
	void run();                                  // FastGoICP::run / GoICP::Register
	void cancel() { cancel_.store(true); }
	Result poll();
	// called on the registering thread after every published snapshot (the reference's worker writes
	// FastGoICP::optR/optT/curR/curT itself, fgoicp.cpp:68-69,85-86)
	void set_progress_callback(std::function<void(const Result&)> cb) { progress_cb_ = std::move(cb); }
	// stepped form for multi-GPU sharding
	void set_shard(int rank, int world) { rank_ = rank; world_ = world; }
	void register_begin();
	StepStatus register_step(int max_rot_pops);
	void offer_global_best(float sse, const float R[9], const float t[3]);   // result of the min all-reduce
	void register_end();
	// rebalancing between ranks (shard.cpp): cubes still worth expanding / give every second one away / take some
	int queue_size() const { return (early_exit_ || converged_) ? 0 : (int)queue_.size(); }
	int donate(int max_nodes, float* nodes7);
	void receive(const float* nodes7, int n);

	// ---- inspection ----
	const DtDesc& dt() const { return dt_; }
	void dt_download(float* grid_linear);        // V^3, [z][y][x]
	size_t n_source() const { return N_; }
	size_t n_target() const { return M_; }
	const float* target_xyz() const { return h_target_.data(); }
	float sse_threshold() const { return sse_thresh_; }
	int device() const { return dev_; }
	int inliers() const { return inliers_; }
	float rot_coeff(int level) const;
	hipStream_t stream() const { return stream_; }
	const float4* d_source() const { return d_src_; }
	void source_transformed(const float R[9], const float t[3], float* out_xyz);  // original order

private:
	struct InnerSearch;
	void init(const float* target_xyz, size_t M, const float* source_xyz, size_t N);
	void release();
	void ensure_batch(size_t B, size_t K);
	void ensure_stage(int k, size_t B);
	void run_inner(std::vector<InnerSearch*>& searches, const std::vector<Rot9>& rots);
	void run_inner_host(std::vector<InnerSearch*>& searches, const std::vector<Rot9>& rots, bool fallback = false);
	bool run_inner_device(std::vector<InnerSearch*>& searches, const std::vector<Rot9>& rots);   // false: a round's lists overflowed, nothing was changed; redo_: searches whose own queue did
	std::vector<InnerSearch*> redo_;
	bool soft_overflow_ = true;       // env GOICP_SOFT_OVERFLOW = 0 (A/B only): a slab overflow sends the whole batch back, as before round 4
	void ensure_queues(size_t nsearch);
	void process_parents(const std::vector<Node>& parents);
	struct Kid { Node node; float R[9]; float parent_lb; };                       // a rotation child and its Rodrigues matrix
	struct SearchOut { float best; bool improved; Node best_node; long long pops, cubes; float min_ub; };   // what an inner search returns (min_ub: smallest upper bound of any cube it evaluated)
	void make_kids(const std::vector<Node>& parents, std::vector<Kid>& kids);
	bool handle_ub(Kid& k, const SearchOut& s);
	void handle_lb(Kid& k, const SearchOut& s);
	// continuous flow of the outer search over the device-resident queues (engine.cpp)
	static constexpr int kFlowSearches = 2048;     // search slots (two per rotation child in flight)
	struct Flight { Kid kid; int rot_slot, s_ub, s_lb; float incumbent; bool handled; };
	bool flow_mode() const { return p_.device_queues && p_.wide_children && p_.trans_batch > 1 && p_.flow; }
	int flow_step(int max_rot_pops);
	void flow_reset();
	void flow_fallback();
	QParams queue_params() const;
	std::vector<Flight> flights_;
	std::vector<int> free_search_, free_rot_;
	int q_hi_ = 0, q_parity_ = 0, flow_active_ = 0;
	QInit* h_qinit_ = nullptr; QInit* d_qinit_ = nullptr;
	void adopt(float err, const float R[9], const float t[3]);
	float icp_from(float R[9], float t[3]);
	void publish(bool finished);
	void icp_state_init(const float R[9], const float t[3], float err_diff, int carry_means, int frozen);
	void icp_state_fetch();

	// HIP's current device is per host thread: every public entry point re-establishes the engine's device
	struct DeviceGuard {
		int prev = -1, want = -1;
		explicit DeviceGuard(int dev);
		~DeviceGuard();
	};
	// A cube is the half-open box [x, x+w)^3, a configured range the CLOSED box [lo, hi]: the cube is kept when the two
	// intersect.  So the high face of a range is inclusive -- a bound (or a fixed value, lo == hi) that coincides with a
	// split plane belongs to exactly one cube per level, the one that starts there -- and the low face of a cube that
	// merely ends at lo is not.  Same rule on the device (bnbqueue.hip in_box).
	bool in_box(const Node& c, const float lo[3], const float hi[3]) const { return cube_in_range(c.x, c.y, c.z, c.w, lo, hi); }
	void* scratch_bytes(size_t bytes);   // grow-only device scratch for the query / transform operators

	Params p_;
	int dev_ = 0;
	Node rot_root_{}, trans_root_{};
	bool rot_boxed_ = false, trans_boxed_ = false;
	float rot_lo_[3], rot_hi_[3], trans_lo_[3], trans_hi_[3];
	std::function<void(const Result&)> progress_cb_;
	void* d_opscratch_ = nullptr; size_t cap_opscratch_ = 0;
	size_t M_ = 0, N_ = 0;
	float sse_thresh_ = 0.f, icp_err_diff_ = 0.f;
	int inliers_ = 0;             // inlierNum = (int)(Nd * (1 - trimFraction)), jly_goicp.cpp:201
	int rank_ = 0, world_ = 1;

	hipStream_t stream_ = nullptr;
	hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
	// A/B only (env GOICP_TILE_CONCURRENT = 1): the tile list's evaluation BESIDE the direct list's (two independent kernels of the same round) on a
	// second stream, forked after the queue kernel and joined before the next one.  Measured slower (EXPERIMENTS R4.8): default off
	hipStream_t stream2_ = nullptr;
	hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
	int tile_concurrent_ = 0;
	float4* d_src_ = nullptr;         // N  (x,y,z,|p|), k-d order (Params::morton_sort)
	std::vector<int32_t> src_perm_;   // sorted position -> original index
	std::vector<float> h_src_sorted_; // N*4
	std::vector<float> h_target_;     // M*3 (kept for viz.ply)
	float src_centroid_[3] = {0, 0, 0}, model_centroid_[3] = {0, 0, 0};
	DtDesc dt_{};
	float* d_dt_ = nullptr;
	DtDesc dt16_{};                   // Params::bounds_fp16: the same grid in half precision (layout 2)
	void* d_dt16_ = nullptr;
	size_t kd_slots_ = 0;                 // float4 slots behind KdDesc::pts
	int32_t* d_nn_ids_ = nullptr;         // nearest-target-point table of the ICP neighbour search (DtDesc::nn_ids)
	bool score_exact_ = false;        // set while eval_sse scores a pose: always the fp32 grid
	const DtDesc& bounds_dt() const { return (d_dt16_ && !score_exact_ && inliers_ >= (int)N_) ? dt16_ : dt_; }
	double* d_overshoot_ = nullptr;
	// k-d tree
	KdDesc kd_{};
	float* d_kd_boxes_[kMaxLevels] = {nullptr, nullptr, nullptr}; float4* d_kd_pts_ = nullptr;
	// bounds staging
	size_t cap_cubes_ = 0, cap_rots_ = 0, cap_scratch_ = 0;
	CubeRec* d_cubes_ = nullptr; CubeRec* h_cubes_ = nullptr;
	Rot9* d_rots_ = nullptr; Rot9* h_rots_ = nullptr;
	float* d_ub_ = nullptr; float* d_lb_ = nullptr; float* h_ub_ = nullptr; float* h_lb_ = nullptr;
	float* d_scratch_ = nullptr;
	struct Stage {   // per-group staging of the pipelined inner-BnB rounds
		ParentRec* d_parents = nullptr; ParentRec* h_parents = nullptr;   // one record per expansion; the kernels derive the 8 children
		float* d_ub = nullptr; float* h_ub = nullptr;   // ub[B] followed by lb[B]
		size_t cap = 0, B = 0; hipEvent_t ev = nullptr;
	} stage_[2];
	// device-resident inner-BnB queues (bnbqueue.hip).  A LANE is one self-contained set of them -- search slots, node slabs, the
	// round's two expansion lists with their bounds and partial sums, the sort buffers, the control block with its pinned snapshots --
	// driven on its own stream.  Lane 0 always exists; lane 1 is created for batches cut in two (Params::lanes, run_inner_device)
	static constexpr int kMaxLanes = 4;
	struct QLane {
		hipStream_t stream = nullptr;
		size_t cap = 0;                                     // search slots
		QSearch* d_search = nullptr; QSearch* h_search = nullptr;
		QNode* d_nodes = nullptr;
		ParentRec* d_parents[2] = {nullptr, nullptr};
		QSort sort{};                                       // footprint-ordered items of large rounds (device.hip); order == nullptr: off
		int list_cap = 0;                                   // expansions the round's lists (parents, bounds, partial sums) hold
		int seg_cap = 0;                                    // segments the tile list holds (list_cap / 64 + search slots)
		int* d_psearch[2] = {nullptr, nullptr};             // per listed expansion: the search that listed it (twin test of the bound evaluation)
		float* d_ub = nullptr; float* d_lb = nullptr; float* d_scratch = nullptr;
		QCtl* d_ctl = nullptr; QCtl* h_ctl = nullptr;       // h_ctl: two pinned snapshots (one per chunk of rounds in flight)
		hipEvent_t ev_ctl[2] = {nullptr, nullptr};
		QTile tile{};                                       // the tile list's buffers (null when lds_tiles == 0 or the DT is not bricked fp32)
		int tile_hint_seen = 0;                             // QCtl::tile_hint at the last read-back
	} ql_[kMaxLanes];
	hipStream_t lane_stream_[kMaxLanes] = {};               // lane 0: stream_, lane 1: stream2_, further lanes: their own
	void free_lane(QLane& L);
	void ensure_lane(int li, size_t nsearch);
	double last_round_work_ = 0, lane_min_work_ = 64e6;    // point-expansions (expansions x source points) of the previous batch's mean round / the auto mode's bar
	int auto_lanes_ = 3;                                    // lanes the auto mode cuts a batch into (env GOICP_AUTO_LANES; measured 2 / 3 / 4 at the end of round 4: bunny mse 3e-5
	                                                        // 5.06 / 5.02 / 5.48 s, synthetic 40 k mse 3e-5 621 / 621 / 681 ms, 3 k points mse 3e-5 1 033 / 1 016 / 1 100 ms, bunny mse 1e-4 259 / 261 / 268 ms)
	int lanes_ = 0, lane_min_searches_ = 64;                // Params::lanes / lane_min_searches (env GOICP_LANES / GOICP_LANE_MIN override, tuning only)
	bool tiles_usable() const;
	long long sel_hist_[4][4] = {};       // verbose: QCtl::sel_hist summed over the registration
	double tile_sticky_share_ = 0.5;      // ... when at least this share of the previous batch's cube bounds came from tiles (env GOICP_TILE_STICKY_SHARE, tuning only)
	bool tile_sticky_ = false;            // lds_tiles == 2: the previous batch evaluated expansions from tiles -> this batch launches the tile list in every round
	long long tile_rounds_ = 0;           // rounds whose tile evaluation was launched
	long long queue_rounds_ = 0, queue_fallbacks_ = 0;
	// icp staging
	float* d_icp_partials_ = nullptr; IcpState* d_icp_state_ = nullptr; IcpState* h_icp_state_ = nullptr;
	unsigned long long* d_icp_acc_ = nullptr;   // fixed-point sums of the small-cloud ICP pass (kIcpAccReplicas x 16, zero between iterations)
	float src_radius_ = 0.f, target_abs_max_ = 0.f;   // extents that bound the pass's terms (IcpState::acc_scale)
	float4* d_nn_cache_ = nullptr;     // per source point: {q_ref, sqrt(best2_ref)}, {neighbour, index} (exact walk-skipping, device.hip)
	bool icp_cache_active_ = false;    // icp_nn_cache = 2: switched on inside a run once the error's decrease per chunk falls under icp_cache_rel_ (the tail of a run)
	float icp_cache_rel_ = 0.02f;
	bool count_hits_ = false;
	int* d_icp_ticket_ = nullptr;      // arrival ticket of the fused ICP iteration (zero between launches)
	float* d_nn_d2_ = nullptr; int* d_nn_slot_ = nullptr; unsigned char* d_include_ = nullptr;   // trimmed ICP only
	void icp_launch_one();
	// nn query staging grows on demand
	float rot_coeff_[20];

	// search state
	std::priority_queue<Node> queue_;
	float opt_err_ = 1e10f;
	float optR_[9], optT_[3], curR_[9], curT_[3];
	bool early_exit_ = false, converged_ = false;
	int rot_ramp_ = 8;
	// experiment knob (env GOICP_ICP_DELAY_BATCHES = k, tools/icp_delay_probe.py): a refinement's result is folded in k rotation batches after
	// the upper bound that triggered it -- what an ICP overlapped with the next batches would do to the search, without the concurrency
	struct LateIcp { float e, R[9], t[3]; long long due; };
	std::vector<LateIcp> late_icp_;
	int icp_delay_ = 0;
	long long batches_done_ = 0;
	void fold_late_icp(bool all);
	Counters cnt_;
	std::atomic<bool> cancel_{false};
	std::mutex mtx_;
	Result snap_{};
	double dt_build_ms_ = 0, register_ms_ = 0, bnb_ms_ = 0, icp_ms_ = 0, t_submit_ = 0, t_wait_ = 0, t_collect_ = 0;
	long long level_hist_[32] = {};   // verbose: translation expansions by parent depth
	// icp_step state
	float stepR_[9], stepT_[3];
};

// ---- host utilities (config_io.cpp, kdtree.cpp) ----
struct KdHost {
	std::vector<std::vector<float>> boxes;   // per level: groups x 384 floats (level 0: the root; level l: F*64^(l-1) groups, F = real children of the root)
	std::vector<float4> pts;                 // kLeafSlots slots per leaf
	int K = 1, L = 64;
};
void build_kdtree(const float* xyz, int M, int leaf_max, KdHost* out);
// fn(0..ntasks-1) on up to `threads` host threads (tasks claimed from a counter; the first exception is rethrown)
void parallel_tasks(int threads, int ntasks, const std::function<void(int)>& fn);
void rodrigues(float ax, float ay, float az, float R[9]);   // jly_goicp.cpp:449-467
void debug_kabsch(const float H[9], float R[9]);            // the device SVD routine on the current device (tests)

}  // namespace goicp
