// Engine: resident data set-up, batched branch-and-bound driver, ICP loop.
//
// Search semantics = reference CPU Go-ICP (src/goicp/jly_goicp.cpp):
//   outer best-first BnB over the angle-axis cube [-pi,pi]^3 (OuterBnB, :342-567), nested
//   best-first BnB over the translation cube [-0.5,0.5]^3 (InnerBnB, :227-340), ICP refinement
//   whenever the upper bound improves (:495-530), stop when best - lb <= SSEThresh.
// What is MI355X-first here: the reference evaluates ONE cube per step; this driver keeps the same
// bounds and the same queues but expands `trans_batch` translation nodes of every active inner search
// per kernel launch, and `rot_batch` rotation nodes at once (their children's upper- and lower-bound
// searches run in lock-step): thousands of cube x point evaluations per launch instead of one pass
// over N points.  Any expansion order of a best-first BnB yields valid bounds; with trans_batch = 1
// and wide_children = 0 the visit order is exactly the reference's.
#include "engine.hpp"
#include "trace.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <stdexcept>

namespace goicp {

namespace {

// constants spelled as the reference does (jly_goicp.h:35-36)
constexpr double kPI = 3.1415926536;
constexpr double kSQRT3 = 1.732050808;
constexpr int kMaxRotLevel = 20;

void hip_check(hipError_t e, const char* what)
{
	if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(x) hip_check((x), #x)

double now_ms()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

uint32_t part1by2(uint32_t x)
{
	x &= 0x3ff;
	x = (x ^ (x << 16)) & 0xff0000ff;
	x = (x ^ (x << 8)) & 0x0300f00f;
	x = (x ^ (x << 4)) & 0x030c30c3;
	x = (x ^ (x << 2)) & 0x09249249;
	return x;
}

// device allocation that frees itself on every exit path (temporaries of init / operators)
template <class T> struct DevBuf {
	T* p = nullptr;
	DevBuf() = default;
	explicit DevBuf(size_t n) { HIPCHK(hipMalloc(&p, sizeof(T) * (n ? n : 1))); }
	~DevBuf() { hipFree(p); }
	DevBuf(const DevBuf&) = delete;
	DevBuf& operator=(const DevBuf&) = delete;
	T* release() { T* q = p; p = nullptr; return q; }
};

}  // namespace

Engine::DeviceGuard::DeviceGuard(int dev) : want(dev)
{
	if (hipGetDevice(&prev) != hipSuccess) prev = -1;
	if (prev != want) HIPCHK(hipSetDevice(want));
}
Engine::DeviceGuard::~DeviceGuard()
{
	if (prev >= 0 && prev != want) hipSetDevice(prev);
}

void* Engine::scratch_bytes(size_t bytes)
{
	if (bytes > cap_opscratch_) {
		HIPCHK(hipStreamSynchronize(stream_));
		hipFree(d_opscratch_);
		d_opscratch_ = nullptr; cap_opscratch_ = 0;
		const size_t cap = std::max(bytes, (size_t)1 << 16);
		HIPCHK(hipMalloc(&d_opscratch_, cap));
		cap_opscratch_ = cap;
	}
	return d_opscratch_;
}

struct Engine::InnerSearch {
	int rot_slot = 0;
	float coeff = 0.f;            // 0 => upper-bound pass (maxRotDisL == NULL)
	float best = 0.f;             // optErrorT
	Node best_node{};
	bool improved = false;
	bool done = false;
	std::priority_queue<Node> pq;
	std::vector<Node> parents;    // popped this round
	long long pops = 0, cubes = 0;
	float min_ub = std::numeric_limits<float>::infinity();   // smallest upper bound of any cube evaluated
	int stale = 0;                // rounds since the incumbent last improved (host fallback: round widening)
};

float Engine::rot_coeff(int level) const
{
	if (level < 0) return 0.f;
	return rot_coeff_[level < kMaxRotLevel ? level : kMaxRotLevel - 1];
}

Engine::Engine(const Params& p, const float* target, size_t M, const float* source, size_t N)
    : p_(p), M_(M), N_(N)
{
	try {
		init(target, M, source, N);
	} catch (...) {
		release();      // a half-built engine must not leak device memory
		throw;
	}
}

void Engine::init(const float* target, size_t M, const float* source, size_t N)
{
	if (!target || !source || M == 0 || N == 0) throw std::invalid_argument("goicp: empty target or source cloud");
	if (M > (size_t)INT32_MAX / 8 || N > (size_t)INT32_MAX / 8) throw std::invalid_argument("goicp: cloud too large");
	if (p_.dt_size < 8 || p_.dt_size > 640) throw std::invalid_argument("goicp: dt_size must be in [8,640]");
	if (!(p_.dt_expand >= 1.0) || !(p_.dt_expand <= 64.0)) throw std::invalid_argument("goicp: dt_expand must be in [1,64]");
	if (!(p_.mse_threshold > 0.f)) p_.mse_threshold = 1e-10f;             // the reference clamps the same way (src/common.cpp:64)
	for (size_t i = 0; i < 3 * M; i++) if (!std::isfinite(target[i])) throw std::invalid_argument("goicp: non-finite coordinate in the target cloud");
	for (size_t i = 0; i < 3 * N; i++) if (!std::isfinite(source[i])) throw std::invalid_argument("goicp: non-finite coordinate in the source cloud");
	double t_mark = now_ms();
	TraceRange tr_create("goicp:create");
	static const char* const kStages[] = {"goicp:create:device_setup", "goicp:create:source_order_upload", "goicp:create:distance_transform",
	                                      "goicp:create:kd_hierarchy", "goicp:create:staging_buffers", "goicp:create:end"};
	int stage = 0;
	Trace::get().push(kStages[0]);
	struct PopLast { ~PopLast() { Trace::get().pop(); } } pop_last;     // the stage range open when init returns or throws
	auto lap = [&](const char* what) {
		if (p_.verbose) std::fprintf(stderr, "[goicp] create: %-28s %8.2f ms\n", what, now_ms() - t_mark);
		t_mark = now_ms();
		if (Trace::get().on()) { Trace::get().pop(); stage = std::min(stage + 1, 5); Trace::get().push(kStages[stage]); }
	};
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
		throw std::runtime_error("goicp: no HIP device available (this engine has no CPU fallback)");
	if (p_.device >= ndev) throw std::invalid_argument("goicp: device ordinal out of range");
	if (p_.device >= 0) dev_ = p_.device; else HIPCHK(hipGetDevice(&dev_));
	DeviceGuard guard(dev_);
	if (p_.stream_priority > 0) {
		int least = 0, greatest = 0;
		HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
		HIPCHK(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, greatest));
	} else
		HIPCHK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
	HIPCHK(hipEventCreate(&ev0_));
	HIPCHK(hipEventCreate(&ev1_));
	HIPCHK(hipStreamCreateWithFlags(&stream2_, hipStreamNonBlocking));
	lane_stream_[0] = stream_; lane_stream_[1] = stream2_;
	HIPCHK(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
	HIPCHK(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
	{ const char* e = std::getenv("GOICP_TILE_CONCURRENT"); if (e) tile_concurrent_ = std::atoi(e) != 0; }     // A/B only (tools/tile_deep.py)
	lanes_ = p_.lanes; lane_min_searches_ = std::max(2, p_.lane_min_searches);
	{ const char* e = std::getenv("GOICP_LANES"); if (e) lanes_ = std::atoi(e); }                                  // tuning only (tools/lanes_probe.py)
	{ const char* e = std::getenv("GOICP_LANE_MIN"); if (e) lane_min_searches_ = std::max(2, std::atoi(e)); }
	{ const char* e = std::getenv("GOICP_LANE_MIN_WORK"); if (e) lane_min_work_ = std::atof(e); }
	{ const char* e = std::getenv("GOICP_SOFT_OVERFLOW"); if (e) soft_overflow_ = std::atoi(e) != 0; }      // A/B only
	{ const char* e = std::getenv("GOICP_TILE_STICKY_SHARE"); if (e) tile_sticky_share_ = std::atof(e); }
	{ const char* e = std::getenv("GOICP_AUTO_LANES"); if (e) auto_lanes_ = std::min(kMaxLanes, std::max(2, std::atoi(e))); }

	h_target_.assign(target, target + 3 * M);
	if (!(p_.trim_fraction >= 0.f) || p_.trim_fraction >= 1.f) throw std::invalid_argument("goicp: trim_fraction must be in [0,1)");
	inliers_ = (int)((float)N_ * (1 - p_.trim_fraction));    // jly_goicp.cpp:201
	if (inliers_ < 1) inliers_ = 1;
	sse_thresh_ = p_.mse_threshold * (float)inliers_;         // jly_goicp.cpp:208
	icp_err_diff_ = p_.mse_threshold / 10000;        // jly_goicp.cpp:186

	// ---- search domain: the CPU path's roots (jly_goicp.cpp:44-53) unless the configured ranges narrow it ----
	rot_root_ = Node{(float)-kPI, (float)-kPI, (float)-kPI, (float)(2 * kPI), 0.f, 0.f, 0};
	trans_root_ = Node{-0.5f, -0.5f, -0.5f, 1.0f, 0.f, 0.f, 0};
	auto cube_of_box = [](const float lo[3], const float hi[3], Node* root) {
		float w = 0.f;
		for (int k = 0; k < 3; k++) {
			if (!(hi[k] >= lo[k]) || !std::isfinite(lo[k]) || !std::isfinite(hi[k])) throw std::invalid_argument("goicp: empty or non-finite search range");
			w = std::max(w, hi[k] - lo[k]);
		}
		if (!(w > 0.f)) w = 1e-6f;          // a point range: one tiny cube
		// An axis of zero width (rotation about one axis only, a fixed translation component): centred, its value would sit
		// exactly on the first split plane, and the strict in_box test would drop BOTH children -- the search would end
		// after one pop with the initial ICP pose.  Such an axis is placed a third of the way into the root instead: a
		// third is not a dyadic fraction, so the value is interior to exactly one cube at every level.
		float c[3];
		for (int k = 0; k < 3; k++) c[k] = hi[k] > lo[k] ? (lo[k] + hi[k]) / 2 - w / 2 : lo[k] - w / 3;
		root->x = c[0]; root->y = c[1]; root->z = c[2];
		root->w = w;
	};
	if (p_.use_rot_range) {
		bool full = true;
		for (int k = 0; k < 3; k++) full = full && p_.rot_min[k] <= -180.f && p_.rot_max[k] >= 180.f;
		if (!full) {
			for (int k = 0; k < 3; k++) {
				rot_lo_[k] = std::max((float)-kPI, (float)((double)p_.rot_min[k] * kPI / 180.0));
				rot_hi_[k] = std::min((float)kPI, (float)((double)p_.rot_max[k] * kPI / 180.0));
			}
			cube_of_box(rot_lo_, rot_hi_, &rot_root_);
			rot_boxed_ = true;
		}
	}
	if (p_.use_trans_range) {
		for (int k = 0; k < 3; k++) { trans_lo_[k] = p_.trans_min[k]; trans_hi_[k] = p_.trans_max[k]; }
		cube_of_box(trans_lo_, trans_hi_, &trans_root_);
		// a cubic range IS the root: nothing to cull
		trans_boxed_ = !(trans_hi_[0] - trans_lo_[0] == trans_root_.w && trans_hi_[1] - trans_lo_[1] == trans_root_.w && trans_hi_[2] - trans_lo_[2] == trans_root_.w);
	}

	// rotation uncertainty coefficients per level (jly_goicp.cpp:153-159); level l = cubes of width root/2^l
	for (int l = 0; l < kMaxRotLevel; l++) {
		float w0 = rot_root_.w;
		float sigma = (float)((double)w0 / std::pow(2.0, l) / 2.0);
		float maxAngle = (float)(kSQRT3 * (double)sigma);
		if ((double)maxAngle > kPI) maxAngle = (float)kPI;
		rot_coeff_[l] = 2 * std::sin(maxAngle / 2);
	}

	lap("validation + device setup");
	// ---- source cloud: (x,y,z,|p|), ordered for gather locality (morton_sort: 0 input order, 1 Morton curve, 2 k-d order) ----
	{
		std::vector<int32_t> perm(N_);
		for (size_t i = 0; i < N_; i++) perm[i] = (int32_t)i;
		if (p_.morton_sort == 1) {
			float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
			for (size_t i = 0; i < N_; i++)
				for (int k = 0; k < 3; k++) {
					mn[k] = std::min(mn[k], source[3 * i + k]);
					mx[k] = std::max(mx[k], source[3 * i + k]);
				}
			float ext = std::max({mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2], 1e-30f});
			std::vector<uint32_t> code(N_);
			for (size_t i = 0; i < N_; i++) {
				uint32_t c = 0;
				for (int k = 0; k < 3; k++) {
					float f = (source[3 * i + k] - mn[k]) / ext;
					uint32_t q = (uint32_t)std::min(1023.f, std::max(0.f, f * 1024.f));
					c |= part1by2(q) << k;
				}
				code[i] = c;
			}
			std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return code[a] < code[b]; });
		} else if (p_.morton_sort >= 2) {
			// k-d order: split the longest axis of the subset's bounding box at the median, rounded so that
			// the left part holds a multiple of 256 / 64 / 16 / 4 points (the largest that is smaller than the
			// subset), down to single points.  Every aligned run of 64 points (one wavefront's gathers) is
			// then a compact box-shaped surface patch, every aligned 4 and 16 lanes a sub-patch of it, and
			// every aligned 256 (one workgroup iteration) a subtree: fewer distinct DT cache lines per gather
			// instruction than a space-filling curve gives.  65 536-cube launch on the bunny: Morton 2.28 ms,
			// Hilbert 2.13 ms, 64-point clusters 1.94 ms (principal-axis splits 2.01 ms), clusters ordered
			// down to 4 points 1.85 ms, down to single points 1.84 ms.
			// one split; returns the size of the left part (0 = nothing to split)
			auto split = [&](size_t lo, size_t hi) -> size_t {
				const size_t n = hi - lo;
				if (n <= 1) return 0;
				float bmn[3] = {INFINITY, INFINITY, INFINITY}, bmx[3] = {-INFINITY, -INFINITY, -INFINITY};
				for (size_t i = lo; i < hi; i++)
					for (int k = 0; k < 3; k++) {
						bmn[k] = std::min(bmn[k], source[3 * perm[i] + k]);
						bmx[k] = std::max(bmx[k], source[3 * perm[i] + k]);
					}
				int ax = 0;
				for (int k = 1; k < 3; k++) if (bmx[k] - bmn[k] > bmx[ax] - bmn[ax]) ax = k;
				const size_t unit = n > 256 ? 256 : (n > 64 ? 64 : (n > 16 ? 16 : (n > 4 ? 4 : 1)));
				size_t nl = ((n / 2 + unit / 2) / unit) * unit;
				if (nl == 0) nl = unit;
				if (nl >= n) nl = n - (n % unit ? n % unit : unit);
				std::nth_element(perm.begin() + lo, perm.begin() + lo + nl, perm.begin() + hi, [&](int32_t a, int32_t b) {
					const float fa = source[3 * a + ax], fb = source[3 * b + ax];
					return fa < fb || (fa == fb && a < b);                    // total order: the permutation is reproducible
				});
				return nl;
			};
			auto order_range = [&](size_t lo0, size_t hi0) {
				std::vector<std::pair<size_t, size_t>> stack{{lo0, hi0}};
				while (!stack.empty()) {
					auto [lo, hi] = stack.back(); stack.pop_back();
					const size_t nl = split(lo, hi);
					if (!nl) continue;
					stack.push_back({lo + nl, hi});
					stack.push_back({lo, lo + nl});
				}
			};
			// the top of the tree level by level (its 1, 2, 4, 8 splits side by side), the (<= 16) ranges below in parallel
			std::vector<std::pair<size_t, size_t>> ranges{{0, N_}};
			while (N_ >= (1u << 13) && ranges.size() < 16) {
				std::vector<size_t> nls(ranges.size());
				parallel_tasks(16, (int)ranges.size(), [&](int t) { nls[t] = split(ranges[t].first, ranges[t].second); });
				std::vector<std::pair<size_t, size_t>> next;
				for (size_t t = 0; t < ranges.size(); t++) {
					const auto [lo, hi] = ranges[t];
					if (nls[t]) { next.push_back({lo, lo + nls[t]}); next.push_back({lo + nls[t], hi}); } else next.push_back({lo, hi});
				}
				if (next.size() == ranges.size()) break;
				ranges.swap(next);
			}
			parallel_tasks(16, (int)ranges.size(), [&](int t) { order_range(ranges[t].first, ranges[t].second); });
		}
		src_perm_ = perm;
		h_src_sorted_.resize(4 * N_);
		double cs[3] = {0, 0, 0};
		for (size_t i = 0; i < N_; i++) {
			const float* s = source + 3 * (size_t)perm[i];
			float x = s[0], y = s[1], z = s[2];
			h_src_sorted_[4 * i] = x; h_src_sorted_[4 * i + 1] = y; h_src_sorted_[4 * i + 2] = z;
			h_src_sorted_[4 * i + 3] = std::sqrt(x * x + y * y + z * z);   // normData, jly_goicp.cpp:145
			cs[0] += x; cs[1] += y; cs[2] += z;
		}
		for (int k = 0; k < 3; k++) src_centroid_[k] = (float)(cs[k] / (double)N_);
		HIPCHK(hipMalloc(&d_src_, sizeof(float4) * N_));
		HIPCHK(hipMemcpy(d_src_, h_src_sorted_.data(), sizeof(float4) * N_, hipMemcpyHostToDevice));
	}

	lap("source order + upload");
	// ---- distance transform geometry (jly_3ddt.cpp:891-923), double ----
	{
		double xMin = target[0], xMax = target[0], yMin = target[1], yMax = target[1], zMin = target[2], zMax = target[2];
		double cm[3] = {0, 0, 0};
		for (size_t i = 0; i < M_; i++) {
			double x = target[3 * i], y = target[3 * i + 1], z = target[3 * i + 2];
			if (xMin > x) xMin = x;
			if (xMax < x) xMax = x;
			if (yMin > y) yMin = y;
			if (yMax < y) yMax = y;
			if (zMin > z) zMin = z;
			if (zMax < z) zMax = z;
			cm[0] += x; cm[1] += y; cm[2] += z;
		}
		for (int k = 0; k < 3; k++) model_centroid_[k] = (float)(cm[k] / (double)M_);
		const double ex = p_.dt_expand;
		double xc = (xMin + xMax) / 2, yc = (yMin + yMax) / 2, zc = (zMin + zMax) / 2;
		xMin = xc - ex * (xMax - xc); xMax = xc + ex * (xMax - xc);
		yMin = yc - ex * (yMax - yc); yMax = yc + ex * (yMax - yc);
		zMin = zc - ex * (zMax - zc); zMax = zc + ex * (zMax - zc);
		double side = xMax - xMin > yMax - yMin ? xMax - xMin : yMax - yMin;
		side = side > zMax - zMin ? side : zMax - zMin;
		if (!(side > 0)) throw std::invalid_argument("goicp: degenerate target cloud (zero extent)");
		dt_.V = p_.dt_size;
		dt_.VB = (p_.dt_size + 3) / 4;
		dt_.layout = p_.dt_layout ? 1 : 0;
		dt_.xmin = xc - side / 2; dt_.ymin = yc - side / 2; dt_.zmin = zc - side / 2;
		dt_.scale = dt_.V / side;
		dt_.scale_f = (float)dt_.scale; dt_.xmin_f = (float)dt_.xmin; dt_.ymin_f = (float)dt_.ymin; dt_.zmin_f = (float)dt_.zmin;
		// float index fast path: 2^-24 * (scale*|min| + 3|F| + 2) bounds the rounding of min_f, scale_f,
		// (q - min_f) and the fma; shipped with a margin (4|F| + 8, x1.05)
		const double s0 = dt_.scale * std::max({std::fabs(dt_.xmin), std::fabs(dt_.ymin), std::fabs(dt_.zmin)});
		dt_.c1 = (float)(1.05 * std::ldexp(1.0, -24) * (s0 + 8.0));
		dt_.c2 = (float)(1.05 * std::ldexp(1.0, -24) * 4.0);
	}
	// ---- DT build on the GPU ----
	{
		double t0 = now_ms();
		const size_t V = dt_.V, nlin = V * V * V;
		const size_t nout = dt_.layout ? (size_t)dt_.VB * dt_.VB * dt_.VB * 64 : nlin;
		DevBuf<float> model_buf(3 * M_);
		DevBuf<int32_t> work_buf(nlin);
		float* d_model = model_buf.p;
		int32_t* d_work = work_buf.p;
		HIPCHK(hipMemcpyAsync(d_model, target, sizeof(float) * 3 * M_, hipMemcpyHostToDevice, stream_));
		if (dt_.layout) {
			HIPCHK(hipMalloc(&d_dt_, sizeof(float) * nout));
			HIPCHK(hipMemsetAsync(d_dt_, 0, sizeof(float) * nout, stream_));
		} else {
			d_dt_ = reinterpret_cast<float*>(work_buf.release());   // the linear grid is built in place: the engine owns it from here
		}
		dt_.grid = d_dt_;
		{   // table of the out-of-grid extension term, same float sqrt + double divide as the kernel's fallback
			const int n = 16384;
			std::vector<double> tab(n);
			for (int i = 0; i < n; i++) tab[i] = (double)std::sqrt((float)i) / dt_.scale;
			HIPCHK(hipMalloc(&d_overshoot_, sizeof(double) * n));
			HIPCHK(hipMemcpyAsync(d_overshoot_, tab.data(), sizeof(double) * n, hipMemcpyHostToDevice, stream_));
			HIPCHK(hipStreamSynchronize(stream_));
			dt_.overshoot = d_overshoot_; dt_.n_overshoot = n;
		}
		HIPCHK(launch_dt_build(d_model, (int)M_, dt_, d_work, d_dt_, stream_));
		if (p_.bounds_fp16 && dt_.layout == 1) {
			HIPCHK(hipMalloc(&d_dt16_, sizeof(unsigned short) * nout));
			HIPCHK(launch_dt_to_half(d_dt_, d_dt16_, nout, stream_));
			dt16_ = dt_;
			dt16_.grid = static_cast<const float*>(d_dt16_);
			dt16_.layout = 2;
		}
		HIPCHK(hipStreamSynchronize(stream_));
		dt_build_ms_ = now_ms() - t0;
	}
	lap("distance transform");
	// ---- k-d tree (64-ary box hierarchy) over the target ----
	{
		// auto = host median splits: at 1 M points the device (Morton) build is 0.3 s quicker to make, but its
		// looser boxes double every ICP pass (3.2 vs 1.65 ms) -- 0.56 s over one registration
		const bool gpu_build = p_.kd_gpu_build > 0;
		if (!gpu_build) {
			KdHost kh;
			build_kdtree(target, (int)M_, kLeafSlots, &kh);
			for (int l = 0; l < kh.K; l++) {
				HIPCHK(hipMalloc(&d_kd_boxes_[l], sizeof(float) * kh.boxes[l].size()));
				HIPCHK(hipMemcpy(d_kd_boxes_[l], kh.boxes[l].data(), sizeof(float) * kh.boxes[l].size(), hipMemcpyHostToDevice));
			}
			HIPCHK(hipMalloc(&d_kd_pts_, sizeof(float4) * kh.pts.size()));
			HIPCHK(hipMemcpy(d_kd_pts_, kh.pts.data(), sizeof(float4) * kh.pts.size(), hipMemcpyHostToDevice));
			kd_.K = kh.K;
			kd_slots_ = kh.pts.size();
		} else {
			// device build: Morton sort + bottom-up boxes, no host tree
			int K = 1;
			while (K < kMaxLevels && (long long)kLeafSlots * (1LL << (6 * K)) < (long long)M_) K++;
			if ((long long)kLeafSlots * (1LL << (6 * K)) < (long long)M_) throw std::invalid_argument("goicp: target cloud too large for the k-d tree");
			DevBuf<float> model_buf(3 * M_);
			float* d_model = model_buf.p;
			HIPCHK(hipMemcpyAsync(d_model, target, sizeof(float) * 3 * M_, hipMemcpyHostToDevice, stream_));
			for (int l = 0; l < K; l++) HIPCHK(hipMalloc(&d_kd_boxes_[l], sizeof(float) * 384 * ((size_t)1 << (6 * l))));
			HIPCHK(hipMalloc(&d_kd_pts_, sizeof(float4) * kLeafSlots * ((size_t)1 << (6 * K))));
			float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
			for (size_t i = 0; i < M_; i++)
				for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], target[3 * i + k]); mx[k] = std::max(mx[k], target[3 * i + k]); }
			const float ext = std::max({mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]});
			HIPCHK(launch_kd_build(d_model, (int)M_, K, mn, ext, d_kd_boxes_, d_kd_pts_, stream_));
			kd_.K = K;
			kd_slots_ = (size_t)kLeafSlots * ((size_t)1 << (6 * K));
		}
		for (int l = 0; l < kMaxLevels; l++) kd_.boxes[l] = d_kd_boxes_[l];
		kd_.pts = d_kd_pts_; kd_.M = (int)M_;
	}
	lap("k-d hierarchy + upload");
	if (p_.icp_point_seed) {
		// nearest-target-point table: per voxel the leaf slot of a target point whose seed voxel is nearest (the exact EDT passes
		// again, carrying their arg-min) -- the neighbour search of the ICP starts from that point instead of from a bound
		const size_t V = dt_.V, nlin = V * V * V;
		const size_t nout = dt_.layout ? (size_t)dt_.VB * dt_.VB * dt_.VB * 64 : nlin;
		const int nslots = (int)kd_slots_;                  // the slots that exist (the root group of the hierarchy is sparse: 16 x 2^D, not 16 x 64^K)
		DevBuf<int32_t> wd(nlin), wi(nlin);
		HIPCHK(hipMalloc(&d_nn_ids_, sizeof(int32_t) * nout));
		HIPCHK(hipMemsetAsync(d_nn_ids_, 0, sizeof(int32_t) * nout, stream_));          // brick padding beyond V: slot 0, never addressed
		HIPCHK(launch_nn_seed_build(d_kd_pts_, nslots, dt_, wd.p, wi.p, d_nn_ids_, stream_));
		HIPCHK(hipStreamSynchronize(stream_));
		dt_.nn_ids = d_nn_ids_;
		lap("nearest-point table");
	}
	HIPCHK(hipMalloc(&d_icp_partials_, sizeof(float) * std::max(icp_partials_floats((int)N_), (size_t)icp_trim_blocks((int)N_) * kIcpAcc)));
	if (inliers_ < (int)N_) {
		HIPCHK(hipMalloc(&d_nn_d2_, sizeof(float) * N_));
		HIPCHK(hipMalloc(&d_nn_slot_, sizeof(int) * N_));
		HIPCHK(hipMalloc(&d_include_, N_));
	}
	HIPCHK(hipMalloc(&d_icp_state_, sizeof(IcpState)));
	HIPCHK(hipMalloc(&d_icp_acc_, sizeof(unsigned long long) * kIcpAccReplicas * kIcpAcc));
	HIPCHK(hipMemsetAsync(d_icp_acc_, 0, sizeof(unsigned long long) * kIcpAccReplicas * kIcpAcc, stream_));
	for (size_t i = 0; i < N_; i++) src_radius_ = std::max(src_radius_, h_src_sorted_[4 * i + 3]);
	for (size_t i = 0; i < 3 * M_; i++) target_abs_max_ = std::max(target_abs_max_, std::fabs(target[i]));
	if (const char* e = std::getenv("GOICP_ICP_CACHE_REL")) { const float v = (float)std::atof(e); if (v > 0.f) icp_cache_rel_ = v; }     // tuning only
	if (p_.icp_nn_cache) {
		HIPCHK(hipMalloc(&d_nn_cache_, sizeof(float4) * 2 * N_));
		HIPCHK(hipMemsetAsync(d_nn_cache_, 0, sizeof(float4) * 2 * N_, stream_));      // sqrt(best2_ref) = 0: the first pass walks
	}
	HIPCHK(hipMalloc(&d_icp_ticket_, 64));
	HIPCHK(hipMemset(d_icp_ticket_, 0, 64));
	HIPCHK(hipHostMalloc(&h_icp_state_, sizeof(IcpState) * 3));      // [0] the state as uploaded / as last fetched, [1], [2] icp_run's two fetch slots
	ensure_batch(4096, 64);
	if (p_.device_queues && p_.trans_batch > 1 && p_.wide_children) {
		// the device-resident inner-BnB queues, sized for a full round of the outer search, and their pinned mirror touched
		// once (the first use of a fresh pinned block costs milliseconds -- measured 8 ms inside the first registration)
		ensure_queues(flow_mode() ? kFlowSearches : 1);
		if (flow_mode()) {
			ensure_batch(1, kFlowSearches / 2);
			HIPCHK(hipHostMalloc(&h_qinit_, sizeof(QInit) * kFlowSearches));
			HIPCHK(hipMalloc(&d_qinit_, sizeof(QInit) * kFlowSearches));
			std::memset(h_qinit_, 0, sizeof(QInit) * kFlowSearches);
		}
		std::memset(ql_[0].h_search, 0, sizeof(QSearch) * ql_[0].cap);
		HIPCHK(hipMemcpyAsync(ql_[0].d_search, ql_[0].h_search, sizeof(QSearch) * ql_[0].cap, hipMemcpyHostToDevice, stream_));
		HIPCHK(hipMemcpyAsync(ql_[0].h_search, ql_[0].d_search, sizeof(QSearch) * ql_[0].cap, hipMemcpyDeviceToHost, stream_));
		HIPCHK(hipMemcpyAsync(ql_[0].h_ctl, ql_[0].d_ctl, sizeof(QCtl), hipMemcpyDeviceToHost, stream_));
		HIPCHK(hipStreamSynchronize(stream_));
		// ... and one dummy search that stops at its root (incumbent 0): the first launch of each queue kernel loads its
		// code object, which belongs to engine creation, not to the first registration
		InnerSearch warm;
		warm.best = 0.f;
		std::vector<InnerSearch*> one{&warm};
		std::vector<Rot9> rot(1);
		const float I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
		std::memcpy(rot[0].r, I9, sizeof(I9));
		run_inner_device(one, rot);
		{
			// ... and the read-back of a batch's search records at the sizes a registration uses (tens of KB): measured on MI355X, the FIRST
			// device-to-host copy of such a size after kernels have run costs 8.6 ms (a one-time set-up inside the runtime; the 80-byte
			// read-back of the dummy search above does not trigger it, neither does the copy made before any kernel ran) -- it belongs to
			// engine creation, not to the first registration (full bunny: first run 43 -> 34 ms)
			const double tw = now_ms();
			for (size_t n = 1; n <= ql_[0].cap; n *= 2) HIPCHK(hipMemcpyAsync(ql_[0].h_search, ql_[0].d_search, sizeof(QSearch) * n, hipMemcpyDeviceToHost, stream_));
			HIPCHK(hipStreamSynchronize(stream_));
			if (p_.verbose) std::fprintf(stderr, "[goicp] create: read-back warm-up %.2f ms\n", now_ms() - tw);
		}
		if (flow_mode()) {
			// the same for the continuous-flow driver: its list-init kernel, the full-size rotation table upload and the
			// full-size read-back of the search records, once, here
			const QParams qp = queue_params();
			flow_reset();
			h_qinit_[0] = QInit{0, 0.f, 0.f, 0, -1};
			std::memcpy(h_rots_[0].r, I9, sizeof(I9));
			HIPCHK(hipMemcpyAsync(d_rots_, h_rots_, sizeof(Rot9) * (kFlowSearches / 2), hipMemcpyHostToDevice, stream_));
			HIPCHK(hipMemcpyAsync(d_qinit_, h_qinit_, sizeof(QInit) * kFlowSearches, hipMemcpyHostToDevice, stream_));
			HIPCHK(launch_bnb_init_list(ql_[0].d_search, ql_[0].d_nodes, d_qinit_, 1, qp, stream_));
			for (int r = 0; r < 3; r++) {
				HIPCHK(launch_bnb_queue(ql_[0].d_search, ql_[0].d_nodes, kFlowSearches, qp, ql_[0].d_parents[q_parity_ ^ 1], ql_[0].d_parents[q_parity_], ql_[0].d_ub, ql_[0].d_lb, ql_[0].d_scratch, ql_[0].d_ctl, q_parity_, stream_));
				HIPCHK(launch_bounds_queue(d_src_, (int)N_, bounds_dt(), d_rots_, ql_[0].d_parents[q_parity_], &ql_[0].d_ctl->n_groups[q_parity_], &ql_[0].d_ctl->work[q_parity_][0], &ql_[0].d_ctl->chunks,
				                           kFlowSearches * qp.K, inliers_, ql_[0].d_scratch, ql_[0].d_ub, ql_[0].d_lb, stream_));
				q_parity_ ^= 1;
			}
			HIPCHK(hipMemcpyAsync(ql_[0].h_ctl, ql_[0].d_ctl, sizeof(QCtl), hipMemcpyDeviceToHost, stream_));
			HIPCHK(hipMemcpyAsync(ql_[0].h_search, ql_[0].d_search, sizeof(QSearch) * kFlowSearches, hipMemcpyDeviceToHost, stream_));
			HIPCHK(hipStreamSynchronize(stream_));
			flow_reset();
			HIPCHK(hipStreamSynchronize(stream_));
		}
		cnt_ = Counters{};
		queue_rounds_ = 0;
	}

	{
		// the first launch of the ICP kernels, of the scoring launch and their first state / result read-backs also belong to engine creation
		// (measured: the first registration's ICP stretch 18.3 ms against 13.2 ms for every later one)
		const double tw = now_ms();
		const float I0[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Z0[3] = {0, 0, 0};
		float Rw[9], tw3[3];
		std::memcpy(Rw, I0, sizeof(I0)); std::memcpy(tw3, Z0, sizeof(Z0));
		int itw = 0;
		icp_run(Rw, tw3, 2 * std::max(1, p_.icp_chunk) + 1, -1e30f, &itw);       // three chunks: both fetch slots and both events are used once
		(void)eval_sse(I0, Z0);
		cnt_ = Counters{};
		if (p_.verbose) std::fprintf(stderr, "[goicp] create: ICP + scoring warm-up %.2f ms\n", now_ms() - tw);
	}
	const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	std::memcpy(optR_, I, sizeof(I)); std::memcpy(curR_, I, sizeof(I)); std::memcpy(stepR_, I, sizeof(I));
	std::memset(optT_, 0, sizeof(optT_)); std::memset(curT_, 0, sizeof(curT_)); std::memset(stepT_, 0, sizeof(stepT_));
	publish(false);
	lap("staging buffers");
}

Engine::~Engine() { release(); }

void Engine::release()
{
	int prev = -1;
	if (stream_ && hipGetDevice(&prev) == hipSuccess && prev != dev_) hipSetDevice(dev_); else prev = -1;
	if (stream_) hipStreamSynchronize(stream_);
	hipFree(d_opscratch_); d_opscratch_ = nullptr; cap_opscratch_ = 0;
	for (QLane& L : ql_) free_lane(L);
	hipFree(d_qinit_); hipHostFree(h_qinit_); d_qinit_ = nullptr; h_qinit_ = nullptr;
	hipFree(d_src_); hipFree(d_dt_); hipFree(d_overshoot_); hipFree(d_dt16_); d_dt16_ = nullptr;
	hipFree(d_nn_ids_); d_nn_ids_ = nullptr;
	for (int l = 0; l < kMaxLevels; l++) hipFree(d_kd_boxes_[l]);
	hipFree(d_kd_pts_);
	hipFree(d_cubes_); hipFree(d_rots_); hipFree(d_ub_); hipFree(d_lb_); hipFree(d_scratch_);
	hipHostFree(h_cubes_); hipHostFree(h_rots_); hipHostFree(h_ub_); hipHostFree(h_lb_);
	hipFree(d_icp_acc_); d_icp_acc_ = nullptr;
	hipFree(d_icp_partials_); hipFree(d_icp_state_); hipHostFree(h_icp_state_); hipFree(d_icp_ticket_); d_icp_ticket_ = nullptr; hipFree(d_nn_cache_); d_nn_cache_ = nullptr;
	hipFree(d_nn_d2_); hipFree(d_nn_slot_); hipFree(d_include_);
	for (Stage& st : stage_) {
		hipFree(st.d_parents); hipFree(st.d_ub);
		hipHostFree(st.h_parents); hipHostFree(st.h_ub);
		if (st.ev) hipEventDestroy(st.ev);
		st = Stage{};
	}
	if (ev0_) hipEventDestroy(ev0_);
	if (ev1_) hipEventDestroy(ev1_);
	if (stream2_) { hipStreamSynchronize(stream2_); hipStreamDestroy(stream2_); }
	for (int k = 2; k < kMaxLanes; k++) if (lane_stream_[k]) { hipStreamSynchronize(lane_stream_[k]); hipStreamDestroy(lane_stream_[k]); }
	for (int k = 0; k < kMaxLanes; k++) lane_stream_[k] = nullptr;
	if (ev_fork_) hipEventDestroy(ev_fork_);
	if (ev_join_) hipEventDestroy(ev_join_);
	stream2_ = nullptr; ev_fork_ = ev_join_ = nullptr;
	if (stream_) hipStreamDestroy(stream_);
	d_src_ = nullptr; d_dt_ = nullptr; d_overshoot_ = nullptr; d_kd_pts_ = nullptr;
	for (int l = 0; l < kMaxLevels; l++) d_kd_boxes_[l] = nullptr;
	d_cubes_ = nullptr; d_rots_ = nullptr; d_ub_ = d_lb_ = d_scratch_ = nullptr;
	h_cubes_ = nullptr; h_rots_ = nullptr; h_ub_ = h_lb_ = nullptr;
	d_icp_partials_ = nullptr; d_icp_state_ = nullptr; h_icp_state_ = nullptr;
	d_nn_d2_ = nullptr; d_nn_slot_ = nullptr; d_include_ = nullptr;
	ev0_ = ev1_ = nullptr; stream_ = nullptr;
	if (prev >= 0) hipSetDevice(prev);
}

void Engine::ensure_batch(size_t B, size_t K)
{
	if (B > cap_cubes_) {
		size_t cap = std::max<size_t>(B, cap_cubes_ * 2);
		hipStreamSynchronize(stream_);
		hipFree(d_cubes_); hipFree(d_ub_); hipFree(d_lb_);
		hipHostFree(h_cubes_); hipHostFree(h_ub_); hipHostFree(h_lb_);
		d_cubes_ = nullptr; d_ub_ = d_lb_ = nullptr; h_cubes_ = nullptr; h_ub_ = h_lb_ = nullptr; cap_cubes_ = 0;   // a throwing hipMalloc must not leave freed pointers behind
		HIPCHK(hipMalloc(&d_cubes_, sizeof(CubeRec) * cap));
		HIPCHK(hipMalloc(&d_ub_, sizeof(float) * cap));
		HIPCHK(hipMalloc(&d_lb_, sizeof(float) * cap));
		HIPCHK(hipHostMalloc(&h_cubes_, sizeof(CubeRec) * cap));
		HIPCHK(hipHostMalloc(&h_ub_, sizeof(float) * cap));
		HIPCHK(hipHostMalloc(&h_lb_, sizeof(float) * cap));
		cap_cubes_ = cap;
	}
	if (K > cap_rots_) {
		size_t cap = std::max<size_t>(K, cap_rots_ * 2);
		hipStreamSynchronize(stream_);
		hipFree(d_rots_); hipHostFree(h_rots_);
		d_rots_ = nullptr; h_rots_ = nullptr; cap_rots_ = 0;
		HIPCHK(hipMalloc(&d_rots_, sizeof(Rot9) * cap));
		HIPCHK(hipHostMalloc(&h_rots_, sizeof(Rot9) * cap));
		cap_rots_ = cap;
	}
	size_t need = bounds_scratch_floats((int)std::max<size_t>(B, 1), (int)N_, nullptr, nullptr);
	// the scratch need is not monotone in B (fewer cubes -> more point chunks): size for the worst case
	size_t worst = (size_t)2 * kGroup * (8 * ((cap_cubes_ + kGroup - 1) / kGroup) + 4096);   // groups x (<= 8 + 2048/groups) chunks
	need = std::max(need, worst);
	if (need > cap_scratch_) {
		hipStreamSynchronize(stream_);
		hipFree(d_scratch_);
		d_scratch_ = nullptr; cap_scratch_ = 0;
		HIPCHK(hipMalloc(&d_scratch_, sizeof(float) * need));
		cap_scratch_ = need;
	}
}

// ------------------------------------------------------------------------------------------------
// operators
// ------------------------------------------------------------------------------------------------
void Engine::eval_bounds_dev(const Rot9* d_rots, const CubeRec* d_cubes, int B, float* d_ub, float* d_lb, hipStream_t s, const ParentRec* d_parents)
{
	DeviceGuard guard(dev_);
	size_t need = bounds_scratch_floats(B, (int)N_, nullptr, nullptr);
	if (need > cap_scratch_) {
		// only the two streams that can still be using the old scratch, not the whole device
		HIPCHK(hipStreamSynchronize(stream_));
		if (s && s != stream_) HIPCHK(hipStreamSynchronize(s));
		hipFree(d_scratch_);
		d_scratch_ = nullptr; cap_scratch_ = 0;
		HIPCHK(hipMalloc(&d_scratch_, sizeof(float) * need));
		cap_scratch_ = need;
	}
	if (inliers_ < (int)N_)
		HIPCHK(launch_bounds_trim(d_src_, (int)N_, dt_, d_rots, d_cubes, d_parents, B, inliers_, d_ub, d_lb, s ? s : stream_));
	else
		HIPCHK(launch_bounds(d_src_, (int)N_, bounds_dt(), d_rots, d_cubes, d_parents, B, d_scratch_, d_ub, d_lb, s ? s : stream_));
	cnt_.bounds_launches++;
}

void Engine::reduce_min_dev(const float* d_v, int n, float* d_min, int* d_idx, hipStream_t s)
{
	DeviceGuard guard(dev_);
	HIPCHK(launch_reduce_min(d_v, n, d_min, d_idx, s ? s : stream_));
}

void Engine::eval_bounds_dev_grouped(const Rot9* d_rots, int nrots, const CubeRec* d_cubes, int B, float* d_ub, float* d_lb, hipStream_t s)
{
	DeviceGuard guard(dev_);
	if (inliers_ < (int)N_) { eval_bounds_dev(d_rots, d_cubes, B, d_ub, d_lb, s); return; }     // the trimmed kernel owns whole cubes: nothing to group
	if (nrots < 1 || nrots > 16) throw std::invalid_argument("goicp: grouped bounds take 1..16 rotations");
	const size_t need = bounds_scratch_floats(B, (int)N_, nullptr, nullptr);
	if (need > cap_scratch_) {
		HIPCHK(hipStreamSynchronize(stream_));
		if (s && s != stream_) HIPCHK(hipStreamSynchronize(s));
		hipFree(d_scratch_);
		d_scratch_ = nullptr; cap_scratch_ = 0;
		HIPCHK(hipMalloc(&d_scratch_, sizeof(float) * need));
		cap_scratch_ = need;
	}
	void* gs = scratch_bytes(bounds_grouped_scratch_bytes(B, nrots));
	HIPCHK(launch_bounds_grouped(d_src_, (int)N_, bounds_dt(), d_rots, nrots, d_cubes, B, gs, d_scratch_, d_ub, d_lb, s ? s : stream_));
	cnt_.bounds_launches++;
}

float Engine::time_bounds_dev(const Rot9* d_rots, const CubeRec* d_cubes, int B, float* d_ub, float* d_lb, int iters, int grouped_nrots)
{
	DeviceGuard guard(dev_);
	auto once = [&] { if (grouped_nrots > 0) eval_bounds_dev_grouped(d_rots, grouped_nrots, d_cubes, B, d_ub, d_lb, stream_); else eval_bounds_dev(d_rots, d_cubes, B, d_ub, d_lb, stream_); };
	once();   // sizes the scratch
	HIPCHK(hipStreamSynchronize(stream_));
	HIPCHK(hipEventRecord(ev0_, stream_));
	for (int i = 0; i < iters; i++) once();
	HIPCHK(hipEventRecord(ev1_, stream_));
	HIPCHK(hipEventSynchronize(ev1_));
	float ms = 0.f;
	HIPCHK(hipEventElapsedTime(&ms, ev0_, ev1_));
	return ms / (float)std::max(iters, 1);
}

void Engine::eval_bounds_batch(const float* rots9, size_t K, const CubeRec* cubes, size_t B, float* ub, float* lb)
{
	if (B == 0) return;
	DeviceGuard guard(dev_);
	for (size_t i = 0; i < B; i++)
		if (cubes[i].rot < 0 || (size_t)cubes[i].rot >= K) throw std::invalid_argument("goicp: cube rotation index out of range");
	ensure_batch(B, K);
	std::memcpy(h_rots_, rots9, sizeof(Rot9) * K);
	if (cubes != h_cubes_) std::memcpy(h_cubes_, cubes, sizeof(CubeRec) * B);
	HIPCHK(hipMemcpyAsync(d_rots_, h_rots_, sizeof(Rot9) * K, hipMemcpyHostToDevice, stream_));
	HIPCHK(hipMemcpyAsync(d_cubes_, h_cubes_, sizeof(CubeRec) * B, hipMemcpyHostToDevice, stream_));
	// an unrelated batch (fewer than half of its groups of eight are the children of one expansion) is grouped on the device
	// first: same bits, ~1.6x faster (eval_bounds_dev_grouped); the search's own batches never take this path
	bool grouped = false;
	if (B >= 256 && K <= 16 && inliers_ >= (int)N_) {
		size_t sib = 0;
		for (size_t g = 0; g + 8 <= B; g += 8) {
			const CubeRec* c = h_cubes_ + g;
			bool s = true;
			for (int j = 1; j < 8 && s; j++)
				s = c[j].rot == c[0].rot && c[j].delta == c[0].delta && c[j].coeff == c[0].coeff && c[j].tx == c[j & 1].tx && c[j].ty == c[j & 2].ty && c[j].tz == c[j & 4].tz;
			sib += s ? 1 : 0;
		}
		grouped = sib * 2 < B / 8;
	}
	if (grouped) eval_bounds_dev_grouped(d_rots_, (int)K, d_cubes_, (int)B, d_ub_, d_lb_, stream_);
	else eval_bounds_dev(d_rots_, d_cubes_, (int)B, d_ub_, d_lb_, stream_);
	HIPCHK(hipMemcpyAsync(h_ub_, d_ub_, sizeof(float) * B, hipMemcpyDeviceToHost, stream_));
	HIPCHK(hipMemcpyAsync(h_lb_, d_lb_, sizeof(float) * B, hipMemcpyDeviceToHost, stream_));
	HIPCHK(hipStreamSynchronize(stream_));
	if (ub && ub != h_ub_) std::memcpy(ub, h_ub_, sizeof(float) * B);
	if (lb && lb != h_lb_) std::memcpy(lb, h_lb_, sizeof(float) * B);
	cnt_.cubes += (long long)B;
}

void Engine::eval_bounds(const float R[9], const float* cubes4, size_t B, int level, float* ub, float* lb)
{
	if (B == 0) return;
	DeviceGuard guard(dev_);
	ensure_batch(B, 1);
	const float coeff = rot_coeff(level);
	for (size_t i = 0; i < B; i++) {
		CubeRec& c = h_cubes_[i];
		c.tx = cubes4[4 * i]; c.ty = cubes4[4 * i + 1]; c.tz = cubes4[4 * i + 2];
		c.delta = (float)(kSQRT3 / 2.0 * (double)cubes4[4 * i + 3]);   // jly_goicp.cpp:263
		c.coeff = coeff;
		c.rot = 0;
	}
	eval_bounds_batch(R, 1, h_cubes_, B, ub, lb);
}

float Engine::eval_sse(const float R[9], const float t[3])
{
	// sum_i Distance(R p_i + t)^2 (jly_goicp.cpp:100-129): one cube with centre t, no radii
	float cube[4] = {t[0], t[1], t[2], 0.f};
	float ub = 0.f, lb = 0.f;
	struct Exact { bool& f; explicit Exact(bool& x) : f(x) { f = true; } ~Exact() { f = false; } } exact(score_exact_);   // a score, not a bound: fp32 grid
	eval_bounds(R, cube, 1, -1, &ub, &lb);
	cnt_.cubes -= 1;   // a score, not a BnB cube bound
	return ub;
}

void Engine::dt_download(float* out)
{
	DeviceGuard guard(dev_);
	const size_t V = dt_.V;
	if (!dt_.layout) {
		HIPCHK(hipMemcpy(out, d_dt_, sizeof(float) * V * V * V, hipMemcpyDeviceToHost));
		return;
	}
	const size_t VB = dt_.VB, nb = VB * VB * VB * 64;
	std::vector<float> tmp(nb);
	HIPCHK(hipMemcpy(tmp.data(), d_dt_, sizeof(float) * nb, hipMemcpyDeviceToHost));
	for (size_t z = 0; z < V; z++)
		for (size_t y = 0; y < V; y++)
			for (size_t x = 0; x < V; x++) {
				size_t b = ((z >> 2) * VB + (y >> 2)) * VB + (x >> 2);
				out[(z * V + y) * V + x] = tmp[b * 64 + (((z & 3) << 4) | ((y & 3) << 2) | (x & 3))];
			}
}

void Engine::nn_query(const float* q, size_t n, int32_t* idx, float* d2)
{
	if (n == 0) return;
	DeviceGuard guard(dev_);
	// one grow-only scratch block: queries | indices | distances
	char* base = static_cast<char*>(scratch_bytes(sizeof(float) * 5 * n));
	float* dq = reinterpret_cast<float*>(base);
	int32_t* di = reinterpret_cast<int32_t*>(base + sizeof(float) * 3 * n);
	float* dd = reinterpret_cast<float*>(base + sizeof(float) * 4 * n);
	HIPCHK(hipMemcpyAsync(dq, q, sizeof(float) * 3 * n, hipMemcpyHostToDevice, stream_));
	HIPCHK(launch_nn_query(dq, (int)n, kd_, dt_, di, dd, stream_));
	HIPCHK(hipMemcpyAsync(idx, di, sizeof(int32_t) * n, hipMemcpyDeviceToHost, stream_));
	HIPCHK(hipMemcpyAsync(d2, dd, sizeof(float) * n, hipMemcpyDeviceToHost, stream_));
	HIPCHK(hipStreamSynchronize(stream_));
}

void Engine::source_transformed(const float R[9], const float t[3], float* out)
{
	// rigid transform apply on the device (kernTransform, src/goicp_kernel.cu:16-22), then back to the
	// caller's point order
	DeviceGuard guard(dev_);
	float4* d_tmp = static_cast<float4*>(scratch_bytes(sizeof(float4) * N_));
	HIPCHK(hipMemcpyAsync(d_tmp, d_src_, sizeof(float4) * N_, hipMemcpyDeviceToDevice, stream_));
	Pose pose;
	std::memcpy(pose.R, R, sizeof(pose.R));
	std::memcpy(pose.t, t, sizeof(pose.t));
	HIPCHK(launch_transform(d_tmp, (int)N_, pose, stream_));
	std::vector<float> h(4 * N_);
	HIPCHK(hipMemcpyAsync(h.data(), d_tmp, sizeof(float4) * N_, hipMemcpyDeviceToHost, stream_));
	HIPCHK(hipStreamSynchronize(stream_));
	for (size_t i = 0; i < N_; i++) {
		float* o = out + 3 * (size_t)src_perm_[i];
		o[0] = h[4 * i]; o[1] = h[4 * i + 1]; o[2] = h[4 * i + 2];
	}
}

// ------------------------------------------------------------------------------------------------
// ICP (ICP3D<float>::Run, jly_icp3d.hpp:181-295; IterativeClosestPoint3D::run, fgoicp/icp3d.cu:83-108)
// The loop state lives on the device (IcpState); iterations are queued in chunks without a host
// round trip, a converged state turns the queued remainder into no-ops.
// ------------------------------------------------------------------------------------------------
void Engine::icp_state_init(const float R[9], const float t[3], float err_diff, int carry_means, int frozen)
{
	IcpState& st = *h_icp_state_;
	icp_cache_active_ = false;          // icp_nn_cache = 2: every run starts with plain walks
	std::memset(&st, 0, sizeof(st));
	std::memcpy(st.R, R, sizeof(st.R));
	std::memcpy(st.t, t, sizeof(st.t));
	for (int i = 0; i < 3; i++) {
		st.src_centroid[i] = src_centroid_[i];
		st.cm[i] = model_centroid_[i];
		st.cq[i] = R[3 * i] * src_centroid_[0] + R[3 * i + 1] * src_centroid_[1] + R[3 * i + 2] * src_centroid_[2] + t[i];
	}
	st.err = -1.f;
	st.err_diff_n = err_diff * (float)inliers_;    // jly_icp3d.hpp:255: err_diff * num
	st.n = (float)inliers_;                        // means over the num correspondences used (the reference divides by n, App. B-12)
	st.carry_means = carry_means;
	st.frozen = frozen;
	{
		// fixed-point scale of the small-cloud pass: every term is a coordinate difference, a product of two, or a squared
		// distance between a moved source point and the target, all below L^2 with L the sum of the extents; N of them must
		// fit 2^62.  ICP moves the cloud towards the target, so the start pose bounds the run.
		const double tl = std::sqrt((double)t[0] * t[0] + (double)t[1] * t[1] + (double)t[2] * t[2]);
		const double L = 2.0 * ((double)src_radius_ + tl + std::sqrt(3.0) * (double)target_abs_max_ + 1.0);
		int e = (int)std::floor(std::log2(4.6e18 / ((double)std::max<size_t>(N_, 1) * L * L)));
		e = std::max(-60, std::min(60, e));
		st.acc_scale = std::ldexp(1.0f, e);
		st.acc_inv = std::ldexp(1.0f, -e);
	}
	HIPCHK(hipMemcpyAsync(d_icp_state_, h_icp_state_, sizeof(IcpState), hipMemcpyHostToDevice, stream_));
}

void Engine::icp_launch_one()
{
	if (inliers_ < (int)N_)
		HIPCHK(launch_icp_iteration_trim(d_src_, (int)N_, inliers_, d_icp_state_, kd_, dt_, d_nn_d2_, d_nn_slot_, d_include_, d_icp_partials_, stream_));
	else
		HIPCHK(launch_icp_iteration(d_src_, (int)N_, d_icp_state_, kd_, dt_, d_icp_partials_, p_.icp_fused ? d_icp_ticket_ : nullptr,
		                            (p_.icp_nn_cache == 1 || icp_cache_active_) ? d_nn_cache_ : nullptr, count_hits_ ? d_icp_ticket_ + 8 : nullptr, stream_, d_icp_acc_));
}

void Engine::icp_state_fetch()
{
	HIPCHK(hipMemcpyAsync(h_icp_state_, d_icp_state_, sizeof(IcpState), hipMemcpyDeviceToHost, stream_));
	HIPCHK(hipStreamSynchronize(stream_));
}

float Engine::icp_run(float R[9], float t[3], int max_iter, float err_diff, int* iters_out)
{
	DeviceGuard guard(dev_);
	icp_state_init(R, t, err_diff, 1, 0);
	// Chunks of iterations, one chunk ahead: while the host waits for and looks at the state after chunk k, chunk k+1 is
	// already queued, so the GPU never idles between chunks (a blocking fetch per chunk cost ~4 us per iteration on the
	// bunny).  Once the state has converged the queued launches are no-ops (every kernel tests the flag first).
	const int chunk = std::max(1, p_.icp_chunk);
	int queued = 0;
	IcpState* slots = h_icp_state_ + 1;
	hipEvent_t evs[2] = {ev0_, ev1_};
	auto submit = [&](int slot) -> bool {
		if (queued >= max_iter) return false;
		const int k = std::min(chunk, max_iter - queued);
		for (int i = 0; i < k; i++) icp_launch_one();
		queued += k;
		HIPCHK(hipMemcpyAsync(&slots[slot], d_icp_state_, sizeof(IcpState), hipMemcpyDeviceToHost, stream_));
		HIPCHK(hipEventRecord(evs[slot], stream_));
		return true;
	};
	int cur = 0;
	bool have = submit(0);
	const IcpState* fin = nullptr;
	float prev_err = -1.f;
	while (have) {
		const bool next = submit(cur ^ 1);
		HIPCHK(hipEventSynchronize(evs[cur]));
		fin = &slots[cur];
		if (fin->converged || cancel_.load()) break;
		// icp_nn_cache = 2: the exact walk-skipping cache pays once the cloud has nearly stopped moving (the tail of a run: a cached neighbour stays
		// provably nearest for many iterations) and loses while it still moves (every miss is a slower 2-nearest walk) -- so it is switched on, for
		// the chunks queued from here on, when the error fell by less than icp_cache_rel_ over the last chunk.  Exact either way: bit-identical states
		if (p_.icp_nn_cache == 2 && d_nn_cache_ && !icp_cache_active_ && prev_err > 0.f && fin->err > 0.f && (prev_err - fin->err) < icp_cache_rel_ * fin->err)
			icp_cache_active_ = true;
		prev_err = fin->err;
		have = next;
		cur ^= 1;
	}
	if (fin) *h_icp_state_ = *fin; else icp_state_fetch();      // max_iter <= 0: the state as uploaded
	const IcpState& st = *h_icp_state_;
	std::memcpy(R, st.R, sizeof(st.R));
	std::memcpy(t, st.t, sizeof(st.t));
	if (iters_out) *iters_out = st.iters;
	cnt_.icp_iters += st.passes;
	cnt_.icp_runs++;
	return st.err_new;
}

float Engine::time_icp_pass(const float R[9], const float t[3], int iters, bool cached)
{
	DeviceGuard guard(dev_);
	// frozen state: every pass does the same work.  cached = false: the neighbour cache is bypassed, every query walks the
	// tree (the cost of a pass at a new pose); cached = true: the pose repeats, so after the first pass every query hits
	const int keep = p_.icp_nn_cache;
	struct Restore { int& r; int v; ~Restore() { r = v; } } restore{p_.icp_nn_cache, keep};
	if (!cached) p_.icp_nn_cache = 0;
	else if (!d_nn_cache_) throw std::invalid_argument("goicp: the neighbour cache is disabled for this engine");
	else p_.icp_nn_cache = 1;
	icp_state_init(R, t, 0.f, 0, 1);
	icp_launch_one();
	HIPCHK(hipStreamSynchronize(stream_));
	HIPCHK(hipEventRecord(ev0_, stream_));
	for (int i = 0; i < iters; i++) icp_launch_one();
	HIPCHK(hipEventRecord(ev1_, stream_));
	HIPCHK(hipEventSynchronize(ev1_));
	float ms = 0.f;
	HIPCHK(hipEventElapsedTime(&ms, ev0_, ev1_));
	return ms / (float)std::max(iters, 1);
}

double Engine::probe_gather(int mode, size_t window_bytes)
{
	DeviceGuard guard(dev_);
	unsigned window = 4096;
	while ((size_t)window * 4 < window_bytes && window < (1u << 30)) window <<= 1;
	const size_t nfl = dt_.layout ? (size_t)dt_.VB * dt_.VB * dt_.VB * 64 : (size_t)dt_.V * dt_.V * dt_.V;
	while ((size_t)window > nfl) window >>= 1;
	if (mode >= 2 && window < 16384u) window = 16384u;          // k runs 256 floats apart need 64 KiB
	int cus = 256;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, dev_) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
	const int blocks = mode == 2 ? cus * 2 : cus * 8, iters = 256;        // mode 2: 64 KiB of LDS per workgroup -> two per CU
	float* sink = static_cast<float*>(scratch_bytes(64));
	HIPCHK(launch_probe_gather(dt_, mode, window, blocks, iters, sink, stream_));   // warm-up
	HIPCHK(hipStreamSynchronize(stream_));
	const int reps = 5;
	HIPCHK(hipEventRecord(ev0_, stream_));
	for (int r = 0; r < reps; r++) HIPCHK(launch_probe_gather(dt_, mode, window, blocks, iters, sink, stream_));
	HIPCHK(hipEventRecord(ev1_, stream_));
	HIPCHK(hipEventSynchronize(ev1_));
	float ms = 0.f;
	HIPCHK(hipEventElapsedTime(&ms, ev0_, ev1_));
	const double lookups = (double)reps * blocks * 256.0 * iters * 8.0;
	return lookups / ((double)ms * 1e-3);
}

void debug_kabsch(const float H[9], float R[9])
{
	DevBuf<float> buf(18);
	HIPCHK(hipMemcpy(buf.p, H, sizeof(float) * 9, hipMemcpyHostToDevice));
	HIPCHK(launch_kabsch_debug(buf.p, buf.p + 9, nullptr));
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpy(R, buf.p + 9, sizeof(float) * 9, hipMemcpyDeviceToHost));
}

void Engine::debug_bounds_tile(const float* rots9, const float* parents4, int nseg, int n, int level, int chunks, float* ub_tile, float* lb_tile,
                               float* ub_direct, float* lb_direct, float ms[2], unsigned stats[2])
{
	DeviceGuard guard(dev_);
	if (nseg < 1 || n < 1 || n > 64 || chunks < 1) throw std::invalid_argument("goicp: debug_bounds_tile: bad shape");
	const size_t G = (size_t)nseg * n, B = G * kGroup;
	struct Seg { int off, n, rot; };
	std::vector<ParentRec> par(G);
	std::vector<Seg> segs((size_t)nseg);
	const float coeff = rot_coeff(level);
	for (int i = 0; i < nseg; i++) {
		segs[(size_t)i] = Seg{i * n, n, i};
		for (int e = 0; e < n; e++) {
			const float* q = parents4 + ((size_t)i * n + e) * 4;
			par[(size_t)i * n + e] = ParentRec{q[0], q[1], q[2], q[3], coeff, i};
		}
	}
	int g2 = 0, c2 = 0;
	const size_t sc_direct = bounds_scratch_floats((int)B, (int)N_, &g2, &c2), sc_tile = G * (size_t)chunks * 2 * kGroup;
	DevBuf<ParentRec> d_par(G);
	DevBuf<Seg> d_seg((size_t)nseg);
	DevBuf<Rot9> d_rot((size_t)nseg);
	DevBuf<float> d_out(4 * B), d_sc(std::max(sc_direct, sc_tile) + 64);
	DevBuf<unsigned> d_stats(2);
	HIPCHK(hipMemcpyAsync(d_par.p, par.data(), sizeof(ParentRec) * G, hipMemcpyHostToDevice, stream_));
	HIPCHK(hipMemcpyAsync(d_seg.p, segs.data(), sizeof(Seg) * (size_t)nseg, hipMemcpyHostToDevice, stream_));
	HIPCHK(hipMemcpyAsync(d_rot.p, rots9, sizeof(Rot9) * (size_t)nseg, hipMemcpyHostToDevice, stream_));
	HIPCHK(hipMemsetAsync(d_stats.p, 0, sizeof(unsigned) * 2, stream_));
	float* t_ub = d_out.p; float* t_lb = d_out.p + B; float* r_ub = d_out.p + 2 * B; float* r_lb = d_out.p + 3 * B;
	const int reps = 5;
	for (int pass = 0; pass < 2; pass++) {              // pass 0 warms up (and counts the sub-patches), pass 1 is timed
		HIPCHK(hipEventRecord(ev0_, stream_));
		for (int r = 0; r < (pass ? reps : 1); r++)
			HIPCHK(launch_bounds_tile(d_src_, (int)N_, dt_, d_rot.p, d_par.p, d_seg.p, nseg, n, chunks, d_sc.p, t_ub, t_lb, pass ? nullptr : d_stats.p, stream_));
		HIPCHK(hipEventRecord(ev1_, stream_));
		HIPCHK(hipEventSynchronize(ev1_));
		if (pass) { HIPCHK(hipEventElapsedTime(&ms[0], ev0_, ev1_)); ms[0] /= reps; }
	}
	for (int pass = 0; pass < 2; pass++) {
		HIPCHK(hipEventRecord(ev0_, stream_));
		for (int r = 0; r < (pass ? reps : 1); r++)
			HIPCHK(launch_bounds(d_src_, (int)N_, dt_, d_rot.p, nullptr, d_par.p, (int)B, d_sc.p, r_ub, r_lb, stream_));
		HIPCHK(hipEventRecord(ev1_, stream_));
		HIPCHK(hipEventSynchronize(ev1_));
		if (pass) { HIPCHK(hipEventElapsedTime(&ms[1], ev0_, ev1_)); ms[1] /= reps; }
	}
	HIPCHK(hipMemcpy(ub_tile, t_ub, sizeof(float) * B, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(lb_tile, t_lb, sizeof(float) * B, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(ub_direct, r_ub, sizeof(float) * B, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(lb_direct, r_lb, sizeof(float) * B, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(stats, d_stats.p, sizeof(unsigned) * 2, hipMemcpyDeviceToHost));
}

void Engine::debug_queue_expand(const float R[9], int level, const float* parents4, int n, float* ub0, float* lb0, float* ub1, float* lb1, int info[2])
{
	DeviceGuard guard(dev_);
	if (n < 1 || n > kQueueRoundPop) throw std::invalid_argument("goicp: debug_queue_expand takes 1..128 nodes");
	if (!(p_.device_queues && p_.trans_batch > 1 && p_.wide_children)) throw std::invalid_argument("goicp: debug_queue_expand needs the device-queue configuration");
	if (inliers_ < (int)N_) throw std::invalid_argument("goicp: debug_queue_expand: untrimmed engines only");
	ensure_queues(2);
	ensure_batch(1, 1);
	std::memcpy(h_rots_[0].r, R, sizeof(float) * 9);
	HIPCHK(hipMemcpyAsync(d_rots_, h_rots_, sizeof(Rot9), hipMemcpyHostToDevice, stream_));
	const bool twins = p_.twin_fusion && ql_[0].d_psearch[0] != nullptr;
	// two searches whose queues hold exactly the given nodes (lower bound 0: all of them pass the stop rule against a huge incumbent and,
	// n <= K, all are selected; the list keeps the queue order)
	std::vector<QNode> nodes((size_t)n);
	for (int i = 0; i < n; i++) nodes[(size_t)i] = QNode{parents4[4 * i], parents4[4 * i + 1], parents4[4 * i + 2], parents4[4 * i + 3], 0.f, 0.f};
	for (int s = 0; s < 2; s++) {
		QSearch& q = ql_[0].h_search[s];
		std::memset(&q, 0, sizeof(q));
		q.best = 1e30f; q.coeff = s ? rot_coeff(level) : 0.f; q.rot = 0; q.count = n; q.min_ub = INFINITY; q.twin = twins ? (s ^ 1) : -1;
		HIPCHK(hipMemcpyAsync(ql_[0].d_nodes + (size_t)s * kQueueCap, nodes.data(), sizeof(QNode) * (size_t)n, hipMemcpyHostToDevice, stream_));
	}
	HIPCHK(hipMemcpyAsync(ql_[0].d_search, ql_[0].h_search, sizeof(QSearch) * 2, hipMemcpyHostToDevice, stream_));
	std::memset(ql_[0].h_ctl, 0, sizeof(QCtl));
	ql_[0].h_ctl->tile_chunks = 1;
	HIPCHK(hipMemcpyAsync(ql_[0].d_ctl, ql_[0].h_ctl, sizeof(QCtl), hipMemcpyHostToDevice, stream_));
	QParams qp = queue_params();
	qp.K = std::max(n, 1); qp.kmax = kQueueRoundPop; qp.tile_on = 0; qp.stale_widen = 0; qp.stale_compact = 0;
	const int parity = 0, max_groups = 2 * n;
	HIPCHK(launch_bnb_queue(ql_[0].d_search, ql_[0].d_nodes, 2, qp, ql_[0].d_parents[parity ^ 1], ql_[0].d_parents[parity], ql_[0].d_ub, ql_[0].d_lb, ql_[0].d_scratch, ql_[0].d_ctl, parity, stream_, nullptr,
	                        twins ? ql_[0].d_psearch[parity] : nullptr));
	HIPCHK(launch_bounds_queue(d_src_, (int)N_, bounds_dt(), d_rots_, ql_[0].d_parents[parity], &ql_[0].d_ctl->n_groups[parity], &ql_[0].d_ctl->work[parity][0], &ql_[0].d_ctl->chunks, max_groups,
	                           inliers_, ql_[0].d_scratch, ql_[0].d_ub, ql_[0].d_lb, stream_, twins ? ql_[0].d_search : nullptr, twins ? ql_[0].d_psearch[parity] : nullptr, nullptr));
	HIPCHK(hipMemcpyAsync(ql_[0].h_ctl, ql_[0].d_ctl, sizeof(QCtl), hipMemcpyDeviceToHost, stream_));
	HIPCHK(hipMemcpyAsync(ql_[0].h_search, ql_[0].d_search, sizeof(QSearch) * 2, hipMemcpyDeviceToHost, stream_));
	HIPCHK(hipStreamSynchronize(stream_));
	if (ql_[0].h_ctl->overflow || ql_[0].h_ctl->n_groups[parity] != 2 * n || ql_[0].h_search[0].n_parents != n || ql_[0].h_search[1].n_parents != n)
		throw std::logic_error("goicp: debug_queue_expand: the round did not list every node");
	const int chunks = ql_[0].h_ctl->chunks;
	info[0] = chunks; info[1] = twins ? 1 : 0;
	std::vector<ParentRec> listed((size_t)2 * n);
	HIPCHK(hipMemcpy(listed.data(), ql_[0].d_parents[parity], sizeof(ParentRec) * 2 * (size_t)n, hipMemcpyDeviceToHost));
	std::vector<float> ub((size_t)16 * n), lb((size_t)16 * n), part;
	if (chunks > 1) {
		part.resize((size_t)2 * n * chunks * 2 * kGroup);
		HIPCHK(hipMemcpy(part.data(), ql_[0].d_scratch, sizeof(float) * part.size(), hipMemcpyDeviceToHost));
	} else {
		HIPCHK(hipMemcpy(ub.data(), ql_[0].d_ub, sizeof(float) * 16 * (size_t)n, hipMemcpyDeviceToHost));
		HIPCHK(hipMemcpy(lb.data(), ql_[0].d_lb, sizeof(float) * 16 * (size_t)n, hipMemcpyDeviceToHost));
	}
	for (int s = 0; s < 2; s++) {
		const int off = ql_[0].h_search[s].parent_off;
		float* ou = s ? ub1 : ub0; float* ol = s ? lb1 : lb0;
		for (int e = 0; e < n; e++) {
			const ParentRec& pr = listed[(size_t)off + e];
			if (pr.x != parents4[4 * e] || pr.y != parents4[4 * e + 1] || pr.z != parents4[4 * e + 2] || pr.w != parents4[4 * e + 3])
				throw std::logic_error("goicp: debug_queue_expand: the list is not in queue order");
			for (int c = 0; c < kGroup; c++) {
				if (chunks > 1) {
					// the chunk partials, added in chunk order as the next round's digest does (bnbqueue.hip)
					float a = 0.f, b = 0.f;
					for (int j = 0; j < chunks; j++) {
						const float* sp = part.data() + ((size_t)(off + e) * chunks + j) * (2 * kGroup);
						a += sp[c]; b += sp[kGroup + c];
					}
					ou[8 * e + c] = a; ol[8 * e + c] = b;
				} else { ou[8 * e + c] = ub[(size_t)8 * (off + e) + c]; ol[8 * e + c] = lb[(size_t)8 * (off + e) + c]; }
			}
		}
	}
}

long long Engine::debug_cache_hits(const float R[9], const float t[3])
{
	// two scoring passes at the same pose: the second one's queries should all hit the neighbour cache
	DeviceGuard guard(dev_);
	if (!d_nn_cache_ || !p_.icp_nn_cache) return -1;
	icp_state_init(R, t, 0.f, 0, 1);
	icp_launch_one();
	HIPCHK(hipMemsetAsync(d_icp_ticket_ + 8, 0, sizeof(int), stream_));
	count_hits_ = true;
	struct Off { bool& b; ~Off() { b = false; } } off{count_hits_};
	icp_launch_one();
	int hits = 0;
	HIPCHK(hipMemcpyAsync(&hits, d_icp_ticket_ + 8, sizeof(int), hipMemcpyDeviceToHost, stream_));
	HIPCHK(hipStreamSynchronize(stream_));
	return hits;
}

void Engine::icp_step()
{
	DeviceGuard guard(dev_);
	// one iteration from the current step pose, fresh means, standard Kabsch (icp_kernel.cu:219-279)
	icp_state_init(stepR_, stepT_, 0.f, 0, 0);
	icp_launch_one();
	icp_state_fetch();
	std::memcpy(stepR_, h_icp_state_->R, sizeof(stepR_));
	std::memcpy(stepT_, h_icp_state_->t, sizeof(stepT_));
	cnt_.icp_iters++;
	std::lock_guard<std::mutex> lk(mtx_);
	std::memcpy(snap_.curR, stepR_, sizeof(stepR_));
	std::memcpy(snap_.curT, stepT_, sizeof(stepT_));
	std::memcpy(snap_.optR, stepR_, sizeof(stepR_));
	std::memcpy(snap_.optT, stepT_, sizeof(stepT_));
	snap_.best_sse = h_icp_state_->err_new;
	snap_.counters = cnt_;
}

// ------------------------------------------------------------------------------------------------
// inner (translation) BnB, batched across searches
// ------------------------------------------------------------------------------------------------
void Engine::ensure_stage(int k, size_t B)
{
	Stage& st = stage_[k];
	if (!st.ev) HIPCHK(hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
	if (B <= st.cap) return;
	const size_t cap = std::max<size_t>(B, st.cap * 2);
	hipFree(st.d_parents); hipFree(st.d_ub);
	hipHostFree(st.h_parents); hipHostFree(st.h_ub);
	st.d_parents = nullptr; st.d_ub = nullptr; st.h_parents = nullptr; st.h_ub = nullptr; st.cap = 0;
	HIPCHK(hipMalloc(&st.d_parents, sizeof(ParentRec) * (cap / 8 + 1)));
	HIPCHK(hipMalloc(&st.d_ub, sizeof(float) * 2 * cap));                 // ub[B] | lb[B]: one copy back per round
	HIPCHK(hipHostMalloc(&st.h_parents, sizeof(ParentRec) * (cap / 8 + 1)));
	HIPCHK(hipHostMalloc(&st.h_ub, sizeof(float) * 2 * cap));
	st.cap = cap;
}

void Engine::free_lane(QLane& L)
{
	hipFree(L.d_search); hipHostFree(L.h_search); hipFree(L.d_nodes);
	for (int k = 0; k < 2; k++) { hipFree(L.d_parents[k]); hipFree(L.d_psearch[k]); hipFree(L.tile.parents[k]); hipFree(L.tile.segs[k]); if (L.ev_ctl[k]) hipEventDestroy(L.ev_ctl[k]); }
	hipFree(L.sort.keys); hipFree(L.sort.order); hipFree(L.sort.hist); hipFree(const_cast<float4*>(L.sort.cen));
	hipFree(L.d_ub); hipFree(L.d_lb); hipFree(L.d_scratch); hipFree(L.d_ctl); hipHostFree(L.h_ctl);
	hipFree(L.tile.ub); hipFree(L.tile.lb); hipFree(L.tile.scratch);
	L = QLane{};          // the stream is the engine's (stream_ / stream2_), not the lane's to destroy
}

void Engine::ensure_queues(size_t nsearch) { ensure_lane(0, nsearch); }

void Engine::ensure_lane(int li, size_t nsearch)
{
	QLane& L = ql_[li];
	if (!lane_stream_[li]) HIPCHK(hipStreamCreateWithFlags(&lane_stream_[li], hipStreamNonBlocking));     // lanes 2.. : on first use (a stream costs ~2 ms to create)
	L.stream = lane_stream_[li];
	if (nsearch <= L.cap) return;
	// first use: room for a full round of the outer search (rot_batch parents x 8 children x {ub, lb} pass) -- growing in
	// steps would re-allocate the 196 KB-per-search slabs several times in the first rounds
	size_t cap = std::max<size_t>(L.cap, p_.wide_children ? (size_t)16 * (size_t)std::max(1, p_.rot_batch) : 16);
	while (cap < nsearch) cap *= 2;
	HIPCHK(hipStreamSynchronize(L.stream));
	hipFree(L.d_search); hipHostFree(L.h_search); hipFree(L.d_nodes); hipFree(L.d_parents[0]); hipFree(L.d_parents[1]); hipFree(L.d_psearch[0]); hipFree(L.d_psearch[1]);
	hipFree(L.sort.keys); hipFree(L.sort.order); hipFree(L.sort.hist); hipFree(const_cast<float4*>(L.sort.cen)); L.sort = QSort{};
	hipFree(L.d_ub); hipFree(L.d_lb); hipFree(L.d_scratch);
	for (int k = 0; k < 2; k++) { hipFree(L.tile.parents[k]); hipFree(L.tile.segs[k]); }
	hipFree(L.tile.ub); hipFree(L.tile.lb); hipFree(L.tile.scratch); L.tile = QTile{};
	L.d_search = nullptr; L.h_search = nullptr; L.d_nodes = nullptr; L.d_parents[0] = L.d_parents[1] = nullptr; L.d_psearch[0] = L.d_psearch[1] = nullptr;
	L.d_ub = L.d_lb = L.d_scratch = nullptr; L.cap = 0;
	const size_t max_groups = cap * kQueueRoundPop;          // what the round's lists hold; QParams::kmax keeps (searches running) x (their steps) inside
	L.list_cap = (int)max_groups;
	// segments (<= 64 expansions of one search) of the tile list: every search contributes floor(n / 64) full ones and at most one partial
	L.seg_cap = (int)(max_groups / 64 + cap);
	HIPCHK(hipMalloc(&L.d_search, sizeof(QSearch) * cap));
	HIPCHK(hipHostMalloc(&L.h_search, sizeof(QSearch) * cap));
	HIPCHK(hipMalloc(&L.d_nodes, sizeof(QNode) * cap * kQueueCap));          // 196 KB per search; HBM is not the scarce resource here
	for (int k = 0; k < 2; k++) HIPCHK(hipMalloc(&L.d_parents[k], sizeof(ParentRec) * max_groups));
	for (int k = 0; k < 2; k++) HIPCHK(hipMalloc(&L.d_psearch[k], sizeof(int) * max_groups));
	HIPCHK(hipMalloc(&L.d_ub, sizeof(float) * max_groups * kGroup));
	HIPCHK(hipMalloc(&L.d_lb, sizeof(float) * max_groups * kGroup));
	// footprint-ordered items for the large rounds (lean grids, untrimmed, clouds of 4..16 chunks of 4 096 points).  Measured, registration in ms,
	// chunk 2 048 | 3 072 | 4 096 | 6 144 | search order: bunny 34.4 | 33.1 | 33.8 | 34.1 | 36.8 (run-to-run +-0.6); bunny mse 1e-4 280 | 277 | 280 | 276 | 295;
	// synthetic 40 k mse 3e-5 798 | 758 | 714 | 725 | 921; spanner 150 k mse 2e-5 223 | 212 | 206 | 201 | 195 -- above ~64 k points the unsorted
	// launch's large chunks win, so the feature stops there
	// round 4: the cloud is cut into TEN chunks (rounded up to 256 points) rather than into chunks of 4 096 points -- re-swept with the threshold at
	// 256 expansions (tools/sort_threshold_probe.py chunks / chunks2, registration in ms): bunny (30 k) 2 048 | 2 560 | 3 072 | 4 096 points =
	// 32.8 | 31.8-32.4 | 32.5 | 34.3, bunny mse 1e-4 271 | 266 | 267 | 270, synthetic 40 k at mse 3e-5 -- | 770 | 765 | 720 (3 584: 744), every second
	// bunny point (15 k) 1 280 .. 4 096: 26.3-27.4, flat: ten chunks is 3 072 / 4 096 / 1 536 points there
	int kSortChunkPts = (int)(((N_ + 9) / 10 + 255) / 256 * 256);
	if (const char* e = std::getenv("GOICP_SORT_CHUNK_PTS")) { const int v = std::atoi(e); if (v >= 256) kSortChunkPts = v; }     // tuning only
	const int sort_chunks = (int)((N_ + kSortChunkPts - 1) / kSortChunkPts);
	if (p_.sort_items && bounds_uses_lean(bounds_dt()) && inliers_ >= (int)N_ && N_ >= 12288 && N_ <= 65536 && sort_chunks >= 4 && sort_chunks <= 16) {
		float4* cen = nullptr;
		HIPCHK(hipMalloc(&cen, sizeof(float4) * sort_chunks));
		HIPCHK(launch_chunk_centroids(d_src_, (int)N_, kSortChunkPts, cen, L.stream));
		L.sort.cen = cen;
		HIPCHK(hipMalloc(&L.sort.keys, sizeof(unsigned) * max_groups * sort_chunks));
		HIPCHK(hipMalloc(&L.sort.order, sizeof(unsigned) * max_groups * sort_chunks));
		HIPCHK(hipMalloc(&L.sort.hist, qsort_hist_bytes()));
		HIPCHK(hipMemsetAsync(L.sort.hist, 0, qsort_hist_bytes(), L.stream));     // kept zero between uses by the kernels themselves
		L.sort.chunk_pts = kSortChunkPts; L.sort.chunks = sort_chunks; L.sort.min_groups = 256;
		if (const char* e = std::getenv("GOICP_SORT_MIN_GROUPS")) { const int v = std::atoi(e); if (v > 0) L.sort.min_groups = v; }     // tuning only (tools/tune_e2e.py); the default is the measured optimum                    // rounds from this many expansions.  Round 3 (512 | 1024 | 2048 | 4096): 35.0 | 34.5 | 33.6 | 37.3 ms.  Re-swept in round 4 with the
		// twin lists and the 4 096-point chunks in place (tools/sort_threshold_probe.py, median of 7): 1 | 128 | 512 | 1024 | 2048 | 4096 | off = 33.8 | 33.3 | 33.5 | 33.6 | 34.4 | 36.6 | 36.3 ms;
		// mse 1e-4 / 3e-5 (0.27 / 6.8 s) flat between 128, 256 and 2048 -- so round 1 of a large batch (230 roots x two passes) is sorted too
		L.sort.shift = qsort_shift(dt_.V);           // 16-voxel cells (32-voxel cells: 35.1 ms)
	}
	HIPCHK(hipMalloc(&L.d_scratch, sizeof(float) * bounds_queue_scratch_floats((int)max_groups, L.sort.order ? L.sort.chunks : 0)));
	if (tiles_usable()) {
		// the second expansion list of a round (LDS-staged DT tiles): same capacity as the direct list
		for (int k = 0; k < 2; k++) {
			HIPCHK(hipMalloc(&L.tile.parents[k], sizeof(ParentRec) * max_groups));
			HIPCHK(hipMalloc(&L.tile.segs[k], sizeof(TileSeg) * L.seg_cap));
		}
		HIPCHK(hipMalloc(&L.tile.ub, sizeof(float) * max_groups * kGroup));
		HIPCHK(hipMalloc(&L.tile.lb, sizeof(float) * max_groups * kGroup));
		HIPCHK(hipMalloc(&L.tile.scratch, sizeof(float) * bounds_tile_queue_scratch_floats(L.seg_cap)));
	}
	if (!L.d_ctl) {
		HIPCHK(hipMalloc(&L.d_ctl, sizeof(QCtl)));
		HIPCHK(hipHostMalloc(&L.h_ctl, sizeof(QCtl) * 2));
		std::memset(L.h_ctl, 0, sizeof(QCtl) * 2);
		for (int k = 0; k < 2; k++) HIPCHK(hipEventCreateWithFlags(&L.ev_ctl[k], hipEventDisableTiming));
	}
	L.cap = cap;
}

// The inner searches with their queues on the device: a round = bnb_queue_kernel (digest the previous round's
// bounds, select the next expansions) + the bound evaluation of the listed expansions; the host queues rounds and
// looks at one word every few rounds.  Same bounds, same stop and prune rules as run_inner_host.
bool Engine::run_inner_device(std::vector<InnerSearch*>& searches, const std::vector<Rot9>& rots)
{
	TraceRange tr("goicp:inner_bnb_rounds");
	const size_t S = searches.size(), nrot = rots.size();
	const int K = std::min(std::max(1, p_.trans_batch), kQueueRoundPop);
	ensure_batch(1, nrot);
	std::memcpy(h_rots_, rots.data(), sizeof(Rot9) * nrot);
	HIPCHK(hipMemcpyAsync(d_rots_, h_rots_, sizeof(Rot9) * nrot, hipMemcpyHostToDevice, stream_));
	// Lanes: the searches of a batch are independent of each other (own queue, own incumbent; only the two passes of one rotation child --
	// twins, same rotation slot -- share loads), so a large batch is cut into lanes by rotation slot and each lane runs its own lock-step rounds
	// on its own stream with its own lists and control block.  Same bounds, same stop and prune rules per search; what changes is that one
	// lane's dependent launches (queue kernel -> sort -> bound evaluation, each draining before the next starts) run beside the others'.
	// That pays when a round is throughput-bound (its kernels' drain tails are what the other lanes fill) and costs when it is latency-bound
	// (the parts of a round take longer than the whole).  Measured (tools/lanes_probe.py; one lane -> two always -> auto with three):
	// rounds of 416 M point-expansions (bunny, mse 3e-5) 6.74 -> 5.71 -> 5.71 s, 183 M (synthetic 40 k, mse 3e-5) 721 -> 669 -> 649 ms, 90 M (bunny,
	// mse 1e-4) 269 -> 264 -> 262 ms, 39 M (3 k points, mse 3e-5) 1.10 -> 1.22 -> 1.02 s (its heavy batches only), 20 M (the default bunny
	// registration) 33.7 -> 34.2 -> 33.1 ms (never cut).  So lanes = 0 (auto) cuts a batch when the PREVIOUS batch's mean round was at least
	// lane_min_work_ point-expansions (swept 16 / 32 / 64 / 128 / 256 M: 64 M) -- a count, not a time: the choice is deterministic
	const bool lanes_wanted = lanes_ >= 2 || (lanes_ == 0 && last_round_work_ >= lane_min_work_);
	int nl = (lanes_wanted && S >= (size_t)lane_min_searches_ && stream2_) ? std::min(kMaxLanes, lanes_ >= 2 ? lanes_ : auto_lanes_) : 1;
	struct Run {
		QLane* L = nullptr;
		std::vector<int> idx;                 // lane slot -> index into `searches`
		QParams qp{};
		int parity = 0, chunk = 3, rounds_done = 0, last = 0;
		long long round_cap = 0;
		bool sort_round = false, tiles = false, twins = false, done = false;
	} run[kMaxLanes];
	if (nl > 1) {
		for (size_t i = 0; i < S; i++) run[searches[i]->rot_slot % nl].idx.push_back((int)i);
		for (int li = 0; li < nl; li++) if (run[li].idx.empty()) nl = 1;           // (a batch whose slots all fall into one class: not worth a lane)
		if (nl == 1) for (Run& r : run) r.idx.clear();
	}
	if (nl == 1) { run[0].idx.resize(S); for (size_t i = 0; i < S; i++) run[0].idx[i] = (int)i; }
	if (nl > 1) {            // the other lanes start behind the rotation upload (and everything else queued on the engine's stream)
		HIPCHK(hipEventRecord(ev_fork_, stream_));
		for (int li = 1; li < nl; li++) {
			if (!lane_stream_[li]) HIPCHK(hipStreamCreateWithFlags(&lane_stream_[li], hipStreamNonBlocking));     // lanes 2.. : on first use (a stream costs ~2 ms to create)
			HIPCHK(hipStreamWaitEvent(lane_stream_[li], ev_fork_, 0));
		}
	}
	for (int li = 0; li < nl; li++) {
		Run& r = run[li];
		const size_t Sl = r.idx.size();
		ensure_lane(li, Sl);
		QLane& L = ql_[li];
		r.L = &L;
		for (size_t i = 0; i < Sl; i++) {
			const InnerSearch& src = *searches[(size_t)r.idx[i]];
			QSearch& q = L.h_search[i];
			std::memset(&q, 0, sizeof(q));
			q.best = src.best; q.coeff = src.coeff; q.rot = src.rot_slot;
			q.twin = -1;
		}
		// the two searches of a rotation child (same rotation slot, one upper-bound pass with coeff 0, one lower-bound pass): each other's twin
		if (p_.twin_fusion && L.d_psearch[0]) {
			std::vector<int> first(nrot, -1);
			for (size_t i = 0; i < Sl; i++) {
				const int rs = L.h_search[i].rot;
				if (first[(size_t)rs] < 0) { first[(size_t)rs] = (int)i; continue; }
				const int j = first[(size_t)rs];
				if (L.h_search[j].twin < 0 && (L.h_search[j].coeff == 0.f) != (L.h_search[i].coeff == 0.f)) { L.h_search[j].twin = (int)i; L.h_search[i].twin = j; }
			}
		}
		HIPCHK(hipMemcpyAsync(L.d_search, L.h_search, sizeof(QSearch) * Sl, hipMemcpyHostToDevice, L.stream));
		r.qp = queue_params();
		r.qp.list_cap = L.list_cap; r.qp.seg_cap = L.seg_cap;
		r.qp.soft_overflow = soft_overflow_ ? 1 : 0;
		r.qp.K = K;
		r.qp.kmax = std::min(kQueueMaxPop, L.list_cap / (int)std::max<size_t>(Sl, 1));   // >= kQueueRoundPop: Sl <= the slots the lists were sized for
		HIPCHK(launch_bnb_init(L.d_search, L.d_nodes, (int)Sl, r.qp, L.d_ctl, L.stream));
		// the tile list: always on (lds_tiles 1), or -- the default -- only for the rounds that follow a read-back in which searches
		// qualified (QCtl::tile_hint moved): shallow batches, i.e. every default registration, never pay for the extra launch
		r.tiles = L.tile.ub != nullptr;
		r.twins = p_.twin_fusion && L.d_psearch[0] != nullptr;
		// footprint-ordered items: the sort is queued only for rounds that can reach sort.min_groups expansions -- the first rounds of a
		// batch by what a search can list in them (1, 8, 64 .. nodes), later ones by what the last read-back saw
		r.round_cap = (long long)Sl;
		r.sort_round = L.sort.order != nullptr;
		r.qp.tile_on = r.tiles && (p_.lds_tiles == 1 || (p_.lds_tiles == 2 && tile_sticky_)) ? 1 : 0;
		L.tile_hint_seen = 0;
	}
	// which build of the queue kernel: the batches of a deep run (following one that was fed from tiles: long bound evaluations, queue kernels of several
	// lanes side by side) take the 64-VGPR one (two searches per CU), the others the 128-VGPR one (no spills).  Measured (all-64 -> by batch): skull 6.3 ->
	// 6.17 ms, synthetic 40 k 11.3 -> 11.05 ms, bunny mse 1e-4 260 -> 255 ms, synthetic 40 k mse 3e-5 617 -> 609 ms; all-128: bunny mse 3e-5 5.02 -> 5.18 s
	const bool deep_batch = tile_sticky_;
	// One chunk of rounds of a lane: queue kernel + (sort) + bound evaluation(s) per round; the control block is read back behind the LAST round's queue kernel.
	auto submit = [&](Run& r) {
		const double t0 = now_ms();
		QLane& L = *r.L;
		const size_t Sl = r.idx.size();
		QParams& qp = r.qp;
		const int max_groups = (int)std::min<size_t>(Sl * (size_t)std::min(qp.kmax, 4 * qp.K), (size_t)L.list_cap);   // most the round can list (the kernel widens a stale search's step up to x4)
		for (int k = 0; k < r.chunk; k++) {
			const int parity = r.parity;
			HIPCHK(launch_bnb_queue(L.d_search, L.d_nodes, (int)Sl, qp, L.d_parents[parity ^ 1], L.d_parents[parity], L.d_ub, L.d_lb, L.d_scratch, L.d_ctl, parity, L.stream, r.tiles ? &L.tile : nullptr,
			                        r.twins ? L.d_psearch[parity] : nullptr, deep_batch));
			if (k == r.chunk - 1) {
				// Everything the host looks at after a chunk -- what the last round listed, how many searches are running, overflow, the tile
				// hint -- is written by THIS kernel; the bound evaluations behind it only fill in the bounds the next queue kernel digests.  So the
				// control block is read back here, ahead of the last round's bound evaluation, and the host decides and queues the next chunk
				// while those bounds are being evaluated: the same information at the same point of the search as a read-back after the chunk
				// (identical decisions, identical counts) for a 4 us copy in the stream instead of ~45 us of idle GPU per chunk.  Measured: bunny
				// 33.1-33.6 -> 32.8 ms, skull 6.4 -> 6.3, the rest within the spread.  (From a SIDE stream the copy lost: its blit kernel cannot
				// start while the persistent bound kernels hold every CU -- bunny 33.9 ms, synthetic 40 k mse 3e-5 610 -> 642 ms.)
				HIPCHK(hipMemcpyAsync(L.h_ctl, L.d_ctl, sizeof(QCtl), hipMemcpyDeviceToHost, L.stream));
				HIPCHK(hipEventRecord(L.ev_ctl[0], L.stream));
			}
			// The round's two lists are independent (own records, own bounds, own partial sums), so the tile list's evaluation CAN be forked onto a
			// second stream right behind the queue kernel and run beside the direct list's.  Built and measured (EXPERIMENTS R4.8): slower -- bunny
			// mse 3e-5 6.73 -> 7.40 s, bunny/10 1.09 -> 1.14 s, identical results -- the VALU-bound tile kernel (32 KB of LDS per workgroup) and the
			// gather kernel (122 VGPRs) take each other's occupancy; opt-in for A/B only (single-lane batches)
			const bool fork_tiles = qp.tile_on && tile_concurrent_ && nl == 1;
			if (fork_tiles) {
				HIPCHK(hipEventRecord(ev_fork_, L.stream));
				HIPCHK(hipStreamWaitEvent(stream2_, ev_fork_, 0));
				HIPCHK(launch_bounds_tile_queue(d_src_, (int)N_, dt_, d_rots_, L.tile, L.d_ctl, parity, stream2_));
				HIPCHK(hipEventRecord(ev_join_, stream2_));
				tile_rounds_++;
			}
			const bool sorted = r.sort_round && std::min<long long>(r.round_cap, max_groups) >= L.sort.min_groups;
			if (sorted) HIPCHK(launch_queue_sort(L.d_parents[parity], d_rots_, &L.d_ctl->n_groups[parity], max_groups, L.sort, bounds_dt(), L.stream));
			HIPCHK(launch_bounds_queue(d_src_, (int)N_, bounds_dt(), d_rots_, L.d_parents[parity], &L.d_ctl->n_groups[parity], &L.d_ctl->work[parity][0], &L.d_ctl->chunks, max_groups,
			                           inliers_, L.d_scratch, L.d_ub, L.d_lb, L.stream, r.twins ? L.d_search : nullptr, r.twins ? L.d_psearch[parity] : nullptr, sorted ? &L.sort : nullptr));
			if (r.round_cap < (1ll << 40)) r.round_cap *= 8;
			r.rounds_done++;
			if (fork_tiles) HIPCHK(hipStreamWaitEvent(L.stream, ev_join_, 0));
			else if (qp.tile_on) { HIPCHK(launch_bounds_tile_queue(d_src_, (int)N_, dt_, d_rots_, L.tile, L.d_ctl, parity, L.stream)); tile_rounds_++; }
			r.last = parity;
			r.parity ^= 1;
			cnt_.bounds_launches++;
			queue_rounds_++;
		}
		t_submit_ += now_ms() - t0;
	};
	// fold a chunk's read-back into the parameters of the chunks still to be queued
	auto adapt = [&](Run& r, const QCtl& c) {
		QLane& L = *r.L;
		QParams& qp = r.qp;
		// (a batch that follows one that used the tile list keeps it on for all of its rounds: an idle tile launch costs a few microseconds, a
		// qualifying search sent through the gather list costs 1.6x per cube bound -- prove-the-optimum bunny 5.18 -> 5.0 s)
		if (r.tiles && p_.lds_tiles == 2) { qp.tile_on = (c.tile_hint != L.tile_hint_seen || tile_sticky_) ? 1 : 0; L.tile_hint_seen = c.tile_hint; }
		// (later rounds of a batch are narrow -- their expansions lie close together whatever the order -- and in long registrations the three
		// extra launches per round are not free on the host side: mse 1e-4 bunny 295 vs 302 ms with the sort queued in every wide round)
		r.sort_round = L.sort.order != nullptr && c.n_groups[r.last] >= L.sort.min_groups && r.rounds_done < 7;
		// the stragglers: when the last round listed few expansions, few searches are still running and the chip is
		// mostly idle -- let each of them expand more nodes per round (fewer latency-bound rounds; the extra speculation
		// costs nothing the chip was using)
		r.chunk = 4;
		// exact: the searches that listed expansions in the last round (a search only ever finishes, so it bounds the rounds to come)
		const int active = std::max(1, c.n_active[r.last]);
		qp.kmax = std::min(kQueueMaxPop, L.list_cap / active);
		if (p_.adaptive_k && K >= 32) {
			// <= 16 running: up to 512 expansions each (swept 1 / 2 / 4 / 8 / 16 searches: the same within the run-to-run spread)
			qp.K = active <= 16 ? kQueueMaxPop : (active <= 64 ? std::min(kQueueRoundPop, 2 * K) : K);
		}
	};
	// lock-step with the host, lane by lane: queue a chunk, wait for it, look at it (a lone lane leaves the GPU idle ~45 us per chunk while
	// the host decides; with two lanes the other lane's chunk is running meanwhile).  Staying a chunk AHEAD of the read-backs instead was
	// built and measured in round 4 (EXPERIMENTS R4.9): the late round width cost more rounds than the bubbles it removed -- removed again.
	bool overflow = false;
	for (int li = 0; li < nl; li++) submit(run[li]);
	for (int live = nl; live > 0;) {
		for (int li = 0; li < nl; li++) {
			Run& r = run[li];
			if (r.done) continue;
			const double t1 = now_ms();
			HIPCHK(hipEventSynchronize(r.L->ev_ctl[0]));
			t_wait_ += now_ms() - t1;
			const QCtl& c = r.L->h_ctl[0];
			if (c.overflow) overflow = true;
			if (c.overflow || overflow || (c.n_groups[r.last] == 0 && c.n_tile_groups[r.last] == 0) || cancel_.load()) { r.done = true; live--; continue; }
			adapt(r, c);
			submit(r);
		}
	}
	if (overflow) {
		for (int li = 0; li < nl; li++) HIPCHK(hipStreamSynchronize(run[li].L->stream));
		queue_fallbacks_++; cnt_.queue_fallbacks++;
		return false;
	}
	// the running totals of the whole batch and the searches' results in one round trip per lane
	const double t2 = now_ms();
	for (int li = 0; li < nl; li++) {
		QLane& L = *run[li].L;
		HIPCHK(hipMemcpyAsync(L.h_ctl, L.d_ctl, sizeof(QCtl), hipMemcpyDeviceToHost, L.stream));
		HIPCHK(hipMemcpyAsync(L.h_search, L.d_search, sizeof(QSearch) * run[li].idx.size(), hipMemcpyDeviceToHost, L.stream));
	}
	for (int li = 0; li < nl; li++) HIPCHK(hipStreamSynchronize(run[li].L->stream));
	for (int li = 0; li < nl; li++) if (run[li].L->h_ctl->overflow) overflow = true;
	if (overflow) { queue_fallbacks_++; cnt_.queue_fallbacks++; return false; }
	if (nl > 1) cnt_.lane_batches++;
	{
		long long cubes = 0;
		for (int li = 0; li < nl; li++) for (size_t i = 0; i < run[li].idx.size(); i++) cubes += run[li].L->h_search[i].cubes;
		int rounds = 1;
		for (int li = 0; li < nl; li++) rounds = std::max(rounds, run[li].rounds_done);
		last_round_work_ = (double)cubes / kGroup / rounds * (double)N_;
	}
	{
		long long tile_total = 0, all = 0;
		for (int li = 0; li < nl; li++) {
			tile_total += run[li].L->h_ctl->tile_total;
			for (size_t i = 0; i < run[li].idx.size(); i++) all += run[li].L->h_search[i].cubes;
		}
		tile_sticky_ = tile_total > 0 && (double)tile_total * kGroup >= tile_sticky_share_ * (double)all;
	}
	for (int li = 0; li < nl; li++) {
		const QLane& L = *run[li].L;
		cnt_.tile_expansions += L.h_ctl->tile_total;
		if (p_.verbose) for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) sel_hist_[a][b] += L.h_ctl->sel_hist[a][b];
		for (size_t i = 0; i < run[li].idx.size(); i++) {
			const QSearch& q = L.h_search[i];
			InnerSearch& s = *searches[(size_t)run[li].idx[i]];
			if (q.done == 2) { redo_.push_back(&s); continue; }          // its queue outgrew the slab: untouched here, re-run by run_inner through the host queues
			s.best = q.best; s.improved = q.improved != 0; s.done = true;
			s.best_node = Node{q.bx, q.by, q.bz, q.bw, 0.f, 0.f, 0};
			s.pops = q.pops; s.cubes = q.cubes; s.min_ub = q.min_ub;
		}
	}
	t_collect_ += now_ms() - t2;
	return true;
}

void Engine::run_inner(std::vector<InnerSearch*>& searches, const std::vector<Rot9>& rots)
{
	DeviceGuard guard(dev_);
	for (auto* s : searches)
		if (s->rot_slot < 0 || (size_t)s->rot_slot >= rots.size()) throw std::logic_error("goicp: rotation slot out of range");
	if (p_.device_queues && p_.trans_batch > 1) {
		const double t_begin = now_ms();
		redo_.clear();
		const bool ok = run_inner_device(searches, rots);
		bnb_ms_ += now_ms() - t_begin;
		if (ok && redo_.empty()) return;
		if (ok) {
			// searches whose queue outgrew its slab (they have not been touched): through the host queues, alone
			queue_fallbacks_++; cnt_.queue_fallbacks++;
			std::vector<InnerSearch*> redo;
			redo.swap(redo_);
			run_inner_host(redo, rots, true);
			return;
		}
		// a round's lists overflowed: the whole batch is re-run through the host queues (the searches have not been touched)
		run_inner_host(searches, rots, true);
		return;
	}
	run_inner_host(searches, rots);
}

// Lock-step rounds of all the given inner searches: pop up to trans_batch nodes per search, evaluate the
// 8 children of every popped node in ONE launch, digest the bounds, repeat until every search stops.
// fallback (a device queue outgrew its slab -- a search with tens of thousands of queued nodes): the round of a search whose
// incumbent did not improve in its last round grows x4, then x16 (the stale-incumbent rule of the device queues: a search that
// is proving expands every node with best - lb >= SSEThresh whatever the order, and the host's heap has no 128-node limit), so
// the re-run is a few hundred large launches instead of thousands of small ones
void Engine::run_inner_host(std::vector<InnerSearch*>& searches, const std::vector<Rot9>& rots, bool fallback)
{
	DeviceGuard guard(dev_);
	const double t_begin = now_ms();
	struct Acc { double& a; double t0; ~Acc() { a += now_ms() - t0; } } acc{bnb_ms_, t_begin};
	const int K0 = std::max(1, p_.trans_batch);
	const bool widen = fallback && p_.adaptive_k && p_.stale_widen && K0 > 1;
	const size_t nrot = rots.size();
	ensure_batch(1, nrot);
	std::memcpy(h_rots_, rots.data(), sizeof(Rot9) * nrot);
	HIPCHK(hipMemcpyAsync(d_rots_, h_rots_, sizeof(Rot9) * nrot, hipMemcpyHostToDevice, stream_));

	// One group per round.  (Measured on MI355X: splitting the searches into two alternating groups so
	// that the host digests one group's results while the GPU evaluates the other's did not pay --
	// 0.087 s vs 0.083 s on the full bunny: the rounds are bound by the small launches themselves, not by
	// host work.  The two-stage plumbing is kept for a device-resident queue.)
	std::vector<InnerSearch*> grp[2];
	grp[0] = searches;

	auto submit = [&](int k) -> bool {
		size_t B = 0;
		for (auto* s : grp[k]) {
			s->parents.clear();
			if (s->done) continue;
			const int K = !widen ? K0 : (s->stale >= 3 ? 16 * K0 : (s->stale >= 1 ? 4 * K0 : K0));
			while ((int)s->parents.size() < K && !s->pq.empty()) {
				const Node n = s->pq.top();
				if (s->best - n.lb < sse_thresh_) {           // jly_goicp.cpp:257
					if (s->parents.empty()) { s->pq.pop(); s->pops++; s->done = true; }
					break;
				}
				s->pq.pop();
				s->pops++;
				s->parents.push_back(n);
			}
			if (s->parents.empty()) { s->done = true; continue; }
			B += 8 * s->parents.size();
		}
		if (B == 0) return false;
		ensure_stage(k, B);
		Stage& st = stage_[k];
		size_t o = 0;
		for (auto* s : grp[k])
			for (const Node& par : s->parents) {
				// the kernels expand the 8 children themselves (load_group in device.hip; jly_goicp.cpp:262-273)
				ParentRec& r = st.h_parents[o++];
				if (p_.verbose) { int lv = 0; for (float w = par.w; w < trans_root_.w && lv < 31; w *= 2) lv++; level_hist_[lv]++; }
				r.x = par.x; r.y = par.y; r.z = par.z; r.w = par.w; r.coeff = s->coeff; r.rot = s->rot_slot;
			}
		st.B = B;
		HIPCHK(hipMemcpyAsync(st.d_parents, st.h_parents, sizeof(ParentRec) * (B / 8), hipMemcpyHostToDevice, stream_));
		eval_bounds_dev(d_rots_, nullptr, (int)B, st.d_ub, st.d_ub + B, stream_, st.d_parents);
		HIPCHK(hipMemcpyAsync(st.h_ub, st.d_ub, sizeof(float) * 2 * B, hipMemcpyDeviceToHost, stream_));
		HIPCHK(hipEventRecord(st.ev, stream_));
		return true;
	};
	auto collect = [&](int k) {
		Stage& st = stage_[k];
		HIPCHK(hipEventSynchronize(st.ev));
		size_t o = 0;
		for (auto* s : grp[k]) {
			const float best_before = s->best;
			for (const Node& par : s->parents) {
				Node c{};
				c.w = par.w / 2;
				for (int j = 0; j < 8; j++, o++) {
					c.x = par.x + (j & 1) * c.w; c.y = par.y + (j >> 1 & 1) * c.w; c.z = par.z + (j >> 2 & 1) * c.w;
					const float ub = st.h_ub[o], lb = st.h_ub[st.B + o];
					s->cubes++;
					if (trans_boxed_ && !in_box(c, trans_lo_, trans_hi_)) continue;   // outside the configured translation range
					if (ub < s->min_ub) s->min_ub = ub;
					if (ub < s->best) { s->best = ub; s->best_node = c; s->improved = true; }   // :319-324
					if (lb >= s->best) continue;                                                  // :327
					if (p_.trans_search_depth > 0 && !(std::ldexp(c.w, p_.trans_search_depth) > trans_root_.w)) continue;   // depth limit reached: evaluated, not expanded
					c.ub = ub; c.lb = lb;
					s->pq.push(c);
				}
			}
			if (!s->parents.empty()) s->stale = s->best < best_before ? 0 : s->stale + 1;
		}
	};
	auto timed = [&](double& acc, auto&& fn) { const double t0 = now_ms(); auto r = fn(); acc += now_ms() - t0; return r; };
	bool fly[2] = {timed(t_submit_, [&] { return submit(0); }), !grp[1].empty() && submit(1)};
	while (fly[0] || fly[1])
		for (int k = 0; k < 2; k++)
			if (fly[k]) {
				timed(t_wait_, [&] { HIPCHK(hipEventSynchronize(stage_[k].ev)); return 0; });
				timed(t_collect_, [&] { collect(k); return 0; });
				fly[k] = !cancel_.load() && timed(t_submit_, [&] { return submit(k); });
			}
	HIPCHK(hipStreamSynchronize(stream_));
}

float Engine::inner_bnb(const float R[9], int level, float incumbent, float best_node[4], Counters* c)
{
	DeviceGuard guard(dev_);
	std::vector<Rot9> rots(1);
	std::memcpy(rots[0].r, R, sizeof(float) * 9);
	InnerSearch s;
	s.rot_slot = 0;
	s.coeff = rot_coeff(level);
	s.best = incumbent;
	s.pq.push(trans_root_);   // jly_goicp.cpp:50-53
	std::vector<InnerSearch*> v{&s};
	run_inner(v, rots);
	if (best_node && s.improved) { best_node[0] = s.best_node.x; best_node[1] = s.best_node.y; best_node[2] = s.best_node.z; best_node[3] = s.best_node.w; }
	cnt_.trans_pops += s.pops; cnt_.cubes += s.cubes; cnt_.inner_calls++;
	if (c) { c->trans_pops += s.pops; c->cubes += s.cubes; c->inner_calls++; c->queue_fallbacks = cnt_.queue_fallbacks; }
	return s.best;
}

// ------------------------------------------------------------------------------------------------
// outer (rotation) BnB
// ------------------------------------------------------------------------------------------------
void Engine::publish(bool finished)
{
	Result copy;
	{
		std::lock_guard<std::mutex> lk(mtx_);
		std::memcpy(snap_.optR, optR_, sizeof(optR_)); std::memcpy(snap_.optT, optT_, sizeof(optT_));
		std::memcpy(snap_.curR, curR_, sizeof(curR_)); std::memcpy(snap_.curT, curT_, sizeof(curT_));
		snap_.best_sse = opt_err_;
		snap_.finished = finished ? 1 : 0;
		snap_.counters = cnt_;
		snap_.dt_build_ms = dt_build_ms_;
		snap_.register_ms = register_ms_;
		copy = snap_;
	}
	if (progress_cb_) progress_cb_(copy);      // outside the lock: the callback may poll
}

Result Engine::poll()
{
	std::lock_guard<std::mutex> lk(mtx_);
	return snap_;
}

void Engine::adopt(float err, const float R[9], const float t[3])
{
	opt_err_ = err;
	std::memcpy(optR_, R, sizeof(optR_));
	std::memcpy(optT_, t, sizeof(optT_));
}

float Engine::icp_from(float R[9], float t[3])
{
	TraceRange tr("goicp:icp_run+rescore");
	const double t0 = now_ms();
	struct Acc { double& a; double t0; ~Acc() { a += now_ms() - t0; } } acc{icp_ms_, t0};
	// GoICP::ICP (jly_goicp.cpp:93-132): ICP3D::Run, then re-score with the DT
	int it = 0;
	icp_run(R, t, p_.icp_max_iter, icp_err_diff_, &it);
	const float e = eval_sse(R, t);
	if (p_.verbose > 1) std::fprintf(stderr, "[goicp] ICP run: %d iterations, %.2f ms, error %.6g (rot pops so far %lld, cube bounds %lld)\n", it, now_ms() - t0, e, cnt_.rot_pops, cnt_.cubes);
	return e;
}

void Engine::offer_global_best(float sse, const float R[9], const float t[3])
{
	DeviceGuard guard(dev_);
	if (sse < opt_err_) {
		adopt(sse, R, t);
		// drop queued nodes that can no longer win (jly_goicp.cpp:533-543)
		std::priority_queue<Node> nq;
		while (!queue_.empty()) {
			Node n = queue_.top(); queue_.pop();
			if (n.lb < opt_err_) nq.push(n); else break;
		}
		queue_.swap(nq);
		if (opt_err_ < sse_thresh_) early_exit_ = true;
		publish(false);
	}
}

void Engine::register_begin()
{
	DeviceGuard guard(dev_);
	cancel_.store(false);
	early_exit_ = converged_ = false;
	rot_ramp_ = 8;
	late_icp_.clear(); batches_done_ = 0;
	last_round_work_ = 0; tile_sticky_ = false;      // every registration starts single-lane (determinism: the choice depends on this registration only)
	{ const char* e = std::getenv("GOICP_ICP_DELAY_BATCHES"); icp_delay_ = e ? std::max(0, std::atoi(e)) : 0; }
	icp_ms_ = 0; t_submit_ = t_wait_ = t_collect_ = 0;
	std::memset(level_hist_, 0, sizeof(level_hist_));
	cnt_ = Counters{};
	while (!queue_.empty()) queue_.pop();
	if (flow_mode()) { ensure_queues(kFlowSearches); flow_reset(); }
	const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	const float Z[3] = {0, 0, 0};
	// initial error (jly_goicp.cpp:357-372) and initial ICP (:375-391)
	opt_err_ = eval_sse(I, Z);
	std::memcpy(optR_, I, sizeof(I)); std::memcpy(optT_, Z, sizeof(Z));
	float R[9], t[3];
	std::memcpy(R, I, sizeof(I)); std::memcpy(t, Z, sizeof(Z));
	float e = icp_from(R, t);
	if (e < opt_err_) adopt(e, R, t);
	std::memcpy(curR_, optR_, sizeof(optR_)); std::memcpy(curT_, optT_, sizeof(optT_));
	if (p_.verbose) std::fprintf(stderr, "[goicp] init error %.6g (after ICP)\n", opt_err_);
	bnb_ms_ = 0;

	const Node root = rot_root_;   // jly_goicp.cpp:44-48 unless [params.rotation] narrows it
	if (world_ <= 1) {
		queue_.push(root);
	} else {
		// shard: expand the root two levels (64 cubes) and deal them round-robin to the ranks; the
		// parents' bounds are 0, so nothing is lost
		int k = 0;
		for (int a = 0; a < 8; a++) {
			Node c1 = root; c1.w = root.w / 2; c1.l = 1;
			c1.x = root.x + (a & 1) * c1.w; c1.y = root.y + (a >> 1 & 1) * c1.w; c1.z = root.z + (a >> 2 & 1) * c1.w;
			for (int b = 0; b < 8; b++, k++) {
				Node c2 = c1; c2.w = c1.w / 2; c2.l = 2;
				c2.x = c1.x + (b & 1) * c2.w; c2.y = c1.y + (b >> 1 & 1) * c2.w; c2.z = c1.z + (b >> 2 & 1) * c2.w;
				if (rot_boxed_ && !in_box(c2, rot_lo_, rot_hi_)) continue;
				if (k % world_ == rank_) queue_.push(c2);
			}
		}
	}
	publish(false);
}

// the rotation children of the given parents that survive the pi-ball and range culls (jly_goicp.cpp:427-467)
void Engine::make_kids(const std::vector<Node>& parents, std::vector<Kid>& kids)
{
	for (const Node& parent : parents) {
		Node c{};
		c.w = parent.w / 2;          // jly_goicp.cpp:427-428
		c.l = parent.l + 1;
		for (int j = 0; j < 8; j++) {
			c.x = parent.x + (j & 1) * c.w; c.y = parent.y + (j >> 1 & 1) * c.w; c.z = parent.z + (j >> 2 & 1) * c.w;
			float v1 = c.x + c.w / 2, v2 = c.y + c.w / 2, v3 = c.z + c.w / 2;
			// pi-ball cull (:443): float sqrt, double subtraction and comparison
			if ((double)std::sqrt(v1 * v1 + v2 * v2 + v3 * v3) - kSQRT3 * (double)c.w / 2 > kPI) continue;
			if (rot_boxed_ && !in_box(c, rot_lo_, rot_hi_)) continue;     // outside the configured rotation range
			Kid k;
			k.node = c;
			k.parent_lb = parent.lb;
			rodrigues(v1, v2, v3, k.R);
			kids.push_back(k);
		}
	}
}

// jly_goicp.cpp:495-544: the upper-bound search of a rotation child is in; returns true on the early exit (:527)
bool Engine::handle_ub(Kid& k, const SearchOut& s)
{
	cnt_.trans_pops += s.pops; cnt_.cubes += s.cubes; cnt_.inner_calls++;
	k.node.ub = s.best;
	// the widened search orders rotation cubes of equal lower bound (all the shallow ones: lb = 0) by the smallest upper bound
	// their own upper-bound search saw -- the basin most likely to refine the optimum is expanded first.  The reference leaves
	// that order to its heap (any order of a best-first BnB is valid); the reference-order mode keeps the key at 0.
	k.node.tie = (p_.wide_children && p_.ub_tiebreak && std::isfinite(s.min_ub)) ? s.min_ub : 0.f;
	std::memcpy(curR_, k.R, sizeof(curR_));
	if (s.improved) { curT_[0] = s.best_node.x + s.best_node.w / 2; curT_[1] = s.best_node.y + s.best_node.w / 2; curT_[2] = s.best_node.z + s.best_node.w / 2; }
	if (!(s.best < opt_err_) || !s.improved) return false;
	float t[3] = {s.best_node.x + s.best_node.w / 2, s.best_node.y + s.best_node.w / 2, s.best_node.z + s.best_node.w / 2};
	adopt(s.best, k.R, t);
	float R[9], ti[3];
	std::memcpy(R, k.R, sizeof(R)); std::memcpy(ti, t, sizeof(ti));
	float e = icp_from(R, ti);
	if (icp_delay_ > 0) {
		LateIcp li; li.e = e; std::memcpy(li.R, R, sizeof(R)); std::memcpy(li.t, ti, sizeof(ti)); li.due = batches_done_ + icp_delay_;
		late_icp_.push_back(li);
		publish(false);
		return false;
	}
	if (e < opt_err_) adopt(e, R, ti);
	if (p_.verbose) std::fprintf(stderr, "[goicp] rank %d  error* %.6g (ub %.6g, level %d)\n", rank_, opt_err_, s.best, k.node.l);
	publish(false);
	if (opt_err_ < sse_thresh_) { early_exit_ = true; return true; }   // :527
	std::priority_queue<Node> nq;                                       // :533-543
	while (!queue_.empty()) {
		Node n = queue_.top(); queue_.pop();
		if (n.lb < opt_err_) nq.push(n); else break;
	}
	queue_.swap(nq);
	return false;
}

void Engine::fold_late_icp(bool all)
{
	for (size_t i = 0; i < late_icp_.size();) {
		if (!all && late_icp_[i].due > batches_done_) { i++; continue; }
		const LateIcp li = late_icp_[i];
		late_icp_.erase(late_icp_.begin() + (long)i);
		if (li.e < opt_err_) {
			adopt(li.e, li.R, li.t);
			if (opt_err_ < sse_thresh_) early_exit_ = true;
			std::priority_queue<Node> nq;
			while (!queue_.empty()) { Node n = queue_.top(); queue_.pop(); if (n.lb < opt_err_) nq.push(n); else break; }
			queue_.swap(nq);
			publish(false);
		}
	}
}

// :551-562: the lower-bound search is in
void Engine::handle_lb(Kid& k, const SearchOut& s)
{
	cnt_.trans_pops += s.pops; cnt_.cubes += s.cubes; cnt_.inner_calls++;
	if (s.best >= opt_err_) return;
	if (p_.rot_search_depth > 0 && k.node.l >= p_.rot_search_depth) return;   // depth limit: evaluated, not expanded
	k.node.lb = s.best;
	queue_.push(k.node);
}

void Engine::process_parents(const std::vector<Node>& parents)
{
	TraceRange tr("goicp:rotation_batch");
	std::vector<Kid> kids;
	make_kids(parents, kids);
	if (kids.empty()) return;
	std::vector<Rot9> rots(kids.size());
	for (size_t i = 0; i < kids.size(); i++) std::memcpy(rots[i].r, kids[i].R, sizeof(float) * 9);
	const Node troot = trans_root_;
	auto fresh = [&](size_t slot, float coeff) {
		InnerSearch s;
		s.rot_slot = (int)slot; s.coeff = coeff; s.best = opt_err_;
		s.pq.push(troot);
		return s;
	};
	auto out = [](const InnerSearch& s) { return SearchOut{s.best, s.improved, s.best_node, s.pops, s.cubes, s.min_ub}; };

	if (p_.wide_children) {
		// every child's upper-bound AND lower-bound search in lock-step: one launch per round covers
		// up to 16 searches per rotation parent.  The searches start from the incumbent of the batch
		// start (the reference lets earlier children tighten it: fewer nodes, same bounds).
		std::vector<InnerSearch> ubs, lbs;
		ubs.reserve(kids.size()); lbs.reserve(kids.size());
		for (size_t i = 0; i < kids.size(); i++) { ubs.push_back(fresh(i, 0.f)); lbs.push_back(fresh(i, rot_coeff(kids[i].node.l))); }
		std::vector<InnerSearch*> ptr;
		// a child's two searches side by side: they share the rotation, expand the same root and a third of the same
		// depth-1 nodes in the same rounds, and the bound kernel walks the expansions in search order -- the second of the
		// pair finds the first one's DT lines in L2 (shallow rounds are bound by compulsory line fetches)
		for (size_t i = 0; i < kids.size(); i++) { ptr.push_back(&ubs[i]); ptr.push_back(&lbs[i]); }
		run_inner(ptr, rots);
		for (size_t i = 0; i < kids.size(); i++) {
			if (handle_ub(kids[i], out(ubs[i]))) return;
			handle_lb(kids[i], out(lbs[i]));
		}
	} else {
		for (size_t i = 0; i < kids.size(); i++) {
			InnerSearch u = fresh(i, 0.f);
			std::vector<InnerSearch*> p1{&u};
			run_inner(p1, rots);
			if (handle_ub(kids[i], out(u))) return;
			InnerSearch l = fresh(i, rot_coeff(kids[i].node.l));
			std::vector<InnerSearch*> p2{&l};
			run_inner(p2, rots);
			handle_lb(kids[i], out(l));
			if (cancel_.load()) return;
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Continuous flow (wide mode + device-resident queues): rotation children are admitted in batches but HARVESTED one by
// one -- a child is handled as soon as both its searches have stopped, its slots are recycled, and the next batch of
// parents is admitted when the number of running searches falls below a low-water mark, instead of waiting for the
// slowest search of the batch.  (Lock-step batches ended in ~5 rounds with a handful of active searches each, at the
// latency floor of three launches; those rounds now carry the next batch's work.)  Any expansion order of a best-first BnB
// is valid; the searches of a child start from the incumbent at its admission.
// ------------------------------------------------------------------------------------------------
void Engine::flow_reset()
{
	flights_.clear();
	free_search_.clear(); free_rot_.clear();
	for (int i = kFlowSearches - 1; i >= 0; i--) free_search_.push_back(i);
	for (int i = kFlowSearches / 2 - 1; i >= 0; i--) free_rot_.push_back(i);
	q_hi_ = 0; q_parity_ = 0; flow_active_ = 0;
	if (ql_[0].d_ctl) HIPCHK(hipMemsetAsync(ql_[0].d_ctl, 0, sizeof(QCtl), stream_));
	// a slot abandoned in flight (early exit, cancel) must not come back as a live search: an all-zero record is an empty
	// queue that marks itself done at first touch
	if (ql_[0].d_search) HIPCHK(hipMemsetAsync(ql_[0].d_search, 0, sizeof(QSearch) * ql_[0].cap, stream_));
}

bool Engine::tiles_usable() const
{
	// the tile kernel stages 4x4x4 fp32 bricks and sums every point (no trimming)
	return p_.lds_tiles != 0 && dt_.layout == 1 && !p_.bounds_fp16 && inliers_ >= (int)N_;
}

QParams Engine::queue_params() const
{
	QParams qp{};
	qp.tile_on = 0; qp.tile_min = std::max(8, p_.tile_min);
	qp.tile_spread = tiles_usable() ? (float)((double)p_.tile_spread_vox / dt_.scale) : 0.f;
	qp.tile_stats = p_.verbose ? 1 : 0; qp.tile_stats_scale = (float)dt_.scale;
	qp.stale_widen = p_.adaptive_k ? p_.stale_widen : 0;
	qp.stale_compact = p_.adaptive_k && p_.stale_widen ? p_.stale_compact : 0;

	qp.thr = sse_thresh_; qp.K = std::min(std::max(1, p_.trans_batch), kQueueRoundPop);
	qp.kmax = kQueueRoundPop; qp.list_cap = ql_[0].list_cap; qp.seg_cap = ql_[0].seg_cap;     // any number of searches fits; run_inner_device raises kmax for the last few
	qp.root_x = trans_root_.x; qp.root_y = trans_root_.y; qp.root_z = trans_root_.z; qp.root_w = trans_root_.w;
	qp.boxed = trans_boxed_ ? 1 : 0; qp.depth = p_.trans_search_depth;
	qp.cap = (p_.queue_cap > 0 && p_.queue_cap < kQueueCap) ? p_.queue_cap : kQueueCap;
	for (int k = 0; k < 3; k++) { qp.lo[k] = trans_boxed_ ? trans_lo_[k] : 0.f; qp.hi[k] = trans_boxed_ ? trans_hi_[k] : 0.f; }
	return qp;
}

// a queue outgrew its slab: every child still in flight is finished through the host queues (from its own incumbent)
void Engine::flow_fallback()
{
	queue_fallbacks_++; cnt_.queue_fallbacks++;
	std::vector<Flight> todo;
	for (const Flight& f : flights_) if (!f.handled) todo.push_back(f);
	flow_reset();
	for (Flight& f : todo) {
		std::vector<Rot9> rots(1);
		std::memcpy(rots[0].r, f.kid.R, sizeof(float) * 9);
		InnerSearch u, l;
		u.rot_slot = l.rot_slot = 0; u.coeff = 0.f; l.coeff = rot_coeff(f.kid.node.l); u.best = l.best = f.incumbent;
		u.pq.push(trans_root_); l.pq.push(trans_root_);
		std::vector<InnerSearch*> ptr{&u, &l};
		run_inner_host(ptr, rots, true);
		if (handle_ub(f.kid, SearchOut{u.best, u.improved, u.best_node, u.pops, u.cubes, u.min_ub})) return;
		handle_lb(f.kid, SearchOut{l.best, l.improved, l.best_node, l.pops, l.cubes, l.min_ub});
	}
}

int Engine::flow_step(int max_rot_pops)
{
	TraceRange tr("goicp:flow_step");
	const double t_begin = now_ms();
	struct Acc { double& a; double t0; ~Acc() { a += now_ms() - t0; } } acc{bnb_ms_, t_begin};
	const QParams qp = queue_params();
	ensure_batch(1, kFlowSearches / 2);
	if (!h_qinit_) {
		HIPCHK(hipHostMalloc(&h_qinit_, sizeof(QInit) * kFlowSearches));
		HIPCHK(hipMalloc(&d_qinit_, sizeof(QInit) * kFlowSearches));
	}
	int pops = 0;
	auto unhandled = [&] { size_t n = 0; for (const Flight& f : flights_) n += f.handled ? 0 : 1; return n; };
	while (!early_exit_ && !cancel_.load()) {
		// ---- admit the next batch of rotation parents when the running searches are few ----
		// at most kFlowSearches / 16 parents per admission: a batch needs 16 search slots per parent, and with a larger P the
		// admission test below could never pass even with every slot free (rot_batch > 128: the loop would spin forever)
		const int P = std::max(1, std::min(std::min(p_.rot_batch, rot_ramp_), kFlowSearches / 16));
		if (!converged_ && !queue_.empty() && pops < max_rot_pops && flow_active_ <= (flights_.empty() ? kFlowSearches : p_.flow) &&
		    free_search_.size() >= (size_t)16 * P && free_rot_.size() >= (size_t)8 * P) {
			std::vector<Node> parents;
			while ((int)parents.size() < P && !queue_.empty() && pops < max_rot_pops) {
				const Node parent = queue_.top();
				if ((opt_err_ - parent.lb) <= sse_thresh_) {      // jly_goicp.cpp:416 -- only final once nothing is in flight
					if (parents.empty() && unhandled() == 0) { queue_.pop(); cnt_.rot_pops++; pops++; converged_ = true; }
					break;
				}
				queue_.pop();
				cnt_.rot_pops++;
				pops++;
				parents.push_back(parent);
			}
			if (!parents.empty()) {
				rot_ramp_ = std::min(rot_ramp_ * 2, 1 << 20);
				std::vector<Kid> kids;
				make_kids(parents, kids);
				int n = 0;
				for (Kid& k : kids) {
					Flight f{};
					f.kid = k; f.incumbent = opt_err_; f.handled = false;
					f.rot_slot = free_rot_.back(); free_rot_.pop_back();
					f.s_ub = free_search_.back(); free_search_.pop_back();
					f.s_lb = free_search_.back(); free_search_.pop_back();
					std::memcpy(h_rots_[f.rot_slot].r, k.R, sizeof(float) * 9);
					h_qinit_[n++] = QInit{f.s_ub, opt_err_, 0.f, f.rot_slot, -1};
					h_qinit_[n++] = QInit{f.s_lb, opt_err_, rot_coeff(k.node.l), f.rot_slot, -1};
					q_hi_ = std::max(q_hi_, std::max(f.s_ub, f.s_lb) + 1);
					flights_.push_back(f);
				}
				if (n) {
					HIPCHK(hipMemcpyAsync(d_rots_, h_rots_, sizeof(Rot9) * (kFlowSearches / 2), hipMemcpyHostToDevice, stream_));
					HIPCHK(hipMemcpyAsync(d_qinit_, h_qinit_, sizeof(QInit) * n, hipMemcpyHostToDevice, stream_));
					HIPCHK(launch_bnb_init_list(ql_[0].d_search, ql_[0].d_nodes, d_qinit_, n, qp, stream_));
					flow_active_ += n;
				}
			}
		}
		if (unhandled() == 0) break;
		// ---- a chunk of rounds over every slot in use ----
		const double t0 = now_ms();
		QParams qr = qp;
		if (p_.adaptive_k && qp.K >= 32) qr.K = flow_active_ <= 16 ? kQueueRoundPop : (flow_active_ <= 64 ? std::min(kQueueRoundPop, 2 * qp.K) : qp.K);
		// most the round can list: the queue kernel widens a stale search's step up to x4 (stale_widen), as in run_inner_device --
		// sizing the evaluation's grid by q_hi_ * K left the expansions listed beyond it unevaluated in trimmed runs (one workgroup each)
		const int max_groups = (int)std::min<size_t>((size_t)q_hi_ * (size_t)std::min(qr.kmax, 4 * qr.K), (size_t)ql_[0].list_cap);
		for (int r = 0; r < 3; r++) {
			HIPCHK(launch_bnb_queue(ql_[0].d_search, ql_[0].d_nodes, q_hi_, qr, ql_[0].d_parents[q_parity_ ^ 1], ql_[0].d_parents[q_parity_], ql_[0].d_ub, ql_[0].d_lb, ql_[0].d_scratch, ql_[0].d_ctl, q_parity_, stream_));
			HIPCHK(launch_bounds_queue(d_src_, (int)N_, bounds_dt(), d_rots_, ql_[0].d_parents[q_parity_], &ql_[0].d_ctl->n_groups[q_parity_], &ql_[0].d_ctl->work[q_parity_][0], &ql_[0].d_ctl->chunks,
			                           max_groups, inliers_, ql_[0].d_scratch, ql_[0].d_ub, ql_[0].d_lb, stream_));
			q_parity_ ^= 1;
			cnt_.bounds_launches++;
			queue_rounds_++;
		}
		HIPCHK(hipMemcpyAsync(ql_[0].h_ctl, ql_[0].d_ctl, sizeof(QCtl), hipMemcpyDeviceToHost, stream_));
		HIPCHK(hipMemcpyAsync(ql_[0].h_search, ql_[0].d_search, sizeof(QSearch) * (size_t)q_hi_, hipMemcpyDeviceToHost, stream_));
		t_submit_ += now_ms() - t0;
		const double t1 = now_ms();
		HIPCHK(hipStreamSynchronize(stream_));
		t_wait_ += now_ms() - t1;
		if (ql_[0].h_ctl->overflow) { flow_fallback(); continue; }
		// ---- harvest: every child whose two searches have stopped, in admission order ----
		const double t2 = now_ms();
		int active = 0;
		bool stop = false;
		for (Flight& f : flights_) {
			if (f.handled) continue;
			const QSearch& u = ql_[0].h_search[f.s_ub];
			const QSearch& l = ql_[0].h_search[f.s_lb];
			// a search that has stopped has no pending children: done is only set in the selection phase, after the digest
			if (!u.done || !l.done) { active += (u.done ? 0 : 1) + (l.done ? 0 : 1); continue; }
			f.handled = true;
			free_search_.push_back(f.s_ub); free_search_.push_back(f.s_lb); free_rot_.push_back(f.rot_slot);
			if (stop) continue;                                                  // early exit taken: the rest is abandoned
			const SearchOut su{u.best, u.improved != 0, Node{u.bx, u.by, u.bz, u.bw, 0.f, 0.f, 0}, u.pops, u.cubes, u.min_ub};
			const SearchOut sl{l.best, l.improved != 0, Node{l.bx, l.by, l.bz, l.bw, 0.f, 0.f, 0}, l.pops, l.cubes, l.min_ub};
			if (handle_ub(f.kid, su)) { stop = true; continue; }
			handle_lb(f.kid, sl);
		}
		t_collect_ += now_ms() - t2;
		flow_active_ = active;
		flights_.erase(std::remove_if(flights_.begin(), flights_.end(), [](const Flight& f) { return f.handled; }), flights_.end());
		if (flights_.empty()) q_hi_ = 0;
		publish(false);
		if (pops >= max_rot_pops && flights_.empty()) break;
	}
	return pops;
}

StepStatus Engine::register_step(int max_rot_pops)
{
	DeviceGuard guard(dev_);
	int pops = 0;
	if (flow_mode()) {
		const double icp0 = icp_ms_;
		pops = flow_step(max_rot_pops);
		bnb_ms_ -= icp_ms_ - icp0;           // flow_step's clock also ran through the ICP runs it triggered
		publish(false);
	} else
	while (true) {
		if (!late_icp_.empty()) fold_late_icp(early_exit_ || converged_ || queue_.empty());      // nothing left to run beside: wait for every refinement
		if (early_exit_ || cancel_.load() || pops >= max_rot_pops) break;
		if (converged_ || queue_.empty()) { if (late_icp_.empty()) break; continue; }
		batches_done_++;
		// Rotation parents expanded together: ramps 8, 16, 32 ... rot_batch.  Easy registrations end in
		// the first rounds and pay for little speculation; long searches run with few, large launches
		// (full bunny: 310 launches / 63 ms at a fixed 8, 52 launches / 55 ms at 64).
		const int P = p_.wide_children ? std::max(1, std::min(p_.rot_batch, rot_ramp_)) : 1;
		rot_ramp_ = std::min(rot_ramp_ * 2, 1 << 20);
		std::vector<Node> parents;
		while ((int)parents.size() < P && !queue_.empty() && pops < max_rot_pops) {
			Node parent = queue_.top();
			if ((opt_err_ - parent.lb) <= sse_thresh_) {      // jly_goicp.cpp:416
				// single rank: global convergence.  sharded: this rank's frontier can no longer improve
				if (parents.empty()) { queue_.pop(); cnt_.rot_pops++; pops++; converged_ = true; }
				break;
			}
			queue_.pop();
			cnt_.rot_pops++;
			pops++;
			parents.push_back(parent);
		}
		if (parents.empty()) break;
		// ub_share: part of a batch goes to the queued cubes with the smallest upper bound seen inside them, whatever their lower
		// bound -- the early exit (jly_goicp.cpp:527) needs a pose below SSEThresh, and that is found by refining a promising
		// basin, not by closing the gap.  Valid: any expansion order keeps the bounds; the batch's first parents are still the
		// smallest lower bounds, so the stop rule (:416) sees the same frontier.
		if (p_.wide_children && p_.ub_share > 0.f && !converged_ && queue_.size() > 1 && pops < max_rot_pops) {
			size_t want = std::min<size_t>({(size_t)((float)P * p_.ub_share), queue_.size(), (size_t)(max_rot_pops - pops)});
			if (want > 0) {
				std::vector<Node> rest;
				rest.reserve(queue_.size());
				while (!queue_.empty()) { rest.push_back(queue_.top()); queue_.pop(); }
				std::partial_sort(rest.begin(), rest.begin() + (long)want, rest.end(), [](const Node& a, const Node& b) { return a.tie < b.tie; });
				for (size_t i = 0; i < want; i++) { parents.push_back(rest[i]); cnt_.rot_pops++; pops++; }
				for (size_t i = want; i < rest.size(); i++) queue_.push(rest[i]);
			}
		}
		process_parents(parents);
		publish(false);
	}
	if (!late_icp_.empty() && (early_exit_ || converged_ || queue_.empty())) fold_late_icp(true);
	StepStatus st{};
	st.early_exit = early_exit_ ? 1 : 0;
	st.finished = (early_exit_ || converged_ || (queue_.empty() && flights_.empty()) || cancel_.load()) ? 1 : 0;
	st.best_sse = opt_err_;
	st.frontier_lb = (queue_.empty() || converged_ || early_exit_) ? std::numeric_limits<float>::infinity() : queue_.top().lb;
	if (!converged_ && !early_exit_)
		for (const Flight& f : flights_) st.frontier_lb = std::min(st.frontier_lb, f.kid.parent_lb);   // children still in flight stand for their parents
	st.rot_pops = cnt_.rot_pops;
	// a step that neither popped a rotation node nor has searches in flight, with work still queued, would make every
	// caller loop (Engine::run, run_sharded) spin forever with the GPU idle: report it instead
	if (!st.finished && pops == 0 && flights_.empty()) throw std::logic_error("goicp: register_step made no progress with a non-empty rotation queue");
	return st;
}

int Engine::donate(int max_nodes, float* out)
{
	// every second cube in priority order leaves (at most max_nodes, never the whole queue): both sides keep
	// cubes of every priority, and the frontier's minimum lower bound stays with the donor
	std::vector<Node> all;
	all.reserve(queue_.size());
	while (!queue_.empty()) { all.push_back(queue_.top()); queue_.pop(); }
	int n = 0;
	for (size_t i = 0; i < all.size(); i++) {
		if ((i & 1) && n < max_nodes) {
			const Node& c = all[i];
			float* o = out + 7 * n++;
			o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w; o[4] = c.tie; o[5] = c.lb; o[6] = (float)c.l;
		} else queue_.push(all[i]);
	}
	return n;
}

void Engine::receive(const float* in, int n)
{
	for (int i = 0; i < n; i++) {
		const float* o = in + 7 * i;
		Node c{o[0], o[1], o[2], o[3], o[4], o[5], (int)o[6]};
		c.tie = o[4];                       // the ub slot of a travelling cube carries its tie-break key
		if (c.lb < opt_err_) queue_.push(c);
	}
	if (!queue_.empty() && !early_exit_) converged_ = false;
	publish(false);
}

void Engine::register_end()
{
	publish(true);
}

void Engine::run()
{
	DeviceGuard guard(dev_);
	TraceRange tr("goicp:register");
	double t0 = now_ms();
	register_begin();
	while (true) {
		StepStatus st = register_step(flow_mode() ? (1 << 20) : std::max(64, p_.rot_batch));     // one batch per step at full ramp (the flow drains its in-flight searches at the end of a step)
		if (st.finished) break;
	}
	register_ms_ = now_ms() - t0;
	if (p_.verbose)
		std::fprintf(stderr, "[goicp] register %.2f ms: inner BnB rounds %.2f ms (%lld launches: host build %.2f, GPU wait %.2f, host digest %.2f), ICP + DT re-score %.2f ms (%lld passes)\n",
		             register_ms_, bnb_ms_, cnt_.bounds_launches, t_submit_, t_wait_, t_collect_, icp_ms_, cnt_.icp_iters);
	if (p_.verbose) {
		std::fprintf(stderr, "[goicp] translation expansions by parent depth:");
		for (int l = 0; l < 32; l++) if (level_hist_[l]) std::fprintf(stderr, " %d:%lld", l, level_hist_[l]);
		std::fprintf(stderr, "\n");
		std::fprintf(stderr, "[goicp] device-queue expansions by [selection size][spread of the selection in voxels <=5 <=10 <=20 >20], tile rounds %lld, from tiles %lld:\n", tile_rounds_, cnt_.tile_expansions);
		const char* rows[4] = {"  n < 16 ", " 16..31  ", " 32..63  ", " 64..128 "};
		for (int a = 0; a < 4; a++) std::fprintf(stderr, "[goicp]  %s %12lld %12lld %12lld %12lld\n", rows[a], sel_hist_[a][0], sel_hist_[a][1], sel_hist_[a][2], sel_hist_[a][3]);
	}
	register_end();
}

}  // namespace goicp
