// Host-side construction of the implicit left-balanced k-d tree the NN kernels walk, plus Rodrigues.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <thread>
#include <functional>
#include <mutex>
#include <vector>
#include <atomic>

#include "engine.hpp"

namespace goicp {

// run fn(0..ntasks-1) on up to `threads` host threads (1 = inline); tasks are claimed from a counter
void parallel_tasks(int threads, int ntasks, const std::function<void(int)>& fn)
{
	threads = std::min({threads, ntasks, (int)std::max(1u, std::thread::hardware_concurrency())});
	if (threads <= 1) { for (int t = 0; t < ntasks; t++) fn(t); return; }
	std::atomic<int> next{0};
	std::exception_ptr err;
	std::mutex err_mtx;
	auto worker = [&] {
		try {
			for (int t = next.fetch_add(1); t < ntasks; t = next.fetch_add(1)) fn(t);
		} catch (...) {
			std::lock_guard<std::mutex> lk(err_mtx);
			if (!err) err = std::current_exception();
		}
	};
	std::vector<std::thread> pool;
	for (int i = 1; i < threads; i++) pool.emplace_back(worker);
	worker();
	for (auto& th : pool) th.join();
	if (err) std::rethrow_exception(err);
}

// Balanced binary k-d tree by median splits along the widest extent, of the smallest depth D whose
// 2^D leaves hold <= leaf_max points each, flattened into K = ceil(D/6) levels of 64-ary box groups.
// The ROOT group is the sparse one: it has F = 2^(D - 6(K-1)) real children (the binary nodes at that
// depth; the other child slots are empty boxes, which no query ever enters), every group below has
// 64.  A level-l group (l >= 1) is the binary node at depth d_l = log2(F) + 6(l-1) with index g, its
// children are the nodes 6 binary levels below (g*64 + c): the walker's node arithmetic needs no
// child pointers and no per-level fan-out.  Sizing the depth to the cloud keeps leaves full (100 k
// points: 8 192 leaves of 12 instead of 262 144 leaves of 0.4; 1 M: 65 536 of 15 instead of 262 144 of 4).
void build_kdtree(const float* xyz, int M, int leaf_max, KdHost* out)
{
	int D = 0;
	while (D < 6 * kMaxLevels && ((long long)leaf_max << D) < (long long)M) D++;
	if (((long long)leaf_max << D) < (long long)M) throw std::invalid_argument("goicp: target cloud too large for the k-d tree");
	const int K = std::max(1, (D + 5) / 6), L = 1 << D;
	const int d1 = D - 6 * (K - 1);              // binary depth of the root group's children; F = 2^d1
	out->K = K; out->L = L;
	std::vector<int> idx(M);
	std::iota(idx.begin(), idx.end(), 0);
	std::vector<int> lo(2 * (size_t)L, 0), hi(2 * (size_t)L, 0);   // heap order: node n -> children 2n, 2n+1
	lo[1] = 0; hi[1] = M;
	auto split_node = [&](int n) {
		const int a = lo[n], b = hi[n];
		const int mid = a + (b - a) / 2;
		if (b - a > 1) {
			float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
			for (int i = a; i < b; i++)
				for (int k = 0; k < 3; k++) {
					float v = xyz[3 * idx[i] + k];
					mn[k] = std::min(mn[k], v);
					mx[k] = std::max(mx[k], v);
				}
			int dim = 0;
			if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
			if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
			std::nth_element(idx.begin() + a, idx.begin() + mid, idx.begin() + b, [&](int p, int q) {
				float fp = xyz[3 * p + dim], fq = xyz[3 * q + dim];
				return fp < fq || (fp == fq && p < q);
			});
		}
		lo[2 * (size_t)n] = a; hi[2 * (size_t)n] = mid;
		lo[2 * (size_t)n + 1] = mid; hi[2 * (size_t)n + 1] = b;
	};
	// the four top levels level by level (1, 2, 4, 8 nodes side by side), then the 16 subtrees below them in parallel (disjoint
	// index ranges and heap slots; the result does not depend on the schedule): 1 M points in ~0.1 s
	// instead of 0.35 s
	constexpr int kTop = 16;
	const int threads = M >= (1 << 13) ? kTop : 1;
	for (int d = 0; (1 << d) < std::min(kTop, L); d++)
		parallel_tasks(threads, 1 << d, [&](int t) { split_node((1 << d) + t); });
	if (L > kTop)
		parallel_tasks(threads, kTop, [&](int t) {
			for (int d = 0; ((kTop + t) << d) < L; d++)
				for (int n = (kTop + t) << d; n < ((kTop + t + 1) << d); n++) split_node(n);
		});
	// leaves: node L + f, left to right
	float pad_w;
	const int pad_id = INT32_MAX;
	std::memcpy(&pad_w, &pad_id, sizeof(float));
	out->pts.assign((size_t)L * kLeafSlots, make_float4(INFINITY, INFINITY, INFINITY, pad_w));
	struct Box { float lo[3], hi[3]; };
	std::vector<Box> box(2 * (size_t)L);
	for (auto& bx : box) for (int k = 0; k < 3; k++) { bx.lo[k] = INFINITY; bx.hi[k] = -INFINITY; }
	for (int f = 0; f < L; f++) {
		const int a = lo[(size_t)L + f], b = hi[(size_t)L + f];
		for (int i = a; i < b; i++) {
			const int id = idx[i];
			float w;
			std::memcpy(&w, &id, sizeof(float));
			out->pts[(size_t)f * kLeafSlots + (i - a)] = make_float4(xyz[3 * id], xyz[3 * id + 1], xyz[3 * id + 2], w);
			for (int k = 0; k < 3; k++) {
				box[(size_t)L + f].lo[k] = std::min(box[(size_t)L + f].lo[k], xyz[3 * id + k]);
				box[(size_t)L + f].hi[k] = std::max(box[(size_t)L + f].hi[k], xyz[3 * id + k]);
			}
		}
	}
	for (int n = L - 1; n >= 1; n--)
		for (int k = 0; k < 3; k++) {
			box[n].lo[k] = std::min(box[2 * (size_t)n].lo[k], box[2 * (size_t)n + 1].lo[k]);
			box[n].hi[k] = std::max(box[2 * (size_t)n].hi[k], box[2 * (size_t)n + 1].hi[k]);
		}
	// level 0: the root group, children = the F nodes at depth d1; level l >= 1: the 2^(d1 + 6(l-1))
	// nodes at that depth, children = the nodes 6 levels below
	const Box empty{{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
	out->boxes.assign(K, std::vector<float>());
	for (int l = 0; l < K; l++) {
		const int dg = l == 0 ? 0 : d1 + 6 * (l - 1), dc = l == 0 ? d1 : dg + 6;   // depth of the groups / of their children
		const size_t groups = (size_t)1 << dg, fan = (size_t)1 << (dc - dg);
		out->boxes[l].assign(groups * 384, 0.f);
		for (size_t g = 0; g < groups; g++)
			for (size_t c = 0; c < 64; c++) {
				const Box& bx = c < fan ? box[((size_t)1 << dc) + g * fan + c] : empty;
				float* rec = &out->boxes[l][g * 384];
				for (int k = 0; k < 3; k++) { rec[64 * k + c] = bx.lo[k]; rec[192 + 64 * k + c] = bx.hi[k]; }
			}
	}
}

void rodrigues(float ax, float ay, float az, float R[9])
{
	// angle-axis vector (a rotation-cube centre) -> rotation matrix.  Float arithmetic with the same
	// operation order as GoICP::OuterBnB (jly_goicp.cpp:449-467), so R is bit-identical to the CPU path's.
	const float theta = std::sqrt(ax * ax + ay * ay + az * az);
	if (!(theta > 0.f)) {
		for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.f : 0.f;
		return;
	}
	const float ux = ax / theta, uy = ay / theta, uz = az / theta;   // unit axis
	const float c = std::cos(theta), omc = 1 - c, sn = std::sin(theta);
	const float xy = ux * uy * omc, xz = ux * uz * omc, yz = uy * uz * omc;
	const float zs = uz * sn, ys = uy * sn, xs = ux * sn;
	R[0] = c + ux * ux * omc; R[1] = xy - zs;           R[2] = xz + ys;
	R[3] = xy + zs;           R[4] = c + uy * uy * omc; R[5] = yz - xs;
	R[6] = xz - ys;           R[7] = yz + xs;           R[8] = c + uz * uz * omc;
}

}  // namespace goicp
