// Host-side construction of the implicit left-balanced k-d tree the NN kernels walk, plus Rodrigues.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>

#include "engine.hpp"

namespace goicp {

// Heap-indexed tree: node n in [1, L) is internal with children 2n, 2n+1; nodes [L, 2L) are the
// leaves.  Median split by count along the widest extent of the node's points, so every leaf holds
// <= leaf_max points and no child pointers are needed (the kernel derives parent/sibling indices
// arithmetically, which is what makes the stackless walk possible).  Every internal node stores the
// tight boxes of its two children, quantised to 16 bits so that the decoded box (the same float
// expression the kernel evaluates) always CONTAINS the true box.
void build_kdtree(const float* xyz, int M, int leaf_max, KdHost* out)
{
	int D = 0;
	while (((M + (1 << D) - 1) >> D) > leaf_max) D++;
	const int L = 1 << D;
	out->L = L;
	std::vector<int> idx(M);
	std::iota(idx.begin(), idx.end(), 0);
	std::vector<int> lo(2 * L + 1, 0), hi(2 * L + 1, 0);
	lo[1] = 0; hi[1] = M;
	for (int n = 1; n < L; n++) {
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
		lo[2 * n] = a; hi[2 * n] = mid;
		lo[2 * n + 1] = mid; hi[2 * n + 1] = b;
	}
	// fixed-size leaves: kLeafSlots float4 slots per leaf (aligned 128-B lines), padded with +inf points
	float pad_w;
	const int pad_id = INT32_MAX;
	std::memcpy(&pad_w, &pad_id, sizeof(float));
	out->pts.assign((size_t)L * kLeafSlots, make_float4(INFINITY, INFINITY, INFINITY, pad_w));
	struct Box { float lo[3], hi[3]; };
	std::vector<Box> box(2 * (size_t)L);
	for (auto& bx : box) for (int k = 0; k < 3; k++) { bx.lo[k] = INFINITY; bx.hi[k] = -INFINITY; }
	for (int leaf = 0; leaf < L; leaf++) {
		const int a = lo[L + leaf], b = hi[L + leaf];
		for (int i = a; i < b; i++) {
			const int id = idx[i];
			float w;
			std::memcpy(&w, &id, sizeof(float));
			out->pts[(size_t)leaf * kLeafSlots + (i - a)] = make_float4(xyz[3 * id], xyz[3 * id + 1], xyz[3 * id + 2], w);
			for (int k = 0; k < 3; k++) {
				box[L + leaf].lo[k] = std::min(box[L + leaf].lo[k], xyz[3 * id + k]);
				box[L + leaf].hi[k] = std::max(box[L + leaf].hi[k], xyz[3 * id + k]);
			}
		}
	}
	for (int n = L - 1; n >= 1; n--)
		for (int k = 0; k < 3; k++) {
			box[n].lo[k] = std::min(box[2 * n].lo[k], box[2 * n + 1].lo[k]);
			box[n].hi[k] = std::max(box[2 * n].hi[k], box[2 * n + 1].hi[k]);
		}
	for (int k = 0; k < 3; k++) {
		out->root_lo[k] = box[1].lo[k];
		float ext = box[1].hi[k] - box[1].lo[k];
		out->step[k] = ext > 0.f ? ext / 65535.0f : 1e-30f;
	}
	// the decode expression of the kernel (device.hip box_lb), float, not contracted
	auto decode = [&](int k, unsigned code) { return out->root_lo[k] + (float)code * out->step[k]; };
	for (int k = 0; k < 3; k++)
		while (decode(k, 65535u) < box[1].hi[k]) out->step[k] *= 1.000001f;   // the top code must reach the root's upper face
	auto quant = [&](const Box& bx, int k) -> unsigned {
		if (!(bx.lo[k] <= bx.hi[k])) return 0xffffu | (0u << 16);            // empty child: inverted box, never entered
		long cl = (long)std::floor((double)(bx.lo[k] - out->root_lo[k]) / (double)out->step[k]);
		long ch = (long)std::ceil((double)(bx.hi[k] - out->root_lo[k]) / (double)out->step[k]);
		cl = std::min(65535L, std::max(0L, cl));
		ch = std::min(65535L, std::max(0L, ch));
		while (cl > 0 && decode(k, (unsigned)cl) > bx.lo[k]) cl--;            // conservative under the kernel's rounding
		while (ch < 65535 && decode(k, (unsigned)ch) < bx.hi[k]) ch++;
		return (unsigned)cl | ((unsigned)ch << 16);
	};
	out->boxes.assign((size_t)L * 3, make_uint2(0xffffu, 0xffffu));
	for (int n = 1; n < L; n++) {
		const Box &l = box[2 * n], &r = box[2 * n + 1];
		out->boxes[3 * (size_t)n] = make_uint2(quant(l, 0), quant(l, 1));
		out->boxes[3 * (size_t)n + 1] = make_uint2(quant(l, 2), quant(r, 0));
		out->boxes[3 * (size_t)n + 2] = make_uint2(quant(r, 1), quant(r, 2));
	}
}

void rodrigues(float v1, float v2, float v3, float R[9])
{
	// angle-axis cube centre -> rotation matrix, float arithmetic in the order of jly_goicp.cpp:449-467
	float t = std::sqrt(v1 * v1 + v2 * v2 + v3 * v3);
	if (!(t > 0.f)) {
		const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
		std::memcpy(R, I, sizeof(I));
		return;
	}
	v1 /= t; v2 /= t; v3 /= t;
	float ct = std::cos(t), ct2 = 1 - ct, st = std::sin(t);
	float tmp121 = v1 * v2 * ct2, tmp122 = v3 * st;
	float tmp131 = v1 * v3 * ct2, tmp132 = v2 * st;
	float tmp231 = v2 * v3 * ct2, tmp232 = v1 * st;
	R[0] = ct + v1 * v1 * ct2; R[1] = tmp121 - tmp122;    R[2] = tmp131 + tmp132;
	R[3] = tmp121 + tmp122;    R[4] = ct + v2 * v2 * ct2; R[5] = tmp231 - tmp232;
	R[6] = tmp131 - tmp132;    R[7] = tmp231 + tmp232;    R[8] = ct + v3 * v3 * ct2;
}

}  // namespace goicp
