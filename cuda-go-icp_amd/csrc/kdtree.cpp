// Host-side construction of the implicit left-balanced k-d tree the NN kernel walks, plus the small
// dense maths of the ICP update (3x3 SVD -> rotation, Rodrigues).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

#include "engine.hpp"

namespace goicp {

// Heap-indexed tree: node n in [1, L) is internal with children 2n, 2n+1; nodes [L, 2L) are the
// leaves.  Median split by count along the widest extent of the node's points, so every leaf holds
// ceil/floor(M / L) <= leaf_max points and no child pointers are needed (the kernel derives
// parent/sibling indices arithmetically, which is what makes the stackless walk possible).
void build_kdtree(const float* xyz, int M, int leaf_max, KdHost* out)
{
	int D = 0;
	while (((M + (1 << D) - 1) >> D) > leaf_max) D++;
	const int L = 1 << D;
	out->L = L;
	out->nodes.assign(L, make_float2(0.f, 0.f));
	out->leaf_start.assign(L + 1, 0);
	std::vector<int> idx(M);
	std::iota(idx.begin(), idx.end(), 0);
	std::vector<int> lo(2 * L, 0), hi(2 * L, 0);
	lo[1] = 0; hi[1] = M;
	for (int n = 1; n < L; n++) {
		const int a = lo[n], b = hi[n];
		int dim = 0;
		float split = 0.f;
		int mid = a + (b - a) / 2;
		if (b > a) {
			float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
			for (int i = a; i < b; i++)
				for (int k = 0; k < 3; k++) {
					float v = xyz[3 * idx[i] + k];
					mn[k] = std::min(mn[k], v);
					mx[k] = std::max(mx[k], v);
				}
			if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
			if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
			if (mid < b) {
				std::nth_element(idx.begin() + a, idx.begin() + mid, idx.begin() + b, [&](int p, int q) {
					float fp = xyz[3 * p + dim], fq = xyz[3 * q + dim];
					return fp < fq || (fp == fq && p < q);
				});
				split = xyz[3 * idx[mid] + dim];   // left coords <= split <= right coords
			}
		}
		int dim_bits = dim;
		float dim_as_float;
		std::memcpy(&dim_as_float, &dim_bits, sizeof(float));
		out->nodes[n] = make_float2(split, dim_as_float);
		lo[2 * n] = a; hi[2 * n] = mid;
		lo[2 * n + 1] = mid; hi[2 * n + 1] = b;
	}
	out->pts.resize(M);
	for (int leaf = 0; leaf < L; leaf++) out->leaf_start[leaf] = lo[L + leaf];
	out->leaf_start[L] = M;
	if (L == 1) { out->leaf_start[0] = 0; }
	for (int i = 0; i < M; i++) {
		int id = idx[i];
		float w;
		std::memcpy(&w, &id, sizeof(float));
		out->pts[i] = make_float4(xyz[3 * id], xyz[3 * id + 1], xyz[3 * id + 2], w);
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

// 3x3 SVD by one-sided Jacobi rotations in double: A*V = U*diag(W)
static void svd3x3(const double A[9], double U[9], double W[3], double V[9])
{
	double B[9];
	std::memcpy(B, A, sizeof(B));
	const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	std::memcpy(V, I, sizeof(I));
	for (int sweep = 0; sweep < 64; sweep++) {
		double off = 0;
		for (int p = 0; p < 2; p++)
			for (int q = p + 1; q < 3; q++) {
				double app = 0, aqq = 0, apq = 0;
				for (int i = 0; i < 3; i++) {
					app += B[3 * i + p] * B[3 * i + p];
					aqq += B[3 * i + q] * B[3 * i + q];
					apq += B[3 * i + p] * B[3 * i + q];
				}
				off += apq * apq;
				if (std::fabs(apq) <= 1e-17 * std::sqrt(app * aqq) || apq == 0.0) continue;
				double zeta = (aqq - app) / (2 * apq);
				double tn = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
				double cs = 1 / std::sqrt(1 + tn * tn), sn = cs * tn;
				for (int i = 0; i < 3; i++) {
					double bp = B[3 * i + p], bq = B[3 * i + q];
					B[3 * i + p] = cs * bp - sn * bq;
					B[3 * i + q] = sn * bp + cs * bq;
					double vp = V[3 * i + p], vq = V[3 * i + q];
					V[3 * i + p] = cs * vp - sn * vq;
					V[3 * i + q] = sn * vp + cs * vq;
				}
			}
		if (off < 1e-60) break;
	}
	for (int j = 0; j < 3; j++) {
		double n = std::sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
		W[j] = n;
		for (int i = 0; i < 3; i++) U[3 * i + j] = n > 0 ? B[3 * i + j] / n : 0.0;
	}
	// rank-2 input: complete the missing left vector so U stays orthogonal
	for (int j = 0; j < 3; j++) {
		if (W[j] > 1e-200) continue;
		int a = (j + 1) % 3, b = (j + 2) % 3;
		if (W[a] <= 1e-200 || W[b] <= 1e-200) continue;
		U[j] = U[3 + a] * U[6 + b] - U[6 + a] * U[3 + b];
		U[3 + j] = U[6 + a] * U[b] - U[a] * U[6 + b];
		U[6 + j] = U[a] * U[3 + b] - U[3 + a] * U[b];
	}
}

void kabsch_rotation(const double H[9], float R[9])
{
	double U[9], W[3], V[9], VUt[9];
	svd3x3(H, U, W, V);
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) {
			double s = 0;
			for (int k = 0; k < 3; k++) s += V[3 * i + k] * U[3 * j + k];
			VUt[3 * i + j] = s;
		}
	double det = VUt[0] * (VUt[4] * VUt[8] - VUt[5] * VUt[7]) - VUt[1] * (VUt[3] * VUt[8] - VUt[5] * VUt[6]) +
	             VUt[2] * (VUt[3] * VUt[7] - VUt[4] * VUt[6]);
	// the reference sorts singular values in decreasing order (matrix.cpp:782-808), so its
	// diag(1,1,det) corrects the direction of the smallest one
	int ks = 0;
	if (W[1] < W[ks]) ks = 1;
	if (W[2] < W[ks]) ks = 2;
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) {
			double s = 0;
			for (int k = 0; k < 3; k++) s += V[3 * i + k] * (k == ks ? det : 1.0) * U[3 * j + k];
			R[3 * i + j] = (float)s;
		}
}

}  // namespace goicp
