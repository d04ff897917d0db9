// HIP kernels of the MI355X Go-ICP engine (gfx950 / CDNA4, wave64).  Compiled with
// -ffp-contract=off: every per-point value follows the reference CPU path's float/double operation
// order (no FMA contraction), so distances, clamps and squares are bit-identical to
// src/goicp/jly_goicp.cpp / jly_3ddt.cpp / jly_icp3d.hpp; only the order of the sums differs.
//
// Kernels
//   bounds_kernel        (a) BnB cube bounds: rotate + translate the source cloud, gather the 3-D
//                        Euclidean distance transform, subtract uncertainty radii, sum of squares.
//                        16 B (ub pass) / 20 B (lb pass) algorithmic per point; sibling fast path.
//   bounds_finalize      fixed-order sum of the per-chunk partials (deterministic, no float atomics)
//   bounds_trim_kernel   trimmed form: exact k-th smallest residual per cube by radix select
//   icp_pass_kernel      (b) one ICP correspondence pass: transform, exact 1-NN (four queries per
//                        wavefront over a 64-ary box hierarchy, DT-seeded bound), pivoted sums
//   icp_finalize_update  fixed-order double-precision reduction + the rest of the ICP loop body
//                        (convergence test, SVD, pose update) on the device-resident state
//   icp_nn/select/accum  trimmed ICP: NN of every point, radix select of the num nearest, sums
//   transform_kernel     rigid transform apply
//   nn_query_kernel      1-NN operator for arbitrary queries
//   dt_*                 exact Euclidean DT build (seed, three separable min-plus passes, sqrt/scale)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cmath>

#include "device.hpp"

namespace goicp {

// ------------------------------------------------------------------------------------------------
// wave64 helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
	return v;
}

// XCD-aware bijective block remap: blocks b and b+8 share an XCD (round-robin dispatch), so give
// each XCD a contiguous slice of the logical grid -> neighbouring cube groups (neighbouring DT
// lines) meet in one L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nb)
{
	int q = nb >> 3, r = nb & 7, xcd = bid & 7;
	return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ------------------------------------------------------------------------------------------------
// DT lookup = DT3D::Distance (jly_3ddt.cpp:981-1026), nearest voxel with int(x+0.5) truncation
// ------------------------------------------------------------------------------------------------
// layout 2 (opt-in, Params::bounds_fp16): the bricked grid in IEEE half precision -- a 4x4x4 brick is then ONE 128-byte
// line instead of two, which halves the distinct lines a 64-lane gather touches (the cost the bounds kernel is bound
// by).  Values are rounded toward zero when the copy is made, so every lower bound stays a valid lower bound; upper
// bounds come out low by at most 2^-10 relative.  Index arithmetic is the bricked layout's.
__device__ __forceinline__ float half_bits_to_float(unsigned short h)
{
	return __half2float(__ushort_as_half(h));
}
// element e of a half grid through a 4-byte load of the aligned pair that holds it (measured: 2-byte gathers run at a
// fraction of the dword rate -- the S2 launch took 80.6 ms with global_load_ushort against 53.8 ms in fp32)
__device__ __forceinline__ float half_fetch(const float* grid, unsigned e)
{
	const unsigned w = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(grid) + (size_t)((e >> 1) << 2));
	return half_bits_to_float((unsigned short)((e & 1u) ? (w >> 16) : (w & 0xffffu)));
}

template <int LAYOUT>
__device__ __forceinline__ float dt_fetch(const DtDesc& dt, int x, int y, int z)
{
	// 24-bit multiplies (full rate) and a 32-bit byte offset from a scalar base: V <= 640 keeps
	// every offset below 2^32 (engine.cpp enforces it)
	unsigned e;
	if (LAYOUT == 0) {
		e = __umul24(__umul24((unsigned)z, (unsigned)dt.V) + (unsigned)y, (unsigned)dt.V) + (unsigned)x;
	} else {
		const unsigned b = __umul24(__umul24((unsigned)z >> 2, (unsigned)dt.VB) + ((unsigned)y >> 2), (unsigned)dt.VB) + ((unsigned)x >> 2);
		e = (b << 6) | ((((unsigned)z & 3u) << 4) | (((unsigned)y & 3u) << 2) | ((unsigned)x & 3u));
	}
	if (LAYOUT == 2) return half_fetch(dt.grid, e);
	return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(dt.grid) + (size_t)(e << 2));
}

// voxel index of one coordinate: the reference computes int((double(q) - min)*scale + 0.5) in double
// (jly_3ddt.cpp:984-986).  Fast path: the same expression in float (1 sub + 1 fma) is within
// eps(F) = c1 + c2*|F| of the exact value (rounding of min, scale, the difference and the fma;
// DtDesc.c1/c2 are computed on the host from the grid geometry); whenever F is farther than eps
// from the nearest integer, truncating F gives the reference's index.  Otherwise -- a few lanes
// in 10^4 -- the caller recomputes in double.  Result: bit-identical indices at ~1/3 of the cost.
__device__ __forceinline__ int voxel_fast(float q, float mn, float scale, float c1, float c2, bool& risky)
{
	const float F = __fmaf_rn(q - mn, scale, 0.5f);
	risky = risky || (fabsf(F - rintf(F)) <= __fmaf_rn(fabsf(F), c2, c1));
	return (int)F;
}

// reference expression (jly_3ddt.cpp:984-986), taken by the few lanes in 10^4 whose float index is not
// provably right.  The empty asm keeps the caller's `if (risky)` a real branch: left to itself the
// compiler if-converts it and puts the five fp64 instructions of every index on every lane (measured:
// 30 of the 255 VALU instructions per point of the sibling path, at half rate).
__device__ __forceinline__ int voxel_exact(float q, double mn, double scale)
{
	__asm__ volatile("");
	return (int)(((double)q - mn) * scale + 0.5);
}

template <int LAYOUT>
__device__ __forceinline__ float dt_distance(const DtDesc& dt, float qx, float qy, float qz)
{
	bool risky = false;
	int x = voxel_fast(qx, dt.xmin_f, dt.scale_f, dt.c1, dt.c2, risky);
	int y = voxel_fast(qy, dt.ymin_f, dt.scale_f, dt.c1, dt.c2, risky);
	int z = voxel_fast(qz, dt.zmin_f, dt.scale_f, dt.c1, dt.c2, risky);
	if (risky) {
		// double index math exactly as the reference: (x - xMin)*scale + 0.5, truncate
		x = voxel_exact(qx, dt.xmin, dt.scale);
		y = voxel_exact(qy, dt.ymin, dt.scale);
		z = voxel_exact(qz, dt.zmin, dt.scale);
	}
	const int V = dt.V;
	if ((unsigned)x < (unsigned)V && (unsigned)y < (unsigned)V && (unsigned)z < (unsigned)V)
		return dt_fetch<LAYOUT>(dt, x, y, z);
	// outside the grid: clamp to the face and add the overshoot (in voxels) / scale
	float a = 0.f, b = 0.f, c = 0.f;
	if (x < 0) { a = (float)x; x = 0; } else if (x >= V) { a = (float)(x - V + 1); x = V - 1; }
	if (y < 0) { b = (float)y; y = 0; } else if (y >= V) { b = (float)(y - V + 1); y = V - 1; }
	if (z < 0) { c = (float)z; z = 0; } else if (z >= V) { c = (float)(z - V + 1); z = V - 1; }
	// sqrt(a^2+b^2+c^2)/scale: a, b, c are small integers, so the term is a function of the integer
	// s = a^2+b^2+c^2 only.  DtDesc.overshoot[s] holds (double)sqrtf(s)/scale computed on the host with
	// the same operations (bit-identical), which replaces a float sqrt and an fp64 division on this path.
	const int si = (int)(a * a + b * b + c * c);
	const double ext = si < dt.n_overshoot ? dt.overshoot[si] : (double)__fsqrt_rn(a * a + b * b + c * c) / dt.scale;
	return (float)(ext + (double)dt_fetch<LAYOUT>(dt, x, y, z));
}

// Per-axis part of a voxel's element offset.  Both layouts are separable: offset = fx(x)+fy(y)+fz(z);
// an out-of-grid index contributes kOutside (>= 2^30 > any in-grid sum, and three of them do not wrap),
// so one unsigned compare of the sum tells whether the fast fetch is valid.
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr unsigned kOutside = 0x40000000u;
template <int LAYOUT, int AXIS>
__device__ __forceinline__ unsigned axis_offset(const DtDesc& dt, int i)
{
	if ((unsigned)i >= (unsigned)dt.V) return kOutside;
	const unsigned u = (unsigned)i;
	if (LAYOUT == 0) return AXIS == 0 ? u : (AXIS == 1 ? __umul24(u, (unsigned)dt.V) : __umul24(u, (unsigned)(dt.V * dt.V)));
	const unsigned hi = u >> 2, lo = u & 3u;
	if (AXIS == 0) return (hi << 6) | lo;
	if (AXIS == 1) return (__umul24(hi, (unsigned)dt.VB) << 6) | (lo << 2);
	return (__umul24(hi, (unsigned)(dt.VB * dt.VB)) << 6) | (lo << 4);
}

// the 8 cube records of workgroup `group`: read from the batch, or derived from one expansion record with
// the search driver's own float operations (engine.cpp / jly_goicp.cpp:262-273: w' = w/2, corner +
// bit*w', centre = corner + w'/2, delta = float(sqrt3/2 * double(w')))
__device__ __forceinline__ void load_group(const CubeRec* __restrict__ cubes, const ParentRec* __restrict__ parents, int group, int B,
                                           CubeRec cr[kGroup])
{
	if (parents) {
		const ParentRec pr = parents[group];
		const float w = pr.w / 2;
		const float delta = (float)(1.732050808 / 2.0 * (double)w);
#pragma unroll
		for (int c = 0; c < kGroup; c++) {
			const float cx = pr.x + (float)(c & 1) * w, cy = pr.y + (float)((c >> 1) & 1) * w, cz = pr.z + (float)((c >> 2) & 1) * w;
			cr[c].tx = cx + w / 2; cr[c].ty = cy + w / 2; cr[c].tz = cz + w / 2;
			cr[c].delta = delta; cr[c].coeff = pr.coeff; cr[c].rot = pr.rot;
		}
	} else {
		const int c0 = group * kGroup;
#pragma unroll
		for (int c = 0; c < kGroup; c++) cr[c] = cubes[c0 + c < B ? c0 + c : B - 1];
	}
}

// The clamped residuals max(DT(R p + t_c) - rho, 0) of one point under the 8 children of one expansion
// (one rotation, per axis two translation values): 6 voxel indices instead of 24.  Every value is the same
// float expression as in the generic per-cube path, so the results are bit-identical.
struct SiblingSet { float tx0, tx1, ty0, ty1, tz0, tz1; };
// rp = R p already formed (xyz)
template <int LAYOUT>
__device__ __forceinline__ void sibling_residuals_rotated(const DtDesc& dt, const float4& rp, const SiblingSet& t, float rho, float m[kGroup]);
template <int LAYOUT>
__device__ __forceinline__ void sibling_residuals(const DtDesc& dt, const Rot9& R0, const SiblingSet& t, const float4& p, float rho,
                                                  float m[kGroup])
{
	const float rx = R0.r[0] * p.x + R0.r[1] * p.y + R0.r[2] * p.z;
	const float ry = R0.r[3] * p.x + R0.r[4] * p.y + R0.r[5] * p.z;
	const float rz = R0.r[6] * p.x + R0.r[7] * p.y + R0.r[8] * p.z;
	sibling_residuals_rotated<LAYOUT>(dt, make_float4(rx, ry, rz, p.w), t, rho, m);
}
template <int LAYOUT>
__device__ __forceinline__ void sibling_residuals_rotated(const DtDesc& dt, const float4& rp, const SiblingSet& t, float rho, float m[kGroup])
{
	const float rx = rp.x, ry = rp.y, rz = rp.z;
	const float qx[2] = {rx + t.tx0, rx + t.tx1}, qy[2] = {ry + t.ty0, ry + t.ty1}, qz[2] = {rz + t.tz0, rz + t.tz1};
	bool risky = false;
	int ix[2], iy[2], iz[2];
#pragma unroll
	for (int k = 0; k < 2; k++) {
		ix[k] = voxel_fast(qx[k], dt.xmin_f, dt.scale_f, dt.c1, dt.c2, risky);
		iy[k] = voxel_fast(qy[k], dt.ymin_f, dt.scale_f, dt.c1, dt.c2, risky);
		iz[k] = voxel_fast(qz[k], dt.zmin_f, dt.scale_f, dt.c1, dt.c2, risky);
	}
	if (risky) {
#pragma unroll
		for (int k = 0; k < 2; k++) {
			ix[k] = voxel_exact(qx[k], dt.xmin, dt.scale);
			iy[k] = voxel_exact(qy[k], dt.ymin, dt.scale);
			iz[k] = voxel_exact(qz[k], dt.zmin, dt.scale);
		}
	}
	const unsigned fx[2] = {axis_offset<LAYOUT, 0>(dt, ix[0]), axis_offset<LAYOUT, 0>(dt, ix[1])};
	const unsigned fy[2] = {axis_offset<LAYOUT, 1>(dt, iy[0]), axis_offset<LAYOUT, 1>(dt, iy[1])};
	const unsigned fz[2] = {axis_offset<LAYOUT, 2>(dt, iz[0]), axis_offset<LAYOUT, 2>(dt, iz[1])};
#pragma unroll
	for (int c = 0; c < kGroup; c++) {
		const unsigned e = fx[c & 1] + fy[(c >> 1) & 1] + fz[(c >> 2) & 1];
		float v;
		if (e < kOutside) {
			if (LAYOUT == 2) v = half_fetch(dt.grid, e);
			else v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(dt.grid) + (size_t)(e << 2));
		}
		else v = dt_distance<LAYOUT>(dt, qx[c & 1], qy[(c >> 1) & 1], qz[(c >> 2) & 1]);   // clamp + overshoot extension
		v = v - rho;
		m[c] = v < 0.f ? 0.f : v;
	}
}
// do the 8 records describe the children of one expansion?  (jly_goicp.cpp:267-273: corner + bit*w + w/2)
__device__ __forceinline__ bool is_sibling_set(const CubeRec cr[kGroup])
{
	bool sib = true;
#pragma unroll
	for (int c = 0; c < kGroup; c++)
		sib = sib && cr[c].rot == cr[0].rot && cr[c].tx == cr[c & 1].tx && cr[c].ty == cr[c & 2].ty && cr[c].tz == cr[c & 4].tz &&
		      cr[c].delta == cr[0].delta && cr[c].coeff == cr[0].coeff;
	return sib;
}

// ------------------------------------------------------------------------------------------------
// (a) cube bounds
// ------------------------------------------------------------------------------------------------
// grid: groups*chunks blocks of 256 threads.  A block owns kGroup consecutive cubes (the siblings
// of one BnB expansion: same rotation, neighbouring translations -> neighbouring DT voxels) and a
// contiguous chunk of the (k-d-ordered) source cloud; each point is loaded once (16 B) and
// reused for the 8 cubes.
//
// The lean sibling path over the points [p0, p1) of a chunk (round 3; fp32 grids that fit the Infinity Cache): one exactness test per
// point on the smallest margin of the six indices; a wavefront whose indices are all inside the grid fetches without per-child checks;
// the eight children are accumulated as four x-sibling pairs in packed fp32 -- the same operations per element as the checked path,
// bit-identical bounds.  NP = 2: the SAME expansion listed by both searches of a rotation child (coeff[0] = the lower-bound pass's
// coefficient, coeff[1] = 0 of the upper-bound pass): one gather serves both, only the subtraction of the rotation radius and the
// sums are per pass -- each pass's sums see exactly the operations of a separate evaluation.
// LAST_ZERO: coeff[NP - 1] is known to be 0 (the upper-bound pass, jly_goicp.cpp:284-285 with maxRotDis = 0): its residual is the looked-up
// distance itself -- v - 0 and max(v, 0) are v bit for bit, the DT holds no negative value -- so the subtraction and the clamp are not issued.
template <int LAYOUT, int NP, bool LAST_ZERO>
__device__ __forceinline__ void lean_points(const float4* __restrict__ src, int p0, int p1, const DtDesc& dt, const Rot9& R0, const SiblingSet& ts, float delta,
                                            const float (&coeff)[NP], f2 (&ub2)[NP][4], f2 (&lb2)[NP][4])
{
	const f2 delta2 = f2{delta, delta}, ndelta2 = f2{-delta, -delta};
	// one bound for the six margins: on the unchecked branch below every index is inside the grid, i.e. |F| < V, and eps(F) = c1 + c2 |F| grows with
	// |F|; a lane whose F lies outside takes the checked branch, which tests for itself (16 instead of 30 instructions for the test)
	const float eps_grid = __fmaf_rn((float)(dt.V + 1), dt.c2, dt.c1);
	for (int i = p0 + (int)threadIdx.x; i < p1; i += kBoundsThreads) {
		const float4 p = src[i];
		const float rx = R0.r[0] * p.x + R0.r[1] * p.y + R0.r[2] * p.z;
		const float ry = R0.r[3] * p.x + R0.r[4] * p.y + R0.r[5] * p.z;
		const float rz = R0.r[6] * p.x + R0.r[7] * p.y + R0.r[8] * p.z;
		const float qx[2] = {rx + ts.tx0, rx + ts.tx1}, qy[2] = {ry + ts.ty0, ry + ts.ty1}, qz[2] = {rz + ts.tz0, rz + ts.tz1};
		float F[6] = {__fmaf_rn(qx[0] - dt.xmin_f, dt.scale_f, 0.5f), __fmaf_rn(qx[1] - dt.xmin_f, dt.scale_f, 0.5f),
		              __fmaf_rn(qy[0] - dt.ymin_f, dt.scale_f, 0.5f), __fmaf_rn(qy[1] - dt.ymin_f, dt.scale_f, 0.5f),
		              __fmaf_rn(qz[0] - dt.zmin_f, dt.scale_f, 0.5f), __fmaf_rn(qz[1] - dt.zmin_f, dt.scale_f, 0.5f)};
		float worst = INFINITY;
#pragma unroll
		for (int k = 0; k < 6; k++) worst = fminf(worst, fabsf(F[k] - rintf(F[k])));
		int ix[2] = {(int)F[0], (int)F[1]}, iy[2] = {(int)F[2], (int)F[3]}, iz[2] = {(int)F[4], (int)F[5]};
		if (worst <= eps_grid) {
#pragma unroll
			for (int k = 0; k < 2; k++) {
				ix[k] = voxel_exact(qx[k], dt.xmin, dt.scale);
				iy[k] = voxel_exact(qy[k], dt.ymin, dt.scale);
				iz[k] = voxel_exact(qz[k], dt.zmin, dt.scale);
			}
		}
		const unsigned V = (unsigned)dt.V;
		const bool inside = (unsigned)ix[0] < V && (unsigned)ix[1] < V && (unsigned)iy[0] < V && (unsigned)iy[1] < V && (unsigned)iz[0] < V && (unsigned)iz[1] < V;
		float rho[NP];
#pragma unroll
		for (int q = 0; q < NP; q++) rho[q] = coeff[q] * p.w;
		if (__all(inside)) {
			unsigned fx[2], fy[2], fz[2];
#pragma unroll
			for (int k = 0; k < 2; k++) {
				// BYTE offsets (the lean variant only runs on grids of at most 256 MB: they fit 32 bits)
				if (LAYOUT == 0) { fx[k] = (unsigned)ix[k] << 2; fy[k] = __umul24((unsigned)iy[k], V) << 2; fz[k] = __umul24((unsigned)iz[k], V * V) << 2; }
				else {
					fx[k] = (((unsigned)ix[k] >> 2) << 8) | (((unsigned)ix[k] & 3u) << 2);
					fy[k] = (__umul24((unsigned)iy[k] >> 2, (unsigned)dt.VB) << 8) | (((unsigned)iy[k] & 3u) << 4);
					fz[k] = (__umul24((unsigned)iz[k] >> 2, (unsigned)(dt.VB * dt.VB)) << 8) | (((unsigned)iz[k] & 3u) << 6);
				}
			}
			const char* gb = reinterpret_cast<const char*>(dt.grid);
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const unsigned yz = fy[k & 1] + fz[k >> 1];
				const f2 v = f2{*reinterpret_cast<const float*>(gb + (fx[0] + yz)), *reinterpret_cast<const float*>(gb + (fx[1] + yz))};
#pragma unroll
				for (int q = 0; q < NP; q++) {
					f2 vq = v, mm = v;
					if (!(LAST_ZERO && q == NP - 1)) {
						vq = v - f2{rho[q], rho[q]};
						mm = __builtin_elementwise_max(vq, f2{0.f, 0.f});
					}
					ub2[q][k] = ub2[q][k] + mm * mm;
					// max(mm - delta, 0) == max(vq - delta, 0) bit for bit (delta >= 0: vq < 0 gives 0 either way, vq >= 0 is mm; x + (-d) is x - d)
					const f2 dis = __builtin_elementwise_max(vq + ndelta2, f2{0.f, 0.f});
					lb2[q][k] = lb2[q][k] + dis * dis;
				}
			}
		} else {
#pragma unroll
			for (int q = 0; q < NP; q++) {
				float m[kGroup];
				sibling_residuals<LAYOUT>(dt, R0, ts, p, rho[q], m);
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const f2 mm = f2{m[2 * k], m[2 * k + 1]};
					ub2[q][k] = ub2[q][k] + mm * mm;
					const f2 dis = __builtin_elementwise_max(mm - delta2, f2{0.f, 0.f});
					lb2[q][k] = lb2[q][k] + dis * dis;
				}
			}
		}
	}
}

// the sums of one (expansion, chunk) work item: wave64 reduce, then the 4 wavefronts through LDS, then the item's row of the scratch block (or,
// unsplit, the bounds themselves)
__device__ __forceinline__ void bounds_item_store(const float (&ub)[kGroup], const float (&lb)[kGroup], int group, int chunk, int chunks, int B,
                                                  float* __restrict__ scratch, float* __restrict__ ub_out, float* __restrict__ lb_out, float (*red)[2 * kGroup])
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c0 = group * kGroup;
#pragma unroll
	for (int c = 0; c < kGroup; c++) {
		float u = wave_sum(ub[c]), l = wave_sum(lb[c]);
		if (lane == 0) { red[wave][c] = u; red[wave][kGroup + c] = l; }
	}
	__syncthreads();
	if (threadIdx.x < 2 * kGroup) {
		float s = red[0][threadIdx.x];
#pragma unroll
		for (int w = 1; w < kBoundsThreads / 64; w++) s += red[w][threadIdx.x];
		if (chunks == 1) {
			int c = threadIdx.x & (kGroup - 1);
			if (c0 + c < B) (threadIdx.x < kGroup ? ub_out : lb_out)[c0 + c] = s;
		} else {
			scratch[((size_t)group * chunks + chunk) * (2 * kGroup) + threadIdx.x] = s;
		}
	}
}

// twin test of the device-resident searches (bnbqueue.hip): the two searches of a rotation child -- upper-bound pass (coeff 0) and
// lower-bound pass -- often list the SAME translation node in the same round (the root always, most depth-1 and depth-2 nodes of the
// upper-bound pass).  The gathers of such a pair are identical; the lower-bound pass's item then evaluates both, the other leaves.
struct TwinCtx { const QSearch* searches; const int* psearch; int* sh; };
// a round of at least min_groups expansions walks its items in footprint order (launch_queue_sort below)
constexpr int kSortBins = 1 << 15;                // Morton cells, 5 bits per axis
__device__ __forceinline__ bool qsort_on(const QSort& qs, int ngroups) { return qs.order != nullptr && ngroups >= qs.min_groups; }

// work item `work` of `total` = groups*chunks: one (cube group, point chunk) pair
template <int LAYOUT, bool LEAN>
__device__ __forceinline__ void bounds_work(
    int work, int total, const float4* __restrict__ src, int N, const DtDesc& dt, const Rot9* __restrict__ rots,
    const CubeRec* __restrict__ cubes, const ParentRec* __restrict__ parents, int B, int groups, int chunks, int chunk_pts,
    float* __restrict__ scratch, float* __restrict__ ub_out, float* __restrict__ lb_out, float (*red)[2 * kGroup],
    const unsigned* __restrict__ order = nullptr, const TwinCtx tw = TwinCtx{nullptr, nullptr, nullptr})
{
	// XCD-aware tiling (speed only): blocks b and b+8 share an XCD (round-robin dispatch).  XCD x owns
	// the point chunks [x*cpx, (x+1)*cpx) -- a compact spatial patch of the k-d-ordered cloud -- and
	// walks the cube groups in order, so at any time one L2 serves gathers into the DT neighbourhood
	// of ONE patch under nearby translations (a few MB) instead of the whole surface band.
	int chunk, group;
	if (order) {
		// footprint-ordered items: XCD x walks the x-th eighth of the sorted list
		const int q = total >> 3, r = total & 7, x = work & 7, s = work >> 3;
		const unsigned t = order[(x <= r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s];
		group = (int)(t / (unsigned)chunks);
		chunk = (int)t - group * chunks;
	} else if ((chunks & 7) == 0) {
		const int cpx = chunks >> 3, xcd = work & 7, slot = work >> 3;
		group = slot / cpx;
		chunk = xcd * cpx + (slot - group * cpx);
	} else {
		const int swz = xcd_remap(work, total);
		chunk = swz / groups; group = swz - chunk * groups;
	}
	const int c0 = group * kGroup;

	CubeRec cr[kGroup];
	load_group(cubes, parents, group, B, cr);
	bool uniform = true;
#pragma unroll
	for (int c = 0; c < kGroup; c++) uniform = uniform && (cr[c].rot == cr[0].rot);
	const Rot9 R0 = rots[cr[0].rot];
	// The 8 children of one BnB expansion share, per axis, only TWO translation values
	// (jly_goicp.cpp:267-273: corner + (j>>a & 1)*w + w/2), one rotation, one delta, one coeff.
	bool siblings = uniform && c0 + kGroup <= B;
#pragma unroll
	for (int c = 0; c < kGroup; c++)
		siblings = siblings && cr[c].tx == cr[c & 1].tx && cr[c].ty == cr[c & 2].ty && cr[c].tz == cr[c & 4].tz &&
		           cr[c].delta == cr[0].delta && cr[c].coeff == cr[0].coeff;

	float ub[kGroup], lb[kGroup];
#pragma unroll
	for (int c = 0; c < kGroup; c++) { ub[c] = 0.f; lb[c] = 0.f; }

	const int p0 = chunk * chunk_pts;
	const int p1 = p0 + chunk_pts < N ? p0 + chunk_pts : N;
	if (siblings) {
		// fast path: 6 voxel-index computations per point instead of 24; every per-point value is the
		// same float expression as in the generic path below, so the results are bit-identical
		const SiblingSet ts{cr[0].tx, cr[1].tx, cr[0].ty, cr[2].ty, cr[0].tz, cr[4].tz};
		const float delta = cr[0].delta, coeff = cr[0].coeff;
		// LEAN: lean_points (above).  Bunny microbench 1.837 -> 1.791 ms per launch, full registration 42.1 -> 40.6 ms; at 1 M points / 512^3
		// (HBM-bound) the same code is 13 % SLOWER (53.3 -> 60.5 ms), so the launch picks it by the size of the grid.
		if constexpr (LEAN && LAYOUT != 2) {
			// twin test (device-resident searches only): is this node also listed by the other search of the rotation child?
			int twin = -1;
			if (tw.searches) {
				const int ts_ = tw.searches[tw.psearch[group]].twin;
				if (ts_ >= 0) {
					const int tn = tw.searches[ts_].n_parents, toff = tw.searches[ts_].parent_off;
					if (tn > 0 && tw.searches[ts_].tile == 0) {
						if (threadIdx.x == 0) *tw.sh = -1;
						__syncthreads();
						for (int j = threadIdx.x; j < tn; j += kBoundsThreads) {       // tn <= kQueueMaxPop
							const ParentRec o = parents[toff + j], me = parents[group];
							if (o.x == me.x && o.y == me.y && o.z == me.z && o.w == me.w && o.rot == me.rot && (o.coeff == 0.f) != (me.coeff == 0.f)) *tw.sh = toff + j;
						}
						__syncthreads();
						twin = *tw.sh;
					}
				}
			}
			if (twin >= 0 && coeff == 0.f) return;              // the lower-bound pass's item evaluates this expansion for both
			if (twin >= 0) {
				f2 ub2[2][4], lb2[2][4];
#pragma unroll
				for (int q = 0; q < 2; q++)
#pragma unroll
					for (int k = 0; k < 4; k++) { ub2[q][k] = f2{0.f, 0.f}; lb2[q][k] = f2{0.f, 0.f}; }
				const float co[2] = {coeff, 0.f};
				lean_points<LAYOUT, 2, true>(src, p0, p1, dt, R0, ts, delta, co, ub2, lb2);
#pragma unroll
				for (int k = 0; k < 4; k++) { ub[2 * k] = ub2[1][k].x; ub[2 * k + 1] = ub2[1][k].y; lb[2 * k] = lb2[1][k].x; lb[2 * k + 1] = lb2[1][k].y; }
				bounds_item_store(ub, lb, twin, chunk, chunks, B, scratch, ub_out, lb_out, red);
				__syncthreads();                                  // `red` is reused for the item's own sums
#pragma unroll
				for (int k = 0; k < 4; k++) { ub[2 * k] = ub2[0][k].x; ub[2 * k + 1] = ub2[0][k].y; lb[2 * k] = lb2[0][k].x; lb[2 * k + 1] = lb2[0][k].y; }
			} else {
				f2 ub2[1][4], lb2[1][4];
#pragma unroll
				for (int k = 0; k < 4; k++) { ub2[0][k] = f2{0.f, 0.f}; lb2[0][k] = f2{0.f, 0.f}; }
				const float co[1] = {coeff};
				if (coeff == 0.f) lean_points<LAYOUT, 1, true>(src, p0, p1, dt, R0, ts, delta, co, ub2, lb2);       // block-uniform: one expansion per workgroup
				else lean_points<LAYOUT, 1, false>(src, p0, p1, dt, R0, ts, delta, co, ub2, lb2);
#pragma unroll
				for (int k = 0; k < 4; k++) { ub[2 * k] = ub2[0][k].x; ub[2 * k + 1] = ub2[0][k].y; lb[2 * k] = lb2[0][k].x; lb[2 * k + 1] = lb2[0][k].y; }
			}
		}
		else
		for (int i = p0 + (int)threadIdx.x; i < p1; i += kBoundsThreads) {
			const float4 p = src[i];
			float m[kGroup];
			sibling_residuals<LAYOUT>(dt, R0, ts, p, coeff * p.w, m);
#pragma unroll
			for (int c = 0; c < kGroup; c++) {
				ub[c] += m[c] * m[c];
				const float dis = fmaxf(m[c] - delta, 0.f);
				lb[c] += dis * dis;
			}
		}
	} else
	for (int i = p0 + (int)threadIdx.x; i < p1; i += kBoundsThreads) {
		const float4 p = src[i];
		// p~ = R p (jly_goicp.cpp:470-476), left-to-right float sums
		float rx = R0.r[0] * p.x + R0.r[1] * p.y + R0.r[2] * p.z;
		float ry = R0.r[3] * p.x + R0.r[4] * p.y + R0.r[5] * p.z;
		float rz = R0.r[6] * p.x + R0.r[7] * p.y + R0.r[8] * p.z;
#pragma unroll
		for (int c = 0; c < kGroup; c++) {
			if (!uniform) {   // wave-uniform branch; generic batches only
				const Rot9 R = rots[cr[c].rot];
				rx = R.r[0] * p.x + R.r[1] * p.y + R.r[2] * p.z;
				ry = R.r[3] * p.x + R.r[4] * p.y + R.r[5] * p.z;
				rz = R.r[6] * p.x + R.r[7] * p.y + R.r[8] * p.z;
			}
			float m = dt_distance<LAYOUT>(dt, rx + cr[c].tx, ry + cr[c].ty, rz + cr[c].tz);
			m = m - cr[c].coeff * p.w;          // rotation uncertainty radius (jly_goicp.cpp:284-285, :159)
			if (m < 0.f) m = 0.f;
			ub[c] += m * m;                      // :302-306
			const float dis = fmaxf(m - cr[c].delta, 0.f);   // :312-314 (adding 0 when dis <= 0 is the same sum)
			lb[c] += dis * dis;
		}
	}

	bounds_item_store(ub, lb, group, chunk, chunks, B, scratch, ub_out, lb_out, red);
}

// ------------------------------------------------------------------------------------------------
// LDS-staged DT tiles (north_star (a)): the bound evaluation for the DEEP expansions of one inner search.
//
// Late in a search the nodes an inner BnB expands are neighbours: the same rotation, translations a few voxels apart.
// For a small patch of the cloud (64 consecutive points of the k-d order) ALL their lookups -- 64 points x 8 siblings x
// up to 64 expansions = 32 768 -- then fall into one box of a few thousand voxels.  That box is copied from the bricked
// grid into LDS once (a few hundred 128-byte lines, coalesced) and the lookups become ds_read_b32.
//
// Mapping: a workgroup = 4 wavefronts = one (search, point chunk); a LANE = one expansion of the search (its two sibling
// translations per axis, delta, coeff live in registers); the 64 points of a sub-patch are dealt 16 to each wavefront and
// broadcast from LDS (rotated once per sub-patch).  No cross-lane reduction at all: a lane owns the 16 sums of its
// expansion over all the points its wavefront sees; the four wavefronts' sums are added in fixed order at the end.
// Every per-point value is the same float expression as in bounds_work (voxel_fast / voxel_exact, the same clamp and
// subtraction order); lookups that fall outside the staged box (or outside the grid) take dt_distance from global memory.
// ------------------------------------------------------------------------------------------------
#ifndef GOICP_TILE_FLOATS
#define GOICP_TILE_FLOATS 8192
#endif
constexpr int kTileFloats = GOICP_TILE_FLOATS;    // 32 KB: up to 128 bricks of 4x4x4 voxels
constexpr int kTilePatch = 64;                    // points per staged box
// split of the cloud for `nseg` tile segments on a grid of `grid` workgroups: as many chunks (multiples of the 64-point
// sub-patch) as it takes to give every workgroup an item, at most one sub-patch per chunk
__host__ __device__ inline void tile_shape(int nseg, int N, int grid, int* chunks, int* chunk_pts)
{
	const int patches = (N + kTilePatch - 1) / kTilePatch;
	int c = nseg > 0 ? (grid + nseg - 1) / nseg : 1;
	if (c < 1) c = 1;
	if (c > patches) c = patches;
	int cp = (patches + c - 1) / c * kTilePatch;
	c = (N + cp - 1) / cp;
	*chunks = c; *chunk_pts = cp;
}

// Round 3: the staged box is kept in LDS as a plain [z][y][x] array (brick-aligned origin and dimensions, so that it is
// still filled brick by brick with coalesced 16-byte loads), and a sub-patch whose box lies wholly inside the grid -- the
// usual case: the grid is the target's bounding cube expanded twice -- takes a path with NO per-lookup checks at all: the
// box contains, by construction, every voxel any lane can index for any of the 64 points, so a lookup is three
// multiply-adds, one three-operand add per child and a ds_read_b32; the eight children are accumulated as four pairs
// (x-siblings) in packed fp32 (v_pk_add / v_pk_mul -- same per-element operations and order as the scalar code:
// bit-identical).  Round 2's form computed brick offsets with a bounds check per axis and child (42 + 16 of its ~180
// vector instructions per point and lane, plus the branches around eight inlined global fallbacks).
__device__ __forceinline__ int add3(int a, int b, int c)
{
	int r;
	__asm__("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}

template <bool QUEUED>
__global__ __launch_bounds__(256) void bounds_tile_kernel(const float4* __restrict__ src, int N, DtDesc dt, const Rot9* __restrict__ rots,
                                                          const ParentRec* __restrict__ parents, const TileSeg* __restrict__ segs, int nseg_host, int chunks_host,
                                                          int chunk_pts_host, const int* __restrict__ d_nseg, int* __restrict__ d_chunks,
                                                          float* __restrict__ scratch, float* __restrict__ ub_out, float* __restrict__ lb_out,
                                                          unsigned* __restrict__ stats)
{
	__shared__ float tile[kTileFloats];
	__shared__ float4 rp[kTilePatch];             // R p of the sub-patch, w = |p|
	static_assert(4 * 64 * 2 * kGroup <= kTileFloats, "the wavefront sums reuse the tile");
	float (*wacc)[64][2 * kGroup] = reinterpret_cast<float (*)[64][2 * kGroup]>(tile);   // after the last sub-patch
	__shared__ float tr[6];                       // translation range of the search's siblings: min xyz, max xyz
	__shared__ int box[8];                        // voxel origin x0 y0 z0, voxel dims DX DY DZ (multiples of 4), staged, wholly inside the grid
	int nseg = nseg_host, chunks = chunks_host, chunk_pts = chunk_pts_host;
	if (QUEUED) {
		// the number of segments is known to the device only (bnb_queue_kernel counted them): the split of the cloud follows from it
		nseg = *d_nseg;
		if (nseg <= 0) return;
		tile_shape(nseg, N, (int)gridDim.x, &chunks, &chunk_pts);
		if (blockIdx.x == 0 && threadIdx.x == 0) *d_chunks = chunks;      // the next round's digest adds the chunk partials up
	}
	for (int item = blockIdx.x; item < nseg * chunks; item += QUEUED ? (int)gridDim.x : nseg * chunks) {
	const int seg = item / chunks, chunk = item - seg * chunks;
	const TileSeg sg = segs[seg];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	// A segment of n <= 32 expansions does not fill a wavefront with one expansion per lane: the lanes are dealt P = 2^ceil(log2 n)
	// expansions x L = 64 / P point subsets instead (lane = sub * P + e; lane group `sub` takes every L-th point of the wavefront's
	// share), and the L partial sums of an expansion are added at the end (xor shuffles, fixed order) -- at least half of the
	// lanes work for any n.
	int P = 1;
	while (P < sg.n) P <<= 1;
	const int L = 64 / P, e_lane = lane & (P - 1), sub = lane / P;
	const bool live = e_lane < sg.n;
	// this lane's expansion: corner + (bit)*w' + w'/2 per axis (load_group's float operations)
	const ParentRec pr = parents[sg.off + (live ? e_lane : 0)];
	const float w = pr.w / 2;
	const float delta = (float)(1.732050808 / 2.0 * (double)w);
	const SiblingSet ts{pr.x + w / 2, pr.x + w + w / 2, pr.y + w / 2, pr.y + w + w / 2, pr.z + w / 2, pr.z + w + w / 2};
	const float coeff = pr.coeff;
	const Rot9 R0 = rots[sg.rot];
	__syncthreads();                              // (queued form: the previous item's sums have been read out of the tile)
	if (wave == 0) {
		float mn[3] = {ts.tx0, ts.ty0, ts.tz0}, mx[3] = {ts.tx1, ts.ty1, ts.tz1};
#pragma unroll
		for (int k = 0; k < 3; k++) {
			if (!live) { mn[k] = INFINITY; mx[k] = -INFINITY; }
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], o, 64)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o, 64)); }
			if (lane == 0) { tr[k] = mn[k]; tr[3 + k] = mx[k]; }
		}
	}
	f2 ub2[4], lb2[4];
#pragma unroll
	for (int k = 0; k < 4; k++) { ub2[k] = f2{0.f, 0.f}; lb2[k] = f2{0.f, 0.f}; }
	const f2 delta2 = f2{delta, delta}, ndelta2 = f2{-delta, -delta};
	const int p0 = chunk * chunk_pts, p1 = p0 + chunk_pts < N ? p0 + chunk_pts : N;
	__syncthreads();
	for (int s0 = p0; s0 < p1; s0 += kTilePatch) {
		// ---- the sub-patch: rotate, find the brick box of everything it can look up, stage it ----
		const int np = p1 - s0 < kTilePatch ? p1 - s0 : kTilePatch;
		if (threadIdx.x < kTilePatch) {
			int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
			if ((int)threadIdx.x < np) {
				const float4 p = src[s0 + threadIdx.x];
				const float rx = R0.r[0] * p.x + R0.r[1] * p.y + R0.r[2] * p.z;
				const float ry = R0.r[3] * p.x + R0.r[4] * p.y + R0.r[5] * p.z;
				const float rz = R0.r[6] * p.x + R0.r[7] * p.y + R0.r[8] * p.z;
				rp[threadIdx.x] = make_float4(rx, ry, rz, p.w);
				const float r3[3] = {rx, ry, rz}, m3[3] = {dt.xmin_f, dt.ymin_f, dt.zmin_f};
#pragma unroll
				for (int k = 0; k < 3; k++) {      // a voxel of margin each way: the box only has to CONTAIN the exact indices
					lo[k] = (int)floorf(((r3[k] + tr[k]) - m3[k]) * dt.scale_f + 0.5f) - 1;
					hi[k] = (int)floorf(((r3[k] + tr[3 + k]) - m3[k]) * dt.scale_f + 0.5f) + 1;
				}
			}
#pragma unroll
			for (int k = 0; k < 3; k++) {
#pragma unroll
				for (int o = 32; o > 0; o >>= 1) { lo[k] = min(lo[k], __shfl_xor(lo[k], o, 64)); hi[k] = max(hi[k], __shfl_xor(hi[k], o, 64)); }
			}
			if (threadIdx.x == 0) {
				// the box in LDS: x brick-aligned (it is filled with 16-byte pieces of brick rows), y and z exact
				bool inside = true;
#pragma unroll
				for (int k = 0; k < 3; k++) inside = inside && lo[k] >= 0 && hi[k] <= dt.V - 1;
				const int x0 = lo[0] & ~3, DX = ((hi[0] | 3) + 1) - x0, DY = hi[1] - lo[1] + 1, DZ = hi[2] - lo[2] + 1;
				const bool ok = inside && (long long)DX * DY * DZ <= kTileFloats;
				box[0] = x0; box[1] = lo[1]; box[2] = lo[2]; box[3] = DX; box[4] = DY; box[5] = DZ;
				box[6] = ok ? 1 : 0; box[7] = ok ? 1 : 0;
				if (stats) atomicAdd(&stats[ok ? 0 : 1], 1u);
			}
		}
		__syncthreads();
		const int x0 = box[0], y0 = box[1], z0 = box[2], DX = box[3], DY = box[4], DZ = box[5];
		const bool fast = box[7] != 0;
		if (fast) {
			// every brick the box touches is read whole (16 consecutive lanes = one 256-byte brick: coalesced), the pieces inside
			// the box are kept
			const int bx0 = x0 >> 2, by0 = y0 >> 2, bz0 = z0 >> 2;
			const int nbx = DX >> 2, nby = ((y0 + DY - 1) >> 2) - by0 + 1, nbz = ((z0 + DZ - 1) >> 2) - bz0 + 1;
			const int n4 = nbx * nby * nbz * 16;
			// brick number -> (bx, by, bz) without integer division (two of them cost ~50 instructions per 16-byte piece): the quotient of b + 0.5 by
			// a count n < 64 in float is exact for b < 2^21 -- it is at least 0.5 / n away from an integer, the rounding errors stay below that
			const float rnbx = 1.0f / (float)nbx, rnby = 1.0f / (float)nby;
			for (int idx = threadIdx.x; idx < n4; idx += 256) {
				const int b = idx >> 4, part = idx & 15;
				const int t2 = (int)(((float)b + 0.5f) * rnbx), bx = b - t2 * nbx;
				const int bz = (int)(((float)t2 + 0.5f) * rnby), by = t2 - bz * nby;
				const int y = (by0 + by) * 4 + (part & 3) - y0, z = (bz0 + bz) * 4 + (part >> 2) - z0;
				const size_t gb = ((size_t)(bz0 + bz) * dt.VB + (by0 + by)) * dt.VB + (bx0 + bx);
				const float4 v = reinterpret_cast<const float4*>(dt.grid)[gb * 16 + part];
				if ((unsigned)y < (unsigned)DY && (unsigned)z < (unsigned)DZ) *reinterpret_cast<float4*>(&tile[(z * DY + y) * DX + bx * 4]) = v;
			}
		}
		__syncthreads();
		// ---- every wavefront: its 16 points x this lane's 8 siblings ----
		if (live) {
			if (fast) {
				// byte address in the tile = 4 (ix - x0) + 4 DX (iy - y0) + 4 DX DY (iz - z0)
				const int sy = 4 * DX, sz = 4 * DX * DY;
				const int cxyz = -4 * x0 - sy * y0 - sz * z0;
				const float eps_box = __fmaf_rn((float)dt.V, dt.c2, dt.c1);
				const char* tb = reinterpret_cast<const char*>(tile);
				// A segment is one search: all its lanes make the same pass.  In the upper-bound pass (coeff == 0, jly_goicp.cpp:284-285 with
				// maxRotDis = 0) the residual is the looked-up distance itself: v - 0 and max(v, 0) are v bit for bit (the DT holds no negative
				// value), so that pass runs a loop without the subtraction and the clamp (12 of ~90 vector instructions per point).
				auto patch_points = [&](auto ub_pass) {
				constexpr bool UB_PASS = decltype(ub_pass)::value;
				for (int j = wave + 4 * sub; j < np; j += 4 * L) {
					const float4 q = rp[j];                               // one address per lane group: a broadcast read
					const float qx[2] = {q.x + ts.tx0, q.x + ts.tx1}, qy[2] = {q.y + ts.ty0, q.y + ts.ty1}, qz[2] = {q.z + ts.tz0, q.z + ts.tz1};
					// six voxel indices: F = (q - min) * scale + 0.5 in float; the float index is provably the reference's unless F is
					// within eps(F) = c1 + c2 |F| of an integer (voxel_fast) -- ONE test on the smallest margin of the six
					float F[6] = {__fmaf_rn(qx[0] - dt.xmin_f, dt.scale_f, 0.5f), __fmaf_rn(qx[1] - dt.xmin_f, dt.scale_f, 0.5f),
					              __fmaf_rn(qy[0] - dt.ymin_f, dt.scale_f, 0.5f), __fmaf_rn(qy[1] - dt.ymin_f, dt.scale_f, 0.5f),
					              __fmaf_rn(qz[0] - dt.zmin_f, dt.scale_f, 0.5f), __fmaf_rn(qz[1] - dt.zmin_f, dt.scale_f, 0.5f)};
					// (the box lies inside the grid, so 0 <= F < V on this path: ONE bound eps(V) >= eps(F) serves all six -- a few more lanes take the
					// exact branch, none takes a wrong index; 16 instead of 29 instructions for the test)
					float margin[6];
#pragma unroll
					for (int k = 0; k < 6; k++) margin[k] = fabsf(F[k] - rintf(F[k]));
					const float worst = fminf(fminf(fminf(margin[0], margin[1]), fminf(margin[2], margin[3])), fminf(margin[4], margin[5]));
					int ix[2] = {(int)F[0], (int)F[1]}, iy[2] = {(int)F[2], (int)F[3]}, iz[2] = {(int)F[4], (int)F[5]};
					if (worst <= eps_box) {
#pragma unroll
						for (int k = 0; k < 2; k++) {
							ix[k] = voxel_exact(qx[k], dt.xmin, dt.scale);
							iy[k] = voxel_exact(qy[k], dt.ymin, dt.scale);
							iz[k] = voxel_exact(qz[k], dt.zmin, dt.scale);
						}
					}
					// the three origin terms ride on the x part; one three-operand add per child (the compiler would share y + z sums: 12 adds for 8)
					const int ax[2] = {4 * ix[0] + cxyz, 4 * ix[1] + cxyz};
					const int ay[2] = {__mul24(iy[0], sy), __mul24(iy[1], sy)};
					const int az[2] = {__mul24(iz[0], sz), __mul24(iz[1], sz)};
					const float rho = coeff * q.w;
					const f2 rho2 = f2{rho, rho};
#pragma unroll
					for (int k = 0; k < 4; k++) {                         // pair k: children 2k (x0) and 2k + 1 (x1) of y[(k & 1)], z[(k >> 1)]
						f2 v = f2{*reinterpret_cast<const float*>(tb + add3(ax[0], ay[k & 1], az[k >> 1])), *reinterpret_cast<const float*>(tb + add3(ax[1], ay[k & 1], az[k >> 1]))};
						f2 m = v;
						if constexpr (!UB_PASS) {
							v = v - rho2;
							m = __builtin_elementwise_max(v, f2{0.f, 0.f});
						}
						ub2[k] = ub2[k] + m * m;
						// max(m - delta, 0) == max(v - delta, 0) bit for bit (delta >= 0: v < 0 gives 0 either way, v >= 0 is m), and the
						// subtraction no longer waits for the clamp (a packed subtract of the packed difference)
						const f2 dis = __builtin_elementwise_max(v + ndelta2, f2{0.f, 0.f});      // x + (-d) is x - d bit for bit
						lb2[k] = lb2[k] + dis * dis;
					}
				}
				};
				if (__builtin_amdgcn_readfirstlane(__float_as_int(coeff)) == 0 && coeff == 0.f) patch_points(std::true_type{});
				else patch_points(std::false_type{});
			} else {
				// the box is too large for the tile, or reaches over the edge of the grid: the direct kernel's sibling path (eight
				// gathers in flight per point, out-of-grid extension included) -- same per-point expressions
				for (int j = wave + 4 * sub; j < np; j += 4 * L) {
					const float4 q = rp[j];
					float m[kGroup];
					sibling_residuals_rotated<1>(dt, q, ts, coeff * q.w, m);
#pragma unroll
					for (int k = 0; k < 4; k++) {
						const f2 m2 = f2{m[2 * k], m[2 * k + 1]};
						ub2[k] = ub2[k] + m2 * m2;
						const f2 dis = __builtin_elementwise_max(m2 - delta2, f2{0.f, 0.f});
						lb2[k] = lb2[k] + dis * dis;
					}
				}
			}
		}
		__syncthreads();                                                  // the box is restaged by the next sub-patch
	}
	// ---- the lane groups' partial sums of an expansion (L > 1), then the four wavefronts' sums, in fixed order ----
	for (int off = P; off < 64; off <<= 1) {
#pragma unroll
		for (int k = 0; k < 4; k++) {
			ub2[k].x += __shfl_xor(ub2[k].x, off, 64); ub2[k].y += __shfl_xor(ub2[k].y, off, 64);
			lb2[k].x += __shfl_xor(lb2[k].x, off, 64); lb2[k].y += __shfl_xor(lb2[k].y, off, 64);
		}
	}
#pragma unroll
	for (int k = 0; k < 4; k++) {
		wacc[wave][lane][2 * k] = ub2[k].x; wacc[wave][lane][2 * k + 1] = ub2[k].y;
		wacc[wave][lane][kGroup + 2 * k] = lb2[k].x; wacc[wave][lane][kGroup + 2 * k + 1] = lb2[k].y;
	}
	__syncthreads();
	for (int idx = threadIdx.x; idx < sg.n * 2 * kGroup; idx += 256) {
		const int e = idx >> 4, k = idx & 15;
		const float s = ((wacc[0][e][k] + wacc[1][e][k]) + wacc[2][e][k]) + wacc[3][e][k];
		const int group = sg.off + e;
		if (chunks == 1) (k < kGroup ? ub_out : lb_out)[group * kGroup + (k & (kGroup - 1))] = s;
		else scratch[((size_t)group * chunks + chunk) * (2 * kGroup) + k] = s;
	}
	}
}

__global__ void bounds_finalize(const float* __restrict__ scratch, int B, int groups, int chunks, float* __restrict__ ub_out, float* __restrict__ lb_out);
// test / measurement entry: nseg searches of n <= 64 expansions each (segment i = parents[i*n .. i*n+n), rotation i)
hipError_t launch_bounds_tile(const float4* src, int N, const DtDesc& dt, const Rot9* rots, const ParentRec* parents, const void* segs, int nseg, int n,
                              int chunks, float* scratch, float* ub, float* lb, unsigned* stats, hipStream_t stream)
{
	if (dt.layout != 1 || n < 1 || n > 64 || nseg < 1 || chunks < 1) return hipErrorInvalidValue;
	int cp = (N + chunks - 1) / chunks;
	cp = (cp + kTilePatch - 1) / kTilePatch * kTilePatch;
	hipLaunchKernelGGL(bounds_tile_kernel<false>, dim3(nseg * chunks), dim3(256), 0, stream, src, N, dt, rots, parents, static_cast<const TileSeg*>(segs), nseg, chunks, cp,
	                   static_cast<const int*>(nullptr), static_cast<int*>(nullptr), scratch, ub, lb, stats);
	if (chunks > 1) {
		const int groups = nseg * n, t = groups * 2 * kGroup;
		hipLaunchKernelGGL(bounds_finalize, dim3((t + 255) / 256), dim3(256), 0, stream, scratch, groups * kGroup, groups, chunks, ub, lb);
	}
	return hipGetLastError();
}

// the lean sibling path pays while the grid is cache-resident (<= the 256 MB Infinity Cache), not when the launch is HBM-bound
static inline bool bounds_lean(const DtDesc& dt) { return (size_t)dt.V * dt.V * dt.V * sizeof(float) <= ((size_t)256 << 20); }
bool bounds_uses_lean(const DtDesc& dt) { return dt.layout == 1 && bounds_lean(dt); }
constexpr int kTileQueueGrid = 256 * (160 * 1024 / (kTileFloats * 4 + 2048));   // as many workgroups per CU as their LDS allows (32 KB tiles: four)
size_t bounds_tile_queue_scratch_floats(int max_groups)
{
	// groups x chunks x 16 with chunks <= ceil(grid / segments) and groups <= 64 segments: at most 64 (grid + segments) rows
	return (size_t)64 * ((size_t)kTileQueueGrid + (size_t)max_groups) * 2 * kGroup;
}
hipError_t launch_bounds_tile_queue(const float4* src, int N, const DtDesc& dt, const Rot9* rots, const QTile& tile, QCtl* ctl, int parity, hipStream_t stream)
{
	if (dt.layout != 1 || N <= 0) return hipErrorInvalidValue;
	hipLaunchKernelGGL(bounds_tile_kernel<true>, dim3(kTileQueueGrid), dim3(256), 0, stream, src, N, dt, rots, tile.parents[parity], tile.segs[parity], 0, 1, 0,
	                   &ctl->n_tile_segs[parity], &ctl->tile_chunks, tile.scratch, tile.ub, tile.lb, static_cast<unsigned*>(nullptr));
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Grouping of an UNRELATED cube batch (the operator API's arbitrary batches; SURVEY 8(d)'s microbench): cubes are bucketed by
// (rotation, pass, 12-bit Morton cell of the translation) with a counting sort, evaluated in that order -- a workgroup's eight
// cubes then share their rotation and lie in one translation cell, and neighbouring workgroups gather from neighbouring DT
// lines -- and the bounds are written back to the caller's order.  Bounds are per cube, so the order changes no bit of them
// (tools/generic_probe.py: 3.75 -> 2.2 ms per 65 536 unrelated cubes on the bunny).  The search never needs this: its expansions
// arrive grouped by search, eight siblings at a time.
// ------------------------------------------------------------------------------------------------
constexpr int kGroupCellBits = 4;                  // per axis: 16 x 16 x 16 translation cells over [-0.5, 0.5]^3 (clamped outside)
__device__ __forceinline__ unsigned cube_bucket(const CubeRec& c, int nrots)
{
	unsigned m = 0;
	const float t[3] = {c.tx, c.ty, c.tz};
#pragma unroll
	for (int k = 0; k < 3; k++) {
		const int q = min(max((int)((t[k] + 0.5f) * (float)(1 << kGroupCellBits)), 0), (1 << kGroupCellBits) - 1);
#pragma unroll
		for (int b = 0; b < kGroupCellBits; b++) m |= (unsigned)((q >> b) & 1) << (3 * b + k);
	}
	const unsigned rot = (unsigned)min(max(c.rot, 0), nrots - 1);
	return ((rot * 2u + (c.coeff > 0.f ? 1u : 0u)) << (3 * kGroupCellBits)) | m;
}
__global__ void cube_hist_kernel(const CubeRec* __restrict__ cubes, int B, int nrots, unsigned* __restrict__ hist)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < B) atomicAdd(&hist[cube_bucket(cubes[i], nrots)], 1u);
}
// exclusive scan of nbins counters, one workgroup of 1024 threads (nbins <= 1024 * 64)
__global__ __launch_bounds__(1024) void cube_scan_kernel(unsigned* __restrict__ hist, int nbins)
{
	__shared__ unsigned wtot[16];
	const int per = (nbins + 1023) / 1024, b0 = threadIdx.x * per;
	unsigned local = 0;
	for (int k = 0; k < per; k++) if (b0 + k < nbins) local += hist[b0 + k];
	unsigned incl = local;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(incl, o, 64); if ((int)(threadIdx.x & 63) >= o) incl += v; }
	if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = incl;
	__syncthreads();
	unsigned before = 0;
	for (int w = 0; w < (int)(threadIdx.x >> 6); w++) before += wtot[w];
	unsigned run = before + incl - local;
	for (int k = 0; k < per; k++) if (b0 + k < nbins) { const unsigned c = hist[b0 + k]; hist[b0 + k] = run; run += c; }
}
__global__ void cube_scatter_kernel(const CubeRec* __restrict__ cubes, int B, int nrots, unsigned* __restrict__ offs, CubeRec* __restrict__ sorted, int* __restrict__ perm)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= B) return;
	const CubeRec c = cubes[i];
	const unsigned pos = atomicAdd(&offs[cube_bucket(c, nrots)], 1u);        // order inside a bucket is arbitrary: the bounds do not depend on it
	sorted[pos] = c;
	perm[pos] = i;
}
__global__ void cube_unpermute_kernel(const float* __restrict__ ub_s, const float* __restrict__ lb_s, const int* __restrict__ perm, int B, float* __restrict__ ub,
                                      float* __restrict__ lb)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= B) return;
	const int o = perm[i];
	ub[o] = ub_s[i]; lb[o] = lb_s[i];
}
size_t bounds_grouped_scratch_bytes(int B, int nrots)
{
	const size_t nbins = (size_t)nrots * 2 << (3 * kGroupCellBits);
	return sizeof(unsigned) * nbins + (sizeof(CubeRec) + sizeof(int) + 2 * sizeof(float)) * (size_t)B + 256;
}
hipError_t launch_bounds_grouped(const float4* src, int N, const DtDesc& dt, const Rot9* rots, int nrots, const CubeRec* cubes, int B, void* group_scratch,
                                 float* scratch, float* ub, float* lb, hipStream_t stream)
{
	if (B <= 0 || N <= 0) return hipSuccess;
	const int nbins = (nrots * 2) << (3 * kGroupCellBits);
	if (nrots < 1 || nbins > 1024 * 64) return hipErrorInvalidValue;
	char* p = static_cast<char*>(group_scratch);
	unsigned* hist = reinterpret_cast<unsigned*>(p); p += sizeof(unsigned) * (size_t)nbins;
	p = reinterpret_cast<char*>(((uintptr_t)p + 63) & ~(uintptr_t)63);
	CubeRec* sorted = reinterpret_cast<CubeRec*>(p); p += sizeof(CubeRec) * (size_t)B;
	int* perm = reinterpret_cast<int*>(p); p += sizeof(int) * (size_t)B;
	float* ub_s = reinterpret_cast<float*>(p); p += sizeof(float) * (size_t)B;
	float* lb_s = reinterpret_cast<float*>(p);
	hipError_t e = hipMemsetAsync(hist, 0, sizeof(unsigned) * (size_t)nbins, stream);
	if (e != hipSuccess) return e;
	const dim3 g((B + 255) / 256), b(256);
	hipLaunchKernelGGL(cube_hist_kernel, g, b, 0, stream, cubes, B, nrots, hist);
	hipLaunchKernelGGL(cube_scan_kernel, dim3(1), dim3(1024), 0, stream, hist, nbins);
	hipLaunchKernelGGL(cube_scatter_kernel, g, b, 0, stream, cubes, B, nrots, hist, sorted, perm);
	e = launch_bounds(src, N, dt, rots, sorted, nullptr, B, scratch, ub_s, lb_s, stream);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(cube_unpermute_kernel, g, b, 0, stream, ub_s, lb_s, perm, B, ub, lb);
	return hipGetLastError();
}

template <int LAYOUT, bool LEAN = false>
__global__ __launch_bounds__(kBoundsThreads) void bounds_kernel(
    const float4* __restrict__ src, int N, DtDesc dt, const Rot9* __restrict__ rots,
    const CubeRec* __restrict__ cubes, const ParentRec* __restrict__ parents, int B, int groups, int chunks, int chunk_pts,
    float* __restrict__ scratch, float* __restrict__ ub_out, float* __restrict__ lb_out)
{
	__shared__ float red[kBoundsThreads / 64][2 * kGroup];
	bounds_work<LAYOUT, LEAN>(blockIdx.x, gridDim.x, src, N, dt, rots, cubes, parents, B, groups, chunks, chunk_pts, scratch, ub_out, lb_out, red);
}

// The same evaluation for a batch whose size only the DEVICE knows (the device-resident BnB queues, bnbqueue.hip):
// *d_groups expansions of 8 children each; a fixed grid walks the (group, chunk) work items, the split of the cloud
// into chunks follows from the count exactly as on the host (bounds_shape).  ub_out / lb_out: [8 * groups] each.
__host__ __device__ inline void bounds_shape(int B, int N, int* groups, int* chunks, int* chunk_pts);
// Work distribution: a launch of exactly-sized grids lets the hardware hand the next block to whichever CU frees up;
// a fixed grid that strides statically over the items loses that (measured on the full bunny: 22.2 ms of bound
// evaluation against 15 ms -- item counts like 3.2 per block leave a quarter of the chip idle in the last pass).
// So the fixed grid draws its items dynamically: eight counters, one per XCD slot (blocks b and b+8 share an XCD), each
// handing out the items whose index is congruent to that slot -- the XCD-aware tiling of bounds_work is kept, and the
// fetch of the next item overlaps the evaluation of the current one.
template <int LAYOUT, bool LEAN = false>
__global__ __launch_bounds__(kBoundsThreads) void bounds_queue_kernel(
    const float4* __restrict__ src, int N, DtDesc dt, const Rot9* __restrict__ rots, const ParentRec* __restrict__ parents,
    const int* __restrict__ d_groups, int* __restrict__ work8, int* __restrict__ d_chunks, float* __restrict__ scratch, float* __restrict__ ub_out,
    float* __restrict__ lb_out, const QSearch* __restrict__ searches, const int* __restrict__ parent_search, QSort qs)
{
	__shared__ float red[kBoundsThreads / 64][2 * kGroup];
	__shared__ int next_item[2];
	__shared__ int twin_sh;
	const TwinCtx twin{LEAN ? searches : nullptr, parent_search, &twin_sh};
	const int ngroups = *d_groups;
	if (ngroups <= 0) return;
	int groups, chunks, chunk_pts;
	const bool sorted = qsort_on(qs, ngroups);
	if (sorted) { groups = ngroups; chunks = qs.chunks; chunk_pts = qs.chunk_pts; }
	else bounds_shape(ngroups * kGroup, N, &groups, &chunks, &chunk_pts);
	if (blockIdx.x == 0 && threadIdx.x == 0) *d_chunks = chunks;          // > 1: the next round's digest adds the chunk partials up
	if (sorted)                                                           // the order is built: leave the histogram zeroed for the next sorted round
		for (int i = blockIdx.x * kBoundsThreads + threadIdx.x; i < kSortBins; i += gridDim.x * kBoundsThreads) qs.hist[i] = 0u;
	const int total = groups * chunks;
	const bool per_xcd = sorted || (total & 7) == 0;
	const int slot = per_xcd ? (int)(blockIdx.x & 7) : 0, stride = per_xcd ? 8 : 1;
	int* ctr = work8 + slot;
	if (threadIdx.x == 0) next_item[0] = atomicAdd(ctr, 1);
	__syncthreads();
	int item = next_item[0], buf = 0;
	while (item * stride + slot < total) {
		if (threadIdx.x == 0) next_item[buf ^ 1] = atomicAdd(ctr, 1);   // in flight while this item is evaluated
		bounds_work<LAYOUT, LEAN>(item * stride + slot, total, src, N, dt, rots, nullptr, parents, ngroups * kGroup, groups, chunks, chunk_pts, scratch, ub_out,
		                    lb_out, red, sorted ? qs.order : nullptr, twin);
		__syncthreads();                                                 // `red` is reused by the next item; next_item is published
		buf ^= 1;
		item = next_item[buf];
	}
}

__global__ void bounds_finalize(const float* __restrict__ scratch, int B, int groups, int chunks,
                                float* __restrict__ ub_out, float* __restrict__ lb_out)
{
	int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= groups * 2 * kGroup) return;
	int group = t / (2 * kGroup), k = t - group * 2 * kGroup;
	int c = group * kGroup + (k & (kGroup - 1));
	if (c >= B) return;
	const float* s = scratch + (size_t)group * chunks * (2 * kGroup) + k;
	float acc = 0.f;
	for (int j = 0; j < chunks; j++) acc += s[(size_t)j * 2 * kGroup];
	(k < kGroup ? ub_out : lb_out)[c] = acc;
}

// ------------------------------------------------------------------------------------------------
// Trimmed cube bounds (trimFraction > 0; jly_goicp.cpp:293-315): only the `inliers` smallest clamped
// residuals of a cube enter the two sums.  One workgroup owns 8 cubes and ALL points; the k-th
// smallest residual of each cube is found exactly by a 3-digit radix select on the float bit
// pattern (11+11+9 bits, integer LDS histograms -> deterministic), recomputing the residuals in
// every pass instead of storing B x N floats; a fourth pass sums the residuals below the threshold
// T, and the `rem` copies of T itself are added once.
// ------------------------------------------------------------------------------------------------
// One workgroup of 1024 threads per expansion: it sweeps the whole cloud four times, and with a single
// workgroup per CU only its own wavefronts can hide the gather latency.
constexpr int kTrimThreads = 1024;
template <int LAYOUT>
__global__ __launch_bounds__(kTrimThreads) void bounds_trim_kernel(
    const float4* __restrict__ src, int N, DtDesc dt, const Rot9* __restrict__ rots,
    const CubeRec* __restrict__ cubes, const ParentRec* __restrict__ parents, int B, int inliers,
    float* __restrict__ ub_out, float* __restrict__ lb_out, const int* __restrict__ d_groups, int* __restrict__ d_chunks = nullptr)
{
	if (d_groups) {                           // batch size known to the device only (device-resident BnB queues)
		if (d_chunks && blockIdx.x == 0 && threadIdx.x == 0) *d_chunks = 1;     // this form writes the final bounds itself
		if ((int)blockIdx.x >= *d_groups) return;
		B = *d_groups * kGroup;
	}
	__shared__ unsigned hist[kGroup][2048];
	__shared__ unsigned sel_prefix[kGroup], sel_rem[kGroup];
	__shared__ float red[kTrimThreads / 64][2 * kGroup];
	const int c0 = blockIdx.x * kGroup;
	CubeRec cr[kGroup];
	load_group(cubes, parents, blockIdx.x, B, cr);
	if (threadIdx.x < kGroup) { sel_prefix[threadIdx.x] = 0u; sel_rem[threadIdx.x] = (unsigned)inliers; }

	// the 8 clamped residuals of one point: the sibling form (one rotation, 6 voxel indices) for an expansion,
	// the per-cube form for anything else; same values either way
	const bool sib = is_sibling_set(cr) && c0 + kGroup <= B;
	const Rot9 R0 = rots[cr[0].rot];
	const SiblingSet ts{cr[0].tx, cr[1].tx, cr[0].ty, cr[2].ty, cr[0].tz, cr[4].tz};
	auto residuals = [&](const float4& p, float m[kGroup]) {
		if (sib) { sibling_residuals<LAYOUT>(dt, R0, ts, p, cr[0].coeff * p.w, m); return; }
#pragma unroll
		for (int c = 0; c < kGroup; c++) {
			const Rot9 R = rots[cr[c].rot];
			const float rx = R.r[0] * p.x + R.r[1] * p.y + R.r[2] * p.z;
			const float ry = R.r[3] * p.x + R.r[4] * p.y + R.r[5] * p.z;
			const float rz = R.r[6] * p.x + R.r[7] * p.y + R.r[8] * p.z;
			float v = dt_distance<LAYOUT>(dt, rx + cr[c].tx, ry + cr[c].ty, rz + cr[c].tz);
			v = v - cr[c].coeff * p.w;
			m[c] = v < 0.f ? 0.f : v;
		}
	};

#pragma unroll 1
	for (int pass = 0; pass < 3; pass++) {
		const int shift = pass == 0 ? 20 : (pass == 1 ? 9 : 0), width = pass == 2 ? 9 : 11, bins = 1 << width;
		for (int i = threadIdx.x; i < kGroup * 2048; i += kTrimThreads) (&hist[0][0])[i] = 0u;
		__syncthreads();
		for (int i = threadIdx.x; i < N; i += kTrimThreads) {
			const float4 p = src[i];
			float m[kGroup];
			residuals(p, m);
#pragma unroll
			for (int c = 0; c < kGroup; c++) {
				const unsigned key = __float_as_uint(m[c]);
				if (pass == 0 || (key >> (shift + width)) == sel_prefix[c])
					atomicAdd(&hist[c][(key >> shift) & (unsigned)(bins - 1)], 1u);
			}
		}
		__syncthreads();
		// 32 threads per cube (the first four wavefronts): locate the bin holding the rem-th smallest of the
		// surviving candidates
		if (threadIdx.x < 32 * kGroup) {
			const int c = threadIdx.x >> 5, j = threadIdx.x & 31, per = bins >> 5;
			unsigned local = 0;
			for (int b = j * per; b < (j + 1) * per; b++) local += hist[c][b];
			unsigned incl = local;
#pragma unroll
			for (int off = 1; off < 32; off <<= 1) {
				const unsigned o = __shfl_up(incl, off, 32);
				if (j >= off) incl += o;
			}
			const unsigned excl = incl - local, rem = sel_rem[c];
			if (excl < rem && rem <= incl) {
				unsigned cum = excl;
				for (int b = j * per; b < (j + 1) * per; b++) {
					const unsigned h = hist[c][b];
					if (cum < rem && rem <= cum + h) {
						sel_prefix[c] = (sel_prefix[c] << width) | (unsigned)b;
						sel_rem[c] = rem - cum;
						break;
					}
					cum += h;
				}
			}
		}
		__syncthreads();
	}

	float ub[kGroup], lb[kGroup];
#pragma unroll
	for (int c = 0; c < kGroup; c++) { ub[c] = 0.f; lb[c] = 0.f; }
	for (int i = threadIdx.x; i < N; i += kTrimThreads) {
		const float4 p = src[i];
		float m[kGroup];
		residuals(p, m);
#pragma unroll
		for (int c = 0; c < kGroup; c++) {
			if (__float_as_uint(m[c]) < sel_prefix[c]) {     // strictly below the k-th smallest
				ub[c] += m[c] * m[c];
				const float dis = fmaxf(m[c] - cr[c].delta, 0.f);
				lb[c] += dis * dis;
			}
		}
	}
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int c = 0; c < kGroup; c++) {
		const float u = wave_sum(ub[c]), l = wave_sum(lb[c]);
		if (lane == 0) { red[wave][c] = u; red[wave][kGroup + c] = l; }
	}
	__syncthreads();
	if (threadIdx.x < 2 * kGroup) {
		float s = red[0][threadIdx.x];
#pragma unroll
		for (int w = 1; w < kTrimThreads / 64; w++) s += red[w][threadIdx.x];
		const int c = threadIdx.x & (kGroup - 1);
		// the residuals equal to the threshold: sel_rem copies of T
		const float T = __uint_as_float(sel_prefix[c]);
		const float n_eq = (float)sel_rem[c];
		if (threadIdx.x < kGroup) s += n_eq * (T * T);
		else {
			const float delta_c = parents ? cr[0].delta : cubes[c0 + c < B ? c0 + c : B - 1].delta;   // read, not cr[c]: no dynamic register indexing
			const float dis = fmaxf(T - delta_c, 0.f);
			s += n_eq * (dis * dis);
		}
		if (c0 + c < B) (threadIdx.x < kGroup ? ub_out : lb_out)[c0 + c] = s;
	}
}

hipError_t launch_bounds_trim(const float4* src, int N, const DtDesc& dt, const Rot9* rots, const CubeRec* cubes, const ParentRec* parents, int B,
                              int inliers, float* ub, float* lb, hipStream_t stream)
{
	if (B <= 0 || N <= 0) return hipSuccess;
	const dim3 grid((B + kGroup - 1) / kGroup), block(kTrimThreads);
	if (dt.layout == 2) return hipErrorInvalidValue;          // the trimmed form is evaluated on the fp32 grid
	if (dt.layout == 0) hipLaunchKernelGGL(bounds_trim_kernel<0>, grid, block, 0, stream, src, N, dt, rots, cubes, parents, B, inliers, ub, lb, (const int*)nullptr, (int*)nullptr);
	else hipLaunchKernelGGL(bounds_trim_kernel<1>, grid, block, 0, stream, src, N, dt, rots, cubes, parents, B, inliers, ub, lb, (const int*)nullptr, (int*)nullptr);
	return hipGetLastError();
}

__host__ __device__ inline void bounds_shape(int B, int N, int* groups, int* chunks, int* chunk_pts)
{
	int g = (B + kGroup - 1) / kGroup;
	// aim for >= 4 blocks of 256 threads per CU (256 CUs) even for small batches, and
	// for a multiple of 8 point chunks (one set per XCD) whenever the cloud is large enough
	const int target_blocks = 1024;   // measured on the BnB's small batches: 512..2048 within 5 %, larger is slower
	const int max_chunks = (N + kBoundsThreads - 1) / kBoundsThreads;
	int c = (target_blocks + g - 1) / g;
	// eight chunk sets only when each of them still gets >= 4 iterations of 256 points: a workgroup's fixed cost (cube records,
	// reduction, partial row) is spread over too few points otherwise (bunny/10, N = 3 038: 8 chunks of 380 points)
	const bool sets = max_chunks >= 32;
	if (sets) c = (c + 7) / 8 * 8;
	if (c > max_chunks) c = sets ? max_chunks / 8 * 8 : max_chunks;
	if (c < 1) c = 1;
	int cp = (N + c - 1) / c;
	cp = (cp + kBoundsThreads - 1) / kBoundsThreads * kBoundsThreads;
	int c2 = (N + cp - 1) / cp;                 // rounding the chunk size up may empty the last chunks
	if (sets && (c & 7) == 0 && c2 != c) { /* keep c: trailing chunks are simply empty */ } else c = c2;
	*groups = g; *chunks = c; *chunk_pts = cp;
}

size_t bounds_scratch_floats(int B, int N, int* groups_out, int* chunks_out)
{
	int g, c, cp;
	bounds_shape(B, N, &g, &c, &cp);
	if (groups_out) *groups_out = g;
	if (chunks_out) *chunks_out = c;
	return (size_t)g * c * 2 * kGroup;
}

size_t bounds_queue_scratch_floats(int max_groups, int sorted_chunks)
{
	// groups x chunks partial rows: chunks <= 8 + 1024/groups (bounds_shape), so groups*chunks <= 8*groups + 4096 (margin included);
	// footprint-ordered rounds cut the cloud into sorted_chunks chunks
	return (size_t)2 * kGroup * ((size_t)std::max(8, sorted_chunks) * (size_t)max_groups + 4096);
}

hipError_t launch_bounds_queue(const float4* src, int N, const DtDesc& dt, const Rot9* rots, const ParentRec* parents, const int* d_groups,
                               int* d_work8, int* d_chunks, int max_groups, int inliers, float* scratch, float* ub, float* lb, hipStream_t stream,
                               const QSearch* searches, const int* parent_search, const QSort* qsort)
{
	if (!parent_search) searches = nullptr;
	QSort qs{};
	if (qsort && dt.layout == 1 && bounds_lean(dt) && inliers >= N) qs = *qsort;
	if (max_groups <= 0 || N <= 0) return hipSuccess;
	if (inliers < N) {
		// trimmed form: one workgroup per expansion, the surplus workgroups of the fixed grid leave at once
		const dim3 grid(max_groups), block(kTrimThreads);
		if (dt.layout == 2) return hipErrorInvalidValue;
		if (dt.layout == 0) hipLaunchKernelGGL(bounds_trim_kernel<0>, grid, block, 0, stream, src, N, dt, rots, (const CubeRec*)nullptr, parents, 0, inliers, ub, lb, d_groups, d_chunks);
		else hipLaunchKernelGGL(bounds_trim_kernel<1>, grid, block, 0, stream, src, N, dt, rots, (const CubeRec*)nullptr, parents, 0, inliers, ub, lb, d_groups, d_chunks);
		return hipGetLastError();
	}
	const dim3 grid(2048), block(kBoundsThreads);                        // 8 workgroups per CU, a multiple of 8 (XCD slots)
	if (dt.layout == 0) hipLaunchKernelGGL(bounds_queue_kernel<0>, grid, block, 0, stream, src, N, dt, rots, parents, d_groups, d_work8, d_chunks, scratch, ub, lb, searches, parent_search, qs);
	else if (dt.layout == 1 && bounds_lean(dt)) hipLaunchKernelGGL((bounds_queue_kernel<1, true>), grid, block, 0, stream, src, N, dt, rots, parents, d_groups, d_work8, d_chunks, scratch, ub, lb, searches, parent_search, qs);
	else if (dt.layout == 1) hipLaunchKernelGGL(bounds_queue_kernel<1>, grid, block, 0, stream, src, N, dt, rots, parents, d_groups, d_work8, d_chunks, scratch, ub, lb, searches, parent_search, qs);
	else hipLaunchKernelGGL(bounds_queue_kernel<2>, grid, block, 0, stream, src, N, dt, rots, parents, d_groups, d_work8, d_chunks, scratch, ub, lb, searches, parent_search, qs);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Footprint-ordered work items (round 3).  In the first rounds of a rotation batch every search expands whole levels of the
// translation tree: hundreds of rotations x all the cubes of a level, i.e. every rotation stamps the cloud all over the grid and
// sweeps the whole distance transform once.  Walked in search order, the items an XCD runs at any time gather from everywhere, its
// 4 MB L2 holds nothing for the next one, and the round runs at the Infinity Cache's bandwidth (a 128-byte line for every few
// 4-byte lookups; 75-110 CU cycles per gather against 37 when the lines are found in L2).  So for large rounds the (expansion, chunk)
// items are bucketed by WHERE their gathers land -- the Morton cell (16 voxels) of R * centroid(chunk) + centre(parent cube) -- with a
// counting sort (key + histogram, scan, scatter), XCD x walks the x-th eighth of the sorted list, and the cloud is cut into smaller
// chunks of 4 096 points (a ~90-voxel patch; the engine picks the size) so that an item's footprint stays compact.  The order changes no bound (an item's sums go to
// its own row); only the chunk count differs from the unsorted launch shape, i.e. the order in which a cube's per-chunk sums are added.
// Measured (tools/round_probe.py, bunny, 230 rotations, both passes): children of level 2 1.22 -> 1.04 ms, of level 3 8.87 -> 6.81 ms.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chunk_centroid_kernel(const float4* __restrict__ src, int N, int chunk_pts, float4* __restrict__ cen)
{
	__shared__ float red[4][3];
	const int p0 = blockIdx.x * chunk_pts, p1 = min(p0 + chunk_pts, N);
	float sx = 0.f, sy = 0.f, sz = 0.f;
	for (int i = p0 + (int)threadIdx.x; i < p1; i += 256) { const float4 p = src[i]; sx += p.x; sy += p.y; sz += p.z; }
	sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
	if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sx; red[threadIdx.x >> 6][1] = sy; red[threadIdx.x >> 6][2] = sz; }
	__syncthreads();
	if (threadIdx.x == 0) {
		const float n = (float)max(p1 - p0, 1);
		cen[blockIdx.x] = make_float4((red[0][0] + red[1][0] + red[2][0] + red[3][0]) / n, (red[0][1] + red[1][1] + red[2][1] + red[3][1]) / n,
		                              (red[0][2] + red[1][2] + red[2][2] + red[3][2]) / n, 0.f);
	}
}
hipError_t launch_chunk_centroids(const float4* src, int N, int chunk_pts, float4* cen, hipStream_t stream)
{
	const int chunks = (N + chunk_pts - 1) / chunk_pts;
	hipLaunchKernelGGL(chunk_centroid_kernel, dim3(chunks), dim3(256), 0, stream, src, N, chunk_pts, cen);
	return hipGetLastError();
}
__global__ void task_key_kernel(const ParentRec* __restrict__ parents, const Rot9* __restrict__ rots, const int* __restrict__ d_groups, QSort qs, DtDesc dt)
{
	const int ngroups = *d_groups;
	if (!qsort_on(qs, ngroups)) return;
	const int T = ngroups * qs.chunks;
	for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
		const int group = t / qs.chunks, chunk = t - group * qs.chunks;
		const ParentRec p = parents[group];
		const Rot9 R = rots[p.rot];
		const float4 c = qs.cen[chunk];
		const float h = 0.5f * p.w;
		const float q[3] = {R.r[0] * c.x + R.r[1] * c.y + R.r[2] * c.z + p.x + h - dt.xmin_f, R.r[3] * c.x + R.r[4] * c.y + R.r[5] * c.z + p.y + h - dt.ymin_f,
		                    R.r[6] * c.x + R.r[7] * c.y + R.r[8] * c.z + p.z + h - dt.zmin_f};
		unsigned key = 0;
#pragma unroll
		for (int k = 0; k < 3; k++) {
			const unsigned v = (unsigned)min(max((int)(q[k] * dt.scale_f), 0), dt.V - 1) >> qs.shift;
#pragma unroll
			for (int b = 0; b < 5; b++) key |= ((v >> b) & 1u) << (3 * b + k);
		}
		qs.keys[t] = key;
		atomicAdd(&qs.hist[key], 1u);
	}
}
__global__ void task_scatter_kernel(const int* __restrict__ d_groups, QSort qs)
{
	const int ngroups = *d_groups;
	if (!qsort_on(qs, ngroups)) return;
	const int T = ngroups * qs.chunks;
	for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x)
		qs.order[atomicAdd(&qs.hist[qs.keys[t]], 1u)] = (unsigned)t;          // order inside a bucket is arbitrary: no bound depends on it
}
// exclusive scan of the 32 768 counters by ONE workgroup: the whole histogram is staged in LDS (128 KB + padding against the 32-way bank
// conflict of "thread t owns bins 32 t .. 32 t + 31"), coalesced both ways
__global__ __launch_bounds__(1024) void task_scan_kernel(const int* __restrict__ d_groups, QSort qs)
{
	constexpr int per = kSortBins / 1024;            // 32 consecutive bins per thread
	__shared__ unsigned lds[kSortBins + kSortBins / per];
	__shared__ unsigned wtot[16];
	if (!qsort_on(qs, *d_groups)) return;
	for (int i = threadIdx.x; i < kSortBins; i += 1024) lds[i + i / per] = qs.hist[i];
	__syncthreads();
	const int b0 = threadIdx.x * (per + 1);
	unsigned local = 0;
#pragma unroll
	for (int k = 0; k < per; k++) local += lds[b0 + k];
	unsigned incl = local;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(incl, o, 64); if ((int)(threadIdx.x & 63) >= o) incl += v; }
	if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = incl;
	__syncthreads();
	unsigned before = 0;
	for (int w = 0; w < (int)(threadIdx.x >> 6); w++) before += wtot[w];
	unsigned run = before + incl - local;
#pragma unroll
	for (int k = 0; k < per; k++) { const unsigned c = lds[b0 + k]; lds[b0 + k] = run; run += c; }
	__syncthreads();
	for (int i = threadIdx.x; i < kSortBins; i += 1024) qs.hist[i] = lds[i + i / per];
}
int qsort_shift(int V) { int s = 0; while (((V - 1) >> s) >= 32) s++; return s; }   // 5 bits per axis: 16-voxel cells at V = 300 and 512
size_t qsort_hist_bytes() { return sizeof(unsigned) * kSortBins; }
// orders the items of the round whose expansions bnb_queue_kernel has just listed (count on the device); a round below qs.min_groups is left alone
hipError_t launch_queue_sort(const ParentRec* parents, const Rot9* rots, const int* d_groups, int max_groups, const QSort& qs, const DtDesc& dt, hipStream_t stream)
{
	if (!qs.order || max_groups < qs.min_groups) return hipSuccess;
	// qs.hist is all zero here: cleared at allocation, and again by the bound evaluation of every round that used it (bounds_queue_kernel)
	const int blocks = (int)std::min<size_t>(((size_t)max_groups * qs.chunks + 255) / 256, 2048);
	hipLaunchKernelGGL(task_key_kernel, dim3(blocks), dim3(256), 0, stream, parents, rots, d_groups, qs, dt);
	hipLaunchKernelGGL(task_scan_kernel, dim3(1), dim3(1024), 0, stream, d_groups, qs);
	hipLaunchKernelGGL(task_scatter_kernel, dim3(blocks), dim3(256), 0, stream, d_groups, qs);
	return hipGetLastError();
}

hipError_t launch_bounds(const float4* src, int N, const DtDesc& dt, const Rot9* rots, const CubeRec* cubes, const ParentRec* parents,
                         int B, float* scratch, float* ub, float* lb, hipStream_t stream)
{
	if (B <= 0 || N <= 0) return hipSuccess;
	int groups, chunks, chunk_pts;
	bounds_shape(B, N, &groups, &chunks, &chunk_pts);
	dim3 grid(groups * chunks), block(kBoundsThreads);
	if (dt.layout == 0)
		hipLaunchKernelGGL(bounds_kernel<0>, grid, block, 0, stream, src, N, dt, rots, cubes, parents, B, groups, chunks, chunk_pts, scratch, ub, lb);
	else if (dt.layout == 1 && bounds_lean(dt))
		hipLaunchKernelGGL((bounds_kernel<1, true>), grid, block, 0, stream, src, N, dt, rots, cubes, parents, B, groups, chunks, chunk_pts, scratch, ub, lb);
	else if (dt.layout == 1)
		hipLaunchKernelGGL(bounds_kernel<1>, grid, block, 0, stream, src, N, dt, rots, cubes, parents, B, groups, chunks, chunk_pts, scratch, ub, lb);
	else
		hipLaunchKernelGGL(bounds_kernel<2>, grid, block, 0, stream, src, N, dt, rots, cubes, parents, B, groups, chunks, chunk_pts, scratch, ub, lb);
	if (chunks > 1) {
		int t = groups * 2 * kGroup;
		hipLaunchKernelGGL(bounds_finalize, dim3((t + 255) / 256), dim3(256), 0, stream, scratch, B, groups, chunks, ub, lb);
	}
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// (b) ICP: exact 1-NN in the flattened k-d tree -- 64-ary, four queries per WAVEFRONT.
// Measured on MI355X, a per-lane walk of a binary tree is hopeless at this size: 30 k queries are
// only 475 wavefronts, every step is a dependent, divergent access, and a wave is as slow as its
// unluckiest lane (1.3 ms per pass, 3.9 ms for queries far from the surface).  So the tree the
// host builds by median splits is flattened into a 64-ary hierarchy of tight bounding boxes, walked
// cooperatively:
//   * a wavefront is four 16-lane rows; a row owns one query, each of its lanes four of the 64
//     child boxes of the current group (six float4 loads per lane), and a leaf's 16 slots map onto
//     the row's 16 lanes;
//   * minima inside a row are DPP butterflies (quad_perm, row_half_mirror, row_mirror) -- no scalar
//     loops, no LDS, no per-lane stack; "any lane of my row" is a slice of a ballot;
//   * a child is one sortable key (distance bits | sub-index); children are TAKEN in ascending key
//     order, so "next key above the best distance" means the group is exhausted -- no pending masks,
//     no re-filtering; the two nearest remaining leaves are scanned per step (two loads in flight,
//     one wait);
//   * the search starts from the upper bound (|q - centre(v)| + DT[v] + 0.9 voxel)^2 read from the
//     distance transform the engine already holds, so queries far from the surface prune as
//     well as near ones; the nearest child of the root is fetched together with that seed;
//   * rows are at different stages of their walks, so one loop iteration runs (at most) the two
//     step kinds "enter a group" and "scan leaves" under the rows' exec masks.
// History (bunny, 30 379 queries, converged pose, per pass): per-lane binary walk 1300 us; one
// wavefront per query with ballot/readlane loops 42 us (instruction-issue bound, ~1250
// wave-instructions per query); this kernel 26 us (~440 per query; what is left is the chain of
// dependent memory round trips of the slowest wavefront).
// The binary depth follows the cloud (full leaves, sparse root group, kdtree.cpp): K = 2 box levels up to
// 65 536 target points, K = 3 up to 4.2 M.
// Exactness: box lower bounds use the same monotone float accumulation as the point distances and
// the boxes are exact, ties go to the lowest original index -> identical to a brute-force scan.
// ------------------------------------------------------------------------------------------------
#ifndef GOICP_ICP_THREADS
#define GOICP_ICP_THREADS 256
#endif
// up to this many source points the pass deals strangers, not neighbours, to a wavefront (icp_pass_kernel).  Measured, iterations/s
// neighbours | strangers: bunny 30 k 26.7 k | 29.0 k; synthetic 40 k 16.4 k | 16.5 k; 100 k 8.6 k | 7.5 k; spanner 150 k 12.6 k | 11.7 k;
// 250 k 4.2 k | 3.3 k; 500 k 2.06 k | 1.45 k
// (kIcpStridedMaxN lives in device.hpp: the engine sizes the accumulators by it)
constexpr int kIcpThreads = GOICP_ICP_THREADS;     // 4 wavefronts = 16 queries per workgroup

// Upper bound on the NN distance from the distance transform.  For ANY voxel v:
//   d(q, NN) <= |q - centre(v)| + DT[v] + sqrt(3)/2 voxel
// (DT[v] is the exact distance from v's centre to the centre of the nearest seed voxel, and a target
// point lies within sqrt(3)/2 voxel of the centre of the voxel it seeded).  v = the grid voxel nearest
// to q (clamped into the grid, so queries outside it are covered too); |q - centre(v)| is computed
// explicitly, so neither the reference's index rounding nor its out-of-grid extension matters here.
// 0.9 voxel + 1e-5 relative cover sqrt(3)/2 = 0.866 and the float rounding of this expression.
template <int LAYOUT>
__device__ __forceinline__ float nn_upper_bound(const DtDesc& dt, float qx, float qy, float qz)
{
	const float inv = 1.0f / dt.scale_f;
	const int V1 = dt.V - 1;
	const int ix = min(max((int)rintf((qx - dt.xmin_f) * dt.scale_f), 0), V1);
	const int iy = min(max((int)rintf((qy - dt.ymin_f) * dt.scale_f), 0), V1);
	const int iz = min(max((int)rintf((qz - dt.zmin_f) * dt.scale_f), 0), V1);
	const float cx = dt.xmin_f + (float)ix * inv, cy = dt.ymin_f + (float)iy * inv, cz = dt.zmin_f + (float)iz * inv;
	const float ex = qx - cx, ey = qy - cy, ez = qz - cz;
	const float d = (__fsqrt_rn(ex * ex + ey * ey + ez * ez) + dt_fetch<LAYOUT>(dt, ix, iy, iz) + 0.9f * inv) * 1.00001f;
	return d * d;
}

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
	return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ unsigned row_min_u32(unsigned v)      // min over the lane's 16-lane row, in every lane
{
	v = min(v, dpp_u32<0xB1>(v));     // quad_perm [1,0,3,2]
	v = min(v, dpp_u32<0x4E>(v));     // quad_perm [2,3,0,1]
	v = min(v, dpp_u32<0x141>(v));    // row_half_mirror
	v = min(v, dpp_u32<0x140>(v));    // row_mirror
	return v;
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row_sum_f32(float v)            // sum over the lane's 16-lane row, in every lane
{
	v += dpp_f32<0xB1>(v);
	v += dpp_f32<0x4E>(v);
	v += dpp_f32<0x141>(v);
	v += dpp_f32<0x140>(v);
	return v;
}

struct Box6x4 { float4 lox, loy, loz, hix, hiy, hiz; };   // the four children 4l..4l+3 of a lane
__device__ __forceinline__ Box6x4 load_child_boxes4(const float* __restrict__ g, int l)
{
	const float4* r = reinterpret_cast<const float4*>(g);
	return Box6x4{r[l], r[16 + l], r[32 + l], r[48 + l], r[64 + l], r[80 + l]};
}
__device__ __forceinline__ float box_lb1(float lox, float loy, float loz, float hix, float hiy, float hiz, float qx, float qy, float qz)
{
	const float ex = fmaxf(fmaxf(lox - qx, qx - hix), 0.f);
	const float ey = fmaxf(fmaxf(loy - qy, qy - hiy), 0.f);
	const float ez = fmaxf(fmaxf(loz - qz, qz - hiz), 0.f);
	float d = ex * ex;
	d += ey * ey;
	d += ez * ez;
	return d;
}
__device__ __forceinline__ void boxes_lb4(const Box6x4& b, float qx, float qy, float qz, float lb[4])
{
	lb[0] = box_lb1(b.lox.x, b.loy.x, b.loz.x, b.hix.x, b.hiy.x, b.hiz.x, qx, qy, qz);
	lb[1] = box_lb1(b.lox.y, b.loy.y, b.loz.y, b.hix.y, b.hiy.y, b.hiz.y, qx, qy, qz);
	lb[2] = box_lb1(b.lox.z, b.loy.z, b.loz.z, b.hix.z, b.hiy.z, b.hiz.z, qx, qy, qz);
	lb[3] = box_lb1(b.lox.w, b.loy.w, b.loz.w, b.hix.w, b.hiy.w, b.hiz.w, qx, qy, qz);
}
// A child is held as one sortable key: the bits of its box distance (>= 0, so they order like the
// float) with the two low mantissa bits replaced by the child's sub-index 0..3; 0xffffffff once taken.
// Pruning with (key & ~3) <= bits(best) is conservative by at most those two bits (a child that could
// just have been skipped is visited): exactness is untouched.
__device__ __forceinline__ void boxes_keys4(const Box6x4& b, float qx, float qy, float qz, unsigned key[4])
{
	float lb[4];
	boxes_lb4(b, qx, qy, qz, lb);
#pragma unroll
	for (int j = 0; j < 4; j++) key[j] = (__float_as_uint(lb[j]) & ~3u) | (unsigned)j;
}
// Take the nearest remaining child of the row: returns its key (uniform in the row; 0xffffffff when
// none is left) and the child 0..63; the owner lane marks it taken.  Because children are taken in
// ascending order, "key above the best distance" means the whole group is exhausted: no pending
// masks, no re-filtering.
__device__ __forceinline__ unsigned row_take(unsigned key[4], int l, int row, int& child)
{
	const unsigned mine = min(min(key[0], key[1]), min(key[2], key[3]));
	const unsigned m = row_min_u32(mine);
	const unsigned owners = (unsigned)(__ballot(mine == m) >> (16 * row)) & 0xffffu;
	const int wl = __ffs((int)owners) - 1;
	const unsigned j = m & 3u;
	if (l == wl) {
		key[0] = j == 0u ? 0xffffffffu : key[0];
		key[1] = j == 1u ? 0xffffffffu : key[1];
		key[2] = j == 2u ? 0xffffffffu : key[2];
		key[3] = j == 3u ? 0xffffffffu : key[3];
	}
	child = 4 * wl + (int)j;
	return m;
}

// Exact 1-NN of the row's query.  `active` rows walk; the others idle through the loop.  On return
// best/idx are uniform in the row and exactly one lane of an active row has `mine` set: the one whose
// leaf slot holds the neighbour (its coordinates and slot ride along in that lane).
//
// TWO: also return best2, a lower bound on the squared distance from the query to EVERY target point other than the
// neighbour (the walk then prunes with the second-best distance; the DT seed takes part as a pseudo-candidate, which can
// only make best2 smaller, i.e. more conservative).  The ICP pass caches (query position, neighbour, sqrt(best2)) per
// source point: on a later pass at position q' the cached point m is still THE nearest neighbour whenever
// |q' - m| + |q' - q_ref| < sqrt(best2_ref) -- every other point is at least sqrt(best2_ref) - |q' - q_ref| away -- so the
// walk is skipped exactly, not approximately.
struct RowNn { float best; int idx; bool mine; float mx, my, mz; int slot; float best2; };

template <int K, int LAYOUT, bool TWO = false, int LEAVES = 2>
__device__ __forceinline__ RowNn rows_nearest(const KdDesc& kd, const DtDesc& dt, const Box6x4& rootb, int l, int row,
                                              float qx, float qy, float qz, bool active)
{
	// no row of this wavefront needs a walk (every query hit the neighbour cache): nothing to fetch, nothing to compute
	if (!__any(active)) return RowNn{INFINITY, INT_MAX, false, 0.f, 0.f, 0.f, 0, INFINITY};
	unsigned key[K][4];
	int node[K];                              // group index at each level; leaves of level K-1 are node*64 + c
	node[0] = 0;
	boxes_keys4(rootb, qx, qy, qz, key[0]);
#pragma unroll
	for (int L = 1; L < K; L++) {
		node[L] = 0;
#pragma unroll
		for (int j = 0; j < 4; j++) key[L][j] = 0xffffffffu;
	}
	// The child of the root with the smallest box distance survives any valid bound (the true
	// neighbour's child has lb <= d_nn), so its boxes are fetched together with the DT seed instead
	// of after it: one dependent memory round trip less.
	Box6x4 fb = rootb;
	if constexpr (K > 1) {
		row_take(key[0], l, row, node[1]);
		fb = load_child_boxes4(kd.boxes[1] + (size_t)node[1] * 384, l);
	}
	RowNn r{0.f, INT_MAX, false, 0.f, 0.f, 0.f, 0, INFINITY};
	if (!TWO && dt.nn_ids) {
		// the walk starts from a REAL candidate: a target point whose seed voxel is nearest to the query's voxel (clamped into
		// the grid) -- nearly always the neighbour itself, so every box beyond the true distance is pruned from the first step
		// (the distance-transform bound below carries up to 1.8 voxels of slack: far from the surface that is a shell with
		// dozens of leaves in it, and the pass lasts as long as its longest walk).  Any target point is a valid candidate, so
		// exactness does not depend on the table; lane 0 of the row holds it until a scan finds a nearer one (or an equally
		// near one with a lower index).
		const int V1 = dt.V - 1;
		const int ix = min(max((int)rintf((qx - dt.xmin_f) * dt.scale_f), 0), V1);
		const int iy = min(max((int)rintf((qy - dt.ymin_f) * dt.scale_f), 0), V1);
		const int iz = min(max((int)rintf((qz - dt.zmin_f) * dt.scale_f), 0), V1);
		size_t off;
		if (LAYOUT == 0) off = ((size_t)iz * dt.V + iy) * dt.V + ix;
		else off = (((size_t)(iz >> 2) * dt.VB + (iy >> 2)) * dt.VB + (ix >> 2)) * 64 + (((iz & 3) << 4) | ((iy & 3) << 2) | (ix & 3));
		const int slot = dt.nn_ids[off];
		const float4 pt = kd.pts[slot];
		const float d0 = qx - pt.x, d1 = qy - pt.y, d2 = qz - pt.z;
		float e = d0 * d0;                                                 // the leaf scan's accumulation order: the same bits
		e += d1 * d1;
		e += d2 * d2;
		r.best = e; r.idx = __float_as_int(pt.w); r.mine = l == 0; r.mx = pt.x; r.my = pt.y; r.mz = pt.z; r.slot = slot;
	} else
		r.best = nn_upper_bound<LAYOUT>(dt, qx, qy, qz);
	unsigned bbits = __float_as_uint(r.best);
	unsigned b2bits = 0x7f800000u;                // +inf: nothing but the neighbour seen yet
	int d = 0;
	if constexpr (K > 1) {
		boxes_keys4(fb, qx, qy, qz, key[1]);
		d = 1;
	}
	bool done = !active;
	while (__any(!done)) {
		// One step per row and iteration: leave exhausted levels (cascading through the unrolled
		// blocks below), then either enter the nearest child group or scan the two nearest leaves.
		bool acted = done;
#pragma unroll
		for (int L = K - 1; L >= 0; L--) {
			if (!acted && d == L) {
				int c;
				const unsigned m = row_take(key[L], l, row, c);
				if ((m & ~3u) > (TWO ? b2bits : bbits)) {            // nothing left within the best (TWO: second-best) distance
					if (L == 0) { done = true; acted = true; }
					else d = L - 1;
				} else if (L == K - 1 && LEAVES == 4 && !TWO) {
					// ---- scan the FOUR nearest remaining leaves of the group: four loads in flight, one wait.  A pass at
					// bunny scale lasts as long as its longest walk (a handful of far-from-surface queries with dozens of
					// near-equidistant leaves; every step is a dependent round trip), so halving the steps of a walk is worth
					// the extra row_takes; leaves beyond the bound re-read the first one and are ignored ----
					const int s0 = (node[L] * 64 + c) * kLeafSlots + l;
					int sl[4] = {s0, s0, s0, s0};
					bool use[4] = {true, false, false, false};
#pragma unroll
					for (int j = 1; j < 4; j++) {
						int cj;
						const unsigned mj = row_take(key[L], l, row, cj);
						use[j] = (mj & ~3u) <= bbits;
						if (use[j]) sl[j] = (node[L] * 64 + cj) * kLeafSlots + l;
					}
					float4 pq[4];
#pragma unroll
					for (int j = 0; j < 4; j++) pq[j] = kd.pts[sl[j]];
					float4 pt = pq[0];
					int sb = sl[0];
					float e;
					{
						const float d0 = qx - pt.x, d1 = qy - pt.y, d2 = qz - pt.z;
						e = d0 * d0;                                     // L2_Simple_Adaptor accumulation order
						e += d1 * d1;
						e += d2 * d2;
					}
#pragma unroll
					for (int j = 1; j < 4; j++) {
						const float d0 = qx - pq[j].x, d1 = qy - pq[j].y, d2 = qz - pq[j].z;
						float dj = d0 * d0;
						dj += d1 * d1;
						dj += d2 * d2;
						const bool wins = use[j] && (dj < e || (dj == e && __float_as_int(pq[j].w) < __float_as_int(pt.w)));
						if (wins) { pt = pq[j]; e = dj; sb = sl[j]; }
					}
					const unsigned db = __float_as_uint(e);
					const unsigned dmin = row_min_u32(db);
					const unsigned id = (unsigned)__float_as_int(pt.w);
					const unsigned idmin = row_min_u32(db == dmin ? id : 0x7fffffffu);   // ties -> lowest original index
					if (dmin < bbits || (dmin == bbits && (int)idmin < r.idx)) {
						bbits = dmin;
						r.idx = (int)idmin;
						r.mine = db == dmin && id == idmin;
						r.mx = pt.x; r.my = pt.y; r.mz = pt.z; r.slot = sb;
					}
					acted = true;
				} else if (L == K - 1) {
					// ---- scan the two nearest remaining leaves of the group: one slot per lane each ----
					int c2;
					const unsigned m2 = row_take(key[L], l, row, c2);
					const int sa = (node[L] * 64 + c) * kLeafSlots + l;
					const bool dup = (m2 & ~3u) > (TWO ? b2bits : bbits);   // no second leaf worth scanning: the first one twice
					int sb = (node[L] * 64 + (dup ? c : c2)) * kLeafSlots + l;
					const float4 pa = kd.pts[sa];
					float4 pt = kd.pts[sb];
					const float d0 = qx - pa.x, d1 = qy - pa.y, d2 = qz - pa.z;
					float da = d0 * d0;                              // L2_Simple_Adaptor accumulation order
					da += d1 * d1;
					da += d2 * d2;
					const float e0 = qx - pt.x, e1 = qy - pt.y, e2 = qz - pt.z;
					float e = e0 * e0;
					e += e1 * e1;
					e += e2 * e2;
					const bool a_wins = da < e || (da == e && __float_as_int(pa.w) < __float_as_int(pt.w));
					const float other = dup ? INFINITY : (a_wins ? e : da);   // this lane's second candidate (TWO)
					if (a_wins) { pt = pa; e = da; sb = sa; }
					const unsigned db = __float_as_uint(e);
					const unsigned dmin = row_min_u32(db);
					const unsigned id = (unsigned)__float_as_int(pt.w);
					const unsigned idmin = row_min_u32(db == dmin ? id : 0x7fffffffu);   // ties -> lowest original index
					if constexpr (TWO) {
						// the two smallest of {best, best2, this scan's candidates}: the scan's runner-up is the row minimum with
						// the winning lane represented by its other candidate
						const unsigned second = row_min_u32((db == dmin && id == idmin) ? __float_as_uint(other) : db);
						b2bits = min(max(bbits, dmin), min(b2bits, second));
					}
					if (dmin < bbits || (dmin == bbits && (int)idmin < r.idx)) {
						bbits = dmin;
						r.idx = (int)idmin;
						r.mine = db == dmin && id == idmin;
						r.mx = pt.x; r.my = pt.y; r.mz = pt.z; r.slot = sb;
					}
					acted = true;
				} else {
					// ---- enter the nearest remaining child group one level down ----
					const int NL = L + 1 < K ? L + 1 : L;
					node[NL] = node[L] * 64 + c;
					const Box6x4 cb = load_child_boxes4(kd.boxes[NL] + (size_t)node[NL] * 384, l);
					boxes_keys4(cb, qx, qy, qz, key[NL]);
					d = NL;
					acted = true;
				}
			}
		}
	}
	r.best = __uint_as_float(bbits);
	r.best2 = __uint_as_float(b2bits);
	return r;
}

// Workgroup-shared state of one ICP iteration: the per-wavefront sums of the pass and the finalize's scratch, in ONE
// __shared__ object (a second one beside it can make the compiler drain the memory pipeline before LDS reads).
constexpr int kFinThreads = 1024;                  // the stand-alone finalize: 256 row streams x four float4 columns
struct FinScratch { double wsum[kFinThreads / 64][kIcpAcc]; double sums[kIcpAcc]; float red[kIcpThreads / 16][kIcpAcc]; int last; IcpState st; };   // red: one row of sums per 16-lane row
template <int T> __device__ __forceinline__ void finalize_reduce(const float* __restrict__ partials, int nblocks, FinScratch& sh);
__device__ __forceinline__ void finalize_rows(const double* __restrict__ sums, IcpState* __restrict__ state, const IcpState& rd, int lane);

// FUSED: the workgroup that arrives last (one ticket per launch; agent-scope release before the ticket, acquire
// after it -- cdna_hip_programming.md Guideline 16, counter form) also runs the finalize, so an ICP iteration is ONE
// launch.  `ticket` is zero between launches: zeroed when the engine is created, reset by the last arriver.
// nn_cache (two float4 per source point: {q_ref.xyz, sqrt(best2_ref)}, {neighbour xyz, its original index}) != nullptr:
// the exact skip test of rows_nearest<TWO>'s comment; entries never go stale (they are statements about the static
// target cloud), an entry with sqrt(best2_ref) = 0 (fresh engine, tied neighbours) always walks.
template <int K, int LAYOUT, bool FUSED, bool CACHE, int LEAVES = 2, bool STRIDED = false, bool ACC = false>
__global__ __launch_bounds__(kIcpThreads, 2048 / kIcpThreads) void icp_pass_kernel(const float4* __restrict__ src, int N,
                                                                  IcpState* __restrict__ st, KdDesc kd, DtDesc dt,
                                                                  float* __restrict__ partials, int* __restrict__ ticket,
                                                                  float4* __restrict__ nn_cache, int* __restrict__ hit_counter)
{
	__shared__ FinScratch sh;
	float (*red)[kIcpAcc] = sh.red;                                   // [16 rows of the workgroup][16 sums]
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane >> 4, l = lane & 15;
	// Which four queries a wavefront walks.  Neighbours in the k-d order (large clouds: their walks touch the same boxes
	// and leaves, and at 1 M points the pass is throughput-bound: 926 vs 583 iterations/s) -- or, STRIDED, four queries a
	// quarter of the cloud apart (small clouds: the pass lasts as long as its slowest wavefront; far-from-surface queries
	// come in neighbourhoods, so with neighbours in one wavefront all four rows walk long and in different stages, with
	// strangers the one long walk of a wavefront runs alone: bunny 27.3 k -> 29.7 k iterations/s)
	const int nw = (N + 3) >> 2, wv = blockIdx.x * (kIcpThreads / 64) + wave;
	const int i = STRIDED ? wv + row * nw : wv * 4 + row;
	const bool valid = wv < nw && i < N;
	const Box6x4 rootb = load_child_boxes4(kd.boxes[0], l);      // issued before the flag is tested: one round trip less
	const int ic = valid ? i : N - 1;
	const float4 p = src[ic];
	float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
	if constexpr (CACHE) { c0 = nn_cache[2 * (size_t)ic]; c1 = nn_cache[2 * (size_t)ic + 1]; }   // same round trip as p
	if (st->converged) return;                                   // loop already finished: queued launches drain
	// jly_icp3d.hpp:222-224, left-to-right float sums
	const float qx = st->R[0] * p.x + st->R[1] * p.y + st->R[2] * p.z + st->t[0];
	const float qy = st->R[3] * p.x + st->R[4] * p.y + st->R[5] * p.z + st->t[1];
	const float qz = st->R[6] * p.x + st->R[7] * p.y + st->R[8] * p.z + st->t[2];
	RowNn r;
	if constexpr (CACHE) {
		// exact skip test: the cached point is THE neighbour at q if |q - m| + |q - q_ref| < sqrt(best2_ref); a relative
		// 1e-5 and an absolute 1e-7 cover the rounding of the three square roots and sums (a failed test only costs a walk)
		const float e0 = qx - c1.x, e1 = qy - c1.y, e2 = qz - c1.z;
		float e = e0 * e0;                                       // the leaf scan's accumulation order: the same bits
		e += e1 * e1;
		e += e2 * e2;
		const float g0 = qx - c0.x, g1 = qy - c0.y, g2 = qz - c0.z;
		const float moved = __fsqrt_rn(g0 * g0 + g1 * g1 + g2 * g2);
		const bool hit = valid && (__fsqrt_rn(e) + moved) * 1.00001f + 1e-7f < c0.w;
		r = rows_nearest<K, LAYOUT, true>(kd, dt, rootb, l, row, qx, qy, qz, valid && !hit);
		if (hit_counter && hit && l == 0) atomicAdd(hit_counter, 1);      // diagnostics only (goicp_debug_cache_hits)
		if (hit) {
			r.best = e; r.idx = __float_as_int(c1.w); r.mine = l == 0; r.mx = c1.x; r.my = c1.y; r.mz = c1.z;
		} else if (valid && r.mine) {
			// the walk's result becomes the entry: position, neighbour, and what everything else is at least away
			nn_cache[2 * (size_t)i] = make_float4(qx, qy, qz, __fsqrt_rn(r.best2) * 0.99999f);
			nn_cache[2 * (size_t)i + 1] = make_float4(r.mx, r.my, r.mz, __int_as_float(r.idx));
		}
	} else {
		r = rows_nearest<K, LAYOUT, false, LEAVES>(kd, dt, rootb, l, row, qx, qy, qz, valid);
	}
	float acc[kIcpAcc];
#pragma unroll
	for (int k = 0; k < kIcpAcc; k++) acc[k] = 0.f;
	if (valid && r.mine) {                                       // exactly one lane of the row
		const float ax = qx - st->cq[0], ay = qy - st->cq[1], az = qz - st->cq[2];   // pivots keep the covariance sums well conditioned
		const float bx = r.mx - st->cm[0], by = r.my - st->cm[1], bz = r.mz - st->cm[2];
		acc[0] = ax; acc[1] = ay; acc[2] = az;
		acc[3] = bx; acc[4] = by; acc[5] = bz;
		acc[6] = ax * bx; acc[7] = ax * by; acc[8] = ax * bz;
		acc[9] = ay * bx; acc[10] = ay * by; acc[11] = ay * bz;
		acc[12] = az * bx; acc[13] = az * by; acc[14] = az * bz;
		acc[15] = r.best;
	}
	// Exactly one lane of a row holds its correspondence (the others hold zeros), so there is nothing to reduce inside a
	// row: that lane stores its 16 terms to LDS (four 16-byte stores), rows without a query store zeros, and 16 threads add
	// the workgroup's 16 rows in fixed order.  (The previous form ran 16 DPP row sums + 32 cross-row shuffles per wavefront:
	// 160 of the ~1 180 VALU instructions of a wavefront, on a kernel the counters show to be VALU-issue-limited.)
	{
		const int wrow = wave * 4 + row;
		const bool owner = valid ? r.mine : l == 0;
		if (owner) {
			float4* dst = reinterpret_cast<float4*>(red[wrow]);
			dst[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
			dst[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
			dst[2] = make_float4(acc[8], acc[9], acc[10], acc[11]);
			dst[3] = make_float4(acc[12], acc[13], acc[14], acc[15]);
		}
	}
	__syncthreads();
	if (threadIdx.x < kIcpAcc) {
		float sum = red[0][threadIdx.x];
#pragma unroll
		for (int x = 1; x < kIcpThreads / 16; x++) sum += red[x][threadIdx.x];
		if constexpr (FUSED) {
			// publish the row WRITE-THROUGH (agent-scope relaxed 8-byte stores = sc1): no release fence, i.e. no L2
			// write-back per workgroup (measured: with a release fence in each of the 1 899 workgroups the pass took 61 us
			// instead of 27)
			const float nb = __shfl_down(sum, 1, 64);
			if ((threadIdx.x & 1) == 0) {
				const unsigned long long pk = (unsigned long long)__float_as_uint(sum) | ((unsigned long long)__float_as_uint(nb) << 32);
				__hip_atomic_store(reinterpret_cast<unsigned long long*>(partials + (size_t)blockIdx.x * kIcpAcc + threadIdx.x), pk, __ATOMIC_RELAXED,
				                   __HIP_MEMORY_SCOPE_AGENT);
			}
		} else if constexpr (ACC) {
			// no row of partial sums per workgroup (on the bunny the finalize spent 4.2 of its 9.4 us fetching and adding
			// 1 899 of them) but 16 device-scope 64-bit integer adds of the sums in fixed point -- integer addition is
			// associative, so the totals are exact and independent of the arrival order: still bit-reproducible
			const long long v = __double2ll_rn((double)sum * (double)st->acc_scale);
			unsigned long long* a = reinterpret_cast<unsigned long long*>(partials) + (size_t)(blockIdx.x & (kIcpAccReplicas - 1)) * kIcpAcc + threadIdx.x;
			__hip_atomic_fetch_add(a, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		} else {
			partials[(size_t)blockIdx.x * kIcpAcc + threadIdx.x] = sum;
		}
	}
	if constexpr (FUSED) {
		// the storing wave drains its stores, workgroup barrier, then ONE lane draws the ticket
		__asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (threadIdx.x == 0) {
			const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const int last = t == (int)gridDim.x - 1;
			if (last) {
				__hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // next launch starts from zero
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
				__asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
			}
			sh.last = last;
		}
		__syncthreads();
		if (!sh.last) return;
		finalize_reduce<kIcpThreads>(partials, (int)gridDim.x, sh);
		if (threadIdx.x < 64) finalize_rows(sh.sums, st, *st, (int)threadIdx.x);
	}
}

// ---- trimmed ICP (trim_fraction > 0; jly_icp3d.hpp:236-252): NN for every point, exact selection of
// the `num` smallest squared distances (radix select, ties in point order), sums over the selected ----
template <int K, int LAYOUT>
__global__ __launch_bounds__(kIcpThreads) void icp_nn_kernel(const float4* __restrict__ src, int N,
                                                             const IcpState* __restrict__ st, KdDesc kd, DtDesc dt,
                                                             float* __restrict__ nn_d2, int* __restrict__ nn_slot)
{
	const int lane = threadIdx.x & 63, row = lane >> 4, l = lane & 15;
	const int i = (blockIdx.x * (kIcpThreads / 64) + (threadIdx.x >> 6)) * 4 + row;
	const bool valid = i < N;
	const Box6x4 rootb = load_child_boxes4(kd.boxes[0], l);
	const float4 p = src[valid ? i : N - 1];
	if (st->converged) return;
	const float qx = st->R[0] * p.x + st->R[1] * p.y + st->R[2] * p.z + st->t[0];
	const float qy = st->R[3] * p.x + st->R[4] * p.y + st->R[5] * p.z + st->t[1];
	const float qz = st->R[6] * p.x + st->R[7] * p.y + st->R[8] * p.z + st->t[2];
	const RowNn r = rows_nearest<K, LAYOUT, false, 4>(kd, dt, rootb, l, row, qx, qy, qz, valid);
	if (valid && r.mine) { nn_d2[i] = r.best; nn_slot[i] = r.slot; }
}

// one workgroup: key of the num-th smallest d2 (3-digit radix select), then inclusion flags with the
// ties taken in point order (deterministic)
__global__ __launch_bounds__(1024) void icp_select_stream_kernel(const float* __restrict__ nn_d2, int N, int num,
                                                          const IcpState* __restrict__ st, unsigned char* __restrict__ include)
{
	if (st->converged) return;
	__shared__ unsigned hist[2048];
	__shared__ unsigned sel_prefix, sel_rem, wave_cnt[16], wave_tot[16];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (threadIdx.x == 0) { sel_prefix = 0u; sel_rem = (unsigned)num; }
#pragma unroll 1
	for (int pass = 0; pass < 3; pass++) {
		const int shift = pass == 0 ? 20 : (pass == 1 ? 9 : 0), width = pass == 2 ? 9 : 11, bins = 1 << width;
		for (int i = threadIdx.x; i < 2048; i += 1024) hist[i] = 0u;
		__syncthreads();
		const unsigned prefix = sel_prefix;
		for (int base = 0; base < N; base += 1024) {
			const int i = base + threadIdx.x;
			const unsigned key = i < N ? __float_as_uint(nn_d2[i]) : 0u;
			bool live = i < N && (pass == 0 || (key >> (shift + width)) == prefix);
			const unsigned bin = (key >> shift) & (unsigned)(bins - 1);
			if (live) atomicAdd(&hist[bin], 1u);      // same-address lanes serialise inside LDS (<= 64 cycles per instruction): cheap from registers
		}
		__syncthreads();
		// the bin holding the rem-th smallest candidate: two bins per thread, wavefront scan, 16 wavefront totals
		{
			const unsigned h0 = hist[2 * threadIdx.x], h1 = hist[2 * threadIdx.x + 1];   // bins beyond `bins` are zero
			const unsigned local = h0 + h1;
			unsigned incl = local;
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) {
				const unsigned o = __shfl_up(incl, off, 64);
				if (lane >= off) incl += o;
			}
			if (lane == 63) wave_tot[wave] = incl;
			__syncthreads();
			unsigned before = 0;
			for (int w2 = 0; w2 < wave; w2++) before += wave_tot[w2];
			const unsigned excl = before + incl - local, rem = sel_rem;
			__syncthreads();                                  // every thread has read sel_rem before one rewrites it
			if (excl < rem && rem <= excl + local) {              // exactly one thread
				const bool first = rem <= excl + h0;
				sel_prefix = (prefix << width) | (unsigned)(2 * threadIdx.x + (first ? 0 : 1));
				sel_rem = rem - (first ? excl : excl + h0);
			}
		}
		__syncthreads();
	}
	// inclusion flags; the sel_rem copies of the threshold itself go to the ties that come first in point
	// order: every thread owns a contiguous run of points, counts its ties, one block scan orders them
	const unsigned T = sel_prefix, rem = sel_rem;
	const int per = (N + 1023) / 1024, lo = min((int)threadIdx.x * per, N), hi = min(lo + per, N);
	unsigned mine = 0;
	for (int i = lo; i < hi; i++) mine += __float_as_uint(nn_d2[i]) == T ? 1u : 0u;
	unsigned incl = mine;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const unsigned o = __shfl_up(incl, off, 64);
		if (lane >= off) incl += o;
	}
	if (lane == 63) wave_cnt[wave] = incl;
	__syncthreads();
	unsigned before = incl - mine;
	for (int w2 = 0; w2 < wave; w2++) before += wave_cnt[w2];
	for (int i = lo; i < hi; i++) {
		const unsigned key = __float_as_uint(nn_d2[i]);
		const bool tie = key == T;
		include[i] = (key < T || (tie && before < rem)) ? 1 : 0;
		before += tie ? 1u : 0u;
	}
}

// The same selection for N <= 32 768 with the distances held in registers (32 per thread, one burst of
// loads): a single workgroup streaming them from memory three times is a chain of ~90 dependent loads
// (160 us at bunny size); from registers the three digit passes touch only LDS.
constexpr int kSelPer = 32;
__global__ __launch_bounds__(1024) void icp_select_kernel(const float* __restrict__ nn_d2, int N, int num,
                                                          const IcpState* __restrict__ st, unsigned char* __restrict__ include)
{
	__shared__ unsigned hist[2048];
	__shared__ unsigned sel_prefix, sel_rem, wave_tot[16];
	__shared__ unsigned short tie_cnt[kSelPer][16];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	unsigned key[kSelPer];
#pragma unroll
	for (int j = 0; j < kSelPer; j++) {
		const int i = j * 1024 + (int)threadIdx.x;
		key[j] = i < N ? __float_as_uint(nn_d2[i]) : 0xffffffffu;      // padding sorts last (distances are >= 0 and finite)
	}
	if (st->converged) return;
	if (threadIdx.x == 0) { sel_prefix = 0u; sel_rem = (unsigned)num; }
#pragma unroll 1
	for (int pass = 0; pass < 3; pass++) {
		const int shift = pass == 0 ? 20 : (pass == 1 ? 9 : 0), width = pass == 2 ? 9 : 11, bins = 1 << width;
		for (int i = threadIdx.x; i < 2048; i += 1024) hist[i] = 0u;
		__syncthreads();
		const unsigned prefix = sel_prefix;
#pragma unroll
		for (int j = 0; j < kSelPer; j++) {
			const bool live = j * 1024 + (int)threadIdx.x < N && (pass == 0 || (key[j] >> (shift + width)) == prefix);
			const unsigned bin = (key[j] >> shift) & (unsigned)(bins - 1);
			if (live) atomicAdd(&hist[bin], 1u);      // same-address lanes serialise inside LDS (<= 64 cycles per instruction): cheap from registers
		}
		__syncthreads();
		// the bin holding the rem-th smallest candidate: two bins per thread, wavefront scan, 16 wavefront totals
		const unsigned h0 = hist[2 * threadIdx.x], h1 = hist[2 * threadIdx.x + 1];       // bins beyond `bins` are zero
		const unsigned local = h0 + h1;
		unsigned incl = local;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const unsigned o = __shfl_up(incl, off, 64);
			if (lane >= off) incl += o;
		}
		if (lane == 63) wave_tot[wave] = incl;
		__syncthreads();
		unsigned before = 0;
		for (int w2 = 0; w2 < wave; w2++) before += wave_tot[w2];
		const unsigned excl = before + incl - local, rem = sel_rem;
		__syncthreads();                                      // every thread has read sel_rem before one rewrites it
		if (excl < rem && rem <= excl + local) {                  // exactly one thread
			const bool first = rem <= excl + h0;
			sel_prefix = (prefix << width) | (unsigned)(2 * threadIdx.x + (first ? 0 : 1));
			sel_rem = rem - (first ? excl : excl + h0);
		}
		__syncthreads();
	}
	// inclusion flags; the sel_rem copies of the threshold itself go to the ties that come first in point order
	// (i = j*1024 + thread): per (j, wavefront) tie counts, one barrier, then every lane ranks its own ties
	const unsigned T = sel_prefix, rem = sel_rem;
#pragma unroll
	for (int j = 0; j < kSelPer; j++) {
		const unsigned long long tb = __ballot(key[j] == T);
		if (lane == 0) tie_cnt[j][wave] = (unsigned short)__popcll(tb);
	}
	__syncthreads();
	// lane j gathers, for iteration j, the ties of the earlier iterations and of the earlier wavefronts
	unsigned wbefore = 0, total = 0;
	if (lane < kSelPer)
		for (int w2 = 0; w2 < 16; w2++) {
			const unsigned c = tie_cnt[lane][w2];
			total += c;
			wbefore += w2 < wave ? c : 0u;
		}
	unsigned incl = total;
#pragma unroll
	for (int off = 1; off < kSelPer; off <<= 1) {
		const unsigned o = __shfl_up(incl, off, 64);
		if (lane >= off) incl += o;
	}
	const unsigned base = incl - total + wbefore;
#pragma unroll
	for (int j = 0; j < kSelPer; j++) {
		const int i = j * 1024 + (int)threadIdx.x;
		const bool tie = key[j] == T;
		const unsigned long long tb = __ballot(tie);
		const unsigned rank = (unsigned)__builtin_amdgcn_readlane((int)base, j) + (unsigned)__popcll(tb & ((1ull << lane) - 1ull));
		if (i < N) include[i] = (key[j] < T || (tie && rank < rem)) ? 1 : 0;
	}
}

__global__ __launch_bounds__(kIcpThreads) void icp_accum_kernel(const float4* __restrict__ src, int N,
                                                                const IcpState* __restrict__ st, KdDesc kd,
                                                                const float* __restrict__ nn_d2, const int* __restrict__ nn_slot,
                                                                const unsigned char* __restrict__ include, float* __restrict__ partials)
{
	__shared__ float red[kIcpThreads / 64][kIcpAcc];
	if (st->converged) return;
	const int i = blockIdx.x * kIcpThreads + threadIdx.x;
	float acc[kIcpAcc];
#pragma unroll
	for (int k = 0; k < kIcpAcc; k++) acc[k] = 0.f;
	if (i < N && include[i]) {
		const float4 p = src[i];
		const float qx = st->R[0] * p.x + st->R[1] * p.y + st->R[2] * p.z + st->t[0];
		const float qy = st->R[3] * p.x + st->R[4] * p.y + st->R[5] * p.z + st->t[1];
		const float qz = st->R[6] * p.x + st->R[7] * p.y + st->R[8] * p.z + st->t[2];
		const float4 m = kd.pts[nn_slot[i]];
		const float ax = qx - st->cq[0], ay = qy - st->cq[1], az = qz - st->cq[2];
		const float bx = m.x - st->cm[0], by = m.y - st->cm[1], bz = m.z - st->cm[2];
		acc[0] = ax; acc[1] = ay; acc[2] = az;
		acc[3] = bx; acc[4] = by; acc[5] = bz;
		acc[6] = ax * bx; acc[7] = ax * by; acc[8] = ax * bz;
		acc[9] = ay * bx; acc[10] = ay * by; acc[11] = ay * bz;
		acc[12] = az * bx; acc[13] = az * by; acc[14] = az * bz;
		acc[15] = nn_d2[i];
	}
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int k = 0; k < kIcpAcc; k++) {
		const float s = wave_sum(acc[k]);
		if (lane == 0) red[wave][k] = s;
	}
	__syncthreads();
	if (threadIdx.x < kIcpAcc) {
		float s = red[0][threadIdx.x];
#pragma unroll
		for (int x = 1; x < kIcpThreads / 64; x++) s += red[x][threadIdx.x];
		partials[(size_t)blockIdx.x * kIcpAcc + threadIdx.x] = s;
	}
}

// Sum the per-block partials in double (fixed order -> deterministic), then -- one lane -- the body of
// the ICP loop after the correspondence pass (jly_icp3d.hpp:253-292): convergence test, means, H,
// SVD, R_ / t_, compose.  The pose lives in device memory, so the host can queue several iterations
// back-to-back without a round trip.  Written for one workgroup of kFinThreads threads; used by the last
// workgroup of the fused iteration kernel and by the stand-alone finalize launch (trimmed ICP, A/B tests):
// the same code and the same summation order, so both forms give bit-identical states.

// every thread of the workgroup (T threads); returns with sh.sums[] valid for all threads
template <int T>
__device__ __forceinline__ void finalize_reduce(const float* __restrict__ partials, int nblocks, FinScratch& sh)
{
	// A chain of dependent memory round trips (~1-2 us each: the partials were written by other XCDs), so the loads
	// of a sweep -- eight float4 per thread -- are requested before anything is waited for.  64 row streams x four
	// float4 columns; fixed-order sums (registers, xor-butterfly inside a wavefront, then the wavefront totals in order).
	constexpr int kRows = T / 4;
	const int q = threadIdx.x & 3, r = threadIdx.x >> 2;
	const float4* __restrict__ P = reinterpret_cast<const float4*>(partials);
	double a[4] = {0.0, 0.0, 0.0, 0.0};
	for (int b0 = 0; b0 < nblocks; b0 += kRows * 8) {
		float4 v[8];
#pragma unroll
		for (int j = 0; j < 8; j++) {
			const int b = b0 + r + kRows * j;
			v[j] = b < nblocks ? P[(size_t)b * 4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
		}
#pragma unroll
		for (int j = 0; j < 8; j++) {
			a[0] += (double)v[j].x;
			a[1] += (double)v[j].y;
			a[2] += (double)v[j].z;
			a[3] += (double)v[j].w;
		}
	}
#pragma unroll
	for (int c = 0; c < 4; c++) {
		double x = a[c];
		x += __shfl_xor(x, 4, 64);
		x += __shfl_xor(x, 8, 64);
		x += __shfl_xor(x, 16, 64);
		x += __shfl_xor(x, 32, 64);
		if ((threadIdx.x & 63) < 4) sh.wsum[threadIdx.x >> 6][4 * q + c] = x;
	}
	__syncthreads();
	if (threadIdx.x < kIcpAcc) {
		double t = 0.0;
		for (int i = 0; i < T / 64; i++) t += sh.wsum[i][threadIdx.x];
		sh.sums[threadIdx.x] = t;
	}
	__syncthreads();
}

// ---- the rest of the loop body, three lanes wide ---------------------------------------------------------------
// Lane i (i = 0, 1, 2 of ONE wavefront; the other lanes idle along) owns coordinate i of every vector and row i of every
// 3x3 matrix: the means, H, the one-sided Jacobi SVD (rows of B and V rotate independently; the column inner products
// are three-lane sums taken in the serial order (x0 + x1) + x2), R_ = V diag(1,1,det) U^T, t_, the composed pose.  Same
// arithmetic as a one-lane version, a third of the dependent fp64 chain, and ~40 registers instead of ~150 -- which is
// what lets the correspondence pass carry this code without losing occupancy.
__device__ __forceinline__ double lane_get(double x, int j) { return __shfl(x, j, 64); }
__device__ __forceinline__ float lane_getf(float x, int j) { return __shfl(x, j, 64); }
// lane J of the wavefront, J a constant: two v_readlane_b32 into scalar registers instead of two ds_bpermute_b32 through
// the LDS crossbar (a Jacobi rotation takes three such sums; measured: Kabsch 3.7 -> 2.x us per iteration)
template <int J>
__device__ __forceinline__ double lane_const(double x)
{
	const unsigned long long u = (unsigned long long)__double_as_longlong(x);
	const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, J), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), J);
	return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double sum3(double x) { return (lane_const<0>(x) + lane_const<1>(x)) + lane_const<2>(x); }

// rows of H in (b0, b1, b2) on lanes 0..2 -> row of the Kabsch rotation in r[3]: 3x3 SVD by one-sided Jacobi sweeps in double,
// R = V diag(1,1,det) U^T with the sign on the smallest singular direction (jly_icp3d.hpp:268-285, matrix.cpp:782-808)
__device__ __forceinline__ void kabsch_rows(double b0, double b1, double b2, int row, float r[3])
{
	double v0 = row == 0 ? 1.0 : 0.0, v1 = row == 1 ? 1.0 : 0.0, v2 = row == 2 ? 1.0 : 0.0;
#define GOICP_JACOBI(bp, bq, vp, vq)                                                                        \
	{                                                                                                       \
		const double app = sum3(bp * bp), aqq = sum3(bq * bq), apq = sum3(bp * bq);                         \
		if (!(apq == 0.0 || apq * apq <= 1e-24 * (app * aqq))) {                                            \
			rotated = true;                                                                                 \
			const double da = aqq - app, db = 2 * apq;                                                      \
			const double tn = (da >= 0 ? db : -db) / (fabs(da) + sqrt(da * da + db * db));                  \
			const double cs = rsqrt(1 + tn * tn), sn = cs * tn;                                             \
			const double nbp = cs * bp - sn * bq, nbq = sn * bp + cs * bq;                                  \
			const double nvp = cs * vp - sn * vq, nvq = sn * vp + cs * vq;                                  \
			bp = nbp; bq = nbq; vp = nvp; vq = nvq;                                                         \
		}                                                                                                   \
	}
#pragma unroll 1
	for (int sweep = 0; sweep < 32; sweep++) {
		bool rotated = false;
		GOICP_JACOBI(b0, b1, v0, v1)
		GOICP_JACOBI(b0, b2, v0, v2)
		GOICP_JACOBI(b1, b2, v1, v2)
		if (!rotated) break;
	}
#undef GOICP_JACOBI
	const double n0 = sum3(b0 * b0), n1 = sum3(b1 * b1), n2 = sum3(b2 * b2);
	const double rn0 = n0 > 0 ? rsqrt(n0) : 0.0, rn1 = n1 > 0 ? rsqrt(n1) : 0.0, rn2 = n2 > 0 ? rsqrt(n2) : 0.0;
	const double W0 = n0 * rn0, W1 = n1 * rn1, W2 = n2 * rn2;
	double u0 = b0 * rn0, u1 = b1 * rn1, u2 = b2 * rn2;                       // row `row` of U
	// rank-2 input: the missing left vector is the cross product of the other two (row i needs rows i+1, i+2)
	const int ra = (row + 1) % 3, rb = (row + 2) % 3;
	{
		const double a1 = lane_get(u1, ra), a2 = lane_get(u2, ra), c1 = lane_get(u1, rb), c2 = lane_get(u2, rb);   // U[ra][1], U[ra][2], U[rb][1], U[rb][2]
		const double a0 = lane_get(u0, ra), c0 = lane_get(u0, rb);
		if (!(W0 > 1e-200) && W1 > 1e-200 && W2 > 1e-200) u0 = a1 * c2 - c1 * a2;       // columns a = 1, b = 2
		if (!(W1 > 1e-200) && W2 > 1e-200 && W0 > 1e-200) u1 = a2 * c0 - c2 * a0;       // columns a = 2, b = 0
		if (!(W2 > 1e-200) && W0 > 1e-200 && W1 > 1e-200) u2 = a0 * c1 - c0 * a1;       // columns a = 0, b = 1
	}
	// VUt[row][j] = sum_k V[row][k] U[j][k]
	double vut[3];
#pragma unroll
	for (int j = 0; j < 3; j++) {
		double t = 0;
		t += v0 * lane_get(u0, j);
		t += v1 * lane_get(u1, j);
		t += v2 * lane_get(u2, j);
		vut[j] = t;
	}
	const double m00 = lane_get(vut[0], 0), m01 = lane_get(vut[1], 0), m02 = lane_get(vut[2], 0);
	const double m10 = lane_get(vut[0], 1), m11 = lane_get(vut[1], 1), m12 = lane_get(vut[2], 1);
	const double m20 = lane_get(vut[0], 2), m21 = lane_get(vut[1], 2), m22 = lane_get(vut[2], 2);
	const double det = m00 * (m11 * m22 - m12 * m21) - m01 * (m10 * m22 - m12 * m20) + m02 * (m10 * m21 - m11 * m20);
	// the reference sorts singular values in decreasing order (matrix.cpp:782-808): its diag(1,1,det)
	// (jly_icp3d.hpp:268-285) corrects the direction of the smallest one
	int ks = 0;
	double Wk = W0;
	if (W1 < Wk) { ks = 1; Wk = W1; }
	if (W2 < Wk) ks = 2;
#pragma unroll
	for (int j = 0; j < 3; j++) {
		double t = 0;
		t += v0 * (ks == 0 ? det : 1.0) * lane_get(u0, j);
		t += v1 * (ks == 1 ? det : 1.0) * lane_get(u1, j);
		t += v2 * (ks == 2 ? det : 1.0) * lane_get(u2, j);
		r[j] = (float)t;
	}
}

// called by every lane of ONE wavefront (lanes >= 3 shadow lane 2 and write nothing)
// state: where the new pose is written; rd: where the old one is read -- the stand-alone finalize hands in a copy its threads
// fetched into LDS beside the partial rows (one round trip instead of a chain of four behind the branches below)
__device__ __forceinline__ void finalize_rows(const double* __restrict__ sums, IcpState* __restrict__ state, const IcpState& rd, int lane)
{
	const int a = lane < 3 ? lane : 2;
	const bool writer = lane < 3;
	const float err_new = (float)sums[15];
	const int passes = rd.passes + 1;
	if (rd.frozen) {                                                     // timing / scoring only
		if (lane == 0) { state->err_new = err_new; state->passes = passes; }
		return;
	}
	const float err = rd.err;
	if (err > 0.f && err - err_new < rd.err_diff_n) {                    // jly_icp3d.hpp:255
		if (lane == 0) { state->err_new = err_new; state->passes = passes; state->converged = 1; }
		return;
	}
	const double nn = (double)rd.n;
	const float cq = rd.cq[a], cm = rd.cm[a];
	const bool carry = rd.carry_means != 0;
	const double sum_q = sums[a] + nn * (double)cq;
	const double sum_m = sums[3 + a] + nn * (double)cm;
	// jly_icp3d.hpp:244-263: the reference accumulates on top of the previous means and divides by n
	const double carry_d = carry ? (double)rd.mu_d[a] : 0.0;
	const double carry_m = carry ? (double)rd.mu_m[a] : 0.0;
	const float mu_d = (float)((carry_d + sum_q) / nn);
	const float mu_m = (float)((carry_m + sum_m) / nn);
	const double alpha = (double)mu_d - (double)cq, beta = (double)mu_m - (double)cm;
	double h[3];
#pragma unroll
	for (int j = 0; j < 3; j++) {
		const double hj = sums[6 + 3 * a + j] - alpha * sums[3 + j] - sums[a] * lane_get(beta, j) + nn * alpha * lane_get(beta, j);
		h[j] = (double)(float)hj;                                            // the reference holds H in float
	}
	float Rrow[3];
	kabsch_rows(h[0], h[1], h[2], a, Rrow);
	float oldR[9], oldt[3], sc[3];
#pragma unroll
	for (int k = 0; k < 9; k++) oldR[k] = rd.R[k];
#pragma unroll
	for (int k = 0; k < 3; k++) { oldt[k] = rd.t[k]; sc[k] = rd.src_centroid[k]; }
	float acc = 0.f;
#pragma unroll
	for (int k = 0; k < 3; k++) acc += Rrow[k] * lane_getf(mu_d, k);
	const float t_ = mu_m - acc;                                             // t_ = mu_m - R_ mu_d
	float Rn[3];
#pragma unroll
	for (int j = 0; j < 3; j++) {
		float s2 = 0.f;
#pragma unroll
		for (int k = 0; k < 3; k++) s2 += Rrow[k] * oldR[3 * k + j];
		Rn[j] = s2;                                                          // R <- R_ R
	}
	float s3 = 0.f;
#pragma unroll
	for (int k = 0; k < 3; k++) s3 += Rrow[k] * oldt[k];
	const float tn = s3 + t_;                                                // t <- R_ t + t_
	const float cq_new = Rn[0] * sc[0] + Rn[1] * sc[1] + Rn[2] * sc[2] + tn;
	const int iters = rd.iters + 1;
	if (writer) {
		state->R[3 * a] = Rn[0]; state->R[3 * a + 1] = Rn[1]; state->R[3 * a + 2] = Rn[2];
		state->t[a] = tn; state->cq[a] = cq_new; state->mu_d[a] = mu_d; state->mu_m[a] = mu_m;
	}
	if (lane == 0) { state->err = err_new; state->err_new = err_new; state->passes = passes; state->iters = iters; }
}

// Large clouds (1 M queries = 62 500 partial rows): a single workgroup reading them all is an 85 us tail on a 1.1 ms pass.
// One more level: workgroup b sums the rows [b * rows_per_block, ...) in double, fixed order, and writes ONE float row; the
// finalize then reads nblocks / rows_per_block rows.
constexpr int kPreThreads = 256, kPreRows = 1024;
__global__ __launch_bounds__(kPreThreads) void icp_partials_reduce(const float* __restrict__ partials, int nblocks, const IcpState* __restrict__ state,
                                                                   float* __restrict__ out)
{
	__shared__ FinScratch sh;
	if (state->converged) return;
	const int b0 = blockIdx.x * kPreRows;
	finalize_reduce<kPreThreads>(partials + (size_t)b0 * kIcpAcc, min(kPreRows, nblocks - b0), sh);
	if (threadIdx.x < kIcpAcc) out[(size_t)blockIdx.x * kIcpAcc + threadIdx.x] = (float)sh.sums[threadIdx.x];
}

__global__ __launch_bounds__(kFinThreads) void icp_finalize_update(const float* __restrict__ partials, int nblocks,
                                                                   IcpState* __restrict__ state)
{
	__shared__ FinScratch sh;
	// the loop state travels to LDS in the same round trip as the flag (one word per thread; finalize_reduce's barriers
	// publish it): finalize_rows then reads it without the chain of dependent global loads its branches would make
	static_assert(sizeof(IcpState) % 4 == 0 && sizeof(IcpState) / 4 <= kFinThreads, "one word per thread");
	if (threadIdx.x < sizeof(IcpState) / 4) reinterpret_cast<unsigned*>(&sh.st)[threadIdx.x] = reinterpret_cast<const unsigned*>(state)[threadIdx.x];
	if (state->converged) return;                 // uniform; the pass kernel left the partials untouched
	finalize_reduce<kFinThreads>(partials, nblocks, sh);
	if (threadIdx.x < 64) finalize_rows(sh.sums, state, sh.st, (int)threadIdx.x);
}

// the finalize of the fixed-point form: 32 replicas x 16 accumulators in, zeroed again on the way out.  ONE wavefront (round 4; it was a
// workgroup of 1 024 threads, one accumulator each): lane t owns accumulator t & 15 of the replicas (t >> 4) + 4 j -- eight loads in flight,
// two shuffles, no cross-wavefront stage -- and a single wavefront finds a free SIMD at once where sixteen had to wait for a whole compute
// unit (tools/overlap_probe.py: beside a bound-evaluation stream the old form waited ~400 us per iteration).  Integer sums: the same totals.
constexpr int kFinAccThreads = 64;
__global__ __launch_bounds__(kFinAccThreads) void icp_finalize_update_acc(unsigned long long* __restrict__ acc, IcpState* __restrict__ state)
{
	__shared__ double sums[kIcpAcc];
	__shared__ IcpState st;
	static_assert(sizeof(IcpState) / 4 <= kFinAccThreads && (kIcpAccReplicas * kIcpAcc) % kFinAccThreads == 0 && kFinAccThreads / kIcpAcc == 4, "one state word per lane; four replicas per pass of the wavefront");
	const int t = threadIdx.x;
	if (t < (int)(sizeof(IcpState) / 4)) reinterpret_cast<unsigned*>(&st)[t] = reinterpret_cast<const unsigned*>(state)[t];
	if (state->converged) return;                 // uniform; the pass kernel added nothing
	long long v[kIcpAccReplicas * kIcpAcc / kFinAccThreads];
#pragma unroll
	for (int j = 0; j < kIcpAccReplicas * kIcpAcc / kFinAccThreads; j++) v[j] = (long long)acc[kFinAccThreads * j + t];
#pragma unroll
	for (int j = 0; j < kIcpAccReplicas * kIcpAcc / kFinAccThreads; j++) acc[kFinAccThreads * j + t] = 0ull;   // the next pass starts from zero (it cannot start before this kernel has finished)
	long long x = 0;
#pragma unroll
	for (int j = 0; j < kIcpAccReplicas * kIcpAcc / kFinAccThreads; j++) x += v[j];
	x += __shfl_xor(x, 16, 64);
	x += __shfl_xor(x, 32, 64);
	__syncthreads();                              // the state words are in LDS
	if (t < kIcpAcc) sums[t] = (double)x * (double)st.acc_inv;
	__syncthreads();
	finalize_rows(sums, state, st, t);
}

// test-only entry (goicp_debug_kabsch): the device SVD on a caller-supplied H, one lane
__global__ void kabsch_debug_kernel(const float* __restrict__ H, float* __restrict__ R)
{
	// the production routine: lane i < 3 holds row i of H, gets row i of R (the other lanes shadow lane 2)
	if (blockIdx.x != 0 || threadIdx.x >= 64) return;
	const int a = threadIdx.x < 3 ? (int)threadIdx.x : 2;
	float Rrow[3];
	kabsch_rows((double)H[3 * a], (double)H[3 * a + 1], (double)H[3 * a + 2], a, Rrow);
	if (threadIdx.x < 3) { R[3 * a] = Rrow[0]; R[3 * a + 1] = Rrow[1]; R[3 * a + 2] = Rrow[2]; }
}
hipError_t launch_kabsch_debug(const float* d_H, float* d_R, hipStream_t stream)
{
	hipLaunchKernelGGL(kabsch_debug_kernel, dim3(1), dim3(64), 0, stream, d_H, d_R);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Gather-ceiling probe (bench.py roofline): the bounds kernel is bound by the vector-memory address /
// L1 path, not by HBM, so its ceiling is measured, in the same run, by a kernel that does nothing
// but issue independent 4-byte loads into the resident distance transform:
//   MODE 0  fully coalesced: the 64 lanes of a wave-instruction read 64 consecutive floats (2 lines)
//   MODE 1  fully divergent: the 64 lanes read 64 different 128-byte lines
// Every workgroup draws its addresses from its own window of `window` floats (a power of two), so the
// footprint per CU -- L1-resident, L2-resident, or the whole grid -- is the caller's choice.  Eight loads
// in flight per lane per iteration, a wave-uniform LCG picks the bases (scalar ALU only).
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void probe_gather_kernel(const float* __restrict__ grid, unsigned n_floats, unsigned window, int iters,
                                                           float* __restrict__ sink)
{
	const unsigned lane = threadIdx.x & 63u;
	const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6);
	const unsigned span = n_floats - window;
	const unsigned wbase = span ? (unsigned)(((unsigned long long)blockIdx.x * 2654435761ull) % span) & ~63u : 0u;
	const float* __restrict__ w = grid + wbase;
	const unsigned mask = window - 1u;
	unsigned h = __builtin_amdgcn_readfirstlane(wave * 2654435761u + 12345u);
	float acc = 0.f;
	// MODE 0: 64 consecutive floats; MODE 1: 64 lines; MODE k >= 2: k distinct 128-byte lines per instruction, the lanes in
	// k runs of 64/k consecutive floats (lines 8 apart so that neighbouring runs never share a line)
	const unsigned mine = MODE == 0 ? lane : (MODE == 1 ? lane * 32u : (lane / (64u / MODE)) * 256u + (lane % (64u / MODE)));
	for (int it = 0; it < iters; it++) {
		float v[8];
#pragma unroll
		for (int u = 0; u < 8; u++) {
			h = h * 1664525u + 1013904223u;
			const unsigned base = MODE == 1 ? ((h >> 7) & mask) : ((h >> 7) & mask & ~31u);
			v[u] = w[(base + mine) & mask];
		}
#pragma unroll
		for (int u = 0; u < 8; u++) acc += v[u];
	}
	if (acc == 1.2345e-30f) sink[0] = acc;      // never true for a distance field; keeps the loads alive
}

// MODE 2 of the probe: what an LDS-staged DT tile would deliver -- a 64 KiB tile of the grid copied into LDS once, then
// the same independent 4-byte lookups at random tile addresses served by ds_read_b32 (bank conflicts included).
__global__ __launch_bounds__(256) void probe_lds_kernel(const float* __restrict__ grid, unsigned n_floats, int iters, float* __restrict__ sink)
{
	__shared__ float tile[16384];
	const unsigned base = (unsigned)(((unsigned long long)blockIdx.x * 2654435761ull) % (n_floats - 16384u)) & ~63u;
	for (int i = threadIdx.x; i < 16384; i += 256) tile[i] = grid[base + i];
	__syncthreads();
	unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;        // per-lane addresses: a gather
	float acc = 0.f;
	for (int it = 0; it < iters; it++) {
		float v[8];
#pragma unroll
		for (int u = 0; u < 8; u++) {
			h = h * 1664525u + 1013904223u;
			v[u] = tile[(h >> 9) & 16383u];
		}
#pragma unroll
		for (int u = 0; u < 8; u++) acc += v[u];
	}
	if (acc == 1.2345e-30f) sink[0] = acc;
}

hipError_t launch_probe_gather(const DtDesc& dt, int mode, unsigned window, int blocks, int iters, float* sink, hipStream_t stream)
{
	if (mode == 2) {
		const unsigned n = dt.layout ? (unsigned)dt.VB * dt.VB * dt.VB * 64u : (unsigned)dt.V * dt.V * dt.V;
		if (n <= 16384u || dt.layout == 2) return hipErrorInvalidValue;
		hipLaunchKernelGGL(probe_lds_kernel, dim3(blocks), dim3(256), 0, stream, dt.grid, n, iters, sink);
		return hipGetLastError();
	}
	const unsigned n = dt.layout ? (unsigned)dt.VB * dt.VB * dt.VB * 64u : (unsigned)dt.V * dt.V * dt.V;
	if (window < 4096u || (window & (window - 1u)) || window > n) return hipErrorInvalidValue;
	switch (mode) {
	case 0: hipLaunchKernelGGL(probe_gather_kernel<0>, dim3(blocks), dim3(256), 0, stream, dt.grid, n, window, iters, sink); break;
	case 1: hipLaunchKernelGGL(probe_gather_kernel<1>, dim3(blocks), dim3(256), 0, stream, dt.grid, n, window, iters, sink); break;
	case 4: hipLaunchKernelGGL(probe_gather_kernel<4>, dim3(blocks), dim3(256), 0, stream, dt.grid, n, window, iters, sink); break;
	case 8: hipLaunchKernelGGL(probe_gather_kernel<8>, dim3(blocks), dim3(256), 0, stream, dt.grid, n, window, iters, sink); break;
	case 16: hipLaunchKernelGGL(probe_gather_kernel<16>, dim3(blocks), dim3(256), 0, stream, dt.grid, n, window, iters, sink); break;
	case 32: hipLaunchKernelGGL(probe_gather_kernel<32>, dim3(blocks), dim3(256), 0, stream, dt.grid, n, window, iters, sink); break;
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

size_t icp_partials_floats(int N)     // the pass's rows + the rows of the intermediate level (large clouds)
{
	const size_t nb = (size_t)icp_blocks(N);
	return (nb + (nb + kPreRows - 1) / kPreRows) * kIcpAcc;
}

int icp_blocks(int N)
{
	const int per_block = (kIcpThreads / 64) * 4;      // four queries per wavefront
	return (N + per_block - 1) / per_block;
}

// the fixed-point form (bricked DT, two launches per iteration): `acc` is the accumulator block; strangers per wavefront up to
// kIcpStridedMaxN points, neighbours above
template <int K>
static void launch_pass_acc(const float4* src, int N, IcpState* st, const KdDesc& kd, const DtDesc& dt, unsigned long long* acc, float4* nn_cache,
                            int* hits, hipStream_t stream)
{
	const dim3 grid(icp_blocks(N)), block(kIcpThreads);
	float* a = reinterpret_cast<float*>(acc);
	const bool strided = N <= kIcpStridedMaxN;
	if (nn_cache) {
		if (strided) hipLaunchKernelGGL((icp_pass_kernel<K, 1, false, true, 2, true, true>), grid, block, 0, stream, src, N, st, kd, dt, a, nullptr, nn_cache, hits);
		else hipLaunchKernelGGL((icp_pass_kernel<K, 1, false, true, 2, false, true>), grid, block, 0, stream, src, N, st, kd, dt, a, nullptr, nn_cache, hits);
	} else {
		if (strided) hipLaunchKernelGGL((icp_pass_kernel<K, 1, false, false, 4, true, true>), grid, block, 0, stream, src, N, st, kd, dt, a, nullptr, nn_cache, hits);
		else hipLaunchKernelGGL((icp_pass_kernel<K, 1, false, false, 4, false, true>), grid, block, 0, stream, src, N, st, kd, dt, a, nullptr, nn_cache, hits);
	}
}

template <int K, bool FUSED>
static void launch_pass_k(const float4* src, int N, IcpState* st, const KdDesc& kd, const DtDesc& dt, float* partials, int* ticket, float4* nn_cache,
                          int* hits, hipStream_t stream)
{
	const dim3 grid(icp_blocks(N)), block(kIcpThreads);
	if (nn_cache) {
		if (dt.layout) hipLaunchKernelGGL((icp_pass_kernel<K, 1, FUSED, true>), grid, block, 0, stream, src, N, st, kd, dt, partials, ticket, nn_cache, hits);
		else hipLaunchKernelGGL((icp_pass_kernel<K, 0, FUSED, true>), grid, block, 0, stream, src, N, st, kd, dt, partials, ticket, nn_cache, hits);
	} else {
		// four leaves per scan step (S2 845 -> 926 iterations/s, S1 21.9 k -> 23.0 k, bunny 27.0 k -> 27.1 k)
		if (dt.layout) hipLaunchKernelGGL((icp_pass_kernel<K, 1, FUSED, false, FUSED ? 2 : 4>), grid, block, 0, stream, src, N, st, kd, dt, partials, ticket, nn_cache, hits);
		else hipLaunchKernelGGL((icp_pass_kernel<K, 0, FUSED, false, FUSED ? 2 : 4>), grid, block, 0, stream, src, N, st, kd, dt, partials, ticket, nn_cache, hits);
	}
}

template <int K>
static void launch_nn_store_k(const float4* src, int N, IcpState* st, const KdDesc& kd, const DtDesc& dt, float* d2, int* slot, hipStream_t stream)
{
	const dim3 grid(icp_blocks(N)), block(kIcpThreads);
	if (dt.layout) hipLaunchKernelGGL((icp_nn_kernel<K, 1>), grid, block, 0, stream, src, N, st, kd, dt, d2, slot);
	else hipLaunchKernelGGL((icp_nn_kernel<K, 0>), grid, block, 0, stream, src, N, st, kd, dt, d2, slot);
}

int icp_trim_blocks(int N) { return (N + kIcpThreads - 1) / kIcpThreads; }

hipError_t launch_icp_iteration_trim(const float4* src, int N, int num, IcpState* st, const KdDesc& kd, const DtDesc& dt,
                                     float* nn_d2, int* nn_slot, unsigned char* include, float* partials, hipStream_t stream)
{
	if (kd.K == 1) launch_nn_store_k<1>(src, N, st, kd, dt, nn_d2, nn_slot, stream);
	else if (kd.K == 2) launch_nn_store_k<2>(src, N, st, kd, dt, nn_d2, nn_slot, stream);
	else launch_nn_store_k<3>(src, N, st, kd, dt, nn_d2, nn_slot, stream);
	if (N <= 1024 * kSelPer) hipLaunchKernelGGL(icp_select_kernel, dim3(1), dim3(1024), 0, stream, nn_d2, N, num, st, include);
	else hipLaunchKernelGGL(icp_select_stream_kernel, dim3(1), dim3(1024), 0, stream, nn_d2, N, num, st, include);
	const int nb = icp_trim_blocks(N);
	hipLaunchKernelGGL(icp_accum_kernel, dim3(nb), dim3(kIcpThreads), 0, stream, src, N, st, kd, nn_d2, nn_slot, include, partials);
	hipLaunchKernelGGL(icp_finalize_update, dim3(1), dim3(kFinThreads), 0, stream, partials, nb, st);
	return hipGetLastError();
}

// ticket != nullptr: one fused launch per iteration; nullptr: pass + stand-alone finalize (same arithmetic, bit-identical)
hipError_t launch_icp_iteration(const float4* src, int N, IcpState* st, const KdDesc& kd, const DtDesc& dt, float* partials,
                                int* ticket, float4* nn_cache, int* hits, hipStream_t stream, unsigned long long* acc)
{
	if (!ticket && acc && dt.layout) {
		// the default form: fixed-point sums, no rows of partial sums
		if (kd.K == 1) launch_pass_acc<1>(src, N, st, kd, dt, acc, nn_cache, hits, stream);
		else if (kd.K == 2) launch_pass_acc<2>(src, N, st, kd, dt, acc, nn_cache, hits, stream);
		else launch_pass_acc<3>(src, N, st, kd, dt, acc, nn_cache, hits, stream);
		hipLaunchKernelGGL(icp_finalize_update_acc, dim3(1), dim3(kFinAccThreads), 0, stream, acc, st);
		return hipGetLastError();
	}
	if (ticket) {
		if (kd.K == 1) launch_pass_k<1, true>(src, N, st, kd, dt, partials, ticket, nn_cache, hits, stream);
		else if (kd.K == 2) launch_pass_k<2, true>(src, N, st, kd, dt, partials, ticket, nn_cache, hits, stream);
		else launch_pass_k<3, true>(src, N, st, kd, dt, partials, ticket, nn_cache, hits, stream);
		return hipGetLastError();
	}
	if (kd.K == 1) launch_pass_k<1, false>(src, N, st, kd, dt, partials, nullptr, nn_cache, hits, stream);
	else if (kd.K == 2) launch_pass_k<2, false>(src, N, st, kd, dt, partials, nullptr, nn_cache, hits, stream);
	else launch_pass_k<3, false>(src, N, st, kd, dt, partials, nullptr, nn_cache, hits, stream);
	const int nb = icp_blocks(N);
	if (nb > 4 * kPreRows) {
		// two-level sum: the reduced rows live behind the pass's own rows in the same buffer (icp_partials_floats sizes it)
		float* reduced = partials + (size_t)nb * kIcpAcc;
		const int nb2 = (nb + kPreRows - 1) / kPreRows;
		hipLaunchKernelGGL(icp_partials_reduce, dim3(nb2), dim3(kPreThreads), 0, stream, partials, nb, st, reduced);
		hipLaunchKernelGGL(icp_finalize_update, dim3(1), dim3(kFinThreads), 0, stream, reduced, nb2, st);
	} else {
		hipLaunchKernelGGL(icp_finalize_update, dim3(1), dim3(kFinThreads), 0, stream, partials, nb, st);
	}
	return hipGetLastError();
}

__global__ void transform_kernel(float4* __restrict__ src, int N, Pose pose)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= N) return;
	float4 p = src[i];
	float x = pose.R[0] * p.x + pose.R[1] * p.y + pose.R[2] * p.z + pose.t[0];
	float y = pose.R[3] * p.x + pose.R[4] * p.y + pose.R[5] * p.z + pose.t[1];
	float z = pose.R[6] * p.x + pose.R[7] * p.y + pose.R[8] * p.z + pose.t[2];
	src[i] = make_float4(x, y, z, __fsqrt_rn(x * x + y * y + z * z));
}

hipError_t launch_transform(float4* src, int N, const Pose& pose, hipStream_t stream)
{
	if (N <= 0) return hipSuccess;
	hipLaunchKernelGGL(transform_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, src, N, pose);
	return hipGetLastError();
}

// min over n floats and the first index that attains it (what a search does with the upper bounds of a batch: the incumbent
// is the smallest ub, first child on ties -- bnb_queue_kernel's digest; exposed so that a caller driving goicp_eval_bounds_device
// itself, like bench.py's step, reduces with the library's own code).  One workgroup: 65 536 values are 64 per thread.
__global__ __launch_bounds__(1024) void reduce_min_kernel(const float* __restrict__ v, int n, float* __restrict__ out_min, int* __restrict__ out_idx)
{
	__shared__ float sv[16];
	__shared__ int si[16];
	float best = __builtin_inff();
	int bi = 0x7fffffff;
	for (int i = threadIdx.x * 4; i < n; i += 4096) {
		if (i + 3 < n) {
			const float4 x = *reinterpret_cast<const float4*>(v + i);
			if (x.x < best) { best = x.x; bi = i; }
			if (x.y < best) { best = x.y; bi = i + 1; }
			if (x.z < best) { best = x.z; bi = i + 2; }
			if (x.w < best) { best = x.w; bi = i + 3; }
		} else
			for (int k = i; k < n; k++) if (v[k] < best) { best = v[k]; bi = k; }
	}
	for (int off = 32; off; off >>= 1) {
		const float ov = __shfl_xor(best, off);
		const int oi = __shfl_xor(bi, off);
		if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
	}
	if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < 16; w++) if (sv[w] < best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
		out_min[0] = best;
		if (out_idx) out_idx[0] = bi;
	}
}

hipError_t launch_reduce_min(const float* v, int n, float* out_min, int* out_idx, hipStream_t stream)
{
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(reduce_min_kernel, dim3(1), dim3(1024), 0, stream, v, n, out_min, out_idx);
	return hipGetLastError();
}

template <int K, int LAYOUT>
__global__ __launch_bounds__(kIcpThreads) void nn_query_kernel(const float* __restrict__ q, int n, KdDesc kd, DtDesc dt,
                                                               int32_t* __restrict__ idx, float* __restrict__ d2)
{
	const int lane = threadIdx.x & 63, row = lane >> 4, l = lane & 15;
	const int i = (blockIdx.x * (kIcpThreads / 64) + (threadIdx.x >> 6)) * 4 + row;   // one 16-lane row per query
	const bool valid = i < n;
	const int iq = valid ? i : n - 1;
	const float qx = q[3 * iq], qy = q[3 * iq + 1], qz = q[3 * iq + 2];
	const RowNn r = rows_nearest<K, LAYOUT, false, 4>(kd, dt, load_child_boxes4(kd.boxes[0], l), l, row, qx, qy, qz, valid);
	if (valid && r.mine) { idx[i] = r.idx; d2[i] = r.best; }
}

template <int K>
static void launch_nn_k(const float* q, int n, const KdDesc& kd, const DtDesc& dt, int32_t* idx, float* d2, hipStream_t stream)
{
	const dim3 grid(icp_blocks(n)), block(kIcpThreads);
	if (dt.layout) hipLaunchKernelGGL((nn_query_kernel<K, 1>), grid, block, 0, stream, q, n, kd, dt, idx, d2);
	else hipLaunchKernelGGL((nn_query_kernel<K, 0>), grid, block, 0, stream, q, n, kd, dt, idx, d2);
}

hipError_t launch_nn_query(const float* q, int n, const KdDesc& kd, const DtDesc& dt, int32_t* idx, float* d2, hipStream_t stream)
{
	if (n <= 0) return hipSuccess;
	if (kd.K == 1) launch_nn_k<1>(q, n, kd, dt, idx, d2, stream);
	else if (kd.K == 2) launch_nn_k<2>(q, n, kd, dt, idx, d2, stream);
	else launch_nn_k<3>(q, n, kd, dt, idx, d2, stream);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Exact Euclidean distance transform of the seed grid (replaces DT3D::Build's 8-pass propagation,
// jly_3ddt.cpp:710-742; same bbox/expand/cubify/int(x+0.5) seeding, jly_3ddt.cpp:889-965).
// Squared distances are integers, so the three separable min-plus passes are exact.
// ------------------------------------------------------------------------------------------------
constexpr int kEdtInf = 1 << 28;

__global__ void dt_fill_kernel(int32_t* __restrict__ w, size_t n)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	for (; i < n; i += (size_t)gridDim.x * blockDim.x) w[i] = kEdtInf;
}

__global__ void dt_seed_kernel(const float* __restrict__ m, int M, DtDesc dt, int32_t* __restrict__ w)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= M) return;
	// the model floats are widened to double before the index math (jly_goicp.cpp:80-85, jly_3ddt.cpp:954-956)
	int x = (int)(((double)m[3 * i] - dt.xmin) * dt.scale + 0.5);
	int y = (int)(((double)m[3 * i + 1] - dt.ymin) * dt.scale + 0.5);
	int z = (int)(((double)m[3 * i + 2] - dt.zmin) * dt.scale + 0.5);
	const int V = dt.V;
	if (x < 0 || x >= V || y < 0 || y >= V || z < 0 || z >= V) return;   // :958
	w[((size_t)z * V + y) * V + x] = 0;                                    // benign race: every writer stores 0
}

// one block per grid row along `axis`; out[x] = min_i (x-i)^2 + g[i]; the row lives in LDS and
// the inner read is a broadcast.
__global__ __launch_bounds__(256) void dt_pass_kernel(int32_t* __restrict__ w, int V, int axis)
{
	extern __shared__ int32_t row[];
	const int r = blockIdx.x;
	size_t base, stride;
	if (axis == 0) { base = (size_t)r * V; stride = 1; }
	else if (axis == 1) { int z = r / V, x = r - z * V; base = (size_t)z * V * V + x; stride = V; }
	else { base = r; stride = (size_t)V * V; }
	for (int i = threadIdx.x; i < V; i += blockDim.x) row[i] = w[base + i * stride];
	__syncthreads();
	for (int x = threadIdx.x; x < V; x += blockDim.x) {
		int best = kEdtInf;
		for (int i = 0; i < V; i++) {
			int d = x - i;
			int v = d * d + row[i];
			best = v < best ? v : best;
		}
		w[base + x * stride] = best;
	}
}

template <int LAYOUT>
__global__ void dt_finish_kernel(const int32_t* w, DtDesc dt, float* out)
{
	const int V = dt.V;
	size_t n = (size_t)V * V * V;
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	// jly_3ddt.cpp:967-978: the float holding sqrt(v^2+h^2+d^2), divided by the double scale
	float dist = (float)sqrt((double)w[i]);
	float v = (float)((double)dist / dt.scale);
	if (v < 0.f) v = 0.f;
	if (LAYOUT == 0) {
		out[i] = v;
	} else {
		int x = (int)(i % V), y = (int)((i / V) % V), z = (int)(i / ((size_t)V * V));
		size_t b = ((size_t)(z >> 2) * dt.VB + (y >> 2)) * dt.VB + (x >> 2);
		out[b * 64 + (((z & 3) << 4) | ((y & 3) << 2) | (x & 3))] = v;
	}
}

// ---- nearest-target-point table: the EDT passes again, carrying which seed attains the minimum ----
__global__ void nnseed_fill_kernel(int32_t* __restrict__ w, int32_t* __restrict__ id, size_t n)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	for (; i < n; i += (size_t)gridDim.x * blockDim.x) { w[i] = kEdtInf; id[i] = 0x7fffffff; }
}
__global__ void nnseed_seed_kernel(const float4* __restrict__ pts, int nslots, DtDesc dt, int32_t* __restrict__ w, int32_t* __restrict__ id)
{
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= nslots) return;
	const float4 p = pts[s];
	if (!(p.x < INFINITY)) return;                                            // padding slot
	const int x = (int)(((double)p.x - dt.xmin) * dt.scale + 0.5);          // the seeding rule of the distance transform (dt_seed_kernel)
	const int y = (int)(((double)p.y - dt.ymin) * dt.scale + 0.5);
	const int z = (int)(((double)p.z - dt.zmin) * dt.scale + 0.5);
	const int V = dt.V;
	if (x < 0 || x >= V || y < 0 || y >= V || z < 0 || z >= V) return;
	const size_t v = ((size_t)z * V + y) * V + x;
	w[v] = 0;
	atomicMin(&id[v], s);                                                     // lowest slot: the table is reproducible
}
__global__ __launch_bounds__(256) void nnseed_pass_kernel(int32_t* __restrict__ w, int32_t* __restrict__ id, int V, int axis)
{
	extern __shared__ int32_t row[];                                          // [V] distances, [V] ids
	int32_t* idrow = row + V;
	const int r = blockIdx.x;
	size_t base, stride;
	if (axis == 0) { base = (size_t)r * V; stride = 1; }
	else if (axis == 1) { int z = r / V, x = r - z * V; base = (size_t)z * V * V + x; stride = V; }
	else { base = r; stride = (size_t)V * V; }
	for (int i = threadIdx.x; i < V; i += blockDim.x) { row[i] = w[base + i * stride]; idrow[i] = id[base + i * stride]; }
	__syncthreads();
	for (int x = threadIdx.x; x < V; x += blockDim.x) {
		int best = kEdtInf, bi = x;
		for (int i = 0; i < V; i++) {
			const int d = x - i;
			const int v = d * d + row[i];
			if (v < best) { best = v; bi = i; }
		}
		w[base + x * stride] = best;
		id[base + x * stride] = idrow[bi];
	}
}
template <int LAYOUT>
__global__ void nnseed_finish_kernel(const int32_t* __restrict__ id, DtDesc dt, int32_t* __restrict__ out)
{
	const int V = dt.V;
	const size_t n = (size_t)V * V * V;
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	if (LAYOUT == 0) { out[i] = id[i]; return; }
	const int x = (int)(i % V), y = (int)((i / V) % V), z = (int)(i / ((size_t)V * V));
	const size_t b = ((size_t)(z >> 2) * dt.VB + (y >> 2)) * dt.VB + (x >> 2);
	out[b * 64 + (((z & 3) << 4) | ((y & 3) << 2) | (x & 3))] = id[i];
}
hipError_t launch_nn_seed_build(const float4* pts, int nslots, const DtDesc& dt, int32_t* work_d, int32_t* work_id, int32_t* out, hipStream_t stream)
{
	const int V = dt.V;
	const size_t n = (size_t)V * V * V;
	hipLaunchKernelGGL(nnseed_fill_kernel, dim3(4096), dim3(256), 0, stream, work_d, work_id, n);
	hipLaunchKernelGGL(nnseed_seed_kernel, dim3((nslots + 255) / 256), dim3(256), 0, stream, pts, nslots, dt, work_d, work_id);
	for (int axis = 0; axis < 3; axis++)
		hipLaunchKernelGGL(nnseed_pass_kernel, dim3(V * V), dim3(256), 2 * V * sizeof(int32_t), stream, work_d, work_id, V, axis);
	const dim3 grid((unsigned)((n + 255) / 256));
	if (dt.layout == 0) hipLaunchKernelGGL(nnseed_finish_kernel<0>, grid, dim3(256), 0, stream, work_id, dt, out);
	else hipLaunchKernelGGL(nnseed_finish_kernel<1>, grid, dim3(256), 0, stream, work_id, dt, out);
	return hipGetLastError();
}

// half-precision copy of a bricked grid, rounded toward zero (|half| <= |float|: lower bounds stay valid)
__global__ void dt_to_half_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, size_t n)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	for (; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = __half_as_ushort(__float2half_rz(in[i]));
}
hipError_t launch_dt_to_half(const float* bricked, void* out_half, size_t n, hipStream_t stream)
{
	hipLaunchKernelGGL(dt_to_half_kernel, dim3(4096), dim3(256), 0, stream, bricked, static_cast<unsigned short*>(out_half), n);
	return hipGetLastError();
}

hipError_t launch_dt_build(const float* model_xyz, int M, const DtDesc& dt, int32_t* work, float* out, hipStream_t stream)
{
	const int V = dt.V;
	const size_t n = (size_t)V * V * V;
	hipLaunchKernelGGL(dt_fill_kernel, dim3(4096), dim3(256), 0, stream, work, n);
	hipLaunchKernelGGL(dt_seed_kernel, dim3((M + 255) / 256), dim3(256), 0, stream, model_xyz, M, dt, work);
	for (int axis = 0; axis < 3; axis++)
		hipLaunchKernelGGL(dt_pass_kernel, dim3(V * V), dim3(256), V * sizeof(int32_t), stream, work, V, axis);
	dim3 grid((unsigned)((n + 255) / 256));
	if (dt.layout == 0) {
		// in-place is safe for the linear layout: element i is read and written by the same thread
		hipLaunchKernelGGL(dt_finish_kernel<0>, grid, dim3(256), 0, stream, work, dt, out);
	} else {
		hipLaunchKernelGGL(dt_finish_kernel<1>, grid, dim3(256), 0, stream, work, dt, out);
	}
	return hipGetLastError();
}

}  // namespace goicp
