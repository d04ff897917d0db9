// HIP kernels of the MI355X Go-ICP engine (gfx950 / CDNA4, wave64).  Compiled with
// -ffp-contract=off: every per-point value follows the reference CPU path's float/double operation
// order (no FMA contraction), so distances, clamps and squares are bit-identical to
// src/goicp/jly_goicp.cpp / jly_3ddt.cpp / jly_icp3d.hpp; only the order of the sums differs.
//
// Kernels
//   bounds_kernel      (a) BnB cube bounds: rotate + translate the source cloud, gather the 3-D
//                      Euclidean distance transform, subtract uncertainty radii, sum of squares.
//                      HBM/L2-bound gather, 16 B (ub pass) / 20 B (lb pass) algorithmic per point.
//   bounds_finalize    fixed-order sum of the per-chunk partials (deterministic, no float atomics)
//   icp_pass_kernel    (b) one ICP correspondence pass: transform, exact 1-NN in the implicit k-d
//                      tree (stackless traversal), pivoted centroid/covariance sums, wave64 reductions
//   icp_finalize       fixed-order double-precision sum of the per-wave partials
//   transform_kernel   in-place rigid transform apply
//   nn_query_kernel    k-d tree 1-NN operator for arbitrary queries
//   dt_*               exact Euclidean DT build (seed, three separable min-plus passes, sqrt/scale)
#include <hip/hip_runtime.h>
#include <climits>
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
__device__ __forceinline__ double wave_sum_d(double v)
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
template <int LAYOUT>
__device__ __forceinline__ float dt_fetch(const DtDesc& dt, int x, int y, int z)
{
	if (LAYOUT == 0) {
		return dt.grid[((size_t)z * dt.V + y) * dt.V + x];
	} else {
		size_t b = ((size_t)(z >> 2) * dt.VB + (y >> 2)) * dt.VB + (x >> 2);
		int in = ((z & 3) << 4) | ((y & 3) << 2) | (x & 3);
		return dt.grid[b * 64 + in];
	}
}

template <int LAYOUT>
__device__ __forceinline__ float dt_distance(const DtDesc& dt, float qx, float qy, float qz)
{
	// double index math exactly as the reference: (x - xMin)*scale + 0.5, truncate
	int x = (int)(((double)qx - dt.xmin) * dt.scale + 0.5);
	int y = (int)(((double)qy - dt.ymin) * dt.scale + 0.5);
	int z = (int)(((double)qz - dt.zmin) * dt.scale + 0.5);
	const int V = dt.V;
	if ((unsigned)x < (unsigned)V && (unsigned)y < (unsigned)V && (unsigned)z < (unsigned)V)
		return dt_fetch<LAYOUT>(dt, x, y, z);
	// outside the grid: clamp to the face and add the overshoot (in voxels) / scale
	float a = 0.f, b = 0.f, c = 0.f;
	if (x < 0) { a = (float)x; x = 0; } else if (x >= V) { a = (float)(x - V + 1); x = V - 1; }
	if (y < 0) { b = (float)y; y = 0; } else if (y >= V) { b = (float)(y - V + 1); y = V - 1; }
	if (z < 0) { c = (float)z; z = 0; } else if (z >= V) { c = (float)(z - V + 1); z = V - 1; }
	float r = __fsqrt_rn(a * a + b * b + c * c);
	return (float)((double)r / dt.scale + (double)dt_fetch<LAYOUT>(dt, x, y, z));
}

// ------------------------------------------------------------------------------------------------
// (a) cube bounds
// ------------------------------------------------------------------------------------------------
// grid: groups*chunks blocks of 256 threads.  A block owns kGroup consecutive cubes (the siblings
// of one BnB expansion: same rotation, neighbouring translations -> neighbouring DT voxels) and a
// contiguous chunk of the (Morton-sorted) source cloud; each point is loaded once (16 B) and
// reused for the 8 cubes.
template <int LAYOUT>
__global__ __launch_bounds__(kBoundsThreads) void bounds_kernel(
    const float4* __restrict__ src, int N, DtDesc dt, const Rot9* __restrict__ rots,
    const CubeRec* __restrict__ cubes, int B, int groups, int chunks, int chunk_pts,
    float* __restrict__ scratch, float* __restrict__ ub_out, float* __restrict__ lb_out)
{
	const int swz = xcd_remap(blockIdx.x, gridDim.x);
	const int chunk = swz / groups, group = swz - chunk * groups;
	const int c0 = group * kGroup;

	CubeRec cr[kGroup];
	bool uniform = true;
#pragma unroll
	for (int c = 0; c < kGroup; c++) {
		int ci = c0 + c < B ? c0 + c : B - 1;
		cr[c] = cubes[ci];
		uniform = uniform && (cr[c].rot == cr[0].rot);
	}
	const Rot9 R0 = rots[cr[0].rot];

	float ub[kGroup], lb[kGroup];
#pragma unroll
	for (int c = 0; c < kGroup; c++) { ub[c] = 0.f; lb[c] = 0.f; }

	const int p0 = chunk * chunk_pts;
	const int p1 = p0 + chunk_pts < N ? p0 + chunk_pts : N;
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
			float dis = m - cr[c].delta;         // :312
			if (dis > 0.f) lb[c] += dis * dis;
		}
	}

	// wave64 reduce, then 4 waves through LDS
	__shared__ float red[kBoundsThreads / 64][2 * kGroup];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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

static void bounds_shape(int B, int N, int* groups, int* chunks, int* chunk_pts)
{
	int g = (B + kGroup - 1) / kGroup;
	// aim for >= 8 blocks of 256 threads per CU (256 CUs) so that the gathers have 32 waves/CU
	const int target_blocks = 2048;
	int max_chunks = (N + kBoundsThreads - 1) / kBoundsThreads;
	int c = (target_blocks + g - 1) / g;
	if (c > max_chunks) c = max_chunks;
	if (c < 1) c = 1;
	int cp = (N + c - 1) / c;
	cp = (cp + kBoundsThreads - 1) / kBoundsThreads * kBoundsThreads;
	c = (N + cp - 1) / cp;
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

hipError_t launch_bounds(const float4* src, int N, const DtDesc& dt, const Rot9* rots, const CubeRec* cubes,
                         int B, float* scratch, float* ub, float* lb, hipStream_t stream)
{
	if (B <= 0 || N <= 0) return hipSuccess;
	int groups, chunks, chunk_pts;
	bounds_shape(B, N, &groups, &chunks, &chunk_pts);
	dim3 grid(groups * chunks), block(kBoundsThreads);
	if (dt.layout == 0)
		hipLaunchKernelGGL(bounds_kernel<0>, grid, block, 0, stream, src, N, dt, rots, cubes, B, groups, chunks, chunk_pts, scratch, ub, lb);
	else
		hipLaunchKernelGGL(bounds_kernel<1>, grid, block, 0, stream, src, N, dt, rots, cubes, B, groups, chunks, chunk_pts, scratch, ub, lb);
	if (chunks > 1) {
		int t = groups * 2 * kGroup;
		hipLaunchKernelGGL(bounds_finalize, dim3((t + 255) / 256), dim3(256), 0, stream, scratch, B, groups, chunks, ub, lb);
	}
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// (b) ICP: exact 1-NN in the implicit k-d tree, stackless (the parent is re-read on the way up:
// no per-thread stack, no scratch, no LDS -> ~40 VGPRs, full occupancy for a latency-bound walk)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pick(int dim, float x, float y, float z) { return dim == 0 ? x : (dim == 1 ? y : z); }

__device__ __forceinline__ void kd_nearest(const KdDesc& kd, float qx, float qy, float qz,
                                           float& best, int& bidx, float& mx, float& my, float& mz)
{
	const int L = kd.L;
	best = INFINITY; bidx = INT_MAX; mx = my = mz = 0.f;
	int node = 1;
	while (true) {
		while (node < L) {                       // descend, near side first
			const float2 nd = kd.nodes[node];
			float diff = pick(__float_as_int(nd.y), qx, qy, qz) - nd.x;
			node = 2 * node + (diff < 0.f ? 0 : 1);
		}
		const int leaf = node - L;
		const int s = kd.leaf_start[leaf], e = kd.leaf_start[leaf + 1];
		for (int k = s; k < e; k++) {
			const float4 p = kd.pts[k];
			// squared L2 in the adaptor's accumulation order (nanoflann_goicp.hpp L2_Simple_Adaptor)
			float d0 = qx - p.x, d1 = qy - p.y, d2 = qz - p.z;
			float d = d0 * d0;
			d += d1 * d1;
			d += d2 * d2;
			int id = __float_as_int(p.w);
			if (d < best || (d == best && id < bidx)) { best = d; bidx = id; mx = p.x; my = p.y; mz = p.z; }
		}
		bool done = false;
		while (true) {                           // ascend until an unvisited far side may hold a closer point
			if (node == 1) { done = true; break; }
			const int parent = node >> 1;
			const float2 nd = kd.nodes[parent];
			float diff = pick(__float_as_int(nd.y), qx, qy, qz) - nd.x;
			int nearc = 2 * parent + (diff < 0.f ? 0 : 1);
			if (node == nearc && diff * diff <= best) { node = nearc ^ 1; break; }
			node = parent;
		}
		if (done) break;
	}
}

constexpr int kIcpThreads = 64;   // one wavefront per block: reductions stay in-register

__global__ __launch_bounds__(kIcpThreads) void icp_pass_kernel(
    const float4* __restrict__ src, int N, Pose pose, KdDesc kd, float cqx, float cqy, float cqz,
    float cmx, float cmy, float cmz, float* __restrict__ partials)
{
	float acc[kIcpAcc];
#pragma unroll
	for (int k = 0; k < kIcpAcc; k++) acc[k] = 0.f;
	for (int i = blockIdx.x * kIcpThreads + threadIdx.x; i < N; i += gridDim.x * kIcpThreads) {
		const float4 p = src[i];
		// jly_icp3d.hpp:222-224
		float qx = pose.R[0] * p.x + pose.R[1] * p.y + pose.R[2] * p.z + pose.t[0];
		float qy = pose.R[3] * p.x + pose.R[4] * p.y + pose.R[5] * p.z + pose.t[1];
		float qz = pose.R[6] * p.x + pose.R[7] * p.y + pose.R[8] * p.z + pose.t[2];
		float d2, mx, my, mz; int id;
		kd_nearest(kd, qx, qy, qz, d2, id, mx, my, mz);
		float ax = qx - cqx, ay = qy - cqy, az = qz - cqz;   // pivots keep the covariance sums well conditioned
		float bx = mx - cmx, by = my - cmy, bz = mz - cmz;
		acc[0] += ax; acc[1] += ay; acc[2] += az;
		acc[3] += bx; acc[4] += by; acc[5] += bz;
		acc[6] += ax * bx; acc[7] += ax * by; acc[8] += ax * bz;
		acc[9] += ay * bx; acc[10] += ay * by; acc[11] += ay * bz;
		acc[12] += az * bx; acc[13] += az * by; acc[14] += az * bz;
		acc[15] += d2;
	}
#pragma unroll
	for (int k = 0; k < kIcpAcc; k++) {
		float s = wave_sum(acc[k]);
		if (threadIdx.x == 0) partials[(size_t)blockIdx.x * kIcpAcc + k] = s;
	}
}

// 16 waves, wave k sums component k over all blocks in double, fixed order -> deterministic
__global__ __launch_bounds__(kIcpAcc * 64) void icp_finalize(const float* __restrict__ partials, int nblocks,
                                                              double* __restrict__ out16)
{
	const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
	double s = 0.0;
	for (int b = lane; b < nblocks; b += 64) s += (double)partials[(size_t)b * kIcpAcc + k];
	s = wave_sum_d(s);
	if (lane == 0) out16[k] = s;
}

int icp_blocks(int N)
{
	int b = (N + kIcpThreads - 1) / kIcpThreads;
	return b > 8192 ? 8192 : (b < 1 ? 1 : b);
}

hipError_t launch_icp_pass(const float4* src, int N, const Pose& pose, const KdDesc& kd, const float cq[3],
                           const float cm[3], float* partials, double* out16, hipStream_t stream)
{
	int nb = icp_blocks(N);
	hipLaunchKernelGGL(icp_pass_kernel, dim3(nb), dim3(kIcpThreads), 0, stream, src, N, pose, kd,
	                   cq[0], cq[1], cq[2], cm[0], cm[1], cm[2], partials);
	hipLaunchKernelGGL(icp_finalize, dim3(1), dim3(kIcpAcc * 64), 0, stream, partials, nb, out16);
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

__global__ void nn_query_kernel(const float* __restrict__ q, int n, KdDesc kd, int32_t* __restrict__ idx,
                                float* __restrict__ d2)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float best, mx, my, mz; int id;
	kd_nearest(kd, q[3 * i], q[3 * i + 1], q[3 * i + 2], best, id, mx, my, mz);
	idx[i] = id;
	d2[i] = best;
}

hipError_t launch_nn_query(const float* q, int n, const KdDesc& kd, int32_t* idx, float* d2, hipStream_t stream)
{
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(nn_query_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, q, n, kd, idx, d2);
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
