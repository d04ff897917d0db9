// Device-side construction of the 64-ary box hierarchy: Morton codes -> rocPRIM radix
// sort -> balanced runs of <= 16 points per leaf -> bottom-up box unions.  Same KdDesc layout as the
// host median-split build (kdtree.cpp); the walk is exact for any hierarchy, only its speed depends
// on how tight the boxes are (median splits are tighter; this build is for clouds of 10^5..10^6+
// points, where the host build costs tenths of a second).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <climits>
#include <cmath>

#include "device.hpp"

namespace goicp {

__device__ __forceinline__ unsigned spread10(unsigned x)
{
	x &= 0x3ffu;
	x = (x ^ (x << 16)) & 0xff0000ffu;
	x = (x ^ (x << 8)) & 0x0300f00fu;
	x = (x ^ (x << 4)) & 0x030c30c3u;
	x = (x ^ (x << 2)) & 0x09249249u;
	return x;
}

__global__ void kd_morton_kernel(const float* __restrict__ xyz, int M, float mnx, float mny, float mnz, float inv_ext,
                                 unsigned* __restrict__ keys, int* __restrict__ vals)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= M) return;
	const float fx = (xyz[3 * i] - mnx) * inv_ext, fy = (xyz[3 * i + 1] - mny) * inv_ext, fz = (xyz[3 * i + 2] - mnz) * inv_ext;
	const unsigned qx = (unsigned)fminf(1023.f, fmaxf(0.f, fx * 1024.f));
	const unsigned qy = (unsigned)fminf(1023.f, fmaxf(0.f, fy * 1024.f));
	const unsigned qz = (unsigned)fminf(1023.f, fmaxf(0.f, fz * 1024.f));
	keys[i] = spread10(qx) | (spread10(qy) << 1) | (spread10(qz) << 2);
	vals[i] = i;
}

// one thread per leaf: leaf f owns the sorted points [f*M/L, (f+1)*M/L) (<= kLeafSlots of them)
__global__ void kd_leaves_kernel(const float* __restrict__ xyz, const int* __restrict__ order, int M, int L,
                                 float4* __restrict__ pts, float* __restrict__ last_level_boxes)
{
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= L) return;
	const long long a = (long long)f * M / L, b = (long long)(f + 1) * M / L;
	float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
	for (int s = 0; s < kLeafSlots; s++) {
		float4 p = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(INT_MAX));
		if (a + s < b) {
			const int id = order[a + s];
			p = make_float4(xyz[3 * id], xyz[3 * id + 1], xyz[3 * id + 2], __int_as_float(id));
			lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
			hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
		}
		pts[(size_t)f * kLeafSlots + s] = p;
	}
	float* rec = last_level_boxes + (size_t)(f >> 6) * 384;
	const int c = f & 63;
	for (int k = 0; k < 3; k++) { rec[64 * k + c] = lo[k]; rec[192 + 64 * k + c] = hi[k]; }
}

// child c of group g on `upper` = union of the 64 children of group 64g+c on `lower`
__global__ void kd_level_kernel(const float* __restrict__ lower, float* __restrict__ upper, int upper_children)
{
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= upper_children) return;
	const float* src = lower + (size_t)t * 384;
	float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
	for (int c = 0; c < 64; c++)
		for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], src[64 * k + c]); hi[k] = fmaxf(hi[k], src[192 + 64 * k + c]); }
	float* rec = upper + (size_t)(t >> 6) * 384;
	for (int k = 0; k < 3; k++) { rec[64 * k + (t & 63)] = lo[k]; rec[192 + 64 * k + (t & 63)] = hi[k]; }
}

// boxes[l] must hold 64^l * 384 floats, pts kLeafSlots * 64^K float4; d_xyz = the M target points on the device
hipError_t launch_kd_build(const float* d_xyz, int M, int K, const float mn[3], float ext, float* const boxes[kMaxLevels],
                           float4* pts, hipStream_t stream)
{
	const int L = 1 << (6 * K);
	unsigned *keys = nullptr, *keys2 = nullptr;
	int *vals = nullptr, *vals2 = nullptr;
	void* tmp = nullptr;
	size_t tmp_bytes = 0;
	hipError_t e;
	// every exit path frees the temporaries
	struct Cleanup {
		unsigned*& a; unsigned*& b; int*& c; int*& d; void*& t;
		~Cleanup() { hipFree(a); hipFree(b); hipFree(c); hipFree(d); hipFree(t); }
	} cleanup{keys, keys2, vals, vals2, tmp};
	if ((e = hipMalloc(&keys, sizeof(unsigned) * M)) != hipSuccess) return e;
	if ((e = hipMalloc(&keys2, sizeof(unsigned) * M)) != hipSuccess) return e;
	if ((e = hipMalloc(&vals, sizeof(int) * M)) != hipSuccess) return e;
	if ((e = hipMalloc(&vals2, sizeof(int) * M)) != hipSuccess) return e;
	hipLaunchKernelGGL(kd_morton_kernel, dim3((M + 255) / 256), dim3(256), 0, stream, d_xyz, M, mn[0], mn[1], mn[2],
	                   ext > 0.f ? 1.f / ext : 0.f, keys, vals);
	if ((e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys2, vals, vals2, (size_t)M, 0, 30, stream)) != hipSuccess) return e;
	if ((e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16)) != hipSuccess) return e;
	if ((e = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, vals, vals2, (size_t)M, 0, 30, stream)) != hipSuccess) return e;
	hipLaunchKernelGGL(kd_leaves_kernel, dim3((L + 255) / 256), dim3(256), 0, stream, d_xyz, vals2, M, L, pts, boxes[K - 1]);
	for (int l = K - 2; l >= 0; l--) {
		const int upper_children = 1 << (6 * (l + 1));
		hipLaunchKernelGGL(kd_level_kernel, dim3((upper_children + 255) / 256), dim3(256), 0, stream, boxes[l + 1], boxes[l], upper_children);
	}
	e = hipStreamSynchronize(stream);            // the temporaries are in use until here
	return e != hipSuccess ? e : hipGetLastError();
}

}  // namespace goicp
