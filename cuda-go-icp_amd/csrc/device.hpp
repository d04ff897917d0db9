// Device layer of the MI355X Go-ICP engine: plain structs + launch wrappers (implemented in
// device.hip, compiled by hipcc for gfx950).  Host C++ (engine.cpp, goicp_api.cpp) only sees this.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstddef>
#include <cstdint>

namespace goicp {

// Distance transform of the target cloud, resident in HBM.
//   layout 0: linear  [z][y][x], x fastest                       (reference order, jly_3ddt.h:53-79)
//   layout 1: bricked 4x4x4 voxels per 256-B brick, bricks [bz][by][bx]; a surface patch touched
//             by one wavefront then spans ~4x fewer cache lines than in the linear layout
struct DtDesc {
	const float* grid;
	int V;          // voxels per side
	int VB;         // bricks per side = ceil(V/4)         (layout 1)
	int layout;
	double scale;   // voxels per unit                      (jly_3ddt.cpp:923)
	double xmin, ymin, zmin;
};

// One translation sub-cube to bound (the inner body of GoICP::InnerBnB, jly_goicp.cpp:262-315).
struct CubeRec {
	float tx, ty, tz;   // cube centre  (jly_goicp.cpp:271-273)
	float delta;        // translation uncertainty radius sqrt(3)/2*w (jly_goicp.cpp:263)
	float coeff;        // rotation uncertainty coefficient 2*sin(min(sqrt3*sigma_l,pi)/2), 0 for the ub pass
	int32_t rot;        // index into the rotation table
};
static_assert(sizeof(CubeRec) == 24, "CubeRec layout");

struct Rot9 { float r[9]; };   // row-major

// Implicit, left-balanced k-d tree over the target cloud ("flattened k-d tree").
// Heap indexing: root = 1, children of n are 2n and 2n+1, leaves are nodes [L, 2L).
struct KdDesc {
	const float2* nodes;      // [L]  (.x = split value, .y = split dim as int bits); entry 0 unused
	const float4* pts;        // [M]  leaf order; .w = original index (int bits)
	const int32_t* leaf_start; // [L+1]
	int L;                    // number of leaves (power of two)
	int M;
};

struct Pose { float R[9]; float t[3]; };

constexpr int kGroup = 8;          // cubes per workgroup pass (the 8 siblings of one BnB expansion)
constexpr int kBoundsThreads = 256;
constexpr int kIcpAcc = 16;        // sum(q-cq)[3] sum(m-cm)[3] sum((q-cq)(m-cm)^T)[9] sum(d^2)

// ---- bounds ---------------------------------------------------------------------------------
// scratch must hold groups*chunks*2*kGroup floats.  ub/lb: [B].
size_t bounds_scratch_floats(int B, int N, int* groups_out, int* chunks_out);
hipError_t launch_bounds(const float4* src, int N, const DtDesc& dt, const Rot9* rots, const CubeRec* cubes,
                         int B, float* scratch, float* ub, float* lb, hipStream_t stream);

// ---- ICP ------------------------------------------------------------------------------------
// One pass: q_i = R p_i + t, exact 1-NN in the k-d tree, pivoted sums -> out16 (double[16]).
// partials must hold icp_blocks(N)*kIcpAcc floats.
int icp_blocks(int N);
hipError_t launch_icp_pass(const float4* src, int N, const Pose& pose, const KdDesc& kd,
                           const float cq[3], const float cm[3], float* partials, double* out16,
                           hipStream_t stream);
// In-place p <- R p + t, norm recomputed (ICP::kdTreeGPUStep's kernTransform, icp_kernel.cu:138-144)
hipError_t launch_transform(float4* src, int N, const Pose& pose, hipStream_t stream);
// NN operator on arbitrary queries (kernKDSearchNearest, icp_kernel.cu:146-157)
hipError_t launch_nn_query(const float* q_xyz, int n, const KdDesc& kd, int32_t* idx, float* d2, hipStream_t stream);

// ---- distance transform build (DT3D::Build, jly_3ddt.cpp:889-979; exact EDT) -------------------
// work: V^3 int32 (linear).  out: V^3 floats in dt.layout (may alias work only for layout 0).
hipError_t launch_dt_build(const float* model_xyz, int M, const DtDesc& dt, int32_t* work, float* out,
                           hipStream_t stream);

}  // namespace goicp
