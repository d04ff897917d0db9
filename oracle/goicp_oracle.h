/* TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
 *
 * Plain-C CPU restatement of the reference's CPU Go-ICP path (zjsun1017/CUDA-Go-ICP,
 * src/goicp/ *), written from its behaviour, used ONLY as the checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
 * (cuda-go-icp_amd/csrc, libgoicp_mi355.so) never includes, links or calls anything here.
 *
 * Parity status: PINNED.  Every function below is checked by tests/test_oracle_vs_golden.py
 * against fixtures under tests/golden/ that were produced by the real reference code compiled
 * in the build container (oracle/ref_harness.cpp + oracle/Makefile -> oracle/_ref/).
 *
 * Known, documented deviations from the reference (all inside the stated tolerances):
 *  - DT *build*: exact Euclidean DT of the same seed grid (Meijster, integer arithmetic) instead of
 *    the reference's 8-pass vector propagation (jly_3ddt.cpp:710-742), which over-estimates
 *    0.005 % of the voxels by <= 0.334 voxel.  Lookup semantics are exact.
 *  - sums run in index order; the reference first permutes minDis with intro_select
 *    (jly_goicp.cpp:298), so the float sums may differ in the last bits.
 *  - 3x3 SVD is a double-precision Jacobi instead of float Numerical-Recipes svdcmp
 *    (matrix.cpp:602-830); R_ = V diag(1,1,det) U^T is sign/ordering invariant.
 */
#ifndef GOICP_ORACLE_H
#define GOICP_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_dt {
	int V;                       /* grid side (reference: 300, jly_goicp.cpp:56) */
	double scale;                /* voxels per unit (jly_3ddt.cpp:923) */
	double xmin, ymin, zmin;     /* cube origin after expand+cubify (jly_3ddt.cpp:905-921) */
	float* grid;                 /* V^3 floats, [z][y][x], x fastest (jly_3ddt.h:53-79) */
	int owns;
} orc_dt;

/* jly_3ddt.cpp:889-923 */
void  orc_dt_geometry(const float* model_xyz, int M, int V, double expand, orc_dt* dt);
/* jly_3ddt.cpp:889-979 (seeding exact, EDT exact instead of propagated) */
int   orc_dt_build(const float* model_xyz, int M, int V, double expand, orc_dt* dt);
/* wrap an existing grid (e.g. one downloaded from the device) */
void  orc_dt_wrap(orc_dt* dt, int V, double scale, double xmin, double ymin, double zmin, float* grid);
void  orc_dt_free(orc_dt* dt);
/* jly_3ddt.cpp:981-1026 */
float orc_dt_distance(const orc_dt* dt, double x, double y, double z);
/* seeds only: 1 where a model point rounds to the voxel (jly_3ddt.cpp:952-965); returns #distinct seeds */
int   orc_dt_seed(const orc_dt* geom, const float* model_xyz, int M, unsigned char* seed);

/* jly_goicp.cpp:139-160: norm[N], rho[20*N] (level-major) */
void  orc_rot_radii(const float* data_xyz, int N, float* norm, float* rho);
/* the per-level coefficient 2*sin(min(sqrt3*sigma_l, pi)/2) as a float */
float orc_rot_coeff(int level);
/* jly_goicp.cpp:449-467 */
void  orc_rodrigues(float ax, float ay, float az, float R[9]);
/* jly_goicp.cpp:470-476 */
void  orc_rotate(const float R[9], const float* data_xyz, int N, float* out_xyz);

/* jly_goicp.cpp:262-315 for one child cube with centre (tx,ty,tz) and width w_child */
void  orc_cube_bound(const orc_dt* dt, const float* prot_xyz, int N, const float* rho_or_null,
                     float tx, float ty, float tz, float w_child, float* ub, float* lb);
/* trimmed form (trimFraction > 0, jly_goicp.cpp:293-315): only the `inliers` smallest residuals are summed */
void  orc_cube_bound_trim(const orc_dt* dt, const float* prot_xyz, int N, const float* rho_or_null,
                          float tx, float ty, float tz, float w_child, int inliers, float* ub, float* lb);
float orc_inner_bnb_trim(const orc_dt* dt, const float* prot_xyz, int N, const float* rho_or_null, int inliers,
                         float incumbent, float sse_thresh, const float root[4],
                         float best_node[4], long long* pops, long long* cubes);
/* same arithmetic per point, OpenMP over points (sum order differs); for the all-core CPU baseline */
void  orc_cube_bound_omp(const orc_dt* dt, const float* prot_xyz, int N, const float* rho_or_null,
                         float tx, float ty, float tz, float w_child, float* ub, float* lb);
/* B cubes {cx,cy,cz,w_child}; parallel != 0: OpenMP over cubes (identical results) */
void  orc_cube_bounds_batch(const orc_dt* dt, const float* prot_xyz, int N, const float* rho_or_null,
                            const float* cubes4, int B, float* ub, float* lb, int parallel);
/* sum_i Distance(R p_i + t)^2 (jly_goicp.cpp:100-129 with trimFraction 0) */
float orc_dt_sse(const orc_dt* dt, const float* data_xyz, int N, const float R[9], const float t[3]);

/* jly_goicp.cpp:227-340.  root = {x,y,z,w} corner+width.  best_node (corner+width) is written only
 * when an improvement over `incumbent` was found (as the reference).  Returns optErrorT. */
float orc_inner_bnb(const orc_dt* dt, const float* prot_xyz, int N, const float* rho_or_null,
                    float incumbent, float sse_thresh, const float root[4],
                    float best_node[4], long long* pops, long long* cubes);

/* exact 1-NN k-d tree over the model (semantics of nanoflann_goicp.hpp:1137-1184: exact, squared L2) */
typedef struct orc_kd orc_kd;
orc_kd* orc_kd_build(const float* model_xyz, int M);
void    orc_kd_free(orc_kd* kd);
void    orc_kd_nn(const orc_kd* kd, const float q[3], int* index, float* dist_sq);
/* brute force, lowest index wins ties */
void    orc_nn_brute(const float* model_xyz, int M, const float q[3], int* index, float* dist_sq);

/* jly_icp3d.hpp:268-285: H (row-major 3x3) -> R_ = V diag(1,1,det(V U^T)) U^T */
void  orc_kabsch_rotation(const float H[9], float R[9]);
/* jly_icp3d.hpp:181-295 (trim_fraction 0, do_trim true => correspondences sorted by distance).
 * R,t in/out.  Returns err_new of the last executed iteration; *iters = executed loop bodies. */
float orc_icp_run(const orc_kd* kd, const float* model_xyz, const float* data_xyz, int N,
                  float R[9], float t[3], int max_iter, float err_diff, int* iters);

/* trimmed forms (see goicp_oracle.c for the one documented deviation: means over num, not n) */
float orc_dt_sse_trim(const orc_dt* dt, const float* data_xyz, int N, const float R[9], const float t[3], int inliers);
float orc_icp_run_trim(const orc_kd* kd, const float* model_xyz, const float* data_xyz, int N, int inliers,
                       float R[9], float t[3], int max_iter, float err_diff, int* iters);

typedef struct orc_result {
	float R[9], t[3], sse;
	long long rot_pops, trans_pops, cubes, inner_calls, icp_runs, icp_iters;
} orc_result;
/* jly_goicp.cpp:342-585 (Initialize + OuterBnB).  data/model are already resized. */
int   orc_register(const orc_dt* dt, const float* model_xyz, int M, const float* data_xyz, int N,
                   float mse_thresh, orc_result* out);

int   orc_register_trim(const orc_dt* dt, const float* model_xyz, int M, const float* data_xyz, int N,
                        float mse_thresh, float trim_fraction, orc_result* out);

#ifdef __cplusplus
}
#endif
#endif
