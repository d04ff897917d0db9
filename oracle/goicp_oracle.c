/* TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.  See goicp_oracle.h for scope, pinning and the
 * documented deviations.  Citations are file:line under the reference checkout (src/goicp/...).
 *
 * Build: oracle/Makefile (gcc -O3 -march=native -ffp-contract=off: no FMA contraction, so each
 * per-point float value is bit-identical to the reference's baseline-x86-64 build).
 */
#include "goicp_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* constants exactly as the reference spells them (jly_goicp.h:35-36, :80) */
#define ORC_PI 3.1415926536
#define ORC_SQRT3 1.732050808
#define ORC_MAXROTLEVEL 20

static int cmp_float_asc(const void* a, const void* b);

/* ------------------------------------------------------------------------------------------
 * Priority queue with the tie behaviour of libstdc++'s std::priority_queue (push_heap /
 * pop_heap = bottom-up hole percolation + sift-up), so that node visit order -- and with it the
 * node counters -- can follow the reference's (jly_goicp.h:44-72: smaller lb first, then larger w).
 * ------------------------------------------------------------------------------------------ */
typedef struct { float x, y, z, w, ub, lb; int l; } node_t;

static int node_less(const node_t* a, const node_t* b) /* "a has lower priority than b" */
{
	if (a->lb != b->lb) return a->lb > b->lb;
	return a->w < b->w;
}

typedef struct { node_t* v; size_t n, cap; } heap_t;

static void heap_init(heap_t* h) { h->v = NULL; h->n = h->cap = 0; }
static void heap_free(heap_t* h) { free(h->v); h->v = NULL; h->n = h->cap = 0; }

static void heap_sift_up(node_t* v, size_t hole, size_t top, node_t val)
{
	while (hole > top) {
		size_t parent = (hole - 1) / 2;
		if (!node_less(&v[parent], &val)) break;
		v[hole] = v[parent];
		hole = parent;
	}
	v[hole] = val;
}

static void heap_push(heap_t* h, node_t val)
{
	if (h->n == h->cap) {
		h->cap = h->cap ? 2 * h->cap : 64;
		h->v = (node_t*)realloc(h->v, h->cap * sizeof(node_t));
	}
	h->n++;
	heap_sift_up(h->v, h->n - 1, 0, val);
}

static node_t heap_pop(heap_t* h)
{
	node_t top = h->v[0];
	node_t last = h->v[h->n - 1];
	size_t len = --h->n;
	if (len == 0) return top;
	size_t hole = 0, child = 0;
	while (child < (len - 1) / 2) {
		child = 2 * (child + 1);
		if (node_less(&h->v[child], &h->v[child - 1])) child--;
		h->v[hole] = h->v[child];
		hole = child;
	}
	if ((len & 1) == 0 && child == (len - 2) / 2) {
		child = 2 * (child + 1);
		h->v[hole] = h->v[child - 1];
		hole = child - 1;
	}
	heap_sift_up(h->v, hole, 0, last);
	return top;
}

/* ------------------------------------------------------------------------------------------
 * Distance transform
 * ------------------------------------------------------------------------------------------ */
void orc_dt_geometry(const float* m, int M, int V, double expand, orc_dt* dt)
{
	/* jly_3ddt.cpp:891-923, all in double; the model floats are widened first (jly_goicp.cpp:80-85) */
	double xMin = m[0], xMax = m[0], yMin = m[1], yMax = m[1], zMin = m[2], zMax = m[2];
	for (int i = 1; i < M; i++) {
		double x = m[3 * i], y = m[3 * i + 1], z = m[3 * i + 2];
		if (xMin > x) xMin = x;
		if (xMax < x) xMax = x;
		if (yMin > y) yMin = y;
		if (yMax < y) yMax = y;
		if (zMin > z) zMin = z;
		if (zMax < z) zMax = z;
	}
	double xc = (xMin + xMax) / 2, yc = (yMin + yMax) / 2, zc = (zMin + zMax) / 2;
	xMin = xc - expand * (xMax - xc); xMax = xc + expand * (xMax - xc);
	yMin = yc - expand * (yMax - yc); yMax = yc + expand * (yMax - yc);
	zMin = zc - expand * (zMax - zc); zMax = zc + expand * (zMax - zc);
	double side = xMax - xMin > yMax - yMin ? xMax - xMin : yMax - yMin;
	side = side > zMax - zMin ? side : zMax - zMin;
	dt->V = V;
	dt->xmin = xc - side / 2;
	dt->ymin = yc - side / 2;
	dt->zmin = zc - side / 2;
	dt->scale = V / side;
}

static inline int voxel_of(double x, double xmin, double scale)
{
	/* ROUND(x) = int(x + 0.5), C truncation toward zero (jly_3ddt.cpp:24, :954, :984) */
	return (int)((x - xmin) * scale + 0.5);
}

int orc_dt_seed(const orc_dt* g, const float* m, int M, unsigned char* seed)
{
	const int V = g->V;
	int n = 0;
	memset(seed, 0, (size_t)V * V * V);
	for (int i = 0; i < M; i++) {
		int x = voxel_of((double)m[3 * i], g->xmin, g->scale);
		int y = voxel_of((double)m[3 * i + 1], g->ymin, g->scale);
		int z = voxel_of((double)m[3 * i + 2], g->zmin, g->scale);
		if (x < 0 || x >= V || y < 0 || y >= V || z < 0 || z >= V) continue; /* jly_3ddt.cpp:958 */
		size_t k = ((size_t)z * V + y) * V + x;
		if (!seed[k]) { seed[k] = 1; n++; }
	}
	return n;
}

#define EDT_INF (1 << 28)

/* 1-D lower envelope of parabolas (Meijster et al. 2000), integers only:
 * out[x] = min_i (x-i)^2 + g[i] */
static void edt_1d(const int32_t* g, int n, int32_t* out, int* s, int* t)
{
	int q = 0;
	s[0] = 0; t[0] = 0;
	for (int u = 1; u < n; u++) {
		while (q >= 0) {
			int64_t a = (int64_t)(t[q] - s[q]) * (t[q] - s[q]) + g[s[q]];
			int64_t b = (int64_t)(t[q] - u) * (t[q] - u) + g[u];
			if (a > b) q--; else break;
		}
		if (q < 0) { q = 0; s[0] = u; }
		else {
			int i = s[q];
			int64_t num = (int64_t)u * u - (int64_t)i * i + (int64_t)g[u] - (int64_t)g[i];
			int64_t den = 2 * (int64_t)(u - i);
			int64_t fl = num / den;
			if ((num % den != 0) && ((num < 0) != (den < 0))) fl--;
			int64_t w = 1 + fl;
			if (w < n) { q++; s[q] = u; t[q] = (int)(w < 0 ? 0 : w); }
		}
	}
	for (int u = n - 1; u >= 0; u--) {
		int64_t v = (int64_t)(u - s[q]) * (u - s[q]) + g[s[q]];
		out[u] = (int32_t)(v > EDT_INF ? EDT_INF : v);
		if (u == t[q]) q--;
	}
}

int orc_dt_build(const float* m, int M, int V, double expand, orc_dt* dt)
{
	orc_dt_geometry(m, M, V, expand, dt);
	size_t nv = (size_t)V * V * V;
	unsigned char* seed = (unsigned char*)malloc(nv);
	int32_t* d2 = (int32_t*)malloc(nv * sizeof(int32_t));
	dt->grid = (float*)malloc(nv * sizeof(float));
	dt->owns = 1;
	if (!seed || !d2 || !dt->grid) return -1;
	orc_dt_seed(dt, m, M, seed);
	for (size_t k = 0; k < nv; k++) d2[k] = seed[k] ? 0 : EDT_INF;
	free(seed);
#pragma omp parallel
	{
		int32_t* in = (int32_t*)malloc(V * sizeof(int32_t));
		int32_t* out = (int32_t*)malloc(V * sizeof(int32_t));
		int* s = (int*)malloc(V * sizeof(int));
		int* t = (int*)malloc(V * sizeof(int));
		/* along x */
#pragma omp for schedule(static)
		for (int r = 0; r < V * V; r++) {
			int32_t* row = d2 + (size_t)r * V;
			edt_1d(row, V, out, s, t);
			memcpy(row, out, V * sizeof(int32_t));
		}
		/* along y */
#pragma omp for schedule(static)
		for (int r = 0; r < V * V; r++) {
			int z = r / V, x = r % V;
			int32_t* base = d2 + (size_t)z * V * V + x;
			for (int y = 0; y < V; y++) in[y] = base[(size_t)y * V];
			edt_1d(in, V, out, s, t);
			for (int y = 0; y < V; y++) base[(size_t)y * V] = out[y];
		}
		/* along z */
#pragma omp for schedule(static)
		for (int r = 0; r < V * V; r++) {
			int32_t* base = d2 + r;
			for (int z = 0; z < V; z++) in[z] = base[(size_t)z * V * V];
			edt_1d(in, V, out, s, t);
			for (int z = 0; z < V; z++) base[(size_t)z * V * V] = out[z];
		}
		free(in); free(out); free(s); free(t);
	}
	/* jly_3ddt.cpp:967-978: distance (a float holding sqrt(v^2+h^2+d^2)) / scale, clamped >= 0 */
	const double scale = dt->scale;
#pragma omp parallel for schedule(static)
	for (size_t k = 0; k < nv; k++) {
		float dist = (float)sqrt((double)d2[k]);
		float v = (float)((double)dist / scale);
		dt->grid[k] = v < 0 ? 0 : v;
	}
	free(d2);
	return 0;
}

void orc_dt_wrap(orc_dt* dt, int V, double scale, double xmin, double ymin, double zmin, float* grid)
{
	dt->V = V; dt->scale = scale; dt->xmin = xmin; dt->ymin = ymin; dt->zmin = zmin;
	dt->grid = grid; dt->owns = 0;
}

void orc_dt_free(orc_dt* dt)
{
	if (dt->owns) free(dt->grid);
	dt->grid = NULL;
}

float orc_dt_distance(const orc_dt* dt, double _x, double _y, double _z)
{
	/* jly_3ddt.cpp:981-1026 */
	const int V = dt->V;
	int x = voxel_of(_x, dt->xmin, dt->scale);
	int y = voxel_of(_y, dt->ymin, dt->scale);
	int z = voxel_of(_z, dt->zmin, dt->scale);
	if (x > -1 && x < V && y > -1 && y < V && z > -1 && z < V)
		return dt->grid[((size_t)z * V + y) * V + x];
	float a = 0, b = 0, c = 0;
	if (x < 0) { a = (float)x; x = 0; } else if (x >= V) { a = (float)(x - V + 1); x = V - 1; }
	if (y < 0) { b = (float)y; y = 0; } else if (y >= V) { b = (float)(y - V + 1); y = V - 1; }
	if (z < 0) { c = (float)z; z = 0; } else if (z >= V) { c = (float)(z - V + 1); z = V - 1; }
	/* sqrt of a float expression -> float overload; then double divide and double add, float return */
	float r = sqrtf(a * a + b * b + c * c);
	return (float)((double)r / dt->scale + (double)dt->grid[((size_t)z * V + y) * V + x]);
}

/* ------------------------------------------------------------------------------------------
 * Rotation helpers
 * ------------------------------------------------------------------------------------------ */
float orc_rot_coeff(int level)
{
	/* jly_goicp.cpp:153-159: sigma = w0/2^l/2 (double -> float), maxAngle = SQRT3*sigma (-> float),
	 * clamp to PI, coefficient 2*sinf(maxAngle/2) */
	float w0 = (float)(2 * ORC_PI);
	float sigma = (float)((double)w0 / pow(2.0, level) / 2.0);
	float maxAngle = (float)(ORC_SQRT3 * (double)sigma);
	if ((double)maxAngle > ORC_PI) maxAngle = (float)ORC_PI;
	return 2 * sinf(maxAngle / 2);
}

void orc_rot_radii(const float* d, int N, float* norm, float* rho)
{
	for (int i = 0; i < N; i++) {
		float x = d[3 * i], y = d[3 * i + 1], z = d[3 * i + 2];
		norm[i] = sqrtf(x * x + y * y + z * z);            /* jly_goicp.cpp:145 */
	}
	for (int l = 0; l < ORC_MAXROTLEVEL; l++) {
		float c = orc_rot_coeff(l);
		for (int j = 0; j < N; j++) rho[(size_t)l * N + j] = c * norm[j];  /* :159 */
	}
}

void orc_rodrigues(float ax, float ay, float az, float R[9])
{
	/* jly_goicp.cpp:449-467: angle-axis -> matrix, float, same operation order */
	float theta = sqrtf(ax * ax + ay * ay + az * az);
	if (!(theta > 0)) {
		for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.f : 0.f;
		return;
	}
	float ux = ax / theta, uy = ay / theta, uz = az / theta;
	float c = cosf(theta), omc = 1 - c, sn = sinf(theta);
	float xy = ux * uy * omc, xz = ux * uz * omc, yz = uy * uz * omc;
	float zs = uz * sn, ys = uy * sn, xs = ux * sn;
	R[0] = c + ux * ux * omc; R[1] = xy - zs;           R[2] = xz + ys;
	R[3] = xy + zs;           R[4] = c + uy * uy * omc; R[5] = yz - xs;
	R[6] = xz - ys;           R[7] = yz + xs;           R[8] = c + uz * uz * omc;
}

void orc_rotate(const float R[9], const float* d, int N, float* o)
{
	for (int i = 0; i < N; i++) {
		float x = d[3 * i], y = d[3 * i + 1], z = d[3 * i + 2];
		o[3 * i]     = R[0] * x + R[1] * y + R[2] * z;
		o[3 * i + 1] = R[3] * x + R[4] * y + R[5] * z;
		o[3 * i + 2] = R[6] * x + R[7] * y + R[8] * z;
	}
}

/* ------------------------------------------------------------------------------------------
 * Cube bounds and inner (translation) BnB
 * ------------------------------------------------------------------------------------------ */
static int cmp_float_asc(const void* a, const void* b)
{
	float fa = *(const float*)a, fb = *(const float*)b;
	return (fa > fb) - (fa < fb);
}

void orc_cube_bound_trim(const orc_dt* dt, const float* p, int N, const float* rho,
                         float tx, float ty, float tz, float w_child, int inliers, float* ub_out, float* lb_out)
{
	/* jly_goicp.cpp:276-315 with trimming: the inlierNum smallest residuals (intro_select, :298) enter the sums.
	 * Any exact selection gives the same multiset; sorted order is used for the sums. */
	float maxTransDis = (float)(ORC_SQRT3 / 2.0 * (double)w_child);
	float* m = (float*)malloc(sizeof(float) * N);
	for (int i = 0; i < N; i++) {
		float v = orc_dt_distance(dt, (double)(p[3 * i] + tx), (double)(p[3 * i + 1] + ty), (double)(p[3 * i + 2] + tz));
		if (rho) v -= rho[i];
		if (v < 0) v = 0;
		m[i] = v;
	}
	if (inliers < N) qsort(m, N, sizeof(float), cmp_float_asc);
	float ub = 0, lb = 0;
	for (int i = 0; i < inliers; i++) {
		ub += m[i] * m[i];
		float dis = m[i] - maxTransDis;
		if (dis > 0) lb += dis * dis;
	}
	free(m);
	*ub_out = ub;
	*lb_out = lb;
}

void orc_cube_bound(const orc_dt* dt, const float* p, int N, const float* rho,
                    float tx, float ty, float tz, float w_child, float* ub_out, float* lb_out)
{
	/* jly_goicp.cpp:263 (maxTransDis), :276-315 */
	float maxTransDis = (float)(ORC_SQRT3 / 2.0 * (double)w_child);
	float ub = 0, lb = 0;
	for (int i = 0; i < N; i++) {
		float m = orc_dt_distance(dt, (double)(p[3 * i] + tx), (double)(p[3 * i + 1] + ty), (double)(p[3 * i + 2] + tz));
		if (rho) m -= rho[i];
		if (m < 0) m = 0;
		ub += m * m;
		float dis = m - maxTransDis;
		if (dis > 0) lb += dis * dis;
	}
	*ub_out = ub;
	*lb_out = lb;
}

void orc_cube_bound_omp(const orc_dt* dt, const float* p, int N, const float* rho,
                        float tx, float ty, float tz, float w_child, float* ub_out, float* lb_out)
{
	float maxTransDis = (float)(ORC_SQRT3 / 2.0 * (double)w_child);
	float ub = 0, lb = 0;
#pragma omp parallel for reduction(+ : ub, lb) schedule(static)
	for (int i = 0; i < N; i++) {
		float m = orc_dt_distance(dt, (double)(p[3 * i] + tx), (double)(p[3 * i + 1] + ty), (double)(p[3 * i + 2] + tz));
		if (rho) m -= rho[i];
		if (m < 0) m = 0;
		ub += m * m;
		float dis = m - maxTransDis;
		if (dis > 0) lb += dis * dis;
	}
	*ub_out = ub;
	*lb_out = lb;
}

void orc_cube_bounds_batch(const orc_dt* dt, const float* p, int N, const float* rho, const float* cubes4, int B,
                           float* ub, float* lb, int parallel)
{
	/* the same per-cube arithmetic and sum order as orc_cube_bound; cubes are independent, so the
	 * all-core CPU baseline parallelises over cubes */
#pragma omp parallel for schedule(dynamic, 4) if (parallel)
	for (int b = 0; b < B; b++)
		orc_cube_bound(dt, p, N, rho, cubes4[4 * b], cubes4[4 * b + 1], cubes4[4 * b + 2], cubes4[4 * b + 3], &ub[b], &lb[b]);
}

float orc_dt_sse_trim(const orc_dt* dt, const float* d, int N, const float R[9], const float t[3], int inliers)
{
	/* jly_goicp.cpp:100-129 with trimming: the inlierNum smallest distances */
	float* m = (float*)malloc(sizeof(float) * N);
	for (int i = 0; i < N; i++) {
		float x = d[3 * i], y = d[3 * i + 1], z = d[3 * i + 2];
		float qx = R[0] * x + R[1] * y + R[2] * z + t[0];
		float qy = R[3] * x + R[4] * y + R[5] * z + t[1];
		float qz = R[6] * x + R[7] * y + R[8] * z + t[2];
		m[i] = orc_dt_distance(dt, qx, qy, qz);
	}
	if (inliers < N) qsort(m, N, sizeof(float), cmp_float_asc);
	float error = 0;
	for (int i = 0; i < inliers; i++) error += m[i] * m[i];
	free(m);
	return error;
}

float orc_dt_sse(const orc_dt* dt, const float* d, int N, const float R[9], const float t[3])
{
	/* jly_goicp.cpp:100-129 */
	float error = 0;
	for (int i = 0; i < N; i++) {
		float x = d[3 * i], y = d[3 * i + 1], z = d[3 * i + 2];
		float qx = R[0] * x + R[1] * y + R[2] * z + t[0];
		float qy = R[3] * x + R[4] * y + R[5] * z + t[1];
		float qz = R[6] * x + R[7] * y + R[8] * z + t[2];
		float dis = orc_dt_distance(dt, qx, qy, qz);
		error += dis * dis;
	}
	return error;
}

float orc_inner_bnb_trim(const orc_dt* dt, const float* p, int N, const float* rho, int inliers, float incumbent, float sse_thresh, const float root[4], float best_node[4], long long* pops, long long* cubes);
float orc_inner_bnb(const orc_dt* dt, const float* p, int N, const float* rho,
                    float incumbent, float sse_thresh, const float root[4],
                    float best_node[4], long long* pops, long long* cubes)
{
	return orc_inner_bnb_trim(dt, p, N, rho, N, incumbent, sse_thresh, root, best_node, pops, cubes);
}

float orc_inner_bnb_trim(const orc_dt* dt, const float* p, int N, const float* rho, int inliers,
                         float incumbent, float sse_thresh, const float root[4],
                         float best_node[4], long long* pops, long long* cubes)
{
	/* jly_goicp.cpp:227-340 */
	heap_t q;
	heap_init(&q);
	float optErrorT = incumbent;
	node_t r = { root[0], root[1], root[2], root[3], 0, 0, 0 };
	heap_push(&q, r);
	long long npop = 0, ncube = 0;
	while (q.n) {
		node_t parent = heap_pop(&q);
		npop++;
		if (optErrorT - parent.lb < sse_thresh) break;
		node_t c;
		c.l = 0;
		c.w = parent.w / 2;
		for (int j = 0; j < 8; j++) {
			c.x = parent.x + (j & 1) * c.w;
			c.y = parent.y + (j >> 1 & 1) * c.w;
			c.z = parent.z + (j >> 2 & 1) * c.w;
			float tx = c.x + c.w / 2, ty = c.y + c.w / 2, tz = c.z + c.w / 2;
			float ub, lb;
			if (inliers < N) orc_cube_bound_trim(dt, p, N, rho, tx, ty, tz, c.w, inliers, &ub, &lb);
			else orc_cube_bound(dt, p, N, rho, tx, ty, tz, c.w, &ub, &lb);
			ncube++;
			if (ub < optErrorT) {
				optErrorT = ub;
				if (best_node) { best_node[0] = c.x; best_node[1] = c.y; best_node[2] = c.z; best_node[3] = c.w; }
			}
			if (lb >= optErrorT) continue;
			c.ub = ub; c.lb = lb;
			heap_push(&q, c);
		}
	}
	heap_free(&q);
	if (pops) *pops += npop;
	if (cubes) *cubes += ncube;
	return optErrorT;
}

/* ------------------------------------------------------------------------------------------
 * Exact 1-NN k-d tree over the model (own layout; semantics = exact NN, squared L2, float)
 * ------------------------------------------------------------------------------------------ */
typedef struct { int leaf, lo, hi, dim, left, right; float split; } kdnode_t;
struct orc_kd { const float* pts; int M; int* idx; kdnode_t* nodes; int nnodes, cap; };

static const float* g_sort_pts;
static int g_sort_dim;
static int cmp_coord(const void* a, const void* b)
{
	float fa = g_sort_pts[3 * (*(const int*)a) + g_sort_dim], fb = g_sort_pts[3 * (*(const int*)b) + g_sort_dim];
	if (fa < fb) return -1;
	if (fa > fb) return 1;
	return (*(const int*)a) - (*(const int*)b);
}

static int kd_build_rec(orc_kd* kd, int lo, int hi)
{
	if (kd->nnodes == kd->cap) {
		kd->cap *= 2;
		kd->nodes = (kdnode_t*)realloc(kd->nodes, kd->cap * sizeof(kdnode_t));
	}
	int me = kd->nnodes++;
	if (hi - lo <= 10) {   /* leaf size 10 as jly_icp3d.hpp:151 */
		kd->nodes[me].leaf = 1; kd->nodes[me].lo = lo; kd->nodes[me].hi = hi;
		return me;
	}
	float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
	for (int i = lo; i < hi; i++)
		for (int a = 0; a < 3; a++) {
			float v = kd->pts[3 * kd->idx[i] + a];
			if (v < mn[a]) mn[a] = v;
			if (v > mx[a]) mx[a] = v;
		}
	int dim = 0;
	if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
	if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
	g_sort_pts = kd->pts; g_sort_dim = dim;
	qsort(kd->idx + lo, hi - lo, sizeof(int), cmp_coord);
	int mid = lo + (hi - lo) / 2;
	float split = kd->pts[3 * kd->idx[mid] + dim];
	int l = kd_build_rec(kd, lo, mid);
	int r = kd_build_rec(kd, mid, hi);
	kd->nodes[me].leaf = 0; kd->nodes[me].dim = dim; kd->nodes[me].split = split;
	kd->nodes[me].left = l; kd->nodes[me].right = r;
	return me;
}

orc_kd* orc_kd_build(const float* m, int M)
{
	orc_kd* kd = (orc_kd*)calloc(1, sizeof(orc_kd));
	kd->pts = m; kd->M = M;
	kd->idx = (int*)malloc(M * sizeof(int));
	for (int i = 0; i < M; i++) kd->idx[i] = i;
	kd->cap = 1024; kd->nodes = (kdnode_t*)malloc(kd->cap * sizeof(kdnode_t));
	kd_build_rec(kd, 0, M);
	return kd;
}

void orc_kd_free(orc_kd* kd)
{
	if (!kd) return;
	free(kd->idx); free(kd->nodes); free(kd);
}

static inline float dist_sq(const float* q, const float* p)
{
	/* L2_Simple_Adaptor accumulation order (nanoflann_goicp.hpp): ((dx^2)+dy^2)+dz^2 */
	float d0 = q[0] - p[0], d1 = q[1] - p[1], d2 = q[2] - p[2];
	float r = d0 * d0;
	r += d1 * d1;
	r += d2 * d2;
	return r;
}

static void kd_search(const orc_kd* kd, int n, const float* q, int* bi, float* bd)
{
	const kdnode_t* nd = &kd->nodes[n];
	if (nd->leaf) {
		for (int i = nd->lo; i < nd->hi; i++) {
			int id = kd->idx[i];
			float d = dist_sq(q, kd->pts + 3 * id);
			if (d < *bd || (d == *bd && id < *bi)) { *bd = d; *bi = id; }
		}
		return;
	}
	float diff = q[nd->dim] - nd->split;
	int near = diff < 0 ? nd->left : nd->right;
	int far = diff < 0 ? nd->right : nd->left;
	kd_search(kd, near, q, bi, bd);
	if (diff * diff <= *bd) kd_search(kd, far, q, bi, bd);
}

void orc_kd_nn(const orc_kd* kd, const float q[3], int* index, float* d2)
{
	int bi = 0x7fffffff; float bd = INFINITY;
	kd_search(kd, 0, q, &bi, &bd);
	*index = bi; *d2 = bd;
}

void orc_nn_brute(const float* m, int M, const float q[3], int* index, float* d2)
{
	int bi = -1; float bd = INFINITY;
	for (int i = 0; i < M; i++) {
		float d = dist_sq(q, m + 3 * i);
		if (d < bd) { bd = d; bi = i; }
	}
	*index = bi; *d2 = bd;
}

/* ------------------------------------------------------------------------------------------
 * 3x3 SVD (one-sided Jacobi, double) and the Kabsch rotation
 * ------------------------------------------------------------------------------------------ */
static void svd3(const double A[9], double U[9], double W[3], double Vm[9])
{
	/* one-sided Jacobi on columns of B = A*V; converges to B = U*diag(W) */
	double B[9], V[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
	memcpy(B, A, sizeof(B));
	for (int sweep = 0; sweep < 60; sweep++) {
		double off = 0;
		for (int p = 0; p < 2; p++)
			for (int q = p + 1; q < 3; q++) {
				double a = 0, b = 0, c = 0;
				for (int i = 0; i < 3; i++) {
					a += B[3 * i + p] * B[3 * i + p];
					b += B[3 * i + q] * B[3 * i + q];
					c += B[3 * i + p] * B[3 * i + q];
				}
				off += c * c;
				if (fabs(c) <= 1e-300 || fabs(c) <= 1e-17 * sqrt(a * b)) continue;
				double zeta = (b - a) / (2 * c);
				double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
				double cs = 1 / sqrt(1 + tt * tt), sn = cs * tt;
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
		double n = sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
		W[j] = n;
		for (int i = 0; i < 3; i++) U[3 * i + j] = n > 0 ? B[3 * i + j] / n : 0;
	}
	/* complete U for (near-)zero singular values so that it stays orthogonal */
	for (int j = 0; j < 3; j++) {
		if (W[j] > 1e-200) continue;
		int a = (j + 1) % 3, b = (j + 2) % 3;
		double ua[3] = { U[a], U[3 + a], U[6 + a] }, ub[3] = { U[b], U[3 + b], U[6 + b] };
		double na = sqrt(ua[0] * ua[0] + ua[1] * ua[1] + ua[2] * ua[2]);
		double nb = sqrt(ub[0] * ub[0] + ub[1] * ub[1] + ub[2] * ub[2]);
		if (na < 0.5 || nb < 0.5) continue; /* rank <= 1: leave as is */
		U[j] = ua[1] * ub[2] - ua[2] * ub[1];
		U[3 + j] = ua[2] * ub[0] - ua[0] * ub[2];
		U[6 + j] = ua[0] * ub[1] - ua[1] * ub[0];
	}
	memcpy(Vm, V, sizeof(V));
}

static double det3(const double M[9])
{
	return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

void orc_kabsch_rotation(const float Hf[9], float R[9])
{
	/* jly_icp3d.hpp:268-285: R_ = V * diag(1,1,det(V*U^T)) * U^T */
	double H[9], U[9], W[3], V[9], VUt[9];
	for (int i = 0; i < 9; i++) H[i] = Hf[i];
	svd3(H, U, W, V);
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) {
			double s = 0;
			for (int k = 0; k < 3; k++) s += V[3 * i + k] * U[3 * j + k];
			VUt[3 * i + j] = s;
		}
	double det = det3(VUt);
	/* the reference's svd sorts singular values in decreasing order (matrix.cpp:782-808), so its
	 * diag(1,1,det) acts on the SMALLEST singular direction */
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

/* ------------------------------------------------------------------------------------------
 * ICP (jly_icp3d.hpp:181-295), float arithmetic in the reference's order
 * ------------------------------------------------------------------------------------------ */
typedef struct { double dis; int id_data, id_model; } pointref_t;
static int cmp_pointref(const void* a, const void* b)
{
	return ((const pointref_t*)a)->dis > ((const pointref_t*)b)->dis ? 1 : -1; /* jly_icp3d.hpp:156-159 */
}

float orc_icp_run(const orc_kd* kd, const float* model, const float* data, int N,
                  float R[9], float t[3], int max_iter, float err_diff, int* iters_out)
{
	return orc_icp_run_trim(kd, model, data, N, N, R, t, max_iter, err_diff, iters_out);
}

float orc_icp_run_trim(const orc_kd* kd, const float* model, const float* data, int N, int inliers,
                       float R[9], float t[3], int max_iter, float err_diff, int* iters_out)
{
	/* num = the `inliers` nearest correspondences (jly_icp3d.hpp:191-199,236-252).  Deviation, documented:
	 * the means are divided by num; the reference divides by n (jly_icp3d.hpp:259-260), which is only
	 * correct for trim_fraction 0. */
	const int n = N, num = inliers;
	pointref_t* points = (pointref_t*)malloc(sizeof(pointref_t) * n);
	float* p_m = (float*)malloc(sizeof(float) * 3 * num);
	float* p_d = (float*)malloc(sizeof(float) * 3 * num);
	float mu_m[3] = { 0, 0, 0 }, mu_d[3] = { 0, 0, 0 }; /* NOT reset per iteration (:205-206, :244-263) */
	float err = -1, err_new = 0;
	int iter;
	for (iter = 0; iter < max_iter; iter++) {
		float r00 = R[0], r01 = R[1], r02 = R[2], r10 = R[3], r11 = R[4], r12 = R[5], r20 = R[6], r21 = R[7], r22 = R[8];
		float t0 = t[0], t1 = t[1], t2 = t[2];
		err_new = 0;
		for (int i = 0; i < n; i++) {
			const float* d = data + 3 * i;
			float q[3];
			q[0] = r00 * d[0] + r01 * d[1] + r02 * d[2] + t0;
			q[1] = r10 * d[0] + r11 * d[1] + r12 * d[2] + t1;
			q[2] = r20 * d[0] + r21 * d[1] + r22 * d[2] + t2;
			int id; float d2;
			orc_kd_nn(kd, q, &id, &d2);
			points[i].dis = d2; points[i].id_data = i; points[i].id_model = id;
		}
		qsort(points, n, sizeof(pointref_t), cmp_pointref); /* do_trim == true (:236-239) */
		for (int i = 0; i < num; i++) {
			const float* m = model + 3 * points[i].id_model;
			p_m[3 * i] = m[0]; mu_m[0] += m[0];
			p_m[3 * i + 1] = m[1]; mu_m[1] += m[1];
			p_m[3 * i + 2] = m[2]; mu_m[2] += m[2];
			const float* d = data + 3 * points[i].id_data;
			p_d[3 * i] = r00 * d[0] + r01 * d[1] + r02 * d[2] + t0; mu_d[0] += p_d[3 * i];
			p_d[3 * i + 1] = r10 * d[0] + r11 * d[1] + r12 * d[2] + t1; mu_d[1] += p_d[3 * i + 1];
			p_d[3 * i + 2] = r20 * d[0] + r21 * d[1] + r22 * d[2] + t2; mu_d[2] += p_d[3 * i + 2];
			err_new = (float)((double)err_new + points[i].dis);
		}
		if (err > 0 && err - err_new < err_diff * num) break;
		err = err_new;
		for (int a = 0; a < 3; a++) { mu_m[a] = mu_m[a] / (float)num; mu_d[a] = mu_d[a] / (float)num; }
		/* H = (p_d - mu_d)^T (p_m - mu_m), sequential float sums over the rows (matrix.cpp:296-299) */
		float H[9] = { 0 };
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) {
				float s = 0;
				for (int k = 0; k < num; k++) s += (p_d[3 * k + i] - mu_d[i]) * (p_m[3 * k + j] - mu_m[j]);
				H[3 * i + j] = s;
			}
		float R_[9];
		orc_kabsch_rotation(H, R_);
		/* t_ = mu_m^T - R_ * mu_d^T */
		float t_[3];
		for (int i = 0; i < 3; i++) {
			float s = 0;
			for (int k = 0; k < 3; k++) s += R_[3 * i + k] * mu_d[k];
			t_[i] = mu_m[i] - s;
		}
		/* R = R_*R ; t = R_*t + t_ */
		float Rn[9], tn[3];
		for (int i = 0; i < 3; i++) {
			for (int j = 0; j < 3; j++) {
				float s = 0;
				for (int k = 0; k < 3; k++) s += R_[3 * i + k] * R[3 * k + j];
				Rn[3 * i + j] = s;
			}
			float s = 0;
			for (int k = 0; k < 3; k++) s += R_[3 * i + k] * t[k];
			tn[i] = s + t_[i];
		}
		memcpy(R, Rn, sizeof(Rn));
		memcpy(t, tn, sizeof(tn));
	}
	free(points); free(p_m); free(p_d);
	if (iters_out) *iters_out = iter;
	return err_new;
}

/* ------------------------------------------------------------------------------------------
 * Outer (rotation) BnB = GoICP::Initialize + OuterBnB (jly_goicp.cpp:134-209, :342-567)
 * ------------------------------------------------------------------------------------------ */
int orc_register(const orc_dt* dt, const float* model, int M, const float* data, int N,
                 float mse_thresh, orc_result* out)
{
	return orc_register_trim(dt, model, M, data, N, mse_thresh, 0.f, out);
}

int orc_register_trim(const orc_dt* dt, const float* model, int M, const float* data, int N,
                      float mse_thresh, float trim_fraction, orc_result* out)
{
	int inliers = (int)(N * (1 - trim_fraction));            /* jly_goicp.cpp:201 */
	if (inliers < 1) inliers = 1;
	float* norm = (float*)malloc(sizeof(float) * N);
	float* rho = (float*)malloc(sizeof(float) * (size_t)ORC_MAXROTLEVEL * N);
	float* prot = (float*)malloc(sizeof(float) * 3 * N);
	orc_rot_radii(data, N, norm, rho);
	orc_kd* kd = orc_kd_build(model, M);
	const float sse_thresh = mse_thresh * inliers;           /* :208 */
	const float icp_err_diff = mse_thresh / 10000;           /* :186 */
	const float troot[4] = { -0.5f, -0.5f, -0.5f, 1.0f };    /* :50-53 */
	memset(out, 0, sizeof(*out));

	float optR[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, optT[3] = { 0, 0, 0 };
	float optError;
	{   /* :357-372 initial error */
		const float I9[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, Z3[3] = { 0, 0, 0 };
		optError = orc_dt_sse_trim(dt, data, N, I9, Z3, inliers);
	}
	{   /* :375-391 initial ICP */
		float R[9], t[3];
		memcpy(R, optR, sizeof(R)); memcpy(t, optT, sizeof(t));
		int it = 0;
		orc_icp_run_trim(kd, model, data, N, inliers, R, t, 10000, icp_err_diff, &it);
		out->icp_runs++; out->icp_iters += it;
		float e = orc_dt_sse_trim(dt, data, N, R, t, inliers);
		if (e < optError) { optError = e; memcpy(optR, R, sizeof(R)); memcpy(optT, t, sizeof(t)); }
	}

	heap_t q;
	heap_init(&q);
	node_t rootR = { (float)-ORC_PI, (float)-ORC_PI, (float)-ORC_PI, (float)(2 * ORC_PI), 0, 0, 0 };
	heap_push(&q, rootR);
	int done = 0;
	while (!done) {
		if (!q.n) break;
		node_t parent = heap_pop(&q);
		out->rot_pops++;
		if ((optError - parent.lb) <= sse_thresh) break;      /* :416 */
		node_t c;
		c.w = parent.w / 2;
		c.l = parent.l + 1;
		for (int j = 0; j < 8 && !done; j++) {
			c.x = parent.x + (j & 1) * c.w;
			c.y = parent.y + (j >> 1 & 1) * c.w;
			c.z = parent.z + (j >> 2 & 1) * c.w;
			float v1 = c.x + c.w / 2, v2 = c.y + c.w / 2, v3 = c.z + c.w / 2;
			/* :443 pi-ball cull: float sqrt, double subtraction and compare */
			if ((double)sqrtf(v1 * v1 + v2 * v2 + v3 * v3) - ORC_SQRT3 * (double)c.w / 2 > ORC_PI) continue;
			float R[9];
			orc_rodrigues(v1, v2, v3, R);
			orc_rotate(R, data, N, prot);
			float best[4] = { 0, 0, 0, 0 };
			out->inner_calls++;
			float ub = orc_inner_bnb_trim(dt, prot, N, NULL, inliers, optError, sse_thresh, troot, best, &out->trans_pops, &out->cubes);
			if (ub < optError) {                              /* :495-544 */
				optError = ub;
				memcpy(optR, R, sizeof(R));
				optT[0] = best[0] + best[3] / 2; optT[1] = best[1] + best[3] / 2; optT[2] = best[2] + best[3] / 2;
				float Ri[9], ti[3];
				memcpy(Ri, optR, sizeof(Ri)); memcpy(ti, optT, sizeof(ti));
				int it = 0;
				orc_icp_run_trim(kd, model, data, N, inliers, Ri, ti, 10000, icp_err_diff, &it);
				out->icp_runs++; out->icp_iters += it;
				float e = orc_dt_sse_trim(dt, data, N, Ri, ti, inliers);
				if (e < optError) { optError = e; memcpy(optR, Ri, sizeof(Ri)); memcpy(optT, ti, sizeof(ti)); }
				if (optError < sse_thresh) { done = 1; break; }   /* :527 */
				/* :533-543 drop queued nodes with lb >= optError (rebuild in pop order) */
				heap_t nq;
				heap_init(&nq);
				while (q.n) {
					node_t nd = heap_pop(&q);
					if (nd.lb < optError) heap_push(&nq, nd); else break;
				}
				heap_free(&q);
				q = nq;
			}
			int level = c.l < ORC_MAXROTLEVEL ? c.l : ORC_MAXROTLEVEL - 1; /* reference would overrun at l >= 20 */
			out->inner_calls++;
			float lb = orc_inner_bnb_trim(dt, prot, N, rho + (size_t)level * N, inliers, optError, sse_thresh, troot, NULL, &out->trans_pops, &out->cubes);
			if (lb >= optError) continue;
			c.ub = ub; c.lb = lb;
			heap_push(&q, c);
		}
	}
	heap_free(&q);
	memcpy(out->R, optR, sizeof(optR));
	memcpy(out->t, optT, sizeof(optT));
	out->sse = optError;
	orc_kd_free(kd);
	free(norm); free(rho); free(prot);
	return 0;
}
