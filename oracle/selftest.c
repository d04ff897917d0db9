/* TEST INFRASTRUCTURE: sanitizer driver for the CPU oracle (make -C oracle asan).  Exercises every entry point of
 * goicp_oracle.c on small seeded clouds under -fsanitize=address,undefined; the golden-vector checks proper live in
 * tests/test_oracle_vs_golden.py.  Exit code 0 = no sanitizer report and the basic invariants hold. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "goicp_oracle.h"

static unsigned long long s_rng = 88172645463325252ull;
static float frand(void)
{
	s_rng ^= s_rng << 13; s_rng ^= s_rng >> 7; s_rng ^= s_rng << 17;
	return (float)((s_rng >> 11) * (1.0 / 9007199254740992.0));
}

int main(void)
{
	enum { M = 400, N = 120, V = 40 };
	float* model = malloc(sizeof(float) * 3 * M);
	float* data = malloc(sizeof(float) * 3 * N);
	for (int i = 0; i < 3 * M; i++) model[i] = frand() * 1.2f - 0.6f;
	const float Rt[9] = {0.9553365f, -0.2955202f, 0.f, 0.2955202f, 0.9553365f, 0.f, 0.f, 0.f, 1.f};
	for (int i = 0; i < N; i++) {
		const float* m = model + 3 * (i * 3);
		for (int r = 0; r < 3; r++) data[3 * i + r] = Rt[3 * r] * m[0] + Rt[3 * r + 1] * m[1] + Rt[3 * r + 2] * m[2] + 0.03f * (float)(r - 1);
	}
	orc_dt dt;
	if (orc_dt_build(model, M, V, 2.0, &dt) != 0) return 2;
	int bad = 0;
	for (int i = 0; i < 200; i++) {                                 /* lookups incl. far outside the grid */
		const float d = orc_dt_distance(&dt, frand() * 8.0 - 4.0, frand() * 8.0 - 4.0, frand() * 8.0 - 4.0);
		if (!(d >= 0.f) || !isfinite(d)) bad++;
	}
	float* norm = malloc(sizeof(float) * N);
	float* rho = malloc(sizeof(float) * 20 * N);
	orc_rot_radii(data, N, norm, rho);
	float R[9];
	orc_rodrigues(0.3f, -0.2f, 0.9f, R);
	float* prot = malloc(sizeof(float) * 3 * N);
	orc_rotate(R, data, N, prot);
	float ub, lb, ub2, lb2;
	orc_cube_bound(&dt, prot, N, NULL, 0.1f, -0.05f, 0.02f, 0.25f, &ub, &lb);
	orc_cube_bound(&dt, prot, N, rho + 4 * N, 0.1f, -0.05f, 0.02f, 0.25f, &ub2, &lb2);
	if (!(lb <= ub) || !(lb2 <= lb + 1e-4f) || !(ub2 <= ub + 1e-4f)) bad++;
	orc_cube_bound_trim(&dt, prot, N, NULL, 0.1f, -0.05f, 0.02f, 0.25f, N * 9 / 10, &ub2, &lb2);
	if (!(ub2 <= ub + 1e-4f)) bad++;
	orc_cube_bound_omp(&dt, prot, N, NULL, 0.1f, -0.05f, 0.02f, 0.25f, &ub2, &lb2);
	if (fabsf(ub2 - ub) > 1e-3f * ub) bad++;
	float cubes[8 * 4], ubs[8], lbs[8];
	for (int i = 0; i < 8; i++) { cubes[4 * i] = frand() - 0.5f; cubes[4 * i + 1] = frand() - 0.5f; cubes[4 * i + 2] = frand() - 0.5f; cubes[4 * i + 3] = 0.125f; }
	orc_cube_bounds_batch(&dt, prot, N, rho + 3 * N, cubes, 8, ubs, lbs, 1);
	const float root[4] = {-0.5f, -0.5f, -0.5f, 1.0f};
	float node[4] = {0, 0, 0, 0};
	long long pops = 0, ncubes = 0;
	const float thr = 1e-3f * N;
	const float v = orc_inner_bnb(&dt, prot, N, NULL, 1e10f, thr, root, node, &pops, &ncubes);
	const float vt = orc_inner_bnb_trim(&dt, prot, N, NULL, N * 9 / 10, 1e10f, thr, root, node, &pops, &ncubes);
	if (!(vt <= v + 1e-4f) || pops <= 0) bad++;
	orc_kd* kd = orc_kd_build(model, M);
	for (int i = 0; i < 100; i++) {
		const float q[3] = {frand() * 2 - 1, frand() * 2 - 1, frand() * 2 - 1};
		int ia, ib; float da, db;
		orc_kd_nn(kd, q, &ia, &da);
		orc_nn_brute(model, M, q, &ib, &db);
		if (da != db) bad++;
	}
	float Ri[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, ti[3] = {0, 0, 0};
	int iters = 0;
	const float e = orc_icp_run(kd, model, data, N, Ri, ti, 50, 1e-7f, &iters);
	if (!isfinite(e) || iters <= 0) bad++;
	orc_icp_run_trim(kd, model, data, N, N * 9 / 10, Ri, ti, 10, 1e-7f, &iters);
	const float H[9] = {0.3f, -0.1f, 0.2f, 0.05f, 0.4f, -0.3f, -0.2f, 0.1f, 0.5f};
	float Rk[9];
	orc_kabsch_rotation(H, Rk);
	const float det = Rk[0] * (Rk[4] * Rk[8] - Rk[5] * Rk[7]) - Rk[1] * (Rk[3] * Rk[8] - Rk[5] * Rk[6]) + Rk[2] * (Rk[3] * Rk[7] - Rk[4] * Rk[6]);
	if (fabsf(det - 1.f) > 1e-4f) bad++;
	orc_result res;
	if (orc_register(&dt, model, M, data, N, 5e-3f, &res) != 0 || !isfinite(res.sse)) bad++;
	if (orc_register_trim(&dt, model, M, data, N, 5e-3f, 0.1f, &res) != 0) bad++;
	(void)orc_dt_sse(&dt, data, N, res.R, res.t);
	(void)orc_dt_sse_trim(&dt, data, N, res.R, res.t, N * 9 / 10);
	orc_kd_free(kd);
	orc_dt_free(&dt);
	free(prot); free(rho); free(norm); free(data); free(model);
	printf("oracle selftest: %s\n", bad ? "FAILED" : "ok");
	return bad ? 1 : 0;
}
