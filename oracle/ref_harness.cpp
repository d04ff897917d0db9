// TEST INFRASTRUCTURE — golden-vector generator driven by the REAL reference code.
//
// This translation unit is our own harness.  It is compiled together with the
// reference's CPU Go-ICP sources *where they lie* under /root/reference
// (src/goicp/jly_goicp.cpp, jly_3ddt.cpp, matrix.cpp; see oracle/Makefile) into
// oracle/_ref/ref_harness.  Nothing from the reference is copied into this repo.
// It only exists in the build container (the GPU box has no /root/reference); its
// outputs are the small fixtures committed under tests/golden/.
//
// Private members of GoICP / ICP3D are reached by compiling THIS file with
// -Dprivate=public (the reference objects themselves are compiled unmodified;
// access specifiers do not change layout for these classes).
//
// usage: ref_harness <cmd> <outdir> [args]
//   e2e    <outdir> <tag> <model.txt> <data.txt> <mse> <stride>
//   units  <outdir> <model.txt> <data.txt> <stride>
//   cloud  <out.f32> <cloud.txt>
//   trim   <outdir> <model.txt> <data.txt> <stride> <trim_fraction>
//   bench  <model.f32> <data.f32> <seconds>      (bench.py's cpu_baseline leg: the reference's own InnerBnB, timed;
//                                                 run from oracle/_ref/ref_harness_bench = Release-flag objects)
//   count  <model.f32> <data.f32> <calls> <out.json>   (ref_harness_count only: exact cube-bound counts of that sequence)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <random>
#include <chrono>

#include "jly_goicp.h"

// globals the reference TU expects (src/goicp/jly_goicp.cpp:36-38)
bool goicp_finished = false;
float mse_threshold = 1e-3f;
float sse_threshold = 0.f;
extern long long tNodeCount;
extern long long rNodeCount;

static std::vector<glm::vec3> load_txt(const char* path, int stride)
{
	FILE* f = fopen(path, "r");
	if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
	int n = 0;
	if (fscanf(f, "%d", &n) != 1) exit(2);
	std::vector<glm::vec3> out;
	for (int i = 0; i < n; i++) {
		float x, y, z;
		if (fscanf(f, "%f %f %f", &x, &y, &z) != 3) exit(2);
		if (i % stride == 0) out.emplace_back(x, y, z);
	}
	fclose(f);
	return out;
}

static void jf(FILE* f, const char* k, float v, bool comma = true) { fprintf(f, "\"%s\": %.9g%s\n", k, v, comma ? "," : ""); }
static void jd(FILE* f, const char* k, double v, bool comma = true) { fprintf(f, "\"%s\": %.17g%s\n", k, v, comma ? "," : ""); }
static void jmat(FILE* f, const char* k, Matrix& M, bool comma = true)
{
	fprintf(f, "\"%s\": [", k);
	for (int i = 0; i < M.m; i++) for (int j = 0; j < M.n; j++)
		fprintf(f, "%.9g%s", M.val[i][j], (i == M.m - 1 && j == M.n - 1) ? "" : ", ");
	fprintf(f, "]%s\n", comma ? "," : "");
}
template <class T> static void jarrf(FILE* f, const char* k, const T* v, size_t n, bool comma = true)
{
	fprintf(f, "\"%s\": [", k);
	for (size_t i = 0; i < n; i++) fprintf(f, "%.9g%s", (double)v[i], i + 1 == n ? "" : ", ");
	fprintf(f, "]%s\n", comma ? "," : "");
}
static void jarrd(FILE* f, const char* k, const double* v, size_t n, bool comma = true)
{
	fprintf(f, "\"%s\": [", k);
	for (size_t i = 0; i < n; i++) fprintf(f, "%.17g%s", v[i], i + 1 == n ? "" : ", ");
	fprintf(f, "]%s\n", comma ? "," : "");
}
static void jarri(FILE* f, const char* k, const long long* v, size_t n, bool comma = true)
{
	fprintf(f, "\"%s\": [", k);
	for (size_t i = 0; i < n; i++) fprintf(f, "%lld%s", v[i], i + 1 == n ? "" : ", ");
	fprintf(f, "]%s\n", comma ? "," : "");
}

static double now_s()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------- e2e
static int cmd_e2e(int argc, char** argv)
{
	if (argc < 8) return 1;
	std::string outdir = argv[2], tag = argv[3];
	auto model = load_txt(argv[4], 1);
	float mse = (float)atof(argv[6]);
	int stride = atoi(argv[7]);
	auto data = load_txt(argv[5], stride);

	mse_threshold = mse;
	sse_threshold = mse * data.size();
	GoICP g(mse);
	g.pModel = model.data(); g.Nm = (int)model.size();
	g.pData = data.data();   g.Nd = (int)data.size();
	double t0 = now_s();
	g.BuildDT();
	double t1 = now_s();
	tNodeCount = rNodeCount = 0;
	float err = g.Register();
	double t2 = now_s();

	FILE* f = fopen((outdir + "/e2e_" + tag + ".json").c_str(), "w");
	fprintf(f, "{\n");
	fprintf(f, "\"tag\": \"%s\", \"Nm\": %d, \"Nd\": %d, \"stride\": %d,\n", tag.c_str(), g.Nm, g.Nd, stride);
	jf(f, "mse_threshold", mse);
	jf(f, "sse_threshold", g.SSEThresh);
	jf(f, "sse", err);
	jmat(f, "R", g.optR);
	jmat(f, "t", g.optT);
	fprintf(f, "\"tNodeCount\": %lld, \"rNodeCount\": %lld,\n", tNodeCount, rNodeCount);
	jd(f, "dt_scale", g.dt.scale);
	jd(f, "dt_xmin", g.dt.xMin); jd(f, "dt_ymin", g.dt.yMin); jd(f, "dt_zmin", g.dt.zMin);
	jd(f, "dt_build_s", t1 - t0);
	jd(f, "register_s", t2 - t1, false);
	fprintf(f, "}\n");
	fclose(f);
	return 0;
}

// ---------------------------------------------------------------- unit fixtures
static void rodrigues(const float v[3], float R[9])
{
	// harness-side rotation set-up (angle-axis -> matrix); the consumer of a fixture receives R itself,
	// so only R*p has to agree with the reference.
	float theta = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
	if (theta <= 0) { for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.f : 0.f; return; }
	float ux = v[0] / theta, uy = v[1] / theta, uz = v[2] / theta;
	float c = cos(theta), omc = 1 - c, sn = sin(theta);
	float xy = ux * uy * omc, xz = ux * uz * omc, yz = uy * uz * omc;
	float zs = uz * sn, ys = uy * sn, xs = ux * sn;
	R[0] = c + ux * ux * omc; R[1] = xy - zs;           R[2] = xz + ys;
	R[3] = xy + zs;           R[4] = c + uy * uy * omc; R[5] = yz - xs;
	R[6] = xz - ys;           R[7] = yz + xs;           R[8] = c + uz * uz * omc;
}

static int cmd_units(int argc, char** argv)
{
	if (argc < 6) return 1;
	std::string outdir = argv[2];
	auto model = load_txt(argv[3], 1);
	int stride = atoi(argv[5]);
	auto data = load_txt(argv[4], stride);
	const float mse = 1e-3f;
	mse_threshold = mse;
	GoICP g(mse);
	g.pModel = model.data(); g.Nm = (int)model.size();
	g.pData = data.data();   g.Nd = (int)data.size();
	g.BuildDT();
	g.Initialize();
	const int N = g.Nd;
	std::mt19937 rng(20241223u);

	// ---- 1. DT lookup: DT3D::Distance at seeded query points, in-grid, out-of-grid, negative overshoot
	{
		const int Q = 4096;
		std::vector<double> q(3 * Q);
		std::vector<float> d(Q);
		double side = g.dt.SIZE / g.dt.scale;
		std::uniform_real_distribution<double> uin(0.0, 1.0);
		for (int i = 0; i < Q; i++) {
			double lo, hi;                       // fraction of the grid side
			if (i < 2048)      { lo = 0.0;  hi = 1.0; }       // inside
			else if (i < 3072) { lo = -0.25; hi = 1.25; }     // straddling the faces
			else if (i < 3584) { lo = -0.004; hi = 0.004; }   // around index 0 (int() truncation toward zero)
			else               { lo = 0.996; hi = 1.004; }    // around index SIZE-1 / SIZE
			double mins[3] = { g.dt.xMin, g.dt.yMin, g.dt.zMin };
			for (int a = 0; a < 3; a++) {
				double fr = lo + (hi - lo) * uin(rng);
				// for the edge groups only perturb one axis strongly, the others stay inside
				if (i >= 3072 && a != (i % 3)) fr = uin(rng);
				q[3 * i + a] = mins[a] + fr * side;
			}
			d[i] = g.dt.Distance(q[3 * i], q[3 * i + 1], q[3 * i + 2]);
		}
		// raw voxel samples (for the DT-build tolerance check): Distance at exact voxel centres
		const int S = 8192;
		std::vector<long long> vox(3 * S);
		std::vector<float> vd(S);
		std::uniform_int_distribution<int> uv(0, g.dt.SIZE - 1);
		for (int i = 0; i < S; i++) {
			int x = uv(rng), y = uv(rng), z = uv(rng);
			vox[3 * i] = x; vox[3 * i + 1] = y; vox[3 * i + 2] = z;
			vd[i] = g.dt.Distance(g.dt.xMin + x / g.dt.scale, g.dt.yMin + y / g.dt.scale, g.dt.zMin + z / g.dt.scale);
		}
		FILE* f = fopen((outdir + "/dt_lookup.json").c_str(), "w");
		fprintf(f, "{\n\"Nm\": %d, \"SIZE\": %d,\n", g.Nm, g.dt.SIZE);
		jd(f, "scale", g.dt.scale);
		jd(f, "xmin", g.dt.xMin); jd(f, "ymin", g.dt.yMin); jd(f, "zmin", g.dt.zMin);
		jarrd(f, "query", q.data(), q.size());
		jarrf(f, "distance", d.data(), d.size());
		jarri(f, "voxel", vox.data(), vox.size());
		jarrf(f, "voxel_distance", vd.data(), vd.size(), false);
		fprintf(f, "}\n");
		fclose(f);
	}

	// ---- 2. rotation uncertainty radii (GoICP::Initialize, jly_goicp.cpp:139-160)
	{
		FILE* f = fopen((outdir + "/rot_radii.json").c_str(), "w");
		fprintf(f, "{\n\"Nd\": %d, \"stride\": %d, \"npts\": 64,\n\"maxRotDis\": [\n", N, stride);
		for (int l = 0; l < MAXROTLEVEL; l++) {
			fprintf(f, "[");
			for (int i = 0; i < 64; i++) fprintf(f, "%.9g%s", g.maxRotDis[l][i], i == 63 ? "" : ", ");
			fprintf(f, "]%s\n", l == MAXROTLEVEL - 1 ? "" : ",");
		}
		fprintf(f, "]\n}\n");
		fclose(f);
	}

	// ---- 3. InnerBnB: (a) single-expansion pins = min over 8 children of one parent cube,
	//                   (b) full inner searches (value, best node, node count)
	{
		const float rotv[3][3] = { { 1.5707963f, -1.5707963f, 1.5707963f },   // a level-1 child centre
		                           { 0.3f, -0.2f, 0.9f },
		                           { -2.1f, 0.4f, 1.1f } };
		FILE* f = fopen((outdir + "/inner_bnb.json").c_str(), "w");
		fprintf(f, "{\n\"Nd\": %d, \"stride\": %d,\n", N, stride);
		jf(f, "sse_threshold", g.SSEThresh);
		fprintf(f, "\"cases\": [\n");
		TRANSNODE rootT = g.initNodeTrans;
		float sseT = g.SSEThresh;
		std::uniform_real_distribution<float> uc(-0.5f, 0.5f);
		for (int r = 0; r < 3; r++) {
			float R[9];
			rodrigues(rotv[r], R);
			for (int i = 0; i < N; i++) {        // jly_goicp.cpp:470-476
				POINT3D& p = g.pData[i];
				g.pDataTemp[i].x = R[0] * p.x + R[1] * p.y + R[2] * p.z;
				g.pDataTemp[i].y = R[3] * p.x + R[4] * p.y + R[5] * p.z;
				g.pDataTemp[i].z = R[6] * p.x + R[7] * p.y + R[8] * p.z;
			}
			fprintf(f, "{");
			jarrf(f, "R", R, 9);
			// (a) single expansions
			fprintf(f, "\"single\": [\n");
			for (int c = 0; c < 64; c++) {
				int lev = c % 7;                 // parent widths 1, 1/2, ... 1/64
				TRANSNODE parent;
				parent.w = 1.0f / (float)(1 << lev);
				parent.x = uc(rng) * (1 - parent.w) - parent.w / 2;
				parent.y = uc(rng) * (1 - parent.w) - parent.w / 2;
				parent.z = uc(rng) * (1 - parent.w) - parent.w / 2;
				parent.lb = 0; parent.ub = 0;
				for (int pass = 0; pass < 2; pass++) {
					int level = 3 + (c % 5);
					g.initNodeTrans = parent;
					g.optError = 1e10f;
					g.SSEThresh = 1e9f;          // => exactly one expansion (jly_goicp.cpp:257)
					TRANSNODE best; best.x = best.y = best.z = best.w = 0;
					long long c0 = tNodeCount;
					float v = g.InnerBnB(pass ? g.maxRotDis[level] : NULL, &best);
					fprintf(f, "{\"parent\": [%.9g, %.9g, %.9g, %.9g], \"level\": %d, \"min_ub\": %.9g, "
					           "\"best\": [%.9g, %.9g, %.9g, %.9g], \"pops\": %lld}%s\n",
					        parent.x, parent.y, parent.z, parent.w, pass ? level : -1, v,
					        best.x, best.y, best.z, best.w, tNodeCount - c0, (c == 63 && pass == 1) ? "" : ",");
				}
			}
			fprintf(f, "],\n");
			// (b) full searches from the standard root, with a realistic incumbent
			g.initNodeTrans = rootT;
			g.SSEThresh = sseT;
			fprintf(f, "\"full\": [\n");
			const float incumbents[2] = { 1e10f, 35.0f * (float)N / 3038.0f };
			for (int k = 0; k < 2; k++) for (int pass = 0; pass < 2; pass++) {
				int level = 6 + r;
				g.optError = incumbents[k];
				TRANSNODE best; best.x = best.y = best.z = best.w = 0;
				long long c0 = tNodeCount;
				float v = g.InnerBnB(pass ? g.maxRotDis[level] : NULL, &best);
				fprintf(f, "{\"incumbent\": %.9g, \"level\": %d, \"value\": %.9g, "
				           "\"best\": [%.9g, %.9g, %.9g, %.9g], \"pops\": %lld}%s\n",
				        incumbents[k], pass ? level : -1, v, best.x, best.y, best.z, best.w,
				        tNodeCount - c0, (k == 1 && pass == 1) ? "" : ",");
			}
			fprintf(f, "]}%s\n", r == 2 ? "" : ",");
		}
		fprintf(f, "]\n}\n");
		fclose(f);
		g.initNodeTrans = rootT;
		g.SSEThresh = sseT;
	}

	// ---- 4. ICP3D<float>::Run (jly_icp3d.hpp:181-295) from given poses, forced iteration counts
	{
		FILE* f = fopen((outdir + "/icp_iter.json").c_str(), "w");
		fprintf(f, "{\n\"Nd\": %d, \"stride\": %d,\n\"cases\": [\n", N, stride);
		const size_t iters[4] = { 1, 2, 10, 10000 };
		const float start[2][3] = { { 0, 0, 0 }, { 0.05f, -0.04f, 0.03f } };
		for (int s = 0; s < 2; s++) for (int k = 0; k < 4; k++) {
			float R0[9]; rodrigues(start[s], R0);
			Matrix R(3, 3, R0), t(3, 1);
			t.val[0][0] = s ? 0.01f : 0.f; t.val[1][0] = s ? -0.02f : 0.f; t.val[2][0] = s ? 0.015f : 0.f;
			Matrix Rin = R, tin = t;
			float err = g.icp3d.Run(g.D_icp, g.Nd, R, t, iters[k], g.icp3d.err_diff_def);
			fprintf(f, "{\"max_iter\": %zu, ", iters[k]);
			jmat(f, "R0", Rin); jmat(f, "t0", tin); jmat(f, "R", R); jmat(f, "t", t);
			jf(f, "err_diff", g.icp3d.err_diff_def);
			jf(f, "err", err, false);
			fprintf(f, "}%s\n", (s == 1 && k == 3) ? "" : ",");
		}
		fprintf(f, "]\n}\n");
		fclose(f);
	}

	// ---- 5. GoICP::ICP = ICP3D::Run then DT re-score (jly_goicp.cpp:93-132)
	{
		FILE* f = fopen((outdir + "/icp_dt_score.json").c_str(), "w");
		Matrix R = Matrix::eye(3), t(3, 1);
		float e = g.ICP(R, t);
		fprintf(f, "{\n\"Nd\": %d, \"stride\": %d,\n", N, stride);
		jmat(f, "R", R); jmat(f, "t", t); jf(f, "dt_sse", e, false);
		fprintf(f, "}\n");
		fclose(f);
	}

	// ---- 6. 3x3 SVD → rotation (jly_icp3d.hpp:268-285 on top of matrix.cpp:602-830)
	{
		FILE* f = fopen((outdir + "/svd3x3.json").c_str(), "w");
		fprintf(f, "{\n\"cases\": [\n");
		std::uniform_real_distribution<float> u(-1.f, 1.f);
		for (int c = 0; c < 32; c++) {
			Matrix H(3, 3);
			for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) H.val[i][j] = u(rng) * (c < 16 ? 1.f : 1000.f);
			Matrix U, W, V;
			H.svd(U, W, V);
			Matrix R_ = V * ~U;
			float a = R_.val[0][0] * (R_.val[1][1] * R_.val[2][2] - R_.val[1][2] * R_.val[2][1]);
			float b = -R_.val[0][1] * (R_.val[1][0] * R_.val[2][2] - R_.val[1][2] * R_.val[2][0]);
			float cc = R_.val[0][2] * (R_.val[1][0] * R_.val[2][1] - R_.val[1][1] * R_.val[2][0]);
			float det = a + b + cc;
			Matrix tmp = Matrix::eye(3);
			tmp.val[2][2] = det;
			R_ = V * tmp * ~U;
			fprintf(f, "{");
			jmat(f, "H", H); jmat(f, "W", W); jf(f, "det", det); jmat(f, "R", R_, false);
			fprintf(f, "}%s\n", c == 31 ? "" : ",");
		}
		fprintf(f, "]\n}\n");
		fclose(f);
	}

	// ---- 7. exact 1-NN through the reference's own k-d tree (nanoflann_goicp.hpp:1137-1184)
	{
		const int Q = 4096;
		std::vector<float> q(3 * Q), d2(Q);
		std::vector<long long> idx(Q);
		std::uniform_real_distribution<float> u(-1.2f, 1.2f);
		for (int i = 0; i < Q; i++) {
			if (i < 1024) {                        // queries near the surface: perturbed model points
				const glm::vec3& m = model[(size_t)i * 31 % model.size()];
				q[3 * i] = m.x + 0.01f * u(rng); q[3 * i + 1] = m.y + 0.01f * u(rng); q[3 * i + 2] = m.z + 0.01f * u(rng);
			} else {
				q[3 * i] = u(rng); q[3 * i + 1] = u(rng); q[3 * i + 2] = u(rng);
			}
			size_t ri; float rd;
			g.icp3d.kdtree->knnSearch(&q[3 * i], 1, &ri, &rd);
			idx[i] = (long long)ri; d2[i] = rd;
		}
		FILE* f = fopen((outdir + "/nn.json").c_str(), "w");
		fprintf(f, "{\n\"Nm\": %d,\n", g.Nm);
		jarrf(f, "query", q.data(), q.size());
		jarri(f, "index", idx.data(), idx.size());
		jarrf(f, "dist_sq", d2.data(), d2.size(), false);
		fprintf(f, "}\n");
		fclose(f);
	}
	return 0;
}

// ---------------------------------------------------------------- trimmed bounds (trimFraction > 0)
// GoICP::trimFraction is a public field (jly_goicp.h:116) the reference hard-wires to 0 (jly_goicp.cpp:55);
// setting it before Initialize() drives the reference's own trimmed InnerBnB (jly_goicp.cpp:293-315).
static int cmd_trim(int argc, char** argv)
{
	if (argc < 7) return 1;
	std::string outdir = argv[2];
	auto model = load_txt(argv[3], 1);
	int stride = atoi(argv[5]);
	float frac = (float)atof(argv[6]);
	auto data = load_txt(argv[4], stride);
	mse_threshold = 1e-3f;
	GoICP g(1e-3f);
	g.pModel = model.data(); g.Nm = (int)model.size();
	g.pData = data.data();   g.Nd = (int)data.size();
	g.trimFraction = frac;
	g.BuildDT();
	g.Initialize();
	const int N = g.Nd;
	std::mt19937 rng(424242u);
	const float rotv[2][3] = { { 0.3f, -0.2f, 0.9f }, { -2.1f, 0.4f, 1.1f } };
	FILE* f = fopen((outdir + "/inner_bnb_trim.json").c_str(), "w");
	fprintf(f, "{\n\"Nd\": %d, \"stride\": %d, \"inlierNum\": %d,\n", N, stride, g.inlierNum);
	jf(f, "trim_fraction", frac);
	jf(f, "sse_threshold", g.SSEThresh);
	fprintf(f, "\"cases\": [\n");
	TRANSNODE rootT = g.initNodeTrans;
	float sseT = g.SSEThresh;
	std::uniform_real_distribution<float> uc(-0.5f, 0.5f);
	for (int r = 0; r < 2; r++) {
		float R[9];
		rodrigues(rotv[r], R);
		for (int i = 0; i < N; i++) {
			POINT3D& p = g.pData[i];
			g.pDataTemp[i].x = R[0] * p.x + R[1] * p.y + R[2] * p.z;
			g.pDataTemp[i].y = R[3] * p.x + R[4] * p.y + R[5] * p.z;
			g.pDataTemp[i].z = R[6] * p.x + R[7] * p.y + R[8] * p.z;
		}
		fprintf(f, "{");
		jarrf(f, "R", R, 9);
		fprintf(f, "\"single\": [\n");
		for (int c = 0; c < 32; c++) {
			int lev = c % 7;
			TRANSNODE parent;
			parent.w = 1.0f / (float)(1 << lev);
			parent.x = uc(rng) * (1 - parent.w) - parent.w / 2;
			parent.y = uc(rng) * (1 - parent.w) - parent.w / 2;
			parent.z = uc(rng) * (1 - parent.w) - parent.w / 2;
			parent.lb = 0; parent.ub = 0;
			for (int pass = 0; pass < 2; pass++) {
				int level = 3 + (c % 5);
				g.initNodeTrans = parent;
				g.optError = 1e10f;
				g.SSEThresh = 1e9f;
				TRANSNODE best; best.x = best.y = best.z = best.w = 0;
				float v = g.InnerBnB(pass ? g.maxRotDis[level] : NULL, &best);
				fprintf(f, "{\"parent\": [%.9g, %.9g, %.9g, %.9g], \"level\": %d, \"min_ub\": %.9g, "
				           "\"best\": [%.9g, %.9g, %.9g, %.9g]}%s\n",
				        parent.x, parent.y, parent.z, parent.w, pass ? level : -1, v,
				        best.x, best.y, best.z, best.w, (c == 31 && pass == 1) ? "" : ",");
			}
		}
		fprintf(f, "],\n");
		g.initNodeTrans = rootT;
		g.SSEThresh = sseT;
		fprintf(f, "\"full\": [\n");
		for (int pass = 0; pass < 2; pass++) {
			int level = 6 + r;
			g.optError = 1e10f;
			TRANSNODE best; best.x = best.y = best.z = best.w = 0;
			long long c0 = tNodeCount;
			float v = g.InnerBnB(pass ? g.maxRotDis[level] : NULL, &best);
			fprintf(f, "{\"incumbent\": 1e10, \"level\": %d, \"value\": %.9g, \"best\": [%.9g, %.9g, %.9g, %.9g], \"pops\": %lld}%s\n",
			        pass ? level : -1, v, best.x, best.y, best.z, best.w, tNodeCount - c0, pass == 1 ? "" : ",");
		}
		fprintf(f, "]}%s\n", r == 1 ? "" : ",");
	}
	fprintf(f, "]\n}\n");
	fclose(f);
	return 0;
}

// ---------------------------------------------------------------- cloud -> float32 blob
static int cmd_cloud(int argc, char** argv)
{
	if (argc < 4) return 1;
	auto c = load_txt(argv[3], 1);      // "%f" parse, as the reference's `stream >> float` does
	FILE* f = fopen(argv[2], "wb");
	fwrite(c.data(), sizeof(glm::vec3), c.size(), f);
	fclose(f);
	printf("%zu\n", c.size());
	return 0;
}


// ---------------------------------------------------------------- bench
// The reference's own GoICP::InnerBnB on the committed fixture clouds (raw little-endian float32 xyz),
// timed for about <seconds> of wall clock after the DT build: alternating upper-bound (no rotation
// uncertainty) and lower-bound searches from the standard translation root, over seeded rotations,
// with a realistic incumbent.  Every popped translation node evaluates its 8 children
// (jly_goicp.cpp:262-315), so cube bounds = 8 x the reference's own tNodeCount.
static std::vector<glm::vec3> load_f32(const char* path)
{
	FILE* f = fopen(path, "rb");
	if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
	fseek(f, 0, SEEK_END);
	const long bytes = ftell(f);
	fseek(f, 0, SEEK_SET);
	std::vector<glm::vec3> out((size_t)bytes / 12);
	if (fread(out.data(), 12, out.size(), f) != out.size()) exit(2);
	fclose(f);
	return out;
}

// Exact count of cube bounds: one cube bound = Nd calls of DT3D::Distance (jly_goicp.cpp:262-315), so the count binary
// (oracle/Makefile: same objects, linked with --wrap on DT3D::Distance) counts the lookups.  The timed binary is
// linked WITHOUT the wrapper: pure reference objects.
#ifdef GOICP_COUNT_DISTANCE
static long long g_distance_calls = 0;
extern "C" float __real__ZN4DT3D8DistanceEddd(DT3D* self, double x, double y, double z);
extern "C" float __wrap__ZN4DT3D8DistanceEddd(DT3D* self, double x, double y, double z)
{
	g_distance_calls++;
	return __real__ZN4DT3D8DistanceEddd(self, x, y, z);
}
#endif

// The seeded sequence of InnerBnB calls both modes replay: call k uses rotation k/2 (seeded, inside the pi-ball),
// even k = upper-bound search (no rotation radii), odd k = lower-bound search with the radii of level 3 + (k/2 % 5),
// each from the standard translation root with the incumbent of the "full" golden cases.
struct BenchSeq {
	GoICP& g;
	TRANSNODE rootT;
	float incumbent;
	std::mt19937 rng{20241223u};
	std::uniform_real_distribution<float> ua{-3.14159265f, 3.14159265f};
	long long calls = 0;
	explicit BenchSeq(GoICP& g_) : g(g_), rootT(g_.initNodeTrans), incumbent(35.0f * (float)g_.Nd / 3038.0f) {}
	void next()
	{
		const int N = g.Nd;
		if (calls % 2 == 0) {
			float v[3];
			do { v[0] = ua(rng); v[1] = ua(rng); v[2] = ua(rng); } while (v[0] * v[0] + v[1] * v[1] + v[2] * v[2] > 9.8696f);
			float R[9];
			rodrigues(v, R);
			for (int i = 0; i < N; i++) {            // jly_goicp.cpp:470-476
				POINT3D& p = g.pData[i];
				g.pDataTemp[i].x = R[0] * p.x + R[1] * p.y + R[2] * p.z;
				g.pDataTemp[i].y = R[3] * p.x + R[4] * p.y + R[5] * p.z;
				g.pDataTemp[i].z = R[6] * p.x + R[7] * p.y + R[8] * p.z;
			}
		}
		const int level = 3 + (int)(calls / 2 % 5);
		g.initNodeTrans = rootT;
		g.optError = incumbent;
		TRANSNODE best; best.x = best.y = best.z = best.w = 0;
		g.InnerBnB(calls % 2 ? g.maxRotDis[level] : NULL, &best);
		calls++;
	}
};

static void bench_setup(GoICP& g, std::vector<glm::vec3>& model, std::vector<glm::vec3>& data, double* dt_build_s)
{
	g.pModel = model.data(); g.Nm = (int)model.size();
	g.pData = data.data();   g.Nd = (int)data.size();
	const double t_dt = now_s();
	g.BuildDT();
	*dt_build_s = now_s() - t_dt;
	g.Initialize();
}

// bench <model.f32> <data.f32> <seconds>: the reference's own GoICP::InnerBnB on the committed fixture clouds, timed for
// about <seconds> of wall clock after the DT build.  Prints the number of completed calls and the reference's own
// tNodeCount (pops, including each call's final non-expanding pop); the caller turns `calls` into the EXACT number of
// cube bounds with tests/golden/ref_bench_counts.json (written by `count`), after checking that the pops agree.
static int cmd_bench(int argc, char** argv)
{
	if (argc < 5) return 1;
	auto model = load_f32(argv[2]);
	auto data = load_f32(argv[3]);
	const double budget = atof(argv[4]);
	mse_threshold = 1e-3f;
	GoICP g(mse_threshold);
	double dt_build_s = 0;
	bench_setup(g, model, data, &dt_build_s);
	BenchSeq seq(g);
	const long long pops0 = tNodeCount;
	const double t0 = now_s();
	while (now_s() - t0 < budget) seq.next();
	const double sec = now_s() - t0;
	const long long pops = tNodeCount - pops0;
	printf("{\"kind\": \"reference\", \"Nd\": %d, \"Nm\": %d, \"dt_build_s\": %.3f, \"seconds\": %.6f, \"inner_bnb_calls\": %lld, "
	       "\"trans_pops\": %lld, \"cube_bounds_upper\": %lld}\n",
	       g.Nd, g.Nm, dt_build_s, sec, seq.calls, pops, 8 * pops);
	return 0;
}

// count <model.f32> <data.f32> <calls> <out.json>: replay the first <calls> calls of the same sequence and write, per
// prefix of calls, the reference's pops and the exact number of cube bounds (Distance calls / Nd).
static int cmd_count(int argc, char** argv)
{
#ifndef GOICP_COUNT_DISTANCE
	fprintf(stderr, "count: this binary was linked without the Distance wrapper (use ref_harness_count)\n");
	return 1;
#else
	if (argc < 6) return 1;
	auto model = load_f32(argv[2]);
	auto data = load_f32(argv[3]);
	const long long ncalls = atoll(argv[4]);
	mse_threshold = 1e-3f;
	GoICP g(mse_threshold);
	double dt_build_s = 0;
	bench_setup(g, model, data, &dt_build_s);
	BenchSeq seq(g);
	std::vector<long long> pops(1, 0), cubes(1, 0);
	const long long pops0 = tNodeCount;
	g_distance_calls = 0;
	for (long long k = 0; k < ncalls; k++) {
		seq.next();
		if (g_distance_calls % g.Nd) { fprintf(stderr, "count: lookups not a multiple of Nd\n"); return 1; }
		pops.push_back(tNodeCount - pops0);
		cubes.push_back(g_distance_calls / g.Nd);
	}
	FILE* f = fopen(argv[5], "w");
	if (!f) return 1;
	fprintf(f, "{\"what\": \"reference InnerBnB bench sequence (oracle/ref_harness.cpp BenchSeq): after k calls, the reference's tNodeCount and the exact number of cube bounds = DT3D::Distance calls / Nd\",\n");
	fprintf(f, "\"Nd\": %d, \"Nm\": %d, \"calls\": %lld,\n\"pops_prefix\": [", g.Nd, g.Nm, ncalls);
	for (size_t i = 0; i < pops.size(); i++) fprintf(f, "%lld%s", pops[i], i + 1 < pops.size() ? "," : "");
	fprintf(f, "],\n\"cube_bounds_prefix\": [");
	for (size_t i = 0; i < cubes.size(); i++) fprintf(f, "%lld%s", cubes[i], i + 1 < cubes.size() ? "," : "");
	fprintf(f, "]\n}\n");
	fclose(f);
	printf("%lld calls: %lld pops, %lld exact cube bounds (8 x pops = %lld)\n", ncalls, pops.back(), cubes.back(), 8 * pops.back());
	return 0;
#endif
}

int main(int argc, char** argv)
{
	if (argc < 3) { fprintf(stderr, "usage: ref_harness e2e|units <outdir> ...\n"); return 1; }
	if (!strcmp(argv[1], "e2e")) return cmd_e2e(argc, argv);
	if (!strcmp(argv[1], "units")) return cmd_units(argc, argv);
	if (!strcmp(argv[1], "cloud")) return cmd_cloud(argc, argv);
	if (!strcmp(argv[1], "trim")) return cmd_trim(argc, argv);
	if (!strcmp(argv[1], "bench")) return cmd_bench(argc, argv);
	if (!strcmp(argv[1], "count")) return cmd_count(argc, argv);
	return 1;
}
