// goicp_mi355.hpp -- header-only C++ shim over the C ABI (goicp_mi355.h) that re-creates the reference's class
// and step-function surface, so that the reference's own call sites compile against this engine:
//     Config / load_cloud                               src/common.h:133-180, src/common.cpp:205-228
//     icp::RotNode / TransNode / StreamPool             src/fgoicp/fgoicp_common.hpp:64-167
//     icp::Registration::compute_sse_error x2           src/fgoicp/registration.hpp:96-97
//     icp::IterativeClosestPoint3D::run                 src/fgoicp/icp3d.hpp:30-35
//     icp::FastGoICP{run,get_best_error,optR,optT,curR,curT,finished}   src/fgoicp/fgoicp.hpp:11-69  (ctor: src/main.cpp:94,
//                                                       worker: src/main.cpp:150)
//     PointCloud::initBuffers / cleanupBuffers          src/kernel.h:57-61
//     ICP::naiveGPUStep / kdTreeGPUStep / CPUStep       src/icp_kernel.h:9-13
//     ICP::goicpGPUStep                                 src/goicp_kernel.h:6-10, src/goicp_kernel.cu:161-177
//
// Matrix / vector types.  The reference uses glm::mat3 / glm::vec3.  When glm is visible (any glm header was included
// before this one, or GOICP_MI355_USE_GLM is defined) Mat3 / Vec3 ARE glm::mat3 / glm::vec3, so reference statements such
// as `prev_optR != fgoicp->optR` or `prev_optR = fgoicp->optR` compile unchanged.  Without glm the shim supplies
// layout-compatible stand-ins (column-major 3x3, `m[col][row]`, like glm).  The C ABI underneath is row-major float[9];
// the conversion happens here.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#if defined(GOICP_MI355_USE_GLM) && !defined(GLM_VERSION)
#include <glm/mat3x3.hpp>
#include <glm/vec3.hpp>
#endif

#include "goicp_mi355.h"

namespace goicp_mi355 {

#if defined(GLM_VERSION)
using Mat3 = glm::mat3;
using Vec3 = glm::vec3;
#else
struct Vec3 {
	float x = 0.f, y = 0.f, z = 0.f;
	Vec3() = default;
	explicit Vec3(float s) : x(s), y(s), z(s) {}
	Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
	float& operator[](int i) { return (&x)[i]; }
	const float& operator[](int i) const { return (&x)[i]; }
	friend bool operator==(const Vec3& a, const Vec3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
	friend bool operator!=(const Vec3& a, const Vec3& b) { return !(a == b); }
};
struct Mat3 {            // column-major like glm::mat3: m[col][row]
	Vec3 c[3];
	Mat3() : Mat3(1.0f) {}
	explicit Mat3(float d) { c[0] = Vec3(d, 0, 0); c[1] = Vec3(0, d, 0); c[2] = Vec3(0, 0, d); }
	Vec3& operator[](int i) { return c[i]; }
	const Vec3& operator[](int i) const { return c[i]; }
	friend bool operator==(const Mat3& a, const Mat3& b) { return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2]; }
	friend bool operator!=(const Mat3& a, const Mat3& b) { return !(a == b); }
};
#endif
static_assert(sizeof(Vec3) == 3 * sizeof(float) && sizeof(Mat3) == 9 * sizeof(float), "packed float vector / matrix types expected");

// row-major float[9] (C ABI) <-> column-major Mat3
inline void to_rows(const Mat3& M, float r[9])
{
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) r[3 * i + j] = M[j][i];
}
inline Mat3 from_rows(const float r[9])
{
	Mat3 M(1.0f);
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) M[j][i] = r[3 * i + j];
	return M;
}
inline Vec3 from_xyz(const float t[3]) { return Vec3(t[0], t[1], t[2]); }

inline void check(int status)
{
	if (status != GOICP_OK) throw std::runtime_error(goicp_last_error());   // the reference throws std::runtime_error too
}

// class Config (src/common.h:133-180): same public fields
class Config {
public:
	int mode = 1;
	bool trim = false;
	float subsample = 1.0f, mse_threshold = 1e-5f, resize = 1.0f;
	struct IO { std::string target, source, output, visualization; } io;
	struct Viz { float phi = 0.4f, theta = 0.0f; bool spin_after_finish = false; } viz;
	struct Range { float xmin, xmax, ymin, ymax, zmin, zmax; int search_depth; } rotation{}, translation{};
	goicp_config raw{};     // the parsed C struct (goicp_params_from_config(&config.raw, &params) hands the ranges to the engine)

	explicit Config(const std::string toml_filepath)
	{
		goicp_config& c = raw;
		check(goicp_config_load(toml_filepath.c_str(), &c));
		mode = c.mode; trim = c.trim != 0; subsample = c.subsample; mse_threshold = c.mse_threshold; resize = c.resize;
		io = {c.target, c.source, c.output, c.visualization};
		viz.phi = c.viz_phi; viz.theta = c.viz_theta; viz.spin_after_finish = c.viz_spin_after_finish != 0;
		rotation = {c.rot_min[0], c.rot_max[0], c.rot_min[1], c.rot_max[1], c.rot_min[2], c.rot_max[2], c.rot_search_depth};
		translation = {c.trans_min[0], c.trans_max[0], c.trans_min[1], c.trans_max[1], c.trans_min[2], c.trans_max[2], c.trans_search_depth};
	}
};

// size_t load_cloud(path, subsample, resize, cloud) (src/common.cpp:205-228): appends to `cloud`
template <class Point3>
size_t load_cloud(const std::string& filepath, const float& subsample, const float& resize, std::vector<Point3>& cloud,
                  uint64_t seed = 0)
{
	static_assert(sizeof(Point3) == 3 * sizeof(float), "Point3 must be three packed floats (e.g. glm::vec3)");
	float* xyz = nullptr;
	size_t n = 0;
	check(goicp_cloud_load(filepath.c_str(), subsample, resize, seed, &xyz, &n));
	const Point3* p = reinterpret_cast<const Point3*>(xyz);
	cloud.insert(cloud.end(), p, p + n);
	goicp_cloud_free(xyz);
	return n;
}

namespace icp {

// src/fgoicp/fgoicp_common.hpp:139-167.  The engine owns its HIP stream; the pool only keeps call sites compiling.
class StreamPool {
public:
	explicit StreamPool(size_t size) : size_(size) {}
	size_t size() const { return size_; }
private:
	size_t size_;
};

// Rotation cube (src/fgoicp/fgoicp_common.hpp:31-104) with the CPU path's parametrisation, which is the one the
// engine implements (src/goicp/jly_goicp.h:44-57, jly_goicp.cpp:433-467): (x,y,z) = the cube CENTRE as an angle-axis
// vector (radians), span = half edge length, R = its Rodrigues matrix.
struct Rotation {
	float x, y, z, r;
	Mat3 R;
	Rotation() : Rotation(0.f, 0.f, 0.f) {}
	Rotation(float x_, float y_, float z_) : x(x_), y(y_), z(z_), r(std::sqrt(x_ * x_ + y_ * y_ + z_ * z_)), R(1.0f)
	{
		const float v[3] = {x, y, z};
		float rows[9];
		goicp_rodrigues(v, rows);
		R = from_rows(rows);
	}
	bool in_SO3() const { return r <= 3.14159265358979f; }
};
struct RotNode {
	Rotation q;
	float span;
	float lb, ub;
	RotNode(float x, float y, float z, float span_, float lb_, float ub_) : q(x, y, z), span(span_), lb(lb_), ub(ub_) {}
	friend bool operator<(const RotNode& a, const RotNode& b) { return a.lb == b.lb ? a.span < b.span : a.lb > b.lb; }
	// pi-ball test of GoICP::OuterBnB (src/goicp/jly_goicp.cpp:443)
	bool overlaps_SO3() const { return (double)q.r - 1.732050808 * (double)(2 * span) / 2 <= 3.1415926536; }
	// level l <=> cube width 2*pi / 2^l (src/goicp/jly_goicp.cpp:148-160: one radius table per level)
	int level() const
	{
		const double l = std::log2(3.1415926536 / (double)span);
		return l <= 0 ? 0 : (int)(l + 0.5);
	}
};
struct TransNode {            // centre + half-width (src/fgoicp/fgoicp_common.hpp:108-129)
	Vec3 t;
	float span;
	float lb, ub;
	TransNode(float x, float y, float z, float span_, float lb_, float ub_) : t(x, y, z), span(span_), lb(lb_), ub(ub_) {}
	friend bool operator<(const TransNode& a, const TransNode& b) { return a.lb == b.lb ? a.span < b.span : a.lb > b.lb; }
};

class Registration {
public:
	template <class Point3>
	Registration(const std::vector<Point3>& pct, size_t nt, const std::vector<Point3>& pcs, size_t ns, float mse_threshold = 1e-3f,
	             const goicp_params* params = nullptr)
	{
		static_assert(sizeof(Point3) == 3 * sizeof(float), "Point3 must be three packed floats");
		goicp_params p;
		if (params) p = *params; else goicp_params_default(&p);
		p.mse_threshold = mse_threshold;
		check(goicp_create(&p, reinterpret_cast<const float*>(pct.data()), nt, reinterpret_cast<const float*>(pcs.data()), ns, &h_));
	}
	~Registration() { goicp_destroy(h_); }
	Registration(const Registration&) = delete;
	Registration& operator=(const Registration&) = delete;

	using BoundsResult_t = std::tuple<std::vector<float>, std::vector<float>>;   // (lb, ub) as the reference

	// float compute_sse_error(glm::mat3 R, glm::vec3 t) const        (registration.hpp:96)
	float compute_sse_error(Mat3 R, Vec3 t) const
	{
		float rows[9], tt[3] = {t[0], t[1], t[2]}, sse = 0.f;
		to_rows(R, rows);
		check(goicp_eval_sse(h_, rows, tt, &sse));
		return sse;
	}
	// BoundsResult_t compute_sse_error(RotNode&, std::vector<TransNode>&, bool fix_rot, StreamPool&) const   (registration.hpp:97)
	BoundsResult_t compute_sse_error(RotNode& rnode, std::vector<TransNode>& tnodes, bool fix_rot, StreamPool&) const
	{
		return bounds(rnode.q.R, fix_rot ? -1 : rnode.level(), tnodes);
	}
	// the same with an explicit rotation level (rot_level < 0 <=> fix_rot)
	BoundsResult_t compute_sse_error(const Mat3& R, int rot_level, const std::vector<TransNode>& tnodes) const { return bounds(R, rot_level, tnodes); }
	goicp_handle handle() const { return h_; }

private:
	BoundsResult_t bounds(const Mat3& R, int rot_level, const std::vector<TransNode>& tnodes) const
	{
		std::vector<float> cubes(4 * tnodes.size()), lb(tnodes.size()), ub(tnodes.size());
		for (size_t i = 0; i < tnodes.size(); i++) {
			cubes[4 * i] = tnodes[i].t[0]; cubes[4 * i + 1] = tnodes[i].t[1]; cubes[4 * i + 2] = tnodes[i].t[2];
			cubes[4 * i + 3] = 2 * tnodes[i].span;
		}
		float rows[9];
		to_rows(R, rows);
		check(goicp_eval_bounds(h_, rows, cubes.data(), tnodes.size(), rot_level, ub.data(), lb.data()));
		return {lb, ub};
	}
	goicp_handle h_ = nullptr;
};

class IterativeClosestPoint3D {
public:
	IterativeClosestPoint3D(const Registration& reg, size_t max_iter, float convergence_threshold, Mat3 R, Vec3 t)
	    : reg_(reg), max_iter_(max_iter), thr_(convergence_threshold), R_(R), t_(t) {}
	// the reference's argument list (icp3d.hpp:30): the clouds already live in `reg`
	template <class PointCloud>
	IterativeClosestPoint3D(const Registration& reg, const PointCloud&, const PointCloud&, size_t max_iter, float convergence_threshold, Mat3 R, Vec3 t)
	    : IterativeClosestPoint3D(reg, max_iter, convergence_threshold, R, t) {}
	using Result_t = std::tuple<float, Mat3, Vec3>;
	Result_t run(Mat3& curR, Vec3& curT)
	{
		float err = 0.f, rows[9], tt[3] = {t_[0], t_[1], t_[2]};
		int32_t it = 0;
		to_rows(R_, rows);
		check(goicp_icp_run(reg_.handle(), rows, tt, (int32_t)max_iter_, thr_, &err, &it));
		R_ = from_rows(rows); t_ = from_xyz(tt);
		curR = R_; curT = t_;
		return Result_t{err, R_, t_};
	}

private:
	const Registration& reg_;
	size_t max_iter_;
	float thr_;
	Mat3 R_;
	Vec3 t_;
};

// icp::FastGoICP (src/fgoicp/fgoicp.hpp:11-69).  As in the reference the worker thread (run()) keeps the public
// members optR / optT / curR / curT / finished current and the reader locks `mtx` while it reads them
// (src/goicp_kernel.cu:164-177) -- here the writes happen under the same mutex (goicp_set_progress_callback), which
// the reference forgot (src/fgoicp/fgoicp.cpp:68-69,85-86).
class FastGoICP {
public:
	template <class Point3>
	FastGoICP(std::vector<Point3>& pct, std::vector<Point3>& pcs, float mse_threshold, std::mutex& mtx_,
	          const goicp_params* params = nullptr)
	    : curR(1.0f), optR(1.0f), curT(0.0f), optT(0.0f), finished(false), mtx(mtx_),
	      registration(pct, pct.size(), pcs, pcs.size(), mse_threshold, params)
	{
		check(goicp_set_progress_callback(registration.handle(), &FastGoICP::on_progress, this));
		sync();
	}
	~FastGoICP() { goicp_set_progress_callback(registration.handle(), nullptr, nullptr); }
	FastGoICP(const FastGoICP&) = delete;

	void run()
	{
		check(goicp_register(registration.handle()));
		sync();
	}
	void cancel() { goicp_cancel(registration.handle()); }
	float get_best_error() const { return best_sse; }       // fgoicp.hpp:34; read under `mtx` like the other members
	void sync()
	{
		goicp_result r;
		check(goicp_poll(registration.handle(), &r));
		std::lock_guard<std::mutex> lk(mtx);
		take(r);
	}
	void write_output(const std::string& path) { check(goicp_result_write_toml(registration.handle(), path.c_str())); }
	void write_visualization(const std::string& path) { check(goicp_result_write_ply(registration.handle(), path.c_str())); }

	// For visualization (fgoicp.hpp:66-69)
	Mat3 curR, optR;
	Vec3 curT, optT;
	bool finished;

private:
	static void on_progress(const goicp_result* r, void* self)
	{
		FastGoICP* f = static_cast<FastGoICP*>(self);
		std::lock_guard<std::mutex> lk(f->mtx);
		f->take(*r);
	}
	void take(const goicp_result& r)
	{
		optR = from_rows(r.optR); curR = from_rows(r.curR);
		optT = from_xyz(r.optT); curT = from_xyz(r.curT);
		best_sse = r.best_sse;
		finished = r.finished != 0;
	}
	float best_sse = 1e10f;
	std::mutex& mtx;

public:
	Registration registration;
};

}  // namespace icp

// ------------------------------------------------------------------------------------------------------------------
// Step API (src/icp_kernel.h:9-13, src/goicp_kernel.h:6-10).  The reference's step functions work on global device
// buffers set up by PointCloud::initBuffers(Ybuffer = data, Xbuffer = model) (src/kernel.cu:60-110) and publish the
// moved cloud to the viewer; here the "global buffers" are one engine instance, and the moved source cloud is
// available from step_positions().  All arithmetic runs in the HIP library -- also for ICP::CPUStep, whose host
// vectors are updated from the device result (this engine has no CPU compute path).
// ------------------------------------------------------------------------------------------------------------------
struct StepGlobals {
	std::unique_ptr<icp::Registration> reg;
	size_t numDataPoints = 0, numModelPoints = 0;
	bool goicp_finished = false;       // the reference's global flag (src/main.cpp:27, goicp_kernel.cu:167)
	float sse_threshold = 0.f;         // src/main.cpp:42
};
inline StepGlobals& step_globals()
{
	static StepGlobals g;
	return g;
}

namespace PointCloud {
template <class Point3>
inline void initBuffers(std::vector<Point3>& Ybuffer, std::vector<Point3>& Xbuffer, const goicp_params* params = nullptr)
{
	StepGlobals& g = step_globals();
	g.reg.reset(new icp::Registration(Xbuffer, Xbuffer.size(), Ybuffer, Ybuffer.size(), 1e-3f, params));
	g.numDataPoints = Ybuffer.size();
	g.numModelPoints = Xbuffer.size();
}
inline void cleanupBuffers() { step_globals().reg.reset(); }
}  // namespace PointCloud

namespace ICP {
inline void require_buffers()
{
	if (!step_globals().reg) throw std::runtime_error("ICP step: PointCloud::initBuffers was not called");
}
// one ICP iteration on the device-resident clouds (exact NN through the engine's own tree in both forms)
inline void naiveGPUStep()
{
	require_buffers();
	check(goicp_icp_step(step_globals().reg->handle()));
}
// kdTreeGPUStep(KDTree&, PointCloudAdaptor&, FlattenedKDTree*): the reference's host / flattened trees are not needed
template <class... TreeArgs>
inline void kdTreeGPUStep(TreeArgs&&...) { naiveGPUStep(); }
// CPUStep(dataBuffer, modelBuffer): one iteration, then the caller's data vector is moved as the reference does in place
template <class Point3>
inline void CPUStep(std::vector<Point3>& dataBuffer, std::vector<Point3>& modelBuffer)
{
	StepGlobals& g = step_globals();
	if (!g.reg) PointCloud::initBuffers(dataBuffer, modelBuffer);
	if (dataBuffer.size() != g.numDataPoints) throw std::invalid_argument("ICP::CPUStep: data buffer size changed");
	naiveGPUStep();
	goicp_result r;
	check(goicp_poll(g.reg->handle(), &r));
	static_assert(sizeof(Point3) == 3 * sizeof(float), "Point3 must be three packed floats");
	check(goicp_transform_source(g.reg->handle(), r.curR, r.curT, reinterpret_cast<float*>(dataBuffer.data())));
}
// accumulated pose of the step API and the source cloud under it (what the viewer draws)
inline void step_pose(Mat3& R, Vec3& t)
{
	require_buffers();
	goicp_result r;
	check(goicp_poll(step_globals().reg->handle(), &r));
	R = from_rows(r.curR); t = from_xyz(r.curT);
}
template <class Point3>
inline void step_positions(std::vector<Point3>& out)
{
	require_buffers();
	StepGlobals& g = step_globals();
	goicp_result r;
	check(goicp_poll(g.reg->handle(), &r));
	out.resize(g.numDataPoints);
	check(goicp_transform_source(g.reg->handle(), r.curR, r.curT, reinterpret_cast<float*>(out.data())));
}
// goicpGPUStep(fgoicp, prev_optR, prev_optT, mtx) (src/goicp_kernel.cu:152-206): the viewer's poll.  Returns whether the
// optimum changed (the reference redraws in that case); sets the global goicp_finished exactly as the reference does.
inline bool goicpGPUStep(const icp::FastGoICP* fgoicp, Mat3& prev_optR, Vec3& prev_optT, std::mutex& mtx)
{
	StepGlobals& g = step_globals();
	bool updated;
	float currentError;
	{
		std::lock_guard<std::mutex> lock(mtx);
		g.goicp_finished = fgoicp->finished;
		updated = (prev_optR != fgoicp->optR || prev_optT != fgoicp->optT);
		currentError = fgoicp->get_best_error();
		prev_optR = fgoicp->optR;
		prev_optT = fgoicp->optT;
	}
	float thr = g.sse_threshold;
	if (thr <= 0.f) goicp_thresholds(fgoicp->registration.handle(), &thr, nullptr);
	if (currentError <= thr) g.goicp_finished = true;
	return updated;
}
}  // namespace ICP

}  // namespace goicp_mi355
