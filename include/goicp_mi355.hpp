// goicp_mi355.hpp -- header-only C++ shim over the C ABI (goicp_mi355.h) that re-creates the
// reference's class surface, so code written against
//     Config / load_cloud                      (src/common.h:133-180, src/common.cpp:205-228)
//     icp::FastGoICP{run,get_best_error,optR,optT,curR,curT,finished}   (src/fgoicp/fgoicp.hpp:11-69)
//     icp::Registration::compute_sse_error x2  (src/fgoicp/registration.hpp:96-97)
//     icp::IterativeClosestPoint3D::run        (src/fgoicp/icp3d.hpp:30-35)
// compiles against this engine.  The reference uses glm::vec3 / glm::mat3; to stay free of a glm
// dependency the shim is templated on any 3-float point type and exposes matrices as
// std::array<float,9> (row-major) -- INTEGRATION.md shows the two-line glm adaptor.
#pragma once
#include <array>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "goicp_mi355.h"

namespace goicp_mi355 {

using Mat3 = std::array<float, 9>;   // row-major
using Vec3 = std::array<float, 3>;

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

	explicit Config(const std::string toml_filepath)
	{
		goicp_config c;
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

struct TransNode { Vec3 t; float span; float lb, ub; };   // centre + half-width (src/fgoicp/fgoicp_common.hpp:108-129)

class Registration {
public:
	template <class Point3>
	Registration(const std::vector<Point3>& pct, size_t nt, const std::vector<Point3>& pcs, size_t ns, float mse_threshold,
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

	using BoundsResult_t = std::tuple<std::vector<float>, std::vector<float>>;   // (lb, ub) as the reference

	float compute_sse_error(const Mat3& R, const Vec3& t) const
	{
		float sse = 0.f;
		check(goicp_eval_sse(h_, R.data(), t.data(), &sse));
		return sse;
	}
	// rot_level < 0 <=> fix_rot
	BoundsResult_t compute_sse_error(const Mat3& R, int rot_level, const std::vector<TransNode>& tnodes) const
	{
		std::vector<float> cubes(4 * tnodes.size()), lb(tnodes.size()), ub(tnodes.size());
		for (size_t i = 0; i < tnodes.size(); i++) {
			cubes[4 * i] = tnodes[i].t[0]; cubes[4 * i + 1] = tnodes[i].t[1]; cubes[4 * i + 2] = tnodes[i].t[2];
			cubes[4 * i + 3] = 2 * tnodes[i].span;
		}
		check(goicp_eval_bounds(h_, R.data(), cubes.data(), tnodes.size(), rot_level, ub.data(), lb.data()));
		return {lb, ub};
	}
	goicp_handle handle() const { return h_; }

private:
	goicp_handle h_ = nullptr;
};

class IterativeClosestPoint3D {
public:
	IterativeClosestPoint3D(const Registration& reg, size_t max_iter, float convergence_threshold, Mat3 R, Vec3 t)
	    : reg_(reg), max_iter_(max_iter), thr_(convergence_threshold), R_(R), t_(t) {}
	using Result_t = std::tuple<float, Mat3, Vec3>;
	Result_t run(Mat3& curR, Vec3& curT)
	{
		float err = 0.f;
		int32_t it = 0;
		check(goicp_icp_run(reg_.handle(), R_.data(), t_.data(), (int32_t)max_iter_, thr_, &err, &it));
		curR = R_; curT = t_;
		return {err, R_, t_};
	}

private:
	const Registration& reg_;
	size_t max_iter_;
	float thr_;
	Mat3 R_;
	Vec3 t_;
};

// icp::FastGoICP (src/fgoicp/fgoicp.hpp:11-69).  optR/optT/curR/curT/finished are refreshed from a
// consistent snapshot by sync() (and at the end of run()); the viewer glue calls sync() where it
// used to lock `mtx` (src/goicp_kernel.cu:164).
class FastGoICP {
public:
	template <class Point3>
	FastGoICP(std::vector<Point3>& pct, std::vector<Point3>& pcs, float mse_threshold, std::mutex& mtx,
	          const goicp_params* params = nullptr)
	    : mtx(mtx), registration(pct, pct.size(), pcs, pcs.size(), mse_threshold, params)
	{
		sync();
	}
	void run()
	{
		check(goicp_register(registration.handle()));
		sync();
	}
	void cancel() { goicp_cancel(registration.handle()); }
	float get_best_error() const
	{
		goicp_result r;
		check(goicp_poll(registration.handle(), &r));
		return r.best_sse;
	}
	void sync()
	{
		goicp_result r;
		check(goicp_poll(registration.handle(), &r));
		std::lock_guard<std::mutex> lk(mtx);
		for (int i = 0; i < 9; i++) { optR[i] = r.optR[i]; curR[i] = r.curR[i]; }
		for (int i = 0; i < 3; i++) { optT[i] = r.optT[i]; curT[i] = r.curT[i]; }
		finished = r.finished != 0;
	}
	void write_output(const std::string& path) { check(goicp_result_write_toml(registration.handle(), path.c_str())); }
	void write_visualization(const std::string& path) { check(goicp_result_write_ply(registration.handle(), path.c_str())); }

	Mat3 curR{}, optR{};
	Vec3 curT{}, optT{};
	bool finished = false;

private:
	std::mutex& mtx;

public:
	Registration registration;
};

}  // namespace icp
}  // namespace goicp_mi355
