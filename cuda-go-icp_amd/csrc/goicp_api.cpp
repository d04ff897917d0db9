// C ABI of libgoicp_mi355.so (declared in include/goicp_mi355.h): exception -> status translation
// around goicp::Engine and the config / cloud IO helpers.
#include "../../include/goicp_mi355.h"

#include <cstring>
#include <new>
#include <string>

#include "config_io.hpp"
#include "engine.hpp"

struct goicp_engine {
	goicp::Engine* e;
};

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
	g_err = msg;
	return code;
}

template <class F> int guarded(F&& f)
{
	try {
		f();
		return GOICP_OK;
	} catch (const goicp::ConfigError& e) { return fail(GOICP_ERR_CONFIG, e.what());
	} catch (const goicp::IoError& e) { return fail(GOICP_ERR_IO, e.what());
	} catch (const std::invalid_argument& e) { return fail(GOICP_ERR_INVALID, e.what());
	} catch (const std::bad_alloc&) { return fail(GOICP_ERR_INTERNAL, "out of host memory");
	} catch (const std::runtime_error& e) {
		std::string m = e.what();
		return fail(m.find("no HIP device") != std::string::npos ? GOICP_ERR_NO_DEVICE : GOICP_ERR_DEVICE, m);
	} catch (const std::exception& e) { return fail(GOICP_ERR_INTERNAL, e.what());
	} catch (...) { return fail(GOICP_ERR_INTERNAL, "unknown error"); }
}

#define REQUIRE(c) do { if (!(c)) return fail(GOICP_ERR_INVALID, std::string(__func__) + ": invalid argument (" #c ")"); } while (0)

void fill_counters(goicp_counters* o, const goicp::Counters& c)
{
	o->rot_pops = c.rot_pops; o->trans_pops = c.trans_pops; o->cubes = c.cubes; o->inner_calls = c.inner_calls;
	o->icp_runs = c.icp_runs; o->icp_iters = c.icp_iters; o->bounds_launches = c.bounds_launches;
}

void fill_result(goicp_result* out, const goicp::Result& r)
{
	std::memcpy(out->optR, r.optR, sizeof(r.optR)); std::memcpy(out->optT, r.optT, sizeof(r.optT));
	std::memcpy(out->curR, r.curR, sizeof(r.curR)); std::memcpy(out->curT, r.curT, sizeof(r.curT));
	out->best_sse = r.best_sse; out->finished = r.finished;
	fill_counters(&out->counters, r.counters);
	out->dt_build_ms = r.dt_build_ms; out->register_ms = r.register_ms;
}

}  // namespace

extern "C" {

const char* goicp_last_error(void) { return g_err.c_str(); }
int goicp_abi_version(void) { return GOICP_ABI_VERSION; }

int goicp_config_load(const char* toml_path, goicp_config* out)
{
	REQUIRE(toml_path && out);
	return guarded([&] { goicp::load_config(toml_path, out); });
}

int goicp_cloud_load(const char* path, float subsample, float resize, uint64_t seed, float** xyz, size_t* n)
{
	REQUIRE(path && xyz && n);
	*xyz = nullptr; *n = 0;
	return guarded([&] {
		std::vector<float> v;
		goicp::load_cloud(path, subsample, resize, seed, v);
		float* p = new float[v.size() ? v.size() : 1];
		std::memcpy(p, v.data(), v.size() * sizeof(float));
		*xyz = p; *n = v.size() / 3;
	});
}

void goicp_cloud_free(float* xyz) { delete[] xyz; }

void goicp_params_default(goicp_params* p)
{
	if (!p) return;
	goicp::Params d;
	p->dt_size = d.dt_size; p->dt_expand = d.dt_expand; p->mse_threshold = d.mse_threshold; p->dt_layout = d.dt_layout;
	p->device = d.device; p->trans_batch = d.trans_batch; p->wide_children = d.wide_children;
	p->icp_max_iter = d.icp_max_iter; p->verbose = d.verbose; p->morton_sort = d.morton_sort; p->rot_batch = d.rot_batch; p->trim_fraction = d.trim_fraction; p->kd_gpu_build = d.kd_gpu_build;
	p->use_rot_range = d.use_rot_range; p->use_trans_range = d.use_trans_range;
	for (int k = 0; k < 3; k++) {
		p->rot_min[k] = d.rot_min[k]; p->rot_max[k] = d.rot_max[k];
		p->trans_min[k] = d.trans_min[k]; p->trans_max[k] = d.trans_max[k];
	}
	p->rot_search_depth = d.rot_search_depth; p->trans_search_depth = d.trans_search_depth;
}

void goicp_params_from_config(const goicp_config* c, goicp_params* p)
{
	if (!p) return;
	goicp_params_default(p);
	if (!c) return;
	p->mse_threshold = c->mse_threshold;
	if (c->has_rotation_range) {
		p->use_rot_range = 1;
		for (int k = 0; k < 3; k++) { p->rot_min[k] = c->rot_min[k]; p->rot_max[k] = c->rot_max[k]; }
		p->rot_search_depth = c->rot_search_depth;
	}
	if (c->has_translation_range) {
		p->use_trans_range = 1;
		for (int k = 0; k < 3; k++) { p->trans_min[k] = c->trans_min[k]; p->trans_max[k] = c->trans_max[k]; }
		p->trans_search_depth = c->trans_search_depth;
	}
}

int goicp_create(const goicp_params* params, const float* target_xyz, size_t n_target, const float* source_xyz,
                 size_t n_source, goicp_handle* out)
{
	REQUIRE(out);
	*out = nullptr;
	REQUIRE(target_xyz && source_xyz && n_target > 0 && n_source > 0);
	return guarded([&] {
		goicp::Params p;
		if (params) {
			p.dt_size = params->dt_size; p.dt_expand = params->dt_expand; p.mse_threshold = params->mse_threshold;
			p.dt_layout = params->dt_layout; p.device = params->device; p.trans_batch = params->trans_batch;
			p.wide_children = params->wide_children; p.icp_max_iter = params->icp_max_iter; p.verbose = params->verbose;
			p.morton_sort = params->morton_sort;
			if (params->rot_batch > 0) p.rot_batch = params->rot_batch;
			p.trim_fraction = params->trim_fraction;
			p.kd_gpu_build = params->kd_gpu_build;
			p.use_rot_range = params->use_rot_range; p.use_trans_range = params->use_trans_range;
			for (int k = 0; k < 3; k++) {
				p.rot_min[k] = params->rot_min[k]; p.rot_max[k] = params->rot_max[k];
				p.trans_min[k] = params->trans_min[k]; p.trans_max[k] = params->trans_max[k];
			}
			p.rot_search_depth = params->rot_search_depth; p.trans_search_depth = params->trans_search_depth;
		}
		goicp_engine* h = new goicp_engine{nullptr};
		try { h->e = new goicp::Engine(p, target_xyz, n_target, source_xyz, n_source); }
		catch (...) { delete h; throw; }
		*out = h;
	});
}

int goicp_destroy(goicp_handle h)
{
	if (!h) return GOICP_OK;
	return guarded([&] { delete h->e; delete h; });
}

int goicp_thresholds(goicp_handle h, float* sse_threshold, int32_t* inliers)
{
	REQUIRE(h);
	if (sse_threshold) *sse_threshold = h->e->sse_threshold();
	if (inliers) *inliers = h->e->inliers();
	return GOICP_OK;
}

int goicp_device(goicp_handle h, int32_t* ordinal)
{
	REQUIRE(h && ordinal);
	*ordinal = h->e->device();
	return GOICP_OK;
}

int goicp_set_progress_callback(goicp_handle h, goicp_progress_fn cb, void* user)
{
	REQUIRE(h);
	if (!cb) { h->e->set_progress_callback(nullptr); return GOICP_OK; }
	h->e->set_progress_callback([cb, user](const goicp::Result& r) {
		goicp_result out;
		fill_result(&out, r);
		cb(&out, user);
	});
	return GOICP_OK;
}

int goicp_probe_gather(goicp_handle h, int32_t mode, size_t window_bytes, double* lookups_per_s)
{
	REQUIRE(h && lookups_per_s && (mode == 0 || mode == 1));
	return guarded([&] { *lookups_per_s = h->e->probe_gather(mode, window_bytes); });
}

int goicp_debug_kabsch(const float H[9], float R[9])
{
	REQUIRE(H && R);
	return guarded([&] { goicp::debug_kabsch(H, R); });
}

int goicp_dt_info(goicp_handle h, int32_t* V, double* scale, double origin[3])
{
	REQUIRE(h);
	const goicp::DtDesc& d = h->e->dt();
	if (V) *V = d.V;
	if (scale) *scale = d.scale;
	if (origin) { origin[0] = d.xmin; origin[1] = d.ymin; origin[2] = d.zmin; }
	return GOICP_OK;
}

int goicp_dt_download(goicp_handle h, float* grid)
{
	REQUIRE(h && grid);
	return guarded([&] { h->e->dt_download(grid); });
}

int goicp_eval_bounds(goicp_handle h, const float R[9], const float* cubes, size_t B, int32_t level, float* ub, float* lb)
{
	REQUIRE(h && R && (B == 0 || (cubes && ub && lb)));
	return guarded([&] { h->e->eval_bounds(R, cubes, B, level, ub, lb); });
}

int goicp_eval_bounds_batch(goicp_handle h, const float* rots, size_t K, const goicp_cube* cubes, size_t B, float* ub, float* lb)
{
	REQUIRE(h && (B == 0 || (rots && K > 0 && cubes && ub && lb)));
	static_assert(sizeof(goicp_cube) == sizeof(goicp::CubeRec), "goicp_cube layout");
	return guarded([&] { h->e->eval_bounds_batch(rots, K, reinterpret_cast<const goicp::CubeRec*>(cubes), B, ub, lb); });
}

int goicp_eval_bounds_device(goicp_handle h, const void* d_rots, const void* d_cubes, size_t B, void* d_ub, void* d_lb, void* stream)
{
	REQUIRE(h && d_rots && d_cubes && d_ub && d_lb && B > 0);
	return guarded([&] {
		h->e->eval_bounds_dev(static_cast<const goicp::Rot9*>(d_rots), static_cast<const goicp::CubeRec*>(d_cubes), (int)B,
		                      static_cast<float*>(d_ub), static_cast<float*>(d_lb), static_cast<hipStream_t>(stream));
	});
}

int goicp_time_bounds_device(goicp_handle h, const void* d_rots, const void* d_cubes, size_t B, void* d_ub, void* d_lb,
                             int32_t iters, float* ms)
{
	REQUIRE(h && d_rots && d_cubes && d_ub && d_lb && B > 0 && iters > 0 && ms);
	return guarded([&] {
		*ms = h->e->time_bounds_dev(static_cast<const goicp::Rot9*>(d_rots), static_cast<const goicp::CubeRec*>(d_cubes),
		                            (int)B, static_cast<float*>(d_ub), static_cast<float*>(d_lb), iters);
	});
}

float goicp_rot_coeff(goicp_handle h, int32_t level) { return h ? h->e->rot_coeff(level) : 0.f; }
float goicp_trans_delta(float w) { return (float)(1.732050808 / 2.0 * (double)w); }
void goicp_rodrigues(const float v[3], float R[9]) { if (v && R) goicp::rodrigues(v[0], v[1], v[2], R); }

int goicp_eval_sse(goicp_handle h, const float R[9], const float t[3], float* sse)
{
	REQUIRE(h && R && t && sse);
	return guarded([&] { *sse = h->e->eval_sse(R, t); });
}

int goicp_inner_bnb(goicp_handle h, const float R[9], int32_t level, float incumbent, float* value, float best_node[4],
                    goicp_counters* counters)
{
	REQUIRE(h && R && value);
	return guarded([&] {
		goicp::Counters c;
		*value = h->e->inner_bnb(R, level, incumbent, best_node, &c);
		if (counters) fill_counters(counters, c);
	});
}

int goicp_icp_run(goicp_handle h, float R[9], float t[3], int32_t max_iter, float err_diff, float* err, int32_t* iters)
{
	REQUIRE(h && R && t && max_iter >= 0);
	return guarded([&] {
		int it = 0;
		float e = h->e->icp_run(R, t, max_iter, err_diff, &it);
		if (err) *err = e;
		if (iters) *iters = it;
	});
}

int goicp_time_icp_pass(goicp_handle h, const float R[9], const float t[3], int32_t iters, float* ms)
{
	REQUIRE(h && R && t && iters > 0 && ms);
	return guarded([&] { *ms = h->e->time_icp_pass(R, t, iters); });
}

int goicp_nn_query(goicp_handle h, const float* q, size_t n, int32_t* index, float* dist_sq)
{
	REQUIRE(h && (n == 0 || (q && index && dist_sq)));
	return guarded([&] { h->e->nn_query(q, n, index, dist_sq); });
}

int goicp_icp_step(goicp_handle h)
{
	REQUIRE(h);
	return guarded([&] { h->e->icp_step(); });
}

int goicp_register(goicp_handle h)
{
	REQUIRE(h);
	return guarded([&] { h->e->run(); });
}

int goicp_cancel(goicp_handle h)
{
	REQUIRE(h);
	h->e->cancel();
	return GOICP_OK;
}

int goicp_poll(goicp_handle h, goicp_result* out)
{
	REQUIRE(h && out);
	return guarded([&] {
		fill_result(out, h->e->poll());
	});
}

int goicp_result_write_toml(goicp_handle h, const char* path)
{
	REQUIRE(h && path);
	return guarded([&] { goicp::write_result_toml(path, h->e->poll(), h->e->n_source(), h->e->n_target(), h->e->sse_threshold()); });
}

int goicp_result_write_ply(goicp_handle h, const char* path)
{
	REQUIRE(h && path);
	return guarded([&] {
		goicp::Result r = h->e->poll();
		std::vector<float> moved(3 * h->e->n_source());
		h->e->source_transformed(r.optR, r.optT, moved.data());
		goicp::write_viz_ply(path, h->e->target_xyz(), h->e->n_target(), moved.data(), h->e->n_source());
	});
}

int goicp_transform_source(goicp_handle h, const float R[9], const float t[3], float* out_xyz)
{
	REQUIRE(h && R && t && out_xyz);
	return guarded([&] { h->e->source_transformed(R, t, out_xyz); });
}

int goicp_set_shard(goicp_handle h, int32_t rank, int32_t world)
{
	REQUIRE(h && world >= 1 && rank >= 0 && rank < world);
	h->e->set_shard(rank, world);
	return GOICP_OK;
}

int goicp_register_begin(goicp_handle h)
{
	REQUIRE(h);
	return guarded([&] { h->e->register_begin(); });
}

int goicp_register_step(goicp_handle h, int32_t max_rot_pops, goicp_step_status* out)
{
	REQUIRE(h && out && max_rot_pops > 0);
	return guarded([&] {
		goicp::StepStatus s = h->e->register_step(max_rot_pops);
		out->finished = s.finished; out->early_exit = s.early_exit; out->best_sse = s.best_sse;
		out->frontier_lb = s.frontier_lb; out->rot_pops = s.rot_pops;
	});
}

int goicp_offer_best(goicp_handle h, float sse, const float R[9], const float t[3])
{
	REQUIRE(h && R && t);
	return guarded([&] { h->e->offer_global_best(sse, R, t); });
}

int goicp_register_end(goicp_handle h)
{
	REQUIRE(h);
	return guarded([&] { h->e->register_end(); });
}

}  // extern "C"
