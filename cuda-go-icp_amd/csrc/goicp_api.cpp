// C ABI of libgoicp_mi355.so (declared in include/goicp_mi355.h): exception -> status translation
// around goicp::Engine and the config / cloud IO helpers.
#include "../../include/goicp_mi355.h"

#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "comm.hpp"
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
	o->icp_runs = c.icp_runs; o->icp_iters = c.icp_iters; o->bounds_launches = c.bounds_launches; o->queue_fallbacks = c.queue_fallbacks;
	o->tile_expansions = c.tile_expansions; o->lane_batches = c.lane_batches;
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
#ifndef GOICP_KERNEL_HASH
#define GOICP_KERNEL_HASH "unknown"
#endif
const char* goicp_kernel_source_hash(void) { return GOICP_KERNEL_HASH; }

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
	p->icp_fused = d.icp_fused; p->bounds_fp16 = d.bounds_fp16; p->icp_nn_cache = d.icp_nn_cache; p->flow = d.flow; p->adaptive_k = d.adaptive_k; p->queue_cap = d.queue_cap; p->device_queues = d.device_queues;
	p->lds_tiles = d.lds_tiles; p->tile_spread_vox = d.tile_spread_vox; p->tile_min = d.tile_min; p->stale_widen = d.stale_widen; p->ub_tiebreak = d.ub_tiebreak; p->icp_point_seed = d.icp_point_seed; p->ub_share = d.ub_share; p->twin_fusion = d.twin_fusion; p->sort_items = d.sort_items; p->stale_compact = d.stale_compact; p->stream_priority = d.stream_priority; p->lanes = d.lanes; p->lane_min_searches = d.lane_min_searches;
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
			p.icp_fused = params->icp_fused; p.bounds_fp16 = params->bounds_fp16; p.icp_nn_cache = params->icp_nn_cache; p.flow = params->flow; p.adaptive_k = params->adaptive_k; p.queue_cap = params->queue_cap; p.device_queues = params->device_queues;
			p.lds_tiles = params->lds_tiles; p.stale_widen = params->stale_widen; p.ub_tiebreak = params->ub_tiebreak; p.icp_point_seed = params->icp_point_seed; p.ub_share = params->ub_share; p.twin_fusion = params->twin_fusion; p.sort_items = params->sort_items; p.stale_compact = params->stale_compact; p.stream_priority = params->stream_priority; p.lanes = params->lanes; p.lane_min_searches = params->lane_min_searches;
			if (params->tile_spread_vox > 0.f) p.tile_spread_vox = params->tile_spread_vox;
			if (params->tile_min > 0) p.tile_min = params->tile_min;
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
	REQUIRE(h && lookups_per_s && (mode == 0 || mode == 1 || mode == 2 || mode == 4 || mode == 8 || mode == 16 || mode == 32));
	return guarded([&] { *lookups_per_s = h->e->probe_gather(mode, window_bytes); });
}

int goicp_debug_cache_hits(goicp_handle h, const float R[9], const float t[3], int64_t* hits)
{
	REQUIRE(h && R && t && hits);
	return guarded([&] { *hits = h->e->debug_cache_hits(R, t); });
}

int goicp_debug_bounds_tile(goicp_handle h, const float* rots9, const float* parents4, int32_t nseg, int32_t n, int32_t level, int32_t chunks,
                            float* ub_tile, float* lb_tile, float* ub_direct, float* lb_direct, float ms[2], uint32_t stats[2])
{
	REQUIRE(h && rots9 && parents4 && ub_tile && lb_tile && ub_direct && lb_direct && ms && stats);
	return guarded([&] { h->e->debug_bounds_tile(rots9, parents4, nseg, n, level, chunks, ub_tile, lb_tile, ub_direct, lb_direct, ms, stats); });
}

int goicp_debug_queue_expand(goicp_handle h, const float R[9], int32_t level, const float* parents4, int32_t n, float* ub_ubpass, float* lb_ubpass,
                             float* ub_lbpass, float* lb_lbpass, int32_t info[2])
{
	REQUIRE(h && R && parents4 && ub_ubpass && lb_ubpass && ub_lbpass && lb_lbpass && info && n >= 1);
	return guarded([&] {
		int inf[2] = {0, 0};
		h->e->debug_queue_expand(R, level, parents4, n, ub_ubpass, lb_ubpass, ub_lbpass, lb_lbpass, inf);
		info[0] = inf[0]; info[1] = inf[1];
	});
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

int goicp_eval_bounds_device_grouped(goicp_handle h, const void* d_rots, size_t n_rots, const void* d_cubes, size_t B, void* d_ub, void* d_lb, void* stream)
{
	REQUIRE(h && d_rots && d_cubes && d_ub && d_lb && B > 0 && n_rots >= 1 && n_rots <= 16);
	return guarded([&] {
		h->e->eval_bounds_dev_grouped(static_cast<const goicp::Rot9*>(d_rots), (int)n_rots, static_cast<const goicp::CubeRec*>(d_cubes), (int)B,
		                              static_cast<float*>(d_ub), static_cast<float*>(d_lb), static_cast<hipStream_t>(stream));
	});
}

int goicp_time_bounds_device_grouped(goicp_handle h, const void* d_rots, size_t n_rots, const void* d_cubes, size_t B, void* d_ub, void* d_lb, int32_t iters, float* ms)
{
	REQUIRE(h && d_rots && d_cubes && d_ub && d_lb && B > 0 && iters > 0 && ms && n_rots >= 1 && n_rots <= 16);
	return guarded([&] {
		*ms = h->e->time_bounds_dev(static_cast<const goicp::Rot9*>(d_rots), static_cast<const goicp::CubeRec*>(d_cubes), (int)B, static_cast<float*>(d_ub),
		                            static_cast<float*>(d_lb), iters, (int)n_rots);
	});
}

int goicp_reduce_min_device(goicp_handle h, const void* d_values, size_t n, void* d_min, void* d_argmin, void* stream)
{
	REQUIRE(h && d_values && d_min && n > 0 && n <= 0x7fffffffu && ((uintptr_t)d_values & 15) == 0);
	return guarded([&] {
		h->e->reduce_min_dev(static_cast<const float*>(d_values), (int)n, static_cast<float*>(d_min), static_cast<int*>(d_argmin), static_cast<hipStream_t>(stream));
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

int goicp_time_icp_pass_cached(goicp_handle h, const float R[9], const float t[3], int32_t iters, float* ms)
{
	REQUIRE(h && R && t && iters > 0 && ms);
	return guarded([&] { *ms = h->e->time_icp_pass(R, t, iters, true); });
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
	return guarded([&] { goicp::write_result_toml(path, h->e->poll(), h->e->n_source(), h->e->n_target(), h->e->sse_threshold(), h->e->inliers()); });
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

// ---- sharded registration: the engine as the protocol's callback table ---------------------------------------------
namespace {
goicp::Engine* E(void* ctx) { return static_cast<goicp_engine*>(ctx)->e; }
int eo_begin(void* c, int32_t rank, int32_t world) { return guarded([&] { E(c)->set_shard(rank, world); E(c)->register_begin(); }); }
int eo_step(void* c, int32_t max_pops, goicp_step_status* out) { return goicp_register_step(static_cast<goicp_engine*>(c), max_pops, out); }
int eo_pose(void* c, float* sse, float R[9], float t[3])
{
	return guarded([&] {
		const goicp::Result r = E(c)->poll();
		*sse = r.best_sse;
		std::memcpy(R, r.optR, sizeof(r.optR));
		std::memcpy(t, r.optT, sizeof(r.optT));
	});
}
int eo_offer(void* c, float sse, const float R[9], const float t[3]) { return guarded([&] { E(c)->offer_global_best(sse, R, t); }); }
int eo_qsize(void* c, int32_t* n) { return guarded([&] { *n = E(c)->queue_size(); }); }
int eo_donate(void* c, int32_t max_nodes, float* nodes7, int32_t* n) { return guarded([&] { *n = E(c)->donate(max_nodes, nodes7); }); }
int eo_receive(void* c, const float* nodes7, int32_t n) { return guarded([&] { E(c)->receive(nodes7, n); }); }
int eo_end(void* c) { return guarded([&] { E(c)->register_end(); }); }
}  // namespace

void goicp_shard_options_default(goicp_shard_options* out)
{
	if (!out) return;
	// the step follows the single-GPU driver's ramp up to 32 parents per batch: measured on 1/2/4/8 ranks (tools/shard_inflation.py) the
	// prove-the-optimum run inflates by 1.02 (4 ranks) / 1.10 (8) with it, in a quarter of the exchanges of a fixed 8
	out->rot_pops_per_step = 8; out->rebalance = 1; out->stale_exchange = 0; out->ramp_to = 32;
}

int goicp_run_sharded_opt(const goicp_shard_engine_ops* engine, const goicp_comm_ops* comm, const goicp_shard_options* opt, goicp_shard_stats* stats)
{
	REQUIRE(engine && comm && opt && engine->begin && engine->step && engine->pose && engine->offer && engine->queue_size && engine->donate &&
	        engine->receive && engine->end && comm->allreduce_min_u64 && comm->bcast && opt->rot_pops_per_step >= 1);
	g_err.clear();                  // a message left by an earlier, unrelated failure on this thread must not be reported for this run
	int rc = GOICP_ERR_INTERNAL;
	const int g = guarded([&] { rc = goicp::run_sharded(engine, comm, opt, stats); });
	if (g != GOICP_OK) return g;
	if (rc == GOICP_ERR_TIMEOUT) g_err = "sharded registration: a collective missed the communicator's deadline (a rank is lost); exit non-zero";
	else if (rc == GOICP_ERR_PEER) g_err = "sharded registration: rank " + std::to_string(stats ? stats->failed_rank : -1) + " reported a failure";
	else if (rc != GOICP_OK && g_err.empty()) g_err = "sharded registration: a callback failed";
	return rc;
}

int goicp_run_sharded(const goicp_shard_engine_ops* engine, const goicp_comm_ops* comm, int32_t rot_pops_per_step, int32_t rebalance,
                      goicp_shard_stats* stats)
{
	goicp_shard_options o;
	goicp_shard_options_default(&o);
	o.rot_pops_per_step = rot_pops_per_step; o.rebalance = rebalance;
	return goicp_run_sharded_opt(engine, comm, &o, stats);
}

int goicp_register_sharded_opt(goicp_handle h, const goicp_comm_ops* comm, const goicp_shard_options* opt, goicp_shard_stats* stats)
{
	REQUIRE(h && comm && opt);
	goicp_shard_engine_ops eo{};
	eo.ctx = h; eo.sse_threshold = h->e->sse_threshold();
	eo.begin = eo_begin; eo.step = eo_step; eo.pose = eo_pose; eo.offer = eo_offer; eo.queue_size = eo_qsize;
	eo.donate = eo_donate; eo.receive = eo_receive; eo.end = eo_end;
	return goicp_run_sharded_opt(&eo, comm, opt, stats);
}

int goicp_register_sharded(goicp_handle h, const goicp_comm_ops* comm, int32_t rot_pops_per_step, int32_t rebalance, goicp_shard_stats* stats)
{
	goicp_shard_options o;
	goicp_shard_options_default(&o);
	o.rot_pops_per_step = rot_pops_per_step; o.rebalance = rebalance;
	return goicp_register_sharded_opt(h, comm, &o, stats);
}

int goicp_comm_set_timeout_ms(goicp_comm_ops* comm, int32_t timeout_ms)
{
	REQUIRE(comm && comm->ctx && timeout_ms >= 1);
	const int rc = goicp::comm_set_timeout_ms(comm, timeout_ms);
	return rc == GOICP_OK ? rc : fail(rc, "goicp_comm_set_timeout_ms: not a communicator made by this library");
}

int goicp_thread_comm_create(int32_t world, goicp_comm_ops* out)
{
	REQUIRE(out && world >= 1 && world <= 1024);
	return guarded([&] { if (goicp::thread_comm_create(world, out) != GOICP_OK) throw std::bad_alloc(); });
}
int goicp_thread_comm_destroy(goicp_comm_ops* comm) { goicp::thread_comm_destroy(comm); return GOICP_OK; }

int goicp_rccl_unique_id(char id128[GOICP_RCCL_ID_BYTES])
{
	REQUIRE(id128);
	const int rc = goicp::rccl_unique_id(id128);
	return rc == GOICP_OK ? rc : fail(rc, "ncclGetUniqueId failed");
}

int goicp_rccl_comm_create(const char id128[GOICP_RCCL_ID_BYTES], int32_t rank, int32_t world, int32_t device, goicp_comm_ops* out)
{
	REQUIRE(id128 && out && world >= 1 && rank >= 0 && rank < world && device >= 0);
	const int rc = goicp::rccl_comm_create(id128, rank, world, device, out);
	return rc == GOICP_OK ? rc : fail(rc, "RCCL communicator creation failed (ncclCommInitRank / staging buffers)");
}

int goicp_rccl_comm_wrap(void* nccl_comm, int32_t rank, int32_t world, int32_t device, goicp_comm_ops* out)
{
	REQUIRE(nccl_comm && out && world >= 1 && rank >= 0 && rank < world && device >= 0);
	const int rc = goicp::rccl_comm_wrap(nccl_comm, rank, world, device, out);
	return rc == GOICP_OK ? rc : fail(rc, "RCCL communicator staging buffers could not be created");
}

int goicp_rccl_comm_destroy(goicp_comm_ops* comm) { return goicp::rccl_comm_destroy(comm); }

int goicp_register_multi_gpu(const goicp_params* params, const float* target_xyz, size_t n_target, const float* source_xyz, size_t n_source,
                             int32_t world, int32_t rot_pops_per_step, goicp_handle* out, goicp_shard_stats* stats)
{
	REQUIRE(out && target_xyz && source_xyz && n_target > 0 && n_source > 0 && world >= 1 && world <= 64 && rot_pops_per_step >= 1);
	*out = nullptr;
	goicp_params base;
	if (params) base = *params; else goicp_params_default(&base);
	std::vector<goicp_handle> hs((size_t)world, nullptr);
	std::vector<void*> raw((size_t)world, nullptr);
	std::vector<goicp_comm_ops> comms((size_t)world);
	std::vector<int> rcs((size_t)world, GOICP_OK);
	std::vector<std::string> errs((size_t)world);
	int rc = GOICP_OK;
	auto cleanup = [&](bool keep0) {
		for (int r = 0; r < world; r++) {
			goicp_rccl_comm_destroy(&comms[(size_t)r]);
			goicp::rccl_comm_destroy_raw(raw[(size_t)r]);
			if (!(keep0 && r == 0)) goicp_destroy(hs[(size_t)r]);
		}
	};
	for (int r = 0; r < world; r++) std::memset(&comms[(size_t)r], 0, sizeof(goicp_comm_ops));
	// one engine per GPU, created side by side (DT + k-d builds overlap)
	{
		std::vector<std::thread> th;
		for (int r = 0; r < world; r++)
			th.emplace_back([&, r] {
				goicp_params p = base;
				p.device = r;
				rcs[(size_t)r] = goicp_create(&p, target_xyz, n_target, source_xyz, n_source, &hs[(size_t)r]);
				if (rcs[(size_t)r] != GOICP_OK) errs[(size_t)r] = goicp_last_error();
			});
		for (auto& t : th) t.join();
	}
	for (int r = 0; r < world; r++) if (rcs[(size_t)r] != GOICP_OK) { rc = rcs[(size_t)r]; g_err = errs[(size_t)r]; }
	if (rc == GOICP_OK && goicp::rccl_comm_init_all(world, raw.data()) != GOICP_OK) rc = fail(GOICP_ERR_DEVICE, "ncclCommInitAll failed");
	for (int r = 0; rc == GOICP_OK && r < world; r++) rc = goicp_rccl_comm_wrap(raw[(size_t)r], r, world, r, &comms[(size_t)r]);
	if (rc != GOICP_OK) { cleanup(false); return rc; }
	{
		std::vector<std::thread> th;
		for (int r = 0; r < world; r++)
			th.emplace_back([&, r] {
				rcs[(size_t)r] = goicp_register_sharded(hs[(size_t)r], &comms[(size_t)r], rot_pops_per_step, 1, stats ? &stats[r] : nullptr);
				if (rcs[(size_t)r] != GOICP_OK) errs[(size_t)r] = goicp_last_error();
			});
		for (auto& t : th) t.join();
	}
	for (int r = 0; r < world; r++) if (rcs[(size_t)r] != GOICP_OK) { rc = rcs[(size_t)r]; g_err = errs[(size_t)r]; }
	cleanup(rc == GOICP_OK);
	if (rc == GOICP_OK) *out = hs[0];
	return rc;
}

}  // extern "C"
