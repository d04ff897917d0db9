// Sanitizer driver for the HOST side of libgoicp_mi355 (make -C cuda-go-icp_amd/csrc asan | tsan).  No HIP call
// is made: it links only the translation units that are pure host code -- the sharding protocol and its in-process
// communicator (shard.cpp), the threaded k-d build (kdtree.cpp: parallel_tasks), the TOML / PLY / TXT readers and
// the result writers (config_io.cpp) -- and drives them from several threads.  Run on the CPU only, never on the GPU box.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <queue>
#include <random>
#include <thread>
#include <vector>

#include "comm.hpp"
#include "config_io.hpp"
#include "engine.hpp"

namespace {

int g_fail = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "host selftest: %s failed (line %d)\n", #c, __LINE__); g_fail++; } } while (0)

// A toy engine with the stepped interface: "cubes" are integers with a fixed pseudo-random lower bound and a leaf value;
// popping a cube may improve the incumbent and spawns two children.  Deterministic, so the sharded runs must all reach
// the same best value as a single rank.
struct ToyEngine {
	struct Cube { float lb; int id, depth; bool operator<(const Cube& o) const { return lb > o.lb; } };
	int fail_at_step = -1, fail_what = 0, steps = 0;     // fault injection: 1 step, 2 offer, 3 donate, 4 begin
	bool ended = false;
	std::priority_queue<Cube> q;
	float best = 1e9f, R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
	int rank = 0, world = 1;
	long long pops = 0;
	std::vector<int> widths;                             // max_pops of every step (the protocol's step rule)
	static float value(int id) { return 10.f + (float)((unsigned)(id * 2654435761u) >> 20) * 1e-3f; }
	static float bound(int id, int depth) { return value(id) - 8.f / (float)(1 + depth); }
	void begin(int r, int w)
	{
		rank = r; world = w; best = 1e9f; pops = 0;
		while (!q.empty()) q.pop();
		for (int k = 0; k < 64; k++) if (k % w == r) q.push(Cube{bound(1000 + k, 2), 1000 + k, 2});
	}
	void step(int max_pops, goicp_step_status* s)
	{
		widths.push_back(max_pops);
		int n = 0;
		while (!q.empty() && n < max_pops) {
			Cube c = q.top(); q.pop(); n++; pops++;
			if (best - c.lb <= 0.01f) { while (!q.empty()) q.pop(); break; }
			if (value(c.id) < best) { best = value(c.id); t[0] = (float)c.id; }
			if (c.depth < 9)
				for (int k = 0; k < 2; k++) {
					const int id = c.id * 2 + k;
					if (bound(id, c.depth + 1) < best) q.push(Cube{bound(id, c.depth + 1), id, c.depth + 1});
				}
		}
		s->finished = q.empty(); s->early_exit = 0; s->best_sse = best;
		s->frontier_lb = q.empty() ? INFINITY : q.top().lb; s->rot_pops = pops;
	}
};
int te_begin(void* c, int32_t r, int32_t w)
{
	ToyEngine* e = static_cast<ToyEngine*>(c);
	e->begin(r, w);
	return e->fail_what == 4 ? GOICP_ERR_DEVICE : 0;
}
int te_step(void* c, int32_t m, goicp_step_status* s)
{
	ToyEngine* e = static_cast<ToyEngine*>(c);
	if (e->fail_what == 1 && ++e->steps == e->fail_at_step) return GOICP_ERR_DEVICE;
	e->step(m, s);
	return 0;
}
int te_pose(void* c, float* sse, float R[9], float t[3])
{
	ToyEngine* e = static_cast<ToyEngine*>(c);
	*sse = e->best; std::memcpy(R, e->R, sizeof(e->R)); std::memcpy(t, e->t, sizeof(e->t));
	return 0;
}
int te_offer(void* c, float sse, const float*, const float t[3])
{
	ToyEngine* e = static_cast<ToyEngine*>(c);
	if (e->fail_what == 2) return GOICP_ERR_INTERNAL;
	if (sse < e->best) { e->best = sse; std::memcpy(e->t, t, sizeof(e->t)); }
	return 0;
}
int te_qsize(void* c, int32_t* n) { *n = (int32_t)static_cast<ToyEngine*>(c)->q.size(); return 0; }
int te_donate(void* c, int32_t max_nodes, float* nodes7, int32_t* n)
{
	ToyEngine* e = static_cast<ToyEngine*>(c);
	if (e->fail_what == 3) return GOICP_ERR_DEVICE;
	std::vector<ToyEngine::Cube> all;
	while (!e->q.empty()) { all.push_back(e->q.top()); e->q.pop(); }
	*n = 0;
	for (size_t i = 0; i < all.size(); i++) {
		if ((i & 1) && *n < max_nodes) { float* o = nodes7 + 7 * (*n)++; o[0] = (float)all[i].id; o[5] = all[i].lb; o[6] = (float)all[i].depth; o[1] = o[2] = o[3] = o[4] = 0.f; }
		else e->q.push(all[i]);
	}
	return 0;
}
int te_receive(void* c, const float* nodes7, int32_t n)
{
	ToyEngine* e = static_cast<ToyEngine*>(c);
	for (int i = 0; i < n; i++) e->q.push(ToyEngine::Cube{nodes7[7 * i + 5], (int)nodes7[7 * i], (int)nodes7[7 * i + 6]});
	return 0;
}
int te_end(void* c) { static_cast<ToyEngine*>(c)->ended = true; return 0; }

goicp_shard_engine_ops toy_ops(ToyEngine* e)
{
	goicp_shard_engine_ops eo{};
	eo.ctx = e; eo.sse_threshold = 0.01f;
	eo.begin = te_begin; eo.step = te_step; eo.pose = te_pose; eo.offer = te_offer; eo.queue_size = te_qsize;
	eo.donate = te_donate; eo.receive = te_receive; eo.end = te_end;
	return eo;
}

float run_world(int world, int rebalance, long long* donations, int stale = 0, int ramp_to = 0)
{
	std::vector<ToyEngine> eng((size_t)world);
	std::vector<goicp_comm_ops> comm((size_t)world);
	CHECK(goicp::thread_comm_create(world, comm.data()) == GOICP_OK);
	std::vector<goicp_shard_stats> st((size_t)world);
	std::vector<std::thread> th;
	for (int r = 0; r < world; r++)
		th.emplace_back([&, r] {
			goicp_shard_engine_ops eo = toy_ops(&eng[(size_t)r]);
			goicp_shard_options o{3, rebalance, stale, ramp_to};
			CHECK(goicp::run_sharded(&eo, &comm[(size_t)r], &o, &st[(size_t)r]) == GOICP_OK);
		});
	for (auto& t : th) t.join();
	for (int r = 0; r < world; r++) {
		// the step rule: a fixed width, or doubling from the first step's width up to ramp_to -- the same sequence on every rank
		const std::vector<int>& w = eng[(size_t)r].widths;
		CHECK(!w.empty() && w == eng[0].widths);
		for (size_t i = 0; i < w.size(); i++) CHECK(w[i] == (ramp_to > 3 ? std::min(ramp_to, 3 << std::min<size_t>(i, 20)) : 3));
		CHECK(eng[(size_t)r].best == eng[0].best);
		CHECK(st[(size_t)r].exchanges == st[0].exchanges && st[(size_t)r].broadcasts == st[0].broadcasts);
		CHECK(st[(size_t)r].failed_rank == -1 && st[(size_t)r].wait_ms >= 0.0);
		goicp::thread_comm_destroy(&comm[(size_t)r]);
	}
	if (donations) *donations = st[0].donations;
	return eng[0].best;
}

// one rank's callback fails: EVERY rank must come back with an error (its own status on the failing rank, GOICP_ERR_PEER
// with failed_rank set on the others), promptly -- long before the 20 s deadline -- and with its registration ended
void run_failure(int world, int bad_rank, int what, int at_step, int stale)
{
	std::vector<ToyEngine> eng((size_t)world);
	eng[(size_t)bad_rank].fail_what = what; eng[(size_t)bad_rank].fail_at_step = at_step;
	std::vector<goicp_comm_ops> comm((size_t)world);
	CHECK(goicp::thread_comm_create(world, comm.data()) == GOICP_OK);
	for (int r = 0; r < world; r++) CHECK(goicp::comm_set_timeout_ms(&comm[(size_t)r], 20000) == GOICP_OK);
	std::vector<goicp_shard_stats> st((size_t)world);
	std::vector<int> rc((size_t)world, 12345);
	const auto t0 = std::chrono::steady_clock::now();
	std::vector<std::thread> th;
	for (int r = 0; r < world; r++)
		th.emplace_back([&, r] {
			goicp_shard_engine_ops eo = toy_ops(&eng[(size_t)r]);
			goicp_shard_options o{3, 1, stale, 0};
			rc[(size_t)r] = goicp::run_sharded(&eo, &comm[(size_t)r], &o, &st[(size_t)r]);
		});
	for (auto& t : th) t.join();
	const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	CHECK(secs < 10.0);
	for (int r = 0; r < world; r++) {
		if (r == bad_rank) CHECK(rc[(size_t)r] == (what == 2 ? GOICP_ERR_INTERNAL : GOICP_ERR_DEVICE));
		else CHECK(rc[(size_t)r] == GOICP_ERR_PEER);
		CHECK(st[(size_t)r].failed_rank == bad_rank);
		CHECK(eng[(size_t)r].ended == (what != 4 || r != bad_rank));       // end() follows every successful begin()
		goicp::thread_comm_destroy(&comm[(size_t)r]);
	}
}

// a rank that never shows up: the others give up at the deadline with GOICP_ERR_TIMEOUT instead of hanging
void run_missing_rank()
{
	const int world = 3;
	std::vector<ToyEngine> eng((size_t)world);
	std::vector<goicp_comm_ops> comm((size_t)world);
	CHECK(goicp::thread_comm_create(world, comm.data()) == GOICP_OK);
	for (int r = 0; r < world; r++) CHECK(goicp::comm_set_timeout_ms(&comm[(size_t)r], 300) == GOICP_OK);
	std::vector<int> rc((size_t)world, 0);
	const auto t0 = std::chrono::steady_clock::now();
	std::vector<std::thread> th;
	for (int r = 0; r < world - 1; r++)                                       // rank 2 is "dead"
		th.emplace_back([&, r] {
			goicp_shard_engine_ops eo = toy_ops(&eng[(size_t)r]);
			goicp_shard_options o{3, 1, r == 0 ? 0 : 0, 0};
			rc[(size_t)r] = goicp::run_sharded(&eo, &comm[(size_t)r], &o, nullptr);
		});
	for (auto& t : th) t.join();
	const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	CHECK(rc[0] == GOICP_ERR_TIMEOUT && rc[1] == GOICP_ERR_TIMEOUT);
	CHECK(secs >= 0.25 && secs < 5.0);
	CHECK(eng[0].ended && eng[1].ended);
	for (int r = 0; r < world; r++) goicp::thread_comm_destroy(&comm[(size_t)r]);
}

}  // namespace

int main()
{
	// ---- sharding protocol + thread communicator, 1 / 2 / 4 / 7 ranks ----
	long long don = 0;
	const float single = run_world(1, 1, nullptr);
	for (int w : {2, 4, 7}) {
		const float b = run_world(w, 1, &don);
		CHECK(std::fabs(b - single) <= 0.01f);
	}
	CHECK(std::fabs(run_world(4, 0, &don) - single) <= 0.01f && don == 0);
	// one-step-stale exchange (helper thread): same optimum, collectives still matched on every rank
	for (int w : {2, 4, 7}) CHECK(std::fabs(run_world(w, 1, &don, 1) - single) <= 0.01f);
	CHECK(std::fabs(run_world(3, 0, &don, 1) - single) <= 0.01f && don == 0);
	// the step ramp (goicp_shard_options.ramp_to): widths 3, 6, 12, 24, 24 ... on every rank, same optimum; with the stale exchange too
	for (int w : {1, 2, 4, 7}) CHECK(std::fabs(run_world(w, 1, &don, 0, 24) - single) <= 0.01f);
	CHECK(std::fabs(run_world(4, 1, &don, 1, 24) - single) <= 0.01f);
	{   // a caller's own communicator is not one of the library's: its ctx is never dereferenced
		goicp_comm_ops own{};
		own.ctx = reinterpret_cast<void*>(0x1); own.world = 1;
		own.allreduce_min_u64 = [](void*, uint64_t*, size_t) { return 0; };
		CHECK(goicp::comm_set_timeout_ms(&own, 100) == GOICP_ERR_INVALID);
	}
	// failure is a collective decision; a lost rank is a timeout
	for (int stale : {0, 1}) {
		run_failure(2, 1, 1, 3, stale);      // step fails on rank 1 at its 3rd step
		run_failure(4, 2, 1, 1, stale);      // ... on the very first step
		run_failure(4, 0, 2, 0, stale);      // rank 0 cannot take the offered global best
		run_failure(4, 3, 4, 0, stale);      // begin() fails
	}
	run_failure(4, 0, 3, 0, 0);              // whoever is asked to donate first cannot (rank 0 owns the largest queue at a tie)
	run_missing_rank();
	// ---- threaded k-d build ----
	{
		std::mt19937 rng(7);
		std::uniform_real_distribution<float> u(-1.f, 1.f);
		std::vector<float> pts(3 * 20000);
		for (float& x : pts) x = u(rng);
		goicp::KdHost kh;
		goicp::build_kdtree(pts.data(), 20000, goicp::kLeafSlots, &kh);
		CHECK(kh.K >= 1 && !kh.pts.empty());
		size_t real = 0;
		for (const float4& p : kh.pts) if (std::isfinite(p.x)) real++;
		CHECK(real == 20000);
		std::atomic<int> hits{0};
		goicp::parallel_tasks(8, 100, [&](int) { hits++; });
		CHECK(hits == 100);
		float R[9];
		goicp::rodrigues(0.3f, -0.2f, 0.9f, R);
		CHECK(std::fabs(R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]) - 1.f) < 1e-5f);
	}
	// ---- config / cloud IO ----
	{
		const char* toml = "/tmp/goicp_selftest.toml";
		const char* txt = "/tmp/goicp_selftest.txt";
		{ std::ofstream f(txt); f << "3\n0 0 0\n1 2 3\n-1 0.5 2\n"; }
		{ std::ofstream f(toml); f << "[info]\ndescription = \"x\"\n[io]\ntarget = \"" << txt << "\"\nsource = \"" << txt << "\"\n[params]\nmode = 4\nmse_threshold = 1e-3\n[params.rotation]\nxmin = -90\nxmax = 90\n"; }
		goicp_config c;
		goicp::load_config(toml, &c);
		CHECK(c.mode == 4 && c.has_rotation_range == 1 && c.has_translation_range == 0 && c.rot_min[0] == -90.f);
		std::vector<float> cloud;
		goicp::load_cloud(txt, 1.0f, 2.0f, 1, cloud);
		CHECK(cloud.size() == 9 && cloud[3] == 2.f);
		bool threw = false;
		try { goicp::load_cloud("/tmp/does_not_exist.ply", 1.f, 1.f, 1, cloud); } catch (const goicp::IoError&) { threw = true; }
		CHECK(threw);
		goicp::write_viz_ply("/tmp/goicp_selftest.ply", cloud.data(), 3, cloud.data(), 3);
		std::vector<float> back;
		goicp::load_cloud("/tmp/goicp_selftest.ply", 1.f, 1.f, 1, back);
		CHECK(back.size() == 18);
	}
	std::printf("host selftest: %s\n", g_fail ? "FAILED" : "ok");
	return g_fail ? 1 : 0;
}
