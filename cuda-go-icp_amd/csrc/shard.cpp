// Multi-GPU Go-ICP: the rotation-cube search sharded over ranks -- the exchange / termination / rebalancing
// protocol.  Pure host code (no HIP, no RCCL in this file): it talks to an engine and to a communicator through
// the two callback tables of include/goicp_mi355.h, so the same code runs over RCCL on xGMI (rccl_comm.cpp), over
// any communicator a caller supplies (the tests use torch.distributed's gloo from Python and an in-process thread
// communicator), and under ThreadSanitizer on the CPU.
//
// New relative to the reference, which drives one CUDA device (src/window.cpp:110); SURVEY.md 8(e):
//   * every rank holds a full replica of the clouds, the distance transform and the k-d tree and owns every
//     world-th cube of the 64 level-2 rotation cubes; it runs its own best-first outer BnB (engine->step);
//   * after every step ONE all-reduce(MIN) of five packed 64-bit words carries everything the ranks must agree on:
//       w0 = orderable(best SSE) << 32 | rank        -> the global best-so-far error and its owner
//       w1 = orderable(frontier lb) << 32            -> min lower bound over every rank's queue (+inf when empty)
//       w2 = 0 if this rank hit the early exit (best < SSEThresh, jly_goicp.cpp:527), else 1
//       w3 = 0 if this rank still has work, else 1
//       w4 = 0 if this rank is idle (nothing queued / converged), else 1
//   * the owner broadcasts R|t (12 floats) only when the global best changed since the previous exchange -- every
//     rank sees the same sequence of global bests, so they agree on that without another collective;
//   * stop: any early exit, no rank active, or global best - min frontier lb <= SSEThresh (jly_goicp.cpp:416 on the
//     union of the queues);
//   * rebalancing: when some rank is idle while others work, the queue sizes are all-gathered (as one all-reduce of
//     a world-sized vector) and each idle rank receives, by broadcast from the currently largest queue, every second
//     cube of that queue in priority order (at most kDonateMax) -- both sides keep cubes of every priority.
#include "../../include/goicp_mi355.h"
#include "trace.hpp"

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <vector>

namespace goicp {

namespace {

constexpr int kDonateMax = 64;               // cubes per donation
constexpr int kNodeWords = 7;                // x y z w ub lb level

// float -> uint32 whose unsigned order is the float order (finite values and +-inf)
uint32_t orderable(float f)
{
	uint32_t u;
	std::memcpy(&u, &f, 4);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
float from_orderable(uint32_t o)
{
	const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
	float f;
	std::memcpy(&f, &u, 4);
	return f;
}

}  // namespace

int run_sharded(const goicp_shard_engine_ops* eng, const goicp_comm_ops* comm, int rot_pops_per_step, int rebalance, goicp_shard_stats* stats)
{
	if (!eng || !comm || comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world || rot_pops_per_step < 1) return GOICP_ERR_INVALID;
	const int rank = comm->rank, world = comm->world;
	goicp_shard_stats st{};
	int rc = eng->begin(eng->ctx, rank, world);
	if (rc != GOICP_OK) return rc;
	const float thr = eng->sse_threshold;
	const float inf = std::numeric_limits<float>::infinity();
	float last_gbest = inf;
	std::vector<uint64_t> sizes((size_t)world);
	std::vector<float> xfer((size_t)1 + kDonateMax * kNodeWords);
	while (true) {
		goicp_step_status ss{};
		rc = eng->step(eng->ctx, rot_pops_per_step, &ss);
		if (rc != GOICP_OK) return rc;
		st.steps++;
		float sse = 0.f, R[9], t[3];
		rc = eng->pose(eng->ctx, &sse, R, t);
		if (rc != GOICP_OK) return rc;
		// ---- one all-reduce(MIN) of five packed words ----
		uint64_t w[5];
		w[0] = ((uint64_t)orderable(sse) << 32) | (uint32_t)rank;
		w[1] = (uint64_t)orderable(ss.finished ? inf : ss.frontier_lb) << 32;
		w[2] = ss.early_exit ? 0u : 1u;
		w[3] = ss.finished ? 1u : 0u;
		w[4] = ss.finished ? 0u : 1u;
		{
			TraceRange tr("goicp:exchange");
			rc = comm->allreduce_min_u64(comm->ctx, w, 5);
		}
		if (rc != GOICP_OK) return rc;
		st.exchanges++;
		const float gbest = from_orderable((uint32_t)(w[0] >> 32));
		const int owner = (int)(uint32_t)(w[0] & 0xffffffffu);
		const float glb = from_orderable((uint32_t)(w[1] >> 32));
		const bool any_early = w[2] == 0, any_active = w[3] == 0, any_idle = w[4] == 0;
		// ---- the winner's pose, only when the global best moved ----
		if (gbest < last_gbest) {
			float pose[12];
			std::memcpy(pose, R, sizeof(R));
			std::memcpy(pose + 9, t, sizeof(t));
			rc = comm->bcast(comm->ctx, pose, sizeof(pose), owner);
			if (rc != GOICP_OK) return rc;
			st.broadcasts++;
			if (sse > gbest) {
				rc = eng->offer(eng->ctx, gbest, pose, pose + 9);
				if (rc != GOICP_OK) return rc;
			}
			last_gbest = gbest;
		}
		if (any_early || !any_active || gbest - glb <= thr) break;
		// ---- rebalancing: idle ranks take half of the largest queues ----
		if (rebalance && any_idle && world > 1) {
			int32_t mine = 0;
			rc = eng->queue_size(eng->ctx, &mine);
			if (rc != GOICP_OK) return rc;
			// all-gather as an all-reduce(MIN): slot r carries the size of rank r, every other slot the MIN-neutral value
			for (int r = 0; r < world; r++) sizes[(size_t)r] = ~(uint64_t)0;
			sizes[(size_t)rank] = (uint64_t)std::max(mine, 0);
			rc = comm->allreduce_min_u64(comm->ctx, sizes.data(), (size_t)world);
			if (rc != GOICP_OK) return rc;
			st.exchanges++;
			std::vector<int64_t> q((size_t)world);
			for (int r = 0; r < world; r++) q[(size_t)r] = (int64_t)sizes[(size_t)r];
			// the same plan on every rank: receivers in rank order, each served by the largest remaining queue
			for (int recv = 0; recv < world; recv++) {
				if (q[(size_t)recv] != 0) continue;
				int donor = -1;
				for (int r = 0; r < world; r++)
					if (q[(size_t)r] >= 2 && (donor < 0 || q[(size_t)r] > q[(size_t)donor])) donor = r;
				if (donor < 0) break;
				int32_t n = 0;
				if (rank == donor) {
					rc = eng->donate(eng->ctx, kDonateMax, xfer.data() + 1, &n);
					if (rc != GOICP_OK) return rc;
					xfer[0] = (float)n;
				}
				rc = comm->bcast(comm->ctx, xfer.data(), xfer.size() * sizeof(float), donor);
				if (rc != GOICP_OK) return rc;
				n = (int32_t)xfer[0];
				if (rank == recv && n > 0) {
					rc = eng->receive(eng->ctx, xfer.data() + 1, n);
					if (rc != GOICP_OK) return rc;
				}
				st.donations++;
				st.donated_cubes += n;
				q[(size_t)donor] -= n;
				q[(size_t)recv] += n;
			}
		}
	}
	rc = eng->end(eng->ctx);
	if (rc != GOICP_OK) return rc;
	float sse = 0.f, R[9], t[3];
	eng->pose(eng->ctx, &sse, R, t);
	st.best_sse = sse;
	if (stats) *stats = st;
	return GOICP_OK;
}

// ---- in-process communicator: `world` host threads of one process (tests, single-GPU rehearsals of the N-rank path) ----
namespace {

struct ThreadGroup {
	int world = 0, refs = 0;
	std::mutex m;
	std::condition_variable cv;
	int arrived = 0, left = 0;
	uint64_t generation = 0;
	std::vector<uint64_t> acc;       // MIN accumulator of the running all-reduce
	std::vector<unsigned char> blob; // broadcast payload
};
struct ThreadComm { ThreadGroup* g; int rank; };

// two-phase rendezvous: everyone contributes, the last arriver publishes, everyone copies out, the last leaver resets
template <class Contribute, class Collect>
void rendezvous(ThreadGroup* g, Contribute contribute, Collect collect)
{
	std::unique_lock<std::mutex> lk(g->m);
	g->cv.wait(lk, [&] { return g->left == 0; });        // the previous collective has been read by all
	contribute();
	const uint64_t gen = g->generation;
	if (++g->arrived == g->world) { g->generation++; g->left = g->world; g->cv.notify_all(); }
	else g->cv.wait(lk, [&] { return g->generation != gen; });
	collect();
	if (--g->left == 0) { g->arrived = 0; g->acc.clear(); g->cv.notify_all(); }
}

int thread_allreduce(void* ctx, uint64_t* words, size_t n)
{
	ThreadComm* c = static_cast<ThreadComm*>(ctx);
	rendezvous(c->g,
	           [&] {
		           if (c->g->acc.empty()) c->g->acc.assign(words, words + n);
		           else for (size_t i = 0; i < n; i++) c->g->acc[i] = std::min(c->g->acc[i], words[i]);
	           },
	           [&] { std::memcpy(words, c->g->acc.data(), n * sizeof(uint64_t)); });
	return GOICP_OK;
}
int thread_bcast(void* ctx, void* buf, size_t bytes, int32_t root)
{
	ThreadComm* c = static_cast<ThreadComm*>(ctx);
	rendezvous(c->g,
	           [&] { if (c->rank == root) c->g->blob.assign(static_cast<unsigned char*>(buf), static_cast<unsigned char*>(buf) + bytes); },
	           [&] { if (c->rank != root) std::memcpy(buf, c->g->blob.data(), bytes); });
	return GOICP_OK;
}

}  // namespace

int thread_comm_create(int world, goicp_comm_ops* out)
{
	ThreadGroup* g = new (std::nothrow) ThreadGroup;
	if (!g) return GOICP_ERR_INTERNAL;
	g->world = world; g->refs = world;
	for (int r = 0; r < world; r++) {
		out[r].ctx = new ThreadComm{g, r};
		out[r].rank = r; out[r].world = world;
		out[r].allreduce_min_u64 = &thread_allreduce;
		out[r].bcast = &thread_bcast;
	}
	return GOICP_OK;
}
void thread_comm_destroy(goicp_comm_ops* comm)
{
	if (!comm || !comm->ctx) return;
	ThreadComm* c = static_cast<ThreadComm*>(comm->ctx);
	bool last;
	{ std::lock_guard<std::mutex> lk(c->g->m); last = --c->g->refs == 0; }
	if (last) delete c->g;
	delete c;
	comm->ctx = nullptr;
}

}  // namespace goicp
