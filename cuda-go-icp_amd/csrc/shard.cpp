// Multi-GPU Go-ICP: the rotation-cube search sharded over ranks -- the exchange / termination / rebalancing
// protocol.  Pure host code (no HIP, no RCCL in this file): it talks to an engine and to a communicator through
// the two callback tables of include/goicp_mi355.h, so the same code runs over RCCL on xGMI (rccl_comm.cpp), over
// any communicator a caller supplies (the tests use torch.distributed's gloo from Python and an in-process thread
// communicator), and under ThreadSanitizer on the CPU.
//
// New relative to the reference, which drives one CUDA device (src/window.cpp:110); SURVEY.md 8(e):
//   * every rank holds a full replica of the clouds, the distance transform and the k-d tree and owns every
//     world-th cube of the 64 level-2 rotation cubes; it runs its own best-first outer BnB (engine->step);
//   * after every step ONE all-reduce(MIN) of six packed 64-bit words carries everything the ranks must agree on:
//       w0 = orderable(best SSE) << 32 | rank        -> the global best-so-far error and its owner
//       w1 = orderable(frontier lb) << 32            -> min lower bound over every rank's queue (+inf when empty)
//       w2 = 0 if this rank hit the early exit (best < SSEThresh, jly_goicp.cpp:527), else 1
//       w3 = 0 if this rank still has work, else 1
//       w4 = 0 if this rank is idle (nothing queued / converged), else 1
//       w5 = all ones while this rank is healthy, else (status + 2^31) << 32 | rank -> FAILURE IS A COLLECTIVE DECISION:
//            a rank whose engine callback failed keeps taking part in the exchange, and every rank leaves the loop
//            in the same iteration, ends its registration and returns an error (its own, or GOICP_ERR_PEER)
//   * the owner broadcasts R|t (12 floats) only when the global best changed since the previous exchange -- every
//     rank sees the same sequence of global bests, so they agree on that without another collective;
//   * stop: any early exit, no rank active, or global best - min frontier lb <= SSEThresh (jly_goicp.cpp:416 on the
//     union of the queues);
//   * rebalancing: when some rank is idle while others work, the queue sizes are all-gathered (as one all-reduce of
//     a world-sized vector + an error word) and each idle rank receives, by broadcast from the currently largest queue,
//     every second cube of that queue in priority order (at most kDonateMax) -- both sides keep cubes of every priority;
//     a donor whose donate() failed broadcasts n = -1;
//   * stale_exchange (opt-in): the exchange of step k runs on a helper thread WHILE step k+1 is evaluated and is consumed
//     after it, so a rank only waits for ranks that are more than one step behind.  Valid because a best error that
//     arrives one step late only delays pruning (the search stays a BnB with valid bounds), and every rank consumes
//     the same sequence of exchange results at the same loop index, so stop decisions still agree; a final blocking
//     exchange after the stop brings every rank to the global best.
//   * a collective that fails (the communicator's deadline passed: GOICP_ERR_TIMEOUT) ends the run on this rank at once --
//     the other ranks meet their own deadlines; the caller must then exit non-zero, never re-execute.
#include "comm.hpp"
#include "trace.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace goicp {

namespace {

constexpr int kDonateMax = 64;               // cubes per donation
constexpr int kNodeWords = 7;                // x y z w ub lb level
constexpr uint64_t kHealthy = ~(uint64_t)0;

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
uint64_t pack_failure(int rc, int rank) { return ((uint64_t)((uint32_t)rc + 0x80000000u) << 32) | (uint32_t)rank; }   // smaller status first, then lower rank

double now_ms()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// what a rank contributes to an exchange: taken right after its step
struct Snapshot {
	int local_rc = GOICP_OK;         // first failure of an engine callback on this rank (sticky)
	float sse = std::numeric_limits<float>::infinity(), R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
	bool finished = true, early_exit = false;
	float frontier_lb = std::numeric_limits<float>::infinity();
};
// what every rank reads out of it (identical on all ranks, except comm_rc)
struct Exchanged {
	int comm_rc = GOICP_OK;          // the communicator itself failed (deadline, transport)
	int fail_rc = GOICP_OK, fail_rank = -1;      // some rank reported a failed callback
	float gbest = 0.f, glb = 0.f;
	int owner = 0;
	bool any_early = false, any_active = false, any_idle = false, pose_moved = false;
	float pose[12];
	int collectives = 0, broadcasts = 0;
};

// One exchange: the packed all-reduce and, when the global best moved, the owner's pose.  `last_gbest` is the exchange
// side's own state (the sequence of global bests is the same on every rank).
Exchanged exchange(const goicp_comm_ops* comm, const Snapshot& s, float* last_gbest)
{
	Exchanged x;
	const float inf = std::numeric_limits<float>::infinity();
	uint64_t w[6];
	w[0] = ((uint64_t)orderable(s.sse) << 32) | (uint32_t)comm->rank;
	w[1] = (uint64_t)orderable(s.finished ? inf : s.frontier_lb) << 32;
	w[2] = s.early_exit ? 0u : 1u;
	w[3] = s.finished ? 1u : 0u;
	w[4] = s.finished ? 0u : 1u;
	w[5] = s.local_rc == GOICP_OK ? kHealthy : pack_failure(s.local_rc, comm->rank);
	{
		TraceRange tr("goicp:exchange");
		x.comm_rc = comm->allreduce_min_u64(comm->ctx, w, 6);
	}
	if (x.comm_rc != GOICP_OK) return x;
	x.collectives = 1;
	if (w[5] != kHealthy) {
		x.fail_rc = (int)((uint32_t)(w[5] >> 32) - 0x80000000u);
		x.fail_rank = (int)(uint32_t)(w[5] & 0xffffffffu);
		return x;                    // every rank sees this word: nobody enters the broadcast
	}
	x.gbest = from_orderable((uint32_t)(w[0] >> 32));
	x.owner = (int)(uint32_t)(w[0] & 0xffffffffu);
	x.glb = from_orderable((uint32_t)(w[1] >> 32));
	x.any_early = w[2] == 0; x.any_active = w[3] == 0; x.any_idle = w[4] == 0;
	if (x.gbest < *last_gbest) {
		std::memcpy(x.pose, s.R, sizeof(s.R));
		std::memcpy(x.pose + 9, s.t, sizeof(s.t));
		x.comm_rc = comm->bcast(comm->ctx, x.pose, sizeof(x.pose), x.owner);
		if (x.comm_rc != GOICP_OK) return x;
		x.broadcasts = 1;
		x.pose_moved = true;
		*last_gbest = x.gbest;
	}
	return x;
}

// the helper thread of the stale mode: runs one posted exchange at a time
class ExchangeWorker {
public:
	ExchangeWorker(const goicp_comm_ops* comm, float* last_gbest) : comm_(comm), last_gbest_(last_gbest), th_([this] { loop(); }) {}
	~ExchangeWorker()
	{
		{ std::lock_guard<std::mutex> lk(m_); quit_ = true; }
		cv_.notify_all();
		th_.join();
	}
	void post(const Snapshot& s)
	{
		{ std::lock_guard<std::mutex> lk(m_); job_ = s; has_job_ = true; done_ = false; }
		cv_.notify_all();
	}
	Exchanged wait()
	{
		std::unique_lock<std::mutex> lk(m_);
		cv_.wait(lk, [&] { return done_; });
		return res_;
	}
private:
	void loop()
	{
		std::unique_lock<std::mutex> lk(m_);
		while (true) {
			cv_.wait(lk, [&] { return has_job_ || quit_; });
			if (quit_) return;
			const Snapshot s = job_;
			has_job_ = false;
			lk.unlock();
			const Exchanged x = exchange(comm_, s, last_gbest_);      // the communicator's own deadline bounds this
			lk.lock();
			res_ = x; done_ = true;
			cv_.notify_all();
		}
	}
	const goicp_comm_ops* comm_;
	float* last_gbest_;
	std::mutex m_;
	std::condition_variable cv_;
	Snapshot job_;
	Exchanged res_;
	bool has_job_ = false, done_ = false, quit_ = false;
	std::thread th_;
};

}  // namespace

int run_sharded(const goicp_shard_engine_ops* eng, const goicp_comm_ops* comm, const goicp_shard_options* opt, goicp_shard_stats* stats)
{
	if (!eng || !comm || !opt || comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world || opt->rot_pops_per_step < 1) return GOICP_ERR_INVALID;
	const int rank = comm->rank, world = comm->world;
	// step width: fixed, or the single-GPU driver's ramp (engine.cpp register_step: 8, 16, 32, 64 parents per batch) -- a rank
	// then runs ONE batch per step and exchanges once per batch.  Every rank computes the same sequence.
	const int pops0 = opt->rot_pops_per_step, ramp_to = opt->ramp_to > pops0 ? opt->ramp_to : 0;
	int pops = pops0;
	const bool rebalance = opt->rebalance != 0, stale = opt->stale_exchange != 0 && world > 1;
	goicp_shard_stats st{};
	st.failed_rank = -1;
	const float thr = eng->sse_threshold;
	const float inf = std::numeric_limits<float>::infinity();
	float last_gbest = inf;
	std::vector<uint64_t> sizes((size_t)world + 1);
	std::vector<float> xfer((size_t)1 + kDonateMax * kNodeWords);

	// a failed callback does not leave the protocol: the failure is remembered, the engine is left alone from then on,
	// and the next exchange tells everybody
	int local_rc = eng->begin(eng->ctx, rank, world);
	const bool begun = local_rc == GOICP_OK;
	int final_rc = GOICP_OK;
	Snapshot snap;

	auto do_step = [&] {
		Snapshot s;
		s.local_rc = local_rc;
		if (local_rc != GOICP_OK) return s;
		goicp_step_status ss{};
		const double t0 = now_ms();
		int rc = eng->step(eng->ctx, pops, &ss);
		st.step_ms += now_ms() - t0;
		if (ramp_to) pops = std::min(ramp_to, pops * 2);
		if (rc == GOICP_OK) rc = eng->pose(eng->ctx, &s.sse, s.R, s.t);
		if (rc != GOICP_OK) { local_rc = s.local_rc = rc; s.sse = inf; return s; }
		st.steps++;
		if (ss.finished) st.steps_idle++;
		s.finished = ss.finished != 0; s.early_exit = ss.early_exit != 0; s.frontier_lb = ss.frontier_lb;
		return s;
	};
	// fold an exchange result into this rank; returns true when the loop must end (final_rc says why)
	auto consume = [&](const Exchanged& x, bool* stop) {
		st.exchanges += x.collectives;
		st.broadcasts += x.broadcasts;
		if (x.comm_rc != GOICP_OK) { final_rc = x.comm_rc; return true; }
		if (x.fail_rc != GOICP_OK) {
			st.failed_rank = x.fail_rank;
			final_rc = local_rc != GOICP_OK ? local_rc : GOICP_ERR_PEER;
			return true;
		}
		if (x.pose_moved && local_rc == GOICP_OK) {
			float cur = inf, R[9], t[3];
			int rc = eng->pose(eng->ctx, &cur, R, t);
			if (rc == GOICP_OK && cur > x.gbest) rc = eng->offer(eng->ctx, x.gbest, x.pose, x.pose + 9);
			if (rc != GOICP_OK) local_rc = rc;                // told to the others by the next exchange
		}
		*stop = x.any_early || !x.any_active || x.gbest - x.glb <= thr;
		return *stop;
	};
	// idle ranks take half of the largest queues; collective, every rank runs the same plan.  false = the run is over.
	auto do_rebalance = [&]() -> bool {
		int32_t mine = 0;
		if (local_rc == GOICP_OK) {
			const int rc = eng->queue_size(eng->ctx, &mine);
			if (rc != GOICP_OK) local_rc = rc;
		}
		// all-gather as an all-reduce(MIN): slot r carries the size of rank r, every other slot the MIN-neutral value
		for (int r = 0; r < world; r++) sizes[(size_t)r] = ~(uint64_t)0;
		sizes[(size_t)rank] = (uint64_t)std::max(mine, 0);
		sizes[(size_t)world] = local_rc == GOICP_OK ? kHealthy : pack_failure(local_rc, rank);
		const double t0 = now_ms();
		int rc = comm->allreduce_min_u64(comm->ctx, sizes.data(), (size_t)world + 1);
		st.wait_ms += now_ms() - t0;
		if (rc != GOICP_OK) { final_rc = rc; return false; }
		st.exchanges++;
		if (sizes[(size_t)world] != kHealthy) {
			st.failed_rank = (int)(uint32_t)(sizes[(size_t)world] & 0xffffffffu);
			final_rc = local_rc != GOICP_OK ? local_rc : GOICP_ERR_PEER;
			return false;
		}
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
				if (rc != GOICP_OK) { local_rc = rc; n = -1; }            // still broadcast: the others are waiting in it
				xfer[0] = (float)n;
			}
			const double t1 = now_ms();
			rc = comm->bcast(comm->ctx, xfer.data(), xfer.size() * sizeof(float), donor);
			st.wait_ms += now_ms() - t1;
			if (rc != GOICP_OK) { final_rc = rc; return false; }
			n = (int32_t)xfer[0];
			if (n < 0) {
				st.failed_rank = donor;
				final_rc = local_rc != GOICP_OK ? local_rc : GOICP_ERR_PEER;
				return false;
			}
			if (rank == recv && n > 0) {
				rc = eng->receive(eng->ctx, xfer.data() + 1, n);
				if (rc != GOICP_OK) local_rc = rc;                       // reported by the next exchange
			}
			st.donations++;
			st.donated_cubes += n;
			q[(size_t)donor] -= n;
			q[(size_t)recv] += n;
		}
		return true;
	};

	if (!stale) {
		while (true) {
			snap = do_step();
			const double t0 = now_ms();
			const Exchanged x = exchange(comm, snap, &last_gbest);
			st.wait_ms += now_ms() - t0;
			bool stop = false;
			if (consume(x, &stop)) break;
			if (rebalance && x.any_idle && world > 1 && !do_rebalance()) break;
		}
	} else {
		ExchangeWorker worker(comm, &last_gbest);
		bool posted = false;
		while (true) {
			snap = do_step();                                   // step k+1 runs while the exchange of step k is in flight
			bool stop = false, over = false;
			if (posted) {
				const double t0 = now_ms();
				const Exchanged x = worker.wait();
				st.wait_ms += now_ms() - t0;
				posted = false;
				over = consume(x, &stop);
				if (!over && rebalance && x.any_idle && !do_rebalance()) over = true;      // no exchange in flight here
			}
			if (over) {
				if (stop && final_rc == GOICP_OK) {
					// every rank stops at this same index: one blocking exchange of the states after the last step
					snap.local_rc = local_rc;
					if (local_rc == GOICP_OK && eng->pose(eng->ctx, &snap.sse, snap.R, snap.t) != GOICP_OK) snap.local_rc = local_rc = GOICP_ERR_INTERNAL;
					const double t0 = now_ms();
					const Exchanged y = exchange(comm, snap, &last_gbest);
					st.wait_ms += now_ms() - t0;
					bool dummy = false;
					consume(y, &dummy);
				}
				break;
			}
			snap.local_rc = local_rc;                           // a failed offer / receive since the step travels with it
			worker.post(snap);
			posted = true;
		}
	}
	if (begun) {
		const int rc = eng->end(eng->ctx);
		if (rc != GOICP_OK && final_rc == GOICP_OK) final_rc = rc;
	}
	if (final_rc == GOICP_OK && local_rc != GOICP_OK) final_rc = local_rc;
	float sse = inf, R[9], t[3];
	if (begun && eng->pose(eng->ctx, &sse, R, t) == GOICP_OK) st.best_sse = sse;
	if (stats) *stats = st;
	return final_rc;
}

// ---- in-process communicator: `world` host threads of one process (tests, single-GPU rehearsals of the N-rank path) ----
namespace {

int env_timeout_ms()
{
	const char* e = std::getenv("GOICP_COMM_TIMEOUT_MS");
	const int v = e ? std::atoi(e) : 0;
	return v > 0 ? v : 60000;
}

struct ThreadGroup {
	int world = 0, refs = 0;
	std::mutex m;
	std::condition_variable cv;
	int arrived = 0, left = 0;
	uint64_t generation = 0;
	bool broken = false;             // a rank gave up waiting: every later (and every waiting) collective fails
	std::vector<uint64_t> acc;       // MIN accumulator of the running all-reduce
	std::vector<unsigned char> blob; // broadcast payload
};
struct ThreadComm {
	CommHeader hdr;                  // first member: goicp_comm_set_timeout_ms finds it through ctx
	ThreadGroup* g;
	int rank;
};

// two-phase rendezvous: everyone contributes, the last arriver publishes, everyone copies out, the last leaver resets.
// Every wait has the communicator's deadline; whoever misses it breaks the group for all.
template <class Contribute, class Collect>
int rendezvous(ThreadComm* c, Contribute contribute, Collect collect)
{
	ThreadGroup* g = c->g;
	// system_clock: libstdc++ then waits with pthread_cond_timedwait, which ThreadSanitizer intercepts (the steady-clock
	// form, pthread_cond_clockwait, it does not -- every wait would be reported as a double lock)
	const auto deadline = std::chrono::system_clock::now() + std::chrono::milliseconds(c->hdr.timeout_ms);
	std::unique_lock<std::mutex> lk(g->m);
	auto give_up = [&] { g->broken = true; g->cv.notify_all(); return GOICP_ERR_TIMEOUT; };
	if (!g->cv.wait_until(lk, deadline, [&] { return g->left == 0 || g->broken; })) return give_up();      // the previous collective has been read by all
	if (g->broken) return GOICP_ERR_TIMEOUT;
	contribute();
	const uint64_t gen = g->generation;
	if (++g->arrived == g->world) { g->generation++; g->left = g->world; g->cv.notify_all(); }
	else if (!g->cv.wait_until(lk, deadline, [&] { return g->generation != gen || g->broken; })) return give_up();
	if (g->generation == gen) return GOICP_ERR_TIMEOUT;         // woken by a broken group
	collect();
	if (--g->left == 0) { g->arrived = 0; g->acc.clear(); g->cv.notify_all(); }
	return GOICP_OK;
}

int thread_allreduce(void* ctx, uint64_t* words, size_t n)
{
	ThreadComm* c = static_cast<ThreadComm*>(ctx);
	return rendezvous(c,
	                  [&] {
		                  if (c->g->acc.empty()) c->g->acc.assign(words, words + n);
		                  else for (size_t i = 0; i < n; i++) c->g->acc[i] = std::min(c->g->acc[i], words[i]);
	                  },
	                  [&] { std::memcpy(words, c->g->acc.data(), n * sizeof(uint64_t)); });
}
int thread_bcast(void* ctx, void* buf, size_t bytes, int32_t root)
{
	ThreadComm* c = static_cast<ThreadComm*>(ctx);
	return rendezvous(c,
	                  [&] { if (c->rank == root) c->g->blob.assign(static_cast<unsigned char*>(buf), static_cast<unsigned char*>(buf) + bytes); },
	                  [&] { if (c->rank != root) std::memcpy(buf, c->g->blob.data(), bytes); });
}

}  // namespace

int comm_default_timeout_ms() { return env_timeout_ms(); }

namespace {
std::mutex g_kinds_m;
std::vector<CommAllreduceFn> g_kinds;
}  // namespace

void comm_register_library_kind(CommAllreduceFn fn)
{
	std::lock_guard<std::mutex> lk(g_kinds_m);
	if (std::find(g_kinds.begin(), g_kinds.end(), fn) == g_kinds.end()) g_kinds.push_back(fn);
}
bool comm_is_library_kind(const goicp_comm_ops* comm)
{
	if (!comm || !comm->allreduce_min_u64) return false;
	std::lock_guard<std::mutex> lk(g_kinds_m);
	return std::find(g_kinds.begin(), g_kinds.end(), (CommAllreduceFn)comm->allreduce_min_u64) != g_kinds.end();
}

int comm_set_timeout_ms(goicp_comm_ops* comm, int ms)
{
	if (!comm || !comm->ctx || ms < 1) return GOICP_ERR_INVALID;
	if (!comm_is_library_kind(comm)) return GOICP_ERR_INVALID;  // a caller's own communicator: its ctx is not ours to read
	CommHeader* h = static_cast<CommHeader*>(comm->ctx);
	if (h->magic != kCommMagic) return GOICP_ERR_INVALID;
	h->timeout_ms = ms;
	return GOICP_OK;
}

int thread_comm_create(int world, goicp_comm_ops* out)
{
	ThreadGroup* g = new (std::nothrow) ThreadGroup;
	if (!g) return GOICP_ERR_INTERNAL;
	g->world = world; g->refs = world;
	comm_register_library_kind(&thread_allreduce);
	for (int r = 0; r < world; r++) {
		out[r].ctx = new ThreadComm{CommHeader{kCommMagic, env_timeout_ms()}, g, r};
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
