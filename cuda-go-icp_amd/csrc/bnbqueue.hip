// Device-resident translation queues of the inner branch-and-bound (GoICP::InnerBnB, src/goicp/jly_goicp.cpp:227-340).
//
// The reference pops ONE translation node per step from a host priority queue and evaluates its 8 children; the
// round-1 driver popped up to 32 nodes per search per round on the host and paid, per round, an upload of the
// expansion records, a download of the bounds, an event wait and the host-side heap work.  Here every active inner
// search (up to 1 024 of them run in lock-step: 64 rotation parents x 8 children x {upper-, lower-bound pass}) keeps
// its queue in HBM, and a round is two launches with no host involvement:
//
//   bnb_queue_kernel   one workgroup per search: DIGEST the bounds of the children evaluated in the previous round
//                      (update the search's incumbent, jly_goicp.cpp:319-324; push the children that survive, :327-335),
//                      then SELECT this round's expansions -- the up-to-K queued nodes with the smallest lower bounds
//                      (ties: wider cube first, jly_goicp.h:64-71), subject to the stop rule best - lb >= SSEThresh
//                      (:257) -- remove them from the queue and append their records to the round's expansion list.
//   bounds_queue_kernel (device.hip) evaluates the 8 children of every listed expansion; the list length is read from
//                      device memory, a fixed grid walks the work items.
//
// The host only queues rounds and, every few rounds, reads one word to learn whether any search is still active.
// Selection = exact K smallest by a 3-digit radix select on a 32-bit key (lb with its five lowest mantissa bits
// replaced by the node's depth: lb ascending to 2^-19 relative, then wider cube first); order of equal keys = queue
// position, so a run is bit-reproducible.  Pruning uses the incumbent after ALL children of the round are known
// (the reference updates it child by child): never prunes a node the reference would keep alive for a valid reason,
// only more of them -- any expansion order of a best-first BnB keeps the bounds valid.
#include <hip/hip_runtime.h>

#include "device.hpp"

namespace goicp {

namespace {

// 1 024 threads per search: one thread per child in the digest (8 x 128 expansions), 8 queue keys per thread in the selection
constexpr int kQThreads = 1024;
constexpr int kQPer = kQueueCap / kQThreads;      // keys per thread held in registers (8)
static_assert(kQueueCap % kQThreads == 0 && kQPer <= 64, "queue capacity / threads = keys per thread");
static_assert(kQueueMaxPop <= kQThreads && kQueueMaxPop % 64 == 0, "one thread per selected node; the digest takes 1 024 children per pass");

struct QShared {
	unsigned hist[2048];
	unsigned sel_prefix, sel_rem;
	unsigned wave_tot[kQThreads / 64];
	unsigned short cnt[kQPer][kQThreads / 64];     // per (j, wavefront) counts: ties, then selected
	int sel_pos[kQueueMaxPop];
	int hole_pos[kQueueMaxPop];
	float red_ub[kQThreads / 64];
	int red_idx[kQThreads / 64];
	unsigned push_tot[kQThreads / 64];
	int n_sel, n_holes, parent_off, bcast;
	int hole_cnt[kQueueMaxPop / 64], fill_cnt[kQueueMaxPop / 64];
	float ext[kQueueMaxPop / 64][6];               // tile-list test: min / max corner of the selected nodes, per wavefront
	float ext_w[kQueueMaxPop / 64];
	unsigned char tail_sel[kQueueMaxPop];          // removal: is tail position m + t one of the selected nodes?
	float psum[2 * kQThreads];                     // digest: partial sums of the chunk partials, [child][part] for ub, then for lb
};

__device__ __forceinline__ int node_depth(float root_w, float w)      // w = root_w * 2^-depth exactly
{
	return (int)((__float_as_uint(root_w) >> 23) & 0xffu) - (int)((__float_as_uint(w) >> 23) & 0xffu);
}
__device__ __forceinline__ unsigned node_key(float lb, int depth)
{
	return (__float_as_uint(lb) & ~31u) | (unsigned)min(max(depth, 0), 31);
}
__device__ __forceinline__ bool in_box(const QParams& qp, float x, float y, float z, float w)
{
	return cube_in_range(x, y, z, w, qp.lo, qp.hi);          // half-open cube against the closed range: the range's high face is inclusive (device.hpp)
}

}  // namespace

// One round of one search (the body of bnb_queue_kernel).  PER = queue slots held per thread: 8 covers the whole slab; 1 is the same code for
// a search whose queue cannot exceed 1 024 nodes this round (count + 8 x last round's expansions) -- every default registration's searches, most
// rounds: the per-slot loops (ballots, rank scans, key loads) then run once instead of eight times.  Measured on the full bunny (460 searches in
// lock-step, rocprofv3 per dispatch): 40-58 -> 35-50 us per large round, 2.27 -> 1.75 ms per registration; with the kernel held to 64 VGPRs
// (__launch_bounds__(1024, 8): four spilled registers on the rare purge path) two searches share a CU instead of one: bound-evaluation + queue
// time of the registration 17.5 -> 17.1 ms.  What is left of a round is its chain of ~15 barriers and ~6 dependent memory round trips.
template <int PER>
__device__ __forceinline__ void queue_round(QShared& sh, QSearch* __restrict__ S, QNode* __restrict__ Q, const QParams& qp,
                                            const ParentRec* __restrict__ prev_parents, ParentRec* __restrict__ parents,
                                            const float* __restrict__ ubs, const float* __restrict__ lbs, const float* __restrict__ scratch,
                                            QCtl* __restrict__ ctl, int parity, const QTile& tile, int* __restrict__ parent_search, int s)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

	float best = S->best;
	int count = S->count;
	int stale = S->stale;
	const int n_prev = S->n_parents;
	// ------------------------------------------------------------------------------------------------------------
	// digest: the bounds of the children evaluated in the previous round (8 per expansion, <= 256 children)
	// ------------------------------------------------------------------------------------------------------------
	if (n_prev > 0) {
		const int Ctot = 8 * n_prev, off = S->parent_off;
		// the previous round listed this search's expansions in the tile list: its bounds, partial sums and records live there
		if (S->tile) { prev_parents = tile.parents[parity ^ 1]; ubs = tile.ub; lbs = tile.lb; scratch = tile.scratch; }
		const int chunks = S->tile ? ctl->tile_chunks : ctl->chunks;
		bool improved_any = false;
		// one thread per child, 1 024 children per pass (a lone search may have expanded up to 512 nodes: four passes; a pass prunes with the
		// incumbent of the children seen so far -- later passes only tighten it, and a node kept too long fails the stop rule when it is selected)
#pragma unroll 1
		for (int cb = 0; cb < Ctot; cb += kQThreads) {
		const int C = min(kQThreads, Ctot - cb);
		const bool mine = tid < C;
		// The evaluation split the cloud into `chunks` chunks: its per-chunk sums are added here (this replaces a finalize launch
		// per round).  Small rounds have few children and many chunks (up to 118), so the sum is spread over the workgroup: P
		// threads per child add every P-th chunk partial, the child's own thread adds the P results in order.  Fixed order
		// for given (children, chunks): deterministic.
		const int P = chunks > 1 ? max(1, min(chunks, kQThreads / C)) : 1;
		if (chunks > 1) {
			if (tid < C * P) {
				const int c = tid / P, p = tid - c * P;
				const float* sp = scratch + ((size_t)(off + ((cb + c) >> 3)) * chunks) * (2 * kGroup) + (c & 7);
				float a = 0.f, b = 0.f;
				for (int j = p; j < chunks; j += P) { a += sp[(size_t)j * 2 * kGroup]; b += sp[(size_t)j * 2 * kGroup + kGroup]; }
				sh.psum[tid] = a; sh.psum[kQThreads + tid] = b;
			}
			__syncthreads();
		}
		float cx = 0.f, cy = 0.f, cz = 0.f, cw = 0.f, ub = INFINITY, lb = INFINITY;
		bool valid = false;
		if (mine) {
			const ParentRec pr = prev_parents[off + ((cb + tid) >> 3)];
			const int c = tid & 7;
			cw = pr.w / 2;                                            // jly_goicp.cpp:262-270
			cx = pr.x + (float)(c & 1) * cw; cy = pr.y + (float)((c >> 1) & 1) * cw; cz = pr.z + (float)((c >> 2) & 1) * cw;
			if (chunks > 1) {
				float a = 0.f, b = 0.f;
				for (int p = 0; p < P; p++) { a += sh.psum[tid * P + p]; b += sh.psum[kQThreads + tid * P + p]; }
				ub = a; lb = b;
			} else {
				ub = ubs[(size_t)8 * off + cb + tid]; lb = lbs[(size_t)8 * off + cb + tid];
			}
			valid = !qp.boxed || in_box(qp, cx, cy, cz, cw);         // outside the configured translation range: not a candidate
		}
		// incumbent: min ub over the valid children, first index on ties (jly_goicp.cpp:319-324 visits them in order)
		float mub = valid ? ub : INFINITY;
		int midx = valid ? tid : INT_MAX;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const float ou = __shfl_xor(mub, o, 64);
			const int oi = __shfl_xor(midx, o, 64);
			if (ou < mub || (ou == mub && oi < midx)) { mub = ou; midx = oi; }
		}
		if (lane == 0) { sh.red_ub[wave] = mub; sh.red_idx[wave] = midx; }
		__syncthreads();
		mub = sh.red_ub[0]; midx = sh.red_idx[0];
#pragma unroll
		for (int w2 = 1; w2 < kQThreads / 64; w2++)
			if (sh.red_ub[w2] < mub || (sh.red_ub[w2] == mub && sh.red_idx[w2] < midx)) { mub = sh.red_ub[w2]; midx = sh.red_idx[w2]; }
		const bool improved = mub < best;
		if (tid == 0 && mub < S->min_ub) S->min_ub = mub;
		improved_any = improved_any || improved;
		if (improved) best = mub;
		if (improved && tid == midx) { S->bx = cx; S->by = cy; S->bz = cz; S->bw = cw; S->improved = 1; }
		// push the children that can still improve on the incumbent (:327-335), unless the depth limit says leaf
		bool push = valid && lb < best;
		if (push && qp.depth > 0 && node_depth(qp.root_w, cw) >= qp.depth) push = false;
		const unsigned long long pb = __ballot(push);
		if (lane == 0) sh.push_tot[wave] = (unsigned)__popcll(pb);
		__syncthreads();
		unsigned before = 0, total = 0;
#pragma unroll
		for (int w2 = 0; w2 < kQThreads / 64; w2++) { if (w2 < wave) before += sh.push_tot[w2]; total += sh.push_tot[w2]; }
		if (count + (int)total > qp.cap) {
			// The slab is full.  First throw out what can never be expanded: a queued node whose lower bound has come within
			// SSEThresh of the incumbent since it was pushed fails the stop rule (jly_goicp.cpp:257) whenever it is popped --
			// in a long upper-bound search the incumbent keeps falling and most of the queue is such dead weight.  Survivors
			// keep their order.  (The reference's heap just grows; the slab cannot.)
			QNode nd[PER];
			bool keep[PER];
#pragma unroll
			for (int j = 0; j < PER; j++) {
				const int i = j * kQThreads + tid;
				keep[j] = false;
				if (i < count) { nd[j] = Q[i]; keep[j] = !(best - nd[j].lb < qp.thr); }
			}
#pragma unroll
			for (int j = 0; j < PER; j++) {
				const unsigned long long kb = __ballot(keep[j]);
				if (lane == 0) sh.cnt[j][wave] = (unsigned short)__popcll(kb);
			}
			__syncthreads();                                              // every node is in registers; the counts are in LDS
			{
				unsigned wbefore = 0, tot = 0;
				if (lane < PER)
					for (int w2 = 0; w2 < kQThreads / 64; w2++) {
						const unsigned c = sh.cnt[lane][w2];
						tot += c;
						wbefore += w2 < wave ? c : 0u;
					}
				unsigned incl = tot;
#pragma unroll
				for (int o = 1; o < PER; o <<= 1) {
					const unsigned v = __shfl_up(incl, o, 64);
					if (lane >= o) incl += v;
				}
				const unsigned base = incl - tot + wbefore;               // lane j: survivors before (slot j, this wavefront)
				if (tid == PER - 1) sh.bcast = (int)incl;               // wavefront 0: the number of survivors
#pragma unroll
				for (int j = 0; j < PER; j++) {
					const unsigned long long kb = __ballot(keep[j]);
					if (keep[j]) Q[(unsigned)__builtin_amdgcn_readlane((int)base, j) + (unsigned)__popcll(kb & ((1ull << lane) - 1ull))] = nd[j];
				}
			}
			__syncthreads();
			count = sh.bcast;
		}
		if (count + (int)total > qp.cap) {
			// still does not fit: this search stops here and the host re-runs it through its own queues, from its original incumbent (nothing is
			// lost) -- alone (soft_overflow: the other searches of the batch are not its hostages; re-running a whole 1 024-search batch on the
			// host cost the prove-the-optimum bunny 0.8 of its 5.0 s), or with the whole batch
			if (tid == 0) {
				if (qp.soft_overflow) S->done = 2;
				else { atomicExch(&ctl->overflow, 1); S->done = 1; }
				S->n_parents = 0;
			}
			return;
		}
		if (push) {
			const int pos = count + (int)before + (int)__popcll(pb & ((1ull << lane) - 1ull));
			Q[pos] = QNode{cx, cy, cz, cw, ub, lb};
		}
		count += (int)total;
		__syncthreads();                                                 // the pushed nodes are read back below; the LDS scratch is reused by the next pass
		}
		stale = improved_any ? 0 : stale + 1;
		if (tid == 0) { S->pops += n_prev; S->cubes += Ctot; }
	}

	// ------------------------------------------------------------------------------------------------------------
	// select: the K queued nodes with the smallest keys
	// ------------------------------------------------------------------------------------------------------------
	// round width: a search whose incumbent did not move in its last round is proving, not finding -- every queued node with
	// best - lb >= SSEThresh has to be expanded whatever the order (the per-node stop rule below admits nothing else), so a wider
	// round wastes no cube bound and saves rounds; a search that is still improving keeps qp.K (its later nodes may yet be pruned)
	const int n = count;
	int K = qp.K;
	if (qp.stale_widen == 1) K = stale >= 3 ? min(4 * qp.K, kQueueMaxPop) : (stale >= 1 ? min(2 * qp.K, kQueueMaxPop) : qp.K);
	else if (qp.stale_widen == 2) K = stale >= 1 ? min(4 * qp.K, kQueueMaxPop) : qp.K;
	else if (qp.stale_widen == 3) K = stale >= 1 ? min(4 * qp.K, kQueueMaxPop) : min(2 * qp.K, kQueueMaxPop);
	// Compact selection (QParams::stale_compact): a proving search with a LARGE queue takes its nodes in Morton order of their corners instead of
	// by lower bound.  Every queued node that passes the stop rule has to be expanded whatever the order (see above), and a run of
	// Morton-neighbours is (a) spatially compact -- LDS-tile material -- and (b) explored depth-first-like, so its children are pruned or finished
	// before the queue has to hold another level of the whole frontier: the slab stops overflowing.  Measured, prove-the-optimum bunny: mse 3e-5
	// 8.47 -> 6.79 s, five host fall-backs -> none; mse 2e-5 26.95 -> 22.3 s; +1.3 % cube bounds.
	K = min(K, qp.kmax);                               // the searches still running x kmax fit the round's lists
	if (K > kQueueRoundPop) K = max(kQueueRoundPop, min(K, (qp.cap - n) / 8));              // ... and the children of this round's expansions fit the slab (a big step into a nearly full
	                                                   // queue would overflow it: the batch would go back to the host); steps of <= 128 are
	                                                   // left alone -- throttling them by the room only adds rounds (measured)
	const bool compact = qp.stale_compact > 0 && stale >= 1 && n >= qp.stale_compact;
	unsigned key[PER];
	float lbv[PER];
#pragma unroll
	for (int j = 0; j < PER; j++) {
		const int i = j * kQThreads + tid;
		if (i < n) {
			const QNode nd = Q[i];
			lbv[j] = nd.lb;
			if (compact) {
				const float sc = 1024.f / qp.root_w;
				const unsigned qx = (unsigned)min(max((int)((nd.x - qp.root_x) * sc), 0), 1023), qy = (unsigned)min(max((int)((nd.y - qp.root_y) * sc), 0), 1023),
				               qz = (unsigned)min(max((int)((nd.z - qp.root_z) * sc), 0), 1023);
				unsigned m = 0;
#pragma unroll
				for (int b = 0; b < 10; b++) m |= (((qx >> b) & 1u) << (3 * b)) | (((qy >> b) & 1u) << (3 * b + 1)) | (((qz >> b) & 1u) << (3 * b + 2));
				key[j] = best - nd.lb < qp.thr ? 0xfffffffeu : m;       // a node that fails the stop rule is never worth a slot
			} else
				key[j] = node_key(nd.lb, node_depth(qp.root_w, nd.w));
		} else { lbv[j] = INFINITY; key[j] = 0xffffffffu; }
	}
	// smallest key and the stop rule on it (jly_goicp.cpp:257): nothing left that could close the gap -> done
	unsigned kmin = 0xffffffffu;
	float lbmin = INFINITY;
#pragma unroll
	for (int j = 0; j < PER; j++) { const unsigned kc = compact ? __float_as_uint(lbv[j]) : key[j]; if (kc < kmin) { kmin = kc; lbmin = lbv[j]; } }
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		const unsigned ok = __shfl_xor(kmin, o, 64);
		const float ol = __shfl_xor(lbmin, o, 64);
		if (ok < kmin) { kmin = ok; lbmin = ol; }
	}
	if (lane == 0) { sh.wave_tot[wave] = kmin; sh.red_ub[wave] = lbmin; }
	__syncthreads();
	kmin = sh.wave_tot[0]; lbmin = sh.red_ub[0];
#pragma unroll
	for (int w2 = 1; w2 < kQThreads / 64; w2++) if (sh.wave_tot[w2] < kmin) { kmin = sh.wave_tot[w2]; lbmin = sh.red_ub[w2]; }
	__syncthreads();
	if (n == 0 || best - lbmin < qp.thr) {
		if (tid == 0) {
			S->best = best; S->count = n; S->n_parents = 0; S->done = 1;
			if (n > 0) S->pops += 1;                                      // the reference counts the node it pops and rejects
		}
		return;
	}

	unsigned T = 0xffffffffu, rem = 0;                                  // selected: key < T, plus the first `rem` keys == T
	if (n > K) {
		if (tid == 0) { sh.sel_prefix = 0u; sh.sel_rem = (unsigned)K; }
#pragma unroll 1
		for (int pass = 0; pass < 3; pass++) {
			const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0), width = pass == 2 ? 10 : 11, bins = 1 << width;
			for (int i = tid; i < 2048; i += kQThreads) sh.hist[i] = 0u;
			__syncthreads();
			const unsigned prefix = sh.sel_prefix;
#pragma unroll
			for (int j = 0; j < PER; j++) {
				const bool live = j * kQThreads + tid < n && (pass == 0 || (key[j] >> (shift + width)) == prefix);
				if (live) atomicAdd(&sh.hist[(key[j] >> shift) & (unsigned)(bins - 1)], 1u);
			}
			__syncthreads();
			// the bin holding the rem-th smallest candidate: 2048 / threads bins per thread, wavefront scan, wavefront totals
			constexpr int kBins = 2048 / kQThreads;
			unsigned h[kBins], local = 0;
#pragma unroll
			for (int b = 0; b < kBins; b++) { h[b] = sh.hist[kBins * tid + b]; local += h[b]; }
			unsigned incl = local;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				const unsigned v = __shfl_up(incl, o, 64);
				if (lane >= o) incl += v;
			}
			if (lane == 63) sh.wave_tot[wave] = incl;
			__syncthreads();
			unsigned before = 0;
			for (int w2 = 0; w2 < wave; w2++) before += sh.wave_tot[w2];
			const unsigned excl = before + incl - local, want = sh.sel_rem;
			__syncthreads();                                              // every thread has read sel_rem before one rewrites it
			if (excl < want && want <= excl + local) {                        // exactly one thread
				unsigned cum = excl;
#pragma unroll
				for (int b = 0; b < kBins; b++) {
					if (cum < want && want <= cum + h[b]) {
						sh.sel_prefix = (prefix << width) | (unsigned)(kBins * tid + b);
						sh.sel_rem = want - cum;
					}
					cum += h[b];
				}
			}
			__syncthreads();
		}
		T = sh.sel_prefix; rem = sh.sel_rem;
	}
	// selected flags; the ties (key == T) are taken in queue order.  Per (j, wavefront) tie counts, then ranks.
	unsigned tie_before[1] = {0};
	bool sel[PER];
	if (n > K) {
#pragma unroll
		for (int j = 0; j < PER; j++) {
			const unsigned long long tb = __ballot(key[j] == T);
			if (lane == 0) sh.cnt[j][wave] = (unsigned short)__popcll(tb);
		}
		__syncthreads();
		unsigned wbefore = 0, total = 0;
		if (lane < PER)
			for (int w2 = 0; w2 < kQThreads / 64; w2++) {
				const unsigned c = sh.cnt[lane][w2];
				total += c;
				wbefore += w2 < wave ? c : 0u;
			}
		unsigned incl = total;
#pragma unroll
		for (int o = 1; o < PER; o <<= 1) {
			const unsigned v = __shfl_up(incl, o, 64);
			if (lane >= o) incl += v;
		}
		const unsigned base = incl - total + wbefore;                    // lane j: ties before (iteration j, this wavefront)
#pragma unroll
		for (int j = 0; j < PER; j++) {
			const bool tie = key[j] == T;
			const unsigned long long tb = __ballot(tie);
			const unsigned rank = (unsigned)__builtin_amdgcn_readlane((int)base, j) + (unsigned)__popcll(tb & ((1ull << lane) - 1ull));
			sel[j] = key[j] < T || (tie && rank < rem);
		}
		__syncthreads();
	} else {
#pragma unroll
		for (int j = 0; j < PER; j++) sel[j] = j * kQThreads + tid < n;
	}
	(void)tie_before;
	// the stop rule per node (:257): a selected node that can no longer close the gap stays queued
#pragma unroll
	for (int j = 0; j < PER; j++) sel[j] = sel[j] && !(best - lbv[j] < qp.thr);

	// rank of every selected node in queue order -> slot in this round's expansion list
#pragma unroll
	for (int j = 0; j < PER; j++) {
		const unsigned long long sb = __ballot(sel[j]);
		if (lane == 0) sh.cnt[j][wave] = (unsigned short)__popcll(sb);
	}
	__syncthreads();
	{
		unsigned wbefore = 0, total = 0;
		if (lane < PER)
			for (int w2 = 0; w2 < kQThreads / 64; w2++) {
				const unsigned c = sh.cnt[lane][w2];
				total += c;
				wbefore += w2 < wave ? c : 0u;
			}
		unsigned incl = total;
#pragma unroll
		for (int o = 1; o < PER; o <<= 1) {
			const unsigned v = __shfl_up(incl, o, 64);
			if (lane >= o) incl += v;
		}
		const unsigned base = incl - total + wbefore;
		if (tid == PER - 1) sh.n_sel = (int)incl;                       // wavefront 0, lane 31: total over all (j, wavefront)
#pragma unroll
		for (int j = 0; j < PER; j++) {
			const unsigned long long sb = __ballot(sel[j]);
			if (sel[j]) {
				const unsigned r = (unsigned)__builtin_amdgcn_readlane((int)base, j) + (unsigned)__popcll(sb & ((1ull << lane) - 1ull));
				sh.sel_pos[r] = j * kQThreads + tid;
			}
		}
	}
	__syncthreads();
	const int n_sel = sh.n_sel;
	if (n_sel == 0) {
		// only possible when nodes share the smallest (truncated) key and the first of them in queue order misses the
		// stop rule by a rounding: the gap is closed to within 2^-19 relative -> done
		if (tid == 0) { S->best = best; S->count = n; S->n_parents = 0; S->done = 1; S->pops += 1; }
		return;
	}
	// ---- which list: the tile list takes a search whose selected nodes lie within a few voxels of each other (then the DT box
	// a 64-point patch of the cloud can reach under ALL of them fits the LDS tile) and that has enough of them to fill lanes ----
	QNode mine_nd{};
	if (tid < n_sel) mine_nd = Q[sh.sel_pos[tid]];
	bool to_tile = false, deep_now = false;
	if (qp.tile_spread > 0.f) {
		float lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY}, wmax = 0.f;
		if (tid < n_sel) { lo3[0] = hi3[0] = mine_nd.x; lo3[1] = hi3[1] = mine_nd.y; lo3[2] = hi3[2] = mine_nd.z; wmax = mine_nd.w; }
		if (wave < kQueueMaxPop / 64) {
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
				for (int k = 0; k < 3; k++) { lo3[k] = fminf(lo3[k], __shfl_xor(lo3[k], o, 64)); hi3[k] = fmaxf(hi3[k], __shfl_xor(hi3[k], o, 64)); }
				wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
			}
			if (lane == 0) {
#pragma unroll
				for (int k = 0; k < 3; k++) { sh.ext[wave][k] = lo3[k]; sh.ext[wave][3 + k] = hi3[k]; }
				sh.ext_w[wave] = wmax;
			}
		}
		__syncthreads();
		float spread = 0.f, wmx = 0.f;
#pragma unroll
		for (int k = 0; k < 3; k++) {
			float lo = INFINITY, hi = -INFINITY;
#pragma unroll
			for (int w2 = 0; w2 < kQueueMaxPop / 64; w2++) { lo = fminf(lo, sh.ext[w2][k]); hi = fmaxf(hi, sh.ext[w2][3 + k]); }
			spread = fmaxf(spread, hi - lo);
		}
#pragma unroll
		for (int w2 = 0; w2 < kQueueMaxPop / 64; w2++) wmx = fmaxf(wmx, sh.ext_w[w2]);
		spread += wmx;                                                  // the children's centres reach w/4 .. 3w/4 beyond the corners
		to_tile = spread <= qp.tile_spread;
		deep_now = to_tile;
		if (qp.tile_stats && tid == 0) {
			const float v = spread * qp.tile_stats_scale;                  // voxels
			atomicAdd(&ctl->sel_hist[n_sel < 16 ? 0 : (n_sel < 32 ? 1 : (n_sel < 64 ? 2 : 3))][v <= 5.f ? 0 : (v <= 10.f ? 1 : (v <= 20.f ? 2 : 3))], n_sel);
		}
		if (to_tile && tid == 0) atomicAdd(&ctl->tile_hint, 1);
		to_tile = to_tile && qp.tile_on != 0 && n_sel >= qp.tile_min;
	}
	if (tid == 0) {
		if (to_tile) {
			const int nseg = (n_sel + 63) >> 6;
			sh.parent_off = atomicAdd(&ctl->n_tile_groups[parity], n_sel);
			atomicAdd(&ctl->tile_total, n_sel);
			const int seg0 = atomicAdd(&ctl->n_tile_segs[parity], nseg);
			// the segment list holds list_cap / 64 + (search slots) entries: sum ceil(n_i / 64) <= sum n_i / 64 + searches, and sum n_i <= list_cap
			// is checked below -- so this cannot trip while the host sized the lists; if it ever does, nothing is written past them
			if (seg0 + nseg > qp.seg_cap || sh.parent_off + n_sel > qp.list_cap) sh.parent_off = qp.list_cap;       // -> the overflow exit below
			else for (int k = 0; k < nseg; k++) tile.segs[parity][seg0 + k] = TileSeg{sh.parent_off + 64 * k, min(64, n_sel - 64 * k), S->rot};
		} else
			sh.parent_off = atomicAdd(&ctl->n_groups[parity], n_sel);
		atomicAdd(&ctl->n_active[parity], 1);
	}
	__syncthreads();
	const int off = sh.parent_off;
	if (off + n_sel > qp.list_cap) {
		// cannot happen while the host's kmax is what it should be; if it ever does, nothing is written past the lists: the batch goes to the host queues
		if (tid == 0) { atomicExch(&ctl->overflow, 1); S->done = 1; S->n_parents = 0; }
		return;
	}
	const float coeff = S->coeff;
	const int rot = S->rot;
	if (tid < n_sel) (to_tile ? tile.parents[parity] : parents)[off + tid] = ParentRec{mine_nd.x, mine_nd.y, mine_nd.z, mine_nd.w, coeff, rot};
	if (parent_search && !to_tile && tid < n_sel) parent_search[off + tid] = s;
	__syncthreads();
	// remove the selected nodes: the holes among the first m = n - n_sel positions are filled, in order, with the
	// unselected nodes of the tail [m, n)  (at most n_sel <= kQueueMaxPop of each: the first eight wavefronts do it)
	const int m = n - n_sel;
	{
		// which of the (at most n_sel) tail positions [m, n) are themselves selected: their owners say so
#pragma unroll
		for (int j = 0; j < PER; j++) {
			const int i = j * kQThreads + tid;
			if (i >= m && i < n) sh.tail_sel[i - m] = sel[j] ? 1 : 0;
		}
		__syncthreads();
		const bool is_hole = tid < n_sel && sh.sel_pos[tid < kQueueMaxPop ? tid : 0] < m;
		const int tailpos = m + tid;
		const bool filler = tid < n_sel && tailpos < n && !sh.tail_sel[tid < kQueueMaxPop ? tid : 0];
		const unsigned long long hb = __ballot(is_hole), fb = __ballot(filler);
		if (wave < kQueueMaxPop / 64 && lane == 0) { sh.hole_cnt[wave] = (int)__popcll(hb); sh.fill_cnt[wave] = (int)__popcll(fb); }
		__syncthreads();
		int hrank = (int)__popcll(hb & ((1ull << lane) - 1ull)), frank = (int)__popcll(fb & ((1ull << lane) - 1ull));
		for (int w2 = 0; w2 < min(wave, kQueueMaxPop / 64); w2++) { hrank += sh.hole_cnt[w2]; frank += sh.fill_cnt[w2]; }
		if (is_hole) sh.hole_pos[hrank] = sh.sel_pos[tid];
		QNode moved{};
		if (filler) moved = Q[tailpos];
		__syncthreads();
		if (filler) Q[sh.hole_pos[frank]] = moved;
	}
	if (tid == 0) { S->best = best; S->count = m; S->n_parents = n_sel; S->parent_off = off; S->tile = to_tile ? 1 : 0; S->deep = deep_now ? 1 : 0; S->stale = stale; }
}

// nodes of search s: q[s * kQueueCap ...]
// WAVES = wavefronts per SIMD the kernel is built for.  8: 64 VGPRs, two searches per CU -- the deep batches, where queue kernels of several
// lanes and long bound evaluations compete for the chip (99 spilled registers, nearly all on the purge path).  4: 128 VGPRs, no spill, one search
// per CU -- every other batch: a default registration's queue kernels are latency chains (bunny 32.5 -> 31.9 ms, skull 6.3 -> 6.2, bunny mse 1e-4
// 260 -> 253 ms; on the prove-the-optimum run the same build costs 5.02 -> 5.18 s, hence the two forms).  Same code, same results.
template <int WAVES>
__global__ __launch_bounds__(kQThreads, WAVES) void bnb_queue_kernel(QSearch* __restrict__ searches, QNode* __restrict__ q, QParams qp,
                                                              const ParentRec* __restrict__ prev_parents, ParentRec* __restrict__ parents,
                                                              const float* __restrict__ ubs, const float* __restrict__ lbs,
                                                              const float* __restrict__ scratch, QCtl* __restrict__ ctl, int parity, QTile tile, int* __restrict__ parent_search)
{
	__shared__ QShared sh;
	const int s = blockIdx.x, tid = threadIdx.x;
	QSearch* __restrict__ S = searches + s;
	QNode* __restrict__ Q = q + (size_t)s * kQueueCap;
	if (s == 0 && tid < 8) {
		if (tid == 0) { ctl->n_groups[parity ^ 1] = 0; ctl->n_tile_groups[parity ^ 1] = 0; ctl->n_tile_segs[parity ^ 1] = 0; ctl->n_active[parity ^ 1] = 0; }   // the next round's counters (their last readers have finished)
		ctl->work[parity][tid] = 0;                                   // this round's work counters of the bound evaluation
	}
	if (S->done) return;
	if (S->count + 8 * S->n_parents <= kQThreads) queue_round<1>(sh, S, Q, qp, prev_parents, parents, ubs, lbs, scratch, ctl, parity, tile, parent_search, s);
	else queue_round<kQPer>(sh, S, Q, qp, prev_parents, parents, ubs, lbs, scratch, ctl, parity, tile, parent_search, s);
}

__global__ void bnb_init_kernel(QSearch* __restrict__ searches, QNode* __restrict__ q, int nsearch, QParams qp, QCtl* __restrict__ ctl)
{
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s == 0) {
		ctl->n_groups[0] = 0; ctl->n_groups[1] = 0; ctl->overflow = 0; ctl->n_active[0] = ctl->n_active[1] = 0;
		for (int k = 0; k < 8; k++) { ctl->work[0][k] = 0; ctl->work[1][k] = 0; }
		ctl->n_tile_groups[0] = ctl->n_tile_groups[1] = 0; ctl->n_tile_segs[0] = ctl->n_tile_segs[1] = 0; ctl->tile_chunks = 1; ctl->tile_hint = 0; ctl->tile_total = 0;
		for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) ctl->sel_hist[a][b] = 0;
	}
	if (s >= nsearch) return;
	q[(size_t)s * kQueueCap] = QNode{qp.root_x, qp.root_y, qp.root_z, qp.root_w, 0.f, 0.f};   // jly_goicp.cpp:50-53, :241
	QSearch& S = searches[s];                                              // best / coeff / rot were uploaded by the host
	S.count = 1; S.done = 0; S.improved = 0; S.n_parents = 0; S.parent_off = 0; S.pops = 0; S.cubes = 0;
	S.bx = S.by = S.bz = S.bw = 0.f; S.tile = 0; S.min_ub = INFINITY; S.deep = 0; S.stale = 0;   // twin: uploaded by the host
}

// the listed slots become fresh searches (continuous flow: slots are recycled while other searches keep running)
__global__ void bnb_init_list_kernel(QSearch* __restrict__ searches, QNode* __restrict__ q, const QInit* __restrict__ list, int n, QParams qp)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const QInit in = list[i];
	q[(size_t)in.slot * kQueueCap] = QNode{qp.root_x, qp.root_y, qp.root_z, qp.root_w, 0.f, 0.f};
	QSearch& S = searches[in.slot];
	S.best = in.best; S.coeff = in.coeff; S.rot = in.rot; S.twin = in.twin;
	S.count = 1; S.done = 0; S.improved = 0; S.n_parents = 0; S.parent_off = 0; S.pops = 0; S.cubes = 0;
	S.bx = S.by = S.bz = S.bw = 0.f; S.tile = 0; S.min_ub = INFINITY; S.deep = 0; S.stale = 0;
}

hipError_t launch_bnb_init_list(QSearch* searches, QNode* q, const QInit* d_list, int n, const QParams& qp, hipStream_t stream)
{
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(bnb_init_list_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, searches, q, d_list, n, qp);
	return hipGetLastError();
}

hipError_t launch_bnb_init(QSearch* searches, QNode* q, int nsearch, const QParams& qp, QCtl* ctl, hipStream_t stream)
{
	if (nsearch <= 0) return hipSuccess;
	hipLaunchKernelGGL(bnb_init_kernel, dim3((nsearch + 255) / 256), dim3(256), 0, stream, searches, q, nsearch, qp, ctl);
	return hipGetLastError();
}

hipError_t launch_bnb_queue(QSearch* searches, QNode* q, int nsearch, const QParams& qp, const ParentRec* prev_parents, ParentRec* parents,
                            const float* ubs, const float* lbs, const float* scratch, QCtl* ctl, int parity, hipStream_t stream, const QTile* tile, int* parent_search,
                            bool deep)
{
	if (nsearch <= 0) return hipSuccess;
	if (qp.K < 1 || qp.K > kQueueMaxPop) return hipErrorInvalidValue;
	QTile t{};
	QParams q2 = qp;
	if (tile && tile->ub) t = *tile;
	else { q2.tile_on = 0; }                                               // no buffers: nothing may be listed there
	if (deep) hipLaunchKernelGGL(bnb_queue_kernel<8>, dim3(nsearch), dim3(kQThreads), 0, stream, searches, q, q2, prev_parents, parents, ubs, lbs, scratch, ctl, parity, t, parent_search);
	else hipLaunchKernelGGL(bnb_queue_kernel<4>, dim3(nsearch), dim3(kQThreads), 0, stream, searches, q, q2, prev_parents, parents, ubs, lbs, scratch, ctl, parity, t, parent_search);
	return hipGetLastError();
}

}  // namespace goicp
