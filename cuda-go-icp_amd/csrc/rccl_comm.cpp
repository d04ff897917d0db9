// RCCL communicator behind goicp_comm_ops: the exchanges of the sharded search (shard.cpp) as ncclAllReduce(MIN) /
// ncclBroadcast over xGMI.  Payloads are 40-48 bytes (the donations 1.8 KB), so the collectives are latency-bound;
// they run on a stream of their own -- never the engine's compute stream -- with one pinned staging block per
// communicator (H2D, collective, D2H on that stream).  The host never blocks in hipStreamSynchronize: it polls
// hipStreamQuery against the communicator's deadline (default 60 s), so a rank that died or left the protocol turns
// into GOICP_ERR_TIMEOUT on the others instead of a hang; a communicator that missed a deadline is marked broken and,
// when the library owns it, torn down with ncclCommAbort.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "comm.hpp"

namespace goicp {

namespace {

struct RcclComm {
	CommHeader hdr{kCommMagic, 60000};     // first member (goicp_comm_set_timeout_ms)
	bool broken = false;                   // a collective missed its deadline
	ncclComm_t comm = nullptr;
	bool owns = false;
	int device = 0;
	hipStream_t stream = nullptr;
	void* d_buf = nullptr;
	void* h_buf = nullptr;
	size_t cap = 0;
	std::string err;
};

constexpr size_t kStageBytes = 1 << 16;

struct DevScope {
	int prev = -1, want;
	explicit DevScope(int d) : want(d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != want) hipSetDevice(want); }
	~DevScope() { if (prev >= 0 && prev != want) hipSetDevice(prev); }
};

// wait for the communicator's stream without ever blocking past the deadline: a short spin (the exchange takes 10-20 us),
// then yields, then 50 us naps
int wait_stream(RcclComm* c)
{
	const auto t0 = std::chrono::steady_clock::now();
	const auto deadline = t0 + std::chrono::milliseconds(c->hdr.timeout_ms);
	for (long spins = 0;; spins++) {
		const hipError_t q = hipStreamQuery(c->stream);
		if (q == hipSuccess) return GOICP_OK;
		if (q != hipErrorNotReady) { (void)hipGetLastError(); c->broken = true; return GOICP_ERR_DEVICE; }
		(void)hipGetLastError();
		ncclResult_t async = ncclSuccess;
		if ((spins & 1023) == 1023 && (ncclCommGetAsyncError(c->comm, &async) != ncclSuccess || (async != ncclSuccess && async != ncclInProgress))) {
			c->broken = true;
			return GOICP_ERR_DEVICE;
		}
		const auto now = std::chrono::steady_clock::now();
		if (now >= deadline) { c->broken = true; return GOICP_ERR_TIMEOUT; }
		if (now - t0 > std::chrono::milliseconds(2)) std::this_thread::sleep_for(std::chrono::microseconds(50));
		else if (now - t0 > std::chrono::microseconds(200)) std::this_thread::yield();
	}
}

int allreduce_min_u64(void* ctx, uint64_t* words, size_t n)
{
	RcclComm* c = static_cast<RcclComm*>(ctx);
	if (c->broken) return GOICP_ERR_TIMEOUT;
	if (n * sizeof(uint64_t) > c->cap) return GOICP_ERR_INVALID;
	DevScope dev(c->device);
	std::memcpy(c->h_buf, words, n * sizeof(uint64_t));
	// any failure to enqueue leaves the stream in an unknown state: the communicator is broken from here on (destroy then aborts
	// it instead of waiting on a possibly wedged stream)
	auto fail = [&] { c->broken = true; (void)hipGetLastError(); return GOICP_ERR_DEVICE; };
	if (hipMemcpyAsync(c->d_buf, c->h_buf, n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream) != hipSuccess) return fail();
	if (ncclAllReduce(c->d_buf, c->d_buf, n, ncclUint64, ncclMin, c->comm, c->stream) != ncclSuccess) return fail();
	if (hipMemcpyAsync(c->h_buf, c->d_buf, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return fail();
	if (const int rc = wait_stream(c)) return rc;
	std::memcpy(words, c->h_buf, n * sizeof(uint64_t));
	return GOICP_OK;
}

int bcast(void* ctx, void* buf, size_t bytes, int32_t root)
{
	RcclComm* c = static_cast<RcclComm*>(ctx);
	if (c->broken) return GOICP_ERR_TIMEOUT;
	if (bytes > c->cap) return GOICP_ERR_INVALID;
	DevScope dev(c->device);
	std::memcpy(c->h_buf, buf, bytes);
	auto fail = [&] { c->broken = true; (void)hipGetLastError(); return GOICP_ERR_DEVICE; };
	if (hipMemcpyAsync(c->d_buf, c->h_buf, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return fail();
	if (ncclBroadcast(c->d_buf, c->d_buf, bytes, ncclUint8, root, c->comm, c->stream) != ncclSuccess) return fail();
	if (hipMemcpyAsync(c->h_buf, c->d_buf, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return fail();
	if (const int rc = wait_stream(c)) return rc;
	std::memcpy(buf, c->h_buf, bytes);
	return GOICP_OK;
}

int finish(RcclComm* c, int32_t rank, int32_t world, goicp_comm_ops* out)
{
	DevScope dev(c->device);
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc(&c->d_buf, kStageBytes) != hipSuccess ||
	    hipHostMalloc(&c->h_buf, kStageBytes) != hipSuccess) {
		if (c->stream) hipStreamDestroy(c->stream);
		hipFree(c->d_buf); hipHostFree(c->h_buf);
		if (c->owns && c->comm) ncclCommDestroy(c->comm);
		delete c;
		return GOICP_ERR_DEVICE;
	}
	c->cap = kStageBytes;
	c->hdr.timeout_ms = comm_default_timeout_ms();
	out->ctx = c; out->rank = rank; out->world = world;
	out->allreduce_min_u64 = &allreduce_min_u64;
	out->bcast = &bcast;
	comm_register_library_kind(&allreduce_min_u64);
	return GOICP_OK;
}

}  // namespace

int rccl_unique_id(char id128[GOICP_RCCL_ID_BYTES])
{
	static_assert(sizeof(ncclUniqueId) == GOICP_RCCL_ID_BYTES, "ncclUniqueId is 128 bytes");
	ncclUniqueId id;
	if (ncclGetUniqueId(&id) != ncclSuccess) return GOICP_ERR_DEVICE;
	std::memcpy(id128, &id, sizeof(id));
	return GOICP_OK;
}

int rccl_comm_create(const char id128[GOICP_RCCL_ID_BYTES], int32_t rank, int32_t world, int32_t device, goicp_comm_ops* out)
{
	RcclComm* c = new (std::nothrow) RcclComm;
	if (!c) return GOICP_ERR_INTERNAL;
	c->device = device; c->owns = true;
	{
		DevScope dev(device);
		ncclUniqueId id;
		std::memcpy(&id, id128, sizeof(id));
		if (ncclCommInitRank(&c->comm, world, id, rank) != ncclSuccess) { delete c; return GOICP_ERR_DEVICE; }
	}
	return finish(c, rank, world, out);
}

int rccl_comm_wrap(void* nccl_comm, int32_t rank, int32_t world, int32_t device, goicp_comm_ops* out)
{
	RcclComm* c = new (std::nothrow) RcclComm;
	if (!c) return GOICP_ERR_INTERNAL;
	c->device = device; c->owns = false; c->comm = static_cast<ncclComm_t>(nccl_comm);
	return finish(c, rank, world, out);
}

int rccl_comm_destroy(goicp_comm_ops* comm)
{
	if (!comm || !comm->ctx) return GOICP_OK;
	RcclComm* c = static_cast<RcclComm*>(comm->ctx);
	{
		DevScope dev(c->device);
		if (c->broken) {
			// a collective is stuck on the stream: abort the communicator first (its kernels then leave), never wait on it;
			// a wrapped communicator stays the caller's to abort, its stream and staging blocks are left alone (leaked)
			if (c->owns && c->comm) {
				ncclCommAbort(c->comm);
				hipStreamDestroy(c->stream);
				hipFree(c->d_buf); hipHostFree(c->h_buf);
			}
		} else {
			hipStreamSynchronize(c->stream);
			hipStreamDestroy(c->stream);
			hipFree(c->d_buf); hipHostFree(c->h_buf);
			if (c->owns && c->comm) ncclCommDestroy(c->comm);
		}
	}
	delete c;
	comm->ctx = nullptr;
	return GOICP_OK;
}

// world communicators of one process (ncclCommInitAll), one per device 0..world-1
int rccl_comm_init_all(int world, void** comms)
{
	std::vector<int> devs((size_t)world);
	for (int i = 0; i < world; i++) devs[(size_t)i] = i;
	std::vector<ncclComm_t> cs((size_t)world);
	if (ncclCommInitAll(cs.data(), world, devs.data()) != ncclSuccess) return GOICP_ERR_DEVICE;
	for (int i = 0; i < world; i++) comms[i] = cs[(size_t)i];
	return GOICP_OK;
}
void rccl_comm_destroy_raw(void* comm) { if (comm) ncclCommDestroy(static_cast<ncclComm_t>(comm)); }

}  // namespace goicp
