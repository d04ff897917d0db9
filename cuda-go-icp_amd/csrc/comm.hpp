// Internal declarations shared by the sharding protocol (shard.cpp), the RCCL communicator (rccl_comm.cpp), the C ABI
// (goicp_api.cpp) and the sanitizer driver (host_selftest.cpp).
#pragma once
#include <cstdint>

#include "../../include/goicp_mi355.h"

namespace goicp {

// first member of the context of every communicator the library makes (thread, RCCL): lets goicp_comm_set_timeout_ms
// find the deadline without knowing the kind
struct CommHeader { uint32_t magic; int32_t timeout_ms; };
constexpr uint32_t kCommMagic = 0x43494f47u;     // "GOIC"

int comm_default_timeout_ms();                   // GOICP_COMM_TIMEOUT_MS, else 60 000
int comm_set_timeout_ms(goicp_comm_ops* comm, int ms);
// A communicator of the library is recognised by its all-reduce FUNCTION (registered here by the kind that owns it), never by
// peeking into ctx: the ABI lets a caller bring its own communicator, whose ctx may be anything -- or nothing readable.
using CommAllreduceFn = int (*)(void*, uint64_t*, size_t);
void comm_register_library_kind(CommAllreduceFn fn);
bool comm_is_library_kind(const goicp_comm_ops* comm);

int run_sharded(const goicp_shard_engine_ops* eng, const goicp_comm_ops* comm, const goicp_shard_options* opt, goicp_shard_stats* stats);
int thread_comm_create(int world, goicp_comm_ops* out);
void thread_comm_destroy(goicp_comm_ops* comm);

int rccl_unique_id(char id128[GOICP_RCCL_ID_BYTES]);
int rccl_comm_create(const char id128[GOICP_RCCL_ID_BYTES], int32_t rank, int32_t world, int32_t device, goicp_comm_ops* out);
int rccl_comm_wrap(void* nccl_comm, int32_t rank, int32_t world, int32_t device, goicp_comm_ops* out);
int rccl_comm_destroy(goicp_comm_ops* comm);
int rccl_comm_init_all(int world, void** comms);
void rccl_comm_destroy_raw(void* comm);

}  // namespace goicp
