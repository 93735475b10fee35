// allreduce.hip -- the data-parallel exchange of the training step behind the C ABI (SURVEY 8b `allreduce_flat`,
// 8e): ONE in-place SUM all-reduce of the flat fp32 gradient buffer over RCCL (xGMI inside a node), in stream order
// between the backward pass and the fused clip + Adam kernel.  The reference loop has no counterpart (sbi trains on one
// device, trainers/base.py:1150-1193).
// librccl.so is resolved at FIRST USE with dlopen / dlsym -- the kernel library itself carries no link-time dependency
// on it (single-GPU users never load it); a process that already runs torch.distributed's `nccl` backend gets the very
// same library instance.  Python's default stays torch.distributed (sbi_amd/utils/collectives.py); this entry point is
// what a C / C++ host binds, and what `NativeAllReduce` uses when asked to.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>

#include <mutex>

#include "../../include/sbi_amd_nsf.h"

namespace {
struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
  ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
  ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void rccl_load() {
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) return;
  g_rccl.get_unique_id = (decltype(g_rccl.get_unique_id))dlsym(g_rccl.handle, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (decltype(g_rccl.comm_init_rank))dlsym(g_rccl.handle, "ncclCommInitRank");
  g_rccl.comm_destroy = (decltype(g_rccl.comm_destroy))dlsym(g_rccl.handle, "ncclCommDestroy");
  g_rccl.all_reduce = (decltype(g_rccl.all_reduce))dlsym(g_rccl.handle, "ncclAllReduce");
  g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.all_reduce;
}
bool rccl_ready() {
  std::call_once(g_rccl_once, rccl_load);
  return g_rccl.ok;
}
// RCCL's result codes are small positive integers as well: keep them apart from HIP's by an offset
inline int rc_of(ncclResult_t r) { return r == ncclSuccess ? 0 : 10000 + (int)r; }
}  // namespace

extern "C" int32_t sbi_amd_rccl_unique_id_bytes(void) { return (int32_t)sizeof(ncclUniqueId); }

extern "C" int sbi_amd_rccl_unique_id(void* id_out) {
  if (!id_out) return SBI_AMD_E_BADARG;
  if (!rccl_ready()) return SBI_AMD_E_UNSUPPORTED;
  ncclUniqueId id;
  const ncclResult_t r = g_rccl.get_unique_id(&id);
  if (r == ncclSuccess) memcpy(id_out, &id, sizeof(id));
  return rc_of(r);
}

extern "C" int sbi_amd_rccl_comm_init(void** comm_out, int32_t world, int32_t rank, const void* id_bytes) {
  if (!comm_out || !id_bytes || world < 1 || rank < 0 || rank >= world) return SBI_AMD_E_BADARG;
  if (!rccl_ready()) return SBI_AMD_E_UNSUPPORTED;
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof(id));
  ncclComm_t comm = nullptr;
  const ncclResult_t r = g_rccl.comm_init_rank(&comm, world, id, rank);     // (on the calling thread's current device)
  *comm_out = r == ncclSuccess ? (void*)comm : nullptr;
  return rc_of(r);
}

extern "C" int sbi_amd_rccl_comm_destroy(void* comm) {
  if (!comm) return SBI_AMD_E_BADARG;
  if (!rccl_ready()) return SBI_AMD_E_UNSUPPORTED;
  return rc_of(g_rccl.comm_destroy((ncclComm_t)comm));
}

// grad_bucket <- sum over the communicator's ranks of grad_bucket (in place, fp32), enqueued on `stream`
extern "C" int sbi_amd_allreduce_flat(void* comm, float* grad_bucket, int64_t count, void* stream) {
  if (!comm || !grad_bucket || count < 0) return SBI_AMD_E_BADARG;
  if (count == 0) return 0;
  if (!rccl_ready()) return SBI_AMD_E_UNSUPPORTED;
  return rc_of(g_rccl.all_reduce(grad_bucket, grad_bucket, (size_t)count, ncclFloat, ncclSum, (ncclComm_t)comm,
                                 (hipStream_t)stream));
}
