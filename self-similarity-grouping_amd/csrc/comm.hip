// comm.hip -- the collective entry points of the C ABI (ssg_comm_init / ssg_allgather / ssg_comm_destroy), RCCL
// over xGMI, one communicator rank per process / GPU.
//
// The reference spreads its extraction over the GPUs of a node with nn.DataParallel (selftraining.py:135: scatter of every
// batch, gather of the features through GPU 0) and runs the N x N work on one CPU.  Here every exchange of the sharded path
// (ssg_amd/dist.py) is an all-gather of equally sized blocks -- the embeddings (C1), the rank lists, the sparse V / V_qe rows,
// the source vector -- plus int64 sum all-reduces for the eps histogram; these four calls are all a host language needs to bind.
// The Python product drives the same exchanges through torch.distributed, whose "nccl" backend IS this library on ROCm;
// `ssg_amd.dist.AbiComm` is the thin wrapper over the entry points below (world-size-1 GPU test: tests/test_abi.py).
#include "ssg_common.h"
#include <rccl/rccl.h>
#include <cstring>

static int ssg_check_nccl(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return SSG_OK;
  ssg_set_error("RCCL error %d (%s) at %s", (int)r, ncclGetErrorString(r), what);
  return SSG_ERR_HIP;
}
#define SSG_NCCL(call) do { int rc_ = ssg_check_nccl((call), #call); if (rc_) return rc_; } while (0)

static_assert(sizeof(ncclUniqueId) == 128, "ssg_comm_unique_id hands out 128 bytes");

// 128 bytes that rank 0 creates and every rank passes to ssg_comm_init (ship them over any side channel: a file, a socket, MPI,
// torch.distributed's store)
extern "C" int ssg_comm_unique_id(void* id128_host) {
  if (!id128_host) { ssg_set_error("ssg_comm_unique_id: null buffer"); return SSG_ERR_INVALID; }
  ncclUniqueId id;
  SSG_NCCL(ncclGetUniqueId(&id));
  memcpy(id128_host, &id, sizeof(id));
  return SSG_OK;
}

// collective over all `world` processes; the communicator is bound to the calling thread's current HIP device
extern "C" int ssg_comm_init(void** comm, int world, int rank, const void* id128_host) {
  if (!comm || !id128_host || world < 1 || rank < 0 || rank >= world) {
    ssg_set_error("ssg_comm_init: bad arguments (world=%d rank=%d)", world, rank);
    return SSG_ERR_INVALID;
  }
  ncclUniqueId id;
  memcpy(&id, id128_host, sizeof(id));
  ncclComm_t c = nullptr;
  SSG_NCCL(ncclCommInitRank(&c, world, id, rank));
  *comm = (void*)c;
  return SSG_OK;
}

// recv[r * bytes_per_rank ...] = rank r's send block, for every r (in place when send == recv + rank * bytes_per_rank)
extern "C" int ssg_allgather(void* comm, const void* send, void* recv, size_t bytes_per_rank, hipStream_t stream) {
  if (!comm || (bytes_per_rank && (!send || !recv))) { ssg_set_error("ssg_allgather: null communicator / buffer"); return SSG_ERR_INVALID; }
  if (bytes_per_rank == 0) return SSG_OK;
  SSG_NCCL(ncclAllGather(send, recv, bytes_per_rank, ncclInt8, (ncclComm_t)comm, stream));
  return SSG_OK;
}

// in-place element-wise sum of int64 vectors over the ranks (eps histograms, candidate counts)
extern "C" int ssg_allreduce_sum_i64(void* comm, int64_t* buf, size_t count, hipStream_t stream) {
  if (!comm || (count && !buf)) { ssg_set_error("ssg_allreduce_sum_i64: null communicator / buffer"); return SSG_ERR_INVALID; }
  if (count == 0) return SSG_OK;
  SSG_NCCL(ncclAllReduce(buf, buf, count, ncclInt64, ncclSum, (ncclComm_t)comm, stream));
  return SSG_OK;
}

extern "C" int ssg_comm_destroy(void* comm) {
  if (!comm) return SSG_OK;
  SSG_NCCL(ncclCommDestroy((ncclComm_t)comm));
  return SSG_OK;
}
