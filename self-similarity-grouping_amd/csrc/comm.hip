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
#include <rccl/rccl.h>          // types and prototypes only: the library itself is bound at run time (below)
#include <dlfcn.h>
#include <link.h>
#include <cstring>
#include <mutex>
#include <string>

// One RCCL per process.  libssg_hip.so does NOT link librccl: a process that also runs torch.distributed already
// maps torch's own build (torch/lib/librccl.so), and a second copy from /opt/rocm/lib would give it two sets of RCCL globals (two
// bootstrap threads, two views of the xGMI topology, two IPC caches).  The first collective call binds the six entry points to
//   1. the RCCL that is ALREADY mapped into the process, whichever it is (dl_iterate_phdr -> dlopen(path, RTLD_NOLOAD)),
//   2. else $SSG_RCCL_PATH, else "librccl.so.1" from the loader's default path (/opt/rocm/lib via the RUNPATH of this library).
namespace {
struct Rccl {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  std::string path;
  int version = 0;
  bool ok = false;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;        // the first collective of two threads must not race on the table (bound once, under the lock)

int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* out) {
  if (info->dlpi_name && strstr(info->dlpi_name, "librccl")) { *static_cast<std::string*>(out) = info->dlpi_name; return 1; }
  return 0;
}

// ncclGetVersion's code: major * 10000 + minor * 100 + patch from 2.9 on (major * 1000 + ... before)
int rccl_major(int code) { return code >= 10000 ? code / 10000 : code / 1000; }

int rccl_bind() {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.ok) return SSG_OK;
  std::string loaded;
  dl_iterate_phdr(find_loaded_rccl, &loaded);
  void* h = nullptr;
  if (!loaded.empty()) h = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD);
  if (!h) {
    const char* e = getenv("SSG_RCCL_PATH");
    loaded = e && *e ? e : "librccl.so.1";
    h = dlopen(loaded.c_str(), RTLD_NOW | RTLD_LOCAL);
  }
  if (!h) { ssg_set_error("RCCL not available: %s", dlerror()); return SSG_ERR_HIP; }
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(h, "ncclAllGather"));
  g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.AllReduce || !g_rccl.CommDestroy || !g_rccl.GetErrorString) {
    ssg_set_error("RCCL at %s lacks an entry point", loaded.c_str());
    return SSG_ERR_HIP;
  }
  // the table is typed by the build-time rccl.h: refuse a library of another major version (the five calls used here have kept
  // their signatures inside NCCL 2.x), and say which file was bound when the process maps torch but no RCCL was found mapped
  // (torch linking RCCL statically / under another soname would make this a second copy: SSG_RCCL_PATH overrides)
  g_rccl.GetVersion = reinterpret_cast<decltype(g_rccl.GetVersion)>(dlsym(h, "ncclGetVersion"));
  int code = 0;
  if (!g_rccl.GetVersion || g_rccl.GetVersion(&code) != ncclSuccess || rccl_major(code) != rccl_major(NCCL_VERSION_CODE)) {
    ssg_set_error("RCCL at %s reports version code %d, this library was built against %d (another major version)", loaded.c_str(), code, (int)NCCL_VERSION_CODE);
    return SSG_ERR_HIP;
  }
  if (getenv("SSG_COMM_VERBOSE")) fprintf(stderr, "[ssg_comm] bound to %s (version code %d, headers %d)\n", loaded.c_str(), code, (int)NCCL_VERSION_CODE);
  g_rccl.path = loaded; g_rccl.version = code; g_rccl.ok = true;
  return SSG_OK;
}
}  // namespace

static int ssg_check_nccl(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return SSG_OK;
  ssg_set_error("RCCL error %d (%s) at %s", (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", what);
  return SSG_ERR_HIP;
}
#define SSG_NCCL(call) do { int rc_ = ssg_check_nccl((call), #call); if (rc_) return rc_; } while (0)
#define SSG_RCCL_BIND() do { int rc_ = rccl_bind(); if (rc_) return rc_; } while (0)

// path of the RCCL library the collectives are bound to ("" before the first collective call / when none could be bound)
extern "C" const char* ssg_comm_library(void) { rccl_bind(); return g_rccl.path.c_str(); }

static_assert(sizeof(ncclUniqueId) == 128, "ssg_comm_unique_id hands out 128 bytes");

// 128 bytes that rank 0 creates and every rank passes to ssg_comm_init (ship them over any side channel: a file, a socket, MPI,
// torch.distributed's store)
extern "C" int ssg_comm_unique_id(void* id128_host) {
  if (!id128_host) { ssg_set_error("ssg_comm_unique_id: null buffer"); return SSG_ERR_INVALID; }
  ncclUniqueId id;
  SSG_RCCL_BIND();
  SSG_NCCL(g_rccl.GetUniqueId(&id));
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
  SSG_RCCL_BIND();
  SSG_NCCL(g_rccl.CommInitRank(&c, world, id, rank));
  *comm = (void*)c;
  return SSG_OK;
}

// recv[r * bytes_per_rank ...] = rank r's send block, for every r (in place when send == recv + rank * bytes_per_rank)
extern "C" int ssg_allgather(void* comm, const void* send, void* recv, size_t bytes_per_rank, hipStream_t stream) {
  if (!comm || (bytes_per_rank && (!send || !recv))) { ssg_set_error("ssg_allgather: null communicator / buffer"); return SSG_ERR_INVALID; }
  if (bytes_per_rank == 0) return SSG_OK;
  SSG_RCCL_BIND();
  SSG_NCCL(g_rccl.AllGather(send, recv, bytes_per_rank, ncclInt8, (ncclComm_t)comm, stream));
  return SSG_OK;
}

// in-place element-wise sum of int64 vectors over the ranks (eps histograms, candidate counts)
extern "C" int ssg_allreduce_sum_i64(void* comm, int64_t* buf, size_t count, hipStream_t stream) {
  if (!comm || (count && !buf)) { ssg_set_error("ssg_allreduce_sum_i64: null communicator / buffer"); return SSG_ERR_INVALID; }
  if (count == 0) return SSG_OK;
  SSG_RCCL_BIND();
  SSG_NCCL(g_rccl.AllReduce(buf, buf, count, ncclInt64, ncclSum, (ncclComm_t)comm, stream));
  return SSG_OK;
}

extern "C" int ssg_comm_destroy(void* comm) {
  if (!comm) return SSG_OK;
  SSG_RCCL_BIND();
  SSG_NCCL(g_rccl.CommDestroy((ncclComm_t)comm));
  return SSG_OK;
}
