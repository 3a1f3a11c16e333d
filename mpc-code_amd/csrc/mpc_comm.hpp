// Multi-GPU collective of the per-model libraries (libmpc_enmpc_<model>.so): one process per GPU, RCCL over xGMI, as
// libmpc_amd.so has it for the linear path (mpc_amd.hip: mpc_comm_*).  Instances are independent, so the only exchanges are the all-gather of the
// controls - straight from the device log, device to device - and the job-level barrier / reductions of a harness.  librccl is opened on first
// use: a single-GPU user of the library never loads it.  The including .hip defines fail(code, fmt, ...) and HIP_TRY.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>

namespace mpc_comm {

constexpr int kIdBytes = 128;

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;

static int rccl_load()
{
    if (g_rccl.lib) return 0;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
    if (!lib) return fail(-12, "librccl.so not found: %s", dlerror());
#define MPC_RCCL_SYM(field, sym) *(void **)(&g_rccl.field) = dlsym(lib, sym); if (!g_rccl.field) { dlclose(lib); return fail(-12, "librccl lacks %s", sym); }
    MPC_RCCL_SYM(GetUniqueId, "ncclGetUniqueId") MPC_RCCL_SYM(CommInitRank, "ncclCommInitRank") MPC_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    MPC_RCCL_SYM(AllGather, "ncclAllGather") MPC_RCCL_SYM(AllReduce, "ncclAllReduce") MPC_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef MPC_RCCL_SYM
    g_rccl.lib = lib;
    return 0;
}
#define MPC_RCCL_TRY(x)                                                                                  \
    do {                                                                                                 \
        ncclResult_t r_ = (x);                                                                           \
        if (r_ != ncclSuccess) return fail(-12, "%s failed: %s (%s:%d)", #x, mpc_comm::g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// RCCL writes its version banner to stdout when it initialises (NCCL_DEBUG=VERSION, as on the benchmark boxes): while one of its set-up
// calls runs, file descriptor 1 points at stderr, so that a caller's stdout carries only what the caller prints (bench.py: one JSON line).
struct StdoutToStderr {
    int saved = -1;
    StdoutToStderr() { fflush(stdout); saved = dup(1); if (saved >= 0) (void)dup2(2, 1); }
    ~StdoutToStderr() { fflush(stdout); if (saved >= 0) { (void)dup2(saved, 1); close(saved); } }
};

struct Buf {
    void *p = nullptr; size_t bytes = 0;
    int ensure(size_t n) { if (n <= bytes) return 0; if (p) (void)hipFree(p); p = nullptr; bytes = 0; HIP_TRY(hipMalloc(&p, n)); bytes = n; return 0; }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

// the communicator of one handle: the ranks of the job, this rank's staging buffers
struct State {
    ncclComm_t comm = nullptr; int rank = 0, world = 1;
    Buf send, recv;
    size_t agreed_n = 0;      // the per-rank count of allgather_dev every rank was last seen to agree on
};

static int unique_id(char *out128)
{
    if (!out128) return fail(-1, "null argument");
    if (rccl_load()) return -12;
    ncclUniqueId id;
    StdoutToStderr quiet;
    MPC_RCCL_TRY(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == kIdBytes, "ncclUniqueId size");
    std::memcpy(out128, &id, sizeof(id));
    return 0;
}
static int init(State &c, int device, int rank, int world, const char *id128)
{
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(-1, "bad argument");
    if (c.comm) return fail(-1, "the handle already has a communicator");
    if (rccl_load()) return -12;
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    {
        StdoutToStderr quiet;
        MPC_RCCL_TRY(g_rccl.CommInitRank(&c.comm, world, id, rank));
    }
    c.rank = rank; c.world = world;
    return 0;
}
static void destroy(State &c, int device, hipStream_t stream)
{
    (void)hipSetDevice(device);
    if (c.comm) { if (stream) (void)hipStreamSynchronize(stream); g_rccl.CommDestroy(c.comm); }
    c.comm = nullptr; c.rank = 0; c.world = 1;
    c.send.release(); c.recv.release();
}
// all-gather of `bytes` bytes per rank between host buffers, staged through device memory (rank r's block lands at recv + r * bytes)
static int allgather_host(State &c, int device, hipStream_t stream, const void *send, size_t bytes, void *recv)
{
    if (!send || !recv || bytes == 0) return fail(-1, "bad argument");
    if (!c.comm) { std::memcpy(recv, send, bytes); return 0; }
    HIP_TRY(hipSetDevice(device));
    if (c.send.ensure(bytes) || c.recv.ensure(bytes * c.world)) return -10;
    HIP_TRY(hipMemcpyAsync(c.send.p, send, bytes, hipMemcpyHostToDevice, stream));
    MPC_RCCL_TRY(g_rccl.AllGather(c.send.p, c.recv.p, bytes, ncclChar, c.comm, stream));
    HIP_TRY(hipMemcpyAsync(recv, c.recv.p, bytes * c.world, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}
static int allreduce_max(State &c, int device, hipStream_t stream, double *inout, int n)
{
    if (!inout || n < 1) return fail(-1, "bad argument");
    if (!c.comm) return 0;
    HIP_TRY(hipSetDevice(device));
    if (c.send.ensure(sizeof(double) * n)) return -10;
    HIP_TRY(hipMemcpyAsync(c.send.p, inout, sizeof(double) * n, hipMemcpyHostToDevice, stream));
    MPC_RCCL_TRY(g_rccl.AllReduce(c.send.p, c.send.p, n, ncclDouble, ncclMax, c.comm, stream));
    HIP_TRY(hipMemcpyAsync(inout, c.send.p, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}
// everything queued on this handle's stream on every rank has completed when this returns
static int barrier(State &c, int device, hipStream_t stream)
{
    HIP_TRY(hipSetDevice(device));
    if (c.comm) { double one = 1.0; const int rc = allreduce_max(c, device, stream, &one, 1); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}
// n doubles per rank from device memory `src` into c.recv ([world][n], device), asynchronous on `stream`: the single RCCL all-gather of the controls
static int allgather_dev(State &c, int device, hipStream_t stream, const double *src, size_t n)
{
    HIP_TRY(hipSetDevice(device));
    if (c.comm && c.world > 1 && c.agreed_n != n) {
        // ncclAllGather wants the same count on every rank (unequal shards hang or corrupt the gather): checked once per count with a small gather of its own
        unsigned long long mine = n, all[64];
        if (c.world > 64) return fail(-1, "more than 64 ranks");
        if (int rc = allgather_host(c, device, stream, &mine, sizeof(mine), all)) return rc;
        for (int r = 0; r < c.world; r++)
            if (all[r] != mine) return fail(-1, "all-gather of a log: rank %d holds %llu doubles per rank, rank %d holds %llu - every rank must allocate the same batch and step count (pad the shards: shard.py)", c.rank, mine, r, all[r]);
        c.agreed_n = n;
    }
    if (c.recv.ensure(n * sizeof(double) * c.world)) return -10;
    if (c.comm) MPC_RCCL_TRY(g_rccl.AllGather(src, c.recv.p, n, ncclDouble, c.comm, stream));
    else HIP_TRY(hipMemcpyAsync(c.recv.p, src, n * sizeof(double), hipMemcpyDeviceToDevice, stream));
    return 0;
}

}  // namespace mpc_comm
