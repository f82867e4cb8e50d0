// pgx_comm.hip -- the exchange step of a sharded MixPE: all-reduce(sum) of the ranks' partial mixes on
// RCCL over xGMI, behind the C ABI (reference: mix_pe.py:91-94 is the sum this distributes; one process
// per GPU, inputs i mod world on rank `rank`).
//
// librccl is resolved with dlopen at pgx_comm_init time: single-GPU users never load it, and the render
// library has no link-time dependency on it.
//
// Stream discipline (the same deferred wait the PE layer had with torch.distributed): the collective runs
// on its own stream, ordered behind whatever the library stream has enqueued so far; it returns a ticket.
// pgx_allreduce_wait(ticket) orders the library stream behind that collective -- a stream-level wait, the
// host never blocks -- so block k+1 renders while block k's partial mix is on the links.

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <rccl/rccl.h>

#include "pgx_common.h"

namespace {

constexpr int kRing = 64;   // tickets whose completion event is still individually addressable

struct Api {
    void *dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

struct Comm {
    Api api;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 0;
    hipStream_t cstream = nullptr;
    hipEvent_t ev_in = nullptr;
    hipEvent_t done[kRing] = {};
    int64_t issued = 0;          // tickets are 1-based; ticket t's event is done[t % kRing]
    double *scratch = nullptr;   // one device double for the scalar reductions
};

Comm &cm() {
    static Comm c;
    return c;
}

int load_api(Api &a) {
    if (a.dl) return PGX_OK;
    const char *names[] = {getenv("PGX_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string tried;
    for (const char *n : names) {
        if (!n || !*n) continue;
        a.dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (a.dl) break;
        tried += std::string(n) + ": " + dlerror() + "; ";
    }
    if (!a.dl) return pgx::fail(PGX_ERR_RUNTIME, "pgx_comm: cannot load librccl (" + tried + ")");
#define PGX_SYM(field, name)                                                                  \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.dl, name));                         \
    if (!a.field) return pgx::fail(PGX_ERR_RUNTIME, std::string("pgx_comm: librccl lacks ") + name)
    PGX_SYM(GetUniqueId, "ncclGetUniqueId");
    PGX_SYM(CommInitRank, "ncclCommInitRank");
    PGX_SYM(CommDestroy, "ncclCommDestroy");
    PGX_SYM(AllReduce, "ncclAllReduce");
    PGX_SYM(GetErrorString, "ncclGetErrorString");
#undef PGX_SYM
    return PGX_OK;
}

#define PGX_NCCL(call)                                                                            \
    do {                                                                                          \
        ncclResult_t _r = (call);                                                                 \
        if (_r != ncclSuccess)                                                                    \
            return pgx::fail(PGX_ERR_RUNTIME, std::string(#call) + ": " + cm().api.GetErrorString(_r)); \
    } while (0)

}  // namespace

extern "C" {

size_t pgx_comm_unique_id_bytes(void) { return sizeof(ncclUniqueId); }

int pgx_comm_unique_id(void *id_host, size_t len) {
    PGX_CHECK_ARG(id_host != nullptr && len >= sizeof(ncclUniqueId), "pgx_comm_unique_id: buffer too small");
    Comm &c = cm();
    if (int rc = load_api(c.api)) return rc;
    ncclUniqueId id;
    PGX_NCCL(c.api.GetUniqueId(&id));
    memcpy(id_host, &id, sizeof(id));
    return PGX_OK;
}

int pgx_comm_init(int rank, int world, const void *id_host, size_t len) {
    PGX_REQUIRE_INIT();
    Comm &c = cm();
    PGX_CHECK_ARG(c.comm == nullptr, "pgx_comm_init: communicator already initialised");
    PGX_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "pgx_comm_init: bad rank / world");
    PGX_CHECK_ARG(id_host != nullptr && len >= sizeof(ncclUniqueId), "pgx_comm_init: bad unique id");
    if (int rc = load_api(c.api)) return rc;
    PGX_HIP(hipSetDevice(pgx::device_index()));
    ncclUniqueId id;
    memcpy(&id, id_host, sizeof(id));
    // Everything is built into locals and published only when every step succeeded: a failure half-way leaves the
    // singleton exactly as it was (no communicator, a retry is welcome) and frees what had been made.
    ncclComm_t comm = nullptr;
    hipStream_t cstream = nullptr;
    hipEvent_t ev_in = nullptr;
    hipEvent_t done[kRing] = {};
    double *scratch = nullptr;
    std::string why;
    auto step = [&](hipError_t e, const char *what) {
        if (e == hipSuccess) return true;
        why = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    };
    bool ok = true;
    {
        const ncclResult_t r = c.api.CommInitRank(&comm, world, id, rank);
        if (r != ncclSuccess) {
            why = std::string("c.api.CommInitRank(&c.comm, world, id, rank): ") + c.api.GetErrorString(r);
            comm = nullptr;
            ok = false;
        }
    }
    int least = 0, greatest = 0;
    ok = ok && step(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
    ok = ok && step(hipStreamCreateWithPriority(&cstream, hipStreamNonBlocking, greatest), "hipStreamCreateWithPriority");
    ok = ok && step(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming), "hipEventCreateWithFlags");
    for (int i = 0; ok && i < kRing; ++i) ok = step(hipEventCreateWithFlags(&done[i], hipEventDisableTiming), "hipEventCreateWithFlags");
    ok = ok && step(hipMalloc(&scratch, sizeof(double)), "hipMalloc");
    if (!ok) {
        if (scratch) (void)hipFree(scratch);
        for (auto &e : done)
            if (e) (void)hipEventDestroy(e);
        if (ev_in) (void)hipEventDestroy(ev_in);
        if (cstream) (void)hipStreamDestroy(cstream);
        if (comm) (void)c.api.CommDestroy(comm);
        return pgx::fail(PGX_ERR_RUNTIME, "pgx_comm_init: " + why);
    }
    c.cstream = cstream;
    c.ev_in = ev_in;
    for (int i = 0; i < kRing; ++i) c.done[i] = done[i];
    c.scratch = scratch;
    c.rank = rank;
    c.world = world;
    c.issued = 0;
    c.comm = comm;                                    // last: its presence is what "initialised" means
    return PGX_OK;
}

int pgx_comm_info(int *rank, int *world) {
    Comm &c = cm();
    if (rank) *rank = c.comm ? c.rank : 0;
    if (world) *world = c.comm ? c.world : 0;      // 0 ranks: no communicator
    return PGX_OK;
}

int pgx_comm_destroy(void) {
    Comm &c = cm();
    if (!c.comm) return PGX_OK;
    (void)hipStreamSynchronize(c.cstream);
    (void)c.api.CommDestroy(c.comm);
    c.comm = nullptr;
    (void)hipStreamDestroy(c.cstream);
    (void)hipEventDestroy(c.ev_in);
    for (auto &e : c.done) (void)hipEventDestroy(e);
    (void)hipFree(c.scratch);
    c.cstream = nullptr;
    c.scratch = nullptr;
    c.world = 0;
    return PGX_OK;
}

int pgx_allreduce_sum(float *out, const float *in, size_t n, int64_t *ticket) {
    PGX_REQUIRE_INIT();
    Comm &c = cm();
    PGX_CHECK_ARG(c.comm != nullptr, "pgx_allreduce_sum: pgx_comm_init has not been called");
    PGX_CHECK_ARG(out != nullptr && in != nullptr && ticket != nullptr, "pgx_allreduce_sum: null argument");
    PGX_HIP(hipEventRecord(c.ev_in, pgx::stream()));             // the local partial mix is complete here
    PGX_HIP(hipStreamWaitEvent(c.cstream, c.ev_in, 0));
    if (n) PGX_NCCL(c.api.AllReduce(in, out, n, ncclFloat32, ncclSum, c.comm, c.cstream));
    const int64_t t = ++c.issued;
    PGX_HIP(hipEventRecord(c.done[t % kRing], c.cstream));
    *ticket = t;
    return PGX_OK;
}

int pgx_allreduce_wait(int64_t ticket) {
    PGX_REQUIRE_INIT();
    Comm &c = cm();
    PGX_CHECK_ARG(c.comm != nullptr, "pgx_allreduce_wait: no communicator");
    PGX_CHECK_ARG(ticket >= 1 && ticket <= c.issued, "pgx_allreduce_wait: unknown ticket");
    // an event slot recycled since then marks a LATER point of the same in-order stream: still sufficient
    const int64_t t = ticket > c.issued - kRing ? ticket : c.issued - kRing + 1;
    PGX_HIP(hipStreamWaitEvent(pgx::stream(), c.done[t % kRing], 0));
    return PGX_OK;
}

// Scalar reductions over the ranks for host-side bookkeeping (bench.py: slowest rank's time, ranks seen).
// Synchronous.  op: 0 = sum, 1 = max.
int pgx_allreduce_scalar_host(double *value_host, int op) {
    PGX_REQUIRE_INIT();
    Comm &c = cm();
    PGX_CHECK_ARG(c.comm != nullptr, "pgx_allreduce_scalar_host: no communicator");
    PGX_CHECK_ARG(value_host != nullptr && (op == 0 || op == 1), "pgx_allreduce_scalar_host: bad argument");
    PGX_HIP(hipMemcpyAsync(c.scratch, value_host, sizeof(double), hipMemcpyHostToDevice, c.cstream));
    PGX_NCCL(c.api.AllReduce(c.scratch, c.scratch, 1, ncclFloat64, op ? ncclMax : ncclSum, c.comm, c.cstream));
    PGX_HIP(hipMemcpyAsync(value_host, c.scratch, sizeof(double), hipMemcpyDeviceToHost, c.cstream));
    PGX_HIP(hipStreamSynchronize(c.cstream));
    return PGX_OK;
}

}  // extern "C"
