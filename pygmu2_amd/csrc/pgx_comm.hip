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
//
// The collective is ISSUED by a thread of the library's own (PGX_COMM_THREAD=0: by the caller): ncclAllReduce costs
// 20-30 us of host time per call, as much as a rank's render thread spends enqueueing a whole block of its 64 C5 voices
// -- with a collective per block the render thread became the bottleneck (53 -> 80 us per block).  pgx_allreduce_sum
// records the "partial mix complete" event on the library stream, queues the job and returns; the issue thread orders
// the collective stream behind that event, calls RCCL and records the ticket's completion event; pgx_allreduce_wait
// first makes sure that event has been recorded (normally long ago), then orders the library stream behind it.  Jobs
// are issued in ticket order, which is the call order -- the same on every rank.  The scalar reductions and
// pgx_comm_destroy drain the queue first.

#include <dlfcn.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <rccl/rccl.h>
#include <thread>

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

struct Job {
    const float *in;
    float *out;
    size_t n;
    int64_t ticket;
};

struct Comm {
    Api api;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 0;
    hipStream_t cstream = nullptr;
    hipEvent_t ev_in[kRing] = {};    // ticket t: the library stream where the partial mix was complete
    hipEvent_t done[kRing] = {};
    int64_t issued = 0;          // tickets are 1-based; ticket t's events are ev_in / done[t % kRing]
    double *scratch = nullptr;   // one device double for the scalar reductions
    // the issue thread
    bool threaded = false;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<Job> jobs;
    int64_t recorded = 0;        // tickets up to here have their completion event recorded
    bool stop = false;
    std::string worker_error;    // first failure of the issue thread (sticky)

    // (a process that exits without pgx_comm_destroy: an idle thread just ends; one that is inside a collective whose
    // peers are gone would never come back -- it is left behind instead of joined, and ends with the process)
    ~Comm() { stop_worker(true); }
    void stop_worker(bool at_exit = false) {
        if (!worker.joinable()) return;
        bool busy;
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            busy = recorded < issued;
        }
        cv_work.notify_all();
        if (at_exit && busy) worker.detach();
        else worker.join();
        stop = false;
    }
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

// what issuing ticket `job.ticket` means, on whichever thread does it
std::string issue(Comm &c, const Job &job) {
    const int slot = (int)(job.ticket % kRing);
    hipError_t e = hipStreamWaitEvent(c.cstream, c.ev_in[slot], 0);
    if (e != hipSuccess) return std::string("hipStreamWaitEvent: ") + hipGetErrorString(e);
    if (job.n) {
        const ncclResult_t r = c.api.AllReduce(job.in, job.out, job.n, ncclFloat32, ncclSum, c.comm, c.cstream);
        if (r != ncclSuccess) return std::string("ncclAllReduce: ") + c.api.GetErrorString(r);
    }
    e = hipEventRecord(c.done[slot], c.cstream);
    if (e != hipSuccess) return std::string("hipEventRecord: ") + hipGetErrorString(e);
    return std::string();
}

void worker_main(Comm *cp, int device) {
    Comm &c = *cp;
    (void)hipSetDevice(device);
    std::unique_lock<std::mutex> lk(c.mu);
    for (;;) {
        c.cv_work.wait(lk, [&] { return c.stop || !c.jobs.empty(); });
        if (c.jobs.empty()) return;                   // stop, nothing left to issue
        const Job job = c.jobs.front();
        c.jobs.pop_front();
        lk.unlock();
        std::string err = issue(c, job);
        lk.lock();
        if (!err.empty() && c.worker_error.empty()) c.worker_error = err;
        c.recorded = job.ticket;                      // (recorded or failed: nobody waits for it for ever)
        c.cv_done.notify_all();
    }
}

// every queued collective has been handed to RCCL (the caller may use the communicator itself)
int drain(Comm &c) {
    if (!c.threaded) return PGX_OK;
    std::unique_lock<std::mutex> lk(c.mu);
    c.cv_done.wait(lk, [&] { return c.recorded >= c.issued; });
    if (!c.worker_error.empty()) return pgx::fail(PGX_ERR_RUNTIME, "pgx_comm issue thread: " + c.worker_error);
    return PGX_OK;
}

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
    hipEvent_t ev_in[kRing] = {};
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
    // Normal priority.  The library's side stream (envelope walks a block ahead) is the high-priority queue of the
    // process; a second one for the collectives made the two take turns: with a 1-rank communicator a rank's 64 C5
    // voices went from 53 us per block to 104-111 (edge search 5 -> 45 us, walk 34 -> 57 us in the kernel trace), at
    // normal priority 55.6.  PGX_COMM_PRIORITY overrides (-1 high, 1 low: 68-173 us).
    (void)least;
    (void)greatest;
    int prio = 0;
    if (getenv("PGX_COMM_PRIORITY")) prio = atoi(getenv("PGX_COMM_PRIORITY"));
    ok = ok && step(hipStreamCreateWithPriority(&cstream, hipStreamNonBlocking, prio), "hipStreamCreateWithPriority");
    for (int i = 0; ok && i < kRing; ++i) ok = step(hipEventCreateWithFlags(&ev_in[i], hipEventDisableTiming), "hipEventCreateWithFlags");
    for (int i = 0; ok && i < kRing; ++i) ok = step(hipEventCreateWithFlags(&done[i], hipEventDisableTiming), "hipEventCreateWithFlags");
    ok = ok && step(hipMalloc(&scratch, sizeof(double)), "hipMalloc");
    if (!ok) {
        if (scratch) (void)hipFree(scratch);
        for (auto &e : done)
            if (e) (void)hipEventDestroy(e);
        for (auto &e : ev_in)
            if (e) (void)hipEventDestroy(e);
        if (cstream) (void)hipStreamDestroy(cstream);
        if (comm) (void)c.api.CommDestroy(comm);
        return pgx::fail(PGX_ERR_RUNTIME, "pgx_comm_init: " + why);
    }
    c.cstream = cstream;
    for (int i = 0; i < kRing; ++i) c.ev_in[i] = ev_in[i];
    for (int i = 0; i < kRing; ++i) c.done[i] = done[i];
    c.scratch = scratch;
    c.rank = rank;
    c.world = world;
    c.issued = 0;
    c.recorded = 0;
    c.jobs.clear();
    c.worker_error.clear();
    c.comm = comm;                                    // last: its presence is what "initialised" means
    c.threaded = !(getenv("PGX_COMM_THREAD") && atoi(getenv("PGX_COMM_THREAD")) == 0);
    if (c.threaded) c.worker = std::thread(worker_main, &c, pgx::device_index());
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
    (void)drain(c);
    c.stop_worker();
    (void)hipStreamSynchronize(c.cstream);
    (void)c.api.CommDestroy(c.comm);
    c.comm = nullptr;
    (void)hipStreamDestroy(c.cstream);
    for (auto &e : c.ev_in) (void)hipEventDestroy(e);
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
    const int64_t t = c.issued + 1;
    const Job job{in, out, n, t};
    if (!c.threaded) {
        PGX_HIP(hipEventRecord(c.ev_in[t % kRing], pgx::stream()));      // the local partial mix is complete here
        const std::string err = issue(c, job);
        if (!err.empty()) return pgx::fail(PGX_ERR_RUNTIME, "pgx_allreduce_sum: " + err);
        c.issued = t;
        *ticket = t;
        return PGX_OK;
    }
    {
        // an event slot is used again kRing tickets later: never while its previous job is still in the queue
        std::unique_lock<std::mutex> lk(c.mu);
        c.cv_done.wait(lk, [&] { return t - c.recorded < kRing / 2; });
        if (!c.worker_error.empty()) return pgx::fail(PGX_ERR_RUNTIME, "pgx_comm issue thread: " + c.worker_error);
    }
    PGX_HIP(hipEventRecord(c.ev_in[t % kRing], pgx::stream()));          // the local partial mix is complete here
    {
        std::lock_guard<std::mutex> lk(c.mu);
        c.jobs.push_back(job);
        c.issued = t;
    }
    c.cv_work.notify_one();
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
    if (c.threaded) {
        std::unique_lock<std::mutex> lk(c.mu);
        c.cv_done.wait(lk, [&] { return c.recorded >= t; });              // (its completion event exists)
        if (!c.worker_error.empty()) return pgx::fail(PGX_ERR_RUNTIME, "pgx_comm issue thread: " + c.worker_error);
    }
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
    if (int rc = drain(c)) return rc;                                    // the communicator is ours from here on
    PGX_HIP(hipMemcpyAsync(c.scratch, value_host, sizeof(double), hipMemcpyHostToDevice, c.cstream));
    PGX_NCCL(c.api.AllReduce(c.scratch, c.scratch, 1, ncclFloat64, op ? ncclMax : ncclSum, c.comm, c.cstream));
    PGX_HIP(hipMemcpyAsync(value_host, c.scratch, sizeof(double), hipMemcpyDeviceToHost, c.cstream));
    PGX_HIP(hipStreamSynchronize(c.cstream));
    return PGX_OK;
}

}  // extern "C"
