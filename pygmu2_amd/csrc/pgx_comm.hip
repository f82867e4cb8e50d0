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
//
// Ranks out of step (round 4).  Every rank must issue the same sequence of collectives with the same element counts:
// the callers above decide "one collective per window or per block" from facts every rank shares, but a rank whose
// caller pulled differently would hang the communicator or corrupt a sum, silently.  So the first kCheckFirst tickets
// and every kCheckEvery-th after them are preceded by a 32-byte all-reduce(max) of {n, -n, h, -h} -- n the element
// count, h a running hash of every count and every word the caller folded in (pgx_comm_fold_check: window or block,
// rows, switches) since the communicator was made.  A fixed-size collective cannot mismatch; max(n) == -max(-n) on
// every rank exactly when all ranks agree.  The issue thread waits for it (with a deadline) before handing the payload
// to RCCL; on disagreement or timeout the error is sticky and every later call on the communicator fails with it.
//
// Ending.  pgx_comm_quiesce(timeout) waits, with a deadline, until everything handed to the communicator has
// completed; pgx_comm_destroy gives up after PGX_COMM_EXIT_TIMEOUT_MS (a peer that died mid-collective never
// completes ours): the communicator is then abandoned -- nothing of it is joined, synchronised or destroyed -- and the
// call reports it, so that the process can exit non-zero instead of hanging.  The singleton lives on the heap and is
// never destroyed: an issue thread left behind never touches destroyed members.

#include <dlfcn.h>

#include <chrono>
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
    bool check;                  // preceded by the cross-rank agreement collective
    int64_t hash;                // the sequence hash including this job
};

constexpr int64_t kHashMask = (int64_t(1) << 62) - 1;
inline int64_t fold(int64_t h, int64_t word) {       // splitmix-style: every earlier word moves every later hash
    uint64_t x = (uint64_t)h ^ ((uint64_t)word + 0x9e3779b97f4a7c15ull + ((uint64_t)h << 6) + ((uint64_t)h >> 2));
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return (int64_t)(x & (uint64_t)kHashMask);
}

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
    // ranks-out-of-step check
    int64_t seq_hash = 0;        // render thread: folded at every pgx_allreduce_sum / pgx_comm_fold_check
    int check_first = 8, check_every = 16, check_timeout_ms = 60000;
    int64_t fault_at = 0;        // test hook: this ticket's check contributes a disagreeing count
    int64_t *chk_dev = nullptr, *chk_host = nullptr;   // 4 device words; 8 pinned words (in, out)
    hipEvent_t chk_ev = nullptr;
    int64_t checks_done = 0;
    bool abandoned = false;      // a destroy that ran into its deadline: nothing of the communicator is touched again

    void stop_worker(bool detach = false) {
        if (!worker.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            if (detach) jobs.clear();
        }
        cv_work.notify_all();
        if (detach) worker.detach();
        else worker.join();
        stop = false;
    }
};

// On the heap and never destroyed: static teardown must not run under an issue thread that was left behind inside a
// collective whose peers are gone (it would lock a destroyed mutex); a process that ends without pgx_comm_destroy
// simply ends.
Comm &cm() {
    static Comm *c = new Comm;
    return *c;
}

bool wait_event_for(hipEvent_t ev, int timeout_ms) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return true;
        if (e != hipErrorNotReady) return false;
        if (std::chrono::steady_clock::now() >= deadline) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

bool wait_stream_for(hipStream_t st, int timeout_ms) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return true;
        if (e != hipErrorNotReady) return false;
        if (std::chrono::steady_clock::now() >= deadline) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
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
    hipError_t e;
    if (job.check) {
        // do all ranks issue this collective with this count, after the same sequence?  (in front of the wait for
        // the partial mix: only the collectives before it are in its way)
        int64_t *in = c.chk_host, *out = c.chk_host + 4;
        const int64_t n = (int64_t)job.n;
        in[0] = n; in[1] = job.ticket == c.fault_at ? -(n + 1) : -n; in[2] = job.hash; in[3] = -job.hash;
        e = hipMemcpyAsync(c.chk_dev, in, 4 * sizeof(int64_t), hipMemcpyHostToDevice, c.cstream);
        if (e != hipSuccess) return std::string("hipMemcpyAsync: ") + hipGetErrorString(e);
        const ncclResult_t r = c.api.AllReduce(c.chk_dev, c.chk_dev, 4, ncclInt64, ncclMax, c.comm, c.cstream);
        if (r != ncclSuccess) return std::string("ncclAllReduce (agreement check): ") + c.api.GetErrorString(r);
        e = hipMemcpyAsync(out, c.chk_dev, 4 * sizeof(int64_t), hipMemcpyDeviceToHost, c.cstream);
        if (e != hipSuccess) return std::string("hipMemcpyAsync: ") + hipGetErrorString(e);
        e = hipEventRecord(c.chk_ev, c.cstream);
        if (e != hipSuccess) return std::string("hipEventRecord: ") + hipGetErrorString(e);
        if (!wait_event_for(c.chk_ev, c.check_timeout_ms))
            return "ranks out of step? the agreement check in front of collective " + std::to_string(job.ticket) +
                   " did not complete within " + std::to_string(c.check_timeout_ms) + " ms";
        ++c.checks_done;
        if (out[0] != -out[1] || out[2] != -out[3])
            return "ranks out of step at collective " + std::to_string(job.ticket) + ": this rank reduces " +
                   std::to_string(n) + " floats (sequence hash " + std::to_string(job.hash) + "), the ranks' counts span " +
                   std::to_string(-out[1]) + " .. " + std::to_string(out[0]) +
                   (out[0] == -out[1] ? " (same count, different history of pulls)" : "") +
                   " -- every rank must make the same sequence of pulls";
    }
    e = hipStreamWaitEvent(c.cstream, c.ev_in[slot], 0);
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
    int64_t *chk_dev = nullptr, *chk_host = nullptr;
    hipEvent_t chk_ev = nullptr;
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
    ok = ok && step(hipMalloc(&chk_dev, 4 * sizeof(int64_t)), "hipMalloc");
    ok = ok && step(hipHostMalloc(&chk_host, 8 * sizeof(int64_t), hipHostMallocDefault), "hipHostMalloc");
    ok = ok && step(hipEventCreateWithFlags(&chk_ev, hipEventDisableTiming), "hipEventCreateWithFlags");
    if (!ok) {
        if (chk_ev) (void)hipEventDestroy(chk_ev);
        if (chk_host) (void)hipHostFree(chk_host);
        if (chk_dev) (void)hipFree(chk_dev);
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
    c.chk_dev = chk_dev;
    c.chk_host = chk_host;
    c.chk_ev = chk_ev;
    c.seq_hash = 0;
    c.checks_done = 0;
    c.abandoned = false;
    auto env_int = [](const char *name, int dflt) { return getenv(name) ? atoi(getenv(name)) : dflt; };
    c.check_first = env_int("PGX_COMM_CHECK_FIRST", 8);
    c.check_every = env_int("PGX_COMM_CHECK_EVERY", 16);
    c.check_timeout_ms = env_int("PGX_COMM_CHECK_TIMEOUT_MS", 60000);
    c.fault_at = getenv("PGX_COMM_CHECK_FAULT_AT") ? atoll(getenv("PGX_COMM_CHECK_FAULT_AT")) : 0;
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

// Everything handed to the communicator so far has completed (or `timeout_ms` have passed: PGX_ERR_RUNTIME).  No
// communicator: PGX_OK.  The atexit hook of the Python layer asks this before it synchronises anything: a rank whose
// peer died mid-collective then exits non-zero instead of hanging in hipStreamSynchronize.
int pgx_comm_quiesce(int timeout_ms) {
    Comm &c = cm();
    if (!c.comm) return c.abandoned ? pgx::fail(PGX_ERR_RUNTIME, "pgx_comm: the communicator was abandoned") : PGX_OK;
    if (timeout_ms < 0) timeout_ms = 0;
    const auto t0 = std::chrono::steady_clock::now();
    if (c.threaded) {
        std::unique_lock<std::mutex> lk(c.mu);
        if (!c.cv_done.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return c.recorded >= c.issued; }))
            return pgx::fail(PGX_ERR_RUNTIME, "pgx_comm_quiesce: " + std::to_string(c.issued - c.recorded) +
                                                  " collective(s) not handed to RCCL after " +
                                                  std::to_string(timeout_ms) + " ms");
    }
    const int spent = (int)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
    if (!wait_stream_for(c.cstream, timeout_ms > spent ? timeout_ms - spent : 0))
        return pgx::fail(PGX_ERR_RUNTIME, "pgx_comm_quiesce: collectives still running after " +
                                              std::to_string(timeout_ms) + " ms (a peer gone, or ranks out of step?)");
    return PGX_OK;
}

int pgx_comm_destroy(void) {
    Comm &c = cm();
    if (!c.comm) return PGX_OK;
    static const int timeout_ms = getenv("PGX_COMM_EXIT_TIMEOUT_MS") ? atoi(getenv("PGX_COMM_EXIT_TIMEOUT_MS")) : 10000;
    const bool failed = !c.worker_error.empty();
    if (failed || pgx_comm_quiesce(timeout_ms) != PGX_OK) {
        // outstanding work that will never complete (or an issue thread that reported a failure: the collective
        // stream may be stuck behind it).  Joining, synchronising or ncclCommDestroy would hang with it: leave
        // everything where it is and say so.
        const std::string why = failed ? "issue thread failed: " + c.worker_error : std::string(pgx_last_error());
        c.stop_worker(true);
        c.comm = nullptr;
        c.world = 0;
        c.abandoned = true;
        return pgx::fail(PGX_ERR_RUNTIME, "pgx_comm_destroy: communicator abandoned (" + why + ")");
    }
    c.stop_worker();
    (void)c.api.CommDestroy(c.comm);
    c.comm = nullptr;
    (void)hipStreamDestroy(c.cstream);
    for (auto &e : c.ev_in) (void)hipEventDestroy(e);
    for (auto &e : c.done) (void)hipEventDestroy(e);
    (void)hipEventDestroy(c.chk_ev);
    (void)hipHostFree(c.chk_host);
    (void)hipFree(c.chk_dev);
    (void)hipFree(c.scratch);
    c.cstream = nullptr;
    c.scratch = nullptr;
    c.chk_dev = c.chk_host = nullptr;
    c.chk_ev = nullptr;
    c.world = 0;
    return PGX_OK;
}

// 1 after a pgx_comm_destroy that ran into its deadline (pgx_shutdown then leaves the HIP objects alone too).
int pgx_comm_abandoned(void) { return cm().abandoned ? 1 : 0; }

// A fact this rank's NEXT collectives depend on and every rank must share (window or block, rows, switches): folded
// into the sequence hash the agreement checks compare.
int pgx_comm_fold_check(int64_t word) {
    Comm &c = cm();
    c.seq_hash = fold(c.seq_hash, word);
    return PGX_OK;
}

// tickets issued, agreement checks completed, the current sequence hash (tests, bench.py)
int pgx_comm_stats(int64_t *issued, int64_t *checks, int64_t *hash) {
    Comm &c = cm();
    if (issued) *issued = c.comm ? c.issued : 0;
    if (checks) {
        std::lock_guard<std::mutex> lk(c.mu);
        *checks = c.checks_done;
    }
    if (hash) *hash = c.seq_hash;
    return PGX_OK;
}

int pgx_allreduce_sum(float *out, const float *in, size_t n, int64_t *ticket) {
    PGX_REQUIRE_INIT();
    Comm &c = cm();
    PGX_CHECK_ARG(c.comm != nullptr, "pgx_allreduce_sum: pgx_comm_init has not been called");
    PGX_CHECK_ARG(out != nullptr && in != nullptr && ticket != nullptr, "pgx_allreduce_sum: null argument");
    const int64_t t = c.issued + 1;
    c.seq_hash = fold(c.seq_hash, (int64_t)n);
    const bool check = (t <= c.check_first) || (c.check_every > 0 && t % c.check_every == 0);
    const Job job{in, out, n, t, check, c.seq_hash};
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
