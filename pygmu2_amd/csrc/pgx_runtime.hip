// pgx_runtime.hip -- device selection, library stream, pooled device memory, copies, events.
//
// Host-side plumbing for the C ABI in include/pygmu_hip.h.  The pool keeps freed blocks in
// power-of-two size classes; because every kernel and copy is ordered on the one library
// stream, a block may be handed out again as soon as it is freed.
//
// Fork / join: two independent sub-graphs of one block (a voice bank's envelopes and its
// oscillator -> filter chain) may run concurrently.  pgx_stream_fork() starts a side stream behind
// everything enqueued so far; between fork and join the caller picks the stream each call goes
// to; pgx_stream_join() orders the main stream after the side stream.  While forked, freed blocks
// are parked instead of re-pooled, because "freed = reusable" only holds in single-stream order.

#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "pgx_common.h"

namespace {

thread_local std::string g_last_error;

struct Runtime {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;          // main stream
    hipStream_t side = nullptr;            // second stream, used between fork and join
    hipStream_t current = nullptr;         // where the next call is enqueued
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_detached = nullptr;
    bool forked = false;
    bool detached = false;                 // the side stream still runs what a fork gave it; nobody waits yet
    std::vector<void *> parked;            // blocks freed while forked (re-pooled at join / when the detached work is waited for)
    std::mutex mu;
    std::map<size_t, std::vector<void *>> free_lists;   // size class -> blocks
    std::unordered_map<void *, size_t> live;             // ptr -> size class
    size_t bytes_cached = 0;
    // pinned host blocks + the device->host copy stream (root Snippets leaving the device)
    std::map<size_t, std::vector<void *>> host_free;
    std::unordered_map<void *, size_t> host_live;
    hipStream_t copy = nullptr;
    hipEvent_t ev_copy_in = nullptr;
    hipEvent_t copy_done[64] = {};
    int64_t copies_issued = 0;
};

Runtime &rt() {
    static Runtime r;
    return r;
}

__global__ void __launch_bounds__(256) k_copy_words(uint32_t *dst, const uint32_t *src, size_t words) {
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < words && i < ((size_t)blockIdx.x + 1) * 1024; i += 256)
        dst[i] = src[i];
}

__global__ void __launch_bounds__(256) k_fill_words(uint32_t *dst, uint32_t value, size_t words) {
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < words && i < ((size_t)blockIdx.x + 1) * 1024; i += 256)
        dst[i] = value;
}

size_t size_class(size_t bytes) {
    size_t c = 256;
    while (c < bytes) c <<= 1;
    return c;
}

}  // namespace

namespace pgx {

void set_error(const std::string &msg) { g_last_error = msg; }

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

hipStream_t stream() { return rt().current; }
hipStream_t main_stream() { return rt().stream; }
bool initialised() { return rt().ready; }
int device_index() { return rt().device; }

}  // namespace pgx

extern "C" {

int pgx_abi_version(void) { return 1; }

const char *pgx_last_error(void) { return g_last_error.c_str(); }

int pgx_device_count(int *count) {
    PGX_CHECK_ARG(count != nullptr, "pgx_device_count: null output");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return pgx::fail(PGX_ERR_RUNTIME, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return PGX_OK;
}

int pgx_init(int device) {
    Runtime &r = rt();
    std::lock_guard<std::mutex> lock(r.mu);
    if (r.ready) {
        if (r.device != device)
            return pgx::fail(PGX_ERR_INVALID, "pgx_init: already initialised on another device");
        return PGX_OK;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return pgx::fail(PGX_ERR_NOT_INIT, "pgx_init: no HIP device available");
    PGX_CHECK_ARG(device >= 0 && device < n, "pgx_init: device index out of range");
    PGX_HIP(hipSetDevice(device));
    PGX_HIP(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
    {
        // the side stream carries the latency-bound branch of a fork: give it queue priority so that its
        // short throughput phases are not stretched by the main stream's VALU-bound kernels
        int least = 0, greatest = 0;
        PGX_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        PGX_HIP(hipStreamCreateWithPriority(&r.side, hipStreamNonBlocking, greatest));
    }
    PGX_HIP(hipEventCreateWithFlags(&r.ev_fork, hipEventDisableTiming));
    PGX_HIP(hipEventCreateWithFlags(&r.ev_join, hipEventDisableTiming));
    PGX_HIP(hipEventCreateWithFlags(&r.ev_detached, hipEventDisableTiming));
    r.detached = false;
    PGX_HIP(hipStreamCreateWithFlags(&r.copy, hipStreamNonBlocking));
    PGX_HIP(hipEventCreateWithFlags(&r.ev_copy_in, hipEventDisableTiming));
    for (auto &e : r.copy_done) PGX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    r.copies_issued = 0;
    r.current = r.stream;
    r.forked = false;
    r.device = device;
    r.ready = true;
    return PGX_OK;
}

int pgx_pool_trim(void) {
    Runtime &r = rt();
    std::lock_guard<std::mutex> lock(r.mu);
    if (!r.ready) return PGX_OK;
    (void)hipStreamSynchronize(r.stream);
    (void)hipStreamSynchronize(r.side);
    for (auto &kv : r.free_lists)
        for (void *p : kv.second) (void)hipFree(p);
    r.free_lists.clear();
    r.bytes_cached = 0;
    return PGX_OK;
}

int pgx_shutdown(void) {
    Runtime &r = rt();
    if (!r.ready) return PGX_OK;
    if (pgx_comm_destroy() != PGX_OK || pgx_comm_abandoned()) {
        // collectives that will never complete sit on the device (a peer died, ranks out of step): freeing, synchronising
        // or destroying anything could wait for them for ever.  The library is marked down and the process is expected
        // to end (the Python layer exits non-zero).
        r.ready = false;
        return PGX_ERR_RUNTIME;                          // (pgx_last_error() still holds pgx_comm_destroy's message)
    }
    pgx_pool_trim();
    std::lock_guard<std::mutex> lock(r.mu);
    for (auto &kv : r.live) (void)hipFree(kv.first);   // parked blocks are still listed in `live`
    r.live.clear();
    r.parked.clear();
    (void)hipStreamSynchronize(r.copy);
    for (auto &kv : r.host_free)
        for (void *p : kv.second) (void)hipHostFree(p);
    r.host_free.clear();
    for (auto &kv : r.host_live) (void)hipHostFree(kv.first);
    r.host_live.clear();
    (void)hipStreamDestroy(r.copy);
    (void)hipEventDestroy(r.ev_copy_in);
    for (auto &e : r.copy_done) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(r.stream);
    (void)hipStreamDestroy(r.side);
    (void)hipEventDestroy(r.ev_fork);
    (void)hipEventDestroy(r.ev_join);
    (void)hipEventDestroy(r.ev_detached);
    r.detached = false;
    r.stream = r.side = r.current = nullptr;
    r.forked = false;
    r.ready = false;
    r.device = -1;
    return PGX_OK;
}

int pgx_device_name(char *buf, size_t len) {
    PGX_REQUIRE_INIT();
    PGX_CHECK_ARG(buf != nullptr && len > 0, "pgx_device_name: bad buffer");
    hipDeviceProp_t prop;
    PGX_HIP(hipGetDeviceProperties(&prop, rt().device));
    snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return PGX_OK;
}

void *pgx_stream_handle(void) { return (void *)rt().stream; }

int pgx_stream_sync(void) {
    PGX_REQUIRE_INIT();
    PGX_HIP(hipStreamSynchronize(rt().stream));
    if (rt().forked || rt().detached) PGX_HIP(hipStreamSynchronize(rt().side));
    return PGX_OK;
}

static void release_parked(Runtime &r) {
    std::lock_guard<std::mutex> lock(r.mu);
    for (void *p : r.parked) {                      // single-stream order holds again from here on
        auto it = r.live.find(p);
        if (it == r.live.end()) continue;
        const size_t cls = it->second;
        r.live.erase(it);
        r.free_lists[cls].push_back(p);
        r.bytes_cached += cls;
    }
    r.parked.clear();
}

int pgx_stream_wait_detached(void) {
    PGX_REQUIRE_INIT();
    Runtime &r = rt();
    if (!r.detached) return PGX_OK;
    PGX_HIP(hipStreamWaitEvent(r.stream, r.ev_detached, 0));
    r.detached = false;
    release_parked(r);
    return PGX_OK;
}

int pgx_stream_detach(void) {
    PGX_REQUIRE_INIT();
    Runtime &r = rt();
    PGX_CHECK_ARG(r.forked, "pgx_stream_detach: not forked");
    PGX_HIP(hipEventRecord(r.ev_detached, r.side));
    r.forked = false;
    r.detached = true;                              // (what was freed while forked stays parked until the wait)
    r.current = r.stream;
    return PGX_OK;
}

int pgx_stream_is_detached(void) { return rt().ready && rt().detached ? 1 : 0; }

int pgx_stream_fork(void) {
    PGX_REQUIRE_INIT();
    Runtime &r = rt();
    PGX_CHECK_ARG(!r.forked, "pgx_stream_fork: already forked");
    if (r.detached) {                               // one side stream: what it still runs is waited for first
        if (int rc = pgx_stream_wait_detached()) return rc;
    }
    PGX_HIP(hipEventRecord(r.ev_fork, r.stream));
    PGX_HIP(hipStreamWaitEvent(r.side, r.ev_fork, 0));
    r.forked = true;
    r.current = r.side;
    return PGX_OK;
}

// A fork whose side stream does not start behind everything the main stream holds, but behind an event recorded on
// the main stream earlier (or behind nothing: event == NULL): for side work that depends on nothing the main stream is
// doing -- the next block's envelopes -- and writes only buffers whose last main-stream reader that event covers.  A
// side queue that has to be woken by the main stream's CURRENT tail starts ~17 us after it; one that waits for something
// long finished starts at once.
int pgx_stream_fork_after(void *event) {
    PGX_REQUIRE_INIT();
    Runtime &r = rt();
    PGX_CHECK_ARG(!r.forked, "pgx_stream_fork_after: already forked");
    if (r.detached) {
        if (int rc = pgx_stream_wait_detached()) return rc;
    }
    if (event != nullptr) PGX_HIP(hipStreamWaitEvent(r.side, (hipEvent_t)event, 0));
    r.forked = true;
    r.current = r.side;
    return PGX_OK;
}

int pgx_stream_is_forked(void) { return rt().ready && rt().forked ? 1 : 0; }

int pgx_stream_select(int side) {
    PGX_REQUIRE_INIT();
    Runtime &r = rt();
    PGX_CHECK_ARG(r.forked, "pgx_stream_select: not forked");
    r.current = side ? r.side : r.stream;
    return PGX_OK;
}

int pgx_stream_join(void) {
    PGX_REQUIRE_INIT();
    Runtime &r = rt();
    PGX_CHECK_ARG(r.forked, "pgx_stream_join: not forked");
    PGX_HIP(hipEventRecord(r.ev_join, r.side));
    PGX_HIP(hipStreamWaitEvent(r.stream, r.ev_join, 0));
    r.forked = false;
    r.current = r.stream;
    release_parked(r);
    return PGX_OK;
}

int pgx_malloc(void **dptr, size_t bytes) {
    PGX_REQUIRE_INIT();
    PGX_CHECK_ARG(dptr != nullptr, "pgx_malloc: null output");
    Runtime &r = rt();
    size_t cls = size_class(bytes ? bytes : 1);
    {
        std::lock_guard<std::mutex> lock(r.mu);
        auto it = r.free_lists.find(cls);
        if (it != r.free_lists.end() && !it->second.empty()) {
            void *p = it->second.back();
            it->second.pop_back();
            r.bytes_cached -= cls;
            r.live[p] = cls;
            *dptr = p;
            return PGX_OK;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, cls);
    if (e == hipErrorOutOfMemory) {
        pgx_pool_trim();
        e = hipMalloc(&p, cls);
    }
    if (e != hipSuccess)
        return pgx::fail(e == hipErrorOutOfMemory ? PGX_ERR_NOMEM : PGX_ERR_RUNTIME,
                         std::string("hipMalloc: ") + hipGetErrorString(e));
    {
        std::lock_guard<std::mutex> lock(r.mu);
        r.live[p] = cls;
    }
    *dptr = p;
    // The first large block of a process announces a streaming caller (a look-ahead window: 32 MB for 8 blocks of
    // 1 M frames): its next windows are 2, 4, 8 times longer, and a hipMalloc of that size in the middle of a stream
    // costs more than rendering the window.  Put one block of each of the next two classes aside now -- the two
    // doublings a stream's windows make before they reach their cap (6 x the block just asked for: 192 MB behind a
    // 32 MB window; every class up to 256 MB, 480 MB, until round 4 -- too much for several ranks sharing a card).
    // PGX_POOL_RESERVE=0 switches it off.
    static const bool reserve = !(getenv("PGX_POOL_RESERVE") && atoi(getenv("PGX_POOL_RESERVE")) == 0);
    static bool reserved = false;
    if (reserve && !reserved && cls >= ((size_t)16 << 20)) {
        reserved = true;
        for (size_t c = cls << 1; c <= (cls << 2) && c <= ((size_t)256 << 20); c <<= 1) {
            void *q = nullptr;
            if (hipMalloc(&q, c) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            std::lock_guard<std::mutex> lock(r.mu);
            r.free_lists[c].push_back(q);                // a cached block like any other (pgx_pool_trim frees it)
            r.bytes_cached += c;
        }
    }
    return PGX_OK;
}

int pgx_free(void *dptr) {
    if (dptr == nullptr) return PGX_OK;
    Runtime &r = rt();
    if (!r.ready) return PGX_OK;   // shutdown already released everything
    std::lock_guard<std::mutex> lock(r.mu);
    auto it = r.live.find(dptr);
    if (it == r.live.end()) return pgx::fail(PGX_ERR_INVALID, "pgx_free: unknown pointer");
    if (r.forked) {                                  // two streams in flight: park until the join
        r.parked.push_back(dptr);
        return PGX_OK;
    }
    size_t cls = it->second;
    r.live.erase(it);
    r.free_lists[cls].push_back(dptr);
    r.bytes_cached += cls;
    return PGX_OK;
}

int pgx_memset(void *dptr, int byte_value, size_t bytes) {
    PGX_REQUIRE_INIT();
    if (bytes == 0) return PGX_OK;
    PGX_CHECK_ARG(dptr != nullptr, "pgx_memset: null pointer");
    if (bytes <= (1u << 20) && bytes % 4 == 0 && (uintptr_t)dptr % 4 == 0) {     // small: a launch of our own (see
        const uint32_t b = (uint32_t)(byte_value & 0xff);                         // pgx_memcpy_d2d)
        const size_t words = bytes / 4;
        hipLaunchKernelGGL(k_fill_words, dim3((unsigned)((words + 1023) / 1024)), dim3(256), 0, rt().current,
                           (uint32_t *)dptr, b * 0x01010101u, words);
        PGX_LAUNCH_CHECK("k_fill_words");
        return PGX_OK;
    }
    PGX_HIP(hipMemsetAsync(dptr, byte_value, bytes, rt().current));
    return PGX_OK;
}

int pgx_memcpy_h2d(void *dst, const void *src_host, size_t bytes) {
    PGX_REQUIRE_INIT();
    if (bytes == 0) return PGX_OK;
    PGX_CHECK_ARG(dst != nullptr && src_host != nullptr, "pgx_memcpy_h2d: null pointer");
    PGX_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, rt().current));
    PGX_HIP(hipStreamSynchronize(rt().current));
    return PGX_OK;
}

int pgx_memcpy_d2h(void *dst_host, const void *src, size_t bytes) {
    PGX_REQUIRE_INIT();
    if (bytes == 0) return PGX_OK;
    PGX_CHECK_ARG(dst_host != nullptr && src != nullptr, "pgx_memcpy_d2h: null pointer");
    PGX_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, rt().current));
    PGX_HIP(hipStreamSynchronize(rt().current));
    return PGX_OK;
}

int pgx_memcpy_d2d(void *dst, const void *src, size_t bytes) {
    PGX_REQUIRE_INIT();
    if (bytes == 0) return PGX_OK;
    PGX_CHECK_ARG(dst != nullptr && src != nullptr, "pgx_memcpy_d2d: null pointer");
    // state blobs (a few doubles per voice: the snapshots of look-ahead windows and of the voice banks) go through
    // a kernel of our own: the runtime's copy path spends 5-8 us of stream time on any size
    if (bytes <= (1u << 20) && bytes % 4 == 0 && ((uintptr_t)dst | (uintptr_t)src) % 4 == 0) {
        const size_t words = bytes / 4;
        const unsigned groups = (unsigned)((words + 1023) / 1024);
        hipLaunchKernelGGL(k_copy_words, dim3(groups), dim3(256), 0, rt().current, (uint32_t *)dst,
                           (const uint32_t *)src, words);
        PGX_LAUNCH_CHECK("k_copy_words");
        return PGX_OK;
    }
    PGX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, rt().current));
    return PGX_OK;
}

int pgx_host_malloc(void **hptr, size_t bytes) {
    PGX_REQUIRE_INIT();
    PGX_CHECK_ARG(hptr != nullptr, "pgx_host_malloc: null output");
    Runtime &r = rt();
    const size_t cls = size_class(bytes ? bytes : 1);
    {
        std::lock_guard<std::mutex> lock(r.mu);
        auto it = r.host_free.find(cls);
        if (it != r.host_free.end() && !it->second.empty()) {
            void *p = it->second.back();
            it->second.pop_back();
            r.host_live[p] = cls;
            *hptr = p;
            return PGX_OK;
        }
    }
    void *p = nullptr;
    PGX_HIP(hipHostMalloc(&p, cls, hipHostMallocDefault));
    {
        std::lock_guard<std::mutex> lock(r.mu);
        r.host_live[p] = cls;
    }
    *hptr = p;
    return PGX_OK;
}

int pgx_host_free(void *hptr) {
    if (hptr == nullptr) return PGX_OK;
    Runtime &r = rt();
    if (!r.ready) return PGX_OK;
    std::lock_guard<std::mutex> lock(r.mu);
    auto it = r.host_live.find(hptr);
    if (it == r.host_live.end()) return pgx::fail(PGX_ERR_INVALID, "pgx_host_free: unknown pointer");
    // a copy still in flight into this block stays ordered before any later one: the copy stream is in-order
    r.host_free[it->second].push_back(hptr);
    r.host_live.erase(it);
    return PGX_OK;
}

int pgx_d2h_begin(void *dst_host, const void *src, size_t bytes, int64_t *ticket) {
    PGX_REQUIRE_INIT();
    PGX_CHECK_ARG(ticket != nullptr && (bytes == 0 || (dst_host != nullptr && src != nullptr)),
                  "pgx_d2h_begin: null argument");
    Runtime &r = rt();
    PGX_HIP(hipEventRecord(r.ev_copy_in, r.current));            // the payload is complete at this point
    PGX_HIP(hipStreamWaitEvent(r.copy, r.ev_copy_in, 0));
    if (bytes) PGX_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, r.copy));
    const int64_t t = ++r.copies_issued;
    PGX_HIP(hipEventRecord(r.copy_done[t % 64], r.copy));
    *ticket = t;
    return PGX_OK;
}

namespace {
// the event that marks `ticket` done: its own while the slot has not been recycled, else a later one
hipEvent_t copy_event(Runtime &r, int64_t ticket) {
    const int64_t t = ticket > r.copies_issued - 64 ? ticket : r.copies_issued - 63;
    return r.copy_done[t % 64];
}
}  // namespace

int pgx_d2h_wait(int64_t ticket) {
    PGX_REQUIRE_INIT();
    Runtime &r = rt();
    PGX_CHECK_ARG(ticket >= 1 && ticket <= r.copies_issued, "pgx_d2h_wait: unknown ticket");
    PGX_HIP(hipEventSynchronize(copy_event(r, ticket)));
    return PGX_OK;
}

// *done = 1 when the copy of `ticket` has landed, 0 while it is on its way: never blocks.
int pgx_d2h_query(int64_t ticket, int *done) {
    PGX_REQUIRE_INIT();
    Runtime &r = rt();
    PGX_CHECK_ARG(done != nullptr && ticket >= 1 && ticket <= r.copies_issued, "pgx_d2h_query: unknown ticket");
    const hipError_t e = hipEventQuery(copy_event(r, ticket));
    if (e != hipSuccess && e != hipErrorNotReady) PGX_HIP(e);
    *done = e == hipSuccess ? 1 : 0;
    return PGX_OK;
}

int pgx_d2h_fence(int64_t ticket) {
    PGX_REQUIRE_INIT();
    Runtime &r = rt();
    PGX_CHECK_ARG(ticket >= 1 && ticket <= r.copies_issued, "pgx_d2h_fence: unknown ticket");
    PGX_HIP(hipStreamWaitEvent(r.current, copy_event(r, ticket), 0));
    return PGX_OK;
}

int pgx_event_create(void **event) {
    PGX_REQUIRE_INIT();
    PGX_CHECK_ARG(event != nullptr, "pgx_event_create: null output");
    hipEvent_t ev;
    PGX_HIP(hipEventCreate(&ev));
    *event = (void *)ev;
    return PGX_OK;
}

int pgx_event_destroy(void *event) {
    if (event == nullptr) return PGX_OK;
    PGX_HIP(hipEventDestroy((hipEvent_t)event));
    return PGX_OK;
}

int pgx_event_record(void *event) {
    PGX_REQUIRE_INIT();
    PGX_CHECK_ARG(event != nullptr, "pgx_event_record: null event");
    PGX_HIP(hipEventRecord((hipEvent_t)event, rt().stream));
    return PGX_OK;
}

int pgx_event_elapsed_ms(void *start, void *stop, float *ms) {
    PGX_REQUIRE_INIT();
    PGX_CHECK_ARG(start && stop && ms, "pgx_event_elapsed_ms: null argument");
    PGX_HIP(hipEventSynchronize((hipEvent_t)stop));
    PGX_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return PGX_OK;
}

}  // extern "C"
