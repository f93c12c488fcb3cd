// loopccl.cpp -- TEST INFRASTRUCTURE, not product: a loop-back stand-in for the few librccl entry points libamps_recc.so binds
// (gr_amps_amd/csrc/recc_rccl.hip.h), so that the library's OWN collective code -- header exchange, status words, scatter +
// all-gather, record gather, bounded waits -- runs with 2, 4 and 8 ranks on the ONE GPU a test box has.  RCCL itself refuses two
// ranks on one device, which is why the real library has only ever been exercised with a one-rank communicator here
// (tests/test_gpu_rccl_abi.py); everything above the transport is the same code either way.  Selected with
// AMPS_RECC_RCCL_LIB=<this .so>; nothing in the product loads it otherwise.
//
// Transport: the ranks are processes on one machine; they meet in a memory-mapped file (its path travels in the 128-byte unique id)
// and move data through an 8 MiB window in it: device -> window by the sender, a barrier, window -> device by the receivers, a
// barrier.  Every call is SYNCHRONOUS: it first synchronises the stream it was given (so everything the caller ordered in front of
// the collective has happened), runs the exchange on the host, and returns with the data in place -- a legal, slow implementation of
// the stream-ordered API.
//
// LOOPCCL_ASYNC=1 (the synchronous form above cannot expose a missing or misplaced event between the library's collective
// stream and the handle's stream -- everything has happened by the time a call returns): a call then only RECORDS an event on its
// stream, queues the exchange for a worker thread of the communicator and puts a gate (a host function) on the stream; the worker
// waits for the event -- the work the caller ordered in front of the collective -- runs the exchange with copies of its own and
// opens the gate.  The call returns at once, the data lands later, in stream order: the real library's contract.  Consumers that do
// not wait for the library's `filled` event, or a collective that does not wait for `freed` / `root_ready`, now read or overwrite
// the wrong block.
//
// A peer that never comes: a barrier gives up after LOOPCCL_TIMEOUT_MS (default 60 s) and the call returns ncclSystemError -- unless
// LOOPCCL_ASYNC_HANG_MS is set: then the call behaves like the real thing with a dead peer -- it returns ncclSuccess and leaves a
// blocked operation on the stream (a host function that sleeps until ncclCommAbort / ncclCommDestroy, or that many milliseconds at
// most), which is what the library's bounded wait (rccl_wait) has to cope with.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

namespace {

constexpr int MAXR = 16, MAXOPS = 32;
constexpr size_t WIN = 8u << 20;
enum { OK = 0, HIP_ERR = 1, SYS_ERR = 2, INTERNAL = 3, BAD_ARG = 4, BAD_USE = 5 };

struct Op { uint32_t kind, peer; uint64_t bytes; };          // kind 1 = send, 2 = receive
struct Shm {
    std::atomic<uint32_t> joined, abort_flag, bar_count, bar_gen;
    uint32_t nranks, pad[3];
    struct { uint32_t n, pad; Op ops[MAXOPS]; } table[MAXR];
    alignas(4096) unsigned char window[WIN];
};
static_assert(std::atomic<uint32_t>::is_always_lock_free, "process-shared atomics");

struct Hang { std::atomic<int> release{0}; long max_ms = 0; };
struct Job { hipEvent_t ready = nullptr; std::function<int()> run; Hang gate; };
struct Comm {
    Shm *shm = nullptr;
    int nranks = 0, rank = 0;
    struct Pending { uint32_t kind, peer; uint64_t bytes; void *ptr; };
    std::vector<Pending> pending;
    std::vector<Hang *> hangs;
    bool broken = false;
    // LOOPCCL_ASYNC: the exchanges run on a worker thread, in issue order
    bool async = false;
    int device = 0;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job *> jobs;
    bool quit = false;
};
thread_local int g_depth = 0;
thread_local Comm *g_group_comm = nullptr;
thread_local hipStream_t g_group_stream = nullptr;
Comm *g_only_comm = nullptr;                                  // the process's communicator (the tests make one per process): who an EMPTY group belongs to

long env_ms(const char *name, long dflt) { const char *v = std::getenv(name); return v && *v ? std::atol(v) : dflt; }
size_t dtype_size(int t) { switch (t) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: case 9: return 2; default: return 0; } }

int barrier(Comm *c)
{
    Shm *s = c->shm;
    if (c->broken || s->abort_flag.load(std::memory_order_acquire)) return SYS_ERR;
    const uint32_t gen = s->bar_gen.load(std::memory_order_acquire);
    if (s->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->nranks) {
        s->bar_count.store(0, std::memory_order_relaxed);
        s->bar_gen.fetch_add(1, std::memory_order_release);
        return OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const long limit = env_ms("LOOPCCL_TIMEOUT_MS", 60000);
    for (uint32_t spins = 0; s->bar_gen.load(std::memory_order_acquire) == gen; spins++) {
        if (s->abort_flag.load(std::memory_order_acquire)) { c->broken = true; return SYS_ERR; }
        if (spins > 2000) {
            if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > limit) {
                s->abort_flag.store(1, std::memory_order_release);     // nobody may use a barrier a rank has walked away from
                c->broken = true;
                return SYS_ERR;
            }
            usleep(50);
        }
    }
    return OK;
}

void sleeper(void *p)
{
    Hang *h = (Hang *)p;
    const auto t0 = std::chrono::steady_clock::now();
    while (!h->release.load(std::memory_order_acquire) &&
           std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() < h->max_ms) usleep(200);
}
// a collective whose peers never came: an error, or (LOOPCCL_ASYNC_HANG_MS) a stream that does not finish, as with the real library
int peer_missing(Comm *c, hipStream_t st)
{
    const long ms = env_ms("LOOPCCL_ASYNC_HANG_MS", 0);
    if (ms <= 0) return SYS_ERR;
    Hang *h = new Hang;
    h->max_ms = ms;
    c->hangs.push_back(h);
    if (hipLaunchHostFunc(st, sleeper, h) != hipSuccess) return HIP_ERR;
    return OK;
}

// `bytes` from rank `src`'s device pointer `from` to the device pointer `to` of every rank in `dst_mask` (bit per rank); all ranks call
int transfer(Comm *c, int src, uint32_t dst_mask, const void *from, void *to, size_t bytes)
{
    for (size_t off = 0; off < bytes; off += WIN) {
        const size_t n = bytes - off < WIN ? bytes - off : WIN;
        if (c->rank == src && hipMemcpy(c->shm->window, (const char *)from + off, n, hipMemcpyDeviceToHost) != hipSuccess) { c->shm->abort_flag.store(1); return HIP_ERR; }
        if (int e = barrier(c)) return e;
        if (((dst_mask >> c->rank) & 1u) && c->rank != src && hipMemcpy((char *)to + off, c->shm->window, n, hipMemcpyHostToDevice) != hipSuccess) { c->shm->abort_flag.store(1); return HIP_ERR; }
        if (int e = barrier(c)) return e;
    }
    return OK;
}

int run_group(Comm *c, hipStream_t st, bool in_worker = false)
{
    if (c->pending.size() > MAXOPS) return BAD_USE;
    if (!in_worker && st && hipStreamSynchronize(st) != hipSuccess) return HIP_ERR;
    auto &mine = c->shm->table[c->rank];
    mine.n = (uint32_t)c->pending.size();
    for (size_t i = 0; i < c->pending.size(); i++) mine.ops[i] = { c->pending[i].kind, c->pending[i].peer, c->pending[i].bytes };
    if (barrier(c)) return in_worker ? (int)SYS_ERR : peer_missing(c, st);
    int rc = OK;
    for (int s = 0; s < c->nranks && !rc; s++) {
        uint32_t nth[MAXR] = { 0 };                                 // how many sends s -> d have been matched so far
        for (uint32_t k = 0; k < c->shm->table[s].n && !rc; k++) {
            const Op op = c->shm->table[s].ops[k];
            if (op.kind != 1) continue;
            const int d = (int)op.peer;
            if (d < 0 || d >= c->nranks || d == s) { rc = BAD_ARG; break; }
            // the matching receive: the nth[d]-th receive from s that rank d posted
            uint32_t seen = 0; int match = -1;
            for (uint32_t j = 0; j < c->shm->table[d].n; j++)
                if (c->shm->table[d].ops[j].kind == 2 && (int)c->shm->table[d].ops[j].peer == s && seen++ == nth[d]) { match = (int)j; break; }
            nth[d]++;
            if (match < 0 || c->shm->table[d].ops[match].bytes != op.bytes) { rc = BAD_USE; break; }
            const void *from = c->rank == s ? c->pending[k].ptr : nullptr;
            void *to = c->rank == d ? c->pending[match].ptr : nullptr;
            rc = transfer(c, s, 1u << d, from, to, op.bytes);
        }
    }
    // every receive must have been matched by a send (all ranks see all tables: the same verdict everywhere)
    for (int d = 0; d < c->nranks && !rc; d++)
        for (uint32_t j = 0; j < c->shm->table[d].n && !rc; j++) {
            const Op r = c->shm->table[d].ops[j];
            if (r.kind != 2) continue;
            uint32_t sends = 0, recvs = 0;
            for (uint32_t k = 0; k < c->shm->table[r.peer % MAXR].n; k++) sends += c->shm->table[r.peer % MAXR].ops[k].kind == 1 && (int)c->shm->table[r.peer % MAXR].ops[k].peer == d;
            for (uint32_t k = 0; k < c->shm->table[d].n; k++) recvs += c->shm->table[d].ops[k].kind == 2 && c->shm->table[d].ops[k].peer == r.peer;
            if (sends != recvs) rc = BAD_USE;
        }
    if (!rc) rc = barrier(c);                                       // the tables may be rewritten from here on
    c->pending.clear();
    return rc;
}

void worker_main(Comm *c)
{
    (void)hipSetDevice(c->device);
    for (;;) {
        Job *j = nullptr;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv.wait(lk, [&] { return c->quit || !c->jobs.empty(); });
            if (c->jobs.empty()) return;
            j = c->jobs.front();
            c->jobs.pop_front();
        }
        // everything the caller ordered on the stream in front of the collective has happened
        const bool ok = hipEventSynchronize(j->ready) == hipSuccess;
        const int rc = ok ? j->run() : HIP_ERR;
        (void)hipEventDestroy(j->ready);
        // a failed exchange (a peer that left) is a collective that never completes: the gate stays shut until the communicator is
        // aborted or destroyed, or LOOPCCL_ASYNC_HANG_MS have passed -- what rccl_wait has to cope with
        if (rc == OK) j->gate.release.store(1, std::memory_order_release);
    }
}
// the asynchronous form of a call: event, job, gate
int submit(Comm *c, hipStream_t st, std::function<int()> fn)
{
    Job *j = new Job;                                               // (leaked on purpose, like Hang: the gate's host function may still look at it)
    j->run = std::move(fn);
    j->gate.max_ms = env_ms("LOOPCCL_ASYNC_HANG_MS", 20000);
    if (hipEventCreateWithFlags(&j->ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(j->ready, st) != hipSuccess) return HIP_ERR;
    c->hangs.push_back(&j->gate);
    if (hipLaunchHostFunc(st, sleeper, &j->gate) != hipSuccess) return HIP_ERR;
    { std::lock_guard<std::mutex> lk(c->mu); c->jobs.push_back(j); }
    c->cv.notify_one();
    return OK;
}

} // namespace

extern "C" {

struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return BAD_ARG;
    const char *dir = std::getenv("LOOPCCL_DIR");
    std::memset(id->internal, 0, sizeof(id->internal));
    static std::atomic<unsigned> seq{0};
    std::snprintf(id->internal, sizeof(id->internal), "%s/loopccl_%ld_%llu_%u", dir && *dir ? dir : "/tmp", (long)getpid(),
                  (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count(), seq++);
    const int fd = open(id->internal, O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0) return SYS_ERR;
    const int rc = ftruncate(fd, sizeof(Shm));                       // zero-filled: all counters start at 0
    close(fd);
    return rc == 0 ? OK : SYS_ERR;
}

int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return BAD_ARG;
    id.internal[sizeof(id.internal) - 1] = 0;
    const int fd = open(id.internal, O_RDWR);
    if (fd < 0) return SYS_ERR;
    void *p = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return SYS_ERR;
    Comm *c = new Comm;
    c->shm = (Shm *)p; c->nranks = nranks; c->rank = rank;
    c->shm->joined.fetch_add(1, std::memory_order_acq_rel);
    const auto t0 = std::chrono::steady_clock::now();
    const long limit = env_ms("LOOPCCL_TIMEOUT_MS", 60000);
    while (c->shm->joined.load(std::memory_order_acquire) < (uint32_t)nranks) {
        if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > limit) { munmap(p, sizeof(Shm)); delete c; return SYS_ERR; }
        usleep(200);
    }
    if (barrier(c)) { munmap(p, sizeof(Shm)); delete c; return SYS_ERR; }
    if (rank == 0) unlink(id.internal);                               // everybody has it mapped: the name can go
    g_only_comm = c;
    c->async = env_ms("LOOPCCL_ASYNC", 0) != 0;
    if (c->async) {
        (void)hipGetDevice(&c->device);
        c->worker = std::thread(worker_main, c);
    }
    *comm = c;
    return OK;
}

int ncclBroadcast(const void *send, void *recv, size_t count, int dtype, int root, void *comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    const size_t bytes = count * dtype_size(dtype);
    if (!c || !bytes || root < 0 || root >= c->nranks) return BAD_ARG;
    auto body = [=]() -> int {
        if (barrier(c)) return SYS_ERR;
        if (c->rank == root && send != recv && hipMemcpy(recv, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return HIP_ERR;
        return transfer(c, root, ~0u, send, recv, bytes);
    };
    if (c->async) return submit(c, st, body);
    if (hipStreamSynchronize(st) != hipSuccess) return HIP_ERR;
    if (barrier(c)) return peer_missing(c, st);
    if (c->rank == root && send != recv && hipMemcpy(recv, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return HIP_ERR;
    return transfer(c, root, ~0u, send, recv, bytes);
}

int ncclAllGather(const void *send, void *recv, size_t sendcount, int dtype, void *comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    const size_t bytes = sendcount * dtype_size(dtype);
    if (!c || !bytes) return BAD_ARG;
    if (c->async)
        return submit(c, st, [=]() -> int {
            if (barrier(c)) return SYS_ERR;
            char *own = (char *)recv + (size_t)c->rank * bytes;
            if ((const void *)own != send && hipMemcpy(own, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return HIP_ERR;
            for (int r = 0; r < c->nranks; r++)
                if (int e = transfer(c, r, ~0u, send, (char *)recv + (size_t)r * bytes, bytes)) return e;
            return OK;
        });
    if (hipStreamSynchronize(st) != hipSuccess) return HIP_ERR;
    if (barrier(c)) return peer_missing(c, st);
    char *mine = (char *)recv + (size_t)c->rank * bytes;
    if ((const void *)mine != send && hipMemcpy(mine, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return HIP_ERR;
    for (int r = 0; r < c->nranks; r++)
        if (int e = transfer(c, r, ~0u, send, (char *)recv + (size_t)r * bytes, bytes)) return e;
    return OK;
}

int ncclGroupStart() { g_depth++; return OK; }
int ncclGroupEnd()
{
    if (g_depth <= 0) return BAD_USE;
    if (--g_depth) return OK;
    Comm *c = g_group_comm;
    hipStream_t st = g_group_stream;
    g_group_comm = nullptr;
    g_group_stream = nullptr;
    // A group in which THIS rank posted nothing (the library's scatter on a rank whose chunk is empty, or with one rank) still has to
    // take part in the others' exchange -- they wait at the barriers: it belongs to the process's one communicator.
    if (!c) c = g_only_comm;
    if (!c) return OK;
    if (c->async) {
        // the posted operations travel with the job (the worker owns c->pending while it runs: jobs are serial); an EMPTY group has no
        // stream of its own -- the library's scatter posts it on its collective stream like the others, which the test harness names
        auto ops = std::make_shared<std::vector<Comm::Pending>>(std::move(c->pending));
        c->pending.clear();
        return submit(c, st, [c, ops]() -> int { c->pending = *ops; return run_group(c, nullptr, true); });
    }
    return run_group(c, st);
}
static int p2p(uint32_t kind, void *ptr, size_t count, int dtype, int peer, void *comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    const size_t bytes = count * dtype_size(dtype);
    if (!c || !bytes || peer < 0 || peer >= c->nranks) return BAD_ARG;
    if (g_depth <= 0) return BAD_USE;                                // (the library always groups its sends and receives)
    if (g_group_comm && g_group_comm != c) return BAD_USE;
    g_group_comm = c; g_group_stream = st;
    c->pending.push_back({ kind, (uint32_t)peer, bytes, ptr });
    return OK;
}
int ncclSend(const void *send, size_t count, int dtype, int peer, void *comm, hipStream_t st) { return p2p(1, (void *)send, count, dtype, peer, comm, st); }
int ncclRecv(void *recv, size_t count, int dtype, int peer, void *comm, hipStream_t st) { return p2p(2, recv, count, dtype, peer, comm, st); }

static void release_hangs(Comm *c) { for (Hang *h : c->hangs) h->release.store(1, std::memory_order_release); }
static void stop_worker(Comm *c)
{
    if (!c->async) return;
    { std::lock_guard<std::mutex> lk(c->mu); c->quit = true; }
    c->cv.notify_all();
    if (c->worker.joinable()) c->worker.detach();                  // it may sit in a barrier until that times out or sees the abort flag: not waited for
}
int ncclCommAbort(void *comm)
{
    Comm *c = (Comm *)comm;
    if (!c) return BAD_ARG;
    c->shm->abort_flag.store(1, std::memory_order_release);
    release_hangs(c);
    stop_worker(c);
    c->broken = true;
    if (g_only_comm == c) g_only_comm = nullptr;
    return OK;                                                       // (the object is leaked on purpose: a sleeper may still look at its Hang)
}
int ncclCommDestroy(void *comm)
{
    Comm *c = (Comm *)comm;
    if (!c) return BAD_ARG;
    release_hangs(c);
    if (c->async) {                                                  // a clean shutdown: the queue is empty (the library synchronises its stream first)
        { std::lock_guard<std::mutex> lk(c->mu); c->quit = true; }
        c->cv.notify_all();
        if (c->worker.joinable()) c->worker.join();
    }
    if (g_only_comm == c) g_only_comm = nullptr;
    munmap(c->shm, sizeof(Shm));
    c->shm = nullptr;
    return OK;
}
int ncclCommCount(void *comm, int *n) { if (!comm || !n) return BAD_ARG; *n = ((Comm *)comm)->nranks; return OK; }
int ncclCommUserRank(void *comm, int *r) { if (!comm || !r) return BAD_ARG; *r = ((Comm *)comm)->rank; return OK; }
const char *ncclGetErrorString(int e)
{
    switch (e) { case OK: return "no error"; case HIP_ERR: return "loopccl: HIP error"; case SYS_ERR: return "loopccl: a peer is missing, left or aborted";
                 case BAD_ARG: return "loopccl: invalid argument"; case BAD_USE: return "loopccl: invalid usage"; default: return "loopccl: internal error"; }
}

} // extern "C"
