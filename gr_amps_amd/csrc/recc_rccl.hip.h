// recc_rccl.hip.h -- one band over the GPUs of a node (BASELINE configs[4]): the wideband block travels root -> everybody by RCCL over
// xGMI INSIDE the C ABI, so that a flow graph (gr::amps::recc_wideband) -- not only bench.py -- can run one handle per GPU on the same
// stream, each decoding its interleaved channel group (cfg.wideband_groups).  Channel independence it rests on: the reference keeps
// per-instance state only (lib/recc_impl.h:31-43).
//
// Two distributions: a flat ncclBroadcast, or scatter (ncclSend / ncclRecv in one group) + ncclAllGather, which puts
// B/N instead of B on every xGMI link per phase.  Both fill one of two receive buffers on a stream of the library's own, ordered by
// events against the handle's stream: the collective of push i runs beside the kernels of push i - 1 and waits only for those of
// push i - 2.
//
// RULES OF THIS FILE (round 5; a rank must never leave its peers inside a collective):
//  * everything that can fail on ONE rank -- buffers, events, argument checks that depend on the rank -- happens BEFORE a collective
//    and is carried through it as a status word: every push and every gather starts with a 16-byte all-gather {status, mode, nsamp},
//    after which all ranks hold the same verdict and either all run the data collective or none does;
//  * the receive buffers, the header words and the gather buffers are allocated in rccl_init for the capacities the ranks agree on
//    there (min push capacity, max record list): nothing is allocated inside a push or a gather;
//  * every wait of the host on a collective is bounded (hipStreamQuery poll, AMPS_RECC_RCCL_TIMEOUT_MS, default 30 s): on expiry
//    the communicator is aborted (ncclCommAbort) and the call returns -ETIMEDOUT; a rank that has to leave the game calls
//    rccl_kill itself, so its peers run into that bound instead of waiting for ever;
//  * the root's n is the push size on every rank (it travels in the header): callers that cannot agree on a block size beforehand
//    (GNU Radio schedulers in different processes) need not.
//
// RCCL is loaded at run time (dlopen of librccl.so, or of the library AMPS_RECC_RCCL_LIB names: another RCCL build, or the
// loop-back stand-in the test suite uses to run several ranks on the one GPU of a test box -- RCCL itself refuses two ranks on one
// device): the library keeps no link-time dependency on it and single-GPU users never touch it.  The communicator is per handle; the
// application carries the 128-byte unique id from rank 0 to the other ranks any way it likes -- the control plane stays the
// application's.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace amps {

struct RcclId { char internal[128]; };                       // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
constexpr int RCCL_FLOAT32 = 7;                              // ncclFloat32 of rccl.h's ncclDataType_t
constexpr int RCCL_UINT8 = 1;                                // ncclUint8
constexpr int RCCL_DIST_BROADCAST = 0, RCCL_DIST_SCATTER_ALLGATHER = 1;   // AMPS_RECC_DIST_* of amps_recc.h
constexpr uint32_t RCCL_HDR_WORDS = 4;                       // {status, mode, nsamp low, nsamp high} (push) / {status, count, 0, 0} (gather)
constexpr uint32_t RCCL_CHUNK_ALIGN = 64;                    // samples: scatter chunks start on 512-byte boundaries

struct RcclApi {
    void *lib = nullptr;
    const char *path = "";
    int (*GetUniqueId)(RcclId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclId, int) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommAbort)(void *) = nullptr;                       // optional
    int (*CommCount)(void *, int *) = nullptr;                // optional
    int (*CommUserRank)(void *, int *) = nullptr;             // optional
    const char *(*GetErrorString)(int) = nullptr;
    bool ok() const { return lib && GetUniqueId && CommInitRank && Broadcast && AllGather && Send && Recv && GroupStart && GroupEnd && CommDestroy; }
};
inline RcclApi &rccl_api()
{
    static RcclApi api = [] {
        RcclApi a;
        const char *env = std::getenv("AMPS_RECC_RCCL_LIB");
        const char *names[] = { env && *env ? env : "librccl.so", "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" };
        for (const char *n : names) { a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (a.lib) { a.path = n; break; } if (env && *env) break; }
        if (a.lib) {
            a.GetUniqueId = (int (*)(RcclId *))dlsym(a.lib, "ncclGetUniqueId");
            a.CommInitRank = (int (*)(void **, int, RcclId, int))dlsym(a.lib, "ncclCommInitRank");
            a.Broadcast = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(a.lib, "ncclBroadcast");
            a.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(a.lib, "ncclAllGather");
            a.Send = (int (*)(const void *, size_t, int, int, void *, hipStream_t))dlsym(a.lib, "ncclSend");
            a.Recv = (int (*)(void *, size_t, int, int, void *, hipStream_t))dlsym(a.lib, "ncclRecv");
            a.GroupStart = (int (*)())dlsym(a.lib, "ncclGroupStart");
            a.GroupEnd = (int (*)())dlsym(a.lib, "ncclGroupEnd");
            a.CommDestroy = (int (*)(void *))dlsym(a.lib, "ncclCommDestroy");
            a.CommAbort = (int (*)(void *))dlsym(a.lib, "ncclCommAbort");
            a.CommCount = (int (*)(void *, int *))dlsym(a.lib, "ncclCommCount");
            a.CommUserRank = (int (*)(void *, int *))dlsym(a.lib, "ncclCommUserRank");
            a.GetErrorString = (const char *(*)(int))dlsym(a.lib, "ncclGetErrorString");
        }
        return a;
    }();
    return api;
}

// what a rank brings to rccl_init: its handle's configuration, compared / combined across the ranks there
struct RcclLocal {
    uint32_t groups = 0, group = 0;                          // cfg.wideband_groups / wideband_group
    uint64_t cap_samples = 0;                                // largest wideband block one push of this handle takes
    uint32_t max_bursts = 0;                                 // cfg.max_bursts: the longest record list a drain of this handle returns
};

struct RcclState {
    void *comm = nullptr;
    bool dead = false;                                       // timed out or aborted: every later call answers -ENOTCONN
    bool stale = false;                                      // the communicator died with collectives in flight: the handle's stream state and record lists are void
                                                             // until amps_recc_reset (the consumer kernels queued behind an aborted collective run on
                                                             // a receive buffer that holds the block of two pushes ago, partly overwritten)
    int nranks = 0, rank = 0;
    uint32_t timeout_ms = 30000;
    bool timeout_set = false;                                // amps_recc_rccl_set_timeout was called: the environment does not override it
    hipStream_t cstream = nullptr;
    float2 *buf[2] = { nullptr, nullptr };
    size_t buf_samples = 0;                                  // allocated per buffer (cap_common + slack for the scatter chunks)
    uint64_t cap_common = 0;                                 // the push size every rank's handle takes: min over the ranks
    uint32_t mb_common = 0;                                  // the longest record list of any rank: max over the ranks
    hipEvent_t filled[2] = { nullptr, nullptr }, freed[2] = { nullptr, nullptr }, root_ready = nullptr;
    bool used[2] = { false, false };
    int slot = 0;
    uint32_t *hdr_dev = nullptr, *hdr_host = nullptr;        // [HDR_WORDS (1 + nranks)]: this rank's words, then everybody's (host copy: pinned)
    uint8_t *g_send = nullptr, *g_recv = nullptr;            // device [mb_common recsz] / [nranks mb_common recsz] bytes
    size_t g_cap = 0;                                        // bytes per rank in g_send / g_recv
    // timing of the data collectives (HIP events on cstream; harvested when a slot comes round again and by rccl_info)
    bool timing = false;
    hipEvent_t t0[2] = { nullptr, nullptr }, t1[2] = { nullptr, nullptr };
    bool t_open[2] = { false, false };
    double coll_ms = 0.0;
    uint64_t coll_bytes = 0, coll_count = 0;
    int last_mode = -1;
};

inline const char *rccl_errstr(int rc)
{
    RcclApi &api = rccl_api();
    return api.GetErrorString ? api.GetErrorString(rc) : "error";
}

inline void rccl_free_buffers(RcclState &r)
{
    for (int i = 0; i < 2; i++) {
        if (r.buf[i]) (void)hipFree(r.buf[i]);
        if (r.filled[i]) (void)hipEventDestroy(r.filled[i]);
        if (r.freed[i]) (void)hipEventDestroy(r.freed[i]);
        if (r.t0[i]) (void)hipEventDestroy(r.t0[i]);
        if (r.t1[i]) (void)hipEventDestroy(r.t1[i]);
    }
    if (r.root_ready) (void)hipEventDestroy(r.root_ready);
    if (r.hdr_dev) (void)hipFree(r.hdr_dev);
    if (r.hdr_host) (void)hipHostFree(r.hdr_host);
    if (r.g_send) (void)hipFree(r.g_send);
    if (r.g_recv) (void)hipFree(r.g_recv);
}

// the communicator is gone after this (ncclCommAbort makes collectives in flight return); peers run into their own bound
inline void rccl_kill(RcclState &r)
{
    if (r.comm && rccl_api().CommAbort) (void)rccl_api().CommAbort(r.comm);
    else if (r.comm && rccl_api().ok()) (void)rccl_api().CommDestroy(r.comm);
    r.comm = nullptr;
    r.dead = true;
    r.stale = true;
}

inline int rccl_wait(RcclState &r, hipStream_t s);
inline void rccl_destroy(RcclState &r)
{
    // bounded like every other wait on a collective: a peer that died inside the last data collective must not hang the
    // destructor; on expiry rccl_wait has aborted the communicator, after which the stream drains by itself
    if (r.cstream && r.comm && !r.dead) (void)rccl_wait(r, r.cstream);
    if (r.cstream) (void)hipStreamSynchronize(r.cstream);
    if (r.comm && rccl_api().ok()) (void)rccl_api().CommDestroy(r.comm);
    rccl_free_buffers(r);
    if (r.cstream) (void)hipStreamDestroy(r.cstream);
    const uint32_t keep_ms = r.timeout_ms;
    const bool keep_set = r.timeout_set, keep_stale = r.stale;
    r = RcclState();
    r.timeout_ms = keep_ms; r.timeout_set = keep_set; r.stale = keep_stale;   // a failed init leaves a fresh state, not a forgotten bound
}

// bounded wait of the host for stream s: 0, -EIO, or -ETIMEDOUT with the communicator aborted
inline int rccl_wait(RcclState &r, hipStream_t s)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; spins++) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) { (void)hipGetLastError(); rccl_kill(r); return -EIO; }
        (void)hipGetLastError();                                 // hipErrorNotReady is sticky in hipGetLastError otherwise
        if (spins > 4000) {
            const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            if ((uint64_t)ms >= r.timeout_ms) {
                std::fprintf(stderr, "amps_recc: rank %d: no answer from the other ranks within %u ms: communicator aborted\n", r.rank, r.timeout_ms);
                rccl_kill(r);
                return -ETIMEDOUT;
            }
            usleep(spins > 40000 ? 200 : 20);
        }
    }
}

// the same for an event (a drain's marker on the handle's stream, a timing event on the collective stream)
inline int rccl_wait_event(RcclState &r, hipEvent_t ev)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; spins++) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) { (void)hipGetLastError(); rccl_kill(r); return -EIO; }
        (void)hipGetLastError();
        if (spins > 4000) {
            const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            if ((uint64_t)ms >= r.timeout_ms) {
                std::fprintf(stderr, "amps_recc: rank %d: work queued behind a collective did not finish within %u ms: communicator aborted\n", r.rank, r.timeout_ms);
                rccl_kill(r);
                return -ETIMEDOUT;
            }
            usleep(spins > 40000 ? 200 : 20);
        }
    }
}

// 16 bytes of every rank to every rank; on return r.hdr_host[HDR_WORDS (1 + k) ...] = rank k's words.  Any failure on the way kills the
// communicator (the peers must not be left waiting for this rank's contribution beyond their bound).
inline int rccl_exchange(RcclState &r, const uint32_t mine[RCCL_HDR_WORDS])
{
    RcclApi &api = rccl_api();
    const size_t hb = sizeof(uint32_t) * RCCL_HDR_WORDS;
    std::memcpy(r.hdr_host, mine, hb);
    if (hipMemcpyAsync(r.hdr_dev, r.hdr_host, hb, hipMemcpyHostToDevice, r.cstream) != hipSuccess) { rccl_kill(r); return -EIO; }
    if (int rc = api.AllGather(r.hdr_dev, r.hdr_dev + RCCL_HDR_WORDS, hb, RCCL_UINT8, r.comm, r.cstream)) {
        std::fprintf(stderr, "amps_recc: ncclAllGather (header): %s\n", rccl_errstr(rc));
        rccl_kill(r);
        return -EIO;
    }
    if (hipMemcpyAsync(r.hdr_host + RCCL_HDR_WORDS, r.hdr_dev + RCCL_HDR_WORDS, hb * (size_t)r.nranks, hipMemcpyDeviceToHost, r.cstream) != hipSuccess) { rccl_kill(r); return -EIO; }
    return rccl_wait(r, r.cstream);
}
inline const uint32_t *rccl_hdr_of(const RcclState &r, int k) { return r.hdr_host + RCCL_HDR_WORDS * (1 + (size_t)k); }

// the verdict of an exchange whose word 0 is a status (0 = fine, else a positive errno): 0 if every rank is fine, this rank's own
// error if it has one, -EREMOTEIO if only others have
inline int rccl_verdict(const RcclState &r, uint32_t own)
{
    if (own) return -(int)own;
    for (int k = 0; k < r.nranks; k++) if (rccl_hdr_of(r, k)[0]) return -EREMOTEIO;
    return 0;
}

inline void rccl_harvest(RcclState &r, int slot, bool wait)
{
    if (!r.t_open[slot]) return;
    if (!wait && hipEventQuery(r.t1[slot]) != hipSuccess) { (void)hipGetLastError(); return; }
    float ms = 0.f;
    if ((wait ? (r.comm && !r.dead ? (rccl_wait_event(r, r.t1[slot]) == 0 ? hipSuccess : hipErrorUnknown) : hipEventSynchronize(r.t1[slot])) : hipSuccess) == hipSuccess &&
        hipEventElapsedTime(&ms, r.t0[slot], r.t1[slot]) == hipSuccess) {
        r.coll_ms += ms;
        r.coll_count++;
    }
    r.t_open[slot] = false;
}

// A collective itself: returns when every rank has joined ncclCommInitRank AND the ranks have compared notes.  Argument errors that
// keep a rank from joining at all (-EINVAL for a missing id or a rank outside 0 .. nranks - 1, -EBUSY, -ENOSYS) are returned at once --
// the application's other ranks then wait in RCCL's bootstrap, whose patience is RCCL's; everything else (a handle built for another
// group split, an allocation that failed) is reported to all ranks, and all of them return an error and no communicator.
inline int rccl_init(RcclState &r, const uint8_t *id, int nranks, int rank, const RcclLocal &loc, size_t recsz)
{
    RcclApi &api = rccl_api();
    if (!api.ok()) return -ENOSYS;                             // no librccl on this machine
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) return -EINVAL;
    if (r.comm || r.dead) return -EBUSY;
    if (!r.timeout_set) if (const char *t = std::getenv("AMPS_RECC_RCCL_TIMEOUT_MS")) { const long v = std::atol(t); if (v > 0) r.timeout_ms = (uint32_t)v; }
    uint32_t own = 0;
    // a handle built with channel groups decodes 1/groups of the band: the communicator must be exactly those groups, this rank its own
    if (loc.groups >= 2 && ((uint32_t)nranks != loc.groups || (uint32_t)rank != loc.group)) own = EINVAL;
    RcclId uid;
    std::memcpy(uid.internal, id, sizeof(uid.internal));
    const size_t hw = RCCL_HDR_WORDS * (1 + (size_t)nranks);
    if (hipStreamCreateWithFlags(&r.cstream, hipStreamNonBlocking) != hipSuccess) { r.cstream = nullptr; own = own ? own : EIO; }
    for (int i = 0; i < 2 && !own; i++)
        if (hipEventCreateWithFlags(&r.filled[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&r.freed[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreate(&r.t0[i]) != hipSuccess || hipEventCreate(&r.t1[i]) != hipSuccess) own = ENOMEM;
    if (!own && hipEventCreateWithFlags(&r.root_ready, hipEventDisableTiming) != hipSuccess) own = ENOMEM;
    if (hipMalloc((void **)&r.hdr_dev, sizeof(uint32_t) * hw) != hipSuccess || hipHostMalloc((void **)&r.hdr_host, sizeof(uint32_t) * hw, hipHostMallocDefault) != hipSuccess || !r.cstream) {
        // without the header words this rank cannot even say what went wrong: it does not join (see above)
        rccl_free_buffers(r);
        if (r.cstream) (void)hipStreamDestroy(r.cstream);
        r = RcclState();
        return -ENOMEM;
    }
    const int rc = api.CommInitRank(&r.comm, nranks, uid, rank);  // collective: returns when every rank has joined
    if (rc != 0) {
        std::fprintf(stderr, "amps_recc: ncclCommInitRank: %s\n", rccl_errstr(rc));
        r.comm = nullptr;
        rccl_destroy(r);
        return -EIO;
    }
    r.nranks = nranks; r.rank = rank;
    auto fail = [&](int code) { rccl_destroy(r); return code; };   // (rccl_destroy leaves a fresh state: the handle may try again)
    // round 1: {status, groups, capacity in units of 64 samples, max_bursts} -> the common capacities
    const uint32_t w1[RCCL_HDR_WORDS] = { own, loc.groups, (uint32_t)std::min<uint64_t>(loc.cap_samples / RCCL_CHUNK_ALIGN, 0xffffffffu), loc.max_bursts };
    if (int e = rccl_exchange(r, w1)) return fail(e);
    if (int v = rccl_verdict(r, own)) return fail(v);
    uint64_t cap = ~0ull;
    uint32_t mb = 0;
    bool same_groups = true;
    for (int k = 0; k < nranks; k++) {
        const uint32_t *w = rccl_hdr_of(r, k);
        same_groups = same_groups && w[1] == loc.groups;
        cap = std::min<uint64_t>(cap, (uint64_t)w[2] * RCCL_CHUNK_ALIGN);
        mb = std::max(mb, w[3]);
    }
    if (!same_groups || cap == 0 || mb == 0) return fail(-EINVAL);   // every rank sees the same words: the same verdict everywhere
    r.cap_common = cap; r.mb_common = mb;
    // round 2: the buffers for those capacities, and whether every rank got them
    own = 0;
    r.buf_samples = (size_t)cap + (size_t)RCCL_CHUNK_ALIGN * (size_t)nranks;
    for (int i = 0; i < 2; i++) if (hipMalloc((void **)&r.buf[i], sizeof(float2) * r.buf_samples) != hipSuccess) { r.buf[i] = nullptr; own = ENOMEM; }
    r.g_cap = (size_t)mb * recsz;
    if (hipMalloc((void **)&r.g_send, r.g_cap) != hipSuccess) { r.g_send = nullptr; own = ENOMEM; }
    if (hipMalloc((void **)&r.g_recv, r.g_cap * (size_t)nranks) != hipSuccess) { r.g_recv = nullptr; own = ENOMEM; }
    if (own) (void)hipGetLastError();
    const uint32_t w2[RCCL_HDR_WORDS] = { own, 0, 0, 0 };
    if (int e = rccl_exchange(r, w2)) return fail(e);
    if (int v = rccl_verdict(r, own)) return fail(v);
    return 0;
}

// The step's block into a receive buffer of this rank.  Every rank calls it in step.  root_block / nsamp_arg count on the root only; the
// root's n comes back in *nsamp_out on every rank.  *out = where the block will be once `consumer` has waited (it does, on return).
//   0            every rank has enqueued the data collective
//   -E...        NO rank has: this rank's own error (-EINVAL: no block or an unknown mode; -E2BIG: larger than the common capacity), or
//                -EREMOTEIO when another rank reported one; the communicator stays usable
//   -ETIMEDOUT / -EIO / -ENOTCONN   the communicator is gone
inline int rccl_distribute(RcclState &r, const float2 *root_block, bool root_block_on_host, size_t nsamp_arg, int root, int mode, hipStream_t consumer,
                           const float2 **out, int *slot_out, size_t *nsamp_out)
{
    RcclApi &api = rccl_api();
    if (r.dead) return -ENOTCONN;
    if (!r.comm) return -ENOSYS;
    if (root < 0 || root >= r.nranks) return -EINVAL;             // the same argument on every rank by contract: the same verdict, no collective
    const size_t N = (size_t)r.nranks;
    uint32_t own = 0;
    if (mode != RCCL_DIST_BROADCAST && mode != RCCL_DIST_SCATTER_ALLGATHER) own = EINVAL;
    if (r.rank == root) {
        // END OF STREAM: the root pushing (NULL, 0) says it has no more samples.  That is not an error: it travels in the
        // header's status word as ENODATA, EVERY rank returns -ENODATA from this call, no data collective follows and the communicator
        // stays up (a later push with samples continues the stream) -- so ranks whose own sources end at other times than the root's can
        // keep joining until they see it instead of running into their bound.
        if (!root_block && nsamp_arg == 0) own = ENODATA;
        else if (!root_block || nsamp_arg == 0) own = EINVAL;
        else if (nsamp_arg > r.cap_common) own = E2BIG;
    }
    const uint32_t mine[RCCL_HDR_WORDS] = { own, (uint32_t)mode, (uint32_t)(nsamp_arg & 0xffffffffu), (uint32_t)((uint64_t)nsamp_arg >> 32) };
    if (int e = rccl_exchange(r, mine)) return e;
    if (rccl_hdr_of(r, root)[0] == (uint32_t)ENODATA && own == (r.rank == root ? (uint32_t)ENODATA : 0u)) {
        bool others_fine = true;
        for (int k = 0; k < r.nranks; k++) if (k != root && rccl_hdr_of(r, k)[0]) others_fine = false;
        if (others_fine) return -ENODATA;
    }
    if (int v = rccl_verdict(r, own)) return v;
    for (size_t k = 0; k < N; k++) if (rccl_hdr_of(r, (int)k)[1] != (uint32_t)mode) return -EINVAL;   // the ranks disagree: everybody sees it
    const size_t nsamp = (size_t)(((uint64_t)rccl_hdr_of(r, root)[3] << 32) | rccl_hdr_of(r, root)[2]);
    *nsamp_out = nsamp;
    // From here on nothing rank-local may fail short of a broken device: every failure below kills the communicator.
    auto broken = [&](const char *what, int rc) {
        if (rc) std::fprintf(stderr, "amps_recc: %s: %s\n", what, rccl_errstr(rc));
        rccl_kill(r);
        return -EIO;
    };
    static const bool force_coll = [] { const char *e = std::getenv("AMPS_RECC_RCCL_FORCE_COLLECTIVE"); return e && e[0] == '1'; }();   // tests: the real librccl's data path with one rank
    if (N == 1 && !root_block_on_host && !force_coll) {
        // a communicator of one rank with its block already on the device: there is nobody to send to, and ncclBroadcast from the
        // caller's block into a receive buffer would be a 1 GiB device-to-device copy per push that fights the filter bank for HBM
        // (measured in round 6: 0.78 ms per step against 0.34 plain).  The block is used in place, as amps_recc_push_wideband uses a
        // device pointer -- it stays the caller's until drained; the header exchange above (and its host round trip) has run as ever.
        *out = root_block;
        *slot_out = -1;
        r.last_mode = mode;
        return 0;
    }
    const int slot = r.slot;
    r.slot ^= 1;
    rccl_harvest(r, slot, false);
    float2 *dst = r.buf[slot];
    const float2 *src = root_block;                              // where the root's samples are read from (root only)
    if (r.rank == root && root_block_on_host) {
        // a host block is staged into the receive buffer by a synchronous copy (the caller may reuse its memory on return; an async copy
        // from pageable memory gives no such guarantee) once the kernels of two pushes ago have read that buffer
        // (bounded: those kernels sit behind the collective that filled the buffer -- a peer that died inside it must not hang this rank here)
        if (r.used[slot]) if (int e = rccl_wait_event(r, r.freed[slot])) return e;
        if (hipMemcpy(dst, root_block, sizeof(float2) * nsamp, hipMemcpyHostToDevice) != hipSuccess) return broken("hipMemcpy (staging)", 0);
        src = dst;                                               // in place
    } else {
        if (r.used[slot] && hipStreamWaitEvent(r.cstream, r.freed[slot], 0) != hipSuccess) return broken("hipStreamWaitEvent", 0);   // the kernels of two pushes ago have read it
        if (r.rank == root) {
            // a device block belongs to the work the caller ordered on the handle's stream (amps_recc_wait_event, or a producer on the
            // stream passed in cfg.stream): the collective reads it on ANOTHER stream and has to wait for that work too
            if (hipEventRecord(r.root_ready, consumer) != hipSuccess || hipStreamWaitEvent(r.cstream, r.root_ready, 0) != hipSuccess) return broken("hipEventRecord", 0);
        }
    }
    if (r.timing && hipEventRecord(r.t0[slot], r.cstream) != hipSuccess) return broken("hipEventRecord", 0);
    if (mode == RCCL_DIST_BROADCAST) {
        if (int rc = api.Broadcast(r.rank == root ? (const void *)src : (const void *)dst, dst, 2 * nsamp, RCCL_FLOAT32, root, r.comm, r.cstream)) return broken("ncclBroadcast", rc);
    } else {
        // scatter: rank k's chunk = samples [k chunk, k chunk + cnt(k)); then everybody's chunk to everybody, in place
        const size_t chunk = ((nsamp + N - 1) / N + RCCL_CHUNK_ALIGN - 1) / RCCL_CHUNK_ALIGN * RCCL_CHUNK_ALIGN;
        auto cnt = [&](size_t k) { return k * chunk >= nsamp ? (size_t)0 : std::min(chunk, nsamp - k * chunk); };
        if (int rc = api.GroupStart()) return broken("ncclGroupStart", rc);
        int grc = 0;
        if (r.rank == root) {
            for (size_t k = 0; k < N && !grc; k++)
                if ((int)k != root && cnt(k)) grc = api.Send(src + k * chunk, 2 * cnt(k), RCCL_FLOAT32, (int)k, r.comm, r.cstream);
        } else if (cnt((size_t)r.rank)) grc = api.Recv(dst + (size_t)r.rank * chunk, 2 * cnt((size_t)r.rank), RCCL_FLOAT32, root, r.comm, r.cstream);
        const int erc = api.GroupEnd();
        if (grc || erc) return broken("ncclSend / ncclRecv", grc ? grc : erc);
        if (r.rank == root && src != dst && cnt((size_t)root) &&
            hipMemcpyAsync(dst + (size_t)root * chunk, src + (size_t)root * chunk, sizeof(float2) * cnt((size_t)root), hipMemcpyDeviceToDevice, r.cstream) != hipSuccess)
            return broken("hipMemcpyAsync (own chunk)", 0);
        if (int rc = api.AllGather(dst + (size_t)r.rank * chunk, dst, 2 * chunk, RCCL_FLOAT32, r.comm, r.cstream)) return broken("ncclAllGather", rc);
    }
    if (r.timing) {
        if (hipEventRecord(r.t1[slot], r.cstream) != hipSuccess) return broken("hipEventRecord", 0);
        r.t_open[slot] = true;
        r.coll_bytes += sizeof(float2) * nsamp;
    }
    r.last_mode = mode;
    if (hipEventRecord(r.filled[slot], r.cstream) != hipSuccess || hipStreamWaitEvent(consumer, r.filled[slot], 0) != hipSuccess) return broken("hipEventRecord", 0);
    *out = dst;
    *slot_out = slot;
    return 0;
}
inline int rccl_block_consumed(RcclState &r, int slot, hipStream_t consumer)
{
    if (slot < 0) return 0;                                      // used in place (one rank, device block)
    if (hipEventRecord(r.freed[slot], consumer) != hipSuccess) return -EIO;
    r.used[slot] = true;
    return 0;
}

// The drained records of every rank to `root` ("ncclGather / host copy of burst records"): a few records of 728 bytes
// per rank and drain, so the simplest collective that is in every RCCL does it -- the 16-byte header exchange ({status, count}), then
// an all-gather of the lists padded to the longest one (RCCL has no gather; grouped send / receive would save the N - 1 copies nobody
// reads, which at these sizes are microseconds).  mine: this rank's n records of recsz bytes, host memory.  On the root: all[r] =
// rank r's records; everywhere: status_or = the OR of the ranks' drain status words.  Blocks (bounded) until the collective is through.
inline int rccl_gather_records(RcclState &r, const void *mine, uint32_t n, uint32_t status, size_t recsz, int root,
                               std::vector<std::vector<uint8_t>> *all, uint32_t *status_or)
{
    RcclApi &api = rccl_api();
    if (r.dead) return -ENOTCONN;
    if (!r.comm) return -ENOSYS;
    if (root < 0 || root >= r.nranks) return -EINVAL;
    const size_t N = (size_t)r.nranks;
    if ((size_t)n * recsz > r.g_cap) { n = (uint32_t)(r.g_cap / recsz); status |= 1u; }   // cannot happen for a list of <= max_bursts records
    const uint32_t w[RCCL_HDR_WORDS] = { 0, n, status, 0 };
    if (int e = rccl_exchange(r, w)) return e;
    uint32_t longest = 0, st = 0;
    std::vector<uint32_t> counts(N);
    for (size_t k = 0; k < N; k++) { counts[k] = rccl_hdr_of(r, (int)k)[1]; longest = std::max(longest, counts[k]); st |= rccl_hdr_of(r, (int)k)[2]; }
    *status_or = st;
    if (all) all->assign(N, std::vector<uint8_t>());
    if (longest == 0) return 0;                               // every rank sees the same counts: nobody enters the second collective
    const size_t bytes = (size_t)longest * recsz;             // <= g_cap on every rank (mb_common is the maximum over the ranks)
    // a synchronous copy from pageable memory (complete on return; cstream is idle: the exchange above ended with a wait)
    if (n && hipMemcpy(r.g_send, mine, (size_t)n * recsz, hipMemcpyHostToDevice) != hipSuccess) { rccl_kill(r); return -EIO; }
    if (int rc = api.AllGather(r.g_send, r.g_recv, bytes, RCCL_UINT8, r.comm, r.cstream)) {
        std::fprintf(stderr, "amps_recc: ncclAllGather (records): %s\n", rccl_errstr(rc));
        rccl_kill(r);
        return -EIO;
    }
    if (int e = rccl_wait(r, r.cstream)) return e;
    if (r.rank == root && all) {
        for (size_t k = 0; k < N; k++) {
            (*all)[k].resize((size_t)counts[k] * recsz);
            if (counts[k] && hipMemcpy((*all)[k].data(), r.g_recv + k * bytes, (*all)[k].size(), hipMemcpyDeviceToHost) != hipSuccess) return -EIO;
        }
    }
    return 0;
}

} // namespace amps
