// recc_rccl.hip.h -- one band over the GPUs of a node (BASELINE configs[4]): the wideband block travels rank 0 -> everybody by RCCL
// ncclBroadcast over xGMI INSIDE the C ABI, so that a flow graph (gr::amps::recc_wideband) -- not only bench.py -- can run one
// handle per GPU on the same stream, each decoding its interleaved channel group (cfg.wideband_groups).
//
// RCCL is loaded at run time (dlopen of librccl.so): the library keeps no link-time dependency on it and single-GPU users never
// touch it.  The communicator is per handle; the application carries the 128-byte unique id from rank 0 to the other ranks any
// way it likes (a file, MPI, torch.distributed's store) -- the control plane stays the application's.
// The collective runs on its own stream into one of two receive buffers, ordered by events against the handle's stream: the
// broadcast of push i overlaps the kernels of push i - 1 and waits only for those of push i - 2 (what bench.py --dist broadcast
// does with torch.distributed, now behind one entry point).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace amps {

struct RcclId { char internal[128]; };                       // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
constexpr int RCCL_FLOAT32 = 7;                              // ncclFloat32 of rccl.h's ncclDataType_t
constexpr int RCCL_UINT8 = 1;                                // ncclUint8

struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(RcclId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclId, int) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok() const { return lib && GetUniqueId && CommInitRank && Broadcast && AllGather && CommDestroy; }
};
inline RcclApi &rccl_api()
{
    static RcclApi api = [] {
        RcclApi a;
        const char *names[] = { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" };
        for (const char *n : names) { a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (a.lib) break; }
        if (a.lib) {
            a.GetUniqueId = (int (*)(RcclId *))dlsym(a.lib, "ncclGetUniqueId");
            a.CommInitRank = (int (*)(void **, int, RcclId, int))dlsym(a.lib, "ncclCommInitRank");
            a.Broadcast = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(a.lib, "ncclBroadcast");
            a.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(a.lib, "ncclAllGather");
            a.CommDestroy = (int (*)(void *))dlsym(a.lib, "ncclCommDestroy");
            a.GetErrorString = (const char *(*)(int))dlsym(a.lib, "ncclGetErrorString");
        }
        return a;
    }();
    return api;
}

struct RcclState {
    void *comm = nullptr;
    int nranks = 0, rank = 0;
    hipStream_t cstream = nullptr;
    float2 *buf[2] = { nullptr, nullptr };
    size_t buf_samples = 0;
    hipEvent_t filled[2] = { nullptr, nullptr }, freed[2] = { nullptr, nullptr };
    bool used[2] = { false, false };
    int slot = 0;
    // record gather (rccl_gather_records): {count, status} of every rank, then the records themselves, padded to the longest list
    uint32_t *g_hdr = nullptr;                               // device [2 + 2 nranks]: this rank's pair, then everybody's
    uint8_t *g_send = nullptr, *g_recv = nullptr;            // device [g_cap] / [nranks g_cap] bytes
    size_t g_cap = 0;
};

inline void rccl_destroy(RcclState &r)
{
    if (r.cstream) (void)hipStreamSynchronize(r.cstream);
    if (r.comm && rccl_api().ok()) (void)rccl_api().CommDestroy(r.comm);
    for (int i = 0; i < 2; i++) {
        if (r.buf[i]) (void)hipFree(r.buf[i]);
        if (r.filled[i]) (void)hipEventDestroy(r.filled[i]);
        if (r.freed[i]) (void)hipEventDestroy(r.freed[i]);
    }
    if (r.g_hdr) (void)hipFree(r.g_hdr);
    if (r.g_send) (void)hipFree(r.g_send);
    if (r.g_recv) (void)hipFree(r.g_recv);
    if (r.cstream) (void)hipStreamDestroy(r.cstream);
    r = RcclState();
}

inline int rccl_init(RcclState &r, const uint8_t *id, int nranks, int rank)
{
    RcclApi &api = rccl_api();
    if (!api.ok()) return -ENOSYS;                             // no librccl on this machine
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) return -EINVAL;
    if (r.comm) return -EBUSY;
    RcclId uid;
    std::memcpy(uid.internal, id, sizeof(uid.internal));
    if (hipStreamCreateWithFlags(&r.cstream, hipStreamNonBlocking) != hipSuccess) return -EIO;
    for (int i = 0; i < 2; i++)
        if (hipEventCreateWithFlags(&r.filled[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&r.freed[i], hipEventDisableTiming) != hipSuccess) { rccl_destroy(r); return -ENOMEM; }
    const int rc = api.CommInitRank(&r.comm, nranks, uid, rank);  // collective: returns when every rank has joined
    if (rc != 0) {
        std::fprintf(stderr, "amps_recc: ncclCommInitRank: %s\n", api.GetErrorString ? api.GetErrorString(rc) : "error");
        r.comm = nullptr;
        rccl_destroy(r);
        return -EIO;
    }
    r.nranks = nranks; r.rank = rank;
    return 0;
}

// the step's block into a receive buffer of this rank: *out = where it will be once `consumer` has waited (it does, on return)
inline int rccl_broadcast_block(RcclState &r, const float2 *root_block, bool root_block_on_host, size_t nsamp, int root, hipStream_t consumer,
                                const float2 **out, int *slot_out)
{
    RcclApi &api = rccl_api();
    if (!r.comm) return -ENOSYS;
    if (root < 0 || root >= r.nranks || nsamp == 0) return -EINVAL;
    if (r.rank == root && !root_block) return -EINVAL;
    if (r.buf_samples < nsamp) {
        (void)hipStreamSynchronize(r.cstream);
        (void)hipStreamSynchronize(consumer);
        for (int i = 0; i < 2; i++) {
            if (r.buf[i]) (void)hipFree(r.buf[i]);
            r.buf[i] = nullptr; r.used[i] = false;
        }
        r.buf_samples = 0;
        for (int i = 0; i < 2; i++)
            if (hipMalloc((void **)&r.buf[i], sizeof(float2) * nsamp) != hipSuccess) return -ENOMEM;
        r.buf_samples = nsamp;
    }
    const int slot = r.slot;
    r.slot ^= 1;
    const void *send = r.rank == root ? (const void *)root_block : (const void *)r.buf[slot];
    if (r.rank == root && root_block_on_host) {
        // a host block is staged into the receive buffer by a synchronous copy (the caller may reuse its memory on return; an async copy
        // from pageable memory gives no such guarantee) once the kernels of two pushes ago have read that buffer
        if (r.used[slot] && hipEventSynchronize(r.freed[slot]) != hipSuccess) return -EIO;
        if (hipMemcpy(r.buf[slot], root_block, sizeof(float2) * nsamp, hipMemcpyHostToDevice) != hipSuccess) return -EIO;
        send = r.buf[slot];                                      // in place
    } else if (r.used[slot] && hipStreamWaitEvent(r.cstream, r.freed[slot], 0) != hipSuccess) return -EIO;   // the kernels of two pushes ago have read it
    const int rc = api.Broadcast(send, r.buf[slot], 2 * nsamp, RCCL_FLOAT32, root, r.comm, r.cstream);
    if (rc != 0) {
        std::fprintf(stderr, "amps_recc: ncclBroadcast: %s\n", api.GetErrorString ? api.GetErrorString(rc) : "error");
        return -EIO;
    }
    if (hipEventRecord(r.filled[slot], r.cstream) != hipSuccess || hipStreamWaitEvent(consumer, r.filled[slot], 0) != hipSuccess) return -EIO;
    *out = r.buf[slot];
    *slot_out = slot;
    return 0;
}
inline int rccl_block_consumed(RcclState &r, int slot, hipStream_t consumer)
{
    if (hipEventRecord(r.freed[slot], consumer) != hipSuccess) return -EIO;
    r.used[slot] = true;
    return 0;
}

// The drained records of every rank to `root` ("ncclGather / host copy of burst records"): a few records of 728 bytes
// per rank and drain, so the simplest collective that is in every RCCL does it -- an all-gather of {count, status}, then an
// all-gather of the lists padded to the longest one (RCCL has no gather; grouped send / receive would save the N - 1 copies nobody
// reads, which at these sizes are microseconds).  mine: this rank's n records of recsz bytes, host memory.  On the root: all[r] =
// rank r's records; everywhere: status_or = the OR of the ranks' drain status words.  Blocks until the collective is through.
inline int rccl_gather_records(RcclState &r, const void *mine, uint32_t n, uint32_t status, size_t recsz, int root,
                               std::vector<std::vector<uint8_t>> *all, uint32_t *status_or)
{
    RcclApi &api = rccl_api();
    if (!r.comm) return -ENOSYS;
    if (root < 0 || root >= r.nranks) return -EINVAL;
    const size_t N = (size_t)r.nranks;
    auto fail = [&](const char *what, int rc) { std::fprintf(stderr, "amps_recc: %s: %s\n", what, api.GetErrorString ? api.GetErrorString(rc) : "error"); return -EIO; };
    if (!r.g_hdr && hipMalloc((void **)&r.g_hdr, sizeof(uint32_t) * (2 + 2 * N)) != hipSuccess) return -ENOMEM;
    const uint32_t pair[2] = { n, status };
    // synchronous copies from / to pageable memory (complete on return), the collectives on the library's RCCL stream, which is idle
    // whenever this function is entered (it ends with a synchronisation, and a broadcast in flight is waited for here)
    if (hipStreamSynchronize(r.cstream) != hipSuccess) return -EIO;
    if (hipMemcpy(r.g_hdr, pair, sizeof(pair), hipMemcpyHostToDevice) != hipSuccess) return -EIO;
    if (int rc = api.AllGather(r.g_hdr, r.g_hdr + 2, sizeof(pair), RCCL_UINT8, r.comm, r.cstream)) return fail("ncclAllGather (counts)", rc);
    if (hipStreamSynchronize(r.cstream) != hipSuccess) return -EIO;
    std::vector<uint32_t> hdr(2 * N);
    if (hipMemcpy(hdr.data(), r.g_hdr + 2, sizeof(uint32_t) * 2 * N, hipMemcpyDeviceToHost) != hipSuccess) return -EIO;
    uint32_t longest = 0, st = 0;
    for (size_t k = 0; k < N; k++) { longest = hdr[2 * k] > longest ? hdr[2 * k] : longest; st |= hdr[2 * k + 1]; }
    *status_or = st;
    if (all) all->assign(N, std::vector<uint8_t>());
    if (longest == 0) return 0;                               // every rank sees the same counts: nobody enters the second collective
    const size_t bytes = (size_t)longest * recsz;
    if (r.g_cap < bytes) {
        if (r.g_send) (void)hipFree(r.g_send);
        if (r.g_recv) (void)hipFree(r.g_recv);
        r.g_send = r.g_recv = nullptr; r.g_cap = 0;
        if (hipMalloc((void **)&r.g_send, bytes) != hipSuccess || hipMalloc((void **)&r.g_recv, bytes * N) != hipSuccess) return -ENOMEM;
        r.g_cap = bytes;
    }
    if (n && hipMemcpy(r.g_send, mine, (size_t)n * recsz, hipMemcpyHostToDevice) != hipSuccess) return -EIO;
    if (int rc = api.AllGather(r.g_send, r.g_recv, bytes, RCCL_UINT8, r.comm, r.cstream)) return fail("ncclAllGather (records)", rc);
    if (hipStreamSynchronize(r.cstream) != hipSuccess) return -EIO;
    if (r.rank == root && all) {
        for (size_t k = 0; k < N; k++) {
            (*all)[k].resize((size_t)hdr[2 * k] * recsz);
            if (hdr[2 * k] && hipMemcpy((*all)[k].data(), r.g_recv + k * bytes, (*all)[k].size(), hipMemcpyDeviceToHost) != hipSuccess) return -EIO;
        }
    }
    return 0;
}

} // namespace amps
