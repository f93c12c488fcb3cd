// amps_recc.hip -- C ABI implementation (include/amps_recc.h) over the gfx950 kernels.
// Host code only orchestrates: buffers, stream, launches, result copy-out.  There is no CPU
// compute path: every entry point that produces data launches HIP kernels, and handle creation
// fails with -ENODEV when no HIP device is usable.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "amps_recc.h"
#include "amps_recc_numerics.h"
#include "recc_front.hip.h"
#include "recc_resolve.hip.h"
#include "recc_symbols.hip.h"
#include "recc_channelizer.hip.h"
#include "recc_rccl.hip.h"
#include "recc_xlate.hip.h"
#include "recc_bits.hip.h"
#include "recc_refchain.hip.h"

static_assert(sizeof(amps_recc_burst_t) == AMPS_RECC_BURST_BYTES, "record layout is part of the ABI");
static_assert(sizeof(amps_recc_burst_t) % 8 == 0, "records are copied as dwords");

namespace {

using namespace amps;

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            std::fprintf(stderr, "amps_recc: %s failed: %s\n", #expr, hipGetErrorString(e_)); \
            return e_ == hipErrorOutOfMemory ? -ENOMEM : -EIO;                          \
        }                                                                               \
    } while (0)

constexpr size_t LIST_WORDS = 4;   // a record list's device-side words: {slot allocator, status, published count (recc_resolve.hip.h: publish_header), pad}

enum { T_FRONT = 0, T_RESOLVE, T_DECODE, T_CARRY, T_SYMBOLS, T_CHANNELIZER, T_XLATE, T_COUNT };

struct TimedSpan { hipEvent_t a, b; int tag; uint64_t samples; };

} // namespace

struct amps_recc {
    amps_recc_cfg_t cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t C = 0, sps = 0;

    // ---- IQ seam ----
    float2 *carry[2] = { nullptr, nullptr };
    int carry_cur = 0;
    uint64_t n_done = 0;
    uint64_t origin = 0;              // absolute index of the stream's first sample (amps_recc_set_origin)
    int slicer = AMPS_SLICER_DEFAULT;       // numeric spec of the slicer (AMPS_RECC_FLAG_SLICER_* select one explicitly)
    RcclState rccl;                         // one band over several GPUs: amps_recc_rccl_init / amps_recc_push_wideband_bcast
    uint32_t r_prev = 0;
    bool origin_locked = false;       // a push has happened since the last reset
    uint64_t *gring = nullptr;
    uint32_t ring_words = 0;
    uint32_t max_waves = 0, max_chunks = 0, det_cap = 0;   // front-launch geometry bounds (see run_iq_device)
    uint32_t max_waves_bits = 0;                           // the same for the bit-domain kernel (more waves fit: 72 VGPRs, 2 KB LDS)
    uint64_t *det = nullptr;
    uint32_t *detcount = nullptr;
    uint64_t *next_allowed = nullptr, *pending = nullptr;
    unsigned long long *done_blocks = nullptr;    // {resolve workgroups of the launch in flight that have finished, record slots they reserved}
    uint64_t *capq = nullptr;                     // queue form of the capture (few channels: resolve_uses_queue)
    uint32_t *capq_count = nullptr;
    amps_recc_burst_t *records = nullptr;
    uint32_t *nrecords = nullptr;
    uint32_t *status = nullptr;
    amps_recc_burst_t *rec_host = nullptr;   // mapped pinned host memory: the capture kernel writes records here directly
    uint32_t *hdr_host = nullptr, *hdr_dev = nullptr;   // mapped pinned {nrecords, status} per record list: written by the capture kernel's last workgroup
    bool list_clean[2] = { true, true };      // the device-side {nrecords, status} of the list are zero (or a launch that zeroes them is enqueued)
    // two record lists: pushes append to the current one; drain_begin closes it (and switches), drain_end collects it
    amps_recc_burst_t *rec_host_buf[2] = { nullptr, nullptr }, *records_buf[2] = { nullptr, nullptr };
    uint8_t *bsym_host_buf[2] = { nullptr, nullptr }, *bsym_dev_buf[2] = { nullptr, nullptr };   // AMPS_RECC_FLAG_KEEP_BURSTS: [max_bursts][PACKED_BURST_BYTES] (a bit per symbol; allocated for 3374 bytes each), mapped pinned
    uint32_t *nrecords_buf[2] = { nullptr, nullptr }, *status_buf[2] = { nullptr, nullptr };
    int cur_buf = 0, open_buf = -1;
    bool open_untouched = false;      // no push has been enqueued since drain_begin: the open list's device counters are still there (header cross-check)
    hipEvent_t drain_event = nullptr;
    float2 *stage_iq = nullptr;       // device staging for host-resident IQ
    size_t stage_iq_samples = 0;
    StageFence stage_iq_fence;

    // ---- channelizer seam ----
    ChannelizerState chz;

    // ---- translate seam (recctest.grc channel filter) ----
    XlateState xl;

    // ---- reference-timing seam (G2 -> G3 -> G4 as the flow graph wires them; created on first use) ----
    RefState ref;

    // ---- symbol seam ----
    uint8_t *symbuf = nullptr;
    uint32_t *sym_len = nullptr;
    int32_t *sym_cur = nullptr;
    uint8_t *sym_stage = nullptr;     // [C][MAX_WORK_ITEMS]
    uint8_t *bursts_dev = nullptr;    // [max_bursts][3374]
    uint32_t *burst_chan_dev = nullptr;
    uint32_t *nbursts_dev = nullptr;
    amps_recc_burst_t *dec_out_dev = nullptr; // decode_bursts output staging
    size_t dec_out_cap = 0;
    uint8_t *dec_in_dev = nullptr;
    uint32_t *dec_chan_dev = nullptr;

    // ---- timing ----
    bool timing = false;
    int timing_mode = 0;              // AMPS_RECC_TIMING_*
    uint32_t dominant_tick = 0;       // launches of the dominant kernel seen in DOMINANT_SAMPLED mode
    std::vector<hipEvent_t> event_pool;   // recycled by collect_spans
    std::vector<TimedSpan> spans;
    double ms[T_COUNT] = { 0 };
    uint32_t launches_front = 0, launches_chz = 0;
    uint64_t samples_front = 0;

    // ---- amps_bch_* scratch (grow-only; no allocation per call) ----
    uint8_t *bch_in = nullptr, *bch_out = nullptr, *bch_val = nullptr, *bch_err = nullptr;
    size_t bch_in_cap = 0, bch_out_cap = 0, bch_n_cap = 0;

    // ---- debug taps (amps_recc_debug_demod) ----
    float *dbg_d = nullptr, *dbg_S = nullptr;
};

namespace {

// bits -> bytes, eight at a time: table entry v = the eight bytes (0 / 1) of the bits of v, bit i in byte i
inline const uint64_t *bit_bytes_lut()
{
    static const std::vector<uint64_t> lut = [] {
        std::vector<uint64_t> t(256);
        for (int v = 0; v < 256; v++) { uint64_t w = 0; for (int i = 0; i < 8; i++) w |= (uint64_t)((v >> i) & 1) << (8 * i); t[v] = w; }
        return t;
    }();
    return lut.data();
}
// a packed record (recc_decode.hip.h: decode_core_store_packed, 216 bytes) -> the ABI's amps_recc_burst_t: the two bit arrays back to
// one byte per bit
inline void expand_packed_record(amps_recc_burst_t *dst, const uint8_t *src)
{
    const uint64_t *lut = bit_bytes_lut();
    uint8_t *d = (uint8_t *)dst;
    std::memcpy(d, src, REC_RAW_OFF);
    const uint8_t *raw = src + 13 * 4, *dec = src + 24 * 4;
    for (int k = 0; k < (REC_DEC_OFF - REC_RAW_OFF) / 8; k++) std::memcpy(d + REC_RAW_OFF + 8 * k, &lut[raw[k]], 8);            // 42 x 8 = 336 bytes
    for (int k = 0; k < (REC_TAIL_OFF - REC_DEC_OFF + 7) / 8; k++) std::memcpy(d + REC_DEC_OFF + 8 * k, &lut[dec[k]], 8);        // 32 x 8: 4 bytes into the tail ...
    std::memcpy(d + REC_TAIL_OFF, src + 32 * 4, sizeof(amps_recc_burst_t) - REC_TAIL_OFF);                                       // ... which is written last
}

// the kept symbol blob: PACKED_BURST_BYTES of bits -> the 3374 bytes (values 0 / 1) gr::amps::recc publishes (lib/recc_impl.cc:126)
inline void expand_packed_burst(uint8_t *dst, const uint8_t *src)
{
    const uint64_t *lut = bit_bytes_lut();
    constexpr int FULL = AMPS_RECC_CAPTURE_SYMS / 8;                                                                             // 421 whole bytes of bits
    for (int k = 0; k < FULL; k++) std::memcpy(dst + 8 * k, &lut[src[k]], 8);
    for (int i = 8 * FULL; i < AMPS_RECC_CAPTURE_SYMS; i++) dst[i] = (uint8_t)((src[i >> 3] >> (i & 7)) & 1u);                    // the last six symbols
}

template <typename T> int dev_alloc(T **p, size_t n)
{
    if (n == 0) n = 1;
    hipError_t e = hipMalloc((void **)p, n * sizeof(T));
    if (e != hipSuccess) { *p = nullptr; return -ENOMEM; }
    return 0;
}

struct SpanGuard {   // records a pair of events around a launch when timing is on
    amps_recc *h; int tag; uint64_t samples; hipEvent_t a = nullptr, b = nullptr; bool on;
    static hipEvent_t take(amps_recc *h)
    {
        if (!h->event_pool.empty()) { hipEvent_t e = h->event_pool.back(); h->event_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        return hipEventCreate(&e) == hipSuccess ? e : nullptr;
    }
    SpanGuard(amps_recc *h_, int tag_, uint64_t samples_ = 0) : h(h_), tag(tag_), samples(samples_), on(h_->timing)
    {
        // "dominant" mode: only the streaming kernel of the seam (front kernel, or the channelizer on the wideband
        // seam) is bracketed -- two event records per push instead of ten, for timed regions that should not be perturbed
        if (on && h->timing_mode >= AMPS_RECC_TIMING_DOMINANT) on = (tag == T_CHANNELIZER) || (tag == T_FRONT && !h->chz.enabled);
        if (on && h->timing_mode == AMPS_RECC_TIMING_DOMINANT_SAMPLED) on = (h->dominant_tick++ % AMPS_RECC_TIMING_SAMPLE_PERIOD) == 0;
        if (!on) return;
        a = take(h); b = take(h);
        if (!a || !b) { on = false; return; }
        (void)hipEventRecord(a, h->stream);
    }
    void end()     // close the span now (the destructor then does nothing)
    {
        if (!on) return;
        (void)hipEventRecord(b, h->stream);
        h->spans.push_back({ a, b, tag, samples });
        on = false;
    }
    static void end_cb(void *g) { static_cast<SpanGuard *>(g)->end(); }
    ~SpanGuard() { end(); }
};

void collect_spans(amps_recc *h)   // collects the spans whose events have completed (all of them after a stream sync)
{
    size_t keep = 0;
    for (auto &s : h->spans) {
        if (hipEventQuery(s.b) != hipSuccess) { h->spans[keep++] = s; continue; }
        float t = 0.f;
        if (hipEventElapsedTime(&t, s.a, s.b) == hipSuccess) {
            h->ms[s.tag] += t;
            if (s.tag == T_FRONT) { h->launches_front++; h->samples_front += s.samples; }
            if (s.tag == T_CHANNELIZER) h->launches_chz++;
        }
        h->event_pool.push_back(s.a);
        h->event_pool.push_back(s.b);
    }
    h->spans.resize(keep);
}

// AMPS_RECC_DEBUG_SYNC=1: synchronise after every launch and say which kernel ran (fault isolation)
bool debug_sync_enabled()
{
    static int v = -1;
    if (v < 0) { const char *e = std::getenv("AMPS_RECC_DEBUG_SYNC"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}
int debug_sync(amps_recc *h, const char *what)
{
    if (!debug_sync_enabled()) return 0;
    std::fprintf(stderr, "amps_recc[debug]: %s ...", what); std::fflush(stderr);
    hipError_t e = hipStreamSynchronize(h->stream);
    std::fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e)); std::fflush(stderr);
    return e == hipSuccess ? 0 : -EIO;
}

// Waits of the host on the handle's stream.  With a live communicator the handle's kernels may be queued behind a data collective
// whose peer has died: every such wait is bounded (rccl_wait / rccl_wait_event: on expiry the communicator is aborted, the collective
// returns and the stream drains) -- ADVICE r05: drain, destroy and reset used to sit in hipStreamSynchronize / hipEventSynchronize.
int sync_stream(amps_recc *h, hipStream_t s)
{
    if (h->rccl.comm && !h->rccl.dead) { if (int rc = rccl_wait(h->rccl, s)) return rc; return 0; }
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -EIO;
}
int sync_event(amps_recc *h, hipEvent_t e)
{
    if (h->rccl.comm && !h->rccl.dead) { if (int rc = rccl_wait_event(h->rccl, e)) return rc; return 0; }
    return hipEventSynchronize(e) == hipSuccess ? 0 : -EIO;
}
// A communicator that died under work in flight leaves the stream state advanced over a block that never arrived: everything the
// handle would report from there on is void.  The data seams and the drains answer -ESTALE until amps_recc_reset.
#define STALE_CHECK(h) do { if ((h)->rccl.stale) return -ESTALE; } while (0)

void select_record_list(amps_recc *h, int b)
{
    h->cur_buf = b;
    h->records = h->records_buf[b]; h->rec_host = h->rec_host_buf[b];
    h->nrecords = h->nrecords_buf[b]; h->status = h->status_buf[b];
}

constexpr int HDR_STRIDE = 16;      // dwords between the two lists' host headers (one 64-byte line each: the CPU clears one while the GPU may write the other)
constexpr uint64_t MIN_SPAN = 16;   // tiles per wave at least: bounds the 2-tile halo overhead to 12.5 % on tiny pushes

uint32_t next_pow2(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return (uint32_t)p; }

int reset_state(amps_recc *h)
{
    hipStream_t s = h->stream;
    if (h->carry[0]) {
        HIP_TRY(hipMemsetAsync(h->carry[0], 0, sizeof(float2) * (size_t)h->C * CARRY_CAP, s));
        HIP_TRY(hipMemsetAsync(h->carry[1], 0, sizeof(float2) * (size_t)h->C * CARRY_CAP, s));
        HIP_TRY(hipMemsetAsync(h->gring, 0xff, sizeof(uint64_t) * (size_t)h->C * h->ring_words, s));
        HIP_TRY(hipMemsetAsync(h->detcount, 0, sizeof(uint32_t) * (size_t)h->C * h->max_chunks, s));
        HIP_TRY(hipMemsetAsync(h->next_allowed, 0, sizeof(uint64_t) * h->C, s));
        HIP_TRY(hipMemsetAsync(h->pending, 0xff, sizeof(uint64_t) * h->C, s));
        HIP_TRY(hipMemsetAsync(h->done_blocks, 0, (1 + DONE_GROUPS) * sizeof(unsigned long long), s));
        if (h->capq_count) HIP_TRY(hipMemsetAsync(h->capq_count, 0, sizeof(uint32_t), s));
    }
    for (int b = 0; b < 2; b++) {
        HIP_TRY(hipMemsetAsync(h->nrecords_buf[b], 0, LIST_WORDS * sizeof(uint32_t), s));
        h->list_clean[b] = true;
    }
    std::memset(h->hdr_host, 0, 2 * HDR_STRIDE * sizeof(uint32_t));
    h->open_buf = -1;
    select_record_list(h, 0);
    HIP_TRY(hipMemsetAsync(h->symbuf, 0, (size_t)h->C * AMPS_RECC_SYMBUF, s));
    HIP_TRY(hipMemsetAsync(h->sym_len, 0, sizeof(uint32_t) * h->C, s));
    HIP_TRY(hipMemsetAsync(h->sym_cur, 0xff, sizeof(int32_t) * h->C, s));
    HIP_TRY(hipMemsetAsync(h->nbursts_dev, 0, sizeof(uint32_t), s));
    h->carry_cur = 0;
    h->n_done = 0;
    h->origin = 0;
    h->origin_locked = false;
    h->r_prev = 0;
    int rc = channelizer_reset(h->chz, s);
    if (rc) return rc;
    rc = xlate_reset(h->xl, s);
    if (rc) return rc;
    rc = ref_reset(h->ref, s);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

// AMPS_RECC_CHECK_HEADER=1 (the GPU test suite sets it): a drain that finds the stream idle behind it compares the header the
// capture kernel's last workgroup published to host memory with the list's device-side counters.  The publish orders three
// relaxed device atomics by their completion (recc_resolve.hip.h); this check is what would notice a compiler or architecture
// change breaking that.
bool check_header_enabled()
{
    static int v = -1;
    if (v < 0) { const char *e = std::getenv("AMPS_RECC_CHECK_HEADER"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

bool bits_kernel_is_front()   // AMPS_RECC_BITS_KERNEL=front: search the bit ring with recc_front_kernel<3,1,BITS> (cross-check)
{
    static int v = -1;
    if (v < 0) { const char *e = std::getenv("AMPS_RECC_BITS_KERNEL"); v = (e && e[0] == 'f') ? 1 : 0; }
    return v == 1;
}

// tiles in flight per wave beyond the one being processed; AMPS_RECC_DEPTH overrides for experiments.  Measured with the
// non-temporal tile loads (832 x 2^18, ms): spec A 0.329 at depth 1 / 0.342 at depth 2 (its discriminator needs the registers:
// depth 2 costs a wave per SIMD); specs B / C 0.298 / 0.286: with the arctangent gone the kernel only waits for HBM
int front_depth(int slicer)
{
    static int env = -1;
    if (env < 0) { const char *e = std::getenv("AMPS_RECC_DEPTH"); env = e ? std::atoi(e) : 0; if (env < 0 || env > 3) env = 0; }
    if (env) return env;
    // round 4: depth 2 is compiled for four waves per SIMD too (AMPS_FRONT_D2_BLOCKS = 4: a handful of prologue spills, none in the tile
    // loop).  Same box, ms: spec A 0.3336 at depth 1 / 0.3343 at depth 2; D 0.3208 / 0.3174; B 0.3118 (three waves) -> 0.3038; C 0.3122 ->
    // 0.3055.  The default spec keeps depth 1 -- 1 % slower and no scratch at all; the opt-in specs B and C take depth 2
    return (slicer == AMPS_SLICER_ATAN_BOXCAR || slicer == AMPS_SLICER_EXACT) ? 1 : 2;
}
typedef void (*front_kernel_t)(FrontArgs);
// the instantiation of the streaming kernel for (samples per symbol, slicer spec, tolerant sync, tiles in flight)
template <int SPS, int SL> front_kernel_t front_kernel_of(bool tol, int depth)
{
    if (tol) return recc_front_kernel<SPS, 1, false, true, SL>;
    if constexpr (SL == AMPS_SLICER_ATAN_BOXCAR) { if (depth == 3) return recc_front_kernel<SPS, 3, false, false, SL>; }
    if (depth == 2) return recc_front_kernel<SPS, 2, false, false, SL>;
    return recc_front_kernel<SPS, 1, false, false, SL>;
}
template <int SPS> front_kernel_t front_kernel_for(int slicer, bool tol)
{
    const int depth = front_depth(slicer);
    switch (slicer) {
    case AMPS_SLICER_PRODUCT: return front_kernel_of<SPS, AMPS_SLICER_PRODUCT>(tol, depth);
    case AMPS_SLICER_SINE: return front_kernel_of<SPS, AMPS_SLICER_SINE>(tol, depth);
    case AMPS_SLICER_EXACT: return front_kernel_of<SPS, AMPS_SLICER_EXACT>(tol, depth);
    default: return front_kernel_of<SPS, AMPS_SLICER_ATAN_BOXCAR>(tol, depth);
    }
}
template <int SPS> int front_blocks_per_cu(int slicer, bool tol)   // of the kernel launch_front<SPS> will pick
{
    int n = 0;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, front_kernel_for<SPS>(slicer, tol), 256, 0);
    return (e == hipSuccess && n > 0) ? n : 2;
}
int front_blocks_per_cu_for(uint32_t sps, int slicer, bool tol)
{
    switch (sps) {
    case 2: return front_blocks_per_cu<2>(slicer, tol);     // the two-kernel (unfused) form of the wideband seam at D = 768 only
    case 3: return front_blocks_per_cu<3>(slicer, tol);
    case 4: return front_blocks_per_cu<4>(slicer, tol);
    case 5: return front_blocks_per_cu<5>(slicer, tol);
    case 6: return front_blocks_per_cu<6>(slicer, tol);
    case 8: return front_blocks_per_cu<8>(slicer, tol);
    case 10: return front_blocks_per_cu<10>(slicer, tol);
    case 12: return front_blocks_per_cu<12>(slicer, tol);
    default: return 2;
    }
}
template <int SPS> void launch_front(const FrontArgs &fa, dim3 grid, hipStream_t s, int slicer)
{
    hipLaunchKernelGGL(front_kernel_for<SPS>(slicer, fa.tol != 0), grid, dim3(256), 0, s, fa);
}

bool sps_supported(uint32_t sps)
{
    switch (sps) { case 3: case 4: case 5: case 6: case 8: case 10: case 12: return true; default: return false; }
}

int dispatch_front(uint32_t sps, const FrontArgs &fa, dim3 grid, hipStream_t s, int slicer)
{
    switch (sps) {
    case 2: launch_front<2>(fa, grid, s, slicer); break;
    case 3: launch_front<3>(fa, grid, s, slicer); break;
    case 4: launch_front<4>(fa, grid, s, slicer); break;
    case 5: launch_front<5>(fa, grid, s, slicer); break;
    case 6: launch_front<6>(fa, grid, s, slicer); break;
    case 8: launch_front<8>(fa, grid, s, slicer); break;
    case 10: launch_front<10>(fa, grid, s, slicer); break;
    case 12: launch_front<12>(fa, grid, s, slicer); break;
    default: return -EINVAL;
    }
    return 0;
}

// The streaming / bit-domain kernel of a push also does the push's housekeeping (thread 0): it clears the capture queue count,
// and the {count, status} of the record list that is NOT current if those are still dirty from its last use -- a list is only
// appended to while it is current, and a drain reads its header from host memory (published by the capture kernel), so the
// idle list's device counters are free to be cleared by any later launch.  Two memsets and one copy fewer per push.
void front_housekeeping_args(amps_recc *h, FrontArgs &fa)
{
    h->open_untouched = false;              // this launch may clear the counters of the list a split drain has open
    fa.zero1 = h->capq_count;               // null in the fused form
    const int idle = h->cur_buf ^ 1;
    fa.zero2 = h->list_clean[idle] ? nullptr : h->nrecords_buf[idle];
    h->list_clean[idle] = true;
    h->list_clean[h->cur_buf] = false;      // the capture kernel of this push may append to the current list
}

// the fused chain on channel-major device IQ: front -> carry -> resolve -> capture/decode
// one workgroup per channel; wide groups when a channel spans more wave segments than 256 lanes cover in one batch
// AMPS_RECC_BITS_KERNEL=separate: the wideband seam's trigger search as its own launch (recc_bits_kernel, rounds 2-5) instead of the
// search stage inside the resolve kernel -- an independent launch structure the GPU suite checks the default against
static bool bits_search_is_separate()
{
    static const bool v = [] { const char *e = std::getenv("AMPS_RECC_BITS_KERNEL"); return e && std::strcmp(e, "separate") == 0; }();
    return v;
}
// the search stage inside the resolve kernel serves the many-channel form (no capture queue) at the wideband seam's two rates
static bool search_in_resolve(const amps_recc *h) { return h->chz.enabled && !h->capq && !bits_kernel_is_front() && !bits_search_is_separate() && (h->sps == 2 || h->sps == 3); }

static void launch_resolve(amps_recc *h, ResolveArgs &ra, hipStream_t s, bool search = false)
{
    // capture + decode side of the kernel
    ra.gring = h->gring; ra.ring_mask = h->ring_words - 1; ra.ring_words = h->ring_words; ra.cap_words = resolve_cap_words(h->sps);
    ra.records = h->records; ra.nrecords = h->nrecords; ra.rec_cap = h->cfg.max_bursts; ra.status = h->status;
    ra.majority = (h->cfg.flags & AMPS_RECC_FLAG_MAJORITY) ? 1u : 0u;
    ra.track = (h->cfg.flags & AMPS_RECC_FLAG_FIXED_TIMING) ? 0u : 1u;
    ra.burst_syms = h->bsym_dev_buf[h->cur_buf];
    ra.done_blocks = h->done_blocks; ra.hdr_host = h->hdr_dev + HDR_STRIDE * h->cur_buf;
    ra.capq = h->capq; ra.capq_count = h->capq_count; ra.capq_cap = h->cfg.max_bursts;
    const size_t lds = h->capq ? 0 : resolve_dyn_lds(h->sps);
#ifdef RESOLVE_TIMELINE
    static unsigned long long *tl_dev = nullptr;
    if (!tl_dev) (void)hipMalloc((void **)&tl_dev, (size_t)24 * 8 * 4096);
    if (tl_dev && h->C <= 4096) { (void)hipMemsetAsync(tl_dev, 0, (size_t)24 * 8 * h->C, s); ra.tl = tl_dev; }
#endif
    // The wide instantiation (a channel cut into more wave segments than 256 lanes compact in one batch) exists for handles with few
    // channels, which always take the queue form: its 36.9 KB of static LDS next to the fused capture form's dynamic LDS is a
    // combination max_chunks never produces for 64 channels or more (Tc / span + 2 <= max_waves / C + 4 <= 131 there).  Held here, so
    // that a change to either threshold cannot turn into a launch failure: a handle without a queue stays on the narrow kernel, whose
    // batches walk any number of segments.
    const bool wide = ra.tiles_per_channel / ra.span + 2 > (uint64_t)RESOLVE_THREADS && h->capq != nullptr;
    // two samples per symbol (the wideband seam at D = 768) have their own capture rule: a second instantiation of the kernels, so that
    // the default ones carry nothing of it
    const bool two = h->sps == 2;
    if (wide) {
        if (two) hipLaunchKernelGGL((recc_resolve_kernel<RESOLVE_THREADS_WIDE, RESOLVE_LDS_HITS_WIDE, true>), dim3(h->C), dim3(RESOLVE_THREADS_WIDE), lds, s, ra);
        else hipLaunchKernelGGL((recc_resolve_kernel<RESOLVE_THREADS_WIDE, RESOLVE_LDS_HITS_WIDE, false>), dim3(h->C), dim3(RESOLVE_THREADS_WIDE), lds, s, ra);
    } else if (search) {
        const dim3 g(h->C), b(RESOLVE_THREADS);
        if (two) {
            if (ra.search_tol) hipLaunchKernelGGL((recc_resolve_kernel<RESOLVE_THREADS, RESOLVE_LDS_HITS, true, 2, true>), g, b, lds, s, ra);
            else hipLaunchKernelGGL((recc_resolve_kernel<RESOLVE_THREADS, RESOLVE_LDS_HITS, true, 2, false>), g, b, lds, s, ra);
        } else {
            if (ra.search_tol) hipLaunchKernelGGL((recc_resolve_kernel<RESOLVE_THREADS, RESOLVE_LDS_HITS, false, 3, true>), g, b, lds, s, ra);
            else hipLaunchKernelGGL((recc_resolve_kernel<RESOLVE_THREADS, RESOLVE_LDS_HITS, false, 3, false>), g, b, lds, s, ra);
        }
    } else {
        if (two) hipLaunchKernelGGL((recc_resolve_kernel<RESOLVE_THREADS, RESOLVE_LDS_HITS, true>), dim3(h->C), dim3(RESOLVE_THREADS), lds, s, ra);
        else hipLaunchKernelGGL((recc_resolve_kernel<RESOLVE_THREADS, RESOLVE_LDS_HITS, false>), dim3(h->C), dim3(RESOLVE_THREADS), lds, s, ra);
    }
    if (h->capq) {
        const dim3 gq(std::min<uint32_t>(h->cfg.max_bursts, 2048u));
        const size_t ldsq = (size_t)resolve_cap_stride(ra.cap_words) * 8;
        if (two) hipLaunchKernelGGL(recc_capture_kernel<true>, gq, dim3(64), ldsq, s, ra);
        else hipLaunchKernelGGL(recc_capture_kernel<false>, gq, dim3(64), ldsq, s, ra);
    }
#ifdef RESOLVE_TIMELINE
    if (const char *path = std::getenv("AMPS_RECC_RESOLVE_TIMELINE")) {   // the last launch's stamps, raw
        std::vector<unsigned long long> tl((size_t)24 * h->C);
        if (ra.tl && hipStreamSynchronize(s) == hipSuccess && hipMemcpy(tl.data(), ra.tl, tl.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE *f = std::fopen(path, "wb")) { std::fwrite(tl.data(), 8, tl.size(), f); std::fclose(f); }
    }
#endif
}

int run_iq_device(amps_recc *h, const float2 *iq, uint64_t ld, uint32_t nsamp)
{
    h->origin_locked = true;
    if (nsamp == 0) return 0;
    hipStream_t s = h->stream;
    const uint32_t avail = h->r_prev + nsamp;
    const uint32_t P = (avail / 64) * 64, r_new = avail - P;
    // geometry of the persistent front launch
    const uint32_t Tc = (P + TILE - 1) / TILE;
    const uint64_t G = (uint64_t)h->C * Tc;
    uint32_t nwaves = (uint32_t)std::min<uint64_t>(h->max_waves, (G + MIN_SPAN - 1) / MIN_SPAN);
    if (nwaves == 0) nwaves = 1;
    const uint32_t span = (uint32_t)((G + nwaves - 1) / nwaves);
    if (P && (uint64_t)(Tc + span - 1) / span + 1 > h->max_chunks) return -E2BIG;
    if (P) {
        FrontArgs fa{};
        fa.block = iq; fa.carry = h->carry[h->carry_cur]; fa.ld = ld;
        fa.r_prev = h->r_prev; fa.avail = avail; fa.P = P; fa.tiles_per_channel = Tc; fa.n_channels = h->C; fa.span = span;
        fa.n_done = h->n_done; fa.gring = h->gring; fa.ring_mask = h->ring_words - 1; fa.ring_words = h->ring_words;
        fa.det = h->det; fa.detcount = h->detcount; fa.max_chunks = h->max_chunks; fa.det_cap = h->det_cap;
        fa.tol = h->cfg.sync_tolerance;
        fa.force_ones = ((h->slicer == AMPS_SLICER_PRODUCT || h->slicer == AMPS_SLICER_EXACT) && h->n_done == h->origin) ? h->sps : 0u;   // specs B, D: no partner yet
        fa.status = h->status; fa.dbg_d = h->dbg_d; fa.dbg_S = h->dbg_S; fa.dbg_channel = 0;
        front_housekeeping_args(h, fa);
        fa.carry_out = h->carry[h->carry_cur ^ 1]; fa.carry_n = HALO + r_new;     // the next push's carry is written by the streaming kernel itself
        SpanGuard g(h, T_FRONT, P);
        if (debug_sync_enabled())
            std::fprintf(stderr, "amps_recc[debug]: front waves=%u span=%u Tc=%u C=%u P=%u avail=%u r_prev=%u ld=%llu n_done=%llu ring_words=%u max_chunks=%u det_cap=%u\n",
                         nwaves, span, Tc, h->C, P, avail, h->r_prev, (unsigned long long)ld, (unsigned long long)h->n_done, h->ring_words, h->max_chunks, h->det_cap);
        int rc = dispatch_front(h->sps, fa, dim3((nwaves + 3) / 4), s, h->slicer);
        if (rc) return rc;
    }
    if (int rc = debug_sync(h, "front")) return rc;
    if (!P) {                                              // a push too short for a 64-sample word only moves the carry
        CarryArgs ca{};
        ca.block = iq; ca.carry_in = h->carry[h->carry_cur]; ca.carry_out = h->carry[h->carry_cur ^ 1];
        ca.ld = ld; ca.r_prev = h->r_prev; ca.avail = avail; ca.P = P; ca.r_new = r_new;
        SpanGuard g(h, T_CARRY);
        hipLaunchKernelGGL(recc_carry_kernel, dim3((HALO + r_new + 255) / 256, h->C), dim3(256), 0, s, ca);
    }
    if (int rc = debug_sync(h, "carry")) return rc;
    if (P) {
        ResolveArgs ra{};
        ra.det = h->det; ra.detcount = h->detcount; ra.max_chunks = h->max_chunks; ra.det_cap = h->det_cap;
        ra.tiles_per_channel = Tc; ra.span = span; ra.sps = h->sps; ra.n_proc = h->n_done + P;
        ra.next_allowed = h->next_allowed; ra.pending = h->pending;
        {
            SpanGuard g(h, T_RESOLVE);
            launch_resolve(h, ra, s);
        }
        if (int rc = debug_sync(h, "resolve + capture")) return rc;
    }
    HIP_TRY(hipGetLastError());
    h->n_done += P;
    h->r_prev = r_new;
    h->carry_cur ^= 1;
    return 0;
}

} // namespace

extern "C" {

int amps_recc_abi_version(void) { return AMPS_RECC_ABI_VERSION; }
int amps_recc_default_slicer(void) { return AMPS_SLICER_DEFAULT; }
uint32_t amps_recc_default_wideband_decim(void) { return (uint32_t)CHZ_D768; }
size_t amps_recc_burst_size(void) { return sizeof(amps_recc_burst_t); }

const char *amps_recc_strerror(int code)
{
    switch (-code) {
    case 0: return "ok";
    case EINVAL: return "invalid argument";
    case ENODEV: return "no usable HIP device (this library has no CPU fallback)";
    case ENOMEM: return "out of device or host memory";
    case EIO: return "HIP runtime error";
    case ENOSPC: return "result list overflowed max_bursts";
    case E2BIG: return "push larger than the configured capacity";
    case EOVERFLOW: return "trigger-hit list overflowed";
    case ENOSYS: return "seam not configured on this handle";
    case EBUSY: return "busy: a split drain is open, the stream has started, or the handle already has a communicator";
    case ETIMEDOUT: return "no answer from the other ranks within the RCCL timeout: communicator aborted";
    case ENOTCONN: return "the handle's communicator has been aborted";
    case ESTALE: return "the communicator died with collectives in flight: stream state and record lists are void until amps_recc_reset";
    case EREMOTEIO: return "another rank reported an error: no rank ran the collective";
    case ENODATA: return "end of stream: the root of the distributed push has no more samples";
    default: return "unknown error";
    }
}

int amps_recc_create(amps_recc_t **out, const amps_recc_cfg_t *cfg_in)
{
    if (!out || !cfg_in) return -EINVAL;
    *out = nullptr;
    if (cfg_in->struct_size != sizeof(amps_recc_cfg_t)) return -EINVAL;
    amps_recc_cfg_t cfg_v = *cfg_in;                                  // wideband handles: decimation 0 = the library default, samples per symbol 0 = what goes with it
    if (cfg_v.wideband_channels) {
        if (cfg_v.wideband_decim == 0) cfg_v.wideband_decim = amps_recc_default_wideband_decim();
        if (cfg_v.samples_per_symbol == 0 && (cfg_v.wideband_decim == (uint32_t)CHZ_D || cfg_v.wideband_decim == (uint32_t)CHZ_D768))
            cfg_v.samples_per_symbol = 1536u / cfg_v.wideband_decim;
    }
    const amps_recc_cfg_t *cfg = &cfg_v;
    if (cfg->n_channels < 1 || cfg->max_bursts < 1) return -EINVAL;
    // the channelizer's geometry (M = 1024 at 30.72 Msps) delivers 60 ksps = 3 samples per Manchester symbol at D = 512 and 40 ksps =
    // 2 at D = 768; the bit-domain kernels behind it are built for exactly those.  Two samples per symbol exist on the wideband seam only
    // (the streaming kernel of the IQ seam has no such instantiation).
    const bool wide768 = cfg->wideband_channels && cfg->wideband_decim == (uint32_t)CHZ_D768;
    if (cfg->max_samples_per_push && !(wide768 ? cfg->samples_per_symbol == 2 : sps_supported(cfg->samples_per_symbol))) return -EINVAL;
    if (cfg->wideband_channels && (cfg->samples_per_symbol != (wide768 ? 2u : 3u) || cfg->max_samples_per_push == 0)) return -EINVAL;
    if (wide768 && bits_kernel_is_front()) return -EINVAL;        // AMPS_RECC_BITS_KERNEL=front: the streaming kernel's bit-domain mode is built for 3 samples per symbol
    if (cfg->sync_tolerance > AMPS_RECC_MAX_SYNC_TOLERANCE) return -EINVAL;
    {
        const uint32_t sl = cfg->flags & (AMPS_RECC_FLAG_SLICER_PRODUCT | AMPS_RECC_FLAG_SLICER_SINE | AMPS_RECC_FLAG_SLICER_ATAN | AMPS_RECC_FLAG_SLICER_EXACT);
        if (sl & (sl - 1)) return -EINVAL;                         // at most one slicer spec
    }
    if (cfg->n_channels >= (1u << (64 - CAPQ_POS_BITS))) return -EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return -ENODEV;
    int dev = cfg->device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) return -ENODEV; }
    if (dev >= ndev) return -ENODEV;
    if (hipSetDevice(dev) != hipSuccess) return -ENODEV;

    // AMPS_RECC_TRACE_CREATE=1: the steps of this function to stderr (which HIP call a creation hangs or crawls in when many
    // processes share one device -- the 8-rank rehearsals of tests/test_gpu_bench_ranks.py)
    const bool trace = std::getenv("AMPS_RECC_TRACE_CREATE") != nullptr;
    auto step = [&](const char *what) { if (trace) { std::fprintf(stderr, "amps_recc_create[%d]: %s\n", (int)getpid(), what); std::fflush(stderr); } };
    step("device set");
    amps_recc *h = new (std::nothrow) amps_recc();
    if (!h) return -ENOMEM;
    h->cfg = *cfg;
    h->device = dev;
    h->C = cfg->n_channels;
    if (cfg->wideband_channels && cfg->wideband_groups > 1) {      // a channel-group handle owns only its group's rows
        if (cfg->flags & AMPS_RECC_FLAG_UNFUSED_WIDEBAND) { delete h; return -EINVAL; }
        const int rows = chz_rows(*cfg, nullptr, nullptr);
        if (rows < 1) { delete h; return -EINVAL; }
        h->C = (uint32_t)rows;
    }
    h->sps = cfg->samples_per_symbol;
    h->slicer = (cfg->flags & AMPS_RECC_FLAG_SLICER_PRODUCT) ? AMPS_SLICER_PRODUCT
              : (cfg->flags & AMPS_RECC_FLAG_SLICER_SINE) ? AMPS_SLICER_SINE
              : (cfg->flags & AMPS_RECC_FLAG_SLICER_EXACT) ? AMPS_SLICER_EXACT
              : (cfg->flags & AMPS_RECC_FLAG_SLICER_ATAN) ? AMPS_SLICER_ATAN_BOXCAR : AMPS_SLICER_DEFAULT;
    h->timing = (cfg->flags & AMPS_RECC_FLAG_TIME_KERNELS) != 0;
    h->timing_mode = h->timing ? AMPS_RECC_TIMING_ALL : AMPS_RECC_TIMING_OFF;
    if (cfg->stream) h->stream = (hipStream_t)cfg->stream;
    else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return -EIO; }
        h->own_stream = true;
    }
    int rc = 0;
    const size_t C = h->C;
    step("stream created");
    // results + symbol seam (always present)
    for (int b = 0; b < 2; b++) {   // {nrecords, status} of a list are adjacent: one 8-byte copy / memset serves both
        rc |= dev_alloc(&h->nrecords_buf[b], LIST_WORDS);
        h->status_buf[b] = h->nrecords_buf[b] ? h->nrecords_buf[b] + 1 : nullptr;
    }
    rc |= dev_alloc(&h->symbuf, C * AMPS_RECC_SYMBUF);
    rc |= dev_alloc(&h->sym_len, C);
    rc |= dev_alloc(&h->sym_cur, C);
    rc |= dev_alloc(&h->sym_stage, C * (size_t)(AMPS_RECC_MAX_WORK_ITEMS + 1));
    rc |= dev_alloc(&h->bursts_dev, (size_t)cfg->max_bursts * AMPS_RECC_CAPTURE_SYMS);
    rc |= dev_alloc(&h->burst_chan_dev, cfg->max_bursts);
    rc |= dev_alloc(&h->nbursts_dev, 1);
    step("symbol seam buffers allocated");
    // result records live in mapped, pinned host memory (zero copy: PACKED_RECORD_BYTES = 216 per burst over PCIe while the
    // kernels run, expanded to the ABI's 728 by drain_end_impl); h->records is the device-side view of the same allocation
    for (int b = 0; b < 2; b++)
        if (hipHostMalloc((void **)&h->rec_host_buf[b], sizeof(amps_recc_burst_t) * (size_t)cfg->max_bursts, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void **)&h->records_buf[b], h->rec_host_buf[b], 0) != hipSuccess) rc |= -ENOMEM;
    if (cfg->flags & AMPS_RECC_FLAG_KEEP_BURSTS)
        for (int b = 0; b < 2; b++)
            if (hipHostMalloc((void **)&h->bsym_host_buf[b], (size_t)cfg->max_bursts * AMPS_RECC_CAPTURE_SYMS, hipHostMallocMapped) != hipSuccess ||
                hipHostGetDevicePointer((void **)&h->bsym_dev_buf[b], h->bsym_host_buf[b], 0) != hipSuccess) rc |= -ENOMEM;
    if (hipHostMalloc((void **)&h->hdr_host, 2 * HDR_STRIDE * sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&h->hdr_dev, h->hdr_host, 0) != hipSuccess) rc |= -ENOMEM;
    if (hipEventCreateWithFlags(&h->drain_event, hipEventDisableTiming) != hipSuccess) rc |= -ENOMEM;
    if (!rc) select_record_list(h, 0);
    step("pinned record lists mapped");
    // IQ seam
    if (!rc && cfg->max_samples_per_push) {
        const uint64_t maxs = cfg->max_samples_per_push;
        h->ring_words = next_pow2(maxs + (uint64_t)h->sps * (AMPS_RECC_CAPTURE_SYMS + 2 * AMPS_RECC_TRIGGER_SYMS + 64) + 2 * TILE) / 64;
        // Front launch = one round of resident waves: 4 workgroups (16 waves) per CU, each wave owning an equal
        // span of the flattened (channel, tile) space.  A channel is covered by at most max_waves/C + 2 segments.
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { amps_recc_destroy(h); return -ENODEV; }
        h->max_waves = (uint32_t)prop.multiProcessorCount * 4u * (uint32_t)front_blocks_per_cu_for(h->sps, h->slicer, cfg->sync_tolerance != 0);   // exactly one resident round
        if (const char *e = std::getenv("AMPS_RECC_MAX_WAVES")) { const long v = std::atol(e); if (v >= 4 && (uint32_t)v <= h->max_waves) h->max_waves = (uint32_t)v & ~3u; }   // experiments: fewer, longer wave streams (span geometry against the HBM channel interleave)
        {
            int nb = 0;
            hipError_t e;
            if (bits_kernel_is_front())
                e = cfg->sync_tolerance ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, recc_front_kernel<3, 1, true, true>, 256, 0)
                                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, recc_front_kernel<3, 1, true, false>, 256, 0);
            else if (h->sps == 2)
                e = cfg->sync_tolerance ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, recc_bits_kernel<2, true>, 256, 0)
                                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, recc_bits_kernel<2, false>, 256, 0);
            else
                e = cfg->sync_tolerance ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, recc_bits_kernel<3, true>, 256, 0)
                                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, recc_bits_kernel<3, false>, 256, 0);
            if (e != hipSuccess || nb < 1) nb = 4;
            if (nb > 8) nb = 8;                                  // max_chunks below assumes at most 32 waves per CU
            h->max_waves_bits = (uint32_t)prop.multiProcessorCount * 4u * (uint32_t)nb;
        }
        const uint64_t max_tiles = (maxs + 63 + TILE - 1) / TILE;
        const uint64_t max_span = std::max<uint64_t>(MIN_SPAN, (C * max_tiles + h->max_waves - 1) / h->max_waves);
        h->max_chunks = (uint32_t)(prop.multiProcessorCount * 32u / C + 3);   // bound for any occupancy
        h->det_cap = (uint32_t)(max_span * TILE / ((uint64_t)AMPS_RECC_TRIGGER_SYMS * h->sps) + 4);
        rc |= dev_alloc(&h->carry[0], C * CARRY_CAP);
        rc |= dev_alloc(&h->carry[1], C * CARRY_CAP);
        rc |= dev_alloc(&h->gring, C * h->ring_words);
        rc |= dev_alloc(&h->det, C * h->max_chunks * h->det_cap);
        rc |= dev_alloc(&h->detcount, C * h->max_chunks);
        rc |= dev_alloc(&h->next_allowed, C);
        rc |= dev_alloc(&h->pending, C);
        rc |= dev_alloc(&h->done_blocks, 1 + DONE_GROUPS);
        if (resolve_uses_queue((uint32_t)C)) { rc |= dev_alloc(&h->capq, cfg->max_bursts); rc |= dev_alloc(&h->capq_count, 1); }
    }
    step("IQ seam buffers allocated");
    if (!rc && cfg->wideband_channels) rc = channelizer_create(h->chz, *cfg, h->stream);
    step("channelizer created");
    if (!rc) rc = reset_state(h);
    step("state reset");
    if (rc) { amps_recc_destroy(h); return rc; }
    *out = h;
    return 0;
}

void amps_recc_destroy(amps_recc_t *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)sync_stream(h, h->stream); (void)hipStreamSynchronize(h->stream); }   // bounded first: a dead peer must not hang the destructor
    collect_spans(h);
    h->stage_iq_fence.destroy();
    for (hipEvent_t e : h->event_pool) (void)hipEventDestroy(e);
    h->event_pool.clear();
    void *bufs[] = { h->carry[0], h->carry[1], h->gring, h->det, h->detcount, h->next_allowed, h->pending,
                     h->done_blocks, h->capq, h->capq_count, h->nrecords_buf[0], h->nrecords_buf[1], h->stage_iq, h->symbuf, h->sym_len, h->sym_cur,
                     h->sym_stage, h->bursts_dev, h->burst_chan_dev, h->nbursts_dev, h->dec_out_dev, h->dec_in_dev,
                     h->dec_chan_dev, h->dbg_d, h->dbg_S, h->bch_in, h->bch_out, h->bch_val, h->bch_err };
    for (void *p : bufs) if (p) (void)hipFree(p);
    for (int b = 0; b < 2; b++) if (h->rec_host_buf[b]) (void)hipHostFree(h->rec_host_buf[b]);
    for (int b = 0; b < 2; b++) if (h->bsym_host_buf[b]) (void)hipHostFree(h->bsym_host_buf[b]);
    if (h->drain_event) (void)hipEventDestroy(h->drain_event);
    if (h->hdr_host) (void)hipHostFree(h->hdr_host);
    rccl_destroy(h->rccl);
    channelizer_destroy(h->chz);
    xlate_destroy(h->xl);
    ref_destroy(h->ref);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int amps_recc_reset(amps_recc_t *h)
{
    if (!h) return -EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    (void)sync_stream(h, h->stream);          // bounded while a communicator lives (on expiry it is aborted and the stream drains)
    HIP_TRY(hipStreamSynchronize(h->stream));
    collect_spans(h);
    h->rccl.stale = false;                    // a fresh stream: whatever an aborted collective left behind is gone
    if (h->open_buf >= 0) { volatile uint32_t *hdr = h->hdr_host + HDR_STRIDE * h->open_buf; hdr[0] = 0u; hdr[1] = 0u; h->open_buf = -1; }
    return reset_state(h);
}

int amps_recc_push_symbols(amps_recc_t *h, const uint8_t *syms, size_t ld, int n, int mem,
                           uint8_t *bursts_out, uint32_t *burst_channel, size_t cap, size_t *nout)
{
    if (!h || !nout) return -EINVAL;
    *nout = 0;
    if (h->chz.enabled && h->chz.groups > 1) return -ENOSYS;   // a channel-group handle owns its group's rows of the wideband seam only
    if (n < 1) return 0;                                   // lib/recc_impl.cc:99-102
    if (n > AMPS_RECC_MAX_WORK_ITEMS) return -EINVAL;      // lib/recc_impl.cc:103
    if (!syms || ld < (size_t)n) return -EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const uint8_t *dsyms = syms;
    uint64_t dld = ld;
    if (mem == AMPS_MEM_HOST) {
        dld = AMPS_RECC_MAX_WORK_ITEMS + 1;
        HIP_TRY(hipMemcpy2DAsync(h->sym_stage, dld, syms, ld, (size_t)n, h->C, hipMemcpyHostToDevice, s));
        dsyms = h->sym_stage;
    }
    HIP_TRY(hipMemsetAsync(h->nbursts_dev, 0, sizeof(uint32_t), s));
    SymbolsArgs a{};
    a.syms = dsyms; a.ld = dld; a.n = n; a.symbuf = h->symbuf; a.len = h->sym_len; a.curstart = h->sym_cur;
    a.bursts = h->bursts_dev; a.burst_chan = h->burst_chan_dev; a.nbursts = h->nbursts_dev;
    a.cap = h->cfg.max_bursts; a.status = h->status;
    {
        SpanGuard g(h, T_SYMBOLS);
        hipLaunchKernelGGL(recc_symbols_kernel, dim3(h->C), dim3(256), 0, s, a);
    }
    HIP_TRY(hipGetLastError());
    uint32_t nb = 0;
    HIP_TRY(hipMemcpyAsync(&nb, h->nbursts_dev, sizeof(nb), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    collect_spans(h);
    int rc = 0;
    if (nb > h->cfg.max_bursts) { nb = h->cfg.max_bursts; rc = -ENOSPC; }
    if (nb == 0) return rc;
    std::vector<uint32_t> chan(nb);
    std::vector<uint8_t> data((size_t)nb * AMPS_RECC_CAPTURE_SYMS);
    HIP_TRY(hipMemcpy(chan.data(), h->burst_chan_dev, sizeof(uint32_t) * nb, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(data.data(), h->bursts_dev, data.size(), hipMemcpyDeviceToHost));
    std::vector<uint32_t> order(nb);
    for (uint32_t i = 0; i < nb; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return chan[x] < chan[y]; });
    size_t k = 0;
    for (; k < nb && k < cap; k++) {
        if (bursts_out) std::memcpy(bursts_out + k * AMPS_RECC_CAPTURE_SYMS, &data[(size_t)order[k] * AMPS_RECC_CAPTURE_SYMS], AMPS_RECC_CAPTURE_SYMS);
        if (burst_channel) burst_channel[k] = chan[order[k]];
    }
    *nout = k;
    if (nb > cap) rc = -ENOSPC;
    return rc;
}

int amps_recc_decode_bursts(amps_recc_t *h, const uint8_t *bursts, size_t nbursts, int mem,
                            const uint32_t *burst_channel, amps_recc_burst_t *out)
{
    if (!h || (!bursts && nbursts) || (!out && nbursts)) return -EINVAL;
    if (nbursts == 0) return 0;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    if (nbursts > h->dec_out_cap) {
        if (h->dec_out_dev) (void)hipFree(h->dec_out_dev);
        if (h->dec_in_dev) (void)hipFree(h->dec_in_dev);
        if (h->dec_chan_dev) (void)hipFree(h->dec_chan_dev);
        h->dec_out_dev = nullptr; h->dec_in_dev = nullptr; h->dec_chan_dev = nullptr; h->dec_out_cap = 0;
        if (dev_alloc(&h->dec_out_dev, nbursts) || dev_alloc(&h->dec_in_dev, nbursts * AMPS_RECC_CAPTURE_SYMS) ||
            dev_alloc(&h->dec_chan_dev, nbursts)) return -ENOMEM;
        h->dec_out_cap = nbursts;
    }
    const uint8_t *din = bursts;
    if (mem == AMPS_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(h->dec_in_dev, bursts, nbursts * AMPS_RECC_CAPTURE_SYMS, hipMemcpyHostToDevice, s));
        din = h->dec_in_dev;
    }
    const uint32_t *dchan = nullptr;
    if (burst_channel) {
        HIP_TRY(hipMemcpyAsync(h->dec_chan_dev, burst_channel, nbursts * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        dchan = h->dec_chan_dev;
    }
    {
        SpanGuard g(h, T_DECODE);
        uint32_t grid = (uint32_t)std::min<size_t>(nbursts, 4096);
        hipLaunchKernelGGL(recc_decode_bursts_kernel, dim3(grid), dim3(64), 0, s, din, dchan, (uint32_t)nbursts, h->dec_out_dev,
                           (h->cfg.flags & AMPS_RECC_FLAG_MAJORITY) ? 1u : 0u);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, h->dec_out_dev, nbursts * sizeof(amps_recc_burst_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    collect_spans(h);
    return 0;
}

int amps_recc_push_iq(amps_recc_t *h, const float *iq, size_t ld, size_t nsamp, int mem)
{
    if (!h) return -EINVAL;
    if (!h->carry[0]) return -ENOSYS;
    if (h->chz.enabled && h->chz.groups > 1) return -ENOSYS;   // a channel-group handle owns its group's rows of the wideband seam only: its
                                                               // records are numbered through row2chan, which means nothing on this seam
    if (nsamp == 0) return 0;
    if (!iq || ld < nsamp) return -EINVAL;
    if (nsamp > h->cfg.max_samples_per_push) return -E2BIG;
    HIP_TRY(hipSetDevice(h->device));
    const float2 *d = (const float2 *)iq;
    uint64_t dld = ld;
    if (mem == AMPS_MEM_HOST) {
        if (int rc = h->stage_iq_fence.wait()) return rc;         // the previous push may still be reading the staging buffer
        if (h->stage_iq_samples < (size_t)h->C * h->cfg.max_samples_per_push) {
            if (h->stage_iq) (void)hipFree(h->stage_iq);
            h->stage_iq = nullptr; h->stage_iq_samples = 0;
            if (dev_alloc(&h->stage_iq, (size_t)h->C * h->cfg.max_samples_per_push)) return -ENOMEM;
            h->stage_iq_samples = (size_t)h->C * h->cfg.max_samples_per_push;
        }
        dld = nsamp;
        // synchronous on purpose: an async copy from pageable memory can return before the source has been read (the
        // fuzzer caught blocks freed right after the call being copied late); the fence above has already made sure
        // nobody is reading the staging buffer, and the kernels below are enqueued after the copy has completed
        HIP_TRY(hipMemcpy2D(h->stage_iq, dld * sizeof(float2), iq, ld * sizeof(float2), nsamp * sizeof(float2),
                            h->C, hipMemcpyHostToDevice));
        d = h->stage_iq;
    }
    int rc = run_iq_device(h, d, dld, (uint32_t)nsamp);
    if (!rc && mem == AMPS_MEM_HOST) rc = h->stage_iq_fence.arm(h->stream);
    return rc;
}

namespace {
// bit-domain tail of the fused wideband seam: the slicer bits of [n_done, n_done + P) are already in the ring
int run_bits_device(amps_recc *h, uint32_t P)
{
    if (P == 0) return 0;
    hipStream_t s = h->stream;
    const uint32_t Tc = (P + TILE - 1) / TILE;
    const uint64_t G = (uint64_t)h->C * Tc;
    uint32_t nwaves = (uint32_t)std::min<uint64_t>(h->max_waves_bits, (G + MIN_SPAN - 1) / MIN_SPAN);
    if (nwaves == 0) nwaves = 1;
    const uint32_t span = (uint32_t)((G + nwaves - 1) / nwaves);
    if (search_in_resolve(h)) {
        // round 6: ONE launch -- every channel's workgroup searches its own slicer bits (a quarter of the push per wave, hits in LDS),
        // then resolves, captures and decodes them as ever; the launch's housekeeping goes with it
        FrontArgs hk{};
        front_housekeeping_args(h, hk);
        ResolveArgs ra{};
        ra.tiles_per_channel = Tc; ra.span = span; ra.sps = h->sps; ra.n_proc = h->n_done + P;
        ra.next_allowed = h->next_allowed; ra.pending = h->pending;
        ra.search_P = P; ra.search_tol = h->cfg.sync_tolerance; ra.zero1 = hk.zero1; ra.zero2 = hk.zero2;
        {
            SpanGuard g(h, T_RESOLVE);
            launch_resolve(h, ra, s, true);
        }
        HIP_TRY(hipGetLastError());
        h->n_done += P;
        return 0;
    }
    if ((uint64_t)(Tc + span - 1) / span + 1 > h->max_chunks) return -E2BIG;
    FrontArgs fa{};
    fa.r_prev = 0; fa.avail = P; fa.P = P; fa.tiles_per_channel = Tc; fa.n_channels = h->C; fa.span = span;
    fa.n_done = h->n_done; fa.gring = h->gring; fa.ring_mask = h->ring_words - 1; fa.ring_words = h->ring_words;
    fa.det = h->det; fa.detcount = h->detcount; fa.max_chunks = h->max_chunks; fa.det_cap = h->det_cap; fa.status = h->status;
    fa.tol = h->cfg.sync_tolerance;
    front_housekeeping_args(h, fa);
    {
        SpanGuard g(h, T_FRONT, P);
        // the dedicated bit-domain kernel; AMPS_RECC_BITS_KERNEL=front selects the bit-domain mode of the streaming kernel
        // instead (an independent implementation of the same search: the two must agree)
        const dim3 grid((nwaves + 3) / 4);
        if (bits_kernel_is_front()) {
            if (fa.tol) hipLaunchKernelGGL((recc_front_kernel<3, 1, true, true>), grid, dim3(256), 0, s, fa);
            else hipLaunchKernelGGL((recc_front_kernel<3, 1, true>), grid, dim3(256), 0, s, fa);
        } else if (h->sps == 2) {
            if (fa.tol) hipLaunchKernelGGL((recc_bits_kernel<2, true>), grid, dim3(256), 0, s, fa);
            else hipLaunchKernelGGL((recc_bits_kernel<2, false>), grid, dim3(256), 0, s, fa);
        } else if (fa.tol) hipLaunchKernelGGL((recc_bits_kernel<3, true>), grid, dim3(256), 0, s, fa);
        else hipLaunchKernelGGL((recc_bits_kernel<3, false>), grid, dim3(256), 0, s, fa);
    }
#ifdef BITS_TIMELINE
    if (const char *path = std::getenv("AMPS_RECC_BITS_TIMELINE")) {
        std::vector<unsigned long long> tl(3 * 16384);
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpyFromSymbol(tl.data(), HIP_SYMBOL(bits_tl), tl.size() * 8) == hipSuccess)
            if (FILE *f = std::fopen(path, "wb")) { unsigned long long nw = nwaves; std::fwrite(&nw, 8, 1, f); std::fwrite(tl.data(), 8, tl.size(), f); std::fclose(f); }
    }
#endif
    ResolveArgs ra{};
    ra.det = h->det; ra.detcount = h->detcount; ra.max_chunks = h->max_chunks; ra.det_cap = h->det_cap;
    ra.tiles_per_channel = Tc; ra.span = span; ra.sps = h->sps; ra.n_proc = h->n_done + P;
    ra.next_allowed = h->next_allowed; ra.pending = h->pending;
    {
        SpanGuard g(h, T_RESOLVE);
        launch_resolve(h, ra, s);
    }
    HIP_TRY(hipGetLastError());
    h->n_done += P;
    return 0;
}
} // namespace

int amps_recc_push_wideband(amps_recc_t *h, const float *iq, size_t nsamp, int mem)
{
    if (!h) return -EINVAL;
    STALE_CHECK(h);
    if (!h->chz.enabled) return -ENOSYS;
    if (nsamp == 0) return 0;
    if (!iq) return -EINVAL;
    h->origin_locked = true;
    HIP_TRY(hipSetDevice(h->device));
    // Fused form (default): filter bank + discriminator + boxcar + slicer in one kernel, then the bit-domain
    // correlator.  AMPS_RECC_FLAG_UNFUSED_WIDEBAND keeps the two-kernel form (channel-major intermediate in HBM).
    const bool fused = !(h->cfg.flags & AMPS_RECC_FLAG_UNFUSED_WIDEBAND);
    const float2 *chan_iq = nullptr;
    uint64_t ld = 0;
    uint32_t nout = 0;
    int rc;
    {
        SpanGuard g(h, T_CHANNELIZER, nsamp);
        rc = channelizer_run(h->chz, (const float2 *)iq, nsamp, mem, h->stream, &chan_iq, &ld, &nout, fused, h->gring, h->ring_words, h->n_done, h->slicer,
                             SpanGuard::end_cb, &g);
    }
    if (rc) return rc;
    if (nout > h->cfg.max_samples_per_push) return -E2BIG;
    if (fused) return run_bits_device(h, nout);
    return run_iq_device(h, chan_iq, ld, nout);
}

int amps_recc_rccl_unique_id(uint8_t *id)
{
    if (!id) return -EINVAL;
    RcclApi &api = rccl_api();
    if (!api.ok()) return -ENOSYS;
    RcclId uid;
    if (api.GetUniqueId(&uid) != 0) return -EIO;
    std::memcpy(id, uid.internal, sizeof(uid.internal));
    return 0;
}

int amps_recc_rccl_init(amps_recc_t *h, const uint8_t *id, int nranks, int rank)
{
    if (!h) return -EINVAL;
    if (!h->chz.enabled) return -ENOSYS;
    HIP_TRY(hipSetDevice(h->device));
    RcclLocal loc;
    loc.groups = h->cfg.wideband_groups >= 2 ? h->cfg.wideband_groups : 0;
    loc.group = h->cfg.wideband_group;
    // the largest block one push takes: max_samples_per_push frames of 512 samples (amps_recc_push_wideband: -E2BIG beyond)
    loc.cap_samples = (uint64_t)h->cfg.max_samples_per_push * (uint64_t)h->chz.D;
    loc.max_bursts = h->cfg.max_bursts;
    const int rc = rccl_init(h->rccl, id, nranks, rank, loc, sizeof(amps_recc_burst_t));
    if (rc == 0) h->rccl.timing = h->timing;
    return rc;
}

int amps_recc_rccl_set_timeout(amps_recc_t *h, uint32_t milliseconds)
{
    if (!h || milliseconds == 0) return -EINVAL;
    h->rccl.timeout_ms = milliseconds;
    h->rccl.timeout_set = true;
    return 0;
}

int amps_recc_rccl_abort(amps_recc_t *h)
{
    if (!h) return -EINVAL;
    if (!h->rccl.comm) return h->rccl.dead ? 0 : -ENOSYS;
    HIP_TRY(hipSetDevice(h->device));
    rccl_kill(h->rccl);
    return 0;
}

int amps_recc_rccl_info(amps_recc_t *h, amps_recc_rccl_info_t *info)
{
    if (!h || !info || info->struct_size != sizeof(amps_recc_rccl_info_t)) return -EINVAL;
    RcclState &r = h->rccl;                                   // (no communicator yet: nranks = 0, the device fields are still filled)
    HIP_TRY(hipSetDevice(h->device));
    std::memset((char *)info + sizeof(info->struct_size), 0, sizeof(*info) - sizeof(info->struct_size));
    info->alive = r.comm ? 1 : 0;
    info->nranks = r.nranks; info->rank = r.rank;
    info->comm_nranks = -1; info->comm_rank = -1;
    // (librccl is loaded only for a handle that has, or had, a communicator: a single-GPU caller who asks for the device identity must
    // not pull a second copy of RCCL into a process that bundles its own -- ADVICE r05)
    const bool touched = r.comm != nullptr || r.dead;
    static RcclApi none;
    RcclApi &api = touched ? rccl_api() : none;
    if (r.comm && api.CommCount) { int v = -1; if (api.CommCount(r.comm, &v) == 0) info->comm_nranks = v; }
    if (r.comm && api.CommUserRank) { int v = -1; if (api.CommUserRank(r.comm, &v) == 0) info->comm_rank = v; }
    info->device = h->device;
    info->max_samples_per_push = r.cap_common;
    info->max_bursts_per_gather = r.mb_common;
    info->timeout_ms = r.timeout_ms;
    if (r.comm) for (int i = 0; i < 2; i++) rccl_harvest(r, i, true);
    info->collectives_timed = r.coll_count;
    info->collective_ms = r.coll_ms;
    info->collective_bytes = r.coll_count ? r.coll_bytes : 0;
    info->last_mode = r.last_mode;
    std::snprintf(info->library, sizeof(info->library), "%s", api.path);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) == hipSuccess) {
        static_assert(sizeof(info->device_uuid) == sizeof(prop.uuid.bytes), "uuid");
        std::memcpy(info->device_uuid, prop.uuid.bytes, sizeof(info->device_uuid));
        info->pci_bus = prop.pciBusID; info->pci_device = prop.pciDeviceID; info->pci_domain = prop.pciDomainID;
    }
    return 0;
}

int amps_recc_push_wideband_dist(amps_recc_t *h, const float *iq, size_t nsamp, int mem, int root, int mode, size_t *npushed)
{
    if (npushed) *npushed = 0;
    if (!h) return -EINVAL;
    if (h->rccl.dead) return -ENOTCONN;
    if (!h->chz.enabled || !h->rccl.comm) return -ENOSYS;
    HIP_TRY(hipSetDevice(h->device));
    const float2 *blk = nullptr;
    int slot = 0;
    size_t n = 0;
    // (a missing block or nsamp = 0 at the root is the root's error -- both together its END OF STREAM, -ENODATA on every rank -- and
    // travels through the header: the other ranks learn of it)
    if (int rc = rccl_distribute(h->rccl, (const float2 *)iq, mem == AMPS_MEM_HOST, nsamp, root, mode, h->stream, &blk, &slot, &n)) return rc;
    const int rc = amps_recc_push_wideband(h, (const float *)blk, n, AMPS_MEM_DEVICE);
    const int rc2 = rccl_block_consumed(h->rccl, slot, h->stream);
    if (npushed) *npushed = n;
    return rc ? rc : rc2;
}

int amps_recc_push_wideband_bcast(amps_recc_t *h, const float *iq, size_t nsamp, int mem, int root)
{
    return amps_recc_push_wideband_dist(h, iq, nsamp, mem, root, AMPS_RECC_DIST_BROADCAST, nullptr);
}

int amps_recc_drain_gather(amps_recc_t *h, amps_recc_burst_t *out, size_t cap, size_t *nout, int root)
{
    if (!h || !nout) return -EINVAL;
    *nout = 0;
    if (h->rccl.dead) return -ENOTCONN;
    if (!h->rccl.comm) return -ENOSYS;
    if (root < 0 || root >= h->rccl.nranks) return -EINVAL;      // (the same verdict on every rank: nobody is left alone in the collective)
    if (!out) cap = 0;
    HIP_TRY(hipSetDevice(h->device));
    // The handle's kernels wait for the data collectives: if a peer has gone, they never start, and an unbounded drain would sit behind
    // them for ever.  So the wait for the stream is bounded here (then the drain below finds it idle); on expiry the communicator is
    // aborted and the peers run into their own bound.
    if (int rc = rccl_wait(h->rccl, h->stream)) return rc;
    // this rank's own list first; whatever it returns, the rank then takes part in the collective (the others are waiting in it) and
    // tells them through the status word: bit 0 = its list overflowed, bit 1 = its drain failed
    std::vector<amps_recc_burst_t> mine(h->cfg.max_bursts);
    size_t n = 0;
    const int lrc = amps_recc_drain(h, mine.data(), mine.size(), &n);
    const uint32_t st = lrc == 0 ? 0u : lrc == -ENOSPC ? 1u : 2u;
    if (st & 2u) n = 0;
    std::vector<std::vector<uint8_t>> all;
    uint32_t st_or = 0;
    if (int rc = rccl_gather_records(h->rccl, mine.data(), (uint32_t)n, st, sizeof(amps_recc_burst_t), root, &all, &st_or)) return rc;
    bool truncated = false;
    if (h->rccl.rank == root) {
        struct Key { uint64_t k; const amps_recc_burst_t *r; };
        std::vector<Key> keys;
        for (const auto &v : all) {
            const amps_recc_burst_t *r = (const amps_recc_burst_t *)v.data();
            for (size_t i = 0; i < v.size() / sizeof(amps_recc_burst_t); i++)
                keys.push_back({ ((uint64_t)r[i].channel << CAPQ_POS_BITS) | (r[i].position & ((1ull << CAPQ_POS_BITS) - 1)), r + i });
        }
        std::sort(keys.begin(), keys.end(), [](const Key &x, const Key &y) { return x.k < y.k; });   // a channel belongs to one rank: no ties
        const size_t k = std::min(keys.size(), cap);
        for (size_t i = 0; i < k; i++) std::memcpy(&out[i], keys[i].r, sizeof(amps_recc_burst_t));
        *nout = k;
        truncated = keys.size() > cap;
    }
    if (st_or & 2u) return (lrc && lrc != -ENOSPC) ? lrc : -EIO;
    return ((st_or & 1u) || truncated) ? -ENOSPC : 0;
}

int amps_recc_set_xlate(amps_recc_t *h, const amps_recc_xlate_cfg_t *x)
{
    if (!h || !x || x->struct_size != sizeof(amps_recc_xlate_cfg_t)) return -EINVAL;
    if (!h->carry[0]) return -ENOSYS;                       // the IQ seam must be configured
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (x->decim == 0) { xlate_destroy(h->xl); return 0; }
    // defaults = the flow graph's values (grc/recctest.grc:115-155, 889-937)
    const double gain = x->gain != 0.0 ? x->gain : 3.0;
    const double cutoff = x->cutoff_hz != 0.0 ? x->cutoff_hz : 10e3;
    const double width = x->width_hz != 0.0 ? x->width_hz : 4.5e3;
    if (!(x->rate_hz > 0.0) || !(cutoff > 0.0) || !(width > 0.0)) return -EINVAL;
    // the filtered stream must arrive at the symbol rate the handle was built for
    const double out_rate = x->rate_hz / x->decim;
    if (std::fabs(out_rate - 20e3 * h->sps) > 1e-6 * out_rate) return -EINVAL;
    std::vector<float> taps = xlate_design_taps(gain, x->rate_hz, cutoff, width);
    return xlate_create(h->xl, h->C, x->decim, h->cfg.max_samples_per_push, x->rate_hz, x->center_hz, taps, h->stream);
}

int amps_recc_push_raw(amps_recc_t *h, const float *iq, size_t ld, size_t nsamp, int mem)
{
    if (!h) return -EINVAL;
    if (!h->xl.enabled) return -ENOSYS;
    if (h->chz.enabled && h->chz.groups > 1) return -ENOSYS;
    if (nsamp == 0) return 0;
    if (!iq || ld < nsamp) return -EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    const float2 *f = nullptr;
    uint64_t fld = 0;
    uint32_t nout = 0;
    int rc;
    {
        SpanGuard g(h, T_XLATE, nsamp);
        rc = xlate_run(h->xl, (const float2 *)iq, ld, nsamp, mem, h->stream, &f, &fld, &nout);
    }
    if (rc) return rc;
    if (int rc2 = debug_sync(h, "xlate")) return rc2;
    if (nout == 0) return 0;
    return run_iq_device(h, f, fld, nout);
}

int amps_recc_debug_xlate(amps_recc_t *h, const float *iq, size_t ld, size_t nsamp, int mem, float *out, size_t out_ld, size_t *nout)
{
    if (!h || !iq || !out || !nout || ld < nsamp) return -EINVAL;
    if (!h->xl.enabled) return -ENOSYS;
    HIP_TRY(hipSetDevice(h->device));
    const float2 *f = nullptr;
    uint64_t fld = 0;
    uint32_t n = 0;
    int rc = xlate_run(h->xl, (const float2 *)iq, ld, nsamp, mem, h->stream, &f, &fld, &n);
    if (rc) return rc;
    *nout = n;
    if (n > out_ld) return -E2BIG;
    if (n)
        HIP_TRY(hipMemcpy2DAsync(out, out_ld * sizeof(float2), f, fld * sizeof(float2), (size_t)n * sizeof(float2), h->C,
                                 hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int amps_recc_refchain_symbols(amps_recc_t *h, const float *iq, size_t ld, size_t nsamp, int mem,
                               uint8_t *symbols_out, size_t sym_ld, uint32_t *nsym_out)
{
    if (!h || !symbols_out || !nsym_out || (nsamp && (!iq || ld < nsamp))) return -EINVAL;
    if (!h->cfg.max_samples_per_push) return -ENOSYS;
    if (h->sps != 10) return -EINVAL;                       // the flow graph's omega = 10 samples per symbol
    if (nsamp > h->cfg.max_samples_per_push) return -E2BIG;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    if (!h->ref.ready) { if (int rc = ref_create(h->ref, h->C, h->cfg.max_samples_per_push, s)) return rc; }
    if (sym_ld < h->ref.sym_cap && sym_ld < nsamp / 9 + 16) return -EINVAL;
    const float2 *d = (const float2 *)iq;
    uint64_t dld = ld;
    if (mem == AMPS_MEM_HOST && nsamp) {
        if (!h->ref.stage && dev_alloc(&h->ref.stage, (size_t)h->C * h->cfg.max_samples_per_push)) return -ENOMEM;
        dld = nsamp;
        HIP_TRY(hipMemcpy2D(h->ref.stage, dld * sizeof(float2), iq, ld * sizeof(float2), nsamp * sizeof(float2), h->C, hipMemcpyHostToDevice));
        d = h->ref.stage;
    }
    if (int rc = ref_run(h->ref, d, dld, (uint32_t)nsamp, s)) return rc;
    HIP_TRY(hipMemcpyAsync(nsym_out, h->ref.nsym, sizeof(uint32_t) * h->C, hipMemcpyDeviceToHost, s));
    const size_t w = std::min<size_t>(sym_ld, h->ref.sym_cap);
    HIP_TRY(hipMemcpy2DAsync(symbols_out, sym_ld, h->ref.syms, h->ref.sym_cap, w, h->C, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int amps_recc_refchain_tables(amps_recc_t *h, float *atan258, float *mmse1032)
{
    if (!h || !atan258 || !mmse1032) return -EINVAL;
    if (!h->cfg.max_samples_per_push) return -ENOSYS;
    HIP_TRY(hipSetDevice(h->device));
    if (!h->ref.ready) { if (int rc = ref_create(h->ref, h->C, h->cfg.max_samples_per_push, h->stream)) return rc; }
    std::memcpy(atan258, h->ref.atan_host.data(), sizeof(float) * 258);
    std::memcpy(mmse1032, h->ref.mmse_host.data(), sizeof(float) * 129 * 8);
    return 0;
}

int amps_recc_wait_event(amps_recc_t *h, void *hip_event)
{
    if (!h || !hip_event) return -EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamWaitEvent(h->stream, (hipEvent_t)hip_event, 0));
    return 0;
}

int amps_recc_record_event(amps_recc_t *h, void *hip_event)
{
    if (!h || !hip_event) return -EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventRecord((hipEvent_t)hip_event, h->stream));
    return 0;
}

int amps_recc_drain_begin(amps_recc_t *h)
{
    if (!h) return -EINVAL;
    STALE_CHECK(h);
    if (h->open_buf >= 0) return -EBUSY;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int b = h->cur_buf;
    // the list's header is already on its way to host memory: the last capture workgroup of every push writes it
    HIP_TRY(hipEventRecord(h->drain_event, s));
    h->open_buf = b;
    h->open_untouched = true;
    select_record_list(h, b ^ 1);           // later pushes append to the other list
    if (!h->list_clean[b ^ 1]) {            // drained twice with no push in between: nobody has cleared it yet
        HIP_TRY(hipMemsetAsync(h->nrecords_buf[b ^ 1], 0, LIST_WORDS * sizeof(uint32_t), s));
        h->list_clean[b ^ 1] = true;
    }
    return 0;
}

static int drain_end_impl(amps_recc_t *h, amps_recc_burst_t *out, uint8_t *bursts_out, size_t cap, size_t *nout);
int amps_recc_drain_end(amps_recc_t *h, amps_recc_burst_t *out, size_t cap, size_t *nout) { return drain_end_impl(h, out, nullptr, cap, nout); }

int amps_recc_drain_bursts(amps_recc_t *h, amps_recc_burst_t *out, uint8_t *bursts_out, size_t cap, size_t *nout)
{
    if (!h || !nout || !bursts_out) return -EINVAL;
    *nout = 0;
    if (!(h->cfg.flags & AMPS_RECC_FLAG_KEEP_BURSTS)) return -ENOSYS;
    if (h->open_buf >= 0) return -EBUSY;
    int rc = amps_recc_drain_begin(h);
    if (rc) return rc;
    return drain_end_impl(h, out, bursts_out, cap, nout);
}

static int drain_end_impl(amps_recc_t *h, amps_recc_burst_t *out, uint8_t *bursts_out, size_t cap, size_t *nout)
{
    if (!h || !nout) return -EINVAL;
    *nout = 0;
    if (h->open_buf < 0) return -EINVAL;
    const int b = h->open_buf;
    volatile uint32_t *hdr = h->hdr_host + HDR_STRIDE * b;
    // an error below still CLOSES the split drain (and empties the list's header): a handle must not answer -EBUSY for ever
    // because one drain failed
    auto fail = [&](int rc) { hdr[0] = 0u; hdr[1] = 0u; h->open_buf = -1; return rc; };
    if (hipSetDevice(h->device) != hipSuccess) return fail(-EIO);
    if (int wrc = sync_event(h, h->drain_event)) return fail(wrc);   // everything enqueued before drain_begin is done; later pushes may still run (bounded behind a collective)
    if (h->rccl.stale) return fail(-ESTALE);
    collect_spans(h);
    uint32_t n = hdr[0];
    const uint32_t st = hdr[1];
    if (check_header_enabled() && h->open_untouched) {
        uint32_t dev[2] = { 0u, 0u };
        if (hipMemcpy(dev, h->nrecords_buf[b], sizeof(dev), hipMemcpyDeviceToHost) != hipSuccess) return fail(-EIO);
        if (dev[0] != n || dev[1] != st) {
            std::fprintf(stderr, "amps_recc: published list header {%u, %u} differs from the device counters {%u, %u}\n", n, st, dev[0], dev[1]);
            return fail(-EIO);
        }
    }
    hdr[0] = 0u; hdr[1] = 0u;               // empty until a capture kernel publishes into it again (the list is not current now)
    int rc = 0;
    if (st & 1u) rc = -EOVERFLOW;
    if ((st & (2u | 4u)) || n > h->cfg.max_bursts) { rc = -ENOSPC; }
    if (n > h->cfg.max_bursts) n = h->cfg.max_bursts;
    if (n) {
        // the records are already in host memory (written by the capture kernel, visible after the event above);
        // order by (channel, position) through compact 16-byte keys, then gather once into the caller's buffer
        struct Key { uint64_t k; uint32_t i; };
        std::vector<Key> keys(n);
        // (packed: PACKED_RECORD_BYTES each, the bit arrays as bits -- recc_decode.hip.h; channel and position sit in the first 16 bytes)
        const uint8_t *r = (const uint8_t *)h->rec_host_buf[b];
        for (uint32_t i = 0; i < n; i++) {
            uint32_t ch; uint64_t pos;
            std::memcpy(&ch, r + (size_t)i * PACKED_RECORD_BYTES + offsetof(amps_recc_burst_t, channel), 4);
            std::memcpy(&pos, r + (size_t)i * PACKED_RECORD_BYTES + offsetof(amps_recc_burst_t, position), 8);
            keys[i] = { ((uint64_t)ch << CAPQ_POS_BITS) | (pos & ((1ull << CAPQ_POS_BITS) - 1)), i };
        }
        std::sort(keys.begin(), keys.end(), [](const Key &x, const Key &y) { return x.k < y.k; });
        size_t k = std::min<size_t>(n, cap);
        if (out) for (size_t i = 0; i < k; i++) expand_packed_record(&out[i], r + (size_t)keys[i].i * PACKED_RECORD_BYTES);
        if (out && h->chz.enabled && h->chz.groups > 1)               // rows of a channel group -> channel numbers of the band selection
            for (size_t i = 0; i < k; i++) out[i].channel = out[i].channel < h->chz.row2chan.size() ? h->chz.row2chan[out[i].channel] : out[i].channel;
        if (bursts_out && h->bsym_host_buf[b])
            for (size_t i = 0; i < k; i++)
                expand_packed_burst(bursts_out + i * AMPS_RECC_CAPTURE_SYMS, h->bsym_host_buf[b] + (size_t)keys[i].i * PACKED_BURST_BYTES);
        *nout = k;
        if (n > cap) rc = -ENOSPC;
    }
    // the list's device counters are cleared by the next push (front_housekeeping_args) or by the next drain_begin
    h->open_buf = -1;
    return rc;
}

int amps_recc_drain(amps_recc_t *h, amps_recc_burst_t *out, size_t cap, size_t *nout)
{
    if (!h || !nout) return -EINVAL;
    *nout = 0;
    if (h->open_buf >= 0) return -EBUSY;     // finish the split drain first
    int rc = amps_recc_drain_begin(h);
    if (rc) return rc;
    return amps_recc_drain_end(h, out, cap, nout);
}

int amps_recc_debug_exact_slice(int form, int sps, const uint32_t *in, uint32_t *out)
{
    if (!in || !out) return -EINVAL;
    if (form == 1) {
        if (sps == 3) out[0] = exact_slice_word3(in[0], in[1], in[2], in[3], in[4], in[5], out[1], out[2]);
        else if (sps == 2) out[0] = exact_slice_word2(in[0], in[1], in[2], in[3], in[4], in[5], out[1], out[2]);
        else return -EINVAL;
        return 0;
    }
    if (form != 0) return -EINVAL;
    switch (sps) {
    case 2: out[0] = exact_slice_word<2>(in[0], in[1], in[2]); break;
    case 3: out[0] = exact_slice_word<3>(in[0], in[1], in[2]); break;
    case 4: out[0] = exact_slice_word<4>(in[0], in[1], in[2]); break;
    case 5: out[0] = exact_slice_word<5>(in[0], in[1], in[2]); break;
    case 6: out[0] = exact_slice_word<6>(in[0], in[1], in[2]); break;
    case 8: out[0] = exact_slice_word<8>(in[0], in[1], in[2]); break;
    case 10: out[0] = exact_slice_word<10>(in[0], in[1], in[2]); break;
    case 12: out[0] = exact_slice_word<12>(in[0], in[1], in[2]); break;
    default: return -EINVAL;
    }
    return 0;
}

int amps_recc_debug_demod(amps_recc_t *h, const float *iq, size_t nsamp, int mem, float *demod, float *soft, uint8_t *hard)
{
    if (!h || !iq) return -EINVAL;
    if (!h->carry[0]) return -ENOSYS;
    if (nsamp == 0 || nsamp > h->cfg.max_samples_per_push) return -E2BIG;
    int rc = amps_recc_reset(h);
    if (rc) return rc;
    const size_t P = (nsamp / 64) * 64;
    if (dev_alloc(&h->dbg_d, nsamp) || dev_alloc(&h->dbg_S, nsamp)) return -ENOMEM;
    // channel 0 only: replicate the single stream on every channel row is not needed, rows other than 0 read garbage-free zeros
    std::vector<float> zeros;
    const float *src = iq;
    size_t ld = nsamp;
    std::vector<float> tmp;
    if (h->C > 1) {
        if (mem != AMPS_MEM_HOST) { rc = -EINVAL; goto done; }
        tmp.assign((size_t)h->C * nsamp * 2, 0.f);
        std::memcpy(tmp.data(), iq, nsamp * 2 * sizeof(float));
        src = tmp.data();
    }
    rc = amps_recc_push_iq(h, src, ld, nsamp, mem);
    if (!rc) {
        if (hipStreamSynchronize(h->stream) != hipSuccess) rc = -EIO;
    }
    if (!rc && P) {
        if (demod && hipMemcpy(demod, h->dbg_d, P * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = -EIO;
        if (soft && hipMemcpy(soft, h->dbg_S, P * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = -EIO;
        if (hard) {
            std::vector<uint64_t> ring(h->ring_words);
            if (hipMemcpy(ring.data(), h->gring, sizeof(uint64_t) * h->ring_words, hipMemcpyDeviceToHost) != hipSuccess) rc = -EIO;
            for (size_t i = 0; i < P; i++) hard[i] = (uint8_t)((ring[(i >> 6) & (h->ring_words - 1)] >> (i & 63)) & 1ull);
        }
    }
done:
    (void)hipFree(h->dbg_d); (void)hipFree(h->dbg_S);
    h->dbg_d = nullptr; h->dbg_S = nullptr;
    {
        int rc2 = amps_recc_reset(h);
        if (!rc) rc = rc2;
    }
    return rc;
}

int amps_recc_debug_channelize(amps_recc_t *h, const float *iq, size_t nsamp, int mem, float *out, size_t out_ld, size_t *nframes)
{
    if (!h || !iq || !out || !nframes) return -EINVAL;
    if (!h->chz.enabled) return -ENOSYS;
    HIP_TRY(hipSetDevice(h->device));
    const float2 *chan_iq = nullptr;
    uint64_t ld = 0;
    uint32_t nout = 0;
    int rc = channelizer_run(h->chz, (const float2 *)iq, nsamp, mem, h->stream, &chan_iq, &ld, &nout);
    if (rc) return rc;
    *nframes = nout;
    if (nout > out_ld) return -E2BIG;
    if (nout)
        HIP_TRY(hipMemcpy2DAsync(out, out_ld * sizeof(float2), chan_iq, ld * sizeof(float2), nout * sizeof(float2), h->C,
                                 hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int amps_recc_set_origin(amps_recc_t *h, uint64_t first_sample)
{
    if (!h || (first_sample & 63u) || first_sample >= (1ull << CAPQ_POS_BITS)) return -EINVAL;
    if (!h->carry[0]) return -ENOSYS;
    if (h->n_done != 0 || h->r_prev != 0 || h->origin_locked) return -EBUSY;     // only on a fresh or reset handle
    h->n_done = first_sample;
    h->origin = first_sample;
    return 0;
}

int amps_recc_set_timing(amps_recc_t *h, int mode)
{
    if (!h || mode < AMPS_RECC_TIMING_OFF || mode > AMPS_RECC_TIMING_DOMINANT_SAMPLED) return -EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    collect_spans(h);
    h->timing_mode = mode;
    h->dominant_tick = 0;
    h->timing = mode != AMPS_RECC_TIMING_OFF;
    h->rccl.timing = h->timing;
    return 0;
}

int amps_recc_get_timing(amps_recc_t *h, amps_recc_timing_t *t, int reset)
{
    if (!h || !t) return -EINVAL;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    collect_spans(h);
    std::memset(t, 0, sizeof(*t));
    t->struct_size = sizeof(*t);
    t->launches_front = h->launches_front;
    t->ms_front = h->ms[T_FRONT];
    t->ms_channelizer = h->ms[T_CHANNELIZER];
    t->launches_channelizer = h->launches_chz;
    t->ms_resolve = h->ms[T_RESOLVE];
    t->ms_decode = h->ms[T_DECODE];
    t->ms_carry = h->ms[T_CARRY];
    t->ms_symbols = h->ms[T_SYMBOLS];
    t->ms_xlate = h->ms[T_XLATE];
    t->samples_front = h->samples_front;
    if (reset) {
        for (double &m : h->ms) m = 0;
        h->launches_front = 0;
        h->launches_chz = 0;
        h->samples_front = 0;
    }
    return 0;
}

// ---- BCH(63,51) shortened: batch encode / decode on the device
// scratch buffers live in the handle and only ever grow: a call costs one launch, its copies and one synchronise
static int bch_grow(uint8_t **p, size_t *cap, size_t need)
{
    if (need <= *cap) return 0;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    size_t want = need < 4096 ? 4096 : need + need / 2;
    if (hipMalloc((void **)p, want) != hipSuccess) return -ENOMEM;
    *cap = want;
    return 0;
}
static int bch_stage_in(amps_recc_t *h, const uint8_t *in, size_t nin, int mem, const uint8_t **din)
{
    if (mem == AMPS_MEM_DEVICE) { *din = in; return 0; }
    if (int rc = bch_grow(&h->bch_in, &h->bch_in_cap, nin)) return rc;
    if (hipMemcpyAsync(h->bch_in, in, nin, hipMemcpyHostToDevice, h->stream) != hipSuccess) return -EIO;
    *din = h->bch_in;
    return 0;
}

int amps_bch_encode_words(amps_recc_t *h, const uint8_t *msg, size_t nwords, int k, int mem, uint8_t *codewords)
{
    if (!h || k < 1 || k > 51 || (nwords && (!msg || !codewords))) return -EINVAL;
    if (nwords == 0) return 0;
    HIP_TRY(hipSetDevice(h->device));
    const uint8_t *din = nullptr;
    if (int rc = bch_stage_in(h, msg, nwords * k, mem, &din)) return rc;
    const size_t nout = nwords * (size_t)(k + 12);
    if (int rc = bch_grow(&h->bch_out, &h->bch_out_cap, nout)) return rc;
    hipLaunchKernelGGL(bch_encode_words_kernel, dim3((unsigned)std::min<size_t>((nwords + 255) / 256, 4096)), dim3(256), 0, h->stream,
                       din, (uint32_t)nwords, k, h->bch_out);
    HIP_TRY(hipMemcpyAsync(codewords, h->bch_out, nout, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int amps_bch_decode_words(amps_recc_t *h, const uint8_t *codewords, size_t nwords, int k, int mem, uint8_t *msg, uint8_t *valid, uint8_t *nerrors)
{
    if (!h || k < 1 || k > 51 || (nwords && (!codewords || !msg || !valid))) return -EINVAL;
    if (nwords == 0) return 0;
    HIP_TRY(hipSetDevice(h->device));
    const uint8_t *din = nullptr;
    if (int rc = bch_stage_in(h, codewords, nwords * (size_t)(k + 12), mem, &din)) return rc;
    if (int rc = bch_grow(&h->bch_out, &h->bch_out_cap, nwords * (size_t)k)) return rc;
    if (nwords > h->bch_n_cap) {
        size_t c1 = h->bch_n_cap, c2 = h->bch_n_cap;
        if (bch_grow(&h->bch_val, &c1, nwords) || bch_grow(&h->bch_err, &c2, nwords)) { h->bch_n_cap = 0; return -ENOMEM; }
        h->bch_n_cap = std::min(c1, c2);
    }
    hipLaunchKernelGGL(bch_decode_words_kernel, dim3((unsigned)std::min<size_t>((nwords + 255) / 256, 4096)), dim3(256), 0, h->stream,
                       din, (uint32_t)nwords, k, h->bch_out, h->bch_val, h->bch_err);
    HIP_TRY(hipMemcpyAsync(msg, h->bch_out, nwords * k, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(valid, h->bch_val, nwords, hipMemcpyDeviceToHost, h->stream));
    if (nerrors) HIP_TRY(hipMemcpyAsync(nerrors, h->bch_err, nwords, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// ---- reply generation: handle_response / handle_registration / handle_origination
// (lib/recc_decode_impl.cc:181-272) with the TX word builders of lib/amps_packet.cc:26-95.
// Host integer code, a few dozen byte stores per burst.
static void put_bits(uint8_t *o, int n, uint64_t v) { for (int i = n - 1; i >= 0; i--) { o[i] = (uint8_t)(v & 1u); v >>= 1; } }
static void word1(uint8_t *w, bool multi, unsigned dcc, uint64_t min1)
{
    w[0] = 0; w[1] = multi; w[2] = (dcc >> 1) & 1u; w[3] = dcc & 1u; put_bits(w + 4, 24, min1);
}
static void word2_general(uint8_t *w, uint64_t min2, unsigned msg_type, unsigned ordq, unsigned order)
{
    w[0] = 1; w[1] = 0; w[2] = 1; w[3] = 1; put_bits(w + 4, 10, min2); w[14] = 0;
    put_bits(w + 15, 5, msg_type); put_bits(w + 20, 3, ordq); put_bits(w + 23, 5, order);
}
static void word2_voice(uint8_t *w, unsigned scc, uint64_t min2, unsigned vmac, unsigned chan)
{
    w[0] = 1; w[1] = 0; w[2] = (scc >> 1) & 1u; w[3] = scc & 1u; put_bits(w + 4, 10, min2);
    put_bits(w + 14, 3, vmac); put_bits(w + 17, 11, chan);
}
static void fvc_general(uint8_t *w, unsigned pscc, unsigned msg_type, unsigned ordq, unsigned order)
{
    std::memset(w, 0, 28);
    w[0] = 1; w[2] = 1; w[3] = 1; w[4] = (pscc >> 1) & 1u; w[5] = pscc & 1u;
    put_bits(w + 15, 5, msg_type); put_bits(w + 20, 3, ordq); put_bits(w + 23, 5, order);
}

int amps_recc_reply_words(const amps_recc_burst_t *b, amps_recc_reply_t *r)
{
    if (!b || !r) return -EINVAL;
    std::memset(r, 0, sizeof(*r));
    const unsigned DCC = 0, SCC = 1;     // GLOBAL_DCC_SHORT, GLOBAL_SCC (lib/amps_packet.h:13-14)
    const int STREAM_BOTH = 3;           // lib/amps_packet.h:33 (the A/B choice at :240-245 is overridden at :247)
    switch (b->msg_class) {
    case AMPS_MSG_REGISTRATION:          // :181-190 order confirmation = audit order 7
        r->has_focc = 1; r->focc_stream = STREAM_BOTH; r->focc_nwords = 2;
        word1(r->focc_word1, true, DCC, b->a_MIN1);
        word2_general(r->focc_word2, b->b_MIN2, 0, 0, 7);
        break;
    case AMPS_MSG_PAGE_RESPONSE:         // :195-222 voice channel 355, alert on the FVC
        r->has_focc = 1; r->focc_stream = STREAM_BOTH; r->focc_nwords = 2;
        word1(r->focc_word1, true, DCC, b->a_MIN1);
        word2_voice(r->focc_word2, SCC, b->b_MIN2, 0, 355);
        r->has_fvc = 1; r->fvc_count = 1; r->fvc_repeat = 35;
        fvc_general(r->fvc_word1, SCC, 0, 0, 1);
        r->has_mutes = 1; r->fvc_mute = 0; r->audio_mute = 1;
        break;
    case AMPS_MSG_ORIGINATION:           // :236-272 voice channel 356 (or reorder 9 for a leading '0')
        r->has_focc = 1; r->focc_stream = STREAM_BOTH; r->focc_nwords = 2;
        word1(r->focc_word1, true, DCC, b->a_MIN1);
        if (b->dialed[0] == '0') word2_general(r->focc_word2, b->b_MIN2, 0, 0, 9);
        else word2_voice(r->focc_word2, SCC, b->b_MIN2, 0, 356);
        r->has_mutes = 1; r->fvc_mute = 1; r->audio_mute = 0;
        r->has_command = 1;
        std::snprintf(r->command, sizeof(r->command), "page %.*s", 32, b->dialed);
        break;
    default: break;
    }
    return 0;
}

} // extern "C"
