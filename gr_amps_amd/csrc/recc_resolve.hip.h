// recc_resolve.hip.h -- per-channel ordering of trigger hits, burst hold-off, capture and decode.
//
// What the reference does serially inside recc_impl::work once memmem has hit
// (lib/recc_impl.cc:121-139: wait until more than 3374 symbols follow, publish them, resume the
// search after the captured region) is ONE small kernel behind the streaming front kernel, one workgroup per channel
// (everything here is channel-local):
//   resolve   walks the channel's ordered hit list, drops hits that fall inside an accepted burst (hold-off = 74+3374
//             symbols), picks the centre of the run of matching sample phases as the symbol timing, and either accepts
//             the capture or parks it as "pending" until its tail has been received;
//   capture   the workgroup's first CAP_WAVES waves each take accepted captures in turn: gather the 3374 slicer bits at the
//             chosen phase from the channel's HBM bit ring, run the recc_decode core (recc_decode.hip.h), append the record.
// (Round 2 had a capture queue and a second kernel of one workgroup per capture: 2048 workgroups to dispatch, a launch and
// ~25 us per push for the same work.  That form is kept for handles with few channels: resolve_uses_queue.)  Latency-bound bookkeeping on kilobytes; the HBM-bound work is in recc_front.hip.h.
#pragma once
#include "recc_decode.hip.h"
#include "recc_bits.hip.h"

namespace amps {

// one packed key (channel, position) for the capture queue and the host's sort: 2^44 samples per channel stream (2.8 years at 200 ksps), 2^20 channels
constexpr int CAPQ_POS_BITS = 44;

struct ResolveArgs {
    const uint64_t *det;       // [C][max_chunks][det_cap]
    const uint32_t *detcount;  // [C][max_chunks]
    uint32_t max_chunks, det_cap;
    uint32_t tiles_per_channel, span;   // segment geometry of the front launch (recc_front.hip.h)
    uint32_t sps;
    uint64_t n_proc;           // absolute samples processed after this push
    uint64_t *next_allowed;    // [C]
    uint64_t *pending;         // [C], ~0 = none (holds n_c)
    uint32_t *status;          // bit 1: capture queue overflow (queue form only), bit 2: record list overflow
    // capture + decode
    const uint64_t *gring;
    uint32_t ring_mask, ring_words;
    uint32_t cap_words;        // ring words one capture spans at most: sizes the dynamic LDS (resolve_dyn_lds)
    amps_recc_burst_t *records;
    uint32_t *nrecords;        // atomic
    uint32_t rec_cap;
    uint32_t majority;         // decode mode (AMPS_RECC_FLAG_MAJORITY)
    uint32_t track;            // timing tracking inside a capture (off: AMPS_RECC_FLAG_FIXED_TIMING)
    uint8_t *burst_syms;       // optional [rec_cap][3374]: the captured symbols of record `slot` (AMPS_RECC_FLAG_KEEP_BURSTS)
    unsigned long long *done_blocks;   // [1 + DONE_GROUPS] {workgroups of this launch that have finished, record slots they reserved} (the last one publishes the header; see DONE_GROUPS)
    uint32_t *hdr_host;        // mapped pinned {nrecords, status} of the record list: what a drain reads, no copy on the stream
    // queue form (few channels, see resolve_uses_queue): accepted captures go to a queue and recc_capture_kernel decodes them
    uint64_t *capq;            // [capq_cap] (channel << CAPQ_POS_BITS | n_c), or null: decode in this kernel
    uint32_t *capq_count;      // atomic; cleared by the streaming kernel's housekeeping
    uint32_t capq_cap;
    // search stage (recc_resolve_kernel<..., SEARCH = samples per symbol>, the wideband seam since round 6): the bit-domain trigger search of
    // recc_bits_kernel runs INSIDE this kernel, a quarter of the channel's push per wave, its hits go to LDS lists instead of det /
    // detcount -- one launch and one kernel boundary fewer per step
    uint32_t search_P;         // samples of this push (multiple of 64); the stream position of its first sample is n_proc - search_P
    uint32_t search_tol;       // accepted mismatching trigger symbols (cfg.sync_tolerance)
    uint32_t *zero1, *zero2;   // the launch's housekeeping (front_housekeeping): the capture queue count / the idle record list's words
    unsigned long long *tl;    // -DRESOLVE_TIMELINE builds: [C][24] s_memtime stamps of every workgroup's thread 0 (scripts/resolve_timeline.py)
};
#ifdef RESOLVE_TIMELINE
#define RTL(k) do { if (a.tl && tid == 0) a.tl[(size_t)blockIdx.x * 24 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RTL(k) do { } while (0)
#endif

constexpr int CAP_WAVES = 4;               // waves of a workgroup that decode captures side by side
// dynamic LDS of the kernel: per decoding wave one DecodeCore and the ring words of one capture
__host__ __device__ constexpr uint32_t resolve_cap_stride(uint32_t cap_words) { return (uint32_t)((sizeof(DecodeCore) + 7) / 8) + cap_words; }   // in 8-byte words
inline uint32_t resolve_cap_words(uint32_t sps) { return (AMPS_RECC_CAPTURE_SYMS * sps + sps + capture_lead(sps) + AMPS_TRACK_BLOCKS) / 64 + 4; }   // + one dword of read-ahead for the funnel shift
inline size_t resolve_dyn_lds(uint32_t sps) { return (size_t)CAP_WAVES * resolve_cap_stride(resolve_cap_words(sps)) * 8; }

// One accepted capture (channel c, symbol-timing position nc), by all 64 lanes of one wave: ring words -> Manchester bits ->
// BCH -> parsed record, staged in the wave's LDS scratch (capture_store_wave writes it out once its slot is known).
#ifdef RESOLVE_TIMELINE
struct RtlStamps {
    unsigned long long *dst;
    __device__ __forceinline__ void mark(int k) { if (dst) dst[k] = __builtin_amdgcn_s_memtime(); }
};
#else
typedef NoTl RtlStamps;
#endif
__device__ __forceinline__ void capture_gather_wave(const ResolveArgs &a, uint32_t c, uint64_t nc, uint64_t *scratch, int lane, RtlStamps tl = RtlStamps())
{
    uint64_t *s_ring = scratch + (sizeof(DecodeCore) + 7) / 8;
    const uint64_t *ring = a.gring + (uint64_t)c * a.ring_words;
    const uint64_t w0 = capture_first_word(nc, a.sps);
    const int nw = (int)(((nc + (uint64_t)a.sps * (AMPS_RECC_CAPTURE_SYMS + 1) + AMPS_TRACK_BLOCKS + 32) >> 6) - w0) + 1;   // + the funnel shift's second dword
    // eight loads in flight per lane, not a load-wait-store loop of one round trip per 64 words (sps 10: nine of them)
    for (int i0 = 0; i0 < nw; i0 += 512) {
        uint64_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + 64 * u + lane;
            v[u] = i < nw ? ring[(w0 + (uint64_t)i) & a.ring_mask] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + 64 * u + lane;
            if (i < nw) s_ring[i] = v[u];
        }
    }
    WaveSync::sync();
    tl.mark(8);
}
template <bool TWO>
__device__ __forceinline__ void capture_decode_wave(const ResolveArgs &a, uint32_t c, uint64_t nc, uint64_t *scratch, int lane, RtlStamps tl = RtlStamps())
{
    DecodeCore &k = *(DecodeCore *)scratch;
    const uint64_t *s_ring = scratch + (sizeof(DecodeCore) + 7) / 8;
    const uint64_t w0 = capture_first_word(nc, a.sps);
    manchester_from_ring<WaveSync, TWO>(k, s_ring, nc, w0, a.sps, lane, a.track != 0, a.burst_syms != nullptr);
    tl.mark(0);
    decode_core_wave<WaveSync>(k, c, nc, nullptr, a.majority != 0, lane, tl);
}
__device__ __forceinline__ void capture_store_wave(const ResolveArgs &a, uint64_t nc, uint32_t slot, const uint64_t *scratch, int lane)
{
    const DecodeCore &k = *(const DecodeCore *)scratch;
    decode_core_store_packed(k, (uint32_t *)((uint8_t *)a.records + (size_t)slot * PACKED_RECORD_BYTES), lane);   // bits as bits over PCIe: expand_packed_record
    if (a.burst_syms) {
        // the kept blob (AMPS_RECC_FLAG_KEEP_BURSTS) crosses PCIe as BITS too (round 5): symbol i = bit i of PACKED_BURST_BYTES, two
        // dwords per lane instead of 53 single-byte stores; amps_recc_drain_bursts expands it to the 3374 bytes gr::amps::recc publishes
        const uint64_t *s_ring = scratch + (sizeof(DecodeCore) + 7) / 8;
        const uint64_t w0 = capture_first_word(nc, a.sps);
        uint32_t *dst = (uint32_t *)(a.burst_syms + (uint64_t)slot * PACKED_BURST_BYTES);
        for (int dw = lane; dw < PACKED_BURST_BYTES / 4; dw += 64) {
            uint32_t word = 0u;
            for (int j = 0; j < 32; j++) {
                const int i = 32 * dw + j;
                if (i < AMPS_RECC_CAPTURE_SYMS) {
                    const int kb = i >> 1, blk = kb < 7 + AMPS_RECC_WORD_BITS ? 1 : 2 + (kb - 7 - AMPS_RECC_WORD_BITS) / AMPS_RECC_WORD_BITS;   // tracking block of the symbol's bit
                    const uint64_t n = (uint64_t)((int64_t)nc + (int64_t)a.sps * (i + 1) + k.dly[blk]);
                    word |= (uint32_t)((s_ring[(n >> 6) - w0] >> (n & 63)) & 1ull) << j;
                }
            }
            dst[dw] = word;
        }
    }
}

// Same-address device-scope atomics are performed at the memory side of the eight XCDs' L2s, one every ~20 ns, and vector memory
// returns in order per CU (measured: 832 workgroups counting themselves done on ONE counter cost 4 us per launch; a slot atomic
// issued in front of a capture's ring loads held the ring words back 7 us).  So the workgroups count themselves done on
// DONE_GROUPS group counters (channel mod DONE_GROUPS) and only the last of a group steps the top counter; and a workgroup reserves
// the record slots of a whole batch of captures with one atomic, issued behind the first ring loads and consumed after the decode.
constexpr uint32_t DONE_GROUPS = 32;       // done_blocks[0] = top counter, [1 + g] = group g
// A workgroup counts itself done, and in the SAME atomic says how many record slots it reserved in this launch: the counters are
// 64 bits, {workgroups done (low word), slots reserved (high word)}.  The workgroup that finishes last therefore knows the launch's
// record count from the value its own read-modify-write returns -- atomicity on ONE address, no ordering between two addresses
// needed -- adds it to the list's published count (a plain word only ever touched by that one thread of a launch, and launches are
// stream-ordered) and writes the header.  Rounds 3-4 had the last workgroup read `nrecords` after its done-count with relaxed
// atomics, which holds on this hardware (device-scope atomics are performed in issue order at the memory side of the L2s, DESIGN.md
// 4.4) but is not a guarantee of the memory model; release / acquire on the old counters would have
// been: measured +15 us per launch (resolve + capture + decode 0.0317 -> 0.0468 ms at 416 bursts, profiles/r05/done_counters.txt),
// because a release waits for the counting wave's 728-byte record stores to cross PCIe.  The packed count needs no ordering at all, so
// the read-modify-writes stay RELAXED (AMPS_RESOLVE_DONE_ORDER: an acquire on them measured +2.5 us).  The status word is read
// relaxed as well: the one bit that can be set inside the publishing launch -- 4, record list overflow -- is implied by the count the
// host compares with max_bursts anyway (drain_end_impl), bit 2 (capture queue overflow) is set by the kernel in FRONT of the one that
// publishes in the queue form, bit 1 by the search kernel in front of both: kernel boundaries order those.
// AMPS_RECC_CHECK_HEADER (on in the test suite) still cross-checks the header against a copy of the device counters.
#ifndef AMPS_RESOLVE_DONE_ORDER
#define AMPS_RESOLVE_DONE_ORDER __ATOMIC_RELAXED
#endif
__device__ __forceinline__ unsigned long long count_done(unsigned long long *p, uint32_t done, unsigned long long reserved)
{
    return __hip_atomic_fetch_add(p, (reserved << 32) | done, AMPS_RESOLVE_DONE_ORDER, __HIP_MEMORY_SCOPE_AGENT);
}
// the last workgroup of the launch: the list's running {count, status} to the device-side mirror and the host header
__device__ __forceinline__ void publish_header(uint32_t *nrecords /* {slot allocator, status, published count} */, uint32_t *status,
                                               volatile uint32_t *hdr_host, unsigned long long launch_total)
{
    const uint32_t total = nrecords[2] + (uint32_t)launch_total;
    nrecords[2] = total;
    hdr_host[0] = total;
    hdr_host[1] = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int RESOLVE_THREADS = 256;       // many channels, few segments each
constexpr int RESOLVE_THREADS_WIDE = 1024; // few channels, thousands of segments each (one channel x 2^26 samples)
constexpr int RESOLVE_LDS_HITS = 512;        // hits walked per pass of the narrow kernel (10 KB of LDS instead of 38: four workgroups per CU become sixteen)
constexpr int RESOLVE_LDS_HITS_WIDE = 2048;

// One workgroup per channel.  The hit lists of a channel's wave segments are ordered but scattered
// (up to thousands of segments when one channel is pushed 2^26 samples at a time): 256 (or 1024) lanes compact them into
// LDS with a block-wide prefix sum (coalesced count loads, independent hit loads); the hold-off walk then runs on LDS,
// split into independent chains (see below).  0.72 ms -> 0.09 (LDS, one lane) -> parallel chains, for one channel x 2^26.
// A batch of THREADS segments with more hits than the LDS window holds is walked in several passes (the hold-off state
// carries from pass to pass exactly as it does from batch to batch), so no hit count overflows this kernel.
// TWO = two samples per symbol (a.sps == 2): the capture rule of the wideband seam at D = 768 (manchester_from_ring)
// SEARCH = 0: the trigger hits come from det / detcount (written by the streaming kernel or by recc_bits_kernel); SEARCH = 2 or 3 (samples
// per symbol): the workgroup searches its channel's slicer bits itself first (bits_search_segment, one quarter of the push per wave).
// The stand-alone search kernel took 13.5 us per wideband step at D = 768, 7 of them instruction issue and the rest what ANY launch costs;
// in here its issue slots are the ones this latency-bound kernel leaves idle, and a kernel boundary (~5 us) goes with it.
constexpr int RESOLVE_SEARCH_CAP = 320;    // hits per quarter: a quarter of the largest push (2^18 / 4 frames) holds at most 65 536 / (74 sps) of them
template <int THREADS, int HITS, bool TWO = false, int SEARCH = 0, bool STOL = false>
__global__ __launch_bounds__(THREADS, 4) void recc_resolve_kernel(ResolveArgs a)
{
    static_assert(SEARCH == 0 || (THREADS == 256 && (SEARCH == 2 || SEARCH == 3)), "the search stage runs in the narrow kernel, at the wideband seam's rates");
    __shared__ __attribute__((aligned(16))) uint32_t s_sw[SEARCH ? 4 : 1][SEARCH ? 8 + 64 * 4 : 1];   // the search's LDS window per wave
    __shared__ uint64_t s_sdet[SEARCH ? 4 : 1][SEARCH ? RESOLVE_SEARCH_CAP : 1];
    __shared__ uint32_t s_scnt[4];
    __shared__ uint64_t s_hits[HITS];
    __shared__ uint32_t s_scan[THREADS / 64];                      // wave totals of the count scan
    __shared__ uint64_t s_acc[HITS + 1];                           // +1: the pending capture of an earlier push
    __shared__ uint8_t  s_head[HITS], s_accf[HITS];
    __shared__ uint64_t s_na_out, s_pend;
    __shared__ uint32_t s_total, s_nacc, s_base;
    extern __shared__ uint64_t s_cap[];                            // [CAP_WAVES][resolve_cap_stride]
    const int c = blockIdx.x, tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    RTL(0);
    const uint64_t span_hold = (uint64_t)a.sps * (AMPS_RECC_CAPTURE_SYMS + AMPS_RECC_TRIGGER_SYMS);
    const uint64_t span_done = (uint64_t)a.sps * (AMPS_RECC_CAPTURE_SYMS + 1) + (a.track ? AMPS_TRACK_BLOCKS : 0);   // + the most the sampling instants can move
    // (the first batch's hit counts are fetched HERE, beside the channel's state: they used to be the third dependent global load of a
    // workgroup's first 3 us)
    uint64_t next_allowed = a.next_allowed[c];                        // uniform across the block
    uint64_t pend = a.pending[c];
    if constexpr (SEARCH != 0) {
        if (blockIdx.x == 0 && tid == 0) {                            // the launch's housekeeping (front_housekeeping)
            if (a.zero1) *a.zero1 = 0u;
            if (a.zero2) { a.zero2[0] = 0u; a.zero2[1] = 0u; a.zero2[2] = 0u; }
        }
        // wave w searches tiles [Tc w / 4, Tc (w + 1) / 4) of the push: the run starts of [512 t_lo - 64, min(512 t_hi, P) - 64), in order
        const uint32_t Tc = (a.search_P + TILE - 1) / TILE;
        const uint32_t t_lo = (uint32_t)((uint64_t)Tc * (uint32_t)wv / 4u), t_hi = (uint32_t)((uint64_t)Tc * ((uint32_t)wv + 1u) / 4u);
        uint32_t nd = 0;
        if (t_hi > t_lo)
            nd = bits_search_segment<SEARCH, STOL>(a.gring + (uint64_t)c * a.ring_words, a.ring_words, a.n_proc - a.search_P, a.search_P, a.search_tol, a.status,
                                                   t_lo, t_hi, s_sw[wv], s_sdet[wv], (uint32_t)RESOLVE_SEARCH_CAP, lane);
        if (lane == 0) s_scnt[wv] = nd < (uint32_t)RESOLVE_SEARCH_CAP ? nd : (uint32_t)RESOLVE_SEARCH_CAP;
        __syncthreads();
    }
    const uint32_t n_first = [&]() -> uint32_t {
        if constexpr (SEARCH != 0) return (uint32_t)tid < 4u ? s_scnt[tid] : 0u;
        const uint64_t gs0 = (uint64_t)c * a.tiles_per_channel;
        const uint32_t nch = (uint32_t)((gs0 + a.tiles_per_channel - 1) / a.span - gs0 / a.span) + 1;
        return (uint32_t)tid < nch ? a.detcount[(uint64_t)c * a.max_chunks + (uint32_t)tid] : 0u;
    }();
    auto centre = [](uint64_t ei) -> uint64_t { return (ei >> 8) + (uint32_t)(ei & 0xff) / 2; };   // of the run of matching phases

    // accepted captures are collected in LDS; the first CAP_WAVES waves then capture and decode them, one each per round.  The
    // slots of the whole batch are reserved by the last wave with one atomic whose round trip hides behind the first decode.
    uint32_t reserved = 0;                                            // record slots this workgroup has reserved in this launch (uniform)
    auto flush = [&]() {                                              // all threads
        __syncthreads();
        const uint32_t m = s_nacc;
        if (!a.capq) reserved += m;
        if (m && a.capq) {                                            // queue form: one atomicAdd per batch
            if (tid == 0) s_base = atomicAdd(a.capq_count, m);
            __syncthreads();
            const uint32_t base = s_base;
            for (uint32_t i = tid; i < m; i += THREADS) {
                if (base + i < a.capq_cap) a.capq[base + i] = ((uint64_t)c << CAPQ_POS_BITS) | (s_acc[i] & ((1ull << CAPQ_POS_BITS) - 1));
                else atomicOr(a.status, 2u);
            }
            __syncthreads();
        } else if (m) {
            for (uint32_t i0 = 0; i0 < m; i0 += CAP_WAVES) {
                // (rotating the capture -> wave assignment by the channel, so that the workgroups sharing a CU do not all decode in their
                // wave 0, was measured in round 5: 0.0311 against 0.0304 ms -- no gain, not kept)
                const uint32_t i = i0 + (uint32_t)wv;
                const bool mine = wv < CAP_WAVES && i < m;
                uint64_t *scr = s_cap + (size_t)wv * resolve_cap_stride(a.cap_words);
#ifdef RESOLVE_TIMELINE
                RtlStamps st{ (a.tl && tid == 0) ? a.tl + (size_t)blockIdx.x * 24 + 8 : nullptr };
#else
                RtlStamps st;
#endif
                if (mine) capture_gather_wave(a, (uint32_t)c, s_acc[i], scr, lane, st);
                if (i0 == 0) {
                    // the slot atomic goes out BEHIND the ring loads: vector memory returns in order per CU, and this one queues at
                    // the memory side behind every other workgroup's (measured: issued first, it held the ring words back 7 us)
                    __syncthreads();
                    if (tid == THREADS - 1) s_base = atomicAdd(a.nrecords, m);
                }
                if (mine) capture_decode_wave<TWO>(a, (uint32_t)c, s_acc[i], scr, lane, st);
                RTL(3);
                __syncthreads();
                RTL(4);
                if (mine) {
                    const uint32_t slot = s_base + i;
                    if (slot < a.rec_cap) capture_store_wave(a, s_acc[i], slot, scr, lane);
                    else if (lane == 0) { atomicOr(a.status, 4u); __threadfence(); }   // rare: performed before this workgroup counts itself done
                }
                __syncthreads();
            }
        }
    };
    if (tid == 0) s_nacc = 0;
    if (pend != ~0ull && pend + span_done < a.n_proc) {
        if (tid == 0) { s_acc[0] = pend; s_nacc = 1; }
        pend = ~0ull;
    }

    // segments of channel c: waves floor(c*Tc/span) .. floor((c*Tc + Tc - 1)/span), in stream order
    const uint64_t gs = (uint64_t)c * a.tiles_per_channel;
    const uint32_t nchunks = SEARCH != 0 ? 4u : (uint32_t)((gs + a.tiles_per_channel - 1) / a.span - gs / a.span) + 1;
    const uint32_t *cnt = SEARCH != 0 ? s_scnt : a.detcount + (uint64_t)c * a.max_chunks;
    const uint64_t *det = SEARCH != 0 ? &s_sdet[0][0] : a.det + (uint64_t)c * a.max_chunks * a.det_cap;
    const uint32_t det_cap = SEARCH != 0 ? (uint32_t)RESOLVE_SEARCH_CAP : a.det_cap;

    for (uint32_t cb = 0; cb < nchunks; cb += THREADS) {
        // ---- compaction of up to THREADS segments into LDS, order preserved
        const uint32_t ch = cb + tid;
        const uint32_t n = cb == 0 ? n_first : (ch < nchunks ? cnt[ch] : 0u);
        // inclusive scan of the counts: inside each wave by lane shifts, then the wave totals (two barriers; the first version
        // was a Hillis-Steele scan over the workgroup in LDS, seventeen)
        uint32_t incl = n;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)incl, off);
            if (lane >= off) incl += y;
        }
        if (lane == 63) s_scan[wv] = incl;
        __syncthreads();
        for (int w = 0; w < wv; w++) incl += s_scan[w];
        const uint32_t excl = incl - n;
        if (tid == THREADS - 1) s_total = incl;
        __syncthreads();
        const uint32_t batch_total = s_total;
        RTL(1);
        const uint64_t *d = det + (uint64_t)ch * det_cap;
        for (uint32_t win = 0; win == 0 || win < batch_total; win += HITS) {
            if (tid == 0) { s_pend = ~0ull; s_na_out = next_allowed; }
            for (uint32_t i = 0; i < n; i++) {
                const uint32_t g = excl + i;
                if (g >= win && g < win + HITS) s_hits[g - win] = d[i];
            }
            __syncthreads();
            const uint32_t total = batch_total - win < (uint32_t)HITS ? batch_total - win : (uint32_t)HITS;
            // ---- hold-off walk.  The rule is sequential (a hit is dropped iff it starts inside the hold-off of the last
            // ACCEPTED hit), but a hit that starts at least one hold-off after the latest possible CENTRE of any earlier hit is
            // outside every earlier burst whatever was accepted: hits are ordered by their START and a run is at most 256 phases
            // long, so every earlier centre is below start[j-1] + 128 (the centre of hit j-1 itself is no bound: an earlier,
            // longer run can end later -- tolerant sync lengthens runs).  Such a hit, if it also lies beyond the hold-off carried
            // in from the previous pass / batch / push, is accepted unconditionally and heads a new chain.  Chains are walked
            // independently, one lane each; real traffic has one hit per chain (one lane did all of it before: 125 ns per
            // hit, 0.09 ms for the 745 bursts of one channel x 2^26).
            for (uint32_t j = tid; j < total; j += THREADS)
                s_head[j] = (j == 0) || ((s_hits[j] >> 8) >= (s_hits[j - 1] >> 8) + 128 + span_hold && (s_hits[j] >> 8) >= next_allowed);
            __syncthreads();
            for (uint32_t j = tid; j < total; j += THREADS) {
                if (!s_head[j]) continue;
                uint64_t na = j == 0 ? next_allowed : 0ull;           // hit 0 continues the state carried into this pass
                uint32_t k = j;
                do {
                    const uint64_t ei = s_hits[k];
                    const bool acc = (ei >> 8) >= na;
                    if (acc) na = centre(ei) + span_hold;
                    s_accf[k] = (uint8_t)acc;
                    k++;
                } while (k < total && !s_head[k]);
                if (k == total) s_na_out = na;                        // the last chain carries the state out
            }
            __syncthreads();
            for (uint32_t j = tid; j < total; j += THREADS) {
                if (!s_accf[j]) continue;
                const uint64_t nc = centre(s_hits[j]);
                if (nc + span_done < a.n_proc) s_acc[atomicAdd(&s_nacc, 1u)] = nc;   // order is irrelevant: drain sorts the records
                else s_pend = nc;                                     // tail not received yet (at most one: the last)
            }
            __syncthreads();
            next_allowed = s_na_out;
            if (s_pend != ~0ull) pend = s_pend;
            RTL(2);
            flush();
            RTL(5);
            if (tid == 0) s_nacc = 0;
            __syncthreads();
        }
    }
    if (tid == 0) { a.next_allowed[c] = next_allowed; a.pending[c] = pend; }
    if (tid == 0 && !a.capq) {
        // the workgroup that finishes last publishes the list's running {count, status} to host memory: drain_begin needs no
        // device-to-host copy on the stream (4.4 us of it per push in the pipelined flow).  No fence here: a fence would wait for
        // this workgroup's record stores to cross PCIe (measured: +30 us per launch); the header needs only the two device-side
        // counters, and every slot atomic of this workgroup has returned its value (the barrier in flush) before the one below.
        RTL(6);
        const uint32_t ng = gridDim.x < DONE_GROUPS ? gridDim.x : DONE_GROUPS;
        const uint32_t g = blockIdx.x % DONE_GROUPS;
        const uint32_t gsize = (gridDim.x - g + DONE_GROUPS - 1) / DONE_GROUPS;
        const unsigned long long gv = count_done(a.done_blocks + 1 + g, 1u, reserved);
        if ((uint32_t)gv == gsize - 1) {
            const unsigned long long gtot = (gv >> 32) + reserved;    // the group's reservations: everybody's before mine, and mine
            atomicExch(a.done_blocks + 1 + g, 0ull);
            const unsigned long long tv = count_done(a.done_blocks, 1u, gtot);
            if ((uint32_t)tv == ng - 1) {
                publish_header(a.nrecords, a.status, a.hdr_host, (tv >> 32) + gtot);
                atomicExch(a.done_blocks, 0ull);
            }
        }
        RTL(7);
    }
}

// Queue form of capture + decode: one single-wave workgroup per queued capture.  Used when the handle has few channels: one
// channel x 2^26 samples holds 745 bursts, which the resolve kernel's one workgroup would decode four at a time (1.5 ms; here
// 0.03).  With many channels the fused form wins: no queue, no second launch, no 2048 workgroups to dispatch.
inline bool resolve_uses_queue(uint32_t n_channels) { return n_channels < 64; }
template <bool TWO = false>
__global__ __launch_bounds__(64) void recc_capture_kernel(ResolveArgs a)
{
    extern __shared__ uint64_t s_cap[];                            // one resolve_cap_stride
    const int lane = threadIdx.x;
    uint32_t ncap = *a.capq_count;
    if (ncap > a.capq_cap) ncap = a.capq_cap;
    uint32_t reserved = 0;
    for (uint32_t q = blockIdx.x; q < ncap; q += gridDim.x) {
        reserved++;
        const uint64_t e = a.capq[q];
        const uint32_t c = (uint32_t)(e >> CAPQ_POS_BITS);
        const uint64_t nc = e & ((1ull << CAPQ_POS_BITS) - 1);
        capture_gather_wave(a, c, nc, s_cap, lane);
        uint32_t slot = 0;
        if (lane == 0) slot = atomicAdd(a.nrecords, 1u);
        capture_decode_wave<TWO>(a, c, nc, s_cap, lane);
        slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
        if (slot < a.rec_cap) capture_store_wave(a, nc, slot, s_cap, lane);
        else if (lane == 0) { atomicOr(a.status, 4u); __threadfence(); }   // rare: performed before this workgroup counts itself done
        WaveSync::sync();
    }
    // the last workgroup publishes the header (see recc_resolve_kernel); only the workgroups that had a capture (and workgroup 0)
    // take part
    const uint32_t nb = ncap < gridDim.x ? (ncap ? ncap : 1u) : gridDim.x;
    if (lane == 0 && blockIdx.x < nb) {
        const unsigned long long t = count_done(a.done_blocks, 1u, reserved);
        if ((uint32_t)t == nb - 1) {
            publish_header(a.nrecords, a.status, a.hdr_host, (t >> 32) + reserved);
            atomicExch(a.done_blocks, 0ull);
        }
    }
}

// recc_decode core on a batch of 3374-byte bursts (amps_recc_decode_bursts)
__global__ __launch_bounds__(64) void recc_decode_bursts_kernel(const uint8_t *bursts, const uint32_t *chan,
                                                                uint32_t nbursts, amps_recc_burst_t *out, uint32_t majority)
{
    __shared__ DecodeScratch s;
    const int lane = threadIdx.x;
    for (uint32_t q = blockIdx.x; q < nbursts; q += gridDim.x) {
        const uint8_t *b = bursts + (uint64_t)q * AMPS_RECC_CAPTURE_SYMS;
        for (int i = lane; i < AMPS_RECC_CAPTURE_SYMS; i += 64) s.sym[i] = b[i];
        __syncthreads();
        decode_burst_wave(s, chan ? chan[q] : 0u, 0ull, out + q, majority != 0);
        __syncthreads();
    }
}

// BCH(63,51) shortened to (k+12, k): one code word per lane (amps_bch_encode_words / amps_bch_decode_words)
__global__ __launch_bounds__(256) void bch_encode_words_kernel(const uint8_t *msg, uint32_t n, int k, uint8_t *cw)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        // systematic encode, parity = m(x) x^12 mod g(x) (bch_short_encode, in registers: no per-lane byte arrays)
        unsigned rem = 0;
        for (int j = 0; j < k; j++) {
            const unsigned bit = msg[(uint64_t)i * k + j] & 1u;
            const unsigned fb = ((rem >> 11) & 1u) ^ bit;
            rem = (rem << 1) & 0xfffu;
            if (fb) rem ^= 0x539u;
            cw[(uint64_t)i * (k + 12) + j] = (uint8_t)bit;
        }
        for (int j = 0; j < 12; j++) cw[(uint64_t)i * (k + 12) + k + j] = (uint8_t)((rem >> (11 - j)) & 1u);
    }
}
__global__ __launch_bounds__(256) void bch_decode_words_kernel(const uint8_t *cw, uint32_t n, int k, uint8_t *msg, uint8_t *valid,
                                                               uint8_t *nerr)
{
    const int nb = k + 12;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        // the code word as a polynomial in one register (bit e = coefficient of x^e = byte nb - 1 - e): corrections are bit flips,
        // not stores into a per-lane byte array at a computed index (which the compiler turned into 63 compare-and-select chains
        // and 124 spilled SGPRs)
        uint64_t w = 0;
        for (int j = 0; j < nb; j++) w |= (uint64_t)(cw[(uint64_t)i * nb + j] & 1u) << (nb - 1 - j);
        const uint64_t raw = w;
        BchResult r = bch63_decode_packed(w);
        int ok = r.ok;
#pragma unroll
        for (int f = 0; f < 3; f++) {
            const int e = r.e[f];
            if (e >= nb) ok = 0;                           // a "correction" inside the shortening zeros
            else if (r.ok && e >= 0) w ^= 1ull << e;
        }
        const uint64_t out = ok ? w : raw;
        for (int j = 0; j < k; j++) msg[(uint64_t)i * k + j] = (uint8_t)((out >> (nb - 1 - j)) & 1ull);
        valid[i] = (uint8_t)ok;
        nerr[i] = (uint8_t)(ok ? r.nflip : 0xff);
    }
}

} // namespace amps
