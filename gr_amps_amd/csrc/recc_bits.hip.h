// recc_bits.hip.h -- bit-domain trigger correlator behind the fused channelizer (wideband seam; exact match, or at most
// cfg.sync_tolerance wrong symbols with TOL = true).
//
// chz_fused_kernel leaves only slicer bits in HBM (1 bit per channel sample).  recc_front_kernel<SPS,1,BITS=true> can
// search them, but it inherits the IQ kernel's layout -- four lanes share one dword of positions, 512 positions per wave
// iteration -- and spends ~9 lane-instructions per position on what is a handful of funnel shifts.  Here a lane owns 32
// consecutive positions outright: it reads its own dword and the K = ceil(SPS*73/32) dwords before it (from a small LDS
// window that the wave fills with ONE coalesced load per 2048 positions), and every trigger tap is one v_alignbit plus
// one and/andn2 with compile-time shift and polarity.  16 taps of the word-sync part prefilter (noise passes with
// probability 2^-16 per position); the other 58 run behind a wave-uniform branch.
//
// Round 3: a lane owns FOUR consecutive dwords of a 256-dword block (one 16-byte load, three 16-byte LDS reads for its 11-dword
// window), so the per-block overhead (window exchange, addressing, loop) is paid once per 8192 positions instead of once per
// 2048; the last block of a segment may reach past its end (the surplus positions are masked off).
//
// Detections are emitted exactly as recc_front_kernel does (same run-start / dedup window / run-length rule, same
// attribution of positions to wave segments, same ordering), so recc_resolve_kernel and everything after it are shared:
// a segment [t_lo, t_hi) of 512-sample tiles emits the run starts located in [512 t_lo - 64, min(512 t_hi, P) - 64).
#pragma once
#include "recc_front.hip.h"

namespace amps {

// Run starts of one finished block of 64 J dwords (lane owns dwords J lane + j): m = the lane's match words, mb / ma = the match
// words of the dword before the block and of the dword after it.  Appends (position << 8 | run length - 1) of the starts inside
// [E0, E1) to dst in stream order and returns the new count.  Out of line on purpose: hits are rare, and inlined at its six call
// sites it took the kernel from 42 to 135 VGPRs (3 waves per SIMD instead of 8: the search became latency-bound).
template <int J, int D>
__device__ __noinline__ uint32_t bits_emit_block(int64_t dblk, uint32_t m0, uint32_t m1, uint32_t m2, uint32_t m3, uint32_t mb, uint32_t ma,
                                                 int64_t E0, int64_t E1, uint64_t n_done, uint64_t *dst, uint32_t ndet, uint32_t det_cap,
                                                 uint32_t *status)
{
    const int lane = threadIdx.x & 63;
    const uint32_t m[4] = { m0, m1, m2, m3 };
    const uint32_t up = (uint32_t)__shfl_up((int)m[J - 1], 1), dn = (uint32_t)__shfl_down((int)m[0], 1);
    uint32_t starts[J], mnext[J];
    uint32_t any = 0u;
#pragma unroll
    for (int j = 0; j < J; j++) {
        const uint32_t mp = j ? m[j - 1] : (lane ? up : mb);
        mnext[j] = j < J - 1 ? m[j + 1] : (lane < 63 ? dn : ma);
        uint32_t smear = 0;
#pragma unroll
        for (int s = 1; s <= D; s++) smear |= (m[j] << s) | (mp >> (32 - s));
        uint32_t st = m[j] & ~smear;
        const int64_t pos0 = 32 * (dblk + J * lane + j);
        {   // keep the positions of [E0, E1)
            int64_t lo = E0 - pos0, hi = E1 - pos0;
            lo = lo < 0 ? 0 : lo > 32 ? 32 : lo;
            hi = hi < 0 ? 0 : hi > 32 ? 32 : hi;
            const uint32_t mlo = lo >= 32 ? 0u : ~0u << lo;
            const uint32_t mhi = hi >= 32 ? ~0u : ~(~0u << hi);
            st &= mlo & mhi;
        }
        starts[j] = st;
        any |= st;
    }
    uint64_t who = __ballot(any != 0);
    while (who) {                                  // ordered append, lane by lane, dword by dword
        const int l = __ffsll((unsigned long long)who) - 1;
        who &= who - 1;
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < J; j++) cnt += __popc((uint32_t)__shfl((int)starts[j], l));
        if (lane == l) {
            uint32_t slot_i = ndet;
#pragma unroll
            for (int j = 0; j < J; j++) {
                uint32_t st = starts[j];
                const int64_t pos0 = 32 * (dblk + J * lane + j);
                while (st) {
                    const int p = __ffs((int)st) - 1;
                    st &= st - 1;
                    uint32_t win = (m[j] >> p) | (p ? (mnext[j] << (32 - p)) : 0u);
                    win &= (1u << D) - 1u;
                    const int last = 31 - __clz((int)win);
                    const uint64_t absn = n_done + (uint64_t)(pos0 + p);
                    if (slot_i < det_cap) dst[slot_i] = (absn << 8) | (uint64_t)last;
                    else atomicOr(status, 1u);
                    slot_i++;
                }
            }
        }
        ndet += (uint32_t)cnt;
    }
    return ndet;
}

#ifdef BITS_TIMELINE
__device__ unsigned long long bits_tl[3 * 16384];     // per wave: s_memtime at entry and exit, XCC id (scripts/bits_timeline.py)
#endif

// One segment of one channel's bit stream: the run starts located in [512 t_lo - 64, min(512 t_hi, P) - 64) (relative to n_done), appended
// in stream order to dst (det_cap entries; more set bit 0 of *status); returns their number.  One wave; s_w = its window of KP + 256
// dwords.  Shared by the stand-alone kernel below and by the search stage inside the resolve kernel (recc_resolve.hip.h, round 6).
template <int SPS, bool TOL>
__device__ __forceinline__ uint32_t bits_search_segment(const uint64_t *gring_c, uint32_t ring_words, uint64_t n_done, uint32_t P, uint32_t tol, uint32_t *status,
                                                        uint32_t t_lo, uint32_t t_hi, uint32_t *s_w, uint64_t *dst, uint32_t det_cap, int lane)
{
    constexpr int D = AMPS_DEDUP_SYMBOLS * SPS;          // dedup / run window in samples
    constexpr int HIST = SPS * (TRIG - 1);               // a match at n looks back to n - HIST
    constexpr int K = (HIST + 31) / 32;                  // history dwords a lane needs besides its own
    constexpr int KP = 8;                                // history slots of the LDS window (padded: the block starts 16-byte aligned)
    constexpr int JW = 4;                                // dwords a lane owns in a block
    constexpr int U = 2;                                 // blocks in flight per wave
    static_assert(D <= 32 && K <= KP - 1, "bit-domain kernel is for small samples-per-symbol");
    const uint32_t *gring32 = (const uint32_t *)gring_c;
    uint32_t ndet = 0;                                   // wave-uniform
    {
        // run starts of this segment: relative sample positions [E0, E1)
        int64_t E0 = (int64_t)t_lo * TILE - 64;
        int64_t E1 = (int64_t)t_hi * TILE; if (E1 > (int64_t)P) E1 = P;
        E1 -= 64;
        if ((int64_t)n_done + E0 < 0) E0 = -(int64_t)n_done;      // nothing before the stream
        if (E1 > E0) {
            // matches are needed on [E0 - D, E1 + D); dwords [d_first, d_last) of the relative bit stream.  d_first is moved down
            // to a ring index that is a multiple of four (16-byte loads; the stream position is a multiple of 64 and the ring a
            // power of two, so the ring index is 32-bit arithmetic)
            const int64_t M0 = E0 - D, M1 = E1 + D;
            const uint32_t mask32 = 2u * ring_words - 1u;
            const uint32_t base32 = (uint32_t)(n_done >> 5) & mask32;
            int64_t d_first = M0 >= 0 ? M0 / 32 : -((-M0 + 31) / 32);
            d_first -= (int64_t)((base32 + (uint32_t)d_first) & (uint32_t)(JW - 1));
            const int64_t d_last = (M1 + 31) / 32;
            const int ndw = (int)(d_last - d_first);
            const int nwide = (ndw + 64 * JW - 1) / (64 * JW);                  // blocks (the last one may reach past d_last: its surplus positions lie outside [E0, E1))
            const bool pre = (int64_t)n_done + 32 * (d_first - K) < 0;        // wave-uniform: the segment reaches before the stream (ones there)
            auto load_word = [&](int64_t dj) -> uint32_t {
                if (pre && (int64_t)n_done + 32 * dj < 0) return ~0u;
                return gring32[(base32 + (uint32_t)dj) & mask32];
            };
            struct Wide { uint32_t w[JW]; };
            auto load_wide = [&](int64_t dj) -> Wide {                          // dwords dj .. dj + JW - 1, dj's ring index a multiple of JW
                Wide r;
                if (pre) {
#pragma unroll
                    for (int j = 0; j < JW; j++) r.w[j] = load_word(dj + j);
                } else {
                    const uint4 q = *(const uint4 *)(gring32 + ((base32 + (uint32_t)dj) & mask32));
                    r.w[0] = q.x; r.w[1] = q.y; r.w[2] = q.z; r.w[3] = q.w;
                }
                return r;
            };

            // state of the block before the current one: its run starts can be emitted once the current block's first match
            // word (the look-ahead of its last dword) exists
            uint32_t m_prev[JW] = {};
            uint32_t mb_prev = 0u, last_m = 0u;          // match words of the dword before the previous block / of the newest dword
            int64_t dblk_prev = 0;
            bool have_prev = false;
            bool hit_prev = false;

            auto emit_prev = [&](uint32_t ma) {          // the block before the current one (rare: it held a match)
                ndet = bits_emit_block<JW, D>(dblk_prev, m_prev[0], m_prev[1], m_prev[2], m_prev[3], mb_prev, ma, E0, E1, n_done, dst, ndet, det_cap, status);
            };

            // one block: window exchange through LDS, match words of the lane's J dwords, deferred emission of the block before
            auto block = [&](int64_t dblk, const uint32_t (&w)[JW], uint32_t hist0) {
                constexpr int J = JW;
                // history: the last K dwords of the previous block (or the K dwords before the segment)
                uint32_t carry = hist0;
                if (have_prev && lane < K) carry = s_w[KP + 64 * J - 1 - lane];
                __builtin_amdgcn_wave_barrier();
                if (lane < K) s_w[KP - 1 - lane] = carry;
                *(uint4 *)(s_w + KP + 4 * lane) = make_uint4(w[0], w[1], w[2], w[3]);
                __builtin_amdgcn_wave_barrier();
                // L[k] = dword (J lane - K + k) of the block, k < K + J; one more (never shifted in) closes the last tap
                uint32_t L[K + J + 1];
                {
                    static_assert(KP == 8 && JW == 4, "three aligned 16-byte reads cover the window");
                    const uint4 q0 = *(const uint4 *)(s_w + 4 * lane), q1 = *(const uint4 *)(s_w + 4 * lane + 4), q2 = *(const uint4 *)(s_w + 4 * lane + 8);
                    const uint32_t all[12] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w };
#pragma unroll
                    for (int k = 0; k < K + J; k++) L[k] = all[KP - K + k];
                }
                L[K + J] = 0u;
                // symbol i of the trigger sits SPS * (73 - i) samples before the position; j = the lane's dword
                auto tapx = [&](int i, int j) -> uint32_t {
                    const int base = 32 * (K + j) - SPS * (TRIG - 1 - i);
                    return __builtin_amdgcn_alignbit(L[(base >> 5) + 1], L[base >> 5], base & 31);
                };
                auto sym = [](int i) -> bool { return ((i < 64 ? TRIG_LO >> i : TRIG_HI >> (i - 64)) & 1ull) != 0; };
                uint32_t m[JW] = {};
                if constexpr (TOL) {
                    // tolerant sync (cfg.sync_tolerance): at most tol of the 74 symbols differ.  The mismatch words of the
                    // taps are summed bit-sliced: carry-save adders (Harley-Seal) keep the weights 1, 2, 4 in three words
                    // and emit one weight-8 word per eight taps, which ripples into the planes 8..64; the 7-bit sums are
                    // then compared with tol plane by plane.  ~2 instructions per tap instead of ~16 for a ripple counter.
                    auto csa = [](uint32_t &h, uint32_t &l, uint32_t x, uint32_t y, uint32_t z) {
                        const uint32_t u = x ^ y;
                        h = (x & y) | (u & z);
                        l = u ^ z;
                    };
                    static_assert(TRIG == 74, "the adder tree below is laid out for 9 x 8 + 2 taps");
#pragma unroll
                    for (int j = 0; j < J; j++) {
                        auto mism = [&](int i) -> uint32_t { const uint32_t x = tapx(i, j); return sym(i) ? ~x : x; };
                        uint32_t ones = 0u, twos = 0u, fours = 0u, hi[4] = { 0u, 0u, 0u, 0u };   // hi[k]: weight 8 << k
                        auto add8 = [&](uint32_t e) {
#pragma unroll
                            for (int k = 0; k < 4; k++) { const uint32_t cy = hi[k] & e; hi[k] ^= e; e = cy; }
                        };
#pragma unroll
                        for (int blk = 0; blk < 9; blk++) {
                            const int i0 = 8 * blk;
                            uint32_t twosA, twosB, foursA, foursB, eights;
                            csa(twosA, ones, ones, mism(i0), mism(i0 + 1));
                            csa(twosB, ones, ones, mism(i0 + 2), mism(i0 + 3));
                            csa(foursA, twos, twos, twosA, twosB);
                            csa(twosA, ones, ones, mism(i0 + 4), mism(i0 + 5));
                            csa(twosB, ones, ones, mism(i0 + 6), mism(i0 + 7));
                            csa(foursB, twos, twos, twosA, twosB);
                            csa(eights, fours, fours, foursA, foursB);
                            add8(eights);
                        }
                        {
                            uint32_t t2;
                            csa(t2, ones, ones, mism(72), mism(73));          // weight-2 carry of the last two taps
                            const uint32_t c4 = twos & t2; twos ^= t2;
                            const uint32_t c8 = fours & c4; fours ^= c4;
                            add8(c8);
                        }
                        const uint32_t plane[7] = { ones, twos, fours, hi[0], hi[1], hi[2], hi[3] };
                        uint32_t gt = 0u, eq = ~0u;
#pragma unroll
                        for (int pl = 6; pl >= 0; pl--) {
                            const uint32_t kb = 0u - ((tol >> pl) & 1u);
                            gt |= eq & plane[pl] & ~kb;
                            eq &= ~(plane[pl] ^ kb);
                        }
                        m[j] = ~gt;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < J; j++) {
                        uint32_t acc = ~0u;
#pragma unroll
                        for (int i = TRIG - 16; i < TRIG; i++) { const uint32_t x = tapx(i, j); acc = sym(i) ? acc & x : acc & ~x; }
                        if (__ballot(acc != 0)) {                             // rare (2048 x 2^-16 on noise): the other 58 symbols
#pragma unroll
                            for (int i = 0; i < TRIG - 16; i++) { const uint32_t x = tapx(i, j); acc = sym(i) ? acc & x : acc & ~x; }
                        }
                        m[j] = acc;
                    }
                }
                uint32_t any = 0u;
#pragma unroll
                for (int j = 0; j < J; j++) any |= m[j];
                const bool hit = __ballot(any != 0) != 0;
                // the block before can be emitted now that its look-ahead word (this block's first) exists
                if (hit_prev) emit_prev((uint32_t)__builtin_amdgcn_readlane((int)m[0], 0));
                mb_prev = last_m;
                last_m = (uint32_t)__builtin_amdgcn_readlane((int)m[J - 1], 63);
#pragma unroll
                for (int j = 0; j < JW; j++) m_prev[j] = m[j];
                dblk_prev = dblk; have_prev = true; hit_prev = hit;
            };

            // U blocks in flight per wave (measured, us event to event for 2^27 wideband samples: U = 1 ... 3 19.0 +- 0.2, U = 4
            // 20.1; round 2's one-dword-per-lane form 20.7: the kernel is bound by instruction issue, ~240 wave instructions per
            // 8192 positions of which 104 are the sixteen prefilter taps -- scripts/bits_timeline.py)
            const uint32_t hist0 = lane < K ? load_word(d_first - 1 - lane) : 0u;
            Wide q[U];
#pragma unroll
            for (int u = 0; u < U; u++) q[u] = u < nwide ? load_wide(d_first + 64 * JW * u + JW * lane) : Wide{};
            for (int t0 = 0; t0 < nwide; t0 += U) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int t = t0 + u;
                    if (t >= nwide) break;
                    const int64_t dblk = d_first + 64 * JW * (int64_t)t;
                    const Wide cur = q[u];
                    if (t + U < nwide) q[u] = load_wide(dblk + 64 * JW * U + JW * lane);
                    block(dblk, cur.w, hist0);
                }
            }
            if (hit_prev) emit_prev(0u);
        }
    }
    return ndet;
}

template <int SPS, bool TOL = false>
__global__ __launch_bounds__(256) void recc_bits_kernel(FrontArgs a)
{
#ifdef BITS_TIMELINE
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memtime();
#endif
    // window of one wave: [KP - h] = the h-th dword before the block (h = 1..K), [KP + q] = dword q of the block
    __shared__ __attribute__((aligned(16))) uint32_t s_w_all[4][8 + 64 * 4];
    front_housekeeping(a);

    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    uint32_t *s_w = s_w_all[wv];
    const uint32_t w_id = blockIdx.x * 4 + wv;
    const uint64_t Tc = a.tiles_per_channel;
    const uint64_t g_end_all = (uint64_t)a.n_channels * Tc;
    uint64_t g0 = (uint64_t)w_id * a.span;
    uint64_t g1 = g0 + a.span; if (g1 > g_end_all) g1 = g_end_all;

    while (g0 < g1) {                                    // one segment = a run of tiles inside one channel
        const int c = (int)(g0 / Tc);
        const uint32_t t_lo = (uint32_t)(g0 - (uint64_t)c * Tc);
        uint32_t t_hi = t_lo + (uint32_t)(g1 - g0); if (t_hi > Tc) t_hi = (uint32_t)Tc;
        const uint32_t chunk = w_id - (uint32_t)(((uint64_t)c * Tc) / a.span);   // k-th segment of this channel
        g0 += (uint64_t)(t_hi - t_lo);
        uint64_t *dst = a.det + ((uint64_t)c * a.max_chunks + chunk) * a.det_cap;
        const uint32_t ndet = bits_search_segment<SPS, TOL>(a.gring + (uint64_t)c * a.ring_words, a.ring_words, a.n_done, a.P, a.tol, a.status, t_lo, t_hi, s_w, dst,
                                                            a.det_cap, lane);
        if (lane == 0) a.detcount[(uint64_t)c * a.max_chunks + chunk] = ndet < a.det_cap ? ndet : a.det_cap;
    }
#ifdef BITS_TIMELINE
    if (lane == 0 && w_id < 16384) {
        bits_tl[3 * w_id] = tl_t0; bits_tl[3 * w_id + 1] = __builtin_amdgcn_s_memtime();
        bits_tl[3 * w_id + 2] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf;   // XCC_ID[3:0]
    }
#endif
}

} // namespace amps
