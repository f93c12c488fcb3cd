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
// Detections are emitted exactly as recc_front_kernel does (same run-start / dedup window / run-length rule, same
// attribution of positions to wave segments, same ordering), so recc_resolve_kernel and everything after it are shared:
// a segment [t_lo, t_hi) of 512-sample tiles emits the run starts located in [512 t_lo - 64, min(512 t_hi, P) - 64).
#pragma once
#include "recc_front.hip.h"

namespace amps {

template <int SPS, bool TOL = false>
__global__ __launch_bounds__(256) void recc_bits_kernel(FrontArgs a)
{
    constexpr int D = AMPS_DEDUP_SYMBOLS * SPS;          // dedup / run window in samples
    constexpr int HIST = SPS * (TRIG - 1);               // a match at n looks back to n - HIST
    constexpr int K = (HIST + 31) / 32;                  // history dwords a lane needs besides its own
    static_assert(D <= 32 && K <= 10, "bit-domain kernel is for small samples-per-symbol");
    __shared__ uint32_t s_w_all[4][K + 64];
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
        const uint32_t *gring32 = (const uint32_t *)(a.gring + (uint64_t)c * a.ring_words);
        uint64_t *dst = a.det + ((uint64_t)c * a.max_chunks + chunk) * a.det_cap;
        uint32_t ndet = 0;                               // wave-uniform

        // run starts of this segment: relative sample positions [E0, E1)
        int64_t E0 = (int64_t)t_lo * TILE - 64;
        int64_t E1 = (int64_t)t_hi * TILE; if (E1 > (int64_t)a.P) E1 = a.P;
        E1 -= 64;
        if ((int64_t)a.n_done + E0 < 0) E0 = -(int64_t)a.n_done;      // nothing before the stream
        if (E1 > E0) {
            // matches are needed on [E0 - D, E1 + D); dwords [d_first, d_last) of the relative bit stream
            const int64_t M0 = E0 - D, M1 = E1 + D;
            const int64_t d_first = M0 >= 0 ? M0 / 32 : -((-M0 + 31) / 32);
            const int64_t d_last = (M1 + 31) / 32;
            const int nblocks = (int)((d_last - d_first + 63) / 64);
            auto load_word = [&](int64_t dj) -> uint32_t {     // dword dj of the relative bit stream; ones before the stream
                const int64_t n32 = (int64_t)a.n_done + 32 * dj;
                return n32 < 0 ? ~0u : gring32[(uint64_t)(n32 >> 5) & (2ull * a.ring_words - 1)];
            };
            // emit the run starts of one finished block of 64 dwords: m = the lane's match word, mb / ma = the match words
            // before lane 0's and after lane 63's
            auto emit = [&](int64_t dblk, uint32_t m, uint32_t mb, uint32_t ma) {
                const uint32_t up = (uint32_t)__shfl_up((int)m, 1), dn = (uint32_t)__shfl_down((int)m, 1);
                const uint32_t mp = lane ? up : mb;
                const uint32_t mn = lane < 63 ? dn : ma;
                uint32_t smear = 0;
#pragma unroll
                for (int s = 1; s <= D; s++) smear |= (m << s) | (mp >> (32 - s));
                uint32_t starts = m & ~smear;
                const int64_t pos0 = 32 * (dblk + lane);
                {   // keep the positions of [E0, E1)
                    int64_t lo = E0 - pos0, hi = E1 - pos0;
                    lo = lo < 0 ? 0 : lo > 32 ? 32 : lo;
                    hi = hi < 0 ? 0 : hi > 32 ? 32 : hi;
                    const uint32_t mlo = lo >= 32 ? 0u : ~0u << lo;
                    const uint32_t mhi = hi >= 32 ? ~0u : ~(~0u << hi);
                    starts &= mlo & mhi;
                }
                uint64_t who = __ballot(starts != 0);
                while (who) {                                  // ordered append, lane by lane
                    const int l = __ffsll((unsigned long long)who) - 1;
                    who &= who - 1;
                    const int cnt = __popc((uint32_t)__shfl((int)starts, l));
                    if (lane == l) {
                        uint32_t slot_i = ndet;
                        while (starts) {
                            const int p = __ffs((int)starts) - 1;
                            starts &= starts - 1;
                            uint32_t win = (m >> p) | (p ? (mn << (32 - p)) : 0u);
                            win &= (1u << D) - 1u;
                            const int last = 31 - __clz((int)win);
                            const uint64_t absn = a.n_done + (uint64_t)(pos0 + p);
                            if (slot_i < a.det_cap) dst[slot_i] = (absn << 8) | (uint64_t)last;
                            else atomicOr(a.status, 1u);
                            slot_i++;
                        }
                    }
                    ndet += (uint32_t)cnt;
                }
            };

            // one block fetched ahead is enough: the kernel is bound by instruction issue (~90 wave instructions per 2048 positions),
            // four blocks in flight made it 30 % slower (profiles/EXPERIMENTS.md)
            uint32_t wnext = load_word(d_first + lane);                  // block 0, fetched ahead
            uint32_t hist0 = lane < K ? load_word(d_first - K + lane) : 0u;
            uint32_t m_prev = 0, m_prev_before = 0;                      // block t-1 (lane's word) and the word before its lane 0
            bool hit_prev = false;
            for (int t = 0; t < nblocks; t++) {
                const int64_t dblk = d_first + 64 * (int64_t)t;
                const uint32_t wcur = wnext;
                if (t + 1 < nblocks) wnext = load_word(dblk + 64 + lane);
                // LDS window: [0, K) = the K dwords before this block, [K, K + 64) = the block
                uint32_t carry = 0;
                if (t > 0 && lane < K) carry = s_w[64 + lane];
                __builtin_amdgcn_wave_barrier();
                if (lane < K) s_w[lane] = t > 0 ? carry : hist0;
                s_w[K + lane] = wcur;
                __builtin_amdgcn_wave_barrier();
                uint32_t L[K + 2];
#pragma unroll
                for (int k = 0; k <= K; k++) L[k] = s_w[lane + k];
                L[K + 1] = 0u;
                // symbol i of the trigger sits SPS * (73 - i) samples before the position
                auto tap = [&](int i, uint32_t acc) -> uint32_t {
                    const int base = 32 * K - SPS * (TRIG - 1 - i);
                    const uint32_t x = __builtin_amdgcn_alignbit(L[(base >> 5) + 1], L[base >> 5], base & 31);
                    const bool sym = ((i < 64 ? TRIG_LO >> i : TRIG_HI >> (i - 64)) & 1ull) != 0;
                    return sym ? acc & x : acc & ~x;
                };
                uint32_t acc = ~0u;
                if constexpr (TOL) {
                    // tolerant sync (cfg.sync_tolerance): at most a.tol of the 74 symbols differ.  The mismatch words of the
                    // taps are summed bit-sliced: carry-save adders (Harley-Seal) keep the weights 1, 2, 4 in three words
                    // and emit one weight-8 word per eight taps, which ripples into the planes 8..64; the 7-bit sums are
                    // then compared with a.tol plane by plane.  ~2 instructions per tap instead of ~16 for a ripple counter.
                    auto mism = [&](int i) -> uint32_t {
                        const int base = 32 * K - SPS * (TRIG - 1 - i);
                        const uint32_t x = __builtin_amdgcn_alignbit(L[(base >> 5) + 1], L[base >> 5], base & 31);
                        const bool sym = ((i < 64 ? TRIG_LO >> i : TRIG_HI >> (i - 64)) & 1ull) != 0;
                        return sym ? ~x : x;
                    };
                    auto csa = [](uint32_t &h, uint32_t &l, uint32_t x, uint32_t y, uint32_t z) {
                        const uint32_t u = x ^ y;
                        h = (x & y) | (u & z);
                        l = u ^ z;
                    };
                    uint32_t ones = 0u, twos = 0u, fours = 0u, hi[4] = { 0u, 0u, 0u, 0u };   // hi[k]: weight 8 << k
                    auto add8 = [&](uint32_t e) {
#pragma unroll
                        for (int k = 0; k < 4; k++) { const uint32_t cy = hi[k] & e; hi[k] ^= e; e = cy; }
                    };
                    static_assert(TRIG == 74, "the adder tree below is laid out for 9 x 8 + 2 taps");
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
                        const uint32_t kb = 0u - ((a.tol >> pl) & 1u);
                        gt |= eq & plane[pl] & ~kb;
                        eq &= ~(plane[pl] ^ kb);
                    }
                    acc = ~gt;
                } else {
#pragma unroll
                for (int i = TRIG - 16; i < TRIG; i++) acc = tap(i, acc);
                if (__ballot(acc != 0)) {                                 // rare: the other 58 symbols
#pragma unroll
                    for (int i = 0; i < TRIG - 16; i++) acc = tap(i, acc);
                }
                }
                const bool hit = __ballot(acc != 0) != 0;
                // block t-1 can be emitted now that its look-ahead word (lane 0 of this block) exists
                if (t > 0 && (hit_prev || hit)) {
                    const uint32_t ma = (uint32_t)__builtin_amdgcn_readlane((int)acc, 0);
                    if (hit_prev) emit(dblk - 64, m_prev, m_prev_before, ma);
                }
                m_prev_before = (uint32_t)__builtin_amdgcn_readlane((int)m_prev, 63);
                m_prev = acc;
                hit_prev = hit;
            }
            if (hit_prev) emit(d_first + 64 * (int64_t)(nblocks - 1), m_prev, m_prev_before, 0u);
        }
        if (lane == 0) a.detcount[(uint64_t)c * a.max_chunks + chunk] = ndet < a.det_cap ? ndet : a.det_cap;
    }
}

} // namespace amps
