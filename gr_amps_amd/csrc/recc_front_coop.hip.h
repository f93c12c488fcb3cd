// recc_front_coop.hip.h -- the streaming kernel of the IQ seam with the waves of a workgroup reading ONE stream together.
// Opt-in (AMPS_RECC_COOP=4); the default is recc_front_kernel.  Kept because it is the measured answer to "is the access
// pattern the limit?", and it is held to the same parity tests (tests/test_gpu_parity.py).
//
// recc_front_kernel gives every wave its own span of a channel: thousands of 4 KB-granular streams into HBM at once, and
// that access pattern tops out at 5.3 TB/s on MI355X whatever the kernel does with the data (scripts/ubench_stream.hip:
// linear read 6.4 TB/s, wave streams 5.3, the W waves of a workgroup on W consecutive tiles 5.6 / 5.8 / 6.0 for W = 4 / 8 /
// 16 with a barrier per step).  Here a workgroup of W waves owns the span: at step i wave w takes tile W i + w, so the
// workgroup reads W x 4 KB contiguous per step.  What a tile needs from its predecessor no longer comes from the same
// wave one iteration earlier:
//   * the discriminator / boxcar history (the SPS - 1 values before the tile) is recomputed from the 16 samples in front
//     of the tile (one extra 128-byte load per tile, an L2 hit: the neighbouring wave reads the same lines);
//   * the slicer bits of the two tiles before it (the trigger spans 73 SPS samples) come from a bit ring in LDS that the
//     whole workgroup shares: slice, write, ONE workgroup barrier, correlate;
//   * the run-start / dedup pass over the match words (which looks one word back and one word ahead of a tile) runs one
//     step late, by wave 0, for the W tiles of the previous step in order -- hits are rare, and the detection list of a
//     segment stays ordered.
// Numerics, ring contents, detections and their attribution to segments are exactly those of recc_front_kernel (a segment
// is now a workgroup's run of tiles inside one channel), so recc_resolve_kernel and everything behind it are shared and
// the parity tests hold both kernels to the same CPU model.  Exact sync only (no TOL form), slicer specs A and C.
//
// Measured (scripts/bench_front_ab.py: all variants alive in one process and pushed in turn, 832 x 2^18, spec C): wave-private
// 0.327 / 0.345 ms in two runs, W = 4: 0.3244 / 0.3249, W = 8: 0.328, W = 16: 0.337 (one workgroup per CU at 128 VGPRs: nothing
// runs while it waits at its barrier); spec A: 0.339 wave-private against 0.385 for W = 4.  The lock step the barrier imposes
// costs what the better locality buys (the kernel reaches 95 % of the micro-benchmark's figure for its pattern, as the
// wave-private kernel does for its own), so only W = 4 is built and it is not the default.
#pragma once
#include "recc_front.hip.h"

namespace amps {

template <int SPS, int W, int SL>
__global__ __launch_bounds__(64 * W, 4) void recc_front_coop_kernel(FrontArgs a)   // 128 VGPRs: 16 waves per CU
{
    static_assert(SL == AMPS_SLICER_ATAN_BOXCAR || SL == AMPS_SLICER_SINE, "specs A and C");
    static_assert(W >= 2 && (W & (W - 1)) == 0, "waves per workgroup: a power of two, at least the two halo tiles");
    constexpr int H = SPS - 1;                  // boxcar history
    constexpr int D = AMPS_DEDUP_SYMBOLS * SPS; // dedup / run window in samples (<= 32)
    constexpr int R = 4 * W;                    // tiles in the shared rings: W being written, W + 2 being read, rounded up
    constexpr int RW32 = R * (TILE / 32);       // ring dwords
    constexpr int GW = RW32 / 2;                // ring 64-bit words
    __shared__ float    s_d_all[W][DBUF];       // per wave: 16 values of history + the tile's 512 discriminator outputs
    __shared__ uint32_t s_g[RW32 + 2];          // slicer bits; [RW32], [RW32 + 1] mirror [0], [1]
    __shared__ uint32_t s_m[RW32];              // match words
    __shared__ uint32_t s_hit[R];               // tile had a match
    front_housekeeping(a);

    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const uint32_t w_id = blockIdx.x;           // a segment owner is a workgroup here
    const uint64_t Tc = a.tiles_per_channel;
    const uint64_t g_end_all = (uint64_t)a.n_channels * Tc;
    uint64_t g0 = (uint64_t)w_id * a.span;
    uint64_t g1 = g0 + a.span; if (g1 > g_end_all) g1 = g_end_all;
    const int64_t words_end = (int64_t)a.P / 64;
    const int r_prev = (int)a.r_prev, avail = (int)a.avail;
    float *const dcur = s_d_all[wv];
    const int dw_off = 2 * lane + (lane >> 2);          // P1 writes: didx(DHIST + 128q + 2*lane + e)
    const int dr_off = 9 * lane;                        // P2 reads:  didx(DHIST - H + 8*lane + m)
    const int wq = lane >> 2;                           // P3: dword of the tile this quad owns (0..15)
    const int part = lane & 3;                          // P3: 4 lanes share a dword

  while (g0 < g1) {                                    // one segment = a run of tiles inside one channel
    const int c = (int)(g0 / Tc);
    const uint32_t t_lo = (uint32_t)(g0 - (uint64_t)c * Tc);
    uint32_t t_hi = t_lo + (uint32_t)(g1 - g0); if (t_hi > Tc) t_hi = (uint32_t)Tc;
    const uint32_t chunk = w_id - (uint32_t)(((uint64_t)c * Tc) / a.span);
    const int64_t chunk_start = (int64_t)t_lo * TILE;
    const int K = (int)(t_hi - t_lo);
    g0 += (uint64_t)K;
    const float2 *blk = a.block + (uint64_t)c * a.ld;
    const float2 *car = a.carry + (uint64_t)c * CARRY_CAP;

    auto fetch = [&](int64_t i) -> float2 {
        if (i >= avail || i < -(int64_t)HALO) return make_float2(0.f, 0.f);
        const float2 *p = (i < r_prev) ? (car + (HALO + i)) : (blk + (i - r_prev));
        return *p;
    };
    auto load_tile = [&](float4 (&r)[4], int64_t s0) {
        if (s0 >= r_prev && s0 + TILE <= avail) {
            const f4a8 *p = (const f4a8 *)(blk + (s0 - r_prev)) + lane;
#pragma unroll
            for (int q = 0; q < 4; q++) { f4a8 v = p[64 * q]; r[q] = make_float4(v.x, v.y, v.z, v.w); }
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float2 u = fetch(s0 + 128 * q + 2 * lane), w = fetch(s0 + 128 * q + 2 * lane + 1);
                r[q] = make_float4(u.x, u.y, w.x, w.y);
            }
        }
    };
    // the 16 samples in front of a tile, two per lane in lanes 56..63 (as if they were the end of a previous tile's r[3])
    auto load_hist = [&](float4 &r, int64_t s0) {
        if (lane >= 56) {
            const int64_t s = s0 - 16 + 2 * (lane - 56);
            if (s0 - 16 >= r_prev && s0 <= avail) { f4a8 v = *(const f4a8 *)(blk + (s - r_prev)); r = make_float4(v.x, v.y, v.z, v.w); }
            else { float2 u = fetch(s), w = fetch(s + 1); r = make_float4(u.x, u.y, w.x, w.y); }
        }
    };

    for (int i = threadIdx.x; i < RW32 + 2; i += 64 * W) s_g[i] = ~0u;
    for (int i = threadIdx.x; i < RW32; i += 64 * W) s_m[i] = 0u;
    if (threadIdx.x < R) s_hit[threadIdx.x] = 0u;
    __syncthreads();

    uint32_t ndet = 0;                           // hits appended for this segment (kept by wave 0, wave-uniform)
    // run starts located in [previous tile word 7, this tile words 0..6] of tile kk (whole wave, rare)
    auto emit_tile = [&](int kk) {
        const int sl = (kk + W) & (R - 1);
        const uint32_t h1 = __builtin_amdgcn_readfirstlane(s_hit[sl]), h0 = __builtin_amdgcn_readfirstlane(s_hit[(sl - 1) & (R - 1)]);
        if (!(h0 | h1)) return;
        const int64_t tt0 = chunk_start + (int64_t)kk * TILE;
        const uint64_t *s_m64 = (const uint64_t *)s_m;
        uint64_t starts = 0, mcur = 0, mnext = 0;
        int64_t relw = 0;
        if (lane < 8) {
            const int ring_w = (sl * (TILE / 64) + lane - 1) & (GW - 1);
            relw = tt0 / 64 + lane - 1;
            const uint64_t mprev = s_m64[(ring_w - 1) & (GW - 1)];
            mcur = s_m64[ring_w];
            mnext = s_m64[(ring_w + 1) & (GW - 1)];
            uint64_t smear = 0;
#pragma unroll
            for (int s = 1; s <= D; s++) smear |= (mcur << s) | (mprev >> (64 - s));
            starts = mcur & ~smear;
            const int64_t absw = (int64_t)(a.n_done / 64) + relw;
            if (absw < 0 || relw + 1 >= words_end) starts = 0;
        }
        uint64_t who = __ballot(starts != 0);
        while (who) {
            const int l = __ffsll((unsigned long long)who) - 1;
            who &= who - 1;
            const int cnt = __popcll(__shfl(starts, l));
            if (lane == l) {
                uint64_t *dst = a.det + ((uint64_t)c * a.max_chunks + chunk) * a.det_cap;
                uint32_t slot_i = ndet;
                while (starts) {
                    int p = __ffsll((unsigned long long)starts) - 1;
                    starts &= starts - 1;
                    uint64_t win = (mcur >> p) | (p ? (mnext << (64 - p)) : 0ull);
                    win &= (1ull << D) - 1ull;
                    int last = 63 - __clzll((long long)win);
                    uint64_t absn = a.n_done + (uint64_t)(relw * 64 + p);
                    if (slot_i < a.det_cap) dst[slot_i] = (absn << 8) | (uint64_t)last;
                    else atomicOr(a.status, 1u);
                    slot_i++;
                }
            }
            ndet += (uint32_t)cnt;
        }
    };

    // step i: wave w works on tile k = -W + W i + w of the segment (k < 0: halo, recomputed, never stored or emitted)
    const int nsteps = 1 + (K + W - 1) / W;
    float4 cur[4], hist = make_float4(0.f, 0.f, 0.f, 0.f);   // the tile in use; reloaded for the next step as soon as P1 has consumed it
    {
        const int64_t s0 = chunk_start + (int64_t)(wv - W) * TILE;
        load_tile(cur, s0);
        load_hist(hist, s0);
    }
    for (int i = 0; i <= nsteps; i++) {                           // the last pass only emits the tiles of step nsteps - 1
        const int k = -W + W * i + wv;
        const bool active = i < nsteps && k < K;                  // wave-uniform
        const int64_t t0 = chunk_start + (int64_t)k * TILE;
        const int slot = (k + W) & (R - 1);
        if (active) {
            // ---- P1: demodulate the tile into LDS, then start the loads of the next step's tile into the same registers ----
            float *const dw = dcur + dw_off;
            {   // history prefix: the values of samples t0-15 .. t0-1 (lane 56's first value, of t0-16, is not used: H <= 15)
                const float pr = shift_in(hist.z, 0.f), pi_ = shift_in(hist.w, 0.f);
                const f2 dd = SL == AMPS_SLICER_SINE ? sine_pair(hist, pr, pi_) : fm_phase_pair(hist, pr, pi_);
                if (lane >= 56) {
                    dcur[didx(2 * (lane - 56))] = dd.x;
                    dcur[didx(2 * (lane - 56)) + 1] = dd.y;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float ex = lane63(q == 0 ? hist.z : cur[q - 1].z);
                const float ey = lane63(q == 0 ? hist.w : cur[q - 1].w);
                const float pr = shift_in(cur[q].z, ex), pi_ = shift_in(cur[q].w, ey);
                const f2 dd = SL == AMPS_SLICER_SINE ? sine_pair(cur[q], pr, pi_) : fm_phase_pair(cur[q], pr, pi_);
                dw[didx(DHIST + 128 * q)] = dd.x;
                dw[didx(DHIST + 128 * q) + 1] = dd.y;
            }
            if (k + W < K) { load_tile(cur, t0 + (int64_t)W * TILE); load_hist(hist, t0 + (int64_t)W * TILE); }
            __builtin_amdgcn_wave_barrier();
            // ---- P2: boxcar over one symbol (aligned pair sums), slice, pack 8 bits per lane ----
            {
                const float *const dr = dcur + dr_off;
                float v[H + 8];
#pragma unroll
                for (int m = 0; m < H + 8; m++) v[m] = dr[didx(DHIST - H + m)];
                unsigned byte = 0;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int lead = (q + H) & 1;
                    const int j0 = q + lead;
                    const int npairs = (q + H - j0 + 1) / 2;
                    const int trail = (q + H - j0 + 1) & 1;
                    float s = lead ? v[q] : (v[j0] + v[j0 + 1]);
#pragma unroll
                    for (int u = lead ? 0 : 1; u < npairs; u++) s = s + (v[j0 + 2 * u] + v[j0 + 2 * u + 1]);
                    if (trail) s = s + v[q + H];
                    if constexpr (SL == AMPS_SLICER_SINE) byte |= (~__float_as_uint(s) >> 31) << q;
                    else byte |= (s >= 0.0f ? 1u : 0u) << q;
                }
                ((uint8_t *)s_g)[slot * (TILE / 8) + lane] = (uint8_t)byte;
                if (slot == 0 && lane < 8) ((uint8_t *)s_g)[RW32 * 4 + lane] = (uint8_t)byte;   // mirror of dwords 0,1
            }
            __builtin_amdgcn_wave_barrier();
            if (part == 0 && k >= 0) {                            // publish the tile's slicer words
                const int64_t relw = t0 / 64 + (wq >> 1);
                if (relw < words_end) {
                    const uint64_t absw = a.n_done / 64 + (uint64_t)relw;
                    uint32_t *g32 = (uint32_t *)(a.gring + (uint64_t)c * a.ring_words + (absw & a.ring_mask));
                    g32[wq & 1] = s_g[slot * (TILE / 32) + wq];
                }
            }
        }
        __syncthreads();                                          // the bits of this step's W tiles are in the ring
        if (active) {
            // ---- P3a: bit-parallel exact match of the 74-symbol trigger ----
            auto and_quad = [&](uint32_t x) -> uint32_t {
                x &= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);
                x &= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);
                return x;
            };
            uint32_t acc = ~0u;
#pragma unroll
            for (int u = 0; u < 4; u++) {                         // prefilter: the last 16 symbols, 4 taps per lane
                const int off = 32 * wq - SPS * (part + 4 * u);   // may be negative: previous tiles
                const int qd = ((off >> 5) + slot * (TILE / 32)) & (RW32 - 1);
                const uint32_t lo = s_g[qd], hi = s_g[qd + 1];
                acc &= __builtin_amdgcn_alignbit(hi, lo, off & 31) ^ trig_xor(TRIG - 1 - part - 4 * u);
            }
            acc = and_quad(acc);
            bool hit = false;
            if (__ballot(acc != 0)) {                             // rare: all 74 taps
                const int bitbase = slot * TILE + 32 * wq;
                acc = ~0u;
#pragma unroll 1
                for (int ii = part; ii < TRIG; ii += 4) {
                    const int B = (bitbase - SPS * (TRIG - 1 - ii)) & (R * TILE - 1);
                    const int qd = B >> 5;
                    acc &= __builtin_amdgcn_alignbit(s_g[qd + 1], s_g[qd], B & 31) ^ trig_xor(ii);
                }
                acc = and_quad(acc);
                hit = __ballot(acc != 0) != 0;
            }
            if (part == 0) s_m[slot * (TILE / 32) + wq] = acc;
            if (lane == 0) s_hit[slot] = hit ? 1u : 0u;
        }
        // ---- P3b, one step late: the match words of the previous step's tiles and of their neighbours are complete ----
        if (wv == 0 && i > 0) {
            const uint64_t any = __ballot(lane < R && s_hit[lane < R ? lane : 0] != 0u);   // R <= 64 slots: one LDS read per step
            if (any) {
#pragma unroll 1
                for (int w2 = 0; w2 < W; w2++) {
                    const int kk = -W + W * (i - 1) + w2;
                    if (kk >= 0 && kk < K) emit_tile(kk);
                }
            }
        }
    }
    if (wv == 0 && lane == 0) a.detcount[(uint64_t)c * a.max_chunks + chunk] = ndet < a.det_cap ? ndet : a.det_cap;
    __syncthreads();                                              // the rings are re-initialised for the next segment
  }
}

} // namespace amps
