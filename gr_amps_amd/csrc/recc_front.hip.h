// recc_front.hip.h -- the fused streaming kernel of the IQ seam for gfx950 (MI355X, wave64).
//
// One pass over interleaved fc32 IQ does what the reference wires as four GNU Radio blocks plus the
// search loop of its own recc block:
//   analog_quadrature_demod_cf          grc/recctest.grc:458      -> FM discriminator (numeric spec:
//                                                                    include/amps_recc_numerics.h)
//   (post-detection low-pass)                                      -> boxcar over one Manchester symbol
//   digital_clock_recovery_mm_ff        grc/recctest.grc:846-874  -> replaced by feed-forward timing:
//                                                                    every sample phase is sliced and
//                                                                    tested; the run of matching phases
//                                                                    is resolved later (recc_resolve)
//   digital_binary_slicer_fb            grc/recctest.grc:807      -> g[n] = S[n] >= 0
//   recc_impl::work trigger memmem      lib/recc_impl.cc:115-119  -> exact 74-symbol match, bit-parallel
//                                                                    over 64 sample phases per lane
//
// Mapping to the hardware ("wave-streams")
//   * every 64-lane wavefront is an independent stream processor: it owns one time-chunk of one
//     channel and walks it in 512-sample tiles.  Nothing is shared between the 4 waves of a
//     workgroup, so there is no s_barrier anywhere; a launch has chunks*channels >> 256*16 waves.
//   * HBM reads: four coalesced 1 KiB `global_load_dwordx4` per tile (two fc32 samples per lane) on
//     a wave-uniform fast path with no per-lane branches; the loads of tile k+1 are in flight while
//     tile k is processed (register double buffer).  IQ is read exactly once, plus a 1024-sample
//     halo per chunk (2-6 %).  Only tiles that touch the inter-push carry or the end of the data take
//     the generic (branchy) path.
//   * the x[n-1] neighbour comes from the lane's own float4 (odd samples) or one `__shfl_up` (even).
//   * LDS (private to the wave): demod floats go through a padded (stride 9/8) 2-tile ring so the
//     contiguous window each lane needs for 8 boxcar outputs is bank-conflict free; slicer bits live
//     in a 2048-bit ring that is the sliding window of the trigger correlator (730 bits of history).
//   * the correlator is bit-parallel: a lane tests 64 consecutive sample phases against one tap of
//     the 74-symbol pattern with one funnel shift + xnor.  8 lanes share a 64-phase word; a 16-tap
//     prefilter on the word-sync symbols (2 taps per lane, three `__shfl_xor` AND steps, one ballot)
//     rejects noise with probability 1-2^-16 per phase; only then are all 74 taps evaluated.
//   * HBM writes: 1 bit per sample of slicer output (1.6 % of the read volume) into a per-channel
//     ring that the capture/decode kernel reads, plus 8 bytes per trigger hit.
// No MFMA: there is no dense contraction on this path; it is HBM-bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "amps_recc.h"
#include "amps_recc_numerics.h"

namespace amps {

constexpr int TILE = AMPS_TILE_SAMPLES;      // 512 samples per wave tile
constexpr int HALO = AMPS_HALO_SAMPLES;      // 1024 = 2 tiles of history per chunk / push
constexpr int CARRY_CAP = HALO + 64;         // samples kept per channel between pushes
constexpr int TRIG = AMPS_RECC_TRIGGER_SYMS; // 74
static_assert(HALO == 2 * TILE, "halo is two wave tiles");

// trigger symbols, lib/recc_impl.cc:76 Manchester coded (bit i of the pair = symbol i)
// "1010101010101010101010101011100010010" -> symbols 01 10 01 10 ...
constexpr uint64_t make_trig(int lo)
{
    const char *bits = "1010101010101010101010101011100010010";
    uint64_t v = 0;
    for (int i = 0; i < 64; i++) {
        int s = lo + i;
        if (s >= 74) break;
        int bit = bits[s / 2] - '0';
        int sym = (s & 1) ? bit : 1 - bit; // '1' -> (0,1), '0' -> (1,0)
        v |= (uint64_t)sym << i;
    }
    return v;
}
constexpr uint64_t TRIG_LO = make_trig(0);   // symbols 0..63
constexpr uint64_t TRIG_HI = make_trig(64);  // symbols 64..73

struct FrontArgs {
    const float2 *block;     // [C][ld] new samples of this push
    const float2 *carry;     // [C][CARRY_CAP]: samples [n_done-HALO, n_done+r_prev)
    uint64_t ld;
    uint32_t r_prev;         // leftover samples of the previous push held in carry after the halo
    uint32_t avail;          // r_prev + nsamp
    uint32_t P;              // samples processed by this launch (multiple of 64)
    uint32_t tiles_per_chunk;// wave tiles per wave chunk
    uint64_t n_done;         // absolute index of rel sample 0 (multiple of 64)
    uint64_t *gring;         // [C][ring_words] slicer bits, word = abs_sample/64 & ring_mask
    uint32_t ring_mask;      // ring_words - 1
    uint32_t ring_words;
    uint64_t *det;           // [C][max_chunks][det_cap]  (abs_sample << 8 | run_len-1), ordered
    uint32_t *detcount;      // [C][max_chunks]
    uint32_t max_chunks;
    uint32_t det_cap;
    uint32_t *status;        // bit 0: detection list overflow
    float    *dbg_d;         // optional taps for channel dbg_channel: d and S of rel samples [0,P)
    float    *dbg_S;
    uint32_t dbg_channel;
};

typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));  // two fc32 samples, 8-byte aligned

__device__ __forceinline__ float fm_phase(float xr, float xi, float pr, float pi_)
{
    float re = __builtin_fmaf(xr, pr, xi * pi_);
    float im = __builtin_fmaf(xi, pr, -(xr * pi_));
    float ax = __builtin_fabsf(re), ay = __builtin_fabsf(im);
    float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    float q = mx > 0.0f ? mn / mx : 0.0f;
    float z = q * q;
    float p = AMPS_ATAN_C5;
    p = __builtin_fmaf(p, z, AMPS_ATAN_C4);
    p = __builtin_fmaf(p, z, AMPS_ATAN_C3);
    p = __builtin_fmaf(p, z, AMPS_ATAN_C2);
    p = __builtin_fmaf(p, z, AMPS_ATAN_C1);
    p = __builtin_fmaf(p, z, AMPS_ATAN_C0);
    float a = p * q;
    if (ay > ax) a = AMPS_PI_2_F - a;
    if (re < 0.0f) a = AMPS_PI_F - a;
    if (im < 0.0f) a = -a;
    return a;
}

constexpr int DRING = 2 * TILE;                                   // demod ring: 2 tiles
__device__ __forceinline__ int didx(int n) { return n + (n >> 3); } // padded LDS index, n in [0, DRING)

template <int SPS>
__global__ __launch_bounds__(256) void recc_front_kernel(FrontArgs a)
{
    static_assert(SPS >= 2 && SPS <= 16, "samples per symbol");
    constexpr int H = SPS - 1;                  // boxcar history
    constexpr int D = AMPS_DEDUP_SYMBOLS * SPS; // dedup / run window in samples (<= 32)
    constexpr int GW = 4 * TILE / 64;           // words in the per-wave bit rings (4 tiles)
    __shared__ float    s_d_all[4][DRING + DRING / 8];
    __shared__ uint64_t s_g_all[4][GW];
    __shared__ uint64_t s_m_all[4][GW];

    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.y;
    const int chunk = blockIdx.x * 4 + wv;
    const int64_t chunk_start = (int64_t)chunk * a.tiles_per_chunk * TILE;   // rel
    if (chunk_start >= (int64_t)a.P) return;     // whole wave leaves; no barriers are used below
    int64_t chunk_len = (int64_t)a.P - chunk_start;
    if (chunk_len > (int64_t)a.tiles_per_chunk * TILE) chunk_len = (int64_t)a.tiles_per_chunk * TILE;
    const int K = (int)((chunk_len + TILE - 1) / TILE);  // real tiles in this chunk
    const int64_t words_end = (int64_t)a.P / 64;          // rel word index limit of this launch

    float *s_d = s_d_all[wv];
    uint64_t *s_g = s_g_all[wv];
    uint64_t *s_m = s_m_all[wv];
    const float2 *blk = a.block + (uint64_t)c * a.ld;
    const float2 *car = a.carry + (uint64_t)c * CARRY_CAP;
    const int r_prev = (int)a.r_prev, avail = (int)a.avail;

    auto fetch = [&](int64_t i) -> float2 {   // generic path: carry then block; zero outside the data
        if (i >= avail || i < -(int64_t)HALO) return make_float2(0.f, 0.f);
        const float2 *p = (i < r_prev) ? (car + (HALO + i)) : (blk + (i - r_prev));
        return *p;
    };
    // one 512-sample tile: r[q] = samples (s0 + 128q + 2*lane, +1)
    auto load_tile = [&](float4 (&r)[4], int64_t s0) {
        if (s0 >= r_prev && s0 + TILE <= avail) {          // wave-uniform: entirely inside the new block
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

    if (lane < GW) { s_g[lane] = ~0ull; s_m[lane] = 0; }
    for (int i = lane; i < DRING + DRING / 8; i += 64) s_d[i] = 0.f;

    float4 cur[4], nxt[4];
    float last_x = 0.f, last_y = 0.f;            // last sample of the previous tile (wave-uniform)
    uint32_t ndet = 0;                           // hits appended by this wave (wave-uniform)
    load_tile(cur, chunk_start - HALO);

    // k = -2, -1 are the halo tiles [chunk_start-1024, chunk_start): recomputed, never stored or emitted
    for (int k = -2; k < K; k++) {
        const int64_t t0 = chunk_start + (int64_t)k * TILE;   // rel start of this tile
        const int slot = (k + 2) & 3;                          // bit-ring slot of this tile
        const int dbase = ((k + 2) & 1) * TILE;                // demod-ring base of this tile
        // ---- P1: prefetch the next tile, demodulate this one into LDS ----
        if (k + 1 < K) load_tile(nxt, t0 + TILE);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float pr = __shfl_up(cur[q].z, 1), pi_ = __shfl_up(cur[q].w, 1);
            float er, ei;
            if (q == 0) { er = last_x; ei = last_y; }
            else { er = __shfl(cur[q - 1].z, 63); ei = __shfl(cur[q - 1].w, 63); }
            if (lane == 0) { pr = er; pi_ = ei; }
            const int n = dbase + 128 * q + 2 * lane;
            s_d[didx(n)] = fm_phase(cur[q].x, cur[q].y, pr, pi_);
            s_d[didx(n + 1)] = fm_phase(cur[q].z, cur[q].w, cur[q].x, cur[q].y);
        }
        last_x = __shfl(cur[3].z, 63);
        last_y = __shfl(cur[3].w, 63);
        __builtin_amdgcn_wave_barrier();
        // ---- P2: boxcar over one symbol (aligned pair sums), slice, pack 8 bits per lane ----
        {
            float v[H + 8];
#pragma unroll
            for (int m = 0; m < H + 8; m++) v[m] = s_d[didx((dbase + 8 * lane - H + m) & (DRING - 1))];
            unsigned byte = 0;
            const bool tap = a.dbg_d && (uint32_t)c == a.dbg_channel && k >= 0;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                // window v[q .. q+H]; v[j] has absolute parity (j + H) & 1 because 8*lane is even
                const int lead = (q + H) & 1;                 // window starts on an odd sample
                const int j0 = q + lead;
                const int npairs = (q + H - j0 + 1) / 2;
                const int trail = (q + H - j0 + 1) & 1;       // window ends on an even sample
                float s = lead ? v[q] : (v[j0] + v[j0 + 1]);
#pragma unroll
                for (int u = lead ? 0 : 1; u < npairs; u++) s = s + (v[j0 + 2 * u] + v[j0 + 2 * u + 1]);
                if (trail) s = s + v[q + H];
                byte |= (s >= 0.0f ? 1u : 0u) << q;
                if (tap) {
                    int64_t rel = t0 + 8 * lane + q;
                    if (rel < (int64_t)a.P) { a.dbg_d[rel] = v[q + H]; a.dbg_S[rel] = s; }
                }
            }
            ((uint8_t *)s_g)[slot * (TILE / 8) + lane] = (uint8_t)byte;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- P3a: bit-parallel exact match of the 74-symbol trigger; publish slicer words ----
        const int wl = lane & 7;             // word within the tile (0..7)
        const int part = lane >> 3;          // 8 lanes share a word
        {
            const int bitbase = slot * TILE + 64 * wl;
            auto tap_word = [&](int i) -> uint64_t {   // 64 phases of tap i, 1 = symbol matches
                int B = (bitbase - SPS * (TRIG - 1 - i)) & (4 * TILE - 1);
                int qw = B >> 6, sh = B & 63;
                uint64_t lo = s_g[qw], hi = s_g[(qw + 1) & (GW - 1)];
                uint64_t val = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
                uint64_t t = (i < 64 ? (TRIG_LO >> i) : (TRIG_HI >> (i - 64))) & 1ull;
                return t ? val : ~val;
            };
            auto and_parts = [&](uint64_t x) -> uint64_t {
                x &= __shfl_xor(x, 8);
                x &= __shfl_xor(x, 16);
                x &= __shfl_xor(x, 32);
                return x;
            };
            // prefilter: the last 16 symbols (all inside the word-sync part), 2 taps per lane
            uint64_t acc = and_parts(tap_word(TRIG - 1 - part) & tap_word(TRIG - 9 - part));
            if (__ballot(acc != 0)) {        // rare: evaluate all 74 taps
                acc = ~0ull;
                for (int i = part; i < TRIG; i += 8) acc &= tap_word(i);
                acc = and_parts(acc);
            }
            if (part == 0) {
                s_m[slot * (TILE / 64) + wl] = acc;
                const int64_t relw = t0 / 64 + wl;      // rel word index (t0 is a multiple of 64)
                if (k >= 0 && relw < words_end) {
                    uint64_t absw = a.n_done / 64 + (uint64_t)relw;
                    a.gring[(uint64_t)c * a.ring_words + (absw & a.ring_mask)] = s_g[slot * (TILE / 64) + wl];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- P3b: emit run starts located in [previous tile word 7, this tile words 0..6] ----
        if (k >= 0) {
            uint64_t starts = 0, mcur = 0, mnext = 0;
            int64_t relw = 0;
            if (lane < 8) {
                const int ring_w = (slot * (TILE / 64) + lane - 1) & (GW - 1);   // word examined
                relw = t0 / 64 + lane - 1;
                const uint64_t mprev = s_m[(ring_w - 1) & (GW - 1)];
                mcur = s_m[ring_w];
                mnext = s_m[(ring_w + 1) & (GW - 1)];
                uint64_t smear = 0;
#pragma unroll
                for (int s = 1; s <= D; s++) smear |= (mcur << s) | (mprev >> (64 - s));
                starts = mcur & ~smear;
                // the examined word and its look-ahead word must both be processed data of this stream
                const int64_t absw = (int64_t)(a.n_done / 64) + relw;
                if (absw < 0 || relw + 1 >= words_end) starts = 0;
            }
            uint64_t who = __ballot(starts != 0);
            while (who) {                                 // rare: ordered append, lane by lane
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
        }
#pragma unroll
        for (int q = 0; q < 4; q++) cur[q] = nxt[q];
    }
    if (lane == 0) a.detcount[(uint64_t)c * a.max_chunks + chunk] = ndet < a.det_cap ? ndet : a.det_cap;
}

// carry[c][k] = V(P - HALO + k) for k in [0, HALO + r_new): the halo the next push recomputes from
struct CarryArgs {
    const float2 *block, *carry_in;
    float2 *carry_out;
    uint64_t ld;
    uint32_t r_prev, avail, P, r_new;
};
__global__ __launch_bounds__(256) void recc_carry_kernel(CarryArgs a)
{
    const int c = blockIdx.y;
    const int n = HALO + (int)a.r_new;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
        int64_t i = (int64_t)a.P - HALO + k;
        float2 v;
        if (i < (int64_t)a.r_prev) v = a.carry_in[(uint64_t)c * CARRY_CAP + HALO + i];
        else v = a.block[(uint64_t)c * a.ld + (i - a.r_prev)];
        a.carry_out[(uint64_t)c * CARRY_CAP + k] = v;
    }
}

} // namespace amps
