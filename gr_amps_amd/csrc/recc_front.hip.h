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
// Mapping to the hardware
//   * grid = (chunks, channels); a 256-thread workgroup (4 waves) walks one chunk of one channel in
//     2048-sample tiles, so a launch has chunks*channels >> 256 workgroups and every CU streams.
//   * HBM reads: each wave owns a 512-sample strip of the tile and issues eight 512-byte coalesced
//     `global_load_dwordx2` per tile (one fc32 sample per lane), all eight for tile k+1 in flight
//     while tile k is processed (register double buffer) -- IQ is read exactly once, plus one halo
//     tile per chunk (3 % at 32 tiles/chunk).
//   * LDS: demod floats are staged in a padded (stride 9/8) array so the contiguous 17-float window
//     every thread needs for 8 boxcar outputs is bank-conflict free; slicer bits live in a 4-tile
//     LDS bit ring that is the sliding window of the trigger correlator (730 bits of history).
//   * the correlator is bit-parallel: a lane tests 64 consecutive sample phases against one tap of
//     the 74-symbol pattern with one funnel shift + xnor; 8 lanes share a 64-phase word and combine
//     with three `__shfl_xor` AND steps.  Hits are rare, so the emit path is a wave-uniform branch.
//   * HBM writes: 1 bit per sample of slicer output (1.6 % of the read volume) into a per-channel
//     ring that the capture/decode kernel reads, plus 8 bytes per trigger hit.
// No MFMA: there is no dense contraction on this path; it is HBM-bound (~45 VALU ops per 8-byte sample).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "amps_recc.h"
#include "amps_recc_numerics.h"

namespace amps {

constexpr int TILE = AMPS_TILE_SAMPLES;      // 2048
constexpr int HALO = AMPS_HALO_SAMPLES;      // 2048
constexpr int CARRY_CAP = HALO + 64;         // samples kept per channel between pushes
constexpr int TRIG = AMPS_RECC_TRIGGER_SYMS; // 74

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
    uint32_t tiles_per_chunk;
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

__device__ __forceinline__ int didx(int n) { return n + (n >> 3); } // padded LDS index of tile-local sample n

template <int SPS>
__global__ __launch_bounds__(256) void recc_front_kernel(FrontArgs a)
{
    static_assert(SPS >= 2 && SPS <= 16, "samples per symbol");
    constexpr int H = SPS - 1;               // boxcar history
    constexpr int D = AMPS_DEDUP_SYMBOLS * SPS; // dedup / run window in samples (<= 32)
    __shared__ float    s_d[TILE + TILE / 8];       // padded demod floats of the current tile
    __shared__ float    s_dhist[2][16];             // last 16 demod floats of the previous tile
    __shared__ uint64_t s_g[4 * TILE / 64];         // slicer bit ring, 4 tiles
    __shared__ uint64_t s_m[4 * TILE / 64];         // trigger-hit bit ring, 4 tiles
    __shared__ uint32_t s_ndet;

    const int c = blockIdx.y, chunk = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t chunk_start = (int64_t)chunk * a.tiles_per_chunk * TILE;   // rel
    if (chunk_start >= (int64_t)a.P) return;
    int64_t chunk_len = (int64_t)a.P - chunk_start;
    if (chunk_len > (int64_t)a.tiles_per_chunk * TILE) chunk_len = (int64_t)a.tiles_per_chunk * TILE;
    const int K = (int)((chunk_len + TILE - 1) / TILE);  // real tiles in this chunk
    const int64_t words_end = (int64_t)a.P / 64;          // rel word index limit of this launch

    const float2 *blk = a.block + (uint64_t)c * a.ld;
    const float2 *car = a.carry + (uint64_t)c * CARRY_CAP;
    const int r_prev = (int)a.r_prev, avail = (int)a.avail;

    auto fetch = [&](int64_t i) -> float2 {   // virtual stream: carry then block; zero beyond the data
        if (i >= avail) return make_float2(0.f, 0.f);
        const float2 *p = (i < r_prev) ? (car + (HALO + i)) : (blk + (i - r_prev));
        return *p;
    };

    if (tid == 0) s_ndet = 0;
    for (int i = tid; i < 4 * TILE / 64; i += 256) { s_g[i] = ~0ull; s_m[i] = 0; }
    if (tid < 32) ((float *)s_dhist)[tid] = 0.f;

    float2 cur[8], nxt[8], cur_edge, nxt_edge;
    {
        const int64_t t0 = chunk_start - TILE + 512 * wv;
#pragma unroll
        for (int j = 0; j < 8; j++) cur[j] = fetch(t0 + 64 * j + lane);
        cur_edge = fetch(t0 - 1);
    }
    __syncthreads();

    // k = 0 is the halo tile [chunk_start-2048, chunk_start): recomputed, never stored or emitted
    for (int k = 0; k <= K; k++) {
        const int64_t t0 = chunk_start + (int64_t)(k - 1) * TILE;  // rel start of this tile
        // ---- P1: prefetch next tile, demodulate this one into LDS ----
        if (k < K) {
            const int64_t n0 = t0 + TILE + 512 * wv;
#pragma unroll
            for (int j = 0; j < 8; j++) nxt[j] = fetch(n0 + 64 * j + lane);
            nxt_edge = fetch(n0 - 1);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float pr = __shfl_up(cur[j].x, 1), pi_ = __shfl_up(cur[j].y, 1);
            float er, ei;
            if (j == 0) { er = cur_edge.x; ei = cur_edge.y; }
            else { er = __shfl(cur[j - 1].x, 63); ei = __shfl(cur[j - 1].y, 63); }
            if (lane == 0) { pr = er; pi_ = ei; }
            s_d[didx(512 * wv + 64 * j + lane)] = fm_phase(cur[j].x, cur[j].y, pr, pi_);
        }
        __syncthreads();
        // ---- P2: boxcar over one symbol, slice, pack 8 bits per thread ----
        {
            float v[H + 8];
            const int base = 8 * tid - H;
#pragma unroll
            for (int m = 0; m < H + 8; m++) {
                int n = base + m;
                v[m] = n >= 0 ? s_d[didx(n)] : s_dhist[k & 1][16 + n];
            }
            unsigned byte = 0;
            const bool tap = a.dbg_d && (uint32_t)c == a.dbg_channel && k > 0;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                float s = v[q];
#pragma unroll
                for (int r = 1; r < SPS; r++) s = s + v[q + r];  // oldest -> newest
                byte |= (s >= 0.0f ? 1u : 0u) << q;
                if (tap) {
                    int64_t rel = t0 + 8 * tid + q;
                    if (rel < (int64_t)a.P) { a.dbg_d[rel] = v[q + H]; a.dbg_S[rel] = s; }
                }
            }
            ((uint8_t *)s_g)[(k & 3) * (TILE / 8) + tid] = (uint8_t)byte;
            if (tid >= 254) {
#pragma unroll
                for (int q = 0; q < 8; q++) s_dhist[(k + 1) & 1][(tid - 254) * 8 + q] = v[H + q];
            }
        }
        __syncthreads();
        // ---- P3a: bit-parallel exact match of the 74-symbol trigger; publish slicer words ----
        {
            const int wl = wv * 8 + (lane & 7);   // word within tile (0..31)
            const int part = lane >> 3;           // 8 lanes share a word, each takes every 8th tap
            uint64_t acc = ~0ull;
            const int bitbase = (k & 3) * TILE + 64 * wl;
            for (int i = part; i < TRIG; i += 8) {
                int B = (bitbase - SPS * (TRIG - 1 - i)) & (4 * TILE - 1);
                int qw = B >> 6, sh = B & 63;
                uint64_t lo = s_g[qw], hi = s_g[(qw + 1) & (4 * TILE / 64 - 1)];
                uint64_t val = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
                uint64_t t = (i < 64 ? (TRIG_LO >> i) : (TRIG_HI >> (i - 64))) & 1ull;
                acc &= t ? val : ~val;
            }
            acc &= __shfl_xor(acc, 8);
            acc &= __shfl_xor(acc, 16);
            acc &= __shfl_xor(acc, 32);
            if (part == 0) {
                s_m[(k & 3) * (TILE / 64) + wl] = acc;
                const int64_t relw = t0 / 64 + wl;      // rel word index (t0 is a multiple of 64)
                if (k > 0 && relw < words_end) {
                    uint64_t absw = a.n_done / 64 + (uint64_t)relw;
                    a.gring[(uint64_t)c * a.ring_words + (absw & a.ring_mask)] = s_g[(k & 3) * (TILE / 64) + wl];
                }
            }
        }
        __syncthreads();
        // ---- P3b: emit run starts located in [previous tile word 31, this tile words 0..30] ----
        if (k > 0 && wv == 0) {
            uint64_t starts = 0, mcur = 0, mnext = 0;
            int64_t relw = 0;
            if (lane < 32) {
                const int ring_w = ((k & 3) * (TILE / 64) + lane - 1) & (4 * TILE / 64 - 1); // word examined
                relw = t0 / 64 + lane - 1;
                const uint64_t mprev = s_m[(ring_w - 1) & (4 * TILE / 64 - 1)];
                mcur = s_m[ring_w];
                mnext = s_m[(ring_w + 1) & (4 * TILE / 64 - 1)];
                uint64_t smear = 0;
#pragma unroll
                for (int s = 1; s <= D; s++) smear |= (mcur << s) | (mprev >> (64 - s));
                starts = mcur & ~smear;
                // the examined word and its look-ahead word must both be processed data of this stream
                const int64_t absw = (int64_t)(a.n_done / 64) + relw;
                if (absw < 0 || relw + 1 >= words_end) starts = 0;
            }
            if (__ballot(starts != 0)) {              // rare: wave-uniform slow path, ordered append
                int cnt = __popcll(starts);
                int incl = cnt;
#pragma unroll
                for (int s = 1; s < 64; s <<= 1) { int o = __shfl_up(incl, s); if (lane >= s) incl += o; }
                int excl = incl - cnt;
                const uint32_t base = s_ndet;
                uint64_t *dst = a.det + ((uint64_t)c * a.max_chunks + chunk) * a.det_cap;
                int slot = (int)base + excl;
                while (starts) {
                    int p = __ffsll((unsigned long long)starts) - 1;
                    starts &= starts - 1;
                    uint64_t win = (mcur >> p) | (p ? (mnext << (64 - p)) : 0ull);
                    win &= (1ull << D) - 1ull;
                    int last = 63 - __clzll((long long)win);
                    uint64_t absn = a.n_done + (uint64_t)(relw * 64 + p);
                    if (slot < (int)a.det_cap) dst[slot] = (absn << 8) | (uint64_t)last;
                    else atomicOr(a.status, 1u);
                    slot++;
                }
                int total = __shfl(incl, 63);
                if (lane == 0) s_ndet = base + (uint32_t)total;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) cur[j] = nxt[j];
        cur_edge = nxt_edge;
        // no barrier needed here: the next P1 only writes s_d (last read before the P2/P3 barriers)
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t n = s_ndet < a.det_cap ? s_ndet : a.det_cap;
        a.detcount[(uint64_t)c * a.max_chunks + chunk] = n;
    }
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
