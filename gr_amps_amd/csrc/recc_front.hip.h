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
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "amps_recc.h"
#include "amps_recc_numerics.h"

namespace amps {

#ifndef AMPS_FRONT_NT
#define AMPS_FRONT_NT 1                        // the tiles are read exactly once: non-temporal loads (measured 832 x 2^18: spec C 0.364 -> 0.327 ms, spec A 0.373 -> 0.349)
#endif
#ifndef AMPS_FRONT_D2_BLOCKS
#define AMPS_FRONT_D2_BLOCKS 4                 // workgroups per CU the depth-2 / depth-3 instantiations are compiled for (register budget 512 / blocks).
                                               // Round 4: with the rare paths' lane addresses no longer hoisted, depth 2 fits 128 registers: four waves per SIMD AND two
                                               // tiles in flight (832 x 2^18, same box: B 0.3118 -> 0.3038 ms, C 0.3122 -> 0.3055, D 0.3208 -> 0.3174)
#endif
#ifndef AMPS_FRONT_D3_BLOCKS
#define AMPS_FRONT_D3_BLOCKS 3
#endif
#ifndef AMPS_FRONT_D1_BLOCKS
#define AMPS_FRONT_D1_BLOCKS 4                 // depth 1: 125 registers as written; compiled for five (96 registers) the tile loop spills (round 6 probe, profiles/EXPERIMENTS.md)
#endif
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
    uint32_t tiles_per_channel; // Tc = ceil(P / 512)
    uint32_t n_channels;
    uint32_t span;           // tiles per wave: wave w owns global tiles [w*span, (w+1)*span) of the C*Tc tile space
    uint64_t n_done;         // absolute index of rel sample 0 (multiple of 64)
    uint64_t *gring;         // [C][ring_words] slicer bits, word = abs_sample/64 & ring_mask
    uint32_t ring_mask;      // ring_words - 1
    uint32_t ring_words;
    uint64_t *det;           // [C][max_chunks][det_cap]  (abs_sample << 8 | run_len-1), ordered;
    uint32_t *detcount;      // [C][max_chunks]           "chunk" = k-th wave segment of the channel
    uint32_t max_chunks;
    uint32_t det_cap;
    uint32_t *status;        // bit 0: detection list overflow
    float    *dbg_d;         // optional taps for channel dbg_channel: d and S of rel samples [0,P)
    float    *dbg_S;
    uint32_t dbg_channel;
    uint32_t tol;            // TOL kernels: accepted mismatching trigger symbols (cfg.sync_tolerance)
    uint32_t force_ones;     // spec B: rel samples [0, force_ones) have no partner yet (stream start): g = 1
    uint32_t *zero1;         // housekeeping done by thread 0 of the launch instead of separate memsets on the stream: one dword to
    uint32_t *zero2;         // clear (the capture queue count) and an optional triple (the idle record list's {count, status, published count})
    float2   *carry_out;     // optional: the NEXT push's carry, [C][CARRY_CAP], written by this launch (a slice per wave) instead of by
    uint32_t carry_n;        // recc_carry_kernel behind it: carry_out[c][k] = sample P - HALO + k, k < carry_n = HALO + leftover
};
__device__ __forceinline__ void front_housekeeping(const FrontArgs &a)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (a.zero1) *a.zero1 = 0u;
        if (a.zero2) { a.zero2[0] = 0u; a.zero2[1] = 0u; a.zero2[2] = 0u; }
    }
}

typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));  // two fc32 samples, 8-byte aligned

typedef float f2 __attribute__((ext_vector_type(2)));

// The discriminator of include/amps_recc_numerics.h for the two samples a lane holds (A = s.xy with
// predecessor p, B = s.zw with predecessor A), written on 2-vectors so that hipcc emits v_pk_mul /
// v_pk_fma / v_pk_add: measured on MI355X a packed fp32 op issues in ~4.8 cycles per wave against
// ~4.3 for a scalar one (scripts/ubench_pk.hip), i.e. 1.8x the arithmetic per issue slot, and this
// kernel is VALU-issue bound.  ~19 instructions per sample instead of ~30.
// re, im = the real and imaginary parts of TWO conj-products (planar: element 0 in .x, element 1 in .y); every operation
// after them is packed over the two
__device__ __forceinline__ f2 fm_phase_planar(f2 re, f2 im)
{
    const float axa = __builtin_fabsf(re.x), aya = __builtin_fabsf(im.x);
    const float axb = __builtin_fabsf(re.y), ayb = __builtin_fabsf(im.y);
    const f2 mx = { __builtin_fmaxf(__builtin_fmaxf(axa, aya), AMPS_MX_FLOOR), __builtin_fmaxf(__builtin_fmaxf(axb, ayb), AMPS_MX_FLOOR) };
    const f2 mn = { __builtin_fminf(axa, aya), __builtin_fminf(axb, ayb) };
    f2 r = { __uint_as_float(AMPS_RCP_MAGIC - __float_as_uint(mx.x)), __uint_as_float(AMPS_RCP_MAGIC - __float_as_uint(mx.y)) };
    const f2 one = { 1.0f, 1.0f };
    f2 e;
    e = __builtin_elementwise_fma(-mx, r, one); r = __builtin_elementwise_fma(r, e, r);
    e = __builtin_elementwise_fma(-mx, r, one); r = __builtin_elementwise_fma(r, e, r);
    e = __builtin_elementwise_fma(-mx, r, one); r = __builtin_elementwise_fma(r, e, r);
    const f2 q = mn * r;
    const f2 z = q * q;
    f2 p = { AMPS_ATAN_C5, AMPS_ATAN_C5 };
    p = __builtin_elementwise_fma(p, z, (f2){ AMPS_ATAN_C4, AMPS_ATAN_C4 });
    p = __builtin_elementwise_fma(p, z, (f2){ AMPS_ATAN_C3, AMPS_ATAN_C3 });
    p = __builtin_elementwise_fma(p, z, (f2){ AMPS_ATAN_C2, AMPS_ATAN_C2 });
    p = __builtin_elementwise_fma(p, z, (f2){ AMPS_ATAN_C1, AMPS_ATAN_C1 });
    p = __builtin_elementwise_fma(p, z, (f2){ AMPS_ATAN_C0, AMPS_ATAN_C0 });
    f2 a = p * q;
    const f2 a_swapped = (f2){ AMPS_PI_2_F, AMPS_PI_2_F } - a;
    a.x = aya > axa ? a_swapped.x : a.x;
    a.y = ayb > axb ? a_swapped.y : a.y;
    const f2 a_reflect = (f2){ AMPS_PI_F, AMPS_PI_F } - a;
    a.x = re.x < 0.0f ? a_reflect.x : a.x;
    a.y = re.y < 0.0f ? a_reflect.y : a.y;
    return (f2){ __builtin_copysignf(a.x, im.x), __builtin_copysignf(a.y, im.y) };
}
// N independent planar evaluations written in lock step (same operations, same order per value, so every result is
// bit-identical to a separate fm_phase_planar call): the Newton and Horner chains are serial and every packed operation that
// consumes the previous one costs a wait state, so one chain at a time runs at a third of the issue rate (the compiler keeps
// the source order: it pads the single chain with s_nop instead of interleaving)
// an empty asm that "uses and redefines" the N values of one row: every chain has reached this row before any goes on
template <int N> __device__ __forceinline__ void fm_row_join(f2 (&v)[N])
{
    static_assert(N == 4 || N == 2, "rows of two or four");
    if constexpr (N == 4) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
    else asm volatile("" : "+v"(v[0]), "+v"(v[1]));
}
template <int N>
__device__ __forceinline__ void fm_phase_planar_n(const f2 (&re)[N], const f2 (&im)[N], f2 (&out)[N])
{
    f2 mx[N], mn[N], r[N], e[N], q[N], z[N], p[N], a[N];
    bool sw[N][2];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const float axa = __builtin_fabsf(re[i].x), aya = __builtin_fabsf(im[i].x), axb = __builtin_fabsf(re[i].y), ayb = __builtin_fabsf(im[i].y);
        mx[i] = (f2){ __builtin_fmaxf(__builtin_fmaxf(axa, aya), AMPS_MX_FLOOR), __builtin_fmaxf(__builtin_fmaxf(axb, ayb), AMPS_MX_FLOOR) };
        mn[i] = (f2){ __builtin_fminf(axa, aya), __builtin_fminf(axb, ayb) };
        sw[i][0] = aya > axa; sw[i][1] = ayb > axb;
        r[i] = (f2){ __uint_as_float(AMPS_RCP_MAGIC - __float_as_uint(mx[i].x)), __uint_as_float(AMPS_RCP_MAGIC - __float_as_uint(mx[i].y)) };
    }
    const f2 one = { 1.0f, 1.0f };
#pragma unroll
    for (int it = 0; it < 3; it++) {
#pragma unroll
        for (int i = 0; i < N; i++) e[i] = __builtin_elementwise_fma(-mx[i], r[i], one);
    fm_row_join(e);
#pragma unroll
        for (int i = 0; i < N; i++) r[i] = __builtin_elementwise_fma(r[i], e[i], r[i]);
    fm_row_join(r);
    }
#pragma unroll
    for (int i = 0; i < N; i++) q[i] = mn[i] * r[i];
    fm_row_join(q);
#pragma unroll
    for (int i = 0; i < N; i++) z[i] = q[i] * q[i];
    fm_row_join(z);
#pragma unroll
    for (int i = 0; i < N; i++) p[i] = __builtin_elementwise_fma((f2){ AMPS_ATAN_C5, AMPS_ATAN_C5 }, z[i], (f2){ AMPS_ATAN_C4, AMPS_ATAN_C4 });
    fm_row_join(p);
#pragma unroll
    for (int i = 0; i < N; i++) p[i] = __builtin_elementwise_fma(p[i], z[i], (f2){ AMPS_ATAN_C3, AMPS_ATAN_C3 });
    fm_row_join(p);
#pragma unroll
    for (int i = 0; i < N; i++) p[i] = __builtin_elementwise_fma(p[i], z[i], (f2){ AMPS_ATAN_C2, AMPS_ATAN_C2 });
    fm_row_join(p);
#pragma unroll
    for (int i = 0; i < N; i++) p[i] = __builtin_elementwise_fma(p[i], z[i], (f2){ AMPS_ATAN_C1, AMPS_ATAN_C1 });
    fm_row_join(p);
#pragma unroll
    for (int i = 0; i < N; i++) p[i] = __builtin_elementwise_fma(p[i], z[i], (f2){ AMPS_ATAN_C0, AMPS_ATAN_C0 });
    fm_row_join(p);
#pragma unroll
    for (int i = 0; i < N; i++) a[i] = p[i] * q[i];
    fm_row_join(a);
#pragma unroll
    for (int i = 0; i < N; i++) {
        const f2 a_swapped = (f2){ AMPS_PI_2_F, AMPS_PI_2_F } - a[i];
        a[i].x = sw[i][0] ? a_swapped.x : a[i].x;
        a[i].y = sw[i][1] ? a_swapped.y : a[i].y;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        const f2 a_reflect = (f2){ AMPS_PI_F, AMPS_PI_F } - a[i];
        a[i].x = re[i].x < 0.0f ? a_reflect.x : a[i].x;
        a[i].y = re[i].y < 0.0f ? a_reflect.y : a[i].y;
        out[i] = (f2){ __builtin_copysignf(a[i].x, im[i].x), __builtin_copysignf(a[i].y, im[i].y) };
    }
}
// ta, tb = the two conj-products as (re, im) pairs
__device__ __forceinline__ f2 fm_phase_core(f2 ta, f2 tb)
{
    return fm_phase_planar((f2){ ta.x, tb.x }, (f2){ ta.y, tb.y });
}

// t = x * conj(p): (re, im) = (xr*pr + xi*pi, xi*pr - xr*pi), packed over (re, im)
__device__ __forceinline__ f2 conj_product(f2 x, f2 p)
{
    const f2 m = (f2){ x.y, x.x } * (f2){ p.y, -p.y };
    return __builtin_elementwise_fma(x, (f2){ p.x, p.x }, m);
}

// spec C: only the imaginary part of the two conj-products, fmaf(xi, pr, -(xr*pi)) -- the `im` of spec A
__device__ __forceinline__ f2 sine_pair(float4 s, float pr, float pi_)
{
    return (f2){ __builtin_fmaf(s.y, pr, -(s.x * pi_)), __builtin_fmaf(s.w, s.x, -(s.z * s.y)) };
}
__device__ __forceinline__ f2 fm_phase_pair(float4 s, float pr, float pi_)
{
    const f2 xa = { s.x, s.y }, xb = { s.z, s.w };
    return fm_phase_core(conj_product(xa, (f2){ pr, pi_ }), conj_product(xb, xa));
}

// the same discriminator for two INDEPENDENT streams (x0 after p0, x1 after p1): used behind the channelizer's
// FFT, where a lane owns four channels and the predecessor is the previous frame's value of the same bin
__device__ __forceinline__ f2 fm_phase_two(f2 x0, f2 p0, f2 x1, f2 p1)
{
    return fm_phase_core(conj_product(x0, p0), conj_product(x1, p1));
}

// Per-wave LDS demod buffers (two, used alternately): 16 floats of history (the tail of the previous
// tile, written by the previous tile's P1 into THIS buffer) then the 512 floats of the tile; every 8
// floats padded by one, so the boxcar reads of lane L hit bank 9*L + const (conflict free) and every
// LDS address is a lane-constant base plus an immediate offset.
constexpr int DHIST = 16;
__host__ __device__ constexpr int didx(int n) { return n + (n >> 3); }   // n = DHIST + tile-local sample
constexpr int DBUF = ((didx(DHIST + TILE - 1) + 1 + 7) / 8) * 8;        // 600 floats per buffer
constexpr int GW32 = 4 * TILE / 32;                                     // 64 dwords: 4-tile bit ring (+1 mirror)
// Spec B stages the tile's raw samples instead: 16 samples of history then the 512 of the tile, as float2, every 8
// padded by one (lane stride 18 banks: the 8-byte reads of a 32-lane group cover all 64 banks once)
constexpr int XHIST = 16;
__host__ __device__ constexpr int xidx(int n) { return n + (n >> 3); }   // n = XHIST + tile-local sample
constexpr int XBUF = ((xidx(XHIST + TILE - 1) + 1 + 7) / 8) * 8;        // 600 float2 per wave

// trigger symbol i as an xor mask for the xnor test (symbol 1 -> 0, symbol 0 -> ~0)
__device__ __forceinline__ uint32_t trig_xor(int i)
{
    uint32_t w = i < 32 ? (uint32_t)TRIG_LO : i < 64 ? (uint32_t)(TRIG_LO >> 32) : (uint32_t)TRIG_HI;
    return ((w >> (i & 31)) & 1u) - 1u;
}

// lane i <- lane i-1 of the same wave, lane 0 <- `lane0` (one DPP mov, no LDS traffic)
__device__ __forceinline__ float shift_in(float v, float lane0)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lane0), __float_as_int(v), 0x138 /* wave_shr:1 */,
                                                      0xf, 0xf, false));
}
__device__ __forceinline__ float lane63(float v)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Bit-sliced small SIGNED integers for slicer spec D: bit i of plane k = bit k of the two's complement number that belongs to sample
// i of a 32-sample word (oldest sample at bit 0, so "j samples earlier" is a left shift); plane W - 1 is the sign.  One chain of
// signed adds for K = w'' - sum w' instead of two unsigned ones (wraps past +pi and past -pi counted apart, then compared): 63
// word operations per 32 samples at 10 samples per symbol instead of 88.  Plain C: the same functions run on the host for
// tests/test_cpu_exact_slicer.py (amps_recc_debug_exact_slice).
template <int W> struct SBits { uint32_t p[W]; };
__host__ __device__ constexpr int sbits_width(int L) { int w = 1; while ((1 << (w - 1)) <= L) w++; return w; }   // planes that hold -L .. L
template <int W> __host__ __device__ __forceinline__ SBits<W> sb_shl(const SBits<W> &a, int j)
{
    SBits<W> r;
#pragma unroll
    for (int k = 0; k < W; k++) r.p[k] = a.p[k] << j;
    return r;
}
// a + b in WR planes, both sign-extended (the caller picks WR wide enough for the sum)
template <int WR, int WA, int WB> __host__ __device__ __forceinline__ SBits<WR> sb_add(const SBits<WA> &a, const SBits<WB> &b)
{
    SBits<WR> r;
    uint32_t c = 0u;
#pragma unroll
    for (int k = 0; k < WR; k++) {
        const uint32_t pa = a.p[k < WA ? k : WA - 1], pb = b.p[k < WB ? k : WB - 1];
        const uint32_t x = pa ^ pb;
        r.p[k] = x ^ c;
        c = (pa & pb) | (x & c);
    }
    return r;
}
// sum_{j < L} (t << j) for t in {-1, 0, +1} per sample: the L samples ending at each position, by doubling (1, 2, 4, 8, 16 samples)
// and the binary digits of L
template <int L> __host__ __device__ __forceinline__ SBits<sbits_width(L)> sb_window(const SBits<2> &t)
{
    static_assert(L >= 1 && L <= 16, "window");
    constexpr int WR = sbits_width(L);
    const SBits<2> s1 = t;
    SBits<3> s2{}; SBits<4> s4{}; SBits<5> s8{}; SBits<6> s16{};
    if constexpr (L >= 2) s2 = sb_add<3>(s1, sb_shl(s1, 1));
    if constexpr (L >= 4) s4 = sb_add<4>(s2, sb_shl(s2, 2));
    if constexpr (L >= 8) s8 = sb_add<5>(s4, sb_shl(s4, 4));
    if constexpr (L >= 16) s16 = sb_add<6>(s8, sb_shl(s8, 8));
    SBits<WR> acc{};
    int off = 0;
    if constexpr (L & 16) { acc = sb_add<WR>(acc, sb_shl(s16, off)); off += 16; }
    if constexpr (L & 8) { acc = sb_add<WR>(acc, sb_shl(s8, off)); off += 8; }
    if constexpr (L & 4) { acc = sb_add<WR>(acc, sb_shl(s4, off)); off += 4; }
    if constexpr (L & 2) { acc = sb_add<WR>(acc, sb_shl(s2, off)); off += 2; }
    if constexpr (L & 1) { acc = sb_add<WR>(acc, sb_shl(s1, off)); off += 1; }
    return acc;
}
// slicer spec D (include/amps_recc_numerics.h) on one 32-sample window, oldest sample at bit 0: SX, ST, SC = the signs of Im x,
// Im(x conj(x[n-1])), Im(x conj(x[n-SPS])).  Bits >= SPS of the result are exact (the wraps need one sample of history, their
// window SPS - 1 more, the partner SPS).
template <int SPS> __host__ __device__ __forceinline__ uint32_t exact_slice_word(uint32_t SX, uint32_t ST, uint32_t SC)
{
    const uint32_t sx1 = SX << 1, sxs = SX << SPS;
    const uint32_t wp = ~SX & sx1 & ST, wm = SX & ~sx1 & ~ST;          // w'  = +1 / -1
    const uint32_t up = ~SX & sxs & SC, um = SX & ~sxs & ~SC;          // w'' = +1 / -1
    const SBits<2> t = { { wp | wm, wp } };                            // -w' per sample: -1 where the step wrapped past +pi, +1 past -pi
    const SBits<2> u = { { up | um, um } };                            // +w''
    constexpr int WK = sbits_width(SPS + 1);
    const SBits<WK> K = sb_add<WK>(sb_window<SPS>(t), u);              // K = w'' - sum w'
    uint32_t any = 0u;
#pragma unroll
    for (int k = 0; k < WK; k++) any |= K.p[k];
    return ~K.p[WK - 1] & (any | ~SC);                                 // K > 0, or K == 0 and the partner product not negative
}
// the same for the filter bank's slicer (3 frames per symbol), whose sign words are shift registers with the NEWEST frame at bit 0
// and the previous 32 frames in a second word: delay j = funnel shift.  wp_out / wm_out: the wrap words the next call needs.
__host__ __device__ __forceinline__ uint32_t exact_funnel(uint32_t prev, uint32_t cur, int j) { return (cur >> j) | (prev << (32 - j)); }
__host__ __device__ __forceinline__ uint32_t exact_slice_word3(uint32_t SX, uint32_t ST, uint32_t SC, uint32_t sx_prev, uint32_t wp_prev, uint32_t wm_prev,
                                                                uint32_t &wp_out, uint32_t &wm_out)
{
    const uint32_t sx1 = exact_funnel(sx_prev, SX, 1), sx3 = exact_funnel(sx_prev, SX, 3);
    const uint32_t wp = ~SX & sx1 & ST, wm = SX & ~sx1 & ~ST;          // the phase step crossed the cut: w' = +1 / -1
    const uint32_t up = ~SX & sx3 & SC, um = SX & ~sx3 & ~SC;          // ... of the three-frame partner: w'' = +1 / -1
    const uint32_t wp1 = exact_funnel(wp_prev, wp, 1), wp2 = exact_funnel(wp_prev, wp, 2);
    const uint32_t wm1 = exact_funnel(wm_prev, wm, 1), wm2 = exact_funnel(wm_prev, wm, 2);
    const SBits<2> t0 = { { wp | wm, wp } }, t1 = { { wp1 | wm1, wp1 } }, t2 = { { wp2 | wm2, wp2 } }, u = { { up | um, um } };
    const SBits<4> K = sb_add<4>(sb_add<3>(t0, t1), sb_add<3>(t2, u));  // K = w'' - (w'[n] + w'[n-1] + w'[n-2]), -4 .. 4
    wp_out = wp; wm_out = wm;
    return ~K.p[3] & (K.p[0] | K.p[1] | K.p[2] | ~SC);                  // K > 0, or K == 0 and Im(y conj(y[n-3])) not negative
}
// ... and two frames per symbol (the D = 768 bank): K = w'' - (w'[n] + w'[n-1]), -3 .. 3
__host__ __device__ __forceinline__ uint32_t exact_slice_word2(uint32_t SX, uint32_t ST, uint32_t SC, uint32_t sx_prev, uint32_t wp_prev, uint32_t wm_prev,
                                                                uint32_t &wp_out, uint32_t &wm_out)
{
    const uint32_t sx1 = exact_funnel(sx_prev, SX, 1), sx2 = exact_funnel(sx_prev, SX, 2);
    const uint32_t wp = ~SX & sx1 & ST, wm = SX & ~sx1 & ~ST;
    const uint32_t up = ~SX & sx2 & SC, um = SX & ~sx2 & ~SC;
    const uint32_t wp1 = exact_funnel(wp_prev, wp, 1), wm1 = exact_funnel(wm_prev, wm, 1);
    const SBits<2> t0 = { { wp | wm, wp } }, t1 = { { wp1 | wm1, wp1 } }, u = { { up | um, um } };
    const SBits<3> K = sb_add<3>(sb_add<3>(t0, t1), u);
    wp_out = wp; wm_out = wm;
    return ~K.p[2] & (K.p[0] | K.p[1] | ~SC);                           // K > 0, or K == 0 and Im(y conj(y[n-2])) not negative
}

// BITS = true is the bit-domain form used behind the fused channelizer: the slicer bits of this launch are
// already in the HBM ring (written by chz_fused_kernel), so a tile is just 16 dwords read from it and only the
// correlator / emit stages (P3a, P3b) run.
// TOL = true replaces the exact trigger match by "at most a.tol of the 74 symbols differ" (a
// divergence from the reference's memmem, off by default): no prefilter is sound then, so every tile pays a
// bit-sliced population count over all 74 taps (~350 instructions per lane per tile instead of ~25).
// SL = AMPS_SLICER_PRODUCT: slicer spec B (sign of Im(x[n] conj(x[n-SPS]))) instead of discriminator + boxcar: the tile's
// raw samples are staged in the wave's LDS buffer and each lane slices 8 consecutive samples with one v_pk_mul, one
// v_sub and one v_alignbit each -- the kernel is then bound by its HBM reads alone.
template <int SPS, int DEPTH, bool BITS = false, bool TOL = false, int SL = AMPS_SLICER_ATAN_BOXCAR>
__global__ __launch_bounds__(256, DEPTH == 1 ? AMPS_FRONT_D1_BLOCKS : DEPTH == 2 ? AMPS_FRONT_D2_BLOCKS : AMPS_FRONT_D3_BLOCKS) void recc_front_kernel(FrontArgs a)
{
    constexpr bool EXACT = SL == AMPS_SLICER_EXACT;
    constexpr bool PROD = SL == AMPS_SLICER_PRODUCT || EXACT;          // specs B and D stage the tile's raw samples in LDS
    static_assert(!PROD || (!BITS && SPS < XHIST), "specs B / D run on IQ");
    front_housekeeping(a);
    static_assert(SPS >= 2 && SPS <= 16, "samples per symbol");
    constexpr int H = SPS - 1;                  // boxcar history
    constexpr int D = AMPS_DEDUP_SYMBOLS * SPS; // dedup / run window in samples (<= 32)
    static_assert(H <= DHIST, "history prefix too small");
    __shared__ float    s_d_all[4][BITS ? 8 : PROD ? 2 * XBUF : 2 * DBUF];   // demod buffers (spec B: one float2 sample buffer): not used in the bit domain (keeps its LDS at 2 KB)
    __shared__ uint32_t s_g_all[4][GW32 + 2];   // [GW32] mirrors [0] so a tap can always read dwords qd, qd+1
    __shared__ uint32_t s_m_all[4][GW32];
    __shared__ uint32_t s_x_all[EXACT ? 4 : 1][EXACT ? 3 * (GW32 + 2) : 1];   // spec D: the three sign streams as bit rings shaped like s_g

    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // Persistent, evenly split: the launch has exactly as many waves as the chip holds at once and
    // wave w owns the contiguous span [w*span, (w+1)*span) of the flattened (channel, tile) space, so
    // every wave does the same amount of work in ONE round (no tail round of a quantised grid).  A
    // span that crosses a channel boundary is processed as two (or more) segments.
    const uint32_t w_id = blockIdx.x * 4 + wv;
    const uint64_t Tc = a.tiles_per_channel;
    const uint64_t g_end_all = (uint64_t)a.n_channels * Tc;
    uint64_t g0 = (uint64_t)w_id * a.span;
    uint64_t g1 = g0 + a.span; if (g1 > g_end_all) g1 = g_end_all;
    if (!BITS && a.carry_out) {
        // round 4: the ~9 KB per channel the next push starts from are copied here, 1-4 samples per lane, instead of by a kernel of
        // their own behind this one (5 us + a launch gap per push); the two carry buffers alternate, so nothing read below is written
        const uint32_t nw_all = gridDim.x * 4u;
        const uint32_t total = a.n_channels * a.carry_n;
        const uint32_t per = (total + nw_all - 1) / nw_all;
        const uint32_t e1 = w_id * per + per < total ? w_id * per + per : total;
        for (uint32_t e = w_id * per + (uint32_t)lane; e < e1; e += 64) {
            const uint32_t cc = e / a.carry_n, k = e - cc * a.carry_n;
            const int64_t i = (int64_t)a.P - HALO + k;
            a.carry_out[(uint64_t)cc * CARRY_CAP + k] = i < (int64_t)a.r_prev ? a.carry[(uint64_t)cc * CARRY_CAP + HALO + i]
                                                                              : a.block[(uint64_t)cc * a.ld + (i - a.r_prev)];
        }
    }
    const int64_t words_end = (int64_t)a.P / 64;          // rel word index limit of this launch
    const int r_prev = (int)a.r_prev, avail = (int)a.avail;
    float *s_d = s_d_all[wv];
    uint32_t *s_g = s_g_all[wv];
    uint32_t *s_m = s_m_all[wv];
    uint32_t *s_x = s_x_all[EXACT ? wv : 0];
    // lane-constant pieces of the LDS addressing (everything else is an immediate offset)
    const int dw_off = 2 * lane + (lane >> 2);          // P1 writes: didx(DHIST + 128q + 2*lane + e)
    const int dr_off = 9 * lane;                        // P2 reads:  didx(DHIST - H + 8*lane + m)
    const int wq = lane >> 2;                           // P3: dword of the tile this quad owns (0..15)
    const int part = lane & 3;                          // P3: 4 lanes share a dword
    // prefilter taps: the symbols part+4u (u = 0..3) before the last one; relative to the slot base the
    // bit offset is lane-constant, so the shift and the xor mask never change and only the dword moves
    uint32_t pre_xor[4];
    int pre_q0[4], pre_sh[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int off = 32 * wq - SPS * (part + 4 * u);          // may be negative: previous tiles
        pre_xor[u] = trig_xor(TRIG - 1 - part - 4 * u);
        pre_q0[u] = off >> 5;                                     // arithmetic shift
        pre_sh[u] = off & 31;
    }

  while (g0 < g1) {                                    // one segment = a run of tiles inside one channel
    const int c = __builtin_amdgcn_readfirstlane((int)(g0 / Tc));   // wave-uniform, but a 64-bit division runs on the VALU: back to an SGPR, and the segment's base pointers with it (they were the kernel's three spilled register pairs)
    const uint32_t t_lo = (uint32_t)(g0 - (uint64_t)c * Tc);
    uint32_t t_hi = t_lo + (uint32_t)(g1 - g0); if (t_hi > Tc) t_hi = (uint32_t)Tc;
    const uint32_t chunk = w_id - (uint32_t)(((uint64_t)c * Tc) / a.span);   // k-th segment of this channel
    const int64_t chunk_start = (int64_t)t_lo * TILE;    // rel
    const int K = (int)(t_hi - t_lo);                    // real tiles in this segment
    g0 += (uint64_t)K;
    const float2 *blk = a.block + (uint64_t)c * a.ld;
    const float2 *car = a.carry + (uint64_t)c * CARRY_CAP;
    int lane2 = 2 * lane;
    asm volatile("" : "+v"(lane2));                      // opaque per segment: "block base + lane" must not be hoisted out of the segment loop
    const float2 *blk_lane = blk + lane2;                // (as a 64-bit invariant of the whole kernel it was spilled at 128 registers)

    auto fetch = [&](int64_t i) -> float2 {   // generic path: carry then block; zero outside the data
        if (i >= avail || i < -(int64_t)HALO) return make_float2(0.f, 0.f);
        const float2 *p = (i < r_prev) ? (car + (HALO + i)) : (blk + (i - r_prev));
        return *p;
    };
    // one 512-sample tile: r[q] = samples (s0 + 128q + 2*lane, +1)
    auto load_tile = [&](float4 (&r)[4], int64_t s0) {
        if (s0 >= r_prev && s0 + TILE <= avail) {          // wave-uniform: entirely inside the new block
            const f4a8 *p = (const f4a8 *)(blk_lane + (s0 - r_prev));
#pragma unroll
#if AMPS_FRONT_NT
            for (int q = 0; q < 4; q++) { f4a8 v = __builtin_nontemporal_load(p + 64 * q); r[q] = make_float4(v.x, v.y, v.z, v.w); }
#else
            for (int q = 0; q < 4; q++) { f4a8 v = p[64 * q]; r[q] = make_float4(v.x, v.y, v.z, v.w); }
#endif
        } else {
            // (the lane's offset is made opaque here: otherwise its 64-bit sign extensions are hoisted out of the segment loop as
            // invariants of this rare path and, at 128 registers, spilled -- the kernel's only scratch traffic in rounds 2 and 3)
            int l2 = 2 * lane;
            asm volatile("" : "+v"(l2));
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float2 u = fetch(s0 + 128 * q + l2), w = fetch(s0 + 128 * q + l2 + 1);
                r[q] = make_float4(u.x, u.y, w.x, w.y);
            }
        }
    };

    s_g[lane] = ~0u;
    if (lane < 2) s_g[GW32 + lane] = ~0u;
    s_m[lane] = 0u;
    if constexpr (EXACT) {
#pragma unroll
        for (int u = 0; u < 3; u++) { s_x[u * (GW32 + 2) + lane] = 0u; if (lane < 2) s_x[u * (GW32 + 2) + GW32 + lane] = 0u; }
    }
    if constexpr (!BITS) for (int i = lane; i < (PROD ? 2 * XBUF : 2 * DBUF); i += 64) s_d[i] = 0.f;

    float4 cur[4], nxt[DEPTH][4];                // tile k in use, tiles k+1..k+DEPTH in flight (DEPTH x 4 KiB per wave)
    float last_x = 0.f, last_y = 0.f;            // last sample of the previous tile (wave-uniform)
    uint32_t ndet = 0;                           // hits appended by this wave (wave-uniform)
    bool hit_prev = false;                       // the previous tile had a trigger hit (wave-uniform)
    if constexpr (!BITS) {
        load_tile(cur, chunk_start - HALO);
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; d++) load_tile(nxt[d], chunk_start - HALO + (d + 1) * TILE);   // K >= 1: all exist up to d = 1
    }
    const uint32_t *gring32 = (const uint32_t *)(a.gring + (uint64_t)c * a.ring_words);
    auto load_bits = [&](int kk) -> uint32_t {        // BITS mode: dword `lane` of tile kk of this segment (lanes 0..15)
        if (lane >= TILE / 32) return 0u;
        const int64_t n32 = (int64_t)a.n_done + chunk_start + (int64_t)kk * TILE + 32 * lane;
        return n32 < 0 ? ~0u : gring32[(uint64_t)(n32 >> 5) & (2ull * a.ring_words - 1)];
    };
    uint32_t bw0 = 0u, bw1 = 0u, bw2 = 0u;           // tiles k, k+1, k+2 of the loop below (K >= 1)
    if constexpr (BITS) { bw0 = load_bits(-2); bw1 = load_bits(-1); bw2 = load_bits(0); }

    // k = -2, -1 are the halo tiles [chunk_start-1024, chunk_start): recomputed, never stored or emitted
    for (int k = -2; k < K; k++) {
        const int64_t t0 = chunk_start + (int64_t)k * TILE;   // rel start of this tile
        const int slot = (k + 2) & 3;                          // bit-ring slot of this tile
        float *const dcur = s_d + ((k + 2) & 1) * DBUF;        // demod buffer of this tile
        float *const dnxt = s_d + ((k + 3) & 1) * DBUF;        // ... of the next tile (gets our tail as history)
        if constexpr (BITS) {
            // the tile's 512 slicer bits come from the HBM ring (fetched three tiles ahead: a bare load per tile is ~1 us of
            // exposed latency); samples before the stream are ones (x = 0 -> g = 1)
            const uint32_t w = bw0;
            bw0 = bw1; bw1 = bw2;
            bw2 = (k + 3 < K) ? load_bits(k + 3) : 0u;
            if (lane < TILE / 32) {
                s_g[slot * (TILE / 32) + lane] = w;
                if (slot == 0 && lane < 2) s_g[GW32 + lane] = w;      // mirror of dwords 0,1
            }
        } else if constexpr (PROD) {
        // ---- spec B: stage the tile's samples in LDS, then every lane slices samples 8*lane .. 8*lane+7 ----
        if (k + DEPTH < K) load_tile(nxt[DEPTH - 1], t0 + DEPTH * TILE);
        f2 *const xs = (f2 *)s_d;
        {
#ifdef AMPS_FRONT_LDS_LINEAR_STORES_EXPERIMENT
            // TIMING EXPERIMENT ONLY (profiles/r05/front_lds_conflicts.txt): the stores go to an unpadded, perfectly linear layout -- no
            // bank conflict at all, and WRONG results (the reads keep the padded layout) -- to see what the real layout's 17 % of
            // conflict cycles cost the kernel
            f2 *const xw = xs + 2 * lane;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                xw[XHIST + 128 * q] = (f2){ cur[q].x, cur[q].y };
                xw[XHIST + 128 * q + 1] = (f2){ cur[q].z, cur[q].w };
            }
#else
            f2 *const xw = xs + 2 * lane + (lane >> 2);              // xidx(XHIST + 128 q + 2 lane + e) = const + this
#pragma unroll
            for (int q = 0; q < 4; q++) {
                xw[xidx(XHIST + 128 * q)] = (f2){ cur[q].x, cur[q].y };
                xw[xidx(XHIST + 128 * q) + 1] = (f2){ cur[q].z, cur[q].w };
            }
#endif
        }
        __builtin_amdgcn_wave_barrier();
        {
            const f2 *const xr = xs + 9 * lane;
            f2 v[SPS + 8];
#pragma unroll
            for (int m = 0; m < SPS + 8; m++) v[m] = xr[xidx(XHIST - SPS + m)];
            const bool dbg = a.dbg_d && (uint32_t)c == a.dbg_channel && k >= 0;
            unsigned byte;
            if constexpr (EXACT) {
                // spec D: three sign bits per sample (Im x, Im(x conj(x[n-1])), Im(x conj(x[n-SPS]))) into three bit rings shaped
                // like s_g; the slicer bits of the lane's eight samples then come out of a 32-sample window of those rings
                uint32_t ax = 0u, at = 0u, ac = 0u;                  // sign bits, newest at bit 0
                // (round 6: the debug taps are tested ONCE per tile, not once per sample -- eight wave-uniform branches sat between the
                // eight samples' products and fenced the scheduler; the tapped tile runs its own copy of the loop)
                auto samples = [&](auto tapc) {
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const f2 x = v[SPS + q], p1 = v[SPS + q - 1], ps = v[q];
                        const float it = __builtin_fmaf(x.y, p1.x, -(x.x * p1.y));
                        const float ic = __builtin_fmaf(x.y, ps.x, -(x.x * ps.y));
                        ax = __builtin_amdgcn_alignbit(ax, __float_as_uint(x.y), 31);
                        at = __builtin_amdgcn_alignbit(at, __float_as_uint(it), 31);
                        ac = __builtin_amdgcn_alignbit(ac, __float_as_uint(ic), 31);
                        if constexpr (decltype(tapc)::value) {
                            int l8 = 8 * lane; asm volatile("" : "+v"(l8));   /* opaque: the debug tap's lane address is not a kernel-wide invariant worth a spilled register pair */
                            int64_t rel = t0 + l8 + q;
                            if (rel < (int64_t)a.P) { a.dbg_d[rel] = it; a.dbg_S[rel] = ic; }
                        }
                    }
                };
                if (__builtin_expect(dbg, 0)) samples(std::true_type{}); else samples(std::false_type{});
                uint8_t *const bx = (uint8_t *)s_x, *const bt = (uint8_t *)(s_x + (GW32 + 2)), *const bc = (uint8_t *)(s_x + 2 * (GW32 + 2));
                const int bo = slot * (TILE / 8) + lane;
                const uint8_t vx = (uint8_t)(__builtin_bitreverse32(ax) >> 24), vt = (uint8_t)(__builtin_bitreverse32(at) >> 24),
                              vc = (uint8_t)(__builtin_bitreverse32(ac) >> 24);      // sample 8*lane+q at bit q
                bx[bo] = vx; bt[bo] = vt; bc[bo] = vc;
                if (slot == 0 && lane < 8) { bx[GW32 * 4 + lane] = vx; bt[GW32 * 4 + lane] = vt; bc[GW32 * 4 + lane] = vc; }   // mirror of dwords 0,1
                __builtin_amdgcn_wave_barrier();                     // sign bytes visible; everybody has read the history prefix
                // the 32 samples that end with this lane's eight: bytes bo-3 .. bo of each ring
                const int b0 = (bo - 3) & (4 * GW32 - 1), qd = b0 >> 2;
                uint32_t W[3];
#pragma unroll
                for (int u = 0; u < 3; u++)
                    W[u] = __builtin_amdgcn_alignbyte(s_x[u * (GW32 + 2) + qd + 1], s_x[u * (GW32 + 2) + qd], (uint32_t)(b0 & 3));
                byte = exact_slice_word<SPS>(W[0], W[1], W[2]) >> 24;
            } else {
            uint32_t acc = 0u;                                       // sign bits, newest at bit 0
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const f2 m = v[SPS + q] * (f2){ v[q].y, v[q].x };    // (xr * pi, xi * pr) = (b, a)
                const float sd = m.y - m.x;
                acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(sd), 31);
                if (dbg) {
                    int l8 = 8 * lane; asm volatile("" : "+v"(l8));   /* opaque: the debug tap's lane address is not a kernel-wide invariant worth a spilled register pair */
                        int64_t rel = t0 + l8 + q;
                    if (rel < (int64_t)a.P) { a.dbg_d[rel] = 0.f; a.dbg_S[rel] = rel < (int64_t)a.force_ones ? 0.f : sd; }
                }
            }
            byte = (~__builtin_bitreverse32(acc)) >> 24;             // g = !signbit, sample 8*lane+q at bit q
            }
            if (t0 == 0 && a.force_ones) {                           // wave-uniform: the first tile of a stream
                const int nf = (int)a.force_ones - 8 * lane;         // samples of this lane that have no partner yet
                if (nf > 0) byte |= nf >= 8 ? 0xffu : (1u << nf) - 1u;
            }
            if constexpr (!EXACT) __builtin_amdgcn_wave_barrier();   // everybody has read the history prefix
            if (lane >= 56) {                                        // samples 496..511 become the next tile's history
                xs[xidx(2 * (lane - 56))] = (f2){ cur[3].x, cur[3].y };
                xs[xidx(2 * (lane - 56)) + 1] = (f2){ cur[3].z, cur[3].w };
            }
            ((uint8_t *)s_g)[slot * (TILE / 8) + lane] = (uint8_t)byte;
            if (slot == 0 && lane < 8) ((uint8_t *)s_g)[GW32 * 4 + lane] = (uint8_t)byte;   // mirror of dwords 0,1
        }
        } else {
        // ---- P1: prefetch the next tile, demodulate this one into LDS ----
        if (k + DEPTH < K) load_tile(nxt[DEPTH - 1], t0 + DEPTH * TILE);
        float *const dw = dcur + dw_off;
        f2 dd4[4];
        if constexpr (SL != AMPS_SLICER_SINE) {
            // the four pairs of a tile are independent: their arctangents run in lock step (fm_phase_planar_n), not one serial
            // Newton / Horner chain after the other
            f2 re4[4], im4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float ex = q == 0 ? last_x : lane63(cur[q - 1].z);
                const float ey = q == 0 ? last_y : lane63(cur[q - 1].w);
                const float pr = shift_in(cur[q].z, ex), pi_ = shift_in(cur[q].w, ey);
                const f2 xa = { cur[q].x, cur[q].y }, xb = { cur[q].z, cur[q].w };
                const f2 ta = conj_product(xa, (f2){ pr, pi_ }), tb = conj_product(xb, xa);
                re4[q] = (f2){ ta.x, tb.x }; im4[q] = (f2){ ta.y, tb.y };
            }
            fm_phase_planar_n<4>(re4, im4, dd4);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f2 dd;
            if constexpr (SL == AMPS_SLICER_SINE) {
                const float ex = q == 0 ? last_x : lane63(cur[q - 1].z);
                const float ey = q == 0 ? last_y : lane63(cur[q - 1].w);
                const float pr = shift_in(cur[q].z, ex), pi_ = shift_in(cur[q].w, ey);
                dd = sine_pair(cur[q], pr, pi_);
            } else dd = dd4[q];
            const float d0 = dd.x, d1 = dd.y;
            dw[didx(DHIST + 128 * q)] = d0;
            dw[didx(DHIST + 128 * q) + 1] = d1;
            if (q == 3 && lane >= 56) {          // samples 496..511: also the history prefix of the next tile
                dnxt[didx(2 * (lane - 56))] = d0;
                dnxt[didx(2 * (lane - 56)) + 1] = d1;
            }
        }
        last_x = lane63(cur[3].z);
        last_y = lane63(cur[3].w);
        __builtin_amdgcn_wave_barrier();
        // ---- P2: boxcar over one symbol (aligned pair sums), slice, pack 8 bits per lane ----
        {
            const float *const dr = dcur + dr_off;
            float v[H + 8];
#pragma unroll
            for (int m = 0; m < H + 8; m++) v[m] = dr[didx(DHIST - H + m)];
            unsigned byte = 0;
            const bool dbg = a.dbg_d && (uint32_t)c == a.dbg_channel && k >= 0;   // diagnostic taps (tests only)
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
                if constexpr (SL == AMPS_SLICER_SINE) byte |= (~__float_as_uint(s) >> 31) << q;   // g = !signbit(S')
                else byte |= (s >= 0.0f ? 1u : 0u) << q;
                if (dbg) {
                    int l8 = 8 * lane; asm volatile("" : "+v"(l8));   /* opaque: the debug tap's lane address is not a kernel-wide invariant worth a spilled register pair */
                        int64_t rel = t0 + l8 + q;
                    if (rel < (int64_t)a.P) { a.dbg_d[rel] = v[q + H]; a.dbg_S[rel] = s; }
                }
            }
            ((uint8_t *)s_g)[slot * (TILE / 8) + lane] = (uint8_t)byte;
            if (slot == 0 && lane < 8) ((uint8_t *)s_g)[GW32 * 4 + lane] = (uint8_t)byte;   // mirror of dwords 0,1
        }
        }   // !BITS
        __builtin_amdgcn_wave_barrier();
        // ---- P3a: bit-parallel exact match of the 74-symbol trigger; publish slicer words ----
        bool hit = false;
        {
            auto and_quad = [&](uint32_t x) -> uint32_t {
                x &= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                x &= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
                return x;
            };
            uint32_t acc = ~0u;
            if constexpr (TOL) {
                // each lane of the quad counts the mismatches of its 18-19 taps in five bit planes (32 positions at
                // once), the quad adds its four counters, and the 7-bit sums are compared with a.tol plane by plane
                const int bitbase = slot * TILE + 32 * wq;
                // the lane's taps are i = part + 4u, u = 0..18 (the 19th exists for part < 2 only); their mismatch words are
                // summed with carry-save adders (Harley-Seal, as in recc_bits.hip.h): ~2 instructions per tap for the count
                // instead of 10 for a five-plane ripple counter
                auto mism = [&](int u) -> uint32_t {
                    const int i = part + 4 * u;
                    const int B = (bitbase - SPS * (TRIG - 1 - i)) & (4 * TILE - 1);
                    const int qd = B >> 5;
                    const uint32_t x = ~(__builtin_amdgcn_alignbit(s_g[qd + 1], s_g[qd], B & 31) ^ trig_xor(i));   // 1 = differs
                    return i < TRIG ? x : 0u;
                };
                auto csa = [](uint32_t &h, uint32_t &l, uint32_t x, uint32_t y, uint32_t z) {
                    const uint32_t w = x ^ y;
                    h = (x & y) | (w & z);
                    l = w ^ z;
                };
                uint32_t ones = 0u, twos = 0u, fours = 0u, e8 = 0u, e16 = 0u;
                auto add8 = [&](uint32_t e) { const uint32_t cy = e8 & e; e8 ^= e; e16 ^= cy; };   // at most 19: no carry out of 16
#pragma unroll
                for (int blk = 0; blk < 2; blk++) {
                    const int u0 = 8 * blk;
                    uint32_t twosA, twosB, foursA, foursB, eights;
                    csa(twosA, ones, ones, mism(u0), mism(u0 + 1));
                    csa(twosB, ones, ones, mism(u0 + 2), mism(u0 + 3));
                    csa(foursA, twos, twos, twosA, twosB);
                    csa(twosA, ones, ones, mism(u0 + 4), mism(u0 + 5));
                    csa(twosB, ones, ones, mism(u0 + 6), mism(u0 + 7));
                    csa(foursB, twos, twos, twosA, twosB);
                    csa(eights, fours, fours, foursA, foursB);
                    add8(eights);
                }
                {
                    uint32_t t2a, c4;
                    csa(t2a, ones, ones, mism(16), mism(17));
                    const uint32_t m18 = mism(18);
                    const uint32_t t2b = ones & m18; ones ^= m18;
                    csa(c4, twos, twos, t2a, t2b);
                    const uint32_t c8 = fours & c4; fours ^= c4;
                    add8(c8);
                }
                uint32_t cnt[7] = { ones, twos, fours, e8, e16, 0u, 0u };
#pragma unroll
                for (int step = 0; step < 2; step++) {
                    uint32_t cy = 0u;
#pragma unroll
                    for (int pl = 0; pl < 5 + step; pl++) {
                        const uint32_t o = step == 0 ? (uint32_t)__builtin_amdgcn_mov_dpp((int)cnt[pl], 0xB1, 0xF, 0xF, true)    // lane ^ 1
                                                     : (uint32_t)__builtin_amdgcn_mov_dpp((int)cnt[pl], 0x4E, 0xF, 0xF, true);   // lane ^ 2
                        const uint32_t x = cnt[pl] ^ o;
                        const uint32_t sum = x ^ cy;
                        cy = (cnt[pl] & o) | (x & cy);
                        cnt[pl] = sum;
                    }
                    cnt[5 + step] = cy;
                }
                uint32_t gt = 0u, eq = ~0u;
#pragma unroll
                for (int pl = 6; pl >= 0; pl--) {
                    const uint32_t kb = 0u - ((a.tol >> pl) & 1u);
                    gt |= eq & cnt[pl] & ~kb;
                    eq &= ~(cnt[pl] ^ kb);
                }
                acc = ~gt;
                hit = __ballot(acc != 0) != 0;
            } else {
            // prefilter: the last 16 symbols (all inside the word-sync part), 4 taps per lane
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int qd = (pre_q0[u] + slot * (TILE / 32)) & (GW32 - 1);
                const uint32_t lo = s_g[qd], hi = s_g[qd + 1];          // [GW32] mirrors [0]
                acc &= __builtin_amdgcn_alignbit(hi, lo, pre_sh[u]) ^ pre_xor[u];
            }
            acc = and_quad(acc);
            if (__ballot(acc != 0)) {        // rare (2^-16 per phase on noise): evaluate all 74 taps
                const int bitbase = slot * TILE + 32 * wq;
                acc = ~0u;
                for (int i = part; i < TRIG; i += 4) {
                    const int B = (bitbase - SPS * (TRIG - 1 - i)) & (4 * TILE - 1);
                    const int qd = B >> 5;
                    acc &= __builtin_amdgcn_alignbit(s_g[qd + 1], s_g[qd], B & 31) ^ trig_xor(i);
                }
                acc = and_quad(acc);
                hit = __ballot(acc != 0) != 0;
            }
            }   // !TOL
            if (part == 0) {
                s_m[slot * (TILE / 32) + wq] = acc;
                const int64_t relw = t0 / 64 + (wq >> 1);      // rel 64-bit word index
                if (!BITS && k >= 0 && relw < words_end) {
                    uint64_t absw = a.n_done / 64 + (uint64_t)relw;
                    uint32_t *g32 = (uint32_t *)(a.gring + (uint64_t)c * a.ring_words + (absw & a.ring_mask));
                    g32[wq & 1] = s_g[slot * (TILE / 32) + wq];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- P3b: emit run starts located in [previous tile word 7, this tile words 0..6] ----
        if (k >= 0 && (hit || hit_prev)) {               // wave-uniform and rare
            const uint64_t *s_m64 = (const uint64_t *)s_m;
            constexpr int GW = GW32 / 2;
            uint64_t starts = 0, mcur = 0, mnext = 0;
            int64_t relw = 0;
            if (lane < 8) {
                const int ring_w = (slot * (TILE / 64) + lane - 1) & (GW - 1);   // word examined
                relw = t0 / 64 + lane - 1;
                const uint64_t mprev = s_m64[(ring_w - 1) & (GW - 1)];
                mcur = s_m64[ring_w];
                mnext = s_m64[(ring_w + 1) & (GW - 1)];
                uint64_t smear = 0;
#pragma unroll
                for (int s = 1; s <= D; s++) smear |= (mcur << s) | (mprev >> (64 - s));
                starts = mcur & ~smear;
                // the examined word and its look-ahead word must both be processed data of this stream
                const int64_t absw = (int64_t)(a.n_done / 64) + relw;
                if (absw < 0 || relw + 1 >= words_end) starts = 0;
            }
            uint64_t who = __ballot(starts != 0);
            while (who) {                                 // ordered append, lane by lane
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
        hit_prev = hit;
        if constexpr (!BITS) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                cur[q] = nxt[0][q];
#pragma unroll
                for (int d = 0; d + 1 < DEPTH; d++) nxt[d][q] = nxt[d + 1][q];
            }
        }
    }
    if (lane == 0) a.detcount[(uint64_t)c * a.max_chunks + chunk] = ndet < a.det_cap ? ndet : a.det_cap;
  }   // next segment of this wave's span
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
