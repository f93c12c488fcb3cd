// recc_channelizer.hip.h -- polyphase channelizer front end for gfx950: one wideband fc32 stream
// (fs = M * 30 kHz) -> C active 30 kHz channels at fs / D, with the RECC slicer fused behind the FFT.
//
// It stands where the reference wires one freq_xlating_fir_filter_ccc (299 complex taps, decim 2) per
// channel in front of the RECC chain (grc/recctest.grc:889-937, taps :115-155).  Replicating that FIR per
// channel is ~150 flop per input byte; a weighted-overlap-add filter bank does all M
// channels at once for ~17 flop/B:
//     frame m:  n0 = (m+1) D - L,  L = P*M taps
//               u[r] = sum_{i : (n0+i) mod M = r} h[i] x[n0+i]          (fold, "polyphase")
//               Y_k[m] = FFT_M(u)[k] = sum_i h[i] x[n0+i] e^{-j 2 pi k (n0+i)/M}
// i.e. channel k (centre k*fs/M) mixed to DC with an absolute phase reference, low-pass filtered by the
// prototype h and decimated by D = M/2 (2x oversampled: 60 ksps = 3 samples per Manchester symbol at M = 1024).
//
// Mapping to the hardware: ONE 768-thread workgroup owns a CU and its twelve waves split three ROLES, four waves each, so
// that every SIMD holds one wave of each role -- a VALU-dense one, and two that alternate LDS round trips with VALU work in
// different phases (chz12_kernel below):
//   fold  : polyphase delay lines in a register ring, the fold as v_pk_fma chains, radix-4 pass 1 on its own registers
//   pass 2: radix 16, one frame per wave, in place in LDS
//   pass 3: radix 16 (FFT-1024 = 4 x 16 x 16), then the slicer: a lane owns the same four channels for ever
// Half-batches of four frames travel through a four-slot LDS ring (16 frame buffers, 136 KB), one workgroup barrier per four
// frames.  Only slicer bits (1/64 of the input) reach HBM.  No MFMA: the contraction per channel is 8 taps deep and differs per
// branch, and the FFT is a butterfly network (a DFT-16 as a matrix product costs 12x its flops at the f32 MFMA rate = the VALU rate).
#pragma once
#include <hip/hip_runtime.h>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "amps_recc.h"
#include "amps_recc_numerics.h"

namespace amps {

// Guard of a device staging buffer for host-resident input.  Copies from pageable host memory are neither ordered after
// earlier kernels of a non-blocking stream nor guaranteed to have read their source when an Async call returns, so
// back-to-back pushes without a drain in between corrupted samples (found by scripts/fuzz_parity.py: intermittent wrong
// slicer bits).  Host pushes therefore: wait() until the previous push's kernels have released the staging buffer,
// copy synchronously, enqueue the kernels, arm().
struct StageFence {
    hipEvent_t ev = nullptr;
    bool armed = false;
    int wait()
    {
        if (armed) { if (hipEventSynchronize(ev) != hipSuccess) return -EIO; armed = false; }
        return 0;
    }
    int arm(hipStream_t s)
    {
        if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return -ENOMEM;
        if (hipEventRecord(ev, s) != hipSuccess) return -EIO;
        armed = true;
        return 0;
    }
    void destroy() { if (ev) (void)hipEventDestroy(ev); ev = nullptr; armed = false; }
};

constexpr int CHZ_M = 1024;          // branches = FFT size
constexpr int CHZ_D = 512;           // input samples per frame of the default form (2x oversampled: 60 ksps per channel, 3 samples per symbol)
constexpr int CHZ_D768 = 768;        // ... of the 4/3 x oversampled form (40 ksps per channel, 2 samples per symbol; round 6, section "D = 768" below)

struct ChzArgs {
    const float2 *block;     // new wideband samples of this push
    const float2 *carry;     // samples [f_done*D + D - L, f_done*D + leftover) of the stream so far
    const float *taps;       // [L] prototype
    float2 *out;             // [C][ld] channel-major output of this push
    uint64_t ld;
    uint32_t carry_len;      // L - D + leftover
    uint32_t nsamp;          // new samples
    uint32_t nframes;        // frames produced by this launch
    uint32_t frames_per_wg;  // multiple of 64
    uint32_t first_bin;      // FFT bin of channel 0 (host bookkeeping; the kernel goes by bin2row)
    uint32_t n_channels;     // rows of `out` / `gring` this handle owns
    const uint16_t *bin2row; // [M]: row of FFT bin k in `out` / `gring`, >= n_channels for a bin this handle does not decode
    uint32_t grp_w, grp_r;   // channel groups (cfg.wideband_groups): this handle's bins are those with (k mod 64) in [grp_r * grp_w,
                             // (grp_r + 1) * grp_w); grp_w = 64, grp_r = 0: every bin
    uint32_t odd_start;      // parity of (absolute frame index of frame 0 of this launch)
    uint32_t hist;           // samples of history the carry holds before v = 0  (L - D + CHZ_PRE * D)
    // fused form: slicer bits go straight to the RECC bit ring
    uint64_t *gring;         // [C][ring_words]
    uint32_t ring_words;
    uint64_t n_done;         // absolute channel-stream sample index of frame 0 (multiple of 64)
    uint32_t stream_start;   // frame 0 of this launch is the first frame of the stream (spec B: its first 3 bits are ones)
    // the carry of the NEXT launch, written by this one (every workgroup copies a slice; the two carry buffers alternate)
    float2 *carry_out;       // [carry_out_len]: virtual samples [consumed - hist, consumed - hist + carry_out_len)
    uint32_t consumed, carry_out_len;
    unsigned long long *tl;  // CHZ_TIMELINE builds: s_memtime stamps of workgroup 0 ([wave][step][8]), else unused
};
constexpr int CHZ_PRE = 4;   // frames of history the carry keeps beyond the filter's own L - D samples: what the exact half of a workgroup's pre-roll reaches back to

typedef float cf2 __attribute__((ext_vector_type(2)));
// Buffer resource (V#) of the fast loader (round 6).  The fold role's prefetch used `global_load_dwordx2 v, v_off, s[base:base+1]` with
// a 64-bit scalar base per frame: clamp the frame index to the block, multiply, subtract, shift, add with carry -- nine scalar
// instructions per frame and two more per chunk, ~65 per half-step in the wave whose instruction count IS the kernel's critical path
// (a wave issues one instruction per slot, scalar or vector).  A raw buffer load takes a 32-bit scalar byte offset (+ a 12-bit
// immediate) against ONE descriptor per workgroup, and the hardware's range check (num_records) replaces the clamp: a prefetch that
// runs past the end of the block returns zeros nobody folds.  One s_add per pair of 2 KB chunks.
typedef int chz_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ chz_rsrc_t chz_make_rsrc(const void *base, uint64_t bytes)
{
    const uint64_t a = (uint64_t)base;
#ifdef CHZ_TIMING_NO_LOADS
    // TIMING EXPERIMENT ONLY (profiles/r06/chz_no_loads.txt): every fast load falls outside the descriptor and returns zeros without touching
    // memory -- WRONG results; what the kernel takes when its input stream costs nothing but the load instructions themselves
    const uint32_t n = 64u; (void)bytes;
#else
    const uint32_t n = bytes > 0xffffffffull ? 0xffffffffu : (uint32_t)bytes;
#endif
    chz_rsrc_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffffu));          // stride 0: a raw buffer, offsets and num_records in bytes
    r.z = __builtin_amdgcn_readfirstlane((int)n);
    r.w = 0x00020000;                                                                   // gfx9 family: DATA_FORMAT = 32 (untyped loads ignore it), type = buffer
    return r;
}

// Packed fp32 with operand modifiers.  hipcc materialises every swap / negate / broadcast of a packed operand with v_mov /
// v_xor and keeps broadcast constants as duplicated register PAIRS (the 32 fold coefficients cost 64 VGPRs, the six pass
// twiddles 24); the VOP3P modifiers do all of that for free, so the few shapes the pipeline needs are written as single
// instructions.  (The compiler treats an inline-asm producer conservatively for the gfx950 forwarding hazard and inserts
// the wait state itself.)  Every form computes exactly the IEEE operations of the plain expression next to it.
//
// (a.x b.x - a.y b.y, a.y b.x + a.x b.y):  m = (a.y * -b.y, a.x * b.y);  r = a * (b.x, b.x) + m
__device__ __forceinline__ cf2 cmul(cf2 a, cf2 b)
{
    cf2 m, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(m) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(m));
    return r;
}
// the same with a wave-uniform twiddle held in an SGPR pair (the W16 constants of the radix-16 pass)
__device__ __forceinline__ cf2 cmul_s(cf2 a, cf2 b)
{
    cf2 m, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(m) : "v"(a), "s"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "s"(b), "v"(m));
    return r;
}
__device__ __forceinline__ cf2 mul_mi(cf2 a) { return (cf2){ a.y, -a.x }; }   // a * (-i)
// v + (-i) d = (v.x + d.y, v.y - d.x)   and   v - (-i) d = (v.x - d.y, v.y + d.x)
__device__ __forceinline__ cf2 add_mi(cf2 v, cf2 d)
{
    cf2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(v), "v"(d));
    return r;
}
__device__ __forceinline__ cf2 sub_mi(cf2 v, cf2 d)
{
    cf2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(v), "v"(d));
    return r;
}
// a * (c.x, c.x) + s  and  a * (c.y, c.y) + s : one coefficient PAIR serves two taps
__device__ __forceinline__ cf2 fma_lo(cf2 a, cf2 c, cf2 s)
{
    cf2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(c), "v"(s));
    return r;
}
__device__ __forceinline__ cf2 fma_hi(cf2 a, cf2 c, cf2 s)
{
    cf2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(c), "v"(s));
    return r;
}


// ---- frame geometry ----
// A half-batch = four frames (what the fold role produces per time step); a frame buffer holds one 1024-point frame.
constexpr int CHZ_BATCH = 4;
constexpr int CHZ_FB = CHZ_M + CHZ_M / 16;                     // padded frame buffer, cf2 elements
constexpr int CHZ_FBF = 2 * CHZ_FB;                            // ... in floats

__device__ __forceinline__ void dft4(cf2 a0, cf2 a1, cf2 a2, cf2 a3, cf2 (&o)[4])
{
    const cf2 v0 = a0 + a2, v1 = a0 - a2, v2 = a1 + a3, d = a1 - a3;
    o[0] = v0 + v2; o[1] = add_mi(v1, d); o[2] = v0 - v2; o[3] = sub_mi(v1, d);
}

// e^{-2 pi i num / den}, argument reduced exactly (sincospif)
__device__ __forceinline__ cf2 chz_twiddle(int num, int den)
{
    float sn, cs;
    sincospif(-2.0f * (float)num / (float)den, &sn, &cs);
    return (cf2){ cs, sn };
}


// LDS layouts of a frame buffer (cf2 elements unless noted); every access of the pipeline is bank-conflict free (PMC:
// SQ_LDS_BANK_CONFLICT = 0; a one-pad-per-16 layout cost 17 % of the LDS cycles):
//   chz_pos1: pass-1 output, element 4 t + k1 at t + 260 k1 -- the fold's stores are stride 1 across lanes, and pass 2's gather
//             (the compiler pairs its reads into ds_read2_b64: 16-lane groups, 16 bank pairs) hits (i >> 2) + 4 (i & 3) mod 16;
//   chz_pos2: pass-2 output, n + 4 (n >> 6): no padding inside 64 consecutive elements (reads i + 64 r of consecutive lanes are
//             consecutive), and the stride-64 write-back of pass 2 advances 8 banks per four lanes;
//   planar  : pass-3 output = the spectrum in natural order, as FLOATS, in blocks of 128 bins: [re of the block's first 64 bins |
//             re of its last 64 | im of the first 64 | im of the last 64] -- the slicer role owns bins l and l + 64 of a block per
//             lane, so one ds_read2st64_b32 delivers (re of its two channels) as the register PAIR that packed fp32
//             instructions take as it is, the next one (im, im); 64 consecutive lanes read 64 consecutive floats.
__host__ __device__ constexpr int chz_pos1(int t, int k1) { return t + 260 * k1; }
__host__ __device__ constexpr int chz_pos2(int n) { return n + 4 * (n >> 6); }
__host__ __device__ constexpr int chz_planar(int n) { return 256 * (n >> 7) + (n & 127); }   // real part; the imaginary part is 128 floats further
static_assert(chz_pos1(255, 3) < CHZ_FB && chz_pos2(1023) < CHZ_FB && chz_planar(1023) + 128 < CHZ_FBF, "frame buffer too small for the layouts");

// pass 2, radix 16, p = 4, one frame per wave, in place.  lane i: k = i & 3, u[r] = A[i + 64 r] e^{-2 pi i r k / 64},
// X = DFT16(u), A[16 (i - k) + k + 4 r] = X[r].  The input twiddles W_64^{r k} have already been applied by the fold role
// (chz_fold2_ring): read from an LDS table here, one pair at a time between the multiplies, they cost eight exposed LDS
// latencies per pass (measured with s_memtime: 2950 cycles per frame against 1780 for pass 3).
__device__ __forceinline__ void chz_p2(cf2 *A, int lane)
{
    const int k = lane & 3;
    cf2 u[16];
    // pass 1 left element 4 t + k1 at chz_pos1(t, k1): lane i wants 4 t + k1 = i + 64 r, i.e. t = (i >> 2) + 16 r, k1 = i & 3
    const cf2 *src = A + chz_pos1(lane >> 2, k);
#pragma unroll
    for (int r = 0; r < 16; r++) u[r] = src[16 * r];
    // DFT16 = 4 x DFT4 over a (s = 4a + b), twiddle W16^{bc}, 4 x DFT4 over b -> X[c + 4d]
    cf2 v[4][4];
#pragma unroll
    for (int b = 0; b < 4; b++) dft4(u[b], u[4 + b], u[8 + b], u[12 + b], v[b]);
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
    v[1][1] = cmul_s(v[1][1], (cf2){ C1, -S1 });                // W16^1
    v[1][2] = cmul_s(v[1][2], (cf2){ R2, -R2 });                // W16^2
    v[1][3] = cmul_s(v[1][3], (cf2){ S1, -C1 });                // W16^3
    v[2][1] = cmul_s(v[2][1], (cf2){ R2, -R2 });                // W16^2
    v[2][2] = mul_mi(v[2][2]);                                  // W16^4 = -i
    v[2][3] = cmul_s(v[2][3], (cf2){ -R2, -R2 });               // W16^6
    v[3][1] = cmul_s(v[3][1], (cf2){ S1, -C1 });                // W16^3
    v[3][2] = cmul_s(v[3][2], (cf2){ -R2, -R2 });               // W16^6
    v[3][3] = cmul_s(v[3][3], (cf2){ -C1, S1 });                // W16^9
    // all reads of this wave precede its writes in program order; nobody else touches this frame during pass 2
    cf2 *dst = A + 17 * (lane - k) + k;                         // chz_pos2(16 (i-k) + k + 4 r) = 17 (i-k) + k + 4 r   (k + 4 r < 64)
#pragma unroll
    for (int c = 0; c < 4; c++) {
        cf2 X[4];
        dft4(v[0][c], v[1][c], v[2][c], v[3][c], X);
#pragma unroll
        for (int d = 0; d < 4; d++) dst[4 * (c + 4 * d)] = X[d];
    }
}

// pass 3 of the 4 x 16 x 16 factorisation, radix 16, p = 64, one frame per wave:
//   lane i: u[r] = A[i + 64 r] W_1024^{r i};  bin i + 64 q = DFT16(u)[q]   (natural order, planar layout over the same buffer:
//   every output depends on all sixteen inputs, so the reads have returned before the first store is issued)
template <int NT>
__device__ __forceinline__ void chz_p3(cf2 *A, const cf2 (&tw)[NT], int lane)
{
    static_assert(NT == 15 || NT == 1, "fifteen twiddles (NT = 1: the role that does not run pass 3)");
    if constexpr (NT == 15) {
    cf2 u[16];
    const cf2 *src = A + lane;                                  // chz_pos2(lane + 64 r) = lane + 68 r
#pragma unroll
    for (int r = 0; r < 16; r++) u[r] = src[68 * r];
#pragma unroll
    for (int r = 1; r < 16; r++) u[r] = cmul(u[r], tw[r - 1]);
    cf2 v[4][4];
#pragma unroll
    for (int b = 0; b < 4; b++) dft4(u[b], u[4 + b], u[8 + b], u[12 + b], v[b]);
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
    v[1][1] = cmul_s(v[1][1], (cf2){ C1, -S1 });
    v[1][2] = cmul_s(v[1][2], (cf2){ R2, -R2 });
    v[1][3] = cmul_s(v[1][3], (cf2){ S1, -C1 });
    v[2][1] = cmul_s(v[2][1], (cf2){ R2, -R2 });
    v[2][2] = mul_mi(v[2][2]);
    v[2][3] = cmul_s(v[2][3], (cf2){ -R2, -R2 });
    v[3][1] = cmul_s(v[3][1], (cf2){ S1, -C1 });
    v[3][2] = cmul_s(v[3][2], (cf2){ -R2, -R2 });
    v[3][3] = cmul_s(v[3][3], (cf2){ -C1, S1 });
    float *dst = (float *)A + lane;                             // chz_planar(lane + 64 q) = chz_planar(64 q) + lane
#pragma unroll
    for (int c = 0; c < 4; c++) {
        cf2 X[4];
        dft4(v[0][c], v[1][c], v[2][c], v[3][c], X);
#pragma unroll
        for (int d = 0; d < 4; d++) { dst[chz_planar(64 * (c + 4 * d))] = X[d].x; dst[chz_planar(64 * (c + 4 * d)) + 128] = X[d].y; }
    }
    }
}

// Input samples of launch-relative frame F for this thread: virtual indices F*D + t and F*D + 256 + t.  A batch that
// lies inside the new block (all but the first and last of a launch) is plain coalesced loads; the generic path walks
// carry / block / zero padding.  The choice is made once per batch, so the common path has no branch between the four
// folds (per-frame branches cost 8 %: they fence the scheduler).
struct ChzIn {
    const float2 *block, *carry;
    int64_t hist, lead, carry_len, nsamp;
    uint32_t f_last;         // last launch-relative frame whose CHZ_D samples lie wholly inside the new block (FAST prefetch clamp)
    __device__ __forceinline__ cf2 generic(int64_t v) const
    {
        const int64_t ci = v + hist;
        if (ci < 0) return (cf2){ 0.f, 0.f };
        float2 s;
        if (ci < carry_len) s = carry[ci];
        else { const int64_t bi = v - lead; if (bi >= nsamp) return (cf2){ 0.f, 0.f }; s = block[bi]; }
        return (cf2){ s.x, s.y };
    }
    // the same without a branch around the load: the edge paths fetch a batch of these and wait ONCE (round 6: sixteen -- at D = 768
    // nineteen -- generic loads of a workgroup's prologue each behind its own drain were that many HBM round trips in a row)
    __device__ __forceinline__ cf2 generic_nb(int64_t v) const
    {
        const int64_t ci = v + hist, bi = v - lead;
        const bool in_carry = ci < carry_len, ok = ci >= 0 && (in_carry || bi < nsamp);
        const float2 *p = in_carry ? carry + ci : block + bi;
        p = ok ? p : carry;                                           // any valid address
        const float2 s = *p;
        return ok ? (cf2){ s.x, s.y } : (cf2){ 0.f, 0.f };
    }
    // all CHZ_BATCH frames starting at F lie inside the new block (wave-uniform)
    __device__ __forceinline__ bool batch_in_block(int64_t F) const
    {
        const int64_t b0 = F * CHZ_D - lead;
        return b0 >= 0 && b0 + CHZ_BATCH * CHZ_D <= nsamp;
    }
    template <bool FAST>
    __device__ __forceinline__ void frame(int64_t F, int t, cf2 &s0, cf2 &s1) const
    {
        if constexpr (FAST) {
            const float2 *p = block + (F * CHZ_D - lead) + t;
            const float2 u = p[0], w = p[256];
            s0 = (cf2){ u.x, u.y }; s1 = (cf2){ w.x, w.y };
        } else {
            s0 = generic(F * CHZ_D + t); s1 = generic(F * CHZ_D + 256 + t);
        }
    }
};

// The delay lines as a register RING: branch jb keeps P + 4 slots; at a half-step with rotation BASE the
// logical element i of the old {delay line, new samples} view is ring[jb][(BASE + i) % (P + 4)]: elements 0..P-1 the delay
// line, P and P+1 the samples of this half-step's four frames, P+2 and P+3 those of the NEXT half-step (in flight).  After
// the fold the two oldest slots are dead and receive the loads of the half-step after next, and BASE advances by two: no
// register ever moves (round 1 shifted 32 register pairs per four frames), and a load has two half-steps to land.
// The half-step loop is unrolled over the ring's period of (P + 4) / 2 rotations.
//
// Two frames at a time: eight independent accumulator chains (2 frames x 4 branches), two taps per asm block.  Measured
// (scripts/ubench_pk2.hip): a lone wave on a SIMD issues v_pk_fma_f32 with three distinct register-pair sources every 7.1
// cycles with 4 chains and one-instruction asm statements (the compiler puts an s_nop behind every group of dependent asm
// statements -- it counts an asm statement as zero wait states -- and a lone wave pays a full issue slot for it), 6.2 with 8
// chains; inside one asm block there is nothing to pad, and a dependent v_pk_fma is eight instructions away.
#define CHZ_FMA_LO(d, x, c) "v_pk_fma_f32 %" #d ", %" #x ", %" #c ", %" #d " op_sel_hi:[1,0,1]\n\t"
#define CHZ_FMA_HI(d, x, c) "v_pk_fma_f32 %" #d ", %" #x ", %" #c ", %" #d " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
#define CHZ_MUL_LO(d, x, c) "v_pk_mul_f32 %" #d ", %" #x ", %" #c " op_sel_hi:[1,0]\n\t"
// acc[f][jb] (+)= x0[f][jb] * c[jb ^ SW_f].lo ; then += x1[f][jb] * c[jb ^ SW_f].hi   (SW_0 = 2, SW_1 = 0: frame 0 has even parity)
template <bool FIRST>
__device__ __forceinline__ void chz_fold_taps2(cf2 (&acc)[2][4], const cf2 (&x0)[2][4], const cf2 (&x1)[2][4], const cf2 (&c)[4])
{
    // operands: 0..7 acc[f][jb] (f major), 8..15 x0, 16..23 x1, 24..27 c[0..3].  Neighbouring instructions share their coefficient
    // pair: a v_pk_fma_f32 with three distinct register-pair sources issues every 6.3 cycles, 5.6 when every second one repeats a
    // source pair of its predecessor (scripts/ubench_pk3.hip, two waves per SIMD)
    if constexpr (FIRST) {
        asm(CHZ_MUL_LO(2, 10, 24) CHZ_MUL_LO(4, 12, 24) CHZ_MUL_LO(3, 11, 25) CHZ_MUL_LO(5, 13, 25)
            CHZ_MUL_LO(0, 8, 26) CHZ_MUL_LO(6, 14, 26) CHZ_MUL_LO(1, 9, 27) CHZ_MUL_LO(7, 15, 27)
            CHZ_FMA_HI(2, 18, 24) CHZ_FMA_HI(4, 20, 24) CHZ_FMA_HI(3, 19, 25) CHZ_FMA_HI(5, 21, 25)
            CHZ_FMA_HI(0, 16, 26) CHZ_FMA_HI(6, 22, 26) CHZ_FMA_HI(1, 17, 27) CHZ_FMA_HI(7, 23, 27)
            : "=&v"(acc[0][0]), "=&v"(acc[0][1]), "=&v"(acc[0][2]), "=&v"(acc[0][3]), "=&v"(acc[1][0]), "=&v"(acc[1][1]), "=&v"(acc[1][2]), "=&v"(acc[1][3])
            : "v"(x0[0][0]), "v"(x0[0][1]), "v"(x0[0][2]), "v"(x0[0][3]), "v"(x0[1][0]), "v"(x0[1][1]), "v"(x0[1][2]), "v"(x0[1][3]),
              "v"(x1[0][0]), "v"(x1[0][1]), "v"(x1[0][2]), "v"(x1[0][3]), "v"(x1[1][0]), "v"(x1[1][1]), "v"(x1[1][2]), "v"(x1[1][3]),
              "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]));
    } else {
        asm(CHZ_FMA_LO(2, 10, 24) CHZ_FMA_LO(4, 12, 24) CHZ_FMA_LO(3, 11, 25) CHZ_FMA_LO(5, 13, 25)
            CHZ_FMA_LO(0, 8, 26) CHZ_FMA_LO(6, 14, 26) CHZ_FMA_LO(1, 9, 27) CHZ_FMA_LO(7, 15, 27)
            CHZ_FMA_HI(2, 18, 24) CHZ_FMA_HI(4, 20, 24) CHZ_FMA_HI(3, 19, 25) CHZ_FMA_HI(5, 21, 25)
            CHZ_FMA_HI(0, 16, 26) CHZ_FMA_HI(6, 22, 26) CHZ_FMA_HI(1, 17, 27) CHZ_FMA_HI(7, 23, 27)
            : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3])
            : "v"(x0[0][0]), "v"(x0[0][1]), "v"(x0[0][2]), "v"(x0[0][3]), "v"(x0[1][0]), "v"(x0[1][1]), "v"(x0[1][2]), "v"(x0[1][3]),
              "v"(x1[0][0]), "v"(x1[0][1]), "v"(x1[0][2]), "v"(x1[0][3]), "v"(x1[1][0]), "v"(x1[1][1]), "v"(x1[1][2]), "v"(x1[1][3]),
              "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]));
    }
}
// frames FA, FA+1 of a half-step (FA = 0: tap windows start at SA = 1 / SB = 0 and 1 / 1; FA = 2: 2 / 1 and 2 / 2), then the
// radix-4 pass 1 of both -> A[FA], A[FA+1]
// `after_first` runs behind the first tap block (q = 0): that block holds the last uses of the two oldest elements of the
// ring's current view, so the caller refills those slots from there (chz_load1_ring)
template <int P, int BASE, int FA, typename Hook>
__device__ __forceinline__ void chz_fold2_ring(const cf2 (&ring)[4][P + 4], const cf2 (&coef)[4][P / 2], const cf2 (&tw1)[3], cf2 *bufA, int t, Hook &&after_first)
{
    constexpr int R = P + 4;
    constexpr int S[2][2] = { { FA == 0 ? 1 : 2, FA == 0 ? 0 : 1 }, { FA == 0 ? 1 : 2, FA == 0 ? 1 : 2 } };   // [frame][jb >= 2]
    cf2 acc[2][4];
#pragma unroll
    for (int q = 0; q < P; q += 2) {
        cf2 x0[2][4], x1[2][4], c[4];
#pragma unroll
        for (int f = 0; f < 2; f++)
#pragma unroll
            for (int jb = 0; jb < 4; jb++) {
                x0[f][jb] = ring[jb][(BASE + S[f][jb >> 1] + q) % R];
                x1[f][jb] = ring[jb][(BASE + S[f][jb >> 1] + q + 1) % R];
            }
#pragma unroll
        for (int j = 0; j < 4; j++) c[j] = coef[j][q / 2];
        if (q == 0) { chz_fold_taps2<true>(acc, x0, x1, c); after_first(); } else chz_fold_taps2<false>(acc, x0, x1, c);
    }
#pragma unroll
    for (int f = 0; f < 2; f++) {
        cf2 o[4];
        dft4(acc[f][0], acc[f][1], acc[f][2], acc[f][3], o);     // radix 4, p = 1: no twiddles
        // the input twiddles of pass 2 (element 4 t + k1 is its point r = t >> 4 of lane 4 (t & 15) + k1: W_64^{r k1}) are applied
        // HERE: the fold waves wait at the barriers more than half of the time, the FFT waves are the critical path of a time step
        cf2 *d = bufA + (FA + f) * CHZ_FB + t;
        d[chz_pos1(0, 0)] = o[0]; d[chz_pos1(0, 1)] = cmul(o[1], tw1[0]); d[chz_pos1(0, 2)] = cmul(o[2], tw1[1]); d[chz_pos1(0, 3)] = cmul(o[3], tw1[2]);
    }
}
// The two samples frame F + G brings for this thread go to branches 2 (G & 1) + {0, 1}, logical element ELEM + (G >> 1).
// FAST (the four frames of the half-batch lie inside the new block): the loads are issued as inline asm, so that the compiler
// does not track them -- its own bookkeeping puts `s_waitcnt vmcnt(0)` behind every barrier of the loop (the fast / generic
// join makes it conservative), which drains the loads issued a moment ago.  The fold waits with chz_ring_wait instead: vmcnt(8)
// = "everything but the eight youngest loads", i.e. exactly the loads of the previous half-step stay in flight.  The address is
// a wave-uniform SGPR base plus the lane's constant byte offset (saddr form: no per-load address arithmetic on the VALU, one
// address dword to move instead of two: with 64-bit VGPR addresses the eight loads of a half-step took the fold waves ~350
// cycles at the END of their step, when every other wave of the CU was already waiting at the barrier).
// The generic path (carry, zero padding: first and last half-steps of a launch) uses ordinary loads and drains them before it
// returns, so vmcnt(8) is right after either path.
template <int P, int BASE, int ELEM, int G, bool FAST>
__device__ __forceinline__ void chz_load1_ring(cf2 (&ring)[4][P + 4], const ChzIn &in, int64_t F, int t, chz_rsrc_t rsrc = chz_rsrc_t{}, uint32_t soff = 0u)
{
    constexpr int R = P + 4;
    constexpr int E = (BASE + ELEM + (G >> 1)) % R, J = 2 * (G & 1);
    if constexpr (FAST) {
        // Round 6: raw buffer loads against the workgroup's descriptor (chz_make_rsrc) -- `soff` = byte offset of the half-step in work,
        // the frame fetched here lies 2 * CHZ_BATCH + G frames behind its first sample; the hardware's range check stands in for the
        // clamp of rounds 3-5 (frame index against the last whole frame of the block, on the scalar unit since round 5), and the
        // 64-bit base arithmetic per frame -- ~45 scalar instructions per half-step in the wave that is the kernel's critical path -- is
        // one s_add.
        // "+v": the destination is TIED to the register that holds the slot's dead value, so the new value is born in the ring's own
        // register -- with "=v" the compiler is free to load into a scratch pair and copy it into place at the next control-flow
        // join, i.e. to READ a register whose load is still in flight (it did: tests/test_cpu_inflight_loads.py scans the
        // assembly for any access to such a register before the wait that covers it)
        // (non-temporal loads change nothing here, 0.380 against 0.381 ms: the kernel is bound by VALU issue, not by its input stream)
        const uint32_t voff = (uint32_t)t * (uint32_t)sizeof(float2);
        const uint32_t so = soff + (uint32_t)((2 * CHZ_BATCH + G) * CHZ_D * sizeof(float2));
        asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "+v"(ring[J][E]) : "v"(voff), "s"(rsrc), "s"(so));
        asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen offset:2048" : "+v"(ring[J + 1][E]) : "v"(voff), "s"(rsrc), "s"(so));
    } else {
        cf2 s0, s1;
        in.template frame<false>(F + G, t, s0, s1);
        // a use of both values: the compiler drains its loads HERE and carries no pending load out of this path; they then move
        // into the slot's own registers through the same tie as the fast path's loads, so that the two paths join with the ring
        // where it is and the compiler has nothing to copy behind an in-flight load
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(s0), "+v"(s1) :: "memory");
        asm volatile("v_mov_b64 %0, %1" : "+v"(ring[J][E]) : "v"(s0));
        asm volatile("v_mov_b64 %0, %1" : "+v"(ring[J + 1][E]) : "v"(s1));
    }
}
// the eight generic loads of a half-step as ONE batch: all of them in flight, one drain, then into their slots through the same ties
template <int P, int BASE, int ELEM>
__device__ __forceinline__ void chz_load_half_ring_generic(cf2 (&ring)[4][P + 4], const ChzIn &in, int64_t F, int t)
{
    constexpr int R = P + 4;
    cf2 v[8];
#pragma unroll
    for (int g = 0; g < 4; g++) { v[2 * g] = in.generic_nb((F + g) * CHZ_D + t); v[2 * g + 1] = in.generic_nb((F + g) * CHZ_D + 256 + t); }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int e = (BASE + ELEM + (g >> 1)) % R, j = 2 * (g & 1);
        asm volatile("v_mov_b64 %0, %1" : "+v"(ring[j][e]) : "v"(v[2 * g]));
        asm volatile("v_mov_b64 %0, %1" : "+v"(ring[j + 1][e]) : "v"(v[2 * g + 1]));
    }
}
template <int P, int BASE, int ELEM, bool FAST>
__device__ __forceinline__ void chz_load_half_ring(cf2 (&ring)[4][P + 4], const ChzIn &in, int64_t F, int t)
{
    if constexpr (!FAST) { chz_load_half_ring_generic<P, BASE, ELEM>(ring, in, F, t); return; }
    chz_load1_ring<P, BASE, ELEM, 0, FAST>(ring, in, F, t);
    chz_load1_ring<P, BASE, ELEM, 1, FAST>(ring, in, F, t);
    chz_load1_ring<P, BASE, ELEM, 2, FAST>(ring, in, F, t);
    chz_load1_ring<P, BASE, ELEM, 3, FAST>(ring, in, F, t);
}
// before a fold: the samples of logical elements P and P+1 (loaded two half-steps ago) have arrived; the empty asm makes
// every use of those registers depend on the wait
template <int P, int BASE>
__device__ __forceinline__ void chz_ring_wait(cf2 (&ring)[4][P + 4])
{
    constexpr int R = P + 4;
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("" : "+v"(ring[0][(BASE + P) % R]), "+v"(ring[1][(BASE + P) % R]), "+v"(ring[2][(BASE + P) % R]), "+v"(ring[3][(BASE + P) % R]),
                      "+v"(ring[0][(BASE + P + 1) % R]), "+v"(ring[1][(BASE + P + 1) % R]), "+v"(ring[2][(BASE + P + 1) % R]), "+v"(ring[3][(BASE + P + 1) % R]));
}



// ---- D = 768: the fold role of the 4/3 x oversampled bank (M = 1024, 40 ksps per channel = two samples per Manchester symbol) ----
// Everything per FRAME is what it is at D = 512 -- eight taps on each of 1024 branches, the same FFT, the same slicer work per bin --
// but a frame now consumes 768 input samples instead of 512: 1.5 x fewer frames per input byte (the D = 512 form ends at
// 0.33 of HBM, bound by VALU issue).  What changes is the bookkeeping of the fold, all of it compile-time:
//   * frame F (F = 0 mod 4 at the start of a half-step; launches and workgroup ranges start on multiples of four frames, where the
//     stream position is a multiple of M) brings the chunks k = 0, 1, 2 of 256 samples; chunk k belongs to branch j = (k - g) & 3,
//     g = F & 3: per half-step of four frames every branch receives THREE samples -- branch j in the frames g with (j + g) & 3 != 3;
//   * branch j of frame g is weighted with coefficient set (j + g + 1) & 3 (the prototype is indexed by the sample's distance from
//     the frame's start, (r - n0) mod M with n0 = (F + 1) D - L = 256 (g + 1) mod M) -- the same 32 coefficients per thread as at
//     D = 512, where the set alternates between j and j ^ 2;
//   * the register ring keeps P + 4 = 12 slots per branch, as at D = 512, and advances by three per half-step: period FOUR
//     half-steps.  In the view of a half-step, elements 0..7 are the delay line, 8..10 its own three samples, 11 the first sample of
//     the next one.  Frame g reads the window [cnt(j, g), cnt(j, g) + 8) of branch j, cnt = samples received so far; every element
//     0..3 dies in the FIRST tap block of a frame pair (position 0 of its last window), and the slot takes a load at once:
//        after the first tap block of frames 0 / 1:  element 0 of branch 3, element 1 of every branch, element 2 of branch 0
//        after the first tap block of frames 2 / 3:  element 2 of branches 1..3, element 3 of branches 0..2
//     The slot of element e receives element e + 12: the samples of the next half-step (9, 10 there) and of the one after (8, 9
//     there), issued in the order they will be needed, so ONE counter serves: s_waitcnt vmcnt(13) in front of either frame pair
//     leaves exactly the thirteen loads issued behind the last one that pair reads.  A load has five to seven frames to land
//     (D = 512: four to eight), 13 x 512 B x 4 waves = 26 KB per CU are in flight at the least.
__host__ __device__ constexpr int chz768_recv(int j, int g) { return ((j + g) & 3) != 3; }           // branch j receives a sample in frame g
__host__ __device__ constexpr int chz768_cnt(int j, int g) { int c = 0; for (int q = 0; q <= g; q++) c += chz768_recv(j, q); return c; }
static_assert(chz768_cnt(0, 0) == 1 && chz768_cnt(0, 3) == 3 && chz768_cnt(1, 2) == 2 && chz768_cnt(2, 1) == 1 && chz768_cnt(3, 0) == 0 && chz768_cnt(3, 3) == 3, "reception table");
constexpr int CHZ768_R = 12;
// frames FA, FA + 1 of a half-step (FA = 0 or 2), then the radix-4 pass 1 of both -> A[FA], A[FA + 1].  The asm block of
// chz_fold_taps2 is the D = 512 one -- its first frame's accumulator jb uses coefficient pair c[jb ^ 2], its second frame's c[jb] --
// handed permuted operands: c[m] = set (m + FA + 2) & 3, second frame = branch jb, first frame = branch (jb + 3) & 3 (all
// indices compile-time: no register moves)
template <int P, int BASE, int FA, typename Hook>
__device__ __forceinline__ void chz768_fold2_ring(const cf2 (&ring)[4][CHZ768_R], const cf2 (&coef)[4][P / 2], const cf2 (&tw1)[3], cf2 *bufA, int t, Hook &&after_first)
{
    static_assert(P == 8 && (FA == 0 || FA == 2), "eight taps per branch, frame pairs 0 / 1 and 2 / 3");
    constexpr int R = CHZ768_R;
    cf2 acc[2][4];
#pragma unroll
    for (int q = 0; q < P; q += 2) {
        cf2 x0[2][4], x1[2][4], c[4];
#pragma unroll
        for (int jb = 0; jb < 4; jb++) {
            const int j0 = (jb + 3) & 3, s0 = chz768_cnt(j0, FA), s1 = chz768_cnt(jb, FA + 1);
            x0[0][jb] = ring[j0][(BASE + s0 + q) % R]; x1[0][jb] = ring[j0][(BASE + s0 + q + 1) % R];
            x0[1][jb] = ring[jb][(BASE + s1 + q) % R]; x1[1][jb] = ring[jb][(BASE + s1 + q + 1) % R];
        }
#pragma unroll
        for (int m = 0; m < 4; m++) c[m] = coef[(m + FA + 2) & 3][q / 2];
        if (q == 0) { chz_fold_taps2<true>(acc, x0, x1, c); after_first(); } else chz_fold_taps2<false>(acc, x0, x1, c);
    }
#pragma unroll
    for (int f = 0; f < 2; f++) {
        cf2 o[4];
        if (f == 0) dft4(acc[0][1], acc[0][2], acc[0][3], acc[0][0], o);     // branch j of the first frame sits in accumulator (j + 1) & 3
        else dft4(acc[1][0], acc[1][1], acc[1][2], acc[1][3], o);
        cf2 *d = bufA + (FA + f) * CHZ_FB + t;
        d[chz_pos1(0, 0)] = o[0]; d[chz_pos1(0, 1)] = cmul(o[1], tw1[0]); d[chz_pos1(0, 2)] = cmul(o[2], tw1[1]); d[chz_pos1(0, 3)] = cmul(o[3], tw1[2]);
    }
}
// The sample branch J receives in frame G of the half-step X half-steps behind the one in work (whose first frame is F0, ring rotation
// BASE): chunk K = (J + G) & 3 of that frame's 768 new samples, into the slot its reception number says.  FAST / generic as at
// D = 512 (chz_load1_ring): untracked inline-asm loads tied to the slot's own registers, or a bounds-checked load drained at once.
// FAST: one raw buffer load (chz_make_rsrc).  `soff` = byte offset of the half-step in work inside the workgroup's descriptor; the
// chunks a half-step fetches are the CONSECUTIVE 2 KB chunks 19 .. 30 behind its own first sample -- the stream is read in order --
// so chunk c is soff + 2048 (c & ~1) with the odd ones on the 12-bit immediate.
template <int BASE, int X, int G, int J, bool FAST>
__device__ __forceinline__ void chz768_load1(cf2 (&ring)[4][CHZ768_R], const ChzIn &in, int64_t F0, int t, chz_rsrc_t rsrc = chz_rsrc_t{}, uint32_t soff = 0u)
{
    constexpr int K = (J + G) & 3;
    static_assert(K != 3, "branch J receives nothing in frame G");
    constexpr int SLOT = (BASE + 7 + chz768_cnt(J, G) + 3 * X) % CHZ768_R;
    if constexpr (FAST) {
        constexpr int CH = 3 * (4 * X + G) + K;                      // chunk of 256 samples behind the half-step's first sample
        const uint32_t voff = (uint32_t)t * (uint32_t)sizeof(float2);
#ifdef CHZ_TIMING_CACHED_LOADS
        // TIMING EXPERIMENT ONLY (profiles/r06/chz_no_loads.txt): every half-step reads the SAME 64 KB of its workgroup's range -- real (non-zero)
        // data out of the L2, no HBM traffic; WRONG results
        const uint32_t so = (soff & 0x7fffu) + 2048u * (uint32_t)(CH & ~1);
#else
        const uint32_t so = soff + 2048u * (uint32_t)(CH & ~1);
#endif
        // "+v": the new value is born in the ring's own register (see chz_load1_ring)
        if constexpr (CH & 1) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen offset:2048" : "+v"(ring[J][SLOT]) : "v"(voff), "s"(rsrc), "s"(so));
        else asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "+v"(ring[J][SLOT]) : "v"(voff), "s"(rsrc), "s"(so));
    } else {
        cf2 s0 = in.generic((F0 + 4 * X + G) * CHZ_D768 + 256 * K + t);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(s0) :: "memory");
        asm volatile("v_mov_b64 %0, %1" : "+v"(ring[J][SLOT]) : "v"(s0));
    }
}
// generic form in two halves, so that a batch of fetches shares one drain
template <int X, int G, int J>
__device__ __forceinline__ cf2 chz768_fetch(const ChzIn &in, int64_t F0, int t) { return in.generic_nb((F0 + 4 * X + G) * CHZ_D768 + 256 * ((J + G) & 3) + t); }
template <int BASE, int X, int G, int J>
__device__ __forceinline__ void chz768_put(cf2 (&ring)[4][CHZ768_R], cf2 v)
{
    constexpr int SLOT = (BASE + 7 + chz768_cnt(J, G) + 3 * X) % CHZ768_R;
    asm volatile("v_mov_b64 %0, %1" : "+v"(ring[J][SLOT]) : "v"(v));
}
// the twelve loads of an EDGE half-step (both lists below), one drain
template <int BASE>
__device__ __forceinline__ void chz768_loads_generic(cf2 (&ring)[4][CHZ768_R], const ChzIn &in, int64_t F0, int t)
{
    cf2 v[12] = { chz768_fetch<1, 2, 3>(in, F0, t), chz768_fetch<1, 2, 0>(in, F0, t), chz768_fetch<1, 3, 1>(in, F0, t), chz768_fetch<1, 3, 2>(in, F0, t),
                  chz768_fetch<1, 3, 3>(in, F0, t), chz768_fetch<2, 0, 0>(in, F0, t), chz768_fetch<2, 0, 1>(in, F0, t), chz768_fetch<2, 0, 2>(in, F0, t),
                  chz768_fetch<2, 1, 3>(in, F0, t), chz768_fetch<2, 1, 0>(in, F0, t), chz768_fetch<2, 1, 1>(in, F0, t), chz768_fetch<2, 2, 2>(in, F0, t) };
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]) :: "memory");
    chz768_put<BASE, 1, 2, 3>(ring, v[0]); chz768_put<BASE, 1, 2, 0>(ring, v[1]); chz768_put<BASE, 1, 3, 1>(ring, v[2]); chz768_put<BASE, 1, 3, 2>(ring, v[3]);
    chz768_put<BASE, 1, 3, 3>(ring, v[4]); chz768_put<BASE, 2, 0, 0>(ring, v[5]); chz768_put<BASE, 2, 0, 1>(ring, v[6]); chz768_put<BASE, 2, 0, 2>(ring, v[7]);
    chz768_put<BASE, 2, 1, 3>(ring, v[8]); chz768_put<BASE, 2, 1, 0>(ring, v[9]); chz768_put<BASE, 2, 1, 1>(ring, v[10]); chz768_put<BASE, 2, 2, 2>(ring, v[11]);
}
// what the half-steps -2 and -1 would have left in flight when half-step 0 (first frame F0, BASE 0) begins: its own three samples per
// branch (slots 8..10), the first of half-step 1 (slot 11) and -- branches 0..2, whose element 0 is dead already -- the second (slot 0)
__device__ __forceinline__ void chz768_prime(cf2 (&ring)[4][CHZ768_R], const ChzIn &in, int64_t F0, int t)
{
    cf2 a[12] = { chz768_fetch<0, 0, 0>(in, F0, t), chz768_fetch<0, 0, 1>(in, F0, t), chz768_fetch<0, 0, 2>(in, F0, t), chz768_fetch<0, 1, 3>(in, F0, t),
                  chz768_fetch<0, 1, 0>(in, F0, t), chz768_fetch<0, 1, 1>(in, F0, t), chz768_fetch<0, 2, 2>(in, F0, t), chz768_fetch<0, 2, 3>(in, F0, t),
                  chz768_fetch<0, 2, 0>(in, F0, t), chz768_fetch<0, 3, 1>(in, F0, t), chz768_fetch<0, 3, 2>(in, F0, t), chz768_fetch<0, 3, 3>(in, F0, t) };
    cf2 b[7] = { chz768_fetch<1, 0, 0>(in, F0, t), chz768_fetch<1, 0, 1>(in, F0, t), chz768_fetch<1, 0, 2>(in, F0, t), chz768_fetch<1, 1, 3>(in, F0, t),
                 chz768_fetch<1, 1, 0>(in, F0, t), chz768_fetch<1, 1, 1>(in, F0, t), chz768_fetch<1, 2, 2>(in, F0, t) };
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]) :: "memory");
    asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]));
    chz768_put<0, 0, 0, 0>(ring, a[0]); chz768_put<0, 0, 0, 1>(ring, a[1]); chz768_put<0, 0, 0, 2>(ring, a[2]); chz768_put<0, 0, 1, 3>(ring, a[3]);
    chz768_put<0, 0, 1, 0>(ring, a[4]); chz768_put<0, 0, 1, 1>(ring, a[5]); chz768_put<0, 0, 2, 2>(ring, a[6]); chz768_put<0, 0, 2, 3>(ring, a[7]);
    chz768_put<0, 0, 2, 0>(ring, a[8]); chz768_put<0, 0, 3, 1>(ring, a[9]); chz768_put<0, 0, 3, 2>(ring, a[10]); chz768_put<0, 0, 3, 3>(ring, a[11]);
    chz768_put<0, 1, 0, 0>(ring, b[0]); chz768_put<0, 1, 0, 1>(ring, b[1]); chz768_put<0, 1, 0, 2>(ring, b[2]); chz768_put<0, 1, 1, 3>(ring, b[3]);
    chz768_put<0, 1, 1, 0>(ring, b[4]); chz768_put<0, 1, 1, 1>(ring, b[5]); chz768_put<0, 1, 2, 2>(ring, b[6]);
}
// the six loads behind the first tap block of frames 0 / 1, and the six behind that of frames 2 / 3, in the order they are needed
template <int BASE, bool FAST>
__device__ __forceinline__ void chz768_loads_a(cf2 (&ring)[4][CHZ768_R], const ChzIn &in, int64_t F0, int t, chz_rsrc_t rs, uint32_t so)
{
    chz768_load1<BASE, 1, 2, 3, FAST>(ring, in, F0, t, rs, so); chz768_load1<BASE, 1, 2, 0, FAST>(ring, in, F0, t, rs, so);
    chz768_load1<BASE, 1, 3, 1, FAST>(ring, in, F0, t, rs, so); chz768_load1<BASE, 1, 3, 2, FAST>(ring, in, F0, t, rs, so);
    chz768_load1<BASE, 1, 3, 3, FAST>(ring, in, F0, t, rs, so); chz768_load1<BASE, 2, 0, 0, FAST>(ring, in, F0, t, rs, so);
}
template <int BASE, bool FAST>
__device__ __forceinline__ void chz768_loads_b(cf2 (&ring)[4][CHZ768_R], const ChzIn &in, int64_t F0, int t, chz_rsrc_t rs, uint32_t so)
{
    chz768_load1<BASE, 2, 0, 1, FAST>(ring, in, F0, t, rs, so); chz768_load1<BASE, 2, 0, 2, FAST>(ring, in, F0, t, rs, so);
    chz768_load1<BASE, 2, 1, 3, FAST>(ring, in, F0, t, rs, so); chz768_load1<BASE, 2, 1, 0, FAST>(ring, in, F0, t, rs, so);
    chz768_load1<BASE, 2, 1, 1, FAST>(ring, in, F0, t, rs, so); chz768_load1<BASE, 2, 2, 2, FAST>(ring, in, F0, t, rs, so);
}
// in front of frames 0 / 1: element 8 of every branch and element 9 of branches 0, 1 have arrived (everything but the thirteen
// youngest loads); in front of frames 2 / 3: element 9 of branches 2, 3 and element 10 of every branch (again thirteen)
template <int BASE, int FA>
__device__ __forceinline__ void chz768_ring_wait(cf2 (&ring)[4][CHZ768_R])
{
    constexpr int R = CHZ768_R;
    asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    if constexpr (FA == 0)
        asm volatile("" : "+v"(ring[0][(BASE + 8) % R]), "+v"(ring[1][(BASE + 8) % R]), "+v"(ring[2][(BASE + 8) % R]), "+v"(ring[3][(BASE + 8) % R]),
                          "+v"(ring[0][(BASE + 9) % R]), "+v"(ring[1][(BASE + 9) % R]));
    else
        asm volatile("" : "+v"(ring[2][(BASE + 9) % R]), "+v"(ring[3][(BASE + 9) % R]),
                          "+v"(ring[0][(BASE + 10) % R]), "+v"(ring[1][(BASE + 10) % R]), "+v"(ring[2][(BASE + 10) % R]), "+v"(ring[3][(BASE + 10) % R]));
}

// ---- the three-role pipeline ----
// Round 2's kernel had two roles (4 fold waves, 8 FFT waves that also ran the slicer): the eight FFT waves moved in lock step
// -- all reading LDS, all computing, all writing -- so LDS bursts and VALU bursts alternated instead of overlapping, two FFT
// waves in the same phase shared every SIMD, and the slicer sat on their critical path behind a second barrier (VALU issue
// slots 0.57 busy, waves parked 39 % of their cycles).  Here a time step is FOUR frames and every SIMD holds three waves in
// three different phases:
//   "fold"  : thread t keeps branches t + 256 j as a register ring, folds the four frames of half-batch h and runs
//                         the radix-4 pass 1 (+ pass 2's input twiddles) on its own registers -> slot h & 3.  Pure VALU +
//                         prefetched global loads.
//   waves 4..7  "pass 2": wave w transforms frame w of half-batch h - 1 (radix 16, p = 4), in place.
//   waves 8..11 "pass 3": wave w first slices half-batch h - 3 (all four frames, four channels per lane, planar operands),
//                         then transforms frame w of half-batch h - 2 (radix 16, p = 64) into the planar layout.
// ONE workgroup barrier per four frames (round 2: two per eight, with the slicer between them).  A half-batch lives four time
// steps in a four-slot ring of 4 x 4 frame buffers = 136 KB of LDS; every role stays below 168 VGPRs: three waves per SIMD.
// The unfused form is the same kernel with a different epilogue (MODE = CHZ12_IQ: the bins leave as 32-byte runs of the
// channel-major block), so fused and unfused forms stay bit-identical by construction.
constexpr int CHZ_SLOTS = 4;                                     // half-batches in flight
constexpr int CHZ_PREROLL = 8;                                   // frames re-run in front of a workgroup's range (two half-batches)
constexpr int CHZ12_IQ = -1;                                     // MODE: write the channel-major block; >= 0: AMPS_SLICER_* fused behind the FFT

// Slicer state of TWO channels of one lane, planar (.x = the first channel, .y = the second): the specs of
// include/amps_recc_numerics.h, operation by operation, two channels per packed instruction.
// SPS = frames per Manchester symbol: 3 behind the D = 512 bank, 2 behind the D = 768 one (the partner of specs B / D is SPS frames
// back, the boxcar of specs A / C is SPS long -- at SPS = 2 its ordered sum is d[n-1] + d[n] whatever the parity)
template <int SL, int SPS = 3> struct ChzSlicePair {
    static_assert(SPS == 2 || SPS == 3, "frames per symbol");
    f2 pr, pi;               // spec A / C: the bins one frame earlier
    f2 d1, d2;               // spec A / C: the last two discriminator outputs
    f2 h1r, h1i, h2r, h2i, h3r, h3i;   // spec B: the bins one, two and three frames earlier
    uint32_t gw[2];          // SIGN bits of the statistic, newest at bit 0 (the slicer bit is the inverted sign)
    // spec D: three sign streams per channel as shift registers (newest at bit 0) -- Im y, Im(y conj(y[n-1])), Im(y conj(y[n-3])) --
    // and, latched every 32 frames, the previous word of Im y's signs and of the two wrap words (exact_word)
    uint32_t sx[2], st[2], sxp[2], wpp[2], wmp[2];   // (the third stream's signs live in gw)
    __device__ __forceinline__ void reset()
    {
        const f2 z = { 0.f, 0.f };
        pr = pi = d1 = d2 = h1r = h1i = h2r = h2i = h3r = h3i = z;
        gw[0] = gw[1] = 0u;                                           // "ones before the stream": sign bits clear
        sx[0] = sx[1] = st[0] = st[1] = sxp[0] = sxp[1] = wpp[0] = wpp[1] = wmp[0] = wmp[1] = 0u;
    }
    // Spec D, once per 32 frames and channel: the slicer bits of the word just completed (newest at bit 0) from the sign words,
    // 32 frames per instruction.  delay(W, Wprev, j) = the stream j frames earlier.
    __device__ __forceinline__ uint32_t exact_word(int e)
    {
        uint32_t wp, wm;
        const uint32_t g = SPS == 3 ? exact_slice_word3(sx[e], st[e], gw[e], sxp[e], wpp[e], wmp[e], wp, wm)    // recc_front.hip.h
                                    : exact_slice_word2(sx[e], st[e], gw[e], sxp[e], wpp[e], wmp[e], wp, wm);
        sxp[e] = sx[e]; wpp[e] = wp; wmp[e] = wm;
        return g;
    }
    template <int PAR> __device__ __forceinline__ void step(f2 yr, f2 yi)   // PAR = parity of the absolute frame index
    {
        if constexpr (SL == AMPS_SLICER_EXACT) {
            const f2 it = __builtin_elementwise_fma(yi, pr, -(yr * pi));       // Im(y conj(y[n-1]))
            const f2 qr = SPS == 3 ? h3r : h2r, qi = SPS == 3 ? h3i : h2i;
            const f2 ic = __builtin_elementwise_fma(yi, qr, -(yr * qi));       // Im(y conj(y[n-SPS]))
            sx[0] = __builtin_amdgcn_alignbit(sx[0], __float_as_uint(yi.x), 31);
            sx[1] = __builtin_amdgcn_alignbit(sx[1], __float_as_uint(yi.y), 31);
            st[0] = __builtin_amdgcn_alignbit(st[0], __float_as_uint(it.x), 31);
            st[1] = __builtin_amdgcn_alignbit(st[1], __float_as_uint(it.y), 31);
            gw[0] = __builtin_amdgcn_alignbit(gw[0], __float_as_uint(ic.x), 31);
            gw[1] = __builtin_amdgcn_alignbit(gw[1], __float_as_uint(ic.y), 31);
            h3r = h2r; h3i = h2i; h2r = pr; h2i = pi; pr = yr; pi = yi;
        } else if constexpr (SL == AMPS_SLICER_PRODUCT) {
            // g = !signbit(yi * pr3 - yr * pi3), the partner SPS frames (one Manchester symbol) earlier
            const f2 sd = SPS == 3 ? yi * h3r - yr * h3i : yi * h2r - yr * h2i;
            gw[0] = __builtin_amdgcn_alignbit(gw[0], __float_as_uint(sd.x), 31);
            gw[1] = __builtin_amdgcn_alignbit(gw[1], __float_as_uint(sd.y), 31);
            h3r = h2r; h3i = h2i; h2r = h1r; h2i = h1i; h1r = yr; h1i = yi;
        } else {
            f2 d;
            if constexpr (SL == AMPS_SLICER_SINE) d = __builtin_elementwise_fma(yi, pr, -(yr * pi));     // Im(y conj(prev))
            else {
                const f2 re = __builtin_elementwise_fma(yr, pr, yi * pi);                                 // y conj(prev)
                const f2 im = __builtin_elementwise_fma(yi, pr, -(yr * pi));
                d = fm_phase_planar(re, im);
            }
            // window [n-2, n]: n even -> (d[n-2] + d[n-1]) + d[n];  n odd -> d[n-2] + (d[n-1] + d[n])
            const f2 s = SPS == 2 ? d1 + d : PAR == 0 ? (d2 + d1) + d : d2 + (d1 + d);
            const f2 sp = SL == AMPS_SLICER_SINE ? s : s + (f2){ 0.0f, 0.0f };   // spec A: g = (S >= 0), i.e. -0 counts as +0 (see step4)
            gw[0] = __builtin_amdgcn_alignbit(gw[0], __float_as_uint(sp.x), 31);
            gw[1] = __builtin_amdgcn_alignbit(gw[1], __float_as_uint(sp.y), 31);
            d2 = d1; d1 = d;
            pr = yr; pi = yi;
        }
    }
    // the four frames of a half-batch (frame 0 has even parity).  Spec A: the four arctangents are independent of the stream
    // state, so they are evaluated in lock step (fm_phase_planar_n) before the sequential boxcar / slicer part
    __device__ __forceinline__ void step4(const f2 (&yr)[4], const f2 (&yi)[4])
    {
        if constexpr (SL == AMPS_SLICER_ATAN_BOXCAR) {
            f2 re[4], im[4], d[4];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f2 qr = g ? yr[g - 1] : pr, qi = g ? yi[g - 1] : pi;
                re[g] = __builtin_elementwise_fma(yr[g], qr, yi[g] * qi);                                   // y conj(prev)
                im[g] = __builtin_elementwise_fma(yi[g], qr, -(yr[g] * qi));
            }
            fm_phase_planar_n<4>(re, im, d);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f2 s = SPS == 2 ? d1 + d[g] : (g & 1) == 0 ? (d2 + d1) + d[g] : d2 + (d1 + d[g]);
                // g = (S >= 0): S + (+0) turns the one negative-signed value that counts as >= 0, -0, into +0 (and leaves every
                // other finite S alone), after which the bit is the inverted sign -- one packed add and two funnel shifts for
                // the pair instead of two compares, two selects and two shifts
                const f2 sp = s + (f2){ 0.0f, 0.0f };
                gw[0] = __builtin_amdgcn_alignbit(gw[0], __float_as_uint(sp.x), 31);
                gw[1] = __builtin_amdgcn_alignbit(gw[1], __float_as_uint(sp.y), 31);
                d2 = d1; d1 = d[g];
            }
            pr = yr[3]; pi = yi[3];
        } else {
#pragma unroll
            for (int g = 0; g < 4; g++) { if (g & 1) step<1>(yr[g], yi[g]); else step<0>(yr[g], yi[g]); }
        }
    }
    // Spec D with the three frames of history handed in instead of kept: hr[0] / hi[0] = the bins one frame before yr[0], hr[1] two,
    // hr[2] three.  Nothing is copied at the end -- the caller's four frames ARE the next step's history (ChzSlicer::half: two
    // buffers used alternately, the parity a compile-time constant).  Round 4 kept (pr, pi, h2, h3) as members: six v_mov_b64 per
    // channel pair and time step across the loop's back edge, 12 of the kernel's 528 VALU instructions per frame.
    __device__ __forceinline__ void step4_hist(const f2 (&yr)[4], const f2 (&yi)[4], const f2 (&hr)[3], const f2 (&hi)[3])
    {
        static_assert(SL == AMPS_SLICER_EXACT, "spec D only");
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const f2 p1r = g >= 1 ? yr[g - 1] : hr[0], p1i = g >= 1 ? yi[g - 1] : hi[0];
            const f2 p3r = g >= SPS ? yr[g - SPS] : hr[SPS - 1 - g], p3i = g >= SPS ? yi[g - SPS] : hi[SPS - 1 - g];
            const f2 it = __builtin_elementwise_fma(yi[g], p1r, -(yr[g] * p1i));       // Im(y conj(y[n-1]))
            const f2 ic = __builtin_elementwise_fma(yi[g], p3r, -(yr[g] * p3i));       // Im(y conj(y[n-SPS]))
            sx[0] = __builtin_amdgcn_alignbit(sx[0], __float_as_uint(yi[g].x), 31);
            sx[1] = __builtin_amdgcn_alignbit(sx[1], __float_as_uint(yi[g].y), 31);
            st[0] = __builtin_amdgcn_alignbit(st[0], __float_as_uint(it.x), 31);
            st[1] = __builtin_amdgcn_alignbit(st[1], __float_as_uint(it.y), 31);
            gw[0] = __builtin_amdgcn_alignbit(gw[0], __float_as_uint(ic.x), 31);
            gw[1] = __builtin_amdgcn_alignbit(gw[1], __float_as_uint(ic.y), 31);
        }
    }
    // the ring word of the 32 frames just completed (oldest frame at bit 0)
    __device__ __forceinline__ uint32_t word(int e)
    {
        if constexpr (SL == AMPS_SLICER_EXACT) return __builtin_bitreverse32(exact_word(e));
        else return ~__builtin_bitreverse32(gw[e]);
    }
};

// The slicer of the pass-3 role: NP = 2 channel pairs per lane (pairs J0 .. J0 + NP - 1 of the wave's eight).
// Bin ownership.  The planar layout keeps a frame in eight blocks of 128 bins, [re of bins 0..63 | re of 64..127 | im | im]; a lane's
// PAIR is two bins 64 apart in one block, so the four floats a frame brings for it sit at base + {0, 64, 128, 192} (two
// ds_read2st64_b32) and arrive as the register pairs (re, re) and (im, im) of its two channels.  With W = grp_w residues per
// block belonging to this handle (64 = all of them) a frame holds 8 W pairs, numbered vp = W * block + i'; pair j of (wave wf, lane)
// is vp = 64 (4 (J0 + j) + wf) + lane.  W = 64: pair j holds bins 512 j + 128 wf + lane and + 64, consecutive lanes read consecutive
// floats.  A pair-wave none of whose bins is decoded by this handle is skipped (wave-uniform).
template <int SL, bool IQ, int SPS = 3> struct ChzSlicer {
    static constexpr int NB = CHZ_BATCH, M = CHZ_M, J0 = 0, NP = 2;   // channel pairs of a lane
    ChzSlicePair<SL, SPS> S[NP];
    uint32_t ch[NP][2];                                           // row of the bin (>= n_channels: not decoded by this handle)
    uint32_t pbase[NP];                                           // float offset of the pair's first real part in a planar frame
    bool pair_on[NP];
    uint32_t hold[NP][2][4];                                      // finished ring words waiting for their 16-byte store
    int nheld;
    uint64_t mask32;
    // spec D: the four frames of the half-batch in work and of the one before it, per channel pair (step4_hist); only frames 1..3 of
    // the older buffer are ever read again, so what stays live across a time step is what the members (pr, pi, h2, h3) used to hold
    static constexpr bool PINGPONG = !IQ && SL == AMPS_SLICER_EXACT;
    f2 Yr[PINGPONG ? 2 : 1][NP][NB], Yi[PINGPONG ? 2 : 1][NP][NB];
    __device__ __forceinline__ void clear_frames()
    {
        if constexpr (PINGPONG) {
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int j = 0; j < NP; j++)
#pragma unroll
                    for (int g = 0; g < NB; g++) { Yr[b][j][g] = (f2){ 0.f, 0.f }; Yi[b][j][g] = (f2){ 0.f, 0.f }; }
        }
    }
    __device__ __forceinline__ void init(const ChzArgs &a, int wf, int lane)
    {
        nheld = 0;
        mask32 = 2ull * a.ring_words - 1;
        clear_frames();
#pragma unroll
        for (int j = 0; j < NP; j++) {
            S[j].reset();
            const uint32_t vp = 64u * (4u * (J0 + j) + (uint32_t)wf) + (uint32_t)lane;
            const bool valid = vp < 8u * a.grp_w;
            const uint32_t blk = vp / a.grp_w, ip = vp - blk * a.grp_w, k0 = 128u * blk + a.grp_r * a.grp_w + ip;
            pbase[j] = valid ? 256u * blk + a.grp_r * a.grp_w + ip : 0u;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                ch[j][e] = valid ? (uint32_t)a.bin2row[(k0 + 64u * e) & (M - 1)] : 0xffffu;
#pragma unroll
                for (int k = 0; k < 4; k++) hold[j][e][k] = 0u;
            }
            pair_on[j] = __ballot(ch[j][0] < a.n_channels || ch[j][1] < a.n_channels) != 0;   // wave-uniform
        }
    }
    // the four frames of half-batch hs: slice (or, unfused, store) this lane's bins.
    // KIND 0 = any half-batch (pre-roll, frames before the stream, the range's last one: every rare case behind a dynamic test);
    // KIND 1 / 2 = a STEADY half-batch -- real frames (F >= f0), not the range's last, no stream-start reset -- that does not /
    // does complete a 32-frame word.  Round 4: with the rare cases tested inside the hot loop every one of them was a control-flow
    // merge through which the compiler carried the whole slicer state in a second register set: ~50 v_mov per time step and wave
    // (8 % of the kernel's VALU instructions) for branches taken once in eight steps or twice per launch.  The role's loop now runs
    // seven KIND-1 steps and one KIND-2 step per word, and KIND 0 only at the two ends of a workgroup's range.
    // PAR = parity of the time step (spec D: which of the two frame buffers this half-batch is read into; the other one holds
    // the half-batch before it.  Every half-batch between the first and the last sliced one is sliced, in consecutive time steps.)
    template <int KIND = 0, int PAR = 0>
    __device__ __forceinline__ void half(const ChzArgs &a, const cf2 *buf, int64_t fs, int64_t f0, int64_t f1, int hs)
    {
        const int64_t F = fs + (int64_t)NB * hs;          // first frame of the half-batch (multiple of 4)
        const float *Af = (const float *)(buf + (hs & (CHZ_SLOTS - 1)) * NB * CHZ_FB);
#pragma unroll
        for (int j = 0; j < NP; j++) {
            if (!pair_on[j]) continue;
            if constexpr (PINGPONG) {
                f2 (&yr)[NB] = Yr[PAR][j];
                f2 (&yi)[NB] = Yi[PAR][j];
#pragma unroll
                for (int g = 0; g < NB; g++) {
                    const float *q = Af + pbase[j] + g * CHZ_FBF;
                    yr[g] = (f2){ q[0], q[64] };
                    yi[g] = (f2){ q[128], q[192] };
                }
                const f2 hr[3] = { Yr[PAR ^ 1][j][3], Yr[PAR ^ 1][j][2], Yr[PAR ^ 1][j][1] };
                const f2 hi[3] = { Yi[PAR ^ 1][j][3], Yi[PAR ^ 1][j][2], Yi[PAR ^ 1][j][1] };
                S[j].step4_hist(yr, yi, hr, hi);
                continue;
            }
            f2 yr[NB], yi[NB];
#pragma unroll
            for (int g = 0; g < NB; g++) {
                const float *q = Af + pbase[j] + g * CHZ_FBF;
                yr[g] = (f2){ q[0], q[64] };
                yi[g] = (f2){ q[128], q[192] };
            }
            if constexpr (IQ) {
                // four frames of a bin leave as one 32-byte run of the channel-major block
                const int ng = KIND != 0 ? NB : (int)(f1 - F < (int64_t)NB ? f1 - F : (int64_t)NB);
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    if (ch[j][e] < a.n_channels) {
                        float2 *dstp = a.out + (uint64_t)ch[j][e] * a.ld + F;
                        if (ng == NB) {
#pragma unroll
                            for (int g = 0; g < NB; g += 2)
                                *(float4 *)(dstp + g) = make_float4(yr[g][e], yi[g][e], yr[g + 1][e], yi[g + 1][e]);
                        } else {
#pragma unroll
                            for (int g = 0; g < NB; g++)
                                if (g < ng) dstp[g] = make_float2(yr[g][e], yi[g][e]);
                        }
                    }
                }
            } else {
                S[j].step4(yr, yi);
            }
        }
        if constexpr (!IQ && KIND != 1) {
            if (KIND == 0 && a.stream_start && F < 0) {   // frames before the stream are exactly +0 (the FFT of zeros may hold -0):
                asm volatile("" ::: "memory");            // the state they leave is that of a fresh stream (a real branch, twice per launch)
#pragma unroll
                for (int j = 0; j < NP; j++) S[j].reset();
                clear_frames();
            }
            if constexpr (SL == AMPS_SLICER_EXACT) {
                // the word boundary inside the pre-roll (f0 - 1): latch the previous-word state the first real word needs
                if (KIND == 0 && F < f0 && ((F + NB - 1) & 31) == 31) {
#pragma unroll
                    for (int j = 0; j < NP; j++) { S[j].exact_word(0); S[j].exact_word(1); }
                }
            }
            if (KIND == 2 || (F >= f0 && ((F + NB - 1) & 31) == 31)) {   // 32 real frames collected (f0 is a multiple of 64)
                // A channel's words leave as ONE 16-byte store per 128 frames (aligned group of four ring dwords): single
                // dwords scattered over the channels' ring rows are counted -- and written -- as 32-byte sectors, 8x the 27 MB
                // of slicer bits per GiB of input (round 1: 215 MB of 1.36 GB traffic).  Ranges start and end on 64-frame
                // boundaries, so a run that is not a whole group is exactly two words.
                const uint64_t w = (a.n_done + (uint64_t)(F + NB - 1)) >> 5;   // absolute ring dword of the finished word
                const bool last = KIND == 0 && F + NB >= f1;
#pragma unroll
                for (int j = 0; j < NP; j++)
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        uint32_t word = S[j].word(e);
                        if ((SL == AMPS_SLICER_PRODUCT || SL == AMPS_SLICER_EXACT) && a.stream_start && F + NB - 1 == 31) word |= (1u << SPS) - 1u;   // no partner yet: g = 1
                        hold[j][e][0] = hold[j][e][1]; hold[j][e][1] = hold[j][e][2]; hold[j][e][2] = hold[j][e][3]; hold[j][e][3] = word;
                    }
                nheld++;
                if ((w & 3) == 3 || last) {
#pragma unroll
                    for (int j = 0; j < NP; j++)
#pragma unroll
                        for (int e = 0; e < 2; e++) {
                            if (ch[j][e] < a.n_channels) {
                                uint32_t *row = (uint32_t *)(a.gring + (uint64_t)ch[j][e] * a.ring_words);
                                if (nheld == 4 && (w & 3) == 3) *(uint4 *)(row + ((w - 3) & mask32)) = make_uint4(hold[j][e][0], hold[j][e][1], hold[j][e][2], hold[j][e][3]);
                                else if (nheld == 2) *(uint2 *)(row + ((w - 1) & mask32)) = make_uint2(hold[j][e][2], hold[j][e][3]);
                                else {                                             // not reached (ranges are multiples of 64 frames); constant register indices
#pragma unroll
                                    for (int k = 0; k < 4; k++)
                                        if (k >= 4 - nheld) row[(w - (uint64_t)(3 - k)) & mask32] = hold[j][e][k];
                                }
                            }
                        }
                    nheld = 0;
                }
            }
        }
    }
};

// Timeline hook (builds with -DCHZ_TIMELINE only; scripts/chz_timeline.py): every wave of workgroup 0 accumulates, in registers,
// the s_memtime spent between its phase boundaries over the time steps >= CHZ_TL_FIRST and writes the sums once, at the end
// (no memory traffic inside the loop: stores would count in the fold role's vmcnt window and stall it)
constexpr int CHZ_TL_FIRST = 40;
#ifdef CHZ_TIMELINE
#define CHZ_TL_DECL long long tl_acc[6] = {}, tl_last = 0
#define CHZ_STAMP(step, k) do { if ((step) >= CHZ_TL_FIRST) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); \
        if (tl_last) tl_acc[k] += now_ - tl_last; tl_last = now_; if ((k) == 4) tl_acc[5]++; } } while (0)
#define CHZ_TL_FLUSH do { if (a.tl && blockIdx.x == 0 && lane == 0) for (int k_ = 0; k_ < 6; k_++) a.tl[wave * 8 + k_] = (unsigned long long)tl_acc[k_]; } while (0)
#else
#define CHZ_TL_DECL
#define CHZ_STAMP(step, k) do { } while (0)
#define CHZ_TL_FLUSH do { } while (0)
#endif

// carry_out[k] = virtual sample (consumed - hist + k), k in [0, hist + leftover_new)
__device__ __forceinline__ float2 chz_carry_sample(const float2 *block, const float2 *carry_in, uint32_t carry_len, uint32_t nsamp,
                                                   uint32_t hist, uint32_t consumed, uint32_t k)
{
    const int64_t lead = (int64_t)carry_len - hist;
    const int64_t v = (int64_t)consumed - hist + k;   // launch-relative virtual index
    const int64_t ci = v + hist;
    float2 s = make_float2(0.f, 0.f);
    if (ci >= 0) {
        if (ci < (int64_t)carry_len) s = carry_in[ci];
        else { const int64_t bi = v - lead; if (bi < (int64_t)nsamp) s = block[bi]; }
    }
    return s;
}
// stand-alone form: a push too short to produce a frame only moves the carry
__global__ __launch_bounds__(256) void chz_carry_kernel(const float2 *block, const float2 *carry_in, float2 *carry_out,
                                                         uint32_t carry_len, uint32_t nsamp, uint32_t hist, uint32_t consumed,
                                                         uint32_t out_len)
{
    for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < out_len; k += gridDim.x * 256)
        carry_out[k] = chz_carry_sample(block, carry_in, carry_len, nsamp, hist, consumed, k);
}

template <int P, int MODE, int DEC = CHZ_D>
__global__ __launch_bounds__(768, 3) void chz12_kernel(ChzArgs a)
{
    static_assert(DEC == CHZ_D || DEC == CHZ_D768, "input samples per frame");
    constexpr int M = CHZ_M, D = DEC, NB = CHZ_BATCH;
    constexpr int SPS = 1536 / DEC;                                   // frames per Manchester symbol (20 ksym/s at 30.72 Msps)
    constexpr bool IQ = MODE == CHZ12_IQ;
    constexpr int SL = IQ ? AMPS_SLICER_ATAN_BOXCAR : MODE;
    __shared__ cf2 buf[CHZ_SLOTS * NB * CHZ_FB];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    // Role of a wave.  The hardware arbitrates VALU issue between the waves of a SIMD by priority, then by age: the fold role is
    // pure VALU and would starve the two roles that alternate LDS round trips with short VALU bursts -- their latency chains
    // would then run AFTER the fold instead of beside it.  So the latency-bound roles get the oldest waves and a higher priority
    // (measured, ms per GiB, spec C / A: fold in the oldest waves and no priorities 0.440 / 0.592; priorities alone 0.390 /
    // 0.505; order alone 0.395 / 0.502; both 0.387 / 0.503; round 2's two-role kernel on the same box 0.387 / 0.532).
    const int role = 2 - (wave >> 2);                                   // 0 fold (waves 8..11), 1 pass 2 (4..7), 2 pass 3 + slicer (0..3)
    // (round 6, under the priorities below: the other wave orders of the three roles -- fold | pass 2 | slicer, fold | slicer | pass 2,
    // slicer | fold | pass 2, pass 2 | slicer | fold -- are within 1 % of this one: profiles/r06/prio_ab.txt)
    // Which role runs pass 3.  Behind the cheap slicers (specs B, C: 7 instructions per channel pair and frame) it shares the
    // slicer's waves; spec A's arctangent makes the slicer the longest chain of a time step (46 instructions per pair and
    // frame), so there pass 3 moves to the pass-2 waves (spec A 0.537 -> 0.517 ms, spec C 0.399 -> 0.414 if it moved too).
    // Spec D keeps pass 3 beside its slicer like specs B / C: either placement 0.413-0.421 ms, and its second channel pair sliced by
    // the pass-2 role's waves (which idle half a step) 0.425-0.428 against 0.420-0.428 -- the kernel is bound by VALU throughput, not
    // by one role's chain (profiles/EXPERIMENTS.md, round 4).
    constexpr bool P3_WITH_P2 = !IQ && SL == AMPS_SLICER_ATAN_BOXCAR;   // (handing one of the slicer's two channel pairs to the pass-2 role instead: 0.518 against 0.494)
    const int wf = wave & 3;                                            // frame of a half-batch this wave transforms (roles 1, 2)
    // Pass 3 produces the bins n = i (mod 64) from the points i + 64 r: a handle that decodes one channel group only needs the grp_w
    // residues of its group, so the four frames of a half-batch pack into 4 grp_w lanes: virtual lane v = 64 wf + lane transforms
    // residue i = grp_r grp_w + v % grp_w of frame v / grp_w (grp_w = 64: lane i of wave wf, frame wf, as ever)
    const uint32_t p3_v = 64u * (uint32_t)wf + (uint32_t)lane;
    const bool p3_on = p3_v < 4u * a.grp_w;
    const int p3_f = (int)(p3_v / a.grp_w) & 3, p3_i = (int)(a.grp_r * a.grp_w + p3_v % a.grp_w);
    // Priorities.  Rounds 3-5: pass 3 + slicer 2, pass 2 1, fold 0 (six other triples within the noise at D = 512 under specs A / C).  Round 6,
    // with the fold the longest chain of a step at either decimation (chz_timeline: 2575 of 3445 cycles at D = 768, the pass-2 role idle for
    // 1650): the FOLD ABOVE PASS 2 -- pass 3 + slicer 2, fold 1, pass 2 0 -- is 1.8-3.3 % faster under spec D at D = 768, 1.4-5 % under
    // B / C, 0.8-3.2 % at D = 512 (every triple with pass 2 lowest gains 2-3 %; profiles/r06/prio_ab.txt).  Spec A, whose pass-2 waves also
    // run pass 3, loses 4-8 % by it and keeps the old order, as does the unfused form.
#ifndef CHZ_PRIO_SLICER
    constexpr bool FOLD_OVER_P2 = !IQ && SL != AMPS_SLICER_ATAN_BOXCAR;
    constexpr int CHZ_PRIO_SLICER = 2, CHZ_PRIO_PASS2 = FOLD_OVER_P2 ? 0 : 1, CHZ_PRIO_FOLD = FOLD_OVER_P2 ? 1 : 0;
#endif
    if (role == 2) __builtin_amdgcn_s_setprio(CHZ_PRIO_SLICER); else if (role == 1) __builtin_amdgcn_s_setprio(CHZ_PRIO_PASS2); else __builtin_amdgcn_s_setprio(CHZ_PRIO_FOLD);
    // The next launch's carry (the last L - D + 4 D samples and the leftover) is a ~80 KB copy: every workgroup moves its slice
    // here, a sample per thread of wave 0, instead of a kernel of its own behind this one (4.4 us + a launch gap per push).  Not
    // in the fold waves: their vmcnt windows count their own loads only.
    if (a.carry_out && wave == 0) {
        const uint32_t per = (a.carry_out_len + gridDim.x - 1) / gridDim.x;
        const uint32_t k0 = blockIdx.x * per;
        const uint32_t k1 = k0 + per < a.carry_out_len ? k0 + per : a.carry_out_len;
        for (uint32_t k = k0 + (uint32_t)lane; k < k1; k += 64)
            a.carry_out[k] = chz_carry_sample(a.block, a.carry, a.carry_len, a.nsamp, a.hist, a.consumed, k);
    }
    const int64_t f0 = (int64_t)blockIdx.x * a.frames_per_wg;   // multiple of 64
    if (f0 >= (int64_t)a.nframes) return;
    int64_t f1 = f0 + a.frames_per_wg; if (f1 > (int64_t)a.nframes) f1 = a.nframes;
    // the slicer state of every bin is rebuilt by two pre-roll half-batches; the first may reach behind the carry (zeros): it
    // only primes the delay lines for the second, which is exact (the carry holds L - D + 4 D samples)
    const int64_t fs = IQ ? f0 : f0 - CHZ_PREROLL;
    const int nh = (int)((f1 - fs + NB - 1) / NB);              // half-batches of this workgroup
    CHZ_TL_DECL;
    const int nsteps = nh + 3;                                    // time step i: fold h = i, pass 2 h = i - 1, pass 3 h = i - 2, slicer h = i - 3

    if (role == 0) {
        // ------------------------------------------------------------------ fold role
        const int t = tid & 255;
        const int64_t lead0 = (int64_t)a.carry_len - (int64_t)a.hist;
        const int64_t fl = ((int64_t)a.nsamp - D + lead0) / D;              // floor for the non-negative values the FAST path sees
        const ChzIn in{ a.block, a.carry, (int64_t)a.hist, lead0, (int64_t)a.carry_len, (int64_t)a.nsamp, (uint32_t)(fl < 0 ? 0 : fl) };
        cf2 coef[4][P / 2];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int q = 0; q < P; q += 2) coef[j][q / 2] = (cf2){ a.taps[t + 256 * j + q * M], a.taps[t + 256 * j + (q + 1) * M] };
        cf2 tw1[3];                                               // pass 2's input twiddles of this thread's outputs k1 = 1..3
#pragma unroll
        for (int k1 = 1; k1 < 4; k1++) tw1[k1 - 1] = chz_twiddle((t >> 4) * k1, 64);
        if constexpr (DEC == CHZ_D768) {
        // ---- D = 768 (section "D = 768" above): twelve ring slots per branch, three new samples per branch and half-step
        cf2 ring[4][CHZ768_R];
        {
            const int64_t vend = fs * D;                          // multiple of M (fs is a multiple of four)
#pragma unroll
            for (int jb = 0; jb < 4; jb++) {
                const int64_t vlast = vend - M + (t + 256 * jb);
#pragma unroll
                for (int q = 0; q < P; q++) ring[jb][q] = in.generic_nb(vlast - (int64_t)M * (P - 1 - q));
            }
        }
        chz768_prime(ring, in, fs, t);
        constexpr int PERIOD = 4;                                 // half-steps until the ring is back where it started (three slots per half-step, twelve slots)
        // half-step h loads frames of the half-steps h + 1 (from its third frame on) and h + 2: the fast loader is right once the
        // first frame of half-step h + 1 lies inside the new block
        // (The unfused form -- a checking mode -- runs every half-step as an edge step at this decimation: with its epilogue's row
        // addresses the kernel does not fit 168 registers, and what the compiler chose to spill were ring slots with a load in flight
        // -- it stores the stale value and reloads it behind the wait; tests/test_cpu_inflight_loads.py scans for exactly that.  One
        // drained batch of twelve loads per half-step is a third of the fast loader's speed, and the arithmetic is the same.)
        int h_edge = 0;
        if (IQ || in.nsamp < D) h_edge = nsteps;
        else if ((fs + NB) * D < in.lead) {
            const int64_t need = (in.lead + D - 1) / D - (fs + NB);
            h_edge = (int)((need + NB - 1) / NB);
            h_edge = (h_edge + PERIOD - 1) / PERIOD * PERIOD;
            if (h_edge > nsteps) h_edge = nsteps;
        }
        // the workgroup's descriptor starts at the first sample of its first FAST half-step (a few frames in front of the block, at the
        // most, for the workgroup that takes over from the carry: nothing down there is ever addressed); a range beyond 2^31 bytes -- a
        // 50 GB push -- keeps the bounds-checked loader
        const int64_t base_s = (fs + (int64_t)NB * h_edge) * D - in.lead;
        if ((int64_t)(nsteps - h_edge + 3) * NB * D * (int64_t)sizeof(float2) >= (1ll << 31)) h_edge = nsteps;
        const chz_rsrc_t rsrc = chz_make_rsrc(in.block + base_s, (uint64_t)(in.nsamp - base_s) * sizeof(float2));
        __syncthreads();                                          // all roles start together
        auto half_step = [&](auto basec, auto edgec, int h) {
            constexpr int BASE = decltype(basec)::value;
            constexpr bool EDGE = decltype(edgec)::value;
            CHZ_STAMP(h, 0);
            if (__builtin_expect(h < nh, 1)) {
                const int64_t F = fs + (int64_t)NB * h;
                cf2 *dst = buf + (h & (CHZ_SLOTS - 1)) * NB * CHZ_FB;
                chz768_ring_wait<BASE, 0>(ring);
                CHZ_STAMP(h, 1);
                if constexpr (EDGE) {
                    chz768_fold2_ring<P, BASE, 0>(ring, coef, tw1, dst, t, [] {});
                    chz768_fold2_ring<P, BASE, 2>(ring, coef, tw1, dst, t, [] {});
                    chz768_loads_generic<BASE>(ring, in, F, t);
                } else {
                    const uint32_t so = (uint32_t)(h - h_edge) * (uint32_t)(NB * D * sizeof(float2));   // this half-step inside the workgroup's descriptor
                    chz768_fold2_ring<P, BASE, 0>(ring, coef, tw1, dst, t, [&] { chz768_loads_a<BASE, true>(ring, in, F, t, rsrc, so); });
                    chz768_ring_wait<BASE, 2>(ring);
                    chz768_fold2_ring<P, BASE, 2>(ring, coef, tw1, dst, t, [&] { chz768_loads_b<BASE, true>(ring, in, F, t, rsrc, so); });
                }
                CHZ_STAMP(h, 2);
            }
            CHZ_STAMP(h, 3);
            __syncthreads();
            CHZ_STAMP(h, 4);
        };
        auto run_steps = [&](auto edgec, int hb, int he) __attribute__((always_inline)) {        // half-steps [hb, he); hb is a multiple of the ring's period
            for (int h = hb; h < he; h += PERIOD) {
                half_step(std::integral_constant<int, 0>{}, edgec, h);
                if (h + 1 >= he) break;
                half_step(std::integral_constant<int, 3>{}, edgec, h + 1);
                if (h + 2 >= he) break;
                half_step(std::integral_constant<int, 6>{}, edgec, h + 2);
                if (h + 3 >= he) break;
                half_step(std::integral_constant<int, 9>{}, edgec, h + 3);
            }
        };
        run_steps(std::true_type{}, 0, h_edge);
        if constexpr (!IQ) run_steps(std::false_type{}, h_edge, nsteps);
        } else {
        cf2 ring[4][P + 4];                                       // delay lines + the inputs of this and the next half-step
        {
            const int64_t vend = fs * D;                          // multiple of M (fs is even)
#pragma unroll
            for (int jb = 0; jb < 4; jb++) {
                const int64_t vlast = vend - M + (t + 256 * jb);
#pragma unroll
                for (int q = 0; q < P; q++) ring[jb][q] = in.generic_nb(vlast - (int64_t)M * (P - 1 - q));
            }
        }
        chz_load_half_ring<P, 0, P, false>(ring, in, fs, t);
        chz_load_half_ring<P, 0, P + 2, false>(ring, in, fs + NB, t);
        constexpr int PERIOD = (P + 4) / 2;                       // half-steps until the ring is back where it started (6)
        static_assert(PERIOD == 6, "the unrolled loop below is written for P = 8");
        // half-step h loads the frames of half-step h + 2: the fast loader is right once those lie inside the new block
        int h_edge = 0;
        if (in.nsamp < D) h_edge = nsteps;
        else if ((fs + 2 * NB) * D < in.lead) {
            const int64_t need = (in.lead + D - 1) / D - (fs + 2 * NB);           // frames from the first loaded one to the first inside the block
            h_edge = (int)((need + NB - 1) / NB);
            h_edge = (h_edge + PERIOD - 1) / PERIOD * PERIOD;
            if (h_edge > nsteps) h_edge = nsteps;
        }
        // the workgroup's descriptor (chz_make_rsrc) starts at the first sample of its first FAST half-step; a range beyond 2^31 bytes keeps the bounds-checked loader
        const int64_t base_s = (fs + (int64_t)NB * h_edge) * D - in.lead;
        if ((int64_t)(nsteps - h_edge + 3) * NB * D * (int64_t)sizeof(float2) >= (1ll << 31)) h_edge = nsteps;
        const chz_rsrc_t rsrc = chz_make_rsrc(in.block + base_s, (uint64_t)(in.nsamp - base_s) * sizeof(float2));
        __syncthreads();                                          // all roles start together 
        // one half-step = four frames: fold them, then load the frames of the half-step after next into the two slots that
        // just died.  A load has eight frames (~3 us) to arrive: with four frames of lead the fold waves were the critical path
        // (4 waves x 8 loads x 512 B = 16 KB in flight per CU do not cover the HBM latency under load).
        // One half-step = four frames.  EDGE half-steps (the head of a launch, where the inputs still come from the carry of the
        // previous push, and pushes shorter than a frame) load with ordinary, bounds-checked loads BEHIND the fold and drain them
        // at once; all others prefetch with untracked asm loads (chz_load1_ring) that go into the ring slots as they die, as early
        // in the step as possible: the oldest slot of branches 0, 1 is not read at all in this half-step; the oldest of branches
        // 2, 3 and the second-oldest of branches 0, 1 are last read by the first tap block of frames 0 / 1; the second-oldest of
        // branches 2, 3 by the first tap block of frame 2.  A load then has almost two time steps to land and is issued beside the
        // other roles' VALU work.  The two kinds never meet inside one loop body: a control-flow join behind an untracked load
        // invites the compiler to copy a register whose load is still in flight (it did; tests/test_cpu_inflight_loads.py scans
        // the assembly for that).
        auto half_step = [&](auto basec, auto edgec, int h) {
            constexpr int BASE = decltype(basec)::value;
            constexpr bool EDGE = decltype(edgec)::value;
            CHZ_STAMP(h, 0);
            if (__builtin_expect(h < nh, 1)) {
                const int64_t F = fs + (int64_t)NB * h;
                cf2 *dst = buf + (h & (CHZ_SLOTS - 1)) * NB * CHZ_FB;
                chz_ring_wait<P, BASE>(ring);
                CHZ_STAMP(h, 1);
                if constexpr (EDGE) {
                    chz_fold2_ring<P, BASE, 0>(ring, coef, tw1, dst, t, [] {});
                    chz_fold2_ring<P, BASE, 2>(ring, coef, tw1, dst, t, [] {});
                    chz_load_half_ring<P, BASE, P + 4, false>(ring, in, F + 2 * NB, t);
                } else {
                    const uint32_t so = (uint32_t)(h - h_edge) * (uint32_t)(NB * D * sizeof(float2));   // this half-step inside the workgroup's descriptor
                    chz_load1_ring<P, BASE, P + 4, 0, true>(ring, in, F + 2 * NB, t, rsrc, so);
                    chz_fold2_ring<P, BASE, 0>(ring, coef, tw1, dst, t, [&] {
                        chz_load1_ring<P, BASE, P + 4, 1, true>(ring, in, F + 2 * NB, t, rsrc, so);
                        chz_load1_ring<P, BASE, P + 4, 2, true>(ring, in, F + 2 * NB, t, rsrc, so);
                    });
                    chz_fold2_ring<P, BASE, 2>(ring, coef, tw1, dst, t, [&] { chz_load1_ring<P, BASE, P + 4, 3, true>(ring, in, F + 2 * NB, t, rsrc, so); });
                }
                CHZ_STAMP(h, 2);
            }
            CHZ_STAMP(h, 3);
            __syncthreads();
            CHZ_STAMP(h, 4);
        };
        auto run_steps = [&](auto edgec, int hb, int he) __attribute__((always_inline)) {        // half-steps [hb, he); hb is a multiple of the ring's period
            for (int h = hb; h < he; h += PERIOD) {
                half_step(std::integral_constant<int, 0>{}, edgec, h);
                if (h + 1 >= he) break;
                half_step(std::integral_constant<int, 2>{}, edgec, h + 1);
                if (h + 2 >= he) break;
                half_step(std::integral_constant<int, 4>{}, edgec, h + 2);
                if (h + 3 >= he) break;
                half_step(std::integral_constant<int, 6>{}, edgec, h + 3);
                if (h + 4 >= he) break;
                half_step(std::integral_constant<int, 8>{}, edgec, h + 4);
                if (h + 5 >= he) break;
                half_step(std::integral_constant<int, 10>{}, edgec, h + 5);
            }
        };
        run_steps(std::true_type{}, 0, h_edge);
        run_steps(std::false_type{}, h_edge, nsteps);
        }
        CHZ_TL_FLUSH;
    } else if (role == 1) {
        // ------------------------------------------------------------------ pass-2 role (+ pass 3 when P3_WITH_P2)
        cf2 tw3[P3_WITH_P2 ? 15 : 1];                             // twiddles of the second radix-16 pass: W_1024^{r lane}
        if constexpr (P3_WITH_P2) {
#pragma unroll
            for (int r = 1; r < 16; r++) tw3[r - 1] = chz_twiddle(r * p3_i, 1024);
        }
        __syncthreads();                                          // all roles start together 
        {
            for (int i = 0; i < nh + 3; i++) {
                const int h = i - 1, h3 = i - 2;
                CHZ_STAMP(i, 0);
                if (h >= 0 && h < nh) chz_p2(buf + ((h & (CHZ_SLOTS - 1)) * NB + wf) * CHZ_FB, lane);
                CHZ_STAMP(i, 1);
                if constexpr (P3_WITH_P2) { if (h3 >= 0 && h3 < nh && p3_on) chz_p3(buf + ((h3 & (CHZ_SLOTS - 1)) * NB + p3_f) * CHZ_FB, tw3, p3_i); }
                CHZ_STAMP(i, 3);
                __syncthreads();
                CHZ_STAMP(i, 4);
            }
        }
        CHZ_TL_FLUSH;
    } else {
        // ------------------------------------------------------------------ pass-3 + slicer role
        cf2 tw3[P3_WITH_P2 ? 1 : 15];                             // twiddles of the second radix-16 pass: W_1024^{r lane}
        if constexpr (!P3_WITH_P2) {
#pragma unroll
            for (int r = 1; r < 16; r++) tw3[r - 1] = chz_twiddle(r * p3_i, 1024);
        }
        ChzSlicer<SL, IQ, SPS> slicer;
        slicer.init(a, wf, lane);
        __syncthreads();                                          // all roles start together 
        {
            // time step i slices half-batch hs = i - 3 and transforms h3 = i - 2.  STEADY steps -- 2 <= hs <= nh - 2: real frames, not
            // the range's last half-batch, h3 inside the range -- run without any of the rare-case tests (ChzSlicer::half<1 / 2>);
            // a 32-frame word completes when hs = 1 (mod 8), i.e. in the last step of every group of eight that starts at i = 5
            auto step = [&](auto kindc, auto parc, int i) __attribute__((always_inline)) {
                constexpr int KIND = decltype(kindc)::value, PAR = decltype(parc)::value;   // PAR = i & 1
                const int hs = i - 3, h3 = i - 2;
                CHZ_STAMP(i, 0);
                if (KIND != 0 || (hs >= 0 && hs < nh)) slicer.template half<KIND, PAR>(a, buf, fs, f0, f1, hs);
                CHZ_STAMP(i, 1);
                if constexpr (!P3_WITH_P2) { if ((KIND != 0 || (h3 >= 0 && h3 < nh)) && p3_on) chz_p3(buf + ((h3 & (CHZ_SLOTS - 1)) * NB + p3_f) * CHZ_FB, tw3, p3_i); }
                CHZ_STAMP(i, 3);
                __syncthreads();
                CHZ_STAMP(i, 4);
            };
            constexpr int I_FIRST = IQ ? 3 + 2 : 5;               // first steady step (hs = 2); ODD, and a group is eight steps: the parities below
            static_assert((I_FIRST & 1) == 1, "parity of the steady groups");
            const int i_last = nh + 1;                            // last steady step (hs = nh - 2, h3 = nh - 1)
            using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;
            using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
            // Every step's parity is a compile-time constant, the edge steps' too: a run-time parity at either end would keep BOTH
            // frame buffers of the slicer alive across the whole steady loop (24 VGPRs: measured as spills in the word step, +10 %).
            int i = 0;
            static_assert(I_FIRST == 5, "the five edge steps in front of the steady groups are written out");
            if (i < nsteps) { step(K0{}, P0{}, i); i++; }
            if (i < nsteps) { step(K0{}, P1{}, i); i++; }
            if (i < nsteps) { step(K0{}, P0{}, i); i++; }
            if (i < nsteps) { step(K0{}, P1{}, i); i++; }
            if (i < nsteps) { step(K0{}, P0{}, i); i++; }
            while (i + 7 <= i_last) {
                // seven plain steps and the one that completes a word; two steps per loop round so that the time step's parity -- which
                // of the slicer's two frame buffers is written -- is a compile-time constant (all eight as straight-line code: 15
                // spilled VGPRs)
#pragma unroll 1
                for (int k = 0; k < 6; k += 2) { step(K1{}, P1{}, i + k); step(K1{}, P0{}, i + k + 1); }
                step(K1{}, P1{}, i + 6);
                step(K2{}, P0{}, i + 7);
                i += 8;
            }
            // (i is odd here -- I_FIRST + 8 n -- or the range was shorter than the five edge steps and nothing is left)
            while (i < nsteps) {
                step(K0{}, P1{}, i); i++;
                if (i >= nsteps) break;
                step(K0{}, P0{}, i); i++;
            }
        }
        CHZ_TL_FLUSH;
    }
}

struct ChannelizerState {
    bool enabled = false;
    int P = 8;
    int D = CHZ_D;                  // input samples per frame: 512 (3 samples per symbol) or 768 (2)
    uint32_t C = 0, first_bin = 0;  // rows this handle decodes (= the band's channels, or one group of them), FFT bin of the band's channel 0
    uint32_t groups = 1, group = 0; // cfg.wideband_groups / wideband_group
    uint16_t *bin2row = nullptr;    // device [M]
    std::vector<uint32_t> row2chan; // row -> channel number within the band selection (what the records carry)
    uint32_t max_frames = 0;        // per push
    uint32_t target_wgs = 256;      // resident workgroups of the filter-bank kernel (one 768-thread workgroup per CU)
    float *taps = nullptr;          // [L]
    float2 *carry[2] = { nullptr, nullptr };
    int carry_cur = 0;
    uint32_t carry_len = 0;         // L - D + leftover
    uint64_t frames_done = 0;
    float2 *out = nullptr;          // [C][ld]
    uint64_t ld = 0;
    float2 *stage = nullptr;        // device staging for host-resident wideband input
    size_t stage_samples = 0;
    StageFence stage_fence;
};

inline double bessel_i0(double x)
{
    double s = 1.0, t = 1.0;
    for (int k = 1; k < 64; k++) { t *= (x / (2.0 * k)) * (x / (2.0 * k)); s += t; if (t < 1e-18 * s) break; }
    return s;
}

// Kaiser(beta = 8) windowed sinc at fs = M * 30 kHz, unit DC gain; -6 dB at 13 kHz behind the D = 512 bank, at 15 kHz behind the D = 768 one
// (40 ksps per channel leave the slicer two sampling phases per symbol instead of three: the wider pass band gives back, under a
// carrier offset, what the coarser timing costs -- profiles/r06/decim768_cpu_gonogo.txt; oracle/channelizer.py: cutoff_for_decim)
inline double chz_cutoff_hz(int D) { return D == CHZ_D768 ? 15.0e3 : 13.0e3; }
inline std::vector<float> chz_design_taps(int P, int D = CHZ_D)
{
    const int L = P * CHZ_M;
    const double fc = chz_cutoff_hz(D) / (CHZ_M * 30.0e3);      // cycles per sample
    const double beta = 8.0, i0b = bessel_i0(beta);
    std::vector<double> h(L);
    double sum = 0.0;
    for (int i = 0; i < L; i++) {
        const double m = i - 0.5 * (L - 1);
        const double x = 2.0 * fc * m;
        const double sinc = std::fabs(x) < 1e-12 ? 1.0 : std::sin(M_PI * x) / (M_PI * x);
        const double r = 2.0 * i / (L - 1) - 1.0;
        const double w = bessel_i0(beta * std::sqrt(std::max(0.0, 1.0 - r * r))) / i0b;
        h[i] = 2.0 * fc * sinc * w;
        sum += h[i];
    }
    std::vector<float> out(L);
    for (int i = 0; i < L; i++) out[i] = (float)(h[i] / sum);
    return out;
}

inline uint32_t chz_hist(int P, int D) { return (uint32_t)(P * CHZ_M - D + CHZ_PRE * D); }
inline size_t chz_carry_cap(int P, int D) { return (size_t)chz_hist(P, D) + 64 * D; }   // + leftover (< 64 frames)

inline int channelizer_reset(ChannelizerState &z, hipStream_t s)
{
    if (!z.enabled) return 0;
    const size_t cap = chz_carry_cap(z.P, z.D);
    if (hipMemsetAsync(z.carry[0], 0, sizeof(float2) * cap, s) != hipSuccess) return -EIO;
    if (hipMemsetAsync(z.carry[1], 0, sizeof(float2) * cap, s) != hipSuccess) return -EIO;
    z.carry_cur = 0;
    z.carry_len = chz_hist(z.P, z.D);                // all-zero history, no leftover
    z.frames_done = 0;
    return 0;
}

inline void channelizer_destroy(ChannelizerState &z)
{
    z.stage_fence.destroy();
    void *bufs[] = { z.taps, z.carry[0], z.carry[1], z.out, z.stage, z.bin2row };
    for (void *p : bufs) if (p) (void)hipFree(p);
    z = ChannelizerState();
}

// bins and rows of a handle: row i = the i-th channel (in band order) of the band selection [first, first + n) that belongs to the
// handle's group -- all of them without groups.  Returns the row count, or a negative errno for an invalid split.
inline int chz_rows(const amps_recc_cfg_t &cfg, std::vector<uint16_t> *bin2row, std::vector<uint32_t> *row2chan)
{
    const uint32_t G = cfg.wideband_groups > 1 ? cfg.wideband_groups : 1u;
    if ((G != 1 && G != 2 && G != 4 && G != 8) || cfg.wideband_group >= G) return -EINVAL;
    if (cfg.n_channels > (uint32_t)CHZ_M || cfg.wideband_first_channel >= (uint32_t)CHZ_M) return -EINVAL;
    const uint32_t W = 64u / G;
    if (bin2row) bin2row->assign(CHZ_M, (uint16_t)0xffffu);
    if (row2chan) row2chan->clear();
    uint32_t rows = 0;
    for (uint32_t c = 0; c < cfg.n_channels; c++) {
        const uint32_t k = (cfg.wideband_first_channel + c) & (CHZ_M - 1);
        if ((k & 63u) / W != cfg.wideband_group) continue;
        if (bin2row) (*bin2row)[k] = (uint16_t)rows;
        if (row2chan) row2chan->push_back(c);
        rows++;
    }
    return (int)rows;
}

inline int channelizer_create(ChannelizerState &z, const amps_recc_cfg_t &cfg, hipStream_t s)
{
    if (cfg.wideband_channels != CHZ_M || (cfg.wideband_decim != CHZ_D && cfg.wideband_decim != CHZ_D768)) return -EINVAL;   // M = 1024, D = 512 or 768
    const int P = cfg.wideband_taps_per_branch ? (int)cfg.wideband_taps_per_branch : 8;
    if (P != 8) return -EINVAL;                                   // the register ring of chz12_kernel is laid out for eight taps per branch
    if (cfg.n_channels > CHZ_M || cfg.wideband_first_channel >= CHZ_M || cfg.max_samples_per_push == 0) return -EINVAL;
    std::vector<uint16_t> b2r;
    const int rows = chz_rows(cfg, &b2r, &z.row2chan);
    if (rows < 1) return rows < 0 ? rows : -EINVAL;
    z.P = P; z.D = (int)cfg.wideband_decim; z.C = (uint32_t)rows; z.first_bin = cfg.wideband_first_channel;
    z.groups = cfg.wideband_groups > 1 ? cfg.wideband_groups : 1u; z.group = cfg.wideband_group;
    if (hipMalloc((void **)&z.bin2row, sizeof(uint16_t) * CHZ_M) != hipSuccess) return -ENOMEM;
    if (hipMemcpy(z.bin2row, b2r.data(), sizeof(uint16_t) * CHZ_M, hipMemcpyHostToDevice) != hipSuccess) return -EIO;
    z.max_frames = cfg.max_samples_per_push;
    z.ld = ((uint64_t)z.max_frames + 7) & ~7ull;
    const size_t L = (size_t)P * CHZ_M;
    std::vector<float> h = chz_design_taps(P, z.D);
    if (hipMalloc((void **)&z.taps, sizeof(float) * L) != hipSuccess) return -ENOMEM;
    if (hipMemcpy(z.taps, h.data(), sizeof(float) * L, hipMemcpyHostToDevice) != hipSuccess) return -EIO;
    if (hipMalloc((void **)&z.carry[0], sizeof(float2) * chz_carry_cap(P, z.D)) != hipSuccess) return -ENOMEM;
    if (hipMalloc((void **)&z.carry[1], sizeof(float2) * chz_carry_cap(P, z.D)) != hipSuccess) return -ENOMEM;
    // z.out (the channel-major block, C x ld x 8 B: 1.7 GB for a full band at 2^18 frames per push) is allocated by the first
    // unfused / debug run: the fused form never touches it
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            z.target_wgs = (uint32_t)prop.multiProcessorCount;
    }
    z.enabled = true;
    (void)s;
    return 0;
}

// Channelise `nsamp` new wideband samples.
//  fused = false: writes the channel-major block; *chan_iq / *ld / *nframes describe it (even number of frames).
//  fused = true : runs discriminator + boxcar + slicer behind the FFT and writes only slicer bits into `gring`
//                 at absolute sample index n_done.. ; consumes a multiple of 64 frames.
inline int channelizer_run(ChannelizerState &z, const float2 *iq, size_t nsamp, int mem, hipStream_t s,
                           const float2 **chan_iq, uint64_t *ld, uint32_t *nframes_out,
                           bool fused = false, uint64_t *gring = nullptr, uint32_t ring_words = 0, uint64_t n_done = 0,
                           int slicer = AMPS_SLICER_ATAN_BOXCAR, void (*after_main)(void *) = nullptr, void *after_ctx = nullptr)
{
    if (!z.enabled) return -ENOSYS;
    if (!fused && z.groups > 1) return -ENOSYS;                   // channel groups exist in the fused form only
    const float2 *d = iq;
    if (mem == AMPS_MEM_HOST) {
        if (int rc = z.stage_fence.wait()) return rc;             // the previous push may still be reading the staging buffer
        if (z.stage_samples < nsamp) {
            if (z.stage) (void)hipFree(z.stage);
            z.stage = nullptr; z.stage_samples = 0;
            if (hipMalloc((void **)&z.stage, sizeof(float2) * nsamp) != hipSuccess) return -ENOMEM;
            z.stage_samples = nsamp;
        }
        // synchronous: the caller may reuse its buffer as soon as the push returns (see amps_recc_push_iq)
        if (hipMemcpy(z.stage, iq, sizeof(float2) * nsamp, hipMemcpyHostToDevice) != hipSuccess) return -EIO;
        d = z.stage;
    }
    if (!fused && !z.out && hipMalloc((void **)&z.out, sizeof(float2) * (size_t)z.C * z.ld) != hipSuccess) return -ENOMEM;
    const uint32_t hist = chz_hist(z.P, z.D);
    const uint32_t leftover = z.carry_len - hist;
    const uint64_t avail = (uint64_t)leftover + nsamp;
    // frames consumed: a multiple of four (a launch starts where the stream position is a multiple of M: frame parity 0 at D = 512,
    // frame phase 0 of 4 at D = 768), and in the fused form a multiple of 64 (whole words of the RECC bit ring); the rest waits in the carry
    const uint32_t nframes = (uint32_t)(avail / (uint32_t)z.D) & (fused ? ~63u : z.D == CHZ_D ? ~1u : ~3u);
    if (nframes > z.max_frames) return -E2BIG;
#ifdef CHZ_TIMELINE
    unsigned long long *a_tl_last = nullptr;
#endif
    const uint32_t consumed = nframes * (uint32_t)z.D;                // virtual samples consumed (incl. leftover)
    const uint32_t new_left = (uint32_t)(avail - consumed);
    bool carry_in_kernel = false;
    if (nframes) {
        ChzArgs a{};
        a.block = d; a.carry = z.carry[z.carry_cur]; a.taps = z.taps; a.out = z.out; a.ld = z.ld;
        a.carry_len = z.carry_len; a.nsamp = (uint32_t)nsamp; a.nframes = nframes; a.hist = hist;
        // one resident round of 768-thread workgroups, one per CU; each refills its delay lines and re-runs eight pre-roll
        // frames when fused, so fewer, longer runs are cheaper
        uint32_t fpw = std::max<uint32_t>(64u, (nframes + z.target_wgs - 1) / z.target_wgs);
        fpw = (fpw + 63) / 64 * 64;
        a.frames_per_wg = fpw; a.first_bin = z.first_bin; a.n_channels = z.C;
        a.bin2row = z.bin2row; a.grp_w = 64u / z.groups; a.grp_r = z.group;
        a.odd_start = 0;
        a.gring = gring; a.ring_words = ring_words; a.n_done = n_done;
        a.stream_start = z.frames_done == 0 ? 1u : 0u;
        const dim3 g12((nframes + fpw - 1) / fpw), b12(768);
        carry_in_kernel = g12.x >= 64;                                // a slice of at most ~700 samples per workgroup; smaller grids leave it to the copy kernel
        if (carry_in_kernel) { a.carry_out = z.carry[z.carry_cur ^ 1]; a.consumed = consumed; a.carry_out_len = hist + new_left; }
#ifdef CHZ_TIMELINE
        static unsigned long long *tl_dev = nullptr;
        constexpr size_t TLN = 12 * 8;
        if (!tl_dev && hipMalloc((void **)&tl_dev, TLN * 8) != hipSuccess) return -ENOMEM;
        (void)hipMemsetAsync(tl_dev, 0, TLN * 8, s);
        a.tl = tl_dev;
        a_tl_last = tl_dev;
#endif
        if (z.D == CHZ_D768) {
            if (!fused) hipLaunchKernelGGL((chz12_kernel<8, CHZ12_IQ, CHZ_D768>), g12, b12, 0, s, a);
            else if (slicer == AMPS_SLICER_PRODUCT) hipLaunchKernelGGL((chz12_kernel<8, AMPS_SLICER_PRODUCT, CHZ_D768>), g12, b12, 0, s, a);
            else if (slicer == AMPS_SLICER_SINE) hipLaunchKernelGGL((chz12_kernel<8, AMPS_SLICER_SINE, CHZ_D768>), g12, b12, 0, s, a);
            else if (slicer == AMPS_SLICER_EXACT) hipLaunchKernelGGL((chz12_kernel<8, AMPS_SLICER_EXACT, CHZ_D768>), g12, b12, 0, s, a);
            else hipLaunchKernelGGL((chz12_kernel<8, AMPS_SLICER_ATAN_BOXCAR, CHZ_D768>), g12, b12, 0, s, a);
        } else
        if (!fused) hipLaunchKernelGGL((chz12_kernel<8, CHZ12_IQ>), g12, b12, 0, s, a);
        else if (slicer == AMPS_SLICER_PRODUCT) hipLaunchKernelGGL((chz12_kernel<8, AMPS_SLICER_PRODUCT>), g12, b12, 0, s, a);
        else if (slicer == AMPS_SLICER_SINE) hipLaunchKernelGGL((chz12_kernel<8, AMPS_SLICER_SINE>), g12, b12, 0, s, a);
        else if (slicer == AMPS_SLICER_EXACT) hipLaunchKernelGGL((chz12_kernel<8, AMPS_SLICER_EXACT>), g12, b12, 0, s, a);
        else hipLaunchKernelGGL((chz12_kernel<8, AMPS_SLICER_ATAN_BOXCAR>), g12, b12, 0, s, a);
    }
    if (after_main) after_main(after_ctx);                            // timing: the span ends behind the filter-bank kernel, before the carry copy
#ifdef CHZ_TIMELINE
    if (const char *path = std::getenv("AMPS_RECC_CHZ_TIMELINE")) {       // the last launch's stamps, raw
        std::vector<unsigned long long> tl(12 * 8);
        if (a_tl_last && hipStreamSynchronize(s) == hipSuccess && hipMemcpy(tl.data(), a_tl_last, tl.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE *f = std::fopen(path, "wb")) { std::fwrite(tl.data(), 8, tl.size(), f); std::fclose(f); }
    }
#endif
    if (!carry_in_kernel)
        hipLaunchKernelGGL(chz_carry_kernel, dim3((hist + new_left + 255) / 256), dim3(256), 0, s, d, z.carry[z.carry_cur],
                           z.carry[z.carry_cur ^ 1], z.carry_len, (uint32_t)nsamp, hist, consumed, hist + new_left);
    if (hipGetLastError() != hipSuccess) return -EIO;
    if (mem == AMPS_MEM_HOST) { if (int rc = z.stage_fence.arm(s)) return rc; }
    z.carry_cur ^= 1;
    z.carry_len = hist + new_left;
    z.frames_done += nframes;
    if (chan_iq) *chan_iq = z.out;
    if (ld) *ld = z.ld;
    *nframes_out = nframes;
    return 0;
}

} // namespace amps
