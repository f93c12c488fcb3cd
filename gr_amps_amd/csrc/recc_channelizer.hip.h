// recc_channelizer.hip.h -- polyphase channelizer front end for gfx950: one wideband fc32 stream
// (fs = M * 30 kHz) -> C active 30 kHz channels at fs / D, channel-major, ready for recc_front_kernel.
//
// It stands where the reference wires one freq_xlating_fir_filter_ccc (299 complex taps, decim 2) per
// channel in front of the RECC chain (grc/recctest.grc:889-937, taps :115-155).  Replicating that FIR per
// channel is ~150 flop per input byte; a weighted-overlap-add filter bank does all M
// channels at once for ~28 flop/B:
//     frame m:  n0 = (m+1) D - L,  L = P*M taps
//               u[r] = sum_{i : (n0+i) mod M = r} h[i] x[n0+i]          (fold, "polyphase")
//               Y_k[m] = FFT_M(u)[k] = sum_i h[i] x[n0+i] e^{-j 2 pi k (n0+i)/M}
// i.e. channel k (centre k*fs/M) mixed to DC with an absolute phase reference, low-pass filtered by the
// prototype h and decimated by D = M/2 (2x oversampled: 60 ksps = 3 samples per Manchester symbol at M = 1024).
//
// Mapping to the hardware (one 256-thread workgroup walks a run of consecutive frames):
//   * polyphase delay lines live in REGISTERS: thread t owns the four branches with residue t + 256 jb (mod M).
//     A frame adds D = 512 new samples, read with one coalesced 4 KiB load (prefetched one frame ahead), and the
//     two samples a thread loads are exactly the ones its own branches need -- there is no LDS sample window at
//     all (a first version kept a 64 KiB LDS ring: 80 KiB per workgroup, two workgroups per CU, 32 LDS reads and
//     64 address instructions per thread per frame);
//   * the fold is one v_pk_fma per tap against 4P register-resident coefficients; which coefficient set a branch
//     uses alternates with the frame parity, so the frame loop is unrolled by two parities (by eight, see below);
//   * FFT-1024 = Stockham passes of radix 4, 16, 4, 4 over a BATCH of four frames: pass 1 runs on the registers the
//     fold just produced, the radix-16 pass is done by one wave per frame entirely in registers (in place in LDS),
//     the last pass leaves bins {t, t+256, t+512, t+768} in registers -- so a thread owns the same four channels in
//     every frame.  Three workgroup barriers per four frames.  Twiddles of the radix-4 passes are per-thread
//     constants (12 registers), those of the radix-16 pass a 512-byte LDS table;
//   * eight frames of a thread's four bins are kept in registers and written as 64-byte runs into the
//     channel-major output (dwordx4 stores), which recc_front_kernel then streams at full rate.
// No MFMA: the contraction per channel is 8..16 taps deep and the FFT is a butterfly network.
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
constexpr int CHZ_D = 512;           // input samples per frame (2x oversampled)
constexpr int CHZ_GROUP = 8;         // frames buffered in registers per output store (64-byte runs)

struct ChzArgs {
    const float2 *block;     // new wideband samples of this push
    const float2 *carry;     // samples [f_done*D + D - L, f_done*D + leftover) of the stream so far
    const float *taps;       // [L] prototype
    float2 *out;             // [C][ld] channel-major output of this push
    uint64_t ld;
    uint32_t carry_len;      // L - D + leftover
    uint32_t nsamp;          // new samples
    uint32_t nframes;        // frames produced by this launch
    uint32_t frames_per_wg;  // multiple of CHZ_GROUP
    uint32_t first_bin;      // FFT bin of channel 0
    uint32_t n_channels;
    uint32_t odd_start;      // parity of (absolute frame index of frame 0 of this launch)
    uint32_t hist;           // samples of history the carry holds before v = 0  (L - D + CHZ_PRE * D)
    // fused form: slicer bits go straight to the RECC bit ring
    uint64_t *gring;         // [C][ring_words]
    uint32_t ring_words;
    uint64_t n_done;         // absolute channel-stream sample index of frame 0 (multiple of 64)
    uint32_t stream_start;   // frame 0 of this launch is the first frame of the stream (spec B: its first 3 bits are ones)
};
constexpr int CHZ_PRE = 4;   // frames re-run in front of a workgroup's range to rebuild per-bin demod state (even)

typedef float cf2 __attribute__((ext_vector_type(2)));

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

// History on MI355X (1 GiB of wideband per launch, fused kernel ms): LDS sample ring 1.49 -> register delay lines 1.26 ->
// fused discriminator 1.02 -> packed complex multiply 0.90 -> lock-step 4-way discriminator 0.87 -> two-frame prefetch 0.83
// -> four-frame batches with a radix-16 pass 0.69 -> per-batch branch-free input path 0.67 -> packed boxcar adds 0.64.
// Measured and rejected (kernel ms at the time): baseline 1.24; XOR-swizzled exchange
// buffers 1.36 (33 % of LDS cycles are bank conflicts, but the kernel is latency- not LDS-bound and the index math
// sits on the critical path); padded buffers 2.0 (LDS over 80 KiB -> one workgroup per CU, LDS-ring version);
// 4-frame output groups + recomputed twiddle powers + __launch_bounds__(256,3) 3.9 (168 VGPRs -> spills).

// ---- the frame pipeline: four frames per batch, FFT-1024 = radix 4 x 16 x 4 x 4 (Stockham) ----
// A workgroup is four waves, and a 1024-point frame is 64 lanes x 16 points: with FOUR frames in flight the middle
// of the FFT becomes one in-register radix-16 pass in which wave w owns frame w of the batch outright.  Per batch:
//   fold + pass 1 (radix 4, registers -> A), all four frames            | barrier
//   pass 2 (radix 16): wave w reads frame w of A entirely, writes it back in place   | barrier
//   pass 3 (radix 4): A -> C, all four frames                           | barrier
//   pass 4 (radix 4): C -> registers (bins t, t+256, t+512, t+768 of every frame)
// = 3 barriers per 4 frames.  The first version ran five radix-4 passes per frame with 4 barriers EACH (0.85 ms -> 0.69 with
// this pipeline; with the barriers compiled out, timing only, the current kernel runs 8 % faster, the old one 20 %).
// Frame buffers are padded by one element per 16 (cpad) so that the radix-16 write-back (stride 16 elements between
// lanes) does not land on four banks.
constexpr int CHZ_BATCH = 4;
constexpr int CHZ_FB = CHZ_M + CHZ_M / 16;                     // padded frame buffer, cf2 elements
__host__ __device__ constexpr int cpad(int n) { return n + (n >> 4); }

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

// One frame: fold x[jb] = sum_q h[t + 256 j + qM] * win[jb][q], j = jb ^ (2 * ((m+1) & 1)), then the radix-4 pass 1 ->
// A[4t .. 4t+3].  The tap window of branch jb is ext[jb][S .. S+P): ext = {delay line, the batch's new samples} and S (SA for
// branches 0,1; SB for 2,3) counts the samples of this batch already shifted in.  PAR = parity of the absolute frame index
// m: which coefficient set a branch uses alternates with it.  All register indices are compile-time constants.
template <int P, int PAR, int SA, int SB>
__device__ __forceinline__ void chz_fold_p1(const cf2 (&ext)[4][P + 2], const cf2 (&coef)[4][P / 2], cf2 *A, int t)
{
    constexpr int SW = 2 * ((PAR + 1) & 1);
    cf2 x[4];
#pragma unroll
    for (int jb = 0; jb < 4; jb++) {
        const int sh = jb < 2 ? SA : SB;
        cf2 s = { 0.f, 0.f };
#pragma unroll
        for (int q = 0; q < P; q += 2) {                            // taps q, q+1 share one coefficient pair
            s = fma_lo(ext[jb][sh + q], coef[jb ^ SW][q / 2], s);
            s = fma_hi(ext[jb][sh + q + 1], coef[jb ^ SW][q / 2], s);
        }
        x[jb] = s;
    }
    cf2 o[4];
    dft4(x[0], x[1], x[2], x[3], o);                            // radix 4, p = 1: no twiddles
    cf2 *d = A + cpad(4 * t);                                   // 4t .. 4t+3 share one 16-group
    d[0] = o[0]; d[1] = o[1]; d[2] = o[2]; d[3] = o[3];
}

// The four frames of a batch.  Frame g brings two new samples per thread, for branches {0,1} (g even) or {2,3} (g odd).
// The four tap windows are views of {delay line, new samples} at compile-time offsets, so within the batch nothing moves;
// the delay lines are shifted once per batch, by two samples per branch (shifting per frame cost 150 v_mov per batch,
// a tenth of the instruction stream).
template <int P>
__device__ __forceinline__ void chz_fold_batch(cf2 (&line)[4][P], const cf2 (&coef)[4][P / 2], const cf2 (&nx)[CHZ_BATCH][2], cf2 *bufA, int t)
{
    cf2 ext[4][P + 2];
#pragma unroll
    for (int jb = 0; jb < 4; jb++) {
#pragma unroll
        for (int q = 0; q < P; q++) ext[jb][q] = line[jb][q];
        ext[jb][P] = nx[jb >> 1][jb & 1];                       // frames 0 / 1 feed branches {0,1} / {2,3}
        ext[jb][P + 1] = nx[2 + (jb >> 1)][jb & 1];             // frames 2 / 3
    }
    chz_fold_p1<P, 0, 1, 0>(ext, coef, bufA, t);
    chz_fold_p1<P, 1, 1, 1>(ext, coef, bufA + CHZ_FB, t);
    chz_fold_p1<P, 0, 2, 1>(ext, coef, bufA + 2 * CHZ_FB, t);
    chz_fold_p1<P, 1, 2, 2>(ext, coef, bufA + 3 * CHZ_FB, t);
#pragma unroll
    for (int jb = 0; jb < 4; jb++)
#pragma unroll
        for (int q = 0; q < P; q++) line[jb][q] = ext[jb][q + 2];
}

// LDS layouts of the 12-wave kernel's frame buffers (cf2 elements).  cpad's one pad element per 16 makes the 8-byte READS of 32
// consecutive lanes span 33 elements, so lane 31 lands on lane 0's banks (one extra LDS cycle on every read of passes 2 and 3
// and of the slicer: 17 % of the LDS cycles were bank conflicts); here every access of the pipeline is conflict free:
//   chz_pos1: pass-1 output, element 4 t + k1 at t + 260 k1 -- the fold's stores are stride 1 across lanes, and pass 2's gather
//             (the compiler pairs its reads into ds_read2_b64: 16-lane groups, 16 bank pairs) hits (i >> 2) + 4 (i & 3) mod 16:
//             with cpad, or with 264 k1, every such read was a 2-way conflict (PMC: exactly 8 conflict cycles per ds_read2);
//   chz_pos2: pass-2 and pass-3 output, n + 4 (n >> 6): no padding inside 64 consecutive elements (reads i + 64 r of
//             consecutive lanes are consecutive), and the stride-64 write-back of pass 2 advances 8 banks per four lanes.
__host__ __device__ constexpr int chz_pos1(int t, int k1) { return t + 260 * k1; }
__host__ __device__ constexpr int chz_pos2(int n) { return n + 4 * (n >> 6); }
static_assert(chz_pos1(255, 3) < CHZ_FB && chz_pos2(1023) < CHZ_FB, "frame buffer too small for the 12-wave layouts");

// pass 2, radix 16, p = 4, one frame per wave, in place.  lane i: k = i & 3, u[r] = A[i + 64 r] e^{-2 pi i r k / 64},
// X = DFT16(u), A[16 (i - k) + k + 4 r] = X[r].  tab[r][k] holds the twiddles (LDS, 512 B).
template <bool REGS>
__device__ __forceinline__ void chz_p2_t(cf2 *A, const cf2 *tab, const cf2 (&twr)[15], int lane)
{
    const int k = lane & 3;
    cf2 u[16];
    if constexpr (REGS) {
        // 12-wave kernel: pass 1 left element 4 t + k1 at chz_pos1(t, k1): lane i wants 4 t + k1 = i + 64 r, i.e.
        // t = (i >> 2) + 16 r, k1 = i & 3
        const cf2 *src = A + chz_pos1(lane >> 2, k);
#pragma unroll
        for (int r = 0; r < 16; r++) u[r] = src[16 * r];
    } else {
        const cf2 *src = A + cpad(lane);                        // cpad(lane + 64 r) = cpad(lane) + 68 r
#pragma unroll
        for (int r = 0; r < 16; r++) u[r] = src[68 * r];
    }
    if constexpr (REGS) {
        // 12-wave kernel: the fold waves have already applied the input twiddles W_64^{r k} (chz_fold2_ring).  (Read from the
        // LDS table here, one pair at a time between the multiplies, they cost eight exposed LDS latencies per pass: measured
        // with s_memtime, pass 2 took 2950 cycles per frame against 1780 for pass 3.)
        (void)twr;
    } else {
        // The 15 twiddles are loop invariant; hoisted out of the batch loop they would pin 30 VGPRs the 4-wave kernel does
        // not have.  The empty asm hides the invariance of the index (laundering the POINTER instead turns the reads
        // into flat loads); re-reading 120 B of LDS per batch is free.
        int kk = k;
        asm volatile("" : "+v"(kk));
#pragma unroll
        for (int r = 1; r < 16; r++) u[r] = cmul(u[r], tab[4 * r + kk]);
    }
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
    if constexpr (REGS) {
        cf2 *dst = A + 17 * (lane - k) + k;                     // chz_pos2(16 (i-k) + k + 4 r) = 17 (i-k) + k + 4 r   (k + 4 r < 64)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            cf2 X[4];
            dft4(v[0][c], v[1][c], v[2][c], v[3][c], X);
#pragma unroll
            for (int d = 0; d < 4; d++) dst[4 * (c + 4 * d)] = X[d];
        }
    } else {
    cf2 *dst = A + 17 * (lane - k) + k;                         // cpad(16 (i-k) + k + 4 r) = 17 (i-k) + k + 4 r + (r >> 2)
#pragma unroll
    for (int c = 0; c < 4; c++) {
        cf2 X[4];
        dft4(v[0][c], v[1][c], v[2][c], v[3][c], X);
#pragma unroll
        for (int d = 0; d < 4; d++) dst[4 * (c + 4 * d) + d] = X[d];   // r = c + 4 d, (r >> 2) = d
    }
    }
}

__device__ __forceinline__ void chz_p2(cf2 *A, const cf2 *tab, int lane)
{
    const cf2 none[15] = {};
    chz_p2_t<false>(A, tab, none, lane);
}

// pass 3, radix 4, p = 64: u[r] = B[t + 256 r] e^{-2 pi i r k / 256}, k = t & 63; C[4 (t - k) + k + 64 r] = X[r]
// The four frames of a batch are independent: all sixteen LDS reads are issued before the first butterfly (the compiler
// otherwise keeps read -> wait -> butterfly per frame, four exposed LDS latencies per pass).
__device__ __forceinline__ void chz_p3_batch(const cf2 *A, cf2 *Cb, const cf2 (&tw)[3], int t)
{
    const cf2 *src = A + cpad(t);                               // cpad(t + 256 r) = cpad(t) + 272 r
    cf2 u[CHZ_BATCH][4];
#pragma unroll
    for (int g = 0; g < CHZ_BATCH; g++)
#pragma unroll
        for (int r = 0; r < 4; r++) u[g][r] = src[g * CHZ_FB + 272 * r];
    const int k = t & 63;
    cf2 *d = Cb + cpad(4 * (t - k) + k);                        // cpad(j + 64 r) = cpad(j) + 68 r
#pragma unroll
    for (int g = 0; g < CHZ_BATCH; g++) {
        cf2 o[4];
        dft4(u[g][0], cmul(u[g][1], tw[0]), cmul(u[g][2], tw[1]), cmul(u[g][3], tw[2]), o);
        d[g * CHZ_FB] = o[0]; d[g * CHZ_FB + 68] = o[1]; d[g * CHZ_FB + 136] = o[2]; d[g * CHZ_FB + 204] = o[3];
    }
}

// pass 4, radix 4, p = 256: bins t + 256 r of the frame; `u` = the frame's four inputs (read by chz_p4_load)
__device__ __forceinline__ void chz_p4_load(const cf2 *Cb, int t, cf2 (&u)[CHZ_BATCH][4])
{
    const cf2 *src = Cb + cpad(t);
#pragma unroll
    for (int g = 0; g < CHZ_BATCH; g++)
#pragma unroll
        for (int r = 0; r < 4; r++) u[g][r] = src[g * CHZ_FB + 272 * r];
}
__device__ __forceinline__ void chz_p4(const cf2 (&u)[4], const cf2 (&tw)[3], cf2 (&y)[4])
{
    dft4(u[0], cmul(u[1], tw[0]), cmul(u[2], tw[1]), cmul(u[3], tw[2]), y);
}

// per-thread constants of the pipeline
template <int P> struct ChzRegs {
    cf2 coef[4][P / 2];      // (h[t + 256 j + qM], h[t + 256 j + (q+1)M]), q even
    cf2 tw3[3], tw4[3];      // pass 3 / pass 4 twiddles
};
template <int P>
__device__ __forceinline__ void chz_setup(ChzRegs<P> &R, const float *taps, cf2 *tab, int t)
{
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int q = 0; q < P; q += 2) R.coef[j][q / 2] = (cf2){ taps[t + 256 * j + q * CHZ_M], taps[t + 256 * j + (q + 1) * CHZ_M] };
#pragma unroll
    for (int r = 1; r < 4; r++) { R.tw3[r - 1] = chz_twiddle(r * (t & 63), 256); R.tw4[r - 1] = chz_twiddle(r * t, 1024); }
    if (t < 64) tab[t] = chz_twiddle((t >> 2) * (t & 3), 64);   // tab[4 r + k]
}

// Input samples of launch-relative frame F for this thread: virtual indices F*D + t and F*D + 256 + t.  A batch that
// lies inside the new block (all but the first and last of a launch) is plain coalesced loads; the generic path walks
// carry / block / zero padding.  The choice is made once per batch, so the common path has no branch between the four
// folds (per-frame branches cost 8 %: they fence the scheduler).
struct ChzIn {
    const float2 *block, *carry;
    int64_t hist, lead, carry_len, nsamp;
    __device__ __forceinline__ cf2 generic(int64_t v) const
    {
        const int64_t ci = v + hist;
        if (ci < 0) return (cf2){ 0.f, 0.f };
        float2 s;
        if (ci < carry_len) s = carry[ci];
        else { const int64_t bi = v - lead; if (bi >= nsamp) return (cf2){ 0.f, 0.f }; s = block[bi]; }
        return (cf2){ s.x, s.y };
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

template <int P>
__global__ __launch_bounds__(256) void chz_pfb_fft_kernel(ChzArgs a)
{
    constexpr int M = CHZ_M, D = CHZ_D;
    __shared__ cf2 bufA[CHZ_BATCH * CHZ_FB];
    __shared__ cf2 bufC[CHZ_BATCH * CHZ_FB];
    __shared__ cf2 tab[64];
    const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
    const uint32_t f0 = blockIdx.x * a.frames_per_wg;          // first frame of this workgroup (launch-relative, multiple of 8)
    if (f0 >= a.nframes) return;
    uint32_t f1 = f0 + a.frames_per_wg; if (f1 > a.nframes) f1 = a.nframes;

    // virtual input stream of this launch: index v in [-(L-D), nsamp + leftover): carry then block.
    // frame f (launch-relative) consumes v in [f*D - (L-D), f*D + D); the absolute frame index of f = 0 is even
    // (the host only ever consumes an even number of frames), so parity(m) = parity(f) and residue(v) = v mod M.
    const ChzIn in{ a.block, a.carry, (int64_t)a.hist, (int64_t)a.carry_len - (int64_t)a.hist, (int64_t)a.carry_len, (int64_t)a.nsamp };
    ChzRegs<P> R;
    chz_setup<P>(R, a.taps, tab, t);
    // delay lines: branch jb (residue r = t + 256 jb) holds its P most recent samples before frame f0:
    // v_last = largest v < f0*D with v mod M == r   (f0*D is a multiple of M because f0 is even)
    cf2 line[4][P];
    {
        const int64_t vend = (int64_t)f0 * D;
#pragma unroll
        for (int jb = 0; jb < 4; jb++) {
            const int64_t vlast = vend - M + (t + 256 * jb);
#pragma unroll
            for (int q = 0; q < P; q++) line[jb][q] = in.generic(vlast - (int64_t)M * (P - 1 - q));
        }
    }
    __syncthreads();                                             // tab

    // eight frames (two batches) of this thread's four bins stay in registers and leave as 64-byte runs of the
    // channel-major output.  Frames past f1 (a partial last group) run on zero padding and are not stored.
    for (uint32_t fg = f0; fg < f1; fg += CHZ_GROUP) {
        cf2 acc[CHZ_GROUP][4];
        const int ng = (int)(f1 - fg < (uint32_t)CHZ_GROUP ? f1 - fg : (uint32_t)CHZ_GROUP);
#pragma unroll
        for (int hb = 0; hb < CHZ_GROUP; hb += CHZ_BATCH) {
            auto fold4 = [&](auto fastc) {
                cf2 nx[CHZ_BATCH][2];
#pragma unroll
                for (int g = 0; g < CHZ_BATCH; g++) in.template frame<decltype(fastc)::value>((int64_t)fg + hb + g, t, nx[g][0], nx[g][1]);
                chz_fold_batch<P>(line, R.coef, nx, bufA, t);
            };
            if (in.batch_in_block((int64_t)fg + hb)) fold4(std::true_type{}); else fold4(std::false_type{});
            __syncthreads();
            chz_p2(bufA + wv * CHZ_FB, tab, lane);
            __syncthreads();
            chz_p3_batch(bufA, bufC, R.tw3, t);
            __syncthreads();
            {
                cf2 u4[CHZ_BATCH][4];
                chz_p4_load(bufC, t, u4);
#pragma unroll
                for (int g = 0; g < CHZ_BATCH; g++) chz_p4(u4[g], R.tw4, acc[hb + g]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t ch = ((uint32_t)(t + 256 * j) - a.first_bin) & (M - 1);
            if (ch < a.n_channels) {
                float2 *dstp = a.out + (uint64_t)ch * a.ld + fg;
                if (ng == CHZ_GROUP) {
#pragma unroll
                    for (int e = 0; e < CHZ_GROUP; e += 2)
                        *(float4 *)(dstp + e) = make_float4(acc[e][j].x, acc[e][j].y, acc[e + 1][j].x, acc[e + 1][j].y);
                } else {
#pragma unroll
                    for (int e = 0; e < CHZ_GROUP; e++)
                        if (e < ng) dstp[e] = make_float2(acc[e][j].x, acc[e][j].y);
                }
            }
        }
    }
}

// ---- fused form: channelizer + FM discriminator + boxcar + slicer; only 1 bit per channel sample reaches HBM ----
// After the FFT a lane owns bins {t, t+256, t+512, t+768} in every frame, so the per-channel stream state of the
// RECC front end is four small register sets: previous frame's value, the last two demod floats (boxcar over
// 3 samples per symbol, ordered aligned-pair sums of include/amps_recc_numerics.h) and a 32-bit slicer shift
// register that is stored to the channel's bit ring every 32 frames.  The arithmetic is the same as
// recc_front_kernel's on the channel-major intermediate, so both forms produce identical bits.
template <int PAR, bool FOUR = true>
__device__ __forceinline__ void chz_bins(const cf2 (&y)[4], cf2 (&prev)[4], f2 (&d1)[2], f2 (&d2)[2], uint32_t (&gw)[4])
{
    f2 d[2];
    if constexpr (FOUR) fm_phase_four(y, prev, d[0], d[1]);       // two chains in lock step (same bits, fewer stalls, more registers)
    else { d[0] = fm_phase_two(y[0], prev[0], y[1], prev[1]); d[1] = fm_phase_two(y[2], prev[2], y[3], prev[3]); }
#pragma unroll
    for (int h = 0; h < 2; h++) {                                 // bins (0,1) and (2,3): the boxcar adds are packed per pair
        // window [n-2, n]: n even -> (d[n-2] + d[n-1]) + d[n];  n odd -> d[n-2] + (d[n-1] + d[n])
        const f2 s = PAR == 0 ? (d2[h] + d1[h]) + d[h] : d2[h] + (d1[h] + d[h]);
        // gw = (gw >> 1) | (s >= 0) << 31 as one v_alignbit per bin
        gw[2 * h] = __builtin_amdgcn_alignbit(s.x >= 0.0f ? 1u : 0u, gw[2 * h], 1);
        gw[2 * h + 1] = __builtin_amdgcn_alignbit(s.y >= 0.0f ? 1u : 0u, gw[2 * h + 1], 1);
        d2[h] = d1[h]; d1[h] = d[h];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) prev[j] = y[j];
}

// Slicer spec B (include/amps_recc_numerics.h): g = !signbit(yi * pr - yr * pi) with p = the same bin three frames (one
// Manchester symbol) earlier.  One v_pk_mul (the partner's halves swapped by op_sel), one v_sub and one v_alignbit per bin:
// the register collects SIGN bits, newest at bit 0, and is bit-reversed and inverted when it is stored.
__device__ __forceinline__ void chz_slice_prod(const cf2 (&y)[4], const cf2 (&p)[4], uint32_t (&gw)[4])
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        cf2 m;                                                    // (yr * pi, yi * pr) = (b, a): the partner's halves swapped by op_sel
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(m) : "v"(y[j]), "v"(p[j]));
        const float sdiff = m.y - m.x;
        gw[j] = __builtin_amdgcn_alignbit(gw[j], __float_as_uint(sdiff), 31);   // (gw << 1) | signbit
    }
}

// Slicer spec C: d' = Im(y conj(prev)) = fmaf(yi, pr, -(yr * pi)) (the `im` of spec A's conj-product, no arctangent), spec A's
// 3-sample boxcar in its aligned-pair order, g = !signbit(S').  Sign bits are collected like spec B's.
template <int PAR>
__device__ __forceinline__ void chz_bins_sine(const cf2 (&y)[4], cf2 (&prev)[4], f2 (&d1)[2], f2 (&d2)[2], uint32_t (&gw)[4])
{
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const cf2 y0 = y[2 * h], p0 = prev[2 * h], y1 = y[2 * h + 1], p1 = prev[2 * h + 1];
        const f2 d = { __builtin_fmaf(y0.y, p0.x, -(y0.x * p0.y)), __builtin_fmaf(y1.y, p1.x, -(y1.x * p1.y)) };
        const f2 s = PAR == 0 ? (d2[h] + d1[h]) + d : d2[h] + (d1[h] + d);
        gw[2 * h] = __builtin_amdgcn_alignbit(gw[2 * h], __float_as_uint(s.x), 31);
        gw[2 * h + 1] = __builtin_amdgcn_alignbit(gw[2 * h + 1], __float_as_uint(s.y), 31);
        d2[h] = d1[h]; d1[h] = d;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) prev[j] = y[j];
}

// P = 8 is held to 256 VGPRs = two waves per SIMD (left alone the allocator takes 258 and halves the occupancy;
// __launch_bounds__(256, 3) would force 168 and spill: 2.9 ms instead of 1.0).  P = 16 needs ~390: one wave per SIMD.
template <int P, int SL = AMPS_SLICER_ATAN_BOXCAR>
__global__ __launch_bounds__(256, P <= 8 ? 2 : 1) void chz_fused_kernel(ChzArgs a)
{
    constexpr int M = CHZ_M, D = CHZ_D;
    __shared__ cf2 bufA[CHZ_BATCH * CHZ_FB];
    __shared__ cf2 bufC[CHZ_BATCH * CHZ_FB];
    __shared__ cf2 tab[64];
    const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
    const int64_t f0 = (int64_t)blockIdx.x * a.frames_per_wg;   // multiple of 64
    if (f0 >= (int64_t)a.nframes) return;
    int64_t f1 = f0 + a.frames_per_wg; if (f1 > (int64_t)a.nframes) f1 = a.nframes;
    const int64_t fs = f0 - CHZ_PRE;                             // pre-roll (one batch): rebuild prev / d1 / d2 of every bin

    const ChzIn in{ a.block, a.carry, (int64_t)a.hist, (int64_t)a.carry_len - (int64_t)a.hist, (int64_t)a.carry_len, (int64_t)a.nsamp };
    ChzRegs<P> R;
    chz_setup<P>(R, a.taps, tab, t);
    cf2 line[4][P];
    {
        const int64_t vend = fs * D;                              // multiple of M (fs is even)
#pragma unroll
        for (int jb = 0; jb < 4; jb++) {
            const int64_t vlast = vend - M + (t + 256 * jb);
#pragma unroll
            for (int q = 0; q < P; q++) line[jb][q] = in.generic(vlast - (int64_t)M * (P - 1 - q));
        }
    }
    cf2 prev[4] = {};
    f2 d1[2] = {}, d2[2] = {};                                    // the last two demod floats of bins (0,1) and (2,3)
    cf2 hist[3][4] = {};                                          // spec B: this lane's bins in frames 1..3 of the previous batch
    uint32_t gw[4];
#pragma unroll
    for (int j = 0; j < 4; j++) gw[j] = SL != AMPS_SLICER_ATAN_BOXCAR ? 0u : ~0u;   // "ones before the stream" in either representation
    const uint64_t mask32 = 2ull * a.ring_words - 1;

    // the inputs of a whole batch are loaded one batch (~7 us) ahead
    cf2 nx[CHZ_BATCH][2];
#pragma unroll
    for (int g = 0; g < CHZ_BATCH; g++) in.template frame<false>(fs + g, t, nx[g][0], nx[g][1]);
    __syncthreads();                                             // tab

    for (int64_t f = fs; f < f1; f += CHZ_BATCH) {               // fs and f1 are multiples of 4
        auto fold4 = [&](auto fastc) {                           // fold this batch, then load the next one (a batch ahead)
            chz_fold_batch<P>(line, R.coef, nx, bufA, t);
#pragma unroll
            for (int g = 0; g < CHZ_BATCH; g++)
                in.template frame<decltype(fastc)::value>(f + CHZ_BATCH + g, t, nx[g][0], nx[g][1]);   // generic: zero beyond the data
        };
        if (in.batch_in_block(f + CHZ_BATCH)) fold4(std::true_type{}); else fold4(std::false_type{});
        __syncthreads();
        chz_p2(bufA + wv * CHZ_FB, tab, lane);
        __syncthreads();
        chz_p3_batch(bufA, bufC, R.tw3, t);
        __syncthreads();
        cf2 u4[CHZ_BATCH][4];
        chz_p4_load(bufC, t, u4);
        if constexpr (SL == AMPS_SLICER_PRODUCT) {
            // frame g pairs with frame g - 3: frames 0..2 with frames 1..3 of the previous batch, frame 3 with frame 0
            cf2 y0[4];
            chz_p4(u4[0], R.tw4, y0);
            chz_slice_prod(y0, hist[0], gw);
            chz_p4(u4[1], R.tw4, hist[0]);                        // frame 1 takes the place of the value it replaces
            chz_slice_prod(hist[0], hist[1], gw);
            chz_p4(u4[2], R.tw4, hist[1]);
            chz_slice_prod(hist[1], hist[2], gw);
            chz_p4(u4[3], R.tw4, hist[2]);
            chz_slice_prod(hist[2], y0, gw);
        } else if constexpr (SL == AMPS_SLICER_SINE) {
#pragma unroll
            for (int g = 0; g < CHZ_BATCH; g++) {
                cf2 y[4];
                chz_p4(u4[g], R.tw4, y);
                if (g & 1) chz_bins_sine<1>(y, prev, d1, d2, gw); else chz_bins_sine<0>(y, prev, d1, d2, gw);
            }
            if (a.stream_start && f < 0) {                        // frames before the stream are exactly +0 (the FFT of zeros may hold -0):
                asm volatile("" ::: "memory");                    // the state they leave is all zeros (a real branch, once per launch)
#pragma unroll
                for (int j = 0; j < 4; j++) { prev[j] = (cf2){ 0.f, 0.f }; gw[j] = 0u; }
                d1[0] = d1[1] = d2[0] = d2[1] = (f2){ 0.f, 0.f };
            }
        } else {
#pragma unroll
        for (int g = 0; g < CHZ_BATCH; g++) {
            cf2 y[4];
            chz_p4(u4[g], R.tw4, y);
            if (g & 1) chz_bins<1>(y, prev, d1, d2, gw); else chz_bins<0>(y, prev, d1, d2, gw);
        }
        }
        if (f >= f0 && ((f + 3) & 31) == 31) {                    // 32 real frames collected (f0 is a multiple of 64)
            const uint64_t n = a.n_done + (uint64_t)(f + 3);      // absolute index of the newest bit
#pragma unroll
            for (int j = 0; j < 4; j++) {                         // channel / ring address recomputed here: 12 fewer live VGPRs
                const uint32_t ch = ((uint32_t)(t + 256 * j) - a.first_bin) & (M - 1);
                uint32_t word = SL != AMPS_SLICER_ATAN_BOXCAR ? ~__builtin_bitreverse32(gw[j]) : gw[j];
                if (SL == AMPS_SLICER_PRODUCT && a.stream_start && f + 3 == 31) word |= 7u;   // no partner yet: g = 1
                if (ch < a.n_channels) ((uint32_t *)(a.gring + (uint64_t)ch * a.ring_words))[(n >> 5) & mask32] = word;
            }
        }
    }
}

// The delay lines as a register RING (12-wave kernel): branch jb keeps P + 4 slots; at a half-step with rotation BASE the
// logical element i of the old {delay line, new samples} view is ring[jb][(BASE + i) % (P + 4)]: elements 0..P-1 the delay
// line, P and P+1 the samples of this half-step's four frames, P+2 and P+3 those of the NEXT half-step (in flight).  After
// the fold the two oldest slots are dead and receive the loads of the half-step after next, and BASE advances by two: no
// register ever moves (the 4-wave kernels shift 32 register pairs per four frames), and a load has two half-steps to land.
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
template <int P, int BASE, int FA>
__device__ __forceinline__ void chz_fold2_ring(const cf2 (&ring)[4][P + 4], const cf2 (&coef)[4][P / 2], const cf2 (&tw1)[3], cf2 *bufA, int t)
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
        if (q == 0) chz_fold_taps2<true>(acc, x0, x1, c); else chz_fold_taps2<false>(acc, x0, x1, c);
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
template <int P, int BASE>
__device__ __forceinline__ void chz_fold_half_ring(const cf2 (&ring)[4][P + 4], const cf2 (&coef)[4][P / 2], const cf2 (&tw1)[3], cf2 *bufA, int t)
{
    chz_fold2_ring<P, BASE, 0>(ring, coef, tw1, bufA, t);
    chz_fold2_ring<P, BASE, 2>(ring, coef, tw1, bufA, t);
}
// the two samples frame F + g brings for this thread go to branches 2 (g & 1) + {0, 1}, logical element ELEM + (g >> 1).
// FAST (the four frames lie inside the new block): the eight loads are issued as inline asm, so that the compiler does not
// track them -- its own bookkeeping puts `s_waitcnt vmcnt(0)` behind every barrier of the loop (the fast / generic join
// makes it conservative), which drains the loads issued a moment ago and halves the lead a load has.  The fold waits with
// chz_ring_wait instead: vmcnt(8) = "everything but the eight youngest loads", i.e. exactly the loads of the previous
// half-step stay in flight.  The generic path (carry, zero padding: first and last half-steps of a launch) uses ordinary
// loads and drains them before it returns, so vmcnt(8) is right after either path.
template <int P, int BASE, int ELEM, bool FAST>
__device__ __forceinline__ void chz_load_half_ring(cf2 (&ring)[4][P + 4], const ChzIn &in, int64_t F, int t)
{
    constexpr int R = P + 4;
    if constexpr (FAST) {
        const float2 *p = in.block + (F * CHZ_D - in.lead) + t;
#pragma unroll
        for (int g = 0; g < CHZ_BATCH; g++) {
            const float2 *q = p + g * CHZ_D;
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(ring[2 * (g & 1)][(BASE + ELEM + (g >> 1)) % R]) : "v"(q));
            asm volatile("global_load_dwordx2 %0, %1, off offset:2048" : "=v"(ring[2 * (g & 1) + 1][(BASE + ELEM + (g >> 1)) % R]) : "v"(q));
        }
    } else {
#pragma unroll
        for (int g = 0; g < CHZ_BATCH; g++)
            in.template frame<false>(F + g, t, ring[2 * (g & 1)][(BASE + ELEM + (g >> 1)) % R], ring[2 * (g & 1) + 1][(BASE + ELEM + (g >> 1)) % R]);
        // a use of all eight destinations: the compiler drains its loads HERE and carries no pending load out of this path
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ring[0][(BASE + ELEM) % R]), "+v"(ring[1][(BASE + ELEM) % R]), "+v"(ring[2][(BASE + ELEM) % R]),
                     "+v"(ring[3][(BASE + ELEM) % R]), "+v"(ring[0][(BASE + ELEM + 1) % R]), "+v"(ring[1][(BASE + ELEM + 1) % R]),
                     "+v"(ring[2][(BASE + ELEM + 1) % R]), "+v"(ring[3][(BASE + ELEM + 1) % R]) :: "memory");
    }
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

// ---- the 12-wave pipeline (P = 8): fold waves and FFT waves ----
// The 4-wave kernels above keep every per-thread state of the filter bank in ONE set of waves: delay lines, coefficients and
// input prefetch (112 VGPRs) next to the radix-16 temporaries (64) and the twiddles -- 222 VGPRs, two waves per SIMD, and a
// 70 KB exchange buffer per workgroup; half of the issue slots stay empty while both waves of a SIMD sit in LDS latency or in
// one of the three barriers per four frames.  Here one 768-thread workgroup owns a CU and splits the ROLES between waves:
//   waves 0..3   "fold":  thread t keeps branches t + 256 j in registers (exactly as above), folds a frame and runs the
//                radix-4 pass 1 on its own registers -> bufA[next][frame].  Pure VALU + prefetched global loads.
//   waves 4..11  "FFT":   wave w owns frame w of a batch of EIGHT outright and runs the rest of the FFT wave-privately, in
//                place: pass 2 (radix 16, p = 4) as above, then ONE more radix-16 pass (p = 64: lane i takes points
//                i + 64 r, twiddles W_1024^{r i} from registers) which leaves the frame in natural order.  After a barrier
//                thread u reads bins u and u + 512 of all eight frames: a lane owns the same two channels for ever, and
//                the slicer state is two small register sets.
// The fold waves work one batch ahead into the other half of a double buffer (2 x 8 frames = 136 KB of LDS), so a batch
// costs two workgroup barriers per EIGHT frames instead of six, the fold's VALU stream fills the issue slots the FFT waves
// leave while they wait for LDS, and both roles fit 168 VGPRs: three waves per SIMD.  FFT-1024 = 4 x 16 x 16 needs one
// LDS round trip less than 4 x 16 x 4 x 4.  The unfused form is the same kernel with a different epilogue (MODE_IQ: the
// bins go to the channel-major block as 64-byte runs), so fused and unfused forms stay bit-identical by construction.
constexpr int CHZ_NB = 8;                                        // frames per batch
constexpr int CHZ12_IQ = -1;                                     // MODE: write the channel-major block; >= 0: AMPS_SLICER_* fused behind the FFT

// pass 3 of the 4 x 16 x 16 factorisation, radix 16, p = 64, one frame per wave, in place:
//   lane i: u[r] = A[i + 64 r] W_1024^{r i};  A[i + 64 r] = DFT16(u)[r]        (natural bin order)
__device__ __forceinline__ void chz_p34(cf2 *A, const cf2 (&tw)[15], int lane)
{
    cf2 u[16];
    cf2 *src = A + lane;                                        // chz_pos2(lane + 64 r) = lane + 68 r
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
#pragma unroll
    for (int c = 0; c < 4; c++) {
        cf2 X[4];
        dft4(v[0][c], v[1][c], v[2][c], v[3][c], X);
#pragma unroll
        for (int d = 0; d < 4; d++) src[68 * (c + 4 * d)] = X[d];
    }
}

// slicer state of the two bins a lane of the FFT role owns (specs of include/amps_recc_numerics.h)
template <int SL> struct ChzSlice2 {
    cf2 prev[2];             // spec A / C: the bins one frame earlier
    f2 d1, d2;               // spec A / C: the last two discriminator outputs of (bin 0, bin 1)
    cf2 h1[2], h2[2], h3[2]; // spec B: the bins one, two and three frames earlier
    uint32_t gw[2];          // spec A: slicer bits, newest at bit 31; specs B / C: SIGN bits, newest at bit 0
    __device__ __forceinline__ void reset()
    {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            prev[j] = h1[j] = h2[j] = h3[j] = (cf2){ 0.f, 0.f };
            gw[j] = SL == AMPS_SLICER_ATAN_BOXCAR ? ~0u : 0u;     // "ones before the stream" in either representation
        }
        d1 = d2 = (f2){ 0.f, 0.f };
    }
    template <int PAR> __device__ __forceinline__ void step(const cf2 (&y)[2])   // PAR = parity of the absolute frame index
    {
        if constexpr (SL == AMPS_SLICER_PRODUCT) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                cf2 m;                                              // (yr * pi, yi * pr): the partner's halves swapped by op_sel
                asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(m) : "v"(y[j]), "v"(h3[j]));
                gw[j] = __builtin_amdgcn_alignbit(gw[j], __float_as_uint(m.y - m.x), 31);
                h3[j] = h2[j]; h2[j] = h1[j]; h1[j] = y[j];
            }
        } else {
            f2 d;
            if constexpr (SL == AMPS_SLICER_SINE)
                d = (f2){ __builtin_fmaf(y[0].y, prev[0].x, -(y[0].x * prev[0].y)), __builtin_fmaf(y[1].y, prev[1].x, -(y[1].x * prev[1].y)) };
            else d = fm_phase_two(y[0], prev[0], y[1], prev[1]);
            // window [n-2, n]: n even -> (d[n-2] + d[n-1]) + d[n];  n odd -> d[n-2] + (d[n-1] + d[n])
            const f2 s = PAR == 0 ? (d2 + d1) + d : d2 + (d1 + d);
            if constexpr (SL == AMPS_SLICER_SINE) {
                gw[0] = __builtin_amdgcn_alignbit(gw[0], __float_as_uint(s.x), 31);
                gw[1] = __builtin_amdgcn_alignbit(gw[1], __float_as_uint(s.y), 31);
            } else {
                gw[0] = __builtin_amdgcn_alignbit(s.x >= 0.0f ? 1u : 0u, gw[0], 1);
                gw[1] = __builtin_amdgcn_alignbit(s.y >= 0.0f ? 1u : 0u, gw[1], 1);
            }
            d2 = d1; d1 = d;
            prev[0] = y[0]; prev[1] = y[1];
        }
    }
    __device__ __forceinline__ uint32_t word(int j) const { return SL == AMPS_SLICER_ATAN_BOXCAR ? gw[j] : ~__builtin_bitreverse32(gw[j]); }
};

template <int P, int MODE>
__global__ __launch_bounds__(768, 3) void chz12_kernel(ChzArgs a)
{
    constexpr int M = CHZ_M, D = CHZ_D, NB = CHZ_NB;
    constexpr bool IQ = MODE == CHZ12_IQ;
    constexpr int SL = IQ ? AMPS_SLICER_ATAN_BOXCAR : MODE;
    __shared__ cf2 bufA[2][NB * CHZ_FB];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int64_t f0 = (int64_t)blockIdx.x * a.frames_per_wg;   // multiple of 64
    if (f0 >= (int64_t)a.nframes) return;
    int64_t f1 = f0 + a.frames_per_wg; if (f1 > (int64_t)a.nframes) f1 = a.nframes;
    // the slicer state of every bin is rebuilt by one pre-roll batch; its first four frames may reach behind the carry
    // (zeros): they only prime the delay lines for the last four, which are exact (the carry holds L - D + 4 D samples)
    const int64_t fs = IQ ? f0 : f0 - NB;
    const int nbatch = (int)((f1 - fs + NB - 1) / NB);

    // time step s: the fold waves produce batch s into bufA[s & 1] while the FFT waves consume batch s - 1 from the other
    // half; both roles pass the same two barriers per step
    if (wave < 4) {
        // ------------------------------------------------------------------ fold role
        const int t = tid;
        const ChzIn in{ a.block, a.carry, (int64_t)a.hist, (int64_t)a.carry_len - (int64_t)a.hist, (int64_t)a.carry_len, (int64_t)a.nsamp };
        cf2 coef[4][P / 2];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int q = 0; q < P; q += 2) coef[j][q / 2] = (cf2){ a.taps[t + 256 * j + q * M], a.taps[t + 256 * j + (q + 1) * M] };
        cf2 tw1[3];                                               // pass 2's input twiddles of this thread's outputs k1 = 1..3
#pragma unroll
        for (int k1 = 1; k1 < 4; k1++) tw1[k1 - 1] = chz_twiddle((t >> 4) * k1, 64);
        cf2 ring[4][P + 4];                                       // delay lines + the inputs of this and the next half-step
        {
            const int64_t vend = fs * D;                          // multiple of M (fs is even)
#pragma unroll
            for (int jb = 0; jb < 4; jb++) {
                const int64_t vlast = vend - M + (t + 256 * jb);
#pragma unroll
                for (int q = 0; q < P; q++) ring[jb][q] = in.generic(vlast - (int64_t)M * (P - 1 - q));
            }
        }
        chz_load_half_ring<P, 0, P, false>(ring, in, fs, t);
        chz_load_half_ring<P, 0, P + 2, false>(ring, in, fs + CHZ_BATCH, t);
        __syncthreads();                                          // both roles start together
        // one half-step = four frames: fold them, then load the frames of the half-step after next into the two slots that
        // just died.  A load has eight frames (~4 us) to arrive: with four frames of lead, as in the 4-wave kernels, the fold
        // waves were the critical path (4 waves x 8 loads x 512 B = 16 KB in flight per CU do not cover the HBM latency under
        // load: fold + slicer alone ran 0.39 ms per GiB, the FFT waves alone 0.31).
        auto half_step = [&](auto basec, int sidx, int half) {
            constexpr int BASE = decltype(basec)::value;
            if (sidx < nbatch) {
                const int64_t F = fs + (int64_t)NB * sidx + CHZ_BATCH * half;
                cf2 *dst = bufA[sidx & 1] + CHZ_BATCH * half * CHZ_FB;
                chz_ring_wait<P, BASE>(ring);
                chz_fold_half_ring<P, BASE>(ring, coef, tw1, dst, t);
                if (in.batch_in_block(F + NB)) chz_load_half_ring<P, BASE, P + 4, true>(ring, in, F + NB, t);
                else chz_load_half_ring<P, BASE, P + 4, false>(ring, in, F + NB, t);   // generic: zero beyond the data
            }
            __syncthreads();
        };
        constexpr int PERIOD = (P + 4) / 2;                       // half-steps until the ring is back where it started (6)
        static_assert(PERIOD == 6, "the unrolled loop below is written for P = 8");
        for (int s = 0; s <= nbatch; s += 3) {                    // three time steps = six half-steps = one ring period
            half_step(std::integral_constant<int, 0>{}, s, 0);
            half_step(std::integral_constant<int, 2>{}, s, 1);
            if (s + 1 > nbatch) break;
            half_step(std::integral_constant<int, 4>{}, s + 1, 0);
            half_step(std::integral_constant<int, 6>{}, s + 1, 1);
            if (s + 2 > nbatch) break;
            half_step(std::integral_constant<int, 8>{}, s + 2, 0);
            half_step(std::integral_constant<int, 10>{}, s + 2, 1);
        }
    } else {
        // ------------------------------------------------------------------ FFT role
        const int u = tid - 256;                                  // 0..511: owns bins u and u + 512
        const int wf = wave - 4;                                  // frame of the batch this wave transforms
        cf2 tw34[15];                                             // twiddles of the second radix-16 pass: W_1024^{r lane}
#pragma unroll
        for (int r = 1; r < 16; r++) tw34[r - 1] = chz_twiddle(r * lane, 1024);
        const cf2 tw2[15] = {};                                   // (pass 2's are applied by the fold waves)
        ChzSlice2<SL> S;
        S.reset();
        uint32_t hold[2][4] = {};                                 // finished ring words of the two bins waiting for their 16-byte store
        int nheld = 0;
        const uint64_t mask32 = 2ull * a.ring_words - 1;
        uint32_t ch[2];
#pragma unroll
        for (int j = 0; j < 2; j++) ch[j] = ((uint32_t)(u + 512 * j) - a.first_bin) & (M - 1);
        __syncthreads();                                          // both roles start together
        for (int s = 0; s <= nbatch; s++) {
            cf2 *A = bufA[(s - 1) & 1];
            if (s >= 1) {
                cf2 *Af = A + wf * CHZ_FB;
                chz_p2_t<true>(Af, nullptr, tw2, lane);
                chz_p34(Af, tw34, lane);
            }
            __syncthreads();
            if (s >= 1) {
                const int64_t F = fs + (int64_t)NB * (s - 1);     // first frame of the batch (multiple of 8)
                cf2 y[NB][2];
#pragma unroll
                for (int g = 0; g < NB; g++) { y[g][0] = A[g * CHZ_FB + chz_pos2(u)]; y[g][1] = A[g * CHZ_FB + chz_pos2(u + 512)]; }
                if constexpr (IQ) {
                    // eight frames of a bin leave as one 64-byte run of the channel-major block
                    const int ng = (int)(f1 - F < (int64_t)NB ? f1 - F : (int64_t)NB);
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        if (ch[j] < a.n_channels) {
                            float2 *dstp = a.out + (uint64_t)ch[j] * a.ld + F;
                            if (ng == NB) {
#pragma unroll
                                for (int e = 0; e < NB; e += 2)
                                    *(float4 *)(dstp + e) = make_float4(y[e][j].x, y[e][j].y, y[e + 1][j].x, y[e + 1][j].y);
                            } else {
#pragma unroll
                                for (int e = 0; e < NB; e++)
                                    if (e < ng) dstp[e] = make_float2(y[e][j].x, y[e][j].y);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < NB; g++) { if (g & 1) S.template step<1>(y[g]); else S.template step<0>(y[g]); }
                    if (a.stream_start && F < 0) {                // frames before the stream are exactly +0 (the FFT of zeros may hold -0):
                        asm volatile("" ::: "memory");            // the state they leave is that of a fresh stream (a real branch, once per launch)
                        S.reset();
                    }
                    if (F >= f0 && ((F + NB - 1) & 31) == 31) {   // 32 real frames collected (f0 is a multiple of 64)
                        // A channel's words leave as ONE 16-byte store per 128 frames (aligned group of four ring dwords): single
                        // dwords scattered over the channels' ring rows are counted -- and written -- as 32-byte sectors, 8x the 27 MB
                        // of slicer bits per GiB of input (round 1: 215 MB of 1.36 GB traffic).  Ranges start and end on 64-frame
                        // boundaries, so a run that is not a whole group is exactly two words.
                        const uint64_t w = (a.n_done + (uint64_t)(F + NB - 1)) >> 5;   // absolute ring dword of the finished word
                        const bool last = F + NB >= f1;
#pragma unroll
                        for (int j = 0; j < 2; j++) {
                            uint32_t word = S.word(j);
                            if (SL == AMPS_SLICER_PRODUCT && a.stream_start && F + NB - 1 == 31) word |= 7u;   // no partner yet: g = 1
                            hold[j][0] = hold[j][1]; hold[j][1] = hold[j][2]; hold[j][2] = hold[j][3]; hold[j][3] = word;
                        }
                        nheld++;
                        if ((w & 3) == 3 || last) {
#pragma unroll
                            for (int j = 0; j < 2; j++) {
                                if (ch[j] < a.n_channels) {
                                    uint32_t *row = (uint32_t *)(a.gring + (uint64_t)ch[j] * a.ring_words);
                                    if (nheld == 4 && (w & 3) == 3) *(uint4 *)(row + ((w - 3) & mask32)) = make_uint4(hold[j][0], hold[j][1], hold[j][2], hold[j][3]);
                                    else if (nheld == 2) *(uint2 *)(row + ((w - 1) & mask32)) = make_uint2(hold[j][2], hold[j][3]);
                                    else for (int k = 0; k < nheld; k++) row[(w - (uint64_t)(nheld - 1 - k)) & mask32] = hold[j][4 - nheld + k];   // not reached: ranges are multiples of 64 frames
                                }
                            }
                            nheld = 0;
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
}

// carry_out[k] = virtual sample (consumed - hist + k), k in [0, hist + leftover_new)
__global__ __launch_bounds__(256) void chz_carry_kernel(const float2 *block, const float2 *carry_in, float2 *carry_out,
                                                         uint32_t carry_len, uint32_t nsamp, uint32_t hist, uint32_t consumed,
                                                         uint32_t out_len)
{
    const int64_t lead = (int64_t)carry_len - hist;
    for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < out_len; k += gridDim.x * 256) {
        int64_t v = (int64_t)consumed - hist + k;   // launch-relative virtual index
        int64_t ci = v + hist;
        float2 s = make_float2(0.f, 0.f);
        if (ci >= 0) {
            if (ci < (int64_t)carry_len) s = carry_in[ci];
            else { int64_t bi = v - lead; if (bi < (int64_t)nsamp) s = block[bi]; }
        }
        carry_out[k] = s;
    }
}

struct ChannelizerState {
    bool enabled = false;
    int P = 8;
    uint32_t C = 0, first_bin = 0;
    uint32_t max_frames = 0;        // per push
    uint32_t target_wgs = 512;      // resident workgroups of the filter-bank kernel (2 per CU)
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

inline bool chz_legacy_kernels()   // AMPS_RECC_CHZ=legacy: the 4-wave kernels also for P = 8 (A/B measurements)
{
    static int v = -1;
    if (v < 0) { const char *e = std::getenv("AMPS_RECC_CHZ"); v = (e && e[0] == 'l') ? 1 : 0; }
    return v == 1;
}

inline double bessel_i0(double x)
{
    double s = 1.0, t = 1.0;
    for (int k = 1; k < 64; k++) { t *= (x / (2.0 * k)) * (x / (2.0 * k)); s += t; if (t < 1e-18 * s) break; }
    return s;
}

// Kaiser(beta = 8) windowed sinc, cutoff 13 kHz at fs = M * 30 kHz, unit DC gain
inline std::vector<float> chz_design_taps(int P)
{
    const int L = P * CHZ_M;
    const double fc = 13.0e3 / (CHZ_M * 30.0e3);      // cycles per sample
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

inline uint32_t chz_hist(int P) { return (uint32_t)(P * CHZ_M - CHZ_D + CHZ_PRE * CHZ_D); }
inline size_t chz_carry_cap(int P) { return (size_t)chz_hist(P) + 64 * CHZ_D; }   // + leftover (< 64 frames)

inline int channelizer_reset(ChannelizerState &z, hipStream_t s)
{
    if (!z.enabled) return 0;
    const size_t cap = chz_carry_cap(z.P);
    if (hipMemsetAsync(z.carry[0], 0, sizeof(float2) * cap, s) != hipSuccess) return -EIO;
    if (hipMemsetAsync(z.carry[1], 0, sizeof(float2) * cap, s) != hipSuccess) return -EIO;
    z.carry_cur = 0;
    z.carry_len = chz_hist(z.P);                     // all-zero history, no leftover
    z.frames_done = 0;
    return 0;
}

inline void channelizer_destroy(ChannelizerState &z)
{
    z.stage_fence.destroy();
    void *bufs[] = { z.taps, z.carry[0], z.carry[1], z.out, z.stage };
    for (void *p : bufs) if (p) (void)hipFree(p);
    z = ChannelizerState();
}

inline int channelizer_create(ChannelizerState &z, const amps_recc_cfg_t &cfg, hipStream_t s)
{
    if (cfg.wideband_channels != CHZ_M || cfg.wideband_decim != CHZ_D) return -EINVAL;   // round 1: M = 1024, D = 512
    const int P = cfg.wideband_taps_per_branch ? (int)cfg.wideband_taps_per_branch : 8;
    if (P != 8 && P != 16) return -EINVAL;
    if (cfg.n_channels > CHZ_M || cfg.wideband_first_channel >= CHZ_M || cfg.max_samples_per_push == 0) return -EINVAL;
    z.P = P; z.C = cfg.n_channels; z.first_bin = cfg.wideband_first_channel;
    z.max_frames = cfg.max_samples_per_push;
    z.ld = ((uint64_t)z.max_frames + 7) & ~7ull;
    const size_t L = (size_t)P * CHZ_M;
    std::vector<float> h = chz_design_taps(P);
    if (hipMalloc((void **)&z.taps, sizeof(float) * L) != hipSuccess) return -ENOMEM;
    if (hipMemcpy(z.taps, h.data(), sizeof(float) * L, hipMemcpyHostToDevice) != hipSuccess) return -EIO;
    if (hipMalloc((void **)&z.carry[0], sizeof(float2) * chz_carry_cap(P)) != hipSuccess) return -ENOMEM;
    if (hipMalloc((void **)&z.carry[1], sizeof(float2) * chz_carry_cap(P)) != hipSuccess) return -ENOMEM;
    // z.out (the channel-major block, C x ld x 8 B: 1.7 GB for a full band at 2^18 frames per push) is allocated by the first
    // unfused / debug run: the fused form never touches it
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            z.target_wgs = 2u * (uint32_t)prop.multiProcessorCount;
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
    const uint32_t hist = chz_hist(z.P);
    const uint32_t leftover = z.carry_len - hist;
    const uint64_t avail = (uint64_t)leftover + nsamp;
    // frames consumed: even (keeps the frame parity of a launch at 0), and in the fused form a multiple of 64
    // (whole words of the RECC bit ring); the rest waits in the carry
    const uint32_t nframes = (uint32_t)(avail / CHZ_D) & (fused ? ~63u : ~1u);
    if (nframes > z.max_frames) return -E2BIG;
    if (nframes) {
        ChzArgs a{};
        a.block = d; a.carry = z.carry[z.carry_cur]; a.taps = z.taps; a.out = z.out; a.ld = z.ld;
        a.carry_len = z.carry_len; a.nsamp = (uint32_t)nsamp; a.nframes = nframes; a.hist = hist;
        // one resident round: two workgroups per CU (register-limited); each refills its delay lines (+ 4 pre-roll
        // frames when fused), so fewer, longer runs are cheaper (measured 1 GiB: 128 frames/WG 0.998 ms, 512 0.976 ms)
        uint32_t fpw = (nframes + z.target_wgs - 1) / z.target_wgs;
        fpw = std::max<uint32_t>(fused ? 128u : 64u, fpw);
        fpw = (fpw + 63) / 64 * 64;
        a.frames_per_wg = fpw; a.first_bin = z.first_bin; a.n_channels = z.C;
        a.odd_start = 0;
        a.gring = gring; a.ring_words = ring_words; a.n_done = n_done;
        a.stream_start = z.frames_done == 0 ? 1u : 0u;
        const uint32_t nwg = (nframes + fpw - 1) / fpw;
        const bool k12 = z.P == 8 && !chz_legacy_kernels();        // the 12-wave pipeline (one workgroup per CU)
        if (k12) {
            fpw = std::max<uint32_t>(64u, (nframes + z.target_wgs / 2 - 1) / (z.target_wgs / 2));
            fpw = (fpw + 63) / 64 * 64;
            a.frames_per_wg = fpw;
            const dim3 g12((nframes + fpw - 1) / fpw), b12(768);
            if (!fused) hipLaunchKernelGGL((chz12_kernel<8, CHZ12_IQ>), g12, b12, 0, s, a);
            else if (slicer == AMPS_SLICER_PRODUCT) hipLaunchKernelGGL((chz12_kernel<8, AMPS_SLICER_PRODUCT>), g12, b12, 0, s, a);
            else if (slicer == AMPS_SLICER_SINE) hipLaunchKernelGGL((chz12_kernel<8, AMPS_SLICER_SINE>), g12, b12, 0, s, a);
            else hipLaunchKernelGGL((chz12_kernel<8, AMPS_SLICER_ATAN_BOXCAR>), g12, b12, 0, s, a);
        } else if (fused && slicer == AMPS_SLICER_PRODUCT) {
            if (z.P == 8) hipLaunchKernelGGL((chz_fused_kernel<8, AMPS_SLICER_PRODUCT>), dim3(nwg), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((chz_fused_kernel<16, AMPS_SLICER_PRODUCT>), dim3(nwg), dim3(256), 0, s, a);
        } else if (fused && slicer == AMPS_SLICER_SINE) {
            if (z.P == 8) hipLaunchKernelGGL((chz_fused_kernel<8, AMPS_SLICER_SINE>), dim3(nwg), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((chz_fused_kernel<16, AMPS_SLICER_SINE>), dim3(nwg), dim3(256), 0, s, a);
        } else if (fused) {
            if (z.P == 8) hipLaunchKernelGGL(chz_fused_kernel<8>, dim3(nwg), dim3(256), 0, s, a);
            else hipLaunchKernelGGL(chz_fused_kernel<16>, dim3(nwg), dim3(256), 0, s, a);
        } else {
            if (z.P == 8) hipLaunchKernelGGL(chz_pfb_fft_kernel<8>, dim3(nwg), dim3(256), 0, s, a);
            else hipLaunchKernelGGL(chz_pfb_fft_kernel<16>, dim3(nwg), dim3(256), 0, s, a);
        }
    }
    if (after_main) after_main(after_ctx);                            // timing: the span ends behind the filter-bank kernel, before the carry copy
    const uint32_t consumed = nframes * CHZ_D;                        // virtual samples consumed (incl. leftover)
    const uint32_t new_left = (uint32_t)(avail - consumed);
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
