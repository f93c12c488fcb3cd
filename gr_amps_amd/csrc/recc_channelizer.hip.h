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
//   * the last L input samples live in an LDS ring (sample n at slot n mod L: 64 KiB for P = 8); a frame
//     adds D = 512 new samples with one coalesced 4 KiB read (the next frame's are prefetched in registers);
//   * thread t folds the four outputs t' = t + 256 j: because it takes r = (t' + n0) mod M the tap indices
//     are always {t' + qM}, so its 4P coefficients stay in registers for the whole kernel and only the ring
//     address rotates; one v_pk_fma per tap (complex sample x real coefficient);
//   * FFT-1024 = five radix-4 Stockham passes; pass 1 runs on the registers the fold just produced, passes
//     2-4 exchange through two 8 KiB LDS buffers, pass 5 leaves bins {t, t+256, t+512, t+768} in registers --
//     so a thread owns the same four channels in every frame.  Twiddles are per-thread constants, computed
//     once (12 complex registers);
//   * eight frames of a thread's four bins are kept in registers and written as 64-byte runs into the
//     channel-major output (dwordx4 stores), which recc_front_kernel then streams at full rate.
// No MFMA: the contraction per channel is 8..16 taps deep and the FFT is a butterfly network.
#pragma once
#include <hip/hip_runtime.h>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <vector>
#include "amps_recc.h"

namespace amps {

constexpr int CHZ_M = 1024;          // branches = FFT size
constexpr int CHZ_D = 512;           // input samples per frame (2x oversampled)
constexpr int CHZ_GROUP = 8;         // frames buffered in registers per output store

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
};

typedef float cf2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ cf2 cmul(cf2 a, cf2 b) { return (cf2){ a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x }; }
__device__ __forceinline__ cf2 mul_mi(cf2 a) { return (cf2){ a.y, -a.x }; }   // a * (-i)

template <int P>
__global__ __launch_bounds__(256) void chz_pfb_fft_kernel(ChzArgs a)
{
    constexpr int M = CHZ_M, D = CHZ_D, L = P * M;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf2 *ring = (cf2 *)smem;                 // [L]
    cf2 *bufA = ring + L;                    // [M]
    cf2 *bufB = bufA + M;                    // [M]
    const int t = threadIdx.x;
    const uint32_t f0 = blockIdx.x * a.frames_per_wg;          // first frame of this workgroup (launch-relative)
    if (f0 >= a.nframes) return;
    uint32_t f1 = f0 + a.frames_per_wg; if (f1 > a.nframes) f1 = a.nframes;

    // virtual input stream of this launch: index v in [-(L-D), nsamp + leftover): carry then block.
    // frame f (launch-relative) consumes v in [f*D - (L-D), f*D + D)
    const int64_t hist = (int64_t)L - D;
    const int64_t lead = (int64_t)a.carry_len - hist;          // leftover samples (< D) that precede the block
    auto fetch = [&](int64_t v) -> cf2 {
        int64_t ci = v + hist;                                  // index into carry
        if (ci < 0) return (cf2){ 0.f, 0.f };
        float2 s;
        if (ci < (int64_t)a.carry_len) s = a.carry[ci];
        else { int64_t bi = v - lead; if (bi >= (int64_t)a.nsamp) return (cf2){ 0.f, 0.f }; s = a.block[bi]; }
        return (cf2){ s.x, s.y };
    };

    // coefficients h[t' + qM] for t' = t + 256 j
    float coef[4][P];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int q = 0; q < P; q++) coef[j][q] = a.taps[t + 256 * j + q * M];
    // twiddles of passes 2..5: w1 = exp(-2 pi i k / (4 Ns)), k = t & (Ns-1), Ns = 4, 16, 64, 256
    cf2 tw[4][3];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int Ns = 4 << (2 * p);
        const int k = t & (Ns - 1);
        float sn, cs;
        sincosf(-6.283185307179586f * (float)k / (float)(4 * Ns), &sn, &cs);
        tw[p][0] = (cf2){ cs, sn };
        tw[p][1] = cmul(tw[p][0], tw[p][0]);
        tw[p][2] = cmul(tw[p][1], tw[p][0]);
    }

    // ring slot of launch-relative sample v: (v + base) mod L, base chosen so that slots line up with
    // n mod L of the absolute stream only up to a constant -- what matters is n mod M, tracked by parity below
    auto slot = [&](int64_t v) -> int { return (int)((v + (int64_t)L * 4) & (L - 1)); };
    // prologue: history of the first frame
    for (int64_t v = (int64_t)f0 * D - hist + t; v < (int64_t)f0 * D; v += 256) ring[slot(v)] = fetch(v);
    cf2 nx0 = fetch((int64_t)f0 * D + t), nx1 = fetch((int64_t)f0 * D + 256 + t);

    cf2 acc[CHZ_GROUP][4] = {};
    for (uint32_t f = f0; f < f1; f++) {
        const int64_t vs = (int64_t)f * D;                      // first new sample of this frame
        ring[slot(vs + t)] = nx0;
        ring[slot(vs + 256 + t)] = nx1;
        if (f + 1 < f1) { nx0 = fetch(vs + D + t); nx1 = fetch(vs + D + 256 + t); }
        __syncthreads();
        // ---- fold: window = v in [vs + D - L, vs + D); tap i <-> v = vs + D - L + i; thread's taps i = t' + qM
        cf2 u[4];
        const int wbase = slot(vs + D - L);                     // ring slot of tap 0
#pragma unroll
        for (int j = 0; j < 4; j++) {
            cf2 s = { 0.f, 0.f };
            const int b = wbase + t + 256 * j;
#pragma unroll
            for (int q = 0; q < P; q++) {
                const cf2 x = ring[(b + q * M) & (L - 1)];
                s = __builtin_elementwise_fma(x, (cf2){ coef[j][q], coef[j][q] }, s);
            }
            u[j] = s;
        }
        // FFT input index r = (n0 + t') mod M: n0 mod M alternates 0 / M/2 with the absolute frame parity
        // (n0 = (m+1) D - L, L multiple of M, D = M/2), so odd (m+1) swaps the halves: j <-> j ^ 2
        const bool half = ((a.odd_start + f + 1) & 1) != 0;
        cf2 x0 = half ? u[2] : u[0], x1 = half ? u[3] : u[1], x2 = half ? u[0] : u[2], x3 = half ? u[1] : u[3];
        // ---- pass 1 (Ns = 1): registers -> bufA[4t .. 4t+3]
        {
            cf2 v0 = x0 + x2, v1 = x0 - x2, v2 = x1 + x3, v3 = mul_mi(x1 - x3);
            bufA[4 * t + 0] = v0 + v2; bufA[4 * t + 1] = v1 + v3; bufA[4 * t + 2] = v0 - v2; bufA[4 * t + 3] = v1 - v3;
        }
        __syncthreads();
        // ---- passes 2..4 through LDS, pass 5 into registers
        cf2 y0, y1, y2, y3;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int Ns = 4 << (2 * p);
            const cf2 *src = (p & 1) ? bufB : bufA;
            cf2 *dst = (p & 1) ? bufA : bufB;
            cf2 a0 = src[t], a1 = cmul(src[t + 256], tw[p][0]), a2 = cmul(src[t + 512], tw[p][1]), a3 = cmul(src[t + 768], tw[p][2]);
            cf2 v0 = a0 + a2, v1 = a0 - a2, v2 = a1 + a3, v3 = mul_mi(a1 - a3);
            y0 = v0 + v2; y1 = v1 + v3; y2 = v0 - v2; y3 = v1 - v3;
            if (p < 3) {
                const int k = t & (Ns - 1);
                const int i = ((t - k) << 2) + k;
                dst[i] = y0; dst[i + Ns] = y1; dst[i + 2 * Ns] = y2; dst[i + 3 * Ns] = y3;
                __syncthreads();
            }
        }
        // bins t, t+256, t+512, t+768 of this frame: shift them into the 8-frame register window
        // (static register indices only; a dynamically indexed array would live in scratch)
#pragma unroll
        for (int e = 0; e + 1 < CHZ_GROUP; e++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[e][j] = acc[e + 1][j];
        acc[CHZ_GROUP - 1][0] = y0; acc[CHZ_GROUP - 1][1] = y1; acc[CHZ_GROUP - 1][2] = y2; acc[CHZ_GROUP - 1][3] = y3;
        const int g = (int)((f - f0) & (CHZ_GROUP - 1));        // frames held = g + 1, newest at acc[7]
        if (g == CHZ_GROUP - 1 || f + 1 == f1) {
            const uint32_t fbase = f - g;                       // first frame of the group (multiple of 8, launch-relative)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t ch = ((uint32_t)(t + 256 * j) - a.first_bin) & (M - 1);
                if (ch < a.n_channels) {
                    float2 *dstp = a.out + (uint64_t)ch * a.ld + fbase;
                    if (g == CHZ_GROUP - 1) {
#pragma unroll
                        for (int e = 0; e < CHZ_GROUP; e += 2)
                            *(float4 *)(dstp + e) = make_float4(acc[e][j].x, acc[e][j].y, acc[e + 1][j].x, acc[e + 1][j].y);
                    } else {                                     // tail of the workgroup's run: frames sit at acc[7-g .. 7]
#pragma unroll
                        for (int e = 0; e < CHZ_GROUP; e++)
                            if (e >= CHZ_GROUP - 1 - g) dstp[e - (CHZ_GROUP - 1 - g)] = make_float2(acc[e][j].x, acc[e][j].y);
                    }
                }
            }
        }
        // the ring slots this frame read are rewritten next frame only after the barriers above; bufA/bufB are
        // rewritten after the next frame's first barrier
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
    float *taps = nullptr;          // [L]
    float2 *carry[2] = { nullptr, nullptr };
    int carry_cur = 0;
    uint32_t carry_len = 0;         // L - D + leftover
    uint64_t frames_done = 0;
    float2 *out = nullptr;          // [C][ld]
    uint64_t ld = 0;
    float2 *stage = nullptr;        // device staging for host-resident wideband input
    size_t stage_samples = 0;
};

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

inline int channelizer_reset(ChannelizerState &z, hipStream_t s)
{
    if (!z.enabled) return 0;
    const size_t cap = (size_t)z.P * CHZ_M;   // hist + leftover < L
    if (hipMemsetAsync(z.carry[0], 0, sizeof(float2) * cap, s) != hipSuccess) return -EIO;
    if (hipMemsetAsync(z.carry[1], 0, sizeof(float2) * cap, s) != hipSuccess) return -EIO;
    z.carry_cur = 0;
    z.carry_len = (uint32_t)(z.P * CHZ_M - CHZ_D);   // all-zero history, no leftover
    z.frames_done = 0;
    return 0;
}

inline void channelizer_destroy(ChannelizerState &z)
{
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
    if (hipMalloc((void **)&z.carry[0], sizeof(float2) * L) != hipSuccess) return -ENOMEM;
    if (hipMalloc((void **)&z.carry[1], sizeof(float2) * L) != hipSuccess) return -ENOMEM;
    if (hipMalloc((void **)&z.out, sizeof(float2) * (size_t)z.C * z.ld) != hipSuccess) return -ENOMEM;
    z.enabled = true;
    (void)s;
    return 0;
}

// Channelise `nsamp` new wideband samples; on return *chan_iq / *ld / *nframes describe the channel-major block.
inline int channelizer_run(ChannelizerState &z, const float2 *iq, size_t nsamp, int mem, hipStream_t s,
                           const float2 **chan_iq, uint64_t *ld, uint32_t *nframes_out)
{
    if (!z.enabled) return -ENOSYS;
    const float2 *d = iq;
    if (mem == AMPS_MEM_HOST) {
        if (z.stage_samples < nsamp) {
            if (z.stage) (void)hipFree(z.stage);
            z.stage = nullptr; z.stage_samples = 0;
            if (hipMalloc((void **)&z.stage, sizeof(float2) * nsamp) != hipSuccess) return -ENOMEM;
            z.stage_samples = nsamp;
        }
        if (hipMemcpyAsync(z.stage, iq, sizeof(float2) * nsamp, hipMemcpyHostToDevice, s) != hipSuccess) return -EIO;
        d = z.stage;
    }
    const uint32_t L = (uint32_t)z.P * CHZ_M, hist = L - CHZ_D;
    const uint32_t leftover = z.carry_len - hist;
    const uint64_t avail = (uint64_t)leftover + nsamp;
    const uint32_t nframes = (uint32_t)(avail / CHZ_D);
    if (nframes > z.max_frames) return -E2BIG;
    if (nframes) {
        ChzArgs a{};
        a.block = d; a.carry = z.carry[z.carry_cur]; a.taps = z.taps; a.out = z.out; a.ld = z.ld;
        a.carry_len = z.carry_len; a.nsamp = (uint32_t)nsamp; a.nframes = nframes;
        uint32_t fpw = (nframes + 2047) / 2048;                       // aim for ~2048 workgroups
        fpw = std::max<uint32_t>(64, fpw);                            // history refill = 2P-1 frames per workgroup
        fpw = (fpw + CHZ_GROUP - 1) / CHZ_GROUP * CHZ_GROUP;
        a.frames_per_wg = fpw; a.first_bin = z.first_bin; a.n_channels = z.C;
        a.odd_start = (uint32_t)(z.frames_done & 1);
        const uint32_t nwg = (nframes + fpw - 1) / fpw;
        const size_t lds = sizeof(float2) * ((size_t)L + 2 * CHZ_M);
        if (z.P == 8) {
            (void)hipFuncSetAttribute((const void *)chz_pfb_fft_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(chz_pfb_fft_kernel<8>, dim3(nwg), dim3(256), lds, s, a);
        } else {
            (void)hipFuncSetAttribute((const void *)chz_pfb_fft_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(chz_pfb_fft_kernel<16>, dim3(nwg), dim3(256), lds, s, a);
        }
    }
    const uint32_t consumed = nframes * CHZ_D;                        // virtual samples consumed (incl. leftover)
    const uint32_t new_left = (uint32_t)(avail - consumed);
    hipLaunchKernelGGL(chz_carry_kernel, dim3((hist + new_left + 255) / 256), dim3(256), 0, s, d, z.carry[z.carry_cur],
                       z.carry[z.carry_cur ^ 1], z.carry_len, (uint32_t)nsamp, hist, consumed - leftover + leftover, hist + new_left);
    if (hipGetLastError() != hipSuccess) return -EIO;
    z.carry_cur ^= 1;
    z.carry_len = hist + new_left;
    z.frames_done += nframes;
    *chan_iq = z.out; *ld = z.ld; *nframes_out = nframes;
    return 0;
}

} // namespace amps
