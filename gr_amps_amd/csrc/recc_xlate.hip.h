// recc_xlate.hip.h -- the single-channel front filter of grc/recctest.grc on gfx950:
// freq_xlating_fir_filter_ccc (grc/recctest.grc:889-937) with firdes.low_pass taps (:115-155): translate the
// channel at `center_hz` to DC, low-pass, decimate by D.  It stands in front of the fused IQ seam for the
// ".raw" fc32 400 ksps captures the reference's test flow graph reads (grc/recctest.grc:591).
//
// The reference block multiplies by COMPLEX composite taps h[i] e^{j i phi} and then by a running rotator
// (4 real MACs per tap, rotator renormalised every 512 outputs).  Algebraically
//     y[k] = e^{-j phi D k} sum_i h[i] e^{j phi i} x[Dk - i] = sum_i h[i] z[Dk - i],   z[n] = x[n] e^{-j phi n}
// so this kernel mixes each input sample ONCE while staging it into LDS (phase from an exact 64-bit phase
// accumulator evaluated per sample: no rotator drift, any push boundary gives the same bits) and then runs a REAL-tap FIR on complex
// data: one v_pk_fma per tap.  A lane produces 8/D adjacent outputs, so consecutive taps reuse the same LDS
// words (14 ds_read_b64 per 32 v_pk_fma at D = 2); the LDS window is padded by one sample per eight so that the
// lane stride of 8 samples is conflict free and the pad term of the address is wave-uniform (scalar) arithmetic.
// This is the file-tool path (one or a few channels); the 832-channel front end is the polyphase channelizer.
#pragma once
#include <hip/hip_runtime.h>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <vector>
#include "recc_channelizer.hip.h"   // cf2, cmul

namespace amps {

constexpr int XL_TILE = 2048;        // input samples per workgroup (256 lanes x 8)
constexpr int XL_MAX_TAPS = 1024;    // padded tap count limit

struct XlateArgs {
    const float2 *block;     // [C][ld_in] new samples
    const float2 *carry;     // [C][carry_cap]: hist samples of history, then the leftover (< D) unconsumed samples
    const float *taps;       // [ntp], zero padded to a multiple of 8
    float2 *out;             // [C][ld_out]
    uint64_t ld_in, ld_out;
    uint64_t n_abs0;         // absolute input index of the first unconsumed sample (virtual index v = hist)
    uint64_t step;           // center_hz / rate_hz as a 0.64 fixed-point fraction of a turn
    uint32_t carry_cap, carry_len, hist, nsamp, nout, ntp;
};

__host__ __device__ constexpr int xl_pad(int n) { return n + (n >> 3); }

// e^{-j 2 pi frac(n * step)}: the top 24 bits of the wrapped product are exact in fp32
__device__ __forceinline__ cf2 xl_phasor(uint64_t turns)
{
    float sn, cs;
    sincospif(-(float)(uint32_t)(turns >> 40) * 0x1p-23f, &sn, &cs);   // argument in half-turns, exact: no Payne-Hanek path
    return (cf2){ cs, sn };
}

template <int D>
__global__ __launch_bounds__(256) void xlate_fir_kernel(XlateArgs a)
{
    constexpr int OPT = 8 / D;                                   // outputs per lane
    __shared__ cf2 zs[xl_pad(XL_TILE + XL_MAX_TAPS) + 8];
    __shared__ float hs[XL_MAX_TAPS];
    const int t = threadIdx.x;
    const uint32_t c = blockIdx.y;
    const uint32_t k0 = blockIdx.x * (XL_TILE / D);
    const int H = (int)a.hist;                                   // = ntp - 1
    const int ntp = (int)a.ntp;
    const float2 *blk = a.block + (uint64_t)c * a.ld_in;
    const float2 *car = a.carry + (uint64_t)c * a.carry_cap;
    const int64_t vtot = (int64_t)a.carry_len + a.nsamp;

    for (int i = t; i < ntp; i += 256) hs[i] = a.taps[i];
    // stage + mix: tile-local sample n <-> virtual index v = D*k0 + n;  lane t takes n = t, t+256, ...
    {
        const int64_t v0 = (int64_t)D * k0;
        // the phasor is evaluated per sample from the absolute index (not a running rotation), so a sample is mixed to
        // the same bits whatever tile or push it lands in
        const uint64_t nabs0 = a.n_abs0 + (uint64_t)(v0 - H);      // wraps consistently for the (zero) pre-stream history
        for (int n = t; n < XL_TILE + H; n += 256) {
            const int64_t v = v0 + n;
            float2 s = make_float2(0.f, 0.f);
            if (v < (int64_t)a.carry_len) s = car[v];
            else if (v < vtot) s = blk[v - a.carry_len];
            zs[xl_pad(n)] = cmul((cf2){ s.x, s.y }, xl_phasor((nabs0 + (uint64_t)n) * a.step));
        }
    }
    __syncthreads();

    cf2 acc[OPT];
#pragma unroll
    for (int j = 0; j < OPT; j++) acc[j] = (cf2){ 0.f, 0.f };
    // tile-local sample of (output j, tap i+e) = 8t + u, u = (H - 7 - i) + (7 + D j - e): H - 7 - i is a multiple of 8,
    // so xl_pad(8t + u) = 9t + 9 (H - 7 - i)/8 + xl_pad(7 + D j - e) -- a moving base plus compile-time offsets
    const cf2 *zp = zs + 9 * t + 9 * ((H - 7) >> 3);
    for (int i = 0; i < ntp; i += 8, zp -= 9) {                  // ascending tap order: the summation order of the spec
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float h = hs[i + e];
#pragma unroll
            for (int j = 0; j < OPT; j++)
                acc[j] = __builtin_elementwise_fma(zp[xl_pad(7 + D * j - e)], (cf2){ h, h }, acc[j]);
        }
    }
    float2 *o = a.out + (uint64_t)c * a.ld_out;
#pragma unroll
    for (int j = 0; j < OPT; j++) {
        const uint32_t k = k0 + OPT * t + j;
        if (k < a.nout) o[k] = make_float2(acc[j].x, acc[j].y);
    }
}

// carry_out[c][i] = virtual[c][consumed + i], i < new_len  (separate buffers: the ranges can overlap)
__global__ void xlate_carry_kernel(const float2 *block, uint64_t ld_in, const float2 *carry_in, float2 *carry_out,
                                   uint32_t carry_cap, uint32_t carry_len, uint32_t consumed, uint32_t new_len)
{
    const uint32_t c = blockIdx.y;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= new_len) return;
    const uint64_t v = (uint64_t)consumed + i;
    carry_out[(uint64_t)c * carry_cap + i] = v < carry_len ? carry_in[(uint64_t)c * carry_cap + v] : block[(uint64_t)c * ld_in + (v - carry_len)];
}

struct XlateState {
    bool enabled = false;
    uint32_t C = 0, D = 0, ntaps = 0, ntp = 0, hist = 0, carry_cap = 0, carry_len = 0, max_out = 0;
    int cur = 0;
    uint64_t n_abs = 0, step = 0;
    float *taps = nullptr;
    float2 *carry[2] = { nullptr, nullptr };
    float2 *out = nullptr;
    float2 *stage = nullptr;
    size_t stage_samples = 0;
    StageFence stage_fence;
    std::vector<float> taps_host;
};

// firdes.low_pass(gain, fs, cutoff, width, WIN_BLACKMAN) as the flow graph calls it (grc/recctest.grc:115-155):
// ntaps = int(74 fs / (22 width)) made odd; windowed sinc normalised to DC gain `gain`
inline std::vector<float> xlate_design_taps(double gain, double fs, double cutoff, double width)
{
    int n = (int)(74.0 * fs / (22.0 * width));
    if (!(n & 1)) n++;
    const int m = (n - 1) / 2;
    std::vector<double> t((size_t)n);
    const double w0 = 2.0 * M_PI * cutoff / fs;
    double sum = 0.0;
    for (int i = 0; i < n; i++) {
        const int k = i - m;
        const double win = 0.42 - 0.5 * std::cos(2.0 * M_PI * i / (n - 1)) + 0.08 * std::cos(4.0 * M_PI * i / (n - 1));
        t[(size_t)i] = (k == 0 ? w0 / M_PI : std::sin(k * w0) / (k * M_PI)) * win;
        sum += t[(size_t)i];
    }
    std::vector<float> out((size_t)n);
    for (int i = 0; i < n; i++) out[(size_t)i] = (float)(t[(size_t)i] * gain / sum);
    return out;
}

inline void xlate_destroy(XlateState &x)
{
    if (x.taps) (void)hipFree(x.taps);
    if (x.carry[0]) (void)hipFree(x.carry[0]);
    if (x.carry[1]) (void)hipFree(x.carry[1]);
    if (x.out) (void)hipFree(x.out);
    if (x.stage) (void)hipFree(x.stage);
    x.stage_fence.destroy();
    x = XlateState{};
}

inline int xlate_reset(XlateState &x, hipStream_t s)
{
    if (!x.enabled) return 0;
    if (hipMemsetAsync(x.carry[0], 0, sizeof(float2) * (size_t)x.C * x.carry_cap, s) != hipSuccess) return -EIO;
    if (hipMemsetAsync(x.carry[1], 0, sizeof(float2) * (size_t)x.C * x.carry_cap, s) != hipSuccess) return -EIO;
    x.cur = 0; x.carry_len = x.hist; x.n_abs = 0;
    return 0;
}

inline int xlate_create(XlateState &x, uint32_t C, uint32_t D, uint32_t max_out, double rate_hz, double center_hz,
                        const std::vector<float> &taps, hipStream_t s)
{
    xlate_destroy(x);
    if (!(D == 1 || D == 2 || D == 4) || taps.empty() || !(rate_hz > 0.0) || std::fabs(center_hz) > rate_hz) return -EINVAL;
    const uint32_t ntp = (uint32_t)((taps.size() + 7) / 8 * 8);
    if (ntp > (uint32_t)XL_MAX_TAPS) return -E2BIG;
    x.C = C; x.D = D; x.ntaps = (uint32_t)taps.size(); x.ntp = ntp; x.hist = ntp - 1; x.carry_cap = ntp + D; x.max_out = max_out;
    x.taps_host = taps;
    // fraction of a turn per input sample, two's complement for negative offsets
    const long double f = (long double)center_hz / (long double)rate_hz;
    const long double fr = f - std::floor(f);
    x.step = (uint64_t)(fr * 18446744073709551616.0L);
    std::vector<float> padded(ntp, 0.0f);
    for (size_t i = 0; i < taps.size(); i++) padded[i] = taps[i];
    if (hipMalloc((void **)&x.taps, sizeof(float) * ntp) != hipSuccess) { xlate_destroy(x); return -ENOMEM; }
    if (hipMalloc((void **)&x.carry[0], sizeof(float2) * (size_t)C * x.carry_cap) != hipSuccess) { xlate_destroy(x); return -ENOMEM; }
    if (hipMalloc((void **)&x.carry[1], sizeof(float2) * (size_t)C * x.carry_cap) != hipSuccess) { xlate_destroy(x); return -ENOMEM; }
    if (hipMalloc((void **)&x.out, sizeof(float2) * (size_t)C * max_out) != hipSuccess) { xlate_destroy(x); return -ENOMEM; }
    if (hipMemcpy(x.taps, padded.data(), sizeof(float) * ntp, hipMemcpyHostToDevice) != hipSuccess) { xlate_destroy(x); return -EIO; }
    x.enabled = true;
    return xlate_reset(x, s);
}

// filter nsamp new samples per channel ([C][ld], host or device); *out_iq is [C][*out_ld] device memory holding *nout samples
inline int xlate_run(XlateState &x, const float2 *iq, uint64_t ld, size_t nsamp, int mem, hipStream_t s,
                     const float2 **out_iq, uint64_t *out_ld, uint32_t *nout)
{
    *out_iq = x.out; *out_ld = x.max_out; *nout = 0;
    if (!x.enabled) return -ENOSYS;
    if (nsamp == 0) return 0;
    if (nsamp > (size_t)x.D * x.max_out) return -E2BIG;
    const float2 *d = iq;
    if (mem == AMPS_MEM_HOST) {
        if (int rc = x.stage_fence.wait()) return rc;             // the previous push may still be reading the staging buffer
        const size_t need = (size_t)x.C * x.D * x.max_out;
        if (x.stage_samples < need) {
            if (x.stage) (void)hipFree(x.stage);
            x.stage = nullptr; x.stage_samples = 0;
            if (hipMalloc((void **)&x.stage, sizeof(float2) * need) != hipSuccess) return -ENOMEM;
            x.stage_samples = need;
        }
        if (hipMemcpy2D(x.stage, nsamp * sizeof(float2), iq, ld * sizeof(float2), nsamp * sizeof(float2), x.C,
                        hipMemcpyHostToDevice) != hipSuccess) return -EIO;       // synchronous: see amps_recc_push_iq
        d = x.stage; ld = nsamp;
    }
    const uint64_t avail = (uint64_t)(x.carry_len - x.hist) + nsamp;
    const uint64_t n_out = avail / x.D;
    if (n_out > x.max_out) return -E2BIG;
    XlateArgs a{};
    a.block = d; a.carry = x.carry[x.cur]; a.taps = x.taps; a.out = x.out; a.ld_in = ld; a.ld_out = x.max_out;
    a.n_abs0 = x.n_abs; a.step = x.step; a.carry_cap = x.carry_cap; a.carry_len = x.carry_len; a.hist = x.hist;
    a.nsamp = (uint32_t)nsamp; a.nout = (uint32_t)n_out; a.ntp = x.ntp;
    if (n_out) {
        const dim3 grid((uint32_t)((n_out * x.D + XL_TILE - 1) / XL_TILE), x.C);
        switch (x.D) {
        case 1: hipLaunchKernelGGL(xlate_fir_kernel<1>, grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL(xlate_fir_kernel<2>, grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(xlate_fir_kernel<4>, grid, dim3(256), 0, s, a); break;
        }
    }
    const uint32_t consumed = (uint32_t)(n_out * x.D);
    const uint32_t new_len = x.hist + (uint32_t)(avail - (uint64_t)consumed);
    hipLaunchKernelGGL(xlate_carry_kernel, dim3((new_len + 255) / 256, x.C), dim3(256), 0, s, d, ld, x.carry[x.cur], x.carry[x.cur ^ 1],
                       x.carry_cap, x.carry_len, consumed, new_len);
    if (hipGetLastError() != hipSuccess) return -EIO;
    if (mem == AMPS_MEM_HOST) { if (int rc = x.stage_fence.arm(s)) return rc; }
    x.cur ^= 1; x.carry_len = new_len; x.n_abs += consumed;
    *nout = (uint32_t)n_out;
    return 0;
}

} // namespace amps
