// recc_refchain.hip.h -- the reference flow graph's OWN symbol timing on the GPU (checking mode):
//
//   analog.quadrature_demod_cf(gain 1)                              grc/recctest.grc:458      G2
//   digital.clock_recovery_mm_ff(omega 10, gain_omega .25*.175^2*3,
//                                mu 0, gain_mu .05, rel. limit .005) grc/recctest.grc:846-874  G3
//   digital.binary_slicer_fb                                        grc/recctest.grc:807      G4
//
// The fused seam replaces this sub-chain by all-phase slicing + run-centre timing (recc_front.hip.h), so its symbol stream
// cannot be compared with the reference chain's symbol for symbol.  This seam computes the chain as GNU Radio 3.7 defines it
// and hands back the very byte symbols `gr::amps::recc::work` would be fed -- amps_recc_refchain_symbols, then
// amps_recc_push_symbols for the (exact) recc replica: G2 -> G3 -> G4 -> R2 -> R5 end to end on the device, symbol stream
// included.  It exists to be checked against, not to be fast:
//   * the discriminator is embarrassingly parallel (ref_demod_kernel, one thread per sample) and uses GNU Radio's
//     fast_atan2f as published: 255-interval table of atan on [0,1], linear interpolation, octant unfolding;
//   * the Mueller & Mueller loop is a sequential feedback recursion per channel (the interpolation instant of symbol k+1
//     depends on the error of symbol k): ref_mm_kernel runs ONE LANE PER CHANNEL, 832 channels = 13 waves, each lane
//     walking its channel's discriminator stream (8-tap, 129-phase MMSE interpolator, the block's arithmetic in its
//     order: every operation is an IEEE binary32 add / mul / compare / rint / floor, so the device reproduces a CPU
//     evaluation of the same statement bit for bit).
// Stream state (last IQ sample, mu, omega, last symbol, unconsumed discriminator tail) is kept per channel across pushes.
#pragma once
#include <hip/hip_runtime.h>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <vector>
#include "amps_recc.h"

namespace amps {

constexpr int REF_TAIL = 64;              // discriminator samples kept in front of a push (the loop leaves < 8 + 11)

struct RefMM { float mu, omega, last; uint32_t tail, skip; };   // per channel; skip = samples the last step reached beyond the data

struct RefState {
    bool ready = false;
    uint32_t C = 0, max_samples = 0;
    float *atan_tab = nullptr;            // [258]
    float *mmse = nullptr;                // [129][8]
    float2 *last_iq = nullptr;            // [C]
    RefMM *mm = nullptr;                  // [C]
    float *d = nullptr;                   // [C][REF_TAIL + max_samples]
    uint8_t *syms = nullptr;              // [C][sym_cap]
    uint32_t *nsym = nullptr;             // [C]
    uint32_t sym_cap = 0;
    float2 *stage = nullptr;
    std::vector<float> atan_host, mmse_host;
};

// gr::filter::mmse_fir_interpolator_ff's table, 129 phases x 8 taps: the taps minimising the mean squared interpolation
// error over |f| <= 1/4; computed here as the closed-form least-squares solution of that objective (normal equations
// R h = p with R_kl = 2B sinc(2B(k-l)), p_k = 2B sinc(2B(3 + mu - k)), B = 1/4), end rows exact unit impulses
inline std::vector<float> ref_design_mmse()
{
    auto sincpi = [](double x) { return std::fabs(x) < 1e-12 ? 1.0 : std::sin(M_PI * x) / (M_PI * x); };
    std::vector<float> tab(129 * 8);
    const double B = 0.25;
    for (int s = 0; s <= 128; s++) {
        const double mu = s / 128.0;
        double A[8][9];
        for (int k = 0; k < 8; k++) {
            for (int l = 0; l < 8; l++) A[k][l] = 2 * B * sincpi(2 * B * (k - l));
            A[k][k] += 1e-9;              // the sinc Gram matrix is ill-conditioned
            A[k][8] = 2 * B * sincpi(2 * B * (3.0 + mu - k));
        }
        for (int c = 0; c < 8; c++) {     // Gauss-Jordan, partial pivoting
            int p = c;
            for (int r = c + 1; r < 8; r++) if (std::fabs(A[r][c]) > std::fabs(A[p][c])) p = r;
            if (p != c) for (int j = 0; j < 9; j++) std::swap(A[c][j], A[p][j]);
            for (int r = 0; r < 8; r++) if (r != c) {
                const double f = A[r][c] / A[c][c];
                for (int j = c; j < 9; j++) A[r][j] -= f * A[c][j];
            }
        }
        for (int k = 0; k < 8; k++) tab[s * 8 + k] = (float)(A[k][8] / A[k][k]);
    }
    for (int k = 0; k < 8; k++) { tab[k] = (k == 3); tab[128 * 8 + k] = (k == 4); }
    return tab;
}

__device__ __forceinline__ float ref_fast_atan2f(float y, float x, const float *tab)
{
    const float ya = fabsf(y), xa = fabsf(x);
    if (!(ya > 0.0f || xa > 0.0f)) return 0.0f;
    const float z = ya < xa ? ya / xa : xa / ya;
    float base;
    if (z < 0.003921569f) base = z;
    else {
        float alpha = z * 255.0f;
        const int idx = ((int)alpha) & 0xff;
        alpha -= (float)idx;
        base = tab[idx] + (tab[idx + 1] - tab[idx]) * alpha;
    }
    if (xa > ya) {
        if (x >= 0.0f) return y >= 0.0f ? base : -base;
        return y >= 0.0f ? 3.14159265358979f - base : base - 3.14159265358979f;
    }
    if (y >= 0.0f) return x >= 0.0f ? 1.5707963267949f - base : 1.5707963267949f + base;
    return x >= 0.0f ? -1.5707963267949f + base : -1.5707963267949f - base;
}

// d[c][REF_TAIL + n] = fast_atan2f(arg of x[n] conj(x[n-1]))
__global__ __launch_bounds__(256) void ref_demod_kernel(const float2 *iq, uint64_t ld, uint32_t nsamp, const float2 *last_iq,
                                                         const float *atan_tab, float *d, uint64_t dld)
{
    __shared__ float tab[258];
    for (int i = threadIdx.x; i < 258; i += 256) tab[i] = atan_tab[i];
    __syncthreads();
    const int c = blockIdx.y;
    const float2 *x = iq + (uint64_t)c * ld;
    for (uint32_t n = blockIdx.x * 256 + threadIdx.x; n < nsamp; n += gridDim.x * 256) {
        const float2 v = x[n], p = n ? x[n - 1] : last_iq[c];
        const float re = v.x * p.x + v.y * p.y, im = v.y * p.x - v.x * p.y;
        d[(uint64_t)c * dld + REF_TAIL + n] = 1.0f * ref_fast_atan2f(im, re, tab);
    }
}

// one lane per channel: the M&M loop over [REF_TAIL - tail, REF_TAIL + nsamp) of the channel's discriminator row
__global__ __launch_bounds__(64) void ref_mm_kernel(float *d, uint64_t dld, uint32_t nsamp, const float *mmse, RefMM *st, uint8_t *syms,
                                                     uint32_t sym_cap, uint32_t *nsym, uint32_t C, float omega_mid, float omega_lim,
                                                     float gain_omega, float gain_mu, const float2 *iq, uint64_t ld, float2 *last_iq)
{
    __shared__ float tab[129 * 8];
    for (int i = threadIdx.x; i < 129 * 8; i += 64) tab[i] = mmse[i];
    __syncthreads();
    const uint32_t c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    RefMM m = st[c];
    float *row = d + (uint64_t)c * dld + (REF_TAIL - m.tail);
    const uint32_t n = m.tail + nsamp;
    uint32_t ii = m.skip, oo = 0;
    uint8_t *out = syms + (uint64_t)c * sym_cap;
    if (n >= 8) {
        const uint32_t ni = n - 8;
        while (oo < sym_cap && ii < ni) {
            const int imu = (int)rintf(m.mu * 128.0f);
            const float *h = tab + 8 * imu;
            float y = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; k++) y += h[k] * row[ii + k];
            const float sl = m.last < 0.0f ? -1.0f : 1.0f, sy = y < 0.0f ? -1.0f : 1.0f;
            const float e = sl * y - sy * m.last;
            m.last = y;
            m.omega += gain_omega * e;
            float dv = m.omega - omega_mid;
            if (dv > omega_lim) dv = omega_lim; else if (dv < -omega_lim) dv = -omega_lim;
            m.omega = omega_mid + dv;
            m.mu += m.omega + gain_mu * e;
            const float fl = floorf(m.mu);
            ii += (uint32_t)(int)fl;
            m.mu -= fl;
            out[oo++] = y >= 0.0f ? 1 : 0;          // binary_slicer_fb
        }
    }
    // what the loop did not consume waits in front of the next push (ii may have stepped past the data by < 11)
    // (source index - destination index = nsamp >= 0: an ascending copy never overwrites what it still has to read)
    // < 8 + 11 <= REF_TAIL when the loop stopped at the end of the data (it stops within 8 samples of it and steps < 11).  If it
    // stopped because the symbol row is full instead (cannot happen while omega is held within 0.5 % of 10 and sym_cap =
    // max/9 + 16, but nothing else enforces it), only the newest REF_TAIL samples can be carried: the older ones are dropped
    // rather than written in front of the row.
    uint32_t keep = ii < n ? n - ii : 0;
    if (keep > (uint32_t)REF_TAIL) { ii += keep - (uint32_t)REF_TAIL; keep = (uint32_t)REF_TAIL; }
    float *base = d + (uint64_t)c * dld;
    for (uint32_t k = 0; k < keep; k++) base[REF_TAIL - keep + k] = row[ii + k];
    m.tail = keep;
    m.skip = ii > n ? ii - n : 0;                     // the last step may land up to two samples beyond the data
    st[c] = m;
    nsym[c] = oo;
    if (nsamp) last_iq[c] = iq[(uint64_t)c * ld + nsamp - 1];
}

inline void ref_destroy(RefState &r)
{
    void *bufs[] = { r.atan_tab, r.mmse, r.last_iq, r.mm, r.d, r.syms, r.nsym, r.stage };
    for (void *p : bufs) if (p) (void)hipFree(p);
    r = RefState();
}

inline int ref_reset(RefState &r, hipStream_t s)
{
    if (!r.ready) return 0;
    if (hipMemsetAsync(r.last_iq, 0, sizeof(float2) * r.C, s) != hipSuccess) return -EIO;
    std::vector<RefMM> init(r.C, RefMM{ 0.0f, 10.0f, 0.0f, 0u, 0u });   // mu 0, omega 10, last_sample 0 (grc/recctest.grc:846-874)
    if (hipMemcpyAsync(r.mm, init.data(), sizeof(RefMM) * r.C, hipMemcpyHostToDevice, s) != hipSuccess) return -EIO;
    if (hipStreamSynchronize(s) != hipSuccess) return -EIO;
    return 0;
}

inline int ref_create(RefState &r, uint32_t C, uint32_t max_samples, hipStream_t s)
{
    r.C = C; r.max_samples = max_samples;
    r.sym_cap = max_samples / 9 + 16;                 // omega is held within 0.5 % of 10 samples per symbol
    r.atan_host.resize(258);
    for (int i = 0; i < 258; i++) r.atan_host[i] = (float)std::atan((double)i / 255.0);
    r.mmse_host = ref_design_mmse();
    const uint64_t dld = (uint64_t)REF_TAIL + max_samples;
    if (hipMalloc((void **)&r.atan_tab, sizeof(float) * 258) != hipSuccess || hipMalloc((void **)&r.mmse, sizeof(float) * 129 * 8) != hipSuccess ||
        hipMalloc((void **)&r.last_iq, sizeof(float2) * C) != hipSuccess || hipMalloc((void **)&r.mm, sizeof(RefMM) * C) != hipSuccess ||
        hipMalloc((void **)&r.d, sizeof(float) * C * dld) != hipSuccess || hipMalloc((void **)&r.syms, (size_t)C * r.sym_cap) != hipSuccess ||
        hipMalloc((void **)&r.nsym, sizeof(uint32_t) * C) != hipSuccess) { ref_destroy(r); return -ENOMEM; }
    if (hipMemcpy(r.atan_tab, r.atan_host.data(), sizeof(float) * 258, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(r.mmse, r.mmse_host.data(), sizeof(float) * 129 * 8, hipMemcpyHostToDevice) != hipSuccess) { ref_destroy(r); return -EIO; }
    r.ready = true;
    return ref_reset(r, s);
}

// G2 -> G3 -> G4 on `nsamp` new samples of every channel (device pointer); symbols / counts stay on the device
inline int ref_run(RefState &r, const float2 *iq, uint64_t ld, uint32_t nsamp, hipStream_t s)
{
    const uint64_t dld = (uint64_t)REF_TAIL + r.max_samples;
    if (nsamp) hipLaunchKernelGGL(ref_demod_kernel, dim3(std::min<uint32_t>((nsamp + 255) / 256, 1024u), r.C), dim3(256), 0, s,
                                  iq, ld, nsamp, r.last_iq, r.atan_tab, r.d, dld);
    // grc/recctest.grc:846-874: omega 10, gain_omega 0.25 * 0.175^2 * 3, gain_mu 0.05, omega_relative_limit 0.005
    hipLaunchKernelGGL(ref_mm_kernel, dim3((r.C + 63) / 64), dim3(64), 0, s, r.d, dld, nsamp, r.mmse, r.mm, r.syms, r.sym_cap, r.nsym, r.C,
                       10.0f, 10.0f * 0.005f, 0.25f * 0.175f * 0.175f * 3.0f, 0.05f, iq, ld, r.last_iq);
    return hipGetLastError() == hipSuccess ? 0 : -EIO;
}

} // namespace amps
