// ubench_mfma_pass.hip -- the kill criterion of "pass 3 of the filter bank's FFT on the matrix pipe" (VERDICT r02, item 2).
// One 768-thread workgroup per CU, three roles of four waves as in chz12_kernel, one barrier per time step:
//   role 0: 190 packed-fp32 VALU instructions (the fold's instruction count per step), 8 independent chains
//   role 1: 141 packed VALU instructions (pass 2 + the cheap slicer)
//   role 2: EITHER 115 packed VALU instructions (pass 3 as the radix-16 butterfly network it is)
//           OR 64 v_mfma_f32_16x16x4_f32 + 30 packed VALU instructions (pass 3 as a 16 x 16 complex matrix product per 16 columns:
//              4 column tiles x 4 k-chunks x 4 real products, twiddles stay on the VALU)
// No LDS traffic, no loads: this is the BEST case for either form (issue and pipe contention only).  Prints cycles per step.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/ubench_mfma_pass.hip -o /tmp/ubm && /tmp/ubm
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void valu_block(f2 (&acc)[8], const f2 (&x)[8], const f2 (&c)[4])
{
#pragma unroll
    for (int i = 0; i < N; i++) acc[i & 7] = __builtin_elementwise_fma(x[(i * 3) & 7], c[i & 3], acc[i & 7]);
}

template <int MODE>   // 0: role 2 on the VALU, 1: role 2 on the matrix pipe, 2: role 2 idle, 3: only role 2 (matrix pipe alone)
__global__ __launch_bounds__(768, 3) void k(float *out, int steps)
{
    const int wave = threadIdx.x >> 6, role = wave >> 2, lane = threadIdx.x & 63;
    f2 acc[8], x[8], c[4];
#pragma unroll
    for (int i = 0; i < 8; i++) { acc[i] = (f2){ 0.f, 0.f }; x[i] = (f2){ 1.0f + lane * 1e-3f + i, 0.5f - i * 1e-2f }; }
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = (f2){ 1e-3f * (i + 1), -2e-3f * (i + 1) };
    f4 d[4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
    if (role == 0) __builtin_amdgcn_s_setprio(0); else if (role == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
    for (int s = 0; s < steps; s++) {
        if (role == 0) { if (MODE != 3) valu_block<190>(acc, x, c); }
        else if (role == 1) { if (MODE != 3) valu_block<141>(acc, x, c); }
        else if (MODE == 0) valu_block<115>(acc, x, c);
        else if (MODE == 1 || MODE == 3) {
            valu_block<30>(acc, x, c);
#pragma unroll
            for (int t = 0; t < 4; t++)                       // 4 column tiles, 16 MFMA each, accumulating into the tile's own registers
#pragma unroll
                for (int m = 0; m < 16; m++)
                    d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[m & 7].x, c[m & 3].y, d[t], 0, 0, 0);
        }
        __syncthreads();
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) r += acc[i].x + acc[i].y;
#pragma unroll
    for (int t = 0; t < 4; t++) r += d[t].x + d[t].y + d[t].z + d[t].w;
    if (r == 12345.678f) out[0] = r;
}

template <int MODE> static int run(const char *name, float *out)
{
    const int steps = 4000;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(768), 0, 0, out, 200);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(768), 0, 0, out, steps);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("%-44s %.3f ms for %d steps = %.0f ns per step (%.0f cycles at 2.4 GHz)\n", name, best, steps, best * 1e6 / steps, best * 1e6 / steps * 2.4);
    return 0;
}

int main()
{
    float *out; CK(hipMalloc(&out, 64));
    if (run<2>("fold + pass-2/slicer roles only", out)) return 1;
    if (run<0>("+ pass 3 on the VALU (115 packed ops)", out)) return 1;
    if (run<1>("+ pass 3 on the matrix pipe (64 MFMA + 30)", out)) return 1;
    if (run<3>("pass 3 on the matrix pipe alone", out)) return 1;
    return 0;
}
