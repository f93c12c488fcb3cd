// ubench_pk.hip -- issue cost of v_pk_fma_f32 vs v_fma_f32 on MI355X (pure VALU, no memory)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 };
    f2 sv = { s, s * 0.5f };
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {           // 8 independent scalar fma chains
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a0 = __builtin_fmaf(a0, s, 1.0f); a1 = __builtin_fmaf(a1, s, 1.0f); a2 = __builtin_fmaf(a2, s, 1.0f); a3 = __builtin_fmaf(a3, s, 1.0f);
                a4 = __builtin_fmaf(a4, s, 1.0f); a5 = __builtin_fmaf(a5, s, 1.0f); a6 = __builtin_fmaf(a6, s, 1.0f); a7 = __builtin_fmaf(a7, s, 1.0f);
            }
        } else {                   // 4 independent packed chains = the same 8 fma per step
#pragma unroll
            for (int u = 0; u < 8; u++) {
                p0 = __builtin_elementwise_fma(p0, sv, (f2){ 1.0f, 1.0f }); p1 = __builtin_elementwise_fma(p1, sv, (f2){ 1.0f, 1.0f });
                p2 = __builtin_elementwise_fma(p2, sv, (f2){ 1.0f, 1.0f }); p3 = __builtin_elementwise_fma(p3, sv, (f2){ 1.0f, 1.0f });
            }
        }
    }
    float r = MODE == 0 ? a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 : p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (r == 12345.f) out[0] = r;
}
int main()
{
    float *out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    for (int wpc : { 4, 8, 16, 32 }) {      // waves per CU
        dim3 g(256 * wpc / 4);
        for (int mode = 0; mode < 2; mode++) {
            auto run = [&] { if (mode == 0) hipLaunchKernelGGL(k<0>, g, dim3(256), 0, 0, out, iters, 0.999f); else hipLaunchKernelGGL(k<1>, g, dim3(256), 0, 0, out, iters, 0.999f); };
            run(); hipDeviceSynchronize();
            hipEventRecord(a); run(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            double fma = (double)g.x * 256 * iters * 64.0;
            printf("waves/CU %2d  %s : %.3f ms  %.1f TFLOP/s  (%.2f cycles@2.4GHz per wave-instr per SIMD)\n", wpc, mode ? "v_pk_fma_f32" : "v_fma_f32   ",
                   ms, 2 * fma / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * (mode ? 32 : 64) * (wpc / 4.0)));
        }
    }
    return 0;
}
