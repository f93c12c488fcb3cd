// ubench_pk2.hip -- issue rate of v_pk_fma_f32 with three DISTINCT register-pair sources (the fold's shape) on MI355X:
// chains = independent accumulators interleaved; the fold of the channelizer runs 4.  Pure VALU, no memory.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma_lo(f2 a, f2 c, f2 s) { f2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(c), "v"(s)); return r; }
__device__ __forceinline__ f2 fma_hi(f2 a, f2 c, f2 s) { f2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(c), "v"(s)); return r; }
template <int CH, int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s)
{
    f2 x[16], c[8], acc[CH];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = (f2){ threadIdx.x + i * 0.5f, threadIdx.x * 0.25f + i };
#pragma unroll
    for (int i = 0; i < 8; i++) c[i] = (f2){ s + i * 1e-3f, s - i * 1e-3f };
#pragma unroll
    for (int i = 0; i < CH; i++) acc[i] = (f2){ 0.f, 0.f };
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
#pragma unroll
            for (int ch = 0; ch < CH; ch++) {
                if (MODE == 0) acc[ch] = (q & 1) ? fma_hi(x[(q + 2 * ch) & 15], c[(q >> 1) + (ch & 3)], acc[ch]) : fma_lo(x[(q + 2 * ch) & 15], c[(q >> 1) + (ch & 3)], acc[ch]);
                else if (MODE == 1) acc[ch] = __builtin_elementwise_fma(x[(q + 2 * ch) & 15], c[(q >> 1) + (ch & 3)], acc[ch]);   // compiler's own form
                else if (MODE == 2) acc[ch] = acc[ch] + x[(q + 2 * ch) & 15];                                                     // v_pk_add, two sources
                else acc[ch] = __builtin_elementwise_fma(acc[ch], c[0], c[1]);                                                    // reused sources
            }
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < CH; i++) r += acc[i].x + acc[i].y;
    if (r == 12345.f) out[0] = r;
}
template <int CH, int MODE> void run(float *out, const char *name)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4000;
    for (int wpc : { 4, 8, 12, 16 }) {
        dim3 g(256 * wpc / 4);
        hipLaunchKernelGGL((k<CH, MODE>), g, dim3(256), 0, 0, out, iters, 0.999f); hipDeviceSynchronize();
        hipEventRecord(a); hipLaunchKernelGGL((k<CH, MODE>), g, dim3(256), 0, 0, out, iters, 0.999f); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-28s chains %d waves/SIMD %d : %.2f cycles@2.4GHz per wave-instr per SIMD\n", name, CH, wpc / 4,
               ms * 1e-3 * 2.4e9 / ((double)iters * 8 * CH * (wpc / 4.0)));
    }
}
int main()
{
    float *out; hipMalloc(&out, 4);
    run<4, 0>(out, "pk_fma asm 3 distinct srcs");
    run<8, 0>(out, "pk_fma asm 3 distinct srcs");
    run<4, 1>(out, "pk_fma compiler form");
    run<4, 2>(out, "pk_add 2 srcs");
    run<8, 2>(out, "pk_add 2 srcs");
    run<4, 3>(out, "pk_fma reused srcs");
    run<8, 3>(out, "pk_fma reused srcs");
    return 0;
}
