// ubench_pk3.hip -- does the ORDER of v_pk_fma_f32 instructions (which consecutive instructions share a source pair) change the issue
// rate?  8 chains, 8 taps; coefficient pair shared by consecutive instructions or not; one asm block per tap pair like the fold.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define FL(d, x, c) "v_pk_fma_f32 %" #d ", %" #x ", %" #c ", %" #d " op_sel_hi:[1,0,1]\n\t"
#define FH(d, x, c) "v_pk_fma_f32 %" #d ", %" #x ", %" #c ", %" #d " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s)
{
    f2 x0[8], x1[8], c[4], acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { x0[i] = (f2){ threadIdx.x + i * 0.5f, threadIdx.x * 0.25f + i }; x1[i] = x0[i] * 1.5f; acc[i] = (f2){ 0.f, 0.f }; }
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = (f2){ s + i * 1e-3f, s - i * 1e-3f };
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (MODE == 0)      // the fold's order: chains 0..7, coefficient 26,27,24,25,24,25,26,27
                asm volatile(FL(0, 8, 26) FL(1, 9, 27) FL(2, 10, 24) FL(3, 11, 25) FL(4, 12, 24) FL(5, 13, 25) FL(6, 14, 26) FL(7, 15, 27)
                    FH(0, 16, 26) FH(1, 17, 27) FH(2, 18, 24) FH(3, 19, 25) FH(4, 20, 24) FH(5, 21, 25) FH(6, 22, 26) FH(7, 23, 27)
                    : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                    : "v"(x0[0]), "v"(x0[1]), "v"(x0[2]), "v"(x0[3]), "v"(x0[4]), "v"(x0[5]), "v"(x0[6]), "v"(x0[7]),
                      "v"(x1[0]), "v"(x1[1]), "v"(x1[2]), "v"(x1[3]), "v"(x1[4]), "v"(x1[5]), "v"(x1[6]), "v"(x1[7]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]));
            else if (MODE == 1) // consecutive instructions share the coefficient pair
                asm volatile(FL(2, 10, 24) FL(4, 12, 24) FL(3, 11, 25) FL(5, 13, 25) FL(0, 8, 26) FL(6, 14, 26) FL(1, 9, 27) FL(7, 15, 27)
                    FH(2, 18, 24) FH(4, 20, 24) FH(3, 19, 25) FH(5, 21, 25) FH(0, 16, 26) FH(6, 22, 26) FH(1, 17, 27) FH(7, 23, 27)
                    : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                    : "v"(x0[0]), "v"(x0[1]), "v"(x0[2]), "v"(x0[3]), "v"(x0[4]), "v"(x0[5]), "v"(x0[6]), "v"(x0[7]),
                      "v"(x1[0]), "v"(x1[1]), "v"(x1[2]), "v"(x1[3]), "v"(x1[4]), "v"(x1[5]), "v"(x1[6]), "v"(x1[7]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]));
            else if (MODE == 2) // all sixteen share ONE coefficient pair (upper bound of what sharing can give)
                asm volatile(FL(0, 8, 24) FL(1, 9, 24) FL(2, 10, 24) FL(3, 11, 24) FL(4, 12, 24) FL(5, 13, 24) FL(6, 14, 24) FL(7, 15, 24)
                    FH(0, 16, 24) FH(1, 17, 24) FH(2, 18, 24) FH(3, 19, 24) FH(4, 20, 24) FH(5, 21, 24) FH(6, 22, 24) FH(7, 23, 24)
                    : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                    : "v"(x0[0]), "v"(x0[1]), "v"(x0[2]), "v"(x0[3]), "v"(x0[4]), "v"(x0[5]), "v"(x0[6]), "v"(x0[7]),
                      "v"(x1[0]), "v"(x1[1]), "v"(x1[2]), "v"(x1[3]), "v"(x1[4]), "v"(x1[5]), "v"(x1[6]), "v"(x1[7]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]));
            else                // v_pk_mul + nothing: two-source packed op for reference
                asm volatile("v_pk_mul_f32 %0, %8, %24\n\tv_pk_mul_f32 %1, %9, %25\n\tv_pk_mul_f32 %2, %10, %26\n\tv_pk_mul_f32 %3, %11, %27\n\t"
                    "v_pk_mul_f32 %4, %12, %24\n\tv_pk_mul_f32 %5, %13, %25\n\tv_pk_mul_f32 %6, %14, %26\n\tv_pk_mul_f32 %7, %15, %27\n\t"
                    "v_pk_mul_f32 %0, %16, %24\n\tv_pk_mul_f32 %1, %17, %25\n\tv_pk_mul_f32 %2, %18, %26\n\tv_pk_mul_f32 %3, %19, %27\n\t"
                    "v_pk_mul_f32 %4, %20, %24\n\tv_pk_mul_f32 %5, %21, %25\n\tv_pk_mul_f32 %6, %22, %26\n\tv_pk_mul_f32 %7, %23, %27\n\t"
                    : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                    : "v"(x0[0]), "v"(x0[1]), "v"(x0[2]), "v"(x0[3]), "v"(x0[4]), "v"(x0[5]), "v"(x0[6]), "v"(x0[7]),
                      "v"(x1[0]), "v"(x1[1]), "v"(x1[2]), "v"(x1[3]), "v"(x1[4]), "v"(x1[5]), "v"(x1[6]), "v"(x1[7]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]));
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) r += acc[i].x + acc[i].y;
    if (r == 12345.f) out[0] = r;
}
template <int MODE> void run(float *out, const char *name)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 4000;
    for (int wpc : { 4, 8, 12 }) {
        dim3 g(256 * wpc / 4);
        hipLaunchKernelGGL((k<MODE>), g, dim3(256), 0, 0, out, iters, 0.999f); (void)hipDeviceSynchronize();
        (void)hipEventRecord(a); hipLaunchKernelGGL((k<MODE>), g, dim3(256), 0, 0, out, iters, 0.999f); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-34s waves/SIMD %d : %.2f cycles@2.4GHz per wave-instr per SIMD\n", name, wpc / 4, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * (wpc / 4.0)));
    }
}
int main()
{
    float *out; (void)hipMalloc(&out, 4);
    run<0>(out, "fold order");
    run<1>(out, "neighbours share coefficient");
    run<2>(out, "one coefficient for all");
    run<3>(out, "v_pk_mul two sources");
    return 0;
}
