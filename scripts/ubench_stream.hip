// ubench_stream.hip -- what HBM read bandwidth does the wave-stream access pattern allow on MI355X?
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/ubench_stream.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// A: plain grid-stride float4 read (sum) -- the linear-stream ceiling
__global__ __launch_bounds__(256) void k_linear(const float4 *p, size_t n4, float *out)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.f) out[0] = acc;
}
// B: one wave per (channel, chunk), 512-sample (4 KiB) tiles, DEPTH tiles in flight, WORK dummy fma per element
template <int DEPTH, int WORK>
__global__ __launch_bounds__(256) void k_wave(const float4 *p, size_t ld4, int tiles_per_chunk, int tiles_total, float *out)
{
    extern __shared__ float occupancy_limiter[];      // dynamic LDS only caps workgroups per CU
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.y, chunk = blockIdx.x * 4 + wv;
    if (ld4 == 1) occupancy_limiter[threadIdx.x] = 0.f;
    const int t0 = chunk * tiles_per_chunk;
    if (t0 >= tiles_total) return;
    int t1 = t0 + tiles_per_chunk; if (t1 > tiles_total) t1 = tiles_total;
    const float4 *base = p + (size_t)c * ld4 + lane;
    float4 buf[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
#pragma unroll
        for (int q = 0; q < 4; q++) buf[d][q] = base[(size_t)(t0 + d < t1 ? t0 + d : t0) * 256 + 64 * q];
    float acc = 0.f;
    for (int t = t0; t < t1; t++) {
        float4 cur[4];
#pragma unroll
        for (int q = 0; q < 4; q++) cur[q] = buf[0][q];
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; d++)
#pragma unroll
            for (int q = 0; q < 4; q++) buf[d][q] = buf[d + 1][q];
        if (t + DEPTH < t1) {
#pragma unroll
            for (int q = 0; q < 4; q++) buf[DEPTH - 1][q] = base[(size_t)(t + DEPTH) * 256 + 64 * q];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float x = cur[q].x, y = cur[q].y, z = cur[q].z, w = cur[q].w;
#pragma unroll
            for (int r = 0; r < WORK; r++) { x = __builtin_fmaf(x, y, z); y = __builtin_fmaf(y, z, w); z = __builtin_fmaf(z, w, x); w = __builtin_fmaf(w, x, y); }
            acc += x + y + z + w;
        }
    }
    if (acc == 12345.f) out[0] = acc;
}

// C: the same, but the four waves of a workgroup share one stream: wave w takes tile t0 + 4 i + w at step i (16 KiB contiguous
// per workgroup step); MODE 1 = every wave of the grid steps through the whole block together (tile i * W + g): the sweep order
template <int DEPTH, int WORK, int MODE, int NW = 4, bool BAR = false>
__global__ __launch_bounds__(64 * NW) void k_wg(const float4 *p, size_t ld4, int tiles_per_chunk, int tiles_total, float *out)
{
    extern __shared__ float occupancy_limiter[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (ld4 == 1) occupancy_limiter[threadIdx.x] = 0.f;
    size_t first, stride; int n;
    if (MODE == 0) {
        const int c = blockIdx.y;
        const int t0 = blockIdx.x * NW * tiles_per_chunk;
        if (t0 >= tiles_total) return;
        int t1 = t0 + NW * tiles_per_chunk; if (t1 > tiles_total) t1 = tiles_total;
        first = (size_t)c * ld4 + (size_t)(t0 + wv) * 256; stride = NW * 256; n = (t1 - t0 - wv + NW - 1) / NW;
        if (BAR) n = (t1 - t0) / NW;                                     // uniform step count (tiles_total is a multiple here)
    } else {
        const size_t W = (size_t)gridDim.x * gridDim.y * 4, g = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv;
        const size_t total = (size_t)gridDim.y * (ld4 / 256);            // all tiles of the block (rows are contiguous here)
        first = g * 256; stride = W * 256; n = (int)((total - g + W - 1) / W);
    }
    const float4 *base = p + first + lane;
    float4 buf[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
#pragma unroll
        for (int q = 0; q < 4; q++) buf[d][q] = base[(size_t)(d < n ? d : 0) * stride + 64 * q];
    float acc = 0.f;
    for (int t = 0; t < n; t++) {
        float4 cur[4];
#pragma unroll
        for (int q = 0; q < 4; q++) cur[q] = buf[0][q];
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; d++)
#pragma unroll
            for (int q = 0; q < 4; q++) buf[d][q] = buf[d + 1][q];
        if (t + DEPTH < n) {
#pragma unroll
            for (int q = 0; q < 4; q++) buf[DEPTH - 1][q] = base[(size_t)(t + DEPTH) * stride + 64 * q];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float x = cur[q].x, y = cur[q].y, z = cur[q].z, w = cur[q].w;
#pragma unroll
            for (int r = 0; r < WORK; r++) { x = __builtin_fmaf(x, y, z); y = __builtin_fmaf(y, z, w); z = __builtin_fmaf(z, w, x); w = __builtin_fmaf(w, x, y); }
            acc += x + y + z + w;
        }
        if (BAR) { occupancy_limiter[threadIdx.x] = acc; __syncthreads(); acc += occupancy_limiter[(threadIdx.x + 64) % (64 * NW)]; }
    }
    if (acc == 12345.f) out[0] = acc;
}

// D: sweep order with self-contained tiles: every wave also re-reads the 2 KiB in front of its tile (the neighbouring wave's data:
// L2 / MALL hits) and does 1.5x the arithmetic -- the shape of a streaming kernel whose tiles carry their own history
template <int WORK>
__global__ __launch_bounds__(256) void k_sweep_halo(const float4 *p, size_t n4, float *out)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t W = (size_t)gridDim.x * 4, g = (size_t)blockIdx.x * 4 + wv;
    const size_t total = n4 / 256;
    float acc = 0.f;
    float4 cur[6];
    auto load = [&](size_t t) {
        const float4 *b = p + t * 256 + lane;
#pragma unroll
        for (int q = 0; q < 4; q++) cur[q] = b[64 * q];
        const float4 *h = t ? b - 128 : b;
        cur[4] = h[0]; cur[5] = h[64];
    };
    if (g < total) load(g);
    for (size_t t = g; t < total; t += W) {
        float4 c[6];
#pragma unroll
        for (int q = 0; q < 6; q++) c[q] = cur[q];
        if (t + W < total) load(t + W);
#pragma unroll
        for (int q = 0; q < 6; q++) {
            float x = c[q].x, y = c[q].y, z = c[q].z, w = c[q].w;
#pragma unroll
            for (int r = 0; r < WORK; r++) { x = __builtin_fmaf(x, y, z); y = __builtin_fmaf(y, z, w); z = __builtin_fmaf(z, w, x); w = __builtin_fmaf(w, x, y); }
            acc += x + y + z + w;
        }
    }
    if (acc == 12345.f) out[0] = acc;
}

template <typename F> float time_ms(F f, int reps = 10)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main()
{
    const int C = 832; const size_t N = 1 << 18; const size_t ld4 = N / 2;     // float4 = 2 samples
    const size_t bytes = (size_t)C * N * 8;
    float4 *d; float *out; CK(hipMalloc(&d, bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(d, 0x11, bytes));
    const int tiles_total = N / 512;
    printf("bytes per pass %.3f GB\n", bytes / 1e9);
    float ms = time_ms([&] { hipLaunchKernelGGL(k_linear, dim3(256 * 8), dim3(256), 0, 0, d, bytes / 16, out); });
    printf("A linear float4 read            : %.3f ms  %.0f GB/s\n", ms, bytes / ms / 1e6);
    {
        // workgroup-stream with a barrier per step (the cooperative front-kernel shape): NW waves, waves/CU held at 16 and 12
        auto run = [&](auto kern, int nw, int waves_per_cu, int tpc, const char *name) {
            int nch = (tiles_total + nw * tpc - 1) / (nw * tpc);
            dim3 g(nch, C);
            size_t lds = 160 * 1024 / (waves_per_cu / nw) - 512; if (lds > 64 * 1024) lds = 64 * 1024;
            hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            float t = time_ms([&] { hipLaunchKernelGGL(kern, g, dim3(64 * nw), lds, 0, d, ld4, tpc, tiles_total, out); });
            printf("%s NW %2d waves/CU %2d tpc %3d : %.3f ms  %.0f GB/s\n", name, nw, waves_per_cu, tpc, t, bytes / t / 1e6);
        };
        for (int wpc : { 16, 12 }) {
            run(k_wg<1, 20, 0, 4, true>, 4, wpc, 64, "coop+barrier");
            run(k_wg<1, 20, 0, 8, true>, 8, wpc == 12 ? 16 : wpc, 32, "coop+barrier");
            run(k_wg<1, 20, 0, 16, true>, 16, 16, 16, "coop+barrier");
            run(k_wg<1, 20, 0, 4, false>, 4, wpc, 64, "coop        ");
            run(k_wg<1, 20, 0, 16, false>, 16, 16, 16, "coop        ");
        }
    }
    for (int wgs_per_cu : { 8, 4, 3 }) {
        ms = time_ms([&] { hipLaunchKernelGGL((k_sweep_halo<20>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, bytes / 16, out); });
        printf("sweep+halo(1.5x) waves/CU %2d : %.3f ms  %.0f GB/s useful\n", wgs_per_cu * 4, ms, bytes / ms / 1e6);
    }
    for (int wgs_per_cu : { 8, 4 }) {
        const int tpc = 52;
        int nch = (tiles_total + 4 * tpc - 1) / (4 * tpc);
        dim3 g(nch, C);
        size_t lds = 160 * 1024 / wgs_per_cu - 512;
        hipFuncSetAttribute((const void *)k_wg<1, 20, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void *)k_wg<2, 20, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void *)k_wg<1, 20, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void *)k_wg<2, 20, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        ms = time_ms([&] { hipLaunchKernelGGL((k_wg<1, 20, 0>), g, dim3(256), lds, 0, d, ld4, tpc, tiles_total, out); });
        printf("WG-stream waves/CU %2d depth1 : %.3f ms  %.0f GB/s\n", wgs_per_cu * 4, ms, bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL((k_wg<2, 20, 0>), g, dim3(256), lds, 0, d, ld4, tpc, tiles_total, out); });
        printf("WG-stream waves/CU %2d depth2 : %.3f ms  %.0f GB/s\n", wgs_per_cu * 4, ms, bytes / ms / 1e6);
        dim3 gs(256 * wgs_per_cu, 1);          // one resident round; MODE 1 needs total tiles: pass ld4 * C as one row
        ms = time_ms([&] { hipLaunchKernelGGL((k_wg<1, 20, 1>), gs, dim3(256), lds, 0, d, ld4 * C, tpc, tiles_total, out); });
        printf("sweep     waves/CU %2d depth1 : %.3f ms  %.0f GB/s\n", wgs_per_cu * 4, ms, bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL((k_wg<2, 20, 1>), gs, dim3(256), lds, 0, d, ld4 * C, tpc, tiles_total, out); });
        printf("sweep     waves/CU %2d depth2 : %.3f ms  %.0f GB/s\n", wgs_per_cu * 4, ms, bytes / ms / 1e6);
    }
    for (int wgs_per_cu : { 4, 3 }) {      // 4 waves per workgroup
        const int tpc = 52;
        int nch = (tiles_total + tpc - 1) / tpc;
        dim3 g((nch + 3) / 4, C);
        size_t lds = 160 * 1024 / wgs_per_cu - 512;
        hipFuncSetAttribute((const void *)k_wave<1, 20>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void *)k_wave<2, 20>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void *)k_wave<3, 20>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        ms = time_ms([&] { hipLaunchKernelGGL((k_wave<1, 20>), g, dim3(256), lds, 0, d, ld4, tpc, tiles_total, out); });
        printf("waves/CU %2d depth1 w20 : %.3f ms  %.0f GB/s\n", wgs_per_cu * 4, ms, bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL((k_wave<2, 20>), g, dim3(256), lds, 0, d, ld4, tpc, tiles_total, out); });
        printf("waves/CU %2d depth2 w20 : %.3f ms  %.0f GB/s\n", wgs_per_cu * 4, ms, bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL((k_wave<3, 20>), g, dim3(256), lds, 0, d, ld4, tpc, tiles_total, out); });
        printf("waves/CU %2d depth3 w20 : %.3f ms  %.0f GB/s\n", wgs_per_cu * 4, ms, bytes / ms / 1e6);
    }
    return 0;
}
