// EXPERIMENT: how the dispatcher places one-per-CU workgroups (768 threads, 136 KB LDS) under a CU mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <tuple>
#include <algorithm>
__global__ void __launch_bounds__(768) big(unsigned long long *out)
{
    extern __shared__ char pad[];
    pad[threadIdx.x] = 0;
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long w0 = wall_clock64();
    while (__builtin_readcyclecounter() - t0 < 200000) { }
    if (threadIdx.x == 0) { out[3 * blockIdx.x] = hw | ((unsigned long long)(xcc & 15) << 32); out[3 * blockIdx.x + 1] = w0; out[3 * blockIdx.x + 2] = wall_clock64() + pad[1]; }
}
int main()
{
    unsigned long long *out; (void)hipMalloc(&out, 24 * 4096);
    (void)hipFuncSetAttribute((const void *)big, hipFuncAttributeMaxDynamicSharedMemorySize, 139264);
    std::vector<unsigned long long> h(3 * 4096);
    auto run = [&](const char *label, const uint32_t *mask, int wgs) {
        hipStream_t s;
        if (mask) { if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("%s: create failed\n", label); return; } }
        else (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(big, dim3(wgs), dim3(768), 139264, s, out);
        (void)hipStreamSynchronize(s);
        }
        (void)hipMemcpy(h.data(), out, 24 * wgs, hipMemcpyDeviceToHost);
        std::map<std::tuple<unsigned, unsigned, unsigned>, int> cus;
        unsigned long long tmin = ~0ull, tmax = 0, smax = 0;
        for (int b = 0; b < wgs; b++) {
            unsigned hw = (unsigned)h[3 * b], xcc = (unsigned)(h[3 * b] >> 32);
            cus[{ xcc, (hw >> 13) & 7, (hw >> 8) & 15 }]++;
            tmin = std::min(tmin, h[3 * b + 1]); smax = std::max(smax, h[3 * b + 1]); tmax = std::max(tmax, h[3 * b + 2]);
        }
        int mx = 0; for (auto &c : cus) mx = std::max(mx, c.second);
        int perse[8][4] = { { 0 } };
        for (auto &c : cus) perse[std::get<0>(c.first) & 7][std::get<1>(c.first) & 3] += c.second;
        printf("%s, %d WGs: %zu CUs used, max %d WGs on one CU, last start - first start %.1f us, span %.1f us; WGs per (xcc: se0 se1 se2 se3):", label, wgs, cus.size(), mx,
               (smax - tmin) / 100.0, (tmax - tmin) / 100.0);
        for (int x = 0; x < 8; x++) printf(" %d:%d,%d,%d,%d", x, perse[x][0], perse[x][1], perse[x][2], perse[x][3]);
        printf("\n");
        (void)hipStreamDestroy(s);
    };
    uint32_t m[8];
    run("no mask", nullptr, 256);
    run("no mask", nullptr, 240);
    for (int n : { 8, 16, 24, 32 }) {
        for (int w = 0; w < 8; w++) m[w] = 0;
        for (int i = 0; i < n; i++) m[i / 32] |= 1u << (i % 32);
        for (int w = 0; w < 8; w++) m[w] = ~m[w];
        char l[32]; snprintf(l, sizeof l, "all but 0..%d", n - 1); run(l, m, 256 - n);
    }
    // two CUs of the SAME shader engine per XCC: bits (se0, cu0) and (se0, cu1) = i in 0..7 and 32..39
    for (int w = 0; w < 8; w++) m[w] = 0xffffffffu;
    m[0] &= ~0xffu; m[1] &= ~0xffu;
    run("all but 0..7 and 32..39 (two CUs of SE0)", m, 240);
    // one CU of each of the four SEs: bits 0..31
    // four CUs of SE0: 0..7, 32..39, 64..71, 96..103
    for (int w = 0; w < 8; w++) m[w] = 0xffffffffu;
    m[0] &= ~0xffu; m[1] &= ~0xffu; m[2] &= ~0xffu; m[3] &= ~0xffu;
    run("all but four CUs of SE0", m, 224);
    return 0;
}
