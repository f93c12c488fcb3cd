// EXPERIMENT: which physical (XCC, SE, CU) a bit of hipExtStreamCreateWithCUMask enables on this box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#include <tuple>
__global__ void probe(unsigned *out)
{
    __shared__ char pad[60000];           // one workgroup per CU at a time (LDS), so the grid spreads over every enabled CU
    pad[threadIdx.x] = 0;
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    long long t0 = clock64();
    while (clock64() - t0 < 200000) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc + pad[1]; }
}
int main(int argc, char **argv)
{
    unsigned *out; hipMalloc(&out, 8 * 4096);
    std::vector<unsigned> h(2 * 4096);
    auto run = [&](const char *label, const uint32_t *mask) {
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("%s: create failed\n", label); return; }
        hipMemsetAsync(out, 0xff, 8 * 4096, s);
        hipLaunchKernelGGL(probe, dim3(2048), dim3(64), 0, s, out);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), out, 8 * 4096, hipMemcpyDeviceToHost);
        std::set<std::tuple<unsigned, unsigned, unsigned, unsigned>> cus;
        for (int b = 0; b < 2048; b++) {
            unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            cus.insert({ xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15 });
        }
        printf("%s: %zu CUs:", label, cus.size());
        if (cus.size() <= 40) for (auto &c : cus) printf(" (x%u se%u sh%u cu%u)", std::get<0>(c), std::get<1>(c), std::get<2>(c), std::get<3>(c));
        else { unsigned per[8] = { 0 }; for (auto &c : cus) per[std::get<0>(c) & 7]++; printf(" per xcc:"); for (int x = 0; x < 8; x++) printf(" %u", per[x]); }
        printf("\n");
        hipStreamDestroy(s);
    };
    uint32_t m[8];
    for (int w = 0; w < 8; w++) m[w] = 0xffffffffu;
    run("all", m);
    for (int bit : { 0, 1, 7, 8, 9, 15, 16, 24, 31, 32, 40, 63, 64, 128, 200, 248, 255 }) {
        for (int w = 0; w < 8; w++) m[w] = 0; m[bit / 32] = 1u << (bit % 32);
        char l[32]; snprintf(l, sizeof l, "bit %d", bit); run(l, m);
    }
    for (int n : { 8, 16, 24, 32 }) {
        for (int w = 0; w < 8; w++) m[w] = 0;
        for (int i = 0; i < n; i++) m[i / 32] |= 1u << (i % 32);
        char l[32]; snprintf(l, sizeof l, "bits 0..%d", n - 1); run(l, m);
        for (int w = 0; w < 8; w++) m[w] = ~m[w];
        snprintf(l, sizeof l, "all but 0..%d", n - 1); run(l, m);
    }
    return 0;
}
