// Stage timeline of one burst decode (recc_decode.hip.h) as the fused resolve kernel runs it: s_memtime stamps of one wave
// decoding a clean burst out of a bit ring at sps = 3, alone on the chip and with every CU loaded.
// build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Igr_amps_amd/csrc scripts/ubench_decode.hip -o /tmp/ubd && /tmp/ubd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "recc_resolve.hip.h"
using namespace amps;

struct Stamps {
    uint64_t t[10];
    __device__ __forceinline__ void mark(int k) { t[k] = __builtin_readcyclecounter(); }
};
struct StampRef {
    Stamps *p;
    __device__ __forceinline__ void mark(int k) { p->mark(k); }
};

__global__ __launch_bounds__(256, 4) void k_decode(const uint64_t *ring, uint32_t ring_words, uint32_t sps, amps_recc_burst_t *out, uint64_t *tl, int nwaves)
{
    extern __shared__ uint64_t s_cap[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wv >= nwaves) return;
    const uint32_t cap_words = resolve_cap_words(sps);
    uint64_t *scratch = s_cap + (size_t)wv * resolve_cap_stride(cap_words);
    DecodeCore &k = *(DecodeCore *)scratch;
    uint64_t *s_ring = scratch + (sizeof(DecodeCore) + 7) / 8;
    Stamps st;
    for (int i = 0; i < 10; i++) st.t[i] = 0;
    st.mark(8);
    const uint64_t nc = 512, w0 = capture_first_word(nc, sps);
    const int nw = (int)(((nc + (uint64_t)sps * (AMPS_RECC_CAPTURE_SYMS + 1) + AMPS_TRACK_BLOCKS + 32) >> 6) - w0) + 1;
    for (int i = lane; i < nw; i += 64) s_ring[i] = ring[(size_t)blockIdx.x * ring_words + w0 + i];
    WaveSync::sync();
    st.mark(9);
    manchester_from_ring<WaveSync>(k, s_ring, nc, w0, sps, lane, true);
    st.mark(0);
    StampRef ref{ &st };
    decode_core_wave<WaveSync>(k, blockIdx.x, nc, out + blockIdx.x * 4 + wv, false, lane, ref);
    if (blockIdx.x == 0 && wv == 0 && lane == 0) {
        tl[0] = st.t[9] - st.t[8];
        tl[1] = st.t[0] - st.t[9];
        for (int i = 1; i <= 7; i++) tl[1 + i] = st.t[i] - st.t[i - 1];
    }
}

int main()
{
    const uint32_t sps = 3, ring_words = 256;
    const int C = 832;
    std::vector<uint64_t> ring((size_t)C * ring_words, 0);
    // a valid-looking burst is not needed for the timing of the clean path: alternate bits -> Manchester pairs (1,0) -> zeros -> BCH syndromes 0
    for (int c = 0; c < C; c++)
        for (int i = 0; i < AMPS_RECC_CAPTURE_SYMS; i++) {
            const uint64_t n = 64 + (uint64_t)sps * (i + 1);
            if ((i & 1) == 0) ring[(size_t)c * ring_words + (n >> 6)] |= 1ull << (n & 63);
        }
    uint64_t *dring, *dtl; amps_recc_burst_t *dout;
    hipMalloc(&dring, ring.size() * 8); hipMalloc(&dtl, 16 * 8); hipMalloc(&dout, sizeof(amps_recc_burst_t) * C * 4);
    hipMemcpy(dring, ring.data(), ring.size() * 8, hipMemcpyHostToDevice);
    const size_t lds = resolve_dyn_lds(sps);
    const char *names[9] = { "ring load", "manchester", "bch", "valid+raw copy", "word_dec copy", "flip+dcc", "pack", "parse", "record copy" };
    for (int grid : { 1, 832 })
        for (int nwaves : { 1, 2, 4 }) {
            uint64_t tl[16];
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_decode, dim3(grid), dim3(256), lds, 0, dring, ring_words, sps, dout, dtl, nwaves);
                hipEventRecord(e1);
                hipDeviceSynchronize();
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(tl, dtl, sizeof(tl), hipMemcpyDeviceToHost);
            uint64_t tot = 0;
            printf("grid %4d, %d decoding waves per workgroup: kernel %.1f us |", grid, nwaves, ms * 1e3);
            for (int i = 0; i < 9; i++) { printf(" %s %llu", names[i], (unsigned long long)tl[i]); tot += tl[i]; }
            printf(" | total %llu ticks (100 MHz: %.1f us)\n", (unsigned long long)tot, tot / 100.0);
        }
    amps_recc_burst_t r;
    hipMemcpy(&r, dout, sizeof(r), hipMemcpyDeviceToHost);
    printf("valid[0..6] = %d %d %d %d %d %d %d\n", r.valid[0], r.valid[1], r.valid[2], r.valid[3], r.valid[4], r.valid[5], r.valid[6]);
    return 0;
}
