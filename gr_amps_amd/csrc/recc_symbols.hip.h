// recc_symbols.hip.h -- device replica of gr::amps::recc_impl::work (lib/recc_impl.cc:93-145),
// one 256-thread workgroup per RECC channel, all channels of a push in one launch.
//
// This is the exact drop-in seam: the per-channel 64 KiB symbol buffer, its length and the pending
// trigger pointer (lib/recc_impl.h:31-43) live in HBM and are mutated exactly as the reference
// mutates them, including the behaviours SURVEY.md 8a lists as Q1-Q4 (strict '>', search only the
// last n+73 symbols and only when nothing is pending, the post-capture tail shuffle, the wrap that
// keeps buf[61440..65536) and forgets a pending trigger).  Only the two data-parallel pieces are
// restructured for the GPU: memmem becomes 256 lanes testing candidate offsets with an LDS
// atomicMin for "first match", and the overlapping memmove becomes a chunked copy through registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "amps_recc.h"
#include "recc_front.hip.h" // TRIG_LO / TRIG_HI

namespace amps {

struct SymbolsArgs {
    const uint8_t *syms;   // [C][ld]
    uint64_t ld;
    int n;                 // noutput_items of this work() call
    uint8_t  *symbuf;      // [C][65536]
    uint32_t *len;         // [C]  d_symbuflen
    int32_t  *curstart;    // [C]  d_curstart as an offset, -1 == NULL
    uint8_t  *bursts;      // [cap][3374]
    uint32_t *burst_chan;  // [cap]
    uint32_t *nbursts;     // atomic
    uint32_t cap;
    uint32_t *status;      // bit 3: burst list overflow
};

__device__ __forceinline__ uint8_t trig_sym(int i)
{
    return (uint8_t)(((i < 64 ? (TRIG_LO >> i) : (TRIG_HI >> (i - 64))) & 1ull));
}

__global__ __launch_bounds__(256) void recc_symbols_kernel(SymbolsArgs a)
{
    constexpr uint32_t BUFSZ = AMPS_RECC_SYMBUF, WIN = AMPS_RECC_WINDOW;
    constexpr uint32_t T = AMPS_RECC_TRIGGER_SYMS, CAP = AMPS_RECC_CAPTURE_SYMS;
    __shared__ uint32_t s_first;
    __shared__ uint32_t s_slot;
    __shared__ uint8_t  s_trig[80];
    const int c = blockIdx.x, tid = threadIdx.x;
    uint8_t *buf = a.symbuf + (uint64_t)c * BUFSZ;
    const uint8_t *in = a.syms + (uint64_t)c * a.ld;
    uint32_t len = a.len[c];
    int32_t cur = a.curstart[c];
    const uint32_t n = (uint32_t)a.n;

    if (tid < (int)T) s_trig[tid] = trig_sym(tid);
    if (tid == 0) s_first = 0xffffffffu;

    // :104-108 wrap -- source [61440,65536) and destination [0,4096) are disjoint
    if (len + n > BUFSZ) {
        for (uint32_t i = tid; i < WIN; i += 256) buf[i] = buf[BUFSZ - WIN + i];
        len = WIN;
        cur = -1;
    }
    __syncthreads();
    // :110-111 append
    for (uint32_t i = tid; i < n; i += 256) buf[len + i] = in[i];
    len += n;
    __syncthreads();

    if (len > T) {                                                    // :114
        const uint32_t searchsz = len < n + T - 1 ? len : n + T - 1; // :115
        if (cur < 0) {                                                // :117-119 first match in the tail
            const uint32_t lo = len - searchsz, hi = len - T;         // candidate offsets lo..hi inclusive
            for (uint32_t p = lo + tid; p <= hi; p += 256) {
                if (p >= s_first) break;                               // a smaller offset already matched
                bool ok = true;
                for (uint32_t i = 0; i < T; i++) if (buf[p + i] != s_trig[i]) { ok = false; break; }
                if (ok) { atomicMin(&s_first, p); break; }
            }
            __syncthreads();
            if (s_first != 0xffffffffu) cur = (int32_t)s_first;
        }
        if (cur >= 0) {                                               // :121-139
            const uint32_t startoff = (uint32_t)cur;
            const uint32_t captured = len - startoff - T;             // :124
            if (captured > CAP) {                                     // :125 strict
                if (tid == 0) s_slot = atomicAdd(a.nbursts, 1u);
                __syncthreads();
                const uint32_t slot = s_slot;
                if (slot < a.cap) {
                    uint8_t *dst = a.bursts + (uint64_t)slot * CAP;   // :126 blob copy
                    for (uint32_t i = tid; i < CAP; i += 256) dst[i] = buf[startoff + T + i];
                    if (tid == 0) a.burst_chan[slot] = (uint32_t)c;
                } else if (tid == 0) atomicOr(a.status, 8u);
                __syncthreads();
                // :129-134 memmove(buf, buf + captured + T, tomove) with tomove == startoff; may overlap
                const uint32_t src = captured + T, tomove = len - src;
                for (uint32_t base = 0; base < tomove; base += 256 * 16) {
                    uint8_t r[16];
                    uint32_t o = base + tid * 16;
#pragma unroll
                    for (int j = 0; j < 16; j++) r[j] = (o + j < tomove) ? buf[src + o + j] : 0;
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < 16; j++) if (o + j < tomove) buf[o + j] = r[j];
                    __syncthreads();
                }
                len -= tomove;                                        // :134
                cur = -1;                                             // :135
            }
        }
    }
    if (tid == 0) { a.len[c] = len; a.curstart[c] = cur; }
}

} // namespace amps
