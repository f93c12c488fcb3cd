// recc_symbols.hip.h -- device replica of gr::amps::recc_impl::work (lib/recc_impl.cc:93-145),
// one 256-thread workgroup per RECC channel, all channels of a push in one launch.
//
// This is the exact drop-in seam: the per-channel 64 KiB symbol buffer, its length and the pending
// trigger pointer (lib/recc_impl.h:31-43) live in HBM and are mutated exactly as the reference
// mutates them, including the behaviours SURVEY.md 8a lists as Q1-Q4 (strict '>', search only the
// last n+73 symbols and only when nothing is pending, the post-capture tail shuffle, the wrap that
// keeps buf[61440..65536) and forgets a pending trigger).  Only the two data-parallel pieces are
// restructured for the GPU: memmem becomes a bit-domain search (the window packed to two bitmaps while it is appended, 32
// candidate offsets per funnel shift + and, LDS atomicMin for "first match"), and the overlapping memmove a copy through
// registers with every load in flight before the first store; all copies move 16 bytes per lane and instruction.
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

typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(1)));   // 16 bytes at any address: gfx950 global memory takes unaligned vector accesses

// 16 symbol bytes -> 16 bits of "byte == 1" and of "byte == 0" (a byte outside {0,1} is neither: it can never be part of a trigger)
__device__ __forceinline__ void pack16(const u4u v, uint32_t &ones, uint32_t &zeros)
{
    ones = 0u; zeros = 0u;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t x = v[d];
        const uint32_t lsb = x & 0x01010101u;
        const uint32_t hi = x & 0xfefefefeu;                              // non-binary bytes have a bit here
        uint32_t nb = hi | (hi >> 1); nb |= nb >> 2; nb |= nb >> 4;      // bit 0 of every byte: byte > 1
        const uint32_t o = lsb & ~nb, z = ~lsb & ~nb & 0x01010101u;
        // gather bits 0, 8, 16, 24 into a nibble
        const uint32_t on = (o | (o >> 7) | (o >> 14) | (o >> 21)) & 0xfu, zn = (z | (z >> 7) | (z >> 14) | (z >> 21)) & 0xfu;
        ones |= on << (4 * d);
        zeros |= zn << (4 * d);
    }
}

// 256 lanes: the trigger search in the bit domain.  After the append the search window (the last n + 73 symbols,
// lib/recc_impl.cc:115) exists twice in LDS as bitmaps -- symbol == 1 and symbol == 0 -- so a lane tests 32 candidate offsets per
// instruction pair (funnel shift + and) instead of comparing 74 bytes per candidate; "first match" (memmem) is an LDS atomicMin.
__global__ __launch_bounds__(256, 2) void recc_symbols_kernel(SymbolsArgs a)
{
    constexpr uint32_t BUFSZ = AMPS_RECC_SYMBUF, WIN = AMPS_RECC_WINDOW;
    constexpr uint32_t T = AMPS_RECC_TRIGGER_SYMS, CAP = AMPS_RECC_CAPTURE_SYMS;
    constexpr int GMAX = (AMPS_RECC_MAX_WORK_ITEMS + T - 1 + 31) / 32;    // 32-symbol groups of the largest search window (1923)
    constexpr int GU = (GMAX + 255) / 256;                                // groups per lane (8)
    constexpr int MU = (BUFSZ / 16 + 255) / 256;                          // 16-byte chunks per lane of the largest tail move (16)
    static_assert(WIN == 256 * 16, "the wrap copy is one 16-byte chunk per lane");
    __shared__ uint32_t s_one[GMAX + 4], s_zero[GMAX + 4];
    __shared__ uint32_t s_first;
    __shared__ uint32_t s_slot;
    const int c = blockIdx.x, tid = threadIdx.x;
    uint8_t *buf = a.symbuf + (uint64_t)c * BUFSZ;
    const uint8_t *in = a.syms + (uint64_t)c * a.ld;
    uint32_t len = a.len[c];
    int32_t cur = a.curstart[c];
    const uint32_t n = (uint32_t)a.n;

    if (tid == 0) s_first = 0xffffffffu;
    // :104-108 wrap -- source [61440,65536) and destination [0,4096) are disjoint
    if (len + n > BUFSZ) {
        *(uint4 *)(buf + 16 * tid) = *(const uint4 *)(buf + (BUFSZ - WIN) + 16 * tid);
        len = WIN;
        cur = -1;
        __syncthreads();
    }
    // :110-111 append, and -- when a search will run -- the bitmaps of the search window in the same pass.  The window is the
    // last searchsz symbols of the buffer after the append (:115); its first h symbols were in the buffer already
    const uint32_t len_old = len;
    len += n;
    const bool search = len > T && cur < 0;                             // :114, :117
    const uint32_t searchsz = len < n + T - 1 ? len : n + T - 1;      // :115
    const uint32_t lo0 = len - searchsz, h = len_old - lo0;           // h <= T - 1
    const uint32_t G = (searchsz + 31) / 32;
    // whole groups inside the new symbols: 32 bytes per lane and step, every load of a batch in flight before its first store
    constexpr int GB = 4;                                               // groups per lane and batch (8 x 16 bytes in flight)
    static_assert(GU % GB == 0, "batches");
#pragma unroll 1
    for (int u0 = 0; u0 < GU; u0 += GB) {
        u4u va[GB], vb[GB];
#pragma unroll
        for (int u = 0; u < GB; u++) {
            const uint32_t w0 = 32u * ((uint32_t)tid + 256u * (uint32_t)(u0 + u));
            if (w0 >= h && w0 + 32 <= searchsz) {
                const uint8_t *src = in + (w0 - h);
                va[u] = *(const u4u *)src;
                vb[u] = *(const u4u *)(src + 16);
            }
        }
#pragma unroll
        for (int u = 0; u < GB; u++) {
            const uint32_t g = (uint32_t)tid + 256u * (uint32_t)(u0 + u), w0 = 32u * g;
            if (w0 >= h && w0 + 32 <= searchsz) {
                uint8_t *dst = buf + len_old + (w0 - h);
                *(u4u *)dst = va[u];
                *(u4u *)(dst + 16) = vb[u];
                if (search) {
                    uint32_t o0, z0, o1, z1;
                    pack16(va[u], o0, z0);
                    pack16(vb[u], o1, z1);
                    s_one[g] = o0 | (o1 << 16); s_zero[g] = z0 | (z1 << 16);
                }
            }
        }
    }
    // the groups at the two ends of the window (the first h <= 73 symbols were in the buffer already; the last group may be
    // partial): byte by byte, one lane each
    {
        const uint32_t nhead = (h + 31) / 32;                             // groups 0 .. nhead-1 touch old symbols (<= 3)
        uint32_t g = 0xffffffffu;
        if ((uint32_t)tid < nhead) g = (uint32_t)tid;
        else if (tid == 3 && G > nhead && 32u * G != searchsz) g = G - 1;    // a partial last group that is not a head group
        if (g != 0xffffffffu && g < G) {
            const uint32_t w0 = 32u * g;
            if (!(w0 >= h && w0 + 32 <= searchsz)) {
                uint32_t ones = 0u, zeros = 0u;
#pragma unroll 1
                for (uint32_t b = 0; b < 32; b++) {
                    const uint32_t w = w0 + b;
                    if (w >= searchsz) break;
                    uint8_t v;
                    if (w < h) v = buf[lo0 + w];
                    else { v = in[w - h]; buf[len_old + (w - h)] = v; }
                    ones |= (uint32_t)(v == 1) << b;
                    zeros |= (uint32_t)(v == 0) << b;
                }
                s_one[g] = ones; s_zero[g] = zeros;
            }
        }
        if (tid >= 4 && tid < 8) { s_one[G + tid - 4] = 0u; s_zero[G + tid - 4] = 0u; }      // read-ahead of the funnel shifts
    }
    __syncthreads();

    if (len > T) {                                                    // :114
        if (cur < 0) {                                                // :117-119 first match in the window
            const uint32_t ncand = searchsz - T + 1;                  // candidate offsets 0 .. searchsz - T of the window
#pragma unroll 1
            for (uint32_t g = (uint32_t)tid; 32u * g < ncand; g += 256u) {
                uint32_t o[4], z[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { o[k] = s_one[g + k]; z[k] = s_zero[g + k]; }
                auto tap = [&](int i) -> uint32_t {                   // candidates whose symbol i equals the trigger's
                    const uint32_t *x = ((i < 64 ? (TRIG_LO >> i) : (TRIG_HI >> (i - 64))) & 1ull) ? o : z;
                    return (i & 31) ? __builtin_amdgcn_alignbit(x[(i >> 5) + 1], x[i >> 5], (uint32_t)(i & 31)) : x[i >> 5];
                };
                uint32_t acc = ~0u;
#pragma unroll
                for (int i = (int)T - 16; i < (int)T; i++) acc &= tap(i);   // the word-sync end of the trigger first: noise rarely passes
                if (acc) {
#pragma unroll
                    for (int i = 0; i < (int)T - 16; i++) acc &= tap(i);
                    const uint32_t left = ncand - 32u * g;
                    if (left < 32u) acc &= (1u << left) - 1u;
                    if (acc) { atomicMin(&s_first, 32u * g + (uint32_t)__builtin_ctz(acc)); break; }   // the lane's groups ascend
                }
            }
            __syncthreads();
            if (s_first != 0xffffffffu) cur = (int32_t)(lo0 + s_first);
        }
        if (cur >= 0) {                                               // :121-139
            const uint32_t startoff = (uint32_t)cur;
            const uint32_t captured = len - startoff - T;             // :124
            if (captured > CAP) {                                     // :125 strict
                if (tid == 0) s_slot = atomicAdd(a.nbursts, 1u);
                // :129-134 memmove(buf, buf + captured + T, tomove) with tomove == startoff; the ranges may overlap, so the whole
                // tail is read into registers (every load in flight at once) before any of it is written
                const uint32_t src = captured + T, tomove = len - src;
                u4u mv[MU];
                uint8_t mt[16];
                const uint32_t nfull = tomove / 16u, ntail = tomove & 15u;
#pragma unroll
                for (int u = 0; u < MU; u++) {
                    const uint32_t j = (uint32_t)tid + 256u * u;
                    if (j < nfull) mv[u] = *(const u4u *)(buf + src + 16u * j);
                }
                if (tid == 255) {
#pragma unroll
                    for (int b = 0; b < 16; b++) mt[b] = (uint32_t)b < ntail ? buf[src + 16u * nfull + b] : (uint8_t)0;
                }
                __syncthreads();
                const uint32_t slot = s_slot;
                if (slot < a.cap) {
                    uint8_t *dst = a.bursts + (uint64_t)slot * CAP;   // :126 blob copy (before the move overwrites it)
                    const uint8_t *bsrc = buf + startoff + T;
                    constexpr uint32_t CF = CAP / 16u;                // 210 full chunks + 14 bytes
                    if ((uint32_t)tid < CF) *(u4u *)(dst + 16 * tid) = *(const u4u *)(bsrc + 16 * tid);
                    else if ((uint32_t)tid == CF) for (uint32_t b = 16u * CF; b < CAP; b++) dst[b] = bsrc[b];
                    if (tid == 0) a.burst_chan[slot] = (uint32_t)c;
                } else if (tid == 0) atomicOr(a.status, 8u);
                __syncthreads();
#pragma unroll
                for (int u = 0; u < MU; u++) {
                    const uint32_t j = (uint32_t)tid + 256u * u;
                    if (j < nfull) *(u4u *)(buf + 16u * j) = mv[u];
                }
                if (tid == 255) {
#pragma unroll
                    for (int b = 0; b < 16; b++) if ((uint32_t)b < ntail) buf[16u * nfull + b] = mt[b];
                }
                len -= tomove;                                        // :134
                cur = -1;                                             // :135
            }
        }
    }
    if (tid == 0) { a.len[c] = len; a.curstart[c] = cur; }
}

} // namespace amps
