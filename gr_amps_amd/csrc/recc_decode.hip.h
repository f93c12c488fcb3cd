// recc_decode.hip.h -- device-side recc_decode core for gfx950 (one 64-lane wavefront per burst).
//
// Replaces, on the device, the per-burst work of the reference's recc_decode block:
//   manchester_decode_binbuf      lib/utils.cc:27-59
//   recc_bch_decode / itpp::BCH   lib/recc_decode_impl.cc:53-79 (BCH(63,51,t=2) shortened to (48,36))
//   bursts_message word loop      lib/recc_decode_impl.cc:96-107 (first valid repeat of 5)
//   recc_word_a/_b/_c/_called     lib/amps_packet.h:103-274, calc_min :277-302,354-363
//   dispatch                      lib/recc_decode_impl.cc:108-168
//
// Design notes (MI355X): a burst is 3374 symbol bytes -> 1687 bits -> 35 BCH blocks.  One wave
// handles one burst: the 64 lanes stride over the symbol pairs (Manchester), lanes 0..34 each
// decode one 48-bit block algebraically (syndromes S1,S3 in GF(64) as parities of constant masks,
// closed-form locator for t=2, 63-step incremental root search, no tables), lanes 0..6 pick the first valid repeat, and
// the record is written back cooperatively.  Everything stays in LDS (3.4 KB symbols + 1.7 KB
// bits); bursts are rare events (<= 1 per 34 480 samples per channel), so this kernel is latency-
// not bandwidth-critical and is kept simple.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "amps_recc.h"
#include "amps_recc_numerics.h"

namespace amps {

// ---- GF(64), primitive polynomial x^6 + x + 1 (the field IT++ uses for q = 64) ----
// Table-free on the device: a first version looked alpha^e / log up in a __constant__ table with per-lane indices
// (vector loads from constant memory inside divergent loops) and one burst took 51 us of pure latency; shift-and-xor
// arithmetic in registers plus parity-of-mask syndromes needs no memory at all.
struct Gf64Tables {
    uint8_t exp[128];
    uint8_t log[64];
};
constexpr Gf64Tables make_gf64()
{
    Gf64Tables t{};
    unsigned v = 1;
    for (int i = 0; i < 63; i++) {
        t.exp[i] = (uint8_t)v;
        t.exp[i + 63] = (uint8_t)v;
        t.log[v] = (uint8_t)i;
        v <<= 1;
        if (v & 0x40) v ^= 0x43;
    }
    t.exp[126] = t.exp[0];
    t.exp[127] = t.exp[1];
    t.log[0] = 0;
    return t;
}
// syndrome masks: bit k of S_m = parity(w & mask[m][k]) where w holds the received polynomial (bit e = coefficient of
// x^e) and mask[m][k] collects the exponents e whose alpha^(m e) has bit k set   (m = 1, 3)
struct SynMasks { uint64_t m1[6], m3[6]; };
constexpr SynMasks make_syn_masks()
{
    SynMasks s{};
    const Gf64Tables t = make_gf64();
    for (int e = 0; e < 63; e++)
        for (int k = 0; k < 6; k++) {
            if ((t.exp[e] >> k) & 1) s.m1[k] |= 1ull << e;
            if ((t.exp[(3 * e) % 63] >> k) & 1) s.m3[k] |= 1ull << e;
        }
    return s;
}
static constexpr SynMasks k_syn = make_syn_masks();

__device__ __forceinline__ unsigned gf_xtime(unsigned a)      // a * alpha
{
    a <<= 1;
    return a ^ ((a & 0x40u) ? 0x43u : 0u);
}
__device__ __forceinline__ unsigned gf_mul(unsigned a, unsigned b)
{
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) { r ^= ((b >> i) & 1u) ? a : 0u; a = gf_xtime(a); }
    return r;
}
__device__ __forceinline__ unsigned gf_inv(unsigned a)        // a^62, a != 0
{
    const unsigned a2 = gf_mul(a, a), a4 = gf_mul(a2, a2), a8 = gf_mul(a4, a4), a16 = gf_mul(a8, a8), a32 = gf_mul(a16, a16);
    return gf_mul(gf_mul(gf_mul(a2, a4), gf_mul(a8, a16)), a32);
}
__device__ __forceinline__ unsigned gf_div(unsigned a, unsigned b) { return gf_mul(a, gf_inv(b)); }   // b != 0

// Result of decoding one 48-bit block: ok + up to 3 error exponents (coefficient of x^e flips).
struct BchResult {
    int ok;
    int nflip;
    int e[3];
};

// w: received polynomial, bit e = coefficient of x^e (the shortening zeros are the bits above the block length).
// Semantics = IT++ BCH(63,2,true)::decode: syndromes, two Berlekamp steps (closed form for t = 2),
// root search over all 63 positions (in the order j = 0..62, root alpha^j <-> exponent (63 - j) % 63), failure iff
// #roots != deg(Lambda).  Roots in the 15 padding positions are NOT rejected (the reference does not check them,
// SURVEY.md 8a R4).
__device__ __forceinline__ BchResult bch63_decode_packed(uint64_t w)
{
    BchResult r;
    r.ok = 0; r.nflip = 0; r.e[0] = r.e[1] = r.e[2] = -1;
    unsigned S1 = 0, S3 = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        S1 |= (unsigned)(__popcll(w & k_syn.m1[k]) & 1) << k;
        S3 |= (unsigned)(__popcll(w & k_syn.m3[k]) & 1) << k;
    }
    if ((S1 | S3) == 0) { r.ok = 1; return r; }
    // the root searches keep their (up to three) results in named registers: an array indexed by the running count lands in
    // scratch memory (260 scratch instructions in the first version of this kernel, each a trip to HBM on the latency path)
    int found = 0, e0 = -1, e1 = -1, e2 = -1;
    auto take = [&](bool root, int j) {
        const int pos = (63 - j) % 63;
        e0 = (root && found == 0) ? pos : e0;
        e1 = (root && found == 1) ? pos : e1;
        e2 = (root && found == 2) ? pos : e2;
        found += (root && found < 3) ? 1 : 0;
    };
    if (S1 != 0) {
        const unsigned S1cube = gf_mul(gf_mul(S1, S1), S1);
        const unsigned delta = S3 ^ S1cube;      // Omega[3] = S3 + S1*S2, S2 = S1^2
        if (delta == 0) {                        // Lambda = 1 + S1 x : single error at log(S1)
            unsigned p = 1;
            int lg = 0;
#pragma unroll 1
            for (int j = 0; j < 63; j++) { if (p == S1) lg = j; p = gf_xtime(p); }
            r.ok = 1; r.nflip = 1; r.e[0] = lg;
            return r;
        }
        const unsigned c2 = gf_div(delta, S1);   // Lambda = 1 + S1 x + (delta/S1) x^2
        unsigned a = S1, b = c2;                 // S1 alpha^j, c2 alpha^(2j)
#pragma unroll 1
        for (int j = 0; j < 63; j++) {
            take((1u ^ a ^ b) == 0, j);
            a = gf_xtime(a);
            b = gf_xtime(gf_xtime(b));
        }
        r.e[0] = e0; r.e[1] = e1; r.e[2] = e2;
        if (found == 2) { r.ok = 1; r.nflip = 2; }
        return r;
    }
    // S1 == 0, S3 != 0: first step leaves Lambda = 1 and T = x^2, second gives Lambda = 1 + S3 x^3
    {
        unsigned c = S3;                         // S3 alpha^(3j)
#pragma unroll 1
        for (int j = 0; j < 63; j++) {
            take((1u ^ c) == 0, j);
            c = gf_xtime(gf_xtime(gf_xtime(c)));
        }
        r.e[0] = e0; r.e[1] = e1; r.e[2] = e2;
        if (found == 3) { r.ok = 1; r.nflip = 3; }
    }
    return r;
}

// bits: nbits bytes (0/1), bit i is the coefficient of x^(nbits-1-i); the shortening zeros sit above x^(nbits-1).
__device__ __forceinline__ BchResult bch_short_decode(const uint8_t *bits, int nbits)
{
    uint64_t w = 0;
    for (int i = 0; i < nbits; i++) w |= (uint64_t)(bits[i] & 1u) << (nbits - 1 - i);
    return bch63_decode_packed(w);
}

__device__ __forceinline__ BchResult bch4836_decode(const uint8_t *bits) { return bch_short_decode(bits, 48); }

// the same for a block of 48 bit-bytes (0/1) at an 8-byte aligned address: six 64-bit reads, each packed to a byte by one
// multiply (first byte -> most significant bit), instead of 48 dependent one-byte LDS reads
__device__ __forceinline__ BchResult bch4836_decode_aligned(const uint8_t *bits)
{
    const uint64_t *p = (const uint64_t *)bits;
    uint64_t w = 0;
#pragma unroll
    for (int c = 0; c < 6; c++) w |= ((p[c] * 0x8040201008040201ull) >> 56) << (40 - 8 * c);
    return bch63_decode_packed(w);
}

// systematic encode of k message bits: parity = m(x) x^12 mod g(x), g = x^12+x^10+x^8+x^5+x^4+x^3+1
__device__ __forceinline__ void bch_short_encode(const uint8_t *msg, int k, uint8_t *cw)
{
    unsigned rem = 0;
    for (int j = 0; j < k; j++) {
        unsigned fb = ((rem >> 11) & 1u) ^ (msg[j] & 1u);
        rem = (rem << 1) & 0xfffu;
        if (fb) rem ^= 0x539u;          // g(x) without the x^12 term: 0b0101_0011_1001
        cw[j] = msg[j] & 1u;
    }
    for (int j = 0; j < 12; j++) cw[k + j] = (uint8_t)((rem >> (11 - j)) & 1u);
}

__device__ __forceinline__ unsigned getbits(const uint8_t *b, int n)
{
    unsigned v = 0;
    for (int i = 0; i < n; i++) v = (v << 1) | (b[i] & 1u);
    return v;
}

// lib/amps_packet.h:277-302 (quirks kept: dig>9 -> 0)
__device__ __forceinline__ void extract_min_3(unsigned val, char *out)
{
    unsigned m2 = val + 111;
    unsigned dig = m2 % 10;
    out[2] = (char)('0' + dig);
    if (dig == 0) m2 -= 10; else m2 -= dig;
    dig = (m2 % 100) / 10;
    out[1] = (char)('0' + dig);
    if (dig == 0) m2 -= 100; else m2 -= (m2 % 100);
    dig = m2 / 100;
    if (dig > 9) dig = 0;
    out[0] = (char)('0' + dig);
}

// LDS scratch one wave needs to decode a burst.  DecodeCore is everything behind the Manchester stage; the symbol bytes are only
// staged when the burst arrives as bytes (amps_recc_decode_bursts) -- a capture out of the slicer-bit ring goes from ring words
// to Manchester bits directly (recc_resolve.hip.h).
constexpr int BOFF = 1;                         // bit k of the burst lives at bits[BOFF + k]: the word blocks start at 8 + 240 w + 48 r
struct DecodeCore {
    alignas(8) uint8_t bits[1696];              // [BOFF + k], k < 7 + 7*240: dcc(7) then the words; BOFF makes every 48-bit block 8-byte aligned
    uint64_t packed[7];                         // the words the field parser reads, bit-reversed: bit 63-i = word bit i
    uint32_t bad[8];                            // [0]=dcc, [1+w]=word w
    uint32_t nonbin;
    int8_t   ok[35];
    int8_t   flip[35][3];
    int8_t   rr[8];                             // repeat whose bits become word_dec[w]
    int8_t   dly[AMPS_TRACK_BLOCKS];            // samples the sampling instants of tracking block b were moved by (capture from the bit ring)
    amps_recc_burst_t rec;                      // staged record (728 B), copied out coalesced
};
struct DecodeScratch {
    DecodeCore k;
    uint8_t sym[AMPS_RECC_CAPTURE_SYMS + 2];    // symbol bytes
};

// Synchronisation of the lanes that decode one burst: a whole (single-wave) workgroup, or one wave of a larger workgroup --
// LDS operations of one wave complete in issue order, so there only the compiler has to be kept from moving them.
struct BlockSync { static __device__ __forceinline__ void sync() { __syncthreads(); } };
struct WaveSync {
    static __device__ __forceinline__ void sync()
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

// clear the per-burst counters and the staged record (all 64 lanes of the wave; followed by a sync in the Manchester stage)
__device__ __forceinline__ void decode_core_begin(DecodeCore &s, int lane)
{
    if (lane < 8) s.bad[lane] = 0;
    if (lane == 8) s.nonbin = 0;
    for (int i = lane; i < (int)(sizeof(amps_recc_burst_t) / 4); i += 64) ((uint32_t *)&s.rec)[i] = 0;
}

// ---- Manchester decode, lib/utils.cc:27-59: (1,0) -> 0, (0,1) -> 1, (1,1) -> 0 + bad, (0,0) -> 1 + bad; a byte outside
// {0,1} is undefined in the reference (assert(0) compiled out): bit 0 + bad + flag.  Branch-free on one 16-bit read per
// pair so the 27 rounds pipeline (the four-way if/else with byte reads cost 5 us per burst).
template <class Sync>
__device__ __forceinline__ void manchester_from_sym(DecodeCore &s, const uint8_t *sym, int lane)
{
    decode_core_begin(s, lane);
    Sync::sync();
    const uint16_t *s16 = (const uint16_t *)sym;
#pragma unroll 9
    for (int it = 0; it < 27; it++) {
        const int k = lane + 64 * it;
        if (k < 1687) {
            const unsigned pr = s16[k];
            const unsigned sa = pr & 0xffu, sb = pr >> 8;
            const bool nonbin = (sa | sb) > 1u;
            const bool same = sa == sb;
            const unsigned bit = nonbin ? 0u : (same ? (sa ^ 1u) : sb);
            s.bits[BOFF + k] = (uint8_t)bit;
            if (nonbin) atomicOr(&s.nonbin, 1u);
            if (nonbin || same) atomicAdd(&s.bad[k < 7 ? 0 : 1 + (k - 7) / 240], 1u);
        }
    }
    Sync::sync();
}

// The same straight from slicer bits, with the capture's timing tracking (CPU model: capture() of
// oracle/fused_model.c).  Symbol i of the capture is bit nc + sps (i + 1) + dly(block of i) of the channel's stream; `ring` holds
// the stream's 64-bit words from word w0 on (LDS), starting early enough for the trigger in front of the capture.  The burst is
// walked in AMPS_TRACK_BLOCKS blocks of at most 55 bits, a lane per bit: the trigger's 37 bits (measured only), the coded DCC
// with the first repeat, then the other 34 repeats.  Every lane fetches the sps + 1 slicer bits from its pair's first sampling
// instant to the second; where the pair is a valid Manchester bit (a != b) the number of bits in between that still equal a says
// how late the mid-bit transition came.  The two polarities are summed separately with ballots per bit plane of that count
// (scalar population counts, no cross-lane adds) and the block moves everything behind it by a sample if
// mean(e | a = 1) + mean(e | a = 0) leaves [-1, 1].  track = false (AMPS_RECC_FLAG_FIXED_TIMING) never moves.
// Bits cannot be non-binary.
constexpr int TRACK_PRE_BITS = AMPS_RECC_TRIGGER_SYMS / 2;     // 37 bits of trigger in front of the capture
__host__ __device__ constexpr uint32_t capture_lead(uint32_t sps) { return sps * (AMPS_RECC_TRIGGER_SYMS - 1) + AMPS_TRACK_BLOCKS; }   // samples in front of n_c a capture reads
__device__ __forceinline__ uint64_t capture_first_word(uint64_t nc, uint32_t sps)
{
    const uint64_t lead = capture_lead(sps);
    return (nc > lead ? nc - lead : 0ull) >> 6;
}
// sum of v over the 64 lanes of the wave (DPP adds inside the rows, two row broadcasts, lane 63 read back): seven instructions, no LDS
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, true);    // row_ror:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);    // row_ror:8   -> every lane holds its row's sum
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// TWO = two samples per symbol (sps == 2, the wideband seam at D = 768), where the rule is another one (capture() of
// oracle/fused_model.c): ONE slicer bit lies between the two instants of a pair -- it says which side the transition was on, not how
// far -- so a block is taken at whichever of the delays d - 1, d, d + 1 (d = the block before) shows the fewest Manchester violations
// (a == b) in the block ITSELF; d wins a tie, then d - 1.  Same rounds, same one wave sum (three 8-bit counts), and the same trick
// for the chain: the window of block b + 1 is fetched TWO samples early at the delay block b starts from, so whatever block b decides
// (-1, 0, +1) the five bits block b + 1 looks at start 0, 1 or 2 bits into it.
template <class Sync, bool TWO = false>
__device__ __forceinline__ void manchester_from_ring(DecodeCore &s, const uint64_t *ring, uint64_t nc, uint64_t w0, uint32_t sps, int lane, bool track,
                                                     bool keep_delays = true)
{
    constexpr int EARLY = TWO ? 2 : 1;                             // samples in front of a pair's first instant that a lane's window starts
    decode_core_begin(s, lane);
    Sync::sync();
    const uint32_t *r32 = (const uint32_t *)ring;
    const int32_t base = (int32_t)(nc - (w0 << 6));              // bit offset of n_c inside the window (< 2^20)
    const uint32_t midmask = (1u << (sps - 1)) - 1u;
    // The loop is a chain: block b's delay depends on block b - 1's measurement, and one wave issues an instruction every ~5 cycles,
    // so what a round costs is its instruction count (first version: 190 instructions and an LDS round trip per round, 12 us per
    // burst).  So: (1) a lane's window of block b + 1 is fetched while block b is being measured, one sample EARLY at block b's
    // delay -- whatever block b decides (-1, 0, +1), the sps + 1 bits block b + 1 needs start 0, 1 or 2 bits into it -- and only the
    // two raw dwords are fetched, the funnel shift that aligns them waits for the round that uses them; (2) the four statistics of a
    // block travel in ONE wave sum (DPP adds); (3) the two irregular blocks (the trigger, the DCC + first repeat) are peeled off, so
    // that the 34 regular rounds carry no per-block selects.
    uint32_t f_lo = 0u, f_hi = 0u, f_sh = 0u;
    int dly = 0, dly_fetched = 0;
    uint32_t badacc = 0u;
    // slicer bits from one before the first sampling instant of bit k at delay d (k, d from the caller; lanes beyond the block idle)
    auto fetch = [&](int k, int d, bool on) {
        const uint32_t na = (uint32_t)(base + (int32_t)sps * (2 * k + 1) + d - EARLY);
        const uint32_t q = on ? na >> 5 : 0u;
        f_lo = r32[q]; f_hi = r32[q + 1]; f_sh = na & 31u;
    };
    auto measure = [&](uint32_t w, bool on) {                      // the round's timing decision from the lanes' windows
        const uint32_t sa = w & 1u, sb = (w >> sps) & 1u;
        const uint32_t mid = (w >> 1) & midmask;
        const uint32_t cnt = (uint32_t)__popc(sa ? mid : (~mid & midmask));           // bits between the two instants that still equal a
        // one wave sum for all four statistics: {sum of cnt, number of bits} of the falling pairs (a = 1) in the low half-word, of
        // the rising ones in the high half-word (cnt <= 11 and at most 55 bits: 10 + 6 bits each)
        const uint32_t contrib = (on && sa != sb) ? ((cnt | (1u << 10)) << (sa ? 0 : 16)) : 0u;
        const uint32_t tot = wave_sum_u32(contrib);
        const int sum1 = (int)(tot & 0x3ffu), n1 = (int)((tot >> 10) & 0x3fu), sum0 = (int)((tot >> 16) & 0x3ffu), n0 = (int)(tot >> 26);
        const int E1 = 2 * sum1 - (int)(sps - 1) * n1, E0 = 2 * sum0 - (int)(sps - 1) * n0;
        const int lhs = E1 * n0 + E0 * n1, rhs = 2 * n0 * n1;
        if (rhs > 0) dly += lhs > rhs ? 1 : lhs < -rhs ? -1 : 0;
    };
    // TWO: w starts one sample in front of the pair at the delay the block before was taken at; returns the block's own move
    auto choose2 = [&](uint32_t w, bool on) -> int {
        const uint32_t e = w ^ (w >> 2);                               // bit i clear: samples i and i + 2 are equal
        // (three ballots + scalar population counts instead of the one DPP wave sum: measured slower, 0.0355-0.0367 against 0.0349-0.0353 ms
        // per wideband step at D = 768 -- a VALU compare feeding the scalar unit waits longer than six v_add_dpp: profiles/EXPERIMENTS.md, round 6)
        const uint32_t contrib = on ? ((~e & 1u) | ((~e & 2u) << 7) | ((~e & 4u) << 14)) : 0u;   // violations at d - 1 | d << 8 | d + 1 << 16
        const uint32_t tot = wave_sum_u32(contrib);
        const uint32_t vm = tot & 0xffu, v0 = (tot >> 8) & 0xffu, vp = tot >> 16;
        int mv = 0;
        uint32_t best = v0;
        if (vm < v0) { mv = -1; best = vm; }
        if (vp < v0 && vp < best) mv = 1;
        return mv;
    };
    // ---- block 0: the 37 bits of the trigger, in front of the capture: measured only.  A trigger at the very start of a stream
    // puts the window of the leading lanes (partly) in front of the stream: an exact trigger by the one early bit of lane 0, which
    // is not used; a TOLERANT one (cfg.sync_tolerance) by up to its tolerated symbols.  What lies in front of the stream reads 1,
    // as in the trigger test and in the CPU model (gbit() of oracle/fused_model.c): s = the number of window bits in front of bit 0
    // (s > 0 used to read stream bit 0 unshifted)
    {
        const int k = -TRACK_PRE_BITS + lane;
        const bool on = lane < TRACK_PRE_BITS;
        const int32_t nas = base + (int32_t)sps * (2 * k + 1) - 1;
        const uint32_t na = nas < 0 ? 0u : (uint32_t)nas, q = on ? na >> 5 : 0u;
        const int32_t sh = -nas - 1;                                  // >= 0 where the window starts at or before bit 0
        const uint32_t w = nas >= 0 ? __builtin_amdgcn_alignbit(r32[q + 1], r32[q], na & 31u) >> 1
                                    : sh >= 32 ? ~0u : ((r32[0] << (uint32_t)sh) | ((1u << (uint32_t)sh) - 1u));
        fetch(lane, 0, lane < 7 + AMPS_RECC_WORD_BITS);           // block 1 = bits 0 .. 54
        if constexpr (TWO) {
            // the window from one sample in front of the pair (nas), ones in front of the stream
            const uint32_t w2 = nas >= 0 ? __builtin_amdgcn_alignbit(r32[q + 1], r32[q], na & 31u)
                                         : sh + 1 >= 32 ? ~0u : ((r32[0] << (uint32_t)(sh + 1)) | ((1u << (uint32_t)(sh + 1)) - 1u));
            if (track) dly += choose2(w2, on);
            if (lane == 0 && keep_delays) s.dly[0] = (int8_t)dly;
        } else {
        if (lane == 0 && keep_delays) s.dly[0] = 0;
        if (track) measure(w, on);
        }
    }
    // ---- block 1: the coded DCC (bits 0..6) and repeat 0 of word 0
    {
        const int k = lane;
        const bool on = lane < 7 + AMPS_RECC_WORD_BITS;
        uint32_t w = __builtin_amdgcn_alignbit(f_hi, f_lo, f_sh) >> (uint32_t)(dly + 1);
        const int d1 = dly;
        fetch(7 + AMPS_RECC_WORD_BITS + lane, d1, lane < AMPS_RECC_WORD_BITS);
        if constexpr (TWO) {                                       // w starts one sample in front of the pair at d1: the block picks its own delay
            const int mv = track ? choose2(w, on) : 0;
            dly += mv;
            w >>= (uint32_t)(mv + 1);
        }
        if (lane == 0 && keep_delays) s.dly[1] = (int8_t)dly;
        const uint32_t sa = w & 1u, sb = (w >> sps) & 1u;
        if (on) s.bits[BOFF + k] = (uint8_t)(sa == sb ? (sa ^ 1u) : sb);
        const uint64_t bad = __ballot(on && sa == sb);
        if (lane == 0) s.bad[0] = (uint32_t)__popcll(bad & 0x7full);
        badacc = (uint32_t)__popcll(bad >> 7);
        if constexpr (!TWO) { if (track) measure(w, on); }
        dly_fetched = d1;
    }
    // ---- blocks 2 .. 35: repeat r of word w (block = 1 + 5 w + r), 48 bits each
    const bool on48 = lane < AMPS_RECC_WORD_BITS;
    uint8_t *bitp = &s.bits[BOFF + 7 + AMPS_RECC_WORD_BITS + lane];
    int kn = 7 + 2 * AMPS_RECC_WORD_BITS + lane;                  // this lane's bit in the block that is fetched next
    int rep = 1, word = 0;
#pragma unroll 1
    for (int b = 2; b < AMPS_TRACK_BLOCKS; b++) {
        uint32_t w = __builtin_amdgcn_alignbit(f_hi, f_lo, f_sh) >> (uint32_t)(dly - dly_fetched + 1);
        dly_fetched = dly;
        fetch(kn, dly, on48 && b + 1 < AMPS_TRACK_BLOCKS);        // (the last round fetches nothing it uses: q = 0)
        kn += AMPS_RECC_WORD_BITS;
        if constexpr (TWO) {
            const int mv = track ? choose2(w, on48) : 0;
            dly += mv;
            w >>= (uint32_t)(mv + 1);
        }
        if (lane == 0 && keep_delays) s.dly[b] = (int8_t)dly;
        const uint32_t sa = w & 1u, sb = (w >> sps) & 1u;
        if (on48) *bitp = (uint8_t)(sa == sb ? (sa ^ 1u) : sb);
        bitp += AMPS_RECC_WORD_BITS;
        badacc += (uint32_t)__popcll(__ballot(on48 && sa == sb));
        if (++rep == AMPS_RECC_REPEATS) {                          // the word's five repeats are through
            if (lane == 0) s.bad[1 + word] = badacc;
            badacc = 0u; rep = 0; word++;
        }
        if constexpr (!TWO) { if (track) measure(w, on48); }
    }
    Sync::sync();
}

// BCH + first-valid-repeat + field parse + record write of the burst whose Manchester bits are in s.bits; all 64 lanes of ONE wave.
// coalesced copy of the staged record to HBM
__device__ __forceinline__ void decode_core_store(const DecodeCore &s, amps_recc_burst_t *__restrict__ out, int lane)
{
    for (int i = lane; i < (int)(sizeof(amps_recc_burst_t) / 4); i += 64) ((uint32_t *)out)[i] = ((const uint32_t *)&s.rec)[i];
}

// ---- the record on its way to the host (round 5).  The capture kernels write their records straight into mapped, pinned HOST memory;
// 588 of a record's 728 bytes are the one-byte-per-bit arrays word_raw[7][48] and word_dec[7][36] the reference's own layout asks for
// (lib/recc_decode_impl.cc:92-95).  1664 records per push of the channel-major bench are 1.2 MB of 728-byte PCIe writes that the kernel
// cannot retire before they have crossed the link: 0.050 ms of "resolve + capture + decode" there was mostly that.  So the bits travel
// as bits -- PACKED_RECORD_BYTES = 216 instead of 728 -- and amps_recc_drain expands them while it gathers the sorted records into the
// caller's buffer anyway (expand_packed_record: one 8-byte table entry per packed byte).  Layout, in dwords:
//    0 .. 12   the record's first 52 bytes as they are (channel .. first_valid_rep)
//   13 .. 23   word_raw: bit 8 j + i of dword 13 + g = byte 32 g + 4 j + i of the array (bytes past the array's 336: don't care)
//   24 .. 31   word_dec likewise (252 bytes)
//   32 .. 53   the record's last 88 bytes as they are (a_F .. _pad4)
constexpr int PACKED_RECORD_BYTES = 216;
constexpr int PACKED_BURST_BYTES = (AMPS_RECC_CAPTURE_SYMS + 31) / 32 * 4;   // 424: the kept 3374-symbol blob, a bit per symbol (recc_resolve.hip.h: capture_store_wave)
constexpr int REC_RAW_OFF = 52, REC_DEC_OFF = 388, REC_TAIL_OFF = 640;
static_assert(offsetof(amps_recc_burst_t, word_raw) == REC_RAW_OFF && offsetof(amps_recc_burst_t, word_dec) == REC_DEC_OFF &&
              offsetof(amps_recc_burst_t, a_F) == REC_TAIL_OFF && sizeof(amps_recc_burst_t) - REC_TAIL_OFF == 88, "packed record layout");
__device__ __forceinline__ void decode_core_store_packed(const DecodeCore &s, uint32_t *__restrict__ out, int lane)
{
    const uint32_t *rec = (const uint32_t *)&s.rec;
    if (lane < PACKED_RECORD_BYTES / 4) {
        uint32_t v;
        if (lane < 13) v = rec[lane];
        else if (lane >= 32) v = rec[REC_TAIL_OFF / 4 + (lane - 32)];
        else {
            const uint32_t *src = rec + (lane < 24 ? REC_RAW_OFF / 4 + 8 * (lane - 13) : REC_DEC_OFF / 4 + 8 * (lane - 24));
            v = 0u;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                // bytes are 0 / 1: the multiply gathers their low bits into bits 24 .. 27 (no two partial products share a bit position)
                // (the last group of either array runs a few bytes into the field behind it -- inside the record; those bits are don't-care)
                v |= ((((src[j] & 0x01010101u) * 0x01020408u) >> 24) & 0xfu) << (4 * j);
            }
        }
        out[lane] = v;
    }
}

// Tl: optional stage stamps (scripts/ubench_decode.hip); NoTl compiles to nothing.
struct NoTl { __device__ __forceinline__ void mark(int) {} };
template <class Sync, class Tl = NoTl>
__device__ __forceinline__ void decode_core_wave(DecodeCore &s, uint32_t channel, uint64_t position,
                                                 amps_recc_burst_t *__restrict__ out, bool majority, int lane, Tl tl = Tl())
{
    amps_recc_burst_t &o = s.rec;
    if (!majority) {
        // ---- BCH: 7 words x 5 repeats, lib/recc_decode_impl.cc:100-107 ----
        if (lane < 35) {
            const int w = lane / 5, r = lane % 5;
            BchResult br = bch4836_decode_aligned(&s.bits[BOFF + 7 + 240 * w + 48 * r]);
            s.ok[lane] = (int8_t)br.ok;
            s.flip[lane][0] = (int8_t)br.e[0];
            s.flip[lane][1] = (int8_t)br.e[1];
            s.flip[lane][2] = (int8_t)br.e[2];
        }
        Sync::sync();
        tl.mark(1);
        if (lane < 7) {
            const int w = lane;
            int r = 0, ok = 0;
            for (; r < 5; r++) if (s.ok[w * 5 + r]) { ok = 1; break; }
            o.valid[w] = (uint8_t)ok;
            o.first_valid_rep[w] = (uint8_t)r;
            o.manch_bad[w] = (uint16_t)s.bad[1 + w];
            s.rr[w] = (int8_t)(ok ? r : 4);
        }
        // raw repeat 0 of every word (what the reference parses)
        for (int i = lane; i < 7 * 48; i += 64) o.word_raw[i / 48][i % 48] = s.bits[BOFF + 7 + 240 * (i / 48) + (i % 48)];
        Sync::sync();
        tl.mark(2);
        // word_dec = the 36 message bits of the first valid repeat (or of repeat 4), copied by all lanes, then corrected
        for (int i = lane; i < 7 * 36; i += 64) {
            const int w = i / 36, b = i % 36;
            o.word_dec[w][b] = s.bits[BOFF + 7 + 240 * w + 48 * s.rr[w] + b];
        }
        Sync::sync();
        tl.mark(3);
        if (lane < 7 && o.valid[lane]) {
            const int w = lane;
            for (int f = 0; f < 3; f++) {
                int e = s.flip[w * 5 + s.rr[w]][f];
                if (e >= 12 && e <= 47) o.word_dec[w][47 - e] ^= 1u;
            }
        }
    } else {
        // ---- majority mode: bitwise 3-of-5 vote, one BCH decode per word, pad corrections rejected ----
        for (int i = lane; i < 7 * 48; i += 64) {
            const int w = i / 48, b = i % 48;
            int cnt = 0;
            for (int r = 0; r < 5; r++) cnt += s.bits[BOFF + 7 + 240 * w + 48 * r + b];
            o.word_raw[w][b] = (uint8_t)(cnt >= 3);
        }
        Sync::sync();
        if (lane < 7) {
            const int w = lane;
            BchResult br = bch4836_decode(o.word_raw[w]);
            int ok = br.ok;
            for (int f = 0; f < 3; f++) if (br.e[f] >= 48) ok = 0;          // a "correction" in the 15 shortening zeros
            o.valid[w] = (uint8_t)ok;
            int agree = 0;
            for (int r = 0; r < 5; r++) {
                // branch-free: an early exit from the unrolled comparison nested 48 saved EXEC masks per repeat -- the ~400 spilled
                // SGPRs of every kernel this function is inlined into (rounds 2-3)
                unsigned diff = 0;
#pragma unroll 8
                for (int b = 0; b < 48; b++) diff |= (unsigned)(s.bits[BOFF + 7 + 240 * w + 48 * r + b] ^ o.word_raw[w][b]);
                agree += diff == 0;
            }
            o.first_valid_rep[w] = (uint8_t)agree;
            o.manch_bad[w] = (uint16_t)s.bad[1 + w];
            for (int i = 0; i < 36; i++) o.word_dec[w][i] = o.word_raw[w][i];
            if (ok) for (int f = 0; f < 3; f++) { int e = br.e[f]; if (e >= 12 && e <= 47) o.word_dec[w][47 - e] ^= 1u; }
        }
    }
    if (lane < 7) o.dcc[lane] = s.bits[BOFF + lane];
    Sync::sync();
    tl.mark(4);
    // pack the seven words the parser reads (one ballot each): the field extraction below is then shifts on registers
    // instead of ~250 dependent one-byte LDS reads by a single lane (7 us of a 25 us burst)
    for (int w = 0; w < 7; w++) {
        const uint8_t *src = majority ? o.word_dec[w] : o.word_raw[w];
        const unsigned bit = lane < 36 ? (src[lane] & 1u) : 0u;   // every field lies inside the 36 message bits
        const uint64_t m = __ballot(bit != 0);
        if (lane == 0) s.packed[w] = __brevll(m);
    }
    Sync::sync();
    tl.mark(5);

    // ---- field parse + dispatch: one lane, negligible work (lib/amps_packet.h, recc_decode_impl.cc:108-168) ----
    if (lane == 0) {
        o.channel = channel;
        o.position = position;
        o.flags = s.nonbin ? AMPS_BURST_FLAG_NONBINARY : 0u;
        o.dcc_bad = (uint8_t)s.bad[0];
        // reference mode parses the raw repeat 0 (lib/recc_decode_impl.cc:112,117); majority mode the corrected word.
        // word_dec holds 36 bits = everything the field parsers read (the last 12 of the 48 are parity).
        bool used_ok = true;             // majority mode: every word the dispatch reads must have decoded
        auto W = [&](int w) -> uint64_t { used_ok = used_ok && o.valid[w]; return s.packed[w]; };
        auto fld = [](uint64_t wp, int off, int n) -> unsigned { return (unsigned)((wp >> (64 - off - n)) & ((1ull << n) - 1ull)); };   // MSB first
        const uint64_t A = W(0), B = W(1);
        if (majority) {   // coded DCC: 0000000 / 0011111 / 1100011 / 1111100, accept within one bit
            const unsigned d = getbits(o.dcc, 7);
            const bool good = __popc(d ^ 0x00u) <= 1 || __popc(d ^ 0x1fu) <= 1 || __popc(d ^ 0x63u) <= 1 || __popc(d ^ 0x7cu) <= 1;
            if (!good) o.flags |= AMPS_BURST_FLAG_DCC_INVALID;
        }
        o.a_F = (uint8_t)fld(A, 0, 1); o.a_NAWC = (uint8_t)fld(A, 1, 3);
        o.a_T = (uint8_t)fld(A, 4, 1); o.a_S = (uint8_t)fld(A, 5, 1); o.a_E = (uint8_t)fld(A, 6, 1); o.a_ER = (uint8_t)fld(A, 7, 1);
        o.a_SCM = (uint8_t)fld(A, 8, 4); o.a_MIN1 = fld(A, 12, 24);
        o.b_F = (uint8_t)fld(B, 0, 1); o.b_NAWC = (uint8_t)fld(B, 1, 3);
        o.b_MSG_TYPE = (uint8_t)fld(B, 4, 5); o.b_ORDQ = (uint8_t)fld(B, 9, 3);
        o.b_ORDER = (uint8_t)fld(B, 12, 5); o.b_LT = (uint8_t)fld(B, 17, 1); o.b_EP = (uint8_t)fld(B, 18, 1);
        o.b_SCM4 = (uint8_t)fld(B, 19, 1); o.b_MPCI = (uint8_t)fld(B, 20, 2); o.b_SDCC1 = (uint8_t)fld(B, 22, 2);
        o.b_SDCC2 = (uint8_t)fld(B, 24, 2); o.b_MIN2 = (uint16_t)fld(B, 26, 10);
        // calc_min, lib/amps_packet.h:354-363
        extract_min_3(o.b_MIN2, o.min);
        extract_min_3((o.a_MIN1 >> 14) & 0x3ff, o.min + 3);
        unsigned thous = (o.a_MIN1 >> 10) & 0xf;
        if (thous > 9) thous = 0;
        o.min[6] = (char)('0' + thous);
        extract_min_3(o.a_MIN1 & 0x3ff, o.min + 7);

        const bool zero_order = (o.b_ORDER == 0 && o.b_ORDQ == 0 && o.b_MSG_TYPE == 0);
        if (!o.valid[0]) o.msg_class = AMPS_MSG_INVALID_WORD_A;
        else if (!o.a_E) o.msg_class = AMPS_MSG_E_ZERO;
        else if (o.a_T == 0 && zero_order) o.msg_class = AMPS_MSG_PAGE_RESPONSE;
        else if (o.a_T == 1 && o.b_ORDER == 0xd) {
            o.msg_class = AMPS_MSG_REGISTRATION;
            o.has_esn = o.a_S;
            if (o.a_S && o.a_NAWC > 1) {
                const uint64_t Cw = W(2);
                o.esn = fld(Cw, 4, 32);
                uint8_t nawc = (uint8_t)(o.a_NAWC - 2);
                if ((uint8_t)fld(Cw, 1, 3) != nawc) o.flags |= AMPS_BURST_FLAG_WORDC_NAWC_MISMATCH;
            }
        } else if (o.a_T == 1 && (o.a_NAWC > 2 || zero_order)) {
            uint8_t nawc = o.a_NAWC;
            unsigned next = 2;
            o.has_esn = o.a_S;
            if (o.a_S) {
                const uint64_t Cw = W(next++);
                o.esn = fld(Cw, 4, 32);
                nawc = (uint8_t)(o.a_NAWC - 2);
                if ((uint8_t)fld(Cw, 1, 3) != nawc) o.flags |= AMPS_BURST_FLAG_WORDC_NAWC_MISMATCH;
            }
            if (nawc < 1 || nawc > 4) o.msg_class = AMPS_MSG_BAD_NAWC;
            else {
                o.msg_class = AMPS_MSG_ORIGINATION;
                int dl = 0;
                for (; nawc > 0; nawc--) {
                    unsigned digs = fld(W(next++), 4, 32);
                    for (int i = 0; i < 8; i++) {       // recc_word_called::digits(), amps_packet.h:211-273
                        unsigned v = (digs >> 28) & 0xf;
                        if (v == 0) break;
                        if (v >= 13) { o.flags |= AMPS_BURST_FLAG_BAD_DIGIT; break; }
                        o.dialed[dl++] = v <= 9 ? (char)('0' + v) : v == 10 ? '0' : v == 11 ? '*' : '#';
                        digs <<= 4;
                    }
                    o.n_called_words++;
                }
            }
        } else o.msg_class = AMPS_MSG_UNKNOWN;
        if (majority && !used_ok && o.msg_class >= AMPS_MSG_PAGE_RESPONSE) o.msg_class = AMPS_MSG_INVALID_WORD_A;
    }
    Sync::sync();
    tl.mark(6);
    if (out) decode_core_store(s, out, lane);
    tl.mark(7);
}

// Decode the burst held in s.sym; all 64 lanes of ONE single-wave workgroup must call this (blockDim.x == 64).
__device__ __forceinline__ void decode_burst_wave(DecodeScratch &s, uint32_t channel, uint64_t position,
                                                  amps_recc_burst_t *__restrict__ out, bool majority = false)
{
    const int lane = threadIdx.x & 63;
    manchester_from_sym<BlockSync>(s.k, s.sym, lane);
    decode_core_wave<BlockSync>(s.k, channel, position, out, majority, lane);
}

} // namespace amps
