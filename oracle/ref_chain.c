/* ref_chain.c -- TEST INFRASTRUCTURE ONLY (see amps_oracle.h).
 *
 * Own-words C restatement of the reference's RECC receive algorithm.  Every function cites the
 * reference file:line it follows (paths relative to the reference tree).  "parity unpinned" for
 * the IT++ and GNU Radio parts: their sources are not in the reference tree; what is restated is
 * their published algorithm as the reference's call sites use it.
 */
#include "amps_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ===================================================================================== R1 */

/* lib/recc_impl.cc:51-65 -- '0' -> symbols (1,0), '1' -> symbols (0,1) */
int orc_manchester_encode(const char *bits, size_t nbits, uint8_t *dst)
{
    for (size_t i = 0; i < nbits; i++) {
        if (bits[i] == '0') { dst[2 * i] = 1; dst[2 * i + 1] = 0; }
        else if (bits[i] == '1') { dst[2 * i] = 0; dst[2 * i + 1] = 1; }
        else return -1;
    }
    return 0;
}

/* lib/recc_impl.cc:76-79 -- 26 dotting bits + word sync 11100010010, Manchester coded: 74 symbols */
void orc_trigger(uint8_t dst[AMPS_RECC_TRIGGER_SYMS])
{
    static const char trig[] = "1010101010101010101010101011100010010";
    orc_manchester_encode(trig, sizeof(trig) - 1, dst);
}

/* ===================================================================================== R2 */

/* state of lib/recc_impl.h:31-43 */
struct orc_recc {
    uint8_t  buf[AMPS_RECC_SYMBUF]; /* d_symbuf                                   */
    size_t   len;                   /* d_symbuflen                                */
    int64_t  curstart;              /* d_curstart as an offset, -1 == NULL        */
    uint8_t  trig[AMPS_RECC_TRIGGER_SYMS];
};

orc_recc_t *orc_recc_new(void)
{
    orc_recc_t *s = (orc_recc_t *)calloc(1, sizeof(*s)); /* new unsigned char[n]() zero-fills, :75 */
    if (s) { orc_recc_reset(s); }
    return s;
}
void orc_recc_free(orc_recc_t *s) { free(s); }
void orc_recc_reset(orc_recc_t *s)
{
    memset(s->buf, 0, sizeof(s->buf));
    s->len = 0;
    s->curstart = -1;
    orc_trigger(s->trig);
}
void orc_recc_peek(const orc_recc_t *s, uint64_t *len, int64_t *curstart, const uint8_t **buf)
{
    if (len) *len = s->len;
    if (curstart) *curstart = s->curstart;
    if (buf) *buf = s->buf;
}

/* first occurrence of the 74-byte trigger inside hay[0..n), or -1 (glibc memmem semantics) */
static int64_t find_trigger(const uint8_t *hay, size_t n, const uint8_t *trig)
{
    if (n < AMPS_RECC_TRIGGER_SYMS) return -1;
    for (size_t p = 0; p + AMPS_RECC_TRIGGER_SYMS <= n; p++)
        if (hay[p] == trig[0] && memcmp(hay + p, trig, AMPS_RECC_TRIGGER_SYMS) == 0) return (int64_t)p;
    return -1;
}

/* lib/recc_impl.cc:93-145, including its stream quirks */
int orc_recc_work(orc_recc_t *s, const uint8_t *in, int n, uint8_t *burst_out)
{
    const size_t T = AMPS_RECC_TRIGGER_SYMS, CAP = AMPS_RECC_CAPTURE_SYMS;
    int published = 0;
    if (n < 1) return 0;                                   /* :99-102 */
    if (n > AMPS_RECC_MAX_WORK_ITEMS) return -1;           /* :103 assert (inert in Release; we refuse) */
    if (s->len + (size_t)n > AMPS_RECC_SYMBUF) {           /* :104-108 wrap: keeps buf[61440..65536), forgets a pending trigger */
        memmove(s->buf, s->buf + (AMPS_RECC_SYMBUF - AMPS_RECC_WINDOW), AMPS_RECC_WINDOW);
        s->len = AMPS_RECC_WINDOW;
        s->curstart = -1;
    }
    memmove(s->buf + s->len, in, (size_t)n);               /* :110-111 */
    s->len += (size_t)n;
    if (s->len > T) {                                      /* :114 */
        size_t searchsz = s->len < (size_t)n + T - 1 ? s->len : (size_t)n + T - 1; /* :115 */
        if (s->curstart < 0) {                             /* :117-119 search only when nothing is pending */
            int64_t rel = find_trigger(s->buf + (s->len - searchsz), searchsz, s->trig);
            if (rel >= 0) s->curstart = (int64_t)(s->len - searchsz) + rel;
        }
        if (s->curstart >= 0) {                            /* :121-139 */
            size_t startoff = (size_t)s->curstart;
            size_t captured = s->len - startoff - T;       /* :124 */
            if (captured > CAP) {                          /* :125 strict '>' (Q1) */
                memcpy(burst_out, s->buf + startoff + T, CAP); /* :126 blob copy */
                published = 1;
                size_t tomove = s->len - (captured + T);   /* :129 == startoff (Q3) */
                if (tomove > 0) memmove(s->buf, s->buf + (captured + T), tomove); /* :131-133 */
                s->len -= tomove;                          /* :134 */
                s->curstart = -1;                          /* :135 */
            }
        }
    }
    return published;
}

/* ===================================================================================== R3 */

/* lib/utils.cc:27-59.  The reference hits assert(0) (a no-op under NDEBUG, leaving the bit
 * uninitialised) for bytes outside {0,1}; here such a pair decodes to 0, counts as bad and is
 * reported through *nonbinary -- the one place this restatement defines what the reference leaves
 * undefined. */
size_t orc_manchester_decode_binbuf(const uint8_t *src, uint8_t *dst, size_t dstsz, int *nonbinary)
{
    size_t bad = 0;
    for (size_t o = 0; o < dstsz; o++) {
        unsigned sval = ((unsigned)src[2 * o] << 8) | src[2 * o + 1]; /* :33 */
        uint8_t bit;
        switch (sval) {
        case 0x101: bit = 0; bad++; break; /* :36-39 */
        case 0x000: bit = 1; bad++; break; /* :40-43 */
        case 0x100: bit = 0; break;        /* :44-46 */
        case 0x001: bit = 1; break;        /* :47-49 */
        default:    bit = 0; bad++; if (nonbinary) *nonbinary = 1; break; /* :50-52 assert(0) */
        }
        dst[o] = bit;
    }
    return bad;
}

/* ===================================================================================== R4 */
/* itpp::BCH(63, 2, true): n = 63, t = 2, systematic; GF(64) with primitive polynomial x^6 + x + 1
 * (IT++'s table entry for q = 64).  IT++ is not vendored in the reference and its version is not
 * pinned (CMakeLists.txt:89, find_package(ITPP) without a version); the algorithm restated here is
 * the one IT++ 4.3.x publishes in itpp/comm/bch.cpp: syndromes S1..S4, the t-step Berlekamp
 * iteration for binary BCH codes, a Chien-style root search over all 63 positions, and "decoder
 * failure" iff the number of roots found differs from deg(Lambda).  parity unpinned. */

static uint8_t gf_exp[126];
static int8_t  gf_log[64];
static int     gf_ready;

static void gf_init(void)
{
    if (gf_ready) return;
    unsigned v = 1;
    for (int i = 0; i < 63; i++) {
        gf_exp[i] = gf_exp[i + 63] = (uint8_t)v;
        gf_log[v] = (int8_t)i;
        v <<= 1;
        if (v & 0x40) v ^= 0x43; /* x^6 = x + 1 */
    }
    gf_log[0] = -1;
    gf_ready = 1;
}
static inline unsigned gf_mul(unsigned a, unsigned b) { return (a && b) ? gf_exp[gf_log[a] + gf_log[b]] : 0; }
static inline unsigned gf_div(unsigned a, unsigned b) { return a ? gf_exp[gf_log[a] + 63 - gf_log[b]] : 0; }
static inline unsigned gf_pow_alpha(int e) { e %= 63; if (e < 0) e += 63; return gf_exp[e]; }

/* generator g(x) = lcm of the minimal polynomials of alpha^1..alpha^4 = m1(x) * m3(x) */
uint32_t orc_bch_generator(void)
{
    gf_init();
    /* minimal polynomial of alpha^r: product over the conjugacy class (x - alpha^(r*2^i)) */
    uint32_t g = 1;
    for (int r = 1; r <= 3; r += 2) {
        unsigned poly[8] = { 1, 0, 0, 0, 0, 0, 0, 0 }; /* coefficients in GF(64), poly[i] = coeff of x^i */
        int deg = 0, e = r;
        do {
            unsigned root = gf_pow_alpha(e);
            for (int i = deg + 1; i > 0; i--) poly[i] = poly[i - 1] ^ gf_mul(poly[i], root);
            poly[0] = gf_mul(poly[0], root);
            deg++;
            e = (e * 2) % 63;
        } while (e != r);
        uint32_t m = 0;
        for (int i = 0; i <= deg; i++) if (poly[i]) m |= 1u << i; /* coefficients are 0/1 */
        /* multiply g by m over GF(2) */
        uint32_t prod = 0;
        for (int i = 0; i <= deg; i++) if (m & (1u << i)) prod ^= g << i;
        g = prod;
    }
    return g;
}

/* systematic encode: c(x) = m(x) x^12 + (m(x) x^12 mod g(x)); bit j of msg = coeff of x^(k-1-j) */
void orc_bch_encode_short(const uint8_t *msg, int k, uint8_t *cw)
{
    uint32_t g = orc_bch_generator();
    uint32_t rem = 0; /* 12-bit remainder register */
    for (int j = 0; j < k; j++) {
        unsigned fb = ((rem >> 11) & 1u) ^ (msg[j] & 1u);
        rem = (rem << 1) & 0xfffu;
        if (fb) rem ^= (g & 0xfffu);
        cw[j] = msg[j] & 1u;
    }
    for (int j = 0; j < 12; j++) cw[k + j] = (rem >> (11 - j)) & 1u;
}
void orc_bch63_encode(const uint8_t msg[51], uint8_t cw[63]) { orc_bch_encode_short(msg, 51, cw); }

/* polynomials over GF(64), small fixed degree */
typedef struct { unsigned c[8]; } gfx_t;
static int gfx_deg(const gfx_t *p) { for (int i = 7; i >= 0; i--) if (p->c[i]) return i; return -1; }

int orc_bch63_decode(const uint8_t rx[63], uint8_t corrected[63], int *nflips)
{
    gf_init();
    memcpy(corrected, rx, 63);
    if (nflips) *nflips = 0;
    /* syndromes S_j = r(alpha^j), j = 1..4; r_i (coefficient of x^i) is bit 62-i */
    unsigned S[5] = { 0, 0, 0, 0, 0 };
    for (int i = 0; i < 63; i++)
        if (rx[62 - i] & 1u)
            for (int j = 1; j <= 4; j++) S[j] ^= gf_pow_alpha(i * j);
    if (!(S[1] | S[2] | S[3] | S[4])) return 1; /* a codeword */
    /* Berlekamp iteration, kk = 0..t-1 (bch.cpp): Omega = Lambda*(1+S); delta = Omega[2kk+1];
       Lambda' = Lambda + delta*x*T; T' = x^2*T if delta==0 or deg(Lambda)>kk else x*Lambda/delta */
    gfx_t Lambda = { { 1 } }, T = { { 1 } };
    for (int kk = 0; kk < 2; kk++) {
        unsigned onepS[5] = { 1, S[1], S[2], S[3], S[4] };
        unsigned delta = 0;
        int want = 2 * kk + 1;
        for (int i = 0; i <= want && i < 8; i++)
            if (want - i <= 4) delta ^= gf_mul(Lambda.c[i], onepS[want - i]);
        gfx_t Old = Lambda;
        for (int i = 0; i < 7; i++) Lambda.c[i + 1] ^= gf_mul(delta, T.c[i]);
        if (delta == 0 || gfx_deg(&Old) > kk) {
            gfx_t nt = { { 0 } };
            for (int i = 0; i < 6; i++) nt.c[i + 2] = T.c[i];
            T = nt;
        } else {
            gfx_t nt = { { 0 } };
            for (int i = 0; i < 7; i++) nt.c[i + 1] = gf_div(Old.c[i], delta);
            T = nt;
        }
    }
    int deg = gfx_deg(&Lambda);
    int found = 0, pos[8];
    for (int j = 0; j < 63 && found < deg; j++) { /* roots alpha^j of Lambda; error position (63-j)%63 */
        unsigned v = 0;
        for (int i = 0; i <= deg; i++) v ^= gf_mul(Lambda.c[i], gf_pow_alpha(i * j));
        if (v == 0) pos[found++] = (63 - j) % 63;
    }
    if (found != deg) return 0; /* decoder failure: output stays uncorrected */
    for (int f = 0; f < found; f++) corrected[62 - pos[f]] ^= 1u;
    if (nflips) *nflips = found;
    return 1;
}

/* lib/recc_decode_impl.cc:53-79: 15 zeros ++ 48 received bits -> decode -> message bits 15..50.
 * The reference then copies final[0..47] although final has 36 elements (:71-77, out of range);
 * only the 36 defined bits are produced here.  On decoder failure IT++'s systematic branch returns
 * the uncorrected message bits. */
int orc_recc_bch_decode(const uint8_t src[48], uint8_t dst[36])
{
    uint8_t padded[63], corr[63];
    memset(padded, 0, 15);
    for (int i = 0; i < 48; i++) padded[15 + i] = src[i] & 1u;
    int ok = orc_bch63_decode(padded, corr, NULL);
    memcpy(dst, corr + 15, 36);
    return ok;
}

/* ===================================================================================== R6 / R7 */

static unsigned getbits(const uint8_t *buf, int bits) /* get8/get32/get64, lib/amps_packet.h:118-143 */
{
    unsigned v = 0;
    for (int i = 0; i < bits; i++) v = (v << 1) | (buf[i] & 1u);
    return v;
}

/* lib/amps_packet.h:277-302 */
static void extract_min_3(uint64_t val, char out[3])
{
    uint64_t m2 = val + 111;
    uint64_t dig = m2 % 10;
    out[2] = (char)('0' + dig);
    if (dig == 0) m2 -= 10; else m2 -= dig;
    dig = (m2 % 100) / 10;
    out[1] = (char)('0' + dig);
    if (dig == 0) m2 -= 100; else m2 -= (m2 % 100);
    dig = m2 / 100;
    if (dig > 9) dig = 0;
    out[0] = (char)('0' + dig);
}
/* lib/amps_packet.h:305-319 */
static uint64_t compute_min_3(char a, char b, char c)
{
    uint64_t d1 = (uint64_t)(a - '0'), d2 = (uint64_t)(b - '0'), d3 = (uint64_t)(c - '0');
    if (d1 == 0) d1 = 10;
    if (d2 == 0) d2 = 10;
    if (d3 == 0) d3 = 10;
    return 100 * d1 + 10 * d2 + d3 - 111;
}
/* lib/amps_packet.h:328-349 (the reference indexes min[0..9] even for shorter strings; 10 digits required here) */
int orc_parse_min(const char *min, uint64_t *min1, uint64_t *min2)
{
    size_t len = strlen(min);
    if (len < 1 || len > 10) return 0;
    for (size_t i = 0; i < len; i++) if (min[i] < '0' || min[i] > '9') return 0;
    if (len != 10) return 0;
    *min2 = compute_min_3(min[0], min[1], min[2]);
    uint64_t om1 = (compute_min_3(min[3], min[4], min[5]) & 0x3ff) << 14;
    uint64_t thous = (uint64_t)(min[6] - '0');
    if (thous == 0) thous = 10;
    om1 |= (thous & 0xf) << 10;
    om1 |= compute_min_3(min[7], min[8], min[9]) & 0x3ff;
    *min1 = om1;
    return 1;
}
/* lib/amps_packet.h:354-363 */
void orc_calc_min(uint64_t min1, uint64_t min2, char out[11])
{
    extract_min_3(min2, out);
    extract_min_3((min1 >> 14) & 0x3ff, out + 3);
    uint64_t thous = (min1 >> 10) & 0xf;
    if (thous > 9) thous = 0;
    out[6] = (char)('0' + thous);
    extract_min_3(min1 & 0x3ff, out + 7);
    out[10] = 0;
}
/* recc_word_called::digits(), lib/amps_packet.h:211-273 */
void orc_called_digits(uint32_t digits, char out[9], int *bad)
{
    int n = 0;
    for (int i = 0; i < 8; i++) {
        unsigned v = (digits >> 28) & 0xf;
        if (v == 0) break;
        if (v >= 13) { if (bad) *bad = 1; break; }
        out[n++] = v <= 9 ? (char)('0' + v) : v == 10 ? '0' : v == 11 ? '*' : '#';
        digits <<= 4;
    }
    out[n] = 0;
}
/* lib/utils.cc:101-108 */
void orc_expandbits(uint8_t *out, size_t nbits, uint64_t val)
{
    while (nbits > 0) { nbits--; out[nbits] = (uint8_t)(val & 1u); val >>= 1; }
}
/* lib/amps_packet.cc:26-32 */
void orc_focc_word1(uint8_t w[28], int multiword, unsigned dcc, uint64_t min1)
{
    w[0] = 0; w[1] = multiword ? 1 : 0; w[2] = (dcc >> 1) & 1u; w[3] = dcc & 1u;
    orc_expandbits(w + 4, 24, min1);
}
/* lib/amps_packet.cc:38-49 */
void orc_focc_word2_general(uint8_t w[28], uint64_t min2, unsigned msg_type, unsigned ordq, unsigned order)
{
    w[0] = 1; w[1] = 0; w[2] = 1; w[3] = 1;
    orc_expandbits(w + 4, 10, min2);
    w[14] = 0;
    orc_expandbits(w + 15, 5, msg_type);
    orc_expandbits(w + 20, 3, ordq);
    orc_expandbits(w + 23, 5, order);
}
/* lib/amps_packet.cc:55-76 */
void orc_fvc_word1_general(uint8_t w[28], unsigned pscc, unsigned msg_type, unsigned ordq, unsigned order)
{
    memset(w, 0, 28);
    w[0] = 1; w[1] = 0; w[2] = 1; w[3] = 1; w[4] = (pscc >> 1) & 1u; w[5] = pscc & 1u;
    orc_expandbits(w + 15, 5, msg_type);
    orc_expandbits(w + 20, 3, ordq);
    orc_expandbits(w + 23, 5, order);
}
/* lib/amps_packet.cc:82-95 */
void orc_focc_word2_voice_channel(uint8_t w[28], unsigned scc, uint64_t min2, unsigned vmac, unsigned chan)
{
    w[0] = 1; w[1] = 0; w[2] = (scc >> 1) & 1u; w[3] = scc & 1u;
    orc_expandbits(w + 4, 10, min2);
    w[14] = (vmac >> 2) & 1u; w[15] = (vmac >> 1) & 1u; w[16] = vmac & 1u;
    orc_expandbits(w + 17, 11, chan);
}

/* ===================================================================================== R5 / R8 */

/* lib/recc_decode_impl.cc:81-169 when majority == 0.  majority != 0 is NOT reference behaviour: it is the CPU
 * model of the product's optional "majority" decode mode (include/amps_recc.h "Decode modes"):
 * bitwise 3-of-5 vote per word, one BCH decode, corrections inside the 15 shortening zeros rejected, fields from
 * the corrected bits, every word the dispatch reads must be valid, coded DCC within one bit of a code word. */
void orc_decode_burst_mode(const uint8_t burst[AMPS_RECC_CAPTURE_SYMS], uint32_t channel, uint64_t position,
                           amps_recc_burst_t *o, int majority)
{
    uint8_t words[AMPS_RECC_WORDS][240];
    int nonbin = 0;
    memset(o, 0, sizeof(*o));
    o->channel = channel;
    o->position = position;
    o->dcc_bad = (uint8_t)orc_manchester_decode_binbuf(burst, o->dcc, 7, &nonbin);          /* :90 */
    for (int i = 0; i < AMPS_RECC_WORDS; i++)                                                   /* :96-99 */
        o->manch_bad[i] = (uint16_t)orc_manchester_decode_binbuf(burst + 14 + 480 * i, words[i], 240, &nonbin);
    for (int w = 0; w < AMPS_RECC_WORDS; w++) {
        if (!majority) {                                                                        /* :100-107 */
            uint8_t dec[36];
            int r, ok = 0;
            for (r = 0; r < AMPS_RECC_REPEATS; r++) {
                ok = orc_recc_bch_decode(&words[w][r * 48], dec);
                if (ok) break;
            }
            o->valid[w] = (uint8_t)ok;
            o->first_valid_rep[w] = (uint8_t)r; /* 5 when none decoded */
            memcpy(o->word_dec[w], dec, 36);    /* last attempt: the valid one, or uncorrected repeat 4 */
            memcpy(o->word_raw[w], words[w], 48);
        } else {
            uint8_t padded[63], corr[63];
            int nflip = 0, agree = 0;
            for (int b = 0; b < 48; b++) {
                int cnt = 0;
                for (int r = 0; r < AMPS_RECC_REPEATS; r++) cnt += words[w][r * 48 + b];
                o->word_raw[w][b] = (uint8_t)(cnt >= 3);
            }
            memset(padded, 0, 15);
            memcpy(padded + 15, o->word_raw[w], 48);
            int ok = orc_bch63_decode(padded, corr, &nflip);
            for (int b = 0; b < 15; b++) if (corr[b]) ok = 0;   /* "correction" inside the shortening zeros */
            o->valid[w] = (uint8_t)ok;
            for (int r = 0; r < AMPS_RECC_REPEATS; r++) agree += memcmp(&words[w][r * 48], o->word_raw[w], 48) == 0;
            o->first_valid_rep[w] = (uint8_t)agree;
            memcpy(o->word_dec[w], ok ? corr + 15 : o->word_raw[w], 36);
        }
    }
    if (nonbin) o->flags |= AMPS_BURST_FLAG_NONBINARY;
    if (majority) {
        static const unsigned codes[4] = { 0x00, 0x1f, 0x63, 0x7c };
        unsigned d = getbits(o->dcc, 7);
        int good = 0;
        for (int i = 0; i < 4; i++) if (__builtin_popcount(d ^ codes[i]) <= 1) good = 1;
        if (!good) o->flags |= AMPS_BURST_FLAG_DCC_INVALID;
    }

    /* reference: fields are always parsed from the raw repeat 0 (:112, :117), also for dropped bursts */
    int used_ok = 1;
#define WORD(w) (used_ok = used_ok && o->valid[(w)], majority ? o->word_dec[(w)] : o->word_raw[(w)])
    const uint8_t *A = WORD(0), *B = WORD(1);
    o->a_F = A[0] & 1u; o->a_NAWC = (uint8_t)getbits(A + 1, 3);                                 /* amps_packet.h:108-113 */
    o->a_T = A[4] & 1u; o->a_S = A[5] & 1u; o->a_E = A[6] & 1u; o->a_ER = A[7] & 1u;            /* :154-161 */
    o->a_SCM = (uint8_t)getbits(A + 8, 4); o->a_MIN1 = getbits(A + 12, 24);
    o->b_F = B[0] & 1u; o->b_NAWC = (uint8_t)getbits(B + 1, 3);
    o->b_MSG_TYPE = (uint8_t)getbits(B + 4, 5); o->b_ORDQ = (uint8_t)getbits(B + 9, 3);         /* :177-188 */
    o->b_ORDER = (uint8_t)getbits(B + 12, 5); o->b_LT = B[17] & 1u; o->b_EP = B[18] & 1u;
    o->b_SCM4 = B[19]; o->b_MPCI = (uint8_t)getbits(B + 20, 2); o->b_SDCC1 = (uint8_t)getbits(B + 22, 2);
    o->b_SDCC2 = (uint8_t)getbits(B + 24, 2); o->b_MIN2 = (uint16_t)getbits(B + 26, 10);
    orc_calc_min(o->a_MIN1, o->b_MIN2, o->min);

    int zero_order = (o->b_ORDER == 0 && o->b_ORDQ == 0 && o->b_MSG_TYPE == 0);
    if (!o->valid[0]) o->msg_class = AMPS_MSG_INVALID_WORD_A;                                   /* :108-111 */
    else if (!o->a_E) o->msg_class = AMPS_MSG_E_ZERO;                                           /* :113-116 */
    else if (o->a_T == 0 && zero_order) {                                                       /* :121-122 */
        o->msg_class = AMPS_MSG_PAGE_RESPONSE;
    } else if (o->a_T == 1 && o->b_ORDER == 0xd) {                                              /* :123-138 */
        o->msg_class = AMPS_MSG_REGISTRATION;
        o->has_esn = o->a_S;
        if (o->a_S && o->a_NAWC > 1) {
            const uint8_t *C = WORD(2);
            o->esn = getbits(C + 4, 32);
            uint8_t nawc = (uint8_t)(o->a_NAWC - 2);
            if ((uint8_t)getbits(C + 1, 3) != nawc) o->flags |= AMPS_BURST_FLAG_WORDC_NAWC_MISMATCH;
        }
    } else if (o->a_T == 1 && (o->a_NAWC > 2 || zero_order)) {                                  /* :139-165 */
        uint8_t nawc = o->a_NAWC;
        unsigned next = 2;
        o->has_esn = o->a_S;
        if (o->a_S) {
            const uint8_t *C = WORD(next); next++;
            o->esn = getbits(C + 4, 32);
            nawc = (uint8_t)(o->a_NAWC - 2); /* unsigned char arithmetic: wraps for NAWC < 2 */
            if ((uint8_t)getbits(C + 1, 3) != nawc) o->flags |= AMPS_BURST_FLAG_WORDC_NAWC_MISMATCH;
        }
        if (nawc < 1 || nawc > 4) o->msg_class = AMPS_MSG_BAD_NAWC;                             /* :155-158 */
        else {
            o->msg_class = AMPS_MSG_ORIGINATION;
            size_t dl = 0;
            for (; nawc > 0; nawc--) {
                char d[9]; int bad = 0;
                const uint8_t *Dw = WORD(next); next++;
                orc_called_digits(getbits(Dw + 4, 32), d, &bad);
                if (bad) o->flags |= AMPS_BURST_FLAG_BAD_DIGIT;
                size_t l = strlen(d);
                memcpy(o->dialed + dl, d, l); dl += l;
                o->n_called_words++;
            }
        }
    } else {
        o->msg_class = AMPS_MSG_UNKNOWN;                                                        /* :166-168 */
    }
#undef WORD
    if (majority && !used_ok && o->msg_class >= AMPS_MSG_PAGE_RESPONSE) o->msg_class = AMPS_MSG_INVALID_WORD_A;
}

void orc_decode_burst(const uint8_t burst[AMPS_RECC_CAPTURE_SYMS], uint32_t channel, uint64_t position,
                      amps_recc_burst_t *o)
{
    orc_decode_burst_mode(burst, channel, position, o, 0);
}

/* lib/recc_decode_impl.cc:181-272; GLOBAL_DCC_SHORT = 0, GLOBAL_SCC = 1 (amps_packet.h:13-14), STREAM_BOTH = 3 (:33) */
void orc_reply_words(const amps_recc_burst_t *b, amps_recc_reply_t *r)
{
    memset(r, 0, sizeof(*r));
    switch (b->msg_class) {
    case AMPS_MSG_REGISTRATION:                                                                 /* :181-190 */
        r->has_focc = 1; r->focc_stream = 3; r->focc_nwords = 2;
        orc_focc_word1(r->focc_word1, 1, 0, b->a_MIN1);
        orc_focc_word2_general(r->focc_word2, b->b_MIN2, 0, 0, 7);
        break;
    case AMPS_MSG_PAGE_RESPONSE:                                                                /* :195-222 */
        r->has_focc = 1; r->focc_stream = 3; r->focc_nwords = 2;
        orc_focc_word1(r->focc_word1, 1, 0, b->a_MIN1);
        orc_focc_word2_voice_channel(r->focc_word2, 1, b->b_MIN2, 0, 355);
        r->has_fvc = 1; r->fvc_count = 1; r->fvc_repeat = 35;
        orc_fvc_word1_general(r->fvc_word1, 1, 0, 0, 1);
        r->has_mutes = 1; r->fvc_mute = 0; r->audio_mute = 1;
        break;
    case AMPS_MSG_ORIGINATION:                                                                  /* :236-272 */
        r->has_focc = 1; r->focc_stream = 3; r->focc_nwords = 2;
        orc_focc_word1(r->focc_word1, 1, 0, b->a_MIN1);
        if (b->dialed[0] == '0') orc_focc_word2_general(r->focc_word2, b->b_MIN2, 0, 0, 9);
        else orc_focc_word2_voice_channel(r->focc_word2, 1, b->b_MIN2, 0, 356);
        r->has_mutes = 1; r->fvc_mute = 1; r->audio_mute = 0;
        r->has_command = 1;
        strcpy(r->command, "page ");
        strncat(r->command, b->dialed, sizeof(r->command) - 6);
        break;
    default: break;
    }
}

/* ===================================================================================== G1-G4 */
/* GNU Radio 3.7 blocks as wired in grc/recctest.grc.  GNU Radio is not part of the reference tree
 * (CMakeLists.txt:96 finds it on the system); these are own-words restatements of the blocks'
 * published behaviour.  parity unpinned. */

/* firdes.low_pass(gain, fs, cutoff, width, WIN_BLACKMAN): grc/recctest.grc:115-155 */
int orc_firdes_low_pass_blackman(double gain, double fs, double cutoff, double width, float *taps, int cap)
{
    int ntaps = (int)(74.0 * fs / (22.0 * width)); /* Blackman: 74 dB */
    if ((ntaps & 1) == 0) ntaps++;
    if (ntaps > cap) return -ntaps;
    int M = (ntaps - 1) / 2;
    double w0 = 2.0 * M_PI * cutoff / fs, sum = 0.0;
    double *t = (double *)malloc(sizeof(double) * (size_t)ntaps);
    for (int n = -M; n <= M; n++) {
        int i = n + M;
        double win = 0.42 - 0.5 * cos(2.0 * M_PI * i / (ntaps - 1)) + 0.08 * cos(4.0 * M_PI * i / (ntaps - 1));
        t[i] = (n == 0 ? w0 / M_PI : sin(n * w0) / (n * M_PI)) * win;
    }
    sum = t[M];
    for (int n = 1; n <= M; n++) sum += 2.0 * t[n + M];
    for (int i = 0; i < ntaps; i++) taps[i] = (float)(t[i] * gain / sum);
    free(t);
    return ntaps;
}

/* freq_xlating_fir_filter_ccc (grc/recctest.grc:889-937): composite taps h[i]*exp(j*i*phi),
 * phi = 2*pi*fc/fs; output k = rotator * sum_i ctaps[i] * x[k*decim - i]; rotator step
 * exp(-j*phi*decim), renormalised every 512 outputs. */
size_t orc_freq_xlating_fir(const float *in, size_t n_in, const float *taps, int ntaps,
                            double fc, double fs, int decim, float *out)
{
    double phi = 2.0 * M_PI * fc / fs;
    float *cr = (float *)malloc(sizeof(float) * (size_t)ntaps), *ci = (float *)malloc(sizeof(float) * (size_t)ntaps);
    for (int i = 0; i < ntaps; i++) { cr[i] = (float)(taps[i] * cos(i * phi)); ci[i] = (float)(taps[i] * sin(i * phi)); }
    float pr = 1.0f, pi_ = 0.0f;
    float ir = (float)cos(-phi * decim), ii = (float)sin(-phi * decim);
    size_t nout = n_in / (size_t)decim;
    unsigned counter = 0;
    for (size_t k = 0; k < nout; k++) {
        float ar = 0.0f, ai = 0.0f;
        size_t base = k * (size_t)decim;
        for (int i = 0; i < ntaps; i++) {
            if ((size_t)i > base) break; /* history before the stream is zero */
            float xr = in[2 * (base - i)], xi = in[2 * (base - i) + 1];
            ar += cr[i] * xr - ci[i] * xi;
            ai += cr[i] * xi + ci[i] * xr;
        }
        counter++;
        out[2 * k] = ar * pr - ai * pi_;
        out[2 * k + 1] = ar * pi_ + ai * pr;
        float npr = pr * ir - pi_ * ii, npi = pr * ii + pi_ * ir;
        pr = npr; pi_ = npi;
        if ((counter % 512) == 0) { float m = sqrtf(pr * pr + pi_ * pi_); pr /= m; pi_ /= m; }
    }
    free(cr); free(ci);
    return nout;
}

/* gr::fast_atan2f: 255-step table of atan on [0,1] + linear interpolation + octant unfolding */
static float atan_tab[258];
static int atan_tab_ready;
float orc_fast_atan2f(float y, float x)
{
    if (!atan_tab_ready) { for (int i = 0; i < 258; i++) atan_tab[i] = (float)atan((double)i / 255.0); atan_tab_ready = 1; }
    float ya = fabsf(y), xa = fabsf(x);
    if (!(ya > 0.0f || xa > 0.0f)) return 0.0f;
    float z = ya < xa ? ya / xa : xa / ya;
    float base;
    if (z < 0.003921569f) base = z;
    else {
        float alpha = z * 255.0f;
        int idx = ((int)alpha) & 0xff;
        alpha -= (float)idx;
        base = atan_tab[idx] + (atan_tab[idx + 1] - atan_tab[idx]) * alpha;
    }
    float ang;
    if (xa > ya) {
        if (x >= 0.0f) ang = y >= 0.0f ? base : -base;
        else ang = y >= 0.0f ? 3.14159265358979f - base : base - 3.14159265358979f;
    } else {
        if (y >= 0.0f) ang = x >= 0.0f ? 1.5707963267949f - base : 1.5707963267949f + base;
        else ang = x >= 0.0f ? -1.5707963267949f + base : -1.5707963267949f - base;
    }
    return ang;
}

/* analog.quadrature_demod_cf(gain) (grc/recctest.grc:458): gain * fast_atan2f(arg of x[n]*conj(x[n-1])) */
void orc_quadrature_demod(const float *iq, size_t n, float gain, float *out)
{
    float pr = 0.0f, pi_ = 0.0f;
    for (size_t i = 0; i < n; i++) {
        float xr = iq[2 * i], xi = iq[2 * i + 1];
        float re = xr * pr + xi * pi_, im = xi * pr - xr * pi_;
        out[i] = gain * orc_fast_atan2f(im, re);
        pr = xr; pi_ = xi;
    }
}

/* mmse_fir_interpolator_ff: 8 taps x 129 phases.  GNU Radio ships a precomputed table minimising
 * the mean squared error over |f| <= 0.25; restated here as the closed-form least-squares solution
 * of the same objective (normal equations R h = p, R_kl = sinc(2B(k-l)), p_k = sinc(2B(3+mu-k))). */
static float mmse_tab[129][8];
static int mmse_ready;
static double sincpi(double x) { return fabs(x) < 1e-12 ? 1.0 : sin(M_PI * x) / (M_PI * x); }
const float *orc_mmse_taps(void)
{
    if (mmse_ready) return &mmse_tab[0][0];
    const double B = 0.25;
    for (int s = 0; s <= 128; s++) {
        double mu = s / 128.0, A[8][9];
        for (int k = 0; k < 8; k++) {
            for (int l = 0; l < 8; l++) A[k][l] = 2 * B * sincpi(2 * B * (k - l));
            A[k][k] += 1e-9; /* the sinc Gram matrix is ill-conditioned; tiny ridge */
            A[k][8] = 2 * B * sincpi(2 * B * (3.0 + mu - k));
        }
        for (int c = 0; c < 8; c++) { /* Gauss-Jordan with partial pivoting */
            int p = c;
            for (int r = c + 1; r < 8; r++) if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
            if (p != c) for (int j = 0; j < 9; j++) { double t = A[c][j]; A[c][j] = A[p][j]; A[p][j] = t; }
            for (int r = 0; r < 8; r++) if (r != c) {
                double f = A[r][c] / A[c][c];
                for (int j = c; j < 9; j++) A[r][j] -= f * A[c][j];
            }
        }
        for (int k = 0; k < 8; k++) mmse_tab[s][k] = (float)(A[k][8] / A[k][k]);
    }
    /* exact end rows like the GNU Radio table */
    for (int k = 0; k < 8; k++) { mmse_tab[0][k] = (k == 3); mmse_tab[128][k] = (k == 4); }
    mmse_ready = 1;
    return &mmse_tab[0][0];
}

/* digital.clock_recovery_mm_ff (grc/recctest.grc:846-874): Mueller & Mueller loop */
void orc_mm_init(orc_mm_t *m, float omega, float gain_omega, float mu, float gain_mu, float rel)
{
    m->mu = mu; m->omega = omega; m->omega_mid = omega; m->omega_lim = omega * rel;
    m->gain_mu = gain_mu; m->gain_omega = gain_omega; m->last_sample = 0.0f;
}
size_t orc_mm_clock_recovery(orc_mm_t *m, const float *in, size_t n, float *out, size_t cap, size_t *consumed)
{
    const float (*tab)[8] = (const float (*)[8])orc_mmse_taps();
    size_t ii = 0, oo = 0;
    if (n < 8) { if (consumed) *consumed = 0; return 0; }
    size_t ni = n - 8;
    while (oo < cap && ii < ni) {
        int imu = (int)rintf(m->mu * 128.0f);
        const float *h = tab[imu];
        float y = 0.0f;
        for (int k = 0; k < 8; k++) y += h[k] * in[ii + (size_t)k];
        float sl = m->last_sample < 0.0f ? -1.0f : 1.0f, sy = y < 0.0f ? -1.0f : 1.0f;
        float e = sl * y - sy * m->last_sample;
        m->last_sample = y;
        m->omega += m->gain_omega * e;
        float dv = m->omega - m->omega_mid;
        if (dv > m->omega_lim) dv = m->omega_lim; else if (dv < -m->omega_lim) dv = -m->omega_lim;
        m->omega = m->omega_mid + dv;
        m->mu += m->omega + m->gain_mu * e;
        float fl = floorf(m->mu);
        ii += (size_t)(int)fl;
        m->mu -= fl;
        out[oo++] = y;
    }
    if (consumed) *consumed = ii;
    return oo;
}

/* digital.binary_slicer_fb (grc/recctest.grc:807) */
void orc_binary_slicer(const float *in, size_t n, uint8_t *out)
{
    for (size_t i = 0; i < n; i++) out[i] = in[i] >= 0.0f ? 1 : 0;
}

/* the chain of grc/recctest.grc:3238-3274 downstream of the channel filter */
size_t orc_chain_iq200(const float *iq, size_t n, uint32_t channel, int chunk,
                       amps_recc_burst_t *out, size_t cap, uint8_t *symbols_out, size_t sym_cap, size_t *nsym_out)
{
    float *d = (float *)malloc(sizeof(float) * (n + 1));
    size_t symcap = n / 9 + 16;
    float *soft = (float *)malloc(sizeof(float) * symcap);
    uint8_t *hard = (uint8_t *)malloc(symcap);
    orc_quadrature_demod(iq, n, 1.0f, d);
    orc_mm_t mm;
    orc_mm_init(&mm, 10.0f, 0.25f * 0.175f * 0.175f * 3.0f, 0.0f, 0.05f, 0.005f); /* grc/recctest.grc:846-874 */
    size_t consumed = 0;
    size_t nsym = orc_mm_clock_recovery(&mm, d, n, soft, symcap, &consumed);
    orc_binary_slicer(soft, nsym, hard);
    if (symbols_out) memcpy(symbols_out, hard, nsym < sym_cap ? nsym : sym_cap);
    if (nsym_out) *nsym_out = nsym;
    orc_recc_t *r = orc_recc_new();
    uint8_t burst[AMPS_RECC_CAPTURE_SYMS];
    size_t nout = 0;
    if (chunk < 1) chunk = 4096;
    for (size_t off = 0; off < nsym; off += (size_t)chunk) {
        int c = (int)(nsym - off < (size_t)chunk ? nsym - off : (size_t)chunk);
        if (orc_recc_work(r, hard + off, c, burst) == 1 && nout < cap)
            orc_decode_burst(burst, channel, 0, &out[nout++]);
    }
    orc_recc_free(r);
    free(d); free(soft); free(hard);
    return nout;
}

size_t orc_chain_iq400(const float *iq, size_t n, double fc, uint32_t channel, int chunk,
                       amps_recc_burst_t *out, size_t cap)
{
    float taps[512];
    int nt = orc_firdes_low_pass_blackman(3.0, 400e3, 10e3, 4.5e3, taps, 512); /* grc/recctest.grc:115-155 */
    float *y = (float *)malloc(sizeof(float) * 2 * (n / 2 + 1));
    size_t ny = orc_freq_xlating_fir(iq, n, taps, nt, fc, 400e3, 2, y);
    size_t r = orc_chain_iq200(y, ny, channel, chunk, out, cap, NULL, 0, NULL);
    free(y);
    return r;
}
