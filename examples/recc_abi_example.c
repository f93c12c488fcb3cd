/* recc_abi_example.c -- the boundary from plain C (C99, gcc): the two reference blocks of the path chained through the C ABI.
 *
 *   gr::amps::recc::work          (lib/recc_impl.cc:93-145)          ->  amps_recc_push_symbols
 *   recc_decode::bursts_message   (lib/recc_decode_impl.cc:81-169)   ->  amps_recc_decode_bursts
 *   recc_decode's reply           (lib/recc_decode_impl.cc:181-272)  ->  amps_recc_reply_words
 *
 * A mobile's page response (TIA/EIA-553: 30 bits of dotting, word sync 11100010010, coded DCC, words A and B five times each, every
 * word BCH(48,36) coded -- by the library's own encoder, amps_bch_encode_words) is Manchester coded as lib/recc_impl.cc:51-65 maps it
 * (bit 0 -> symbols 1,0; bit 1 -> 0,1), dropped into a stream of idle symbols and handed to the library in work()-sized calls, as a flow
 * graph's scheduler would.  The burst the first call publishes goes through the decode call; the program prints what came back and
 * exits 0 iff it is the MIN that was sent.
 *
 *   gcc -std=c99 -Wall -Iinclude examples/recc_abi_example.c -o recc_abi_example -Lgr_amps_amd -lamps_recc -Wl,-rpath,$PWD/gr_amps_amd
 *
 * No HIP header, no C++: this file sees include/amps_recc.h and nothing else of the library.  It needs an MI355X to RUN (the library has
 * no CPU fallback: amps_recc_create answers -ENODEV without one, and this program says so and exits 77).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>
#include "amps_recc.h"

static void put_bits(uint8_t *o, int n, unsigned long v) { int i; for (i = n - 1; i >= 0; i--) { o[i] = (uint8_t)(v & 1u); v >>= 1; } }

/* 10-digit MIN -> MIN1 (24 bits), MIN2 (10 bits): TIA-553 2.3.1, the inverse of calc_min (lib/amps_packet.h:305-349) */
static unsigned d3(const char *s)
{
    unsigned a = s[0] == '0' ? 10u : (unsigned)(s[0] - '0'), b = s[1] == '0' ? 10u : (unsigned)(s[1] - '0'), c = s[2] == '0' ? 10u : (unsigned)(s[2] - '0');
    return (100u * a + 10u * b + c - 111u) & 0x3ffu;
}
static void min_fields(const char *m, unsigned long *min1, unsigned *min2)
{
    const unsigned thous = m[6] == '0' ? 10u : (unsigned)(m[6] - '0');
    *min1 = ((unsigned long)d3(m + 3) << 14) | ((unsigned long)(thous & 0xfu) << 10) | d3(m + 7);
    *min2 = d3(m);
}

#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s: %s\n", #call, amps_recc_strerror(rc_)); return 1; } } while (0)

int main(void)
{
    const char *min_sent = "2065551234";
    enum { NSYM = 40000, OFFSET = 5000, WORK = 8192 };
    static uint8_t syms[NSYM], burst[AMPS_RECC_CAPTURE_SYMS], burst_out[2 * AMPS_RECC_CAPTURE_SYMS];
    uint8_t msg[2][AMPS_RECC_MSG_BITS], cw[2][AMPS_RECC_WORD_BITS], bits[48 + 7 * 240];
    amps_recc_cfg_t cfg;
    amps_recc_t *h = NULL;
    amps_recc_burst_t rec;
    amps_recc_reply_t reply;
    uint32_t chan[2];
    unsigned long min1;
    unsigned min2, seed = 12345u;
    size_t nout = 0, nfound = 0;
    int nbits = 0, i, w, r, rc, done;

    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.n_channels = 1;
    cfg.max_bursts = 16;
    cfg.device = -1;
    rc = amps_recc_create(&h, &cfg);          /* symbol seam only: no IQ capacity asked for */
    if (rc == -ENODEV) { fprintf(stderr, "no MI355X here: %s\n", amps_recc_strerror(rc)); return 77; }
    if (rc != 0) { fprintf(stderr, "amps_recc_create: %s\n", amps_recc_strerror(rc)); return 1; }
    printf("ABI version %d, record size %u bytes\n", amps_recc_abi_version(), (unsigned)amps_recc_burst_size());

    /* word A: F=1 NAWC=1 T=0 S=0 E=1 ER=0 SCM=0110 MIN1;  word B: F=0 NAWC=0, all order fields 0, MIN2  (lib/amps_packet.h:103-170) */
    min_fields(min_sent, &min1, &min2);
    put_bits(msg[0], 1, 1); put_bits(msg[0] + 1, 3, 1); put_bits(msg[0] + 4, 4, 0x2 /* T S E ER = 0 0 1 0 */); put_bits(msg[0] + 8, 4, 0x6); put_bits(msg[0] + 12, 24, min1);
    put_bits(msg[1], 26, 0); put_bits(msg[1] + 26, 10, min2);
    CHECK(amps_bch_encode_words(h, &msg[0][0], 2, AMPS_RECC_MSG_BITS, AMPS_MEM_HOST, &cw[0][0]));

    /* the burst: dotting, word sync, coded DCC 0 (0000000), each word five times */
    for (i = 0; i < 30; i++) bits[nbits++] = (uint8_t)((i & 1) ^ 1);
    for (i = 0; i < 11; i++) bits[nbits++] = (uint8_t)("11100010010"[i] - '0');
    for (i = 0; i < 7; i++) bits[nbits++] = 0;
    for (w = 0; w < 2; w++) for (r = 0; r < AMPS_RECC_REPEATS; r++) { memcpy(bits + nbits, cw[w], AMPS_RECC_WORD_BITS); nbits += AMPS_RECC_WORD_BITS; }

    /* idle symbols (a small LCG: any 0/1 noise does), then the Manchester-coded burst at OFFSET */
    for (i = 0; i < NSYM; i++) { seed = seed * 1103515245u + 12345u; syms[i] = (uint8_t)((seed >> 16) & 1u); }
    for (i = 0; i < nbits; i++) { syms[OFFSET + 2 * i] = (uint8_t)(1 - bits[i]); syms[OFFSET + 2 * i + 1] = bits[i]; }

    /* one amps_recc_push_symbols per work() call */
    for (done = 0; done < NSYM; done += WORK) {
        const int n = NSYM - done < WORK ? NSYM - done : WORK;
        CHECK(amps_recc_push_symbols(h, syms + done, (size_t)n, n, AMPS_MEM_HOST, burst_out, chan, 2, &nout));
        if (nout) { memcpy(burst, burst_out, sizeof burst); nfound += nout; printf("work() call at symbol %d published a burst on channel %u\n", done, (unsigned)chan[0]); }
    }
    if (nfound != 1) { fprintf(stderr, "expected one burst, got %u\n", (unsigned)nfound); amps_recc_destroy(h); return 1; }

    CHECK(amps_recc_decode_bursts(h, burst, 1, AMPS_MEM_HOST, NULL, &rec));
    CHECK(amps_recc_reply_words(&rec, &reply));
    printf("message class %u, MIN %.10s, word A valid %u (repeat %u), word B valid %u, NAWC %u, SCM %u\n", (unsigned)rec.msg_class, rec.min,
           (unsigned)rec.valid[0], (unsigned)rec.first_valid_rep[0], (unsigned)rec.valid[1], (unsigned)rec.a_NAWC, (unsigned)rec.a_SCM);
    printf("reply: %s FOCC words, %s FVC word\n", reply.has_focc ? "has" : "no", reply.has_fvc ? "has" : "no");
    amps_recc_destroy(h);
    if (strncmp(rec.min, min_sent, 10) != 0 || !rec.valid[0] || !rec.valid[1]) { fprintf(stderr, "decoded MIN differs from the one sent\n"); return 1; }
    printf("ok\n");
    return 0;
}
