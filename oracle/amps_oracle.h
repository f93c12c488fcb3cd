/* amps_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference algorithm (unsynchronized/gr-amps) for the RECC receive path,
 * plus a CPU model of the fused MI355X algorithm.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product (gr_amps_amd/, include/) never
 * includes, links or calls anything in oracle/.
 *
 * PARITY STATUS (see DESIGN.md "Oracle"):  the reference cannot be built in this image -- every
 * source file on the path includes GNU Radio / IT++ / Boost headers that are absent, and writing
 * stand-ins for them is not allowed -- and the reference's own test-suite is empty
 * (lib/qa_amps.cc:9-15): NO reference-produced vector exists, so parity stays "partial".  What pins
 * each row instead (tests/test_cpu_oracle.py, tests/test_cpu_oracle_pins.py, tests/test_gpu_pins.py):
 *   R1 trigger, R3 Manchester, R6/R7 word + MIN parse, TX word builders
 *        the constants embedded in the reference sources + the known-answer values SURVEY.md 8a
 *        recorded from the reference's compiled code (tests/golden/survey_kats.json)
 *   R2 recc_impl::work
 *        the recorded stream behaviours Q1-Q4 (3/3/3/2 bursts for chunk 1000/4096/333/8191; wrap loss)
 *   R4 itpp::BCH(63,2,true)
 *        an INDEPENDENT brute-force decoder (tests/bchref.py: integer polynomial arithmetic, coset
 *        leaders) on all 4096 syndromes and all 39711 weight-3 patterns, including the documented
 *        IT++ rule "#roots == deg Lambda" (S1 = 0, S3 a cube: 21 syndromes are "corrected" at weight
 *        3); the encoder against polynomial division.  The IT++ VERSION is unpinned
 *        (CMakeLists.txt:89 names none): a release whose decode() differs from the published 4.x
 *        algorithm on uncorrectable words would differ from this restatement there
 *   G1 firdes.low_pass, G2 fast_atan2f, G3's MMSE table
 *        the blocks' published invariants (tap count / DC gain / symmetry / -6 dB cutoff / Blackman
 *        stop band; exact special cases and the 255-interval interpolation error bound; unit rows,
 *        mirror symmetry, DC gain, band-limited interpolation error) -- NOT GNU Radio's literal
 *        tables: the MMSE rows are the closed-form least-squares solution of the same objective,
 *        equal to the shipped table only to its print precision
 *   R5/R8 bursts_message control flow
 *        a SECOND restatement written apart from this one (tests/refdecode.py: Python over lists, BCH verdict from
 *        tests/bchref.py) agrees with this library and with the HIP kernel on every field of random bursts steered
 *        into all seven message classes, with bit errors and non-Manchester pairs (tests/test_second_restatement.py)
 *        -- two restatements by the same hand, not a reference vector
 *   still unpinned: what only a build of the reference could confirm:
 *        G3's loop arithmetic (clock_recovery_mm_ff) and G1's rotator renormalisation period, which
 *        only a GNU Radio build could confirm; word-level equality with the restated chain is
 *        self-consistency, not reference parity.
 */
#ifndef AMPS_ORACLE_H
#define AMPS_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "amps_recc.h" /* record layout only: the checker depends on the product's interface, never the reverse */

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- R1: trigger (lib/recc_impl.cc:51-65, 76-79) ---------------- */
int  orc_manchester_encode(const char *bits, size_t nbits, uint8_t *dst); /* returns -1 on a char outside '0'/'1' */
void orc_trigger(uint8_t dst[AMPS_RECC_TRIGGER_SYMS]);

/* ---------------- R2: recc_impl::work (lib/recc_impl.cc:93-145) ---------------- */
typedef struct orc_recc orc_recc_t;
orc_recc_t *orc_recc_new(void);
void        orc_recc_free(orc_recc_t *);
void        orc_recc_reset(orc_recc_t *);
/* one work() call; returns 1 and fills burst_out[3374] if the call published a burst, else 0;
 * -1 if n > AMPS_RECC_MAX_WORK_ITEMS (the reference asserts) */
int         orc_recc_work(orc_recc_t *, const uint8_t *in, int n, uint8_t *burst_out);
void        orc_recc_peek(const orc_recc_t *, uint64_t *len, int64_t *curstart, const uint8_t **buf);

/* ---------------- R3: manchester_decode_binbuf (lib/utils.cc:27-59) ---------------- */
size_t orc_manchester_decode_binbuf(const uint8_t *src, uint8_t *dst, size_t dstsz, int *nonbinary);

/* ---------------- R4: itpp::BCH(63,2,true) as used at lib/recc_decode_impl.cc:33,53-79 ---------------- */
/* one 63-bit codeword, bit j = coefficient of x^(62-j); returns 1 if no decoder failure */
int  orc_bch63_decode(const uint8_t rx[63], uint8_t corrected[63], int *nflips);
void orc_bch63_encode(const uint8_t msg[51], uint8_t cw[63]);
uint32_t orc_bch_generator(void);                 /* g(x) as a bit mask, x^12 = bit 12 */
int  orc_recc_bch_decode(const uint8_t src[48], uint8_t dst[36]);      /* recc_bch_decode */
void orc_bch_encode_short(const uint8_t *msg, int k, uint8_t *cw);     /* (k+12,k): (48,36) and (40,28) */

/* ---------------- R5-R8: bursts_message (lib/recc_decode_impl.cc:81-169) ---------------- */
void orc_decode_burst(const uint8_t burst[AMPS_RECC_CAPTURE_SYMS], uint32_t channel, uint64_t position,
                      amps_recc_burst_t *out);
/* majority != 0: CPU model of the product's optional majority decode mode (not reference behaviour) */
void orc_decode_burst_mode(const uint8_t burst[AMPS_RECC_CAPTURE_SYMS], uint32_t channel, uint64_t position,
                           amps_recc_burst_t *out, int majority);
void orc_reply_words(const amps_recc_burst_t *burst, amps_recc_reply_t *reply); /* :181-272 */

/* R6/R7 helpers (lib/amps_packet.h, lib/amps_packet.cc) */
int      orc_parse_min(const char *min, uint64_t *min1, uint64_t *min2);
void     orc_calc_min(uint64_t min1, uint64_t min2, char out[11]);
void     orc_called_digits(uint32_t digits, char out[9], int *bad);
void     orc_expandbits(uint8_t *out, size_t nbits, uint64_t val);
void     orc_focc_word1(uint8_t w[28], int multiword, unsigned dcc, uint64_t min1);
void     orc_focc_word2_general(uint8_t w[28], uint64_t min2, unsigned msg_type, unsigned ordq, unsigned order);
void     orc_fvc_word1_general(uint8_t w[28], unsigned pscc, unsigned msg_type, unsigned ordq, unsigned order);
void     orc_focc_word2_voice_channel(uint8_t w[28], unsigned scc, uint64_t min2, unsigned vmac, unsigned chan);

/* ---------------- G1-G4: the GNU Radio 3.7 blocks wired in grc/recctest.grc (own restatement, unpinned) ---------------- */
int    orc_firdes_low_pass_blackman(double gain, double fs, double cutoff, double width, float *taps, int cap);
/* freq_xlating_fir_filter_ccc: returns number of outputs; in is the whole stream (history = zeros) */
size_t orc_freq_xlating_fir(const float *iq_in, size_t n_in, const float *taps, int ntaps,
                            double center_freq, double fs, int decim, float *iq_out);
float  orc_fast_atan2f(float y, float x);
void   orc_quadrature_demod(const float *iq, size_t n, float gain, float *out); /* history sample = 0 */
typedef struct orc_mm {
    float mu, omega, omega_mid, omega_lim, gain_mu, gain_omega, last_sample;
} orc_mm_t;
void   orc_mm_init(orc_mm_t *, float omega, float gain_omega, float mu, float gain_mu, float omega_rel_limit);
size_t orc_mm_clock_recovery(orc_mm_t *, const float *in, size_t n, float *out, size_t cap, size_t *consumed);
void   orc_binary_slicer(const float *in, size_t n, uint8_t *out);
const float *orc_mmse_taps(void); /* [129][8] */

/* whole reference chain from 200 ksps IQ (after the channel filter): G2->G3->G4->R2->R5..R8.
 * chunk = symbols handed to recc::work per call.  symbols_out (cap sym_cap) is optional. */
size_t orc_chain_iq200(const float *iq, size_t n, uint32_t channel, int chunk,
                       amps_recc_burst_t *out, size_t cap, uint8_t *symbols_out, size_t sym_cap, size_t *nsym);
/* the recctest.grc chain from 400 ksps IQ: G1 (299-tap xlating FIR, decim 2) then the above */
size_t orc_chain_iq400(const float *iq, size_t n, double center_freq, uint32_t channel, int chunk,
                       amps_recc_burst_t *out, size_t cap);

/* ---------------- CPU model of the fused MI355X algorithm (include/amps_recc_numerics.h) ---------------- */
typedef struct orc_fused orc_fused_t;
orc_fused_t *orc_fused_new(uint32_t channel, int sps);
void         orc_fused_free(orc_fused_t *);
/* model of the product's optional tolerant sync (cfg.sync_tolerance): accept <= k mismatching trigger symbols */
void         orc_fused_set_tolerance(orc_fused_t *, int k);
void         orc_fused_set_majority(orc_fused_t *, int on);   /* captures decoded in the product's majority mode */
void         orc_fused_set_tracking(orc_fused_t *, int on);   /* timing tracking in the capture (default on; off = AMPS_RECC_FLAG_FIXED_TIMING) */
void         orc_fused_set_slicer(orc_fused_t *, int spec);   /* AMPS_SLICER_* (include/amps_recc_numerics.h); default = AMPS_SLICER_DEFAULT (spec D) */
/* push n new samples of this channel; returns number of records appended to out */
size_t       orc_fused_push(orc_fused_t *, const float *iq, size_t n, amps_recc_burst_t *out, size_t cap);
/* taps for tolerance tests: d, S and g of the stream pushed so far (valid for the processed prefix) */
size_t       orc_fused_processed(const orc_fused_t *);
const float   *orc_fused_demod(const orc_fused_t *);
const float   *orc_fused_soft(const orc_fused_t *);
const uint8_t *orc_fused_hard(const orc_fused_t *);
void   orc_fm_discriminator(const float *iq, size_t n, float *d); /* spec arithmetic, previous sample of x[0] is 0 */

#ifdef __cplusplus
}
#endif
#endif
