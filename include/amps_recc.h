/* amps_recc.h -- C ABI of the MI355X-native AMPS reverse-control-channel (RECC) receive path.
 *
 * This is the drop-in boundary.  Every entry point replaces one
 * interface of the reference (unsynchronized/gr-amps); citations are file:line under the
 * reference tree:
 *
 *   amps_recc_push_symbols      <-  gr::amps::recc_impl::work            lib/recc_impl.cc:93-145
 *                                   (io signature: 1 x unsigned char in, lib/recc_impl.cc:71-73;
 *                                    "bursts" blob of 3374 bytes out,    lib/recc_impl.cc:82,126)
 *   amps_recc_decode_bursts     <-  recc_decode_impl::bursts_message     lib/recc_decode_impl.cc:81-169
 *                                   recc_decode_impl::recc_bch_decode    lib/recc_decode_impl.cc:53-79
 *                                   manchester_decode_binbuf             lib/utils.cc:27-59
 *                                   recc_word_a/_b/_c_serial/_called     lib/amps_packet.h:103-274
 *                                   extract_min_3 / calc_min             lib/amps_packet.h:277-302,354-366
 *   amps_recc_push_iq / _drain  <-  the flow-graph sub-chain  quadrature_demod_cf -> clock_recovery_mm_ff
 *                                   -> binary_slicer_fb -> amps_recc -> amps_recc_decode
 *                                   (grc/recctest.grc:458,846-874,807,310,349 and connections :3238-3274)
 *   amps_recc_push_wideband     <-  N x (freq_xlating_fir_filter_ccc -> the chain above), one per 30 kHz
 *                                   channel (grc/recctest.grc:889-937, taps :115-155); polyphase channelizer
 *                                   front end: M = 1024 branches at fs = 30.72 Msps, 8 taps per branch,
 *                                   D = 768 (40 ksps per channel, samples_per_symbol = 2; the default) or
 *                                   D = 512 (60 ksps, samples_per_symbol = 3): the two geometries
 *                                   amps_recc_create accepts (anything else: -EINVAL)
 *   amps_recc_reply_words       <-  handle_response / handle_registration / handle_origination
 *                                   lib/recc_decode_impl.cc:181-272 + word builders lib/amps_packet.cc:26-95
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative errno-style
 * code (never throws, never aborts); one handle is single-threaded (same rule as a GNU Radio block,
 * SURVEY.md 8b "Threading"); all compute runs in hand-written HIP kernels on the handle's device --
 * there is NO CPU fallback: without a usable HIP device amps_recc_create() fails with -ENODEV.
 */
#ifndef AMPS_RECC_H
#define AMPS_RECC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMPS_RECC_ABI_VERSION 4   /* 2: amps_recc_cfg_t gained wideband_groups / wideband_group; 3: default slicer = spec D, captures track the bit
                                     clock unless AMPS_RECC_FLAG_FIXED_TIMING, amps_recc_rccl_* / _push_wideband_bcast / _drain_gather / _debug_exact_slice added;
                                     4: amps_recc_push_wideband_dist (scatter + all-gather), amps_recc_rccl_info / _abort / _set_timeout; rccl_init checks the
                                     group split and allocates; every collective entry is bounded and carries a status word */

/* protocol constants of the reference */
#define AMPS_RECC_TRIGGER_SYMS 74   /* lib/recc_impl.cc:76-77: 37 bits x 2 Manchester symbols   */
#define AMPS_RECC_CAPTURE_SYMS 3374 /* lib/recc_impl.cc:70: (7 + 7*240) bits x 2                */
#define AMPS_RECC_WORDS        7    /* lib/recc_decode_impl.cc:92                                */
#define AMPS_RECC_REPEATS      5    /* lib/recc_decode_impl.cc:101                               */
#define AMPS_RECC_WORD_BITS    48   /* BCH(48,36) = BCH(63,51) shortened by 15                   */
#define AMPS_RECC_MSG_BITS     36
#define AMPS_RECC_SYMBUF       65536 /* lib/recc_impl.cc:68 d_symbufsz                           */
#define AMPS_RECC_WINDOW       4096  /* lib/recc_impl.cc:69 d_windowsz                           */
#define AMPS_RECC_MAX_WORK_ITEMS 61439 /* lib/recc_impl.cc:103: noutput_items < bufsz - windowsz */

/* where a data pointer lives */
#define AMPS_MEM_HOST   0
#define AMPS_MEM_DEVICE 1

/* amps_recc_cfg_t.flags */
#define AMPS_RECC_FLAG_TIME_KERNELS 0x1u /* record HIP events around every kernel (amps_recc_get_timing) */
#define AMPS_RECC_FLAG_UNFUSED_WIDEBAND 0x4u /* channelizer seam: keep the channel-major intermediate in HBM (two kernels)
                                               instead of fusing the RECC front end behind the FFT; same results */
#define AMPS_RECC_FLAG_SLICER_PRODUCT 0x8u /* IQ / wideband seams: slicer spec B of amps_recc_numerics.h (sign of
                                               Im(x[n] conj(x[n-sps])), the telescoped form of discriminator + boxcar) */
#define AMPS_RECC_FLAG_SLICER_SINE  0x10u /* IQ / wideband seams: slicer spec C (boxcar over Im(x[n] conj(x[n-1])): spec A without the
                                               arctangent) */
#define AMPS_RECC_FLAG_SLICER_EXACT 0x80u /* IQ / wideband seams: slicer spec D (the sign of spec A's boxcar sum computed exactly from sign
                                               bits and the winding number: spec A's decisions at spec C's cost).  At most one of the four
                                               SLICER flags may be set */
#define AMPS_RECC_FLAG_SLICER_ATAN  0x40u /* IQ / wideband seams: slicer spec A (arctangent discriminator + boxcar), explicitly.
                                               With no SLICER flag a handle uses amps_recc_default_slicer() */
#define AMPS_RECC_FLAG_KEEP_BURSTS  0x20u /* IQ / wideband seams: also keep the 3374 captured symbol bytes of every burst (what
                                               gr::amps::recc publishes on "bursts", lib/recc_impl.cc:126) for amps_recc_drain_bursts */
#define AMPS_RECC_FLAG_FIXED_TIMING 0x100u /* IQ / wideband seams: sample all 3374 symbols of a capture at the one phase the trigger run
                                               gives (rounds 1-3).  Default: the capture tracks the mobile's bit clock, one sample per
                                               repeat at most, from where the mid-bit transitions fall -- the fast path's
                                               stand-in for clock_recovery_mm_ff's loop (grc/recctest.grc:846-874) */
#define AMPS_RECC_FLAG_MAJORITY     0x2u /* decode mode "majority" instead of "reference" (SURVEY.md 8f.2), see below */

/* message classes, the branches of lib/recc_decode_impl.cc:108-168 */
enum amps_recc_msg_class {
    AMPS_MSG_INVALID_WORD_A = 0, /* validwords[0]==false  -> dropped   (:108-111) */
    AMPS_MSG_E_ZERO         = 1, /* worda.E==false        -> dropped   (:113-116) */
    AMPS_MSG_PAGE_RESPONSE  = 2, /* handle_response                    (:121-122) */
    AMPS_MSG_REGISTRATION   = 3, /* handle_registration                (:123-138) */
    AMPS_MSG_ORIGINATION    = 4, /* handle_origination                 (:139-165) */
    AMPS_MSG_BAD_NAWC       = 5, /* origination with nawc outside 1..4 (:155-158) */
    AMPS_MSG_UNKNOWN        = 6  /* "got unknown RECC message"         (:166-168) */
};

/* Decode modes.
 * reference (default): exactly lib/recc_decode_impl.cc:96-117 -- the five repeats are BCH-decoded in order and
 *   the first that decodes wins; only validwords[0] gates; fields are parsed from the RAW repeat 0.
 * majority (AMPS_RECC_FLAG_MAJORITY): what TIA/EIA-553 specifies and the reference leaves as "XXX" -- each of
 *   the 48 bit positions of a word is a 3-of-5 vote over the repeats; the voted word is BCH-decoded once and is
 *   valid only if it decodes AND no correction falls into the 15 shortening positions; fields are parsed from the
 *   corrected bits; every word the dispatch reads must be valid; the 7-bit coded DCC must be within one bit of a
 *   code word.  In this mode word_raw = voted bits, first_valid_rep = number of repeats equal to the voted word.
 *   On error-free bursts both modes produce the same words, fields and class. */

/* amps_recc_burst_t.flags */
#define AMPS_BURST_FLAG_NONBINARY 0x1u /* a symbol byte outside {0,1} was seen (reference: assert(0), UB in Release) */
#define AMPS_BURST_FLAG_WORDC_NAWC_MISMATCH 0x2u /* "protocol violation" warning of :134-136 / :150-152 */
#define AMPS_BURST_FLAG_BAD_DIGIT 0x4u /* digit code 13..15 truncated a called-address word (amps_packet.h:219-223) */
#define AMPS_BURST_FLAG_DCC_INVALID 0x8u /* majority mode: coded DCC further than one bit from every code word */

/* One decoded seizure burst.  Bits are one byte per bit (0/1), MSB first, exactly as the reference
 * holds them (lib/recc_decode_impl.cc:92-95).  Layout is fixed: natural alignment, 728 bytes. */
typedef struct amps_recc_burst {
    uint32_t channel;        /* RECC instance index                                                     */
    uint32_t flags;          /* AMPS_BURST_FLAG_*                                                       */
    uint64_t position;       /* symbol seam: 0.  IQ seams: absolute sample index of the last trigger
                                symbol's decision instant (capture symbol i is sliced at position+sps*(i+1)) */
    uint8_t  dcc[7];         /* manchester_decode_binbuf(bdata, dcc, 7)            (:90)               */
    uint8_t  dcc_bad;        /* its return value (bad Manchester pairs)                                 */
    uint16_t manch_bad[AMPS_RECC_WORDS];                 /* errs[i]                 (:97)               */
    uint8_t  valid[AMPS_RECC_WORDS];                     /* validwords[w]           (:102)              */
    uint8_t  first_valid_rep[AMPS_RECC_WORDS];           /* r at the break, 5 if none valid            */
    uint8_t  word_raw[AMPS_RECC_WORDS][AMPS_RECC_WORD_BITS]; /* words[w][0..47]: repeat 0, uncorrected --
                                                            what the reference parses (:112,117,130,146,161) */
    uint8_t  word_dec[AMPS_RECC_WORDS][AMPS_RECC_MSG_BITS];  /* corrected message bits of the first valid
                                                            repeat; uncorrected bits of repeat 4 if none   */
    /* ---- parsed fields (lib/amps_packet.h:103-198), always from word_raw like the reference ---- */
    uint8_t  a_F, a_NAWC, a_T, a_S, a_E, a_ER, a_SCM, _pad0;
    uint32_t a_MIN1;
    uint8_t  b_F, b_NAWC, b_MSG_TYPE, b_ORDQ, b_ORDER, b_LT, b_EP, b_SCM4, b_MPCI, b_SDCC1, b_SDCC2, _pad1;
    uint16_t b_MIN2;
    uint16_t _pad2;
    uint32_t esn;            /* recc_word_c_serial::SERIAL when read, else 0                            */
    uint8_t  has_esn;        /* registration: worda.S (:128); origination: worda.S (:145)               */
    uint8_t  msg_class;      /* enum amps_recc_msg_class                                                */
    uint8_t  n_called_words; /* origination: number of called-address words consumed                    */
    uint8_t  _pad3;
    char     min[12];        /* calc_min(): 10 digits, NUL padded                                        */
    char     dialed[36];     /* concatenated recc_word_called::digits(), NUL padded (<= 32 chars)        */
    uint32_t _pad4;
} amps_recc_burst_t;
#define AMPS_RECC_BURST_BYTES 728

/* beyond ~8 wrong symbols the dotting pattern shifted by one bit period starts to pass as a trigger */
#define AMPS_RECC_MAX_SYNC_TOLERANCE 8

typedef struct amps_recc_cfg {
    uint32_t struct_size;          /* sizeof(amps_recc_cfg_t), for ABI evolution                          */
    uint32_t n_channels;           /* independent RECC instances handled per push (1 .. 2^20-1); a channel's
                                    * stream may run for 2^44 samples between resets (2.8 years at 200 ksps) */
    uint32_t samples_per_symbol;   /* IQ seam: samples per Manchester symbol (10 at 200 ksps); 2..16      */
    uint32_t max_samples_per_push; /* IQ seam: capacity per channel per push (0 = IQ seam unused)         */
    uint32_t max_bursts;           /* capacity of the device-side result list per push/drain (>=1)        */
    int32_t  device;               /* HIP device ordinal, -1 = current device                             */
    uint32_t flags;                /* AMPS_RECC_FLAG_*                                                    */
    uint32_t wideband_channels;    /* channelizer seam: M branches: 1024 (0 = seam unused); other values: -EINVAL */
    uint32_t wideband_decim;       /* channelizer seam: D input samples per output frame; 0 = amps_recc_default_wideband_decim().
                                    * 768 (4/3 x oversampled: 40 ksps per channel, samples_per_symbol = 2; the default since round 6)
                                    * or 512 (2x oversampled: 60 ksps, samples_per_symbol = 3; 1.5 x the frames per input byte,
                                    * 0.2 - 0.7 dB more sensitive)                                                     */
    uint32_t wideband_taps_per_branch; /* prototype length = taps_per_branch * M: 8 (0 selects 8)              */
    uint32_t wideband_first_channel;   /* first FFT bin that is an active RECC channel                    */
    uint32_t sync_tolerance;       /* IQ / wideband seams: accept a trigger with up to this many of its 74
                                    * symbols wrong.  0 = exact match, the reference's memmem
                                    * (lib/recc_impl.cc:118) -- keep 0 for parity runs.  <= AMPS_RECC_MAX_SYNC_TOLERANCE */
    uint32_t wideband_groups;      /* channelizer seam: 0 / 1 = this handle decodes the whole band selection; G = 2, 4 or 8 = the band's
                                    * channels are split into G interleaved groups and this handle decodes ONE of them (one handle per
                                    * GPU of a node, every one fed the same wideband stream: BASELINE configs[4]).  Group r = the active
                                    * channels whose FFT bin k has (k mod 64) in [r * 64/G, (r+1) * 64/G): blocks of 64/G adjacent channels,
                                    * every 64 -- the split that lets a rank skip the last FFT pass and the slicer for everybody else's
                                    * bins.  n_channels / wideband_first_channel still describe the WHOLE band selection; records carry
                                    * whole-band channel numbers; fused form only, and the handle serves the wideband seam only:
                                    * amps_recc_push_iq / _push_raw / _push_symbols answer -ENOSYS on it                                */
    uint32_t wideband_group;       /* channelizer seam: which group, 0 .. wideband_groups - 1                                          */
    void    *stream;               /* hipStream_t to launch on, NULL = library-owned stream               */
} amps_recc_cfg_t;

typedef struct amps_recc_timing {
    uint32_t struct_size;
    uint32_t launches_front;   /* number of timed launches of the front kernel since last reset           */
    double   ms_front;         /* summed HIP-event time of the dominant streaming kernel (demod+sync)     */
    double   ms_resolve;       /* per-channel detection ordering / capture scheduling kernel              */
    double   ms_decode;        /* burst extract + Manchester + BCH + parse kernel                         */
    double   ms_carry;         /* inter-push halo copy kernel                                             */
    double   ms_symbols;       /* symbol-seam work() kernel                                               */
    uint64_t samples_front;    /* per-channel IQ samples consumed by the timed front launches             */
    double   ms_channelizer;   /* polyphase channelizer kernel (wideband seam)                            */
    uint32_t launches_channelizer;
    uint32_t _pad;
    double   ms_xlate;         /* translate seam: mixing + channel FIR + decimation kernel                */
} amps_recc_timing_t;

typedef struct amps_recc amps_recc_t; /* opaque; owns device buffers + per-channel stream state */

/* version / introspection */
int         amps_recc_abi_version(void);
/* the numeric slicer spec (AMPS_SLICER_* of amps_recc_numerics.h) of a handle created with no SLICER flag */
int         amps_recc_default_slicer(void);
/* The decimation a wideband handle created with cfg.wideband_decim = 0 uses (cfg.samples_per_symbol = 0 is then filled in to match:
 * 1536 / decim).  768 since round 6: 4/3 x oversampled, 2 samples per symbol -- 1.4 x the throughput of the 2x oversampled bank for
 * 0.2 dB at 1 % burst loss without a carrier offset and 0.7 dB at +-2 kHz (profiles/r06/decim768_sensitivity.txt).  512 (3 samples per
 * symbol) stays one field away: the more sensitive form, and what rounds 1-5 shipped. */
uint32_t    amps_recc_default_wideband_decim(void);
const char *amps_recc_strerror(int code);
size_t      amps_recc_burst_size(void);   /* sizeof(amps_recc_burst_t), for binding self-checks */

/* lifetime */
int  amps_recc_create(amps_recc_t **out, const amps_recc_cfg_t *cfg);
void amps_recc_destroy(amps_recc_t *h);
int  amps_recc_reset(amps_recc_t *h);     /* back to the just-constructed state of every channel */
/* Resume / offset a stream: the first sample pushed after create or reset gets this absolute index (a multiple of 64,
 * < 2^44), so `position` of the records continues a numbering kept elsewhere (checkpoint / restart of a receiver).
 * What precedes the origin is treated like what precedes a stream that starts at 0.  -EBUSY after the first push. */
int  amps_recc_set_origin(amps_recc_t *h, uint64_t first_sample);

/* (i) exact drop-in seam.  One call == one recc_impl::work() call on EVERY channel with
 * noutput_items = n; syms is [n_channels][ld] bytes (values 0/1).  Bursts published by this call
 * are copied to bursts_out ([cap][3374] host bytes) with their channel in burst_channel[], sorted
 * by channel (a channel publishes at most one burst per work() call).  n<1 returns 0 with *nout=0
 * (lib/recc_impl.cc:99-102); n>AMPS_RECC_MAX_WORK_ITEMS returns -EINVAL (the reference asserts). */
int amps_recc_push_symbols(amps_recc_t *h, const uint8_t *syms, size_t ld, int n, int mem,
                           uint8_t *bursts_out, uint32_t *burst_channel, size_t cap, size_t *nout);

/* recc_decode core on a batch of 3374-byte bursts ([nbursts][3374], host or device per `mem`); out and burst_channel are
 * always HOST memory. */
int amps_recc_decode_bursts(amps_recc_t *h, const uint8_t *bursts, size_t nbursts, int mem,
                            const uint32_t *burst_channel /* may be NULL */, amps_recc_burst_t *out);

/* (ii) fused seam: interleaved fc32 IQ, channel-major: sample i of channel c at iq[2*(c*ld + i)].
 * Enqueues the fused demod->sync->capture->decode kernels on the handle's stream and returns
 * without synchronising; results accumulate on the device until amps_recc_drain().
 * Ownership: a HOST buffer has been copied to the device when the call returns (synchronous copy into a
 * fenced staging buffer) and may be reused at once.  A DEVICE buffer is read in place by the enqueued
 * kernels: it must stay valid and unmodified until a drain (or drain_end) that covers this push has
 * returned, or the caller has synchronised the handle's stream.  The same holds for push_wideband /
 * push_raw. */
int amps_recc_push_iq(amps_recc_t *h, const float *iq, size_t ld, size_t nsamp, int mem);

/* channelizer seam: one wideband interleaved fc32 stream (fs = M * 30 kHz) -> polyphase
 * channelizer -> the same fused path on every active channel.  nsamp wideband samples. */
int amps_recc_push_wideband(amps_recc_t *h, const float *iq, size_t nsamp, int mem);

/* translate seam: the channel filter the reference's test flow graph wires in front of the
 * chain -- freq_xlating_fir_filter_ccc(decim, firdes.low_pass(gain, rate, cutoff, width), center, rate),
 * grc/recctest.grc:889-937 with taps :115-155 -- so that its ".raw" fc32 captures (400 ksps, channel at
 * +-160 kHz, :591) can be pushed directly.  rate_hz / decim must equal samples_per_symbol * 20 kHz.
 * gain / cutoff_hz / width_hz = 0 select the flow graph's 3.0 / 10 kHz / 4.5 kHz (299 taps at 400 ksps).
 * decim = 0 removes the stage.  Resets nothing else; call before the first push. */
typedef struct amps_recc_xlate_cfg {
    uint32_t struct_size;
    uint32_t decim;            /* 1, 2 or 4 */
    double   rate_hz;          /* input sample rate */
    double   center_hz;        /* channel centre relative to the input's centre, |center| <= rate */
    double   gain, cutoff_hz, width_hz;
} amps_recc_xlate_cfg_t;
int amps_recc_set_xlate(amps_recc_t *h, const amps_recc_xlate_cfg_t *x);
/* like amps_recc_push_iq but iq is [n_channels][ld] at rate_hz; every channel uses the same centre.
 * nsamp <= decim * max_samples_per_push; any nsamp (leftover samples wait for the next push). */
int amps_recc_push_raw(amps_recc_t *h, const float *iq, size_t ld, size_t nsamp, int mem);
/* test tap: run only the translate stage (continuing its stream); out is host [n_channels][out_ld] fc32 */
int amps_recc_debug_xlate(amps_recc_t *h, const float *iq, size_t ld, size_t nsamp, int mem,
                          float *out, size_t out_ld, size_t *nout);

/* Reference-timing seam (checking mode): the flow graph's OWN sub-chain in front of amps_recc, computed on the device as GNU
 * Radio 3.7 defines it -- analog.quadrature_demod_cf(1) -> digital.clock_recovery_mm_ff(omega 10, gain_omega .25*.175^2*3,
 * mu 0, gain_mu .05, omega_relative_limit .005) -> digital.binary_slicer_fb (grc/recctest.grc:458, 846-874, 807) -- for
 * every channel of a channel-major fc32 block at 200 ksps (samples_per_symbol must be 10).  The Mueller & Mueller loop is
 * a sequential recursion: one lane per channel.  symbols_out is host memory [n_channels][sym_ld] (values 0/1: exactly the
 * byte stream gr::amps::recc::work is fed), nsym_out[c] the number produced for channel c by this call (<= nsamp/9 + 16);
 * stream state continues across calls.  Feed the symbols to amps_recc_push_symbols for the reference chain end to end. */
int amps_recc_refchain_symbols(amps_recc_t *h, const float *iq, size_t ld, size_t nsamp, int mem,
                               uint8_t *symbols_out, size_t sym_ld, uint32_t *nsym_out);
/* test tap: the two tables the seam uses (fast_atan2f: 258 floats, MMSE interpolator: 129 x 8 floats) */
int amps_recc_refchain_tables(amps_recc_t *h, float *atan258, float *mmse1032);

/* Stream ordering for DEVICE buffers.  A handle created with cfg.stream = NULL launches on its own non-blocking stream,
 * which is not ordered against any other stream: a device buffer must have been completely written before it is pushed.
 * Either synchronise the producing stream first, or record an event behind the producer and hand it over here: work
 * enqueued by LATER calls on this handle waits for it (hipStreamWaitEvent; nothing blocks on the host).  `hip_event` is
 * a hipEvent_t. */
int amps_recc_wait_event(amps_recc_t *h, void *hip_event);
/* the converse: records `hip_event` (a hipEvent_t of the caller) behind everything enqueued on the handle so far, so that
 * another stream can wait for the handle's kernels before it overwrites a buffer they read */
int amps_recc_record_event(amps_recc_t *h, void *hip_event);

/* Synchronise the handle's stream and copy out the decoded bursts accumulated since the last
 * drain, sorted by (channel, position).  -ENOSPC if the device list overflowed max_bursts
 * (the first max_bursts records are still returned). */
int amps_recc_drain(amps_recc_t *h, amps_recc_burst_t *out, size_t cap, size_t *nout);
/* Split drain for streaming callers that must not idle the GPU while the host collects results:
 *     push(n); drain_begin();  push(n+1);  drain_end(&records of everything pushed before drain_begin) ...
 * drain_begin closes the current record list without waiting (later pushes append to a second list);
 * drain_end waits only for the work enqueued before drain_begin.  At most one split drain is open
 * (-EBUSY otherwise); amps_recc_drain == drain_begin + drain_end. */
int amps_recc_drain_begin(amps_recc_t *h);
int amps_recc_drain_end(amps_recc_t *h, amps_recc_burst_t *out, size_t cap, size_t *nout);
/* amps_recc_drain that also hands out the captured symbols of every record, bursts_out[i] = the 3374 bytes (0/1) the reference's
 * recc block would have published for record i ([cap][3374] host bytes; handle created with AMPS_RECC_FLAG_KEEP_BURSTS, -ENOSYS
 * otherwise).  decode_bursts(bursts_out[i]) gives out[i] again (position and channel aside). */
int amps_recc_drain_bursts(amps_recc_t *h, amps_recc_burst_t *out, uint8_t *bursts_out, size_t cap, size_t *nout);

/* test/diagnostic taps (not on the hot path): FM-demod floats and sliced symbol bits of one
 * channel for the last push_iq() call.  demod/soft/hard are host arrays of length n (may be NULL). */
int amps_recc_debug_demod(amps_recc_t *h, const float *iq, size_t nsamp, int mem,
                          float *demod, float *soft, uint8_t *hard);

/* ---- one band over the GPUs of a node (BASELINE configs[4]: "RCCL broadcast of wideband IQ over xGMI") ----
 * One process and one handle per GPU, every handle created with cfg.wideband_groups = N, wideband_group = its rank (the reference's
 * channels are independent -- per-instance state only, lib/recc_impl.h:31-43 -- so any split of them is exact).  Rank 0 (or any one
 * rank) calls amps_recc_rccl_unique_id and the application carries the 128 bytes to the other ranks (a file, MPI,
 * torch.distributed's store: the control plane is the application's); every rank then calls amps_recc_rccl_init and from then on
 * amps_recc_push_wideband_dist (or _bcast) in step.  RCCL is loaded at run time (librccl.so, or the library the environment
 * variable AMPS_RECC_RCCL_LIB names); -ENOSYS where it is absent.  nranks = 1 is valid.
 *
 * No rank is ever left waiting inside a collective (round 5):
 *  - everything that can fail on one rank is checked or allocated BEFORE a collective and travels through it as a status word (a
 *    16-byte all-gather in front of every push and every gather): either all ranks run the data collective or none does, and all of
 *    them return an error -- the rank's own, or -EREMOTEIO on the ranks that were fine.  The communicator stays usable;
 *  - every wait of the host is bounded (amps_recc_rccl_set_timeout, default 30 s, or AMPS_RECC_RCCL_TIMEOUT_MS at init): when the
 *    other ranks do not answer the communicator is aborted and the call returns -ETIMEDOUT; from then on the collective entry
 *    points answer -ENOTCONN.  Since round 6 the bound holds for EVERY wait that can sit behind a collective -- amps_recc_drain /
 *    _drain_end / _drain_bursts, amps_recc_reset, amps_recc_rccl_info and amps_recc_destroy as well;
 *  - an aborted collective lets the kernels queued behind it run on a receive buffer that never received: what the handle found
 *    since its last drain is void and its stream state has advanced over garbage.  The data seams and the drains therefore answer
 *    -ESTALE after a communicator died (timeout, error or amps_recc_rccl_abort) until amps_recc_reset has been called; then the handle
 *    decodes again, on its own (amps_recc_push_wideband) or with a new communicator;
 *  - a rank that has to leave (its flow graph stops, its device failed) calls amps_recc_rccl_abort: its peers then run into their
 *    bound instead of waiting for ever.
 *
 * amps_recc_rccl_init is itself a collective: it returns when all nranks have joined AND compared notes.  -EINVAL on a handle built
 * with wideband_groups = G >= 2 unless nranks == G and rank == wideband_group (it would decode part of the band and nobody the
 * rest), or when the ranks' handles were built for different group counts; -EREMOTEIO where another rank failed; in all those
 * cases every rank returns an error and no rank keeps a communicator.  The receive buffers (2 x the largest push) and the gather
 * buffers are allocated here, for the capacities the ranks have in common: the smallest max_samples_per_push, the largest
 * max_bursts.  Only a rank that cannot join at all (no id, rank outside 0 .. nranks - 1: -EINVAL at once) leaves the others
 * waiting in RCCL's own bootstrap. */
#define AMPS_RECC_RCCL_ID_BYTES 128
int amps_recc_rccl_unique_id(uint8_t id[AMPS_RECC_RCCL_ID_BYTES]);
int amps_recc_rccl_init(amps_recc_t *h, const uint8_t id[AMPS_RECC_RCCL_ID_BYTES], int nranks, int rank);
int amps_recc_rccl_set_timeout(amps_recc_t *h, uint32_t milliseconds);
int amps_recc_rccl_abort(amps_recc_t *h);
/* How the step's block travels root -> everybody:
 *   BROADCAST          one flat ncclBroadcast: every xGMI link out of the root carries the whole block B;
 *   SCATTER_ALLGATHER  the root sends rank k its N-th of the block (ncclSend / ncclRecv in one group), then one ncclAllGather, in place:
 *                      B/N per link and phase (~4x less time at N = 8).
 * Both deliver the same bytes: the records do not depend on the mode. */
#define AMPS_RECC_DIST_BROADCAST         0
#define AMPS_RECC_DIST_SCATTER_ALLGATHER 1
/* Every rank in step, with the same root and mode.  The root passes its block (`mem` says where it lives) and its size; the other
 * ranks' iq / nsamp are ignored (NULL, 0): the ROOT's nsamp is what every rank pushes -- it travels in the header, so callers whose
 * block sizes cannot be agreed beforehand (GNU Radio schedulers in different processes) need no side channel -- and comes back in
 * *npushed (may be NULL).  The block is distributed on a stream of the library's own into one of two receive buffers and pushed
 * through the wideband seam of every rank (amps_recc_push_wideband on the received block), the collective of push i beside the
 * kernels of push i - 1.  Ownership of the root's block: a HOST block has been staged when the call returns and is free; a DEVICE
 * block is read in place, behind everything enqueued on the handle's stream so far (so amps_recc_wait_event orders it behind its
 * producer), and may be overwritten by work that is ordered behind a LATER call on this handle (amps_recc_record_event) or after a
 * drain that covers this push.  Records are drained per rank as ever (or by amps_recc_drain_gather) and carry whole-band channel numbers.
 * END OF STREAM: the root passing (iq = NULL, nsamp = 0) says it has no more samples -- every rank returns -ENODATA from that call, no
 * data collective runs, nothing is pushed, the communicator stays up (a later call with samples continues the stream).  Ranks whose
 * own sources end at other times than the root's keep calling until they see it.
 * Errors: the root's -EINVAL (a block without samples or samples without a block, unknown mode) / -E2BIG (beyond the smallest rank's
 * capacity) with -EREMOTEIO on the other ranks; -EINVAL everywhere when the ranks pass different modes; -ETIMEDOUT / -ENOTCONN / -EIO:
 * see above. */
int amps_recc_push_wideband_dist(amps_recc_t *h, const float *iq, size_t nsamp, int mem, int root, int mode, size_t *npushed);
/* = amps_recc_push_wideband_dist(h, iq, nsamp, mem, root, AMPS_RECC_DIST_BROADCAST, NULL) */
int amps_recc_push_wideband_bcast(amps_recc_t *h, const float *iq, size_t nsamp, int mem, int root);
/* The collective drain that goes with it (the burst records back to one place): every rank calls it in step; each
 * drains its own list as amps_recc_drain does (no split drain may be open) and the records of all ranks arrive at `root`, merged
 * and sorted by (channel, position) -- what one whole-band handle would have returned.  The other ranks get *nout = 0 and may pass
 * out = NULL, cap = 0.  -ENOSPC on every rank if any rank's list overflowed max_bursts (and on the root if cap is too small; what
 * fits is returned), -EIO (or the rank's own error) on every rank if any rank's drain failed -- that rank still takes part and says
 * so in the status word.  The header exchange ({status, count}), then one ncclAllGather of the lists padded to the longest; the wait
 * for the handle's own kernels (which wait for the data collectives) is bounded like every other: -ETIMEDOUT. */
int amps_recc_drain_gather(amps_recc_t *h, amps_recc_burst_t *out, size_t cap, size_t *nout, int root);
/* What the communicator of this handle is, as RCCL itself reports it, and on which device -- enough for a record of an N-GPU run to
 * prove N distinct devices and N ranks by itself (a handle without a communicator: nranks = 0, the device fields are filled).  collective_* : the data collectives of push_wideband_dist timed with HIP events
 * on the library's stream (handles with kernel timing on: AMPS_RECC_FLAG_TIME_KERNELS / amps_recc_set_timing), bytes = block sizes. */
typedef struct amps_recc_rccl_info {
    uint32_t struct_size;
    int32_t  alive;                 /* 1: communicator usable; 0: aborted (timeout / amps_recc_rccl_abort)            */
    int32_t  nranks, rank;          /* as passed to amps_recc_rccl_init                                                */
    int32_t  comm_nranks, comm_rank;/* ncclCommCount / ncclCommUserRank of the communicator (-1: not available)       */
    int32_t  device;                /* HIP device ordinal of the handle                                                */
    int32_t  pci_domain, pci_bus, pci_device;
    uint8_t  device_uuid[16];       /* hipDeviceProp_t::uuid                                                           */
    uint64_t max_samples_per_push;  /* common capacity of the ranks' handles (samples per block)                      */
    uint32_t max_bursts_per_gather; /* longest record list of any rank                                                 */
    uint32_t timeout_ms;
    uint64_t collectives_timed;     /* data collectives whose events have been read                                    */
    double   collective_ms;         /* their summed duration                                                           */
    uint64_t collective_bytes;      /* their summed block bytes                                                        */
    int32_t  last_mode;             /* AMPS_RECC_DIST_* of the last push, -1 before the first                          */
    int32_t  _pad;
    char     library[96];           /* the name librccl was loaded by                                                  */
} amps_recc_rccl_info_t;
int amps_recc_rccl_info(amps_recc_t *h, amps_recc_rccl_info_t *info);

/* test tap of slicer spec D's bit logic, evaluated ON THE HOST by the very functions the kernels inline (no device needed):
 *   form 0: the streaming kernel's 32-sample window, oldest sample at bit 0: in = {SX, ST, SC}; out[0] = the slicer bits, exact from bit
 *           `sps` on (sps = any supported samples-per-symbol);
 *   form 1: the filter bank's word, newest frame at bit 0, 3 frames per symbol: in = {SX, ST, SC, SX of the previous 32 frames, wp and wm
 *           of the previous call}; out = {slicer bits, wp, wm}.
 * 0, or -EINVAL. */
int amps_recc_debug_exact_slice(int form, int sps, const uint32_t *in, uint32_t *out);

/* test tap of the channelizer seam: channelise nsamp wideband samples (continuing the handle's wideband
 * stream) WITHOUT running the RECC kernels; out is host memory [n_channels][out_ld] fc32, *nframes the
 * number of output samples per channel produced. */
int amps_recc_debug_channelize(amps_recc_t *h, const float *iq, size_t nsamp, int mem,
                               float *out, size_t out_ld, size_t *nframes);

int amps_recc_get_timing(amps_recc_t *h, amps_recc_timing_t *t, int reset);
/* HIP-event timing of the launches: OFF, ALL kernels (what AMPS_RECC_FLAG_TIME_KERNELS selects at creation), or only the
 * DOMINANT streaming kernel of the seam in use (front kernel; channelizer on the wideband seam) -- two event records
 * per push instead of ten, for timed regions that should not be perturbed -- or the dominant kernel of every
 * AMPS_RECC_TIMING_SAMPLE_PERIOD-th push only (DOMINANT_SAMPLED: the two event records cost a wideband step 6.5 us = 1.9 %,
 * profiles/r06/event_cost.txt).  Synchronises the stream. */
#define AMPS_RECC_TIMING_OFF      0
#define AMPS_RECC_TIMING_ALL      1
#define AMPS_RECC_TIMING_DOMINANT 2
#define AMPS_RECC_TIMING_DOMINANT_SAMPLED 3
#define AMPS_RECC_TIMING_SAMPLE_PERIOD 8
int amps_recc_set_timing(amps_recc_t *h, int mode);

/* reply generation of recc_decode: fills the focc_words / fvc_words payloads the
 * reference would publish for this burst.  Pure host integer code. */
typedef struct amps_recc_reply {
    uint8_t  has_focc;  int32_t focc_stream; int32_t focc_nwords; uint8_t focc_word1[28]; uint8_t focc_word2[28];
    uint8_t  has_fvc;   int32_t fvc_count;   uint8_t fvc_word1[28]; uint64_t fvc_repeat;
    uint8_t  has_mutes; uint8_t fvc_mute;    uint8_t audio_mute;
    uint8_t  has_command; char command[48];
} amps_recc_reply_t;
int amps_recc_reply_words(const amps_recc_burst_t *burst, amps_recc_reply_t *reply);

/* BCH(63,51,t=2) shortened to (k+12,k) on the device, one code word per lane: k = 36 -> RECC (48,36),
 * k = 28 -> FOCC/FVC (40,28) as encoded by focc_impl::focc_bch / fvc_impl::fvc_bch (lib/focc_impl.cc:156-176,
 * lib/fvc_impl.cc:98-107).  Bits are one byte each, MSB first; arrays are [nwords][k] / [nwords][k+12], host or
 * device per `mem`; results are host arrays.  valid[i] = 1 iff the word decodes with no correction inside the
 * 63-(k+12) shortening positions. */
int amps_bch_encode_words(amps_recc_t *h, const uint8_t *msg, size_t nwords, int k, int mem, uint8_t *codewords);
int amps_bch_decode_words(amps_recc_t *h, const uint8_t *codewords, size_t nwords, int k, int mem,
                          uint8_t *msg, uint8_t *valid, uint8_t *nerrors);

#ifdef __cplusplus
}
#endif
#endif /* AMPS_RECC_H */
