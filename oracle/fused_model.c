/* fused_model.c -- TEST INFRASTRUCTURE ONLY (see amps_oracle.h).
 *
 * Scalar CPU model ("port") of the fused MI355X seam: the arithmetic of
 * include/amps_recc_numerics.h and the detection / capture rules of DESIGN.md section 4,
 * written as plain sequential loops.  It shares no code with the HIP kernels; the burst decode it
 * ends in is the reference restatement (orc_decode_burst).  Used (a) to check the kernels bit for
 * bit, (b) as the `cpu_baseline` "port" leg of bench.py together with the reference chain.
 */
#include "amps_oracle.h"
#include "amps_recc_numerics.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* d = arg(x * conj(p)) per the numeric spec */
static inline float fm_phase(float xr, float xi, float pr, float pi_)
{
    float re = fmaf(xr, pr, xi * pi_);
    float im = fmaf(xi, pr, -(xr * pi_));
    float ax = fabsf(re), ay = fabsf(im);
    float mx = fmaxf(fmaxf(ax, ay), AMPS_MX_FLOOR), mn = fminf(ax, ay);
    union { float f; uint32_t u; } cv;
    cv.f = mx;
    cv.u = AMPS_RCP_MAGIC - cv.u;
    float r = cv.f, e;
    e = fmaf(-mx, r, 1.0f); r = fmaf(r, e, r);
    e = fmaf(-mx, r, 1.0f); r = fmaf(r, e, r);
    e = fmaf(-mx, r, 1.0f); r = fmaf(r, e, r);
    float q = mn * r;
    float z = q * q;
    float p = AMPS_ATAN_C5;
    p = fmaf(p, z, AMPS_ATAN_C4);
    p = fmaf(p, z, AMPS_ATAN_C3);
    p = fmaf(p, z, AMPS_ATAN_C2);
    p = fmaf(p, z, AMPS_ATAN_C1);
    p = fmaf(p, z, AMPS_ATAN_C0);
    float a = p * q;
    if (ay > ax) a = AMPS_PI_2_F - a;
    if (re < 0.0f) a = AMPS_PI_F - a;
    return copysignf(a, im);
}

void orc_fm_discriminator(const float *iq, size_t n, float *d)
{
    float pr = 0.0f, pi_ = 0.0f;
    for (size_t i = 0; i < n; i++) {
        d[i] = fm_phase(iq[2 * i], iq[2 * i + 1], pr, pi_);
        pr = iq[2 * i]; pi_ = iq[2 * i + 1];
    }
}

struct orc_fused {
    uint32_t channel;
    int      sps;
    size_t   n_in, cap;       /* samples received so far / allocated                      */
    size_t   n_done;          /* samples processed so far (multiple of 64)                 */
    float   *x;               /* 2*cap interleaved IQ                                      */
    float   *d, *S;           /* demod and boxcar, valid for [0, n_done)                   */
    uint8_t *g, *M;           /* slicer bits and trigger-hit bits, valid for [0, n_done)   */
    uint8_t *sx, *wp, *wm;    /* spec D: sign of Im x and the two wrap bits per sample      */
    size_t   wcap;
    uint64_t next_allowed;    /* run starts below this are inside an accepted burst        */
    int      have_pending;
    uint64_t pending_nc;
    uint8_t  trig[AMPS_RECC_TRIGGER_SYMS];
    int      tol;             /* accepted mismatching symbols of the 74 (0 = exact match = reference behaviour) */
    int      majority;        /* decode captures in the product's majority mode (AMPS_RECC_FLAG_MAJORITY)       */
    int      slicer;          /* AMPS_SLICER_* of include/amps_recc_numerics.h                                   */
    int      track;           /* per-repeat timing tracking in the capture (AMPS_RECC_FLAG_FIXED_TIMING clears it)          */
};

orc_fused_t *orc_fused_new(uint32_t channel, int sps)
{
    orc_fused_t *f = (orc_fused_t *)calloc(1, sizeof(*f));
    f->channel = channel; f->sps = sps;
    f->slicer = AMPS_SLICER_DEFAULT;
    f->track = 1;
    orc_trigger(f->trig);
    return f;
}
void orc_fused_free(orc_fused_t *f)
{
    if (!f) return;
    free(f->x); free(f->d); free(f->S); free(f->g); free(f->M); free(f->sx); free(f->wp); free(f->wm); free(f);
}
void orc_fused_set_tolerance(orc_fused_t *f, int k) { f->tol = k < 0 ? 0 : k; }
void orc_fused_set_majority(orc_fused_t *f, int on) { f->majority = on != 0; }
void orc_fused_set_slicer(orc_fused_t *f, int spec) { f->slicer = spec; }
void orc_fused_set_tracking(orc_fused_t *f, int on) { f->track = on != 0; }
size_t orc_fused_processed(const orc_fused_t *f) { return f->n_done; }
const float *orc_fused_demod(const orc_fused_t *f) { return f->d; }
const float *orc_fused_soft(const orc_fused_t *f) { return f->S; }
const uint8_t *orc_fused_hard(const orc_fused_t *f) { return f->g; }

static void grow(orc_fused_t *f, size_t need)
{
    if (need <= f->cap) return;
    size_t nc = f->cap ? f->cap : 4096;
    while (nc < need) nc *= 2;
    f->x = (float *)realloc(f->x, sizeof(float) * 2 * nc);
    f->d = (float *)realloc(f->d, sizeof(float) * nc);
    f->S = (float *)realloc(f->S, sizeof(float) * nc);
    f->g = (uint8_t *)realloc(f->g, nc);
    f->M = (uint8_t *)realloc(f->M, nc);
    f->cap = nc;
}

static void grow_wraps(orc_fused_t *f)
{
    if (f->wcap >= f->cap) return;
    f->sx = (uint8_t *)realloc(f->sx, f->cap);
    f->wp = (uint8_t *)realloc(f->wp, f->cap);
    f->wm = (uint8_t *)realloc(f->wm, f->cap);
    f->wcap = f->cap;
}

static inline int gbit(const orc_fused_t *f, int64_t n) { return n < 0 ? 1 : f->g[n]; }

/* Symbol i of a capture is slicer bit nc + sps (i + 1) + delay(block of i).  With timing tracking (the default) the
 * burst is walked in 36 blocks -- the 37 bits of the trigger itself (measured only: they lie in front of the capture), then the
 * coded DCC together with the first repeat (55 bits), then the other 34 repeats of 48 bits -- and after each block the sampling
 * instants of everything behind it move by one sample if the mid-bit transitions of that block sat, on average, more than half a
 * sample late or early.  For a Manchester pair (a, b), a != b, sampled at t and t + sps the boxcar output changes sign half way
 * between, so the number of slicer bits equal to a among t+1 .. t+sps-1 says where the transition really was: e = cnt - (sps-1)/2
 * samples late.  A carrier offset moves the falling transitions one way and the rising ones the other, so the two polarities are
 * averaged separately and a block with only one of them measures nothing:
 *     move by +1 if mean(e | a = 1) + mean(e | a = 0) >  1,   by -1 if < -1        (integers: E1 n0 + E0 n1 vs 2 n0 n1, E = sum 2e)
 * A first-order timing loop on hard decisions, one step per 96 symbols: it follows a mobile whose bit clock is off by up to
 * ~1/(96 sps) (3400 ppm at 3 samples per symbol, 1000 ppm at 10; TIA-553 allows 100) and never moves on an exact clock. */
static void capture(orc_fused_t *f, uint64_t nc, amps_recc_burst_t *out)
{
    uint8_t burst[AMPS_RECC_CAPTURE_SYMS];
    const int sps = f->sps;
    int dly = 0, k0 = -(AMPS_RECC_TRIGGER_SYMS / 2);          /* bit index relative to the capture: the trigger is bits -37 .. -1 */
    for (int b = 0; b < AMPS_TRACK_BLOCKS; b++) {
        const int nb = b == 0 ? AMPS_RECC_TRIGGER_SYMS / 2 : b == 1 ? 7 + AMPS_RECC_WORD_BITS : AMPS_RECC_WORD_BITS;
        if (sps == 2) {
            /* Two samples per symbol (the wideband seam at D = 768): ONE slicer bit lies between the two instants of a pair, which
             * says which side of it the transition was on but not how far -- no first-order loop can be steered by that.  What the
             * hard bits do carry is the number of Manchester violations (pairs a == b): the block is taken at whichever of the delays
             * d - 1, d, d + 1 (d = the block before; 0 in front of the trigger) shows the fewest of them in THIS block, d on a
             * tie, then d - 1.  (The other sample phase one symbol early or late pairs symbols across bit boundaries and violates in
             * every second bit, so at most one neighbour is ever a candidate.)  The block that decides is the block that is taken:
             * the first repeat of word A -- which the reference parses uncorrected -- is already sampled at the better phase. */
            if (f->track) {
                int v[3] = { 0, 0, 0 };
                for (int c = 0; c < 3; c++)
                    for (int k = k0; k < k0 + nb; k++) {
                        const int64_t ta = (int64_t)nc + (int64_t)sps * (2 * k + 1) + dly + c - 1;
                        v[c] += gbit(f, ta) == gbit(f, ta + sps);
                    }
                int best = 1;
                if (v[0] < v[1]) best = 0;
                if (v[2] < v[1] && v[2] < v[best]) best = 2;
                dly += best - 1;
            }
            for (int k = k0 < 0 ? 0 : k0; k < k0 + nb; k++) {
                const int64_t ta = (int64_t)nc + (int64_t)sps * (2 * k + 1) + dly;
                burst[2 * k] = (uint8_t)gbit(f, ta); burst[2 * k + 1] = (uint8_t)gbit(f, ta + sps);
            }
            k0 += nb;
            continue;
        }
        int E[2] = { 0, 0 }, n[2] = { 0, 0 };
        for (int k = k0; k < k0 + nb; k++) {
            /* (samples in front of the stream read 1, as they do in the trigger test: only a TOLERANT match can begin there -- the
             * trigger's first symbol is 0 -- and then only block 0, which is measured and not captured, looks at them) */
            const int64_t ta = (int64_t)nc + (int64_t)sps * (2 * k + 1) + dly;
            const int a = gbit(f, ta), bb = gbit(f, ta + sps);
            if (k >= 0) { burst[2 * k] = (uint8_t)a; burst[2 * k + 1] = (uint8_t)bb; }
            if (f->track && a != bb) {
                int cnt = 0;
                for (int m = 1; m < sps; m++) cnt += gbit(f, ta + m) == a;
                E[a] += 2 * cnt - (sps - 1);
                n[a]++;
            }
        }
        const int lhs = E[1] * n[0] + E[0] * n[1], rhs = 2 * n[0] * n[1];
        if (rhs > 0) { if (lhs > rhs) dly++; else if (lhs < -rhs) dly--; }
        k0 += nb;
    }
    orc_decode_burst_mode(burst, f->channel, nc, out, f->majority);
}
/* samples that must follow n_c before its capture is taken: one symbol more than the capture (the reference's strict '>',
 * lib/recc_impl.cc:125), plus -- with tracking -- the most the sampling instants can have moved */
static uint64_t span_done(const orc_fused_t *f)
{
    return (uint64_t)f->sps * (AMPS_RECC_CAPTURE_SYMS + 1) + (f->track ? AMPS_TRACK_BLOCKS : 0);
}

size_t orc_fused_push(orc_fused_t *f, const float *iq, size_t n, amps_recc_burst_t *out, size_t cap)
{
    const int sps = f->sps, T = AMPS_RECC_TRIGGER_SYMS;
    const int D = AMPS_DEDUP_SYMBOLS * sps;
    size_t nout = 0;
    grow(f, f->n_in + n + 64);
    memcpy(f->x + 2 * f->n_in, iq, sizeof(float) * 2 * n);
    f->n_in += n;
    size_t P = ((f->n_in - f->n_done) / AMPS_WORD_SAMPLES) * AMPS_WORD_SAMPLES;
    size_t lo = f->n_done, hi = f->n_done + P;
    /* demod, boxcar, slicer */
    if (f->slicer == AMPS_SLICER_PRODUCT) {
        /* spec B: g[n] = !signbit(xi[n] xr[n-sps] - xr[n] xi[n-sps]); no demod float exists (d = 0, S = the statistic) */
        for (size_t i = lo; i < hi; i++) {
            f->d[i] = 0.0f;
            if (i < (size_t)sps) { f->S[i] = 0.0f; f->g[i] = 1; continue; }   /* no partner yet: g = 1 by definition */
            const float pr = f->x[2 * (i - sps)], pi_ = f->x[2 * (i - sps) + 1];
            volatile float a = f->x[2 * i + 1] * pr, b = f->x[2 * i] * pi_;   /* volatile: two rounded products, never an fma */
            const float s = a - b;
            f->S[i] = s;
            f->g[i] = signbit(s) ? 0 : 1;
        }
    } else if (f->slicer == AMPS_SLICER_EXACT) {
        /* spec D: the sign of spec A's boxcar sum without the arctangent -- S[n] = phi[n] + 2 pi K[n], the winding number K from
         * sign bits only (include/amps_recc_numerics.h).  d = Im(x conj(x[n-1])), S = Im(x conj(x[n-sps])): the two float
         * intermediates that exist; the wrap bits of sample i live in f->M's sibling arrays below */
        grow_wraps(f);
        for (size_t i = lo; i < hi; i++) {
            const float xr = f->x[2 * i], xi = f->x[2 * i + 1];
            const float pr = i ? f->x[2 * (i - 1)] : 0.0f, pi_ = i ? f->x[2 * (i - 1) + 1] : 0.0f;
            const float qr = i >= (size_t)sps ? f->x[2 * (i - sps)] : 0.0f, qi = i >= (size_t)sps ? f->x[2 * (i - sps) + 1] : 0.0f;
            const float it = fmaf(xi, pr, -(xr * pi_));
            const float ic = fmaf(xi, qr, -(xr * qi));
            const int sx = signbit(xi) != 0, st = signbit(it) != 0, sc = signbit(ic) != 0;
            const int sx1 = i ? f->sx[i - 1] : 0, sxs = i >= (size_t)sps ? f->sx[i - sps] : 0;
            f->sx[i] = (uint8_t)sx;
            f->wp[i] = (uint8_t)(!sx && sx1 && st);        /* arg x[i] - arg x[i-1] wrapped past +pi */
            f->wm[i] = (uint8_t)(sx && !sx1 && !st);       /* ... past -pi */
            int K = (!sx && sxs && sc) - (sx && !sxs && !sc);
            for (int j = 0; j < sps; j++) if (i >= (size_t)j) K += (int)f->wm[i - j] - (int)f->wp[i - j];
            f->d[i] = it;
            f->S[i] = ic;
            f->g[i] = i < (size_t)sps ? 1 : (uint8_t)(K > 0 || (K == 0 && !sc));   /* no partner yet: g = 1, as in spec B */
        }
    } else
    for (size_t i = lo; i < hi; i++) {
        float pr = i ? f->x[2 * (i - 1)] : 0.0f, pi_ = i ? f->x[2 * (i - 1) + 1] : 0.0f;
        f->d[i] = f->slicer == AMPS_SLICER_SINE ? fmaf(f->x[2 * i + 1], pr, -(f->x[2 * i] * pi_))     /* spec C: Im(x conj(p)) = |x||p| sin(d) */
                                 : fm_phase(f->x[2 * i], f->x[2 * i + 1], pr, pi_);
        /* boxcar per the numeric spec: aligned pair sums, oldest to newest */
        float s = 0.0f;
        int first = 1;
        int64_t lo_n = (int64_t)i - (sps - 1), hi_n = (int64_t)i, m = lo_n;
        while (m <= hi_n) {
            float v;
            if ((m & 1) == 0 && m + 1 <= hi_n) {            /* aligned pair (m even; negative m: both zero) */
                float a0 = m < 0 ? 0.0f : f->d[m], a1 = m + 1 < 0 ? 0.0f : f->d[m + 1];
                v = a0 + a1;
                m += 2;
            } else {
                v = m < 0 ? 0.0f : f->d[m];
                m += 1;
            }
            if (first) { s = v; first = 0; } else s = s + v;
        }
        f->S[i] = s;
        f->g[i] = f->slicer == AMPS_SLICER_SINE ? (signbit(s) ? 0 : 1) : (s >= 0.0f ? 1 : 0);
    }
    /* 74-symbol trigger test ending at sample i: at most `tol` symbols may differ (tol = 0: the reference's exact
     * memmem, lib/recc_impl.cc:118) */
    for (size_t i = lo; i < hi; i++) {
        int bad = 0;
        for (int k = 0; k < T && bad <= f->tol; k++)
            if (gbit(f, (int64_t)i - (int64_t)sps * (T - 1 - k)) != f->trig[k]) bad++;
        f->M[i] = (uint8_t)(bad <= f->tol);
    }
    /* run starts located in word w are examined when word w+1 has been processed */
    int64_t w_lo = (int64_t)(lo / AMPS_WORD_SAMPLES) - 1, w_hi = (int64_t)(hi / AMPS_WORD_SAMPLES) - 1;
    /* previously accepted burst waiting for its tail */
    if (f->have_pending && f->pending_nc + span_done(f) < hi) {
        if (nout < cap) capture(f, f->pending_nc, &out[nout++]);
        f->have_pending = 0;
    }
    for (int64_t w = w_lo < 0 ? 0 : w_lo; w < w_hi; w++) {
        for (int b = 0; b < AMPS_WORD_SAMPLES; b++) {
            int64_t p = w * AMPS_WORD_SAMPLES + b;
            if (!f->M[p]) continue;
            int dup = 0;
            for (int k = 1; k <= D; k++) if (p - k >= 0 && f->M[p - k]) dup = 1;
            if (dup) continue;
            int last = 0;
            for (int k = 0; k < D; k++) if (f->M[p + k]) last = k; /* p+k < (w+2)*64 since D <= 64 */
            uint64_t a = (uint64_t)p;
            if (a < f->next_allowed) continue;
            uint64_t nc = a + (uint64_t)(last / 2);
            f->next_allowed = nc + (uint64_t)sps * (AMPS_RECC_CAPTURE_SYMS + AMPS_RECC_TRIGGER_SYMS);
            if (nc + span_done(f) < hi) {
                if (nout < cap) capture(f, nc, &out[nout++]);
            } else {
                f->have_pending = 1; f->pending_nc = nc;
            }
        }
    }
    f->n_done = hi;
    return nout;
}
