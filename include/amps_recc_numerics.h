/* amps_recc_numerics.h -- the numeric specification of the fused IQ seam (part of the C ABI contract).
 *
 * The float stages of the reference chain are stock GNU Radio blocks (quadrature_demod_cf with
 * fast_atan2f, grc/recctest.grc:458) whose source is not part of the reference tree.  The fused
 * MI355X path defines its own FM discriminator arithmetic; it is specified here operation by
 * operation in IEEE-754 binary32 so that the device kernel and the CPU model under oracle/ produce
 * bit-identical intermediates (both are compiled with floating-point contraction off and use
 * explicit fmaf):
 *
 *   re = fmaf(xr, pr, xi*pi)          t = x[n] * conj(x[n-1])
 *   im = fmaf(xi, pr, -(xr*pi))
 *   ax = |re|  ay = |im|   mx = max(ax, ay, 2^-100)   mn = min(ax, ay)
 *   r  = as_float(0x7EF311C7 - as_uint(mx))            reciprocal seed (12 % error), then three
 *   e  = fmaf(-mx, r, 1); r = fmaf(r, e, r)   (x3)     Newton steps -> |r*mx - 1| < 1e-7
 *   q  = mn * r                                         (no hardware rcp/div: their rounding is not
 *                                                        reproducible on a CPU, this sequence is;
 *                                                        x = 0 gives mn = 0 -> q = 0 -> d = 0)
 *   z  = q*q
 *   p  = C5; p = fmaf(p,z,C4); ... ; p = fmaf(p,z,C0)        (Horner)
 *   a  = p*q
 *   if (ay > ax) a = PI_2 - a
 *   if (re < 0)  a = PI   - a
 *   a  = copysign(a, im)                               (atan2 sign convention, -0 included)
 *   d[n] = a                                          |d[n] - atan2(im,re)| <= 4e-6 rad
 *
 *   boxcar over one Manchester symbol (sps samples ending at n), summed oldest to newest over
 *   aligned pair sums  p[m] = d[2m] + d[2m+1]  (m = absolute sample index / 2):
 *     the window [n-sps+1, n] is cut into: a leading single d[n-sps+1] if that index is odd,
 *     then every aligned pair inside the window, then a trailing single d[n] if n is even;
 *     S[n] = ((first + second) + third) + ...   in that order
 *   g[n] = S[n] >= 0 ? 1 : 0                           binary_slicer_fb semantics (x >= 0 -> 1)
 *
 * Samples before the start of the stream are zero (x = 0 -> d = 0 -> g = 1).
 *
 * ---- slicer spec B, "product detector" (AMPS_RECC_FLAG_SLICER_PRODUCT) ----
 * The boxcar of spec A sums the sps phase increments that end at n; that sum telescopes:
 *     S[n] = sum_{k=n-sps+1..n} arg(x[k] conj(x[k-1]))  ==  arg(x[n] conj(x[n-sps]))   (mod 2 pi)
 * so as long as |S[n]| < pi -- the phase advance over one Manchester symbol at the +-8 kHz AMPS
 * deviation is 2.51 rad -- the slicer bit is the sign of Im(x[n] conj(x[n-sps])).  Spec B computes
 * exactly that and nothing else (no arctangent, no reciprocal, no boxcar):
 *
 *   a = xi[n] * xr[n-sps]              binary32 products, round to nearest even
 *   b = xr[n] * xi[n-sps]
 *   s = a - b                          binary32 subtraction
 *   g[n] = signbit(s) ? 0 : 1          (-0 counts as negative)
 *   g[n] = 1 for the first sps samples of a stream (no partner yet), as in spec A
 *
 * Inputs must be finite with |x| < 2^60 (no overflow to inf - inf).  Spec B produces no FM-demod float
 * intermediate; d[n] of spec A stays available through amps_recc_debug_demod on a spec-A handle.
 * The two specs give the same bit wherever |S[n]| < pi and neither statistic is within rounding of
 * zero; they differ on phase wraps (noise without carrier, low SNR: the margin to the wrap is only
 * pi - 2.51 = 0.63 rad, so spec B loses bursts from ~12 dB SNR down while spec A holds to ~6 dB).
 *
 * ---- slicer spec C, "sine discriminator" (AMPS_RECC_FLAG_SLICER_SINE) ----
 * Spec A with the arctangent left out: the discriminator output is the imaginary part of the
 * conj-product itself, d'[n] = |x[n]| |x[n-1]| sin(phase increment), i.e. the classic polar
 * discriminator; everything after it is spec A's boxcar, and only the sign leaves the kernel:
 *
 *   d'[n] = fmaf(xi, pr, -(xr*pi))                      the `im` of spec A, p = x[n-1]
 *   S'[n] = spec A's ordered aligned-pair boxcar over d'[n-sps+1 .. n]
 *   g[n]  = signbit(S'[n]) ? 0 : 1                      (x = 0 -> d' = +0 -> g = 1)
 *
 * No wrap can occur inside the window (every increment is below pi), low-amplitude (noisy) samples
 * weigh less, and the cost is 2 flops per sample instead of ~25.  Measured on the synthetic bursts
 * it decodes like spec A down to 8 dB SNR.  d' is not an FM-demod float in radians; the
 * stated-tolerance intermediate of spec A stays available through amps_recc_debug_demod.
 *
 * ---- slicer spec D, "exact-sign discriminator" (AMPS_RECC_FLAG_SLICER_EXACT) ----
 * Only the SIGN of spec A's boxcar sum S[n] = sum_{k=n-sps+1..n} theta[k], theta[k] = arg(x[k] conj(x[k-1])) in (-pi, pi], leaves
 * the float stage, and that sign can be had exactly without evaluating one arctangent.  With a[k] = arg x[k] (principal value):
 *     theta[k]        = a[k] - a[k-1] - 2 pi w'[k]          w'[k]  in {-1, 0, +1}: did the phase step cross the +-pi cut
 *     a[n] - a[n-sps] = phi[n]        + 2 pi w''[n]         phi[n] = arg(x[n] conj(x[n-sps])), the statistic of spec B
 *  => S[n] = phi[n] + 2 pi K[n],     K[n] = w''[n] - sum_{k=n-sps+1..n} w'[k]      (the winding number spec B ignores)
 *  => S[n] >= 0  <=>  K[n] > 0, or K[n] == 0 and phi[n] >= 0.
 * A difference of two principal values exceeds +pi only from the lower half plane to the upper one and only if the conj-product
 * then has a negative imaginary part (and mirrored for -pi), so every wrap is a function of three sign bits:
 *
 *   it = fmaf(xi[n], xr[n-1],   -(xr[n] * xi[n-1]))      Im(x[n] conj(x[n-1]))    -- the d' of spec C
 *   ic = fmaf(xi[n], xr[n-sps], -(xr[n] * xi[n-sps]))    Im(x[n] conj(x[n-sps]))  -- the statistic of spec B (as one fma)
 *   sx[n] = signbit(xi[n])   st = signbit(it)   sc = signbit(ic)
 *   wp[n] = !sx[n] &  sx[n-1]   &  st        (w'[n]  = +1)        up = !sx[n] &  sx[n-sps] &  sc     (w''[n] = +1)
 *   wm[n] =  sx[n] & !sx[n-1]   & !st        (w'[n]  = -1)        um =  sx[n] & !sx[n-sps] & !sc     (w''[n] = -1)
 *   K     = (up + sum_{j<sps} wm[n-j]) - (um + sum_{j<sps} wp[n-j])
 *   g[n]  = (K > 0) | (K == 0 & !sc)
 *   g[n]  = 1 for the first sps samples of a stream (no partner yet), as in spec B; samples before the stream are +0
 *           (sx = 0, wp = wm = 0)
 *
 * Four multiply-adds and three sign bits per sample; the rest is bitwise logic on 32 samples at a time.  In exact arithmetic
 * g[n] equals spec A's bit with an ideal arctangent; in binary32 it equals the bit of the float64 libm discriminator wherever
 * |S[n]| exceeds the rounding of the two products (tests/test_gpu_slicer_specs.py holds it to: no difference at |S| > 1e-4 rad,
 * against spec A's own 1e-5 rad per sample) -- so spec D keeps spec A's sensitivity (no wrap, no amplitude weighting) at spec C's
 * cost.  The FM-demod float d[n] in radians is not materialised; it stays available through amps_recc_debug_demod on a spec-A
 * handle, and the two floats spec D does compute (it, ic) are what its debug taps return.
 *
 * ---- which spec is the default, and what the others cost in sensitivity ----
 * AMPS_SLICER_DEFAULT = spec D (round 4; rounds 1-3: spec A).  scripts/slicer_sensitivity.py (profiles/r04/slicer_sensitivity.txt;
 * 1000 bursts per point on the IQ seam, 1248 on the wideband seam, C/N stated in a 30 kHz channel) gives the C/N at which 1 % of the
 * seizure bursts are lost:
 *
 *                          spec A     spec B            spec C            spec D            restated reference chain (M&M timing)
 *   IQ seam (behind the    10.23 dB   10.10 dB (-0.1)   11.20 dB (+1.0)   10.23 dB (0.00)   24.8 dB  (its loop must lock inside the
 *   flow graph's 299-tap                                                                             four spare dotting bits)
 *   channel filter)
 *   wideband seam           9.64 dB   11.68 dB (+2.0)   13.30 dB (+3.7)    9.64 dB (0.00)   24.3 dB
 *
 * Spec D's loss column equals spec A's burst for burst (same seeds): it makes the same decisions except within rounding of S = 0.
 * It costs the filter-bank kernel 0.42 ms per GiB against 0.51-0.53 for spec A (0.40 for B / C), so it is the default: spec A's
 * sensitivity without the arctangent.  Spec A stays selectable (AMPS_RECC_FLAG_SLICER_ATAN) and is the spec whose FM-demod float
 * d[n] is held to AMPS_DEMOD_TOL_RAD against libm; specs B and C stay opt-in: C costs 3.7 dB on the wideband seam (three samples
 * per symbol: the amplitude-weighted sine of a 0.84 rad step is a poorer statistic than the angle itself), B is free on
 * band-limited input but wraps on white noise (8.1 dB -> 15.0 dB at 200 ksps without a channel filter) and under a carrier offset
 * (+-2 kHz uses up its whole margin to the wrap).  Undetected wrong words (flagged valid, different from what was sent) stay below
 * 1e-3 of the valid words for every spec from 10 dB up.
 */
#ifndef AMPS_RECC_NUMERICS_H
#define AMPS_RECC_NUMERICS_H

/* minimax fit of atan(q)/q in z = q*q on [0,1]; max abs error of the binary32 Horner form 1.75e-6 rad */
#define AMPS_ATAN_C0  0x1.fffd04p-1f
#define AMPS_ATAN_C1 -0x1.549b12p-2f
#define AMPS_ATAN_C2  0x1.8c5ed6p-3f
#define AMPS_ATAN_C3 -0x1.dce1c0p-4f
#define AMPS_ATAN_C4  0x1.af48f4p-5f
#define AMPS_ATAN_C5 -0x1.800270p-7f
#define AMPS_RCP_MAGIC 0x7EF311C7u
#define AMPS_MX_FLOOR  0x1p-100f
#define AMPS_PI_F     0x1.921fb6p+1f
#define AMPS_PI_2_F   0x1.921fb6p+0f

/* stated tolerance of the FM-demod float intermediate against libm atan2 (radians) */
#define AMPS_DEMOD_TOL_RAD 1.0e-5f

/* fused-seam geometry shared by kernel, host code and CPU model */
#define AMPS_TILE_SAMPLES   512    /* samples per wavefront tile (64 lanes x 8 samples)            */
#define AMPS_HALO_SAMPLES   1024   /* history recomputed at the head of every chunk / kept per push */
#define AMPS_WORD_SAMPLES   64     /* samples per packed slicer word                              */
#define AMPS_DEDUP_SYMBOLS  2      /* trigger hits closer than this many symbols form one run     */
#define AMPS_TRACK_BLOCKS   36     /* timing-tracking blocks of a burst: the trigger's 37 bits, the coded DCC with the first repeat,
                                      then 34 repeats of 48 bits; also the most the sampling instants can move (one sample per block) */

/* slicer specs */
#define AMPS_SLICER_ATAN_BOXCAR 0  /* spec A: discriminator + boxcar (the default of rounds 1-3)      */
#define AMPS_SLICER_PRODUCT     1  /* spec B: sign of Im(x[n] conj(x[n-sps]))                        */
#define AMPS_SLICER_SINE        2  /* spec C: boxcar over Im(x[n] conj(x[n-1])), no arctangent       */
#define AMPS_SLICER_EXACT       3  /* spec D: sign of spec A's boxcar sum from sign bits and the winding number, no arctangent */
#define AMPS_SLICER_DEFAULT     AMPS_SLICER_EXACT        /* what a handle created with no SLICER flag uses (amps_recc_default_slicer) */

#endif
