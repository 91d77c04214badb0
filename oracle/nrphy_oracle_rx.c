/* TEST INFRASTRUCTURE -- not part of the product (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it).
 *
 * CPU restatement of the reference's soft demodulator ("demodulation mapper"):
 *   R/lib/phy/upper/channel_modulation/demodulation_mapper_impl.cpp:33-106 (BPSK, pi/2-BPSK, dispatch)
 *   R/lib/phy/upper/channel_modulation/demodulation_mapper_qpsk.cpp:38-77, 124-169
 *   R/lib/phy/upper/channel_modulation/demodulation_mapper_qam16.cpp:40-120, 192-273
 *   R/lib/phy/upper/channel_modulation/demodulation_mapper_qam64.cpp:49-92, 206-300, 398-463
 *   R/lib/phy/upper/channel_modulation/demodulation_mapper_qam256.cpp:47-160, 213-262, 339-427
 *   R/lib/phy/upper/channel_modulation/demodulation_mapper_intervals.h:34-64, avx2_helpers.h:103-262
 *   R/lib/phy/upper/log_likelihood_ratio.cpp:89-98 (quantize)
 *
 * The reference's result depends on the position of a symbol in the span: on x86 with AVX2 (the build the rest of this
 * oracle is pinned to) the first floor(n / B) * B symbols go through the vector code (B = 16 QPSK, 8 16-QAM, 16 64-QAM,
 * 4 256-QAM), the rest through the generic code, and the two differ in arithmetic:
 *   vector : reciprocal of the noise variance (0 when the variance is not > 0), interval index from floor(v * (1 / width)),
 *            quantisation = clip(l * (120 / range)) rounded to nearest even; a component with |v| <= 1e-9 gives zeros for
 *            that component (16/64/256-QAM); NaN -> 0.
 *   generic: division by the noise variance (QPSK, 16-QAM) or multiplication by its reciprocal (64/256-QAM), interval index
 *            from floor(v / width), quantisation = round-half-away(clip(l) / range * 120); a symbol with |z|^2 < 1e-9 gives
 *            zeros for all its bits (16/64/256-QAM).
 * slope * v + intercept is one fused multiply-add in both (GCC contracts it at the reference's flags, -O2 -mfma, C++
 * default -ffp-contract=fast), and so is the generic 16-QAM's 0.8 - g * |x|; tests/test_oracle.py pins all of this against the
 * compiled reference (oracle/_ref), including ties, interval boundaries and non-positive variances.
 * The interval tables are the max-log LLR of Gray-mapped PAM and are derived here, not transcribed: in the interval whose
 * nearest constellation points with the bit 0 / 1 are a0 / a1, LLR(v) = 2 (a0 - a1) v + (a1^2 - a0^2). */
#include "nrphy_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LLR_MAX 120
static const float NEAR_ZERO = 1e-9f;

/* ---- interval tables ------------------------------------------------------------------------------------------- */
typedef struct {
  float    width;      /* interval width */
  float    rcp_width;  /* 1.0F / width, as the vector path computes it */
  unsigned n;          /* number of intervals */
  float    slope[16], intercept[16];
} interval_table;

/* Gray-mapped PAM of TS 38.211 Section 5.1 in units of A = 1/sqrt(norm): m bits per dimension, levels the odd integers of
 * (-2^m, 2^m).  Bit 0 is the sign (1 = negative).  Bit k >= 1 follows from the standard's nested form
 * (1 - 2 b0) [2^(m-1) - (1 - 2 b1) [2^(m-2) - ...]]: with t = |level| in a range (0, R), R = 2^m, the bit is 1 when t lies in
 * the outer half, and the next bit sees t folded around the middle, |t - R/2|, in the range (0, R/2). */
static unsigned pam_bit(int level, unsigned k, unsigned m)
{
  if (k == 0) {
    return level < 0;
  }
  int t = abs(level);
  int r = 1 << m;
  for (unsigned j = 1;; ++j) {
    const unsigned bit = t > r / 2;
    if (j == k) {
      return bit;
    }
    t = abs(t - r / 2);
    r /= 2;
  }
}

static void build_table(interval_table* t, unsigned m, unsigned k, float a /* 1/sqrt(norm) */, unsigned norm, unsigned width_units,
                        unsigned n)
{
  const int M  = 1 << m;
  t->width     = (float)width_units * a;
  t->rcp_width = 1.0f / t->width;
  t->n         = n;
  for (unsigned j = 0; j != n; ++j) {
    /* centre of interval j in units of A */
    const double c = ((double)j - (double)n / 2 + 0.5) * (double)width_units;
    int          best0 = 0, best1 = 0;
    double       d0 = 1e30, d1 = 1e30;
    for (int p = 0; p != M; ++p) {
      const int    level = 2 * p - (M - 1);
      const double d     = fabs(c - level);
      if (pam_bit(level, k, m)) {
        if (d < d1) {
          d1 = d, best1 = level;
        }
      } else if (d < d0) {
        d0 = d, best0 = level;
      }
    }
    t->slope[j]     = (float)(2 * (best0 - best1)) * a;
    t->intercept[j] = (float)(best1 * best1 - best0 * best0) / (float)norm;
  }
}

static interval_table T64[3], T256[4];
static float          SQRT1_10, SQRT1_42, SQRT1_170;
static int            tables_ready;

static void build_tables(void)
{
  if (tables_ready) {
    return;
  }
  SQRT1_10  = 1.0f / sqrtf(10.0f);
  SQRT1_42  = 1.0f / sqrtf(42.0f);
  SQRT1_170 = 1.0f / sqrtf(170.0f);
  /* 64-QAM: bit pairs 01 and 23 on 8 intervals of 2A, 45 on 4 intervals of 4A (demodulation_mapper_qam64.cpp:49-92);
   * the intercepts there are reduced fractions of (a1^2 - a0^2) / 42, the same real numbers. */
  build_table(&T64[0], 3, 0, SQRT1_42, 42, 2, 8);
  build_table(&T64[1], 3, 1, SQRT1_42, 42, 2, 8);
  build_table(&T64[2], 3, 2, SQRT1_42, 42, 4, 4);
  /* 256-QAM: 01, 23, 45 on 16 intervals of 2A, 67 on 8 intervals of 4A (demodulation_mapper_qam256.cpp:47-160). */
  build_table(&T256[0], 4, 0, SQRT1_170, 170, 2, 16);
  build_table(&T256[1], 4, 1, SQRT1_170, 170, 2, 16);
  build_table(&T256[2], 4, 2, SQRT1_170, 170, 2, 16);
  build_table(&T256[3], 4, 3, SQRT1_170, 170, 4, 8);
  tables_ready = 1;
}

void oracle_demod_tables(unsigned qm, unsigned pair, float* width, unsigned* n, float* slope, float* intercept)
{
  build_tables();
  const interval_table* t = qm == 6 ? &T64[pair] : &T256[pair];
  *width                  = t->width;
  *n                      = t->n;
  memcpy(slope, t->slope, sizeof(float) * t->n);
  memcpy(intercept, t->intercept, sizeof(float) * t->n);
}

/* ---- quantisation ------------------------------------------------------------------------------------------------ */
/* log_likelihood_ratio::quantize (log_likelihood_ratio.cpp:89-98). */
static int8_t quantize_generic(float value, float range)
{
  float clipped = value;
  if (fabsf(value) > range) {
    clipped = copysignf(range, value);
  }
  return (int8_t)roundf(clipped / range * (float)LLR_MAX);
}

/* mm256::quantize_ps (avx2_helpers.h:118-170): scale, clip, round to nearest even, NaN -> 0. */
static int8_t quantize_vector(float value, float range)
{
  float v = value * ((float)LLR_MAX / range);
  if (v > (float)LLR_MAX) {
    v = (float)LLR_MAX;
  }
  if (v < -(float)LLR_MAX) {
    v = -(float)LLR_MAX;
  }
  if (isnan(v)) {
    return 0;
  }
  return (int8_t)nearbyintf(v); /* default rounding mode: to nearest even */
}

static float safe_rcp(float noise)
{
  return noise > 0 ? 1.0f / noise : 0.0f;
}

static int clampi(int v, int lo, int hi)
{
  return v < lo ? lo : (v > hi ? hi : v);
}

/* interval function, vector path (avx2_helpers.h:186-262) and generic path (demodulation_mapper_intervals.h:34-64) */
static float interval_vector(const interval_table* t, float v, float rcp_noise)
{
  const int idx = clampi((int)floorf(v * t->rcp_width) + (int)t->n / 2, 0, (int)t->n - 1);
  const float r = fmaf(t->slope[idx], v, t->intercept[idx]) * rcp_noise;
  return fabsf(v) <= NEAR_ZERO ? 0.0f : r;
}
static float interval_generic(const interval_table* t, float v, float rcp_noise)
{
  const int idx = clampi((int)floorf(v / t->width) + (int)t->n / 2, 0, (int)t->n - 1);
  return fmaf(t->slope[idx], v, t->intercept[idx]) * rcp_noise;
}

/* ---- per-modulation ---------------------------------------------------------------------------------------------- */
static int8_t demod_bpsk(float re, float im, float noise)
{
  if (!(noise > 0)) {
    return 0;
  }
  const float gain = 2.0f * 1.41421356237309504880f;
  return quantize_generic(gain * (re + im) / noise, 24.0f);
}

static void demod_qpsk(int8_t* llr, const float* sym, const float* noise, size_t n)
{
  const float  gain = 2.0f * 1.41421356237309504880f;
  const size_t nv   = n / 16 * 16;
  for (size_t i = 0; i != n; ++i) {
    for (unsigned c = 0; c != 2; ++c) {
      const float v = sym[2 * i + c];
      if (i < nv) {
        llr[2 * i + c] = quantize_vector((gain * v) * safe_rcp(noise[i]), 24.0f);
      } else {
        llr[2 * i + c] = !(noise[i] > 0) ? 0 : quantize_generic(gain * v / noise[i], 24.0f);
      }
    }
  }
}

static void demod_qam16(int8_t* llr, const float* sym, const float* noise, size_t n)
{
  const float  g1 = 4.0f * SQRT1_10, thr = 2 * SQRT1_10;
  const size_t nv = n / 8 * 8;
  for (size_t i = 0; i != n; ++i) {
    const float re = sym[2 * i], im = sym[2 * i + 1];
    if (i < nv) {
      const float rcp = safe_rcp(noise[i]);
      for (unsigned c = 0; c != 2; ++c) {
        const float v     = c ? im : re;
        const float first = g1 * v;
        const float l01   = fabsf(v) > thr ? 2.0f * first - copysignf(0.8f, v) : first;
        const float l23   = 0.8f - fabsf(first);
        const int   zero  = fabsf(v) <= NEAR_ZERO;
        llr[4 * i + c]     = quantize_vector(zero ? 0.0f : l01 * rcp, 20.0f);
        llr[4 * i + 2 + c] = quantize_vector(zero ? 0.0f : l23 * rcp, 20.0f);
      }
    } else if (re * re + im * im < NEAR_ZERO) {
      memset(llr + 4 * i, 0, 4);
    } else {
      for (unsigned c = 0; c != 2; ++c) {
        const float v = c ? im : re;
        if (!(noise[i] > 0)) {
          llr[4 * i + c] = llr[4 * i + 2 + c] = 0;
          continue;
        }
        float l = g1 * v;
        if (fabsf(v) > thr) {
          l = 2 * l - copysignf(0.8f, v);
        }
        llr[4 * i + c]     = quantize_generic(l / noise[i], 20.0f);
        llr[4 * i + 2 + c] = quantize_generic(fmaf(-g1, fabsf(v), 0.8f) / noise[i], 20.0f);
      }
    }
  }
}

static void demod_tables(int8_t* llr, const float* sym, const float* noise, size_t n, const interval_table* t, unsigned pairs,
                         unsigned batch)
{
  const size_t nv = n / batch * batch;
  const unsigned qm = 2 * pairs;
  for (size_t i = 0; i != n; ++i) {
    const float re = sym[2 * i], im = sym[2 * i + 1];
    const float rcp = safe_rcp(noise[i]);
    if (i < nv) {
      for (unsigned p = 0; p != pairs; ++p) {
        llr[qm * i + 2 * p]     = quantize_vector(interval_vector(&t[p], re, rcp), 20.0f);
        llr[qm * i + 2 * p + 1] = quantize_vector(interval_vector(&t[p], im, rcp), 20.0f);
      }
    } else if (re * re + im * im < NEAR_ZERO) {
      memset(llr + qm * i, 0, qm);
    } else {
      for (unsigned p = 0; p != pairs; ++p) {
        llr[qm * i + 2 * p]     = quantize_generic(interval_generic(&t[p], re, rcp), 20.0f);
        llr[qm * i + 2 * p + 1] = quantize_generic(interval_generic(&t[p], im, rcp), 20.0f);
      }
    }
  }
}

/* demodulation_mapper::demodulate_soft.  modulation: NRPHY_MOD_* (0 pi/2-BPSK, 1 BPSK, 2 QPSK, 4 16-QAM, 6 64-QAM, 8 256-QAM);
 * symbols: n complex floats; noise_vars: n floats; llr: n * bits-per-symbol int8. */
int oracle_demodulate_soft(uint32_t modulation, size_t n, const float* symbols, const float* noise_vars, int8_t* llr)
{
  build_tables();
  switch (modulation) {
    case 1:
      for (size_t i = 0; i != n; ++i) {
        llr[i] = demod_bpsk(symbols[2 * i], symbols[2 * i + 1], noise_vars[i]);
      }
      return NRPHY_OK;
    case 0:
      for (size_t i = 0; i != n; ++i) {
        /* odd symbols are rotated by -90 degrees first: (im, -re) */
        llr[i] = (i & 1) ? demod_bpsk(symbols[2 * i + 1], -symbols[2 * i], noise_vars[i])
                         : demod_bpsk(symbols[2 * i], symbols[2 * i + 1], noise_vars[i]);
      }
      return NRPHY_OK;
    case 2:
      demod_qpsk(llr, symbols, noise_vars, n);
      return NRPHY_OK;
    case 4:
      demod_qam16(llr, symbols, noise_vars, n);
      return NRPHY_OK;
    case 6:
      demod_tables(llr, symbols, noise_vars, n, T64, 3, 16);
      return NRPHY_OK;
    case 8:
      demod_tables(llr, symbols, noise_vars, n, T256, 4, 4);
      return NRPHY_OK;
    default:
      return NRPHY_ERR_ARGUMENT;
  }
}
