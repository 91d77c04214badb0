/* TEST INFRASTRUCTURE -- CPU oracle, lower-PHY tail: amplitude controller, complex float ->
 * complex int16 conversion, Open Fronthaul IQ compression.  Part of oracle/liboracle.so; see nrphy_oracle.h for who may
 * load it.  R/ = srsRAN-5G-ER/.  Pinned against the compiled reference in tests/test_oracle.py
 * (test_oracle_vs_ref_amplitude_*, _ci16_*, _ofh_*).
 */
#include "nrphy_oracle.h"

#include <math.h>
#include <string.h>

/* One value through the reference's vector conversion (R/lib/srsvec/conversion.cpp:29-65, 202-230, built for AVX2):
 * _mm256_cvtps_epi32 rounds to nearest even, _mm256_packs_epi32 saturates; the scalar tail of a call rounds half away
 * from zero (std::round) and is only defined inside the int16 range. */
static int16_t to_int16(float v, int vector_lane)
{
  if (vector_lane) {
    long r = lrintf(v);
    return (int16_t)(r > 32767 ? 32767 : (r < -32768 ? -32768 : r));
  }
  return (int16_t)roundf(v);
}

/* amplitude_controller_{clipping,scaling}_impl::process (R/lib/phy/lower/amplitude_controller/
 * amplitude_controller_clipping_impl.cpp:31-68, amplitude_controller_scaling_impl.cpp:28-37). */
int oracle_amplitude_control(const nrphy_amplitude_cfg_t* c, const float* in, uint32_t nof_samples, float* out,
                             nrphy_amplitude_stats_t* stats)
{
  const float gain    = powf(10.0F, c->input_gain_dB / 20.0F);
  const float ceiling = c->full_scale_lin * powf(10.0F, c->ceiling_dBFS / 20.0F);
  double      sum     = 0;
  float       peak    = 0;
  uint32_t    clipped = 0;
  for (uint32_t i = 0; i != 2 * nof_samples; ++i) {
    out[i] = in[i] * gain;
  }
  if (c->kind == 0) {
    for (uint32_t i = 0; i != nof_samples; ++i) {
      const float p = out[2 * i] * out[2 * i] + out[2 * i + 1] * out[2 * i + 1];
      sum += p;
      peak = p > peak ? p : peak;
    }
    const float avg = (float)(sum / nof_samples);
    if (c->enable_clipping && isnormal(avg) && isnormal(peak)) {
      for (uint32_t i = 0; i != 2 * nof_samples; ++i) {
        if (out[i] > ceiling) {
          out[i] = ceiling;
          ++clipped;
        } else if (out[i] < -ceiling) {
          out[i] = -ceiling;
          ++clipped;
        }
      }
    }
  }
  if (stats) {
    stats->sum_power   = (float)sum;
    stats->peak_power  = peak;
    stats->nof_clipped = clipped;
    stats->nof_samples = nof_samples;
  }
  return NRPHY_OK;
}

int oracle_amplitude_metrics(const nrphy_amplitude_cfg_t* c, const nrphy_amplitude_stats_t* s, nrphy_amplitude_metrics_t* m)
{
  if (c->kind != 0) { /* the scaling implementation returns empty metrics */
    memset(m, 0, sizeof(*m));
    return NRPHY_OK;
  }
  const float full_scale_pwr = c->full_scale_lin * c->full_scale_lin;
  const float avg = s->sum_power / (float)s->nof_samples, peak = s->peak_power;
  m->clipping_enabled = c->enable_clipping ? 1 : 0;
  m->gain_dB          = 20.0F * log10f(powf(10.0F, c->input_gain_dB / 20.0F));
  m->avg_power_fs     = avg / full_scale_pwr;
  m->peak_power_fs    = peak / full_scale_pwr;
  if (!isnormal(avg) || !isnormal(peak)) {
    m->papr_lin = 1.0F;
    return NRPHY_OK;
  }
  m->papr_lin = peak / avg;
  if (c->enable_clipping) {
    m->nof_processed_samples += s->nof_samples;
    m->nof_clipped_samples += s->nof_clipped;
    m->clipping_probability = (double)m->nof_clipped_samples / (double)m->nof_processed_samples;
  }
  return NRPHY_OK;
}

/* srsvec::convert(span<const cf_t>, float, span<int16_t>) (conversion.cpp:29-65, 323-328). */
int oracle_iq_convert_ci16(const float* in, uint32_t nof_samples, float scale, int16_t* out)
{
  const uint32_t n = 2 * nof_samples, n_vec = (n / 16) * 16;
  for (uint32_t i = 0; i != n; ++i) {
    out[i] = to_int16(in[i] * scale, i < n_vec);
  }
  return NRPHY_OK;
}

static float bf16_value(uint16_t raw)
{
  uint32_t u = (uint32_t)raw << 16;
  float    f;
  memcpy(&f, &u, 4);
  return f;
}

/* MSB-first bit packer of compressed_prb_packer::pack (R/lib/ofh/compression/compressed_prb_packer.cpp:28-62). */
static void pack_samples(const int16_t* s, unsigned width, uint8_t* out)
{
  memset(out, 0, 3 * width);
  unsigned pos = 0;
  for (unsigned i = 0; i != 24; ++i) {
    for (int b = (int)width - 1; b >= 0; --b, ++pos) {
      out[pos >> 3] |= (uint8_t)((((uint16_t)s[i] >> b) & 1U) << (7 - (pos & 7)));
    }
  }
}

uint32_t oracle_ofh_compressed_prb_bytes(const nrphy_ofh_compression_cfg_t* c)
{
  return 3 * c->data_width + (c->type == 1 ? 1U : 0U);
}

/* iq_compressor::compress of one call's PRBs + the user-plane serialisation (iq_compression_none_impl.cpp:31-55,
 * iq_compression_bfp_impl.cpp:31-98, quantizer.h:36-95, ofh_uplane_message_builder_impl.cpp:137-144).  Returns bytes. */
int oracle_ofh_compress(const nrphy_ofh_compression_cfg_t* c, const uint16_t* prbs, uint32_t nof_prb, uint8_t* out)
{
  /* Below 8 bits the reference's packer hands bit_buffer::insert a value wider than the field (negative samples smear
   * into the neighbouring bits, compressed_prb_packer.cpp:39-47): no defined result to reproduce. */
  if (c->type > 1 || c->data_width < 8 || c->data_width > 16) {
    return -1;
  }
  const unsigned w = c->data_width, rec = oracle_ofh_compressed_prb_bytes(c);
  /* The AVX2 compressors (what create_iq_compressor returns on an AVX2 host, compression_factory.cpp:47-92) convert
   * all PRBs of a call in one go for BFP and for the widths their packer supports (9, 16); other widths fall back to
   * the generic compressor, which converts PRB by PRB. */
  const int      whole_span = c->type == 1 || w == 9 || w == 16;
  const uint32_t n_all = 24 * nof_prb, n_all_vec = (n_all / 16) * 16;
  for (uint32_t p = 0; p != nof_prb; ++p) {
    int16_t q[24];
    if (c->type == 0) {
      /* per-PRB conversion: 24 values, the first 16 in the vector loop */
      const float scale = (float)((1 << (w - 1)) - 1.0F) * c->iq_scaling;
      for (unsigned i = 0; i != 24; ++i) {
        q[i] = to_int16(bf16_value(prbs[24 * p + i]) * scale, whole_span ? 24 * p + i < n_all_vec : i < 16);
      }
      pack_samples(q, w, out + (size_t)p * rec);
      continue;
    }
    /* BFP: all PRBs of the call quantised to 16 bits in one conversion call, then per PRB the exponent that makes the
     * largest magnitude fit data_width bits and an arithmetic shift (O-RAN.WG4.CUS Annex A.1.2) */
    const float    scale = 32767.0F * c->iq_scaling;
    int            max_v = -32768, min_v = 32767;
    for (unsigned i = 0; i != 24; ++i) {
      q[i]  = to_int16(bf16_value(prbs[24 * p + i]) * scale, 24 * p + i < n_all_vec);
      max_v = q[i] > max_v ? q[i] : max_v;
      min_v = q[i] < min_v ? q[i] : min_v;
    }
    int a = max_v < 0 ? -max_v : max_v, b = (min_v < 0 ? -min_v : min_v) - 1;
    const unsigned max_abs   = (unsigned)(a > b ? a : b);
    const unsigned max_shift = 16 - w;
    unsigned       lz        = max_shift;
    if ((uint16_t)max_abs > 0 && max_shift > 0) {
      lz = (unsigned)__builtin_clz((unsigned)(uint16_t)max_abs) - 16U - 1U;
    }
    const int raw_exp  = (int)(max_shift < lz ? max_shift : lz);
    int       exponent = (int)max_shift - raw_exp;
    exponent           = exponent < 0 ? 0 : exponent;
    for (unsigned i = 0; i != 24; ++i) {
      q[i] = (int16_t)(q[i] >> exponent);
    }
    out[(size_t)p * rec] = (uint8_t)exponent;
    pack_samples(q, w, out + (size_t)p * rec + 1);
  }
  return (int)(rec * nof_prb);
}
