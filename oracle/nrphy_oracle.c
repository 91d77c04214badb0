/* TEST INFRASTRUCTURE -- CPU oracle: a plain-C restatement of the reference's PDSCH processor + OFDM
 * modulator algorithm (ushasigh/srsran-edgeric-5g, srsRAN-5G-ER/lib/phy; cited below as R/...).
 *
 * It is the checker for the HIP path, never the thing shipped or measured (except as bench.py's
 * "cpu_baseline" of kind "port").  Parity of this file is PINNED against the compiled reference
 * (oracle/_ref) in tests/test_oracle_vs_ref.py and against tests/golden/ vectors generated from it.
 *
 * Written from the 3GPP procedures the reference implements (TS 38.211 / 38.212 / 38.214) with the
 * reference's own choices where the standard leaves freedom (bf16 grid, un-normalised iDFT, LBRM handling,
 * FMA order of the AVX2 precoder).  One bit per byte internally: clarity over speed.
 */
#include "nrphy_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------------ */
/* bit helpers (MSB-first packing, R/include/srsran/adt/bit_buffer.h:98-190)                        */
/* ------------------------------------------------------------------------------------------------ */
static inline unsigned get_bit(const uint8_t* p, size_t i)
{
  return (p[i >> 3] >> (7 - (i & 7))) & 1U;
}
static inline __attribute__((unused)) void put_bit(uint8_t* p, size_t i, unsigned b)
{
  uint8_t m = (uint8_t)(0x80U >> (i & 7));
  if (b) {
    p[i >> 3] |= m;
  } else {
    p[i >> 3] &= (uint8_t)~m;
  }
}
static void pack_bits(uint8_t* out, const uint8_t* bits, size_t n)
{
  memset(out, 0, (n + 7) / 8);
  for (size_t i = 0; i != n; ++i) {
    if (bits[i]) {
      out[i >> 3] |= (uint8_t)(0x80U >> (i & 7));
    }
  }
}
static inline unsigned mask_test(const uint64_t* w, unsigned i)
{
  return (unsigned)((w[i >> 6] >> (i & 63)) & 1U);
}
static unsigned divide_ceil(unsigned a, unsigned b)
{
  return (a + b - 1) / b;
}

/* ------------------------------------------------------------------------------------------------ */
/* CRC (TS 38.212 Section 5.1; R/lib/phy/upper/channel_coding/crc_calculator_lut_impl.cpp:33-152)    */
/* MSB-first, zero initial value, no final XOR.                                                     */
/* ------------------------------------------------------------------------------------------------ */
static void crc_params(uint32_t poly_id, uint32_t* poly, unsigned* order)
{
  switch (poly_id) {
    case 16:
      *poly  = 0x11021;
      *order = 16;
      break;
    case 0x24B:
      *poly  = 0x1800063;
      *order = 24;
      break;
    default:
      *poly  = 0x1864CFB;
      *order = 24;
      break;
  }
}

uint32_t oracle_crc_bits(uint32_t poly_id, const uint8_t* data, uint32_t nbits)
{
  uint32_t poly;
  unsigned order;
  crc_params(poly_id, &poly, &order);
  uint32_t top = 1U << order, reg = 0;
  for (uint32_t i = 0; i != nbits; ++i) {
    reg = (reg << 1) | get_bit(data, i);
    if (reg & top) {
      reg ^= poly;
    }
  }
  for (unsigned i = 0; i != order; ++i) { /* append `order` zeros */
    reg <<= 1;
    if (reg & top) {
      reg ^= poly;
    }
  }
  return reg & (top - 1);
}

/* Byte-table form, used for whole transport blocks (same result as the bit-serial division). */
uint32_t oracle_crc(uint32_t poly_id, const uint8_t* data, uint32_t nbytes)
{
  uint32_t poly;
  unsigned order;
  crc_params(poly_id, &poly, &order);
  uint32_t table[256];
  uint32_t top = 1U << order, mask = top - 1;
  for (unsigned b = 0; b != 256; ++b) {
    uint32_t r = (uint32_t)b << (order - 8);
    for (unsigned k = 0; k != 8; ++k) {
      r <<= 1;
      if (r & top) {
        r ^= poly;
      }
    }
    table[b] = r & mask;
  }
  uint32_t reg = 0;
  for (uint32_t i = 0; i != nbytes; ++i) {
    unsigned idx = ((reg >> (order - 8)) ^ data[i]) & 0xFFU;
    reg          = ((reg << 8) & mask) ^ table[idx];
  }
  return reg;
}

/* ------------------------------------------------------------------------------------------------ */
/* TBS (TS 38.214 Section 5.1.3.2; R/lib/ran/sch/tbs_calculator.cpp:31-144)                          */
/* ------------------------------------------------------------------------------------------------ */
static const uint16_t TBS_TABLE[93] = {
    24,   32,   40,   48,   56,   64,   72,   80,   88,   96,   104,  112,  120,  128,  136,  144,  152,  160,  168,
    176,  184,  192,  208,  224,  240,  256,  272,  288,  304,  320,  336,  352,  368,  384,  408,  432,  456,  480,
    504,  528,  552,  576,  608,  640,  672,  704,  736,  768,  808,  848,  888,  928,  984,  1032, 1064, 1128, 1160,
    1192, 1224, 1256, 1288, 1320, 1352, 1416, 1480, 1544, 1608, 1672, 1736, 1800, 1864, 1928, 2024, 2088, 2152, 2216,
    2280, 2408, 2472, 2536, 2600, 2664, 2728, 2792, 2856, 2976, 3104, 3240, 3368, 3496, 3624, 3752, 3824};

uint32_t oracle_tbs_calculate(uint32_t nof_symb_sh, uint32_t nof_dmrs_prb, uint32_t nof_oh_prb, uint32_t qm,
                              float target_code_rate, uint32_t nof_layers, uint32_t n_prb)
{
  unsigned nof_re_prime = 12 * nof_symb_sh - nof_dmrs_prb - nof_oh_prb;
  unsigned nof_re       = (nof_re_prime < 156 ? nof_re_prime : 156) * n_prb;
  float    tcr          = target_code_rate * (1.F / 1024);
  float    nof_info     = 1.0F * (float)nof_re * tcr * (float)qm * (float)nof_layers;
  if (nof_info <= 3824) {
    unsigned n = 3;
    if (nof_info > 512) {
      n = (unsigned)floorf(log2f(nof_info)) - 6U;
    }
    unsigned p2    = 1U << n;
    unsigned prime = p2 * (unsigned)floorf(nof_info / (float)p2);
    if (prime < 24) {
      prime = 24;
    }
    for (unsigned i = 0; i != 93; ++i) {
      if (TBS_TABLE[i] >= prime) {
        return TBS_TABLE[i];
      }
    }
    return 3824;
  }
  unsigned n     = (unsigned)(floorf(log2f(nof_info - 24)) - 5.0F);
  unsigned p2    = 1U << n;
  unsigned prime = p2 * (unsigned)roundf((nof_info - 24) / (float)p2);
  if (prime < 3840) {
    prime = 3840;
  }
  unsigned C = 1;
  if (tcr <= 0.25F) {
    C = divide_ceil(prime + 24, 3816);
  } else if (prime > 8424) {
    C = divide_ceil(prime + 24, 8424);
  }
  return 8 * C * divide_ceil(prime + 24, 8 * C) - 24;
}

/* ------------------------------------------------------------------------------------------------ */
/* RE masks (R/lib/phy/support/re_pattern.cpp:27-198, R/include/srsran/phy/upper/dmrs_mapping.h:69-123) */
/* ------------------------------------------------------------------------------------------------ */
/* mask[k] = 1 when subcarrier k of OFDM symbol l carries PDSCH data for this PDU. */
static void data_re_mask(const nrphy_pdsch_pdu_t* pdu, unsigned l, uint8_t* mask, unsigned nof_subc)
{
  memset(mask, 0, nof_subc);
  if (l < pdu->start_symbol_index || l >= pdu->start_symbol_index + pdu->nof_symbols) {
    return;
  }
  for (unsigned p = 0; p * 12 < nof_subc && p < NRPHY_MAX_RB; ++p) {
    if (mask_test(pdu->prb_mask, p)) {
      memset(mask + 12 * p, 1, 12);
    }
  }
  /* Reserved patterns. */
  for (unsigned r = 0; r != pdu->nof_reserved; ++r) {
    const nrphy_re_pattern_t* pat = &pdu->reserved[r];
    if (!((pat->symbol_mask >> l) & 1U)) {
      continue;
    }
    for (unsigned p = 0; p * 12 < nof_subc && p < NRPHY_MAX_RB; ++p) {
      if (mask_test(pat->prb_mask, p)) {
        for (unsigned k = 0; k != 12; ++k) {
          if ((pat->re_mask >> k) & 1U) {
            mask[12 * p + k] = 0;
          }
        }
      }
    }
  }
  /* DM-RS pattern: whole BWP, CDM groups without data (type 1: group g = subcarriers g, g+2, ...). */
  if ((pdu->dmrs_symbol_mask >> l) & 1U) {
    for (unsigned p = pdu->bwp_start_rb; p < pdu->bwp_start_rb + pdu->bwp_size_rb && p * 12 < nof_subc; ++p) {
      for (unsigned k = 0; k != 12; ++k) {
        unsigned group = (pdu->dmrs_type == 1) ? (k % 2) : ((k % 6) / 2);
        if (group < pdu->nof_cdm_groups_without_data) {
          mask[12 * p + k] = 0;
        }
      }
    }
  }
}

static __attribute__((unused)) unsigned prb_count(const uint64_t* w)
{
  unsigned c = 0;
  for (unsigned i = 0; i != NRPHY_PRB_WORDS; ++i) {
    c += (unsigned)__builtin_popcountll(w[i]);
  }
  return c;
}
static int prb_lowest(const uint64_t* w)
{
  for (unsigned i = 0; i != NRPHY_MAX_RB; ++i) {
    if (mask_test(w, i)) {
      return (int)i;
    }
  }
  return -1;
}
static int prb_highest(const uint64_t* w)
{
  int hi = -1;
  for (unsigned i = 0; i != 64 * NRPHY_PRB_WORDS; ++i) {
    if (mask_test(w, i)) {
      hi = (int)i;
    }
  }
  return hi;
}

/* ------------------------------------------------------------------------------------------------ */
/* Validator (R/lib/phy/upper/channel_processors/pdsch_processor_validator_impl.cpp:99-181)           */
/* ------------------------------------------------------------------------------------------------ */
int oracle_pdsch_validate(const nrphy_pdsch_pdu_t* pdu)
{
  unsigned nsymb = pdu->cp ? 12 : 14;
  int      hi    = prb_highest(pdu->prb_mask);
  int      lo    = prb_lowest(pdu->prb_mask);
  /* freq_alloc.is_bwp_valid: the allocation lies inside the BWP. */
  if (lo < 0 || (unsigned)lo < pdu->bwp_start_rb || (unsigned)hi >= pdu->bwp_start_rb + pdu->bwp_size_rb ||
      pdu->bwp_start_rb + pdu->bwp_size_rb > NRPHY_MAX_RB) {
    return NRPHY_ERR_INVALID_PDU;
  }
  if (pdu->dmrs_symbol_mask == 0 || (pdu->dmrs_symbol_mask >> nsymb) != 0) {
    return NRPHY_ERR_INVALID_PDU;
  }
  unsigned first_dmrs = (unsigned)__builtin_ctz(pdu->dmrs_symbol_mask);
  unsigned last_dmrs  = 31U - (unsigned)__builtin_clz(pdu->dmrs_symbol_mask);
  if (first_dmrs < pdu->start_symbol_index || last_dmrs >= pdu->start_symbol_index + pdu->nof_symbols ||
      nsymb < pdu->start_symbol_index + pdu->nof_symbols) {
    return NRPHY_ERR_INVALID_PDU;
  }
  if (pdu->dmrs_type != 1 || pdu->nof_cdm_groups_without_data > 2) {
    return NRPHY_ERR_INVALID_PDU;
  }
  if (!pdu->vrb_contiguous) {
    return NRPHY_ERR_INVALID_PDU;
  }
  for (int prb = lo; prb <= hi; ++prb) { /* freq_alloc.is_contiguous(): the mask itself must have no hole */
    if (!mask_test(pdu->prb_mask, (unsigned)prb)) {
      return NRPHY_ERR_INVALID_PDU;
    }
  }
  if (pdu->nof_ports == 0 || pdu->nof_ports > 4 || pdu->nof_layers == 0 || pdu->nof_layers > pdu->nof_ports) {
    return NRPHY_ERR_INVALID_PDU;
  }
  if (pdu->nof_codewords != 1 || pdu->tbs_lbrm_bytes == 0 || pdu->nof_reserved > NRPHY_MAX_RESERVED) {
    return NRPHY_ERR_INVALID_PDU;
  }
  if (pdu->qm != 2 && pdu->qm != 4 && pdu->qm != 6 && pdu->qm != 8) {
    return NRPHY_ERR_INVALID_PDU;
  }
  if (pdu->rv > 3 || (pdu->ldpc_base_graph != 1 && pdu->ldpc_base_graph != 2) || pdu->tb_size_bytes == 0 ||
      pdu->tb_size_bytes > NRPHY_MAX_TB_BYTES ||
      pdu->nof_prg == 0 || pdu->nof_prg > NRPHY_MAX_PRG || pdu->prg_size_rb == 0 || pdu->prg_size_rb > NRPHY_MAX_RB || pdu->precoding == NULL || pdu->cp > 1) {
    return NRPHY_ERR_INVALID_PDU;
  }
  /* DM-RS and reserved RE must not collide (check_dmrs_and_reserved_collision, :28-40): no reserved pattern may
   * touch an OFDM symbol that carries DM-RS. */
  for (unsigned r = 0; r != pdu->nof_reserved; ++r) {
    if ((pdu->reserved[r].symbol_mask & pdu->dmrs_symbol_mask) != 0) {
      return NRPHY_ERR_INVALID_PDU;
    }
  }
  return NRPHY_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* Per-PDU scalars (R/lib/phy/upper/channel_processors/pdsch_processor_impl.cpp:75-136,               */
/* R/lib/phy/upper/channel_coding/ldpc/ldpc_segmenter_impl.cpp:58-160, R/include/.../ldpc/ldpc.h:128-228, */
/* R/lib/phy/upper/channel_coding/ldpc/ldpc_rate_matcher_impl.cpp:37-91)                              */
/* ------------------------------------------------------------------------------------------------ */
static const uint16_t LIFTING_SIZES[51] = {2,   3,   4,   5,   6,   7,   8,   9,   10,  11,  12,  13,  14,
                                           15,  16,  18,  20,  22,  24,  26,  28,  30,  32,  36,  40,  44,
                                           48,  52,  56,  60,  64,  72,  80,  88,  96,  104, 112, 120, 128,
                                           144, 160, 176, 192, 208, 224, 240, 256, 288, 320, 352, 384};

static unsigned nof_data_re(const nrphy_pdsch_pdu_t* pdu)
{
  uint8_t  mask[NRPHY_MAX_RB * 12];
  unsigned count = 0;
  for (unsigned l = 0; l != 14; ++l) {
    data_re_mask(pdu, l, mask, NRPHY_MAX_RB * 12);
    for (unsigned k = 0; k != NRPHY_MAX_RB * 12; ++k) {
      count += mask[k];
    }
  }
  return count;
}

static void derive_coding(uint32_t bg, uint32_t tb_bits, uint32_t rv, uint32_t qm, uint32_t nref_cfg,
                          uint32_t nof_layers, uint32_t nof_re, nrphy_pdsch_derived_t* d)
{
  unsigned tb_crc = (tb_bits <= 3824) ? 16 : 24;
  unsigned b      = tb_bits + tb_crc;
  unsigned kcb    = (bg == 1) ? 8448 : 3840;
  unsigned C      = (b <= kcb) ? 1 : divide_ceil(b, kcb - 24);
  unsigned b_out  = b + ((C > 1) ? 24 * C : 0);
  unsigned ref_len = 22;
  if (bg == 2) {
    ref_len = (b > 640) ? 10 : (b > 560) ? 9 : (b > 192) ? 8 : 6;
  }
  unsigned zc = 0;
  for (unsigned i = 0; i != 51; ++i) {
    if (LIFTING_SIZES[i] * C * ref_len >= b_out) {
      zc = LIFTING_SIZES[i];
      break;
    }
  }
  unsigned K       = ((bg == 1) ? 22 : 10) * zc;
  unsigned cb_crc  = (C > 1) ? 24 : 0;
  unsigned info    = divide_ceil(b_out, C) - cb_crc;
  unsigned N       = ((bg == 1) ? 66 : 50) * zc;
  d->nof_re             = nof_re;
  d->nof_codeblocks     = C;
  d->lifting_size       = zc;
  d->segment_length     = K;
  d->cb_info_bits       = info;
  d->nof_filler_bits    = K - info - cb_crc;
  d->nof_tb_crc_bits    = tb_crc;
  d->nof_cb_crc_bits    = cb_crc;
  d->zero_pad           = (info + cb_crc) * C - b_out;
  d->full_length        = N;
  d->n_ref              = nref_cfg;
  d->n_cb               = (nref_cfg > 0 && nref_cfg < N) ? nref_cfg : N;
  static const double shift_bg1[4] = {0, 17, 33, 56};
  static const double shift_bg2[4] = {0, 13, 25, 43};
  double tmp = (((bg == 1) ? shift_bg1 : shift_bg2)[rv] * d->n_cb) / N;
  d->k0                 = (uint16_t)floor(tmp) * zc;
  d->nof_short_segments = C - (nof_re % C);
  d->rm_length_short    = (nof_re / C) * nof_layers * qm;
  d->rm_length_long     = divide_ceil(nof_re, C) * nof_layers * qm;
  d->codeword_bits      = nof_re * nof_layers * qm;
}

int oracle_pdsch_derive(const nrphy_pdsch_pdu_t* pdu, nrphy_pdsch_derived_t* d)
{
  unsigned nof_re  = nof_data_re(pdu);
  unsigned tb_bits = 8 * pdu->tb_size_bytes;
  unsigned tb_crc  = (tb_bits <= 3824) ? 16 : 24;
  unsigned kcb     = (pdu->ldpc_base_graph == 1) ? 8448 : 3840;
  unsigned C       = (tb_bits + tb_crc <= kcb) ? 1 : divide_ceil(tb_bits + tb_crc, kcb - 24);
  /* ldpc::compute_N_ref (ldpc.h:225-228). */
  uint64_t nref = ((uint64_t)pdu->tbs_lbrm_bytes * 8 * 3) / (2 * C);
  if (nref > 66 * 384) {
    nref = 66 * 384;
  }
  derive_coding(pdu->ldpc_base_graph, tb_bits, pdu->rv, pdu->qm, (uint32_t)nref, pdu->nof_layers, nof_re, d);
  return NRPHY_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* Segmentation (TS 38.212 Section 5.2.2; R/lib/phy/upper/channel_coding/ldpc/ldpc_segmenter_impl.cpp:90-235) */
/* Produces codeblock `i` as K unpacked bits; filler bits are zeros (as in the reference).          */
/* ------------------------------------------------------------------------------------------------ */
static void segment_cb(const nrphy_pdsch_derived_t* d, const uint8_t* tb, uint32_t tb_bytes, uint32_t tb_crc,
                       unsigned i, uint8_t* cb_bits /* K */)
{
  unsigned K    = d->segment_length;
  unsigned info = d->cb_info_bits;
  unsigned C    = d->nof_codeblocks;
  memset(cb_bits, 0, K);
  unsigned tb_offset = i * info;
  unsigned used      = info;
  if (i == C - 1) {
    used -= d->nof_tb_crc_bits + d->zero_pad;
  }
  (void)tb_bytes;
  for (unsigned k = 0; k != used; ++k) {
    cb_bits[k] = (uint8_t)get_bit(tb, tb_offset + k);
  }
  if (i == C - 1) {
    for (unsigned k = 0; k != d->nof_tb_crc_bits; ++k) {
      cb_bits[used + k] = (uint8_t)((tb_crc >> (d->nof_tb_crc_bits - 1 - k)) & 1U);
    }
    used += d->nof_tb_crc_bits + d->zero_pad; /* zero pad already zero */
  }
  if (d->nof_cb_crc_bits) {
    uint8_t packed[8448 / 8 + 8];
    pack_bits(packed, cb_bits, used);
    uint32_t crc = oracle_crc_bits(0x24B, packed, used);
    for (unsigned k = 0; k != 24; ++k) {
      cb_bits[used + k] = (uint8_t)((crc >> (23 - k)) & 1U);
    }
  }
}

static unsigned cb_rm_length(const nrphy_pdsch_derived_t* d, unsigned i)
{
  return (i < d->nof_short_segments) ? d->rm_length_short : d->rm_length_long;
}

int oracle_ldpc_segment(uint32_t bg, uint32_t rv, uint32_t qm, uint32_t nref, uint32_t nof_layers,
                        uint32_t nof_ch_symbols, const uint8_t* tb, uint32_t tb_bytes, uint8_t* segments,
                        uint32_t stride_bytes, uint32_t* meta, uint32_t* lifting_size)
{
  nrphy_pdsch_derived_t d;
  derive_coding(bg, 8 * tb_bytes, rv, qm, nref, nof_layers, nof_ch_symbols / nof_layers, &d);
  uint32_t tb_crc    = oracle_crc(d.nof_tb_crc_bits == 16 ? 16 : 0x24A, tb, tb_bytes);
  uint8_t* bits      = (uint8_t*)malloc(d.segment_length);
  unsigned cw_offset = 0;
  for (unsigned i = 0; i != d.nof_codeblocks; ++i) {
    segment_cb(&d, tb, tb_bytes, tb_crc, i, bits);
    pack_bits(segments + (size_t)i * stride_bytes, bits, d.segment_length);
    meta[5 * i + 0] = cb_rm_length(&d, i);
    meta[5 * i + 1] = cw_offset;
    meta[5 * i + 2] = d.nof_filler_bits;
    meta[5 * i + 3] = d.full_length;
    meta[5 * i + 4] = (d.nof_codeblocks == 1) ? d.nof_tb_crc_bits : 24;
    cw_offset += meta[5 * i];
  }
  *lifting_size = d.lifting_size;
  free(bits);
  return (int)d.nof_codeblocks;
}

/* ------------------------------------------------------------------------------------------------ */
/* LDPC encoding (TS 38.212 Section 5.3.2; R/lib/phy/upper/channel_coding/ldpc/ldpc_encoder_impl.cpp:44-81, */
/* ldpc_encoder_generic.cpp:58-230).  Base-graph data: 3GPP Tables 5.3.2-2/-3 (generated, see           */
/* oracle/gen_bg_tables.py).                                                                          */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
  uint8_t  row;
  uint8_t  col;
  uint16_t shift[8];
} nr_ldpc_edge_t;
#include "srsran-edgeric-5g_amd/csrc/nr_ldpc_bg.inc"

/* Lifting-set index i_LS: Zc = a * 2^j with a in {2,3,5,7,9,11,13,15} (TS 38.212 Table 5.3.2-1). */
static int lifting_set_index(unsigned zc)
{
  static const unsigned base[8] = {2, 3, 5, 7, 9, 11, 13, 15};
  for (int i = 0; i != 8; ++i) {
    for (unsigned v = base[i]; v <= 384; v *= 2) {
      if (v == zc) {
        return i;
      }
    }
  }
  return -1;
}

/* codeblock: (Kb + nof_parity_rows) * Zc unpacked bits, systematic part filled on entry. */
static void ldpc_encode_bits(unsigned bg, unsigned zc, uint8_t* cb, unsigned nof_rows)
{
  const nr_ldpc_edge_t* edges   = (bg == 1) ? NR_LDPC_BG1_EDGES : NR_LDPC_BG2_EDGES;
  unsigned              n_edges = (bg == 1) ? NR_LDPC_BG1_NOF_EDGES : NR_LDPC_BG2_NOF_EDGES;
  unsigned              kb      = (bg == 1) ? 22 : 10;
  int                   ils     = lifting_set_index(zc);
  uint8_t               aux[4][384];
  memset(aux, 0, sizeof(aux));
  memset(cb + kb * zc, 0, (size_t)nof_rows * zc);

  /* Systematic contributions: rows 0..3 into aux, extension rows straight into their parity block. */
  int      core_shift[4] = {-1, -1, -1, -1}; /* shift of column Kb in the four core rows */
  for (unsigned e = 0; e != n_edges; ++e) {
    unsigned m = edges[e].row, c = edges[e].col, s = edges[e].shift[ils] % zc;
    if (c == kb && m < 4) {
      core_shift[m] = (int)s;
    }
    if (c >= kb || m >= nof_rows) {
      continue;
    }
    uint8_t*       dst = (m < 4) ? aux[m] : cb + (kb + m) * zc;
    const uint8_t* src = cb + c * zc;
    for (unsigned k = 0; k != zc; ++k) {
      dst[k] ^= src[(k + s) % zc];
    }
  }
  /* Core parity.  Column Kb has three edges in rows {0, r, 3}, two with equal shifts; adding the four core
   * rows cancels the dual diagonal and leaves P^b p0 = sum(aux) with b the odd shift out. */
  unsigned b = 0, mid = 1;
  {
    int s0 = core_shift[0], s3 = core_shift[3];
    mid    = (core_shift[1] >= 0) ? 1 : 2;
    int sm = core_shift[mid];
    b      = (s0 == s3) ? (unsigned)sm : ((s0 == sm) ? (unsigned)s3 : (unsigned)s0);
  }
  uint8_t* p0 = cb + kb * zc;
  uint8_t* p1 = p0 + zc;
  uint8_t* p2 = p1 + zc;
  uint8_t* p3 = p2 + zc;
  for (unsigned k = 0; k != zc; ++k) {
    unsigned i = (k + zc - b) % zc;
    p0[k]      = aux[0][i] ^ aux[1][i] ^ aux[2][i] ^ aux[3][i];
  }
  for (unsigned k = 0; k != zc; ++k) {
    p1[k] = aux[0][k] ^ p0[(k + (unsigned)core_shift[0]) % zc]; /* row 0: aux0 + P^s0 p0 + p1 = 0 */
    p3[k] = aux[3][k] ^ p0[(k + (unsigned)core_shift[3]) % zc]; /* row 3: aux3 + P^s3 p0 + p3 = 0 */
  }
  for (unsigned k = 0; k != zc; ++k) {
    /* The core row without an edge in column Kb closes the chain: row 2 (BG1): aux2 + p2 + p3 = 0;
     * row 1 (BG2): aux1 + p1 + p2 = 0. */
    p2[k] = (mid == 1) ? (uint8_t)(aux[2][k] ^ p3[k]) : (uint8_t)(aux[1][k] ^ p1[k]);
  }
  /* Extension rows: add the contributions of the four core parity blocks. */
  for (unsigned e = 0; e != n_edges; ++e) {
    unsigned m = edges[e].row, c = edges[e].col, s = edges[e].shift[ils] % zc;
    if (m < 4 || m >= nof_rows || c < kb || c >= kb + 4) {
      continue;
    }
    uint8_t*       dst = cb + (kb + m) * zc;
    const uint8_t* src = cb + c * zc;
    for (unsigned k = 0; k != zc; ++k) {
      dst[k] ^= src[(k + s) % zc];
    }
  }
}

int oracle_ldpc_encode(uint32_t bg, uint32_t zc, const uint8_t* msg, uint32_t out_bits, uint8_t* out)
{
  unsigned kb = (bg == 1) ? 22 : 10, nfull = (bg == 1) ? 68 : 52;
  if (lifting_set_index(zc) < 0 || out_bits > (nfull - 2) * zc) {
    return NRPHY_ERR_ARGUMENT;
  }
  /* Codeblock length the encoder works with (ldpc_encoder_impl.cpp:63-72). */
  unsigned len = out_bits + 2 * zc;
  if (len < (kb + 4) * zc) {
    len = (kb + 4) * zc;
  }
  len              = divide_ceil(len, zc) * zc;
  unsigned nof_rows = len / zc - kb;
  uint8_t* cb       = (uint8_t*)calloc(nfull * 384, 1);
  for (unsigned k = 0; k != kb * zc; ++k) {
    cb[k] = (uint8_t)get_bit(msg, k);
  }
  ldpc_encode_bits(bg, zc, cb, nof_rows);
  pack_bits(out, cb + 2 * zc, out_bits);
  free(cb);
  return NRPHY_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* Rate matching + bit interleaving (TS 38.212 Section 5.4.2; R/.../ldpc_rate_matcher_impl.cpp:37-294) */
/* in: codeblock without its first 2Zc bits (N = 66Zc / 50Zc), unpacked.  out: E unpacked bits.       */
/* ------------------------------------------------------------------------------------------------ */
static void rate_match_bits(const uint8_t* in, unsigned n_cb, unsigned k0, unsigned filler_start,
                            unsigned filler_stop, unsigned qm, unsigned E, uint8_t* out)
{
  uint8_t* sel = (uint8_t*)malloc(E);
  unsigned idx = k0;
  for (unsigned t = 0; t != E; ++t) {
    if (idx >= filler_start && idx < filler_stop) {
      /* Past the filler bits; a circular buffer that ends inside them (filler_start < Ncb < filler_stop, only possible with a
       * limited-buffer size below the transport block's own) wraps to its first bit.  The reference is undefined there
       * (select_bits jumps to filler_stop, beyond the buffer: ldpc_rate_matcher_impl.cpp:115-137, a crash in the compiled
       * reference); this is TS 38.212 Section 5.4.2.1 evaluated literally, which is also what the device computes. */
      idx = filler_stop < n_cb ? filler_stop : 0;
    }
    sel[t] = in[idx];
    idx    = (idx + 1) % n_cb;
  }
  unsigned rows = E / qm;
  for (unsigned i = 0; i != rows; ++i) {
    for (unsigned j = 0; j != qm; ++j) {
      out[i * qm + j] = sel[j * rows + i];
    }
  }
  free(sel);
}

int oracle_ldpc_rate_match(uint32_t bg, uint32_t zc, uint32_t rv, uint32_t qm, uint32_t nref, uint32_t nof_filler,
                           const uint8_t* in, uint32_t in_bits, uint8_t* out, uint32_t rm_length)
{
  unsigned N = ((bg == 1) ? 66 : 50) * zc;
  if (in_bits != N || rm_length % qm != 0) {
    return NRPHY_ERR_ARGUMENT;
  }
  unsigned n_cb = (nref > 0 && nref < N) ? nref : N;
  static const double shift_bg1[4] = {0, 17, 33, 56};
  static const double shift_bg2[4] = {0, 13, 25, 43};
  double   tmp  = (((bg == 1) ? shift_bg1 : shift_bg2)[rv] * n_cb) / N;
  unsigned k0   = (uint16_t)floor(tmp) * zc;
  unsigned nsys = ((bg == 1) ? 20 : 8) * zc;
  uint8_t* bits = (uint8_t*)malloc(N);
  for (unsigned k = 0; k != N; ++k) {
    bits[k] = (uint8_t)get_bit(in, k);
  }
  uint8_t* o = (uint8_t*)malloc(rm_length);
  rate_match_bits(bits, n_cb, k0, nsys - nof_filler, nsys, qm, rm_length, o);
  pack_bits(out, o, rm_length);
  free(o);
  free(bits);
  return NRPHY_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* Gold sequence (TS 38.211 Section 5.2.1; R/.../pseudo_random_generator_impl.cpp:58-316)             */
/* c(n) = x1(n+Nc) ^ x2(n+Nc), Nc = 1600, x1(n+31) = x1(n+3)^x1(n), x2(n+31) = x2(n+3)^x2(n+2)^x2(n+1)^x2(n). */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
  uint32_t x1, x2; /* bit k = x(n + k), k = 0..30 */
} gold_t;

static inline unsigned gold_step(gold_t* g)
{
  unsigned c  = (g->x1 ^ g->x2) & 1U;
  uint32_t f1 = ((g->x1 >> 3) ^ g->x1) & 1U;
  uint32_t f2 = ((g->x2 >> 3) ^ (g->x2 >> 2) ^ (g->x2 >> 1) ^ g->x2) & 1U;
  g->x1       = (g->x1 >> 1) | (f1 << 30);
  g->x2       = (g->x2 >> 1) | (f2 << 30);
  return c;
}
static void gold_init(gold_t* g, uint32_t c_init, uint32_t offset)
{
  g->x1 = 1;
  g->x2 = c_init & 0x7FFFFFFFU;
  for (uint32_t i = 0; i != 1600 + offset; ++i) {
    gold_step(g);
  }
}

void oracle_prg_apply_xor(uint32_t c_init, uint32_t offset, uint8_t* data, uint32_t nbits)
{
  gold_t g;
  gold_init(&g, c_init, offset);
  for (uint32_t i = 0; i != nbits; ++i) {
    if (gold_step(&g)) {
      data[i >> 3] ^= (uint8_t)(0x80U >> (i & 7));
    }
  }
}

/* pseudo_random_generator::apply_xor on soft bits (pseudo_random_generator_impl.cpp:423-523): the log-likelihood ratio of
 * every position whose sequence bit is one changes sign.  The 16-at-a-time path negates in 8-bit two's complement
 * (-128 stays -128) and the scalar tail multiplies by -1 and narrows again, which is the same value.
 * pusch_demodulator_impl performs the same operation on the generated sequence (revert_scrambling,
 * lib/phy/upper/channel_processors/pusch/pusch_demodulator_impl.cpp:38-100, 254-259). */
void oracle_prg_apply_xor_llr(uint32_t c_init, uint32_t offset, const int8_t* in, int8_t* out, uint32_t n)
{
  gold_t g;
  gold_init(&g, c_init, offset);
  for (uint32_t i = 0; i != n; ++i) {
    out[i] = gold_step(&g) ? (int8_t)(uint8_t)(0U - (uint8_t)in[i]) : in[i];
  }
}

void oracle_prg_generate_float(uint32_t c_init, uint32_t offset, float value, float* out, uint32_t n)
{
  gold_t g;
  gold_init(&g, c_init, offset);
  for (uint32_t i = 0; i != n; ++i) {
    out[i] = gold_step(&g) ? -value : value;
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* Modulation mapper (TS 38.211 Section 5.1; R/.../modulation_mapper_lut_impl.cpp:39-65,268-302)       */
/* ci8 output: un-normalised odd integers; returns the normalisation the caller folds into the weights. */
/* ------------------------------------------------------------------------------------------------ */
static void map_symbol(unsigned qm, unsigned index /* Qm bits, first bit = MSB */, int* re, int* im)
{
  /* b(i) of TS 38.211: bit i of the group, i = 0 first. */
  int b[8];
  for (unsigned i = 0; i != qm; ++i) {
    b[i] = (int)((index >> (qm - 1 - i)) & 1U);
  }
  int r = 1, q = 1; /* innermost term */
  /* d = (1-2b0)[2^(h-1) - (1-2b2)[2^(h-2) - ... ]] with h = Qm/2 levels; build from the inside out. */
  unsigned h = qm / 2;
  r          = 1 - 2 * b[2 * (h - 1)];
  q          = 1 - 2 * b[2 * (h - 1) + 1];
  for (unsigned lvl = 1; lvl != h; ++lvl) {
    unsigned i = h - 1 - lvl;
    r          = (1 - 2 * b[2 * i]) * ((1 << lvl) - r);
    q          = (1 - 2 * b[2 * i + 1]) * ((1 << lvl) - q);
  }
  *re = r;
  *im = q;
}

static float modulation_scaling(unsigned qm)
{
  /* sqrt(1 / mean power) evaluated in single precision like the reference's table constructor. */
  float avg = (qm == 2) ? 2.0F : (qm == 4) ? 10.0F : (qm == 6) ? 42.0F : 170.0F;
  return sqrtf(1 / avg);
}

float oracle_modulate_ci8(uint32_t qm, const uint8_t* bits, uint32_t nsym, int8_t* out)
{
  for (uint32_t s = 0; s != nsym; ++s) {
    unsigned index = 0;
    for (unsigned j = 0; j != qm; ++j) {
      index = (index << 1) | get_bit(bits, (size_t)s * qm + j);
    }
    int re, im;
    map_symbol(qm, index, &re, &im);
    out[2 * s]     = (int8_t)re;
    out[2 * s + 1] = (int8_t)im;
  }
  return modulation_scaling(qm);
}

/* ------------------------------------------------------------------------------------------------ */
/* bf16 + precoding arithmetic (R/include/srsran/adt/bf16.h:39-56;                                    */
/* R/lib/phy/generic_functions/precoding/channel_precoder_avx2.cpp:51-73,214-342)                     */
/* ------------------------------------------------------------------------------------------------ */
static inline uint16_t to_bf16(float v)
{
  uint32_t u;
  memcpy(&u, &v, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static inline float from_bf16(uint16_t v)
{
  uint32_t u = (uint32_t)v << 16;
  float    f;
  memcpy(&f, &u, 4);
  return f;
}
/* x * w as the reference's SIMD kernel evaluates it: fmaddsub(x, w.re, swap(x) * w.im). */
static inline void cmul_fmaddsub(float xr, float xi, float wr, float wi, float* outr, float* outi)
{
  float t0 = xi * wi;
  float t1 = xr * wi;
  *outr    = fmaf(xr, wr, -t0);
  *outi    = fmaf(xi, wr, t1);
}

/* ------------------------------------------------------------------------------------------------ */
/* PDSCH processor                                                                                  */
/* ------------------------------------------------------------------------------------------------ */
static int encode_codeword(const nrphy_pdsch_pdu_t* pdu, const nrphy_pdsch_derived_t* d, const uint8_t* tb,
                           uint8_t* cw_bits /* G unpacked */)
{
  unsigned bg = pdu->ldpc_base_graph, zc = d->lifting_size, kb = (bg == 1) ? 22 : 10;
  unsigned nfull  = (bg == 1) ? 68 : 52;
  uint32_t tb_crc = oracle_crc(d->nof_tb_crc_bits == 16 ? 16 : 0x24A, tb, pdu->tb_size_bytes);
  uint8_t* cb     = (uint8_t*)malloc(nfull * 384);
  unsigned nsys   = (kb - 2) * zc;
  unsigned offset = 0;
  for (unsigned i = 0; i != d->nof_codeblocks; ++i) {
    segment_cb(d, tb, pdu->tb_size_bytes, tb_crc, i, cb);
    /* The reference always encodes the full codeblock (pdsch_encoder_impl.cpp:52-55). */
    ldpc_encode_bits(bg, zc, cb, nfull - kb);
    unsigned E = cb_rm_length(d, i);
    rate_match_bits(cb + 2 * zc, d->n_cb, d->k0, nsys - d->nof_filler_bits, nsys, pdu->qm, E, cw_bits + offset);
    offset += E;
  }
  free(cb);
  return (offset == d->codeword_bits) ? NRPHY_OK : NRPHY_ERR_ARGUMENT;
}

int oracle_pdsch_encode(const nrphy_pdsch_pdu_t* pdu, const uint8_t* tb, uint8_t* cw_rm)
{
  nrphy_pdsch_derived_t d;
  oracle_pdsch_derive(pdu, &d);
  uint8_t* bits = (uint8_t*)malloc(d.codeword_bits);
  int      rc   = encode_codeword(pdu, &d, tb, bits);
  pack_bits(cw_rm, bits, d.codeword_bits);
  free(bits);
  return rc;
}

/* DM-RS (TS 38.211 Section 7.4.1.1; R/lib/phy/upper/signal_processors/dmrs_pdsch_processor_impl.cpp:84-262, */
/* dmrs_helper.h:44-109, R/lib/phy/support/resource_grid_mapper_impl.cpp:47-133).                      */
static void put_dmrs(const nrphy_pdsch_pdu_t* pdu, uint16_t* grid, unsigned nof_subc)
{
  float amp_cfg   = powf(10.0F, -pdu->ratio_pdsch_dmrs_to_sss_dB / 20.0F);
  float amplitude = (float)(M_SQRT1_2 * (double)amp_cfg);
  unsigned ref_rb = (pdu->ref_point == 1) ? pdu->bwp_start_rb : 0;
  for (unsigned l = 0; l != 14; ++l) {
    if (!((pdu->dmrs_symbol_mask >> l) & 1U)) {
      continue;
    }
    uint32_t c_init = (uint32_t)((((uint64_t)(14 * pdu->slot_index + l + 1) * (2 * pdu->scrambling_id + 1) << 17) +
                                  (2 * pdu->scrambling_id + (pdu->n_scid ? 1 : 0))) &
                                 0x7FFFFFFFULL);
    for (unsigned prb = 0; prb != NRPHY_MAX_RB && 12 * prb < nof_subc; ++prb) {
      if (!mask_test(pdu->prb_mask, prb)) {
        continue;
      }
      /* Six QPSK pilots per PRB and CDM group, r(6*(prb - ref) + k'). */
      float seq[12];
      oracle_prg_generate_float(c_init, (prb - ref_rb) * 12, amplitude, seq, 12);
      const float* w = pdu->precoding; /* DM-RS uses the wideband (first PRG) weights, unscaled */
      unsigned     prg = (pdu->nof_prg > 1) ? (prb / pdu->prg_size_rb) : 0;
      if (prg >= pdu->nof_prg) {
        prg = pdu->nof_prg - 1;
      }
      w += 2 * (size_t)prg * pdu->nof_ports * pdu->nof_layers;
      for (unsigned kp = 0; kp != 6; ++kp) {
        for (unsigned group = 0; group != (pdu->nof_layers + 1) / 2; ++group) {
          unsigned subc = 12 * prb + group + 2 * kp;
          for (unsigned port = 0; port != pdu->nof_ports; ++port) {
            float accr = 0, acci = 0;
            for (unsigned j = 2 * group; j != 2 * group + 2 && j != pdu->nof_layers; ++j) {
              /* CDM: w_f = {+1, -1} on odd DM-RS ports flips every other pilot; w_t = +1 (single symbol). */
              float sign = ((j & 1U) && (kp & 1U)) ? -1.0F : 1.0F;
              float xr = sign * seq[2 * kp], xi = sign * seq[2 * kp + 1];
              float pr, pi;
              cmul_fmaddsub(xr, xi, w[2 * (port * pdu->nof_layers + j)], w[2 * (port * pdu->nof_layers + j) + 1], &pr,
                            &pi);
              if (j == 2 * group) {
                accr = pr;
                acci = pi;
              } else {
                accr += pr;
                acci += pi;
              }
            }
            size_t o        = 2 * (((size_t)port * 14 + l) * nof_subc + subc);
            grid[o]         = to_bf16(accr);
            grid[o + 1]     = to_bf16(acci);
          }
        }
      }
    }
  }
}

/* ---- NZP-CSI-RS ("next" row) ---------------------------------------------------------------
 * nzp_csi_rs_generator_impl::map (R/lib/phy/upper/signal_processors/nzp_csi_rs_generator_impl.cpp:96-352), the RE
 * patterns of get_csi_rs_pattern (R/lib/ran/csi_rs/csi_rs_pattern.cpp: rows 1-5 of TS 38.211 Table 7.4.1.5.3-1) and
 * the generic branch of resource_grid_mapper_impl::map (resource_grid_mapper_impl.cpp:150-277).
 * Per CDM group: one QPSK sequence per OFDM symbol of the group (c_init of TS 38.211 Section 7.4.1.5.2, the elements
 * below the first occupied PRB skipped), the group's second port is the first with every other element negated
 * (FD-CDM2), the group is precoded onto all ports and written over whatever the grid holds there. */
#define CSI_DENSITY_DOT5_EVEN 0
#define CSI_DENSITY_DOT5_ODD 1
#define CSI_DENSITY_ONE 2
#define CSI_DENSITY_THREE 3

int oracle_csi_rs_validate(const nrphy_csi_rs_cfg_t* c)
{
  static const unsigned row_ports[6] = {0, 1, 1, 2, 4, 4};
  if (c == NULL || c->row < 1 || c->row > 5 || c->nof_k_ref != 1 || c->cp > 1 || c->nof_rb == 0 ||
      c->nof_ports != row_ports[c->row] || c->precoding == NULL || c->nof_prg != 1 || c->prg_size_rb == 0 || c->prg_size_rb > NRPHY_MAX_RB ||
      c->start_rb + c->nof_rb > NRPHY_MAX_RB) {
    return NRPHY_ERR_ARGUMENT;
  }
  unsigned nsymb = c->cp ? 12 : 14, k0 = c->k_ref[0];
  switch (c->row) { /* the assertions of mapping_row_1 .. mapping_row_5 */
    case 1:
      return (k0 <= 3 && c->density == CSI_DENSITY_THREE && c->cdm == 0 && c->symbol_l0 < nsymb) ? NRPHY_OK : NRPHY_ERR_ARGUMENT;
    case 2:
      return (k0 < 12 && c->density < CSI_DENSITY_THREE && c->cdm == 0 && c->symbol_l0 < nsymb) ? NRPHY_OK : NRPHY_ERR_ARGUMENT;
    case 3:
      return (k0 < 11 && c->density < CSI_DENSITY_THREE && c->cdm == 1 && c->symbol_l0 < nsymb) ? NRPHY_OK : NRPHY_ERR_ARGUMENT;
    case 4:
      return (k0 < 9 && c->density == CSI_DENSITY_ONE && c->cdm == 1 && c->symbol_l0 < nsymb) ? NRPHY_OK : NRPHY_ERR_ARGUMENT;
    default:
      return (k0 < 11 && c->density == CSI_DENSITY_ONE && c->cdm == 1 && c->symbol_l0 + 1 < nsymb) ? NRPHY_OK : NRPHY_ERR_ARGUMENT;
  }
}

int oracle_csi_rs_map(const nrphy_csi_rs_cfg_t* c, uint16_t* grid, uint32_t nof_ports, uint32_t nof_subc)
{
  if (oracle_csi_rs_validate(c) != NRPHY_OK || nof_ports < c->nof_ports || nof_subc < 12 * (c->start_rb + c->nof_rb)) {
    return NRPHY_ERR_ARGUMENT;
  }
  const unsigned group_size = c->cdm ? 2 : 1, nof_groups = c->nof_ports / group_size;
  const int      half = c->density == CSI_DENSITY_DOT5_EVEN || c->density == CSI_DENSITY_DOT5_ODD;
  /* PRB range and stride (build_re_patterns) */
  unsigned rb_begin = c->start_rb, rb_end = c->start_rb + c->nof_rb, rb_stride = half ? 2 : 1;
  if (half && (((c->start_rb % 2) != 0) == (c->density == CSI_DENSITY_DOT5_EVEN))) {
    ++rb_begin;
  }
  /* sequence elements below the first occupied PRB (get_nof_skipped_elements) and per symbol (get_seq_len) */
  unsigned first_prb = c->start_rb;
  if (c->density == CSI_DENSITY_DOT5_EVEN) {
    first_prb = c->start_rb + c->start_rb % 2;
  } else if (c->density == CSI_DENSITY_DOT5_ODD) {
    first_prb = c->start_rb + (1 - c->start_rb % 2);
  }
  unsigned advance = 0;
  if (c->density == CSI_DENSITY_THREE) {
    advance = 3 * first_prb;
  } else if (c->density == CSI_DENSITY_ONE) {
    advance = (c->row == 2) ? first_prb : 2 * first_prb;
  } else {
    advance = (c->row == 2) ? first_prb / 2 : first_prb;
  }
  unsigned seq_len = c->nof_rb;
  if (half) {
    seq_len /= 2;
    if (c->nof_rb % 2 != 0 && (((c->start_rb % 2) != 0) == (c->density == CSI_DENSITY_DOT5_ODD))) {
      ++seq_len;
    }
  } else if (c->density == CSI_DENSITY_THREE) {
    seq_len *= 3;
  }
  if (c->cdm) {
    seq_len *= 2;
  }
  const unsigned nsymb     = c->cp ? 12 : 14;
  const float    amplitude = (float)(M_SQRT1_2 * (double)c->amplitude);
  float*         seq       = (float*)malloc(sizeof(float) * 2 * (seq_len ? seq_len : 1));
  for (unsigned g = 0; g != nof_groups; ++g) {
    /* k_bar / l_bar of the group's first port (mapping_row_n) and its RE mask within a PRB */
    unsigned k_bar = c->k_ref[0], l_bar = c->symbol_l0, re_mask = 0;
    if (c->row == 4) {
      k_bar += 2 * g;
    } else if (c->row == 5) {
      l_bar += g;
    }
    if (c->row == 1) {
      re_mask = (1U << k_bar) | (1U << (k_bar + 4)) | (1U << (k_bar + 8));
    } else if (c->cdm == 0) {
      re_mask = 1U << k_bar;
    } else {
      re_mask = 3U << k_bar;
    }
    /* one symbol per group for these rows */
    const unsigned l      = l_bar;
    const uint32_t c_init = (uint32_t)((1024ULL * (nsymb * c->slot_index + l + 1) * (2 * c->scrambling_id + 1) + c->scrambling_id) & 0x7FFFFFFFULL);
    oracle_prg_generate_float(c_init, 2 * advance, amplitude, seq, 2 * seq_len);
    unsigned m = 0; /* element of the sequence = RE of the pattern in ascending subcarrier order */
    for (unsigned prb = rb_begin; prb < rb_end; prb += rb_stride) {
      for (unsigned k = 0; k != 12; ++k) {
        if (!((re_mask >> k) & 1U)) {
          continue;
        }
        const unsigned subc = 12 * prb + k;
        /* one PRG: the generator builds the group's precoding with a single PRG (nzp_csi_rs_generator_impl.cpp:259) */
        const float*   w    = c->precoding;
        for (unsigned port = 0; port != c->nof_ports; ++port) {
          float accr = 0, acci = 0;
          for (unsigned j = 0; j != group_size; ++j) {
            const unsigned layer = g * group_size + j;
            const float    sign  = (j == 1 && (m & 1U)) ? -1.0F : 1.0F; /* fd_cdm2_table: w_f = {+1, -1} */
            float          pr, pi;
            cmul_fmaddsub(sign * seq[2 * m], sign * seq[2 * m + 1], w[2 * (port * c->nof_ports + layer)],
                          w[2 * (port * c->nof_ports + layer) + 1], &pr, &pi);
            if (j == 0) {
              accr = pr;
              acci = pi;
            } else {
              accr += pr;
              acci += pi;
            }
          }
          size_t o    = 2 * (((size_t)port * 14 + l) * nof_subc + subc);
          grid[o]     = to_bf16(accr);
          grid[o + 1] = to_bf16(acci);
        }
        ++m;
      }
    }
    if (m != seq_len) { /* the reference asserts that the pattern consumes the whole sequence */
      free(seq);
      return NRPHY_ERR_ARGUMENT;
    }
  }
  free(seq);
  return NRPHY_OK;
}

int oracle_pdsch_process(const nrphy_pdsch_pdu_t* pdu, const uint8_t* tb, uint16_t* grid, uint32_t nof_ports,
                         uint32_t nof_subc, uint8_t* cw_rm, uint8_t* cw_scrambled)
{
  if (oracle_pdsch_validate(pdu) != NRPHY_OK) {
    return NRPHY_ERR_INVALID_PDU;
  }
  if (grid != NULL && (nof_ports < pdu->nof_ports || nof_subc < 12 * (unsigned)(prb_highest(pdu->prb_mask) + 1))) {
    return NRPHY_ERR_ARGUMENT;
  }
  nrphy_pdsch_derived_t d;
  oracle_pdsch_derive(pdu, &d);
  /* Nothing to map, or more codeblocks than resource elements (a codeblock without a single channel bit: the reference's rate
   * matcher asserts): refused, like the product's plan creation does. */
  if (d.nof_re == 0 || d.lifting_size == 0 || d.nof_codeblocks > NRPHY_MAX_CODEBLOCKS || d.nof_codeblocks > d.nof_re ||
      d.rm_length_short == 0) {
    return NRPHY_ERR_INVALID_PDU;
  }
  unsigned G    = d.codeword_bits;
  uint8_t* bits = (uint8_t*)malloc(G);
  int      rc   = encode_codeword(pdu, &d, tb, bits);
  if (rc != NRPHY_OK) {
    free(bits);
    return rc;
  }
  if (cw_rm) {
    pack_bits(cw_rm, bits, G);
  }
  /* Scrambling (TS 38.211 Section 7.3.1.1; pdsch_modulator_impl.cpp:30-44), q = 0. */
  uint8_t* packed = (uint8_t*)malloc((G + 7) / 8);
  pack_bits(packed, bits, G);
  oracle_prg_apply_xor((pdu->rnti << 15) + pdu->n_id, 0, packed, G);
  if (cw_scrambled) {
    memcpy(cw_scrambled, packed, (G + 7) / 8);
  }
  if (grid == NULL) {
    free(packed);
    free(bits);
    return NRPHY_OK;
  }
  /* Modulation (ci8) and its scaling folded into the weights (pdsch_modulator_impl.cpp:98-105). */
  unsigned nsym    = G / pdu->qm;
  int8_t*  symbols = (int8_t*)malloc(2 * (size_t)nsym);
  float    scaling = oracle_modulate_ci8(pdu->qm, packed, nsym, symbols);
  float    cfg_scaling = powf(10.0F, -pdu->ratio_pdsch_data_to_sss_dB / 20.0F);
  if (isnormal(cfg_scaling)) {
    scaling *= cfg_scaling;
  }
  unsigned L = pdu->nof_layers, P = pdu->nof_ports;
  float*   w = (float*)malloc(sizeof(float) * 2 * pdu->nof_prg * P * L);
  for (unsigned i = 0; i != 2 * pdu->nof_prg * P * L; ++i) {
    w[i] = pdu->precoding[i] * scaling;
  }
  /* Layer mapping + precoding + RE mapping (resource_grid_mapper_impl.cpp:279-437, channel_precoder_avx2.cpp). */
  uint8_t* mask = (uint8_t*)malloc(nof_subc);
  unsigned re   = 0;
  for (unsigned l = 0; l != 14; ++l) {
    data_re_mask(pdu, l, mask, nof_subc);
    for (unsigned k = 0; k != nof_subc; ++k) {
      if (!mask[k]) {
        continue;
      }
      unsigned prg = k / (12 * pdu->prg_size_rb);
      if (prg >= pdu->nof_prg) {
        prg = pdu->nof_prg - 1;
      }
      const float* wp = w + 2 * (size_t)prg * P * L;
      for (unsigned port = 0; port != P; ++port) {
        float accr = 0, acci = 0;
        for (unsigned j = 0; j != L; ++j) {
          float xr = (float)symbols[2 * (re * L + j)], xi = (float)symbols[2 * (re * L + j) + 1];
          float pr, pi;
          cmul_fmaddsub(xr, xi, wp[2 * (port * L + j)], wp[2 * (port * L + j) + 1], &pr, &pi);
          if (j == 0) {
            accr = pr;
            acci = pi;
          } else {
            accr += pr;
            acci += pi;
          }
        }
        size_t o    = 2 * (((size_t)port * 14 + l) * nof_subc + k);
        grid[o]     = to_bf16(accr);
        grid[o + 1] = to_bf16(acci);
      }
      ++re;
    }
  }
  rc = (re == d.nof_re) ? NRPHY_OK : NRPHY_ERR_ARGUMENT;
  put_dmrs(pdu, grid, nof_subc);
  free(mask);
  free(w);
  free(symbols);
  free(packed);
  free(bits);
  return rc;
}

/* ------------------------------------------------------------------------------------------------ */
/* DFT + OFDM modulator (TS 38.211 Section 5.3.1 / 5.4; R/lib/phy/lower/modulation/ofdm_modulator_impl.cpp:33-139, */
/* phase_compensation_lut.h:49-82, R/include/srsran/ran/cyclic_prefix.h:93-104,                        */
/* R/lib/phy/generic_functions/dft_processor_generic_impl.cpp:14-218)                                  */
/* ------------------------------------------------------------------------------------------------ */
/* Un-normalised DFT, sign -1 direct / +1 inverse.  Sizes 2^a * 3^b (radix-2 recursion over a naive core),
 * evaluated in double precision and rounded once: this is the accuracy reference for the fp32 kernels. */
static void dft_rec(unsigned n, unsigned stride, const double* in, double* out, double sign)
{
  if (n % 2 != 0 || n <= 4) {
    for (unsigned k = 0; k != n; ++k) {
      double sr = 0, si = 0;
      for (unsigned t = 0; t != n; ++t) {
        double ang = sign * 2.0 * M_PI * (double)((k * t) % n) / (double)n;
        double c = cos(ang), s = sin(ang);
        double xr = in[2 * t * stride], xi = in[2 * t * stride + 1];
        sr += xr * c - xi * s;
        si += xr * s + xi * c;
      }
      out[2 * k]     = sr;
      out[2 * k + 1] = si;
    }
    return;
  }
  dft_rec(n / 2, 2 * stride, in, out, sign);
  dft_rec(n / 2, 2 * stride, in + 2 * stride, out + n, sign);
  for (unsigned k = 0; k != n / 2; ++k) {
    double ang = sign * 2.0 * M_PI * (double)k / (double)n;
    double c = cos(ang), s = sin(ang);
    double pr = out[2 * k], pi = out[2 * k + 1];
    double qr = out[2 * (k + n / 2)] * c - out[2 * (k + n / 2) + 1] * s;
    double qi = out[2 * (k + n / 2)] * s + out[2 * (k + n / 2) + 1] * c;
    out[2 * k]               = pr + qr;
    out[2 * k + 1]           = pi + qi;
    out[2 * (k + n / 2)]     = pr - qr;
    out[2 * (k + n / 2) + 1] = pi - qi;
  }
}

int oracle_dft(uint32_t n, int inverse, const float* in, float* out)
{
  if (n == 0) {
    return NRPHY_ERR_ARGUMENT;
  }
  double* a = (double*)malloc(sizeof(double) * 2 * n);
  double* b = (double*)malloc(sizeof(double) * 2 * n);
  for (uint32_t i = 0; i != 2 * n; ++i) {
    a[i] = in[i];
  }
  dft_rec(n, 1, a, b, inverse ? 1.0 : -1.0);
  for (uint32_t i = 0; i != 2 * n; ++i) {
    out[i] = (float)b[i];
  }
  free(a);
  free(b);
  return NRPHY_OK;
}

static unsigned cp_length(const nrphy_ofdm_config_t* c, unsigned symbol_index)
{
  /* In units of kappa*Tc at 15 kHz reference: 144 (160 for symbols 0 and 7*2^mu) >> mu, extended 512 >> mu;
   * samples = units * sampling_rate / 30.72e6 = units * dft_size * 2^mu / 2048. */
  unsigned mu    = c->numerology;
  unsigned units = 144U >> mu;
  if (c->cp) {
    units = 512U >> mu;
  } else if (symbol_index == 0 || symbol_index == 7U * (1U << mu)) {
    units += 16;
  }
  return (unsigned)(((uint64_t)units * c->dft_size * (1U << mu)) / 2048U);
}

uint32_t oracle_ofdm_symbol_size(const nrphy_ofdm_config_t* c, uint32_t symbol_index)
{
  return cp_length(c, symbol_index) + c->dft_size;
}

uint32_t oracle_ofdm_slot_size(const nrphy_ofdm_config_t* c, uint32_t slot_index)
{
  unsigned nsymb = c->cp ? 12 : 14, n = 0;
  for (unsigned l = 0; l != nsymb; ++l) {
    n += oracle_ofdm_symbol_size(c, nsymb * slot_index + l);
  }
  return n;
}

/* Fast single-precision iterative radix-2 inverse FFT used by the timed CPU baseline (power-of-two sizes). */
static void ifft_pow2_f32(unsigned n, float* x, const float* tw /* n/2 complex, e^{+j2pi k/n} */)
{
  for (unsigned i = 1, j = 0; i < n; ++i) {
    unsigned bit = n >> 1;
    for (; j & bit; bit >>= 1) {
      j ^= bit;
    }
    j ^= bit;
    if (i < j) {
      float tr = x[2 * i], ti = x[2 * i + 1];
      x[2 * i]     = x[2 * j];
      x[2 * i + 1] = x[2 * j + 1];
      x[2 * j]     = tr;
      x[2 * j + 1] = ti;
    }
  }
  for (unsigned len = 2; len <= n; len <<= 1) {
    unsigned step = n / len;
    for (unsigned i = 0; i < n; i += len) {
      for (unsigned k = 0; k != len / 2; ++k) {
        float wr = tw[2 * k * step], wi = tw[2 * k * step + 1];
        float* a = x + 2 * (i + k);
        float* b = x + 2 * (i + k + len / 2);
        float  qr = b[0] * wr - b[1] * wi, qi = b[0] * wi + b[1] * wr;
        b[0] = a[0] - qr;
        b[1] = a[1] - qi;
        a[0] += qr;
        a[1] += qi;
      }
    }
  }
}

static int ofdm_modulate_slot(const nrphy_ofdm_config_t* c, const uint16_t* grid, uint32_t nof_ports,
                              uint32_t slot_index, float* iq, int fast)
{
  unsigned N = c->dft_size, rg = 12 * c->bw_rb, nsymb = c->cp ? 12 : 14;
  if (N <= rg || (fast && (N & (N - 1)) != 0)) {
    return NRPHY_ERR_ARGUMENT;
  }
  unsigned slot_size = oracle_ofdm_slot_size(c, slot_index);
  double   srate     = 15000.0 * (double)(1U << c->numerology) * (double)N;
  float*   in        = (float*)malloc(sizeof(float) * 2 * N);
  float*   out       = (float*)malloc(sizeof(float) * 2 * N);
  float*   tw        = NULL;
  if (fast) {
    tw = (float*)malloc(sizeof(float) * N);
    for (unsigned k = 0; k != N / 2; ++k) {
      tw[2 * k]     = (float)cos(2.0 * M_PI * k / N);
      tw[2 * k + 1] = (float)sin(2.0 * M_PI * k / N);
    }
  }
  for (uint32_t port = 0; port != nof_ports; ++port) {
    float* o = iq + 2 * (size_t)port * slot_size;
    for (unsigned l = 0; l != nsymb; ++l) {
      unsigned sym = nsymb * slot_index + l;
      unsigned cp  = cp_length(c, sym);
      /* Start of the symbol's useful part counted from the subframe start (phase_compensation_lut.h:63-80). */
      unsigned offset = 0;
      for (unsigned s = 0; s <= sym; ++s) {
        offset += cp_length(c, s);
        if (s != sym) {
          offset += N;
        }
      }
      double phase = -2.0 * M_PI * c->center_freq_hz * ((double)offset / srate);
      float  pr = (float)cos(phase), pi = (float)sin(phase);
      float  cr = pr * c->scale, ci = pi * c->scale;
      const uint16_t* g = grid + 2 * (((size_t)port * 14 + l) * rg);
      memset(in, 0, sizeof(float) * 2 * N);
      for (unsigned k = 0; k != rg / 2; ++k) { /* lower half of the grid -> top DFT bins */
        in[2 * (N - rg / 2 + k)]     = from_bf16(g[2 * k]);
        in[2 * (N - rg / 2 + k) + 1] = from_bf16(g[2 * k + 1]);
      }
      for (unsigned k = rg / 2; k != rg; ++k) { /* upper half -> bins from DC */
        in[2 * (k - rg / 2)]     = from_bf16(g[2 * k]);
        in[2 * (k - rg / 2) + 1] = from_bf16(g[2 * k + 1]);
      }
      if (fast) {
        memcpy(out, in, sizeof(float) * 2 * N);
        ifft_pow2_f32(N, out, tw);
      } else {
        oracle_dft(N, 1, in, out);
      }
      for (unsigned n = 0; n != N; ++n) {
        float xr = out[2 * n], xi = out[2 * n + 1];
        o[2 * (cp + n)]     = xr * cr - xi * ci;
        o[2 * (cp + n) + 1] = xr * ci + xi * cr;
      }
      memcpy(o, o + 2 * N, sizeof(float) * 2 * cp);
      o += 2 * (cp + N);
    }
  }
  free(tw);
  free(in);
  free(out);
  return (int)slot_size;
}

int oracle_ofdm_modulate_slot(const nrphy_ofdm_config_t* c, const uint16_t* grid, uint32_t nof_ports,
                              uint32_t slot_index, float* iq)
{
  return ofdm_modulate_slot(c, grid, nof_ports, slot_index, iq, 0);
}

/* ------------------------------------------------------------------------------------------------ */
/* LDPC decoder: layered scaled min-sum on int8 log-likelihood ratios                                  */
/* (R/lib/phy/upper/channel_coding/ldpc/ldpc_decoder_impl.cpp:60-318, ldpc_decoder_generic.cpp:30-128,   */
/*  LLR arithmetic R/lib/phy/upper/log_likelihood_ratio.cpp:37-87).                                    */
/* llr: nof_llr soft bits of the codeblock WITHOUT its first 2*Zc (punctured) bits, as the rate          */
/* dematcher delivers them.  message_bits: Kb*Zc hard bits, one per byte.  crc_poly_id: 0 = no early     */
/* stop, else the polynomial of oracle_crc_bits checked over the first Kb*Zc - nof_filler bits.          */
/* ------------------------------------------------------------------------------------------------ */
#define LLR_MAX_VALUE 120
#define LLR_INF_VALUE 127

static int llr_isinf(int v)
{
  return v > LLR_MAX_VALUE || v < -LLR_MAX_VALUE;
}

static int llr_special_sum(int a, int b, int* out)
{
  if (a == -b) {
    *out = 0;
    return 1;
  }
  if (llr_isinf(a)) {
    *out = a;
    return 1;
  }
  if (llr_isinf(b)) {
    *out = b;
    return 1;
  }
  return 0;
}

static int llr_add(int a, int b) /* saturating sum */
{
  int r;
  if (llr_special_sum(a, b, &r)) {
    return r;
  }
  r = a + b;
  return r > LLR_MAX_VALUE ? LLR_MAX_VALUE : (r < -LLR_MAX_VALUE ? -LLR_MAX_VALUE : r);
}

static int llr_promotion_sum(int a, int b) /* beyond the range the bit becomes certain */
{
  int r;
  if (llr_special_sum(a, b, &r)) {
    return r;
  }
  r = a + b;
  return r > LLR_MAX_VALUE ? LLR_INF_VALUE : (r < -LLR_MAX_VALUE ? -LLR_INF_VALUE : r);
}

/* ---- LDPC rate dematcher ("next" row, receive side) ----------------------------------------------------------------
 * ldpc_rate_dematcher_impl::rate_dematch (R/lib/phy/upper/channel_coding/ldpc/ldpc_rate_dematcher_impl.cpp:43-103):
 * undo the bit interleaver (:202-256), then walk the circular buffer from k0 the way the rate matcher read it
 * (allot_llrs, :118-200), skipping the filler bits, wrapping at the (Nref-limited) buffer length.  With new data the
 * first visit of a position copies, every later visit adds with the saturating LLR sum; without, every visit adds to
 * what the soft buffer held.  Filler bits become +infinity (new data only).
 *
 * The walk below keeps the reference's order of operations including what it leaves untouched: with new data it
 * clears [0, k0) of the systematic part, the stretch after the last written position only when the input ended
 * before the first wrap -- and then counted from the END of the full-length block (which differs from "the rest of
 * the buffer" when Nref shortens it) -- and nothing between the systematic part and a k0 beyond it.  Whatever is not
 * mentioned keeps its old content, so out is an in/out argument in every mode. */
int oracle_ldpc_rate_dematch(uint32_t bg, uint32_t zc, uint32_t rv, uint32_t qm, uint32_t nref, uint32_t nof_filler,
                             int new_data, const int8_t* in, uint32_t e, int8_t* out)
{
  static const double shift_bg1[4] = {0, 17, 33, 56}, shift_bg2[4] = {0, 13, 25, 43};
  const uint32_t      n_short = (bg == 1) ? 66 : 50, bg_k = (bg == 1) ? 22 : 10;
  const uint32_t      block_length = n_short * zc;
  const uint32_t      buffer_length = (nref > 0 && nref < block_length) ? nref : block_length;
  const uint32_t      nof_systematic = (bg_k - 2) * zc;
  if (rv > 3 || qm == 0 || e % qm != 0 || nof_filler >= nof_systematic || buffer_length <= nof_systematic) {
    return -1;
  }
  const uint32_t nof_info = nof_systematic - nof_filler;
  const double   frac     = (((bg == 1) ? shift_bg1 : shift_bg2)[rv] * buffer_length) / block_length;
  const uint32_t k0       = (uint32_t)((uint16_t)floor(frac)) * zc;

  /* deinterleave: row j of the Qm x (E / Qm) table is the j-th bit of every symbol */
  int8_t* seq = (int8_t*)malloc(e ? e : 1);
  if (qm == 1) {
    memcpy(seq, in, e);
  } else {
    const uint32_t cols = e / qm;
    for (uint32_t i = 0; i != cols; ++i) {
      for (uint32_t j = 0; j != qm; ++j) {
        seq[cols * j + i] = in[i * qm + j];
      }
    }
  }

  int      copying = new_data != 0;
  uint32_t pos = k0, taken = 0;
  while (taken != e) {
    uint32_t left = e - taken;
    if (pos < nof_info) { /* information bits up to the first filler bit */
      uint32_t n = nof_info - pos < left ? nof_info - pos : left;
      if (copying) {
        memset(out, 0, pos);
        memcpy(out + pos, seq + taken, n);
      } else {
        for (uint32_t i = 0; i != n; ++i) {
          out[pos + i] = (int8_t)llr_add(seq[taken + i], out[pos + i]); /* operator+ adds the left operand to the right */
        }
      }
      pos += n;
      taken += n;
      left -= n;
    } else if (copying) {
      memset(out, 0, nof_info);
    }
    if (copying) {
      memset(out + nof_info, LLR_INF_VALUE, nof_filler);
    }
    if (pos < nof_systematic) {
      pos = nof_systematic;
    }
    { /* parity bits up to the end of the buffer */
      uint32_t n = buffer_length - pos < left ? buffer_length - pos : left;
      if (copying) {
        memcpy(out + pos, seq + taken, n);
      } else {
        for (uint32_t i = 0; i != n; ++i) {
          out[pos + i] = (int8_t)llr_add(seq[taken + i], out[pos + i]); /* operator+ adds the left operand to the right */
        }
      }
      pos = (pos + n) % buffer_length;
      taken += n;
    }
    if (taken != e) {
      copying = 0; /* after the first wrap everything is combined */
    }
  }
  if (copying && pos != 0) {
    memset(out + block_length - (buffer_length - pos), 0, buffer_length - pos);
  }
  free(seq);
  return 0;
}

int oracle_ldpc_decode(uint32_t bg, uint32_t zc, uint32_t nof_filler, uint32_t crc_poly_id, uint32_t max_iterations,
                       float scaling_factor, const int8_t* llr, uint32_t nof_llr, uint8_t* message_bits)
{
  const nr_ldpc_edge_t* edges   = (bg == 1) ? NR_LDPC_BG1_EDGES : NR_LDPC_BG2_EDGES;
  const unsigned        n_edges = (bg == 1) ? NR_LDPC_BG1_NOF_EDGES : NR_LDPC_BG2_NOF_EDGES;
  const unsigned        bg_k = (bg == 1) ? 22 : 10, n_full = (bg == 1) ? 68 : 52, n_short = n_full - 2;
  const unsigned        n_hr = bg_k + 4, bg_m = n_full - bg_k;
  const int             ils  = lifting_set_index(zc);
  const unsigned        K    = bg_k * zc;
  if (ils < 0 || nof_llr > n_short * zc || nof_llr < K + 2 * zc || max_iterations == 0 ||
      !(scaling_factor > 0.f && scaling_factor < 1.f)) {
    return NRPHY_ERR_ARGUMENT;
  }
  /* Last non-zero soft bit (ldpc_decoder_impl.cpp:88-97). */
  unsigned input_size = nof_llr;
  while (input_size != 0 && llr[input_size - 1] == 0) {
    --input_size;
  }
  if (input_size == 0) {
    if (crc_poly_id == 0) {
      memset(message_bits, 1, K);
    }
    return 0;
  }
  int8_t* soft = (int8_t*)calloc((size_t)n_full * zc, 1);
  int8_t* v2c  = (int8_t*)calloc((size_t)(n_hr + 1) * zc, 1);
  int8_t* c2v  = (int8_t*)calloc((size_t)bg_m * (n_hr + 1) * zc, 1);
  uint8_t init[46];
  memset(init, 0, sizeof(init));
  /* load_soft_bits (:128-164): whole nodes clamped to +-64, the tail copied as it is, two punctured nodes first. */
  {
    unsigned full = nof_llr / zc;
    for (unsigned i = 0; i != full * zc; ++i) {
      int v            = llr[i];
      soft[2 * zc + i] = (int8_t)(v > 64 ? 64 : (v < -64 ? -64 : v));
    }
    for (unsigned i = full * zc; i != nof_llr; ++i) {
      soft[2 * zc + i] = llr[i];
    }
  }
  unsigned cb_len = input_size + 2 * zc;
  if (cb_len < K + 4 * zc) {
    cb_len = K + 4 * zc;
  }
  if (cb_len % zc != 0) {
    cb_len = (cb_len / zc + 1) * zc;
  }
  const unsigned nof_layers = cb_len / zc - bg_k;
  int            result     = 0;
  int8_t         min1[384], min2[384];
  uint8_t        min_idx[384], sign_prod[384];
  for (unsigned it = 0; it != max_iterations && result == 0; ++it) {
    for (unsigned m = 0; m != nof_layers; ++m) {
      /* Edges of check row m in adjacency order (ascending variable index). */
      unsigned first = 0, deg = 0;
      for (unsigned e = 0; e != n_edges; ++e) {
        if (edges[e].row == m) {
          if (deg == 0) {
            first = e;
          }
          ++deg;
        }
      }
      /* Variable-to-check messages (:166-212): slot = variable index, every extension variable in slot n_hr. */
      for (unsigned t = 0; t != deg; ++t) {
        unsigned var = edges[first + t].col, slot = var < n_hr ? var : n_hr;
        for (unsigned j = 0; j != zc; ++j) {
          int s = soft[var * zc + j];
          v2c[slot * zc + j] =
              (int8_t)(init[m] ? llr_add(s, -(int)c2v[((size_t)m * (n_hr + 1) + slot) * zc + j]) : s);
        }
      }
      /* Two smallest magnitudes and the sign product per check (ldpc_decoder_generic.cpp:44-67). */
      for (unsigned j = 0; j != zc; ++j) {
        min1[j]      = LLR_MAX_VALUE;
        min2[j]      = LLR_MAX_VALUE;
        min_idx[j]   = 0;
        sign_prod[j] = 0;
      }
      for (unsigned t = 0; t != deg; ++t) {
        unsigned var = edges[first + t].col, slot = var < n_hr ? var : n_hr;
        unsigned shift = edges[first + t].shift[ils] % zc;
        for (unsigned j = 0; j != zc; ++j) {
          int v = v2c[slot * zc + (j + shift) % zc];
          int a = v < 0 ? -v : v;
          int is_min = a < min1[j];
          int new2   = is_min ? min1[j] : a;
          if (a < min2[j]) {
            min2[j] = (int8_t)new2;
          }
          if (is_min) {
            min1[j]    = (int8_t)a;
            min_idx[j] = (uint8_t)t;
          }
          sign_prod[j] ^= (v >= 0) ? 0U : 1U;
        }
      }
      /* Check-to-variable messages (:82-107) and soft-bit update (:109-121, ldpc_decoder_impl.cpp:231-245). */
      for (unsigned t = 0; t != deg; ++t) {
        unsigned var = edges[first + t].col, slot = var < n_hr ? var : n_hr;
        unsigned shift = edges[first + t].shift[ils] % zc;
        for (unsigned j = 0; j != zc; ++j) {
          unsigned k = (j + zc - shift) % zc;
          int      v = (t != min_idx[k]) ? min1[k] : min2[k];
          if (!llr_isinf(v)) {
            v = (int)roundf((float)v * scaling_factor);
          }
          int vc   = v2c[slot * zc + j];
          int sign = sign_prod[k] ^ ((vc >= 0) ? 0U : 1U);
          v        = sign ? -(v < 0 ? -v : v) : (v < 0 ? -v : v);
          c2v[((size_t)m * (n_hr + 1) + slot) * zc + j] = (int8_t)v;
          soft[var * zc + j]                            = (int8_t)llr_promotion_sum(v, vc);
        }
      }
      init[m] = 1;
    }
    if (crc_poly_id != 0) {
      int ok = 1;
      for (unsigned i = 0; i != K; ++i) {
        message_bits[i] = soft[i] <= 0;
        ok &= soft[i] != 0;
      }
      if (ok) {
        uint8_t* packed = (uint8_t*)calloc((K + 7) / 8, 1);
        pack_bits(packed, message_bits, K - nof_filler);
        if (oracle_crc_bits(crc_poly_id, packed, K - nof_filler) == 0) {
          result = (int)it + 1;
        }
        free(packed);
      }
    }
  }
  if (crc_poly_id == 0) {
    for (unsigned i = 0; i != K; ++i) {
      message_bits[i] = soft[i] <= 0;
    }
  }
  free(soft);
  free(v2c);
  free(c2v);
  return result;
}

/* ------------------------------------------------------------------------------------------------ */
/* OFDM demodulator (R/lib/phy/lower/modulation/ofdm_demodulator_impl.cpp:98-171): per symbol skip the  */
/* cyclic prefix minus the window offset, direct DFT, x (receive phase compensation x scale)            */
/* [phase_compensation_lut.h:49-82 with is_tx = false], x exp(+j 2 pi window_offset i / N) when the     */
/* window is advanced, top bins -> lower half of the grid, bins from DC -> upper half, stored as cbf16. */
/* iq: [nof_ports][slot_size] complex float, grid: [nof_ports][14][12*bw_rb] (re, im) bf16.            */
/* ------------------------------------------------------------------------------------------------ */
int oracle_ofdm_demodulate_slot(const nrphy_ofdm_config_t* c, const float* iq, uint32_t nof_ports,
                                uint32_t slot_index, uint32_t window_offset, uint16_t* grid)
{
  unsigned N = c->dft_size, rg = 12 * c->bw_rb, nsymb = c->cp ? 12 : 14;
  if (N <= rg || window_offset >= (144 * N) / 2048) {
    return NRPHY_ERR_ARGUMENT;
  }
  unsigned slot_size = oracle_ofdm_slot_size(c, slot_index);
  double   srate     = 15000.0 * (double)(1U << c->numerology) * (double)N;
  float*   out       = (float*)malloc(sizeof(float) * 2 * N);
  for (uint32_t port = 0; port != nof_ports; ++port) {
    const float* x = iq + 2 * (size_t)port * slot_size;
    for (unsigned l = 0; l != nsymb; ++l) {
      unsigned sym = nsymb * slot_index + l;
      unsigned cp  = cp_length(c, sym);
      unsigned offset = 0;
      for (unsigned s = 0; s <= sym; ++s) {
        offset += cp_length(c, s);
        if (s != sym) {
          offset += N;
        }
      }
      double phase = 2.0 * M_PI * c->center_freq_hz * ((double)offset / srate);
      float  cr = (float)cos(phase) * c->scale, ci = (float)sin(phase) * c->scale;
      oracle_dft(N, 0, x + 2 * (cp - window_offset), out);
      uint16_t* g = grid + 2 * (((size_t)port * 14 + l) * rg);
      for (unsigned k = 0; k != rg; ++k) {
        unsigned bin = (k < rg / 2) ? N - rg / 2 + k : k - rg / 2;
        float    xr = out[2 * bin], xi = out[2 * bin + 1];
        float    yr = xr * cr - xi * ci, yi = xr * ci + xi * cr;
        if (window_offset != 0) {
          float omega = (float)window_offset * (float)(2.0 * M_PI) / (float)N;
          float wr = cosf(omega * (float)bin), wi = sinf(omega * (float)bin);
          float tr = yr * wr - yi * wi, ti = yr * wi + yi * wr;
          yr = tr;
          yi = ti;
        }
        g[2 * k]     = to_bf16(yr);
        g[2 * k + 1] = to_bf16(yi);
      }
      x += 2 * (cp + N);
    }
  }
  free(out);
  return (int)slot_size;
}

/* ------------------------------------------------------------------------------------------------ */
/* CPU baseline ("port"): T threads x reps slots, one processor instance per thread                   */
/* (scheme of R/tests/benchmarks/phy/upper/channel_processors/pdsch_processor_benchmark.cpp:684-737).   */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
  const nrphy_pdsch_pdu_t*   pdu;
  const uint8_t*             tb;
  uint32_t                   nof_ports, nof_subc, reps;
  const nrphy_ofdm_config_t* ofdm;
} bench_arg_t;

static void* bench_worker(void* p)
{
  bench_arg_t* a    = (bench_arg_t*)p;
  size_t       gsz  = 2 * (size_t)a->nof_ports * 14 * a->nof_subc;
  uint16_t*    grid = (uint16_t*)malloc(gsz * sizeof(uint16_t));
  float*       iq   = NULL;
  if (a->ofdm) {
    iq = (float*)malloc(sizeof(float) * 2 * (size_t)a->nof_ports * oracle_ofdm_slot_size(a->ofdm, 0));
  }
  for (uint32_t r = 0; r != a->reps; ++r) {
    memset(grid, 0, gsz * sizeof(uint16_t));
    oracle_pdsch_process(a->pdu, a->tb, grid, a->nof_ports, a->nof_subc, NULL, NULL);
    if (a->ofdm) {
      ofdm_modulate_slot(a->ofdm, grid, a->nof_ports, 0, iq, 1);
    }
  }
  free(iq);
  free(grid);
  return NULL;
}

double oracle_bench(const nrphy_pdsch_pdu_t* pdu, const uint8_t* tb, uint32_t nof_ports, uint32_t nof_subc,
                    const nrphy_ofdm_config_t* ofdm, uint32_t threads, uint32_t reps)
{
  bench_arg_t     arg = {pdu, tb, nof_ports, nof_subc, reps, ofdm};
  pthread_t*      th  = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (uint32_t t = 0; t != threads; ++t) {
    pthread_create(&th[t], NULL, bench_worker, &arg);
  }
  for (uint32_t t = 0; t != threads; ++t) {
    pthread_join(th[t], NULL);
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  free(th);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
