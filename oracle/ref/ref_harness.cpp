// TEST INFRASTRUCTURE -- not part of the product.
//
// Thin C entry points over the REFERENCE's own implementation classes, compiled from the sources where
// they lie under /root/reference (never copied into this repository).  The resulting library
// oracle/_ref/libsrsref.so is used to (a) pin the C restatement in oracle/nrphy_oracle.c, (b) generate
// the golden vectors under tests/golden/ and the LDPC base-graph data file, and (c) optionally serve
// as the "reference" CPU baseline in bench.py.  This file is original code: it only *calls* the
// reference through its public/private headers (include root = srsRAN-5G-ER/).
#include "mi355_nrphy.h"

#include "lib/phy/generic_functions/dft_processor_generic_impl.h"
#include "lib/phy/generic_functions/precoding/channel_precoder_avx2.h"
#include "lib/phy/generic_functions/precoding/channel_precoder_generic.h"
#include "lib/phy/lower/modulation/ofdm_demodulator_impl.h"
#include "lib/phy/upper/channel_coding/ldpc/ldpc_decoder_avx2.h"
#include "lib/phy/upper/channel_coding/ldpc/ldpc_decoder_generic.h"
#include "lib/phy/lower/modulation/ofdm_modulator_impl.h"
#include "lib/phy/support/resource_grid_impl.h"
#include "lib/phy/upper/channel_coding/crc_calculator_lut_impl.h"
#include "lib/phy/upper/channel_coding/ldpc/ldpc_encoder_avx2.h"
#include "lib/phy/upper/channel_coding/ldpc/ldpc_encoder_generic.h"
#include "lib/phy/upper/channel_coding/ldpc/ldpc_graph_impl.h"
#include "lib/phy/upper/channel_coding/ldpc/ldpc_luts_impl.h"
#include "lib/phy/upper/channel_coding/ldpc/ldpc_rate_dematcher_avx2_impl.h"
#include "lib/phy/upper/channel_coding/ldpc/ldpc_rate_dematcher_impl.h"
#include "lib/phy/upper/channel_coding/ldpc/ldpc_rate_matcher_impl.h"
#include "lib/phy/upper/channel_coding/ldpc/ldpc_segmenter_impl.h"
#include "lib/phy/upper/channel_modulation/modulation_mapper_lut_impl.h"
#include "lib/phy/upper/channel_processors/pdsch_encoder_impl.h"
#include "lib/phy/upper/channel_processors/pusch/pusch_codeblock_decoder.h"
#include "lib/phy/upper/channel_processors/pusch/pusch_decoder_impl.h"
#include "srsran/phy/upper/channel_processors/pusch/pusch_decoder_notifier.h"
#include "srsran/phy/upper/channel_processors/pusch/pusch_decoder_result.h"
#include "srsran/phy/upper/rx_buffer_pool.h"
#include "lib/phy/upper/channel_processors/pdsch_modulator_impl.h"
#include "lib/phy/upper/channel_processors/pdsch_processor_concurrent_impl.h"
#include "lib/phy/upper/channel_processors/pdsch_processor_impl.h"
#include "lib/phy/upper/channel_processors/pdsch_processor_lite_impl.h"
#include "lib/phy/upper/channel_processors/pdsch_processor_validator_impl.h"
#include "lib/phy/upper/sequence_generators/pseudo_random_generator_impl.h"
#include "lib/phy/upper/signal_processors/dmrs_pdsch_processor_impl.h"
#include "lib/phy/upper/signal_processors/nzp_csi_rs_generator_impl.h"
#include "srsran/ran/precoding/precoding_codebooks.h"
#include "srsran/ran/sch/tbs_calculator.h"
#include "srsran/srsvec/bit.h"

#include <chrono>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

using namespace srsran;

namespace {

std::unique_ptr<ldpc_encoder> make_ldpc_encoder(int simd)
{
  if (simd) {
    return std::make_unique<ldpc_encoder_avx2>();
  }
  return std::make_unique<ldpc_encoder_generic>();
}

std::unique_ptr<channel_precoder> make_precoder(int simd)
{
  if (simd) {
    return std::make_unique<channel_precoder_avx2>();
  }
  return std::make_unique<channel_precoder_generic>();
}

std::unique_ptr<ldpc_segmenter_tx> make_segmenter()
{
  ldpc_segmenter_impl::sch_crc crcs;
  crcs.crc16  = std::make_unique<crc_calculator_lut_impl>(crc_generator_poly::CRC16);
  crcs.crc24A = std::make_unique<crc_calculator_lut_impl>(crc_generator_poly::CRC24A);
  crcs.crc24B = std::make_unique<crc_calculator_lut_impl>(crc_generator_poly::CRC24B);
  return ldpc_segmenter_impl::create_ldpc_segmenter_impl_tx(crcs);
}

std::unique_ptr<pdsch_encoder> make_pdsch_encoder(int simd)
{
  return std::make_unique<pdsch_encoder_impl>(
      make_segmenter(), make_ldpc_encoder(simd), std::make_unique<ldpc_rate_matcher_impl>());
}

std::unique_ptr<pdsch_processor> make_processor(int simd)
{
  return std::make_unique<pdsch_processor_impl>(
      make_pdsch_encoder(simd),
      std::make_unique<pdsch_modulator_impl>(std::make_unique<modulation_mapper_lut_impl>(),
                                             std::make_unique<pseudo_random_generator_impl>()),
      std::make_unique<dmrs_pdsch_processor_impl>(std::make_unique<pseudo_random_generator_impl>()));
}

modulation_scheme to_mod(unsigned qm)
{
  switch (qm) {
    case 1:
      return modulation_scheme::BPSK;
    case 2:
      return modulation_scheme::QPSK;
    case 4:
      return modulation_scheme::QAM16;
    case 6:
      return modulation_scheme::QAM64;
    default:
      return modulation_scheme::QAM256;
  }
}

template <size_t N>
void mask_to_bitset(bounded_bitset<N>& out, const uint64_t* words, unsigned nbits)
{
  out.resize(nbits);
  for (unsigned i = 0; i != nbits; ++i) {
    if ((words[i / 64] >> (i % 64)) & 1U) {
      out.set(i);
    }
  }
}

unsigned highest_bit(const uint64_t* words)
{
  int hi = -1;
  for (unsigned i = 0; i != 64 * NRPHY_PRB_WORDS; ++i) {
    if ((words[i / 64] >> (i % 64)) & 1U) {
      hi = i;
    }
  }
  return static_cast<unsigned>(hi + 1);
}

int lowest_bit(const uint64_t* words)
{
  for (unsigned i = 0; i != 64 * NRPHY_PRB_WORDS; ++i) {
    if ((words[i / 64] >> (i % 64)) & 1U) {
      return i;
    }
  }
  return -1;
}

unsigned count_bits(const uint64_t* words)
{
  unsigned c = 0;
  for (unsigned i = 0; i != NRPHY_PRB_WORDS; ++i) {
    c += __builtin_popcountll(words[i]);
  }
  return c;
}

// Translates the POD descriptor into the reference's pdu_t.
pdsch_processor::pdu_t to_pdu(const nrphy_pdsch_pdu_t& in)
{
  pdsch_processor::pdu_t pdu;
  pdu.context      = std::nullopt;
  pdu.slot         = slot_point(4, 0, in.slot_index);
  pdu.rnti         = static_cast<uint16_t>(in.rnti);
  pdu.bwp_size_rb  = in.bwp_size_rb;
  pdu.bwp_start_rb = in.bwp_start_rb;
  pdu.cp           = in.cp ? cyclic_prefix::EXTENDED : cyclic_prefix::NORMAL;
  for (unsigned i = 0; i != in.nof_codewords; ++i) {
    pdu.codewords.push_back({to_mod(in.qm), in.rv});
  }
  pdu.n_id      = in.n_id;
  pdu.ref_point = in.ref_point ? pdsch_processor::pdu_t::PRB0 : pdsch_processor::pdu_t::CRB0;
  // The validator wants a mask as long as the slot (12 symbols with extended cyclic prefix); a bit beyond it makes the
  // PDU invalid, which a 14-bit mask reproduces.
  const unsigned mask_size = (in.cp && (in.dmrs_symbol_mask >> 12) == 0) ? 12 : 14;
  pdu.dmrs_symbol_mask.resize(mask_size);
  for (unsigned l = 0; l != mask_size; ++l) {
    if ((in.dmrs_symbol_mask >> l) & 1U) {
      pdu.dmrs_symbol_mask.set(l);
    }
  }
  pdu.dmrs                        = (in.dmrs_type == 2) ? dmrs_type::TYPE2 : dmrs_type::TYPE1;
  pdu.scrambling_id               = in.scrambling_id;
  pdu.n_scid                      = in.n_scid != 0;
  pdu.nof_cdm_groups_without_data = in.nof_cdm_groups_without_data;
  // Contiguous, non-interleaved allocation: VRB = PRB - bwp_start.
  int      first = lowest_bit(in.prb_mask);
  unsigned count = count_bits(in.prb_mask);
  if (first >= 0 && highest_bit(in.prb_mask) - first == count) {
    pdu.freq_alloc = rb_allocation::make_type1(first - in.bwp_start_rb, count);
  } else {
    bounded_bitset<MAX_RB> vrb(in.bwp_size_rb);
    for (unsigned i = 0; i != in.bwp_size_rb; ++i) {
      unsigned p = i + in.bwp_start_rb;
      if ((in.prb_mask[p / 64] >> (p % 64)) & 1U) {
        vrb.set(i);
      }
    }
    pdu.freq_alloc = rb_allocation::make_type0(vrb);
  }
  pdu.start_symbol_index = in.start_symbol_index;
  pdu.nof_symbols        = in.nof_symbols;
  pdu.ldpc_base_graph    = (in.ldpc_base_graph == 2) ? ldpc_base_graph_type::BG2 : ldpc_base_graph_type::BG1;
  pdu.tbs_lbrm           = units::bytes(in.tbs_lbrm_bytes);
  for (unsigned i = 0; i != in.nof_reserved; ++i) {
    re_pattern p;
    mask_to_bitset(p.prb_mask, in.reserved[i].prb_mask, highest_bit(in.reserved[i].prb_mask));
    for (unsigned k = 0; k != 12; ++k) {
      p.re_mask.set(k, (in.reserved[i].re_mask >> k) & 1U);
    }
    for (unsigned l = 0; l != 14; ++l) {
      p.symbols.set(l, (in.reserved[i].symbol_mask >> l) & 1U);
    }
    pdu.reserved.merge(p);
  }
  pdu.ratio_pdsch_dmrs_to_sss_dB = in.ratio_pdsch_dmrs_to_sss_dB;
  pdu.ratio_pdsch_data_to_sss_dB = in.ratio_pdsch_data_to_sss_dB;
  pdu.precoding = precoding_configuration(in.nof_layers, in.nof_ports, in.nof_prg, in.prg_size_rb);
  for (unsigned g = 0; g != in.nof_prg; ++g) {
    for (unsigned p = 0; p != in.nof_ports; ++p) {
      for (unsigned l = 0; l != in.nof_layers; ++l) {
        const float* w = in.precoding + 2 * ((g * in.nof_ports + p) * in.nof_layers + l);
        pdu.precoding.set_coefficient(cf_t(w[0], w[1]), l, p, g);
      }
    }
  }
  return pdu;
}

class notifier_flag : public pdsch_processor_notifier
{
public:
  bool done = false;
  void on_finish_processing() override { done = true; }
};

void copy_grid_out(uint16_t* out, const resource_grid_reader& reader, unsigned nof_ports, unsigned nof_subc)
{
  for (unsigned p = 0; p != nof_ports; ++p) {
    for (unsigned l = 0; l != 14; ++l) {
      span<const cbf16_t> view = reader.get_view(p, l);
      std::memcpy(out + 2 * (static_cast<size_t>(p * 14 + l) * nof_subc), view.data(), nof_subc * sizeof(cbf16_t));
    }
  }
}

} // namespace

// For oracle/ref/adaptor_harness.cpp: the reference objects behind the PODs and a processor of the reference.
srsran::pdsch_processor::pdu_t ref_make_pdsch_pdu(const nrphy_pdsch_pdu_t& in)
{
  return to_pdu(in);
}
std::unique_ptr<srsran::pdsch_processor> ref_make_pdsch_processor(int simd)
{
  return make_processor(simd);
}
std::unique_ptr<srsran::channel_precoder> ref_make_precoder(int simd)
{
  return make_precoder(simd);
}
std::unique_ptr<srsran::pdsch_encoder> ref_make_pdsch_encoder(int simd)
{
  return make_pdsch_encoder(simd);
}
std::unique_ptr<srsran::ldpc_segmenter_tx> ref_make_segmenter()
{
  return make_segmenter();
}

extern "C" {

// ---- 3GPP TS 38.212 Tables 5.3.2-2 / 5.3.2-3 as the reference holds them -------------------------
// Raw V(i,j) for lifting-set index i_ls: the lifted value for the largest lifting size of the set is
// the raw value because every raw value is below that size.  Returns 0xffff for "no edge".
unsigned ref_bg_raw_shift(unsigned bg, unsigned i_ls, unsigned row, unsigned col)
{
  static const unsigned max_ls[8] = {256, 384, 320, 224, 288, 352, 208, 240};
  ldpc::BG_matrix_t     g =
      ldpc::get_graph(bg == 2 ? ldpc_base_graph_type::BG2 : ldpc_base_graph_type::BG1,
                      static_cast<ldpc::lifting_size_t>(max_ls[i_ls]));
  return g[row][col];
}

unsigned ref_lifting_index(unsigned ls)
{
  return ldpc::get_lifting_index(static_cast<ldpc::lifting_size_t>(ls));
}

unsigned ref_tbs_calculate(unsigned nof_symb_sh,
                           unsigned nof_dmrs_prb,
                           unsigned nof_oh_prb,
                           unsigned qm,
                           float    target_code_rate,
                           unsigned nof_layers,
                           unsigned n_prb)
{
  tbs_calculator_configuration cfg;
  cfg.nof_symb_sh      = nof_symb_sh;
  cfg.nof_dmrs_prb     = nof_dmrs_prb;
  cfg.nof_oh_prb       = nof_oh_prb;
  cfg.mcs_descr        = {to_mod(qm), target_code_rate};
  cfg.nof_layers       = nof_layers;
  cfg.tb_scaling_field = 0;
  cfg.n_prb            = n_prb;
  return tbs_calculator_calculate(cfg);
}

// Writes the 3GPP TS 38.214 precoding codebook matrices the reference benchmark uses:
// kind 0 identity(nof_layers), 1 make_single_port, 2 make_one_layer_two_ports(i), 3 make_two_layer_two_ports(i),
// 4 make_four_layer_four_ports_type1_sp(i11, i2).  Output [nof_ports][nof_layers] complex.  Returns ports<<8|layers.
unsigned ref_precoding_codebook(unsigned kind, unsigned a, unsigned b, unsigned c, float* out)
{
  precoding_weight_matrix m;
  switch (kind) {
    case 0:
      m = make_identity(a);
      break;
    case 1:
      m = make_single_port();
      break;
    case 2:
      m = make_one_layer_two_ports(a);
      break;
    case 3:
      m = make_two_layer_two_ports(a);
      break;
    case 4:
      m = make_one_layer_four_ports_type1_sp_mode1(a, b);
      break;
    case 5:
      m = make_two_layer_four_ports_type1_sp_mode1(a, b, c);
      break;
    case 6:
      m = make_three_layer_four_ports_type1_sp(a, b);
      break;
    default:
      m = make_four_layer_four_ports_type1_sp(a, b);
      break;
  }
  for (unsigned p = 0; p != m.get_nof_ports(); ++p) {
    for (unsigned l = 0; l != m.get_nof_layers(); ++l) {
      cf_t w                                  = m.get_coefficient(l, p);
      out[2 * (p * m.get_nof_layers() + l)]     = w.real();
      out[2 * (p * m.get_nof_layers() + l) + 1] = w.imag();
    }
  }
  return (m.get_nof_ports() << 8) | m.get_nof_layers();
}

int ref_pdsch_validate(const nrphy_pdsch_pdu_t* in)
{
  pdsch_processor_validator_impl validator;
  return validator.is_valid(to_pdu(*in)) ? NRPHY_OK : NRPHY_ERR_INVALID_PDU;
}

// pdsch_processor::process on a zeroed grid; grid = [nof_ports][14][nof_subc] cbf16 (raw bits).
// impl: 0 generic processor, 1 concurrent-less "lite" processor.  simd: 0 generic kernels, 1 AVX2 kernels.
int ref_pdsch_process(const nrphy_pdsch_pdu_t* in,
                      const uint8_t*           tb,
                      uint16_t*                grid_out,
                      unsigned                 nof_ports,
                      unsigned                 nof_subc,
                      int                      impl,
                      int                      simd)
{
  pdsch_processor::pdu_t pdu = to_pdu(*in);
  pdsch_processor_validator_impl validator;
  if (!validator.is_valid(pdu)) {
    return NRPHY_ERR_INVALID_PDU;
  }
  resource_grid_impl grid(nof_ports, 14, nof_subc, make_precoder(simd));
  grid.set_all_zero();

  std::unique_ptr<pdsch_processor> proc;
  if (impl == 1) {
    proc = std::make_unique<pdsch_processor_lite_impl>(
        make_segmenter(),
        make_ldpc_encoder(simd),
        std::make_unique<ldpc_rate_matcher_impl>(),
        std::make_unique<pseudo_random_generator_impl>(),
        std::make_unique<modulation_mapper_lut_impl>(),
        std::make_unique<dmrs_pdsch_processor_impl>(std::make_unique<pseudo_random_generator_impl>()));
  } else {
    proc = make_processor(simd);
  }
  notifier_flag notifier;
  proc->process(grid.get_mapper(), notifier, {span<const uint8_t>(tb, in->tb_size_bytes)}, pdu);
  if (!notifier.done) {
    return NRPHY_ERR_DEVICE;
  }
  copy_grid_out(grid_out, grid.get_reader(), nof_ports, nof_subc);
  return NRPHY_OK;
}

// pdsch_encoder::encode: rate-matched, interleaved codeword as UNPACKED bits (one per byte).
int ref_pdsch_encode(unsigned       bg,
                     unsigned       rv,
                     unsigned       qm,
                     unsigned       nref,
                     unsigned       nof_layers,
                     unsigned       nof_ch_symbols,
                     const uint8_t* tb,
                     unsigned       tb_bytes,
                     uint8_t*       codeword_unpacked,
                     int            simd)
{
  std::unique_ptr<pdsch_encoder> enc = make_pdsch_encoder(simd);
  pdsch_encoder::configuration   cfg;
  cfg.base_graph     = (bg == 2) ? ldpc_base_graph_type::BG2 : ldpc_base_graph_type::BG1;
  cfg.rv             = rv;
  cfg.mod            = to_mod(qm);
  cfg.Nref           = nref;
  cfg.nof_layers     = nof_layers;
  cfg.nof_ch_symbols = nof_ch_symbols;
  enc->encode(span<uint8_t>(codeword_unpacked, static_cast<size_t>(nof_ch_symbols) * qm),
              span<const uint8_t>(tb, tb_bytes),
              cfg);
  return NRPHY_OK;
}

// ldpc_segmenter_tx::segment: writes C segments of K bits packed, each at stride_bytes; returns C.
// meta_out receives per segment {rm_length, cw_offset, nof_filler_bits, full_length, nof_crc_bits}.
int ref_ldpc_segment(unsigned       bg,
                     unsigned       rv,
                     unsigned       qm,
                     unsigned       nref,
                     unsigned       nof_layers,
                     unsigned       nof_ch_symbols,
                     const uint8_t* tb,
                     unsigned       tb_bytes,
                     uint8_t*       segments,
                     unsigned       stride_bytes,
                     uint32_t*      meta_out,
                     uint32_t*      lifting_size_out)
{
  std::unique_ptr<ldpc_segmenter_tx>                   seg = make_segmenter();
  static_vector<described_segment, MAX_NOF_SEGMENTS>   segs;
  segmenter_config                                     cfg;
  cfg.base_graph     = (bg == 2) ? ldpc_base_graph_type::BG2 : ldpc_base_graph_type::BG1;
  cfg.rv             = rv;
  cfg.mod            = to_mod(qm);
  cfg.Nref           = nref;
  cfg.nof_layers     = nof_layers;
  cfg.nof_ch_symbols = nof_ch_symbols;
  seg->segment(segs, span<const uint8_t>(tb, tb_bytes), cfg);
  for (unsigned i = 0; i != segs.size(); ++i) {
    const bit_buffer&   data  = segs[i].get_data();
    span<const uint8_t> bytes = data.get_buffer();
    std::memcpy(segments + static_cast<size_t>(i) * stride_bytes, bytes.data(), bytes.size());
    const codeblock_metadata& md = segs[i].get_metadata();
    meta_out[5 * i + 0]          = md.cb_specific.rm_length;
    meta_out[5 * i + 1]          = md.cb_specific.cw_offset;
    meta_out[5 * i + 2]          = md.cb_specific.nof_filler_bits;
    meta_out[5 * i + 3]          = md.cb_specific.full_length;
    meta_out[5 * i + 4]          = md.cb_specific.nof_crc_bits;
    *lifting_size_out            = md.tb_common.lifting_size;
  }
  return static_cast<int>(segs.size());
}

// ldpc_encoder::encode for one codeblock: msg = Kb*Zc bits packed, out = out_bits bits packed.
int ref_ldpc_encode(unsigned bg, unsigned zc, const uint8_t* msg, unsigned out_bits, uint8_t* out, int simd)
{
  std::unique_ptr<ldpc_encoder> enc = make_ldpc_encoder(simd);
  unsigned                      kb  = (bg == 2) ? 10 : 22;
  dynamic_bit_buffer            in_bits(kb * zc);
  std::memcpy(in_bits.get_buffer().data(), msg, in_bits.get_buffer().size());
  dynamic_bit_buffer                     out_bitbuf(out_bits);
  codeblock_metadata::tb_common_metadata cfg;
  cfg.base_graph   = (bg == 2) ? ldpc_base_graph_type::BG2 : ldpc_base_graph_type::BG1;
  cfg.lifting_size = static_cast<ldpc::lifting_size_t>(zc);
  enc->encode(out_bitbuf, in_bits, cfg);
  std::memcpy(out, out_bitbuf.get_buffer().data(), out_bitbuf.get_buffer().size());
  return NRPHY_OK;
}

// ldpc_rate_matcher::rate_match for one codeblock: in = full_length bits packed, out = rm_length bits packed.
int ref_ldpc_rate_match(unsigned       bg,
                        unsigned       zc,
                        unsigned       rv,
                        unsigned       qm,
                        unsigned       nref,
                        unsigned       nof_filler_bits,
                        const uint8_t* in,
                        unsigned       in_bits,
                        uint8_t*       out,
                        unsigned       rm_length)
{
  ldpc_rate_matcher_impl rm;
  dynamic_bit_buffer     in_buf(in_bits);
  std::memcpy(in_buf.get_buffer().data(), in, in_buf.get_buffer().size());
  dynamic_bit_buffer out_buf(rm_length);
  codeblock_metadata md;
  md.tb_common.base_graph        = (bg == 2) ? ldpc_base_graph_type::BG2 : ldpc_base_graph_type::BG1;
  md.tb_common.lifting_size      = static_cast<ldpc::lifting_size_t>(zc);
  md.tb_common.rv                = rv;
  md.tb_common.mod               = to_mod(qm);
  md.tb_common.Nref              = nref;
  md.cb_specific.nof_filler_bits = nof_filler_bits;
  rm.rate_match(out_buf, in_buf, md);
  std::memcpy(out, out_buf.get_buffer().data(), out_buf.get_buffer().size());
  return NRPHY_OK;
}

// crc_calculator::calculate_byte; poly: 16, 0x24A (CRC24A), 0x24B (CRC24B), 0x24C, 11, 6.
unsigned ref_crc(unsigned poly, const uint8_t* data, unsigned nbytes)
{
  crc_generator_poly p = crc_generator_poly::CRC24A;
  switch (poly) {
    case 16:
      p = crc_generator_poly::CRC16;
      break;
    case 0x24B:
      p = crc_generator_poly::CRC24B;
      break;
    case 0x24C:
      p = crc_generator_poly::CRC24C;
      break;
    case 11:
      p = crc_generator_poly::CRC11;
      break;
    case 6:
      p = crc_generator_poly::CRC6;
      break;
    default:
      break;
  }
  crc_calculator_lut_impl crc(p);
  return crc.calculate_byte(span<const uint8_t>(data, nbytes));
}

// pseudo_random_generator: init(c_init), advance(offset), then XOR onto nbits packed bits of `data`.
void ref_prg_apply_xor(unsigned c_init, unsigned offset, uint8_t* data, unsigned nbits)
{
  pseudo_random_generator_impl prg;
  prg.init(c_init);
  prg.advance(offset);
  dynamic_bit_buffer in(nbits);
  std::memcpy(in.get_buffer().data(), data, in.get_buffer().size());
  dynamic_bit_buffer out(nbits);
  prg.apply_xor(out, in);
  std::memcpy(data, out.get_buffer().data(), out.get_buffer().size());
}

// pseudo_random_generator::apply_xor(span<log_likelihood_ratio>, span<const log_likelihood_ratio>) after init + advance.
void ref_prg_apply_xor_llr(unsigned c_init, unsigned offset, const int8_t* in, int8_t* out, unsigned n)
{
  pseudo_random_generator_impl prg;
  prg.init(c_init);
  prg.advance(offset);
  static_assert(sizeof(log_likelihood_ratio) == 1, "one byte per soft bit");
  prg.apply_xor(span<log_likelihood_ratio>(reinterpret_cast<log_likelihood_ratio*>(out), n),
                span<const log_likelihood_ratio>(reinterpret_cast<const log_likelihood_ratio*>(in), n));
}

// pseudo_random_generator::generate(span<float>, value) after init + advance.
void ref_prg_generate_float(unsigned c_init, unsigned offset, float value, float* out, unsigned n)
{
  pseudo_random_generator_impl prg;
  prg.init(c_init);
  prg.advance(offset);
  prg.generate(span<float>(out, n), value);
}

// modulation_mapper::modulate(span<ci8_t>, ...): returns the scaling, writes nsym (re, im) int8 pairs.
float ref_modulate_ci8(unsigned qm, const uint8_t* bits, unsigned nsym, int8_t* out)
{
  modulation_mapper_lut_impl mapper;
  dynamic_bit_buffer         in(nsym * qm);
  std::memcpy(in.get_buffer().data(), bits, in.get_buffer().size());
  std::vector<ci8_t> symbols(nsym);
  float              scaling = mapper.modulate(span<ci8_t>(symbols), in, to_mod(qm));
  for (unsigned i = 0; i != nsym; ++i) {
    out[2 * i]     = symbols[i].real();
    out[2 * i + 1] = symbols[i].imag();
  }
  return scaling;
}

// dft_processor_generic_impl: n complex floats in, n out.
int ref_dft(unsigned n, int inverse, const float* in, float* out)
{
  dft_processor::configuration cfg;
  cfg.size = n;
  cfg.dir  = inverse ? dft_processor::direction::INVERSE : dft_processor::direction::DIRECT;
  dft_processor_generic_impl dft(cfg);
  if (!dft.is_valid()) {
    return NRPHY_ERR_ARGUMENT;
  }
  std::memcpy(dft.get_input().data(), in, sizeof(cf_t) * n);
  span<const cf_t> o = dft.run();
  std::memcpy(out, o.data(), sizeof(cf_t) * n);
  return NRPHY_OK;
}

// ofdm_slot_modulator::modulate of one grid ([nof_ports][14][12*bw_rb] cbf16 raw) for every port.
// iq_out: [nof_ports][slot_size] complex float.  Returns the slot size in samples, or < 0.
int ref_ofdm_modulate_slot(const nrphy_ofdm_config_t* c,
                           const uint16_t*            grid_in,
                           unsigned                   nof_ports,
                           unsigned                   slot_index,
                           float*                     iq_out)
{
  unsigned           nof_subc = c->bw_rb * 12;
  resource_grid_impl grid(nof_ports, 14, nof_subc, make_precoder(0));
  grid.set_all_zero();
  for (unsigned p = 0; p != nof_ports; ++p) {
    for (unsigned l = 0; l != 14; ++l) {
      std::vector<cbf16_t> row(nof_subc);
      std::memcpy(row.data(), grid_in + 2 * (static_cast<size_t>(p * 14 + l) * nof_subc), nof_subc * sizeof(cbf16_t));
      grid.get_writer().put(p, l, 0, 1, span<const cbf16_t>(row));
    }
  }
  dft_processor::configuration dft_cfg;
  dft_cfg.size = c->dft_size;
  dft_cfg.dir  = dft_processor::direction::INVERSE;
  ofdm_modulator_common_configuration common;
  common.dft = std::make_unique<dft_processor_generic_impl>(dft_cfg);
  ofdm_modulator_configuration cfg;
  cfg.numerology     = c->numerology;
  cfg.bw_rb          = c->bw_rb;
  cfg.dft_size       = c->dft_size;
  cfg.cp             = c->cp ? cyclic_prefix::EXTENDED : cyclic_prefix::NORMAL;
  cfg.scale          = c->scale;
  cfg.center_freq_hz = c->center_freq_hz;
  ofdm_slot_modulator_impl mod(common, cfg);
  unsigned                 slot_size = mod.get_slot_size(slot_index);
  for (unsigned p = 0; p != nof_ports; ++p) {
    mod.modulate(span<cf_t>(reinterpret_cast<cf_t*>(iq_out) + static_cast<size_t>(p) * slot_size, slot_size),
                 grid.get_reader(),
                 p,
                 slot_index);
  }
  return static_cast<int>(slot_size);
}

// ldpc_decoder::decode (generic or AVX2 implementation) of one codeblock: llr = nof_llr soft bits (the codeblock without
// its first 2*Zc bits), message_bits = Kb*Zc hard bits one per byte.  crc_poly_id: 0 none, 16 CRC16, 0x24A, 0x24B.
// Returns the iteration count when the CRC passed, 0 otherwise.
int ref_ldpc_decode(uint32_t      bg,
                    uint32_t      zc,
                    uint32_t      nof_filler,
                    uint32_t      crc_poly_id,
                    uint32_t      max_iterations,
                    float         scaling_factor,
                    const int8_t* llr,
                    uint32_t      nof_llr,
                    uint8_t*      message_bits,
                    int           simd)
{
  std::unique_ptr<ldpc_decoder> dec;
  if (simd) {
    dec = std::make_unique<ldpc_decoder_avx2>();
  } else {
    dec = std::make_unique<ldpc_decoder_generic>();
  }
  ldpc_decoder::configuration cfg;
  cfg.block_conf.tb_common.base_graph       = (bg == 1) ? ldpc_base_graph_type::BG1 : ldpc_base_graph_type::BG2;
  cfg.block_conf.tb_common.lifting_size     = static_cast<ldpc::lifting_size_t>(zc);
  cfg.block_conf.cb_specific.nof_filler_bits = nof_filler;
  cfg.block_conf.cb_specific.nof_crc_bits   = (crc_poly_id == 16) ? 16 : 24;
  cfg.algorithm_conf.max_iterations         = max_iterations;
  cfg.algorithm_conf.scaling_factor         = scaling_factor;
  std::unique_ptr<crc_calculator> crc;
  if (crc_poly_id != 0) {
    crc = std::make_unique<crc_calculator_lut_impl>(crc_poly_id == 16      ? crc_generator_poly::CRC16
                                                    : crc_poly_id == 0x24B ? crc_generator_poly::CRC24B
                                                                           : crc_generator_poly::CRC24A);
  }
  unsigned                          K = ((bg == 1) ? 22 : 10) * zc;
  std::vector<log_likelihood_ratio> in(nof_llr);
  for (unsigned i = 0; i != nof_llr; ++i) {
    in[i] = log_likelihood_ratio(llr[i]);
  }
  dynamic_bit_buffer      out(K);
  std::optional<unsigned> r = dec->decode(out, in, crc.get(), cfg);
  for (unsigned i = 0; i != K; ++i) {
    message_bits[i] = out.extract(i, 1);
  }
  return r.has_value() ? static_cast<int>(r.value()) : 0;
}

// ldpc_rate_dematcher::rate_dematch (generic or AVX2 implementation) of one codeblock.  out: the soft buffer of
// block_length = N - 2*Zc LLRs, read and written (HARQ combining unless new_data); in: E rate-matched LLRs.
int ref_ldpc_rate_dematch(uint32_t      bg,
                          uint32_t      zc,
                          uint32_t      rv,
                          uint32_t      qm,
                          uint32_t      nref,
                          uint32_t      nof_filler,
                          int           new_data,
                          const int8_t* in,
                          uint32_t      e,
                          int8_t*       out,
                          int           simd)
{
  std::unique_ptr<ldpc_rate_dematcher> dm;
  if (simd) {
    dm = std::make_unique<ldpc_rate_dematcher_avx2_impl>();
  } else {
    dm = std::make_unique<ldpc_rate_dematcher_impl>();
  }
  codeblock_metadata cfg         = {};
  cfg.tb_common.base_graph       = (bg == 1) ? ldpc_base_graph_type::BG1 : ldpc_base_graph_type::BG2;
  cfg.tb_common.lifting_size     = static_cast<ldpc::lifting_size_t>(zc);
  cfg.tb_common.rv               = rv;
  cfg.tb_common.mod              = to_mod(qm);
  cfg.tb_common.Nref             = nref;
  cfg.cb_specific.nof_filler_bits = nof_filler;
  unsigned                          block_length = ((bg == 1) ? 66 : 50) * zc;
  std::vector<log_likelihood_ratio> vin(e), vout(block_length);
  for (unsigned i = 0; i != e; ++i) {
    vin[i] = log_likelihood_ratio(in[i]);
  }
  for (unsigned i = 0; i != block_length; ++i) {
    vout[i] = log_likelihood_ratio(out[i]);
  }
  dm->rate_dematch(vout, vin, new_data != 0, cfg);
  for (unsigned i = 0; i != block_length; ++i) {
    out[i] = static_cast<int8_t>(vout[i].to_int());
  }
  return 0;
}

// pusch_decoder_impl (generic rate dematcher and LDPC decoder underneath) on one transport block, with the HARQ state
// kept in a reference rx_buffer_pool between calls: harq_id names the soft buffer (0..7); new_data starts it afresh.
// llr: the nof_llr = G codeword soft bits.  tb_out: tb_size_bytes.  result[4]: tb_crc_ok, decoder observations
// (codeblocks decoded in this call), sum and maximum of their iteration counts.  Returns 0, or < 0 when no buffer.
namespace {
struct pusch_notifier : pusch_decoder_notifier {
  pusch_decoder_result result;
  bool                 done = false;
  void                 on_sch_data(const pusch_decoder_result& r) override
  {
    result = r;
    done   = true;
  }
};
std::unique_ptr<rx_buffer_pool_controller>& pusch_pool()
{
  static std::unique_ptr<rx_buffer_pool_controller> pool;
  if (!pool) {
    rx_buffer_pool_config cfg;
    cfg.max_codeblock_size   = ldpc::MAX_CODEBLOCK_SIZE;
    cfg.nof_buffers          = 8;
    cfg.nof_codeblocks       = 8 * 170;
    cfg.expire_timeout_slots = 100000;
    cfg.external_soft_bits   = false;
    pool                     = create_rx_buffer_pool(cfg);
  }
  return pool;
}
} // namespace

int ref_pusch_decode(uint32_t      bg,
                     uint32_t      qm,
                     uint32_t      rv,
                     uint32_t      nof_layers,
                     uint32_t      nref,
                     uint32_t      tb_size_bytes,
                     uint32_t      max_iterations,
                     int           use_early_stop,
                     int           new_data,
                     uint32_t      harq_id,
                     uint32_t      nof_codeblocks,
                     const int8_t* llr,
                     uint32_t      nof_llr,
                     uint8_t*      tb_out,
                     uint32_t*     result)
{
  auto make_crcs = [](auto& set) {
    set.crc16  = std::make_unique<crc_calculator_lut_impl>(crc_generator_poly::CRC16);
    set.crc24A = std::make_unique<crc_calculator_lut_impl>(crc_generator_poly::CRC24A);
    set.crc24B = std::make_unique<crc_calculator_lut_impl>(crc_generator_poly::CRC24B);
  };
  pusch_codeblock_decoder::sch_crc cb_crcs;
  make_crcs(cb_crcs);
  std::vector<std::unique_ptr<pusch_codeblock_decoder>> instances;
  instances.push_back(std::make_unique<pusch_codeblock_decoder>(
      std::make_unique<ldpc_rate_dematcher_impl>(), std::make_unique<ldpc_decoder_generic>(), cb_crcs));
  auto decoder_pool = std::make_shared<pusch_decoder_impl::codeblock_decoder_pool>(std::move(instances));
  pusch_decoder_impl::sch_crc tb_crcs;
  make_crcs(tb_crcs);
  pusch_decoder_impl decoder(
      ldpc_segmenter_impl::create_ldpc_segmenter_impl_rx(), decoder_pool, std::move(tb_crcs), nullptr, 275, 4);

  unique_rx_buffer buffer = pusch_pool()->get_pool().reserve(
      slot_point(1, 0), trx_buffer_identifier(0x4601, static_cast<uint8_t>(harq_id)), nof_codeblocks, new_data != 0);
  if (!buffer.is_valid()) {
    return -1;
  }
  pusch_decoder::configuration cfg;
  cfg.base_graph          = (bg == 1) ? ldpc_base_graph_type::BG1 : ldpc_base_graph_type::BG2;
  cfg.rv                  = rv;
  cfg.mod                 = to_mod(qm);
  cfg.Nref                = nref;
  cfg.nof_layers          = nof_layers;
  cfg.nof_ldpc_iterations = max_iterations;
  cfg.use_early_stop      = use_early_stop != 0;
  cfg.new_data            = new_data != 0;
  std::vector<uint8_t>              tb(tb_size_bytes);
  std::vector<log_likelihood_ratio> soft(nof_llr);
  for (unsigned i = 0; i != nof_llr; ++i) {
    soft[i] = log_likelihood_ratio(llr[i]);
  }
  pusch_notifier        notifier;
  pusch_decoder_buffer& in = decoder.new_data(tb, std::move(buffer), notifier, cfg);
  in.on_new_softbits(soft);
  in.on_end_softbits();
  if (!notifier.done) {
    return -2;
  }
  std::memcpy(tb_out, tb.data(), tb_size_bytes);
  const auto& st = notifier.result.ldpc_decoder_stats;
  result[0]      = notifier.result.tb_crc_ok ? 1 : 0;
  result[1]      = static_cast<uint32_t>(st.get_nof_observations());
  result[2]      = static_cast<uint32_t>(std::lround(st.get_mean() * st.get_nof_observations()));
  result[3]      = st.get_nof_observations() ? st.get_max() : 0;
  return 0;
}

// nzp_csi_rs_generator_impl::map into a grid that already holds `grid_io` ([nof_ports][14][nof_subc] cbf16 raw): what the
// signal overwrites and what it leaves alone are both visible.  simd: 0 generic precoder, 1 AVX2.
int ref_csi_rs_map(const nrphy_csi_rs_cfg_t* c, uint16_t* grid_io, unsigned nof_ports, unsigned nof_subc, int simd)
{
  resource_grid_impl grid(nof_ports, 14, nof_subc, make_precoder(simd));
  grid.set_all_zero();
  for (unsigned p = 0; p != nof_ports; ++p) {
    for (unsigned l = 0; l != 14; ++l) {
      const cbf16_t* row = reinterpret_cast<const cbf16_t*>(grid_io) + (static_cast<size_t>(p) * 14 + l) * nof_subc;
      grid.get_writer().put(p, l, 0, 1, span<const cbf16_t>(row, nof_subc));
    }
  }
  nzp_csi_rs_generator::config_t cfg;
  cfg.slot                     = slot_point(4, 0, c->slot_index);
  cfg.cp                       = c->cp ? cyclic_prefix::EXTENDED : cyclic_prefix::NORMAL;
  cfg.start_rb                 = c->start_rb;
  cfg.nof_rb                   = c->nof_rb;
  cfg.csi_rs_mapping_table_row = c->row;
  for (unsigned i = 0; i != c->nof_k_ref; ++i) {
    cfg.freq_allocation_ref_idx.push_back(c->k_ref[i]);
  }
  cfg.symbol_l0     = c->symbol_l0;
  cfg.symbol_l1     = c->symbol_l1;
  cfg.cdm           = static_cast<csi_rs_cdm_type>(c->cdm);
  cfg.freq_density  = static_cast<csi_rs_freq_density_type>(c->density);
  cfg.scrambling_id = c->scrambling_id;
  cfg.amplitude     = c->amplitude;
  cfg.precoding     = precoding_configuration(c->nof_ports, c->nof_ports, c->nof_prg, c->prg_size_rb);
  for (unsigned g = 0; g != c->nof_prg; ++g) {
    for (unsigned p = 0; p != c->nof_ports; ++p) {
      for (unsigned l = 0; l != c->nof_ports; ++l) {
        const float* w = c->precoding + 2 * ((g * c->nof_ports + p) * c->nof_ports + l);
        cfg.precoding.set_coefficient(cf_t(w[0], w[1]), l, p, g);
      }
    }
  }
  nzp_csi_rs_generator_impl generator(std::make_unique<pseudo_random_generator_impl>());
  generator.map(grid.get_mapper(), cfg);
  copy_grid_out(grid_io, grid.get_reader(), nof_ports, nof_subc);
  return NRPHY_OK;
}

// ofdm_slot_demodulator::demodulate of every port of one slot: iq_in [nof_ports][slot_size] complex float ->
// grid_out [nof_ports][14][12*bw_rb] cbf16 raw.  Returns the slot size in samples, or < 0.
int ref_ofdm_demodulate_slot(const nrphy_ofdm_config_t* c,
                             const float*               iq_in,
                             unsigned                   nof_ports,
                             unsigned                   slot_index,
                             unsigned                   window_offset,
                             uint16_t*                  grid_out)
{
  unsigned           nof_subc = c->bw_rb * 12;
  resource_grid_impl grid(nof_ports, 14, nof_subc, make_precoder(0));
  grid.set_all_zero();
  dft_processor::configuration dft_cfg;
  dft_cfg.size = c->dft_size;
  dft_cfg.dir  = dft_processor::direction::DIRECT;
  ofdm_demodulator_common_configuration common;
  common.dft = std::make_unique<dft_processor_generic_impl>(dft_cfg);
  ofdm_demodulator_configuration cfg;
  cfg.numerology                = c->numerology;
  cfg.bw_rb                     = c->bw_rb;
  cfg.dft_size                  = c->dft_size;
  cfg.cp                        = c->cp ? cyclic_prefix::EXTENDED : cyclic_prefix::NORMAL;
  cfg.nof_samples_window_offset = window_offset;
  cfg.scale                     = c->scale;
  cfg.center_freq_hz            = c->center_freq_hz;
  ofdm_slot_demodulator_impl demod(common, cfg);
  unsigned                   slot_size = demod.get_slot_size(slot_index);
  for (unsigned p = 0; p != nof_ports; ++p) {
    demod.demodulate(grid.get_writer(),
                     span<const cf_t>(reinterpret_cast<const cf_t*>(iq_in) + static_cast<size_t>(p) * slot_size, slot_size),
                     p,
                     slot_index);
  }
  copy_grid_out(grid_out, grid.get_reader(), nof_ports, nof_subc);
  return static_cast<int>(slot_size);
}

// CPU baseline: `threads` workers, each owning a processor (+ optional OFDM modulator) instance and
// running `reps` PDUs back to back, the scheme of the reference benchmark
// (tests/benchmarks/phy/upper/channel_processors/pdsch_processor_benchmark.cpp:684-737).
// Returns elapsed seconds for threads*reps slots.
double ref_bench_pdsch(const nrphy_pdsch_pdu_t*   in,
                       const uint8_t*             tb,
                       unsigned                   nof_ports,
                       unsigned                   nof_subc,
                       const nrphy_ofdm_config_t* ofdm, /* may be null: PDSCH only */
                       unsigned                   threads,
                       unsigned                   reps,
                       int                        simd)
{
  pdsch_processor::pdu_t   pdu = to_pdu(*in);
  std::vector<std::thread> pool;
  auto                     t0 = std::chrono::steady_clock::now();
  for (unsigned t = 0; t != threads; ++t) {
    pool.emplace_back([&, t]() {
      (void)t;
      resource_grid_impl               grid(nof_ports, 14, nof_subc, make_precoder(simd));
      std::unique_ptr<pdsch_processor> proc = make_processor(simd);
      std::unique_ptr<ofdm_slot_modulator_impl> mod;
      std::vector<cf_t>                         iq;
      if (ofdm != nullptr) {
        dft_processor::configuration dft_cfg;
        dft_cfg.size = ofdm->dft_size;
        dft_cfg.dir  = dft_processor::direction::INVERSE;
        ofdm_modulator_common_configuration common;
        common.dft = std::make_unique<dft_processor_generic_impl>(dft_cfg);
        ofdm_modulator_configuration cfg;
        cfg.numerology     = ofdm->numerology;
        cfg.bw_rb          = ofdm->bw_rb;
        cfg.dft_size       = ofdm->dft_size;
        cfg.cp             = cyclic_prefix::NORMAL;
        cfg.scale          = ofdm->scale;
        cfg.center_freq_hz = ofdm->center_freq_hz;
        mod                = std::make_unique<ofdm_slot_modulator_impl>(common, cfg);
        iq.resize(mod->get_slot_size(0));
      }
      for (unsigned r = 0; r != reps; ++r) {
        grid.set_all_zero();
        notifier_flag notifier;
        proc->process(grid.get_mapper(), notifier, {span<const uint8_t>(tb, in->tb_size_bytes)}, pdu);
        if (mod) {
          for (unsigned p = 0; p != nof_ports; ++p) {
            mod->modulate(iq, grid.get_reader(), p, 0);
          }
        }
      }
    });
  }
  for (auto& th : pool) {
    th.join();
  }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// CPU baseline of the receive-side coding chain (BASELINE config 5): `threads` workers, each with its own
// pusch_decoder_impl -- rate dematcher + LDPC decoder (simd 1: ldpc_rate_dematcher_avx2_impl + ldpc_decoder_avx2, the
// reference's defaults on this host; 0: the generic ones) + codeblock and transport-block CRCs -- and its own receive-buffer
// pool, decoding the same codeword LLRs `reps` times as new data: the threads x batch scheme of the reference's benchmarks
// (tests/benchmarks/phy/upper/channel_coding/ldpc/ldpc_decoder_benchmark.cpp:36-182 times ldpc_decoder::decode alone;
// pdsch_processor_benchmark.cpp:684-737 is the threads x batch harness).  Returns the elapsed seconds for threads * reps
// transport blocks; *nof_ok receives how many of them passed their CRC.
double ref_bench_pusch_decode(uint32_t      bg,
                              uint32_t      qm,
                              uint32_t      rv,
                              uint32_t      nof_layers,
                              uint32_t      nref,
                              uint32_t      tb_size_bytes,
                              uint32_t      max_iterations,
                              int           use_early_stop,
                              uint32_t      nof_codeblocks,
                              const int8_t* llr,
                              uint32_t      nof_llr,
                              unsigned      threads,
                              unsigned      reps,
                              int           simd,
                              uint32_t*     nof_ok)
{
  std::vector<log_likelihood_ratio> soft(nof_llr);
  for (unsigned i = 0; i != nof_llr; ++i) {
    soft[i] = log_likelihood_ratio(llr[i]);
  }
  pusch_decoder::configuration cfg;
  cfg.base_graph          = (bg == 1) ? ldpc_base_graph_type::BG1 : ldpc_base_graph_type::BG2;
  cfg.rv                  = rv;
  cfg.mod                 = to_mod(qm);
  cfg.Nref                = nref;
  cfg.nof_layers          = nof_layers;
  cfg.nof_ldpc_iterations = max_iterations;
  cfg.use_early_stop      = use_early_stop != 0;
  cfg.new_data            = true;
  std::vector<uint32_t>    ok(threads, 0);
  std::vector<std::thread> pool;
  auto                     t0 = std::chrono::steady_clock::now();
  for (unsigned t = 0; t != threads; ++t) {
    pool.emplace_back([&, t]() {
      auto make_crcs = [](auto& set) {
        set.crc16  = std::make_unique<crc_calculator_lut_impl>(crc_generator_poly::CRC16);
        set.crc24A = std::make_unique<crc_calculator_lut_impl>(crc_generator_poly::CRC24A);
        set.crc24B = std::make_unique<crc_calculator_lut_impl>(crc_generator_poly::CRC24B);
      };
      pusch_codeblock_decoder::sch_crc cb_crcs;
      make_crcs(cb_crcs);
      std::unique_ptr<ldpc_rate_dematcher> dematcher;
      std::unique_ptr<ldpc_decoder>        ldpc;
      if (simd) {
        dematcher = std::make_unique<ldpc_rate_dematcher_avx2_impl>();
        ldpc      = std::make_unique<ldpc_decoder_avx2>();
      } else {
        dematcher = std::make_unique<ldpc_rate_dematcher_impl>();
        ldpc      = std::make_unique<ldpc_decoder_generic>();
      }
      std::vector<std::unique_ptr<pusch_codeblock_decoder>> instances;
      instances.push_back(std::make_unique<pusch_codeblock_decoder>(std::move(dematcher), std::move(ldpc), cb_crcs));
      auto decoder_pool = std::make_shared<pusch_decoder_impl::codeblock_decoder_pool>(std::move(instances));
      pusch_decoder_impl::sch_crc tb_crcs;
      make_crcs(tb_crcs);
      pusch_decoder_impl decoder(
          ldpc_segmenter_impl::create_ldpc_segmenter_impl_rx(), decoder_pool, std::move(tb_crcs), nullptr, 275, 4);
      rx_buffer_pool_config pool_cfg;
      pool_cfg.max_codeblock_size   = ldpc::MAX_CODEBLOCK_SIZE;
      pool_cfg.nof_buffers          = 2;
      pool_cfg.nof_codeblocks       = 2 * 170;
      pool_cfg.expire_timeout_slots = 100000;
      pool_cfg.external_soft_bits   = false;
      std::unique_ptr<rx_buffer_pool_controller> buffers = create_rx_buffer_pool(pool_cfg);
      std::vector<uint8_t>                       tb(tb_size_bytes);
      for (unsigned r = 0; r != reps; ++r) {
        unique_rx_buffer buffer = buffers->get_pool().reserve(
            slot_point(1, 0), trx_buffer_identifier(static_cast<uint16_t>(0x4601 + t), 0), nof_codeblocks, true);
        if (!buffer.is_valid()) {
          return;
        }
        pusch_notifier        notifier;
        pusch_decoder_buffer& in = decoder.new_data(tb, std::move(buffer), notifier, cfg);
        in.on_new_softbits(soft);
        in.on_end_softbits();
        ok[t] += (notifier.done && notifier.result.tb_crc_ok) ? 1u : 0u;
      }
    });
  }
  for (auto& th : pool) {
    th.join();
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (nof_ok != nullptr) {
    *nof_ok = 0;
    for (uint32_t v : ok) {
      *nof_ok += v;
    }
  }
  return dt;
}

} // extern "C"
