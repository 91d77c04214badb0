// TEST INFRASTRUCTURE -- not part of the product.
//
// C entry points over the REFERENCE's own PDCCH and SS/PBCH block processors, the lower-PHY amplitude controllers and
// the Open Fronthaul IQ compressors, compiled from the sources where they lie under
// /root/reference by oracle/Makefile into oracle/_ref/libsrsref.so.  Original code: it only calls the reference.
#include "mi355_nrphy.h"

#include "lib/phy/generic_functions/precoding/channel_precoder_avx2.h"
#include "lib/phy/generic_functions/precoding/channel_precoder_generic.h"
#include "lib/phy/support/resource_grid_impl.h"
#include "lib/phy/upper/channel_coding/crc_calculator_lut_impl.h"
#include "lib/phy/upper/channel_coding/polar/polar_allocator_impl.h"
#include "lib/phy/upper/channel_coding/polar/polar_code_impl.h"
#include "lib/phy/upper/channel_coding/polar/polar_encoder_impl.h"
#include "lib/phy/upper/channel_coding/polar/polar_interleaver_impl.h"
#include "lib/phy/upper/channel_coding/polar/polar_rate_matcher_impl.h"
#include "lib/phy/upper/channel_modulation/modulation_mapper_lut_impl.h"
#include "lib/phy/upper/channel_processors/pbch_encoder_impl.h"
#include "lib/phy/upper/channel_processors/pbch_modulator_impl.h"
#include "lib/phy/upper/channel_processors/pdcch_encoder_impl.h"
#include "lib/phy/upper/channel_processors/pdcch_modulator_impl.h"
#include "lib/phy/upper/channel_processors/pdcch_processor_impl.h"
#include "lib/phy/upper/channel_processors/ssb_processor_impl.h"
#include "lib/phy/upper/sequence_generators/pseudo_random_generator_impl.h"
#include "lib/phy/upper/signal_processors/dmrs_pbch_processor_impl.h"
#include "lib/phy/upper/signal_processors/dmrs_pdcch_processor_impl.h"
#include "lib/phy/upper/signal_processors/pss_processor_impl.h"
#include "lib/phy/upper/signal_processors/sss_processor_impl.h"
#include "lib/ofh/compression/iq_compression_bfp_avx2.h"
#include "lib/ofh/compression/iq_compression_bfp_impl.h"
#include "lib/ofh/compression/iq_compression_none_avx2.h"
#include "lib/ofh/compression/iq_compression_none_impl.h"
#include "lib/phy/lower/amplitude_controller/amplitude_controller_clipping_impl.h"
#include "lib/phy/lower/amplitude_controller/amplitude_controller_scaling_impl.h"
#include "srsran/srslog/srslog.h"
#include "srsran/srsvec/conversion.h"

#include <cstring>
#include <memory>

using namespace srsran;

namespace {

// The reliability sequence of TS 38.212 Table 5.3.1.2-1 is a private static member of polar_code_impl; an explicit
// template instantiation may name it (the standard exempts those from access checking).
template <const std::array<uint16_t, 1024>* Ptr>
struct mother_code_access {
  friend const std::array<uint16_t, 1024>* get_mother_code_10() { return Ptr; }
};
const std::array<uint16_t, 1024>* get_mother_code_10();
template struct mother_code_access<&polar_code_impl::mother_code_10>;

std::unique_ptr<channel_precoder> make_precoder_dl(int simd)
{
  if (simd) {
    return std::make_unique<channel_precoder_avx2>();
  }
  return std::make_unique<channel_precoder_generic>();
}

void load_grid(resource_grid_impl& grid, const uint16_t* grid_io, unsigned nof_ports, unsigned nof_subc)
{
  grid.set_all_zero();
  for (unsigned p = 0; p != nof_ports; ++p) {
    for (unsigned l = 0; l != 14; ++l) {
      const cbf16_t* row = reinterpret_cast<const cbf16_t*>(grid_io) + (static_cast<size_t>(p) * 14 + l) * nof_subc;
      grid.get_writer().put(p, l, 0, 1, span<const cbf16_t>(row, nof_subc));
    }
  }
}

void store_grid(uint16_t* out, const resource_grid_reader& reader, unsigned nof_ports, unsigned nof_subc)
{
  for (unsigned p = 0; p != nof_ports; ++p) {
    for (unsigned l = 0; l != 14; ++l) {
      span<const cbf16_t> view = reader.get_view(p, l);
      std::memcpy(out + 2 * (static_cast<size_t>(p * 14 + l) * nof_subc), view.data(), nof_subc * sizeof(cbf16_t));
    }
  }
}

std::unique_ptr<pdcch_encoder> make_pdcch_encoder()
{
  return std::make_unique<pdcch_encoder_impl>(std::make_unique<crc_calculator_lut_impl>(crc_generator_poly::CRC24C),
                                              std::make_unique<polar_interleaver_impl>(),
                                              std::make_unique<polar_allocator_impl>(),
                                              std::make_unique<polar_code_impl>(),
                                              std::make_unique<polar_encoder_impl>(),
                                              std::make_unique<polar_rate_matcher_impl>());
}

std::unique_ptr<pbch_encoder> make_pbch_encoder()
{
  return std::make_unique<pbch_encoder_impl>(std::make_unique<crc_calculator_lut_impl>(crc_generator_poly::CRC24C),
                                             std::make_unique<pseudo_random_generator_impl>(),
                                             std::make_unique<polar_interleaver_impl>(),
                                             std::make_unique<polar_allocator_impl>(),
                                             std::make_unique<polar_code_impl>(),
                                             std::make_unique<polar_encoder_impl>(),
                                             std::make_unique<polar_rate_matcher_impl>());
}

pbch_encoder::pbch_msg_t to_pbch_msg(const nrphy_ssb_pdu_t& in)
{
  const slot_point         slot(in.numerology, in.sfn, in.slot_index);
  pbch_encoder::pbch_msg_t msg;
  msg.N_id    = in.phys_cell_id;
  msg.ssb_idx = in.ssb_idx;
  msg.L_max   = in.L_max;
  msg.hrf     = slot.is_odd_hrf();
  for (unsigned i = 0; i != msg.payload.size(); ++i) {
    msg.payload[i] = in.bch_payload[i];
  }
  msg.sfn   = slot.sfn();
  msg.k_ssb = in.subcarrier_offset;
  return msg;
}

} // namespace

// For oracle/ref/adaptor_harness.cpp too: the reference objects behind the PODs.
pdcch_processor::pdu_t ref_make_pdcch_pdu(const nrphy_pdcch_pdu_t& in_)
{
  const nrphy_pdcch_pdu_t* in = &in_;
  pdcch_processor::pdu_t pdu;
  pdu.context                    = std::nullopt;
  pdu.slot                       = slot_point(4, 0, in->slot_index);
  pdu.cp                         = in->cp ? cyclic_prefix::EXTENDED : cyclic_prefix::NORMAL;
  pdu.coreset.bwp_size_rb        = in->bwp_size_rb;
  pdu.coreset.bwp_start_rb       = in->bwp_start_rb;
  pdu.coreset.start_symbol_index = in->start_symbol_index;
  pdu.coreset.duration           = in->duration;
  pdu.coreset.frequency_resources.resize(pdcch_constants::MAX_NOF_FREQ_RESOURCES);
  for (unsigned i = 0; i != pdcch_constants::MAX_NOF_FREQ_RESOURCES; ++i) {
    pdu.coreset.frequency_resources.set(i, (in->frequency_resources >> i) & 1U);
  }
  pdu.coreset.cce_to_reg_mapping = static_cast<pdcch_processor::cce_to_reg_mapping_type>(in->cce_to_reg_mapping);
  pdu.coreset.reg_bundle_size    = in->reg_bundle_size;
  pdu.coreset.interleaver_size   = in->interleaver_size;
  pdu.coreset.shift_index        = in->shift_index;
  pdu.dci.rnti                   = in->rnti;
  pdu.dci.n_id_pdcch_dmrs        = in->n_id_pdcch_dmrs;
  pdu.dci.n_id_pdcch_data        = in->n_id_pdcch_data;
  pdu.dci.n_rnti                 = in->n_rnti;
  pdu.dci.cce_index              = in->cce_index;
  pdu.dci.aggregation_level      = in->aggregation_level;
  pdu.dci.dmrs_power_offset_dB   = in->dmrs_power_offset_dB;
  pdu.dci.data_power_offset_dB   = in->data_power_offset_dB;
  for (unsigned i = 0; i != in->payload_size; ++i) {
    pdu.dci.payload.push_back(in->payload[i]);
  }
  pdu.dci.precoding = precoding_configuration(1, in->nof_ports, in->nof_prg, in->prg_size_rb);
  for (unsigned g = 0; g != in->nof_prg; ++g) {
    for (unsigned p = 0; p != in->nof_ports; ++p) {
      const float* w = in->precoding + 2 * (g * in->nof_ports + p);
      pdu.dci.precoding.set_coefficient(cf_t(w[0], w[1]), 0, p, g);
    }
  }
  return pdu;
}

ssb_processor::pdu_t ref_make_ssb_pdu(const nrphy_ssb_pdu_t& in_)
{
  const nrphy_ssb_pdu_t* in = &in_;
  ssb_processor::pdu_t pdu;
  pdu.slot              = slot_point(in->numerology, in->sfn, in->slot_index);
  pdu.phys_cell_id      = static_cast<pci_t>(in->phys_cell_id);
  pdu.beta_pss          = in->beta_pss_dB;
  pdu.ssb_idx           = in->ssb_idx;
  pdu.L_max             = in->L_max;
  pdu.common_scs        = to_subcarrier_spacing(in->common_scs);
  pdu.subcarrier_offset = in->subcarrier_offset;
  pdu.offset_to_pointA  = in->offset_to_pointA;
  pdu.pattern_case      = static_cast<ssb_pattern_case>(in->pattern_case);
  for (unsigned i = 0; i != 32; ++i) {
    pdu.bch_payload[i] = in->bch_payload[i];
  }
  for (unsigned i = 0; i != in->nof_ports; ++i) {
    pdu.ports.push_back(in->ports[i]);
  }
  return pdu;
}

std::unique_ptr<pdcch_processor> ref_make_pdcch_processor()
{
  return std::make_unique<pdcch_processor_impl>(
      make_pdcch_encoder(),
      std::make_unique<pdcch_modulator_impl>(std::make_unique<modulation_mapper_lut_impl>(), std::make_unique<pseudo_random_generator_impl>()),
      std::make_unique<dmrs_pdcch_processor_impl>(std::make_unique<pseudo_random_generator_impl>()));
}

std::unique_ptr<ssb_processor> ref_make_ssb_processor()
{
  ssb_processor_config cfg;
  cfg.encoder   = make_pbch_encoder();
  cfg.modulator = std::make_unique<pbch_modulator_impl>(std::make_unique<modulation_mapper_lut_impl>(),
                                                        std::make_unique<pseudo_random_generator_impl>());
  cfg.dmrs      = std::make_unique<dmrs_pbch_processor_impl>(std::make_unique<pseudo_random_generator_impl>());
  cfg.pss       = std::make_unique<pss_processor_impl>();
  cfg.sss       = std::make_unique<sss_processor_impl>();
  return std::make_unique<ssb_processor_impl>(std::move(cfg));
}

extern "C" {

// TS 38.212 Table 5.3.1.2-1 (polar sequence, 1024 entries in ascending reliability) and Table 5.3.1.1-1 (interleaving
// pattern, 164 entries) as the reference holds them; read by oracle/gen_polar_tables.py.
int ref_polar_tables(uint16_t* reliability, uint8_t* interleaver_pattern)
{
  const std::array<uint16_t, 1024>* mc = get_mother_code_10();
  std::memcpy(reliability, mc->data(), sizeof(uint16_t) * 1024);
  // The interleaver's pattern through its public interface: a full-length (K = 164) identity input comes out permuted.
  std::array<uint8_t, 164> in, out;
  for (unsigned i = 0; i != 164; ++i) {
    in[i] = static_cast<uint8_t>(i);
  }
  polar_interleaver_impl il;
  il.interleave(out, in, polar_interleaver_direction::tx);
  std::memcpy(interleaver_pattern, out.data(), 164);
  return NRPHY_OK;
}

// polar_code::set: returns N; k_set_mask gets N bytes (1 = information or parity-check position).
int ref_polar_code(unsigned K, unsigned E, unsigned n_max, uint8_t* k_set_mask)
{
  polar_code_impl code;
  code.set(K, E, static_cast<uint8_t>(n_max), polar_code_ibil::not_present);
  const auto& mask = code.get_K_set();
  for (unsigned i = 0; i != code.get_N(); ++i) {
    k_set_mask[i] = mask.test(i) ? 1 : 0;
  }
  return static_cast<int>(code.get_N());
}

// pdcch_encoder::encode: payload bits (one per byte) -> E rate-matched bits (one per byte).
int ref_pdcch_encode(const uint8_t* payload, unsigned payload_size, unsigned rnti, unsigned E, uint8_t* encoded)
{
  std::unique_ptr<pdcch_encoder> enc = make_pdcch_encoder();
  pdcch_encoder::config_t        cfg;
  cfg.E    = E;
  cfg.rnti = rnti;
  enc->encode(span<uint8_t>(encoded, E), span<const uint8_t>(payload, payload_size), cfg);
  return NRPHY_OK;
}

// pdcch_processor::process into a copy of the caller's grid [nof_ports][14][nof_subc] cbf16 (raw), read and written.
int ref_pdcch_process(const nrphy_pdcch_pdu_t* in, uint16_t* grid_io, unsigned nof_ports, unsigned nof_subc, int simd)
{
  resource_grid_impl grid(nof_ports, 14, nof_subc, make_precoder_dl(simd));
  load_grid(grid, grid_io, nof_ports, nof_subc);
  pdcch_processor::pdu_t pdu = ref_make_pdcch_pdu(*in);
  pdcch_processor_impl proc(make_pdcch_encoder(),
                            std::make_unique<pdcch_modulator_impl>(std::make_unique<modulation_mapper_lut_impl>(),
                                                                   std::make_unique<pseudo_random_generator_impl>()),
                            std::make_unique<dmrs_pdcch_processor_impl>(std::make_unique<pseudo_random_generator_impl>()));
  proc.process(grid.get_mapper(), pdu);
  store_grid(grid_io, grid.get_reader(), nof_ports, nof_subc);
  return NRPHY_OK;
}

// pbch_encoder::encode: the 864 rate-matched bits (one per byte).
int ref_pbch_encode(const nrphy_ssb_pdu_t* in, uint8_t* encoded)
{
  std::unique_ptr<pbch_encoder> enc = make_pbch_encoder();
  enc->encode(span<uint8_t>(encoded, pbch_encoder::E), to_pbch_msg(*in));
  return NRPHY_OK;
}

// ssb_processor::process into a copy of the caller's grid, read and written.
int ref_ssb_process(const nrphy_ssb_pdu_t* in, uint16_t* grid_io, unsigned nof_ports, unsigned nof_subc)
{
  resource_grid_impl grid(nof_ports, 14, nof_subc, make_precoder_dl(0));
  load_grid(grid, grid_io, nof_ports, nof_subc);
  ssb_processor::pdu_t pdu = ref_make_ssb_pdu(*in);
  ssb_processor_config cfg;
  cfg.encoder   = make_pbch_encoder();
  cfg.modulator = std::make_unique<pbch_modulator_impl>(std::make_unique<modulation_mapper_lut_impl>(),
                                                        std::make_unique<pseudo_random_generator_impl>());
  cfg.dmrs      = std::make_unique<dmrs_pbch_processor_impl>(std::make_unique<pseudo_random_generator_impl>());
  cfg.pss       = std::make_unique<pss_processor_impl>();
  cfg.sss       = std::make_unique<sss_processor_impl>();
  ssb_processor_impl proc(std::move(cfg));
  proc.process(grid.get_writer(), pdu);
  store_grid(grid_io, grid.get_reader(), nof_ports, nof_subc);
  return NRPHY_OK;
}

// ---- lower-PHY tail ---------------------------------------------------------------------------
// amplitude_controller::process on one buffer of nof_samples complex floats.  kind 0: clipping implementation
// (gain, measurements, optional clipping), 1: scaling implementation (gain only).  metrics_out: avg_power_fs,
// peak_power_fs, papr_lin, gain_dB; counters_out: nof_processed_samples, nof_clipped_samples.
int ref_amplitude_control(int          kind,
                          int          enable_clipping,
                          float        input_gain_dB,
                          float        full_scale_lin,
                          float        ceiling_dBFS,
                          const float* in,
                          unsigned     nof_samples,
                          float*       out,
                          float*       metrics_out,
                          uint64_t*    counters_out)
{
  span<const cf_t> x(reinterpret_cast<const cf_t*>(in), nof_samples);
  span<cf_t>       y(reinterpret_cast<cf_t*>(out), nof_samples);
  amplitude_controller_metrics m;
  if (kind == 0) {
    amplitude_controller_clipping_impl ctl(enable_clipping != 0, input_gain_dB, full_scale_lin, ceiling_dBFS);
    m = ctl.process(y, x);
  } else {
    amplitude_controller_scaling_impl ctl(input_gain_dB);
    m = ctl.process(y, x);
  }
  metrics_out[0]  = m.avg_power_fs;
  metrics_out[1]  = m.peak_power_fs;
  metrics_out[2]  = m.papr_lin;
  metrics_out[3]  = m.gain_dB;
  counters_out[0] = m.nof_processed_samples;
  counters_out[1] = m.nof_clipped_samples;
  return NRPHY_OK;
}

// srsvec::convert(span<const cf_t>, float scale, span<int16_t>): what the radio layer does with the baseband buffers.
int ref_convert_cf_to_ci16(const float* in, unsigned nof_samples, float scale, int16_t* out)
{
  srsvec::convert(span<const cf_t>(reinterpret_cast<const cf_t*>(in), nof_samples), scale, span<int16_t>(out, 2 * nof_samples));
  return NRPHY_OK;
}

// iq_compressor::compress of nof_prb PRBs (cbf16, raw) into the serialised form of the Open Fronthaul user plane:
// per PRB [udCompParam (BFP only)] + packed IQ (ofh_uplane_message_builder_impl.cpp:137-144).  type 0 none, 1 BFP;
// simd selects the AVX2 compressor.  Returns the number of bytes written.
int ref_ofh_compress(int type, int simd, unsigned data_width, float iq_scaling, const uint16_t* grid_prbs, unsigned nof_prb, uint8_t* out)
{
  static srslog::basic_logger& logger = srslog::fetch_basic_logger("OFH_REF");
  std::unique_ptr<ofh::iq_compressor> comp;
  if (type == 0) {
    if (simd) {
      comp = std::make_unique<ofh::iq_compression_none_avx2>(logger, iq_scaling);
    } else {
      comp = std::make_unique<ofh::iq_compression_none_impl>(logger, iq_scaling);
    }
  } else if (simd) {
    comp = std::make_unique<ofh::iq_compression_bfp_avx2>(logger, iq_scaling);
  } else {
    comp = std::make_unique<ofh::iq_compression_bfp_impl>(logger, iq_scaling);
  }
  std::vector<ofh::compressed_prb> prbs(nof_prb);
  ofh::ru_compression_params       params;
  params.type       = type == 0 ? ofh::compression_type::none : ofh::compression_type::BFP;
  params.data_width = data_width;
  comp->compress(prbs, span<const cbf16_t>(reinterpret_cast<const cbf16_t*>(grid_prbs), 12 * nof_prb), params);
  unsigned n = 0;
  for (const ofh::compressed_prb& c : prbs) {
    if (type != 0) {
      out[n++] = c.get_compression_param();
    }
    span<const uint8_t> d = c.get_packed_data();
    std::memcpy(out + n, d.data(), d.size());
    n += d.size();
  }
  return static_cast<int>(n);
}

} // extern "C"

// demodulation_mapper_impl::demodulate_soft (AVX2 + generic paths as compiled here).  modulation = NRPHY_MOD_*.
#include "lib/phy/upper/channel_modulation/demodulation_mapper_impl.h"
extern "C" int ref_demodulate_soft(uint32_t modulation, size_t n, const float* symbols, const float* noise_vars, int8_t* llr)
{
  using namespace srsran;
  modulation_scheme mod;
  switch (modulation) {
    case 0:
      mod = modulation_scheme::PI_2_BPSK;
      break;
    case 1:
      mod = modulation_scheme::BPSK;
      break;
    case 2:
      mod = modulation_scheme::QPSK;
      break;
    case 4:
      mod = modulation_scheme::QAM16;
      break;
    case 6:
      mod = modulation_scheme::QAM64;
      break;
    case 8:
      mod = modulation_scheme::QAM256;
      break;
    default:
      return NRPHY_ERR_ARGUMENT;
  }
  demodulation_mapper_impl mapper;
  mapper.demodulate_soft(span<log_likelihood_ratio>(reinterpret_cast<log_likelihood_ratio*>(llr), n * get_bits_per_symbol(mod)),
                         span<const cf_t>(reinterpret_cast<const cf_t*>(symbols), n),
                         span<const float>(noise_vars, n),
                         mod);
  return NRPHY_OK;
}
