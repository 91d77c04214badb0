// Test infrastructure (build container only): the CONFIGURATIONS of the reference's own unit-test vectors, read by compiling
// against the reference's test-data headers where they lie under /root/reference/srsRAN-5G-ER/tests/unittests/phy, and
// handed to tests/golden/generate.py as plain numbers.  The .dat files those headers name are not in the reference
// checkout, so nothing here calls file_vector::read(); generate.py runs the compiled reference on seeded payloads for
// every configuration and commits configuration + output as a fixture.
//
// Every header defines its own `test_case_t`, so this file is compiled once per header with -DWHICH=n (oracle/Makefile):
//   1 upper/channel_processors/pdsch_processor_test_data.h     (24 cases)
//   2 upper/channel_processors/pdsch_encoder_test_data.h       (168 cases)
//   3 upper/channel_processors/pdsch_modulator_test_data.h     (36 cases)
//   4 upper/channel_coding/ldpc/ldpc_segmenter_test_data.h     (11 cases)
//   5 lower/modulation/ofdm_modulator_test_data.h              (20 cases)
//   6 upper/signal_processors/dmrs_pdsch_processor_test_data.h (192 cases)
// and, for the "next" rows built in round 2 (section 8f of SURVEY.md):
//   7 upper/channel_processors/pdcch_processor_test_data.h     (114 cases)
//   8 upper/channel_processors/ssb_processor_test_data.h       (240 cases)
//   9 upper/signal_processors/nzp_csi_rs_generator_test_data.h (102 cases)
//  10 upper/channel_modulation/demodulation_mapper_test_data.h (12 cases)
//  11 lower/modulation/ofdm_demodulator_test_data.h            (20 cases)
//  12 ofh/compression/ofh_compression_test_data.h              (36 cases; under tests/unittests/ofh)
//  13 upper/channel_processors/pdcch_encoder_test_data.h       (29 cases)
//  14 upper/channel_processors/pbch_encoder_test_data.h        (232 cases)
// Each export returns the number of cases when called with a null output.

#include "mi355_nrphy.h"
#include <cmath>
// One name per header for the struct they all call test_case_t: the objects are linked into one library, and
// std::vector<srsran::test_case_t> instantiated over different layouts under one symbol name would be merged.
#define TESTDATA_PASTE2(a, b) a##b
#define TESTDATA_PASTE(a, b) TESTDATA_PASTE2(a, b)
#define test_case_t TESTDATA_PASTE(test_case_of_header_, WHICH)
#include <cstring>
#include <vector>

#if WHICH == 1 || WHICH == 3 || WHICH == 6 || WHICH == 7 || WHICH == 8 || WHICH == 9
#include "mi355_nrphy_srsran.h"
#endif
#if WHICH == 1 || WHICH == 3 || WHICH == 6


namespace {
int emit_pdu(const srsran::pdsch_processor::pdu_t& pdu, nrphy_pdsch_pdu_t* pod, float* weights, unsigned weights_cap)
{
  std::vector<float> w;
  *pod = mi355::to_pod(pdu, 0, w);
  if (w.size() > weights_cap) {
    return -1;
  }
  std::memcpy(weights, w.data(), w.size() * sizeof(float));
  pod->precoding = nullptr;
  return static_cast<int>(w.size());
}
} // namespace
#endif

#if WHICH == 1
#include "upper/channel_processors/pdsch_processor_test_data.h"

// rg[0..1] = {rg_nof_rb, rg_nof_symb}; returns the number of weight floats written.
extern "C" int
ref_testdata_pdsch_processor(unsigned i, nrphy_pdsch_pdu_t* pod, float* weights, unsigned weights_cap, unsigned* rg)
{
  using namespace srsran;
  if (pod == nullptr) {
    return static_cast<int>(pdsch_processor_test_data.size());
  }
  const test_case_context& c = pdsch_processor_test_data[i].context;
  rg[0]                      = c.rg_nof_rb;
  rg[1]                      = c.rg_nof_symb;
  return emit_pdu(c.pdu, pod, weights, weights_cap);
}
#endif

#if WHICH == 2
#include "upper/channel_processors/pdsch_encoder_test_data.h"

// out = {base graph (1|2), rv, bits per symbol, Nref, nof_layers, nof_ch_symbols}.
extern "C" int ref_testdata_pdsch_encoder(unsigned i, unsigned* out)
{
  using namespace srsran;
  if (out == nullptr) {
    return static_cast<int>(pdsch_encoder_test_data.size());
  }
  const segmenter_config& c = pdsch_encoder_test_data[i].config;
  out[0]                    = (c.base_graph == ldpc_base_graph_type::BG1) ? 1 : 2;
  out[1]                    = c.rv;
  out[2]                    = get_bits_per_symbol(c.mod);
  out[3]                    = c.Nref;
  out[4]                    = c.nof_layers;
  out[5]                    = c.nof_ch_symbols;
  return 0;
}
#endif

#if WHICH == 3
#include "upper/channel_processors/pdsch_modulator_test_data.h"

// The modulator's configuration as the PDSCH PDU that makes pdsch_processor_impl build exactly this config_t
// (pdsch_processor_impl.cpp:139-158): the fields the modulator does not see take fixed values (slot 0, CRB0, DM-RS
// scrambling identity 0, rv 0, BG1, no power offset for the DM-RS).
extern "C" int ref_testdata_pdsch_modulator(unsigned i, nrphy_pdsch_pdu_t* pod, float* weights, unsigned weights_cap)
{
  using namespace srsran;
  if (pod == nullptr) {
    return static_cast<int>(pdsch_modulator_test_data.size());
  }
  const pdsch_modulator::config_t& c = pdsch_modulator_test_data[i].config;
  pdsch_processor::pdu_t           pdu;
  pdu.context      = std::nullopt;
  pdu.slot         = slot_point(0, 0);
  pdu.rnti         = c.rnti;
  pdu.bwp_size_rb  = c.bwp_size_rb;
  pdu.bwp_start_rb = c.bwp_start_rb;
  pdu.cp           = cyclic_prefix::NORMAL;
  pdu.codewords.push_back({c.modulation1, 0});
  pdu.n_id                        = c.n_id;
  pdu.ref_point                   = pdsch_processor::pdu_t::CRB0;
  pdu.dmrs_symbol_mask            = c.dmrs_symb_pos;
  pdu.dmrs                        = c.dmrs_config_type;
  pdu.scrambling_id               = 0;
  pdu.n_scid                      = false;
  pdu.nof_cdm_groups_without_data = c.nof_cdm_groups_without_data;
  pdu.freq_alloc                  = c.freq_allocation;
  pdu.start_symbol_index          = c.start_symbol_index;
  pdu.nof_symbols                 = c.nof_symbols;
  pdu.ldpc_base_graph             = ldpc_base_graph_type::BG1;
  pdu.tbs_lbrm                    = units::bytes(159749); // the value every pdsch_processor_test_data.h case carries
  pdu.reserved                    = c.reserved;
  pdu.ratio_pdsch_dmrs_to_sss_dB  = 0.0F;
  pdu.ratio_pdsch_data_to_sss_dB  = (c.scaling == 1.0F) ? 0.0F : -20.0F * std::log10(c.scaling);
  pdu.precoding                   = c.precoding;
  return emit_pdu(pdu, pod, weights, weights_cap);
}
#endif

#if WHICH == 4
#include "upper/channel_coding/ldpc/ldpc_segmenter_test_data.h"

// out = {transport block size in bits, base graph, number of segments, segment length}: the header's known answers.
extern "C" int ref_testdata_ldpc_segmenter(unsigned i, unsigned* out)
{
  using namespace srsran;
  if (out == nullptr) {
    return static_cast<int>(ldpc_segmenter_test_data.size());
  }
  const test_case_t& c = ldpc_segmenter_test_data[i];
  out[0]               = c.tbs;
  out[1]               = c.bg;
  out[2]               = c.nof_segments;
  out[3]               = c.segment_length;
  return 0;
}
#endif

#if WHICH == 5
#include "support/resource_grid_test_doubles.h"
#include "lower/modulation/ofdm_modulator_test_data.h"

// extra = {port_idx, slot_idx}.
extern "C" int ref_testdata_ofdm_modulator(unsigned i, nrphy_ofdm_config_t* cfg, unsigned* extra)
{
  using namespace srsran;
  if (cfg == nullptr) {
    return static_cast<int>(ofdm_modulator_test_data.size());
  }
  const ofdm_modulator_test_configuration& c = ofdm_modulator_test_data[i].test_config;
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->numerology     = c.config.numerology;
  cfg->bw_rb          = c.config.bw_rb;
  cfg->dft_size       = c.config.dft_size;
  cfg->cp             = (c.config.cp == cyclic_prefix::NORMAL) ? 0 : 1;
  cfg->scale          = c.config.scale;
  cfg->center_freq_hz = c.config.center_freq_hz;
  extra[0]            = c.port_idx;
  extra[1]            = c.slot_idx;
  return 0;
}
#endif

#if WHICH == 6
#include "upper/signal_processors/dmrs_pdsch_processor_test_data.h"

namespace {
// The power offset whose amplitude (convert_dB_to_amplitude(-ratio), pdsch_processor_impl.cpp:177) is exactly `amplitude`.
bool ratio_for_amplitude(float amplitude, float& ratio_dB)
{
  float r = -20.0F * std::log10(amplitude);
  float lo = r, hi = r;
  for (unsigned n = 0; n != 256; ++n) {
    if (srsran::convert_dB_to_amplitude(-lo) == amplitude) {
      ratio_dB = lo;
      return true;
    }
    if (srsran::convert_dB_to_amplitude(-hi) == amplitude) {
      ratio_dB = hi;
      return true;
    }
    lo = std::nextafter(lo, -1e9F);
    hi = std::nextafter(hi, 1e9F);
  }
  return false;
}
} // namespace

// The DM-RS configuration as the PDSCH PDU that makes pdsch_processor_impl build exactly this config_t
// (pdsch_processor_impl.cpp:166-183): allocation = the configuration's RB mask over a bandwidth part that starts at CRB 0
// and spans the mask, all 14 symbols, QPSK, two CDM groups without data.  info = {dmrs type (1|2), slot numerology,
// amplitude representable as a power offset (0|1), reference_point_k_rb}.  Type-2 cases are returned as well: the PDSCH processor's validator refuses them
// (pdsch_processor_validator_impl.cpp) and so does nrphy_pdsch_validate.
extern "C" int
ref_testdata_dmrs_pdsch(unsigned i, nrphy_pdsch_pdu_t* pod, float* weights, unsigned weights_cap, unsigned* info)
{
  using namespace srsran;
  if (pod == nullptr) {
    return static_cast<int>(dmrs_pdsch_processor_test_data.size());
  }
  const dmrs_pdsch_processor::config_t& c = dmrs_pdsch_processor_test_data[i].config;
  pdsch_processor::pdu_t                pdu;
  pdu.context      = std::nullopt;
  pdu.slot         = c.slot;
  pdu.rnti         = 1;
  pdu.bwp_size_rb  = c.rb_mask.size();
  pdu.bwp_start_rb = c.reference_point_k_rb;
  pdu.cp           = cyclic_prefix::NORMAL;
  pdu.codewords.push_back({modulation_scheme::QPSK, 0});
  pdu.n_id             = 0;
  pdu.ref_point        = (c.reference_point_k_rb == 0) ? pdsch_processor::pdu_t::CRB0 : pdsch_processor::pdu_t::PRB0;
  pdu.dmrs_symbol_mask = c.symbols_mask;
  pdu.dmrs             = c.type;
  pdu.scrambling_id    = c.scrambling_id;
  pdu.n_scid           = c.n_scid;
  pdu.nof_cdm_groups_without_data = 2;
  pdu.freq_alloc                  = rb_allocation::make_type0(c.rb_mask);
  pdu.start_symbol_index          = 0;
  pdu.nof_symbols                 = 14;
  pdu.ldpc_base_graph             = ldpc_base_graph_type::BG2;
  pdu.tbs_lbrm                    = units::bytes(159749); // the value every pdsch_processor_test_data.h case carries
  float ratio                     = 0.0F;
  info[2]                         = ratio_for_amplitude(c.amplitude, ratio) ? 1 : 0;
  pdu.ratio_pdsch_dmrs_to_sss_dB  = ratio;
  pdu.ratio_pdsch_data_to_sss_dB  = 0.0F;
  pdu.precoding                   = c.precoding;
  info[0]                         = (c.type == dmrs_type::TYPE1) ? 1 : 2;
  info[1]                         = c.slot.numerology();
  info[3]                         = c.reference_point_k_rb;
  return emit_pdu(pdu, pod, weights, weights_cap);
}

// Runs dmrs_pdsch_processor_impl::map on case i into a grid that already holds `grid_io` ([nof_ports][14][nof_subc] cbf16
// raw): the caller pre-fills it with a marker to find the written positions.
#include "lib/phy/support/resource_grid_impl.h"
#include "lib/phy/upper/sequence_generators/pseudo_random_generator_impl.h"
#include "lib/phy/upper/signal_processors/dmrs_pdsch_processor_impl.h"
std::unique_ptr<srsran::channel_precoder> ref_make_precoder(int simd);

extern "C" int ref_testdata_dmrs_pdsch_map(unsigned i, uint16_t* grid_io, unsigned nof_ports, unsigned nof_subc, int simd)
{
  using namespace srsran;
  const dmrs_pdsch_processor::config_t& c = dmrs_pdsch_processor_test_data[i].config;
  if (c.precoding.get_nof_ports() != nof_ports || c.rb_mask.size() * NRE > nof_subc) {
    return NRPHY_ERR_ARGUMENT;
  }
  resource_grid_impl grid(nof_ports, 14, nof_subc, ref_make_precoder(simd));
  grid.set_all_zero();
  for (unsigned p = 0; p != nof_ports; ++p) {
    for (unsigned l = 0; l != 14; ++l) {
      const cbf16_t* row = reinterpret_cast<const cbf16_t*>(grid_io) + (static_cast<size_t>(p) * 14 + l) * nof_subc;
      grid.get_writer().put(p, l, 0, 1, span<const cbf16_t>(row, nof_subc));
    }
  }
  dmrs_pdsch_processor_impl dmrs(std::make_unique<pseudo_random_generator_impl>());
  dmrs.map(grid.get_mapper(), c);
  for (unsigned p = 0; p != nof_ports; ++p) {
    for (unsigned l = 0; l != 14; ++l) {
      span<const cbf16_t> view = grid.get_reader().get_view(p, l);
      std::memcpy(grid_io + 2 * (static_cast<size_t>(p * 14 + l) * nof_subc), view.data(), nof_subc * sizeof(cbf16_t));
    }
  }
  return NRPHY_OK;
}
#endif

#if WHICH == 7
#include "upper/channel_processors/pdcch_processor_test_data.h"

// Returns the number of weight floats written (> 0).
extern "C" int ref_testdata_pdcch_processor(unsigned i, nrphy_pdcch_pdu_t* pod, float* weights, unsigned weights_cap)
{
  using namespace srsran;
  if (pod == nullptr) {
    return static_cast<int>(pdcch_processor_test_data.size());
  }
  std::vector<float> w;
  *pod = mi355::to_pod(pdcch_processor_test_data[i].config, w);
  if (w.size() > weights_cap) {
    return -1;
  }
  std::memcpy(weights, w.data(), w.size() * sizeof(float));
  pod->precoding = nullptr;
  return static_cast<int>(w.size());
}
#endif

#if WHICH == 8
#include "upper/channel_processors/ssb_processor_test_data.h"

extern "C" int ref_testdata_ssb_processor(unsigned i, nrphy_ssb_pdu_t* pod)
{
  using namespace srsran;
  if (pod == nullptr) {
    return static_cast<int>(ssb_processor_test_data.size());
  }
  *pod = mi355::to_pod(ssb_processor_test_data[i].config);
  return 0;
}
#endif

#if WHICH == 9
#include "upper/signal_processors/nzp_csi_rs_generator_test_data.h"

// Returns the number of weight floats written (> 0).
extern "C" int ref_testdata_nzp_csi_rs(unsigned i, nrphy_csi_rs_cfg_t* pod, float* weights, unsigned weights_cap)
{
  using namespace srsran;
  if (pod == nullptr) {
    return static_cast<int>(nzp_csi_rs_generator_test_data.size());
  }
  std::vector<float> w;
  *pod = mi355::to_pod(nzp_csi_rs_generator_test_data[i].config, w);
  if (w.size() > weights_cap) {
    return -1;
  }
  std::memcpy(weights, w.data(), w.size() * sizeof(float));
  pod->precoding = nullptr;
  return static_cast<int>(w.size());
}
#endif

#if WHICH == 10
#include "srsran/phy/upper/log_likelihood_ratio.h"
#include "srsran/ran/sch/modulation_scheme.h"
#include "upper/channel_modulation/demodulation_mapper_test_data.h"

// out = {number of symbols, modulation as NRPHY_MOD_*}.
extern "C" int ref_testdata_demodulation_mapper(unsigned i, unsigned* out)
{
  using namespace srsran;
  if (out == nullptr) {
    return static_cast<int>(demodulation_mapper_test_data.size());
  }
  const test_case_t& c = demodulation_mapper_test_data[i];
  out[0]               = c.nsymbols;
  out[1]               = c.scheme == modulation_scheme::PI_2_BPSK ? NRPHY_MOD_PI2_BPSK : get_bits_per_symbol(c.scheme);
  return 0;
}
#endif

#if WHICH == 11
#include "support/resource_grid_test_doubles.h"
#include "lower/modulation/ofdm_demodulator_test_data.h"

// extra = {port_idx, slot_idx, nof_samples_window_offset}.
extern "C" int ref_testdata_ofdm_demodulator(unsigned i, nrphy_ofdm_config_t* cfg, unsigned* extra)
{
  using namespace srsran;
  if (cfg == nullptr) {
    return static_cast<int>(ofdm_demodulator_test_data.size());
  }
  const ofdm_demodulator_test_configuration& c = ofdm_demodulator_test_data[i].test_config;
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->numerology     = c.config.numerology;
  cfg->bw_rb          = c.config.bw_rb;
  cfg->dft_size       = c.config.dft_size;
  cfg->cp             = (c.config.cp == cyclic_prefix::NORMAL) ? 0 : 1;
  cfg->scale          = c.config.scale;
  cfg->center_freq_hz = c.config.center_freq_hz;
  extra[0]            = c.port_idx;
  extra[1]            = c.slot_idx;
  extra[2]            = c.config.nof_samples_window_offset;
  return 0;
}
#endif

#if WHICH == 12
#include "srsran/adt/complex.h"
#include "srsran/ofh/compression/compression_params.h"
#include "ofh_compression_test_data.h"

// out = {nof_prb, compression type (0 none, 1 BFP, other values as the reference's enum), data width}; *iq_scaling.
extern "C" int ref_testdata_ofh_compression(unsigned i, unsigned* out, float* iq_scaling)
{
  using namespace srsran;
  if (out == nullptr) {
    return static_cast<int>(ofh_compression_test_data.size());
  }
  const test_case_t& c = ofh_compression_test_data[i];
  out[0]               = c.nof_prb;
  out[1]               = c.type == ofh::compression_type::none ? 0 : (c.type == ofh::compression_type::BFP ? 1 : 2);
  out[2]               = c.cIQ_width;
  *iq_scaling          = c.iq_scaling;
  return 0;
}
#endif

#if WHICH == 13
#include "srsran/phy/upper/channel_processors/pdcch_encoder.h"
#include "upper/channel_processors/pdcch_encoder_test_data.h"

// out = {E, rnti}.
extern "C" int ref_testdata_pdcch_encoder(unsigned i, unsigned* out)
{
  using namespace srsran;
  if (out == nullptr) {
    return static_cast<int>(pdcch_encoder_test_data.size());
  }
  out[0] = pdcch_encoder_test_data[i].config.E;
  out[1] = pdcch_encoder_test_data[i].config.rnti;
  return 0;
}
#endif

#if WHICH == 14
#include "srsran/phy/upper/channel_processors/pbch_encoder.h"
#include "upper/channel_processors/pbch_encoder_test_data.h"

// The PBCH message as the SS/PBCH block PDU whose processor builds exactly this pbch_msg_t (ssb_processor_impl.cpp:46-53):
// 15 kHz slot 0 or 5 for the half-frame bit, the other fields verbatim; the fields the encoder does not see stay zero.
extern "C" int ref_testdata_pbch_encoder(unsigned i, nrphy_ssb_pdu_t* pod)
{
  using namespace srsran;
  if (pod == nullptr) {
    return static_cast<int>(pbch_encoder_test_data.size());
  }
  const pbch_encoder::pbch_msg_t& m = pbch_encoder_test_data[i].pbch_msg;
  std::memset(pod, 0, sizeof(*pod));
  pod->numerology        = 0;
  pod->sfn               = m.sfn;
  pod->slot_index        = m.hrf ? 5 : 0;
  pod->phys_cell_id      = m.N_id;
  pod->ssb_idx           = m.ssb_idx;
  pod->L_max             = m.L_max;
  pod->subcarrier_offset = m.k_ssb.to_uint();
  for (unsigned k = 0; k != m.payload.size() && k != 32; ++k) {
    pod->bch_payload[k] = m.payload[k];
  }
  pod->nof_ports = 1;
  return 0;
}
#endif
