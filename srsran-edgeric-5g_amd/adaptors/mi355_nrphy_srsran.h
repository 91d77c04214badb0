// Header-only srsRAN-side adaptors: the reference's plugin interfaces implemented on top of the C ABI of
// include/mi355_nrphy.h.  A gNB maintainer drops this header into the srsRAN tree, links libmi355nrphy.so and hands
// the factories below to upper_phy / lower_phy where the software ones are created today.
//
//   mi355::pdsch_processor_adaptor / pdsch_processor_factory_adaptor
//        -> srsran::pdsch_processor, pdsch_pdu_validator, pdsch_processor_factory
//           (include/srsran/phy/upper/channel_processors/pdsch_processor.h:50-184, channel_processor_factories.h:153-160)
//   mi355::hw_accelerator_pdsch_enc_adaptor / _factory
//        -> srsran::hal::hw_accelerator_pdsch_enc (include/srsran/hal/phy/upper/channel_processors/hw_accelerator_pdsch_enc.h:75-99)
//   mi355::ofdm_symbol_modulator_adaptor / ofdm_slot_modulator_adaptor / ofdm_modulator_factory_adaptor
//        -> srsran::ofdm_symbol_modulator, ofdm_slot_modulator, ofdm_modulator_factory
//           (include/srsran/phy/lower/modulation/ofdm_modulator.h:54-101, modulation_factories.h:34-51)
//   mi355::dft_processor_adaptor / dft_processor_factory_adaptor
//        -> srsran::dft_processor (include/srsran/phy/generic_functions/dft_processor.h:34-73)
//
// These are the host-span (drop-in) forms: every call copies through PCIe.  Throughput comes from the batched
// device-pointer entry points (nrphy_pdsch_run / nrphy_ofdm_run) once the caller keeps grids on the device.
#pragma once

#include "mi355_nrphy.h"

#include "srsran/hal/phy/upper/channel_processors/hw_accelerator_pdsch_enc.h"
#include "srsran/hal/phy/upper/channel_processors/hw_accelerator_pdsch_enc_factory.h"
#include "srsran/hal/phy/upper/channel_processors/pusch/hw_accelerator_pusch_dec.h"
#include "srsran/hal/phy/upper/channel_processors/pusch/hw_accelerator_pusch_dec_factory.h"
#include "srsran/phy/generic_functions/dft_processor.h"
#include "srsran/phy/generic_functions/generic_functions_factories.h"
#include "srsran/phy/lower/modulation/modulation_factories.h"
#include "srsran/phy/lower/modulation/ofdm_demodulator.h"
#include "srsran/phy/lower/modulation/ofdm_modulator.h"
#include "srsran/gateways/baseband/buffer/baseband_gateway_buffer_writer.h"
#include "srsran/phy/lower/processors/downlink/pdxch/pdxch_processor.h"
#include "srsran/phy/lower/processors/downlink/pdxch/pdxch_processor_baseband.h"
#include "srsran/phy/lower/processors/downlink/pdxch/pdxch_processor_factories.h"
#include "srsran/phy/lower/processors/downlink/pdxch/pdxch_processor_notifier.h"
#include "srsran/phy/lower/processors/downlink/pdxch/pdxch_processor_request_handler.h"
#include "srsran/phy/support/resource_grid_context.h"
#include "srsran/phy/support/resource_grid_mapper.h"
#include "srsran/phy/upper/channel_coding/channel_coding_factories.h"
#include "srsran/phy/support/resource_grid_reader.h"
#include "srsran/phy/support/resource_grid_writer.h"
#include "srsran/phy/upper/channel_processors/channel_processor_factories.h"
#include "srsran/phy/upper/channel_processors/pdsch_processor.h"
#include "srsran/phy/upper/signal_processors/nzp_csi_rs_generator.h"
#include "srsran/ran/csi_rs/csi_rs_pattern.h"
#include "srsran/srsvec/bit.h"

#include "srsran/ofh/compression/iq_compressor.h"
#include "srsran/phy/lower/amplitude_controller/amplitude_controller.h"
#include "srsran/phy/lower/amplitude_controller/amplitude_controller_factories.h"
#include "srsran/phy/support/re_buffer.h"
#include "srsran/phy/support/resource_grid.h"
#include "srsran/phy/support/support_factories.h"
#include "srsran/phy/upper/channel_processors/pdcch_processor.h"
#include "srsran/phy/upper/channel_processors/ssb_processor.h"
#include "srsran/ran/precoding/precoding_codebooks.h"
#include "srsran/fapi/messages.h"
#include "srsran/fapi_adaptor/precoding_matrix_repository.h"
#include "srsran/phy/upper/channel_modulation/channel_modulation_factories.h"
#include "srsran/phy/upper/channel_modulation/demodulation_mapper.h"
#include "srsran/phy/upper/vrb_to_prb_mapper.h"
#include "srsran/ran/sch/sch_dmrs_power.h"

#include <array>
#include <atomic>
#include <chrono>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

namespace mi355 {

/// Failure of a library call behind an interface that cannot return one: always logged, never compiled out.
inline void report_failure(const char* what, int rc)
{
  if (rc != NRPHY_OK) {
    fmt::print(stderr, "mi355: {} failed: {}\n", what, nrphy_strerror(rc));
  }
}

/// Shared ownership of one nrphy context per process/device.
class context
{
public:
  explicit context(int device_id = 0)
  {
    int rc = nrphy_create(&ctx, device_id);
    report_failure("nrphy_create", rc);
  }
  ~context() { nrphy_destroy(ctx); }
  context(const context&)            = delete;
  context& operator=(const context&) = delete;
  nrphy_ctx_t* get() const { return ctx; }

private:
  nrphy_ctx_t* ctx = nullptr;
};

/// pdu_t -> POD.  \c weights receives the precoding coefficients the POD points to.
inline nrphy_pdsch_pdu_t to_pod(const srsran::pdsch_processor::pdu_t& pdu, size_t tb_size, std::vector<float>& weights)
{
  using namespace srsran;
  nrphy_pdsch_pdu_t p;
  std::memset(&p, 0, sizeof(p));
  p.slot_index                  = pdu.slot.slot_index();
  p.rnti                        = pdu.rnti;
  p.bwp_start_rb                = pdu.bwp_start_rb;
  p.bwp_size_rb                 = pdu.bwp_size_rb;
  p.cp                          = (pdu.cp == cyclic_prefix::NORMAL) ? 0 : 1;
  p.nof_codewords               = pdu.codewords.size();
  p.qm                          = pdu.codewords.empty() ? 0 : get_bits_per_symbol(pdu.codewords[0].modulation);
  p.rv                          = pdu.codewords.empty() ? 0 : pdu.codewords[0].rv;
  p.n_id                        = pdu.n_id;
  p.ref_point                   = (pdu.ref_point == pdsch_processor::pdu_t::PRB0) ? 1 : 0;
  p.dmrs_symbol_mask            = 0;
  for (unsigned l = 0; l != pdu.dmrs_symbol_mask.size(); ++l) {
    p.dmrs_symbol_mask |= pdu.dmrs_symbol_mask.test(l) ? (1U << l) : 0U;
  }
  p.dmrs_type                   = (pdu.dmrs == dmrs_type::TYPE1) ? 1 : 2;
  p.scrambling_id               = pdu.scrambling_id;
  p.n_scid                      = pdu.n_scid ? 1 : 0;
  p.nof_cdm_groups_without_data = pdu.nof_cdm_groups_without_data;
  p.start_symbol_index          = pdu.start_symbol_index;
  p.nof_symbols                 = pdu.nof_symbols;
  p.ldpc_base_graph             = (pdu.ldpc_base_graph == ldpc_base_graph_type::BG1) ? 1 : 2;
  p.tbs_lbrm_bytes              = pdu.tbs_lbrm.value();
  p.vrb_contiguous              = pdu.freq_alloc.is_contiguous() ? 1 : 0;
  bounded_bitset<MAX_RB> prb    = pdu.freq_alloc.get_prb_mask(pdu.bwp_start_rb, pdu.bwp_size_rb);
  for (unsigned i = 0; i != prb.size(); ++i) {
    if (prb.test(i)) {
      p.prb_mask[i / 64] |= uint64_t(1) << (i % 64);
    }
  }
  p.nof_reserved = 0;
  for (const re_pattern& pat : pdu.reserved.get_re_patterns()) {
    nrphy_re_pattern_t& o = p.reserved[p.nof_reserved++];
    for (unsigned i = 0; i != pat.prb_mask.size(); ++i) {
      if (pat.prb_mask.test(i)) {
        o.prb_mask[i / 64] |= uint64_t(1) << (i % 64);
      }
    }
    for (unsigned k = 0; k != NRE; ++k) {
      o.re_mask |= pat.re_mask.test(k) ? (1U << k) : 0U;
    }
    for (unsigned l = 0; l != pat.symbols.size(); ++l) {
      o.symbol_mask |= pat.symbols.test(l) ? (1U << l) : 0U;
    }
  }
  p.tb_size_bytes              = tb_size;
  p.ratio_pdsch_dmrs_to_sss_dB = pdu.ratio_pdsch_dmrs_to_sss_dB;
  p.ratio_pdsch_data_to_sss_dB = pdu.ratio_pdsch_data_to_sss_dB;
  p.nof_layers                 = pdu.precoding.get_nof_layers();
  p.nof_ports                  = pdu.precoding.get_nof_ports();
  p.prg_size_rb                = pdu.precoding.get_prg_size();
  p.nof_prg                    = pdu.precoding.get_nof_prg();
  weights.resize(2 * p.nof_prg * p.nof_ports * p.nof_layers);
  for (unsigned g = 0; g != p.nof_prg; ++g) {
    for (unsigned port = 0; port != p.nof_ports; ++port) {
      for (unsigned l = 0; l != p.nof_layers; ++l) {
        cf_t w                                                    = pdu.precoding.get_coefficient(l, port, g);
        weights[2 * ((g * p.nof_ports + port) * p.nof_layers + l)]     = w.real();
        weights[2 * ((g * p.nof_ports + port) * p.nof_layers + l) + 1] = w.imag();
      }
    }
  }
  p.precoding = weights.data();
  return p;
}

/// pdsch_pdu_validator over nrphy_pdsch_validate.
class pdsch_pdu_validator_adaptor : public srsran::pdsch_pdu_validator
{
public:
  bool is_valid(const srsran::pdsch_processor::pdu_t& pdu) const override
  {
    std::vector<float> weights;
    nrphy_pdsch_pdu_t  pod = to_pod(pdu, 1, weights);
    return nrphy_pdsch_validate(&pod) == NRPHY_OK;
  }
};

// ---- from a staging grid into the caller's grid ----------------------------------------------------------------------
// srsRAN 24.04 hands channel processors a resource_grid_mapper, not the grid.  Two ways in, neither a patch to the
// reference: (1) the gNB creates its grids through resource_grid_factory_adaptor below (a factory swap where upper_phy
// builds its grid pool): the mapper of such a grid also gives its writer, and the adaptors put cbf16 words straight
// into the grid -- bit-exact; (2) with any other grid the adaptors go through resource_grid_mapper::map with an
// identity precoding, one call per run of PRBs that share a RE mask -- numerically identical, a -0.0 may come out
// as +0.0 (the precoder adds the zero contributions of the other ports).

/// A mapper that forwards to the grid's own and also gives the grid's writer.
class resource_grid_mapper_with_writer : public srsran::resource_grid_mapper
{
public:
  resource_grid_mapper_with_writer(srsran::resource_grid_mapper& inner_, srsran::resource_grid_writer& writer_) :
    inner(inner_), writer(writer_)
  {
  }
  void map(const srsran::re_buffer_reader<srsran::cf_t>& input,
           const srsran::re_pattern&                     pattern,
           const srsran::precoding_configuration&        precoding) override
  {
    inner.map(input, pattern, precoding);
  }
  void map(symbol_buffer&                         buffer,
           const srsran::re_pattern_list&         pattern,
           const srsran::re_pattern_list&         reserved,
           const srsran::precoding_configuration& precoding,
           unsigned                               re_skip = 0) override
  {
    inner.map(buffer, pattern, reserved, precoding, re_skip);
  }
  srsran::resource_grid_writer& get_writer() { return writer; }

private:
  srsran::resource_grid_mapper& inner;
  srsran::resource_grid_writer& writer;
};

/// A resource grid of the reference behind a mapper that gives its writer.
class resource_grid_adaptor : public srsran::resource_grid
{
public:
  explicit resource_grid_adaptor(std::unique_ptr<srsran::resource_grid> inner_) :
    inner(std::move(inner_)), mapper(inner->get_mapper(), inner->get_writer())
  {
  }
  void                                set_all_zero() override { inner->set_all_zero(); }
  srsran::resource_grid_writer&       get_writer() override { return inner->get_writer(); }
  const srsran::resource_grid_reader& get_reader() const override { return inner->get_reader(); }
  srsran::resource_grid_mapper&       get_mapper() override { return mapper; }

private:
  std::unique_ptr<srsran::resource_grid> inner;
  resource_grid_mapper_with_writer       mapper;
};

class resource_grid_factory_adaptor : public srsran::resource_grid_factory
{
public:
  explicit resource_grid_factory_adaptor(std::shared_ptr<srsran::resource_grid_factory> inner_) : inner(std::move(inner_)) {}
  std::unique_ptr<srsran::resource_grid> create(unsigned nof_ports, unsigned nof_symbols, unsigned nof_subc) override
  {
    return std::make_unique<resource_grid_adaptor>(inner->create(nof_ports, nof_symbols, nof_subc));
  }

private:
  std::shared_ptr<srsran::resource_grid_factory> inner;
};

// (the device-mirrored grid of the downlink slot pipeline: see "the downlink slot pipeline" further down for the lower-PHY half)
/// nrphy_dl_slots_t shared by the grid factory (upper PHY) and the pdxch factory (lower PHY) of one cell.
class dl_slot_pool
{
public:
  dl_slot_pool(std::shared_ptr<context> ctx_, const nrphy_ofdm_config_t& ofdm, unsigned nof_ports_, unsigned depth, unsigned max_tb_bytes) :
    ctx(std::move(ctx_)), cfg(ofdm), nof_ports(nof_ports_), nof_subc(ofdm.bw_rb * srsran::NRE)
  {
    nrphy_dl_slots_cfg_t c;
    std::memset(&c, 0, sizeof(c));
    c.ofdm         = ofdm;
    c.nof_ports    = nof_ports;
    c.depth        = depth;
    c.max_tb_bytes = max_tb_bytes;
    c.iq_format    = 0; // baseband_gateway_buffer_writer carries cf_t
    int rc         = nrphy_dl_slots_create(ctx->get(), &c, &pool);
    report_failure("nrphy_dl_slots_create", rc);
    srsran_assert(rc == NRPHY_OK, "nrphy_dl_slots_create failed.");
  }
  ~dl_slot_pool() { nrphy_dl_slots_destroy(pool); }
  dl_slot_pool(const dl_slot_pool&)            = delete;
  dl_slot_pool& operator=(const dl_slot_pool&) = delete;

  nrphy_dl_slots_t*          get() const { return pool; }
  const nrphy_ofdm_config_t& ofdm_config() const { return cfg; }
  unsigned                   get_nof_ports() const { return nof_ports; }
  unsigned                   get_nof_subc() const { return nof_subc; }

private:
  std::shared_ptr<context> ctx;
  nrphy_dl_slots_t*        pool = nullptr;
  nrphy_ofdm_config_t      cfg;
  unsigned                 nof_ports;
  unsigned                 nof_subc;
};

/// The pool slot a device-mirrored grid holds, shared with whoever was handed the grid (the pdxch adaptor keeps it in its request
/// until the slot has been transmitted): it outlives the grid object, so a deferred release is safe whatever the order of teardown.
struct dl_slot_lease {
  explicit dl_slot_lease(std::shared_ptr<dl_slot_pool> pool_) : pool(std::move(pool_)) {}
  ~dl_slot_lease() { release_locked(); }
  /// Gives the slot back.  Caller holds `mutex` (or is the destructor).
  void release_locked()
  {
    if (have_slot) {
      nrphy_dl_slot_close(pool->get(), slot_id);
    }
    have_slot   = false;
    handed_over = false;
  }
  /// `use`: what hand_over() reported -- the grid pool may have handed the grid out again (set_all_zero) since; then a no-op.
  void release(uint64_t use)
  {
    std::lock_guard<std::mutex> lock(mutex);
    if (use == generation) {
      release_locked();
    }
  }
  std::shared_ptr<dl_slot_pool> pool;
  std::mutex                    mutex;               // also guards the owning grid's bookkeeping
  uint32_t                      slot_id     = 0;
  std::atomic<bool>             have_slot{false};    // the device layer exists (a slot is held)
  bool                          handed_over = false; // its modulation has been submitted
  uint64_t                      generation  = 0;     // uses of the grid (set_all_zero starts a new one)
};

/// A resource grid whose accelerated channels live in a slot of the pool (HBM) and whose other channels -- whatever is
/// written through get_writer() or mapped through get_mapper() by processors that are not adaptors of this header -- live
/// in a host grid of the reference.  The two layers meet on the device when the grid is handed over (hand_over(): the host
/// layer's non-zero resource elements are put into the slot's grid, nrphy_dl_slot_put; channels of one slot never share
/// resource elements, so the order does not matter), or on the host if somebody reads the grid (get_reader().get...():
/// one blocking device-to-host copy, then the host layer's elements over it).
class device_resource_grid : public srsran::resource_grid
{
public:
  device_resource_grid(std::shared_ptr<dl_slot_pool> pool_, std::unique_ptr<srsran::resource_grid> host_layer_, std::unique_ptr<srsran::resource_grid> merged_) :
    pool(pool_), lease(std::make_shared<dl_slot_lease>(std::move(pool_))), host_layer(std::move(host_layer_)), merged(std::move(merged_)),
    mapper(*this), writer(*this), reader(*this)
  {
  }
  ~device_resource_grid() override
  {
    std::lock_guard<std::mutex> lock(lease->mutex);
    lease->release_locked();
    ++lease->generation; // (a release still queued somewhere finds nothing to do)
  }

  // resource_grid
  void set_all_zero() override
  {
    std::lock_guard<std::mutex> lock(lease->mutex);
    host_layer->set_all_zero();
    lease->release_locked();
    host_touched = false;
    merged_valid = false;
    ++lease->generation; // a new use of the grid object: a release still queued for the previous use must not touch it
  }
  srsran::resource_grid_writer&       get_writer() override { return writer; }
  const srsran::resource_grid_reader& get_reader() const override { return reader; }
  srsran::resource_grid_mapper&       get_mapper() override { return mapper; }

  /// The mapper / writer / reader an adaptor is handed belong to a device-mirrored grid: this is how it finds out.
  static device_resource_grid* from(srsran::resource_grid_mapper& m)
  {
    auto* p = dynamic_cast<mapper_type*>(&m);
    return p ? &p->owner : nullptr;
  }
  static device_resource_grid* from(srsran::resource_grid_writer& w)
  {
    auto* p = dynamic_cast<writer_type*>(&w);
    return p ? &p->owner : nullptr;
  }
  static const device_resource_grid* from(const srsran::resource_grid_reader& r)
  {
    auto* p = dynamic_cast<const reader_type*>(&r);
    return p ? &p->owner : nullptr;
  }

  /// Runs `fn(pool, slot_id)` -- a writer of the device layer -- on the grid's slot, opening one at the first use.
  /// NRPHY_ERR_CAPACITY when every slot of the pool is taken: the caller falls back to its host-span form.
  template <typename Fn>
  int on_device(Fn&& fn)
  {
    std::lock_guard<std::mutex> lock(lease->mutex);
    if (lease->handed_over) {
      return NRPHY_ERR_ARGUMENT;
    }
    if (!lease->have_slot) {
      int rc = nrphy_dl_slot_open(pool->get(), &lease->slot_id);
      if (rc != NRPHY_OK) {
        return rc;
      }
      lease->have_slot = true;
    }
    merged_valid = false;
    return fn(pool->get(), lease->slot_id);
  }

  /// Grid hand-over (pdxch_processor_request_handler::handle_request): merges the host layer into the slot's grid and
  /// submits the slot's modulation.  Returns the slot to poll, or -1 when the grid is empty (nothing to transmit) or
  /// no slot could be had (`status` then says why).  Runs on the caller's (upper PHY) thread; const because the
  /// reference hands the grid over as a const reader.
  int hand_over(unsigned subframe_slot_index, int& status, std::shared_ptr<dl_slot_lease>& held, uint64_t& use) const
  {
    device_resource_grid& self = const_cast<device_resource_grid&>(*this);
    dl_slot_lease&        ls   = *self.lease;
    std::lock_guard<std::mutex> lock(ls.mutex);
    status = NRPHY_OK;
    held   = self.lease;
    use    = ls.generation;
    if (!ls.have_slot && self.host_layer->get_reader().is_empty()) {
      return -1;
    }
    if (ls.handed_over) { // the same grid requested twice for one slot: the IQ is (being) computed already
      return static_cast<int>(ls.slot_id);
    }
    bool device_layer_empty = false;
    if (!ls.have_slot) { // no processor of this repository wrote the grid: everything in it came through the host layer
      status = nrphy_dl_slot_open(pool->get(), &ls.slot_id);
      if (status != NRPHY_OK) {
        return -1;
      }
      ls.have_slot       = true;
      device_layer_empty = true;
    }
    status = self.flush_host_layer(device_layer_empty);
    if (status == NRPHY_OK) {
      status = nrphy_dl_slot_modulate(pool->get(), ls.slot_id, subframe_slot_index, nullptr, nullptr);
    }
    if (status != NRPHY_OK) {
      return -1;
    }
    ls.handed_over = true;
    return static_cast<int>(ls.slot_id);
  }

  /// (Giving the slot back early -- once it has been transmitted -- goes through the lease hand_over() returned: dl_slot_lease::release.)
  const std::shared_ptr<dl_slot_pool>& get_pool() const { return pool; }

private:
  struct mapper_type : public srsran::resource_grid_mapper {
    explicit mapper_type(device_resource_grid& o) : owner(o) {}
    void map(const srsran::re_buffer_reader<srsran::cf_t>& input, const srsran::re_pattern& pattern, const srsran::precoding_configuration& precoding) override
    {
      owner.touch_host();
      owner.host_layer->get_mapper().map(input, pattern, precoding);
    }
    void map(symbol_buffer& buffer, const srsran::re_pattern_list& pattern, const srsran::re_pattern_list& reserved, const srsran::precoding_configuration& precoding, unsigned re_skip = 0) override
    {
      owner.touch_host();
      owner.host_layer->get_mapper().map(buffer, pattern, reserved, precoding, re_skip);
    }
    device_resource_grid& owner;
  };
  struct writer_type : public srsran::resource_grid_writer {
    explicit writer_type(device_resource_grid& o) : owner(o) {}
    unsigned get_nof_ports() const override { return owner.host_layer->get_writer().get_nof_ports(); }
    unsigned get_nof_subc() const override { return owner.host_layer->get_writer().get_nof_subc(); }
    unsigned get_nof_symbols() const override { return owner.host_layer->get_writer().get_nof_symbols(); }
    srsran::span<const srsran::cf_t> put(unsigned port, unsigned l, unsigned k_init, const srsran::bounded_bitset<srsran::NRE * srsran::MAX_RB>& mask, srsran::span<const srsran::cf_t> symbols) override
    {
      owner.touch_host();
      return owner.host_layer->get_writer().put(port, l, k_init, mask, symbols);
    }
    srsran::span<const srsran::cbf16_t> put(unsigned port, unsigned l, unsigned k_init, const srsran::bounded_bitset<srsran::NRE * srsran::MAX_RB>& mask, srsran::span<const srsran::cbf16_t> symbols) override
    {
      owner.touch_host();
      return owner.host_layer->get_writer().put(port, l, k_init, mask, symbols);
    }
    void put(unsigned port, unsigned l, unsigned k_init, srsran::span<const srsran::cf_t> symbols) override
    {
      owner.touch_host();
      owner.host_layer->get_writer().put(port, l, k_init, symbols);
    }
    void put(unsigned port, unsigned l, unsigned k_init, unsigned stride, srsran::span<const srsran::cbf16_t> symbols) override
    {
      owner.touch_host();
      owner.host_layer->get_writer().put(port, l, k_init, stride, symbols);
    }
    srsran::span<srsran::cbf16_t> get_view(unsigned port, unsigned l) override
    {
      owner.touch_host();
      return owner.host_layer->get_writer().get_view(port, l);
    }
    device_resource_grid& owner;
  };
  struct reader_type : public srsran::resource_grid_reader {
    explicit reader_type(const device_resource_grid& o) : owner(o) {}
    unsigned get_nof_ports() const override { return owner.host_layer->get_reader().get_nof_ports(); }
    unsigned get_nof_subc() const override { return owner.host_layer->get_reader().get_nof_subc(); }
    unsigned get_nof_symbols() const override { return owner.host_layer->get_reader().get_nof_symbols(); }
    // No device access here: pdxch_processor_impl::process_symbol asks on the real-time thread.  A port the device layer
    // may have written counts as not empty.
    bool is_empty(unsigned port) const override { return !owner.lease->have_slot && owner.host_layer->get_reader().is_empty(port); }
    bool is_empty() const override { return !owner.lease->have_slot && owner.host_layer->get_reader().is_empty(); }
    srsran::span<srsran::cf_t> get(srsran::span<srsran::cf_t> symbols, unsigned port, unsigned l, unsigned k_init, const srsran::bounded_bitset<srsran::MAX_RB * srsran::NRE>& mask) const override
    {
      return owner.host_view().get(symbols, port, l, k_init, mask);
    }
    srsran::span<srsran::cbf16_t> get(srsran::span<srsran::cbf16_t> symbols, unsigned port, unsigned l, unsigned k_init, const srsran::bounded_bitset<srsran::MAX_RB * srsran::NRE>& mask) const override
    {
      return owner.host_view().get(symbols, port, l, k_init, mask);
    }
    void get(srsran::span<srsran::cf_t> symbols, unsigned port, unsigned l, unsigned k_init, unsigned stride = 1) const override
    {
      owner.host_view().get(symbols, port, l, k_init, stride);
    }
    void get(srsran::span<srsran::cbf16_t> symbols, unsigned port, unsigned l, unsigned k_init) const override
    {
      owner.host_view().get(symbols, port, l, k_init);
    }
    srsran::span<const srsran::cbf16_t> get_view(unsigned port, unsigned l) const override { return owner.host_view().get_view(port, l); }
    const device_resource_grid& owner;
  };

  void touch_host()
  {
    std::lock_guard<std::mutex> lock(lease->mutex);
    host_touched = true;
    merged_valid = false;
  }

  // The host layer's resource elements as sparse entries (a value of zero is "not written": what the reference's writers
  // leave untouched stays zero too, and channels of a slot do not overlap).
  void host_layer_entries(std::vector<nrphy_grid_re_t>& entries) const
  {
    const srsran::resource_grid_reader& r = host_layer->get_reader();
    for (unsigned port = 0, nof_ports = r.get_nof_ports(); port != nof_ports; ++port) {
      if (r.is_empty(port)) {
        continue;
      }
      for (unsigned l = 0, nof_symbols = r.get_nof_symbols(); l != nof_symbols; ++l) {
        srsran::span<const srsran::cbf16_t> row = r.get_view(port, l);
        for (unsigned k = 0; k != row.size(); ++k) {
          uint32_t word;
          std::memcpy(&word, &row[k], sizeof(word));
          if ((word & 0x7FFF7FFFU) != 0) {
            entries.push_back({static_cast<uint16_t>(port), static_cast<uint16_t>(l), k, word});
          }
        }
      }
    }
  }

  /// The host layer joins the slot's device grid: its non-zero elements as one sparse put -- or, when nothing else has written
  /// the slot (`device_layer_empty`) and the host layer is dense (a slot whose channels were all generated on the host: the
  /// device only modulates), the whole layer as ONE grid copy, which is cheaper than twelve bytes per element and a scatter.
  int flush_host_layer(bool device_layer_empty = false)
  {
    if (!host_touched) {
      return NRPHY_OK;
    }
    std::vector<nrphy_grid_re_t> entries;
    host_layer_entries(entries);
    host_touched = false; // (what the host layer holds stays there: a later read merges it again, idempotently)
    if (entries.empty()) {
      return NRPHY_OK;
    }
    const srsran::resource_grid_reader& r        = host_layer->get_reader();
    const unsigned                      nof_subc = r.get_nof_subc();
    const size_t grid_bytes = static_cast<size_t>(pool->get_nof_ports()) * NRPHY_NSYMB * nof_subc * sizeof(srsran::cbf16_t);
    if (device_layer_empty && entries.size() * sizeof(nrphy_grid_re_t) >= grid_bytes) {
      whole.assign(static_cast<size_t>(pool->get_nof_ports()) * NRPHY_NSYMB * nof_subc, srsran::cbf16_t());
      for (unsigned port = 0; port != r.get_nof_ports() && port != pool->get_nof_ports(); ++port) {
        for (unsigned l = 0; l != r.get_nof_symbols() && l != NRPHY_NSYMB; ++l) {
          srsran::span<const srsran::cbf16_t> row = r.get_view(port, l);
          std::memcpy(&whole[(static_cast<size_t>(port) * NRPHY_NSYMB + l) * nof_subc], row.data(), row.size() * sizeof(srsran::cbf16_t));
        }
      }
      return nrphy_dl_slot_load_grid(pool->get(), lease->slot_id, whole.data());
    }
    return nrphy_dl_slot_put(pool->get(), lease->slot_id, entries.size(), entries.data());
  }

  // The whole grid on the host, for whoever reads it there: blocking.
  const srsran::resource_grid_reader& host_view() const
  {
    device_resource_grid& self = const_cast<device_resource_grid&>(*this);
    std::lock_guard<std::mutex> lock(self.lease->mutex);
    if (!lease->have_slot) {
      return host_layer->get_reader();
    }
    if (!merged_valid) {
      const srsran::resource_grid_reader& r        = host_layer->get_reader();
      const unsigned                      nof_subc = r.get_nof_subc(), nof_ports = r.get_nof_ports();
      std::vector<srsran::cbf16_t>        raw(static_cast<size_t>(pool->get_nof_ports()) * NRPHY_NSYMB * nof_subc);
      int rc = nrphy_dl_slot_read_grid(pool->get(), lease->slot_id, raw.data());
      report_failure("nrphy_dl_slot_read_grid", rc);
      self.merged->set_all_zero();
      for (unsigned port = 0; port != nof_ports && port != pool->get_nof_ports(); ++port) {
        for (unsigned l = 0; l != r.get_nof_symbols(); ++l) {
          self.merged->get_writer().put(port, l, 0, 1, srsran::span<const srsran::cbf16_t>(&raw[(static_cast<size_t>(port) * NRPHY_NSYMB + l) * nof_subc], nof_subc));
        }
      }
      std::vector<nrphy_grid_re_t> entries;
      host_layer_entries(entries);
      for (const nrphy_grid_re_t& e : entries) {
        srsran::cbf16_t v;
        std::memcpy(&v, &e.value, sizeof(v));
        self.merged->get_writer().put(e.port, e.symbol, e.subc, 1, srsran::span<const srsran::cbf16_t>(&v, 1));
      }
      self.merged_valid = true;
    }
    return merged->get_reader();
  }

  std::shared_ptr<dl_slot_pool>          pool;
  std::shared_ptr<dl_slot_lease>         lease;      // the pool slot of the device layer (shared with the lower PHY's request)
  std::unique_ptr<srsran::resource_grid> host_layer; // what host-side processors write
  std::unique_ptr<srsran::resource_grid> merged;     // device + host layers, built only when somebody reads on the host
  mapper_type                            mapper;
  writer_type                            writer;
  reader_type                            reader;
  std::vector<srsran::cbf16_t>           whole;      // staging of a dense host layer on its way to the device (flush_host_layer)
  bool                                   host_touched = false;
  bool                                   merged_valid = false;
};

/// resource_grid_factory whose grids keep the accelerated channels on the device.  `inner` makes the host layers (the
/// reference's own factory, create_resource_grid_factory()).  Grids of another geometry than the pool's come out as plain
/// writer-access grids (resource_grid_adaptor), e.g. the uplink grids if the same factory is used for them.
class device_resource_grid_factory : public srsran::resource_grid_factory
{
public:
  device_resource_grid_factory(std::shared_ptr<dl_slot_pool> pool_, std::shared_ptr<srsran::resource_grid_factory> inner_) :
    pool(std::move(pool_)), inner(std::move(inner_))
  {
  }
  std::unique_ptr<srsran::resource_grid> create(unsigned nof_ports, unsigned nof_symbols, unsigned nof_subc) override
  {
    if (nof_ports != pool->get_nof_ports() || nof_subc != pool->get_nof_subc() || nof_symbols != NRPHY_NSYMB) {
      return std::make_unique<resource_grid_adaptor>(inner->create(nof_ports, nof_symbols, nof_subc));
    }
    return std::make_unique<device_resource_grid>(pool, inner->create(nof_ports, nof_symbols, nof_subc), inner->create(nof_ports, nof_symbols, nof_subc));
  }

private:
  std::shared_ptr<dl_slot_pool>                  pool;
  std::shared_ptr<srsran::resource_grid_factory> inner;
};

/// The RE of one transmission: per OFDM symbol the subcarriers it owns.
using re_symbol_masks = std::array<srsran::bounded_bitset<srsran::MAX_RB * srsran::NRE>, NRPHY_NSYMB>;

/// Puts the RE `masks` select from a staging grid [port][14][nof_subc] into the grid behind `mapper`, ports 0..nof_ports-1.
inline void put_staging_grid(srsran::resource_grid_mapper& mapper,
                             const srsran::cbf16_t*        staging,
                             unsigned                      nof_ports,
                             unsigned                      nof_subc,
                             const re_symbol_masks&        masks)
{
  using namespace srsran;
  // A grid that gives its writer: resource_grid_adaptor, or the host layer of a device-mirrored grid.
  resource_grid_writer* direct = nullptr;
  if (auto* with_writer = dynamic_cast<resource_grid_mapper_with_writer*>(&mapper)) {
    direct = &with_writer->get_writer();
  } else if (device_resource_grid* dg = device_resource_grid::from(mapper)) {
    direct = &dg->get_writer();
  }
  std::vector<cbf16_t> packed(nof_subc);
  for (unsigned l = 0; l != NRPHY_NSYMB; ++l) {
    const bounded_bitset<MAX_RB * NRE>& mask = masks[l];
    if (mask.size() == 0 || mask.none()) {
      continue;
    }
    if (direct != nullptr) {
      for (unsigned port = 0; port != nof_ports; ++port) {
        const cbf16_t* row = &staging[(static_cast<size_t>(port) * NRPHY_NSYMB + l) * nof_subc];
        unsigned       n   = 0;
        mask.for_each(0, mask.size(), [&](unsigned k) { packed[n++] = row[k]; });
        direct->put(port, l, 0, mask, span<const cbf16_t>(packed).first(n));
      }
      continue;
    }
    // Through the mapper: PRBs that share a RE mask make one re_pattern; identity precoding keeps every port's values.
    // ones on the diagonal (the codebook's make_identity is normalised by 1 / sqrt(ports))
    precoding_configuration identity(nof_ports, nof_ports, 1, MAX_RB);
    for (unsigned layer = 0; layer != nof_ports; ++layer) {
      for (unsigned port = 0; port != nof_ports; ++port) {
        identity.set_coefficient(layer == port ? cf_t(1.0F, 0.0F) : cf_t(0.0F, 0.0F), layer, port, 0);
      }
    }
    std::array<bool, MAX_RB>      done     = {};
    const unsigned                nof_prb  = mask.size() / NRE;
    for (unsigned first = 0; first != nof_prb; ++first) {
      uint32_t bits = 0;
      for (unsigned k = 0; k != NRE; ++k) {
        bits |= mask.test(NRE * first + k) ? (1U << k) : 0U;
      }
      if (done[first] || bits == 0) {
        continue;
      }
      re_pattern pattern;
      pattern.prb_mask.resize(nof_prb);
      pattern.symbols.set(l);
      for (unsigned k = 0; k != NRE; ++k) {
        pattern.re_mask.set(k, (bits >> k) & 1U);
      }
      unsigned nof_re = 0;
      for (unsigned prb = first; prb != nof_prb; ++prb) {
        uint32_t other = 0;
        for (unsigned k = 0; k != NRE; ++k) {
          other |= mask.test(NRE * prb + k) ? (1U << k) : 0U;
        }
        if (!done[prb] && other == bits) {
          done[prb] = true;
          pattern.prb_mask.set(prb);
          nof_re += __builtin_popcount(bits);
        }
      }
      dynamic_re_buffer<cf_t> values(nof_ports, nof_re);
      for (unsigned port = 0; port != nof_ports; ++port) {
        const cbf16_t* row = &staging[(static_cast<size_t>(port) * NRPHY_NSYMB + l) * nof_subc];
        span<cf_t>     dst = values.get_slice(port);
        unsigned       n   = 0;
        for (unsigned prb = 0; prb != nof_prb; ++prb) {
          if (pattern.prb_mask.test(prb)) {
            for (unsigned k = 0; k != NRE; ++k) {
              if ((bits >> k) & 1U) {
                dst[n++] = to_cf(row[NRE * prb + k]);
              }
            }
          }
        }
      }
      mapper.map(values, pattern, identity);
    }
  }
}

/// The RE a PDSCH transmission owns: data RE built like pdsch_modulator_impl::map does (pdsch_modulator_impl.cpp:52-106)
/// plus the DM-RS RE of the CDM groups of its layers.
inline void pdsch_re_masks(re_symbol_masks& masks, const srsran::pdsch_processor::pdu_t& pdu, unsigned nof_subc)
{
  using namespace srsran;
  const bounded_bitset<MAX_RB> prb_mask = pdu.freq_alloc.get_prb_mask(pdu.bwp_start_rb, pdu.bwp_size_rb);
  re_pattern_list              reserved(pdu.reserved);
  reserved.merge(pdu.dmrs.get_dmrs_pattern(pdu.bwp_start_rb, pdu.bwp_size_rb, pdu.nof_cdm_groups_without_data, pdu.dmrs_symbol_mask));
  re_pattern alloc;
  alloc.prb_mask = prb_mask;
  alloc.re_mask  = ~re_prb_mask();
  alloc.symbols.fill(pdu.start_symbol_index, pdu.start_symbol_index + pdu.nof_symbols);
  const unsigned nof_groups = (pdu.precoding.get_nof_layers() + 1) / 2;
  for (unsigned l = 0; l != NRPHY_NSYMB; ++l) {
    masks[l].resize(nof_subc);
    masks[l].reset();
    if (l >= get_nsymb_per_slot(pdu.cp)) {
      continue;
    }
    alloc.get_inclusion_mask(masks[l], l);
    reserved.get_exclusion_mask(masks[l], l);
    if (pdu.dmrs_symbol_mask.test(l)) {
      for (unsigned prb = 0; prb != prb_mask.size(); ++prb) {
        if (prb_mask.test(prb)) {
          for (unsigned k = 0; k != NRE; ++k) {
            if ((k % 2) < nof_groups) {
              masks[l].set(NRE * prb + k);
            }
          }
        }
      }
    }
  }
}

/// pdsch_processor over the asynchronous queue of the C ABI (nrphy_pdsch_async_*): process() copies the transport block,
/// enqueues the PDU on one of `depth` streams and returns; when the PDU's grid has reached the host, a thread of the HIP
/// runtime merges its RE into the caller's grid and calls the notifier -- exactly once per process(), also on failure
/// (pdsch_processor.h:157-170; the contract of pdsch_processor_asynchronous_pool.h:124-130).  The grid behind `mapper`
/// must stay valid until then, as for the reference's concurrent processor.
class pdsch_processor_adaptor : public srsran::pdsch_processor
{
public:
  pdsch_processor_adaptor(std::shared_ptr<context> ctx_, unsigned nof_ports_, unsigned nof_subc_, unsigned depth = 4) :
    ctx(std::move(ctx_)), nof_ports(nof_ports_), nof_subc(nof_subc_), ops(2 * depth)
  {
    int rc = nrphy_pdsch_async_create(ctx->get(), depth, nof_ports, nof_subc, 1277992 / 8 + 8, &queue);
    report_failure("nrphy_pdsch_async_create", rc);
    srsran_assert(rc == NRPHY_OK, "nrphy_pdsch_async_create failed.");
  }
  ~pdsch_processor_adaptor() override { nrphy_pdsch_async_destroy(queue); } // waits for what is in flight

  void process(srsran::resource_grid_mapper&                                                mapper,
               srsran::pdsch_processor_notifier&                                            notifier,
               srsran::static_vector<srsran::span<const uint8_t>, MAX_NOF_TRANSPORT_BLOCKS> data,
               const pdu_t&                                                                 pdu) override
  {
    using namespace srsran;
    std::vector<float> weights;
    nrphy_pdsch_pdu_t  pod = to_pod(pdu, data[0].size(), weights);
    // The reference asserts on invalid PDUs (pdsch_processor_validator_impl::assert_pdu).
    srsran_assert(nrphy_pdsch_validate(&pod) == NRPHY_OK, "Invalid PDSCH PDU.");
    if (device_resource_grid* dg = device_resource_grid::from(mapper)) {
      // A device-mirrored grid: the PDU goes into the slot's grid in HBM, ordered on the slot's stream with everything
      // else of the slot; the transport block is copied at the call, so the PDU is "done" as far as the caller's spans
      // and the grid's hand-over are concerned (a device failure surfaces when the slot is transmitted).
      const uint8_t* tb = data[0].data();
      int rc = dg->on_device([&](nrphy_dl_slots_t* p, uint32_t id) { return nrphy_dl_slot_pdsch(p, id, 1, &pod, &tb); });
      if (rc != NRPHY_ERR_CAPACITY) {
        report_failure("nrphy_dl_slot_pdsch", rc);
        notifier.on_finish_processing();
        return;
      }
      // no slot free, or the slot's staging is full: through the host, into the grid's host layer
    }
    for (;;) {
      operation* op = nullptr;
      for (operation& o : ops) {
        bool expected = false;
        if (o.busy.compare_exchange_strong(expected, true)) {
          op = &o;
          break;
        }
      }
      if (op != nullptr) {
        op->self      = this;
        op->mapper    = &mapper;
        op->notifier  = &notifier;
        op->nof_ports = pod.nof_ports;
        pdsch_re_masks(op->masks, pdu, nof_subc);
        int rc = nrphy_pdsch_async_submit(queue, &pod, data[0].data(), &pdsch_processor_adaptor::on_done, op);
        if (rc == NRPHY_OK) {
          return;
        }
        op->busy.store(false);
        if (rc != NRPHY_ERR_CAPACITY) {
          // Like the reference's pool when it runs out of processors: log, and still notify exactly once.
          report_failure("nrphy_pdsch_async_submit", rc);
          notifier.on_finish_processing();
          return;
        }
      }
      // `depth` operations in flight: wait until ONE has completed (not for all of them), then retry.  There are
      // twice as many operation records as queue slots, because a record is released by the completion handler a moment
      // before the queue marks its slot free: a record is then always at hand and the wait below is for the queue alone.
      nrphy_pdsch_async_wait_slot(queue);
    }
  }

private:
  struct operation {
    std::atomic<bool>                 busy{false};
    pdsch_processor_adaptor*          self     = nullptr;
    srsran::resource_grid_mapper*     mapper   = nullptr;
    srsran::pdsch_processor_notifier* notifier = nullptr;
    unsigned                          nof_ports = 0;
    re_symbol_masks                   masks;
  };

  // Runs on a thread of the HIP runtime (no HIP or nrphy calls here).
  static void on_done(void* user, int status, const void* grid)
  {
    operation* op = static_cast<operation*>(user);
    if (status == NRPHY_OK) {
      put_staging_grid(*op->mapper, static_cast<const srsran::cbf16_t*>(grid), op->nof_ports, op->self->nof_subc, op->masks);
    } else {
      report_failure("PDSCH processing", status);
    }
    srsran::pdsch_processor_notifier* notifier = op->notifier;
    op->busy.store(false);
    notifier->on_finish_processing();
  }

  std::shared_ptr<context> ctx;
  unsigned                 nof_ports;
  unsigned                 nof_subc;
  nrphy_pdsch_async_t*     queue = nullptr;
  std::vector<operation>   ops;
};

class pdsch_processor_factory_adaptor : public srsran::pdsch_processor_factory
{
public:
  pdsch_processor_factory_adaptor(std::shared_ptr<context> ctx_, unsigned nof_ports_, unsigned nof_subc_, unsigned depth_ = 4) :
    ctx(std::move(ctx_)), nof_ports(nof_ports_), nof_subc(nof_subc_), depth(depth_)
  {
  }
  std::unique_ptr<srsran::pdsch_processor> create() override
  {
    return std::make_unique<pdsch_processor_adaptor>(ctx, nof_ports, nof_subc, depth);
  }
  std::unique_ptr<srsran::pdsch_pdu_validator> create_validator() override
  {
    return std::make_unique<pdsch_pdu_validator_adaptor>();
  }

private:
  std::shared_ptr<context> ctx;
  unsigned                 nof_ports;
  unsigned                 nof_subc;
  unsigned                 depth;
};

/// hal::hw_accelerator_pdsch_enc over nrphy_pdsch_encode_host, transport-block mode (get_cb_mode() == false): what
/// pdsch_encoder_hw_impl::encode (pdsch_encoder_hw_impl.cpp:34-180) drives -- configure, enqueue the raw transport
/// block, dequeue the rate-matched codeword (unpacked bits in `data`, packed in `aux_data`).  One operation in flight.
class hw_accelerator_pdsch_enc_adaptor : public srsran::hal::hw_accelerator_pdsch_enc
{
public:
  explicit hw_accelerator_pdsch_enc_adaptor(std::shared_ptr<context> ctx_) : ctx(std::move(ctx_)) {}

  void reserve_queue() override {}
  void free_queue() override {}
  bool get_cb_mode() const override { return false; }
  unsigned get_max_tb_size() const override { return 1277992 / 8; }

  void configure_operation(const srsran::hal::hw_pdsch_encoder_configuration& config, unsigned cb_index = 0) override
  {
    (void)cb_index;
    cfg = config;
  }

  bool enqueue_operation(srsran::span<const uint8_t> data, srsran::span<const uint8_t> aux_data = {}, unsigned cb_index = 0) override
  {
    (void)aux_data;
    (void)cb_index;
    if (pending) {
      return false; // queue depth 1: the caller dequeues, then retries
    }
    tb.assign(data.begin(), data.end());
    pending = true;
    return true;
  }

  bool dequeue_operation(srsran::span<uint8_t> data, srsran::span<uint8_t> aux_data = {}, unsigned segment_index = 0) override
  {
    (void)segment_index;
    using namespace srsran;
    if (!pending) {
      return false;
    }
    pending = false;
    nrphy_pdsch_encoder_cfg_t enc;
    enc.base_graph    = (cfg.base_graph_index == ldpc_base_graph_type::BG1) ? 1 : 2;
    enc.rv            = cfg.rv;
    enc.qm            = get_bits_per_symbol(cfg.modulation);
    enc.nref          = cfg.Nref;
    enc.tb_size_bytes = tb.size();
    // The HAL configuration carries the per-segment lengths instead of the layer count: pick the layer count whose
    // rate-matching split (TS 38.212 Section 5.4.2.1) reproduces them.
    unsigned cw_bits   = cfg.nof_short_segments * cfg.cw_length_a + (cfg.nof_segments - cfg.nof_short_segments) * cfg.cw_length_b;
    enc.nof_ch_symbols = cw_bits / enc.qm;
    enc.nof_layers     = 1;
    for (unsigned layers = 1; layers <= NRPHY_MAX_LAYERS; ++layers) {
      if (enc.nof_ch_symbols % layers != 0) {
        continue;
      }
      unsigned per_layer = enc.nof_ch_symbols / layers, c = cfg.nof_segments;
      unsigned n_short = c - per_layer % c, e_short = layers * enc.qm * (per_layer / c);
      unsigned e_long = layers * enc.qm * ((per_layer + c - 1) / c);
      if (n_short == cfg.nof_short_segments && e_short == cfg.cw_length_a && (n_short == c || e_long == cfg.cw_length_b)) {
        enc.nof_layers = layers;
        break;
      }
    }
    srsran_assert(data.size() == cw_bits, "Invalid codeword size.");
    int rc = nrphy_pdsch_encode_host(ctx->get(), &enc, tb.data(), data.data(), aux_data.empty() ? nullptr : aux_data.data());
    report_failure("nrphy_pdsch_encode_host", rc);
    return true;
  }

private:
  std::shared_ptr<context>                     ctx;
  srsran::hal::hw_pdsch_encoder_configuration  cfg;
  std::vector<uint8_t>                         tb;
  bool                                         pending = false;
};

class hw_accelerator_pdsch_enc_factory_adaptor : public srsran::hal::hw_accelerator_pdsch_enc_factory
{
public:
  explicit hw_accelerator_pdsch_enc_factory_adaptor(std::shared_ptr<context> ctx_) : ctx(std::move(ctx_)) {}
  std::unique_ptr<srsran::hal::hw_accelerator_pdsch_enc> create() override
  {
    return std::make_unique<hw_accelerator_pdsch_enc_adaptor>(ctx);
  }

private:
  std::shared_ptr<context> ctx;
};

/// ofdm_symbol_modulator over nrphy_ofdm_modulate_slot_host.  The real-time loop asks one (port, symbol) at a time;
/// the adaptor modulates the whole slot of every port on the first request of a slot and serves the rest from its cache.
class ofdm_symbol_modulator_adaptor : public srsran::ofdm_symbol_modulator
{
public:
  ofdm_symbol_modulator_adaptor(std::shared_ptr<context> ctx_, const srsran::ofdm_modulator_configuration& config, unsigned nof_ports_) :
    ctx(std::move(ctx_)), nof_ports(nof_ports_)
  {
    cfg.numerology     = config.numerology;
    cfg.bw_rb          = config.bw_rb;
    cfg.dft_size       = config.dft_size;
    cfg.cp             = (config.cp == srsran::cyclic_prefix::NORMAL) ? 0 : 1;
    cfg.scale          = config.scale;
    cfg.center_freq_hz = config.center_freq_hz;
    int rc             = nrphy_ofdm_plan_create(ctx->get(), &cfg, nof_ports, &plan);
    report_failure("nrphy_ofdm_plan_create", rc);
    staging.resize(static_cast<size_t>(nof_ports) * NRPHY_NSYMB * cfg.bw_rb * srsran::NRE);
  }
  ~ofdm_symbol_modulator_adaptor() override { nrphy_ofdm_plan_destroy(plan); }

  unsigned get_symbol_size(unsigned symbol_index) const override { return nrphy_ofdm_symbol_size(&cfg, symbol_index); }

  void modulate(srsran::span<srsran::cf_t> output, const srsran::resource_grid_reader& grid, unsigned port_index, unsigned symbol_index) override
  {
    using namespace srsran;
    srsran_assert(output.size() == get_symbol_size(symbol_index), "Invalid output size."); // ofdm_modulator_impl.cpp:68-75
    if (grid.is_empty(port_index)) {
      std::fill(output.begin(), output.end(), cf_t());
      return;
    }
    const unsigned nsymb = cfg.cp ? 12 : 14; // symbols per slot; the staging grid keeps NRPHY_NSYMB rows per port
    unsigned       slot = symbol_index / nsymb, l = symbol_index % nsymb;
    if (&grid != cached_grid || slot != cached_slot || (l == 0 && port_index == 0)) {
      unsigned nof_subc = cfg.bw_rb * NRE;
      for (unsigned port = 0; port != nof_ports; ++port) {
        for (unsigned sym = 0; sym != nsymb; ++sym) {
          span<const cbf16_t> view = grid.get_view(port, sym);
          std::memcpy(&staging[(static_cast<size_t>(port) * NRPHY_NSYMB + sym) * nof_subc], view.data(), nof_subc * sizeof(cbf16_t));
        }
      }
      slot_size = nrphy_ofdm_slot_size(&cfg, slot);
      iq.resize(static_cast<size_t>(nof_ports) * slot_size);
      int rc = nrphy_ofdm_modulate_slot_host(plan, staging.data(), slot, reinterpret_cast<float*>(iq.data()));
      report_failure("nrphy_ofdm_modulate_slot_host", rc);
      if (rc != NRPHY_OK) {
        std::fill(iq.begin(), iq.end(), cf_t()); // silence rather than stale samples
      }
      cached_grid = &grid;
      cached_slot = slot;
    }
    unsigned offset = 0;
    for (unsigned sym = 0; sym != l; ++sym) {
      offset += get_symbol_size(slot * nsymb + sym);
    }
    std::memcpy(output.data(), &iq[static_cast<size_t>(port_index) * slot_size + offset], output.size() * sizeof(cf_t));
  }

private:
  std::shared_ptr<context>            ctx;
  nrphy_ofdm_config_t                 cfg;
  nrphy_ofdm_plan_t*                  plan = nullptr;
  unsigned                            nof_ports;
  std::vector<srsran::cbf16_t>        staging;
  std::vector<srsran::cf_t>           iq;
  const srsran::resource_grid_reader* cached_grid = nullptr;
  unsigned                            cached_slot = ~0U;
  unsigned                            slot_size   = 0;
};

class ofdm_slot_modulator_adaptor : public srsran::ofdm_slot_modulator
{
public:
  ofdm_slot_modulator_adaptor(std::shared_ptr<context> ctx_, const srsran::ofdm_modulator_configuration& config, unsigned nof_ports_) :
    symbol_modulator(std::move(ctx_), config, nof_ports_), nsymb(srsran::get_nsymb_per_slot(config.cp))
  {
  }
  unsigned get_slot_size(unsigned slot_index) const override
  {
    unsigned n = 0;
    for (unsigned l = 0; l != nsymb; ++l) {
      n += symbol_modulator.get_symbol_size(nsymb * slot_index + l);
    }
    return n;
  }
  void modulate(srsran::span<srsran::cf_t> output, const srsran::resource_grid_reader& grid, unsigned port_index, unsigned slot_index) override
  {
    for (unsigned l = 0; l != nsymb; ++l) { // ofdm_modulator_impl.cpp:115-139
      unsigned size = symbol_modulator.get_symbol_size(nsymb * slot_index + l);
      symbol_modulator.modulate(output.first(size), grid, port_index, nsymb * slot_index + l);
      output = output.last(output.size() - size);
    }
  }

private:
  ofdm_symbol_modulator_adaptor symbol_modulator;
  unsigned                      nsymb;
};

class ofdm_modulator_factory_adaptor : public srsran::ofdm_modulator_factory
{
public:
  ofdm_modulator_factory_adaptor(std::shared_ptr<context> ctx_, unsigned nof_ports_) : ctx(std::move(ctx_)), nof_ports(nof_ports_) {}
  std::unique_ptr<srsran::ofdm_symbol_modulator> create_ofdm_symbol_modulator(const srsran::ofdm_modulator_configuration& config) override
  {
    return std::make_unique<ofdm_symbol_modulator_adaptor>(ctx, config, nof_ports);
  }
  std::unique_ptr<srsran::ofdm_slot_modulator> create_ofdm_slot_modulator(const srsran::ofdm_modulator_configuration& config) override
  {
    return std::make_unique<ofdm_slot_modulator_adaptor>(ctx, config, nof_ports);
  }

private:
  std::shared_ptr<context> ctx;
  unsigned                 nof_ports;
};

// ---- the downlink slot pipeline: a device-mirrored resource grid and the lower PHY's pdxch_processor ------------------
// Seam C as SURVEY.md section 8b asks for it: the slot is modulated WHEN THE GRID IS HANDED OVER
// (pdxch_processor_request_handler::handle_request, lib/phy/lower/processors/downlink/pdxch/pdxch_processor_impl.cpp:97-112)
// and the real-time thread (pdxch_processor_baseband::process_symbol, :47-95) only copies from pinned host memory -- and
// seam A with the grid staying in HBM in between: the channel processor adaptors of this header, handed a grid made by
// device_resource_grid_factory, write the slot's DEVICE grid (nrphy_dl_slot_pdsch / _pdcch / _ssb / _csi_rs) instead of
// bringing their resource elements to the host.  What the gNB swaps: the resource_grid_factory where upper_phy builds its
// downlink grid pool (upper_phy_factories.cpp:313-328) and the pdxch_processor_factory where the lower PHY's downlink
// processor is built (create_pdxch_processor_factory_sw, pdxch_processor_factories.h:52-59); see INTEGRATION.md section 4.

/// pdxch_processor over the slot pipeline.  handle_request (upper PHY thread) submits the slot: a device-mirrored grid is
/// handed over where it is (device_resource_grid::hand_over), any other grid is copied into a slot of the pool
/// (nrphy_dl_slot_load_grid: one host-to-device copy, asynchronous) -- either way the whole slot is modulated on the
/// device at once and its IQ lands in pinned host memory.  process_symbol (real-time thread) makes no device or library
/// call beyond an atomic load (nrphy_dl_slot_poll) and pointer arithmetic (nrphy_dl_slot_iq): it copies one symbol per port.
/// A slot whose IQ has not arrived when its first symbol is due is treated like a late request: on_pdxch_request_late,
/// silence for the slot (the reference's own behaviour for a request that misses its slot, pdxch_processor_impl.cpp:65-74)
/// -- unless `max_wait_us` allows the real-time thread to wait that long once per slot.
class pdxch_processor_adaptor : public srsran::pdxch_processor,
                                private srsran::pdxch_processor_baseband,
                                private srsran::pdxch_processor_request_handler
{
public:
  pdxch_processor_adaptor(std::shared_ptr<dl_slot_pool> pool_, srsran::cyclic_prefix cp, unsigned nof_tx_ports_, unsigned max_wait_us_ = 0) :
    pool(std::move(pool_)), nof_symbols_per_slot(srsran::get_nsymb_per_slot(cp)), nof_tx_ports(nof_tx_ports_), max_wait_us(max_wait_us_)
  {
    srsran_assert(nof_tx_ports <= pool->get_nof_ports(), "More transmit ports than the slot pool has.");
    const nrphy_ofdm_config_t& cfg = pool->ofdm_config();
    const unsigned nof_symbols_subframe = nof_symbols_per_slot << cfg.numerology;
    symbol_offset.resize(nof_symbols_subframe);
    symbol_size.resize(nof_symbols_subframe);
    for (unsigned s = 0, offset = 0; s != nof_symbols_subframe; ++s) {
      if (s % nof_symbols_per_slot == 0) {
        offset = 0;
      }
      symbol_offset[s] = offset;
      symbol_size[s]   = nrphy_ofdm_symbol_size(&cfg, s);
      offset += symbol_size[s];
    }
    staging.resize(static_cast<size_t>(pool->get_nof_ports()) * NRPHY_NSYMB * pool->get_nof_subc());
  }
  ~pdxch_processor_adaptor() override
  {
    for (request_slot& r : requests) {
      release(r.take());
    }
    release(current);
    drain_retired();
  }

  void                                     connect(srsran::pdxch_processor_notifier& notifier_) override { notifier = &notifier_; }
  srsran::pdxch_processor_request_handler& get_request_handler() override { return *this; }
  srsran::pdxch_processor_baseband&        get_baseband() override { return *this; }

private:
  // A request of the pool: the slot it is for and where its IQ will be.
  struct request {
    srsran::slot_point          slot;
    std::shared_ptr<dl_slot_lease> lease;            // a device-mirrored grid was handed over: its pool slot ...
    uint64_t                       lease_use = 0;    // ... and which use of the grid object this request belongs to
    int                         pool_slot = -1;      // pool slot with the request's IQ; -1: nothing to transmit
    bool                        own_slot  = false;   // the adaptor opened pool_slot (a plain host grid was loaded into it)
    bool                        valid     = false;   // a grid was given (what the reference encodes as grid != nullptr)
  };
  // resource_grid_request_pool (lib/phy/lower/processors/resource_grid_request_pool.h:37-77): one request per slot modulo 16,
  // exchanged under a lock that is only ever held for a copy.
  struct request_slot {
    request exchange(const request& r)
    {
      std::lock_guard<std::mutex> lock(mutex);
      return std::exchange(value, r);
    }
    request take() { return exchange(request()); }
    request    value;
    std::mutex mutex;
  };
  static constexpr unsigned REQUEST_ARRAY_SIZE = 16;

  void handle_request(const srsran::resource_grid_reader& grid, const srsran::resource_grid_context& context) override
  {
    using namespace srsran;
    srsran_assert(notifier != nullptr, "Notifier has not been connected.");
    drain_retired(); // slots the real-time thread is done with (it never calls the library itself)
    request r;
    r.slot  = context.slot;
    r.valid = true;
    int status = NRPHY_OK;
    if (const device_resource_grid* dg = device_resource_grid::from(grid)) {
      r.pool_slot = dg->hand_over(context.slot.subframe_slot_index(), status, r.lease, r.lease_use);
    } else if (!grid.is_empty()) {
      uint32_t id = 0;
      status      = nrphy_dl_slot_open(pool->get(), &id);
      if (status == NRPHY_OK) {
        std::lock_guard<std::mutex> staging_lock(staging_mutex);
        const unsigned nof_subc = pool->get_nof_subc();
        std::fill(staging.begin(), staging.end(), cbf16_t());
        for (unsigned port = 0; port != pool->get_nof_ports() && port != grid.get_nof_ports(); ++port) {
          if (grid.is_empty(port)) {
            continue;
          }
          for (unsigned l = 0; l != nof_symbols_per_slot; ++l) {
            span<const cbf16_t> view = grid.get_view(port, l);
            std::memcpy(&staging[(static_cast<size_t>(port) * NRPHY_NSYMB + l) * nof_subc], view.data(), nof_subc * sizeof(cbf16_t));
          }
        }
        status = nrphy_dl_slot_load_grid(pool->get(), id, staging.data());
        if (status == NRPHY_OK) {
          status = nrphy_dl_slot_modulate(pool->get(), id, context.slot.subframe_slot_index(), nullptr, nullptr);
        }
        if (status == NRPHY_OK) {
          r.pool_slot = static_cast<int>(id);
          r.own_slot  = true;
        } else {
          nrphy_dl_slot_close(pool->get(), id);
        }
      }
    }
    report_failure("downlink slot submission", status);
    request previous = requests[context.slot.system_slot() % REQUEST_ARRAY_SIZE].exchange(r);
    if (previous.valid) {
      // A request for this position was never transmitted (pdxch_processor_impl.cpp:104-111).
      resource_grid_context late_context;
      late_context.slot   = previous.slot;
      late_context.sector = context.sector;
      notifier->on_pdxch_request_late(late_context);
      release(previous);
    }
  }

  bool process_symbol(srsran::baseband_gateway_buffer_writer& samples, const symbol_context& context) override
  {
    using namespace srsran;
    srsran_assert(notifier != nullptr, "Notifier has not been connected.");
    if (context.slot != current_slot) {
      retire(current);
      current      = request();
      current_slot = context.slot;
      request r    = requests[context.slot.system_slot() % REQUEST_ARRAY_SIZE].take();
      if (!r.valid) {
        return false;
      }
      bool late = r.slot != current_slot;
      if (!late && r.pool_slot >= 0) {
        int rc = nrphy_dl_slot_poll(pool->get(), r.pool_slot);
        for (unsigned waited = 0; rc == NRPHY_ERR_NOT_READY && waited < max_wait_us; waited += 5) {
          std::this_thread::sleep_for(std::chrono::microseconds(5));
          rc = nrphy_dl_slot_poll(pool->get(), r.pool_slot);
        }
        late = rc != NRPHY_OK; // not there yet, or the device failed: silence either way
      }
      if (late) {
        resource_grid_context late_context;
        late_context.slot   = r.slot;
        late_context.sector = context.sector;
        notifier->on_pdxch_request_late(late_context);
        retire(r);
        return false;
      }
      current = r;
    }
    if (current.pool_slot < 0) {
      return false; // no request, or an empty grid
    }
    const unsigned symbol_index_subframe = context.symbol + context.slot.subframe_slot_index() * nof_symbols_per_slot;
    const unsigned size = symbol_size[symbol_index_subframe], offset = symbol_offset[symbol_index_subframe];
    for (unsigned i_port = 0; i_port != nof_tx_ports; ++i_port) {
      span<cf_t> out = samples.get_channel_buffer(i_port);
      srsran_assert(out.size() == size, "Invalid output size."); // ofdm_modulator_impl.cpp:68-75
      const cf_t* iq = static_cast<const cf_t*>(nrphy_dl_slot_iq(pool->get(), current.pool_slot, i_port, nullptr));
      std::memcpy(out.data(), iq + offset, size * sizeof(cf_t));
    }
    return true;
  }

  // Real-time side: a request it is done with goes onto a list the upper PHY's thread empties (closing a pool slot
  // synchronises its stream -- idle by then, but still a runtime call).
  void retire(const request& r)
  {
    if (!r.valid || (!r.lease && !r.own_slot)) {
      return;
    }
    std::lock_guard<std::mutex> lock(retired_mutex);
    retired.push_back(r);
  }
  void drain_retired()
  {
    std::vector<request> list;
    {
      std::lock_guard<std::mutex> lock(retired_mutex);
      list.swap(retired);
    }
    for (const request& r : list) {
      release(r);
    }
  }
  void release(const request& r)
  {
    if (r.lease) {
      r.lease->release(r.lease_use);
    } else if (r.own_slot && r.pool_slot >= 0) {
      nrphy_dl_slot_close(pool->get(), r.pool_slot);
    }
  }

  std::shared_ptr<dl_slot_pool>                 pool;
  unsigned                                      nof_symbols_per_slot;
  unsigned                                      nof_tx_ports;
  unsigned                                      max_wait_us;
  srsran::pdxch_processor_notifier*             notifier = nullptr;
  std::vector<unsigned>                         symbol_offset, symbol_size; // within the slot, per symbol of the subframe
  std::vector<srsran::cbf16_t>                  staging;
  std::mutex                                    staging_mutex;
  std::array<request_slot, REQUEST_ARRAY_SIZE>  requests;
  srsran::slot_point                            current_slot;
  request                                       current;
  std::mutex                                    retired_mutex;
  std::vector<request>                          retired;
};

/// Drop-in for create_pdxch_processor_factory_sw (pdxch_processor_factories.h:52-59).
class pdxch_processor_factory_adaptor : public srsran::pdxch_processor_factory
{
public:
  explicit pdxch_processor_factory_adaptor(std::shared_ptr<dl_slot_pool> pool_, unsigned max_wait_us_ = 0) : pool(std::move(pool_)), max_wait_us(max_wait_us_) {}
  std::unique_ptr<srsran::pdxch_processor> create(const srsran::pdxch_processor_configuration& config) override
  {
    // The pool was made for one modulator configuration (pdxch_processor_factories.cpp:41-48 derives the same from `config`).
    const nrphy_ofdm_config_t& c = pool->ofdm_config();
    srsran_assert(c.numerology == srsran::to_numerology_value(config.scs) && c.bw_rb == config.bandwidth_rb &&
                      c.dft_size == config.srate.get_dft_size(config.scs) && (c.cp != 0) == (config.cp == srsran::cyclic_prefix::EXTENDED) &&
                      c.center_freq_hz == config.center_freq_Hz && config.nof_tx_ports <= pool->get_nof_ports(),
                  "The slot pool was created for another carrier configuration.");
    return std::make_unique<pdxch_processor_adaptor>(pool, config.cp, config.nof_tx_ports, max_wait_us);
  }

private:
  std::shared_ptr<dl_slot_pool> pool;
  unsigned                      max_wait_us;
};

/// ofdm_symbol_demodulator / ofdm_slot_demodulator over the host-span demodulator entry points (receive side,
/// ofdm_demodulator_impl.cpp).  One plan with a single port: the reference calls per port.
class ofdm_demodulator_adaptor_base
{
protected:
  ofdm_demodulator_adaptor_base(std::shared_ptr<context> ctx_, const srsran::ofdm_demodulator_configuration& config) :
    ctx(std::move(ctx_)), window_offset(config.nof_samples_window_offset)
  {
    cfg.numerology     = config.numerology;
    cfg.bw_rb          = config.bw_rb;
    cfg.dft_size       = config.dft_size;
    cfg.cp             = (config.cp == srsran::cyclic_prefix::NORMAL) ? 0 : 1;
    cfg.scale          = config.scale;
    cfg.center_freq_hz = config.center_freq_hz;
    int rc             = nrphy_ofdm_plan_create(ctx->get(), &cfg, 1, &plan);
    report_failure("nrphy_ofdm_plan_create", rc);
    staging.resize(static_cast<size_t>(NRPHY_NSYMB) * cfg.bw_rb * srsran::NRE);
  }
  ~ofdm_demodulator_adaptor_base() { nrphy_ofdm_plan_destroy(plan); }

  std::shared_ptr<context>     ctx;
  nrphy_ofdm_config_t          cfg;
  nrphy_ofdm_plan_t*           plan = nullptr;
  unsigned                     window_offset;
  std::vector<srsran::cbf16_t> staging;
};

class ofdm_symbol_demodulator_adaptor : public srsran::ofdm_symbol_demodulator, private ofdm_demodulator_adaptor_base
{
public:
  ofdm_symbol_demodulator_adaptor(std::shared_ptr<context> ctx_, const srsran::ofdm_demodulator_configuration& config) :
    ofdm_demodulator_adaptor_base(std::move(ctx_), config)
  {
  }
  unsigned get_symbol_size(unsigned symbol_index) const override { return nrphy_ofdm_symbol_size(&cfg, symbol_index); }
  void demodulate(srsran::resource_grid_writer& grid, srsran::span<const srsran::cf_t> input, unsigned port_index, unsigned symbol_index) override
  {
    using namespace srsran;
    unsigned nof_subc = cfg.bw_rb * NRE;
    int      rc       = nrphy_ofdm_demodulate_symbol_host(plan, reinterpret_cast<const float*>(input.data()), input.size(),
                                               symbol_index, window_offset, staging.data());
    report_failure("nrphy_ofdm_demodulate_symbol_host", rc);
    grid.put(port_index, symbol_index % (cfg.cp ? 12 : 14), 0, 1, span<const cbf16_t>(staging.data(), nof_subc));
  }
};

class ofdm_slot_demodulator_adaptor : public srsran::ofdm_slot_demodulator, private ofdm_demodulator_adaptor_base
{
public:
  ofdm_slot_demodulator_adaptor(std::shared_ptr<context> ctx_, const srsran::ofdm_demodulator_configuration& config) :
    ofdm_demodulator_adaptor_base(std::move(ctx_), config)
  {
  }
  unsigned get_slot_size(unsigned slot_index) const override { return nrphy_ofdm_slot_size(&cfg, slot_index); }
  void demodulate(srsran::resource_grid_writer& grid, srsran::span<const srsran::cf_t> input, unsigned port_index, unsigned slot_index) override
  {
    using namespace srsran;
    srsran_assert(input.size() == get_slot_size(slot_index), "Invalid input size.");
    unsigned nof_subc = cfg.bw_rb * NRE;
    int rc = nrphy_ofdm_demodulate_slot_host(plan, reinterpret_cast<const float*>(input.data()), slot_index, window_offset, staging.data());
    report_failure("nrphy_ofdm_demodulate_slot_host", rc);
    for (unsigned l = 0, nsymb = cfg.cp ? 12 : 14; l != nsymb; ++l) {
      grid.put(port_index, l, 0, 1, span<const cbf16_t>(&staging[static_cast<size_t>(l) * nof_subc], nof_subc));
    }
  }
};

class ofdm_demodulator_factory_adaptor : public srsran::ofdm_demodulator_factory
{
public:
  explicit ofdm_demodulator_factory_adaptor(std::shared_ptr<context> ctx_) : ctx(std::move(ctx_)) {}
  std::unique_ptr<srsran::ofdm_symbol_demodulator> create_ofdm_symbol_demodulator(const srsran::ofdm_demodulator_configuration& config) override
  {
    return std::make_unique<ofdm_symbol_demodulator_adaptor>(ctx, config);
  }
  std::unique_ptr<srsran::ofdm_slot_demodulator> create_ofdm_slot_demodulator(const srsran::ofdm_demodulator_configuration& config) override
  {
    return std::make_unique<ofdm_slot_demodulator_adaptor>(ctx, config);
  }

private:
  std::shared_ptr<context> ctx;
};

/// dft_processor over nrphy_dft_run_host (owns its input/output buffers like the reference's implementations).
class dft_processor_adaptor : public srsran::dft_processor
{
public:
  dft_processor_adaptor(std::shared_ptr<context> ctx_, const configuration& config) :
    ctx(std::move(ctx_)), dir(config.dir), input(config.size), output(config.size)
  {
  }
  direction                  get_direction() const override { return dir; }
  unsigned                   get_size() const override { return input.size(); }
  srsran::span<srsran::cf_t> get_input() override { return input; }
  srsran::span<const srsran::cf_t> run() override
  {
    int rc = nrphy_dft_run_host(ctx->get(), input.size(), dir == direction::INVERSE, reinterpret_cast<const float*>(input.data()),
                                reinterpret_cast<float*>(output.data()));
    report_failure("nrphy_dft_run_host", rc);
    return output;
  }

private:
  std::shared_ptr<context>  ctx;
  direction                 dir;
  std::vector<srsran::cf_t> input;
  std::vector<srsran::cf_t> output;
};

class dft_processor_factory_adaptor : public srsran::dft_processor_factory
{
public:
  explicit dft_processor_factory_adaptor(std::shared_ptr<context> ctx_) : ctx(std::move(ctx_)) {}
  std::unique_ptr<srsran::dft_processor> create(const srsran::dft_processor::configuration& config) override
  {
    switch (config.size) { // the sizes of dft_processor_generic_impl.cpp:190-208
      case 128:
      case 256:
      case 384:
      case 512:
      case 768:
      case 1024:
      case 1536:
      case 2048:
      case 3072:
      case 4096:
      case 4608:
      case 6144:
      case 9216:
      case 12288:
      case 18432:
      case 24576:
      case 36864:
      case 49152:
        return std::make_unique<dft_processor_adaptor>(ctx, config);
      default:
        return nullptr;
    }
  }

private:
  std::shared_ptr<context> ctx;
};

// ---- receive side ("next" row): LDPC rate dematcher and decoder ----------------------------------------------------
// Drop-ins for create_ldpc_rate_dematcher_factory_sw / create_ldpc_decoder_factory_sw
// (R/include/srsran/phy/upper/channel_coding/channel_coding_factories.h:52-77): one codeblock per call on host spans,
// the calling convention of pusch_codeblock_decoder (R/lib/phy/upper/channel_processors/pusch/pusch_codeblock_decoder.cpp).
// A device-resident receive chain calls nrphy_ldpc_rate_dematch / nrphy_ldpc_decode on whole batches instead.
class ldpc_rate_dematcher_adaptor : public srsran::ldpc_rate_dematcher
{
public:
  explicit ldpc_rate_dematcher_adaptor(std::shared_ptr<context> ctx_) : ctx(std::move(ctx_)) {}
  void rate_dematch(srsran::span<srsran::log_likelihood_ratio>       output,
                    srsran::span<const srsran::log_likelihood_ratio> input,
                    bool                                             new_data,
                    const srsran::codeblock_metadata&                cfg) override
  {
    using namespace srsran;
    // As the reference, the base graph and the lifting size follow from the soft-buffer length (66 Zc or 50 Zc).
    nrphy_ldpc_rate_dematcher_cfg_t c = {};
    c.base_graph                      = (output.size() % 66 == 0) ? 1 : 2;
    c.lifting_size                    = output.size() / ((c.base_graph == 1) ? 66 : 50);
    c.rv                              = cfg.tb_common.rv;
    c.qm                              = get_bits_per_symbol(cfg.tb_common.mod);
    c.nref                            = cfg.tb_common.Nref;
    c.nof_filler_bits                 = cfg.cb_specific.nof_filler_bits;
    c.rm_length                       = input.size();
    static_assert(sizeof(log_likelihood_ratio) == sizeof(int8_t), "LLRs are plain int8");
    int rc = nrphy_ldpc_rate_dematch_host(ctx->get(), &c, reinterpret_cast<const int8_t*>(input.data()),
                                          reinterpret_cast<int8_t*>(output.data()), new_data ? 1 : 0);
    report_failure("nrphy_ldpc_rate_dematch_host", rc);
  }

private:
  std::shared_ptr<context> ctx;
};

class ldpc_rate_dematcher_factory_adaptor : public srsran::ldpc_rate_dematcher_factory
{
public:
  explicit ldpc_rate_dematcher_factory_adaptor(std::shared_ptr<context> ctx_) : ctx(std::move(ctx_)) {}
  std::unique_ptr<srsran::ldpc_rate_dematcher> create() override
  {
    return std::make_unique<ldpc_rate_dematcher_adaptor>(ctx);
  }

private:
  std::shared_ptr<context> ctx;
};

class ldpc_decoder_adaptor : public srsran::ldpc_decoder
{
public:
  explicit ldpc_decoder_adaptor(std::shared_ptr<context> ctx_) : ctx(std::move(ctx_)) {}
  std::optional<unsigned> decode(srsran::bit_buffer&                              output,
                                 srsran::span<const srsran::log_likelihood_ratio> input,
                                 srsran::crc_calculator*                          crc,
                                 const configuration&                             cfg) override
  {
    using namespace srsran;
    nrphy_ldpc_decoder_cfg_t c = {};
    c.base_graph               = (cfg.block_conf.tb_common.base_graph == ldpc_base_graph_type::BG1) ? 1 : 2;
    c.lifting_size             = cfg.block_conf.tb_common.lifting_size;
    c.nof_filler_bits          = cfg.block_conf.cb_specific.nof_filler_bits;
    c.nof_llr                  = input.size();
    c.max_iterations           = cfg.algorithm_conf.max_iterations;
    c.scaling_factor           = cfg.algorithm_conf.scaling_factor;
    c.crc_poly                 = 0;
    if (crc != nullptr) {
      switch (crc->get_generator_poly()) {
        case crc_generator_poly::CRC16:
          c.crc_poly = 16;
          break;
        case crc_generator_poly::CRC24A:
          c.crc_poly = 0x24A;
          break;
        case crc_generator_poly::CRC24B:
          c.crc_poly = 0x24B;
          break;
        default:
          srsran_assert(false, "CRC polynomial not used for LDPC codeblocks.");
      }
    }
    const unsigned k = ((c.base_graph == 1) ? 22 : 10) * c.lifting_size;
    packed.resize((k + 7) / 8);
    uint32_t iterations = 0;
    int rc = nrphy_ldpc_decode_host(ctx->get(), &c, reinterpret_cast<const int8_t*>(input.data()), packed.data(), &iterations);
    report_failure("nrphy_ldpc_decode_host", rc);
    if (rc != NRPHY_OK) {
      return std::nullopt; // reported like a codeblock that did not converge
    }
    // output holds the message without (or with) its filler bits: the first output.size() hard bits
    const unsigned nbits = std::min<unsigned>(output.size(), k);
    for (unsigned i = 0; i < nbits; i += 8) {
      unsigned n = std::min(8U, nbits - i);
      output.insert(static_cast<uint8_t>(packed[i / 8] >> (8 - n)), i, n);
    }
    if (crc != nullptr && iterations != 0) {
      return iterations;
    }
    return std::nullopt;
  }

private:
  std::shared_ptr<context> ctx;
  std::vector<uint8_t>     packed;
};

class ldpc_decoder_factory_adaptor : public srsran::ldpc_decoder_factory
{
public:
  explicit ldpc_decoder_factory_adaptor(std::shared_ptr<context> ctx_) : ctx(std::move(ctx_)) {}
  std::unique_ptr<srsran::ldpc_decoder> create() override { return std::make_unique<ldpc_decoder_adaptor>(ctx); }

private:
  std::shared_ptr<context> ctx;
};

// hal::hw_accelerator_pusch_dec in codeblock mode (hw_accelerator_pusch_dec.h:36-110): the drop-in for the bbdev
// accelerator that pusch_decoder_hw_impl drives (pusch_decoder_hw_impl.cpp:180-345).  An operation = rate dematching
// of one codeblock into its HARQ buffer + LDPC decoding.  The soft buffers live with the caller
// (is_external_harq_supported() == false): enqueue receives the previous soft bits, dequeue returns the updated ones.
class hw_accelerator_pusch_dec_adaptor : public srsran::hal::hw_accelerator_pusch_dec
{
public:
  explicit hw_accelerator_pusch_dec_adaptor(std::shared_ptr<context> ctx_) : ctx(std::move(ctx_)) {}

  void reserve_queue() override {}
  void free_queue() override {}
  void free_harq_context_entry(unsigned absolute_cb_id) override { (void)absolute_cb_id; }
  bool is_external_harq_supported() const override { return false; }

  void configure_operation(const srsran::hal::hw_pusch_decoder_configuration& config, unsigned cb_index = 0) override
  {
    slot(cb_index).cfg = config;
  }

  bool enqueue_operation(srsran::span<const int8_t> data, srsran::span<const int8_t> soft_data = {}, unsigned cb_index = 0) override
  {
    using namespace srsran;
    operation& op = slot(cb_index);
    if (op.pending) {
      return false;
    }
    const hal::hw_pusch_decoder_configuration& cfg = op.cfg;
    nrphy_ldpc_rate_dematcher_cfg_t            dm  = {};
    dm.base_graph      = (cfg.base_graph_index == ldpc_base_graph_type::BG1) ? 1 : 2;
    dm.lifting_size    = cfg.lifting_size;
    dm.rv              = cfg.rv;
    dm.qm              = get_bits_per_symbol(cfg.modulation);
    dm.nref            = cfg.Nref;
    dm.nof_filler_bits = cfg.nof_filler_bits;
    dm.rm_length       = data.size();
    const unsigned n = ((dm.base_graph == 1) ? 66 : 50) * dm.lifting_size, k = ((dm.base_graph == 1) ? 22 : 10) * dm.lifting_size;
    op.soft.assign(n, 0);
    if (!soft_data.empty()) {
      std::memcpy(op.soft.data(), soft_data.data(), std::min<size_t>(n, soft_data.size()));
    }
    op.message.assign((k + 7) / 8, 0);
    uint32_t crc_poly = 0;
    if (cfg.use_early_stop) {
      crc_poly = (cfg.cb_crc_type == hal::hw_dec_cb_crc_type::CRC16) ? 16 : (cfg.cb_crc_type == hal::hw_dec_cb_crc_type::CRC24A) ? 0x24A : 0x24B;
    }
    uint32_t iterations = 0;
    int      rc = nrphy_pusch_decode_codeblock_host(ctx->get(), &dm, crc_poly, cfg.max_nof_ldpc_iterations, 0.8F, data.data(),
                                               op.soft.data(), cfg.new_data ? 1 : 0, op.message.data(), &iterations);
    report_failure("nrphy_pusch_decode_codeblock_host", rc);
    if (rc != NRPHY_OK) {
      iterations = 0; // reported like a codeblock whose CRC failed
      std::fill(op.message.begin(), op.message.end(), 0xFF);
    }
    op.out.nof_ldpc_iterations = (iterations != 0) ? iterations : cfg.max_nof_ldpc_iterations;
    // Without early stop the decoder did not look at the CRC: divide the payload + CRC by the generator here.
    op.out.CRC_pass = (iterations != 0) || (rc == NRPHY_OK && !cfg.use_early_stop && crc_is_zero(op.message, cfg));
    op.pending      = true;
    return true;
  }

  bool dequeue_operation(srsran::span<uint8_t> data, srsran::span<int8_t> soft_data = {}, unsigned segment_index = 0) override
  {
    operation& op = slot(segment_index);
    if (!op.pending) {
      return false;
    }
    std::memcpy(data.data(), op.message.data(), std::min<size_t>(data.size(), op.message.size()));
    if (!soft_data.empty()) {
      std::memcpy(soft_data.data(), op.soft.data(), std::min<size_t>(soft_data.size(), op.soft.size()));
    }
    op.pending = false;
    return true;
  }

  void read_operation_outputs(srsran::hal::hw_pusch_decoder_outputs& out, unsigned cb_index = 0, unsigned absolute_cb_id = 0) override
  {
    (void)absolute_cb_id;
    out = slot(cb_index).out;
  }

private:
  struct operation {
    srsran::hal::hw_pusch_decoder_configuration cfg = {};
    srsran::hal::hw_pusch_decoder_outputs       out = {};
    std::vector<int8_t>                         soft;
    std::vector<uint8_t>                        message;
    bool                                        pending = false;
  };
  operation& slot(unsigned cb_index)
  {
    if (ops.size() <= cb_index) {
      ops.resize(cb_index + 1);
    }
    return ops[cb_index];
  }
  // Remainder of the message without its filler bits modulo the codeblock CRC polynomial (the early-stop criterion
  // of ldpc_decoder_impl.cpp:118-126).
  static bool crc_is_zero(const std::vector<uint8_t>& message, const srsran::hal::hw_pusch_decoder_configuration& cfg)
  {
    using namespace srsran;
    const unsigned order = (cfg.cb_crc_type == hal::hw_dec_cb_crc_type::CRC16) ? 16 : 24;
    const uint32_t poly  = (cfg.cb_crc_type == hal::hw_dec_cb_crc_type::CRC16) ? 0x11021U : (cfg.cb_crc_type == hal::hw_dec_cb_crc_type::CRC24A) ? 0x1864CFBU : 0x1800063U;
    const unsigned nbits = ((cfg.base_graph_index == ldpc_base_graph_type::BG1) ? 22 : 10) * cfg.lifting_size - cfg.nof_filler_bits;
    uint32_t       reg   = 0;
    for (unsigned i = 0; i != nbits && i / 8 < message.size(); ++i) {
      reg = (reg << 1) | ((message[i / 8] >> (7 - i % 8)) & 1U);
      if (reg & (1U << order)) {
        reg ^= poly;
      }
    }
    return reg == 0;
  }

  std::shared_ptr<context> ctx;
  std::vector<operation>   ops;
};

class hw_accelerator_pusch_dec_factory_adaptor : public srsran::hal::hw_accelerator_pusch_dec_factory
{
public:
  explicit hw_accelerator_pusch_dec_factory_adaptor(std::shared_ptr<context> ctx_) : ctx(std::move(ctx_)) {}
  std::unique_ptr<srsran::hal::hw_accelerator_pusch_dec> create() override
  {
    return std::make_unique<hw_accelerator_pusch_dec_adaptor>(ctx);
  }

private:
  std::shared_ptr<context> ctx;
};

/// nzp_csi_rs_generator::config_t -> POD.  \c weights receives the precoding coefficients the POD points to.
inline nrphy_csi_rs_cfg_t to_pod(const srsran::nzp_csi_rs_generator::config_t& config, std::vector<float>& weights)
{
  using namespace srsran;
  weights.assign(2 * config.precoding.get_nof_ports() * config.precoding.get_nof_layers(), 0.0F);
  nrphy_csi_rs_cfg_t c = {};
  c.slot_index         = config.slot.slot_index();
  c.cp                 = (config.cp == cyclic_prefix::NORMAL) ? 0 : 1;
  c.start_rb           = config.start_rb;
  c.nof_rb             = config.nof_rb;
  c.row                = config.csi_rs_mapping_table_row;
  c.nof_k_ref          = config.freq_allocation_ref_idx.size();
  for (unsigned i = 0; i != c.nof_k_ref && i != 6; ++i) {
    c.k_ref[i] = config.freq_allocation_ref_idx[i];
  }
  c.symbol_l0     = config.symbol_l0;
  c.symbol_l1     = config.symbol_l1;
  c.cdm           = static_cast<uint32_t>(config.cdm);
  c.density       = static_cast<uint32_t>(config.freq_density);
  c.scrambling_id = config.scrambling_id;
  c.amplitude     = config.amplitude;
  c.nof_ports     = config.precoding.get_nof_ports();
  c.prg_size_rb   = config.precoding.get_prg_size();
  c.nof_prg       = config.precoding.get_nof_prg();
  for (unsigned port = 0; port != c.nof_ports; ++port) {
    for (unsigned layer = 0; layer != config.precoding.get_nof_layers(); ++layer) {
      cf_t w                                                              = config.precoding.get_coefficient(layer, port, 0);
      weights[2 * (port * config.precoding.get_nof_layers() + layer)]     = w.real();
      weights[2 * (port * config.precoding.get_nof_layers() + layer) + 1] = w.imag();
    }
  }
  c.precoding = weights.data();
  return c;
}

// ---- other downlink grid writers ("next" row): NZP-CSI-RS generator -------------------------------------------------
// Drop-in for create_nzp_csi_rs_generator_factory_sw (R/include/srsran/phy/upper/signal_processors/
// signal_processor_factories.h) for rows 1-5 of TS 38.211 Table 7.4.1.5.3-1.  Host-span form: the signal is computed
// into a staging grid and its resource elements are put into the caller's grid through the writer, every CDM group on
// all ports, as the reference's mapper does.  A device-resident L1 calls nrphy_csi_rs_map on its device grids instead.
class nzp_csi_rs_generator_adaptor : public srsran::nzp_csi_rs_generator
{
public:
  nzp_csi_rs_generator_adaptor(std::shared_ptr<context> ctx_, unsigned nof_ports_, unsigned nof_subc_) :
    ctx(std::move(ctx_)), nof_ports(nof_ports_), nof_subc(nof_subc_),
    staging(static_cast<size_t>(nof_ports_) * NRPHY_NSYMB * nof_subc_)
  {
  }

  void map(srsran::resource_grid_mapper& mapper, const config_t& config) override
  {
    using namespace srsran;
    std::vector<float> weights;
    nrphy_csi_rs_cfg_t c = to_pod(config, weights);
    srsran_assert(nrphy_csi_rs_validate(&c) == NRPHY_OK, "CSI-RS configuration outside rows 1-5 / wideband precoding.");
    if (device_resource_grid* dg = device_resource_grid::from(mapper)) {
      int rc = dg->on_device([&](nrphy_dl_slots_t* p, uint32_t id) { return nrphy_dl_slot_csi_rs(p, id, 1, &c); });
      if (rc != NRPHY_ERR_CAPACITY) {
        report_failure("nrphy_dl_slot_csi_rs", rc);
        return;
      }
    }
    std::fill(staging.begin(), staging.end(), cbf16_t());
    int rc = nrphy_csi_rs_map_host(ctx->get(), &c, staging.data(), nof_ports, nof_subc);
    report_failure("nrphy_csi_rs_map_host", rc);

    // The RE of the signal: the union of the per-port patterns, every one of them written on all precoding ports.
    csi_rs_pattern_configuration pc;
    pc.start_rb                 = config.start_rb;
    pc.nof_rb                   = config.nof_rb;
    pc.csi_rs_mapping_table_row = config.csi_rs_mapping_table_row;
    pc.freq_allocation_ref_idx  = config.freq_allocation_ref_idx;
    pc.symbol_l0                = config.symbol_l0;
    pc.symbol_l1                = config.symbol_l1;
    pc.cdm                      = config.cdm;
    pc.freq_density             = config.freq_density;
    csi_rs_pattern  pattern = get_csi_rs_pattern(pc);
    re_symbol_masks masks;
    for (unsigned l = 0; l != NRPHY_NSYMB; ++l) {
      masks[l].resize(nof_subc);
      masks[l].reset();
      if (l >= get_nsymb_per_slot(config.cp)) {
        continue;
      }
      for (const csi_rs_pattern_port& port_pattern : pattern.prb_patterns) {
        if (!port_pattern.symbol_mask.test(l)) {
          continue;
        }
        for (unsigned prb = pattern.rb_begin; prb < pattern.rb_end; prb += pattern.rb_stride) {
          for (unsigned k = 0; k != NRE; ++k) {
            if (port_pattern.re_mask.test(k)) {
              masks[l].set(NRE * prb + k);
            }
          }
        }
      }
    }
    put_staging_grid(mapper, staging.data(), c.nof_ports, nof_subc, masks);
  }

private:
  std::shared_ptr<context>     ctx;
  unsigned                     nof_ports;
  unsigned                     nof_subc;
  std::vector<srsran::cbf16_t> staging;
};

// ---- other downlink grid writers: PDCCH and SS/PBCH block processors --------------------------------------------------
/// pdcch_processor::pdu_t -> POD.  \c weights receives the precoding coefficients the POD points to.
inline nrphy_pdcch_pdu_t to_pod(const srsran::pdcch_processor::pdu_t& pdu, std::vector<float>& weights)
{
  using namespace srsran;
  nrphy_pdcch_pdu_t p;
  std::memset(&p, 0, sizeof(p));
  p.slot_index         = pdu.slot.slot_index();
  p.cp                 = (pdu.cp == cyclic_prefix::NORMAL) ? 0 : 1;
  p.bwp_size_rb        = pdu.coreset.bwp_size_rb;
  p.bwp_start_rb       = pdu.coreset.bwp_start_rb;
  p.start_symbol_index = pdu.coreset.start_symbol_index;
  p.duration           = pdu.coreset.duration;
  for (unsigned i = 0; i != pdu.coreset.frequency_resources.size(); ++i) {
    p.frequency_resources |= pdu.coreset.frequency_resources.test(i) ? (uint64_t(1) << i) : 0;
  }
  p.cce_to_reg_mapping   = static_cast<uint32_t>(pdu.coreset.cce_to_reg_mapping);
  p.reg_bundle_size      = pdu.coreset.reg_bundle_size;
  p.interleaver_size     = pdu.coreset.interleaver_size;
  p.shift_index          = pdu.coreset.shift_index;
  p.rnti                 = pdu.dci.rnti;
  p.n_id_pdcch_dmrs      = pdu.dci.n_id_pdcch_dmrs;
  p.n_id_pdcch_data      = pdu.dci.n_id_pdcch_data;
  p.n_rnti               = pdu.dci.n_rnti;
  p.cce_index            = pdu.dci.cce_index;
  p.aggregation_level    = pdu.dci.aggregation_level;
  p.dmrs_power_offset_dB = pdu.dci.dmrs_power_offset_dB;
  p.data_power_offset_dB = pdu.dci.data_power_offset_dB;
  p.payload_size         = pdu.dci.payload.size();
  for (unsigned i = 0; i != p.payload_size && i != NRPHY_PDCCH_MAX_PAYLOAD; ++i) {
    p.payload[i] = pdu.dci.payload[i];
  }
  p.nof_ports   = pdu.dci.precoding.get_nof_ports();
  p.prg_size_rb = pdu.dci.precoding.get_prg_size();
  p.nof_prg     = pdu.dci.precoding.get_nof_prg();
  weights.resize(2 * p.nof_prg * p.nof_ports);
  for (unsigned g = 0; g != p.nof_prg; ++g) {
    for (unsigned port = 0; port != p.nof_ports; ++port) {
      cf_t w                                    = pdu.dci.precoding.get_coefficient(0, port, g);
      weights[2 * (g * p.nof_ports + port)]     = w.real();
      weights[2 * (g * p.nof_ports + port) + 1] = w.imag();
    }
  }
  p.precoding = weights.data();
  return p;
}

/// pdcch_processor over nrphy_pdcch_process_host (host-span form; a device-resident L1 calls nrphy_pdcch_process on
/// its device grids).  The candidate's RE are found in the staging grid by the words the kernel wrote.
class pdcch_processor_adaptor : public srsran::pdcch_processor
{
public:
  pdcch_processor_adaptor(std::shared_ptr<context> ctx_, unsigned nof_ports_, unsigned nof_subc_) :
    ctx(std::move(ctx_)), nof_ports(nof_ports_), nof_subc(nof_subc_), staging(static_cast<size_t>(nof_ports_) * NRPHY_NSYMB * nof_subc_)
  {
  }
  void process(srsran::resource_grid_mapper& mapper, const pdu_t& pdu) override
  {
    using namespace srsran;
    std::vector<float> weights;
    nrphy_pdcch_pdu_t  pod = to_pod(pdu, weights);
    srsran_assert(nrphy_pdcch_validate(&pod) == NRPHY_OK, "Invalid PDCCH PDU.");
    if (device_resource_grid* dg = device_resource_grid::from(mapper)) {
      int rc = dg->on_device([&](nrphy_dl_slots_t* p, uint32_t id) { return nrphy_dl_slot_pdcch(p, id, 1, &pod); });
      if (rc != NRPHY_ERR_CAPACITY) {
        report_failure("nrphy_dl_slot_pdcch", rc);
        return;
      }
    }
    // A marker no precoder output equals (a NaN pattern) tells the candidate's RE from the rest of the staging grid.
    cbf16_t marker;
    const uint32_t marker_bits = 0x7FC17FC1U;
    std::memcpy(&marker, &marker_bits, sizeof(marker));
    std::fill(staging.begin(), staging.end(), marker);
    int rc = nrphy_pdcch_process_host(ctx->get(), &pod, staging.data(), nof_ports, nof_subc);
    report_failure("nrphy_pdcch_process_host", rc);
    if (rc != NRPHY_OK) {
      return;
    }
    re_symbol_masks masks;
    for (unsigned l = 0; l != NRPHY_NSYMB; ++l) {
      masks[l].resize(nof_subc);
      masks[l].reset();
      const cbf16_t* row = &staging[static_cast<size_t>(l) * nof_subc]; // port 0 carries every RE of the candidate
      for (unsigned k = 0; k != nof_subc; ++k) {
        if (std::memcmp(&row[k], &marker, sizeof(marker)) != 0) {
          masks[l].set(k);
        }
      }
    }
    put_staging_grid(mapper, staging.data(), pod.nof_ports, nof_subc, masks);
  }

private:
  std::shared_ptr<context>     ctx;
  unsigned                     nof_ports;
  unsigned                     nof_subc;
  std::vector<srsran::cbf16_t> staging;
};

class pdcch_pdu_validator_adaptor : public srsran::pdcch_pdu_validator
{
public:
  bool is_valid(const srsran::pdcch_processor::pdu_t& pdu) const override
  {
    std::vector<float> weights;
    nrphy_pdcch_pdu_t  pod = to_pod(pdu, weights);
    return nrphy_pdcch_validate(&pod) == NRPHY_OK;
  }
};

class pdcch_processor_factory_adaptor : public srsran::pdcch_processor_factory
{
public:
  pdcch_processor_factory_adaptor(std::shared_ptr<context> ctx_, unsigned nof_ports_, unsigned nof_subc_) :
    ctx(std::move(ctx_)), nof_ports(nof_ports_), nof_subc(nof_subc_)
  {
  }
  std::unique_ptr<srsran::pdcch_processor>     create() override { return std::make_unique<pdcch_processor_adaptor>(ctx, nof_ports, nof_subc); }
  std::unique_ptr<srsran::pdcch_pdu_validator> create_validator() override { return std::make_unique<pdcch_pdu_validator_adaptor>(); }

private:
  std::shared_ptr<context> ctx;
  unsigned                 nof_ports;
  unsigned                 nof_subc;
};

inline nrphy_ssb_pdu_t to_pod(const srsran::ssb_processor::pdu_t& pdu)
{
  using namespace srsran;
  nrphy_ssb_pdu_t p;
  std::memset(&p, 0, sizeof(p));
  p.numerology        = pdu.slot.numerology();
  p.sfn               = pdu.slot.sfn();
  p.slot_index        = pdu.slot.slot_index();
  p.phys_cell_id      = pdu.phys_cell_id;
  p.beta_pss_dB       = pdu.beta_pss;
  p.ssb_idx           = pdu.ssb_idx;
  p.L_max             = pdu.L_max;
  p.common_scs        = to_numerology_value(pdu.common_scs);
  p.subcarrier_offset = pdu.subcarrier_offset.to_uint();
  p.offset_to_pointA  = pdu.offset_to_pointA.to_uint();
  p.pattern_case      = static_cast<uint32_t>(pdu.pattern_case);
  for (unsigned i = 0; i != 32; ++i) {
    p.bch_payload[i] = pdu.bch_payload[i];
  }
  p.nof_ports = pdu.ports.size();
  for (unsigned i = 0; i != p.nof_ports && i != NRPHY_MAX_PORTS; ++i) {
    p.ports[i] = pdu.ports[i];
  }
  return p;
}

/// ssb_processor over nrphy_ssb_process_host: the block's 4 symbols x 240 subcarriers are computed into a staging grid
/// and written through the grid's writer (PBCH, its DM-RS, PSS, SSS -- the zeros around PSS / SSS are not written, as
/// in the reference).
class ssb_processor_adaptor : public srsran::ssb_processor
{
public:
  ssb_processor_adaptor(std::shared_ptr<context> ctx_, unsigned nof_ports_, unsigned nof_subc_) :
    ctx(std::move(ctx_)), nof_ports(nof_ports_), nof_subc(nof_subc_), staging(static_cast<size_t>(nof_ports_) * NRPHY_NSYMB * nof_subc_)
  {
  }
  void process(srsran::resource_grid_writer& grid, const pdu_t& pdu) override
  {
    using namespace srsran;
    nrphy_ssb_pdu_t pod = to_pod(pdu);
    srsran_assert(nrphy_ssb_validate(&pod) == NRPHY_OK, "Invalid SS/PBCH block PDU.");
    if (device_resource_grid* dg = device_resource_grid::from(grid)) {
      int rc = dg->on_device([&](nrphy_dl_slots_t* p, uint32_t id) { return nrphy_dl_slot_ssb(p, id, 1, &pod); });
      if (rc != NRPHY_ERR_CAPACITY) {
        report_failure("nrphy_dl_slot_ssb", rc);
        return;
      }
    }
    cbf16_t        marker;
    const uint32_t marker_bits = 0x7FC17FC1U;
    std::memcpy(&marker, &marker_bits, sizeof(marker));
    std::fill(staging.begin(), staging.end(), marker);
    int rc = nrphy_ssb_process_host(ctx->get(), &pod, staging.data(), nof_ports, nof_subc);
    report_failure("nrphy_ssb_process_host", rc);
    if (rc != NRPHY_OK) {
      return;
    }
    std::vector<cbf16_t> packed(nof_subc);
    for (unsigned port : pdu.ports) {
      for (unsigned l = 0; l != NRPHY_NSYMB; ++l) {
        const cbf16_t*               row = &staging[(static_cast<size_t>(port) * NRPHY_NSYMB + l) * nof_subc];
        bounded_bitset<MAX_RB * NRE> mask(nof_subc);
        unsigned                     n = 0;
        for (unsigned k = 0; k != nof_subc; ++k) {
          if (std::memcmp(&row[k], &marker, sizeof(marker)) != 0) {
            mask.set(k);
            packed[n++] = row[k];
          }
        }
        if (n != 0) {
          grid.put(port, l, 0, mask, span<const cbf16_t>(packed).first(n));
        }
      }
    }
  }

private:
  std::shared_ptr<context>     ctx;
  unsigned                     nof_ports;
  unsigned                     nof_subc;
  std::vector<srsran::cbf16_t> staging;
};

// ---- lower-PHY tail: amplitude controller, Open Fronthaul compressor ---------------------------------------------------
/// amplitude_controller over nrphy_amplitude_control_host (one buffer per call, as the lower PHY calls it per port).
class amplitude_controller_adaptor : public srsran::amplitude_controller
{
public:
  amplitude_controller_adaptor(std::shared_ptr<context> ctx_, const nrphy_amplitude_cfg_t& cfg_) : ctx(std::move(ctx_)), cfg(cfg_)
  {
    std::memset(&metrics, 0, sizeof(metrics));
  }
  srsran::amplitude_controller_metrics process(srsran::span<srsran::cf_t> output, srsran::span<const srsran::cf_t> input) override
  {
    srsran_srsvec_assert_size(output, input);
    int rc = nrphy_amplitude_control_host(ctx->get(), &cfg, reinterpret_cast<const float*>(input.data()), input.size(),
                                          reinterpret_cast<float*>(output.data()), &metrics);
    report_failure("nrphy_amplitude_control_host", rc);
    if (rc != NRPHY_OK) {
      std::fill(output.begin(), output.end(), srsran::cf_t()); // silence rather than an unbounded signal
    }
    return {metrics.avg_power_fs, metrics.peak_power_fs, metrics.papr_lin, metrics.gain_dB, metrics.nof_processed_samples,
            metrics.nof_clipped_samples, static_cast<long double>(metrics.clipping_probability), metrics.clipping_enabled != 0};
  }

private:
  std::shared_ptr<context>  ctx;
  nrphy_amplitude_cfg_t     cfg;
  nrphy_amplitude_metrics_t metrics;
};

/// ofh::iq_compressor over nrphy_ofh_compress_host: the serialised records come back and are unpacked into the
/// reference's compressed_prb objects (the device-resident form, nrphy_ofh_compress, writes the user-plane payload
/// itself).
class iq_compressor_adaptor : public srsran::ofh::iq_compressor
{
public:
  iq_compressor_adaptor(std::shared_ptr<context> ctx_, float iq_scaling_) : ctx(std::move(ctx_)), iq_scaling(iq_scaling_) {}
  void compress(srsran::span<srsran::ofh::compressed_prb> compressed_prbs,
                srsran::span<const srsran::cbf16_t>       iq_data,
                const srsran::ofh::ru_compression_params& params) override
  {
    using namespace srsran;
    nrphy_ofh_compression_cfg_t cfg = {params.type == ofh::compression_type::BFP ? 1U : 0U, params.data_width, iq_scaling};
    const unsigned              rec = nrphy_ofh_compressed_prb_bytes(&cfg), nof_prb = compressed_prbs.size();
    packed.assign(static_cast<size_t>(rec) * nof_prb, 0);
    int rc = nrphy_ofh_compress_host(ctx->get(), &cfg, nof_prb, iq_data.data(), packed.data());
    report_failure("nrphy_ofh_compress_host", rc); // on failure the PRBs go out as zeros
    for (unsigned i = 0; i != nof_prb; ++i) {
      const uint8_t* r = &packed[static_cast<size_t>(i) * rec];
      if (cfg.type == 1) {
        compressed_prbs[i].set_compression_param(*r++);
      }
      std::memcpy(compressed_prbs[i].get_byte_buffer().data(), r, 3 * params.data_width);
      compressed_prbs[i].set_stored_size(3 * params.data_width);
    }
  }

private:
  std::shared_ptr<context> ctx;
  float                    iq_scaling;
  std::vector<uint8_t>     packed;
};

// ---- soft demodulator ---------------------------------------------------------------------------------------------------
/// demodulation_mapper over nrphy_demodulate_soft_host (one span per call, as the reference's callers use it:
/// pusch_demodulator_impl.cpp:235-245, one OFDM symbol's worth of equalised resource elements at a time).  On failure the
/// soft bits are zeros (undecided), which the decoder then reports as a failed CRC.
class demodulation_mapper_adaptor : public srsran::demodulation_mapper
{
public:
  explicit demodulation_mapper_adaptor(std::shared_ptr<context> ctx_) : ctx(std::move(ctx_)) {}
  void demodulate_soft(srsran::span<srsran::log_likelihood_ratio> llrs,
                       srsran::span<const srsran::cf_t>           symbols,
                       srsran::span<const float>                  noise_vars,
                       srsran::modulation_scheme                  mod) override
  {
    using namespace srsran;
    srsran_assert(symbols.size() == noise_vars.size(), "Inputs symbols and noise_vars must have the same length.");
    srsran_assert(symbols.size() * get_bits_per_symbol(mod) == llrs.size(), "Input and output lengths are incompatible.");
    const uint32_t m = mod == modulation_scheme::PI_2_BPSK ? NRPHY_MOD_PI2_BPSK : get_bits_per_symbol(mod);
    int rc = nrphy_demodulate_soft_host(ctx->get(), m, symbols.size(), reinterpret_cast<const float*>(symbols.data()),
                                        noise_vars.data(), reinterpret_cast<int8_t*>(llrs.data()));
    if (rc != NRPHY_OK) {
      report_failure("nrphy_demodulate_soft_host", rc);
      std::memset(static_cast<void*>(llrs.data()), 0, llrs.size());
    }
  }

private:
  std::shared_ptr<context> ctx;
};

/// channel_modulation_factory that hands out the device-backed demodulation mapper and leaves the modulation mapper and
/// the EVM calculator to the reference's factory it wraps (create_channel_modulation_sw_factory()): what
/// create_pusch_demodulator_factory_sw takes as its demodulation factory (R/lib/phy/upper/channel_processors/pusch/factories.cpp:160).
class channel_modulation_factory_adaptor : public srsran::channel_modulation_factory
{
public:
  channel_modulation_factory_adaptor(std::shared_ptr<context> ctx_, std::shared_ptr<srsran::channel_modulation_factory> inner_) :
    ctx(std::move(ctx_)), inner(std::move(inner_))
  {
  }
  std::unique_ptr<srsran::modulation_mapper>   create_modulation_mapper() override { return inner->create_modulation_mapper(); }
  std::unique_ptr<srsran::demodulation_mapper> create_demodulation_mapper() override
  {
    return std::make_unique<demodulation_mapper_adaptor>(ctx);
  }
  std::unique_ptr<srsran::evm_calculator> create_evm_calculator() override { return inner->create_evm_calculator(); }

private:
  std::shared_ptr<context>                            ctx;
  std::shared_ptr<srsran::channel_modulation_factory> inner;
};

// ---- FAPI batching shim --------------------------------------------------------------------------
/// dl_pdsch_pdu -> POD, field for field what convert_pdsch_fapi_to_phy builds as a pdu_t
/// (R/lib/fapi_adaptor/phy/messages/pdsch.cpp:152-206) and to_pod() then flattens, without the pdu_t in between.  The
/// resource allocation goes through the reference's own rb_allocation / vrb_to_prb_mapper (pdsch.cpp:100-150), the power
/// offsets follow pdsch.cpp:57-82.  \c weights receives the precoding coefficients the POD points to.
inline nrphy_pdsch_pdu_t fapi_to_pod(const srsran::fapi::dl_pdsch_pdu&                         fapi_pdu,
                                     uint16_t                                                   sfn,
                                     uint16_t                                                   slot,
                                     srsran::span<const srsran::re_pattern_list>                csi_re_pattern_list,
                                     const srsran::fapi_adaptor::precoding_matrix_repository&   pm_repo,
                                     size_t                                                     tb_size,
                                     std::vector<float>&                                        weights)
{
  using namespace srsran;
  nrphy_pdsch_pdu_t p;
  std::memset(&p, 0, sizeof(p));
  p.slot_index    = slot_point(fapi_pdu.scs, sfn, slot).slot_index();
  p.rnti          = to_value(fapi_pdu.rnti);
  p.bwp_start_rb  = fapi_pdu.bwp_start;
  p.bwp_size_rb   = fapi_pdu.bwp_size;
  p.cp            = (fapi_pdu.cp == cyclic_prefix::NORMAL) ? 0 : 1;
  p.nof_codewords = fapi_pdu.cws.size();
  p.qm            = fapi_pdu.cws.empty() ? 0 : static_cast<unsigned>(fapi_pdu.cws[0].qam_mod_order);
  p.rv            = fapi_pdu.cws.empty() ? 0 : fapi_pdu.cws[0].rv_index;
  p.n_id          = fapi_pdu.nid_pdsch;
  p.ref_point     = (fapi_pdu.ref_point == fapi::pdsch_ref_point_type::point_a) ? 0 : 1;
  p.dmrs_symbol_mask            = fapi_pdu.dl_dmrs_symb_pos & 0x3FFFU;
  p.dmrs_type                   = (fapi_pdu.dmrs_type == fapi::dmrs_cfg_type::type_1) ? 1 : 2;
  p.scrambling_id               = fapi_pdu.pdsch_dmrs_scrambling_id;
  p.n_scid                      = (fapi_pdu.nscid == 1U) ? 1 : 0;
  p.nof_cdm_groups_without_data = fapi_pdu.num_dmrs_cdm_grps_no_data;
  p.start_symbol_index          = fapi_pdu.start_symbol_index;
  p.nof_symbols                 = fapi_pdu.nr_of_symbols;
  p.ldpc_base_graph             = (fapi_pdu.pdsch_maintenance_v3.ldpc_base_graph == ldpc_base_graph_type::BG1) ? 1 : 2;
  p.tbs_lbrm_bytes              = fapi_pdu.pdsch_maintenance_v3.tb_size_lbrm_bytes.value();
  p.tb_size_bytes               = tb_size;

  // Resource allocation: VRB-to-PRB mapping by transmission type, then type 1 (start, length) or type 0 (bitmap, LSB of
  // byte 0 = VRB 0).
  const unsigned bwp_start = fapi_pdu.bwp_start, bwp_size = fapi_pdu.bwp_size;
  const unsigned n_start_coreset = fapi_pdu.pdsch_maintenance_v3.coreset_start_point - bwp_start;
  const unsigned n_bwp_init      = fapi_pdu.pdsch_maintenance_v3.initial_dl_bwp_size;
  unsigned       bundle          = 0;
  if (fapi_pdu.vrb_to_prb_mapping == fapi::vrb_to_prb_mapping_type::interleaved_rb_size2) {
    bundle = 2;
  } else if (fapi_pdu.vrb_to_prb_mapping == fapi::vrb_to_prb_mapping_type::interleaved_rb_size4) {
    bundle = 4;
  }
  vrb_to_prb_mapper vrb_map = vrb_to_prb_mapper::create_non_interleaved_other();
  switch (fapi_pdu.pdsch_maintenance_v3.trans_type) {
    case fapi::pdsch_trans_type::non_interleaved_common_ss:
      vrb_map = vrb_to_prb_mapper::create_non_interleaved_common_ss(n_start_coreset);
      break;
    case fapi::pdsch_trans_type::interleaved_common_type0_coreset0:
      vrb_map = vrb_to_prb_mapper::create_interleaved_coreset0(n_start_coreset, n_bwp_init);
      break;
    case fapi::pdsch_trans_type::interleaved_common_any_coreset0_present:
      vrb_map = vrb_to_prb_mapper::create_interleaved_common(n_start_coreset, bwp_start, n_bwp_init);
      break;
    case fapi::pdsch_trans_type::interleaved_common_any_coreset0_not_present:
      vrb_map = vrb_to_prb_mapper::create_interleaved_common(n_start_coreset, bwp_start, bwp_size);
      break;
    case fapi::pdsch_trans_type::interleaved_other:
      vrb_map = vrb_to_prb_mapper::create_interleaved_other(bwp_start, bwp_size, bundle);
      break;
    default:
      break;
  }
  rb_allocation alloc;
  if (fapi_pdu.resource_alloc == fapi::resource_allocation_type::type_1) {
    alloc = rb_allocation::make_type1(fapi_pdu.rb_start, fapi_pdu.rb_size, vrb_map);
  } else {
    bounded_bitset<MAX_RB> vrb_bitmap(bwp_size);
    for (unsigned vrb = 0; vrb != bwp_size; ++vrb) {
      if ((fapi_pdu.rb_bitmap[vrb / 8] >> (vrb % 8)) & 1U) {
        vrb_bitmap.set(vrb);
      }
    }
    alloc = rb_allocation::make_type0(vrb_bitmap, vrb_map);
  }
  p.vrb_contiguous           = alloc.is_contiguous() ? 1 : 0;
  bounded_bitset<MAX_RB> prb = alloc.get_prb_mask(bwp_start, bwp_size);
  for (unsigned i = 0; i != prb.size(); ++i) {
    if (prb.test(i)) {
      p.prb_mask[i / 64] |= uint64_t(1) << (i % 64);
    }
  }

  // Power: data offset from the two profile fields, DM-RS offset from TS 38.214 Table 4.1-1.
  float ss_dB = 6.0F;
  switch (fapi_pdu.power_control_offset_ss_profile_nr) {
    case fapi::power_control_offset_ss::dB_minus_3:
      ss_dB = -3.0F;
      break;
    case fapi::power_control_offset_ss::dB0:
      ss_dB = 0.0F;
      break;
    case fapi::power_control_offset_ss::dB3:
      ss_dB = 3.0F;
      break;
    default:
      break;
  }
  p.ratio_pdsch_data_to_sss_dB = ss_dB + static_cast<float>(fapi_pdu.power_control_offset_profile_nr);
  p.ratio_pdsch_dmrs_to_sss_dB = p.ratio_pdsch_data_to_sss_dB + get_sch_to_dmrs_ratio_dB(fapi_pdu.num_dmrs_cdm_grps_no_data);

  // Reserved RE: the CSI-RS patterns this PDU rate-matches around, merged as re_pattern_list::merge does.
  re_pattern_list reserved;
  for (auto csi_index : fapi_pdu.pdsch_maintenance_v3.csi_for_rm) {
    srsran_assert(csi_index < csi_re_pattern_list.size(), "CSI-RS PDU index out of bounds.");
    reserved.merge(csi_re_pattern_list[csi_index]);
  }
  for (const re_pattern& pat : reserved.get_re_patterns()) {
    nrphy_re_pattern_t& o = p.reserved[p.nof_reserved++];
    for (unsigned i = 0; i != pat.prb_mask.size(); ++i) {
      if (pat.prb_mask.test(i)) {
        o.prb_mask[i / 64] |= uint64_t(1) << (i % 64);
      }
    }
    for (unsigned k = 0; k != NRE; ++k) {
      o.re_mask |= pat.re_mask.test(k) ? (1U << k) : 0U;
    }
    for (unsigned l = 0; l != pat.symbols.size(); ++l) {
      o.symbol_mask |= pat.symbols.test(l) ? (1U << l) : 0U;
    }
  }

  // Wideband precoding from the repository (one PRG, as the reference asserts).
  srsran_assert(fapi_pdu.precoding_and_beamforming.prgs.size() == 1U, "Unsupported number of PRGs.");
  const precoding_weight_matrix& w = pm_repo.get_precoding_matrix(fapi_pdu.precoding_and_beamforming.prgs.front().pm_index);
  p.nof_layers                     = w.get_nof_layers();
  p.nof_ports                      = w.get_nof_ports();
  p.prg_size_rb                    = MAX_RB;
  p.nof_prg                        = 1;
  weights.resize(2 * p.nof_ports * p.nof_layers);
  for (unsigned port = 0; port != p.nof_ports; ++port) {
    for (unsigned l = 0; l != p.nof_layers; ++l) {
      cf_t c                                  = w.get_coefficient(l, port);
      weights[2 * (port * p.nof_layers + l)]     = c.real();
      weights[2 * (port * p.nof_layers + l) + 1] = c.imag();
    }
  }
  p.precoding = weights.data();
  return p;
}

/// All PDSCH PDUs of one DL_TTI.request: add() converts and keeps each PDU with its transport block, process() runs them as
/// ONE plan and one launch into the slot's grid (nrphy_pdsch_process_slot_host) -- where the reference's translator hands
/// every PDU to its own pdsch_processor::process (fapi_to_phy_translator.cpp).  Blocking; the grid is read first, so what
/// the other channels wrote stays.
class fapi_pdsch_slot_batch
{
public:
  fapi_pdsch_slot_batch(std::shared_ptr<context> ctx_, unsigned nof_ports_, unsigned nof_subc_) :
    ctx(std::move(ctx_)), nof_ports(nof_ports_), nof_subc(nof_subc_)
  {
  }

  /// Returns false (and keeps nothing) when the PDU is one the reference's validator refuses.
  bool add(const srsran::fapi::dl_pdsch_pdu&                       fapi_pdu,
           uint16_t                                                 sfn,
           uint16_t                                                 slot,
           srsran::span<const srsran::re_pattern_list>              csi_re_pattern_list,
           const srsran::fapi_adaptor::precoding_matrix_repository& pm_repo,
           srsran::span<const uint8_t>                              transport_block)
  {
    weights.emplace_back();
    nrphy_pdsch_pdu_t pod = fapi_to_pod(fapi_pdu, sfn, slot, csi_re_pattern_list, pm_repo, transport_block.size(), weights.back());
    if (nrphy_pdsch_validate(&pod) != NRPHY_OK) {
      weights.pop_back();
      return false;
    }
    pods.push_back(pod);
    tbs.push_back(transport_block.data());
    return true;
  }

  unsigned size() const { return pods.size(); }

  /// Runs the batch into `grid` and empties it.  On failure the grid is left as it was and false is returned.
  bool process(srsran::resource_grid& grid)
  {
    using namespace srsran;
    if (pods.empty()) {
      return true;
    }
    for (unsigned i = 0; i != pods.size(); ++i) {
      pods[i].precoding = weights[i].data(); // (the vectors may have moved while the batch grew)
    }
    staging.resize(static_cast<size_t>(nof_ports) * MAX_NSYMB_PER_SLOT * nof_subc);
    const resource_grid_reader& reader = grid.get_reader();
    for (unsigned p = 0; p != nof_ports; ++p) {
      for (unsigned l = 0; l != MAX_NSYMB_PER_SLOT; ++l) {
        span<const cbf16_t> view = reader.get_view(p, l);
        std::memcpy(&staging[(static_cast<size_t>(p) * MAX_NSYMB_PER_SLOT + l) * nof_subc], view.data(), nof_subc * sizeof(cbf16_t));
      }
    }
    int rc = nrphy_pdsch_process_slot_host(ctx->get(), pods.size(), pods.data(), tbs.data(), staging.data(), nof_ports, nof_subc);
    pods.clear();
    tbs.clear();
    weights.clear();
    if (rc != NRPHY_OK) {
      report_failure("nrphy_pdsch_process_slot_host", rc);
      return false;
    }
    resource_grid_writer& writer = grid.get_writer();
    for (unsigned p = 0; p != nof_ports; ++p) {
      for (unsigned l = 0; l != MAX_NSYMB_PER_SLOT; ++l) {
        writer.put(p, l, 0, 1, span<const cbf16_t>(&staging[(static_cast<size_t>(p) * MAX_NSYMB_PER_SLOT + l) * nof_subc], nof_subc));
      }
    }
    return true;
  }

private:
  std::shared_ptr<context>        ctx;
  unsigned                        nof_ports, nof_subc;
  std::vector<nrphy_pdsch_pdu_t>  pods;
  std::vector<const uint8_t*>     tbs;
  std::vector<std::vector<float>> weights;
  std::vector<srsran::cbf16_t>    staging;
};

} // namespace mi355
