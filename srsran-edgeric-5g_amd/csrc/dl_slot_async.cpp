// Seams A and C on one device-resident grid: the downlink slot pipeline (nrphy_dl_slots_* / nrphy_dl_slot_*).
//
// The reference fills a slot's resource grid on the host (downlink_processor_single_executor_impl.cpp:52-215), hands it to
// the lower PHY when the last PDU is done (pdxch_processor_impl::handle_request, pdxch_processor_impl.cpp:97-112) and
// modulates it one OFDM symbol at a time on the real-time thread (process_symbol, :47-95).  A device cannot be asked for one
// symbol at a time by a thread that must never wait, so the slot is modulated as a whole WHEN THE GRID IS HANDED OVER and the
// real-time thread only copies from pinned memory.  And the grid need not visit the host at
// all in between: every writer of the slot (PDSCH, PDCCH, SS/PBCH, CSI-RS, sparse puts from the host) works on the slot's
// grid in HBM, in call order on the slot's stream, and the modulator reads it there.
//
// Rules as in pdsch_async.cpp: a submit builds what it needs into the slot's pinned staging, enqueues copies and kernels on
// the slot's stream and returns; completion is a stream callback.  No device allocation (the control-channel writers use
// their stand-alone forms' stream-ordered staging), no blocking copy, except where the header says "blocking".
#include "nrphy_host_internal.h"

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <new>

namespace {

enum SlotState : uint32_t { SLOT_FREE = 0, SLOT_OPEN, SLOT_MODULATING, SLOT_DONE };

struct DlSlot {
  nrphy_dl_slots*     pool    = nullptr;
  uint32_t            id      = 0;
  hipStream_t         stream  = nullptr;
  uint32_t*           d_grid  = nullptr;
  void*               d_iq    = nullptr;
  uint8_t*            d_stage = nullptr; // transport blocks and plan tables, call after call
  uint8_t*            h_stage = nullptr; // pinned twin of d_stage
  void*               h_grid  = nullptr; // pinned: nrphy_dl_slot_load_grid / _read_grid
  void*               h_iq    = nullptr; // pinned: [port][slot_stride] samples
  nrphy_amplitude_stats_t* d_stats = nullptr; // wire-format pools: the amplitude controller's measurements per port
  nrphy_amplitude_stats_t* h_stats = nullptr; // ... and their pinned twin
  uint32_t*           d_scratch = nullptr; // sequences and TB-CRC shares of the slot's PDSCH runs (ordered on the stream)
  PlanShapeCache*     shapes  = nullptr;
  std::vector<nrphy_pdsch_plan_t*> plans; // of this open; destroyed when the slot is opened again
  std::mutex          mutex;              // serialises the calls on this slot
  size_t              stage_used = 0;
  size_t              tb_used    = 0;
  bool                grid_defined = false; // false: nothing has written the grid since open (it still has to be zeroed)
  uint32_t            slot_index = 0;       // of the modulate submitted
  std::atomic<uint32_t> state{SLOT_FREE};
  std::atomic<int>    status{NRPHY_OK};
  nrphy_dl_slot_done_fn done = nullptr;
  void*               user   = nullptr;
};

} // namespace

struct nrphy_dl_slots {
  nrphy_ctx*              ctx = nullptr;
  nrphy_dl_slots_cfg_t    cfg;
  nrphy_ofdm_plan_t*      ofdm = nullptr;
  uint32_t                nof_subc = 0, slot_stride = 0, sample_bytes = 8;
  size_t                  grid_bytes = 0, iq_bytes = 0, stage_bytes = 0, scratch_words = 0, tb_cap = 0;
  uint32_t*               d_slot_numbers = nullptr; // 0, 1, ... : nrphy_ofdm_run takes the slot index from device memory
  size_t                  iq_block_bytes = 0; // the IQ of a slot and, behind it on a 256-byte boundary, a wire-format pool's measurements
  // NRPHY_DL_SLOT_ZERO_COPY (read at creation): bit 0 = the kernels read transport blocks and plan tables from the pinned staging
  // in place (no copy down), bit 1 = the modulator writes IQ and measurements into the pinned buffer in place (no copy up).
  // hipMemcpyAsync is the most expensive call of a submit (9 us of host time each, DESIGN_HISTORY.md round 2).
  uint32_t                zero_copy = 0;
  std::vector<DlSlot>     slots;
  std::mutex              mutex; // open / close
  std::condition_variable changed;
  uint32_t                nof_open = 0;
};

namespace {

// Runs on a thread of the HIP runtime when the slot's IQ has reached the pinned buffer (or the stream has failed).
void on_slot_done(hipStream_t, hipError_t error, void* arg)
{
  DlSlot*         slot = static_cast<DlSlot*>(arg);
  nrphy_dl_slots* pool = slot->pool;
  const int       rc   = error == hipSuccess ? NRPHY_OK : NRPHY_ERR_DEVICE;
  slot->status.store(rc, std::memory_order_relaxed);
  slot->state.store(SLOT_DONE, std::memory_order_release);
  if (slot->done != nullptr) {
    slot->done(slot->user, rc, slot->id);
  }
  {
    std::lock_guard<std::mutex> lock(pool->mutex);
  }
  pool->changed.notify_all();
}

DlSlot* open_slot(nrphy_dl_slots* pool, uint32_t slot_id)
{
  if (pool == nullptr || slot_id >= pool->slots.size()) {
    return nullptr;
  }
  DlSlot* s = &pool->slots[slot_id];
  return s->state.load(std::memory_order_acquire) == SLOT_FREE ? nullptr : s;
}

// Before the first writer of an open slot: the grid is all zeros (resource_grid::set_all_zero).  A PDSCH run clears what it
// does not map by itself (zero_grids), so only the other writers need the memset.
int define_grid(DlSlot* s)
{
  if (!s->grid_defined) {
    HIP_TRY(hipMemsetAsync(s->d_grid, 0, s->pool->grid_bytes, s->stream));
    s->grid_defined = true;
  }
  return NRPHY_OK;
}

} // namespace

extern "C" int nrphy_dl_slots_create(nrphy_ctx_t* ctx, const nrphy_dl_slots_cfg_t* cfg, nrphy_dl_slots_t** out)
{
  if (ctx == nullptr || cfg == nullptr || out == nullptr || cfg->depth == 0 || cfg->depth > 64 || cfg->nof_ports == 0 ||
      cfg->nof_ports > NRPHY_MAX_PORTS || cfg->iq_format > 1 || cfg->ofdm.bw_rb == 0 || cfg->ofdm.bw_rb > NRPHY_MAX_RB) {
    return NRPHY_ERR_ARGUMENT;
  }
  *out = nullptr;
  HIP_TRY(hipSetDevice(ctx->device));
  nrphy_dl_slots* pool = new (std::nothrow) nrphy_dl_slots;
  if (pool == nullptr) {
    return NRPHY_ERR_CAPACITY;
  }
  pool->ctx = ctx;
  pool->cfg = *cfg;
  int rc    = nrphy_ofdm_plan_create(ctx, &cfg->ofdm, cfg->nof_ports, &pool->ofdm);
  if (rc != NRPHY_OK) {
    delete pool;
    return rc;
  }
  pool->nof_subc     = 12 * cfg->ofdm.bw_rb;
  pool->slot_stride  = nrphy_ofdm_plan_slot_stride(pool->ofdm);
  pool->sample_bytes = cfg->iq_format == 1 ? 4 : 8;
  pool->grid_bytes   = (size_t)cfg->nof_ports * NRPHY_NSYMB * pool->nof_subc * 4;
  pool->iq_bytes     = (size_t)cfg->nof_ports * pool->slot_stride * pool->sample_bytes;
  const size_t stats_bytes = cfg->iq_format == 1 ? sizeof(nrphy_amplitude_stats_t) * cfg->nof_ports : 0;
  pool->iq_block_bytes     = stats_bytes != 0 ? ((pool->iq_bytes + 255) & ~(size_t)255) + stats_bytes : pool->iq_bytes;
  if (const char* e = std::getenv("NRPHY_DL_SLOT_ZERO_COPY")) {
    pool->zero_copy = (uint32_t)std::atoi(e) & 3u;
  }
  // Staging: the slot's transport blocks (each call's rounded up to 256 bytes) and, behind each call's blocks, its plan
  // tables -- a few KB for wideband PDUs, two bytes per RE for fragmented allocations (pdsch_async.cpp sizes one operation
  // the same way); room for four calls' worth of tables, more calls share what is left.
  const size_t table_cap = ((size_t)64 * 1024 + (size_t)NRPHY_NSYMB * pool->nof_subc * 6 + 255) & ~(size_t)255;
  pool->tb_cap           = ((size_t)cfg->max_tb_bytes + 7 + 255) & ~(size_t)255;
  pool->stage_bytes      = pool->tb_cap + 4 * table_cap + 64 * 256; // (every call's blocks and tables start on 256 bytes)
  pool->scratch_words    = (size_t)NRPHY_NSYMB * pool->nof_subc * 2 + 16384;
  if (const char* e = std::getenv("NRPHY_DL_SLOT_STAGE_BYTES")) { // tests: a small value exercises the capacity paths
    pool->stage_bytes = ((size_t)std::max(1024, std::atoi(e)) + 255) & ~(size_t)255;
  }
  std::vector<uint32_t> numbers(16);
  for (uint32_t i = 0; i != 16; ++i) {
    numbers[i] = i;
  }
  if (upload(&pool->d_slot_numbers, numbers.data(), numbers.size() * sizeof(uint32_t)) != hipSuccess) {
    nrphy_dl_slots_destroy(pool);
    return NRPHY_ERR_DEVICE;
  }
  pool->slots = std::vector<DlSlot>(cfg->depth);
  uint32_t id = 0;
  for (DlSlot& s : pool->slots) {
    s.pool = pool;
    s.id   = id++;
    const bool stage_in_place = (pool->zero_copy & 1u) != 0, iq_in_place = (pool->zero_copy & 2u) != 0;
    if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&s.d_grid, pool->grid_bytes) != hipSuccess ||
        hipMalloc((void**)&s.d_scratch, pool->scratch_words * sizeof(uint32_t)) != hipSuccess ||
        hipHostMalloc((void**)&s.h_stage, pool->stage_bytes, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(&s.h_grid, pool->grid_bytes, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(&s.h_iq, pool->iq_block_bytes, hipHostMallocDefault) != hipSuccess ||
        (stage_in_place ? hipHostGetDevicePointer((void**)&s.d_stage, s.h_stage, 0)
                        : hipMalloc((void**)&s.d_stage, pool->stage_bytes)) != hipSuccess ||
        (iq_in_place ? hipHostGetDevicePointer(&s.d_iq, s.h_iq, 0) : hipMalloc(&s.d_iq, pool->iq_block_bytes)) != hipSuccess) {
      nrphy_dl_slots_destroy(pool);
      return NRPHY_ERR_DEVICE;
    }
    std::memset(s.h_stage, 0, pool->stage_bytes);
    std::memset(s.h_iq, 0, pool->iq_block_bytes);
    if (stats_bytes != 0) { // (behind the IQ on both sides: one copy brings both up)
      const size_t at = pool->iq_block_bytes - stats_bytes;
      s.d_stats       = reinterpret_cast<nrphy_amplitude_stats_t*>(static_cast<uint8_t*>(s.d_iq) + at);
      s.h_stats       = reinterpret_cast<nrphy_amplitude_stats_t*>(static_cast<uint8_t*>(s.h_iq) + at);
    }
    s.shapes = plan_shape_cache_create();
  }
  if (cfg->iq_format == 1 && cfg->wire.amplitude.kind == 0) {
    // A wire-format run with measurements keeps per-workgroup records in a buffer of the OFDM plan that grows with the batch:
    // the pool runs one grid at a time, so one run here sizes it and the submit path never reallocates.
    DlSlot& s = pool->slots[0];
    rc        = hipMemsetAsync(s.d_grid, 0, pool->grid_bytes, s.stream) == hipSuccess ? NRPHY_OK : NRPHY_ERR_DEVICE;
    if (rc == NRPHY_OK) {
      rc = nrphy_ofdm_run_ci16(pool->ofdm, 1, s.d_grid, pool->d_slot_numbers, &cfg->wire, (int16_t*)s.d_iq, s.d_stats, s.stream);
    }
    if (rc == NRPHY_OK && hipStreamSynchronize(s.stream) != hipSuccess) {
      rc = NRPHY_ERR_DEVICE;
    }
    if (rc != NRPHY_OK) {
      nrphy_dl_slots_destroy(pool);
      return rc;
    }
  }
  *out = pool;
  return NRPHY_OK;
}

extern "C" int nrphy_dl_slots_destroy(nrphy_dl_slots_t* pool)
{
  if (pool == nullptr) {
    return NRPHY_OK;
  }
  (void)hipSetDevice(pool->ctx->device);
  for (DlSlot& s : pool->slots) {
    if (s.stream) {
      (void)hipStreamSynchronize(s.stream); // callbacks included
    }
    for (nrphy_pdsch_plan_t* p : s.plans) {
      nrphy_pdsch_plan_destroy(p);
    }
    plan_shape_cache_destroy(s.shapes);
    (void)hipFree(s.d_grid);
    if ((pool->zero_copy & 2u) == 0) {
      (void)hipFree(s.d_iq);
    }
    if ((pool->zero_copy & 1u) == 0) {
      (void)hipFree(s.d_stage);
    }
    (void)hipFree(s.d_scratch);
    (void)hipHostFree(s.h_stage);
    (void)hipHostFree(s.h_grid);
    (void)hipHostFree(s.h_iq); // (the measurements live in the same blocks)
    if (s.stream) {
      (void)hipStreamDestroy(s.stream);
    }
  }
  (void)hipFree(pool->d_slot_numbers);
  nrphy_ofdm_plan_destroy(pool->ofdm);
  delete pool;
  return NRPHY_OK;
}

extern "C" int nrphy_dl_slots_wait_free(nrphy_dl_slots_t* pool)
{
  if (pool == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  std::unique_lock<std::mutex> lock(pool->mutex);
  pool->changed.wait(lock, [pool] { return pool->nof_open < pool->slots.size(); });
  return NRPHY_OK;
}

extern "C" int nrphy_dl_slot_open(nrphy_dl_slots_t* pool, uint32_t* slot_id)
{
  if (pool == nullptr || slot_id == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  DlSlot* slot = nullptr;
  {
    std::lock_guard<std::mutex> lock(pool->mutex);
    for (DlSlot& s : pool->slots) {
      if (s.state.load(std::memory_order_acquire) == SLOT_FREE) {
        slot = &s;
        s.state.store(SLOT_OPEN, std::memory_order_release);
        ++pool->nof_open;
        break;
      }
    }
  }
  if (slot == nullptr) {
    return NRPHY_ERR_CAPACITY;
  }
  std::lock_guard<std::mutex> lock(slot->mutex);
  // The slot was closed, so nothing of its previous use is in flight: its plans go.
  for (nrphy_pdsch_plan_t* p : slot->plans) {
    nrphy_pdsch_plan_destroy(p);
  }
  slot->plans.clear();
  slot->stage_used   = 0;
  slot->tb_used      = 0;
  slot->grid_defined = false;
  slot->done         = nullptr;
  slot->user         = nullptr;
  slot->status.store(NRPHY_OK, std::memory_order_relaxed);
  *slot_id = slot->id;
  return NRPHY_OK;
}

extern "C" int nrphy_dl_slot_close(nrphy_dl_slots_t* pool, uint32_t slot_id)
{
  DlSlot* s = open_slot(pool, slot_id);
  if (s == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  {
    std::lock_guard<std::mutex> lock(s->mutex);
    if (hipSetDevice(pool->ctx->device) != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) {
      // (the slot is given back all the same: a failed stream fails the next use too, and says so)
    }
    // A completion callback of this slot may still be running on its runtime thread: wait for its state.
    if (s->state.load(std::memory_order_acquire) == SLOT_MODULATING) {
      std::unique_lock<std::mutex> pool_lock(pool->mutex);
      pool->changed.wait(pool_lock, [s] { return s->state.load(std::memory_order_acquire) == SLOT_DONE; });
    }
  }
  {
    std::lock_guard<std::mutex> lock(pool->mutex);
    s->state.store(SLOT_FREE, std::memory_order_release);
    --pool->nof_open;
  }
  pool->changed.notify_all();
  return NRPHY_OK;
}

extern "C" int nrphy_dl_slot_pdsch(nrphy_dl_slots_t* pool, uint32_t slot_id, uint32_t n_pdu, const nrphy_pdsch_pdu_t* pdus,
                                   const uint8_t* const* tbs)
{
  DlSlot* s = open_slot(pool, slot_id);
  if (s == nullptr || pdus == nullptr || tbs == nullptr || n_pdu == 0) {
    return NRPHY_ERR_ARGUMENT;
  }
  std::lock_guard<std::mutex> lock(s->mutex);
  if (s->state.load(std::memory_order_acquire) != SLOT_OPEN) {
    return NRPHY_ERR_ARGUMENT; // the grid has been handed over
  }
  std::vector<uint64_t> tb_off(n_pdu);
  size_t                tb_total = 0;
  for (uint32_t i = 0; i != n_pdu; ++i) {
    if (tbs[i] == nullptr || pdus[i].tb_size_bytes == 0) {
      return NRPHY_ERR_ARGUMENT;
    }
    tb_off[i] = tb_total;
    tb_total += ((size_t)pdus[i].tb_size_bytes + 7) & ~(size_t)3; // readable to the next multiple of 4
  }
  const size_t tables_at = (tb_total + 255) & ~(size_t)255; // relative to this call's region of the staging
  size_t       tb_bytes  = 0;
  for (uint32_t i = 0; i != n_pdu; ++i) {
    tb_bytes += pdus[i].tb_size_bytes;
  }
  if (s->tb_used + tb_bytes > pool->cfg.max_tb_bytes || s->stage_used + tables_at + 256 > pool->stage_bytes) {
    return NRPHY_ERR_CAPACITY;
  }
  if (hipSetDevice(pool->ctx->device) != hipSuccess) {
    return NRPHY_ERR_DEVICE;
  }
  uint8_t* const        h_base = s->h_stage + s->stage_used;
  uint8_t* const        d_base = s->d_stage + s->stage_used;
  std::vector<uint32_t> grid_of(n_pdu, 0);
  PlanPlacement         place;
  place.h_tables               = h_base + tables_at;
  place.d_tables               = d_base + tables_at;
  place.table_capacity         = (pool->stage_bytes - s->stage_used - tables_at) & ~(size_t)255;
  place.d_scratch              = s->d_scratch;
  place.scratch_capacity_words = pool->scratch_words;
  place.cache                  = s->shapes;
  nrphy_pdsch_plan_t* plan     = nullptr;
  int rc = nrphy_pdsch_plan_create_placed(pool->ctx, n_pdu, pdus, tb_off.data(), grid_of.data(), 1, pool->cfg.nof_ports, pool->nof_subc,
                                          &place, &plan);
  bool own_memory = false;
  if (rc == NRPHY_ERR_CAPACITY) {
    // Tables too big for what is left of the staging: a plan with device memory of its own (an allocation and a blocking
    // copy -- rare: hundreds of PDUs, or RE tables for most of a fragmented grid, in one slot).
    rc         = nrphy_pdsch_plan_create(pool->ctx, n_pdu, pdus, tb_off.data(), grid_of.data(), 1, pool->cfg.nof_ports, pool->nof_subc, &plan);
    own_memory = true;
  }
  if (rc != NRPHY_OK) {
    return rc;
  }
  s->plans.push_back(plan);
  for (uint32_t i = 0; i != n_pdu; ++i) {
    const size_t span = (i + 1 != n_pdu ? tb_off[i + 1] : tb_total) - tb_off[i];
    std::memcpy(h_base + tb_off[i], tbs[i], pdus[i].tb_size_bytes);
    std::memset(h_base + tb_off[i] + pdus[i].tb_size_bytes, 0, span - pdus[i].tb_size_bytes);
  }
  const size_t copy_bytes = own_memory ? tb_total : tables_at + place.table_bytes;
  if ((pool->zero_copy & 1u) == 0) {
    HIP_TRY(hipMemcpyAsync(d_base, h_base, copy_bytes, hipMemcpyHostToDevice, s->stream));
  }
  // The first writer of the slot clears what it does not map in the same launch; later ones leave the rest alone.
  const int zero_grids = s->grid_defined ? 0 : 1;
  rc                   = nrphy_pdsch_run(plan, d_base, s->d_grid, nullptr, nullptr, zero_grids, s->stream);
  if (rc != NRPHY_OK) {
    return rc;
  }
  s->grid_defined = true;
  s->tb_used += tb_bytes;
  s->stage_used += (copy_bytes + 255) & ~(size_t)255;
  return NRPHY_OK;
}

extern "C" int nrphy_dl_slot_pdcch(nrphy_dl_slots_t* pool, uint32_t slot_id, uint32_t n, const nrphy_pdcch_pdu_t* pdus)
{
  DlSlot* s = open_slot(pool, slot_id);
  if (s == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  std::lock_guard<std::mutex> lock(s->mutex);
  if (s->state.load(std::memory_order_acquire) != SLOT_OPEN) {
    return NRPHY_ERR_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(pool->ctx->device));
  int rc = define_grid(s);
  return rc != NRPHY_OK ? rc : nrphy_pdcch_process(pool->ctx, n, pdus, nullptr, s->d_grid, pool->cfg.nof_ports, pool->nof_subc, s->stream);
}

extern "C" int nrphy_dl_slot_ssb(nrphy_dl_slots_t* pool, uint32_t slot_id, uint32_t n, const nrphy_ssb_pdu_t* pdus)
{
  DlSlot* s = open_slot(pool, slot_id);
  if (s == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  std::lock_guard<std::mutex> lock(s->mutex);
  if (s->state.load(std::memory_order_acquire) != SLOT_OPEN) {
    return NRPHY_ERR_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(pool->ctx->device));
  int rc = define_grid(s);
  return rc != NRPHY_OK ? rc : nrphy_ssb_process(pool->ctx, n, pdus, nullptr, s->d_grid, pool->cfg.nof_ports, pool->nof_subc, s->stream);
}

extern "C" int nrphy_dl_slot_csi_rs(nrphy_dl_slots_t* pool, uint32_t slot_id, uint32_t n, const nrphy_csi_rs_cfg_t* cfgs)
{
  DlSlot* s = open_slot(pool, slot_id);
  if (s == nullptr || n == 0 || n > 4096) {
    return NRPHY_ERR_ARGUMENT;
  }
  std::lock_guard<std::mutex> lock(s->mutex);
  if (s->state.load(std::memory_order_acquire) != SLOT_OPEN) {
    return NRPHY_ERR_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(pool->ctx->device));
  int rc = define_grid(s);
  if (rc != NRPHY_OK) {
    return rc;
  }
  const std::vector<uint32_t> grid_of(n, 0);
  return nrphy_csi_rs_map(pool->ctx, n, cfgs, grid_of.data(), s->d_grid, pool->cfg.nof_ports, pool->nof_subc, s->stream);
}

extern "C" int nrphy_dl_slot_put(nrphy_dl_slots_t* pool, uint32_t slot_id, uint32_t n, const nrphy_grid_re_t* entries)
{
  DlSlot* s = open_slot(pool, slot_id);
  if (s == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  std::lock_guard<std::mutex> lock(s->mutex);
  if (s->state.load(std::memory_order_acquire) != SLOT_OPEN) {
    return NRPHY_ERR_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(pool->ctx->device));
  int rc = define_grid(s);
  return rc != NRPHY_OK ? rc : nrphy_grid_put(pool->ctx, s->d_grid, pool->cfg.nof_ports, pool->nof_subc, n, entries, s->stream);
}

extern "C" int nrphy_dl_slot_load_grid(nrphy_dl_slots_t* pool, uint32_t slot_id, const void* grid)
{
  DlSlot* s = open_slot(pool, slot_id);
  if (s == nullptr || grid == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  std::lock_guard<std::mutex> lock(s->mutex);
  if (s->state.load(std::memory_order_acquire) != SLOT_OPEN) {
    return NRPHY_ERR_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(pool->ctx->device));
  // (h_grid may still feed an earlier load of this open: ordered behind it by waiting for the stream only then)
  if (s->grid_defined) {
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  std::memcpy(s->h_grid, grid, pool->grid_bytes);
  HIP_TRY(hipMemcpyAsync(s->d_grid, s->h_grid, pool->grid_bytes, hipMemcpyHostToDevice, s->stream));
  s->grid_defined = true;
  return NRPHY_OK;
}

extern "C" int nrphy_dl_slot_modulate(nrphy_dl_slots_t* pool, uint32_t slot_id, uint32_t subframe_slot_index,
                                      nrphy_dl_slot_done_fn done, void* user)
{
  DlSlot* s = open_slot(pool, slot_id);
  if (s == nullptr || subframe_slot_index >= (1U << pool->cfg.ofdm.numerology)) {
    return NRPHY_ERR_ARGUMENT;
  }
  std::lock_guard<std::mutex> lock(s->mutex);
  if (s->state.load(std::memory_order_acquire) != SLOT_OPEN) {
    return NRPHY_ERR_ARGUMENT; // once per open
  }
  HIP_TRY(hipSetDevice(pool->ctx->device));
  int rc = define_grid(s); // nothing written: an all-zero grid gives silence
  if (rc != NRPHY_OK) {
    return rc;
  }
  const uint32_t* d_slot = pool->d_slot_numbers + subframe_slot_index;
  if (pool->cfg.iq_format == 1) {
    rc = nrphy_ofdm_run_ci16(pool->ofdm, 1, s->d_grid, d_slot, &pool->cfg.wire, (int16_t*)s->d_iq, s->d_stats, s->stream);
  } else {
    rc = nrphy_ofdm_run(pool->ofdm, 1, s->d_grid, d_slot, (float*)s->d_iq, s->stream);
  }
  if (rc != NRPHY_OK) {
    return rc;
  }
  s->slot_index = subframe_slot_index;
  s->done       = done;
  s->user       = user;
  s->state.store(SLOT_MODULATING, std::memory_order_release);
  // IQ and, for a wire-format pool, the measurements behind it come up in ONE copy -- or were written in place.
  if (((pool->zero_copy & 2u) == 0 &&
       hipMemcpyAsync(s->h_iq, s->d_iq, pool->iq_block_bytes, hipMemcpyDeviceToHost, s->stream) != hipSuccess) ||
      hipStreamAddCallback(s->stream, on_slot_done, s, 0) != hipSuccess) {
    (void)hipStreamSynchronize(s->stream);
    s->status.store(NRPHY_ERR_DEVICE, std::memory_order_relaxed);
    s->state.store(SLOT_DONE, std::memory_order_release);
    return NRPHY_ERR_DEVICE;
  }
  return NRPHY_OK;
}

extern "C" int nrphy_dl_slot_poll(nrphy_dl_slots_t* pool, uint32_t slot_id)
{
  if (pool == nullptr || slot_id >= pool->slots.size()) {
    return NRPHY_ERR_ARGUMENT;
  }
  const DlSlot& s = pool->slots[slot_id];
  switch (s.state.load(std::memory_order_acquire)) {
    case SLOT_DONE:
      return s.status.load(std::memory_order_relaxed);
    case SLOT_FREE:
      return NRPHY_ERR_ARGUMENT;
    default:
      return NRPHY_ERR_NOT_READY;
  }
}

extern "C" int nrphy_dl_slot_wait(nrphy_dl_slots_t* pool, uint32_t slot_id)
{
  DlSlot* s = open_slot(pool, slot_id);
  if (s == nullptr || s->state.load(std::memory_order_acquire) == SLOT_OPEN) {
    return NRPHY_ERR_ARGUMENT; // nothing to wait for
  }
  std::unique_lock<std::mutex> lock(pool->mutex);
  pool->changed.wait(lock, [s] { return s->state.load(std::memory_order_acquire) != SLOT_MODULATING; });
  return s->state.load(std::memory_order_acquire) == SLOT_DONE ? s->status.load(std::memory_order_relaxed) : NRPHY_ERR_ARGUMENT;
}

extern "C" const void* nrphy_dl_slot_iq(nrphy_dl_slots_t* pool, uint32_t slot_id, uint32_t port, uint32_t* nof_samples)
{
  if (pool == nullptr || slot_id >= pool->slots.size() || port >= pool->cfg.nof_ports) {
    return nullptr;
  }
  const DlSlot& s = pool->slots[slot_id];
  if (nof_samples != nullptr) {
    *nof_samples = nrphy_ofdm_slot_size(&pool->cfg.ofdm, s.slot_index);
  }
  return static_cast<const uint8_t*>(s.h_iq) + (size_t)port * pool->slot_stride * pool->sample_bytes;
}

extern "C" const nrphy_amplitude_stats_t* nrphy_dl_slot_amplitude_stats(nrphy_dl_slots_t* pool, uint32_t slot_id, uint32_t port)
{
  if (pool == nullptr || slot_id >= pool->slots.size() || port >= pool->cfg.nof_ports || pool->slots[slot_id].h_stats == nullptr) {
    return nullptr;
  }
  return pool->slots[slot_id].h_stats + port;
}

extern "C" int nrphy_dl_slot_read_grid(nrphy_dl_slots_t* pool, uint32_t slot_id, void* grid)
{
  DlSlot* s = open_slot(pool, slot_id);
  if (s == nullptr || grid == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  std::lock_guard<std::mutex> lock(s->mutex);
  HIP_TRY(hipSetDevice(pool->ctx->device));
  int rc = define_grid(s);
  if (rc != NRPHY_OK) {
    return rc;
  }
  HIP_TRY(hipMemcpyAsync(s->h_grid, s->d_grid, pool->grid_bytes, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  std::memcpy(grid, s->h_grid, pool->grid_bytes);
  return NRPHY_OK;
}

extern "C" void* nrphy_dl_slot_device_grid(nrphy_dl_slots_t* pool, uint32_t slot_id)
{
  DlSlot* s = open_slot(pool, slot_id);
  return s ? s->d_grid : nullptr;
}

extern "C" void* nrphy_dl_slot_stream(nrphy_dl_slots_t* pool, uint32_t slot_id)
{
  DlSlot* s = open_slot(pool, slot_id);
  return s ? (void*)s->stream : nullptr;
}
