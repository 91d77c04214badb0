// Seam A, asynchronous host-span form: what a drop-in pdsch_processor needs to keep several PDUs in flight.
//
// pdsch_processor::process may return before the PDU is done and signal completion from any thread
// (R/include/srsran/phy/upper/channel_processors/pdsch_processor.h:157-170; the reference's own asynchronous pool:
// R/lib/phy/upper/channel_processors/pdsch_processor_asynchronous_pool.h:39-143).  A queue owns `depth` operation
// slots, each with a stream, device buffers and pinned host staging; a submit builds the PDU's plan INTO the slot's
// staging (nrphy_pdsch_plan_create_placed: no allocation, no blocking copy -- the reference derives its per-PDU state on
// every call too, pdsch_processor_concurrent_impl.cpp:55-207, and every slot of live traffic brings a new pdu_t), copies
// the transport block behind it, enqueues ONE H2D copy of both + the PDSCH kernels + D2H of the grid + a completion
// callback on the slot's stream and returns; the callback hands the grid to the caller's handler on a runtime thread.
// What depends only on the SHAPE of the PDUs (RE mapping tables, zero-fill run lists) is kept per slot across submits.
#include "nrphy_host_internal.h"

#include <atomic>
#include <cstdlib>
#include <condition_variable>

namespace {

struct AsyncSlot {
  nrphy_pdsch_async*  queue   = nullptr;
  hipStream_t         stream  = nullptr;
  uint8_t*            d_tb    = nullptr;
  void*               d_grid  = nullptr;
  uint8_t*            h_tb    = nullptr; // pinned
  void*               h_grid  = nullptr; // pinned
  bool                busy    = false;
  int                 status  = NRPHY_OK;
  nrphy_pdsch_done_fn done    = nullptr;
  void*               user    = nullptr;
  // The staging buffers carry the transport block(s) and, behind them (256-byte aligned), the plan's tables: one copy
  // moves both.  d_scratch: what every run rewrites (sequences, TB-CRC shares).
  uint32_t*           d_scratch = nullptr;
  nrphy_pdsch_plan_t* plan    = nullptr;  // the plan of the operation in flight (or of the last one), tables in the staging
  PlanShapeCache*     shapes  = nullptr;  // RE mapping tables and zero-fill lists by PDU shape, kept across submits
};

} // namespace

struct nrphy_pdsch_async {
  nrphy_ctx*              ctx = nullptr;
  uint32_t                nof_ports = 0, nof_subc = 0, max_tb_bytes = 0;
  size_t                  grid_bytes = 0, table_cap = 0, scratch_words = 0;
  std::vector<AsyncSlot>  slots;
  std::mutex              mutex;
  std::condition_variable idle;
  uint32_t                in_flight = 0;
  uint32_t                zero_copy = 0; // NRPHY_ASYNC_ZERO_COPY: bit 0 = the kernels read the pinned transport block, bit 1 = and write the pinned grid
};

namespace {

// Runs on a thread of the HIP runtime when everything before it on the slot's stream is done; `error` is the stream's
// status (a kernel fault or a failed copy surfaces here, and reaches the caller's handler as NRPHY_ERR_DEVICE).
void on_stream_done(hipStream_t, hipError_t error, void* arg)
{
  AsyncSlot*         slot = static_cast<AsyncSlot*>(arg);
  nrphy_pdsch_async* q    = slot->queue;
  if (error != hipSuccess) {
    slot->status = NRPHY_ERR_DEVICE;
  }
  if (slot->done != nullptr) {
    slot->done(slot->user, slot->status, slot->h_grid);
  }
  {
    std::lock_guard<std::mutex> lock(q->mutex);
    slot->busy = false;
    --q->in_flight;
  }
  q->idle.notify_all();
}

} // namespace

extern "C" int nrphy_pdsch_async_create(nrphy_ctx_t* ctx, uint32_t depth, uint32_t grid_nof_ports, uint32_t grid_nof_subc,
                                        uint32_t max_tb_bytes, nrphy_pdsch_async_t** out)
{
  if (ctx == nullptr || out == nullptr || depth == 0 || depth > 64 || grid_nof_ports == 0 || grid_nof_ports > NRPHY_MAX_PORTS ||
      grid_nof_subc == 0 || grid_nof_subc % 12 != 0 || grid_nof_subc > NRPHY_MAX_RB * 12 || max_tb_bytes == 0) {
    return NRPHY_ERR_ARGUMENT;
  }
  *out = nullptr;
  HIP_TRY(hipSetDevice(ctx->device));
  nrphy_pdsch_async* q = new (std::nothrow) nrphy_pdsch_async;
  if (q == nullptr) {
    return NRPHY_ERR_CAPACITY;
  }
  q->ctx          = ctx;
  q->nof_ports    = grid_nof_ports;
  q->nof_subc     = grid_nof_subc;
  q->max_tb_bytes = max_tb_bytes;
  q->grid_bytes   = (size_t)grid_nof_ports * NRPHY_NSYMB * grid_nof_subc * 4;
  if (const char* e = std::getenv("NRPHY_ASYNC_ZERO_COPY")) {
    q->zero_copy = (uint32_t)std::atoi(e);
  }
  q->slots.resize(depth);
  // Room for the tables of one operation (PDU descriptors, work lists, weights, RE tables, zero-fill lists: a few KB for a
  // wideband PDU; non-contiguous allocations add two bytes per RE) and for what its run rewrites (one sequence word per
  // 32 codeword bits -- at most 32 bits per RE -- plus DM-RS sequences).  An operation that needs more gets a plan with
  // memory of its own (the slow path).
  q->table_cap     = ((size_t)64 * 1024 + (size_t)NRPHY_NSYMB * grid_nof_subc * 6 + 255) & ~(size_t)255;
  q->scratch_words = (size_t)NRPHY_NSYMB * grid_nof_subc * 2 + 16384;
  if (const char* e = std::getenv("NRPHY_ASYNC_TABLE_CAP")) { // tests: a small value sends every operation down the slow path
    q->table_cap = ((size_t)std::max(256, std::atoi(e)) + 255) & ~(size_t)255;
  }
  const size_t tb_alloc = q->table_cap + (((size_t)max_tb_bytes + 7 + 255) & ~(size_t)255);
  for (AsyncSlot& s : q->slots) {
    s.queue = q;
    if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&s.d_tb, tb_alloc) != hipSuccess || hipMalloc(&s.d_grid, q->grid_bytes) != hipSuccess ||
        hipMalloc((void**)&s.d_scratch, q->scratch_words * sizeof(uint32_t)) != hipSuccess ||
        hipHostMalloc((void**)&s.h_tb, tb_alloc, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(&s.h_grid, q->grid_bytes, hipHostMallocDefault) != hipSuccess) {
      nrphy_pdsch_async_destroy(q);
      return NRPHY_ERR_DEVICE;
    }
    std::memset(s.h_tb, 0, tb_alloc);
    s.shapes = plan_shape_cache_create();
  }
  *out = q;
  return NRPHY_OK;
}

extern "C" int nrphy_pdsch_async_wait(nrphy_pdsch_async_t* q)
{
  if (q == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  std::unique_lock<std::mutex> lock(q->mutex);
  q->idle.wait(lock, [q] { return q->in_flight == 0; });
  return NRPHY_OK;
}

// Blocks while all `depth` operations are in flight: what a caller does after NRPHY_ERR_CAPACITY instead of draining
// the whole queue with nrphy_pdsch_async_wait.
extern "C" int nrphy_pdsch_async_wait_slot(nrphy_pdsch_async_t* q)
{
  if (q == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  std::unique_lock<std::mutex> lock(q->mutex);
  q->idle.wait(lock, [q] { return q->in_flight < q->slots.size(); });
  return NRPHY_OK;
}

extern "C" int nrphy_pdsch_async_destroy(nrphy_pdsch_async_t* q)
{
  if (q == nullptr) {
    return NRPHY_OK;
  }
  (void)hipSetDevice(q->ctx->device);
  nrphy_pdsch_async_wait(q);
  for (AsyncSlot& s : q->slots) {
    if (s.stream) {
      (void)hipStreamSynchronize(s.stream);
    }
    nrphy_pdsch_plan_destroy(s.plan);
    plan_shape_cache_destroy(s.shapes);
    (void)hipFree(s.d_scratch);
    (void)hipFree(s.d_tb);
    (void)hipFree(s.d_grid);
    (void)hipHostFree(s.h_tb);
    (void)hipHostFree(s.h_grid);
    if (s.stream) {
      (void)hipStreamDestroy(s.stream);
    }
  }
  delete q;
  return NRPHY_OK;
}

// n PDUs whose transport blocks sit back to back (each readable to the next multiple of 4) in the slot's staging buffer, all
// into the slot's one grid: one plan, one run.
static int submit_pdus(nrphy_pdsch_async_t* q, uint32_t n_pdu, const nrphy_pdsch_pdu_t* pdus, const uint8_t* const* tbs,
                       nrphy_pdsch_done_fn done, void* user)
{
  if (q == nullptr || pdus == nullptr || tbs == nullptr || n_pdu == 0) {
    return NRPHY_ERR_ARGUMENT;
  }
  // The transport blocks, back to back (each readable to the next multiple of 4); the plan's tables follow them.
  std::vector<uint64_t> tb_off(n_pdu);
  size_t                tb_total = 0;
  for (uint32_t i = 0; i != n_pdu; ++i) {
    if (tbs[i] == nullptr || pdus[i].tb_size_bytes == 0) {
      return NRPHY_ERR_ARGUMENT;
    }
    tb_off[i] = tb_total;
    tb_total += ((size_t)pdus[i].tb_size_bytes + 7) & ~(size_t)3;
  }
  if (tb_total > (((size_t)q->max_tb_bytes + 7) & ~(size_t)3)) {
    return NRPHY_ERR_ARGUMENT;
  }
  AsyncSlot* slot = nullptr;
  {
    std::lock_guard<std::mutex> lock(q->mutex);
    for (AsyncSlot& s : q->slots) {
      if (!s.busy) {
        slot       = &s;
        s.busy     = true;
        ++q->in_flight;
        break;
      }
    }
  }
  if (slot == nullptr) {
    return NRPHY_ERR_CAPACITY; // `depth` operations in flight: the caller waits (nrphy_pdsch_async_wait_slot) or retries
  }
  auto give_back = [q, slot](int rc) {
    {
      std::lock_guard<std::mutex> lock(q->mutex);
      slot->busy = false;
      --q->in_flight;
    }
    q->idle.notify_all();
    return rc;
  };
  if (hipSetDevice(q->ctx->device) != hipSuccess) {
    return give_back(NRPHY_ERR_DEVICE);
  }
  const bool tb_direct = (q->zero_copy & 1u) != 0, grid_direct = (q->zero_copy & 2u) != 0;
  // The slot's previous operation is complete (the slot was free): its plan goes, this one's is built in its place --
  // tables into the pinned staging, device pointers into the slot's device buffer, nothing allocated or copied here.
  nrphy_pdsch_plan_destroy(slot->plan);
  slot->plan = nullptr;
  std::vector<uint32_t> grid_of(n_pdu, 0);
  PlanPlacement         place;
  const size_t tables_at       = (tb_total + 255) & ~(size_t)255;
  place.h_tables               = slot->h_tb + tables_at;
  place.d_tables               = slot->d_tb + tables_at;
  place.table_capacity         = q->table_cap;
  place.d_scratch              = slot->d_scratch;
  place.scratch_capacity_words = q->scratch_words;
  place.cache                  = slot->shapes;
  // (the transport-block offsets the plan records are relative to the pointer nrphy_pdsch_run gets: the staging's start)
  int rc = nrphy_pdsch_plan_create_placed(q->ctx, n_pdu, pdus, tb_off.data(), grid_of.data(), 1, q->nof_ports, q->nof_subc, &place,
                                          &slot->plan);
  bool own_memory = false;
  if (rc == NRPHY_ERR_CAPACITY) {
    // Too big for the slot's table space: a plan with device memory of its own (allocation + blocking copy; freed when the
    // slot is used again).  Rare: hundreds of PDUs or RE tables for most of a fragmented grid.
    rc        = nrphy_pdsch_plan_create(q->ctx, n_pdu, pdus, tb_off.data(), grid_of.data(), 1, q->nof_ports, q->nof_subc, &slot->plan);
    own_memory = true;
  }
  if (rc != NRPHY_OK) {
    return give_back(rc);
  }
  for (uint32_t i = 0; i != n_pdu; ++i) {
    const size_t span = (i + 1 != n_pdu ? tb_off[i + 1] : tb_total) - tb_off[i];
    std::memcpy(slot->h_tb + tb_off[i], tbs[i], pdus[i].tb_size_bytes);
    std::memset(slot->h_tb + tb_off[i] + pdus[i].tb_size_bytes, 0, span - pdus[i].tb_size_bytes); // readable to the next multiple of 4
  }
  slot->done   = done;
  slot->user   = user;
  slot->status = NRPHY_OK;
  // ONE copy carries the tables and the transport blocks (zero-copy transport blocks: the tables alone -- every wave reads
  // them, they belong in device memory).
  const size_t copy_from = tb_direct ? tables_at : 0;
  const size_t copy_to   = own_memory ? tb_total : tables_at + place.table_bytes;
  if (copy_to > copy_from &&
      hipMemcpyAsync(slot->d_tb + copy_from, slot->h_tb + copy_from, copy_to - copy_from, hipMemcpyHostToDevice, slot->stream) !=
          hipSuccess) {
    return give_back(NRPHY_ERR_DEVICE);
  }
  rc = nrphy_pdsch_run(slot->plan, tb_direct ? slot->h_tb : slot->d_tb, grid_direct ? slot->h_grid : slot->d_grid, nullptr,
                       nullptr, 1, slot->stream);
  if (rc != NRPHY_OK) {
    (void)hipStreamSynchronize(slot->stream);
    return give_back(rc);
  }
  if ((!grid_direct && hipMemcpyAsync(slot->h_grid, slot->d_grid, q->grid_bytes, hipMemcpyDeviceToHost, slot->stream) != hipSuccess) ||
      hipStreamAddCallback(slot->stream, on_stream_done, slot, 0) != hipSuccess) {
    (void)hipStreamSynchronize(slot->stream);
    return give_back(NRPHY_ERR_DEVICE);
  }
  return NRPHY_OK;
}

extern "C" int nrphy_pdsch_async_submit(nrphy_pdsch_async_t* q, const nrphy_pdsch_pdu_t* pdu, const uint8_t* tb,
                                        nrphy_pdsch_done_fn done, void* user)
{
  return submit_pdus(q, 1, pdu, &tb, done, user);
}

extern "C" int nrphy_pdsch_async_submit_slot(nrphy_pdsch_async_t* q, uint32_t n_pdu, const nrphy_pdsch_pdu_t* pdus,
                                             const uint8_t* const* tbs, nrphy_pdsch_done_fn done, void* user)
{
  return submit_pdus(q, n_pdu, pdus, tbs, done, user);
}

// A ready-made completion handler: `user` points at a uint64_t that counts completions (benchmarks, tests).
extern "C" void nrphy_pdsch_async_count_done(void* user, int status, const void* grid)
{
  (void)grid;
  if (user != nullptr && status == NRPHY_OK) {
    reinterpret_cast<std::atomic<uint64_t>*>(user)->fetch_add(1, std::memory_order_relaxed);
  }
}
