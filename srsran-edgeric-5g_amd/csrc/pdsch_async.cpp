// Seam A, asynchronous host-span form: what a drop-in pdsch_processor needs to keep several PDUs in flight.
//
// pdsch_processor::process may return before the PDU is done and signal completion from any thread
// (R/include/srsran/phy/upper/channel_processors/pdsch_processor.h:157-170; the reference's own asynchronous pool:
// R/lib/phy/upper/channel_processors/pdsch_processor_asynchronous_pool.h:39-143).  A queue owns `depth` operation
// slots, each with a stream, device buffers, pinned host staging and the plans of the PDU shapes it has seen; a submit
// copies the transport block, enqueues H2D + the PDSCH kernels + D2H of the grid + a host function on the slot's
// stream and returns; the host function hands the grid to the caller's completion handler on a runtime thread.
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
  // plans of the PDU shapes this slot has run, most recent first (a plan must not run on two streams at once, so
  // every slot keeps its own)
  std::vector<std::pair<std::vector<uint8_t>, nrphy_pdsch_plan_t*>> plans;
};

constexpr size_t PLANS_PER_SLOT = 32;

} // namespace

struct nrphy_pdsch_async {
  nrphy_ctx*              ctx = nullptr;
  uint32_t                nof_ports = 0, nof_subc = 0, max_tb_bytes = 0;
  size_t                  grid_bytes = 0;
  std::vector<AsyncSlot>  slots;
  std::mutex              mutex;
  std::condition_variable idle;
  uint32_t                in_flight = 0;
  uint32_t                zero_copy = 0; // NRPHY_ASYNC_ZERO_COPY: bit 0 = the kernels read the pinned transport block, bit 1 = and write the pinned grid
};

namespace {

void signature_of(const nrphy_pdsch_pdu_t& pdu, std::vector<uint8_t>& sig)
{
  nrphy_pdsch_pdu_t copy = pdu;
  copy.precoding         = nullptr;
  sig.insert(sig.end(), reinterpret_cast<const uint8_t*>(&copy), reinterpret_cast<const uint8_t*>(&copy) + sizeof(copy));
  const size_t nw = 2 * (size_t)pdu.nof_prg * pdu.nof_ports * pdu.nof_layers * sizeof(float);
  if (pdu.precoding != nullptr && nw != 0 && nw <= 2 * NRPHY_MAX_RB * NRPHY_MAX_PORTS * NRPHY_MAX_LAYERS * sizeof(float)) {
    const uint8_t* w = reinterpret_cast<const uint8_t*>(pdu.precoding);
    sig.insert(sig.end(), w, w + nw);
  }
}

// Runs on a thread of the HIP runtime when everything before it on the slot's stream is done.
void on_stream_done(void* arg)
{
  AsyncSlot*         slot = static_cast<AsyncSlot*>(arg);
  nrphy_pdsch_async* q    = slot->queue;
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
  const size_t tb_alloc = ((size_t)max_tb_bytes + 7) & ~(size_t)3;
  for (AsyncSlot& s : q->slots) {
    s.queue = q;
    if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&s.d_tb, tb_alloc) != hipSuccess || hipMalloc(&s.d_grid, q->grid_bytes) != hipSuccess ||
        hipHostMalloc((void**)&s.h_tb, tb_alloc, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(&s.h_grid, q->grid_bytes, hipHostMallocDefault) != hipSuccess) {
      nrphy_pdsch_async_destroy(q);
      return NRPHY_ERR_DEVICE;
    }
    std::memset(s.h_tb, 0, tb_alloc);
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
    for (auto& kv : s.plans) {
      nrphy_pdsch_plan_destroy(kv.second);
    }
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
    return NRPHY_ERR_CAPACITY; // `depth` operations in flight: the caller waits or retries
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
  // The plan of this shape (every PDU's bytes and weights): the slot's own, created on first sight.
  std::vector<uint8_t> sig;
  for (uint32_t i = 0; i != n_pdu; ++i) {
    signature_of(pdus[i], sig);
  }
  nrphy_pdsch_plan_t* plan = nullptr;
  for (size_t i = 0; i != slot->plans.size(); ++i) {
    if (slot->plans[i].first == sig) {
      plan = slot->plans[i].second;
      std::rotate(slot->plans.begin(), slot->plans.begin() + i, slot->plans.begin() + i + 1); // most recent first
      break;
    }
  }
  if (plan == nullptr) {
    std::vector<uint32_t> grid_of(n_pdu, 0);
    const int rc = nrphy_pdsch_plan_create(q->ctx, n_pdu, pdus, tb_off.data(), grid_of.data(), 1, q->nof_ports, q->nof_subc, &plan);
    if (rc != NRPHY_OK) {
      return give_back(rc);
    }
    if (slot->plans.size() == PLANS_PER_SLOT) {
      nrphy_pdsch_plan_destroy(slot->plans.back().second);
      slot->plans.pop_back();
    }
    slot->plans.insert(slot->plans.begin(), std::make_pair(std::move(sig), plan));
  }
  for (uint32_t i = 0; i != n_pdu; ++i) {
    const size_t span = (i + 1 != n_pdu ? tb_off[i + 1] : tb_total) - tb_off[i];
    std::memcpy(slot->h_tb + tb_off[i], tbs[i], pdus[i].tb_size_bytes);
    std::memset(slot->h_tb + tb_off[i] + pdus[i].tb_size_bytes, 0, span - pdus[i].tb_size_bytes); // readable to the next multiple of 4
  }
  slot->done   = done;
  slot->user   = user;
  slot->status = NRPHY_OK;
  const bool tb_direct = (q->zero_copy & 1u) != 0, grid_direct = (q->zero_copy & 2u) != 0;
  if (!tb_direct && hipMemcpyAsync(slot->d_tb, slot->h_tb, tb_total, hipMemcpyHostToDevice, slot->stream) != hipSuccess) {
    return give_back(NRPHY_ERR_DEVICE);
  }
  const int rc = nrphy_pdsch_run(plan, tb_direct ? slot->h_tb : slot->d_tb, grid_direct ? slot->h_grid : slot->d_grid, nullptr,
                                 nullptr, 1, slot->stream);
  if (rc != NRPHY_OK) {
    (void)hipStreamSynchronize(slot->stream);
    return give_back(rc);
  }
  if ((!grid_direct && hipMemcpyAsync(slot->h_grid, slot->d_grid, q->grid_bytes, hipMemcpyDeviceToHost, slot->stream) != hipSuccess) ||
      hipLaunchHostFunc(slot->stream, on_stream_done, slot) != hipSuccess) {
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
