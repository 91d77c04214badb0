// Host-side internals shared by the translation units behind the C ABI (nrphy_host.cpp, dl_control_host.cpp,
// lower_phy_host.cpp): the context, its staging buffers, small helpers.  Not part of the ABI.
#pragma once

#include "nrphy_internal.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

using namespace nrphy;

#define HIP_TRY(expr)                                                                                                  \
  do {                                                                                                                 \
    if ((expr) != hipSuccess) {                                                                                        \
      return NRPHY_ERR_DEVICE;                                                                                         \
    }                                                                                                                  \
  } while (0)

namespace {

// Remainder arithmetic in GF(2)[x] / poly on the host (plan-time constants of the CRC kernels).
struct CrcField {
  uint32_t poly, order;
  uint32_t mul(uint32_t a, uint32_t b) const // a below 2^order, b any 32-bit polynomial
  {
    const uint32_t top = 1U << order;
    uint32_t       r   = 0;
    for (int k = 31; k >= 0; --k) {
      r <<= 1;
      if (r & top) {
        r ^= poly;
      }
      if ((b >> k) & 1U) {
        r ^= a;
      }
    }
    return r;
  }
  uint32_t xpow(int64_t e) const // x^e mod g; g(0) = 1 makes x invertible: x^-1 = (g - 1) / x
  {
    uint32_t base = (e >= 0) ? 2U : (poly >> 1);
    uint64_t n    = (uint64_t)(e >= 0 ? e : -e);
    uint32_t r    = 1;
    while (n) {
      if (n & 1U) {
        r = mul(r, base);
      }
      base = mul(base, base);
      n >>= 1;
    }
    return r;
  }
};
const CrcField CRC24A_FIELD = {0x1864CFBU, 24};
const CrcField CRC16_FIELD  = {0x11021U, 16};

template <typename T>
hipError_t upload(T** dptr, const void* src, size_t bytes)
{
  *dptr = nullptr;
  if (bytes == 0) {
    bytes = 16;
    hipError_t e = hipMalloc((void**)dptr, bytes);
    return e;
  }
  hipError_t e = hipMalloc((void**)dptr, bytes);
  if (e != hipSuccess) {
    return e;
  }
  return hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice);
}

} // namespace

// Environment knobs (INTEGRATION.md section 5 lists them): A/B and test aids, none changes a result.  Read ONCE, when the
// context is created -- never on a submit path, where several threads of the host may be running -- so a process that wants
// another setting creates another context.  -1 / 0 = not set.
struct Tunables {
  int      cb_dispatch     = 0;  // NRPHY_CB_DISPATCH: 1 = the mixed codeblock kernel, 2 = one launch per bucket, 0 = by plan shape
  int      crc_regions     = 0;  // NRPHY_CRC_REGIONS: 16 KiB regions per TB-CRC workgroup (0: by batch size)
  int      scr_parts_big   = 0;  // NRPHY_SCR_PARTS_BIG: parts per scrambling sequence in a batch of 128 PDUs or more (0: 1)
  uint32_t extras_nt       = 1;  // NRPHY_EXTRAS_NT=0: DM-RS / zero-fill stores with the default cache policy
  uint32_t prologue_order  = 0;  // NRPHY_PROLOGUE_ORDER=1: sequence workgroups spread among the TB-CRC workgroups
  int      decoder_pairs   = -1; // NRPHY_DECODER_PAIRS=0: one check per lane whatever the lifting size
  int      decoder_msg     = -1; // NRPHY_DECODER_MSG=0: compressed check records instead of messages per edge
  int      decoder_ldsmsg  = -1; // NRPHY_DECODER_LDSMSG: 0 = messages never in LDS, 2 = wherever a workgroup's LDS can hold them
  bool     decoder_slots_all = false; // NRPHY_DECODER_SLOTS_ALL=1: a scratch slot per codeblock (no pooling)
#ifdef NRPHY_PROBES
  // Profiling variants only (profiles/make_variant.sh ... -DNRPHY_PROBES): with these set the outputs are INCOMPLETE.
  uint32_t profile_stage   = 0;  // NRPHY_PROFILE_STAGE: the codeblock waves stop after a stage
  uint32_t ofdm_probe      = 0;  // NRPHY_OFDM_PROBE: bit 0 drops the IQ stores, bit 1 the grid loads, bit 2 takes the grids first to last
#endif
};
Tunables read_tunables();

struct nrphy_ctx {
  int          device   = 0;
  Tunables     tune;
  uint32_t     nof_cus  = 256; // compute units of the device (sizes the decoder's scratch pool)
  hipStream_t  stream   = nullptr;
  LiftedGraph* d_graphs = nullptr;
  GoldTables*  d_gold   = nullptr;
  TbCrcTables* d_tbcrc  = nullptr;
  uint32_t*    d_x1     = nullptr;
  std::map<uint32_t, float2*> d_twiddle; // exp(+j 2 pi k / N) per DFT size, built on first use (under host_mutex)
  DecoderGraph* d_dec_graph[NOF_GRAPHS] = {}; // decoder graphs, built on first use
  uint32_t*     d_dec_addr[NOF_GRAPHS]  = {}; // ... and, for even lifting sizes, the soft-bit addresses of every (edge, pair of checks)
  std::map<uint64_t, uint32_t*> d_dec_crc;     // early-stop CRC weights per (polynomial, message length)
  std::map<uint32_t, uint32_t*> d_tb_crc_w;    // transport-block CRC weights of the PUSCH assembly kernel per block size
  std::vector<LiftedGraph> graphs; // host copy (plan creation sizes the LDS staging of graph rows from it)
  // Device staging of the host-span entry points (*_host): grow-only buffers, one call at a time per context.
  // One lock for everything context-owned and shared: the staging buffers below and the lazily built tables.  The
  // host-span entry points (*_host) hold it from their first staging access to their last copy, so two of them never
  // interleave on a buffer; it is recursive because they are built from device-pointer calls that take it briefly.
  std::recursive_mutex host_mutex;
  void*      scratch[8]       = {};
  size_t     scratch_bytes[8] = {};
};

// ---- PDSCH plans in caller-owned memory (the asynchronous queue) ----------------------------------------------------------
// nrphy_pdsch_plan_create allocates device memory and copies the plan's tables there with a blocking copy: right for a
// plan that is built once and run many times, wrong on a path that sees a new PDU per call.  The placed form writes the
// tables into host memory of the caller (pinned staging it copies in stream order, together with the transport block)
// and points the plan at the device addresses they will have; it makes no HIP call.  NRPHY_ERR_CAPACITY when the plan
// does not fit.  The plan must be destroyed before the memory is reused.
struct PlanShapeCache;
PlanShapeCache* plan_shape_cache_create();
void            plan_shape_cache_destroy(PlanShapeCache* cache);
struct PlanPlacement {
  uint8_t*        h_tables = nullptr; // host memory the tables are written to
  uint8_t*        d_tables = nullptr; // device address they will be copied to (256-byte aligned)
  size_t          table_capacity = 0;
  size_t          table_bytes    = 0; // out: bytes to copy
  uint32_t*       d_scratch = nullptr; // device memory for what every run rewrites (sequences, TB-CRC shares)
  size_t          scratch_capacity_words = 0;
  PlanShapeCache* cache = nullptr;    // may be null; one cache per thread of use
};
int nrphy_pdsch_plan_create_placed(nrphy_ctx_t* ctx, uint32_t n_pdu, const nrphy_pdsch_pdu_t* pdus, const uint64_t* tb_offset,
                                   const uint32_t* grid_index, uint32_t nof_grids, uint32_t grid_nof_ports,
                                   uint32_t grid_nof_subc, PlanPlacement* place, nrphy_pdsch_plan_t** out);

namespace {

// Staging buffer `slot` of the context with room for `bytes` (reallocated only when it has to grow).
enum ScratchSlot { SCRATCH_TB = 0, SCRATCH_GRID, SCRATCH_CW_RM, SCRATCH_CW_SCR, SCRATCH_IQ, SCRATCH_SMALL,
                   SCRATCH_DECODER, SCRATCH_RX, SCRATCH_COUNT };
void* ctx_scratch(nrphy_ctx* ctx, ScratchSlot slot, size_t bytes)
{
  if (bytes > ctx->scratch_bytes[slot]) {
    (void)hipFree(ctx->scratch[slot]);
    ctx->scratch[slot]       = nullptr;
    ctx->scratch_bytes[slot] = 0;
    const size_t want        = (bytes + (bytes >> 2) + 4095) & ~(size_t)4095;
    if (hipMalloc(&ctx->scratch[slot], want) != hipSuccess) {
      return nullptr;
    }
    ctx->scratch_bytes[slot] = want;
  }
  return ctx->scratch[slot];
}

// Device staging that lives for ONE call: host-built work lists a kernel of this call reads.  Allocated and released
// in stream order (hipMallocAsync / hipFreeAsync), so calls in flight on different streams never share a buffer and
// nothing waits for the device.  Usage: alloc(), copy + launch on the same stream, then the destructor frees.
class StreamStaging
{
public:
  explicit StreamStaging(hipStream_t s) : stream(s) {}
  StreamStaging(const StreamStaging&)            = delete;
  StreamStaging& operator=(const StreamStaging&) = delete;
  ~StreamStaging()
  {
    if (ptr != nullptr) {
      (void)hipFreeAsync(ptr, stream);
    }
  }
  void* alloc(size_t bytes)
  {
    if (hipMallocAsync(&ptr, std::max<size_t>(bytes, 16), stream) != hipSuccess) {
      ptr = nullptr;
    }
    return ptr;
  }

private:
  hipStream_t stream;
  void*       ptr = nullptr;
};

} // namespace
