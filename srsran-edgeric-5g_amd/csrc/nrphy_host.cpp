// Host side of libmi355nrphy.so: the C ABI of include/mi355_nrphy.h.
//
// Per-PDU scalar derivation (what pdsch_processor_impl / ldpc_segmenter_impl / ldpc_rate_matcher_impl compute on
// the CPU before their loops), plan construction and kernel launches.  No compute happens here and there is no CPU
// fallback: every entry point that produces PHY output needs a HIP device.
#include "nrphy_host_internal.h"
#include "nrphy_trace.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <array>
#include <map>
#include <mutex>
#include <new>
#include <vector>

using namespace nrphy;

namespace {

struct nr_ldpc_edge_t {
  uint8_t  row;
  uint8_t  col;
  uint16_t shift[8];
};
#include "nr_ldpc_bg.inc"

const uint16_t LIFTING_SIZES[NOF_LIFTING_SIZES] = {
    2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13,  14,  15,  16,  18,  20,  22,  24,  26,  28,  30,  32,  36,  40, 44,
    48, 52, 56, 60, 64, 72, 80, 88, 96, 104, 112, 120, 128, 144, 160, 176, 192, 208, 224, 240, 256, 288, 320, 352, 384};

int lifting_position(unsigned zc)
{
  for (int i = 0; i != NOF_LIFTING_SIZES; ++i) {
    if (LIFTING_SIZES[i] == zc) {
      return i;
    }
  }
  return -1;
}

// Lifting-set index i_LS: Zc = a * 2^j, a in {2, 3, 5, 7, 9, 11, 13, 15} (TS 38.212 Table 5.3.2-1).
int lifting_set_index(unsigned zc)
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

unsigned divide_ceil(unsigned a, unsigned b)
{
  return (a + b - 1) / b;
}

bool mask_test(const uint64_t* w, unsigned i)
{
  return (w[i >> 6] >> (i & 63)) & 1U;
}

int mask_lowest(const uint64_t* w)
{
  for (unsigned i = 0; i != 64 * NRPHY_PRB_WORDS; ++i) {
    if (mask_test(w, i)) {
      return (int)i;
    }
  }
  return -1;
}

int mask_highest(const uint64_t* w)
{
  int hi = -1;
  for (unsigned i = 0; i != 64 * NRPHY_PRB_WORDS; ++i) {
    if (mask_test(w, i)) {
      hi = (int)i;
    }
  }
  return hi;
}

void build_lifted_graph(unsigned bg, unsigned zc, LiftedGraph& g)
{
  const nr_ldpc_edge_t* edges   = (bg == 1) ? NR_LDPC_BG1_EDGES : NR_LDPC_BG2_EDGES;
  const unsigned        n_edges = (bg == 1) ? NR_LDPC_BG1_NOF_EDGES : NR_LDPC_BG2_NOF_EDGES;
  const unsigned        kb      = (bg == 1) ? 22 : 10;
  const unsigned        rows    = (bg == 1) ? 46 : 42;
  const int             ils     = lifting_set_index(zc);
  std::memset(&g, 0, sizeof(g));
  int      core_shift[4] = {-1, -1, -1, -1};
  unsigned count         = 0;
  for (unsigned m = 0; m != rows; ++m) {
    g.row_ptr[m] = (uint16_t)count;
    for (unsigned e = 0; e != n_edges; ++e) {
      if (edges[e].row != m) {
        continue;
      }
      unsigned col = edges[e].col, shift = edges[e].shift[ils] % zc;
      if (m < 4) {
        if (col == kb) {
          core_shift[m] = (int)shift;
        }
        if (col >= kb) {
          continue; // core rows: systematic part only, the parity part is solved in closed form
        }
      } else if (col >= kb + 4) {
        continue; // the identity column of an extension row is its own parity block
      }
      if (m < 4 && (zc & 31U) == 0) {
        // Core rows of word-aligned lifting sizes read a doubled copy of the systematic blocks (ldpc_device.h,
        // core_row_word_dbl): byte offset of word q' of the doubled block, and the funnel shift s'.
        const unsigned wpb = zc / 32, sh = shift & 31U, q = shift >> 5;
        const unsigned qd = (sh != 0) ? q : (q + wpb - 1) % wpb, sd = (32U - sh) & 31U;
        g.edge[count++] = ((4U * (2U * wpb * col + qd)) << 16) | sd;
        continue;
      }
      g.edge[count++] = ((col * zc) << 16) | shift; // bit offset of the block in the codeblock (< 2^16), lifted shift
    }
  }
  for (unsigned m = rows; m != MAX_BG_ROWS + 2; ++m) {
    g.row_ptr[m] = (uint16_t)count;
  }
  unsigned mid = (core_shift[1] >= 0) ? 1 : 2;
  int      s0 = core_shift[0], s3 = core_shift[3], sm = core_shift[mid];
  g.core_mid = (uint16_t)mid;
  g.core_s0  = (uint16_t)s0;
  g.core_s3  = (uint16_t)s3;
  g.core_b   = (uint16_t)((s0 == s3) ? sm : ((s0 == sm) ? s3 : s0));
}

// ---- Gold sequence tables (TS 38.211 Section 5.2.1) -----------------------------------------------------------
// 31x31 matrices over GF(2) as 31 row masks.
void mat_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) // out = a * b
{
  uint32_t bt[31];                      // columns of b
  for (int c = 0; c != 31; ++c) {
    uint32_t col = 0;
    for (int r = 0; r != 31; ++r) {
      col |= ((b[r] >> c) & 1U) << r;
    }
    bt[c] = col;
  }
  uint32_t tmp[31];
  for (int r = 0; r != 31; ++r) {
    uint32_t row = 0;
    for (int c = 0; c != 31; ++c) {
      row |= (uint32_t)(__builtin_popcount(a[r] & bt[c]) & 1) << c;
    }
    tmp[r] = row;
  }
  std::memcpy(out, tmp, sizeof(tmp));
}

// GF(2)[x] / g arithmetic of the transport-block CRC polynomials (order 24 or 16, `poly` with its leading term).

void build_tbcrc_tables(TbCrcTables& t)
{
  const CrcField* field[2] = {&CRC24A_FIELD, &CRC16_FIELD};
  for (unsigned s = 0; s != 2; ++s) {
    const CrcField& f  = *field[s];
    const uint32_t  y32 = f.xpow(32), y1k = f.xpow(32 * 1024), y8k = f.xpow(128 * 64), yz = f.xpow(8 * (int64_t)TB_CRC_REGION_BYTES);
    for (unsigned k = 0; k != 4; ++k) {
      for (uint32_t b = 0; b != 256; ++b) {
        t.y32[s][k][b] = f.mul(y32, b << (8 * k));
        t.y1k[s][k][b] = f.mul(y1k, b << (8 * k));
        t.y8k[s][k][b] = f.mul(y8k, b << (8 * k));
        t.yz[s][k][b]  = f.mul(yz, b << (8 * k));
      }
    }
    for (unsigned l = 0; l != 64; ++l) {
      t.lane[s][l] = f.xpow(128 * (63 - (int)l));
    }
  }
}

void build_gold_tables(GoldTables& t, std::vector<uint32_t>& x1_words)
{
  std::memset(&t, 0, sizeof(t));
  // One step of x2: state'[j] = state[j+1], state'[30] = state[3]^state[2]^state[1]^state[0].
  uint32_t m[31];
  for (int r = 0; r != 30; ++r) {
    m[r] = 1U << (r + 1);
  }
  m[30] = 0xF;
  for (int k = 0; k != GOLD_JUMP_BITS; ++k) {
    std::memcpy(t.x2_jump[k], m, sizeof(m));
    mat_mul(m, m, m);
  }
  // The next 992 bits are linear in the state: x2_head[t][w] is the state mask of bit 32 w + t.
  for (int i = 0; i != 31; ++i) {
    uint32_t st = 1U << i;
    for (int n = 0; n != 992; ++n) {
      if (st & 1U) {
        t.x2_head[n & 31][n >> 5] |= 1U << i;
      }
      uint32_t f = ((st >> 3) ^ (st >> 2) ^ (st >> 1) ^ st) & 1U;
      st         = (st >> 1) | (f << 30);
    }
  }
  // x^(32 m) mod CRC24B and the CRC24B byte table.
  const uint32_t poly = 0x1800063U, top = 1U << 24;
  uint32_t       v    = 1;
  for (int i = 0; i != CRC_POW_WORDS; ++i) {
    t.crc24b_pow32[i] = v;
    for (int s = 0; s != 32; ++s) {
      v <<= 1;
      if (v & top) {
        v ^= poly;
      }
    }
  }
  for (uint32_t b = 0; b != 256; ++b) {
    uint32_t r = b << 16;
    for (int k = 0; k != 8; ++k) {
      r <<= 1;
      if (r & top) {
        r ^= poly;
      }
    }
    t.crc24b_table[b] = r & (top - 1);
  }
  // (b x^(8k) x^24) mod g for k = 0..3: table k is table k - 1 advanced by one byte.
  for (uint32_t b = 0; b != 256; ++b) {
    uint32_t r            = t.crc24b_table[b];
    t.crc24b_slice[0][b] = r;
    for (int k = 1; k != 4; ++k) {
      for (int s = 0; s != 8; ++s) {
        r <<= 1;
        if (r & top) {
          r ^= poly;
        }
      }
      t.crc24b_slice[k][b] = r;
    }
  }
  {
    auto mulx = [&](uint32_t a, unsigned e) { // a x^e mod g
      for (unsigned s = 0; s != e; ++s) {
        a <<= 1;
        if (a & top) {
          a ^= poly;
        }
      }
      return a;
    };
    for (unsigned m = 0; m != CRC_POW_WORDS; ++m) {
      for (unsigned n = 0; n != 6; ++n) {
        uint32_t e = mulx(t.crc24b_pow32[m], 4 * n); // x^(32 m + 4 n)
        // (v x^(4n)) x^(32m) for the 16 values of v, from the four single-bit products.
        uint32_t bit[4];
        for (unsigned k = 0; k != 4; ++k) {
          bit[k] = mulx(e, k);
        }
        for (uint32_t nib = 0; nib != 16; ++nib) {
          uint32_t acc = 0;
          for (unsigned k = 0; k != 4; ++k) {
            acc ^= ((nib >> k) & 1U) ? bit[k] : 0U;
          }
          t.crc24b_mul[m][n][nib] = acc;
        }
      }
    }
  }
  // Modulation tables: d = (1-2b0)[2^(h-1) - (1-2b2)[2^(h-2) - ...]] on the even bits, same on the odd bits for
  // the imaginary part (TS 38.211 Sections 5.1.3-5.1.6), h = Qm / 2.
  for (unsigned q = 0; q != 4; ++q) {
    const unsigned qm = 2 * (q + 1), h = q + 1;
    for (unsigned idx = 0; idx != (1U << qm); ++idx) {
      int re = 1 - 2 * (int)((idx >> 1) & 1U), im = 1 - 2 * (int)(idx & 1U);
      for (unsigned lvl = 1; lvl < h; ++lvl) {
        re = (1 - 2 * (int)((idx >> (2 * lvl + 1)) & 1U)) * ((1 << lvl) - re);
        im = (1 - 2 * (int)((idx >> (2 * lvl)) & 1U)) * ((1 << lvl) - im);
      }
      t.qam_lut[q][idx] = make_float2((float)re, (float)im);
    }
  }
  // x1(n + 1600), MSB-first words: x1(n+31) = x1(n+3) ^ x1(n), x1(0) = 1.
  x1_words.assign(GOLD_X1_WORDS, 0);
  uint32_t x1 = 1;
  for (int i = 0; i != 1600; ++i) {
    uint32_t f = ((x1 >> 3) ^ x1) & 1U;
    x1         = (x1 >> 1) | (f << 30);
  }
  for (size_t n = 0; n != (size_t)GOLD_X1_WORDS * 32; ++n) {
    if (x1 & 1U) {
      x1_words[n >> 5] |= 0x80000000U >> (n & 31);
    }
    uint32_t f = ((x1 >> 3) ^ x1) & 1U;
    x1         = (x1 >> 1) | (f << 30);
  }
}


// All tables of a plan in ONE device allocation filled by ONE copy (a plan of a single PDU used to spend most of its
// creation time in a dozen hipMalloc / hipMemcpy pairs).  add() registers a table; commit() allocates, copies and
// points every registered device pointer into the block, followed by `scratch_bytes` of uninitialised device memory.
class DeviceArena
{
public:
  template <typename T>
  void add(T** dptr, const void* src, size_t bytes)
  {
    items.push_back({(void**)dptr, src, bytes, total});
    total += (bytes + 255) & ~(size_t)255;
  }
  size_t bytes() const { return std::max<size_t>(total, 256); }
  // The same layout in memory the caller owns: the tables are written to `h_base`, the device pointers point into
  // `d_base`; copying [h_base, h_base + bytes()) there is the caller's business (no HIP call here).
  void place(uint8_t* h_base, uint8_t* d_base)
  {
    for (const Item& it : items) {
      if (it.bytes != 0) {
        std::memcpy(h_base + it.offset, it.src, it.bytes);
      }
      *it.dptr = d_base + it.offset;
    }
  }
  hipError_t commit(void** base, size_t scratch_bytes, void** scratch)
  {
    std::vector<uint8_t> staging(std::max<size_t>(total, 256), 0);
    for (const Item& it : items) {
      if (it.bytes != 0) {
        std::memcpy(&staging[it.offset], it.src, it.bytes);
      }
    }
    hipError_t e = hipMalloc(base, staging.size() + scratch_bytes);
    if (e != hipSuccess) {
      return e;
    }
    for (const Item& it : items) {
      *it.dptr = (uint8_t*)*base + it.offset;
    }
    *scratch = (uint8_t*)*base + staging.size();
    return hipMemcpy(*base, staging.data(), staging.size(), hipMemcpyHostToDevice);
  }

private:
  struct Item {
    void**      dptr;
    const void* src;
    size_t      bytes, offset;
  };
  std::vector<Item> items;
  size_t            total = 0;
};

} // namespace


namespace {


} // namespace

struct nrphy_pdsch_plan {
  nrphy_ctx*            ctx = nullptr;
  std::vector<PduDev>   pdus;
  std::vector<uint64_t> cw_offset;
  uint64_t              cw_bits = 0;
  uint32_t              nof_grids = 0, grid_nof_ports = 0, grid_nof_subc = 0;
  void*                 d_arena = nullptr; // the one device allocation every d_* pointer below points into
  bool                  arena_external = false; // the tables live in memory the caller owns (nrphy_pdsch_plan_create_placed)
  PduDev*               d_pdus = nullptr;
  CbWork*               d_work = nullptr;
  DmrsWork*             d_dmrs = nullptr;
  float*                d_weights = nullptr;
  uint16_t*             d_re_table = nullptr;
  uint32_t*             d_tb_crc = nullptr;
  CrcWork*              d_crc_work = nullptr;
  ScrWork*              d_scr_work = nullptr;
  uint32_t              n_scr_work = 0;
  ZeroWork*             d_zero_work = nullptr;
  ZeroSeg*              d_zero_segs = nullptr;
  uint32_t*             d_scr = nullptr;    // scrambling sequences, rewritten by every run's prologue
  uint64_t              scr_words = 0;   // words of the run's scratch: DM-RS sequences, then the work items' seeds
  uint64_t              seed_offset = 0; // where the seeds start
  uint32_t              n_zero_work = 0;
  bool                  encode_only = false;   // seam B plan: no RE mapping, nrphy_pdsch_run only with d_grid = NULL
  bool                  dmrs_separate = false; // DM-RS must overwrite data RE: keep it in its own, later launch
  uint32_t              n_work = 0, n_dmrs = 0, n_cb = 0, n_crc_work = 0;
  uint32_t              lds_lin_words = 0, lds_symb_words = 0, lds_graph_words = 0, lds_u_words = 0;
  uint32_t              bucket_begin[CB_BUCKETS + 1] = {}; // work items sorted by (modulation order, layers)
  // A batch with several big buckets runs their launches side by side on streams of the plan's own (created at the first
  // such run), forked from and joined to the caller's stream with events.
  static constexpr uint32_t MAX_AUX = 3;
  hipStream_t           aux_stream[MAX_AUX] = {};
  hipEvent_t            fork_event = nullptr, join_event[MAX_AUX] = {};
  uint32_t              n_aux = 0;
  std::vector<hipEvent_t> events; // 4 per recorded run: start, after tb_crc, after codeblocks, after dmrs (only when a DM-RS launch follows)
  std::vector<uint8_t>  timed_dmrs; // per recorded run: its fourth event was recorded
  uint32_t              timed_runs = 0, max_timed_runs = 0;
  uint32_t              timing_stride = 1, timing_counter = 0; // every timing_stride-th run is recorded
};

static bool plan_side_streams(nrphy_pdsch_plan* plan, uint32_t want);

struct nrphy_ofdm_plan {
  nrphy_ctx*          ctx = nullptr;
  nrphy_ofdm_config_t cfg;
  uint32_t            nof_ports = 0, nsymb = 14, slot_stride = 0, nsym_subframe = 0;
  const float2*       d_twiddle = nullptr; // the context's table of the DFT size
  float2*             d_phase = nullptr;
  float2*             d_phase_rx = nullptr; // demodulator: conjugate phase x scale (phase_compensation_lut, is_tx = false)
  uint32_t*           d_cp = nullptr;
  uint32_t*           d_off = nullptr;
  std::vector<uint32_t> cp, off;
  std::map<uint32_t, float2*> d_window_phase; // demodulator: per window offset, built on first use
  std::mutex          window_mutex;           // guards the map (the host-span entry points already hold ctx->host_mutex)
  std::vector<hipEvent_t> events; // 2 per recorded run
  uint32_t            timed_runs = 0, max_timed_runs = 0;
  uint32_t            timing_stride = 1, timing_counter = 0; // every timing_stride-th run is recorded
  uint4*              d_wire_partials = nullptr; // wire-format runs with measurements: one record per workgroup
  size_t              wire_partials_cap = 0;     // records
};

// ================================================================================================================
// Scalars
// ================================================================================================================
extern "C" const char* nrphy_version(void)
{
  return "mi355-nrphy 0.1 (gfx950)";
}

extern "C" int nrphy_trace_enabled(void)
{
  return trace_enabled() ? 1 : 0;
}

extern "C" const char* nrphy_strerror(int status)
{
  switch (status) {
    case NRPHY_OK:
      return "ok";
    case NRPHY_ERR_INVALID_PDU:
      return "invalid PDSCH PDU (pdsch_pdu_validator::is_valid is false)";
    case NRPHY_ERR_ARGUMENT:
      return "invalid argument";
    case NRPHY_ERR_DEVICE:
      return "HIP device error (no GPU or kernel launch failure); there is no CPU fallback";
    case NRPHY_ERR_CAPACITY:
      return "capacity exceeded";
    default:
      return "unknown status";
  }
}

// pdsch_processor_validator_impl::is_valid (R/lib/phy/upper/channel_processors/pdsch_processor_validator_impl.cpp:99-181),
// plus the checks the reference leaves to assertions deeper in the chain (modulation, rv, base graph, sizes).
extern "C" int nrphy_pdsch_validate(const nrphy_pdsch_pdu_t* pdu)
{
  if (pdu == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  const unsigned nsymb = pdu->cp ? 12 : 14;
  const int      lo = mask_lowest(pdu->prb_mask), hi = mask_highest(pdu->prb_mask);
  if (lo < 0 || (unsigned)lo < pdu->bwp_start_rb || (unsigned)hi >= pdu->bwp_start_rb + pdu->bwp_size_rb ||
      pdu->bwp_start_rb + pdu->bwp_size_rb > NRPHY_MAX_RB) {
    return NRPHY_ERR_INVALID_PDU; // freq_alloc.is_bwp_valid
  }
  if (pdu->dmrs_symbol_mask == 0 || (pdu->dmrs_symbol_mask >> nsymb) != 0) {
    return NRPHY_ERR_INVALID_PDU;
  }
  const unsigned first_dmrs = (unsigned)__builtin_ctz(pdu->dmrs_symbol_mask);
  const unsigned last_dmrs  = 31U - (unsigned)__builtin_clz(pdu->dmrs_symbol_mask);
  if (first_dmrs < pdu->start_symbol_index || last_dmrs >= pdu->start_symbol_index + pdu->nof_symbols ||
      nsymb < pdu->start_symbol_index + pdu->nof_symbols) {
    return NRPHY_ERR_INVALID_PDU;
  }
  if (pdu->dmrs_type != 1 || pdu->nof_cdm_groups_without_data > 2 || !pdu->vrb_contiguous) {
    return NRPHY_ERR_INVALID_PDU;
  }
  for (int prb = lo; prb <= hi; ++prb) { // "only contiguous allocation": the flag and the mask must tell the same story
    if (!mask_test(pdu->prb_mask, (unsigned)prb)) {
      return NRPHY_ERR_INVALID_PDU;
    }
  }
  if (pdu->nof_ports == 0 || pdu->nof_ports > NRPHY_MAX_PORTS || pdu->nof_layers == 0 ||
      pdu->nof_layers > pdu->nof_ports) {
    return NRPHY_ERR_INVALID_PDU;
  }
  if (pdu->nof_codewords != 1 || pdu->tbs_lbrm_bytes == 0 || pdu->nof_reserved > NRPHY_MAX_RESERVED) {
    return NRPHY_ERR_INVALID_PDU;
  }
  for (unsigned r = 0; r != pdu->nof_reserved; ++r) {
    if (pdu->reserved[r].symbol_mask & pdu->dmrs_symbol_mask) {
      return NRPHY_ERR_INVALID_PDU; // check_dmrs_and_reserved_collision
    }
  }
  if ((pdu->qm != 2 && pdu->qm != 4 && pdu->qm != 6 && pdu->qm != 8) || pdu->rv > 3 ||
      (pdu->ldpc_base_graph != 1 && pdu->ldpc_base_graph != 2) || pdu->tb_size_bytes == 0 ||
      pdu->tb_size_bytes > NRPHY_MAX_TB_BYTES || pdu->nof_prg == 0 || pdu->nof_prg > NRPHY_MAX_PRG || pdu->prg_size_rb == 0 || pdu->prg_size_rb > NRPHY_MAX_RB || pdu->precoding == nullptr || pdu->cp > 1) {
    // (nof_prg sizes the read of the caller's weight array: at most one PRG per resource block)
    return NRPHY_ERR_INVALID_PDU;
  }
  return NRPHY_OK;
}

namespace {

// Data-RE mask of OFDM symbol l: allocation minus reserved minus DM-RS pattern
// (pdsch_modulator_impl.cpp:52-106, re_pattern.cpp:27-60, dmrs_mapping.h:69-123).
void data_re_mask(const nrphy_pdsch_pdu_t& pdu, unsigned l, std::vector<uint8_t>& mask)
{
  std::fill(mask.begin(), mask.end(), 0);
  if (l < pdu.start_symbol_index || l >= pdu.start_symbol_index + pdu.nof_symbols) {
    return;
  }
  const unsigned nof_prb = (unsigned)mask.size() / 12;
  for (unsigned p = 0; p != nof_prb; ++p) {
    if (mask_test(pdu.prb_mask, p)) {
      std::fill(mask.begin() + 12 * p, mask.begin() + 12 * p + 12, 1);
    }
  }
  for (unsigned r = 0; r != pdu.nof_reserved; ++r) {
    const nrphy_re_pattern_t& pat = pdu.reserved[r];
    if (!((pat.symbol_mask >> l) & 1U)) {
      continue;
    }
    for (unsigned p = 0; p != nof_prb; ++p) {
      if (!mask_test(pat.prb_mask, p)) {
        continue;
      }
      for (unsigned k = 0; k != 12; ++k) {
        if ((pat.re_mask >> k) & 1U) {
          mask[12 * p + k] = 0;
        }
      }
    }
  }
  if ((pdu.dmrs_symbol_mask >> l) & 1U) {
    for (unsigned p = pdu.bwp_start_rb; p < pdu.bwp_start_rb + pdu.bwp_size_rb && p < nof_prb; ++p) {
      for (unsigned k = 0; k != 12; ++k) {
        if ((k % 2) < pdu.nof_cdm_groups_without_data) {
          mask[12 * p + k] = 0;
        }
      }
    }
  }
}

// nref_override: the limited-buffer size given directly (seam B hands N_ref, not TBS_LBRM); nullptr = from the PDU.
void derive(const nrphy_pdsch_pdu_t& pdu, unsigned nof_re, nrphy_pdsch_derived_t& d,
            const uint32_t* nref_override = nullptr)
{
  const unsigned bg      = pdu.ldpc_base_graph;
  const unsigned tb_bits = 8 * pdu.tb_size_bytes;
  const unsigned tb_crc  = (tb_bits <= 3824) ? 16 : 24;
  const unsigned b       = tb_bits + tb_crc;
  const unsigned kcb     = (bg == 1) ? 8448 : 3840;
  const unsigned C       = (b <= kcb) ? 1 : divide_ceil(b, kcb - 24);
  const unsigned b_out   = b + ((C > 1) ? 24 * C : 0);
  unsigned       ref_len = 22;
  if (bg == 2) {
    ref_len = (b > 640) ? 10 : (b > 560) ? 9 : (b > 192) ? 8 : 6;
  }
  unsigned zc = 0;
  for (unsigned i = 0; i != NOF_LIFTING_SIZES; ++i) {
    if (LIFTING_SIZES[i] * C * ref_len >= b_out) {
      zc = LIFTING_SIZES[i];
      break;
    }
  }
  const unsigned K      = ((bg == 1) ? 22 : 10) * zc;
  const unsigned cb_crc = (C > 1) ? 24 : 0;
  const unsigned info   = divide_ceil(b_out, C) - cb_crc;
  const unsigned N      = ((bg == 1) ? 66 : 50) * zc;
  uint64_t       nref   = ((uint64_t)pdu.tbs_lbrm_bytes * 8 * 3) / (2 * C); // ldpc::compute_N_ref
  if (nref_override != nullptr) {
    nref = *nref_override;
  }
  nref                  = std::min<uint64_t>(nref, 66 * 384);
  d.nof_re              = nof_re;
  d.nof_codeblocks      = C;
  d.lifting_size        = zc;
  d.segment_length      = K;
  d.cb_info_bits        = info;
  d.nof_filler_bits     = K - info - cb_crc;
  d.nof_tb_crc_bits     = tb_crc;
  d.nof_cb_crc_bits     = cb_crc;
  d.zero_pad            = (info + cb_crc) * C - b_out;
  d.full_length         = N;
  d.n_ref               = (uint32_t)nref;
  d.n_cb                = (nref > 0 && nref < N) ? (uint32_t)nref : N;
  static const double shift_bg1[4] = {0, 17, 33, 56};
  static const double shift_bg2[4] = {0, 13, 25, 43};
  const double tmp      = (((bg == 1) ? shift_bg1 : shift_bg2)[pdu.rv] * d.n_cb) / N; // ldpc_rate_matcher_impl.cpp:89-90
  d.k0                  = (uint32_t)((uint16_t)std::floor(tmp)) * zc;
  d.nof_short_segments  = C - (nof_re % C);
  d.rm_length_short     = (nof_re / C) * pdu.nof_layers * pdu.qm;
  d.rm_length_long      = divide_ceil(nof_re, C) * pdu.nof_layers * pdu.qm;
  d.codeword_bits       = nof_re * pdu.nof_layers * pdu.qm;
}

unsigned count_data_re(const nrphy_pdsch_pdu_t& pdu)
{
  std::vector<uint8_t> mask(NRPHY_MAX_RB * 12);
  unsigned             count = 0;
  for (unsigned l = 0; l != NRPHY_NSYMB; ++l) {
    data_re_mask(pdu, l, mask);
    for (uint8_t m : mask) {
      count += m;
    }
  }
  return count;
}

} // namespace

extern "C" int nrphy_pdsch_derive(const nrphy_pdsch_pdu_t* pdu, nrphy_pdsch_derived_t* out)
{
  if (pdu == nullptr || out == nullptr || pdu->tb_size_bytes == 0 || pdu->nof_layers == 0 || pdu->qm == 0 ||
      (pdu->ldpc_base_graph != 1 && pdu->ldpc_base_graph != 2) || pdu->rv > 3) {
    return NRPHY_ERR_ARGUMENT;
  }
  derive(*pdu, count_data_re(*pdu), *out);
  return NRPHY_OK;
}

// TS 38.214 Section 5.1.3.2 (tbs_calculator_calculate, R/lib/ran/sch/tbs_calculator.cpp:31-144).
extern "C" uint32_t nrphy_tbs_calculate(uint32_t nof_symb_sh, uint32_t nof_dmrs_prb, uint32_t nof_oh_prb, uint32_t qm,
                                        float target_code_rate, uint32_t nof_layers, uint32_t n_prb)
{
  static const uint16_t table[93] = {
      24,   32,   40,   48,   56,   64,   72,   80,   88,   96,   104,  112,  120,  128,  136,  144,  152,  160,  168,
      176,  184,  192,  208,  224,  240,  256,  272,  288,  304,  320,  336,  352,  368,  384,  408,  432,  456,  480,
      504,  528,  552,  576,  608,  640,  672,  704,  736,  768,  808,  848,  888,  928,  984,  1032, 1064, 1128, 1160,
      1192, 1224, 1256, 1288, 1320, 1352, 1416, 1480, 1544, 1608, 1672, 1736, 1800, 1864, 1928, 2024, 2088, 2152, 2216,
      2280, 2408, 2472, 2536, 2600, 2664, 2728, 2792, 2856, 2976, 3104, 3240, 3368, 3496, 3624, 3752, 3824};
  const unsigned nof_re_prime = 12 * nof_symb_sh - nof_dmrs_prb - nof_oh_prb;
  const unsigned nof_re       = std::min(nof_re_prime, 156U) * n_prb;
  const float    tcr          = target_code_rate * (1.F / 1024);
  const float    nof_info     = 1.0F * (float)nof_re * tcr * (float)qm * (float)nof_layers;
  if (nof_info <= 3824) {
    unsigned n = 3;
    if (nof_info > 512) {
      n = (unsigned)std::floor(std::log2(nof_info)) - 6U;
    }
    const unsigned p2    = 1U << n;
    const unsigned prime = std::max(24U, p2 * (unsigned)std::floor(nof_info / (float)p2));
    for (uint16_t v : table) {
      if (v >= prime) {
        return v;
      }
    }
    return 3824;
  }
  const unsigned n     = (unsigned)(std::floor(std::log2(nof_info - 24)) - 5.0F);
  const unsigned p2    = 1U << n;
  const unsigned prime = std::max(3840U, p2 * (unsigned)std::round((nof_info - 24) / (float)p2));
  unsigned       C     = 1;
  if (tcr <= 0.25F) {
    C = divide_ceil(prime + 24, 3816);
  } else if (prime > 8424) {
    C = divide_ceil(prime + 24, 8424);
  }
  return 8 * C * divide_ceil(prime + 24, 8 * C) - 24;
}

namespace {

// cyclic_prefix::get_length + phy_time_unit::to_samples (R/include/srsran/ran/cyclic_prefix.h:93-104,
// phy_time_unit.h:100-111): units of kappa*Tc at 15 kHz -> samples at 15 kHz * 2^mu * dft_size.
unsigned cp_length(const nrphy_ofdm_config_t& c, unsigned symbol_index)
{
  const unsigned mu    = c.numerology;
  unsigned       units = 144U >> mu;
  if (c.cp) {
    units = 512U >> mu;
  } else if (symbol_index == 0 || symbol_index == 7U * (1U << mu)) {
    units += 16;
  }
  return (unsigned)(((uint64_t)units * c.dft_size * (1U << mu)) / 2048U);
}

} // namespace

extern "C" uint32_t nrphy_ofdm_symbol_size(const nrphy_ofdm_config_t* cfg, uint32_t symbol_index)
{
  return cp_length(*cfg, symbol_index) + cfg->dft_size;
}

extern "C" uint32_t nrphy_ofdm_slot_size(const nrphy_ofdm_config_t* cfg, uint32_t slot_index)
{
  const unsigned nsymb = cfg->cp ? 12 : 14;
  unsigned       n     = 0;
  for (unsigned l = 0; l != nsymb; ++l) {
    n += nrphy_ofdm_symbol_size(cfg, nsymb * slot_index + l);
  }
  return n;
}

// ================================================================================================================
// Context
// ================================================================================================================

Tunables read_tunables()
{
  Tunables t;
  auto     number = [](const char* name, int unset) {
    const char* e = std::getenv(name);
    return (e != nullptr && e[0] != 0) ? std::atoi(e) : unset;
  };
  t.cb_dispatch       = number("NRPHY_CB_DISPATCH", 0);
  t.crc_regions       = number("NRPHY_CRC_REGIONS", 0);
  t.scr_parts_big     = number("NRPHY_SCR_PARTS_BIG", 0);
  t.extras_nt         = (uint32_t)number("NRPHY_EXTRAS_NT", 1);
  t.prologue_order    = (uint32_t)number("NRPHY_PROLOGUE_ORDER", 0);
  t.decoder_pairs     = number("NRPHY_DECODER_PAIRS", -1);
  t.decoder_msg       = number("NRPHY_DECODER_MSG", -1);
  t.decoder_ldsmsg    = number("NRPHY_DECODER_LDSMSG", -1);
  t.decoder_slots_all = number("NRPHY_DECODER_SLOTS_ALL", 0) == 1;
#ifdef NRPHY_PROBES
  t.profile_stage = (uint32_t)number("NRPHY_PROFILE_STAGE", 0);
  t.ofdm_probe    = (uint32_t)number("NRPHY_OFDM_PROBE", 0);
#endif
  return t;
}

extern "C" int nrphy_create(nrphy_ctx_t** out, int device_id)
{
  if (out == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  *out      = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count) {
    return NRPHY_ERR_DEVICE;
  }
  HIP_TRY(hipSetDevice(device_id));
  nrphy_ctx* ctx = new (std::nothrow) nrphy_ctx;
  if (ctx == nullptr) {
    return NRPHY_ERR_CAPACITY;
  }
  ctx->device = device_id;
  ctx->tune   = read_tunables();
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) {
      ctx->nof_cus = (uint32_t)prop.multiProcessorCount;
    }
  }
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return NRPHY_ERR_DEVICE;
  }
  std::vector<LiftedGraph> graphs(NOF_GRAPHS);
  for (unsigned bg = 1; bg <= 2; ++bg) {
    for (int i = 0; i != NOF_LIFTING_SIZES; ++i) {
      build_lifted_graph(bg, LIFTING_SIZES[i], graphs[(bg - 1) * NOF_LIFTING_SIZES + i]);
    }
  }
  GoldTables            gold;
  std::vector<uint32_t> x1;
  build_gold_tables(gold, x1);
  ctx->graphs = graphs;
  std::vector<TbCrcTables> tbcrc(1);
  build_tbcrc_tables(tbcrc[0]);
  if (upload(&ctx->d_tbcrc, tbcrc.data(), sizeof(TbCrcTables)) != hipSuccess ||
      upload(&ctx->d_graphs, graphs.data(), graphs.size() * sizeof(LiftedGraph)) != hipSuccess ||
      upload(&ctx->d_gold, &gold, sizeof(gold)) != hipSuccess ||
      upload(&ctx->d_x1, x1.data(), x1.size() * sizeof(uint32_t)) != hipSuccess) {
    nrphy_destroy(ctx);
    return NRPHY_ERR_DEVICE;
  }
  *out = ctx;
  return NRPHY_OK;
}

extern "C" int nrphy_destroy(nrphy_ctx_t* ctx)
{
  if (ctx == nullptr) {
    return NRPHY_OK;
  }
  (void)hipSetDevice(ctx->device);
  (void)hipFree(ctx->d_graphs);
  (void)hipFree(ctx->d_gold);
  (void)hipFree(ctx->d_tbcrc);
  for (void* b : ctx->scratch) {
    (void)hipFree(b);
  }
  for (DecoderGraph* g : ctx->d_dec_graph) {
    (void)hipFree(g);
  }
  for (uint32_t* a : ctx->d_dec_addr) {
    (void)hipFree(a);
  }
  for (auto& kv : ctx->d_dec_crc) {
    (void)hipFree(kv.second);
  }
  for (auto& kv : ctx->d_tb_crc_w) {
    (void)hipFree(kv.second);
  }
  (void)hipFree(ctx->d_x1);
  for (auto& kv : ctx->d_twiddle) {
    (void)hipFree(kv.second);
  }
  if (ctx->stream) {
    (void)hipStreamDestroy(ctx->stream);
  }
  delete ctx;
  return NRPHY_OK;
}

extern "C" int nrphy_synchronize(nrphy_ctx_t* ctx, void* stream)
{
  if (ctx == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  HIP_TRY(hipStreamSynchronize(stream ? (hipStream_t)stream : ctx->stream));
  return NRPHY_OK;
}

namespace {

// exp(+j 2 pi k / N) computed in double precision and rounded once.
const float2* get_twiddle(nrphy_ctx* ctx, uint32_t size)
{
  if (!dft_size_supported(size)) {
    return nullptr;
  }
  std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
  auto it = ctx->d_twiddle.find(size);
  if (it != ctx->d_twiddle.end()) {
    return it->second;
  }
  std::vector<float2> tw(size);
  for (uint32_t k = 0; k != size; ++k) {
    double ang = 2.0 * M_PI * (double)k / (double)size;
    tw[k]      = make_float2((float)std::cos(ang), (float)std::sin(ang));
  }
  float2* d = nullptr;
  if (upload(&d, tw.data(), tw.size() * sizeof(float2)) != hipSuccess) {
    return nullptr;
  }
  ctx->d_twiddle[size] = d;
  return d;
}

} // namespace

// ================================================================================================================
// PDSCH plan
// ================================================================================================================
namespace {

// Everything the RE mapping of a PDU depends on (data_re_mask + the DM-RS comb): PDUs of a batch that repeat an
// allocation share its tables instead of rebuilding them.
void append_allocation_signature(const nrphy_pdsch_pdu_t& pdu, std::vector<uint64_t>& sig)
{
  sig.insert(sig.end(), std::begin(pdu.prb_mask), std::end(pdu.prb_mask));
  sig.push_back(((uint64_t)pdu.start_symbol_index << 48) | ((uint64_t)pdu.nof_symbols << 40) |
                ((uint64_t)pdu.nof_cdm_groups_without_data << 36) | ((uint64_t)pdu.nof_layers << 32) |
                pdu.dmrs_symbol_mask);
  sig.push_back(((uint64_t)pdu.bwp_start_rb << 32) | ((uint64_t)pdu.bwp_size_rb << 8) | pdu.nof_reserved);
  for (unsigned r = 0; r != pdu.nof_reserved; ++r) {
    sig.insert(sig.end(), std::begin(pdu.reserved[r].prb_mask), std::end(pdu.reserved[r].prb_mask));
    sig.push_back(((uint64_t)pdu.reserved[r].re_mask << 32) | pdu.reserved[r].symbol_mask);
  }
}

struct ReMapping {
  uint32_t sym_re_start[NRPHY_NSYMB + 1];
  uint32_t sym_kind[NRPHY_NSYMB];
  uint32_t sym_arg[NRPHY_NSYMB];
};

} // namespace

// What a plan derives from the SHAPE of its PDUs alone -- allocation, symbols, DM-RS and reserved patterns, ports and
// layers -- kept across plans by a caller that builds one plan per PDU (the asynchronous queue): RE mapping tables and
// zero-fill run lists.  Everything else in a plan (slot index, RNTI, scrambling identities, transport-block size and
// the sizes derived from it, weights) is per PDU and rebuilt every time; it costs a few microseconds.
struct PlanShapeCache {
  struct Remap {
    ReMapping             m;     // sym_arg of SYM_TABLE symbols relative to `table`
    std::vector<uint16_t> table;
  };
  struct Zero {
    std::vector<ZeroSeg> segs; // long runs first
    uint32_t             nof_long = 0;
  };
  std::map<std::vector<uint64_t>, Remap> remap;
  std::map<std::vector<uint64_t>, Zero>  zero; // key: grid size + port + the allocation signatures of the PDUs on the port
  static constexpr size_t MAX_ENTRIES = 256;   // shapes in use at a time are few; a full cache starts over
};

PlanShapeCache* plan_shape_cache_create()
{
  return new (std::nothrow) PlanShapeCache;
}
void plan_shape_cache_destroy(PlanShapeCache* c)
{
  delete c;
}

namespace {

// Seam B (encode + rate match + interleave only): the codeword size and N_ref are given, there is no allocation.
struct EncodeOnly {
  uint32_t nof_re; // channel symbols per layer
  uint32_t nref;
};

int plan_create(nrphy_ctx_t* ctx, uint32_t n_pdu, const nrphy_pdsch_pdu_t* pdus, const uint64_t* tb_offset,
                const uint32_t* grid_index, uint32_t nof_grids, uint32_t grid_nof_ports, uint32_t grid_nof_subc,
                const EncodeOnly* enc, nrphy_pdsch_plan_t** out, PlanPlacement* place = nullptr);

} // namespace

int nrphy_pdsch_plan_create_placed(nrphy_ctx_t* ctx, uint32_t n_pdu, const nrphy_pdsch_pdu_t* pdus, const uint64_t* tb_offset,
                                   const uint32_t* grid_index, uint32_t nof_grids, uint32_t grid_nof_ports,
                                   uint32_t grid_nof_subc, PlanPlacement* place, nrphy_pdsch_plan_t** out)
{
  return plan_create(ctx, n_pdu, pdus, tb_offset, grid_index, nof_grids, grid_nof_ports, grid_nof_subc, nullptr, out, place);
}

extern "C" int nrphy_pdsch_plan_create(nrphy_ctx_t* ctx, uint32_t n_pdu, const nrphy_pdsch_pdu_t* pdus,
                                       const uint64_t* tb_offset, const uint32_t* grid_index, uint32_t nof_grids,
                                       uint32_t grid_nof_ports, uint32_t grid_nof_subc, nrphy_pdsch_plan_t** out)
{
  return plan_create(ctx, n_pdu, pdus, tb_offset, grid_index, nof_grids, grid_nof_ports, grid_nof_subc, nullptr, out);
}

namespace {

int plan_create(nrphy_ctx_t* ctx, uint32_t n_pdu, const nrphy_pdsch_pdu_t* pdus, const uint64_t* tb_offset,
                const uint32_t* grid_index, uint32_t nof_grids, uint32_t grid_nof_ports, uint32_t grid_nof_subc,
                const EncodeOnly* enc, nrphy_pdsch_plan_t** out, PlanPlacement* place)
{
  if (ctx == nullptr || out == nullptr || (n_pdu != 0 && (pdus == nullptr || tb_offset == nullptr)) ||
      grid_nof_ports == 0 || grid_nof_ports > NRPHY_MAX_PORTS || grid_nof_subc == 0 || grid_nof_subc % 12 != 0 ||
      grid_nof_subc > NRPHY_MAX_RB * 12) {
    return NRPHY_ERR_ARGUMENT;
  }
  *out = nullptr;
  if (place == nullptr) {
    HIP_TRY(hipSetDevice(ctx->device));
  }
  nrphy_pdsch_plan* plan = new (std::nothrow) nrphy_pdsch_plan;
  if (plan == nullptr) {
    return NRPHY_ERR_CAPACITY;
  }
  PlanShapeCache* shapes = place ? place->cache : nullptr;
  if (shapes != nullptr && shapes->remap.size() + shapes->zero.size() > PlanShapeCache::MAX_ENTRIES) {
    shapes->remap.clear();
    shapes->zero.clear();
  }
  plan->ctx            = ctx;
  plan->nof_grids      = nof_grids;
  plan->grid_nof_ports = grid_nof_ports;
  plan->grid_nof_subc  = grid_nof_subc;
  plan->encode_only    = enc != nullptr;

  std::vector<CbWork>   work;
  std::vector<DmrsWork> dmrs;
  std::vector<CrcWork>  crc_work;
  std::vector<uint64_t> remap_sig;
  std::map<std::vector<uint64_t>, ReMapping> remap_cache;
  std::vector<ScrWork>  scr_work;
  std::vector<std::vector<uint32_t>> pdus_of_grid(nof_grids);
  std::vector<float>    weights;
  std::vector<uint16_t> re_table;
  std::vector<uint8_t>  mask(grid_nof_subc);
  std::vector<uint16_t> list;
  uint64_t              cw_bits = 0;
  int                   status  = NRPHY_OK;

  for (uint32_t i = 0; i != n_pdu && status == NRPHY_OK; ++i) {
    const nrphy_pdsch_pdu_t& pdu = pdus[i];
    if (enc == nullptr ? nrphy_pdsch_validate(&pdu) != NRPHY_OK
                       : (pdu.qm < 2 || pdu.qm > 8 || (pdu.qm & 1U) || pdu.rv > 3 || pdu.nof_layers == 0 ||
                          pdu.nof_layers > NRPHY_MAX_LAYERS || pdu.tb_size_bytes == 0 ||
                          pdu.tb_size_bytes > NRPHY_MAX_TB_BYTES || (pdu.ldpc_base_graph != 1 && pdu.ldpc_base_graph != 2))) {
      status = NRPHY_ERR_INVALID_PDU;
      break;
    }
    const uint32_t g = grid_index ? grid_index[i] : 0;
    if (g >= nof_grids || pdu.nof_ports > grid_nof_ports || (tb_offset[i] & 3U) != 0 ||
        (enc == nullptr && 12U * (unsigned)(mask_highest(pdu.prb_mask) + 1) > grid_nof_subc)) {
      status = NRPHY_ERR_ARGUMENT;
      break;
    }
    PduDev pd;
    std::memset(&pd, 0, sizeof(pd));
    // RE mapping tables (shared by the PDUs of the batch that repeat this allocation).
    unsigned nof_re = 0;
    remap_sig.clear();
    append_allocation_signature(pdu, remap_sig);
    auto cached = remap_cache.find(remap_sig);
    if (enc != nullptr) {
      nof_re = enc[i].nof_re; // no RE mapping: every symbol empty, the count given
      for (unsigned l = 0; l <= NRPHY_NSYMB; ++l) {
        pd.sym_re_start[l] = (l == NRPHY_NSYMB) ? nof_re : 0;
      }
    } else if (cached != remap_cache.end()) {
      std::memcpy(pd.sym_re_start, cached->second.sym_re_start, sizeof(pd.sym_re_start));
      std::memcpy(pd.sym_kind, cached->second.sym_kind, sizeof(pd.sym_kind));
      std::memcpy(pd.sym_arg, cached->second.sym_arg, sizeof(pd.sym_arg));
      nof_re = pd.sym_re_start[NRPHY_NSYMB];
    } else {
      // The mapping with its table entries numbered from zero (`rel`): from the caller's shape cache, or built here.
      PlanShapeCache::Remap        built;
      const PlanShapeCache::Remap* rel = nullptr;
      if (shapes != nullptr) {
        auto known = shapes->remap.find(remap_sig);
        if (known != shapes->remap.end()) {
          rel = &known->second;
        }
      }
      if (rel == nullptr) {
        for (unsigned l = 0; l != NRPHY_NSYMB; ++l) {
          built.m.sym_re_start[l] = nof_re;
          built.m.sym_arg[l]      = 0;
          data_re_mask(pdu, l, mask);
          list.clear();
          for (unsigned k = 0; k != grid_nof_subc; ++k) {
            if (mask[k]) {
              list.push_back((uint16_t)k);
            }
          }
          if (list.empty()) {
            built.m.sym_kind[l] = SYM_NONE;
          } else if ((unsigned)(list.back() - list.front()) + 1 == list.size()) {
            built.m.sym_kind[l] = SYM_CONTIGUOUS;
            built.m.sym_arg[l]  = list.front();
          } else {
            built.m.sym_kind[l] = SYM_TABLE;
            built.m.sym_arg[l]  = (uint32_t)built.table.size();
            // Reuse an earlier symbol's list when identical (the common case).
            for (unsigned lp = 0; lp != l; ++lp) {
              if (built.m.sym_kind[lp] == SYM_TABLE && built.m.sym_re_start[lp + 1] - built.m.sym_re_start[lp] == list.size() &&
                  std::equal(list.begin(), list.end(), built.table.begin() + built.m.sym_arg[lp])) {
                built.m.sym_arg[l] = built.m.sym_arg[lp];
                break;
              }
            }
            if (built.m.sym_arg[l] == built.table.size()) {
              built.table.insert(built.table.end(), list.begin(), list.end());
            }
          }
          nof_re += (unsigned)list.size();
          built.m.sym_re_start[l + 1] = nof_re;
        }
        rel = &built;
        if (shapes != nullptr) {
          rel = &shapes->remap.insert({remap_sig, built}).first->second;
        }
      }
      // Into this plan: the table entries behind what the plan holds already.
      const uint32_t base = (uint32_t)re_table.size();
      re_table.insert(re_table.end(), rel->table.begin(), rel->table.end());
      std::memcpy(pd.sym_re_start, rel->m.sym_re_start, sizeof(pd.sym_re_start));
      std::memcpy(pd.sym_kind, rel->m.sym_kind, sizeof(pd.sym_kind));
      for (unsigned l = 0; l != NRPHY_NSYMB; ++l) {
        pd.sym_arg[l] = rel->m.sym_arg[l] + (rel->m.sym_kind[l] == SYM_TABLE ? base : 0U);
      }
      nof_re = pd.sym_re_start[NRPHY_NSYMB];
      ReMapping m;
      std::memcpy(m.sym_re_start, pd.sym_re_start, sizeof(m.sym_re_start));
      std::memcpy(m.sym_kind, pd.sym_kind, sizeof(m.sym_kind));
      std::memcpy(m.sym_arg, pd.sym_arg, sizeof(m.sym_arg));
      remap_cache.insert({remap_sig, m});
    }
    if (nof_re == 0) {
      status = NRPHY_ERR_INVALID_PDU;
      break;
    }
    nrphy_pdsch_derived_t d;
    derive(pdu, nof_re, d, enc ? &enc[i].nref : nullptr);
    if (d.lifting_size == 0 || d.nof_codeblocks > NRPHY_MAX_CODEBLOCKS || d.nof_codeblocks > nof_re ||
        d.rm_length_short == 0) {
      status = NRPHY_ERR_INVALID_PDU;
      break;
    }
    const unsigned kb = (pdu.ldpc_base_graph == 1) ? 22 : 10;
    pd.tb_offset      = tb_offset[i];
    pd.cw_bit_offset  = cw_bits;
    pd.tb_bytes       = pdu.tb_size_bytes;
    pd.grid_index     = g;
    pd.graph          = (pdu.ldpc_base_graph - 1) * NOF_LIFTING_SIZES + (uint32_t)lifting_position(d.lifting_size);
    pd.zc             = d.lifting_size;
    pd.kb             = kb;
    pd.K              = d.segment_length;
    pd.info_bits      = d.cb_info_bits;
    pd.filler         = d.nof_filler_bits;
    pd.tb_crc_bits    = d.nof_tb_crc_bits;
    pd.cb_crc_bits    = d.nof_cb_crc_bits;
    pd.zero_pad       = d.zero_pad;
    pd.C              = d.nof_codeblocks;
    pd.n_short        = d.nof_short_segments;
    pd.e_short        = d.rm_length_short;
    pd.e_long         = d.rm_length_long;
    pd.n_cb           = d.n_cb;
    pd.k0             = d.k0;
    pd.qm             = pdu.qm;
    pd.nof_layers     = pdu.nof_layers;
    pd.nof_ports      = pdu.nof_ports;
    pd.c_init         = (pdu.rnti << 15) + pdu.n_id; // q = 0 (pdsch_modulator_impl.cpp:35)
    pd.nof_re         = nof_re;
    // Parity rows rate matching can reach (the reference always computes all of them, pdsch_encoder_impl.cpp:52).
    {
      const unsigned nsys = (kb - 2) * d.lifting_size;
      unsigned       fs = std::min(nsys - d.nof_filler_bits, d.n_cb), fe = std::min(nsys, d.n_cb);
      const unsigned flen = fe - fs, n_valid = d.n_cb - flen;
      const unsigned rank0 = d.k0 < fs ? d.k0 : (d.k0 < fe ? fs : d.k0 - flen);
      unsigned       last; // highest circular-buffer position read
      if (rank0 + d.rm_length_long > n_valid) {
        last = d.n_cb - 1;
      } else {
        unsigned u = rank0 + d.rm_length_long - 1;
        last       = u < fs ? u : u + flen;
      }
      const unsigned nodes = divide_ceil(last + 1 + 2 * d.lifting_size, d.lifting_size);
      pd.nof_rows          = std::max(4U, nodes > kb ? nodes - kb : 0U);
    }
    // Precoding weights: data weights carry the modulation and power scaling (pdsch_modulator_impl.cpp:98-102).
    {
      const float avg     = (pdu.qm == 2) ? 2.0F : (pdu.qm == 4) ? 10.0F : (pdu.qm == 6) ? 42.0F : 170.0F;
      float       scaling = std::sqrt(1 / avg);
      const float cfg     = std::pow(10.0F, -pdu.ratio_pdsch_data_to_sss_dB / 20.0F);
      if (std::isnormal(cfg)) {
        scaling *= cfg;
      }
      const unsigned nw      = 2 * pdu.nof_prg * pdu.nof_ports * pdu.nof_layers;
      pd.weights_offset      = (uint32_t)weights.size();
      for (unsigned k = 0; k != nw; ++k) {
        weights.push_back(pdu.precoding ? pdu.precoding[k] * scaling : 0.0F); // no weights in an encode-only plan
      }
      pd.dmrs_weights_offset = (uint32_t)weights.size();
      for (unsigned k = 0; k != nw; ++k) {
        weights.push_back(pdu.precoding ? pdu.precoding[k] : 0.0F);
      }
      pd.nof_prg       = pdu.nof_prg;
      pd.prg_size_subc = pdu.prg_size_rb * 12;
    }
    // DM-RS (dmrs_pdsch_processor_impl.cpp:84-106).
    pd.dmrs_symbol_mask = pdu.dmrs_symbol_mask;
    pd.dmrs_zero_other_group = (pdu.nof_cdm_groups_without_data == 2 && (pdu.nof_layers + 1) / 2 == 1) ? 1U : 0U;
    pd.dmrs_ref_rb      = (pdu.ref_point == 1) ? pdu.bwp_start_rb : 0;
    {
      const float amp   = std::pow(10.0F, -pdu.ratio_pdsch_dmrs_to_sss_dB / 20.0F);
      pd.dmrs_amplitude = (float)(M_SQRT1_2 * (double)amp);
    }
    for (unsigned l = 0; l != NRPHY_NSYMB; ++l) {
      // 14 symbols per slot also with extended cyclic prefix: the reference takes get_nsymb_per_slot(NORMAL) here
      // (dmrs_pdsch_processor_impl.cpp:95), and a drop-in has to produce the same pilots.
      const uint64_t a  = (uint64_t)(14 * pdu.slot_index + l + 1) * (2 * pdu.scrambling_id + 1);
      pd.dmrs_c_init[l] = (uint32_t)(((a << 17) + (2 * pdu.scrambling_id + (pdu.n_scid ? 1 : 0))) & 0x7FFFFFFFULL);
      if ((pdu.dmrs_symbol_mask >> l) & 1U) {
        const uint32_t first = (uint32_t)mask_lowest(pdu.prb_mask), end = (uint32_t)mask_highest(pdu.prb_mask) + 1;
        for (uint32_t b = first; b < end; b += DMRS_PRB_CHUNK) {
          dmrs.push_back({i, l, b, std::min<uint32_t>(end, b + DMRS_PRB_CHUNK)});
        }
      }
    }
    for (unsigned w = 0; w != NRPHY_PRB_WORDS; ++w) {
      pd.prb_mask[2 * w]     = (uint32_t)pdu.prb_mask[w];
      pd.prb_mask[2 * w + 1] = (uint32_t)(pdu.prb_mask[w] >> 32);
    }
    pd.first_prb = (uint32_t)mask_lowest(pdu.prb_mask);
    pd.end_prb   = (uint32_t)mask_highest(pdu.prb_mask) + 1;
    if (pdu.nof_cdm_groups_without_data < (pdu.nof_layers + 1) / 2) {
      plan->dmrs_separate = true; // data is mapped on RE that also carry DM-RS: the reference lets DM-RS win
    }
    pdus_of_grid[g].push_back(i);
    // TB-CRC work: the transport block in 16 KiB regions, a workgroup per run of regions.  A small batch gets a workgroup
    // per region (latency); a big one has workgroups enough and lets each walk several regions, the next one's words in
    // flight while it reduces the current one (a workgroup per region spent two thirds of its time waiting for its loads:
    // profiles/r03_prologue_trace.txt).
    {
      const CrcField& f = (d.nof_tb_crc_bits == 16) ? CRC16_FIELD : CRC24A_FIELD;
      const uint32_t  n = pdu.tb_size_bytes;
      const uint32_t  regions = divide_ceil(n, TB_CRC_REGION_BYTES);
      const uint32_t  want  = std::max<uint32_t>(1, std::min<uint32_t>(regions, TB_CRC_TARGET_WORK / std::max<uint32_t>(1, n_pdu)));
      uint32_t        per   = std::min<uint32_t>(TB_CRC_MAX_REGIONS_PER_WORK, divide_ceil(regions, want));
      if (ctx->tune.crc_regions > 0) { // (A/B and test knob: regions per workgroup)
        per = std::max(1, std::min((int)TB_CRC_MAX_REGIONS_PER_WORK, ctx->tune.crc_regions));
      }
      pd.crc_first      = (uint32_t)crc_work.size();
      pd.crc_count      = divide_ceil(regions, per);
      if (pd.crc_count > 64) { // one lane of the attaching wave per share
        status = NRPHY_ERR_INVALID_PDU;
        break;
      }
      for (uint32_t region = 0; region < regions; region += per) {
        const uint32_t count      = std::min(per, regions - region);
        const int64_t  region_end = (int64_t)(region + count) * TB_CRC_REGION_BYTES;
        crc_work.push_back({i, region, f.xpow((int64_t)f.order + 8 * ((int64_t)n - region_end)), count});
      }
    }
    // Work items: every codeblock owns a whole number of RE (rm_length is a multiple of nof_layers * Qm).
    const unsigned lq = pdu.nof_layers * pdu.qm;
    for (unsigned cb = 0; cb != d.nof_codeblocks; ++cb) {
      const unsigned nre = ((cb < d.nof_short_segments) ? d.rm_length_short : d.rm_length_long) / lq;
      for (unsigned begin = 0; begin < nre; begin += RE_CHUNK) {
        const unsigned count = std::min<unsigned>(RE_CHUNK, nre - begin);
        work.push_back({i, cb, begin, count});
        {
          // The wave expands its scrambling words from a 31-word seed into the LDS that held the codeblock: room for them
          // (the chunk's words from the one its first bit lies in, plus the word a misaligned read runs into).
          const uint64_t bit0 = (uint64_t)(cb < d.nof_short_segments ? cb * d.rm_length_short
                                                                     : d.nof_short_segments * d.rm_length_short +
                                                                           (cb - d.nof_short_segments) * d.rm_length_long) +
                                (uint64_t)begin * lq;
          const uint32_t need = std::max<uint32_t>(31U, (uint32_t)(((bit0 & 31U) + (uint64_t)count * lq + 31U) / 32U) + 1U);
          plan->lds_lin_words = std::max<uint32_t>(plan->lds_lin_words, (need + 3U) & ~3U);
        }
        // LDS the wave needs for the symbol bytes (32 per block + 8 words).
        plan->lds_symb_words = std::max<uint32_t>(plan->lds_symb_words,
                                                  (((count * pdu.nof_layers + 31) / 32) * 8 + 8 + 3) & ~3U);
      }
    }
    plan->lds_lin_words = std::max<uint32_t>(plan->lds_lin_words,
                                             ((((kb + pd.nof_rows) * d.lifting_size + 31) / 32) + 2 + 3) & ~3U);
    plan->lds_graph_words = std::max<uint32_t>(
        plan->lds_graph_words, (48U + ctx->graphs[pd.graph].row_ptr[std::min<uint32_t>(pd.nof_rows, MAX_BG_ROWS)] + 3U) & ~3U);
    // Scrambling sequence of the PDU: one word per 32 codeword bits plus the word a misaligned read runs into, plus the
    // length of a seed (the last work item's 31 words may reach beyond the codeword; the sequence simply goes on).  Only
    // the work items' seeds are stored (behind the DM-RS sequences, below).
    pd.scr_words  = (d.codeword_bits + 31U) / 32U + 1U + 31U;
    // One DM-RS sequence per DM-RS symbol.
    pd.dmrs_seq_offset = (uint32_t)plan->scr_words;
    pd.dmrs_seq_words  = (12U * (pd.end_prb - pd.dmrs_ref_rb) + 31U) / 32U + 1U;
    plan->scr_words += ((uint64_t)pd.dmrs_seq_words * (unsigned)__builtin_popcount(pdu.dmrs_symbol_mask) + 3U) & ~3ULL;
    {
      // A big batch has enough PDUs to fill the device with one workgroup each (seeding a generator is the costly
      // part: measured 0.111 / 0.098 / 0.096 ms per 1024 config-3 PDUs with 4 / 2 / 1 parts); a small one is split
      // for latency.
      // (A/B and test knob: parts of a sequence in a big batch)
      const uint32_t     parts_big = ctx->tune.scr_parts_big > 0 ? (uint32_t)std::min((int)SCR_PARTS, ctx->tune.scr_parts_big) : 1U;
      const uint32_t parts_max = n_pdu >= 128 ? parts_big : SCR_PARTS;
      const uint32_t parts     = std::min<uint32_t>(parts_max, std::max<uint32_t>(1, pd.scr_words >> 11));
      const uint32_t chunk = divide_ceil(pd.scr_words, parts);
      for (uint32_t first = 0, k = 0; first < pd.scr_words; first += chunk, ++k) {
        scr_work.push_back({i, first, std::min(chunk, pd.scr_words - first), k == 0 ? 1U : 0U});
      }
    }
    plan->n_cb += d.nof_codeblocks;
    plan->cw_offset.push_back(cw_bits);
    cw_bits += (d.codeword_bits + 31U) & ~31ULL;
    plan->pdus.push_back(pd);
  }
  if (status != NRPHY_OK) {
    delete plan;
    return status;
  }
  // Zero-fill work: per (grid, port) the runs of subcarriers no PDU maps (data or DM-RS).
  std::vector<ZeroWork> zero_work;
  std::vector<ZeroSeg>  zero_segs;
  {
    std::map<std::vector<uint64_t>, std::array<uint32_t, 3>> seen; // segment list -> (begin, count, long runs)
    std::vector<uint8_t>  cov((size_t)NRPHY_NSYMB * grid_nof_subc);
    std::vector<uint64_t> key, sig;
    std::map<std::vector<uint64_t>, std::array<uint32_t, 3>> by_signature; // allocation -> (begin, count, long runs)
    for (uint32_t g = 0; g != (enc ? 0U : nof_grids); ++g) { // an encode-only plan writes no grid
      for (uint32_t port = 0; port != grid_nof_ports; ++port) {
        // Everything the coverage of this (grid, port) depends on: grids that repeat an allocation (the normal case
        // in a batch of slots) reuse its segment list without rebuilding the RE masks.
        sig.clear();
        for (uint32_t i : pdus_of_grid[g]) {
          const nrphy_pdsch_pdu_t& pdu = pdus[i];
          if (port >= pdu.nof_ports) {
            continue;
          }
          append_allocation_signature(pdu, sig);
        }
        auto known = by_signature.find(sig);
        if (known != by_signature.end()) {
          if (known->second[1] != 0) {
            zero_work.push_back({g, port, known->second[0], known->second[1], known->second[2]});
          }
          continue;
        }
        if (shapes != nullptr) {
          // The run list of this coverage from an earlier plan of the caller (the grid size is part of the key).
          sig.push_back(((uint64_t)grid_nof_subc << 32) | port);
          auto kept = shapes->zero.find(sig);
          sig.pop_back();
          if (kept != shapes->zero.end()) {
            const std::array<uint32_t, 3> where = {(uint32_t)zero_segs.size(), (uint32_t)kept->second.segs.size(),
                                                   kept->second.nof_long};
            zero_segs.insert(zero_segs.end(), kept->second.segs.begin(), kept->second.segs.end());
            by_signature.insert({sig, where});
            if (where[1] != 0) {
              zero_work.push_back({g, port, where[0], where[1], where[2]});
            }
            continue;
          }
        }
        std::fill(cov.begin(), cov.end(), 0);
        for (uint32_t i : pdus_of_grid[g]) {
          const nrphy_pdsch_pdu_t& pdu = pdus[i];
          if (port >= pdu.nof_ports) {
            continue;
          }
          for (unsigned l = 0; l != NRPHY_NSYMB; ++l) {
            uint8_t* row = &cov[(size_t)l * grid_nof_subc];
            data_re_mask(pdu, l, mask);
            for (unsigned k = 0; k != grid_nof_subc; ++k) {
              row[k] |= mask[k];
            }
            if ((pdu.dmrs_symbol_mask >> l) & 1U) {
              // RE of a CDM group that is reserved (no data) but carries no pilots of this PDU are zeroed by the
              // DM-RS waves themselves (dmrs_zero_other_group): as zero-fill work they would be 1-RE segments.
              const unsigned groups = (pdu.nof_cdm_groups_without_data == 2) ? 2 : (pdu.nof_layers + 1) / 2;
              for (unsigned prb = 0; 12 * prb < grid_nof_subc; ++prb) {
                if (mask_test(pdu.prb_mask, prb)) {
                  for (unsigned k = 0; k != 12; ++k) {
                    row[12 * prb + k] |= (k % 2) < groups;
                  }
                }
              }
            }
          }
        }
        key.clear();
        for (unsigned l = 0; l != NRPHY_NSYMB; ++l) {
          const uint8_t* row = &cov[(size_t)l * grid_nof_subc];
          unsigned       k   = 0;
          while (k < grid_nof_subc) {
            if (row[k]) {
              ++k;
              continue;
            }
            unsigned k0 = k;
            while (k < grid_nof_subc && !row[k]) {
              ++k;
            }
            key.push_back(((uint64_t)l << 32) | ((uint64_t)k0 << 16) | (k - k0));
          }
        }
        if (key.empty()) {
          by_signature.insert({sig, {0, 0, 0}});
          if (shapes != nullptr) {
            sig.push_back(((uint64_t)grid_nof_subc << 32) | port);
            shapes->zero.insert({sig, PlanShapeCache::Zero()});
            sig.pop_back();
          }
          continue;
        }
        auto it = seen.find(key);
        if (it == seen.end()) {
          // Long runs first (the wave clears each one together), then the short ones (one lane per run).
          const uint32_t begin = (uint32_t)zero_segs.size();
          uint32_t       nof_long = 0;
          for (int pass = 0; pass != 2; ++pass) {
            for (uint64_t v : key) {
              const bool is_long = (v & 0xFFFF) >= ZERO_LONG_RUN;
              if (is_long == (pass == 0)) {
                zero_segs.push_back({(uint16_t)(v >> 32), (uint16_t)((v >> 16) & 0xFFFF), (uint16_t)(v & 0xFFFF), 0});
                nof_long += is_long ? 1U : 0U;
              }
            }
          }
          it = seen.insert({key, {begin, (uint32_t)key.size(), nof_long}}).first;
        }
        by_signature.insert({sig, it->second});
        zero_work.push_back({g, port, it->second[0], it->second[1], it->second[2]});
        if (shapes != nullptr) {
          PlanShapeCache::Zero z;
          z.segs.assign(zero_segs.begin() + it->second[0], zero_segs.begin() + it->second[0] + it->second[1]);
          z.nof_long = it->second[2];
          sig.push_back(((uint64_t)grid_nof_subc << 32) | port);
          shapes->zero.insert({sig, std::move(z)});
          sig.pop_back();
        }
      }
    }
  }
  plan->n_zero_work = (uint32_t)zero_work.size();
  plan->cw_bits = cw_bits;
  plan->n_work  = (uint32_t)work.size();
  {
    // One bucket per (modulation order, layers), PDU and codeblock order kept inside (launch_codeblocks).
    const auto bucket_of = [&](const CbWork& w) { return cb_bucket(plan->pdus[w.pdu].qm, plan->pdus[w.pdu].nof_layers); };
    std::stable_sort(work.begin(), work.end(), [&](const CbWork& a, const CbWork& b) { return bucket_of(a) < bucket_of(b); });
    for (const CbWork& w : work) {
      ++plan->bucket_begin[bucket_of(w) + 1];
    }
    for (uint32_t b = 0; b != CB_BUCKETS; ++b) {
      plan->bucket_begin[b + 1] += plan->bucket_begin[b];
    }
    // A PDU's work items stay together and in order (one bucket per PDU, stable sort): where they start.
    for (size_t k = work.size(); k-- != 0;) {
      plan->pdus[work[k].pdu].item_first = (uint32_t)k;
    }
    // The seeds of the work items' scrambling sequences: 32 words each, behind the DM-RS sequences.
    plan->scr_words  = (plan->scr_words + 3U) & ~3ULL;
    plan->seed_offset = plan->scr_words;
    plan->scr_words += 32ULL * work.size();
    // The codeblock waves load 2 * NRPHY_MAX_PORTS * layers weights whatever the port count (pdsch_kernels.hip, phase_b).
    weights.insert(weights.end(), 2 * NRPHY_MAX_PORTS * NRPHY_MAX_PORTS, 0.0F);
  }
  plan->n_dmrs  = (uint32_t)dmrs.size();
  plan->n_crc_work = (uint32_t)crc_work.size();
  plan->n_scr_work = (uint32_t)scr_work.size();
  {
    DeviceArena          arena;
    arena.add(&plan->d_crc_work, crc_work.data(), crc_work.size() * sizeof(CrcWork));
    arena.add(&plan->d_scr_work, scr_work.data(), scr_work.size() * sizeof(ScrWork));
    arena.add(&plan->d_pdus, plan->pdus.data(), plan->pdus.size() * sizeof(PduDev));
    arena.add(&plan->d_work, work.data(), work.size() * sizeof(CbWork));
    arena.add(&plan->d_dmrs, dmrs.data(), dmrs.size() * sizeof(DmrsWork));
    arena.add(&plan->d_weights, weights.data(), weights.size() * sizeof(float));
    arena.add(&plan->d_re_table, re_table.data(), re_table.size() * sizeof(uint16_t));
    arena.add(&plan->d_zero_work, zero_work.data(), zero_work.size() * sizeof(ZeroWork));
    arena.add(&plan->d_zero_segs, zero_segs.data(), zero_segs.size() * sizeof(ZeroSeg));
    void* scratch = nullptr;
    // Behind the tables: what every run rewrites before it reads it -- the sequences and the TB-CRC shares.
    const uint64_t scr_alloc = (std::max<uint64_t>(4, plan->scr_words) + 3U) & ~3ULL;
    const uint64_t scratch_words = scr_alloc + std::max<size_t>(4, crc_work.size());
    if (place != nullptr) {
      // Caller-owned memory: no allocation, no copy, no synchronisation here (the asynchronous queue's submit path).
      if (arena.bytes() > place->table_capacity || scratch_words > place->scratch_capacity_words) {
        delete plan;
        return NRPHY_ERR_CAPACITY;
      }
      arena.place(place->h_tables, place->d_tables);
      place->table_bytes  = arena.bytes();
      plan->arena_external = true;
      scratch              = place->d_scratch;
    } else if (arena.commit(&plan->d_arena, sizeof(uint32_t) * scratch_words, &scratch) != hipSuccess) {
      nrphy_pdsch_plan_destroy(plan);
      return NRPHY_ERR_DEVICE;
    }
    plan->d_scr    = (uint32_t*)scratch;
    plan->d_tb_crc = plan->d_scr + scr_alloc;
  }
  {
    // A batch that mixes modulations and is big enough for one launch per bucket (launch_codeblocks) runs those launches side by
    // side: its side streams exist from here on, so that a run makes no HIP object and can be captured in a graph.
    uint32_t nof_buckets = 0;
    for (uint32_t b = 0; b != CB_BUCKETS; ++b) {
      nof_buckets += plan->bucket_begin[b + 1] != plan->bucket_begin[b] ? 1U : 0U;
    }
    if (place == nullptr && nof_buckets > 1 && plan->n_work >= CB_MIXED_MAX_WORK &&
        !plan_side_streams(plan, std::min<uint32_t>(nof_buckets - 1, nrphy_pdsch_plan::MAX_AUX))) {
      nrphy_pdsch_plan_destroy(plan);
      return NRPHY_ERR_DEVICE;
    }
  }
  // The dynamic LDS of the codeblock launch also serves the DM-RS waves it may carry.
  plan->lds_lin_words = std::max<uint32_t>(plan->lds_lin_words, 64);
  // The scratch region the stages of a codeblock wave share (pdsch_kernels.hip, CbShared): CRC tables, then doubled
  // systematic blocks + graph rows, then modulation table + symbol bytes.
  plan->lds_u_words = std::max<uint32_t>({256U * NRPHY_CRC_SLICES, NRPHY_CB_U_GRAPH_OFFSET + plan->lds_graph_words, 512U + plan->lds_symb_words});
  *out = plan;
  return NRPHY_OK;
}

} // namespace

// The plan's side streams and fork / join events for bucket launches that run side by side (nrphy_pdsch_run).
static bool plan_side_streams(nrphy_pdsch_plan* plan, uint32_t want)
{
  while (plan->n_aux < want) {
    const uint32_t k = plan->n_aux;
    if (plan->fork_event == nullptr && hipEventCreateWithFlags(&plan->fork_event, hipEventDisableTiming) != hipSuccess) {
      return false;
    }
    if (hipStreamCreateWithFlags(&plan->aux_stream[k], hipStreamNonBlocking) != hipSuccess) {
      return false;
    }
    if (hipEventCreateWithFlags(&plan->join_event[k], hipEventDisableTiming) != hipSuccess) {
      (void)hipStreamDestroy(plan->aux_stream[k]);
      return false;
    }
    ++plan->n_aux;
  }
  return true;
}

extern "C" int nrphy_pdsch_plan_destroy(nrphy_pdsch_plan_t* plan)
{
  if (plan == nullptr) {
    return NRPHY_OK;
  }
  if (!plan->arena_external) {
    (void)hipSetDevice(plan->ctx->device);
    (void)hipFree(plan->d_arena); // null for a plan whose creation failed half-way
  }
  for (uint32_t k = 0; k != plan->n_aux; ++k) {
    (void)hipStreamDestroy(plan->aux_stream[k]);
    (void)hipEventDestroy(plan->join_event[k]);
  }
  if (plan->fork_event != nullptr) {
    (void)hipEventDestroy(plan->fork_event);
  }
  for (hipEvent_t e : plan->events) {
    (void)hipEventDestroy(e);
  }
  delete plan;
  return NRPHY_OK;
}

extern "C" uint32_t nrphy_pdsch_plan_nof_codeblocks(const nrphy_pdsch_plan_t* plan)
{
  return plan ? plan->n_cb : 0;
}

extern "C" uint64_t nrphy_pdsch_plan_codeword_bits(const nrphy_pdsch_plan_t* plan)
{
  return plan ? plan->cw_bits : 0;
}

extern "C" uint64_t nrphy_pdsch_plan_codeword_offset(const nrphy_pdsch_plan_t* plan, uint32_t pdu)
{
  return (plan && pdu < plan->cw_offset.size()) ? plan->cw_offset[pdu] : 0;
}

extern "C" int nrphy_pdsch_run(nrphy_pdsch_plan_t* plan, const uint8_t* d_tb, void* d_grid, uint8_t* d_cw_rm,
                               uint8_t* d_cw_scrambled, int zero_grids, void* stream)
{
  if (plan == nullptr || d_tb == nullptr || (plan->encode_only && d_grid != nullptr)) {
    return NRPHY_ERR_ARGUMENT;
  }
  const TraceRange trace_run("process_pdsch");
  nrphy_ctx*  ctx = plan->ctx;
  hipStream_t s   = stream ? (hipStream_t)stream : ctx->stream;
  PdschLaunch p;
  p.pdus           = plan->d_pdus;
  p.work           = plan->d_work;
  p.dmrs_work      = plan->d_dmrs;
  p.crc_work       = plan->d_crc_work;
  p.scr_work       = plan->d_scr_work;
  p.n_scr_work     = plan->n_scr_work;
  p.tbcrc          = ctx->d_tbcrc;
  p.n_crc_work     = plan->n_crc_work;
  p.weights        = plan->d_weights;
  p.re_table       = plan->d_re_table;
  p.graphs         = ctx->d_graphs;
  p.gold           = ctx->d_gold;
  p.x1_words       = ctx->d_x1;
  p.tb_crc_part        = plan->d_tb_crc;
  const bool merge_dmrs = d_grid != nullptr && !plan->dmrs_separate;
  p.zero_work          = plan->d_zero_work;
  p.zero_segs          = plan->d_zero_segs;
  p.scr                = plan->d_scr;
  p.scr_seed           = plan->d_scr + plan->seed_offset;
  p.n_zero_work        = (d_grid != nullptr && zero_grids) ? plan->n_zero_work : 0;
  p.zero_fill          = (d_grid != nullptr && zero_grids) ? 1U : 0U;
  p.n_dmrs_in_launch   = merge_dmrs ? plan->n_dmrs : 0;
  p.n_pdu          = (uint32_t)plan->pdus.size();
  p.n_work         = plan->n_work;
  p.work_base      = 0;
  p.n_dmrs_work    = plan->n_dmrs;
  p.grid_nof_ports = plan->grid_nof_ports;
  p.grid_nof_subc  = plan->grid_nof_subc;
  p.lds_lin_words  = plan->lds_lin_words;
  p.lds_u_words    = plan->lds_u_words;
  // Store policy of the DM-RS / zero-fill waves at the tail of the codeblock launch: non-temporal (NRPHY_EXTRAS_NT=0: default
  // policy).  It moves time from the OFDM launch that follows to the codeblock launch.  Before the OFDM launch took its grids
  // last to first the balance depended on the box (+1.2 % whole step where the OFDM launch is slow, 0 ... -1 % where it is
  // fast); with that order, A/B on one box, two rounds (profiles/r03_codeblock_experiments.txt): codeblock 0.301 -> 0.314 ms,
  // OFDM 0.500 -> 0.466 ms, whole step +2.0 %.
  // (Placing those waves first or between the codeblock waves instead: the codeblock launch 0.44 / 0.46 ms -- their stores push
  // the transport blocks and sequences out of the cache.)
  p.extras_nt      = ctx->tune.extras_nt;
  p.prologue_order = ctx->tune.prologue_order;
#ifdef NRPHY_PROBES
  // Profiling variant: stop the codeblock waves after a stage to time the stages apart (outputs are then incomplete).
  p.profile_stage = ctx->tune.profile_stage;
#else
  p.profile_stage = 0;
#endif
  const size_t cw_bytes = (size_t)(plan->cw_bits / 8);
  if (d_cw_rm) {
    HIP_TRY(hipMemsetAsync(d_cw_rm, 0, cw_bytes, s));
  }
  if (d_cw_scrambled) {
    HIP_TRY(hipMemsetAsync(d_cw_scrambled, 0, cw_bytes, s));
  }
  hipEvent_t* ev = nullptr;
  if (plan->timed_runs < plan->max_timed_runs && plan->timing_counter++ % plan->timing_stride == 0) {
    ev = &plan->events[4 * plan->timed_runs++];
    HIP_TRY(hipEventRecord(ev[0], s));
  }
  HIP_TRY(launch_prologue(p, d_tb, s));
  if (ev) {
    HIP_TRY(hipEventRecord(ev[1], s));
  }
  {
    const TraceRange trace_cb("CB batch");
    // NRPHY_CB_DISPATCH: 1 = the one-launch mixed kernel, 2 = one launch per (Qm, layers) bucket, unset = by plan shape.
    const int   dispatch     = ctx->tune.cb_dispatch;
    uint32_t    nof_buckets  = 0;
    hipStream_t streams[1 + nrphy_pdsch_plan::MAX_AUX] = {s};
    uint32_t    n_streams = 1;
    if (codeblocks_take_bucket_launches(p, plan->bucket_begin, dispatch, &nof_buckets) && nof_buckets > 1) {
      const uint32_t want = std::min<uint32_t>(nof_buckets - 1, nrphy_pdsch_plan::MAX_AUX);
      // (made at plan creation for a plan that takes bucket launches by its shape; here only when NRPHY_CB_DISPATCH forces
      // them on a small one -- such a first run creates streams and is not for graph capture)
      if (!plan_side_streams(plan, want)) {
        return NRPHY_ERR_DEVICE;
      }
      HIP_TRY(hipEventRecord(plan->fork_event, s));
      for (uint32_t k = 0; k != want; ++k) {
        HIP_TRY(hipStreamWaitEvent(plan->aux_stream[k], plan->fork_event, 0));
        streams[n_streams++] = plan->aux_stream[k];
      }
    }
    HIP_TRY(launch_codeblocks(p, plan->bucket_begin, dispatch, d_tb, (uint32_t*)d_grid, (uint32_t*)d_cw_rm,
                              (uint32_t*)d_cw_scrambled, streams, n_streams));
    for (uint32_t k = 1; k < n_streams; ++k) {
      HIP_TRY(hipEventRecord(plan->join_event[k - 1], streams[k]));
      HIP_TRY(hipStreamWaitEvent(s, plan->join_event[k - 1], 0));
    }
  }
  if (ev) {
    HIP_TRY(hipEventRecord(ev[2], s));
  }
  if (d_grid && !merge_dmrs) {
    // After the data: when data RE share a CDM group with DM-RS the reference lets DM-RS overwrite them.
    const TraceRange trace_dmrs("process_dmrs");
    HIP_TRY(launch_dmrs(p, (uint32_t*)d_grid, s));
    if (ev) {
      HIP_TRY(hipEventRecord(ev[3], s));
    }
  }
  if (ev) { // (an event between two launches costs the stream a few microseconds: none where no launch follows)
    plan->timed_dmrs[plan->timed_runs - 1] = (d_grid && !merge_dmrs) ? 1 : 0;
  }
  return NRPHY_OK;
}

extern "C" int nrphy_pdsch_plan_enable_timing(nrphy_pdsch_plan_t* plan, uint32_t max_runs)
{
  if (plan == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  for (hipEvent_t e : plan->events) {
    (void)hipEventDestroy(e);
  }
  plan->events.assign(4 * (size_t)max_runs, nullptr);
  for (hipEvent_t& e : plan->events) {
    HIP_TRY(hipEventCreate(&e));
  }
  plan->timed_dmrs.assign(max_runs, 0);
  plan->max_timed_runs = max_runs;
  plan->timed_runs     = 0;
  plan->timing_counter = 0;
  return NRPHY_OK;
}

extern "C" int nrphy_pdsch_plan_timing_stride(nrphy_pdsch_plan_t* plan, uint32_t stride)
{
  if (plan == nullptr || stride == 0) {
    return NRPHY_ERR_ARGUMENT;
  }
  plan->timing_stride  = stride;
  plan->timing_counter = 0;
  return NRPHY_OK;
}

extern "C" int nrphy_pdsch_plan_kernel_times(nrphy_pdsch_plan_t* plan, float avg_ms[4], uint32_t* nof_runs)
{
  if (plan == nullptr || avg_ms == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  double sum[4] = {0, 0, 0, 0};
  for (uint32_t r = 0; r != plan->timed_runs; ++r) {
    hipEvent_t*    ev   = &plan->events[4 * r];
    const unsigned last = plan->timed_dmrs[r] ? 3 : 2;
    HIP_TRY(hipEventSynchronize(ev[last]));
    float ms = 0;
    for (unsigned k = 0; k != last; ++k) {
      HIP_TRY(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
      sum[k] += ms;
    }
    HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[last]));
    sum[3] += ms;
  }
  for (int k = 0; k != 4; ++k) {
    avg_ms[k] = plan->timed_runs ? (float)(sum[k] / plan->timed_runs) : 0.f;
  }
  if (nof_runs) {
    *nof_runs = plan->timed_runs;
  }
  plan->timed_runs = 0;
  return NRPHY_OK;
}

extern "C" int nrphy_pdsch_process_host(nrphy_ctx_t* ctx, const nrphy_pdsch_pdu_t* pdu, const uint8_t* tb, void* grid,
                                        uint32_t grid_nof_ports, uint32_t grid_nof_subc, uint8_t* cw_rm,
                                        uint8_t* cw_scrambled)
{
  if (ctx == nullptr || pdu == nullptr || tb == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  nrphy_pdsch_plan_t* plan   = nullptr;
  uint64_t            tb_off = 0;
  uint32_t            gi     = 0;
  int rc = nrphy_pdsch_plan_create(ctx, 1, pdu, &tb_off, &gi, 1, grid_nof_ports, grid_nof_subc, &plan);
  if (rc != NRPHY_OK) {
    return rc;
  }
  const size_t tb_alloc   = ((size_t)pdu->tb_size_bytes + 7) & ~(size_t)3;
  const size_t grid_bytes = (size_t)grid_nof_ports * NRPHY_NSYMB * grid_nof_subc * 4;
  const size_t cw_bytes   = (size_t)(plan->cw_bits / 8);
  std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
  uint8_t* d_tb   = (uint8_t*)ctx_scratch(ctx, SCRATCH_TB, tb_alloc);
  uint8_t* d_grid = grid ? (uint8_t*)ctx_scratch(ctx, SCRATCH_GRID, grid_bytes) : nullptr;
  uint8_t* d_rm   = cw_rm ? (uint8_t*)ctx_scratch(ctx, SCRATCH_CW_RM, cw_bytes) : nullptr;
  uint8_t* d_scr  = cw_scrambled ? (uint8_t*)ctx_scratch(ctx, SCRATCH_CW_SCR, cw_bytes) : nullptr;
  rc = NRPHY_ERR_DEVICE;
  do {
    if (d_tb == nullptr || (grid && d_grid == nullptr) || (cw_rm && d_rm == nullptr) ||
        (cw_scrambled && d_scr == nullptr)) {
      break;
    }
    // The transport block is readable to the next multiple of 4: clear the tail word, then the bytes.
    if (hipMemsetAsync(d_tb + (tb_alloc - 8), 0, 8, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(d_tb, tb, pdu->tb_size_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
      break;
    }
    if (grid && hipMemcpyAsync(d_grid, grid, grid_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
      break;
    }
    rc = nrphy_pdsch_run(plan, d_tb, d_grid, d_rm, d_scr, 0, ctx->stream);
    if (rc != NRPHY_OK) {
      break;
    }
    rc = NRPHY_ERR_DEVICE;
    nrphy_pdsch_derived_t d;
    nrphy_pdsch_derive(pdu, &d);
    const size_t cw_out = (d.codeword_bits + 7) / 8;
    if (grid && hipMemcpyAsync(grid, d_grid, grid_bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) {
      break;
    }
    if (cw_rm && hipMemcpyAsync(cw_rm, d_rm, cw_out, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) {
      break;
    }
    if (cw_scrambled && hipMemcpyAsync(cw_scrambled, d_scr, cw_out, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) {
      break;
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
      break;
    }
    rc = NRPHY_OK;
  } while (false);
  nrphy_pdsch_plan_destroy(plan);
  return rc;
}

extern "C" int nrphy_pdsch_process_slot_host(nrphy_ctx_t* ctx, uint32_t n_pdu, const nrphy_pdsch_pdu_t* pdus,
                                             const uint8_t* const* tbs, void* grid, uint32_t grid_nof_ports, uint32_t grid_nof_subc)
{
  if (ctx == nullptr || grid == nullptr || (n_pdu != 0 && (pdus == nullptr || tbs == nullptr))) {
    return NRPHY_ERR_ARGUMENT;
  }
  if (n_pdu == 0) {
    return NRPHY_OK;
  }
  // One plan for the slot: every PDU's codeblocks in one launch, all into grid 0.
  std::vector<uint64_t> tb_off(n_pdu);
  std::vector<uint32_t> grid_of(n_pdu, 0);
  size_t                tb_total = 0;
  for (uint32_t i = 0; i != n_pdu; ++i) {
    if (tbs[i] == nullptr) {
      return NRPHY_ERR_ARGUMENT;
    }
    tb_off[i] = tb_total;
    tb_total += ((size_t)pdus[i].tb_size_bytes + 7) & ~(size_t)3; // readable to the next multiple of 4
  }
  nrphy_pdsch_plan_t* plan = nullptr;
  int rc = nrphy_pdsch_plan_create(ctx, n_pdu, pdus, tb_off.data(), grid_of.data(), 1, grid_nof_ports, grid_nof_subc, &plan);
  if (rc != NRPHY_OK) {
    return rc;
  }
  const size_t grid_bytes = (size_t)grid_nof_ports * NRPHY_NSYMB * grid_nof_subc * 4;
  std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
  uint8_t* d_tb   = (uint8_t*)ctx_scratch(ctx, SCRATCH_TB, tb_total + 8);
  uint8_t* d_grid = (uint8_t*)ctx_scratch(ctx, SCRATCH_GRID, grid_bytes);
  rc              = NRPHY_ERR_DEVICE;
  do {
    if (d_tb == nullptr || d_grid == nullptr || hipMemsetAsync(d_tb, 0, tb_total + 8, ctx->stream) != hipSuccess) {
      break;
    }
    bool ok = true;
    for (uint32_t i = 0; ok && i != n_pdu; ++i) {
      ok = hipMemcpyAsync(d_tb + tb_off[i], tbs[i], pdus[i].tb_size_bytes, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    }
    if (!ok || hipMemcpyAsync(d_grid, grid, grid_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
      break;
    }
    rc = nrphy_pdsch_run(plan, d_tb, d_grid, nullptr, nullptr, 0, ctx->stream);
    if (rc != NRPHY_OK) {
      break;
    }
    rc = NRPHY_ERR_DEVICE;
    if (hipMemcpyAsync(grid, d_grid, grid_bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
      break;
    }
    rc = NRPHY_OK;
  } while (false);
  nrphy_pdsch_plan_destroy(plan);
  return rc;
}

extern "C" int nrphy_pdsch_encode_host(nrphy_ctx_t* ctx, const nrphy_pdsch_encoder_cfg_t* cfg, const uint8_t* tb,
                                       uint8_t* codeword_bits, uint8_t* codeword_packed)
{
  if (ctx == nullptr || cfg == nullptr || tb == nullptr || cfg->nof_layers == 0 ||
      cfg->nof_ch_symbols % cfg->nof_layers != 0) {
    return NRPHY_ERR_ARGUMENT;
  }
  nrphy_pdsch_pdu_t pdu;
  std::memset(&pdu, 0, sizeof(pdu));
  pdu.qm              = cfg->qm;
  pdu.rv              = cfg->rv;
  pdu.nof_codewords   = 1;
  pdu.ldpc_base_graph = cfg->base_graph;
  pdu.tb_size_bytes   = cfg->tb_size_bytes;
  pdu.nof_layers      = cfg->nof_layers;
  pdu.nof_ports       = 1;
  pdu.nof_prg         = 1;
  pdu.prg_size_rb     = NRPHY_MAX_RB;
  pdu.tbs_lbrm_bytes  = 1; // unused: N_ref is given
  const EncodeOnly    enc    = {cfg->nof_ch_symbols / cfg->nof_layers, cfg->nref};
  nrphy_pdsch_plan_t* plan   = nullptr;
  uint64_t            tb_off = 0;
  uint32_t            gi     = 0;
  int                 rc     = plan_create(ctx, 1, &pdu, &tb_off, &gi, 1, 1, 12, &enc, &plan);
  if (rc != NRPHY_OK) {
    return rc;
  }
  const size_t cw_bits  = (size_t)cfg->nof_ch_symbols * cfg->qm;
  const size_t cw_bytes = (size_t)(plan->cw_bits / 8);
  const size_t tb_alloc = ((size_t)cfg->tb_size_bytes + 7) & ~(size_t)3;
  std::vector<uint8_t> packed_local;
  uint8_t*             packed = codeword_packed;
  if (packed == nullptr) {
    packed_local.resize((cw_bits + 7) / 8);
    packed = packed_local.data();
  }
  {
    std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
    uint8_t* d_tb = (uint8_t*)ctx_scratch(ctx, SCRATCH_TB, tb_alloc);
    uint8_t* d_rm = (uint8_t*)ctx_scratch(ctx, SCRATCH_CW_RM, cw_bytes);
    rc            = NRPHY_ERR_DEVICE;
    do {
      if (d_tb == nullptr || d_rm == nullptr ||
          hipMemsetAsync(d_tb + (tb_alloc - 8), 0, 8, ctx->stream) != hipSuccess ||
          hipMemcpyAsync(d_tb, tb, cfg->tb_size_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        break;
      }
      rc = nrphy_pdsch_run(plan, d_tb, nullptr, d_rm, nullptr, 0, ctx->stream);
      if (rc != NRPHY_OK) {
        break;
      }
      rc = NRPHY_ERR_DEVICE;
      if (hipMemcpyAsync(packed, d_rm, (cw_bits + 7) / 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess) {
        break;
      }
      rc = NRPHY_OK;
    } while (false);
  }
  nrphy_pdsch_plan_destroy(plan);
  if (rc == NRPHY_OK && codeword_bits != nullptr) { // the reference's codeword span: one bit per byte
    for (size_t i = 0; i != cw_bits; ++i) {
      codeword_bits[i] = (packed[i >> 3] >> (7U - (i & 7U))) & 1U;
    }
  }
  return rc;
}

namespace {

// The walk of ldpc_rate_dematcher_impl::allot_llrs (ldpc_rate_dematcher_impl.cpp:118-200) over the soft buffer, with
// the data taken out: which ranges it clears, fills, copies into and adds to, in its order.  Returns false when the
void build_dematch_ops(std::vector<DematchOp>& ops, unsigned block_length, unsigned buffer_length, unsigned k0,
                       unsigned nof_systematic, unsigned nof_filler, unsigned e, bool new_data)
{
  const unsigned nof_info = nof_systematic - nof_filler;
  bool           copying  = new_data;
  unsigned       pos = k0, taken = 0;
  ops.clear();
  auto emit = [&](uint32_t kind, unsigned begin, unsigned count, unsigned src) {
    if (count != 0) {
      ops.push_back({kind, begin, count, src});
    }
  };
  while (taken != e) {
    unsigned left = e - taken;
    if (pos < nof_info) {
      const unsigned n = std::min(nof_info - pos, left);
      if (copying) {
        emit(DEMATCH_ZERO, 0, pos, 0);
      }
      emit(copying ? DEMATCH_COPY : DEMATCH_COMBINE, pos, n, taken);
      pos += n;
      taken += n;
      left -= n;
    } else if (copying) {
      emit(DEMATCH_ZERO, 0, nof_info, 0);
    }
    if (copying) {
      emit(DEMATCH_FILL, nof_info, nof_filler, 0);
    }
    pos = std::max(pos, nof_systematic);
    const unsigned n = std::min(buffer_length - pos, left);
    emit(copying ? DEMATCH_COPY : DEMATCH_COMBINE, pos, n, taken);
    pos = (pos + n) % buffer_length;
    taken += n;
    if (taken != e) {
      copying = false;
    }
  }
  if (copying && pos != 0) {
    // the reference clears this many soft bits at the end of the full-length block, wherever the buffer ends
    emit(DEMATCH_ZERO, block_length - (buffer_length - pos), buffer_length - pos, 0);
  }
}

// n_outer groups (transport blocks) of n_cb codeblocks: codeblock (g, i) reads d_in + g * in_outer + i * in_stride and
// owns the soft buffer d_soft + g * soft_outer + i * soft_stride.
int rate_dematch_batch(nrphy_ctx_t* ctx, const nrphy_ldpc_rate_dematcher_cfg_t* cfg, uint32_t n_cb, uint32_t n_outer,
                       const int8_t* d_in, uint32_t in_stride_bytes, size_t in_outer, int8_t* d_soft,
                       uint32_t soft_stride_bytes, size_t soft_outer, int new_data, void* stream);

} // namespace

extern "C" int nrphy_ldpc_rate_dematch(nrphy_ctx_t* ctx, const nrphy_ldpc_rate_dematcher_cfg_t* cfg, uint32_t n_cb,
                                       const int8_t* d_in, uint32_t in_stride_bytes, int8_t* d_soft,
                                       uint32_t soft_stride_bytes, int new_data, void* stream)
{
  return rate_dematch_batch(ctx, cfg, n_cb, 1, d_in, in_stride_bytes, 0, d_soft, soft_stride_bytes, 0, new_data, stream);
}

namespace {

int rate_dematch_batch(nrphy_ctx_t* ctx, const nrphy_ldpc_rate_dematcher_cfg_t* cfg, uint32_t n_cb, uint32_t n_outer,
                       const int8_t* d_in, uint32_t in_stride_bytes, size_t in_outer, int8_t* d_soft,
                       uint32_t soft_stride_bytes, size_t soft_outer, int new_data, void* stream)
{
  if (ctx == nullptr || cfg == nullptr || d_in == nullptr || d_soft == nullptr ||
      (cfg->base_graph != 1 && cfg->base_graph != 2) || cfg->rv > 3 || lifting_position(cfg->lifting_size) < 0 ||
      (cfg->qm != 1 && cfg->qm != 2 && cfg->qm != 4 && cfg->qm != 6 && cfg->qm != 8) || cfg->rm_length == 0 ||
      cfg->rm_length > 35 * 8448 /* ldpc::MAX_CODEBLOCK_RM_SIZE */ || cfg->rm_length % cfg->qm != 0) {
    return NRPHY_ERR_ARGUMENT;
  }
  static const double shift_bg1[4] = {0, 17, 33, 56}, shift_bg2[4] = {0, 13, 25, 43};
  const unsigned      zc = cfg->lifting_size, n_short = (cfg->base_graph == 1) ? 66 : 50;
  const unsigned      block_length   = n_short * zc;
  const unsigned      buffer_length  = (cfg->nref > 0 && cfg->nref < block_length) ? cfg->nref : block_length;
  const unsigned      nof_systematic = (((cfg->base_graph == 1) ? 22 : 10) - 2) * zc;
  if (cfg->nof_filler_bits >= nof_systematic || buffer_length <= nof_systematic || in_stride_bytes < cfg->rm_length ||
      soft_stride_bytes < block_length) {
    return NRPHY_ERR_ARGUMENT;
  }
  // ldpc_rate_dematcher_impl.cpp:94-95: k0 of TS 38.212 Table 5.4.2.1-2 in double precision
  const double   frac = (((cfg->base_graph == 1) ? shift_bg1 : shift_bg2)[cfg->rv] * buffer_length) / block_length;
  const unsigned k0   = (unsigned)((uint16_t)std::floor(frac)) * zc;
  DematchLaunch          p;
  std::vector<DematchOp> ops;
  build_dematch_ops(ops, block_length, buffer_length, k0, nof_systematic, cfg->nof_filler_bits, cfg->rm_length,
                    new_data != 0);
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t   s = stream ? (hipStream_t)stream : ctx->stream;
  StreamStaging staging(s);
  p.n_ops   = (uint32_t)ops.size();
  p.ops_ext = nullptr;
  if (ops.size() <= MAX_DEMATCH_OPS) {
    std::copy(ops.begin(), ops.end(), p.ops);
  } else {
    // Heavy repetition (rm_length of dozens of buffer lengths): the list goes through device memory of this call's
    // own, allocated, filled and released in stream order.
    DematchOp* d_ops = (DematchOp*)staging.alloc(ops.size() * sizeof(DematchOp));
    if (d_ops == nullptr) {
      return NRPHY_ERR_DEVICE;
    }
    HIP_TRY(hipMemcpyAsync(d_ops, ops.data(), ops.size() * sizeof(DematchOp), hipMemcpyHostToDevice, s));
    p.ops_ext = d_ops;
  }
  {
    // Do the copying / clearing / filling operations cover the whole block?  Then no soft bit keeps its old value and the
    // kernel skips reading the buffer (always so for a first transmission that does not wrap; never with combining).
    std::vector<std::pair<uint32_t, uint32_t>> ranges;
    bool combines = false;
    for (const DematchOp& op : ops) {
      combines = combines || op.kind == DEMATCH_COMBINE;
      ranges.emplace_back(op.begin, op.begin + op.count);
    }
    std::sort(ranges.begin(), ranges.end());
    uint32_t reach = 0;
    for (const auto& r : ranges) {
      if (r.first > reach) {
        break;
      }
      reach = std::max(reach, r.second);
    }
    p.skip_load = (!combines && reach >= block_length) ? 1U : 0U;
    // Pairwise disjoint destination ranges (sorted by begin: each starts where the previous one has ended or later)?
    p.disjoint = 1U;
    for (size_t i = 1; i < ranges.size(); ++i) {
      if (ranges[i].first < ranges[i - 1].second) {
        p.disjoint = 0U;
      }
    }
  }
  p.in           = d_in;
  p.out          = d_soft;
  p.in_stride    = in_stride_bytes;
  p.out_stride   = soft_stride_bytes;
  p.block_length = block_length;
  p.qm           = cfg->qm;
  p.cols         = cfg->rm_length / cfg->qm;
  p.in_stride_outer  = (uint32_t)in_outer;
  p.out_stride_outer = (uint32_t)soft_outer;
  if (in_outer > 0xFFFFFFFFULL || soft_outer > 0xFFFFFFFFULL) {
    return NRPHY_ERR_CAPACITY;
  }
  HIP_TRY(launch_ldpc_dematch(p, n_cb, s, n_outer));
  return NRPHY_OK;
}

} // namespace

extern "C" int nrphy_ldpc_rate_dematch_host(nrphy_ctx_t* ctx, const nrphy_ldpc_rate_dematcher_cfg_t* cfg,
                                            const int8_t* in, int8_t* soft_buffer, int new_data)
{
  // Everything that sizes a copy below is checked here, before the validating device-pointer call runs.
  if (ctx == nullptr || cfg == nullptr || in == nullptr || soft_buffer == nullptr ||
      (cfg->base_graph != 1 && cfg->base_graph != 2) || lifting_position(cfg->lifting_size) < 0 || cfg->rm_length == 0 ||
      cfg->rm_length > 35 * 8448) {
    return NRPHY_ERR_ARGUMENT;
  }
  const unsigned block_length = ((cfg->base_graph == 1) ? 66U : 50U) * cfg->lifting_size;
  int8_t *       d_in = nullptr, *d_soft = nullptr;
  int            rc = NRPHY_ERR_DEVICE;
  do {
    if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc((void**)&d_in, cfg->rm_length + 16) != hipSuccess ||
        hipMalloc((void**)&d_soft, block_length + 16) != hipSuccess ||
        hipMemcpy(d_in, in, cfg->rm_length, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_soft, soft_buffer, block_length, hipMemcpyHostToDevice) != hipSuccess) {
      break;
    }
    rc = nrphy_ldpc_rate_dematch(ctx, cfg, 1, d_in, cfg->rm_length, d_soft, block_length, new_data, ctx->stream);
    if (rc != NRPHY_OK) {
      break;
    }
    rc = NRPHY_ERR_DEVICE;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy(soft_buffer, d_soft, block_length, hipMemcpyDeviceToHost) != hipSuccess) {
      break;
    }
    rc = NRPHY_OK;
  } while (false);
  (void)hipFree(d_in);
  (void)hipFree(d_soft);
  return rc;
}

namespace {

// Label bit k of a Gray-mapped PAM level (TS 38.211 Section 5.1): bit 0 is the sign; bit k >= 1 is set in the outer half of
// the magnitude's current range, which then folds around its middle.
unsigned pam_label_bit(int level, unsigned k, unsigned m)
{
  if (k == 0) {
    return level < 0;
  }
  int t = std::abs(level), r = 1 << m;
  for (unsigned j = 1;; ++j) {
    const unsigned bit = t > r / 2;
    if (j == k) {
      return bit;
    }
    t = std::abs(t - r / 2);
    r /= 2;
  }
}

// Max-log LLR of bit pair k as a piecewise-linear function on n intervals of width_units / sqrt(norm): in the interval whose
// nearest points with the bit 0 / 1 are a0 / a1, LLR(v) = 2 (a0 - a1) v + (a1^2 - a0^2).  The values are those of the
// reference's tables (demodulation_mapper_qam64.cpp:49-92, demodulation_mapper_qam256.cpp:47-160), derived, not copied.
void demod_interval_table(DemodLaunch& p, unsigned pair, unsigned m, float a, unsigned norm, unsigned width_units, unsigned n)
{
  const int M            = 1 << m;
  p.nof_intervals[pair]  = n;
  p.width[pair]          = (float)width_units * a;
  p.rcp_width[pair]      = 1.0F / p.width[pair];
  for (unsigned j = 0; j != n; ++j) {
    const double centre = ((double)j - (double)n / 2 + 0.5) * (double)width_units;
    int          a0 = 0, a1 = 0;
    double       d0 = 1e30, d1 = 1e30;
    for (int q = 0; q != M; ++q) {
      const int    level = 2 * q - (M - 1);
      const double d     = std::fabs(centre - level);
      if (pam_label_bit(level, pair, m)) {
        if (d < d1) {
          d1 = d, a1 = level;
        }
      } else if (d < d0) {
        d0 = d, a0 = level;
      }
    }
    p.slope[pair][j]     = (float)(2 * (a0 - a1)) * a;
    p.intercept[pair][j] = (float)(a1 * a1 - a0 * a0) / (float)norm;
  }
}

bool demod_launch_params(uint32_t modulation, uint32_t span_len, DemodLaunch& p)
{
  std::memset(&p, 0, sizeof(p));
  p.modulation = modulation;
  p.span_len   = span_len;
  uint32_t batch = 0;
  switch (modulation) {
    case NRPHY_MOD_PI2_BPSK:
    case NRPHY_MOD_BPSK:
      p.range = 24.0F;
      break;
    case NRPHY_MOD_QPSK:
      p.range = 24.0F;
      batch   = 16;
      break;
    case NRPHY_MOD_QAM16:
      p.range           = 20.0F;
      batch             = 8;
      p.qam16_gain      = 4.0F * (1.0F / std::sqrt(10.0F));
      p.qam16_threshold = 2 * (1.0F / std::sqrt(10.0F));
      break;
    case NRPHY_MOD_QAM64: {
      p.range       = 20.0F;
      batch         = 16;
      const float a = 1.0F / std::sqrt(42.0F);
      demod_interval_table(p, 0, 3, a, 42, 2, 8);
      demod_interval_table(p, 1, 3, a, 42, 2, 8);
      demod_interval_table(p, 2, 3, a, 42, 4, 4);
      break;
    }
    case NRPHY_MOD_QAM256: {
      p.range       = 20.0F;
      batch         = 4;
      const float a = 1.0F / std::sqrt(170.0F);
      demod_interval_table(p, 0, 4, a, 170, 2, 16);
      demod_interval_table(p, 1, 4, a, 170, 2, 16);
      demod_interval_table(p, 2, 4, a, 170, 2, 16);
      demod_interval_table(p, 3, 4, a, 170, 4, 8);
      break;
    }
    default:
      return false;
  }
  p.scale      = 120.0F / p.range;
  p.nof_vector = batch ? span_len / batch * batch : 0;
  return true;
}

} // namespace

extern "C" int nrphy_demodulate_soft(nrphy_ctx_t* ctx, uint32_t modulation, uint32_t nof_spans, uint32_t span_len,
                                     const float* d_symbols, const float* d_noise_vars, int8_t* d_llr, void* stream)
{
  DemodLaunch p;
  if (ctx == nullptr || !demod_launch_params(modulation, span_len, p) || nof_spans > 65535U) {
    return NRPHY_ERR_ARGUMENT;
  }
  if (nof_spans == 0 || span_len == 0) {
    return NRPHY_OK;
  }
  if (d_symbols == nullptr || d_noise_vars == nullptr || d_llr == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(launch_demodulate_soft(p, nof_spans, d_symbols, d_noise_vars, d_llr, stream ? (hipStream_t)stream : ctx->stream));
  return NRPHY_OK;
}

extern "C" int nrphy_demodulate_soft_host(nrphy_ctx_t* ctx, uint32_t modulation, uint32_t nof_symbols, const float* symbols,
                                          const float* noise_vars, int8_t* llr)
{
  DemodLaunch p;
  if (ctx == nullptr || !demod_launch_params(modulation, nof_symbols, p)) {
    return NRPHY_ERR_ARGUMENT;
  }
  if (nof_symbols == 0) {
    return NRPHY_OK;
  }
  if (symbols == nullptr || noise_vars == nullptr || llr == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  const uint32_t              qm = modulation == NRPHY_MOD_PI2_BPSK ? 1U : modulation;
  std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
  HIP_TRY(hipSetDevice(ctx->device));
  StreamStaging sym(ctx->stream), nv(ctx->stream), out(ctx->stream);
  float*  d_sym = (float*)sym.alloc((size_t)nof_symbols * 8);
  float*  d_nv  = (float*)nv.alloc((size_t)nof_symbols * 4);
  int8_t* d_out = (int8_t*)out.alloc((size_t)nof_symbols * qm + 16);
  if (d_sym == nullptr || d_nv == nullptr || d_out == nullptr) {
    return NRPHY_ERR_DEVICE;
  }
  HIP_TRY(hipMemcpyAsync(d_sym, symbols, (size_t)nof_symbols * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(d_nv, noise_vars, (size_t)nof_symbols * 4, hipMemcpyHostToDevice, ctx->stream));
  const int rc = nrphy_demodulate_soft(ctx, modulation, 1, nof_symbols, d_sym, d_nv, d_out, ctx->stream);
  if (rc != NRPHY_OK) {
    return rc;
  }
  HIP_TRY(hipMemcpyAsync(llr, d_out, (size_t)nof_symbols * qm, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return NRPHY_OK;
}

extern "C" int nrphy_llr_descramble(nrphy_ctx_t* ctx, uint32_t n_cw, const uint32_t* d_c_init, uint32_t length,
                                    const int8_t* d_in, size_t in_stride, int8_t* d_out, size_t out_stride, void* stream)
{
  if (ctx == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  if (n_cw == 0 || length == 0) {
    return NRPHY_OK;
  }
  // The x1 table holds the first 2^21 sequence bits (the longest PUSCH / PDSCH codeword is 1.47 Mbit).
  if (d_c_init == nullptr || d_in == nullptr || d_out == nullptr || length > (uint32_t)GOLD_X1_WORDS * 32U ||
      (n_cw > 1 && (in_stride < length || out_stride < length)) || n_cw > 65535U) {
    return NRPHY_ERR_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(launch_llr_descramble(ctx->d_gold, ctx->d_x1, d_c_init, n_cw, length, d_in, in_stride, d_out, out_stride,
                                stream ? (hipStream_t)stream : ctx->stream));
  return NRPHY_OK;
}

extern "C" int nrphy_llr_descramble_host(nrphy_ctx_t* ctx, uint32_t c_init, uint32_t length, const int8_t* in, int8_t* out)
{
  if (ctx == nullptr || (length != 0 && (in == nullptr || out == nullptr))) {
    return NRPHY_ERR_ARGUMENT;
  }
  if (length == 0) {
    return NRPHY_OK;
  }
  int8_t*   d_buf = nullptr;
  uint32_t* d_ci  = nullptr;
  int       rc    = NRPHY_ERR_DEVICE;
  do {
    if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc((void**)&d_buf, (size_t)length + 16) != hipSuccess ||
        hipMalloc((void**)&d_ci, 16) != hipSuccess || hipMemcpy(d_buf, in, length, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_ci, &c_init, sizeof(c_init), hipMemcpyHostToDevice) != hipSuccess) {
      break;
    }
    rc = nrphy_llr_descramble(ctx, 1, d_ci, length, d_buf, length, d_buf, length, ctx->stream); // in place, as the caller does
    if (rc != NRPHY_OK) {
      break;
    }
    rc = NRPHY_ERR_DEVICE;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipMemcpy(out, d_buf, length, hipMemcpyDeviceToHost) != hipSuccess) {
      break;
    }
    rc = NRPHY_OK;
  } while (false);
  (void)hipFree(d_buf);
  (void)hipFree(d_ci);
  return rc;
}

namespace {

// Decoder graph of (base graph, lifting size): all edges of TS 38.212 Tables 5.3.2-2/-3, row by row.
const DecoderGraph* get_decoder_graph(nrphy_ctx* ctx, unsigned bg, unsigned zc)
{
  const int pos = lifting_position(zc);
  if (pos < 0) {
    return nullptr;
  }
  const unsigned slot = (bg - 1) * NOF_LIFTING_SIZES + (unsigned)pos;
  if (ctx->d_dec_graph[slot] == nullptr) {
    std::vector<DecoderGraph> g(1);
    std::memset(&g[0], 0, sizeof(DecoderGraph));
    const nr_ldpc_edge_t* edges   = (bg == 1) ? NR_LDPC_BG1_EDGES : NR_LDPC_BG2_EDGES;
    const unsigned        n_edges = (bg == 1) ? NR_LDPC_BG1_NOF_EDGES : NR_LDPC_BG2_NOF_EDGES;
    const unsigned        rows    = (bg == 1) ? 46 : 42;
    const int             ils     = lifting_set_index(zc);
    unsigned              count   = 0;
    for (unsigned m = 0; m != rows; ++m) {
      g[0].row_ptr[m] = count;
      for (unsigned e = 0; e != n_edges; ++e) {
        if (edges[e].row == m) {
          g[0].edge[count++] = (((uint32_t)edges[e].col * zc) << 16) | (edges[e].shift[ils] % zc); // 67 * 384 < 2^16
        }
      }
    }
    for (unsigned m = rows; m != MAX_BG_ROWS + 2; ++m) {
      g[0].row_ptr[m] = count;
    }
    for (unsigned m = 0, pairs = 0; m != MAX_BG_ROWS + 2; ++m) {
      g[0].pair_ptr[m] = pairs;
      pairs += m + 1 < MAX_BG_ROWS + 2 ? (g[0].row_ptr[m + 1] - g[0].row_ptr[m] + 1) / 2 : 0;
    }
    for (unsigned m = 0; m != rows; ++m) {
      // the kernel is specialised for the row degrees the two base graphs have
      const unsigned deg = g[0].row_ptr[m + 1] - g[0].row_ptr[m];
      if (!((deg >= 3 && deg <= 10) || deg == 19)) {
        return nullptr;
      }
    }
    for (unsigned m = 0, quads = 0; m != MAX_BG_ROWS + 2; ++m) {
      g[0].quad_ptr[m] = quads;
      quads += m + 1 < MAX_BG_ROWS + 2 ? (g[0].row_ptr[m + 1] - g[0].row_ptr[m] + 3) / 4 : 0;
    }
    if ((zc & 1U) == 0) {
      // Two checks per lane with the messages in LDS: where lane j finds the soft bits of checks j and j + Zc / 2 on every edge
      // -- (variable node * Zc + (check + shift) mod Zc), the first in the low half of a word, the second in the high half --
      // as [row of four edges][lane][edge in the row]: the same for every codeblock and iteration, so the kernel reads the
      // words of a layer (sixteen bytes per lane and row, a layer ahead) instead of computing them (7 vector instructions per
      // edge and pass).  Five spare rows: the kernel always reads five rows from a layer's first.
      const unsigned        half = zc / 2, total = g[0].quad_ptr[rows] + 5;
      std::vector<uint32_t> addr((size_t)total * half * 4, 0);
      for (unsigned m = 0; m != rows; ++m) {
        const unsigned e0 = g[0].row_ptr[m], deg = g[0].row_ptr[m + 1] - e0;
        for (unsigned t = 0; t != deg; ++t) {
          const unsigned base = g[0].edge[e0 + t] >> 16, shift = g[0].edge[e0 + t] & 0xFFFFU;
          for (unsigned j = 0; j != half; ++j) {
            const unsigned a1 = base + (j + shift) % zc, a2 = base + (j + half + shift) % zc; // < 68 * 384 < 2^16
            addr[((size_t)(g[0].quad_ptr[m] + t / 4) * half + j) * 4 + t % 4] = a1 | (a2 << 16);
          }
        }
      }
      if (upload(&ctx->d_dec_addr[slot], addr.data(), addr.size() * sizeof(uint32_t)) != hipSuccess) {
        return nullptr;
      }
    }
    if (upload(&ctx->d_dec_graph[slot], g.data(), sizeof(DecoderGraph)) != hipSuccess) {
      return nullptr;
    }
  }
  return ctx->d_dec_graph[slot];
}

const uint32_t* get_decoder_pair_addresses(nrphy_ctx* ctx, unsigned bg, unsigned zc)
{
  const int pos = lifting_position(zc);
  return pos < 0 ? nullptr : ctx->d_dec_addr[(bg - 1) * NOF_LIFTING_SIZES + (unsigned)pos];
}

// Early-stop tables: word w of the n-bit message (32 bits, the last one n mod 32) is followed by n - 32 w - bits(w)
// bits; the message is a multiple of the generator g iff the sum of word(w) * x^(that) vanishes mod g.  Per word eight
// nibble tables, tab[w][k][v] = (v x^(4 k)) * x^(bits after word w) mod g: the product is eight independent look-ups
// instead of a 32-step shift-and-add per word and iteration (which was about 15 % of an early-stop iteration).
const uint32_t* get_decoder_crc_weights(nrphy_ctx* ctx, uint32_t poly, uint32_t order, uint32_t n_msg)
{
  const uint64_t key = ((uint64_t)poly << 32) | n_msg;
  auto           it  = ctx->d_dec_crc.find(key);
  if (it != ctx->d_dec_crc.end()) {
    return it->second;
  }
  const CrcField        field = {poly, order};
  const uint32_t        nw    = (n_msg + 31) / 32;
  std::vector<uint32_t> w((size_t)nw * DEC_CRC_TABLE_WORDS);
  for (uint32_t i = 0; i != nw; ++i) {
    const uint32_t bits   = std::min<uint32_t>(32, n_msg - 32 * i);
    const uint32_t weight = field.xpow((int64_t)n_msg - 32 * i - bits);
    for (uint32_t k = 0; k != 8; ++k) {
      for (uint32_t v = 0; v != 16; ++v) {
        w[(size_t)i * DEC_CRC_TABLE_WORDS + 16 * k + v] = field.mul(weight, v << (4 * k));
      }
    }
  }
  uint32_t* d = nullptr;
  if (upload(&d, w.data(), w.size() * sizeof(uint32_t)) != hipSuccess) {
    return nullptr;
  }
  ctx->d_dec_crc[key] = d;
  return d;
}

} // namespace

namespace {
// skip / ok_flags: per-codeblock HARQ state of a transport-block decoder; crc_at_end: no early stop.
// expected_extent: how many of the nof_llr soft bits the caller expects to be in use (the rest zero), 0 = all of them; it
// only sizes the launch's LDS (messages per edge in LDS when the layers that extent needs are few), never the result.
int ldpc_decode_batch(nrphy_ctx_t* ctx, const nrphy_ldpc_decoder_cfg_t* cfg, uint32_t n_cb, const int8_t* d_llr,
                      uint32_t llr_stride_bytes, uint8_t* d_out, uint32_t out_stride_bytes, uint32_t* d_iterations,
                      const uint8_t* d_skip, uint8_t* d_ok_flags, bool crc_at_end, void* d_scratch, void* stream,
                      uint32_t expected_extent = 0);

// The caller-owned scratch of a decoder launch: a pool of check-record slots + the bitmap that hands them out.
struct DecoderScratch {
  uint32_t nof_slots, nof_layers_max;
  uint64_t slot_bytes, records_bytes, flags_bytes, total_bytes;
};
bool decoder_scratch_layout(const nrphy_ctx_t* ctx, const nrphy_ldpc_decoder_cfg_t& cfg, uint32_t n_cb, DecoderScratch& s)
{
  const unsigned bg_k = (cfg.base_graph == 1) ? 22 : 10, zc = cfg.lifting_size;
  if ((cfg.base_graph != 1 && cfg.base_graph != 2) || lifting_position(zc) < 0 || n_cb == 0) {
    return false;
  }
  const uint32_t nof_nodes = std::max<uint32_t>(divide_ceil(cfg.nof_llr, zc) + 2, bg_k + 4);
  s.nof_layers_max         = nof_nodes - bg_k;
  // Workgroups of the decoder the device can hold at once: 32 wavefronts per CU (the kernel is built for 8 per SIMD),
  // ceil(Zc / 64) per workgroup; one slot per CU more as a margin.  A smaller batch gets a slot per codeblock.
  const uint32_t waves    = divide_ceil(zc, 64);
  const uint32_t resident = ctx->nof_cus * (32U / waves) + ctx->nof_cus;
  s.nof_slots             = std::min<uint32_t>(n_cb, resident);
  if (ctx->tune.decoder_slots_all) { // profiling aid (profiles/): a slot per codeblock, i.e. no pooling
    s.nof_slots = n_cb;
  }
  // A slot holds a codeblock's check records (8 bytes per lifted check and layer) or, with two checks per lane, its messages
  // per edge: rows of two edges, 2 Zc bytes each, five spare rows (the kernel requests five rows from a layer's first).
  {
    const nr_ldpc_edge_t* edges   = (cfg.base_graph == 1) ? NR_LDPC_BG1_EDGES : NR_LDPC_BG2_EDGES;
    const unsigned        n_edges = (cfg.base_graph == 1) ? NR_LDPC_BG1_NOF_EDGES : NR_LDPC_BG2_NOF_EDGES;
    std::vector<uint32_t> degree(s.nof_layers_max, 0);
    for (unsigned e = 0; e != n_edges; ++e) {
      if (edges[e].row < s.nof_layers_max) {
        ++degree[edges[e].row];
      }
    }
    uint64_t rows = 5;
    for (uint32_t dg : degree) {
      rows += (dg + 1) / 2;
    }
    s.slot_bytes = (std::max<uint64_t>((uint64_t)s.nof_layers_max * zc * sizeof(uint2), rows * 2 * zc) + 255) & ~(uint64_t)255;
  }
  s.records_bytes         = (uint64_t)s.nof_slots * s.slot_bytes;
  s.flags_bytes           = ((uint64_t)s.nof_slots * 4 + 255) & ~(uint64_t)255;
  s.total_bytes           = ((s.records_bytes + 255) & ~(uint64_t)255) + s.flags_bytes;
  return true;
}
} // namespace

extern "C" int nrphy_ldpc_decoder_scratch_bytes(nrphy_ctx_t* ctx, const nrphy_ldpc_decoder_cfg_t* cfg, uint32_t n_cb,
                                                uint64_t* bytes)
{
  DecoderScratch s;
  if (ctx == nullptr || cfg == nullptr || bytes == nullptr || !decoder_scratch_layout(ctx, *cfg, n_cb, s)) {
    return NRPHY_ERR_ARGUMENT;
  }
  *bytes = s.total_bytes;
  return NRPHY_OK;
}

extern "C" int nrphy_ldpc_decoder_prepare(nrphy_ctx_t* ctx, const nrphy_ldpc_decoder_cfg_t* cfg)
{
  if (ctx == nullptr || cfg == nullptr || (cfg->base_graph != 1 && cfg->base_graph != 2) || lifting_position(cfg->lifting_size) < 0 ||
      (cfg->crc_poly != 0 && cfg->crc_poly != 16 && cfg->crc_poly != 0x24A && cfg->crc_poly != 0x24B)) {
    return NRPHY_ERR_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
  if (get_decoder_graph(ctx, cfg->base_graph, cfg->lifting_size) == nullptr) {
    return NRPHY_ERR_DEVICE;
  }
  if (cfg->crc_poly != 0) {
    const uint32_t order = cfg->crc_poly == 16 ? 16 : 24;
    const uint32_t poly  = (cfg->crc_poly == 16) ? 0x11021U : (cfg->crc_poly == 0x24B ? 0x1800063U : 0x1864CFBU);
    const uint32_t K     = ((cfg->base_graph == 1) ? 22U : 10U) * cfg->lifting_size;
    if (cfg->nof_filler_bits >= K || get_decoder_crc_weights(ctx, poly, order, K - cfg->nof_filler_bits) == nullptr) {
      return NRPHY_ERR_DEVICE;
    }
  }
  return NRPHY_OK;
}

extern "C" int nrphy_ldpc_decode(nrphy_ctx_t* ctx, const nrphy_ldpc_decoder_cfg_t* cfg, uint32_t n_cb,
                                 const int8_t* d_llr, uint32_t llr_stride_bytes, uint8_t* d_out,
                                 uint32_t out_stride_bytes, uint32_t* d_iterations, void* d_scratch, void* stream)
{
  return ldpc_decode_batch(ctx, cfg, n_cb, d_llr, llr_stride_bytes, d_out, out_stride_bytes, d_iterations, nullptr,
                           nullptr, false, d_scratch, stream);
}

namespace {

int ldpc_decode_batch(nrphy_ctx_t* ctx, const nrphy_ldpc_decoder_cfg_t* cfg, uint32_t n_cb, const int8_t* d_llr,
                      uint32_t llr_stride_bytes, uint8_t* d_out, uint32_t out_stride_bytes, uint32_t* d_iterations,
                      const uint8_t* d_skip, uint8_t* d_ok_flags, bool crc_at_end, void* d_scratch, void* stream,
                      uint32_t expected_extent)
{
  if (ctx == nullptr || cfg == nullptr || d_llr == nullptr || d_out == nullptr || d_scratch == nullptr ||
      (cfg->base_graph != 1 && cfg->base_graph != 2) || cfg->max_iterations == 0 ||
      !(cfg->scaling_factor > 0.0F && cfg->scaling_factor < 1.0F) ||
      (cfg->crc_poly != 0 && cfg->crc_poly != 16 && cfg->crc_poly != 0x24A && cfg->crc_poly != 0x24B)) {
    return NRPHY_ERR_ARGUMENT;
  }
  const unsigned bg_k = (cfg->base_graph == 1) ? 22 : 10, n_full = (cfg->base_graph == 1) ? 68 : 52;
  const unsigned zc = cfg->lifting_size, K = bg_k * zc;
  // ldpc_decoder_impl.cpp:70-86: between the message plus two blocks and the whole (shortened) codeblock.
  if (lifting_position(zc) < 0 || cfg->nof_llr < K + 2 * zc || cfg->nof_llr > (n_full - 2) * zc ||
      cfg->nof_filler_bits >= K || llr_stride_bytes < cfg->nof_llr || out_stride_bytes < (K + 7) / 8) {
    return NRPHY_ERR_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  LdpcDecodeLaunch p;
  {
    std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
    p.graph     = get_decoder_graph(ctx, cfg->base_graph, zc);
    p.pair_addr = get_decoder_pair_addresses(ctx, cfg->base_graph, zc);
  }
  if (p.graph == nullptr) {
    return NRPHY_ERR_DEVICE;
  }
  p.zc             = zc;
  p.bg_k           = bg_k;
  p.nof_nodes      = std::max<uint32_t>(divide_ceil(cfg->nof_llr, zc) + 2, bg_k + 4);
  p.nof_layers_max = p.nof_nodes - bg_k;
  p.nof_llr        = cfg->nof_llr;
  p.llr_stride     = llr_stride_bytes;
  p.out_stride     = out_stride_bytes;
  p.nof_filler     = cfg->nof_filler_bits;
  p.crc_order      = (cfg->crc_poly == 0) ? 0 : (cfg->crc_poly == 16 ? 16 : 24);
  p.crc_poly       = (cfg->crc_poly == 16) ? 0x11021U : (cfg->crc_poly == 0x24B ? 0x1800063U : 0x1864CFBU);
  p.max_iterations = cfg->max_iterations;
  p.scaling_factor = cfg->scaling_factor;
  p.llr            = d_llr;
  p.out            = d_out;
  p.iterations     = d_iterations;
  p.skip           = d_skip;
  p.ok_flags       = d_ok_flags;
  p.crc_at_end     = crc_at_end ? 1U : 0U;
  p.knob_pairs     = ctx->tune.decoder_pairs;
  p.knob_msg       = ctx->tune.decoder_msg;
  p.knob_ldsmsg    = ctx->tune.decoder_ldsmsg;
  {
    // LDS for the messages-per-edge form (ldpc_decoder.hip) at the expected extent: the soft bits of the layers it needs (the
    // kernel's own rule: ldpc_decoder_impl.cpp:88-116), then one byte per edge and lifted check of those layers.
    const uint32_t extent = (expected_extent == 0 || expected_extent > cfg->nof_llr) ? cfg->nof_llr : expected_extent;
    uint32_t       cb_len = std::max<uint32_t>(extent + 2 * zc, K + 4 * zc);
    cb_len                = divide_ceil(cb_len, zc) * zc;
    const uint32_t        layers  = cb_len / zc - bg_k;
    const nr_ldpc_edge_t* edges   = (cfg->base_graph == 1) ? NR_LDPC_BG1_EDGES : NR_LDPC_BG2_EDGES;
    const unsigned        n_edges = (cfg->base_graph == 1) ? NR_LDPC_BG1_NOF_EDGES : NR_LDPC_BG2_NOF_EDGES;
    std::vector<uint32_t> degree(layers, 0);
    for (unsigned e = 0; e != n_edges; ++e) {
      if (edges[e].row < layers) {
        ++degree[edges[e].row];
      }
    }
    uint32_t rows = 0; // of two edges each
    for (uint32_t dg : degree) {
      rows += (dg + 1) / 2;
    }
    p.lm_lds_bytes = ((cb_len + 48U + 15U) & ~15U) + rows * 2 * zc;
    // The scaling of the minima by arithmetic instead of a table look-up, where it gives the table's values
    // (round half away from zero: ldpc_decoder_generic.cpp:69-79).
    p.scale_arithmetic = 1;
    p.scale_fixed      = (uint32_t)std::lround((double)cfg->scaling_factor * 512.0); // < 512: 120 * F + 256 stays below 2^16
    bool fixed_ok      = p.scale_fixed < 512;
    for (unsigned m = 0; m <= 120; ++m) {
      const float    x    = (float)m * cfg->scaling_factor;
      const uint32_t want = (uint32_t)(uint8_t)roundf(x);
      if ((uint32_t)(x + 0.5F) != want) {
        p.scale_arithmetic = 0;
      }
      if (((m * p.scale_fixed + 256U) & 0xFFFFU) >> 9 != want) {
        fixed_ok = false;
      }
    }
    if (fixed_ok) {
      p.scale_arithmetic = 2;
    }
  }
  p.crc_weight     = nullptr;
  if (p.crc_order != 0) {
    // Uploaded by nrphy_ldpc_decoder_prepare(); a first use without it allocates and copies here (not capturable).
    std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
    p.crc_weight = get_decoder_crc_weights(ctx, p.crc_poly, p.crc_order, K - cfg->nof_filler_bits);
    if (p.crc_weight == nullptr) {
      return NRPHY_ERR_DEVICE;
    }
  }
  if (n_cb == 0) {
    return NRPHY_OK;
  }
  // Check records: the caller's scratch, a pool of slots shared by the workgroups resident at once.
  DecoderScratch sl;
  if (!decoder_scratch_layout(ctx, *cfg, n_cb, sl)) {
    return NRPHY_ERR_ARGUMENT;
  }
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  p.scratch     = (uint2*)d_scratch;
  p.slot_flags = (uint32_t*)((uint8_t*)d_scratch + ((sl.records_bytes + 255) & ~(uint64_t)255));
  p.nof_slots   = sl.nof_slots;
  p.slot_bytes  = (uint32_t)sl.slot_bytes;
  if (sl.nof_slots < n_cb) {
    HIP_TRY(hipMemsetAsync(p.slot_flags, 0, sl.flags_bytes, s));
  }
  {
    const TraceRange trace("cb_decode");
    HIP_TRY(launch_ldpc_decode(p, n_cb, s));
  }
  return NRPHY_OK;
}

} // namespace

extern "C" int nrphy_ldpc_decode_host(nrphy_ctx_t* ctx, const nrphy_ldpc_decoder_cfg_t* cfg, const int8_t* llr,
                                      uint8_t* message_packed, uint32_t* iterations)
{
  if (ctx == nullptr || cfg == nullptr || llr == nullptr || message_packed == nullptr ||
      (cfg->base_graph != 1 && cfg->base_graph != 2) || lifting_position(cfg->lifting_size) < 0 ||
      cfg->nof_llr > 66U * cfg->lifting_size) {
    return NRPHY_ERR_ARGUMENT;
  }
  const unsigned K     = ((cfg->base_graph == 1) ? 22U : 10U) * cfg->lifting_size;
  int8_t*        d_llr = nullptr;
  uint8_t*       d_out = nullptr;
  void*          d_scratch = nullptr;
  int            rc    = NRPHY_ERR_DEVICE;
  do {
    if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc((void**)&d_llr, cfg->nof_llr + 16) != hipSuccess ||
        hipMalloc((void**)&d_out, (K + 7) / 8 + 16) != hipSuccess ||
        hipMemcpy(d_llr, llr, cfg->nof_llr, hipMemcpyHostToDevice) != hipSuccess) {
      break;
    }
    uint32_t* d_it = (uint32_t*)(d_out + (((K + 7) / 8 + 3) & ~3U));
    uint64_t  scratch_bytes = 0;
    if (nrphy_ldpc_decoder_scratch_bytes(ctx, cfg, 1, &scratch_bytes) != NRPHY_OK) {
      rc = NRPHY_ERR_ARGUMENT;
      break;
    }
    if (hipMalloc(&d_scratch, scratch_bytes) != hipSuccess) {
      break;
    }
    rc = nrphy_ldpc_decode(ctx, cfg, 1, d_llr, cfg->nof_llr, d_out, (K + 7) / 8, d_it, d_scratch, ctx->stream);
    if (rc != NRPHY_OK) {
      break;
    }
    rc = NRPHY_ERR_DEVICE;
    uint32_t it = 0;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy(message_packed, d_out, (K + 7) / 8, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&it, d_it, sizeof(it), hipMemcpyDeviceToHost) != hipSuccess) {
      break;
    }
    if (iterations) {
      *iterations = it;
    }
    rc = NRPHY_OK;
  } while (false);
  (void)hipFree(d_llr);
  (void)hipFree(d_out);
  (void)hipFree(d_scratch);
  return rc;
}

// One codeblock through rate dematcher and decoder with one round trip over the link: what a per-codeblock
// accelerator interface (hal::hw_accelerator_pusch_dec) asks for.
extern "C" int nrphy_pusch_decode_codeblock_host(nrphy_ctx_t* ctx, const nrphy_ldpc_rate_dematcher_cfg_t* dm,
                                                 uint32_t crc_poly, uint32_t max_iterations, float scaling_factor,
                                                 const int8_t* llr, int8_t* soft_buffer, int new_data,
                                                 uint8_t* message_packed, uint32_t* iterations)
{
  if (ctx == nullptr || dm == nullptr || llr == nullptr || soft_buffer == nullptr || message_packed == nullptr ||
      (dm->base_graph != 1 && dm->base_graph != 2) || lifting_position(dm->lifting_size) < 0) {
    return NRPHY_ERR_ARGUMENT;
  }
  const unsigned zc = dm->lifting_size, n = ((dm->base_graph == 1) ? 66U : 50U) * zc;
  const unsigned k = ((dm->base_graph == 1) ? 22U : 10U) * zc, kbytes = (k + 7) / 8;
  const size_t   off_soft = ((size_t)dm->rm_length + 63) & ~(size_t)63, off_out = off_soft + (((size_t)n + 63) & ~(size_t)63);
  const size_t   off_it = off_out + (((size_t)kbytes + 63) & ~(size_t)63);
  std::lock_guard<std::recursive_mutex> host_lock(ctx->host_mutex);
  HIP_TRY(hipSetDevice(ctx->device));
  uint8_t* base = nullptr;
  {
    std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
    base = (uint8_t*)ctx_scratch(ctx, SCRATCH_RX, off_it + 64);
  }
  if (base == nullptr) {
    return NRPHY_ERR_DEVICE;
  }
  HIP_TRY(hipMemcpyAsync(base, llr, dm->rm_length, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(base + off_soft, soft_buffer, n, hipMemcpyHostToDevice, ctx->stream));
  int rc = nrphy_ldpc_rate_dematch(ctx, dm, 1, (const int8_t*)base, dm->rm_length, (int8_t*)(base + off_soft), n, new_data,
                                   ctx->stream);
  if (rc != NRPHY_OK) {
    return rc;
  }
  nrphy_ldpc_decoder_cfg_t dec;
  dec.base_graph      = dm->base_graph;
  dec.lifting_size    = zc;
  dec.nof_filler_bits = dm->nof_filler_bits;
  dec.crc_poly        = crc_poly;
  dec.nof_llr         = n;
  dec.max_iterations  = max_iterations;
  dec.scaling_factor  = scaling_factor;
  uint64_t scratch_bytes = 0;
  if (nrphy_ldpc_decoder_scratch_bytes(ctx, &dec, 1, &scratch_bytes) != NRPHY_OK) {
    return NRPHY_ERR_ARGUMENT;
  }
  void* d_scratch = ctx_scratch(ctx, SCRATCH_DECODER, scratch_bytes); // host_mutex is held for the whole call
  if (d_scratch == nullptr) {
    return NRPHY_ERR_DEVICE;
  }
  rc = nrphy_ldpc_decode(ctx, &dec, 1, (const int8_t*)(base + off_soft), n, base + off_out, kbytes,
                         (uint32_t*)(base + off_it), d_scratch, ctx->stream);
  if (rc != NRPHY_OK) {
    return rc;
  }
  uint32_t it = 0;
  HIP_TRY(hipMemcpyAsync(soft_buffer, base + off_soft, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(message_packed, base + off_out, kbytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(&it, base + off_it, sizeof(it), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (iterations) {
    *iterations = it;
  }
  return NRPHY_OK;
}

// ================================================================================================================
// PUSCH decoder, transport-block level
// ================================================================================================================
namespace {

struct PuschLayout {
  nrphy_pdsch_derived_t d;
  uint32_t              msg_stride; // bytes per decoded message
  uint64_t              off_ok, off_skip, off_iter, off_msg, state_bytes;
};

bool pusch_layout(const nrphy_pusch_decoder_cfg_t& cfg, uint32_t n_tb, PuschLayout& l)
{
  if ((cfg.base_graph != 1 && cfg.base_graph != 2) || (cfg.qm != 1 && cfg.qm != 2 && cfg.qm != 4 && cfg.qm != 6 && cfg.qm != 8) ||
      cfg.rv > 3 || cfg.nof_layers == 0 || cfg.nof_layers > NRPHY_MAX_LAYERS || cfg.tb_size_bytes == 0 ||
      cfg.tb_size_bytes > NRPHY_MAX_TB_BYTES || cfg.nof_ch_symbols == 0 || cfg.nof_ch_symbols % cfg.nof_layers != 0 || cfg.max_iterations == 0) {
    return false;
  }
  // Segmentation is the transmitter's (ldpc_segmenter_rx_impl mirrors ldpc_segmenter_tx): reuse its derivation.
  nrphy_pdsch_pdu_t pdu;
  std::memset(&pdu, 0, sizeof(pdu));
  pdu.ldpc_base_graph = cfg.base_graph;
  pdu.tb_size_bytes   = cfg.tb_size_bytes;
  pdu.rv              = cfg.rv;
  pdu.nof_layers      = cfg.nof_layers;
  pdu.qm              = cfg.qm;
  const uint32_t nref = cfg.nref;
  derive(pdu, cfg.nof_ch_symbols / cfg.nof_layers, l.d, &nref);
  if (l.d.lifting_size == 0 || l.d.nof_codeblocks == 0 || l.d.nof_codeblocks > NRPHY_MAX_CODEBLOCKS ||
      l.d.rm_length_short == 0) {
    return false;
  }
  const uint64_t n_cb = (uint64_t)n_tb * l.d.nof_codeblocks;
  l.msg_stride        = ((l.d.segment_length + 7) / 8 + 8 + 15) & ~15U;
  l.off_ok            = 0;
  l.off_skip          = (n_cb + 63) & ~(uint64_t)63;
  l.off_iter          = 2 * l.off_skip;
  l.off_msg           = (l.off_iter + 4 * n_cb + 63) & ~(uint64_t)63;
  l.state_bytes       = l.off_msg + n_cb * l.msg_stride;
  return true;
}

} // namespace

namespace {
// Weights of the assembly kernel's per-thread transport-block CRC pieces (pusch_decoder.hip): fixed by the block size.
const uint32_t* get_pusch_tb_crc_weights(nrphy_ctx* ctx, uint32_t tb_size_bytes)
{
  std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
  auto                                  it = ctx->d_tb_crc_w.find(tb_size_bytes);
  if (it == ctx->d_tb_crc_w.end()) {
    const uint32_t        piece = divide_ceil(tb_size_bytes, PUSCH_ASSEMBLE_THREADS);
    std::vector<uint32_t> w(PUSCH_ASSEMBLE_THREADS);
    for (uint32_t t = 0; t != PUSCH_ASSEMBLE_THREADS; ++t) {
      const uint32_t end = std::min<uint32_t>(std::min<uint32_t>(t * piece, tb_size_bytes) + piece, tb_size_bytes);
      w[t]               = CRC24A_FIELD.xpow(8 * (int64_t)(tb_size_bytes - end));
    }
    uint32_t* d_w = nullptr;
    if (upload(&d_w, w.data(), w.size() * sizeof(uint32_t)) != hipSuccess) {
      return nullptr;
    }
    it = ctx->d_tb_crc_w.emplace(tb_size_bytes, d_w).first;
  }
  return it->second;
}

nrphy_ldpc_decoder_cfg_t pusch_ldpc_cfg(const nrphy_pusch_decoder_cfg_t& cfg, const nrphy_pdsch_derived_t& d)
{
  // Decoding of the codeblocks whose CRC has not passed yet (pusch_codeblock_decoder.cpp:36-71): CRC24B per codeblock,
  // the transport block's own CRC when it is a single codeblock.
  nrphy_ldpc_decoder_cfg_t dec;
  dec.base_graph      = cfg.base_graph;
  dec.lifting_size    = d.lifting_size;
  dec.nof_filler_bits = d.nof_filler_bits;
  dec.crc_poly        = (d.nof_codeblocks > 1) ? 0x24B : (d.nof_tb_crc_bits == 16 ? 16 : 0x24A);
  dec.nof_llr         = d.full_length;
  dec.max_iterations  = cfg.max_iterations;
  dec.scaling_factor  = 0.8F; // ldpc_decoder::configuration::algorithm_details default, which pusch_codeblock_decoder keeps
  return dec;
}
} // namespace

extern "C" int nrphy_pusch_decoder_prepare(nrphy_ctx_t* ctx, const nrphy_pusch_decoder_cfg_t* cfg)
{
  PuschLayout l;
  if (ctx == nullptr || cfg == nullptr || !pusch_layout(*cfg, 1, l)) {
    return NRPHY_ERR_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  const nrphy_ldpc_decoder_cfg_t dec = pusch_ldpc_cfg(*cfg, l.d);
  const int                      rc  = nrphy_ldpc_decoder_prepare(ctx, &dec);
  if (rc != NRPHY_OK) {
    return rc;
  }
  if (l.d.nof_codeblocks > 1 && get_pusch_tb_crc_weights(ctx, cfg->tb_size_bytes) == nullptr) {
    return NRPHY_ERR_DEVICE;
  }
  return NRPHY_OK;
}

extern "C" int nrphy_pusch_decoder_sizes(nrphy_ctx_t* ctx, const nrphy_pusch_decoder_cfg_t* cfg, uint32_t n_tb,
                                         uint64_t* soft_bytes_per_tb, uint64_t* state_bytes, uint64_t* scratch_bytes,
                                         uint32_t* nof_codeblocks)
{
  PuschLayout l;
  if (ctx == nullptr || cfg == nullptr || !pusch_layout(*cfg, n_tb, l)) {
    return NRPHY_ERR_ARGUMENT;
  }
  if (scratch_bytes) {
    const nrphy_ldpc_decoder_cfg_t dec = pusch_ldpc_cfg(*cfg, l.d);
    if (nrphy_ldpc_decoder_scratch_bytes(ctx, &dec, std::max<uint32_t>(1, n_tb * l.d.nof_codeblocks), scratch_bytes) != NRPHY_OK) {
      return NRPHY_ERR_ARGUMENT;
    }
  }
  if (soft_bytes_per_tb) {
    *soft_bytes_per_tb = (uint64_t)l.d.nof_codeblocks * l.d.full_length;
  }
  if (state_bytes) {
    *state_bytes = l.state_bytes;
  }
  if (nof_codeblocks) {
    *nof_codeblocks = l.d.nof_codeblocks;
  }
  return NRPHY_OK;
}

extern "C" int nrphy_pusch_decode_batch(nrphy_ctx_t* ctx, const nrphy_pusch_decoder_cfg_t* cfg, uint32_t n_tb,
                                        const int8_t* d_llr, uint64_t llr_stride_bytes, int8_t* d_soft, uint8_t* d_state,
                                        void* d_scratch, uint8_t* d_tb, uint32_t tb_stride_bytes, uint32_t* d_result, void* stream)
{
  const TraceRange trace("process_pusch");
  PuschLayout l;
  if (ctx == nullptr || cfg == nullptr || d_llr == nullptr || d_soft == nullptr || d_state == nullptr || d_tb == nullptr ||
      d_scratch == nullptr ||
      d_result == nullptr || !pusch_layout(*cfg, n_tb, l) || tb_stride_bytes < cfg->tb_size_bytes ||
      llr_stride_bytes < (uint64_t)cfg->nof_ch_symbols * cfg->qm) {
    return NRPHY_ERR_ARGUMENT;
  }
  if (n_tb == 0) {
    return NRPHY_OK;
  }
  const nrphy_pdsch_derived_t& d = l.d;
  const uint32_t               C = d.nof_codeblocks, N = d.full_length;
  const uint64_t               n_cb = (uint64_t)n_tb * C;
  hipStream_t                  s    = stream ? (hipStream_t)stream : ctx->stream;
  uint8_t *                    ok = d_state + l.off_ok, *skip = d_state + l.off_skip, *msg = d_state + l.off_msg;
  uint32_t*                    iter = (uint32_t*)(d_state + l.off_iter);
  HIP_TRY(hipSetDevice(ctx->device));
  if (cfg->new_data) { // a fresh soft buffer has no codeblock CRC flags
    HIP_TRY(hipMemsetAsync(ok, 0, n_cb, s));
  }
  HIP_TRY(hipMemcpyAsync(skip, ok, n_cb, hipMemcpyDeviceToDevice, s));
  // Rate dematching of every codeblock, also of those decoded earlier (pusch_decoder_impl.cpp:340-350): the short
  // segments, then the long ones.
  nrphy_ldpc_rate_dematcher_cfg_t dm;
  dm.base_graph      = cfg->base_graph;
  dm.lifting_size    = d.lifting_size;
  dm.rv              = cfg->rv;
  dm.qm              = cfg->qm;
  dm.nref            = d.n_ref;
  dm.nof_filler_bits = d.nof_filler_bits;
  const uint32_t n_short = d.nof_short_segments;
  int            rc      = NRPHY_OK;
  if (n_short != 0) {
    dm.rm_length = d.rm_length_short;
    rc = rate_dematch_batch(ctx, &dm, n_short, n_tb, d_llr, d.rm_length_short, llr_stride_bytes, d_soft, N, (size_t)C * N,
                            cfg->new_data, s);
  }
  if (rc == NRPHY_OK && n_short != C) {
    dm.rm_length = d.rm_length_long;
    rc = rate_dematch_batch(ctx, &dm, C - n_short, n_tb, d_llr + (size_t)n_short * d.rm_length_short, d.rm_length_long,
                            llr_stride_bytes, d_soft + (size_t)n_short * N, N, (size_t)C * N, cfg->new_data, s);
  }
  if (rc != NRPHY_OK) {
    return rc;
  }
  const nrphy_ldpc_decoder_cfg_t dec = pusch_ldpc_cfg(*cfg, d);
  // How far into the soft buffers a first transmission reaches (filler bits included): what the decoder can expect to be in
  // use when the buffers held nothing else -- a hint for the launch's LDS size only (see ldpc_decode_batch).
  uint32_t expected_extent = 0;
  if (cfg->new_data) {
    const unsigned bg_k = (cfg->base_graph == 1) ? 22 : 10, zc = d.lifting_size;
    const unsigned buffer_length = (d.n_ref > 0 && d.n_ref < N) ? d.n_ref : N;
    static const double shift_bg1[4] = {0, 17, 33, 56}, shift_bg2[4] = {0, 13, 25, 43};
    const double   frac = (((cfg->base_graph == 1) ? shift_bg1 : shift_bg2)[cfg->rv] * buffer_length) / N;
    const unsigned k0   = (unsigned)((uint16_t)std::floor(frac)) * zc;
    std::vector<DematchOp> ops;
    for (uint32_t e : {d.rm_length_short, d.rm_length_long}) {
      if (e == 0) {
        continue;
      }
      build_dematch_ops(ops, N, buffer_length, k0, (bg_k - 2) * zc, d.nof_filler_bits, e, true);
      for (const DematchOp& op : ops) {
        if (op.kind != DEMATCH_ZERO) {
          expected_extent = std::max<uint32_t>(expected_extent, op.begin + op.count);
        }
      }
    }
  }
  rc = ldpc_decode_batch(ctx, &dec, (uint32_t)n_cb, d_soft, N, msg, l.msg_stride, iter, skip, ok, cfg->use_early_stop == 0,
                         d_scratch, s, expected_extent);
  if (rc != NRPHY_OK) {
    return rc;
  }
  PuschAssembleLaunch a;
  a.crc_weight = nullptr;
  if (C > 1) {
    a.crc_weight = get_pusch_tb_crc_weights(ctx, cfg->tb_size_bytes); // uploaded by nrphy_pusch_decoder_prepare() or here
    if (a.crc_weight == nullptr) {
      return NRPHY_ERR_DEVICE;
    }
  }
  a.cb_msg         = msg;
  a.cb_ok          = ok;
  a.cb_iter        = iter;
  a.skipped        = skip;
  a.tb             = d_tb;
  a.result         = d_result;
  a.C              = C;
  a.msg_stride     = l.msg_stride;
  a.tb_stride      = tb_stride_bytes;
  a.tb_bytes       = cfg->tb_size_bytes;
  a.cb_info_bits   = d.cb_info_bits;
  a.max_iterations = cfg->max_iterations;
  // The transport-block check by 16 KiB regions (tbcrc_regions_workgroup, the transmit side's TB-CRC role): the context's
  // tables and the factor that turns the remainder of the zero-extended block into the block's CRC24A.
  a.tbcrc          = ctx->d_tbcrc;
  a.crc_factor     = CRC24A_FIELD.xpow((int64_t)CRC24A_FIELD.order +
                                   8 * ((int64_t)cfg->tb_size_bytes -
                                        (int64_t)divide_ceil(cfg->tb_size_bytes, TB_CRC_REGION_BYTES) * TB_CRC_REGION_BYTES));
  HIP_TRY(launch_pusch_assemble(a, n_tb, s));
  return NRPHY_OK;
}

// ================================================================================================================
// NZP-CSI-RS
// ================================================================================================================
extern "C" int nrphy_csi_rs_validate(const nrphy_csi_rs_cfg_t* c)
{
  static const unsigned row_ports[6] = {0, 1, 1, 2, 4, 4};
  if (c == nullptr || c->row < 1 || c->row > 5 || c->nof_k_ref != 1 || c->cp > 1 || c->nof_rb == 0 ||
      c->nof_ports != row_ports[c->row] || c->precoding == nullptr || c->nof_prg != 1 || c->prg_size_rb == 0 || c->prg_size_rb > NRPHY_MAX_RB ||
      c->start_rb + c->nof_rb > NRPHY_MAX_RB) {
    return NRPHY_ERR_ARGUMENT;
  }
  // The assertions of mapping_row_1 .. mapping_row_5 (csi_rs_pattern.cpp:34-158); densities as csi_rs_freq_density_type.
  const unsigned nsymb = c->cp ? 12 : 14, k0 = c->k_ref[0];
  const bool     three = c->density == 3, one = c->density == 2, valid_density = c->density <= 3;
  bool           ok    = false;
  switch (c->row) {
    case 1:
      ok = k0 <= 3 && three && c->cdm == 0 && c->symbol_l0 < nsymb;
      break;
    case 2:
      ok = k0 < 12 && valid_density && !three && c->cdm == 0 && c->symbol_l0 < nsymb;
      break;
    case 3:
      ok = k0 < 11 && valid_density && !three && c->cdm == 1 && c->symbol_l0 < nsymb;
      break;
    case 4:
      ok = k0 < 9 && one && c->cdm == 1 && c->symbol_l0 < nsymb;
      break;
    default:
      ok = k0 < 11 && one && c->cdm == 1 && c->symbol_l0 + 1 < nsymb;
      break;
  }
  return ok ? NRPHY_OK : NRPHY_ERR_ARGUMENT;
}

extern "C" int nrphy_csi_rs_map(nrphy_ctx_t* ctx, uint32_t n, const nrphy_csi_rs_cfg_t* cfgs, const uint32_t* grid_index,
                                void* d_grid, uint32_t grid_nof_ports, uint32_t grid_nof_subc, void* stream)
{
  const TraceRange trace("process_nzp_csi_rs");
  if (ctx == nullptr || (n != 0 && (cfgs == nullptr || d_grid == nullptr))) {
    return NRPHY_ERR_ARGUMENT;
  }
  std::vector<CsiRsWork> work;
  std::vector<float>     weights;
  for (uint32_t i = 0; i != n; ++i) {
    const nrphy_csi_rs_cfg_t& c = cfgs[i];
    if (nrphy_csi_rs_validate(&c) != NRPHY_OK || c.nof_ports > grid_nof_ports || 12 * (c.start_rb + c.nof_rb) > grid_nof_subc) {
      return NRPHY_ERR_ARGUMENT;
    }
    const unsigned group_size = c.cdm ? 2 : 1, nof_groups = c.nof_ports / group_size;
    const bool     half       = c.density <= 1, even = c.density == 0;
    // PRB range and stride (build_re_patterns, csi_rs_pattern.cpp:374-392)
    unsigned rb_begin = c.start_rb, rb_stride = half ? 2 : 1;
    if (half && (((c.start_rb % 2) != 0) == even)) {
      ++rb_begin;
    }
    // sequence elements below the first occupied PRB and per symbol (nzp_csi_rs_generator_impl.cpp:66-159)
    unsigned first_prb = c.start_rb;
    if (half) {
      first_prb = even ? c.start_rb + c.start_rb % 2 : c.start_rb + (1 - c.start_rb % 2);
    }
    unsigned advance = 0;
    if (c.density == 3) {
      advance = 3 * first_prb;
    } else if (c.density == 2) {
      advance = (c.row == 2) ? first_prb : 2 * first_prb;
    } else {
      advance = (c.row == 2) ? first_prb / 2 : first_prb;
    }
    unsigned seq_len = c.nof_rb;
    if (half) {
      seq_len /= 2;
      if (c.nof_rb % 2 != 0 && (((c.start_rb % 2) != 0) == !even)) {
        ++seq_len;
      }
    } else if (c.density == 3) {
      seq_len *= 3;
    }
    seq_len *= c.cdm ? 2 : 1;
    const unsigned nsymb = c.cp ? 12 : 14;
    for (unsigned g = 0; g != nof_groups; ++g) {
      CsiRsWork w;
      unsigned  k_bar = c.k_ref[0], l_bar = c.symbol_l0;
      if (c.row == 4) {
        k_bar += 2 * g;
      } else if (c.row == 5) {
        l_bar += g;
      }
      w.re_mask    = (c.row == 1) ? ((1U << k_bar) | (1U << (k_bar + 4)) | (1U << (k_bar + 8))) : ((c.cdm ? 3U : 1U) << k_bar);
      w.n_re_prb   = (uint32_t)__builtin_popcount(w.re_mask);
      w.grid_index = grid_index ? grid_index[i] : 0;
      w.symbol     = l_bar;
      w.c_init     = (uint32_t)((1024ULL * (nsymb * c.slot_index + l_bar + 1) * (2 * c.scrambling_id + 1) + c.scrambling_id) & 0x7FFFFFFFULL);
      w.advance    = advance;
      w.seq_len    = seq_len;
      w.rb_begin   = rb_begin;
      w.rb_stride  = rb_stride;
      w.nof_ports  = c.nof_ports;
      w.first_layer = g * group_size;
      w.group_size  = group_size;
      w.weights_offset = (uint32_t)weights.size();
      w.amplitude   = (float)(M_SQRT1_2 * (double)c.amplitude);
      if (2 * (advance + seq_len) + 32 > 32 * CSI_RS_MAX_SEQ_WORDS) {
        return NRPHY_ERR_CAPACITY;
      }
      work.push_back(w);
    }
    weights.insert(weights.end(), c.precoding, c.precoding + 2 * (size_t)c.nof_ports * c.nof_ports);
  }
  if (work.empty()) {
    return NRPHY_OK;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  // Work list and weights go through a buffer of this call's own (stream-ordered allocation).
  const size_t  work_bytes = work.size() * sizeof(CsiRsWork), off_w = (work_bytes + 63) & ~(size_t)63;
  StreamStaging staging(s);
  uint8_t*      base = (uint8_t*)staging.alloc(off_w + weights.size() * sizeof(float));
  if (base == nullptr) {
    return NRPHY_ERR_DEVICE;
  }
  // pageable host memory: hipMemcpyAsync stages it before returning, the vectors may go out of scope
  HIP_TRY(hipMemcpyAsync(base, work.data(), work_bytes, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(base + off_w, weights.data(), weights.size() * sizeof(float), hipMemcpyHostToDevice, s));
  CsiRsLaunch p;
  p.work           = (const CsiRsWork*)base;
  p.weights        = (const float*)(base + off_w);
  p.gold           = ctx->d_gold;
  p.x1_words       = ctx->d_x1;
  p.grid           = (uint32_t*)d_grid;
  p.grid_nof_ports = grid_nof_ports;
  p.grid_nof_subc  = grid_nof_subc;
  HIP_TRY(launch_csi_rs(p, (uint32_t)work.size(), s));
  return NRPHY_OK;
}

extern "C" int nrphy_csi_rs_map_host(nrphy_ctx_t* ctx, const nrphy_csi_rs_cfg_t* cfg, void* grid, uint32_t nof_ports,
                                     uint32_t nof_subc)
{
  if (ctx == nullptr || cfg == nullptr || grid == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  std::lock_guard<std::recursive_mutex> host_lock(ctx->host_mutex);
  const size_t bytes  = (size_t)nof_ports * NRPHY_NSYMB * nof_subc * 4;
  void*        d_grid = nullptr;
  HIP_TRY(hipSetDevice(ctx->device));
  {
    std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
    d_grid = ctx_scratch(ctx, SCRATCH_GRID, bytes);
  }
  if (d_grid == nullptr) {
    return NRPHY_ERR_DEVICE;
  }
  HIP_TRY(hipMemcpyAsync(d_grid, grid, bytes, hipMemcpyHostToDevice, ctx->stream));
  const int rc = nrphy_csi_rs_map(ctx, 1, cfg, nullptr, d_grid, nof_ports, nof_subc, ctx->stream);
  if (rc != NRPHY_OK) {
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
  }
  HIP_TRY(hipMemcpyAsync(grid, d_grid, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return NRPHY_OK;
}

extern "C" int nrphy_grid_put(nrphy_ctx_t* ctx, void* d_grid, uint32_t nof_ports, uint32_t nof_subc, uint32_t n,
                              const nrphy_grid_re_t* entries, void* stream)
{
  if (ctx == nullptr || d_grid == nullptr || (n != 0 && entries == nullptr) || nof_ports > NRPHY_MAX_PORTS ||
      nof_subc > NRPHY_MAX_RB * 12) { // (also keeps the 32-bit element index below from wrapping)
    return NRPHY_ERR_ARGUMENT;
  }
  if (n == 0) {
    return NRPHY_OK;
  }
  std::vector<uint32_t> packed(2 * (size_t)n);
  for (uint32_t i = 0; i != n; ++i) {
    const nrphy_grid_re_t& e = entries[i];
    if (e.port >= nof_ports || e.symbol >= NRPHY_NSYMB || e.subc >= nof_subc) {
      return NRPHY_ERR_ARGUMENT;
    }
    packed[i]     = ((uint32_t)e.port * NRPHY_NSYMB + e.symbol) * nof_subc + e.subc;
    packed[n + i] = e.value;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s    = stream ? (hipStream_t)stream : ctx->stream;
  StreamStaging staging(s);
  uint32_t*     base = (uint32_t*)staging.alloc(packed.size() * sizeof(uint32_t));
  if (base == nullptr) {
    return NRPHY_ERR_DEVICE;
  }
  HIP_TRY(hipMemcpyAsync(base, packed.data(), packed.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
  HIP_TRY(launch_grid_put(base, base + n, n, (uint32_t*)d_grid, s));
  return NRPHY_OK;
}

extern "C" int nrphy_ldpc_encode(nrphy_ctx_t* ctx, uint32_t base_graph, uint32_t lifting_size, uint32_t n_cb,
                                 const uint8_t* d_msg, uint32_t msg_stride_bytes, uint32_t out_bits, uint8_t* d_out,
                                 uint32_t out_stride_bytes, void* stream)
{
  if (ctx == nullptr || (base_graph != 1 && base_graph != 2) || d_msg == nullptr || d_out == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  const int pos = lifting_position(lifting_size);
  const unsigned kb = (base_graph == 1) ? 22 : 10, nfull = (base_graph == 1) ? 68 : 52;
  if (pos < 0 || out_bits == 0 || out_bits > (nfull - 2) * lifting_size ||
      msg_stride_bytes < (kb * lifting_size + 7) / 8 || out_stride_bytes < (out_bits + 7) / 8) {
    return NRPHY_ERR_ARGUMENT;
  }
  HIP_TRY(launch_ldpc_encode(ctx->d_graphs, (base_graph - 1) * NOF_LIFTING_SIZES + (uint32_t)pos, kb, lifting_size,
                             n_cb, d_msg, msg_stride_bytes, out_bits, d_out, out_stride_bytes,
                             stream ? (hipStream_t)stream : ctx->stream));
  return NRPHY_OK;
}

// ================================================================================================================
// OFDM
// ================================================================================================================
extern "C" int nrphy_ofdm_plan_create(nrphy_ctx_t* ctx, const nrphy_ofdm_config_t* cfg, uint32_t nof_ports,
                                      nrphy_ofdm_plan_t** out)
{
  if (ctx == nullptr || cfg == nullptr || out == nullptr || nof_ports == 0 || cfg->numerology > 4 ||
      !dft_size_in_lds(cfg->dft_size) || cfg->dft_size <= 12 * cfg->bw_rb || cfg->bw_rb == 0 || cfg->cp > 1 ||
      !std::isnormal(cfg->scale)) {
    return NRPHY_ERR_ARGUMENT;
  }
  *out = nullptr;
  HIP_TRY(hipSetDevice(ctx->device));
  if (get_twiddle(ctx, cfg->dft_size) == nullptr) {
    return NRPHY_ERR_DEVICE;
  }
  nrphy_ofdm_plan* plan = new (std::nothrow) nrphy_ofdm_plan;
  if (plan == nullptr) {
    return NRPHY_ERR_CAPACITY;
  }
  plan->ctx           = ctx;
  plan->cfg           = *cfg;
  plan->d_twiddle     = get_twiddle(ctx, cfg->dft_size);
  plan->nof_ports     = nof_ports;
  plan->nsymb         = cfg->cp ? 12 : 14; // get_nsymb_per_slot (cyclic_prefix.h:108-114)
  plan->nsym_subframe = plan->nsymb << cfg->numerology;
  plan->slot_stride   = nrphy_ofdm_slot_size(cfg, 0);
  // Phase compensation (TS 38.211 Section 5.4; phase_compensation_lut.h:49-82) times the scale.
  const double        srate = 15000.0 * (double)(1U << cfg->numerology) * (double)cfg->dft_size;
  std::vector<float2> phase(plan->nsym_subframe), phase_rx(plan->nsym_subframe);
  plan->cp.resize(plan->nsym_subframe);
  plan->off.resize(plan->nsym_subframe);
  unsigned offset = 0, in_slot = 0;
  for (unsigned s = 0; s != plan->nsym_subframe; ++s) {
    if (s % plan->nsymb == 0) {
      in_slot = 0;
    }
    const unsigned cp = cp_length(*cfg, s);
    offset += cp;
    const double ph = -2.0 * M_PI * cfg->center_freq_hz * ((double)offset / srate);
    const float  pr = (float)std::cos(ph), pi = (float)std::sin(ph);
    phase[s]        = make_float2(pr * cfg->scale, pi * cfg->scale);
    phase_rx[s]     = make_float2(pr * cfg->scale, -pi * cfg->scale);
    plan->cp[s]     = cp;
    plan->off[s]    = in_slot;
    in_slot += cp + cfg->dft_size;
    offset += cfg->dft_size;
  }
  if (upload(&plan->d_phase, phase.data(), phase.size() * sizeof(float2)) != hipSuccess ||
      upload(&plan->d_phase_rx, phase_rx.data(), phase_rx.size() * sizeof(float2)) != hipSuccess ||
      upload(&plan->d_cp, plan->cp.data(), plan->cp.size() * sizeof(uint32_t)) != hipSuccess ||
      upload(&plan->d_off, plan->off.data(), plan->off.size() * sizeof(uint32_t)) != hipSuccess) {
    nrphy_ofdm_plan_destroy(plan);
    return NRPHY_ERR_DEVICE;
  }
  *out = plan;
  return NRPHY_OK;
}

extern "C" int nrphy_ofdm_plan_destroy(nrphy_ofdm_plan_t* plan)
{
  if (plan == nullptr) {
    return NRPHY_OK;
  }
  (void)hipSetDevice(plan->ctx->device);
  (void)hipFree(plan->d_phase);
  (void)hipFree(plan->d_phase_rx);
  (void)hipFree(plan->d_cp);
  (void)hipFree(plan->d_off);
  (void)hipFree(plan->d_wire_partials);
  for (auto& kv : plan->d_window_phase) {
    (void)hipFree(kv.second);
  }
  for (hipEvent_t e : plan->events) {
    (void)hipEventDestroy(e);
  }
  delete plan;
  return NRPHY_OK;
}

extern "C" uint32_t nrphy_ofdm_plan_slot_stride(const nrphy_ofdm_plan_t* plan)
{
  return plan ? plan->slot_stride : 0;
}

namespace {

// Amplitude controller parameters as the reference's constructors derive them
// (amplitude_controller_clipping_impl.h:52-62, amplitude_controller_scaling_impl.h).
struct AmplitudeParams {
  float gain, ceiling;
  bool  measure, clip;
};
bool amplitude_params(const nrphy_amplitude_cfg_t& c, AmplitudeParams& a)
{
  if (c.kind > 1) {
    return false;
  }
  a.gain    = std::pow(10.0F, c.input_gain_dB / 20.0F);
  a.ceiling = c.full_scale_lin * std::pow(10.0F, c.ceiling_dBFS / 20.0F);
  a.measure = c.kind == 0;
  a.clip    = c.kind == 0 && c.enable_clipping != 0;
  // a ceiling that is not a normal number cannot be told from the reference's zero-power exception (see the header)
  return std::isfinite(a.gain) && (!a.clip || std::isnormal(a.ceiling));
}

int ofdm_run(nrphy_ofdm_plan_t* plan, uint32_t nof_grids, const void* d_grid, const uint32_t* slot_index, void* d_iq,
             const nrphy_iq_wire_cfg_t* wire, nrphy_amplitude_stats_t* d_stats, void* stream)
{
  if (plan == nullptr || d_grid == nullptr || d_iq == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  nrphy_ctx* ctx = plan->ctx;
  OfdmLaunch p;
  p.window_phase = nullptr;
  p.dft_size    = plan->cfg.dft_size;
  p.rg_size     = 12 * plan->cfg.bw_rb;
  p.nof_ports   = plan->nof_ports;
  p.nsymb       = plan->nsymb;
  p.slot_stride = plan->slot_stride;
  p.twiddle     = plan->d_twiddle;
  p.phase       = plan->d_phase;
  p.cp_len      = plan->d_cp;
  p.sym_offset  = plan->d_off;
#ifdef NRPHY_PROBES
  p.probe = ctx->tune.ofdm_probe; // profiling variant: 1 = drop the IQ stores, 2 = drop the grid loads, 3 = both (outputs are then wrong)
#else
  p.probe = 0;
#endif
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  if (wire != nullptr) {
    AmplitudeParams a;
    if (!amplitude_params(wire->amplitude, a) || !std::isfinite(wire->ci16_scale)) {
      return NRPHY_ERR_ARGUMENT;
    }
    p.wire         = 1;
    p.wire_clip    = a.clip ? 1U : 0U;
    p.wire_gain    = a.gain;
    p.wire_ceiling = a.ceiling;
    p.wire_scale   = wire->ci16_scale;
    // Largest magnitude whose scaled value stays within int16 (so the saturation is an identity below it).
    float sat = wire->ci16_scale != 0.f ? 32767.0f / std::fabs(wire->ci16_scale) : INFINITY;
    while ((double)sat * std::fabs((double)wire->ci16_scale) > 32767.0) {
      sat = std::nextafterf(sat, 0.f);
    }
    // ... squared, and a hair lower: the kernel compares the rounded power re^2 + im^2 of a sample with it
    const float lim = a.clip ? std::fmin(a.ceiling, sat) : sat;
    p.wire_limit    = std::isfinite(lim) ? (float)((double)lim * (double)lim * (1.0 - 1e-6)) : lim;
    p.wire_stats   = a.measure ? d_stats : nullptr;
    if (p.wire_stats != nullptr) {
      // One record per workgroup (at most one workgroup per symbol), added up by a second small kernel.  The buffer belongs to
      // the plan and grows with the largest batch seen (a synchronous reallocation, on growth only): runs of one plan are
      // ordered, as the conventions in mi355_nrphy.h say.
      const size_t need = (size_t)nof_grids * plan->nof_ports * plan->nsymb;
      if (need > plan->wire_partials_cap) {
        if (plan->d_wire_partials != nullptr) {
          HIP_TRY(hipFree(plan->d_wire_partials));
          plan->d_wire_partials   = nullptr;
          plan->wire_partials_cap = 0;
        }
        HIP_TRY(hipMalloc(&plan->d_wire_partials, need * sizeof(uint4)));
        plan->wire_partials_cap = need;
      }
      p.wire_partials = plan->d_wire_partials;
    } else if (d_stats != nullptr) {
      HIP_TRY(hipMemsetAsync(d_stats, 0, sizeof(nrphy_amplitude_stats_t) * (size_t)nof_grids * plan->nof_ports, s));
    }
  }
  hipEvent_t* ev = nullptr;
  if (plan->timed_runs < plan->max_timed_runs && plan->timing_counter++ % plan->timing_stride == 0) {
    ev = &plan->events[2 * plan->timed_runs++];
    HIP_TRY(hipEventRecord(ev[0], s));
  }
  HIP_TRY(launch_ofdm(p, nof_grids, (const uint32_t*)d_grid, slot_index, (float2*)d_iq, s));
  if (ev) {
    HIP_TRY(hipEventRecord(ev[1], s));
  }
  return NRPHY_OK;
}

} // namespace

extern "C" int nrphy_ofdm_run(nrphy_ofdm_plan_t* plan, uint32_t nof_grids, const void* d_grid,
                              const uint32_t* slot_index, float* d_iq, void* stream)
{
  const TraceRange trace("downlink_baseband");
  return ofdm_run(plan, nof_grids, d_grid, slot_index, d_iq, nullptr, nullptr, stream);
}

extern "C" int nrphy_ofdm_run_ci16(nrphy_ofdm_plan_t* plan, uint32_t nof_grids, const void* d_grid, const uint32_t* slot_index,
                                   const nrphy_iq_wire_cfg_t* cfg, int16_t* d_iq, nrphy_amplitude_stats_t* d_stats, void* stream)
{
  if (cfg == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  const TraceRange trace("downlink_baseband");
  return ofdm_run(plan, nof_grids, d_grid, slot_index, d_iq, cfg, d_stats, stream);
}

// ================================================================================================================
// Lower-PHY tail: amplitude controller, radio sample format, fronthaul compression
// ================================================================================================================
extern "C" int nrphy_amplitude_control(nrphy_ctx_t* ctx, const nrphy_amplitude_cfg_t* cfg, uint32_t n_buffers, uint32_t nof_samples,
                                       const float* d_in, size_t in_stride, float* d_out, size_t out_stride,
                                       nrphy_amplitude_stats_t* d_stats, void* stream)
{
  AmplitudeParams a;
  if (ctx == nullptr || cfg == nullptr || d_in == nullptr || d_out == nullptr || !amplitude_params(*cfg, a) ||
      (n_buffers > 1 && (in_stride < nof_samples || out_stride < nof_samples))) {
    return NRPHY_ERR_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  if (d_stats != nullptr) {
    HIP_TRY(hipMemsetAsync(d_stats, 0, sizeof(nrphy_amplitude_stats_t) * (size_t)n_buffers, s));
  }
  for (uint32_t first = 0; first < n_buffers; first += 32768) { // the launch's second grid dimension holds 65535
    AmplitudeLaunch p;
    p.in          = d_in + 2 * first * in_stride;
    p.out         = d_out + 2 * first * out_stride;
    p.in_stride   = in_stride;
    p.out_stride  = out_stride;
    p.nof_samples = nof_samples;
    p.measure     = a.measure ? 1U : 0U;
    p.clip        = a.clip ? 1U : 0U;
    p.gain        = a.gain;
    p.ceiling     = a.ceiling;
    p.stats       = d_stats ? d_stats + first : nullptr;
    HIP_TRY(launch_amplitude(p, std::min<uint32_t>(32768, n_buffers - first), s));
  }
  return NRPHY_OK;
}

extern "C" int nrphy_amplitude_metrics(const nrphy_amplitude_cfg_t* cfg, const nrphy_amplitude_stats_t* stats,
                                       nrphy_amplitude_metrics_t* m)
{
  if (cfg == nullptr || stats == nullptr || m == nullptr || cfg->kind > 1) {
    return NRPHY_ERR_ARGUMENT;
  }
  if (cfg->kind == 1) { // amplitude_controller_scaling_impl returns empty metrics
    std::memset(m, 0, sizeof(*m));
    return NRPHY_OK;
  }
  // amplitude_controller_clipping_impl::process, the part after the vector operations
  const float full_scale_pwr = cfg->full_scale_lin * cfg->full_scale_lin;
  const float avg            = stats->nof_samples ? stats->sum_power / (float)stats->nof_samples : 0.0F;
  m->clipping_enabled        = cfg->enable_clipping ? 1U : 0U;
  m->gain_dB                 = 20.0F * std::log10(std::pow(10.0F, cfg->input_gain_dB / 20.0F)); // convert_amplitude_to_dB
  m->avg_power_fs            = avg / full_scale_pwr;
  m->peak_power_fs           = stats->peak_power / full_scale_pwr;
  if (!std::isnormal(avg) || !std::isnormal(stats->peak_power)) {
    m->papr_lin = 1.0F;
    return NRPHY_OK;
  }
  m->papr_lin = stats->peak_power / avg;
  if (cfg->enable_clipping) {
    m->nof_processed_samples += stats->nof_samples;
    m->nof_clipped_samples += stats->nof_clipped;
    m->clipping_probability = (double)m->nof_clipped_samples / (double)m->nof_processed_samples;
  }
  return NRPHY_OK;
}

extern "C" int nrphy_amplitude_control_host(nrphy_ctx_t* ctx, const nrphy_amplitude_cfg_t* cfg, const float* in,
                                            uint32_t nof_samples, float* out, nrphy_amplitude_metrics_t* metrics)
{
  if (ctx == nullptr || cfg == nullptr || in == nullptr || out == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)nof_samples * sizeof(float2);
  float*       d_buf = (float*)ctx_scratch(ctx, SCRATCH_IQ, bytes + 64);
  if (d_buf == nullptr) {
    return NRPHY_ERR_DEVICE;
  }
  nrphy_amplitude_stats_t* d_stats = (nrphy_amplitude_stats_t*)((uint8_t*)d_buf + ((bytes + 15) & ~(size_t)15));
  HIP_TRY(hipMemcpyAsync(d_buf, in, bytes, hipMemcpyHostToDevice, ctx->stream));
  const int rc = nrphy_amplitude_control(ctx, cfg, 1, nof_samples, d_buf, nof_samples, d_buf, nof_samples, d_stats, ctx->stream);
  if (rc != NRPHY_OK) {
    return rc;
  }
  nrphy_amplitude_stats_t st;
  HIP_TRY(hipMemcpyAsync(out, d_buf, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(&st, d_stats, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return metrics ? nrphy_amplitude_metrics(cfg, &st, metrics) : NRPHY_OK;
}

extern "C" int nrphy_iq_convert_ci16(nrphy_ctx_t* ctx, uint32_t n_buffers, uint32_t nof_samples, const float* d_in, size_t in_stride,
                                     float scale, int16_t* d_out, size_t out_stride, void* stream)
{
  if (ctx == nullptr || d_in == nullptr || d_out == nullptr || !std::isfinite(scale) ||
      (n_buffers > 1 && (in_stride < nof_samples || out_stride < nof_samples))) {
    return NRPHY_ERR_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  for (uint32_t first = 0; first < n_buffers; first += 32768) {
    HIP_TRY(launch_convert_ci16(d_in + 2 * first * in_stride, in_stride, d_out + 2 * first * out_stride, out_stride,
                                std::min<uint32_t>(32768, n_buffers - first), nof_samples, scale, s));
  }
  return NRPHY_OK;
}

extern "C" int nrphy_iq_convert_ci16_host(nrphy_ctx_t* ctx, const float* in, uint32_t nof_samples, float scale, int16_t* out)
{
  if (ctx == nullptr || in == nullptr || out == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)nof_samples * sizeof(float2), obytes = (size_t)nof_samples * 4;
  uint8_t*     d_buf = (uint8_t*)ctx_scratch(ctx, SCRATCH_IQ, bytes + obytes + 64);
  if (d_buf == nullptr) {
    return NRPHY_ERR_DEVICE;
  }
  int16_t* d_out = (int16_t*)(d_buf + ((bytes + 15) & ~(size_t)15));
  HIP_TRY(hipMemcpyAsync(d_buf, in, bytes, hipMemcpyHostToDevice, ctx->stream));
  const int rc = nrphy_iq_convert_ci16(ctx, 1, nof_samples, (const float*)d_buf, nof_samples, scale, d_out, nof_samples, ctx->stream);
  if (rc != NRPHY_OK) {
    return rc;
  }
  HIP_TRY(hipMemcpyAsync(out, d_out, obytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return NRPHY_OK;
}

extern "C" uint32_t nrphy_ofh_compressed_prb_bytes(const nrphy_ofh_compression_cfg_t* cfg)
{
  return cfg ? 3U * cfg->data_width + (cfg->type == 1 ? 1U : 0U) : 0U;
}

extern "C" int nrphy_ofh_compress(nrphy_ctx_t* ctx, const nrphy_ofh_compression_cfg_t* cfg, uint32_t n_rows, uint32_t nof_prb,
                                  const void* d_prbs, size_t row_stride, uint8_t* d_out, size_t out_row_stride, void* stream)
{
  // Below 8 bits the reference's packer has no defined result (it hands bit_buffer::insert values wider than the
  // field, compressed_prb_packer.cpp:39-47).
  if (ctx == nullptr || cfg == nullptr || d_prbs == nullptr || d_out == nullptr || cfg->type > 1 || cfg->data_width < 8 ||
      cfg->data_width > 16 || !std::isfinite(cfg->iq_scaling) || nof_prb > NRPHY_MAX_RB ||
      (n_rows > 1 && (row_stride < 12 * (size_t)nof_prb || out_row_stride < (size_t)nof_prb * nrphy_ofh_compressed_prb_bytes(cfg)))) {
    return NRPHY_ERR_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  for (uint32_t first = 0; first < n_rows; first += 32768) {
    OfhCompressLaunch p;
    p.prbs           = (const uint32_t*)d_prbs + first * row_stride;
    p.out            = d_out + first * out_row_stride;
    p.row_stride     = row_stride;
    p.out_row_stride = out_row_stride;
    p.nof_prb        = nof_prb;
    p.data_width     = cfg->data_width;
    p.bfp            = cfg->type == 1 ? 1U : 0U;
    // the AVX2 compressors convert the whole call at once for BFP and for the widths their packer has (9, 16)
    p.whole_span = (cfg->type == 1 || cfg->data_width == 9 || cfg->data_width == 16) ? 1U : 0U;
    // quantizer: gain = 2^(width - 1) - 1, 16 bits for BFP (Q_BIT_WIDTH); scale = gain * iq_scaling in float
    const float gain = (float)((1 << ((cfg->type == 1 ? 16 : (int)cfg->data_width) - 1)) - 1.0F);
    p.scale          = gain * cfg->iq_scaling;
    HIP_TRY(launch_ofh_compress(p, std::min<uint32_t>(32768, n_rows - first), s));
  }
  return NRPHY_OK;
}

extern "C" int nrphy_ofh_compress_host(nrphy_ctx_t* ctx, const nrphy_ofh_compression_cfg_t* cfg, uint32_t nof_prb, const void* prbs,
                                       uint8_t* out)
{
  if (ctx == nullptr || cfg == nullptr || prbs == nullptr || out == nullptr || nof_prb == 0 || nof_prb > NRPHY_MAX_RB) {
    return NRPHY_ERR_ARGUMENT;
  }
  std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t in_bytes = (size_t)nof_prb * 48, out_bytes = (size_t)nof_prb * 49;
  uint8_t*     d_buf    = (uint8_t*)ctx_scratch(ctx, SCRATCH_GRID, in_bytes + out_bytes + 64);
  if (d_buf == nullptr) {
    return NRPHY_ERR_DEVICE;
  }
  uint8_t* d_out = d_buf + ((in_bytes + 15) & ~(size_t)15);
  HIP_TRY(hipMemcpyAsync(d_buf, prbs, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  const int rc = nrphy_ofh_compress(ctx, cfg, 1, nof_prb, d_buf, 12 * (size_t)nof_prb, d_out, out_bytes, ctx->stream);
  if (rc != NRPHY_OK) {
    return rc;
  }
  HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)nof_prb * nrphy_ofh_compressed_prb_bytes(cfg), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return NRPHY_OK;
}

extern "C" int nrphy_ofdm_demod_run(nrphy_ofdm_plan_t* plan, uint32_t nof_grids, const float* d_iq,
                                    const uint32_t* slot_index, uint32_t window_offset, void* d_grid, void* stream)
{
  // ofdm_demodulator_impl.cpp:58-63: the window offset must stay inside the shortest cyclic prefix.
  if (plan == nullptr || d_grid == nullptr || d_iq == nullptr ||
      window_offset >= (144U * plan->cfg.dft_size) / 2048U) {
    return NRPHY_ERR_ARGUMENT;
  }
  nrphy_ctx* ctx = plan->ctx;
  OfdmLaunch p;
  p.window_phase = nullptr;
  p.dft_size    = plan->cfg.dft_size;
  p.rg_size     = 12 * plan->cfg.bw_rb;
  p.nof_ports   = plan->nof_ports;
  p.nsymb       = plan->nsymb;
  p.slot_stride = plan->slot_stride;
  p.twiddle     = plan->d_twiddle;
  p.phase       = plan->d_phase_rx;
  p.cp_len      = plan->d_cp;
  p.sym_offset  = plan->d_off;
  p.probe       = 0;
  p.window_phase = nullptr;
  if (window_offset != 0) {
    // The reference rotates bin i by std::polar(1.0F, omega * i) with omega = offset * 2 pi / N rounded to float
    // (ofdm_demodulator_impl.cpp:68-76): omega * i reaches hundreds of radians, so its float rounding shows in the
    // output (1e-5 of the largest bin); the table is built here with the same arithmetic instead of exactly.
    std::lock_guard<std::mutex> lock(plan->window_mutex);
    auto                        it = plan->d_window_phase.find(window_offset);
    if (it == plan->d_window_phase.end()) {
      const unsigned      n = plan->cfg.dft_size;
      std::vector<float2> w(n);
      const float         omega = static_cast<float>(window_offset) * static_cast<float>(2.0 * M_PI) / static_cast<float>(n);
      for (unsigned i = 0; i != n; ++i) {
        const std::complex<float> v = std::polar(1.0F, omega * static_cast<float>(i));
        w[i]                        = make_float2(v.real(), v.imag());
      }
      float2* d_w = nullptr;
      HIP_TRY(hipSetDevice(ctx->device));
      if (upload(&d_w, w.data(), w.size() * sizeof(float2)) != hipSuccess) {
        return NRPHY_ERR_DEVICE;
      }
      it = plan->d_window_phase.emplace(window_offset, d_w).first;
    }
    p.window_phase = it->second;
  }
  HIP_TRY(launch_ofdm_demod(p, nof_grids, (const float2*)d_iq, slot_index, window_offset, (uint32_t*)d_grid,
                            stream ? (hipStream_t)stream : ctx->stream));
  return NRPHY_OK;
}

extern "C" int nrphy_ofdm_demodulate_slot_host(nrphy_ofdm_plan_t* plan, const float* iq, uint32_t slot_index,
                                               uint32_t window_offset, void* grid)
{
  if (plan == nullptr || iq == nullptr || grid == nullptr || slot_index >= (1U << plan->cfg.numerology)) {
    return NRPHY_ERR_ARGUMENT;
  }
  nrphy_ctx*   ctx        = plan->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
  const size_t grid_words = (size_t)plan->nof_ports * NRPHY_NSYMB * 12 * plan->cfg.bw_rb;
  const size_t slot_size  = nrphy_ofdm_slot_size(&plan->cfg, slot_index);
  uint32_t *   d_grid = nullptr, *d_slot = nullptr;
  float2*      d_iq   = nullptr;
  int          rc     = NRPHY_ERR_DEVICE;
  do {
    if (hipSetDevice(ctx->device) != hipSuccess || (d_grid = (uint32_t*)ctx_scratch(ctx, SCRATCH_GRID, grid_words * 4)) == nullptr ||
        (d_iq = (float2*)ctx_scratch(ctx, SCRATCH_IQ, (size_t)plan->nof_ports * plan->slot_stride * sizeof(float2))) == nullptr ||
        (d_slot = (uint32_t*)ctx_scratch(ctx, SCRATCH_SMALL, 16)) == nullptr ||
        hipMemcpy(d_slot, &slot_index, sizeof(slot_index), hipMemcpyHostToDevice) != hipSuccess) {
      break;
    }
    // Host: ports back to back, slot_size samples each; device: slot_stride apart.
    if (hipMemcpy2D(d_iq, (size_t)plan->slot_stride * sizeof(float2), iq, slot_size * sizeof(float2),
                    slot_size * sizeof(float2), plan->nof_ports, hipMemcpyHostToDevice) != hipSuccess) {
      break;
    }
    rc = nrphy_ofdm_demod_run(plan, 1, (const float*)d_iq, d_slot, window_offset, d_grid, ctx->stream);
    if (rc != NRPHY_OK) {
      break;
    }
    rc = NRPHY_ERR_DEVICE;
    // Only the symbols a slot has (12 with extended cyclic prefix) are written, as by the reference's demodulator.
    const size_t port_bytes = (size_t)NRPHY_NSYMB * 12 * plan->cfg.bw_rb * 4;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy2D(grid, port_bytes, d_grid, port_bytes, (size_t)plan->nsymb * 12 * plan->cfg.bw_rb * 4,
                    plan->nof_ports, hipMemcpyDeviceToHost) != hipSuccess) {
      break;
    }
    rc = NRPHY_OK;
  } while (false);
  return rc;
}

extern "C" int nrphy_ofdm_demodulate_symbol_host(nrphy_ofdm_plan_t* plan, const float* input, uint32_t input_size,
                                                 uint32_t symbol_index, uint32_t window_offset, void* grid_row)
{
  if (plan == nullptr || input == nullptr || grid_row == nullptr || symbol_index >= plan->nsym_subframe ||
      input_size != plan->cp[symbol_index] + plan->cfg.dft_size) { // ofdm_demodulator_impl.cpp:110-118
    return NRPHY_ERR_ARGUMENT;
  }
  // One symbol of one port: the samples are placed where the slot kernel expects them (the other symbols of the
  // staging slot transform zeros), the symbol's row is read back.
  nrphy_ctx*     ctx  = plan->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
  const uint32_t rg   = 12 * plan->cfg.bw_rb;
  const uint32_t slot = symbol_index / plan->nsymb, l = symbol_index % plan->nsymb;
  const size_t   iq_samples = (size_t)plan->nof_ports * plan->slot_stride;
  uint32_t *     d_grid = nullptr, *d_slot = nullptr;
  float2*        d_iq   = nullptr;
  int            rc     = NRPHY_ERR_DEVICE;
  do {
    if (hipSetDevice(ctx->device) != hipSuccess ||
        (d_grid = (uint32_t*)ctx_scratch(ctx, SCRATCH_GRID, (size_t)plan->nof_ports * NRPHY_NSYMB * rg * 4)) == nullptr ||
        (d_iq = (float2*)ctx_scratch(ctx, SCRATCH_IQ, iq_samples * sizeof(float2))) == nullptr ||
        hipMemset(d_iq, 0, iq_samples * sizeof(float2)) != hipSuccess ||
        hipMemcpy(d_iq + plan->off[symbol_index], input, (size_t)input_size * sizeof(float2),
                  hipMemcpyHostToDevice) != hipSuccess ||
        (d_slot = (uint32_t*)ctx_scratch(ctx, SCRATCH_SMALL, 16)) == nullptr ||
        hipMemcpy(d_slot, &slot, sizeof(slot), hipMemcpyHostToDevice) != hipSuccess) {
      break;
    }
    rc = nrphy_ofdm_demod_run(plan, 1, (const float*)d_iq, d_slot, window_offset, d_grid, ctx->stream);
    if (rc != NRPHY_OK) {
      break;
    }
    rc = NRPHY_ERR_DEVICE;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy(grid_row, d_grid + (size_t)l * rg, (size_t)rg * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      break;
    }
    rc = NRPHY_OK;
  } while (false);
  return rc;
}

extern "C" int nrphy_ofdm_plan_enable_timing(nrphy_ofdm_plan_t* plan, uint32_t max_runs)
{
  if (plan == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  for (hipEvent_t e : plan->events) {
    (void)hipEventDestroy(e);
  }
  plan->events.assign(2 * (size_t)max_runs, nullptr);
  for (hipEvent_t& e : plan->events) {
    HIP_TRY(hipEventCreate(&e));
  }
  plan->max_timed_runs = max_runs;
  plan->timed_runs     = 0;
  plan->timing_counter = 0;
  return NRPHY_OK;
}

extern "C" int nrphy_ofdm_plan_timing_stride(nrphy_ofdm_plan_t* plan, uint32_t stride)
{
  if (plan == nullptr || stride == 0) {
    return NRPHY_ERR_ARGUMENT;
  }
  plan->timing_stride  = stride;
  plan->timing_counter = 0;
  return NRPHY_OK;
}

extern "C" int nrphy_ofdm_plan_kernel_time(nrphy_ofdm_plan_t* plan, float* avg_ms, uint32_t* nof_runs)
{
  if (plan == nullptr || avg_ms == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  double sum = 0;
  for (uint32_t r = 0; r != plan->timed_runs; ++r) {
    HIP_TRY(hipEventSynchronize(plan->events[2 * r + 1]));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, plan->events[2 * r], plan->events[2 * r + 1]));
    sum += ms;
  }
  *avg_ms = plan->timed_runs ? (float)(sum / plan->timed_runs) : 0.f;
  if (nof_runs) {
    *nof_runs = plan->timed_runs;
  }
  plan->timed_runs = 0;
  return NRPHY_OK;
}

extern "C" int nrphy_ofdm_modulate_symbol_host(nrphy_ofdm_plan_t* plan, const void* grid, uint32_t port_index,
                                               uint32_t symbol_index, float* output, uint32_t output_size)
{
  if (plan == nullptr || grid == nullptr || output == nullptr || port_index >= plan->nof_ports ||
      symbol_index >= plan->nsym_subframe ||
      output_size != plan->cp[symbol_index] + plan->cfg.dft_size) { // ofdm_modulator_impl.cpp:68-75
    return NRPHY_ERR_ARGUMENT;
  }
  nrphy_ctx*   ctx        = plan->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
  const size_t grid_words = (size_t)plan->nof_ports * NRPHY_NSYMB * 12 * plan->cfg.bw_rb;
  const size_t iq_samples = (size_t)plan->nof_ports * plan->slot_stride;
  uint32_t *   d_grid = nullptr, *d_slot = nullptr;
  float2*      d_iq   = nullptr;
  const uint32_t slot = symbol_index / plan->nsymb;
  int            rc   = NRPHY_ERR_DEVICE;
  do {
    if ((d_grid = (uint32_t*)ctx_scratch(ctx, SCRATCH_GRID, grid_words * 4)) == nullptr ||
        hipMemcpy(d_grid, grid, grid_words * 4, hipMemcpyHostToDevice) != hipSuccess ||
        (d_iq = (float2*)ctx_scratch(ctx, SCRATCH_IQ, iq_samples * sizeof(float2))) == nullptr ||
        (d_slot = (uint32_t*)ctx_scratch(ctx, SCRATCH_SMALL, 16)) == nullptr ||
        hipMemcpy(d_slot, &slot, sizeof(slot), hipMemcpyHostToDevice) != hipSuccess) {
      break;
    }
    rc = nrphy_ofdm_run(plan, 1, d_grid, d_slot, (float*)d_iq, ctx->stream);
    if (rc != NRPHY_OK) {
      break;
    }
    rc = NRPHY_ERR_DEVICE;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
      break;
    }
    const size_t src = (size_t)port_index * plan->slot_stride + plan->off[symbol_index];
    if (hipMemcpy(output, d_iq + src, (size_t)output_size * sizeof(float2), hipMemcpyDeviceToHost) != hipSuccess) {
      break;
    }
    rc = NRPHY_OK;
  } while (false);
  return rc;
}

extern "C" int nrphy_dft_run(nrphy_ctx_t* ctx, uint32_t size, int inverse, uint32_t batch, const float* d_in,
                             float* d_out, void* stream)
{
  if (ctx == nullptr || d_in == nullptr || d_out == nullptr || !dft_size_supported(size)) {
    return NRPHY_ERR_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t   s  = stream ? (hipStream_t)stream : ctx->stream;
  uint32_t      n1 = 1, n2 = size;
  dft_split(size, &n1, &n2);
  const float2* tw    = get_twiddle(ctx, size);
  const float2* tw_n2 = (n1 == 1) ? tw : get_twiddle(ctx, n2);
  if (tw == nullptr || tw_n2 == nullptr) {
    return NRPHY_ERR_DEVICE;
  }
  // The sizes beyond 6144 take two passes through a scratch copy of the batch (stream-ordered allocation).
  StreamStaging staging(s);
  float2*       d_tmp = nullptr;
  if (n1 != 1 && batch != 0) {
    d_tmp = (float2*)staging.alloc((size_t)batch * size * sizeof(float2));
    if (d_tmp == nullptr) {
      return NRPHY_ERR_DEVICE;
    }
  }
  HIP_TRY(launch_dft(size, inverse, batch, tw, tw_n2, d_tmp, (const float2*)d_in, (float2*)d_out, s));
  return NRPHY_OK;
}

extern "C" int nrphy_dft_run_host(nrphy_ctx_t* ctx, uint32_t size, int inverse, const float* in, float* out)
{
  if (ctx == nullptr || in == nullptr || out == nullptr || !dft_size_supported(size)) {
    return NRPHY_ERR_ARGUMENT;
  }
  const size_t bytes = (size_t)size * sizeof(float2);
  float *      d_in = nullptr, *d_out = nullptr;
  int          rc   = NRPHY_ERR_DEVICE;
  do {
    if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc((void**)&d_in, bytes) != hipSuccess ||
        hipMalloc((void**)&d_out, bytes) != hipSuccess ||
        hipMemcpy(d_in, in, bytes, hipMemcpyHostToDevice) != hipSuccess) {
      break;
    }
    rc = nrphy_dft_run(ctx, size, inverse, 1, d_in, d_out, ctx->stream);
    if (rc != NRPHY_OK) {
      break;
    }
    rc = NRPHY_ERR_DEVICE;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
      break;
    }
    rc = NRPHY_OK;
  } while (false);
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  return rc;
}

extern "C" int nrphy_ofdm_modulate_slot_host(nrphy_ofdm_plan_t* plan, const void* grid, uint32_t slot_index, float* iq)
{
  if (plan == nullptr || grid == nullptr || iq == nullptr || slot_index >= (1U << plan->cfg.numerology)) {
    return NRPHY_ERR_ARGUMENT;
  }
  nrphy_ctx*     ctx        = plan->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
  const size_t   grid_words = (size_t)plan->nof_ports * NRPHY_NSYMB * 12 * plan->cfg.bw_rb;
  const size_t   iq_samples = (size_t)plan->nof_ports * plan->slot_stride;
  const uint32_t slot_size  = nrphy_ofdm_slot_size(&plan->cfg, slot_index);
  uint32_t *     d_grid = nullptr, *d_slot = nullptr;
  float2*        d_iq   = nullptr;
  int            rc     = NRPHY_ERR_DEVICE;
  do {
    if (hipSetDevice(ctx->device) != hipSuccess || (d_grid = (uint32_t*)ctx_scratch(ctx, SCRATCH_GRID, grid_words * 4)) == nullptr ||
        hipMemcpy(d_grid, grid, grid_words * 4, hipMemcpyHostToDevice) != hipSuccess ||
        (d_iq = (float2*)ctx_scratch(ctx, SCRATCH_IQ, iq_samples * sizeof(float2))) == nullptr ||
        (d_slot = (uint32_t*)ctx_scratch(ctx, SCRATCH_SMALL, 16)) == nullptr ||
        hipMemcpy(d_slot, &slot_index, sizeof(slot_index), hipMemcpyHostToDevice) != hipSuccess) {
      break;
    }
    rc = nrphy_ofdm_run(plan, 1, d_grid, d_slot, (float*)d_iq, ctx->stream);
    if (rc != NRPHY_OK) {
      break;
    }
    rc = NRPHY_ERR_DEVICE;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
      break;
    }
    // Ports back to back on the host, slot_stride apart on the device: one strided copy.
    if (hipMemcpy2D(iq, (size_t)slot_size * sizeof(float2), d_iq, (size_t)plan->slot_stride * sizeof(float2),
                    (size_t)slot_size * sizeof(float2), plan->nof_ports, hipMemcpyDeviceToHost) == hipSuccess) {
      rc = NRPHY_OK;
    }
  } while (false);
  return rc;
}
