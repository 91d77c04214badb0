// PDSCH processor kernels for gfx950 (MI355X).
//
//   prologue_kernel    per-PDU work the codeblock waves consume: the transport-block CRC (a workgroup per run of 16 KiB
//                      regions, Horner's rule through byte tables in LDS, one share per workgroup), a 31-word seed of
//                      the scrambling sequence per codeblock work item, and the DM-RS sequences
//   codeblock_kernel   one wavefront per codeblock (or per 512-RE chunk of it): segmentation, CB-CRC, LDPC
//                      base-graph expansion in LDS, rate matching + bit interleaving as a word-level bit-matrix
//                      transposition, Gold scrambling, QAM mapping through an LDS table, layer mapping, precoding
//                      and RE mapping straight into the grid
//   dmrs_kernel        PDSCH DM-RS generation, CDM, precoding and mapping, one wavefront per 32 PRBs of a symbol
//   ldpc_encode_kernel the LDPC encoder alone (seam B / unit parity)
//
// Together they replace pdsch_processor_impl::process (R/lib/phy/upper/channel_processors/pdsch_processor_impl.cpp:30-184)
// in its per-codeblock form (pdsch_processor_concurrent_impl.cpp:55-338, pdsch_codeblock_processor.cpp:27-141).
#include "ldpc_device.h"

#ifndef NRPHY_PHASE_A_STAGED
#define NRPHY_PHASE_A_STAGED 1
#endif

// Stage stops of the profiling variants (profiles/make_variant.sh NAME "pdsch_kernels.hip ofdm_kernels.hip nrphy_host.cpp"
// "-DNRPHY_PROBES"; profiles/stage_pmc.sh, stage_times.sh): the codeblock waves return after stage n, the outputs are then
// incomplete.  The product library is built without them -- the tests compile to nothing.
#ifdef NRPHY_PROBES
#define NRPHY_STAGE(p) ((p).profile_stage)
#else
#define NRPHY_STAGE(p) 0u
#endif

namespace nrphy {

// ================================================================================================================
// Prologue: per-PDU work the codeblock waves consume.
//
// Transport block CRC (TS 38.212 Section 5.1; reference: ldpc_segmenter_impl.cpp:126, crc_calculator_lut_impl.cpp).
// A CRC is the remainder of a polynomial, so it splits: the transport block is cut into 16 KiB regions, a workgroup
// takes a run of them (CrcWork: one region each in a small batch, up to eight in a big one, the next region's words in
// flight while the current one is reduced) and the codeblock wave that attaches the CRC adds the workgroups' shares up.
// Inside a region thread t owns four groups of four consecutive words (16-byte loads, coalesced, straight from HBM):
// Horner's rule in x^32 inside a group, in y1k = x^(32 * 1024) across the groups -- four independent table look-ups per
// word; the 256 partials are folded by one wavefront the same way with y8k = x^(128 * 64); from region to region those
// 64 lanes step with yz = x^(8 * 16384), and only once per workgroup do they pay for a multiplication by a per-lane
// constant.
//
// Scrambling sequences (seeds per work item) and DM-RS sequences: see gold_sequence_blocks_wave() and gold_sequence_wave().
// ================================================================================================================
// Blocks [0, n_scr_work): the seeds of the scrambling sequence of one PDU (a part of them in a small batch) and its DM-RS
// sequences (TS 38.211 Sections 7.3.1.1, 7.4.1.1.1; reference: pdsch_modulator_impl.cpp:43-60,
// dmrs_pdsch_processor_impl.cpp:84-106).  The blocks after them: transport-block CRC.
__global__ __launch_bounds__(TB_CRC_THREADS) void prologue_kernel(PdschLaunch p, const uint8_t* __restrict__ d_tb)
{
  constexpr uint32_t LDS_WORDS = TB_CRC_LDS_WORDS; // CRC role: four tables and two buffers of partials (18 KB: 8 workgroups per CU)
  __shared__ __attribute__((aligned(16))) uint32_t lds[LDS_WORDS]; // sequence role: seed rows + DM-RS scratch
  static_assert(LDS_WORDS >= GOLD_RING_WORDS, "LDS of the sequence role");
  const uint32_t tid = threadIdx.x;

  // Order of the two roles in the grid (p.prologue_order, NRPHY_PROLOGUE_ORDER):
  //  0  sequence workgroups first, then the CRC workgroups;
  //  1  the sequence workgroups spread evenly among the CRC workgroups, every XCD taking a contiguous run of that list
  //     (block b runs on XCD b % 8).  A sequence workgroup is one long wave (plus three short DM-RS waves) and holds its slot
  //     on the CU for the whole time.  Measured: within 1 % of order 0 (profiles/r03_prologue_trace.txt); an A/B aid.
  // (With the workgroup-wide generator of rounds 1-3 -- a ring in LDS walked by all four waves -- interleaving cost +95 %:
  // ring and byte tables slowed each other down; and CRC first +11 %.  profiles/r02_codeblock_experiments.txt.)
  uint32_t scr_index = blockIdx.x, crc_index = blockIdx.x - p.n_scr_work;
  bool     scr_role  = blockIdx.x < p.n_scr_work;
  if (p.prologue_order == 1 && p.n_scr_work != 0 && p.n_crc_work != 0) {
    const uint32_t total = p.n_scr_work + p.n_crc_work;
    const uint32_t xcd = blockIdx.x & 7u, turn = blockIdx.x >> 3, q = total >> 3, r = total & 7u;
    const uint32_t pos = xcd * q + (xcd < r ? xcd : r) + turn; // position in the interleaved list
    // sequence workgroups before position x: ceil(x n_scr / total)
    const uint32_t s0 = (uint32_t)(((uint64_t)pos * p.n_scr_work + total - 1u) / total);
    const uint32_t s1 = (uint32_t)(((uint64_t)(pos + 1u) * p.n_scr_work + total - 1u) / total);
    scr_role  = s1 > s0;
    scr_index = s0;
    crc_index = pos - s0;
  }
  NRPHY_WG_TRACE_MARK(0);
  NRPHY_WG_TRACE_WHERE(scr_role ? 1 : 2);
  if (scr_role) { // workgroup-uniform
    if (NRPHY_STAGE(p) == 8) {
      return;
    }
    const auto*    swc   = to_constant(&p.scr_work[scr_index]);
    const uint32_t first = swc->first, count = swc->count;
    const bool     with_dmrs = swc->with_dmrs != 0;
    PduRef         pd    = *to_constant(&p.pdus[swc->pdu]);
    const uint32_t wave = tid / WAVE, lane = tid % WAVE;
    // The four waves of the workgroup work on their own, each on LDS of its own, without a barrier.  Wave 0 walks the part's
    // scrambling sequence (its x2 part: the codeblock waves add the x1 words, which every sequence shares) with the
    // recurrence in registers (gold_sequence_blocks_wave) and stores, for every work item of the PDU, the 31 words that
    // start at the word the item's first codeword bit lies in: the codeblock wave expands them to the few hundred words it
    // needs (gold_expand_seed_wave).  13 MB of seeds per 1024 config-3 slots instead of 121 MB of sequences out and back.
    constexpr uint32_t DMRS_WAVES = TB_CRC_THREADS / WAVE - 1, DMRS_SCRATCH = (GOLD_RING_WORDS - 2048u) / DMRS_WAVES;
    static_assert(GOLD_SEED_WORDS <= 2048u && GOLD_RING_WORDS > 2048u, "LDS of the sequence role");
    static_assert((12u * NRPHY_MAX_RB + 31u) / 32u + 1u <= DMRS_SCRATCH, "DM-RS scratch per wave");
    if (wave == 0) {
      // The PDU's work items in the order the plan lists them (nrphy_host.cpp: codeblock by codeblock, RE_CHUNK resource
      // elements per item), walked with scalar arithmetic alongside the blocks: w0 = the word of the item's first bit.
      const uint32_t lq = pd.qm * pd.nof_layers, n_short = pd.n_short, e_short = pd.e_short, e_long = pd.e_long, C = pd.C;
      const uint32_t nre_short = e_short / lq, nre_long = e_long / lq; // (the only divisions: the walk itself has none)
      struct Cursor {
        uint32_t cb, re_begin, item, bit_cb, nre;
      };
      Cursor resume = {0u, 0u, pd.item_first, 0u, n_short != 0 ? nre_short : nre_long}; // the first item whose seed is not complete yet
      auto   advance = [&](Cursor& c) {
        c.re_begin += RE_CHUNK;
        if (c.re_begin >= c.nre) {
          c.bit_cb += c.cb < n_short ? e_short : e_long;
          ++c.cb;
          c.re_begin = 0;
          c.nre      = c.cb < n_short ? nre_short : nre_long;
        }
        ++c.item;
      };
      gold_sequence_blocks_wave(p.gold, pd.c_init, first, count, lds, lane, [&](uint32_t base, uint32_t avail) {
        const uint32_t lo = first + base, hi = lo + avail; // the block holds words [lo, hi) of the PDU's sequence
        // Every item whose 31 words meet the block gets what the block holds of them.  Seeds of neighbouring items overlap
        // when an item is short (the tail of a codeblock cut into RE_CHUNK pieces): an item that runs beyond the block does not
        // end the walk, it only marks where the next block (or the next part's first) takes it up again.
        Cursor c        = resume;
        bool   resuming = false;
        while (c.cb < C) { // wave-uniform
          const uint32_t w0 = (c.bit_cb + c.re_begin * lq) >> 5;
          if (w0 >= hi) {
            break;
          }
          if (w0 + 31u > lo) {
            const uint32_t k = w0 + lane;
            if (lane < 31u && k >= lo && k < hi) {
              p.scr_seed[(size_t)c.item * 32u + lane] = lds[k - lo];
            }
          }
          if (w0 + 31u > hi && !resuming) {
            resume   = c;
            resuming = true;
          }
          advance(c);
        }
        if (!resuming) {
          resume = c;
        }
      });
      NRPHY_WG_TRACE_MARK(6);
    } else if (with_dmrs && pd.dmrs_seq_words <= DMRS_SCRATCH) {
      // Waves 1-3 of the PDU's first workgroup: the DM-RS sequences, from bit 0 to the last allocated PRB -- short, so one
      // wave generates one; the waves take the DM-RS symbols in turn.
      uint32_t ordinal = 0;
      for (uint32_t mask = pd.dmrs_symbol_mask; mask != 0; mask &= mask - 1u, ++ordinal) { // uniform
        if (ordinal % DMRS_WAVES == wave - 1u) {
          const uint32_t l = (uint32_t)__ffs(mask) - 1u;
          gold_sequence_wave(p.gold, p.x1_words, pd.dmrs_c_init[l], pd.dmrs_seq_words,
                             p.scr + pd.dmrs_seq_offset + ordinal * pd.dmrs_seq_words,
                             lds + 2048u + (wave - 1u) * DMRS_SCRATCH, lane);
        }
      }
    }
    return;
  }

  if (NRPHY_STAGE(p) == 9) {
    return;
  }
  const auto*     wkc = to_constant(&p.crc_work[crc_index]);
  const uint32_t  wk_pdu = wkc->pdu, wk_region = wkc->region, wk_factor = wkc->factor, wk_count = wkc->count;
  PduRef          pd  = *to_constant(&p.pdus[wk_pdu]);
  const uint32_t  sel = (pd.tb_crc_bits == 16) ? 1u : 0u;
  const CrcPoly   c   = sel ? crc16() : crc24a();
  const uint32_t  n   = pd.tb_bytes;
  const uint32_t* tbw = reinterpret_cast<const uint32_t*>(d_tb + pd.tb_offset);
#ifdef NRPHY_WG_TRACE
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (trace builds: the descriptors have arrived)
  NRPHY_WG_TRACE_MARK(3);
#endif
  const uint32_t share = tbcrc_regions_workgroup(p.tbcrc, sel, c, tbw, n, wk_region, wk_count, wk_factor, lds, tid);
  if (tid == 0) {
    // The workgroup's share of the PDU's CRC in a slot of its own: the codeblock wave that attaches the CRC adds the
    // shares up, so a run neither relies on nor leaves behind any accumulator state.
    p.tb_crc_part[crc_index] = share;
  }
  NRPHY_WG_TRACE_MARK(6);
}

#ifdef NRPHY_WG_TRACE
extern "C" int nrphy_debug_wg_trace(uint64_t* out, uint32_t nof_blocks)
{
  const size_t n = 8 * (size_t)(nof_blocks < WG_TRACE_MAX ? nof_blocks : WG_TRACE_MAX) * sizeof(uint64_t);
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_trace), n, 0, hipMemcpyDeviceToHost);
}
#endif

hipError_t launch_prologue(const PdschLaunch& p, const uint8_t* d_tb, hipStream_t stream)
{
  if (p.n_crc_work + p.n_scr_work == 0) {
    return hipSuccess;
  }
  // (The zero-fill waves were tried as a third role of this launch, which leaves most of the memory bandwidth unused: the
  // prologue 0.084 -> 0.095 ms, the codeblock launch 0.303 -> 0.352 ms, the OFDM launch 0.50 -> 0.475 ms, the whole step
  // -4 %: profiles/r03_codeblock_experiments.txt.  They stay at the tail of the codeblock launch.)
  hipLaunchKernelGGL(prologue_kernel, dim3(p.n_crc_work + p.n_scr_work), dim3(TB_CRC_THREADS), 0, stream, p, d_tb);
  return hipGetLastError();
}

// ================================================================================================================
// Codeblock construction in LDS (TS 38.212 Section 5.2.2; reference: ldpc_segmenter_impl.cpp:148-217).
// ================================================================================================================
// LDS of one codeblock wavefront: `lin` (the codeblock, sized per launch from the largest one the plan contains) and one
// scratch region `u` that the stages of the wave use one after the other:
//   building the codeblock   u[0, 256 * NRPHY_CRC_SLICES)      byte tables of the codeblock CRC
//   LDPC                     u[0, LDPC_DBL_WORDS)              the systematic blocks doubled (ldpc_device.h)
//                            then LdpcScratch (core rows), then row pointers + edges of the lifted graph rows needed
//   output stage             u[0, 512)                         the modulation table (index = Qm bits, value = ci8 symbol as floats)
//                            u[512, ...)                       interleaver output: one byte per modulation symbol
// At the headline configuration that is 1.3 + 3.2 KB of LDS per wave: 8 waves per SIMD fit the CU's 160 KB.  (Measured,
// A/B on one box: four CRC tables -- a 32-bit word per step, but 4 KB of scratch and 7 waves -- 0.347 ms per 1024 slots,
// three tables 0.328 ms.)
constexpr uint32_t CB_U_QAM_WORDS     = 512;
constexpr uint32_t CB_U_LDPC_WORDS    = (sizeof(LdpcScratch) / 4u + 3u) & ~3u;
constexpr uint32_t CB_U_GRAPH_OFFSET  = LDPC_DBL_WORDS + CB_U_LDPC_WORDS;
static_assert(CB_U_GRAPH_OFFSET == NRPHY_CB_U_GRAPH_OFFSET, "host and device agree on the layout of the scratch region");

struct CbShared {
  uint32_t* lin;   // codeblock bits, (Kb + rows) * Zc bits (+ read-ahead)
  uint32_t* u;     // the scratch region
  uint32_t* symb;  // u + CB_U_QAM_WORDS
  uint32_t* graph; // u + CB_U_GRAPH_OFFSET
  LdpcScratch* ldpc; // u + LDPC_DBL_WORDS
};

// Bounding experiment of round 4 (profiles/r04_lds_conflicts.txt; variant builds only, the results are then WRONG): what the
// LDS bank conflicts of the data-indexed table look-ups cost.  NRPHY_LDS_PROBE bit 0: the modulation table is read at the
// lane's own entry, bit 1: the codeblock CRC's byte tables likewise -- the same instructions and dependences (the index still
// depends on the data through an opaque zero), consecutive addresses instead of data-dependent ones: no conflicts.
#ifndef NRPHY_LDS_PROBE
#define NRPHY_LDS_PROBE 0
#endif
__device__ __forceinline__ uint32_t lds_probe_index(uint32_t idx, uint32_t bit)
{
  if ((NRPHY_LDS_PROBE >> bit) & 1) {
    uint32_t zero = 0;
    asm volatile("" : "+v"(zero));
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return (idx & zero) + lane;
  }
  return idx;
}

// reg <- CRC24B register after the 32 bits of `word`: independent look-ups (tab[k][b] = (b x^(8k) x^24) mod g) instead of
// four dependent byte steps.
__device__ __forceinline__ uint32_t crc24_word_step(const uint32_t* tab, uint32_t reg, uint32_t word)
{
#if NRPHY_CRC_SLICES == 4
  const uint32_t v = (reg << 8) ^ word;
  return tab[v & 0xFFu] ^ tab[256u + ((v >> 8) & 0xFFu)] ^ tab[512u + ((v >> 16) & 0xFFu)] ^ tab[768u + (v >> 24)];
#else
  const uint32_t v = reg ^ (word >> 8); // 24 bits through three tables, then the last byte
  uint32_t       r = tab[lds_probe_index(v & 0xFFu, 1)] ^ tab[256u + lds_probe_index((v >> 8) & 0xFFu, 1)] ^ tab[512u + lds_probe_index(v >> 16, 1)];
  return ((r << 8) & 0xFFFFFFu) ^ tab[lds_probe_index(((r >> 16) ^ word) & 0xFFu, 1)];
#endif
}

// Fills lin with the K bits of codeblock `cb` (payload, TB CRC + zero padding on the last codeblock, CB CRC, filler
// zeros) and zeroes the parity region up to `total_words`.
__device__ inline void build_codeblock(PduRef pd, uint32_t cb, const uint32_t* tbw, const uint32_t* tb_crc_part,
                                       const GoldTables* tables, CbShared* sh, uint32_t total_words, uint32_t lane,
                                       uint32_t profile_stage)
{
  const bool     last    = (cb == pd.C - 1);
  const uint32_t used    = pd.info_bits - (last ? pd.tb_crc_bits + pd.zero_pad : 0u);
  const uint32_t tb_pos  = cb * pd.info_bits;
  const uint32_t tb_bits = pd.tb_bytes * 8u;
  // The CRC's byte tables (16 bytes per lane and step) are requested first and stored behind the segmentation
  // (unconditionally: a transport block of one codeblock carries no codeblock CRC and simply does not use them).
  const uint4* crc_src = reinterpret_cast<const uint4*>(&tables->crc24b_slice[0][0]);
  const uint4  crc_tab0 = crc_src[lane], crc_tab1 = crc_src[lane + WAVE], crc_tab2 = crc_src[lane + 2 * WAVE];
#if NRPHY_CRC_SLICES == 4
  const uint4 crc_tab3 = crc_src[lane + 3 * WAVE];
#endif
  // The transport-block words of SEG_UNROLL trips are requested before the first is used: taken trip by trip, every trip
  // waited out a trip to memory (stage timing: this loop alone cost 0.045 ms per 1024 slots for a hundred instructions).
  constexpr uint32_t SEG_UNROLL = 5; // 320 words: a whole high-rate codeblock with its four core parity blocks
  for (uint32_t base = lane; base < total_words; base += WAVE * SEG_UNROLL) {
    uint32_t hi[SEG_UNROLL], lo[SEG_UNROLL];
#pragma unroll
    for (uint32_t k = 0; k != SEG_UNROLL; ++k) {
      const uint32_t pos = 32u * (base + WAVE * k);
      hi[k] = 0;
      lo[k] = 0;
      if (pos < used) { // never touch words that lie entirely beyond the transport block
        const uint32_t abs_pos = tb_pos + pos;
        const uint32_t i = abs_pos >> 5, sft = abs_pos & 31u;
        hi[k] = tbw[i];
        if (sft != 0 && 32u * (i + 1) < tb_bits) {
          lo[k] = tbw[i + 1];
        }
      }
    }
#pragma unroll
    for (uint32_t k = 0; k != SEG_UNROLL; ++k) {
      const uint32_t j = base + WAVE * k, pos = 32u * j;
      if (j < total_words) {
        uint32_t v = 0;
        if (pos < used) {
          v = __funnelshift_l(__builtin_bswap32(lo[k]), __builtin_bswap32(hi[k]), (tb_pos + pos) & 31u);
          const uint32_t remaining = used - pos;
          if (remaining < 32u) {
            v &= topmask(remaining);
          }
        }
        sh->lin[j] = v;
      }
    }
  }
  {
    uint4* dst           = reinterpret_cast<uint4*>(sh->u);
    dst[lane]            = crc_tab0;
    dst[lane + WAVE]     = crc_tab1;
    dst[lane + 2 * WAVE] = crc_tab2;
#if NRPHY_CRC_SLICES == 4
    dst[lane + 3 * WAVE] = crc_tab3;
#endif
  }
  wave_sync();
  if (profile_stage == 7) {
    return;
  }
  if (last) { // wave-uniform: the transport-block CRC = XOR of the shares of its regions (prologue)
    const uint32_t tb_crc = wave_xor(lane < pd.crc_count ? tb_crc_part[pd.crc_first + lane] : 0u);
    if (lane == 0) {
      or_bits_lds_exclusive(sh->lin, used, tb_crc << (32u - pd.tb_crc_bits), pd.tb_crc_bits);
    }
  }
  wave_sync();
  if (pd.cb_crc_bits) {
    // CRC24B over the first info_bits bits.  Leading zeros do not change a zero-initialised CRC, so the bit string
    // is right-aligned to a word boundary: word j holds message bits [32j - pad, 32j - pad + 32).
    const uint32_t n   = pd.info_bits;
    const uint32_t pad = (32u - (n & 31u)) & 31u;
    const uint32_t nw  = (n + pad) >> 5;
    const uint32_t per = (nw + WAVE - 1) / WAVE;
    uint32_t       a   = lane * per;
    uint32_t       b   = a + per;
    a                  = a > nw ? nw : a;
    b                  = b > nw ? nw : b;
    uint32_t reg       = 0;
    for (uint32_t j = a; j < b; ++j) {
      uint32_t word;
      if (pad == 0) {
        word = sh->lin[j];
      } else if (j == 0) {
        word = sh->lin[0] >> pad;
      } else {
        word = ext32(sh->lin, 32u * j - pad);
      }
      reg = crc24_word_step(sh->u, reg, word);
    }
    // The lane's partial times x^(32 (nw - b)): six independent table look-ups (one per nibble).
    uint32_t part = 0;
    {
      const uint32_t m = nw - b;
#pragma unroll
      for (uint32_t k = 0; k != 6; ++k) {
        part ^= tables->crc24b_mul[m][k][(reg >> (4u * k)) & 15u];
      }
    }
    uint32_t crc = wave_xor(part);
    if (lane == 0) {
      or_bits_lds_exclusive(sh->lin, n, crc << 8, 24);
    }
    wave_sync();
  }
}

// ================================================================================================================
// Rate matching index arithmetic (TS 38.212 Section 5.4.2; reference: ldpc_rate_matcher_impl.cpp:37-144).
// Bit t of the selected sequence e_0, e_1, ... is circular-buffer bit pos(t); the buffer skips the filler interval
// [fs, fs + flen) and wraps at Ncb.  lin holds the codeblock including the 2*Zc punctured bits, hence the + 2*Zc.
// ================================================================================================================
struct RmIndex {
  uint32_t fs;       // first filler position (clamped to Ncb)
  uint32_t flen;     // filler positions inside the circular buffer
  uint32_t n_valid;  // Ncb - flen
  uint32_t rank0;    // rank of k0 among the valid positions
  float    inv_valid;
};

__device__ __forceinline__ RmIndex rm_index_init(PduRef pd)
{
  RmIndex  r;
  uint32_t nsys = (pd.kb - 2u) * pd.zc;
  uint32_t fs   = nsys - pd.filler;
  uint32_t fe   = nsys;
  fs            = fs > pd.n_cb ? pd.n_cb : fs;
  fe            = fe > pd.n_cb ? pd.n_cb : fe;
  r.fs          = fs;
  r.flen        = fe - fs;
  r.n_valid     = pd.n_cb - r.flen;
  r.rank0       = pd.k0 < fs ? pd.k0 : (pd.k0 < fe ? fs : pd.k0 - r.flen);
  r.inv_valid   = 1.0f / (float)r.n_valid;
  return r;
}

// 32 consecutive selected bits e_t .. e_(t+31), first bit in the MSB (bits beyond the codeblock's E are don't-care).
// A run of selected bits is contiguous in lin until it meets the filler gap or the end of the circular buffer; the
// common case is one funnel read, crossings are stitched piece by piece.  WRAP = false: rank0 + E <= n_valid.
template <bool WRAP>
__device__ __forceinline__ uint32_t rm_gather32(const RmIndex& r, const uint32_t* lin, uint32_t zc2, uint32_t t)
{
  uint32_t u = r.rank0 + t;
  if (WRAP && u >= r.n_valid) {
    uint32_t q = (uint32_t)((float)u * r.inv_valid);
    u -= q * r.n_valid;
    if ((int32_t)u < 0) {
      u += r.n_valid;
    } else if (u >= r.n_valid) {
      u -= r.n_valid;
    }
  }
  uint32_t run = (u < r.fs) ? r.fs - u : r.n_valid - u; // selected bits until the next discontinuity
  uint32_t out = ext32(lin, (u < r.fs ? u : u + r.flen) + zc2);
  if (run >= 32u) {
    return out;
  }
  out &= topmask(run);
  uint32_t filled = run;
  while (filled < 32u) {
    u += run;
    if (u >= r.n_valid) {
      if (!WRAP) {
        break; // past the end of the selection: the remaining bits are never used
      }
      u = 0;
    }
    run            = (u < r.fs) ? r.fs - u : r.n_valid - u;
    run            = run > 32u - filled ? 32u - filled : run;
    uint32_t piece = ext32(lin, (u < r.fs ? u : u + r.flen) + zc2);
    out |= (piece & topmask(run)) >> filled;
    filled += run;
  }
  return out;
}

// 8x8 bit-matrix transposition of the 64-bit word (hi:lo), rows = bytes (most significant first), columns = bits
// (most significant first): three rounds of masked swaps (Hacker's Delight 7-3), on 32-bit halves.
// Bits of a where m is set, bits of b elsewhere: ONE v_bitop3_b32 (truth table 0xCA).  Written as (a & m) | (b & ~m) with
// constant masks the compiler re-associated the shifts and masks of transpose8x8 into some sixty and / or / shift
// instructions per transposition; this way it is two shifts and two of these per half and round.
__device__ __forceinline__ uint32_t bit_select(uint32_t m, uint32_t a, uint32_t b)
{
  return __builtin_amdgcn_bitop3_b32(m, a, b, 0xCA);
}

__device__ __forceinline__ void transpose8x8(uint32_t& hi, uint32_t& lo)
{
  // Each round keeps the bits under one mask, takes the left-shifted word under a second one and the right-shifted
  // word elsewhere (the three masks partition the word): two shifts and two bit-selects.
  hi = bit_select(0xAA55AA55u, hi, bit_select(0x55005500u, hi << 7, hi >> 7));
  lo = bit_select(0xAA55AA55u, lo, bit_select(0x55005500u, lo << 7, lo >> 7));
  hi = bit_select(0xCCCC3333u, hi, bit_select(0x33330000u, hi << 14, hi >> 14));
  lo = bit_select(0xCCCC3333u, lo, bit_select(0x33330000u, lo << 14, lo >> 14));
  const uint32_t nhi = bit_select(0xF0F0F0F0u, hi, lo >> 4);
  const uint32_t nlo = bit_select(0x0F0F0F0Fu, lo, hi << 4);
  hi                 = nhi;
  lo                 = nlo;
}

// x * w evaluated like the reference's SIMD precoder (channel_precoder_avx2.cpp:51-56):
// fmaddsub(x, w.re, swap(x) * w.im) -- one rounding on the imaginary-weight product, one fused on the rest.
__device__ __forceinline__ void cmul_ref(float xr, float xi, float wr, float wi, float& outr, float& outi)
{
  float t0 = __fmul_rn(xi, wi);
  float t1 = __fmul_rn(xr, wi);
  outr     = __fmaf_rn(xr, wr, -t0);
  outi     = __fmaf_rn(xi, wr, t1);
}

// The same on packed FP32: a complex number is one 64-bit register pair (re, im), the product two v_pk_*_f32
// instructions instead of four scalar ones -- t = (x.im w.im, x.re w.im) rounded, then (fma(x.re, w.re, -t.lo),
// fma(x.im, w.re, t.hi)): operation for operation what cmul_ref does.  The codeblock launch is VALU-bound (PMC: the
// vector units are busy 87 % of its cycles) and the 4 x 4 precoding products are its largest block of arithmetic.
typedef float cf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf2 cmul_ref_packed(cf2 x, cf2 w)
{
  cf2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(t) : "v"(x), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(x), "v"(w), "v"(t));
  return r;
}
// Wave-uniform weight held in an SGPR pair (wideband precoding).
__device__ __forceinline__ cf2 cmul_ref_packed_uniform(cf2 x, cf2 w)
{
  cf2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(t) : "v"(x), "s"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(x), "s"(w), "v"(t));
  return r;
}

// (re, im) -> cbf16 word, round to nearest even like to_bf16 (R/include/srsran/adt/bf16.h:39-56); v_cvt_pk_bf16_f32.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float  f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_cbf16(float re, float im)
{
  f32x2_t  v = {re, im};
  bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
  return *reinterpret_cast<uint32_t*>(&b);
}

// Data RE go to the grid with non-temporal stores: 0.75 GB per 1024 slots that this kernel never reads back would
// otherwise displace the transport blocks, sequences and tables it does read.  Measured (A/B on one box): codeblock
// launch 0.486 -> 0.427 ms, whole step +7 %.  The same policy on the DM-RS / zero-fill stores, on the loads of the
// transport block or of the scrambling words, or on the prologue's sequence stores did not pay (0 ... -4 %).
__device__ __forceinline__ void grid_store(uint32_t* p, uint32_t v)
{
  __builtin_nontemporal_store(v, p);
}

#ifndef NRPHY_GRID_STORE_AUX
#define NRPHY_GRID_STORE_AUX 2 // buffer-store cache policy of the data RE: 2 = non-temporal (0 = default, for A/B builds)
#endif
#ifndef NRPHY_GRID_BUFFER_STORES
#define NRPHY_GRID_BUFFER_STORES 1 // A/B on one box: -1.3 % on the codeblock launch against 64-bit flat addressing
#endif

struct ChunkGeom {
  uint32_t E;      // rate-matched length of the codeblock
  uint32_t cw_cb;  // first codeword bit of the codeblock
  uint32_t bit0;   // first codeword bit of the chunk
};

// ================================================================================================================
// Output stage of the codeblock kernel for one (modulation order, layer count).
//
// Phase A -- rate matching + bit interleaving (TS 38.212 Section 5.4.2.2) as a bit-matrix transposition: the
//   interleaver writes E/Qm-bit rows and reads columns, so 32 consecutive modulation symbols are the columns of
//   Qm row words.  A lane gathers the Qm row words of its 32 symbols and transposes them 8x8-block-wise into one
//   byte per symbol (symbol bits in the byte's MSBs) -- 8x less work than extracting bit by bit.
// Phase B -- per RE: scramble, QAM table lookup, layer mapping + precoding, bf16, coalesced stores.
//
// The two phases are separate functions and read the PDU descriptor where they use it (scalar loads through the
// constant address space) instead of carrying its fields from the top of the kernel: as one function inlined 32 times
// into one kernel the scalar register file overflowed (106 SGPRs, 393 spilled, 2,387 v_readlane restores -- vector
// instructions in a kernel bound by vector issue).
// ================================================================================================================
template <int QM, int L, bool WRAP>
__device__ __forceinline__ void phase_a(const PdschLaunch& p, PduRef pd, const CbWork& wk, const CbShared& sh,
                                        uint32_t E, uint32_t lane)
{
  const uint32_t esym = E / QM; // rows of the bit interleaver have esym bits
  const uint32_t zc2  = 2u * pd.zc;
  const RmIndex  rm   = rm_index_init(pd);
  const uint32_t s0   = wk.re_begin * L;             // first modulation symbol of the chunk (multiple of 32)
  const uint32_t nblk = (wk.re_count * L + 31u) >> 5;
  // The modulation table's trip to memory starts here and ends behind the two passes below.
  constexpr uint32_t LUT_TRIPS = ((1u << QM) + WAVE - 1u) / WAVE;
  float2             lut[LUT_TRIPS];
#pragma unroll
  for (uint32_t k = 0; k != LUT_TRIPS; ++k) {
    lut[k] = lane + WAVE * k < (1u << QM) ? p.gold->qam_lut[QM / 2 - 1][lane + WAVE * k] : make_float2(0.f, 0.f);
  }
#if NRPHY_PHASE_A_STAGED
  // Two passes over the wave's 64 lanes instead of one lane per block (a 281-RE codeblock has 36 blocks: 36 busy lanes doing
  // Qm gathers and four transpositions each).  First the Qm * nblk row words, one gather per item, into the block's eight
  // words of `symb` (word 8 blk + j = row j); then the 4 * nblk (block, byte column) items: eight byte reads, one 8x8
  // transposition, two words back into the same eight words (the four items of a block sit in neighbouring lanes of one
  // instruction: all their reads precede their writes).
  for (uint32_t item = lane; item < QM * nblk; item += WAVE) {
    const uint32_t blk = item / QM, j = item - blk * QM;
    sh.symb[8u * blk + j] = rm_gather32<WRAP>(rm, sh.lin, zc2, j * esym + s0 + 32u * blk);
  }
  wave_sync();
  const uint8_t* rows = reinterpret_cast<const uint8_t*>(sh.symb);
  for (uint32_t item = lane; item < 4u * nblk; item += WAVE) {
    const uint32_t blk = item >> 2, m = item & 3u;
    const uint8_t* b   = rows + 32u * blk + (3u - m); // byte 3 - m of a row word holds its columns 8m .. 8m + 7
    uint32_t       r8[8];
#pragma unroll
    for (int j = 0; j != 8; ++j) {
      r8[j] = (j < QM) ? b[4 * j] : 0u;
    }
    uint32_t hi = (r8[0] << 24) | (r8[1] << 16) | (r8[2] << 8) | r8[3];
    uint32_t lo = (r8[4] << 24) | (r8[5] << 16) | (r8[6] << 8) | r8[7];
    transpose8x8(hi, lo);
    sh.symb[8u * blk + 2u * m]      = hi;
    sh.symb[8u * blk + 2u * m + 1u] = lo;
  }
#else
  for (uint32_t blk = lane; blk < nblk; blk += WAVE) {
    uint32_t row[8];
#pragma unroll
    for (int j = 0; j != 8; ++j) {
      row[j] = (j < QM) ? rm_gather32<WRAP>(rm, sh.lin, zc2, (uint32_t)j * esym + s0 + 32u * blk) : 0u;
    }
#pragma unroll
    for (int m = 0; m != 4; ++m) { // columns 8m .. 8m+7
      const int sft = 24 - 8 * m;
      uint32_t  hi  = (((row[0] >> sft) & 0xFFu) << 24) | (((row[1] >> sft) & 0xFFu) << 16) |
                    (((row[2] >> sft) & 0xFFu) << 8) | ((row[3] >> sft) & 0xFFu);
      uint32_t lo = (((row[4] >> sft) & 0xFFu) << 24) | (((row[5] >> sft) & 0xFFu) << 16) |
                    (((row[6] >> sft) & 0xFFu) << 8) | ((row[7] >> sft) & 0xFFu);
      transpose8x8(hi, lo);
      sh.symb[8u * blk + 2u * m]      = hi; // symbols 8m .. 8m+3 of the block
      sh.symb[8u * blk + 2u * m + 1u] = lo; // symbols 8m+4 .. 8m+7
    }
  }
#endif
  if (lane < 8) {
    sh.symb[8u * nblk + lane] = 0; // read-ahead padding
  }
  // Modulation table, requested before the gathers (`lut` above), into the scratch region behind the symbol bytes' use of it.
#pragma unroll
  for (uint32_t k = 0; k != LUT_TRIPS; ++k) {
    if (lane + WAVE * k < (1u << QM)) {
      reinterpret_cast<float2*>(sh.u)[lane + WAVE * k] = lut[k];
    }
  }
  wave_sync();
}

// A wave-uniform complex weight as an opaque value in one aligned scalar register pair: the compiler can neither
// re-load it inside the RE loop (it used to: one s_load_dwordx8 + s_waitcnt per port and 64 RE) nor split the pair.
__device__ __forceinline__ void pin_scalar(cf2& w)
{
  asm volatile("" : "+s"(w));
}

// What phase B knows about the chunk before it looks at a resource element.
struct ChunkMap {
  uint32_t re0;      // first RE of the chunk within the PDU
  uint32_t word0;    // first scrambling word of the chunk (valid when the chunk starts on a word and L * Qm = 32)
  bool     aligned;  // L * Qm = 32 and the chunk starts on a word boundary: one scrambling word per RE
};

// The L * Qm scrambling bits of RE r of the chunk (MSB first) are the XOR of two parts.  x1 is the same for every sequence:
// a table in global memory (L2), requested a trip ahead; zero beyond the chunk.  L * Qm = 32 with a word-aligned chunk (the
// headline shape) makes the bits one word of each part.
template <int QM, int L>
__device__ __forceinline__ uint32_t x1_bits(const PdschLaunch& p, const ChunkGeom& g, const ChunkMap& cm, uint32_t re_count,
                                            uint32_t r)
{
  uint32_t bits = 0;
  if (r < re_count) {
    if (QM * L == 32 && cm.aligned) { // wave-uniform: scalar base, small per-lane index
      bits = (p.x1_words + cm.word0)[r];
    } else {
      bits = ext32(p.x1_words, g.bit0 + r * (uint32_t)(QM * L));
    }
  }
  return bits;
}
// x2 depends on the PDU: the wave has expanded its seed into LDS, x2[0] = the word the chunk's first bit lies in (r < re_count).
template <int QM, int L>
__device__ __forceinline__ uint32_t x2_bits(const uint32_t* x2, const ChunkGeom& g, const ChunkMap& cm, uint32_t r)
{
  if (QM * L == 32 && cm.aligned) { // wave-uniform
    return x2[r];
  }
  return ext32(x2, (g.bit0 & 31u) + r * (uint32_t)(QM * L));
}

// The RE's L symbol bytes (first in the MSB) and its L*Qm scrambling bits.  Four layers make the first a whole word.
template <int QM, int L>
__device__ __forceinline__ void re_bits(const PdschLaunch& p, const CbShared& sh, const ChunkGeom& g, const ChunkMap& cm,
                                        uint32_t re_count, uint32_t r, uint32_t& bytes, uint32_t& gbits)
{
  if constexpr (L == 4) {
    bytes = sh.symb[r];
  } else {
    bytes = ext32(sh.symb, 8u * r * L);
  }
  gbits = x1_bits<QM, L>(p, g, cm, re_count, r) ^ x2_bits<QM, L>(sh.lin, g, cm, r);
}

// The grid loop of phase B for P ports with wideband precoding (every weight in a scalar register pair for the whole
// loop, no guards, no loads), or -- P = 0 -- for any port count with weights per PRG read from memory per RE.
template <int QM, int L, int P>
__device__ __forceinline__ void phase_b_grid(const PdschLaunch& p, PduRef pd, const PduDev* __restrict__ pd_global,
                                             const CbWork& wk, const CbShared& sh, const ChunkGeom& g, const ChunkMap& cm,
                                             uint32_t first_gbits, uint32_t lane, uint32_t* __restrict__ d_grid)
{
  const size_t   grid_base   = (size_t)pd.grid_index * p.grid_nof_ports * NRPHY_NSYMB * p.grid_nof_subc;
  const uint32_t plane_words = NRPHY_NSYMB * p.grid_nof_subc;
  // The PDU's grid through a buffer descriptor (scalar base, 32-bit per-lane offsets, the port plane in the scalar offset):
  // no 64-bit address arithmetic in the vector unit.  (Profiling aid, NRPHY_PROFILE_STAGE=11: a zero-sized descriptor
  // drops the data stores, everything else runs.)
  const __amdgpu_buffer_rsrc_t grid_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      d_grid + grid_base, 0, (int)(NRPHY_STAGE(p) == 11 ? 0u : p.grid_nof_ports * plane_words * 4u), 0x00020000);
  cf2 wu[P > 0 ? P : 1][L];
  if constexpr (P > 0) {
    const float NRPHY_CONSTANT* wuni = to_constant(p.weights + pd.weights_offset);
#pragma unroll
    for (int port = 0; port != P; ++port) {
#pragma unroll
      for (int l = 0; l != L; ++l) {
        wu[port][l] = cf2{wuni[2 * (port * L + l)], wuni[2 * (port * L + l) + 1]};
      }
    }
#pragma unroll
    for (int port = 0; port != P; ++port) {
#pragma unroll
      for (int l = 0; l != L; ++l) {
        pin_scalar(wu[port][l]);
      }
    }
  }
  // The OFDM symbol of the chunk's first RE, found once; the loop below only checks (on the scalar unit) whether the
  // next 64 RE stay inside the current symbol and walks on when they do not.  What the loop needs of the current symbol
  // travels in scalar registers (loaded when the symbol changes, not per 64 RE).
  uint32_t l_cur = 0;
#pragma unroll 1
  while (l_cur + 1u < NRPHY_NSYMB && cm.re0 >= pd.sym_re_start[l_cur + 1u]) {
    ++l_cur;
  }
  uint32_t cur_start = pd.sym_re_start[l_cur], cur_end = pd.sym_re_start[l_cur + 1u];
  uint32_t cur_arg = pd.sym_arg[l_cur], cur_row = l_cur * p.grid_nof_subc;
  bool     cur_table = pd.sym_kind[l_cur] == SYM_TABLE;
  // The x1 part of the scrambling bits comes from global memory (L2): the words of the NEXT 64 RE are requested before a
  // trip's arithmetic starts, those of the first 64 RE were requested before rate matching (map_chunk).  The x2 part is in
  // LDS (sh.lin, expanded from the work item's seed).
  uint32_t gbits_next = first_gbits;

  for (uint32_t r0 = 0; r0 < wk.re_count; r0 += WAVE) { // r0 is wave-uniform
    const uint32_t r        = r0 + lane;
    const uint32_t x1       = gbits_next;
    gbits_next              = x1_bits<QM, L>(p, g, cm, wk.re_count, r + WAVE);
    const uint32_t re_first = cm.re0 + r0;
    const uint32_t re_last  = re_first + ((wk.re_count - r0 < WAVE ? wk.re_count - r0 : WAVE) - 1u);
    if (re_first >= cur_end && l_cur + 1u < NRPHY_NSYMB) { // wave-uniform
#pragma unroll 1
      do { // skips symbols without data RE
        ++l_cur;
        cur_start = cur_end;
        cur_end   = pd.sym_re_start[l_cur + 1u];
      } while (re_first >= cur_end && l_cur + 1u < NRPHY_NSYMB);
      cur_arg   = pd.sym_arg[l_cur];
      cur_row   = l_cur * p.grid_nof_subc;
      cur_table = pd.sym_kind[l_cur] == SYM_TABLE;
    }
    if (r >= wk.re_count) {
      continue;
    }
    __builtin_assume(r < (uint32_t)RE_CHUNK + WAVE);
    const uint32_t gbits = x1 ^ x2_bits<QM, L>(sh.lin, g, cm, r);
    // The RE's L symbol bytes (first in the MSB): four layers make them a whole word.
    uint32_t bytes;
    if constexpr (L == 4) {
      bytes = sh.symb[r];
    } else {
      bytes = ext32(sh.symb, 8u * r * L);
    }
    // RE position: OFDM symbol from the per-symbol prefix counts, subcarrier from the symbol's pattern.
    const uint32_t re_pdu = re_first + lane;
    uint32_t       subc, word_offset;
    if (re_last < cur_end) { // wave-uniform: the whole group lies in the current symbol
      subc = cur_arg + (re_pdu - cur_start);
      if (cur_table) {
        subc = (uint32_t)p.re_table[subc];
      }
      word_offset = cur_row + subc;
    } else {
      // The group runs into the next symbol(s): rare (once per symbol and codeblock at most), so each lane walks on by
      // itself -- the scalar formulation of this search kept 27 prefix counts and pattern arguments in scalar registers
      // through the whole loop.
      uint32_t l_sym = l_cur, start = cur_start, end = cur_end;
#pragma unroll 1
      while (re_pdu >= end && l_sym + 1u < NRPHY_NSYMB) {
        ++l_sym;
        start = end;
        end   = pd_global->sym_re_start[l_sym + 1u];
      }
      subc = pd_global->sym_arg[l_sym] + (re_pdu - start);
      if (pd_global->sym_kind[l_sym] == SYM_TABLE) {
        subc = (uint32_t)p.re_table[subc];
      }
      word_offset = l_sym * p.grid_nof_subc + subc;
    }
    // Scrambling (TS 38.211 Section 7.3.1.1), modulation (table lookup), layer mapping + precoding
    // (resource_grid_mapper_impl.cpp:279-437, channel_precoder_avx2.cpp:214-342).
    cf2 x[L];
#pragma unroll
    for (int l = 0; l != L; ++l) {
      const uint32_t raw   = ((bytes >> (24 - 8 * l)) & 0xFFu) >> (8 - QM);
      const uint32_t idx   = raw ^ ((gbits >> (32 - (l + 1) * QM)) & ((1u << QM) - 1u));
      const float2   point = reinterpret_cast<const float2*>(sh.u)[lds_probe_index(idx, 0)];
      x[l]                 = cf2{point.x, point.y};
    }
    if constexpr (P > 0) {
#pragma unroll
      for (int port = 0; port != P; ++port) {
        cf2 acc = cmul_ref_packed_uniform(x[0], wu[port][0]);
#pragma unroll
        for (int l = 1; l != L; ++l) {
          acc += cmul_ref_packed_uniform(x[l], wu[port][l]);
        }
        __builtin_amdgcn_raw_buffer_store_b32(pack_cbf16(acc.x, acc.y), grid_rsrc, (int)(word_offset * 4u),
                                              (int)(port * plane_words * 4u), NRPHY_GRID_STORE_AUX);
      }
    } else {
      const uint32_t nof_ports = pd.nof_ports;
      uint32_t       prg       = subc / pd.prg_size_subc;
      prg                      = prg >= pd.nof_prg ? pd.nof_prg - 1 : prg;
      const float* w = p.weights + pd.weights_offset + 2u * prg * nof_ports * L;
#pragma unroll 1
      for (uint32_t port = 0; port != nof_ports; ++port) {
        cf2 acc = cmul_ref_packed(x[0], cf2{w[2 * port * L], w[2 * port * L + 1]});
#pragma unroll
        for (int l = 1; l != L; ++l) {
          acc += cmul_ref_packed(x[l], cf2{w[2 * (port * L + l)], w[2 * (port * L + l) + 1]});
        }
        __builtin_amdgcn_raw_buffer_store_b32(pack_cbf16(acc.x, acc.y), grid_rsrc, (int)(word_offset * 4u),
                                              (int)(port * plane_words * 4u), NRPHY_GRID_STORE_AUX);
      }
    }
  }
}

template <int QM, int L>
__device__ __forceinline__ void phase_b(const PdschLaunch& p, PduRef pd, const PduDev* __restrict__ pd_global,
                                        const CbWork& wk, const CbShared& sh, const ChunkGeom& g, const ChunkMap& cm,
                                        uint32_t first_gbits, uint32_t lane, uint32_t* __restrict__ d_grid,
                                        uint32_t* __restrict__ d_cw_rm, uint32_t* __restrict__ d_cw_scr)
{
  constexpr uint32_t LQ = QM * L;
  // Codeword taps (parity tests, seam B): a loop of their own, so that the grid loop carries none of this.
  if (d_cw_rm != nullptr || d_cw_scr != nullptr) { // wave-uniform
    const uint64_t cw_bit0 = pd.cw_bit_offset + g.cw_cb + (uint64_t)wk.re_begin * LQ;
    for (uint32_t r = lane; r < wk.re_count; r += WAVE) {
      uint32_t bytes, gbits;
      re_bits<QM, L>(p, sh, g, cm, wk.re_count, r, bytes, gbits);
      uint32_t v_rm = 0;
#pragma unroll
      for (int l = 0; l != L; ++l) {
        const uint32_t raw = ((bytes >> (24 - 8 * l)) & 0xFFu) >> (8 - QM);
        v_rm |= raw << (32 - (l + 1) * QM);
      }
      if (d_cw_rm) {
        or_bits_global(d_cw_rm, cw_bit0 + (uint64_t)r * LQ, v_rm, LQ);
      }
      if (d_cw_scr) {
        or_bits_global(d_cw_scr, cw_bit0 + (uint64_t)r * LQ, v_rm ^ (gbits & topmask(LQ)), LQ);
      }
    }
  }
  if (d_grid == nullptr) {
    return;
  }
  // One copy of the grid loop per port count (wave-uniform): straight-line port code with its weights in registers.
  const uint32_t nof_ports = pd.nof_prg == 1 ? pd.nof_ports : 0u;
  if (nof_ports == 4u) {
    phase_b_grid<QM, L, 4>(p, pd, pd_global, wk, sh, g, cm, first_gbits, lane, d_grid);
  } else if (L <= 3 && nof_ports == 3u) {
    phase_b_grid<QM, (L <= 3 ? L : 1), 3>(p, pd, pd_global, wk, sh, g, cm, first_gbits, lane, d_grid);
  } else if (L <= 2 && nof_ports == 2u) {
    phase_b_grid<QM, (L <= 2 ? L : 1), 2>(p, pd, pd_global, wk, sh, g, cm, first_gbits, lane, d_grid);
  } else if (L == 1 && nof_ports == 1u) {
    phase_b_grid<QM, 1, 1>(p, pd, pd_global, wk, sh, g, cm, first_gbits, lane, d_grid);
  } else {
    phase_b_grid<QM, L, 0>(p, pd, pd_global, wk, sh, g, cm, first_gbits, lane, d_grid); // weights per PRG (or fewer ports than layers)
  }
}

// Stages 3 and 4 of a codeblock wave for one (Qm, layers): this wave's slice of the codeword, rate matching ... RE mapping.
template <int QM, int L>
__device__ __forceinline__ void map_chunk(const PdschLaunch& p, PduRef pd, const PduDev* pd_global, const CbWork& wk,
                                          uint32_t item, const CbShared& sh, uint32_t lane, uint32_t* d_grid,
                                          uint32_t* d_cw_rm, uint32_t* d_cw_scr)
{
  const bool is_long = wk.cb >= pd.n_short;
  ChunkGeom  g;
  g.E     = is_long ? pd.e_long : pd.e_short;
  g.cw_cb = is_long ? pd.n_short * pd.e_short + (wk.cb - pd.n_short) * pd.e_long : wk.cb * pd.e_short;
  g.bit0  = g.cw_cb + wk.re_begin * (uint32_t)(QM * L);
  if (NRPHY_STAGE(p) == 3) {
    return;
  }
  ChunkMap cm;
  cm.re0     = g.cw_cb / (uint32_t)(QM * L) + wk.re_begin;
  cm.word0   = g.bit0 >> 5;
  cm.aligned = (g.bit0 & 31u) == 0;
  // The work item's seed (the 31 x2 words from the one its first bit lies in; prologue) and the x1 words of the chunk's
  // first 64 RE are requested here: their trip to memory rides under rate matching.
  const uint32_t seed_word   = lane < 31u ? p.scr_seed[(size_t)item * 32u + lane] : 0u;
  const uint32_t first_gbits = x1_bits<QM, L>(p, g, cm, wk.re_count, lane);
  {
    const RmIndex rm = rm_index_init(pd);
    if (rm.rank0 + g.E > rm.n_valid) { // wave-uniform: the selection wraps around Ncb
      phase_a<QM, L, true>(p, pd, wk, sh, g.E, lane);
    } else {
      phase_a<QM, L, false>(p, pd, wk, sh, g.E, lane);
    }
  }
  if (NRPHY_STAGE(p) == 4) {
    return;
  }
  NRPHY_WG_TRACE_MARK(3); // rate matched and interleaved
  // The codeblock's LDS is free now: the chunk's x2 words go there, from the word its first bit lies in up to the word a
  // misaligned read of its last bits runs into (the plan sized the region for it).
  static_assert((31u + (uint32_t)RE_CHUNK * 32u + 31u) / 32u + 1u <= GOLD_EXPAND_MAX_WORDS, "a work item's scrambling words");
  gold_expand_seed_wave(sh.lin, seed_word, ((g.bit0 & 31u) + wk.re_count * (uint32_t)(QM * L) + 31u) / 32u + 1u, lane);
  phase_b<QM, L>(p, pd, pd_global, wk, sh, g, cm, first_gbits, lane, d_grid, d_cw_rm, d_cw_scr);
}

template <int QM>
__device__ __forceinline__ void map_chunk_layers(const PdschLaunch& p, PduRef pd, const PduDev* pd_global,
                                                 const CbWork& wk, uint32_t item, const CbShared& sh, uint32_t lane,
                                                 uint32_t* d_grid, uint32_t* d_cw_rm, uint32_t* d_cw_scr)
{
  switch (pd.nof_layers) { // wave-uniform
    case 1:
      map_chunk<QM, 1>(p, pd, pd_global, wk, item, sh, lane, d_grid, d_cw_rm, d_cw_scr);
      break;
    case 2:
      map_chunk<QM, 2>(p, pd, pd_global, wk, item, sh, lane, d_grid, d_cw_rm, d_cw_scr);
      break;
    case 3:
      map_chunk<QM, 3>(p, pd, pd_global, wk, item, sh, lane, d_grid, d_cw_rm, d_cw_scr);
      break;
    default:
      map_chunk<QM, 4>(p, pd, pd_global, wk, item, sh, lane, d_grid, d_cw_rm, d_cw_scr);
      break;
  }
}

// ================================================================================================================
// DM-RS for PDSCH (TS 38.211 Section 7.4.1.1; reference: dmrs_pdsch_processor_impl.cpp:84-262, dmrs_helper.h:44-109,
// resource_grid_mapper_impl.cpp:47-133).  One wavefront per (PDU, DM-RS symbol, 32-PRB chunk).
// ================================================================================================================
// The pilots of the wave's PRBs for one layer count: r(n) = a ((1 - 2 c(2n)) + j (1 - 2 c(2n + 1))) from the symbol's
// sequence (generated by the prologue), CDM weights, precoding, mapping.  W: where the weights are read from
// (constant address space = scalar loads for wideband precoding, global memory per lane otherwise).
// The pilot of one port on the (L + 1) / 2 CDM groups: one cbf16 word per group.
template <int L, typename W>
__device__ __forceinline__ void dmrs_port_words(float dr, float di, bool odd, uint32_t port, W w, uint32_t (&word)[(L + 1) / 2])
{
  // CDM (TS 38.211 Table 7.4.1.1.2-1): w_f = {+1, -1} on odd DM-RS ports flips every other pilot; w_t = +1 for
  // ports 1000-1003.
  const float sr = odd ? -dr : dr, si = odd ? -di : di;
#pragma unroll
  for (int g = 0; g != (L + 1) / 2; ++g) {
    float accr, acci;
    cmul_ref(dr, di, w[2 * (port * L + 2 * g)], w[2 * (port * L + 2 * g) + 1], accr, acci);
    if (2 * g + 1 < L) {
      float pr, pi;
      cmul_ref(sr, si, w[2 * (port * L + 2 * g + 1)], w[2 * (port * L + 2 * g + 1) + 1], pr, pi);
      accr = __fadd_rn(accr, pr);
      acci = __fadd_rn(acci, pi);
    }
    word[g] = pack_cbf16(accr, acci);
  }
}

// Both CDM groups of a pilot position are neighbours in the grid (subcarriers 2k', 2k' + 1, 8-byte aligned): one
// 8-byte store per lane makes the wave's store contiguous instead of every other word.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// nt: non-temporal (the launch's DM-RS and zero-fill stores: see PdschLaunch::extras_nt).
template <int L>
__device__ __forceinline__ void dmrs_store(uint32_t* out, const uint32_t (&word)[(L + 1) / 2], bool zero_other_group, bool nt = false)
{
  if ((L + 1) / 2 == 2 || zero_other_group) { // (zero_other_group is wave-uniform: the reserved, pilot-less neighbour RE is zero)
    const u32x2_t v = {word[0], (L + 1) / 2 == 2 ? word[(L + 1) / 2 - 1] : 0u};
    if (nt) {
      __builtin_nontemporal_store(v, reinterpret_cast<u32x2_t*>(out));
    } else {
      *reinterpret_cast<u32x2_t*>(out) = v;
    }
  } else if (nt) {
    __builtin_nontemporal_store(word[0], out);
  } else {
    out[0] = word[0];
  }
}

template <int L, typename W>
__device__ __forceinline__ void dmrs_precode(float dr, float di, bool odd, uint32_t P, W w, uint32_t* out,
                                             size_t port_stride, bool zero_other_group)
{
#pragma unroll 1
  for (uint32_t port = 0; port != P; ++port) {
    uint32_t word[(L + 1) / 2];
    dmrs_port_words<L>(dr, di, odd, port, w, word);
    dmrs_store<L>(out + port * port_stride, word, zero_other_group);
  }
}

// tab: 64 words of LDS private to the wave.
template <int L>
__device__ __forceinline__ void dmrs_items(const PdschLaunch& p, PduRef pd, const DmrsWork& wk,
                                           uint32_t* __restrict__ d_grid, uint32_t lane, uint32_t* tab)
{
  const uint32_t  P       = pd.nof_ports;
  const uint32_t  ordinal = __popc(pd.dmrs_symbol_mask & ((1u << wk.symbol) - 1u));
  const uint32_t* seq     = p.scr + pd.dmrs_seq_offset + ordinal * pd.dmrs_seq_words;
  const float     a       = pd.dmrs_amplitude;
  const uint32_t  nof_items = (wk.prb_end - wk.prb_begin) * 6u;
  const size_t    port_stride = (size_t)NRPHY_NSYMB * p.grid_nof_subc;
  uint32_t*       row = d_grid + (size_t)pd.grid_index * p.grid_nof_ports * port_stride + (size_t)wk.symbol * p.grid_nof_subc;
  const bool      wideband = pd.nof_prg == 1;
  const bool      zero_other = pd.dmrs_zero_other_group != 0 && p.zero_fill != 0;
  if (wideband) { // wave-uniform
    // With one set of weights a pilot position has eight possible values per port -- r(n) is one of four points and the CDM
    // sign one of two: lane v < 8 computes variant v = c(2n) << 2 | c(2n + 1) << 1 | odd for every port, once per wave, and
    // the items below only look theirs up (the arithmetic of a value is the same wherever it is evaluated).
    if (lane < 8u) {
      const float dr = (lane & 4u) ? -a : a, di = (lane & 2u) ? -a : a;
#pragma unroll 1
      for (uint32_t port = 0; port != P; ++port) {
        uint32_t word[(L + 1) / 2];
        dmrs_port_words<L>(dr, di, (lane & 1u) != 0, port, to_constant(p.weights + pd.dmrs_weights_offset), word);
        tab[(lane * NRPHY_MAX_PORTS + port) * 2u]      = word[0];
        tab[(lane * NRPHY_MAX_PORTS + port) * 2u + 1u] = word[(L + 1) / 2 - 1];
      }
    }
    wave_sync();
  }
  for (uint32_t item = lane; item < nof_items; item += WAVE) {
    const uint32_t prb = wk.prb_begin + item / 6u;
    const uint32_t kp  = item % 6u;
    if (!((pd.prb_mask[prb >> 5] >> (prb & 31u)) & 1u)) {
      continue;
    }
    // Pilot r(n) uses c(2n), c(2n+1); PRB prb holds n = 6 (prb - ref) .. +5: an even bit and its neighbour.
    const uint32_t bit  = 12u * (prb - pd.dmrs_ref_rb) + 2u * kp;
    const uint32_t word = seq[bit >> 5] << (bit & 31u);
    uint32_t*      out  = row + 12u * prb + 2u * kp;
    if (wideband) {
      const uint32_t variant = ((word >> 29) & 6u) | (kp & 1u);
#pragma unroll 1
      for (uint32_t port = 0; port != P; ++port) {
        uint32_t words[(L + 1) / 2];
        words[0] = tab[(variant * NRPHY_MAX_PORTS + port) * 2u];
        if ((L + 1) / 2 == 2) {
          words[(L + 1) / 2 - 1] = tab[(variant * NRPHY_MAX_PORTS + port) * 2u + 1u];
        }
        dmrs_store<L>(out + port * port_stride, words, zero_other, p.extras_nt != 0);
      }
      continue;
    }
    const float dr = (word & 0x80000000u) ? -a : a;
    const float di = (word & 0x40000000u) ? -a : a;
    uint32_t    prg = (12u * prb) / pd.prg_size_subc;
    prg             = prg >= pd.nof_prg ? pd.nof_prg - 1 : prg;
    dmrs_precode<L>(dr, di, (kp & 1u) != 0, P, p.weights + pd.dmrs_weights_offset + 2u * prg * P * L, out, port_stride,
                    zero_other);
  }
}

__device__ __forceinline__ void dmrs_wave(const PdschLaunch& p, uint32_t item_index, uint32_t* __restrict__ d_grid,
                                          uint32_t lane, uint32_t* tab)
{
  const auto*    wkc = to_constant(&p.dmrs_work[item_index]);
  const DmrsWork wk  = {wkc->pdu, wkc->symbol, wkc->prb_begin, wkc->prb_end};
  PduRef         pd  = *to_constant(&p.pdus[wk.pdu]);
  switch (pd.nof_layers) { // wave-uniform
    case 1:
      dmrs_items<1>(p, pd, wk, d_grid, lane, tab);
      break;
    case 2:
      dmrs_items<2>(p, pd, wk, d_grid, lane, tab);
      break;
    case 3:
      dmrs_items<3>(p, pd, wk, d_grid, lane, tab);
      break;
    default:
      dmrs_items<4>(p, pd, wk, d_grid, lane, tab);
      break;
  }
}

// Writes zeros to the grid words of one (grid, port) that no PDU of the plan maps.
__device__ __forceinline__ void zero_wave(const PdschLaunch& p, uint32_t item_index, uint32_t* __restrict__ d_grid, uint32_t lane)
{
  const auto*    wkc  = to_constant(&p.zero_work[item_index]);
  const uint32_t seg_begin = wkc->seg_begin, seg_count = wkc->seg_count, seg_long = wkc->seg_long;
  uint32_t*      base = d_grid + ((size_t)wkc->grid * p.grid_nof_ports + wkc->port) * NRPHY_NSYMB * p.grid_nof_subc;
  // Long runs: the wave clears each one together, 16 bytes per lane and store.  The run descriptors are fetched 64 at a time,
  // one per lane, and handed round with v_readlane: read one by one through the scalar cache, every run began with a trip to
  // memory (a wave's two dozen runs took 30 us, and the zero-fill waves are the tail of their launch).
  static_assert(sizeof(ZeroSeg) == 8, "one run descriptor per lane as two dwords");
  for (uint32_t first = 0; first < seg_long; first += WAVE) {
    const uint32_t n = seg_long - first < WAVE ? seg_long - first : WAVE;
    uint2          d = make_uint2(0u, 0u);
    if (lane < n) {
      d = *reinterpret_cast<const uint2*>(&p.zero_segs[seg_begin + first + lane]);
    }
    for (uint32_t i = 0; i != n; ++i) {
      const uint32_t lo_word = (uint32_t)__builtin_amdgcn_readlane((int)d.x, (int)i); // symbol | k0 << 16
      const uint32_t count   = (uint32_t)__builtin_amdgcn_readlane((int)d.y, (int)i) & 0xFFFFu;
      uint32_t*      row     = base + (size_t)(lo_word & 0xFFFFu) * p.grid_nof_subc + (lo_word >> 16);
      // 16-byte chunks counted from the aligned address at or below the run's first word; the two chunks at the ends may
      // be partial and go word by word.
      const uint32_t off      = (uint32_t)((reinterpret_cast<uintptr_t>(row) >> 2) & 3u);
      uint32_t*      aligned  = row - off;
      const uint32_t n_chunks = (off + count + 3u) >> 2;
      for (uint32_t c = lane; c < n_chunks; c += WAVE) {
        const uint32_t lo = 4u * c;
        if (lo >= off && lo + 4u <= off + count) {
          if (p.extras_nt != 0) {
          __builtin_nontemporal_store(u32x4_t{0u, 0u, 0u, 0u}, reinterpret_cast<u32x4_t*>(aligned + lo));
        } else {
          *reinterpret_cast<uint4*>(aligned + lo) = make_uint4(0u, 0u, 0u, 0u);
        }
        } else {
#pragma unroll
          for (uint32_t w = 0; w != 4; ++w) {
            if (lo + w >= off && lo + w < off + count) {
              aligned[lo + w] = 0u;
            }
          }
        }
      }
    }
  }
  for (uint32_t i = seg_long + lane; i < seg_count; i += WAVE) { // short runs (reserved-RE combs): one lane per run
    const ZeroSeg  sg  = p.zero_segs[seg_begin + i];
    uint32_t*      row = base + (size_t)sg.symbol * p.grid_nof_subc + sg.k0;
    for (uint32_t k = 0; k != sg.count; ++k) {
      row[k] = 0u;
    }
  }
}

// ================================================================================================================
// The codeblock kernels.  Blocks [0, n_work) build codeblocks; the launch may carry two more kinds of wave behind
// them so that their short, latency-bound work overlaps the codeblock waves instead of paying for own launches:
// DM-RS waves and zero-fill waves for the grid words nobody maps.
//
// The plan sorts its work items by (modulation order, layers) -- a "bucket".  A plan with one bucket (every batch of
// like PDUs: the headline workload) or with big buckets runs codeblock_kernel_t<QM, L>, one launch per bucket, whose
// output stage exists once and whose scalar registers hold only what that stage needs; a small mixed plan (one slot
// with PDUs of several modulations) runs the one-launch codeblock_kernel, which selects the output stage per wave.
// ================================================================================================================

// A workgroup is NRPHY_CB_WAVES independent waves, each with its own work unit and its own slice of the dynamic LDS (no
// barrier between them): the dispatcher hands out a quarter of the workgroups for the same waves (an empty launch of the
// headline's 138 k one-wave workgroups alone took 0.035 ms).
#ifndef NRPHY_CB_WAVES
#define NRPHY_CB_WAVES 4
#endif
constexpr uint32_t CB_WAVES = NRPHY_CB_WAVES;

struct CbWave {
  uint32_t  lane;
  uint32_t  wave;  // wave-uniform (scalar register)
  uint32_t* lds;   // this wave's slice of the dynamic LDS
};
__device__ __forceinline__ CbWave cb_wave(const PdschLaunch& p, uint32_t* dyn_lds)
{
  CbWave w;
  w.lane = threadIdx.x % WAVE;
  w.wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
  w.lds  = dyn_lds + w.wave * (p.lds_lin_words + p.lds_u_words);
  return w;
}
__host__ __device__ __forceinline__ uint32_t cb_blocks(uint32_t n_units)
{
  return (n_units + CB_WAVES - 1u) / CB_WAVES;
}

// The launch's extra waves (DM-RS, zero fill) sit in the workgroups behind the codeblock ones; true when this wave
// belongs to one of those (whether or not a unit was left for it).
__device__ __forceinline__ bool extra_wave(const PdschLaunch& p, const CbWave& w, uint32_t* __restrict__ d_grid)
{
  const uint32_t nb = cb_blocks(p.n_work);
  if (blockIdx.x < nb) { // workgroup-uniform
    return false;
  }
  const uint32_t extra = (blockIdx.x - nb) * CB_WAVES + w.wave;
  if (extra < p.n_dmrs_in_launch) {
    dmrs_wave(p, extra, d_grid, w.lane, w.lds);
  } else if (extra - p.n_dmrs_in_launch < p.n_zero_work) {
    zero_wave(p, extra - p.n_dmrs_in_launch, d_grid, w.lane);
  }
  return true;
}

// Workgroups go to the eight XCDs round-robin (block b runs on XCD b % 8).  Give each XCD a contiguous run of work
// items, so that codeblocks which share cache lines -- neighbours in the transport block and in the grid rows --
// meet in one L2 instead of leaving partial lines in two.  Returns the wave's work item, or n_work when none is left.
__device__ __forceinline__ uint32_t xcd_work_item(uint32_t n_work, const CbWave& w)
{
  const uint32_t nb  = cb_blocks(n_work);
  const uint32_t xcd = blockIdx.x & 7u, turn = blockIdx.x >> 3;
  const uint32_t q = nb >> 3, r = nb & 7u;
  const uint32_t item = (xcd * q + (xcd < r ? xcd : r) + turn) * CB_WAVES + w.wave;
  return item < n_work ? item : n_work;
}

// Stages 1 and 2 of a codeblock wave: segmentation + CRC attachment (the graph rows ride along: their loads overlap the
// transport block's), LDPC encoding (only the parity rows that rate matching can reach).  False: a profiling stage stopped the wave.
__device__ __forceinline__ bool codeblock_front(const PdschLaunch& p, PduRef pd, const CbWork& wk, CbShared& sh,
                                                const uint8_t* __restrict__ d_tb, uint32_t lane)
{
  const uint32_t zc = pd.zc, kb = pd.kb;
  if (NRPHY_STAGE(p) == 5 || NRPHY_STAGE(p) == 6) {
    return false;
  }
  const uint32_t total_words = (((kb + pd.nof_rows) * zc + 31u) >> 5) + 2u;
  // The graph rows are requested now and stored where the CRC tables were once the codeblock is built: their trip to memory
  // rides under the segmentation and the CRC instead of standing between the CRC and the encoder.
  GraphRows rows;
  rows.fetch(&p.graphs[pd.graph], pd.nof_rows, lane);
  build_codeblock(pd, wk.cb, reinterpret_cast<const uint32_t*>(d_tb + pd.tb_offset), p.tb_crc_part, p.gold, &sh,
                  total_words, lane, NRPHY_STAGE(p));
  if (NRPHY_STAGE(p) == 1 || NRPHY_STAGE(p) == 7) {
    return false;
  }
  NRPHY_WG_TRACE_MARK(1); // codeblock built (segmentation, CRCs)
  rows.store(pd.nof_rows, sh.graph, lane); // (build_codeblock ends with a wave barrier; ldpc_encode_wave synchronises before it reads)
  ldpc_encode_wave(&p.graphs[pd.graph], sh.graph, kb, zc, pd.nof_rows, sh.lin, sh.u, sh.ldpc, lane);
  NRPHY_WG_TRACE_MARK(2); // encoded
  return NRPHY_STAGE(p) != 2;
}

template <int QM, int L>
__global__ __launch_bounds__(WAVE * CB_WAVES) void codeblock_kernel_t(PdschLaunch p, const uint8_t* __restrict__ d_tb,
                                                              uint32_t* __restrict__ d_grid, uint32_t* __restrict__ d_cw_rm,
                                                              uint32_t* __restrict__ d_cw_scr)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
  const CbWave   w    = cb_wave(p, dyn_lds);
  const uint32_t lane = w.lane;
  CbShared       sh;
  sh.lin   = w.lds;
  sh.u     = w.lds + p.lds_lin_words;
  sh.symb  = sh.u + CB_U_QAM_WORDS;
  sh.graph = sh.u + CB_U_GRAPH_OFFSET;
  sh.ldpc  = reinterpret_cast<LdpcScratch*>(sh.u + LDPC_DBL_WORDS);
  if (NRPHY_STAGE(p) == 10 || extra_wave(p, w, d_grid)) {
    return;
  }
  const uint32_t item = xcd_work_item(p.n_work, w);
  if (item == p.n_work) { // wave-uniform: the last workgroup's spare waves
    return;
  }
  NRPHY_WG_TRACE_MARK(0);
  NRPHY_WG_TRACE_WHERE(3);
  const auto*  wkc = to_constant(&p.work[item]);
  const CbWork wk  = {wkc->pdu, wkc->cb, wkc->re_begin, wkc->re_count};
  PduRef       pd  = *to_constant(&p.pdus[wk.pdu]);
  if (!codeblock_front(p, pd, wk, sh, d_tb, lane)) {
    return;
  }
  map_chunk<QM, L>(p, pd, &p.pdus[wk.pdu], wk, p.work_base + item, sh, lane, d_grid, d_cw_rm, d_cw_scr);
  NRPHY_WG_TRACE_MARK(6);
}

__global__ __launch_bounds__(WAVE * CB_WAVES) void codeblock_kernel(PdschLaunch p, const uint8_t* __restrict__ d_tb,
                                                         uint32_t* __restrict__ d_grid, uint32_t* __restrict__ d_cw_rm,
                                                         uint32_t* __restrict__ d_cw_scr)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
  const CbWave   w    = cb_wave(p, dyn_lds);
  const uint32_t lane = w.lane;
  CbShared       sh;
  sh.lin   = w.lds;
  sh.u     = w.lds + p.lds_lin_words;
  sh.symb  = sh.u + CB_U_QAM_WORDS;
  sh.graph = sh.u + CB_U_GRAPH_OFFSET;
  sh.ldpc  = reinterpret_cast<LdpcScratch*>(sh.u + LDPC_DBL_WORDS);
  if (NRPHY_STAGE(p) == 10 || extra_wave(p, w, d_grid)) {
    return;
  }
  const uint32_t item = xcd_work_item(p.n_work, w);
  if (item == p.n_work) { // wave-uniform: the last workgroup's spare waves
    return;
  }
  const auto*  wkc = to_constant(&p.work[item]);
  const CbWork wk  = {wkc->pdu, wkc->cb, wkc->re_begin, wkc->re_count};
  PduRef       pd  = *to_constant(&p.pdus[wk.pdu]);
  if (!codeblock_front(p, pd, wk, sh, d_tb, lane)) {
    return;
  }
  switch (pd.qm) { // wave-uniform
    case 2:
      map_chunk_layers<2>(p, pd, &p.pdus[wk.pdu], wk, p.work_base + item, sh, lane, d_grid, d_cw_rm, d_cw_scr);
      break;
    case 4:
      map_chunk_layers<4>(p, pd, &p.pdus[wk.pdu], wk, p.work_base + item, sh, lane, d_grid, d_cw_rm, d_cw_scr);
      break;
    case 6:
      map_chunk_layers<6>(p, pd, &p.pdus[wk.pdu], wk, p.work_base + item, sh, lane, d_grid, d_cw_rm, d_cw_scr);
      break;
    default:
      map_chunk_layers<8>(p, pd, &p.pdus[wk.pdu], wk, p.work_base + item, sh, lane, d_grid, d_cw_rm, d_cw_scr);
      break;
  }
}

typedef void (*CodeblockKernel)(PdschLaunch, const uint8_t*, uint32_t*, uint32_t*, uint32_t*);

static CodeblockKernel bucket_kernel(uint32_t bucket)
{
  static const CodeblockKernel table[CB_BUCKETS] = {
      codeblock_kernel_t<2, 1>, codeblock_kernel_t<2, 2>, codeblock_kernel_t<2, 3>, codeblock_kernel_t<2, 4>,
      codeblock_kernel_t<4, 1>, codeblock_kernel_t<4, 2>, codeblock_kernel_t<4, 3>, codeblock_kernel_t<4, 4>,
      codeblock_kernel_t<6, 1>, codeblock_kernel_t<6, 2>, codeblock_kernel_t<6, 3>, codeblock_kernel_t<6, 4>,
      codeblock_kernel_t<8, 1>, codeblock_kernel_t<8, 2>, codeblock_kernel_t<8, 3>, codeblock_kernel_t<8, 4>};
  return table[bucket];
}

// bucket_begin[b] .. bucket_begin[b + 1]: the work items of bucket b = cb_bucket(Qm, layers) (the plan sorts them).
// dispatch: 0 = by plan shape, 1 = always the one-launch mixed kernel, 2 = always one launch per bucket.
bool codeblocks_take_bucket_launches(const PdschLaunch& p, const uint32_t* bucket_begin, int dispatch, uint32_t* nof_buckets)
{
  uint32_t n = 0;
  for (uint32_t b = 0; b != CB_BUCKETS; ++b) {
    n += bucket_begin[b + 1] != bucket_begin[b] ? 1u : 0u;
  }
  *nof_buckets = n;
  return !(dispatch == 1 || (dispatch == 0 && n > 1 && p.n_work < CB_MIXED_MAX_WORK));
}

// streams[0] is the caller's stream; with more than one bucket and n_streams > 1 the bucket launches are dealt to the
// streams in turn, biggest bucket first (the caller has made streams[1 ...] wait for what precedes on streams[0] and
// joins them afterwards): the buckets of a mixed batch then share the device like the waves of one launch instead of
// running one after the other, each with a tail of its own.
hipError_t launch_codeblocks(const PdschLaunch& p, const uint32_t* bucket_begin, int dispatch, const uint8_t* d_tb,
                             uint32_t* d_grid, uint32_t* d_cw_rm, uint32_t* d_cw_scr, const hipStream_t* streams, uint32_t n_streams)
{
  if (p.n_work == 0) {
    return hipSuccess;
  }
  const size_t   lds_bytes = 4u * (size_t)(p.lds_lin_words + p.lds_u_words) * CB_WAVES;
  const uint32_t extras    = d_grid ? p.n_dmrs_in_launch + p.n_zero_work : 0u;
  uint32_t       nof_buckets = 0;
  if (!codeblocks_take_bucket_launches(p, bucket_begin, dispatch, &nof_buckets)) {
    hipLaunchKernelGGL(codeblock_kernel, dim3(cb_blocks(p.n_work) + cb_blocks(extras)), dim3(WAVE * CB_WAVES), lds_bytes, streams[0], p,
                       d_tb, d_grid, d_cw_rm, d_cw_scr);
    return hipGetLastError();
  }
  uint32_t order[CB_BUCKETS], n_order = 0; // non-empty buckets, biggest first
  for (uint32_t b = 0; b != CB_BUCKETS; ++b) {
    if (bucket_begin[b + 1] != bucket_begin[b]) {
      uint32_t k = n_order++;
      for (; k != 0 && bucket_begin[order[k - 1] + 1] - bucket_begin[order[k - 1]] < bucket_begin[b + 1] - bucket_begin[b]; --k) {
        order[k] = order[k - 1];
      }
      order[k] = b;
    }
  }
  for (uint32_t i = 0; i != n_order; ++i) {
    const uint32_t b = order[i], n = bucket_begin[b + 1] - bucket_begin[b];
    PdschLaunch    q = p;
    q.work           = p.work + bucket_begin[b];
    q.work_base      = bucket_begin[b];
    q.n_work         = n;
    if (i != 0) { // the DM-RS and zero-fill waves ride at the end of the biggest bucket's launch
      q.n_dmrs_in_launch = 0;
      q.n_zero_work      = 0;
    }
    const uint32_t blocks = cb_blocks(n) + (i == 0 ? cb_blocks(extras) : 0u);
    hipLaunchKernelGGL(bucket_kernel(b), dim3(blocks), dim3(WAVE * CB_WAVES), lds_bytes, streams[n_streams > 1 ? i % n_streams : 0], q,
                       d_tb, d_grid, d_cw_rm, d_cw_scr);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      return e;
    }
  }
  return hipSuccess;
}

__global__ __launch_bounds__(WAVE) void dmrs_kernel(PdschLaunch p, uint32_t* __restrict__ d_grid)
{
  __shared__ uint32_t tab[8 * NRPHY_MAX_PORTS * 2];
  dmrs_wave(p, blockIdx.x, d_grid, threadIdx.x, tab);
}

hipError_t launch_dmrs(const PdschLaunch& p, uint32_t* d_grid, hipStream_t stream)
{
  if (p.n_dmrs_work == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(dmrs_kernel, dim3(p.n_dmrs_work), dim3(WAVE), 0, stream, p, d_grid);
  return hipGetLastError();
}

// ================================================================================================================
// Stand-alone LDPC encoder: ldpc_encoder::encode for n_cb codeblocks of one (base graph, lifting size).
// ================================================================================================================
__global__ __launch_bounds__(WAVE) void ldpc_encode_kernel(const LiftedGraph* graphs, uint32_t graph, uint32_t kb,
                                                           uint32_t zc, const uint8_t* __restrict__ d_msg,
                                                           uint32_t msg_stride, uint32_t out_bits,
                                                           uint8_t* __restrict__ d_out, uint32_t out_stride)
{
  __shared__ uint32_t    lin[LDPC_LIN_WORDS];
  __shared__ uint32_t    gbuf[LDPC_GRAPH_ROWPTR + MAX_BG_EDGES];
  __shared__ uint32_t    dbl[LDPC_DBL_WORDS];
  __shared__ LdpcScratch scratch;
  const uint32_t         lane = threadIdx.x;
  const uint8_t*         msg  = d_msg + (size_t)blockIdx.x * msg_stride;
  uint8_t*               out  = d_out + (size_t)blockIdx.x * out_stride;

  // Codeblock length the encoder has to produce (ldpc_encoder_impl.cpp:63-72).
  const uint32_t K   = kb * zc;
  uint32_t       len = out_bits + 2u * zc;
  len                = len < (kb + 4u) * zc ? (kb + 4u) * zc : len;
  const uint32_t nof_rows    = (len + zc - 1u) / zc - kb;
  const uint32_t total_words = (((kb + nof_rows) * zc + 31u) >> 5) + 2u;
  const uint32_t msg_bytes   = (K + 7u) >> 3;
  for (uint32_t j = lane; j < total_words; j += WAVE) {
    uint32_t v = 0;
    for (uint32_t k = 0; k != 4; ++k) {
      uint32_t byte = 4u * j + k;
      v             = (v << 8) | (byte < msg_bytes ? (uint32_t)msg[byte] : 0u);
    }
    uint32_t pos = 32u * j;
    v            = pos >= K ? 0u : (K - pos < 32u ? v & topmask(K - pos) : v);
    lin[j]       = v;
  }
  stage_graph(&graphs[graph], nof_rows, gbuf, lane);
  wave_sync();
  ldpc_encode_wave(&graphs[graph], gbuf, kb, zc, nof_rows, lin, dbl, &scratch, lane);
  const uint32_t out_bytes = (out_bits + 7u) >> 3;
  for (uint32_t j = lane; 4u * j < out_bytes; j += WAVE) {
    uint32_t v   = ext32(lin, 2u * zc + 32u * j);
    uint32_t pos = 32u * j;
    if (out_bits - pos < 32u) {
      v &= topmask(out_bits - pos);
    }
    for (uint32_t k = 0; k != 4 && 4u * j + k < out_bytes; ++k) {
      out[4u * j + k] = (uint8_t)(v >> (24 - 8 * k));
    }
  }
}

hipError_t launch_ldpc_encode(const LiftedGraph* graphs, uint32_t graph, uint32_t kb, uint32_t zc, uint32_t n_cb,
                              const uint8_t* d_msg, uint32_t msg_stride, uint32_t out_bits, uint8_t* d_out,
                              uint32_t out_stride, hipStream_t stream)
{
  if (n_cb == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(ldpc_encode_kernel, dim3(n_cb), dim3(WAVE), 0, stream, graphs, graph, kb, zc, d_msg, msg_stride,
                     out_bits, d_out, out_stride);
  return hipGetLastError();
}

} // namespace nrphy
