// PDSCH processor kernels for gfx950 (MI355X).
//
//   tb_crc_kernel      transport-block CRC: every thread reduces a 4-byte-aligned chunk with a byte table, scales
//                      its remainder by x^(8 * bytes after) (table built at plan time) and XORs it into the result
//   codeblock_kernel   one wavefront per codeblock (or per 512-RE chunk of it): segmentation, CB-CRC, LDPC
//                      base-graph expansion in LDS, rate matching + bit interleaving by index arithmetic, Gold
//                      scrambling, QAM mapping, layer mapping, precoding and RE mapping straight into the grid
//   dmrs_kernel        PDSCH DM-RS generation, CDM, precoding and mapping, one wavefront per 32 PRBs of a symbol
//   ldpc_encode_kernel the LDPC encoder alone (seam B / unit parity)
//
// Together they replace pdsch_processor_impl::process (R/lib/phy/upper/channel_processors/pdsch_processor_impl.cpp:30-184)
// in its per-codeblock form (pdsch_processor_concurrent_impl.cpp:55-338, pdsch_codeblock_processor.cpp:27-141).
#include "ldpc_device.h"

namespace nrphy {

// ================================================================================================================
// Transport block CRC (TS 38.212 Section 5.1; reference: ldpc_segmenter_impl.cpp:126, crc_calculator_lut_impl.cpp).
// CRC(M) = sum_t CRC(chunk_t) * x^(8 * bytes after chunk_t) mod g(x): the chunks are independent, the combine is an
// XOR, so a transport block is spread over as many 256-thread workgroups as it takes to give each thread ~64 bytes.
// ================================================================================================================
constexpr int TB_CRC_THREADS = 256;

__global__ __launch_bounds__(TB_CRC_THREADS) void tb_crc_kernel(PdschLaunch p, const uint8_t* __restrict__ d_tb)
{
  __shared__ uint32_t table[256];
  __shared__ uint32_t partial[TB_CRC_THREADS / WAVE];

  const CrcWork   wk  = p.crc_work[blockIdx.x];
  const PduDev&   pd  = p.pdus[wk.pdu];
  const uint32_t  tid = threadIdx.x;
  const CrcPoly   c   = (pd.tb_crc_bits == 16) ? crc16() : crc24a();
  const uint32_t  n   = pd.tb_bytes;
  const uint32_t* w   = reinterpret_cast<const uint32_t*>(d_tb + pd.tb_offset);

  table[tid] = crc_table_entry(tid, c);
  __syncthreads();

  const uint32_t g     = wk.thread_begin + tid;
  const uint32_t begin = g * wk.chunk;
  uint32_t       reg   = 0;
  if (begin < n) {
    const uint32_t len   = (n - begin < wk.chunk) ? n - begin : wk.chunk;
    const uint32_t nfull = len >> 2;
    for (uint32_t i = 0; i != nfull; ++i) {
      reg = crc_update_word(reg, be_word(w, (begin >> 2) + i), table, c);
    }
    const uint32_t tail = len & 3u;
    if (tail) {
      // The last word is only partially inside the transport block: shift its bytes in one by one.
      const uint32_t word = be_word(w, (begin >> 2) + nfull);
      const uint32_t mask = (1u << c.order) - 1u, sh = c.order - 8u;
      for (uint32_t k = 0; k != tail; ++k) {
        uint32_t byte = (word >> (24 - 8 * k)) & 0xFFu;
        uint32_t idx  = ((reg >> sh) ^ byte) & 0xFFu;
        reg           = ((reg << 8) & mask) ^ table[idx];
      }
    }
    reg = crc_mulmod(reg, p.crc_pow[wk.pow_offset + g], c);
  }
  reg = wave_xor(reg);
  if ((tid & (WAVE - 1)) == 0) {
    partial[tid / WAVE] = reg;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t crc = 0;
    for (int i = 0; i != TB_CRC_THREADS / WAVE; ++i) {
      crc ^= partial[i];
    }
    atomicXor(&p.tb_crc[wk.pdu], crc); // tb_crc is zeroed by the run before this kernel
  }
}

hipError_t launch_tb_crc(const PdschLaunch& p, const uint8_t* d_tb, hipStream_t stream)
{
  if (p.n_crc_work == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(tb_crc_kernel, dim3(p.n_crc_work), dim3(TB_CRC_THREADS), 0, stream, p, d_tb);
  return hipGetLastError();
}

// ================================================================================================================
// Codeblock construction in LDS (TS 38.212 Section 5.2.2; reference: ldpc_segmenter_impl.cpp:148-217).
// ================================================================================================================
struct CbShared {
  uint32_t    lin[LDPC_LIN_WORDS];
  LdpcScratch ldpc;
  uint32_t    crc_table[256];
  uint32_t    gold[RE_CHUNK + 40]; // scrambling words of the chunk (+ misalignment, + read-ahead)
  float       w[2 * NRPHY_MAX_PORTS * NRPHY_MAX_LAYERS]; // wideband precoding weights [port][layer] (re, im)
};

// Fills lin with the K bits of codeblock `cb` (payload, TB CRC + zero padding on the last codeblock, CB CRC, filler
// zeros) and zeroes the parity region up to `total_words`.
__device__ inline void build_codeblock(const PduDev& pd, uint32_t cb, const uint32_t* tbw, const uint32_t* tb_crc_ptr,
                                       const GoldTables* tables, CbShared* sh, uint32_t total_words, uint32_t lane)
{
  const bool     last    = (cb == pd.C - 1);
  const uint32_t used    = pd.info_bits - (last ? pd.tb_crc_bits + pd.zero_pad : 0u);
  const uint32_t tb_pos  = cb * pd.info_bits;
  const uint32_t tb_bits = pd.tb_bytes * 8u;
  for (uint32_t j = lane; j < total_words; j += WAVE) {
    uint32_t pos = 32u * j, v = 0;
    if (pos < used) {
      // Never touch words that lie entirely beyond the transport block.
      uint32_t abs_pos = tb_pos + pos;
      uint32_t i = abs_pos >> 5, s = abs_pos & 31u;
      uint32_t hi = be_word(tbw, i);
      uint32_t lo = (s != 0 && 32u * (i + 1) < tb_bits) ? be_word(tbw, i + 1) : 0u;
      v           = __funnelshift_l(lo, hi, s);
      uint32_t remaining = used - pos;
      if (remaining < 32u) {
        v &= topmask(remaining);
      }
    }
    sh->lin[j] = v;
  }
  if (pd.cb_crc_bits) {
#pragma unroll
    for (int k = 0; k != 4; ++k) {
      sh->crc_table[lane + WAVE * k] = tables->crc24b_table[lane + WAVE * k];
    }
  }
  wave_sync();
  if (last && lane == 0) {
    or_bits_lds(sh->lin, used, *tb_crc_ptr << (32u - pd.tb_crc_bits), pd.tb_crc_bits);
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
      reg = crc_update_word(reg, word, sh->crc_table, crc24b());
    }
    if (b < nw && reg != 0) {
      reg = crc_mulmod(reg, tables->crc24b_pow32[nw - b], crc24b());
    }
    uint32_t crc = wave_xor(reg);
    if (lane == 0) {
      or_bits_lds(sh->lin, n, crc << 8, 24);
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

__device__ __forceinline__ RmIndex rm_index_init(const PduDev& pd)
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

// Circular-buffer position of selected bit t.  WRAP = false: the caller knows rank0 + t < n_valid.
template <bool WRAP>
__device__ __forceinline__ uint32_t rm_pos(const RmIndex& r, uint32_t t)
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
  return u < r.fs ? u : u + r.flen;
}

// ================================================================================================================
// Modulation mapper (TS 38.211 Section 5.1; reference: modulation_mapper_lut_impl.cpp:39-65): Qm bits (first bit
// in the MSB of idx) -> un-normalised odd integers, as the reference's ci8 table.
// ================================================================================================================
template <int QM>
__device__ __forceinline__ void qam_map(uint32_t idx, float& re, float& im)
{
  // Even bit positions (from the MSB) drive the real axis, odd positions the imaginary axis:
  // d = (1-2b0)[2^(h-1) - (1-2b2)[2^(h-2) - ...]], evaluated from the innermost term outwards.
  constexpr int h = QM / 2;
  int           r = 1 - 2 * (int)((idx >> 1) & 1u);
  int           q = 1 - 2 * (int)(idx & 1u);
#pragma unroll
  for (int lvl = 1; lvl < h; ++lvl) {
    int br = (int)((idx >> (2 * lvl + 1)) & 1u);
    int bq = (int)((idx >> (2 * lvl)) & 1u);
    r      = (1 - 2 * br) * ((1 << lvl) - r);
    q      = (1 - 2 * bq) * ((1 << lvl) - q);
  }
  re = (float)r;
  im = (float)q;
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

struct ChunkGeom {
  uint32_t E;      // rate-matched length of the codeblock
  uint32_t cw_cb;  // first codeword bit of the codeblock
  uint32_t gmis;   // misalignment of the chunk's first bit within its first scrambling word
};

// ================================================================================================================
// Output stage of the codeblock kernel for one (modulation order, layer count): rate matching, interleaving,
// scrambling, modulation, layer mapping, precoding, RE mapping.  Compile-time QM and L unroll every inner loop.
// ================================================================================================================
template <int QM, int L, bool WRAP>
__device__ __forceinline__ void map_chunk(const PdschLaunch& p, const PduDev& pd, const CbWork& wk, const CbShared& sh,
                                          const ChunkGeom& g, uint32_t lane, uint32_t* __restrict__ d_grid,
                                          uint32_t* __restrict__ d_cw_rm, uint32_t* __restrict__ d_cw_scr)
{
  constexpr uint32_t LQ    = QM * L;
  const uint32_t     esym  = g.E / QM;           // rows of the bit interleaver
  const uint32_t     zc2   = 2u * pd.zc;
  const RmIndex      rm    = rm_index_init(pd);
  const uint32_t     re_cb = g.cw_cb / LQ;       // first RE of the codeblock within the PDU
  const uint32_t     P     = pd.nof_ports;
  const bool         one_prg = pd.nof_prg == 1;
  const float*       wbase = p.weights + pd.weights_offset;
  const size_t       grid_base = (size_t)pd.grid_index * p.grid_nof_ports * NRPHY_NSYMB * p.grid_nof_subc;
  const uint64_t     cw_bit0 = pd.cw_bit_offset + g.cw_cb + (uint64_t)wk.re_begin * LQ;

  bool any_table = false;
#pragma unroll
  for (int l = 0; l != NRPHY_NSYMB; ++l) {
    any_table |= pd.sym_kind[l] == SYM_TABLE;
  }

  for (uint32_t r = lane; r < wk.re_count; r += WAVE) {
    const uint32_t re_in_cb = wk.re_begin + r;
    const uint32_t sym0     = re_in_cb * L; // first modulation symbol of the RE within the codeblock
    // Rate matching + bit interleaving: bit j of symbol s is selected bit j*esym + s; the L symbols of an RE are
    // consecutive, so for each j one funnel read yields the bit of every layer unless the run crosses the filler
    // gap or the end of the circular buffer.
    uint32_t v = 0; // the RE's L*Qm codeword bits, first bit in the MSB
#pragma unroll
    for (int j = 0; j != QM; ++j) {
      const uint32_t t     = (uint32_t)j * esym + sym0;
      const uint32_t first = rm_pos<WRAP>(rm, t);
      uint32_t       bits;
      if (L == 1 || rm_pos<WRAP>(rm, t + L - 1) == first + (L - 1)) {
        bits = ext32(sh.lin, first + zc2);
      } else {
        bits = 0;
#pragma unroll
        for (int l = 0; l != L; ++l) {
          uint32_t pl = rm_pos<WRAP>(rm, t + l) + zc2;
          bits |= ((sh.lin[pl >> 5] >> (31u - (pl & 31u))) & 1u) << (31 - l);
        }
      }
#pragma unroll
      for (int l = 0; l != L; ++l) { // layer l's bit goes to position l*Qm + j of the RE's bit group
        v |= ((bits >> (31 - l)) & 1u) << (31 - (l * QM + j));
      }
    }
    if (d_cw_rm) {
      or_bits_global(d_cw_rm, cw_bit0 + (uint64_t)r * LQ, v, LQ);
    }
    // Scrambling (TS 38.211 Section 7.3.1.1).
    v ^= ext32(sh.gold, g.gmis + r * LQ) & topmask(LQ);
    if (d_cw_scr) {
      or_bits_global(d_cw_scr, cw_bit0 + (uint64_t)r * LQ, v, LQ);
    }
    if (d_grid == nullptr) {
      continue;
    }
    // RE position: OFDM symbol from the per-symbol prefix counts, subcarrier from the symbol's pattern.
    const uint32_t re_pdu = re_cb + re_in_cb;
    uint32_t       l_sym = 0, start = 0, arg = pd.sym_arg[0];
#pragma unroll
    for (int l = 1; l != NRPHY_NSYMB; ++l) {
      bool ge = re_pdu >= pd.sym_re_start[l];
      l_sym += ge ? 1u : 0u;
      start = ge ? pd.sym_re_start[l] : start;
      arg   = ge ? pd.sym_arg[l] : arg;
    }
    uint32_t subc = arg + (re_pdu - start);
    if (any_table && pd.sym_kind[l_sym] == SYM_TABLE) {
      subc = (uint32_t)p.re_table[subc];
    }
    // Modulation + layer mapping + precoding (resource_grid_mapper_impl.cpp:279-437, channel_precoder_avx2.cpp:214-342).
    float xr[L], xi[L];
#pragma unroll
    for (int l = 0; l != L; ++l) {
      qam_map<QM>((v >> (32 - (l + 1) * QM)) & ((1u << QM) - 1u), xr[l], xi[l]);
    }
    uint32_t* out = d_grid + grid_base + (size_t)l_sym * p.grid_nof_subc + subc;
    if (one_prg) {
      // Wideband precoding (the common case): the weights sit in LDS, every lane reads the same words (broadcast).
#pragma unroll 1
      for (uint32_t port = 0; port != P; ++port) {
        const float* w = &sh.w[2 * port * L];
        float        accr, acci;
        cmul_ref(xr[0], xi[0], w[0], w[1], accr, acci);
#pragma unroll
        for (int l = 1; l != L; ++l) {
          float pr, pi;
          cmul_ref(xr[l], xi[l], w[2 * l], w[2 * l + 1], pr, pi);
          accr = __fadd_rn(accr, pr);
          acci = __fadd_rn(acci, pi);
        }
        out[(size_t)port * NRPHY_NSYMB * p.grid_nof_subc] = to_bf16_bits(accr) | (to_bf16_bits(acci) << 16);
      }
    } else {
      uint32_t prg = subc / pd.prg_size_subc;
      prg          = prg >= pd.nof_prg ? pd.nof_prg - 1 : prg;
      const float* w = wbase + 2u * prg * P * L;
      for (uint32_t port = 0; port != P; ++port) {
        float accr, acci;
        cmul_ref(xr[0], xi[0], w[2 * (port * L)], w[2 * (port * L) + 1], accr, acci);
#pragma unroll
        for (int l = 1; l != L; ++l) {
          float pr, pi;
          cmul_ref(xr[l], xi[l], w[2 * (port * L + l)], w[2 * (port * L + l) + 1], pr, pi);
          accr = __fadd_rn(accr, pr);
          acci = __fadd_rn(acci, pi);
        }
        out[(size_t)port * NRPHY_NSYMB * p.grid_nof_subc] = to_bf16_bits(accr) | (to_bf16_bits(acci) << 16);
      }
    }
  }
}

template <int QM, int L>
__device__ __forceinline__ void map_chunk_select(const PdschLaunch& p, const PduDev& pd, const CbWork& wk,
                                                 const CbShared& sh, const ChunkGeom& g, bool wrap, uint32_t lane,
                                                 uint32_t* d_grid, uint32_t* d_cw_rm, uint32_t* d_cw_scr)
{
  if (wrap) {
    map_chunk<QM, L, true>(p, pd, wk, sh, g, lane, d_grid, d_cw_rm, d_cw_scr);
  } else {
    map_chunk<QM, L, false>(p, pd, wk, sh, g, lane, d_grid, d_cw_rm, d_cw_scr);
  }
}

template <int QM>
__device__ __forceinline__ void map_chunk_layers(const PdschLaunch& p, const PduDev& pd, const CbWork& wk,
                                                 const CbShared& sh, const ChunkGeom& g, bool wrap, uint32_t lane,
                                                 uint32_t* d_grid, uint32_t* d_cw_rm, uint32_t* d_cw_scr)
{
  switch (pd.nof_layers) { // wave-uniform
    case 1:
      map_chunk_select<QM, 1>(p, pd, wk, sh, g, wrap, lane, d_grid, d_cw_rm, d_cw_scr);
      break;
    case 2:
      map_chunk_select<QM, 2>(p, pd, wk, sh, g, wrap, lane, d_grid, d_cw_rm, d_cw_scr);
      break;
    case 3:
      map_chunk_select<QM, 3>(p, pd, wk, sh, g, wrap, lane, d_grid, d_cw_rm, d_cw_scr);
      break;
    default:
      map_chunk_select<QM, 4>(p, pd, wk, sh, g, wrap, lane, d_grid, d_cw_rm, d_cw_scr);
      break;
  }
}

// ================================================================================================================
// The codeblock kernel.
// ================================================================================================================
__global__ __launch_bounds__(WAVE) void codeblock_kernel(PdschLaunch p, const uint8_t* __restrict__ d_tb,
                                                         uint32_t* __restrict__ d_grid, uint32_t* __restrict__ d_cw_rm,
                                                         uint32_t* __restrict__ d_cw_scr)
{
  __shared__ CbShared sh;
  const uint32_t      lane = threadIdx.x;
  const CbWork        wk   = p.work[blockIdx.x];
  const PduDev&       pd   = p.pdus[wk.pdu];
  const uint32_t      zc = pd.zc, kb = pd.kb;

  // 1. Segmentation + CRC attachment.
  const uint32_t total_words = (((kb + pd.nof_rows) * zc + 31u) >> 5) + 2u;
  build_codeblock(pd, wk.cb, reinterpret_cast<const uint32_t*>(d_tb + pd.tb_offset), &p.tb_crc[wk.pdu], p.gold, &sh,
                  total_words, lane);

  // 2. LDPC encoding (only the parity rows that rate matching can reach).
  ldpc_encode_wave(&p.graphs[pd.graph], kb, zc, pd.nof_rows, sh.lin, &sh.ldpc, lane);

  // 3. This wave's slice of the codeword and its scrambling sequence.
  const uint32_t lq      = pd.nof_layers * pd.qm; // bits per RE
  const bool     is_long = wk.cb >= pd.n_short;
  ChunkGeom      g;
  g.E     = is_long ? pd.e_long : pd.e_short;
  g.cw_cb = is_long ? pd.n_short * pd.e_short + (wk.cb - pd.n_short) * pd.e_long : wk.cb * pd.e_short;
  const uint32_t bit0   = g.cw_cb + wk.re_begin * lq; // first codeword bit of the chunk
  g.gmis                = bit0 & 31u;
  const uint32_t gwords = (g.gmis + wk.re_count * lq + 31u) >> 5;
  gold_generate_wave(p.gold, p.x1_words, pd.c_init, bit0 >> 5, gwords, sh.gold, lane);
  if (lane < 8) {
    sh.gold[gwords + lane] = 0;
  }
  if (lane < 2 * pd.nof_ports * pd.nof_layers) {
    sh.w[lane] = p.weights[pd.weights_offset + lane];
  }
  wave_sync();

  // 4. Rate matching ... RE mapping, specialised per (Qm, layers); `wrap` = the selection wraps around Ncb.
  const RmIndex rm   = rm_index_init(pd);
  const bool    wrap = rm.rank0 + g.E > rm.n_valid;
  switch (pd.qm) { // wave-uniform
    case 2:
      map_chunk_layers<2>(p, pd, wk, sh, g, wrap, lane, d_grid, d_cw_rm, d_cw_scr);
      break;
    case 4:
      map_chunk_layers<4>(p, pd, wk, sh, g, wrap, lane, d_grid, d_cw_rm, d_cw_scr);
      break;
    case 6:
      map_chunk_layers<6>(p, pd, wk, sh, g, wrap, lane, d_grid, d_cw_rm, d_cw_scr);
      break;
    default:
      map_chunk_layers<8>(p, pd, wk, sh, g, wrap, lane, d_grid, d_cw_rm, d_cw_scr);
      break;
  }
}

hipError_t launch_codeblocks(const PdschLaunch& p, const uint8_t* d_tb, uint32_t* d_grid, uint32_t* d_cw_rm,
                             uint32_t* d_cw_scr, hipStream_t stream)
{
  if (p.n_work == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(codeblock_kernel, dim3(p.n_work), dim3(WAVE), 0, stream, p, d_tb, d_grid, d_cw_rm, d_cw_scr);
  return hipGetLastError();
}

// ================================================================================================================
// DM-RS for PDSCH (TS 38.211 Section 7.4.1.1; reference: dmrs_pdsch_processor_impl.cpp:84-262, dmrs_helper.h:44-109,
// resource_grid_mapper_impl.cpp:47-133).  One wavefront per (PDU, DM-RS symbol, 32-PRB chunk).
// ================================================================================================================
constexpr int DMRS_GOLD_WORDS = (DMRS_PRB_CHUNK * 12) / 32 + 8;

__global__ __launch_bounds__(WAVE) void dmrs_kernel(PdschLaunch p, uint32_t* __restrict__ d_grid)
{
  __shared__ uint32_t gold[DMRS_GOLD_WORDS];
  const uint32_t      lane = threadIdx.x;
  const DmrsWork      wk   = p.dmrs_work[blockIdx.x];
  const PduDev&       pd   = p.pdus[wk.pdu];
  const uint32_t      L = pd.nof_layers, P = pd.nof_ports;

  // Pilot r(n) uses c(2n), c(2n+1); PRB prb holds n = 6*(prb - ref) .. +5, i.e. sequence bits 12*(prb - ref) .. +11.
  const uint32_t bit_first = 12u * (wk.prb_begin - pd.dmrs_ref_rb);
  const uint32_t bit_end   = 12u * (wk.prb_end - pd.dmrs_ref_rb);
  const uint32_t w0        = bit_first >> 5;
  const uint32_t nwords    = ((bit_end + 31u) >> 5) - w0;
  gold_generate_wave(p.gold, p.x1_words, pd.dmrs_c_init[wk.symbol], w0, nwords, gold, lane);

  const float    a          = pd.dmrs_amplitude;
  const uint32_t nof_items  = (wk.prb_end - wk.prb_begin) * 6u;
  const size_t   grid_base  = (size_t)pd.grid_index * p.grid_nof_ports * NRPHY_NSYMB * p.grid_nof_subc;
  const uint32_t nof_groups = (L + 1u) >> 1;
  for (uint32_t item = lane; item < nof_items; item += WAVE) {
    const uint32_t prb = wk.prb_begin + item / 6u;
    const uint32_t kp  = item % 6u;
    if (!((pd.prb_mask[prb >> 5] >> (prb & 31u)) & 1u)) {
      continue;
    }
    const uint32_t bit = 12u * (prb - pd.dmrs_ref_rb) + 2u * kp - 32u * w0;
    const uint32_t c0  = (gold[bit >> 5] >> (31u - (bit & 31u))) & 1u;
    const uint32_t c1  = (gold[(bit + 1u) >> 5] >> (31u - ((bit + 1u) & 31u))) & 1u;
    const float    dr  = c0 ? -a : a;
    const float    di  = c1 ? -a : a;
    const float*   w   = p.weights + pd.dmrs_weights_offset;
    if (pd.nof_prg > 1) {
      uint32_t prg = (12u * prb) / pd.prg_size_subc;
      prg          = prg >= pd.nof_prg ? pd.nof_prg - 1 : prg;
      w += 2u * prg * P * L;
    }
    for (uint32_t group = 0; group != nof_groups; ++group) {
      const uint32_t subc = 12u * prb + group + 2u * kp;
      for (uint32_t port = 0; port != P; ++port) {
        float accr = 0.f, acci = 0.f;
        for (uint32_t j = 2u * group; j != 2u * group + 2u && j != L; ++j) {
          // CDM: w_f = {+1, -1} on odd DM-RS ports flips every other pilot; w_t = +1 for ports 1000-1003.
          const float sign = ((j & 1u) && (kp & 1u)) ? -1.f : 1.f;
          float       pr, pi;
          cmul_ref(sign * dr, sign * di, w[2 * (port * L + j)], w[2 * (port * L + j) + 1], pr, pi);
          if (j == 2u * group) {
            accr = pr;
            acci = pi;
          } else {
            accr = __fadd_rn(accr, pr);
            acci = __fadd_rn(acci, pi);
          }
        }
        d_grid[grid_base + ((size_t)port * NRPHY_NSYMB + wk.symbol) * p.grid_nof_subc + subc] =
            to_bf16_bits(accr) | (to_bf16_bits(acci) << 16);
      }
    }
  }
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
  wave_sync();
  ldpc_encode_wave(&graphs[graph], kb, zc, nof_rows, lin, &scratch, lane);
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
