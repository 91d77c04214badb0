// LDPC base-graph expansion for one codeblock held in LDS, executed by one wavefront.
//
// Replaces ldpc_encoder_impl::encode and the generic/AVX2 kernels behind it
// (R/lib/phy/upper/channel_coding/ldpc/ldpc_encoder_impl.cpp:44-81, ldpc_encoder_generic.cpp:58-230,
// ldpc_encoder_avx2.cpp:144-469).  The reference keeps one byte per bit and rotates by copying; here the codeblock
// is a packed MSB-first bit array `lin` in LDS (variable node n occupies bits [n*Zc, (n+1)*Zc)), a lane owns one
// 32-bit word of one check row, and a circulant rotation is two LDS reads and a funnel shift.
#pragma once

#include "bits_device.h"

namespace nrphy {

constexpr int LDPC_MAX_WPB   = 12;                      // words per Zc-bit block, Zc <= 384
constexpr int LDPC_LIN_WORDS = (68 * 384) / 32 + 2;     // whole BG1 codeblock + read-ahead padding
constexpr int LDPC_DBL_WORDS = 2 * 22 * LDPC_MAX_WPB;   // the systematic blocks, doubled

// Word j of block `base` (bit offset of a Zc-bit block) rotated by s: output bit t of the word is block bit
// (s + 32j + t) mod Zc; bits at or beyond Zc are zero.  Needs 0 <= s < Zc and 32j < Zc.
// ALIGNED (Zc a multiple of 32, which covers every large lifting size): whole words, the wrap is a word index wrap.
// Otherwise one wrap at most, because a word never holds more than Zc valid bits.
template <bool ALIGNED>
__device__ __forceinline__ uint32_t rot_word(const uint32_t* a, uint32_t base, uint32_t zc, uint32_t s, uint32_t j)
{
  uint32_t start = s + 32u * j;
  start          = start >= zc ? start - zc : start;
  if (ALIGNED) {
    // Both words are always read and combined without a branch (the shift differs from lane to lane; a guarded
    // second read would be a divergent branch per edge).
    const uint32_t wpb = zc >> 5, bw = base >> 5;
    const uint32_t w0 = start >> 5, sh = s & 31u;
    const uint32_t w1 = (w0 + 1u == wpb) ? 0u : w0 + 1u;
    return (a[bw + w0] << sh) | ((a[bw + w1] >> 1) >> (31u - sh));
  }
  uint32_t n_valid = zc - 32u * j;
  n_valid          = n_valid > 32u ? 32u : n_valid;
  uint32_t first   = zc - start;
  first            = first > n_valid ? n_valid : first;
  uint32_t v       = ext32(a, base + start) & topmask(first);
  if (first < n_valid) {
    v |= (ext32(a, base) & topmask(n_valid - first)) >> first;
  }
  return v;
}

struct LdpcScratch {
  uint32_t aux[4][LDPC_MAX_WPB + 1]; // systematic part of the four core check rows
  uint32_t sum[LDPC_MAX_WPB + 2];    // aux0^aux1^aux2^aux3 (+ read-ahead padding)
  uint32_t p0[LDPC_MAX_WPB + 2];     // first core parity block
};

// Stores word j of a parity block at variable node `node` of lin.  Word-aligned blocks are plain stores; other
// lifting sizes merge with LDS atomics because neighbouring lanes share words (the parity region is zeroed first).
template <bool ALIGNED>
__device__ __forceinline__ void put_block_word(uint32_t* lin, uint32_t node, uint32_t zc, uint32_t j, uint32_t v)
{
  uint32_t pos = node * zc + 32u * j;
  if (ALIGNED) {
    lin[pos >> 5] = v;
  } else {
    uint32_t n_valid = zc - 32u * j;
    or_bits_lds(lin, pos, v, n_valid > 32u ? 32u : n_valid);
  }
}

// The rows of the lifted graph a codeblock needs, staged in LDS: gbuf[0 .. LDPC_GRAPH_ROWPTR) = row_ptr,
// gbuf[LDPC_GRAPH_ROWPTR ..) = packed edges.  (Read from global memory inside the edge loop, every edge would cost a
// dependent L2 round trip.)
constexpr uint32_t LDPC_GRAPH_ROWPTR = 48;

__device__ __forceinline__ void stage_graph(const LiftedGraph* g, uint32_t nof_rows, uint32_t* gbuf, uint32_t lane)
{
  if (lane <= nof_rows) {
    gbuf[lane] = g->row_ptr[lane];
  }
  const uint32_t nedges = g->row_ptr[nof_rows];
  for (uint32_t e = lane; e < nedges; e += WAVE) {
    gbuf[LDPC_GRAPH_ROWPTR + e] = g->edge[e];
  }
}

// The same in two halves, so that the trip to memory can be started long before the LDS region is free: fetch() requests the
// rows into registers, store() puts them into LDS.
struct GraphRows {
  static constexpr uint32_t TRIPS = (MAX_BG_EDGES + WAVE - 1) / WAVE; // 5
  uint32_t row_ptr, nedges, edge[TRIPS];
  __device__ __forceinline__ void fetch(const LiftedGraph* g, uint32_t nof_rows, uint32_t lane)
  {
    row_ptr = lane <= nof_rows ? g->row_ptr[lane] : 0u;
    nedges  = g->row_ptr[nof_rows];
#pragma unroll
    for (uint32_t k = 0; k != TRIPS; ++k) {
      edge[k] = lane + WAVE * k < nedges ? g->edge[lane + WAVE * k] : 0u;
    }
  }
  __device__ __forceinline__ void store(uint32_t nof_rows, uint32_t* gbuf, uint32_t lane) const
  {
    if (lane <= nof_rows) {
      gbuf[lane] = row_ptr;
    }
#pragma unroll
    for (uint32_t k = 0; k != TRIPS; ++k) {
      if (lane + WAVE * k < nedges) {
        gbuf[LDPC_GRAPH_ROWPTR + lane + WAVE * k] = edge[k];
      }
    }
  }
};

// XOR over the edges of check row m of the rotated blocks, word j.
template <bool ALIGNED>
__device__ __forceinline__ uint32_t row_word(const uint32_t* gbuf, const uint32_t* lin, uint32_t zc, uint32_t m,
                                             uint32_t j)
{
  uint32_t acc = 0;
  for (uint32_t e = gbuf[m], end = gbuf[m + 1]; e != end; ++e) {
    uint32_t edge = gbuf[LDPC_GRAPH_ROWPTR + e];
    acc ^= rot_word<ALIGNED>(lin, edge >> 16, zc, edge & 0xFFFFu, j);
  }
  return acc;
}

// Core rows of a word-aligned lifting size (Zc a multiple of 32: every large one) read the systematic blocks from a
// DOUBLED copy -- block n as its Zc / 32 words twice in a row at dbl[2 (Zc / 32) n] -- so that a rotated word never wraps:
// the edge descriptor is (byte offset of word q' of the doubled block) << 16 | s', and word j of the rotated block is
// alignbit(dbl[q' + j], dbl[q' + j + 1], s')  with  s' = (32 - shift % 32) % 32, q' = shift / 32, one less (mod Zc / 32) when
// shift % 32 = 0 -- one address addition, one two-word LDS read and one funnel shift per edge instead of two wrapped
// indices, two reads and two shifts (the lifted graph carries these descriptors for its four core rows, see
// build_lifted_graph).  The reference's AVX2 encoder rotates through a doubled buffer too (ldpc_encoder_avx2.cpp:206-256).
__device__ __forceinline__ uint32_t core_row_word_dbl(const uint32_t* gbuf, const uint32_t* dbl, uint32_t m, uint32_t j)
{
  uint32_t       acc = 0;
  const uint8_t* at  = reinterpret_cast<const uint8_t*>(dbl) + 4u * j;
  uint32_t       e = gbuf[m], end = gbuf[m + 1];
  // Four edges per trip: their descriptors, then their block words, are requested together -- one LDS round trip each per
  // four edges instead of two per edge (a core row has 15 ... 19 systematic edges).
  for (; e + 4u <= end; e += 4u) {
    uint32_t        edge[4];
    u32x2_unaligned w[4];
#pragma unroll
    for (int k = 0; k != 4; ++k) {
      edge[k] = gbuf[LDPC_GRAPH_ROWPTR + e + k];
    }
#pragma unroll
    for (int k = 0; k != 4; ++k) {
      w[k] = *reinterpret_cast<const u32x2_unaligned*>(at + (edge[k] >> 16));
    }
#pragma unroll
    for (int k = 0; k != 4; ++k) {
      acc ^= __builtin_amdgcn_alignbit(w[k].x, w[k].y, edge[k]);
    }
  }
  for (; e != end; ++e) {
    const uint32_t        edge = gbuf[LDPC_GRAPH_ROWPTR + e];
    const u32x2_unaligned w    = *reinterpret_cast<const u32x2_unaligned*>(at + (edge >> 16));
    acc ^= __builtin_amdgcn_alignbit(w.x, w.y, edge);
  }
  return acc;
}

// Fills the doubled copy of the Kb systematic blocks (word-aligned lifting sizes; lin holds them back to back).
__device__ __forceinline__ void ldpc_double_blocks(const uint32_t* lin, uint32_t kb, uint32_t wpb, uint32_t* dbl, uint32_t lane)
{
  const uint32_t magic = (65536u + wpb - 1u) / wpb; // item / wpb = item * magic >> 16 for item < 22 * 12 + 64
  for (uint32_t item = lane; item < kb * wpb; item += WAVE) {
    const uint32_t n = __umul24(item, magic) >> 16; // (24-bit multiplies: full rate, v_mul_lo_u32 is quarter rate)
    const uint32_t v = lin[item];
    dbl[item + __umul24(n, wpb)]       = v;
    dbl[item + __umul24(n, wpb) + wpb] = v;
  }
}

// Computes parity blocks Kb .. Kb + nof_rows - 1 of the codeblock whose Kb systematic blocks are in lin.
// lin words from ceil(Kb*Zc/32) on must be zero on entry.  All 64 lanes of the wave call this.
template <bool ALIGNED>
__device__ inline void ldpc_encode_wave_impl(const LiftedGraph* g, const uint32_t* gbuf, uint32_t kb, uint32_t zc,
                                             uint32_t nof_rows, uint32_t* lin, uint32_t* dbl, LdpcScratch* sc, uint32_t lane)
{
  const uint32_t wpb = (zc + 31u) >> 5;
  // Core rows 0..3: XOR of the rotated systematic blocks (TS 38.212 Section 5.3.2, H restricted to columns < Kb).
  if (ALIGNED) {
    ldpc_double_blocks(lin, kb, wpb, dbl, lane);
    wave_sync();
  }
  const uint32_t wpb_magic = (65536u + wpb - 1u) / wpb; // item / wpb for item < 4 * 12 + 64 (wpb <= 12)
  for (uint32_t item = lane; item < 4u * wpb; item += WAVE) {
    const uint32_t m = __umul24(item, wpb_magic) >> 16, j = item - __umul24(m, wpb);
    sc->aux[m][j] = ALIGNED ? core_row_word_dbl(gbuf, dbl, m, j) : row_word<ALIGNED>(gbuf, lin, zc, m, j);
  }
  wave_sync();
  if (lane < wpb) {
    sc->sum[lane] = sc->aux[0][lane] ^ sc->aux[1][lane] ^ sc->aux[2][lane] ^ sc->aux[3][lane];
  }
  if (lane == 0) {
    sc->sum[wpb] = 0;
    sc->p0[wpb]  = 0;
  }
  wave_sync();
  // Adding the four core rows cancels the dual diagonal: P^b p0 = sum  =>  p0 = P^(-b) sum.
  uint32_t p0w = 0;
  if (lane < wpb) {
    uint32_t b = g->core_b;
    p0w        = rot_word<ALIGNED>(sc->sum, 0, zc, b == 0 ? 0 : zc - b, lane);
    sc->p0[lane] = p0w;
  }
  wave_sync();
  if (lane < wpb) {
    // Row 0: aux0 + P^s0 p0 + p1 = 0.  Row 3: aux3 + P^s3 p0 + p3 = 0.  The core row without an edge in column Kb
    // closes the chain: BG1 row 2: aux2 + p2 + p3 = 0;  BG2 row 1: aux1 + p1 + p2 = 0.
    uint32_t p1 = sc->aux[0][lane] ^ rot_word<ALIGNED>(sc->p0, 0, zc, g->core_s0, lane);
    uint32_t p3 = sc->aux[3][lane] ^ rot_word<ALIGNED>(sc->p0, 0, zc, g->core_s3, lane);
    uint32_t p2 = (g->core_mid == 1) ? (sc->aux[2][lane] ^ p3) : (sc->aux[1][lane] ^ p1);
    put_block_word<ALIGNED>(lin, kb + 0, zc, lane, p0w);
    put_block_word<ALIGNED>(lin, kb + 1, zc, lane, p1);
    put_block_word<ALIGNED>(lin, kb + 2, zc, lane, p2);
    put_block_word<ALIGNED>(lin, kb + 3, zc, lane, p3);
  }
  wave_sync();
  // Extension rows: the parity bit is the XOR of every other entry of its check row (columns < Kb + 4).
  uint32_t ext_items = (nof_rows > 4u ? nof_rows - 4u : 0u) * wpb;
  for (uint32_t item = lane; item < ext_items; item += WAVE) {
    uint32_t m = item / wpb, j = item - m * wpb;
    m += 4u;
    put_block_word<ALIGNED>(lin, kb + m, zc, j, row_word<ALIGNED>(gbuf, lin, zc, m, j));
  }
  wave_sync();
}

// gbuf: LDS, LDPC_GRAPH_ROWPTR + (edges of rows < nof_rows) words, already filled by stage_graph() and synchronised.
// dbl: LDS, 2 * Kb * Zc / 32 words of scratch (used when Zc is a multiple of 32).
__device__ inline void ldpc_encode_wave(const LiftedGraph* g, const uint32_t* gbuf, uint32_t kb, uint32_t zc,
                                        uint32_t nof_rows, uint32_t* lin, uint32_t* dbl, LdpcScratch* sc, uint32_t lane)
{
  if ((zc & 31u) == 0) { // wave-uniform
    ldpc_encode_wave_impl<true>(g, gbuf, kb, zc, nof_rows, lin, dbl, sc, lane);
  } else {
    ldpc_encode_wave_impl<false>(g, gbuf, kb, zc, nof_rows, lin, dbl, sc, lane);
  }
}

} // namespace nrphy
