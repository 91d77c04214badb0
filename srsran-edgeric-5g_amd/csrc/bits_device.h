// Device helpers for MSB-first packed bit arrays held in 32-bit words, CRC arithmetic and the Gold sequence.
// gfx950 only: 64-lane wavefronts are assumed throughout.
#pragma once

#include "nrphy_internal.h"

namespace nrphy {

constexpr int WAVE = 64;

// Workgroup timeline of the prologue launch (profiles/prologue_trace.py; a variant build with -DNRPHY_WG_TRACE, never the
// product): thread 0 of every workgroup records the 100 MHz wall clock at its start (slot 0), at up to five points on
// the way (1-5) and at its end (6), and where it ran (7: HW_ID, XCC_ID).
#ifdef NRPHY_WG_TRACE
constexpr uint32_t WG_TRACE_MAX = 16384;
__device__ uint64_t g_wg_trace[8 * WG_TRACE_MAX];
#define NRPHY_WG_TRACE_MARK(slot)                                              \
  do {                                                                         \
    if (threadIdx.x == 0 && blockIdx.x < WG_TRACE_MAX) {                       \
      g_wg_trace[8 * blockIdx.x + (slot)] = wall_clock64();                    \
    }                                                                          \
  } while (0)
#define NRPHY_WG_TRACE_WHERE(role)                                                                              \
  do {                                                                                                          \
    if (threadIdx.x == 0 && blockIdx.x < WG_TRACE_MAX) {                                                        \
      const uint64_t hw = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20); \
      g_wg_trace[8 * blockIdx.x + 7] = ((uint64_t)(role) << 60) | ((xcc & 0xFu) << 32) | hw;                     \
    }                                                                                                           \
  } while (0)
#else
#define NRPHY_WG_TRACE_MARK(slot) ((void)0)
#define NRPHY_WG_TRACE_WHERE(role) ((void)0)
#endif

// Plan tables (PDU descriptors, work lists) never change while a kernel runs.  Reading them through the constant
// address space tells the compiler so: wave-uniform reads become scalar loads into SGPRs (one s_load_dwordx16 for
// sixteen fields) instead of vector loads + v_readfirstlane with a full memory round trip each.
#define NRPHY_CONSTANT __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const NRPHY_CONSTANT T* to_constant(const T* p)
{
  return (const NRPHY_CONSTANT T*)p;
}
typedef const NRPHY_CONSTANT PduDev& PduRef;

// Orders the LDS traffic among the lanes of ONE wavefront: what a wave that works on LDS of its own needs between a
// phase that writes and a phase that reads.  The LDS executes the instructions of a wave in order, so waiting for the
// wave's outstanding LDS operations (and keeping the compiler from moving memory accesses across) is enough; no s_barrier
// -- the codeblock kernels pack several independent waves into a workgroup -- and no wait for global memory, which
// __syncthreads() would add.
__device__ __forceinline__ void wave_sync()
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// n most significant bits set, n in [0, 32].
__device__ __forceinline__ uint32_t topmask(uint32_t n)
{
  return n == 0 ? 0u : (0xFFFFFFFFu << (32u - n));
}

// 32 bits starting at bit `pos` of the MSB-first bit array `a` (reads words pos/32 and pos/32 + 1).
// Both words come from one unconditional two-dword load (ds_read2_b32 / global_load_dwordx2): written as two scalar
// loads, the compiler guards the second one with a divergent branch on pos % 32 != 0.
typedef uint32_t u32x2_unaligned __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ uint32_t ext32(const uint32_t* a, uint32_t pos)
{
  const u32x2_unaligned v  = *reinterpret_cast<const u32x2_unaligned*>(a + (pos >> 5));
  const uint32_t        sh = pos & 31u;
  return (v.x << sh) | ((v.y >> 1) >> (31u - sh)); // upper word of (v.x : v.y) << sh, sh = 0 included
}

// Big-endian 32-bit load from a byte stream viewed as words: word i = bytes 4i..4i+3, first byte in the MSBs.
__device__ __forceinline__ uint32_t be_word(const uint32_t* words, uint32_t i)
{
  return __builtin_bswap32(words[i]);
}

// ORs the `nbits` most significant bits of `value` into the bit array at bit `pos` (LDS, shared by lanes).
__device__ __forceinline__ void or_bits_lds(uint32_t* a, uint32_t pos, uint32_t value, uint32_t nbits)
{
  value &= topmask(nbits);
  uint32_t w = pos >> 5, sh = pos & 31u;
  uint32_t hi = value >> sh;
  if (hi) {
    atomicOr(&a[w], hi);
  }
  if (sh) {
    uint32_t lo = value << (32u - sh);
    if (lo) {
      atomicOr(&a[w + 1], lo);
    }
  }
}

// The same by ONE lane while no other lane touches the words (plain read-modify-write: an LDS atomic issued by a single
// lane goes through the compiler's wave-level atomic optimiser, some forty instructions of readlane loops).
__device__ __forceinline__ void or_bits_lds_exclusive(uint32_t* a, uint32_t pos, uint32_t value, uint32_t nbits)
{
  value &= topmask(nbits);
  const uint32_t w = pos >> 5, sh = pos & 31u;
  a[w] |= value >> sh;
  a[w + 1] |= (value << 1) << (31u - sh); // sh = 0: nothing (the array has a word of read-ahead)
}

// Same on global memory (codeword taps; the buffer is zeroed by the caller).
__device__ __forceinline__ void or_bits_global(uint32_t* a, uint64_t pos, uint32_t value, uint32_t nbits)
{
  value &= topmask(nbits);
  uint64_t w  = pos >> 5;
  uint32_t sh = (uint32_t)pos & 31u;
  // The tap buffers are byte streams: store words big-endian.
  uint32_t hi = value >> sh;
  if (hi) {
    atomicOr(&a[w], __builtin_bswap32(hi));
  }
  if (sh) {
    uint32_t lo = value << (32u - sh);
    if (lo) {
      atomicOr(&a[w + 1], __builtin_bswap32(lo));
    }
  }
}

// ---- CRC (TS 38.212 Section 5.1): remainder arithmetic in GF(2)[x] / poly, `order` in {16, 24} ---------------
struct CrcPoly {
  uint32_t poly;  // including the leading term
  uint32_t order;
};
__device__ __forceinline__ CrcPoly crc24a()
{
  return {0x1864CFBu, 24u};
}
__device__ __forceinline__ CrcPoly crc24b()
{
  return {0x1800063u, 24u};
}
__device__ __forceinline__ CrcPoly crc16()
{
  return {0x11021u, 16u};
}

// a * b mod poly, operands below 2^order.
__device__ __forceinline__ uint32_t crc_mulmod(uint32_t a, uint32_t b, CrcPoly c)
{
  uint32_t top = 1u << c.order, r = 0;
  for (int i = (int)c.order - 1; i >= 0; --i) {
    r <<= 1;
    if (r & top) {
      r ^= c.poly;
    }
    if ((b >> i) & 1u) {
      r ^= a;
    }
  }
  return r;
}

// a * b mod poly for a below 2^order and any 32-bit polynomial b (Horner over the bits of b).
__device__ __forceinline__ uint32_t crc_mulmod32(uint32_t a, uint32_t b, CrcPoly c)
{
  uint32_t top = 1u << c.order, r = 0;
  for (int i = 31; i >= 0; --i) {
    r <<= 1;
    if (r & top) {
      r ^= c.poly;
    }
    if ((b >> i) & 1u) {
      r ^= a;
    }
  }
  return r;
}

// x^e mod poly.
__device__ __forceinline__ uint32_t crc_xpow(uint32_t e, CrcPoly c)
{
  uint32_t result = 1, base = 2; // x
  while (e) {
    if (e & 1u) {
      result = crc_mulmod(result, base, c);
    }
    base = crc_mulmod(base, base, c);
    e >>= 1;
  }
  return result;
}

// Table entry b: (b(x) * x^order) mod poly; 256 entries in LDS, filled by fill_crc_table().
__device__ __forceinline__ uint32_t crc_table_entry(uint32_t b, CrcPoly c)
{
  uint32_t top = 1u << c.order;
  uint32_t r   = b << (c.order - 8u);
  for (int k = 0; k != 8; ++k) {
    r <<= 1;
    if (r & top) {
      r ^= c.poly;
    }
  }
  return r & (top - 1u);
}

// reg <- remainder after shifting the 4 bytes of `word` (MSB first) through the register.
__device__ __forceinline__ uint32_t crc_update_word(uint32_t reg, uint32_t word, const uint32_t* table, CrcPoly c)
{
  uint32_t mask = (1u << c.order) - 1u, sh = c.order - 8u;
#pragma unroll
  for (int k = 0; k != 4; ++k) {
    uint32_t byte = (word >> (24 - 8 * k)) & 0xFFu;
    uint32_t idx  = ((reg >> sh) ^ byte) & 0xFFu;
    reg           = ((reg << 8) & mask) ^ table[idx];
  }
  return reg;
}

// XOR reduction over the 64 lanes of a wave, all of which must be active; the result is wave-uniform.  Four DPP steps fold
// every row of 16 lanes (pairs, quads, half rows, rows), four v_readlane + scalar XORs fold the rows: 8 vector
// instructions.  (As six __shfl_xor steps it was six ds_bpermute round trips and 36 vector instructions.)
__device__ __forceinline__ uint32_t wave_xor(uint32_t v)
{
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true); // row_half_mirror
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true); // row_mirror
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 16) ^
         (uint32_t)__builtin_amdgcn_readlane((int)v, 32) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

// ---- Gold sequence (TS 38.211 Section 5.2.1) --------------------------------------------------------------
// Generates `nwords` 32-bit words (MSB-first) of c(n) for n in [32*first_word, 32*(first_word + nwords)) into
// the LDS array `out`, executed by one full wavefront.  x1 comes from the precomputed table, x2 from c_init:
//  1. jump the 31-bit x2 state to offset Nc + 32*first_word with the matrices (M2)^(2^k): lane r evaluates row r
//     and the wave assembles the new state with a ballot ("wavefront-ballot parity");
//  2. the first 31 words in parallel, lane w evaluating the 32 parities of word w (they are linear in the state);
//  3. every further word in parallel from W[k] = W[k-28] ^ W[k-29] ^ W[k-30] ^ W[k-31], the x2 recurrence lifted
//     to 32-bit words (squaring the characteristic polynomial five times: p(x)^32 = p(x^32) over GF(2)).
// Replaces pseudo_random_generator_impl::{init,advance,apply_xor}
// (R/lib/phy/upper/sequence_generators/pseudo_random_generator_impl.cpp:58-82,248-316).
__device__ __forceinline__ uint32_t gold_matvec(const uint32_t* rows, uint32_t state, uint32_t lane)
{
  uint32_t bit = (lane < 31u) ? (__popc(rows[lane & 31u] & state) & 1u) : 0u;
  return (uint32_t)__ballot(bit != 0) & 0x7FFFFFFFu;
}

// Orders the LDS traffic among the lanes of one wavefront when the workgroup has several (no s_barrier: the waves of
// the caller run different amounts of work).  LDS operations of a wave complete in order.
__device__ __forceinline__ void wave_lds_fence()
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// Short sequence (DM-RS: at most a few hundred words) generated by ONE wavefront: c(n) for n in [0, 32 nwords) into
// global memory, steps 1-3 of the scheme above, `scratch` = nwords words of LDS private to the wave.
__device__ inline void gold_sequence_wave(const GoldTables* gold, const uint32_t* x1_words, uint32_t c_init,
                                          uint32_t nwords, uint32_t* __restrict__ out, uint32_t* scratch,
                                          uint32_t lane)
{
  uint32_t state = c_init & 0x7FFFFFFFu;
  // Nc = 1600 = 2^6 + 2^9 + 2^10.
  state = gold_matvec(gold->x2_jump[6], state, lane);
  state = gold_matvec(gold->x2_jump[9], state, lane);
  state = gold_matvec(gold->x2_jump[10], state, lane);
  if (lane < 31u && lane < nwords) {
    uint32_t word = 0;
#pragma unroll 8
    for (uint32_t t = 0; t != 32; ++t) {
      word |= (__popc(gold->x2_head[t][lane] & state) & 1u) << (31u - t);
    }
    scratch[lane] = word;
  }
  wave_lds_fence();
  for (uint32_t base = 31; base < nwords; base += 28) { // wave-uniform
    const uint32_t k = base + lane;
    if (lane < 28u && k < nwords) {
      scratch[k] = scratch[k - 28] ^ scratch[k - 29] ^ scratch[k - 30] ^ scratch[k - 31];
    }
    wave_lds_fence();
  }
  for (uint32_t k = lane; k < nwords; k += WAVE) {
    out[k] = scratch[k] ^ x1_words[k];
  }
}

// Sequence generator for a 256-thread workgroup: c(n) for n in [32 first_word, 32 (first_word + nwords)) into global
// memory (out[0] is word first_word).  Wave 0 seeds as above (jump to Nc + 32 first_word, 31 parallel words).
// Squaring the characteristic polynomial 5 + j times lifts the recurrence to
//   W[k] = W[k - 28 m] ^ W[k - 29 m] ^ W[k - 30 m] ^ W[k - 31 m],   m = 2^j,
// which yields 28 m new words in parallel from 31 m known ones: the level rises as the known prefix grows, and from
// m = 64 on every step produces 1792 words from a 4096-word LDS ring.  The steps synchronise on LDS only; the x1
// words of a block are requested one step before they are needed and the block is written out one step after it was
// computed, so no step waits for global memory.
#ifndef NRPHY_GOLD_LEVEL
#define NRPHY_GOLD_LEVEL 64 // top level m of the lifted recurrence: 28 m words per step from a ring of 64 m words (>= 59 m)
#endif
constexpr uint32_t GOLD_MAX_LEVEL  = NRPHY_GOLD_LEVEL;
constexpr uint32_t GOLD_RING_WORDS = 64 * GOLD_MAX_LEVEL;

// Workgroup barrier that orders LDS accesses only (a __syncthreads() would also drain the global loads and stores).
__device__ __forceinline__ void lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// WITH_X1 = false: the x2 part alone (the caller adds the x1 words, which are the same for every sequence, where it consumes
// the result) -- the loop then has no global load whose latency a step must wait for.
template <uint32_t NT, bool WITH_X1 = true>
__device__ inline void gold_sequence_workgroup(const GoldTables* gold, const uint32_t* x1_words, uint32_t c_init,
                                               uint32_t first_word, uint32_t nwords, uint32_t* __restrict__ out,
                                               uint32_t* ring, uint32_t tid)
{
  constexpr uint32_t MASK = GOLD_RING_WORDS - 1;
  constexpr uint32_t PER  = (28u * GOLD_MAX_LEVEL + NT - 1u) / NT; // words per thread and step at the top level
  const uint32_t     lane = tid & (WAVE - 1);
  if (nwords == 0) { // workgroup-uniform
    return;
  }
  // Seeding.  Wave 0 jumps the state to the part's offset; then all waves share the 31 seed words: wave w evaluates
  // bit positions 8w .. 8w + 7 of every word (eight independent table rows per lane, requested together) and ORs its
  // share into the ring.  (The seed used to be wave 0's alone, 32 dependent-latency rows per lane, while the other
  // waves waited: at four parts per PDU it was most of the sequence role's time.)
  static_assert(NT == 4 * WAVE, "four waves share the seed words");
  if (tid < 31u) {
    ring[tid] = 0u;
  }
  if (tid < WAVE) {
    uint32_t       state  = c_init & 0x7FFFFFFFu;
    const uint32_t offset = 1600u + 32u * first_word;
    // The rows of every jump matrix the offset needs are requested before the first one is used: the products depend on
    // each other (the state), the loads do not -- taken one by one, every step paid a trip to L2.
    uint32_t row[GOLD_JUMP_BITS];
#pragma unroll
    for (uint32_t k = 0; k != GOLD_JUMP_BITS; ++k) {
      row[k] = ((offset >> k) & 1u) ? gold->x2_jump[k][lane & 31u] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k != GOLD_JUMP_BITS; ++k) {
      if ((offset >> k) & 1u) { // wave-uniform
        const uint32_t bit = (lane < 31u) ? (__popc(row[k] & state) & 1u) : 0u;
        state              = (uint32_t)__ballot(bit != 0) & 0x7FFFFFFFu;
      }
    }
    if (lane == 0) {
      ring[GOLD_RING_WORDS - 1u] = state; // a slot the recurrence does not reach before the ring wraps
    }
  }
  lds_barrier();
  {
    const uint32_t state = ring[GOLD_RING_WORDS - 1u];
    const uint32_t t0    = 8u * (tid / WAVE);
    if (lane < 31u) {
      uint32_t head[8];
#pragma unroll
      for (uint32_t t = 0; t != 8; ++t) {
        head[t] = gold->x2_head[t0 + t][lane];
      }
      uint32_t word = 0;
#pragma unroll
      for (uint32_t t = 0; t != 8; ++t) {
        word |= (__popc(head[t] & state) & 1u) << (31u - (t0 + t));
      }
      atomicOr(&ring[lane], word);
    }
  }
  lds_barrier();
  // The block computed last and its x1 words: written out while the next block is being computed.
  uint32_t pw[PER], px[PER];
  uint32_t pbeg = 0, pend = nwords < 31u ? nwords : 31u;
#pragma unroll
  for (uint32_t i = 0; i != PER; ++i) {
    const uint32_t k = tid + i * NT;
    pw[i]            = (k < pend) ? ring[k] : 0u;
    px[i]            = (WITH_X1 && k < pend) ? x1_words[first_word + k] : 0u;
  }
  // One step: request the x1 words of the next block, compute the block into `nw`, write out the previous block
  // (`ow`, `ox`: its x1 words were requested a step ago).  The two register sets swap roles from step to step --
  // copying them would make every step wait for the loads it has just issued.
  uint32_t m = 1, have = 31;
  auto step = [&](uint32_t (&nw)[PER], uint32_t (&nx)[PER], const uint32_t (&ow)[PER], const uint32_t (&ox)[PER]) {
    while (m < GOLD_MAX_LEVEL && have >= 62u * m) {
      m *= 2u;
    }
    const uint32_t end = have + 28u * m < nwords ? have + 28u * m : nwords;
#pragma unroll
    for (uint32_t i = 0; i != PER; ++i) {
      const uint32_t k = have + tid + i * NT;
      nx[i]            = (WITH_X1 && k < end) ? x1_words[first_word + k] : 0u;
    }
#pragma unroll
    for (uint32_t i = 0; i != PER; ++i) {
      const uint32_t k = have + tid + i * NT;
      nw[i]            = 0;
      if (k < end) {
        nw[i] = ring[(k - 28u * m) & MASK] ^ ring[(k - 29u * m) & MASK] ^ ring[(k - 30u * m) & MASK] ^
                ring[(k - 31u * m) & MASK];
        ring[k & MASK] = nw[i];
      }
    }
#pragma unroll
    for (uint32_t i = 0; i != PER; ++i) {
      const uint32_t k = pbeg + tid + i * NT;
      if (k < pend) {
        out[k] = ow[i] ^ ox[i];
      }
    }
    pbeg = have;
    pend = end;
    lds_barrier();
    have = end;
  };
  uint32_t qw[PER], qx[PER];
  bool     in_q = false; // which register set holds the block that is still to be written out
  while (have < nwords) { // workgroup-uniform
    step(qw, qx, pw, px);
    in_q = true;
    if (have >= nwords) {
      break;
    }
    step(pw, px, qw, qx);
    in_q = false;
  }
#pragma unroll
  for (uint32_t i = 0; i != PER; ++i) {
    const uint32_t k = pbeg + tid + i * NT;
    if (k < pend) {
      out[k] = in_q ? (qw[i] ^ qx[i]) : (pw[i] ^ px[i]);
    }
  }
  lds_barrier(); // the ring is reused by the caller's next sequence
}

// Long sequence generated by ONE wavefront with the recurrence in registers: the x2 part of c(n) for
// n in [32 first_word, 32 (first_word + nwords)), handed to the caller block by block in LDS.
//
// At level m = 64 the lifted recurrence W[k] = W[k - 28 m] ^ W[k - 29 m] ^ W[k - 30 m] ^ W[k - 31 m] relates words that are
// whole rows of 64 apart: with word 64 r + l in lane l, row r = row(r-28) ^ row(r-29) ^ row(r-30) ^ row(r-31) lane by
// lane.  The wave keeps the last 31 rows in 31 registers and produces a row with one v_bitop3 (three-way XOR) and one
// v_xor: no barrier, no address arithmetic, and 28 rows between a value and its first use.  The 31 seed rows (1984 words)
// come from the scheme of gold_sequence_wave() with the level doubling in `area` (1984 words of LDS private to the
// wave): jump, 31 head words, then levels 1, 1, 2, 4, 8, 8, 16, 32.  Every block of 31 rows is written to `area` (the
// seed rows are there already) and `on_block(base, avail)` is called with area[0, avail) = words [base, base + avail) of
// the part: the caller takes what it needs of them (the prologue: 31 words per codeblock work item; a 16-byte copy of
// everything to global memory was the first form -- 121 MB per 1024 config-3 slots written here and read back by the
// codeblock waves).  The LDS executes a wave's instructions in order: the caller's reads need no wait before the next
// block's writes.
// (The workgroup form above spent 27 vector instructions per word and thread -- four ring reads with their addresses, a
// ring write, bounds -- and a barrier per 1792 words; this one spends 2 per 64 words after a seed of about 650.)
constexpr uint32_t GOLD_SEED_ROWS  = 31;
constexpr uint32_t GOLD_SEED_WORDS = GOLD_SEED_ROWS * WAVE;

template <class OnBlock>
__device__ inline void gold_sequence_blocks_wave(const GoldTables* gold, uint32_t c_init, uint32_t first_word, uint32_t nwords,
                                                 uint32_t* area, uint32_t lane, OnBlock&& on_block)
{
  if (nwords == 0) { // wave-uniform
    return;
  }
  // 1. The state at the part's offset (wave-uniform chain of ballots; every matrix row requested before the first use).
  uint32_t state = c_init & 0x7FFFFFFFu;
  {
    const uint32_t offset = 1600u + 32u * first_word;
    uint32_t       row[GOLD_JUMP_BITS];
#pragma unroll
    for (uint32_t k = 0; k != GOLD_JUMP_BITS; ++k) {
      row[k] = ((offset >> k) & 1u) ? gold->x2_jump[k][lane & 31u] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k != GOLD_JUMP_BITS; ++k) {
      if ((offset >> k) & 1u) { // wave-uniform
        const uint32_t bit = (lane < 31u) ? (__popc(row[k] & state) & 1u) : 0u;
        state              = (uint32_t)__ballot(bit != 0) & 0x7FFFFFFFu;
      }
    }
  }
  NRPHY_WG_TRACE_MARK(1);
  // 2. The first 31 words, lane w the 32 parities of word w.
  if (lane < 31u) {
    uint32_t head[32];
#pragma unroll
    for (uint32_t t = 0; t != 32; ++t) {
      head[t] = gold->x2_head[t][lane];
    }
    uint32_t word = 0;
#pragma unroll
    for (uint32_t t = 0; t != 32; ++t) {
      word |= (__popc(head[t] & state) & 1u) << (31u - t);
    }
    area[lane] = word;
  }
  wave_lds_fence();
  NRPHY_WG_TRACE_MARK(2);
  // 3. Level doubling in LDS up to the 31 seed rows (or the whole sequence if it is shorter).
  const uint32_t need = nwords < GOLD_SEED_WORDS ? nwords : GOLD_SEED_WORDS;
  for (uint32_t have = 31u, m = 1u; have < need;) { // wave-uniform
    while (m < 32u && have >= 62u * m) {
      m *= 2u;
    }
    const uint32_t end = have + 28u * m < need ? have + 28u * m : need;
    for (uint32_t k = have + lane; k < end; k += WAVE) {
      area[k] = __builtin_amdgcn_bitop3_b32(area[k - 28u * m], area[k - 29u * m], area[k - 30u * m], 0x96) ^
                area[k - 31u * m];
    }
    wave_lds_fence();
    have = end;
  }
  NRPHY_WG_TRACE_MARK(3);
  on_block(0u, need);
  NRPHY_WG_TRACE_MARK(4);
  // 4. Rows in registers, a block of 31 at a time through the same area.
  const uint32_t rows = (nwords + WAVE - 1u) / WAVE;
  if (rows <= GOLD_SEED_ROWS) { // wave-uniform
    return;
  }
  uint32_t w[GOLD_SEED_ROWS];
#pragma unroll
  for (uint32_t i = 0; i != GOLD_SEED_ROWS; ++i) {
    w[i] = area[i * WAVE + lane];
  }
  for (uint32_t base = GOLD_SEED_ROWS; base < rows; base += GOLD_SEED_ROWS) { // wave-uniform
#pragma unroll
    for (uint32_t i = 0; i != GOLD_SEED_ROWS; ++i) {
      w[i] ^= __builtin_amdgcn_bitop3_b32(w[(i + 1u) % GOLD_SEED_ROWS], w[(i + 2u) % GOLD_SEED_ROWS],
                                          w[(i + 3u) % GOLD_SEED_ROWS], 0x96);
      area[i * WAVE + lane] = w[i];
    }
    wave_lds_fence(); // (orders the compiler; the hardware needs nothing between the writes and the caller's reads)
    const uint32_t left = nwords - base * WAVE;
    on_block(base * WAVE, left < GOLD_SEED_WORDS ? left : GOLD_SEED_WORDS);
  }
}

// The other end of the seeds: x2[0, need) = the x2 words that follow a 31-word seed (lane l < 31 holds word l), by level
// doubling in LDS private to the wave.  The levels are a fixed schedule -- 1, 1, 2, 4, 8, 8 reach 703 words, more than a
// work item of RE_CHUNK resource elements needs -- so that every step has its level as a compile-time constant: one address
// per lane and 64 words (the four reads and the write are immediate offsets from it) instead of five computed ones.
constexpr uint32_t GOLD_EXPAND_MAX_WORDS = 703;

template <uint32_t M>
__device__ __forceinline__ uint32_t gold_expand_step(uint32_t* x2, uint32_t have, uint32_t need, uint32_t lane)
{
  if (have >= need) { // wave-uniform
    return have;
  }
  const uint32_t end = have + 28u * M < need ? have + 28u * M : need;
#pragma unroll
  for (uint32_t s = 0; s != (28u * M + WAVE - 1u) / WAVE; ++s) {
    if (have + WAVE * s >= end) { // wave-uniform
      break;
    }
    const uint32_t k = have + lane + WAVE * s;
    if (k < end) {
      uint32_t* b = x2 + (k - 31u * M); // b[0], b[M], b[2 M], b[3 M] -> b[31 M]
      b[31u * M]  = __builtin_amdgcn_bitop3_b32(b[3u * M], b[2u * M], b[M], 0x96) ^ b[0];
    }
  }
  wave_lds_fence();
  return end;
}

__device__ inline void gold_expand_seed_wave(uint32_t* x2, uint32_t seed_word, uint32_t need, uint32_t lane)
{
  if (lane < 31u) {
    x2[lane] = seed_word;
  }
  wave_lds_fence();
  uint32_t have = 31u;
  have = gold_expand_step<1>(x2, have, need, lane); //  59
  have = gold_expand_step<1>(x2, have, need, lane); //  87
  have = gold_expand_step<2>(x2, have, need, lane); // 143
  have = gold_expand_step<4>(x2, have, need, lane); // 255
  have = gold_expand_step<8>(x2, have, need, lane); // 479
  have = gold_expand_step<8>(x2, have, need, lane); // 703
}

// ---- Transport-block CRC by regions (prologue_kernel's TB-CRC role; the receive side's transport-block check) ----------
// A CRC is the remainder of a polynomial, so it splits: the block is cut into 16 KiB regions; a 256-thread workgroup
// reduces regions [region0, region0 + count) one after the other, the next region's words in flight while the current one
// is reduced, and returns (in thread 0) remainder * factor, factor = x^(order + 8 (bytes - end of the last region)) mod g:
// its share of the block's CRC.  Inside a region thread t owns four groups of four consecutive words (16-byte loads,
// coalesced, straight from HBM): Horner's rule in x^32 inside a group, in y1k = x^(32 * 1024) across the groups -- four
// independent table look-ups per word; the 256 partials are folded by one wavefront the same way with y8k = x^(128 * 64);
// from region to region those 64 lanes step with yz = x^(8 * 16384), and only once per workgroup do they pay for a
// multiplication by a per-lane constant.  `lds`: TB_CRC_LDS_WORDS words.  Workgroup-uniform arguments.
constexpr int TB_CRC_THREADS = 256;

constexpr int TB_CRC_WPT = NRPHY_CRC_WORDS_PER_THREAD;
static_assert(TB_CRC_REGION_WORDS == TB_CRC_WPT * TB_CRC_THREADS, "words per thread");

// reg * y mod g for a 32-bit partial, y's table in LDS: tab[k * 256 + b] = (b x^(8k)) y mod g.
__device__ __forceinline__ uint32_t crc_advance(const uint32_t* tab, uint32_t reg)
{
  return tab[reg & 0xFFu] ^ tab[256u + ((reg >> 8) & 0xFFu)] ^ tab[512u + ((reg >> 16) & 0xFFu)] ^ tab[768u + (reg >> 24)];
}

constexpr uint32_t TB_CRC_LDS_WORDS = 4 * 1024 + 2 * TB_CRC_THREADS;

__device__ inline uint32_t tbcrc_regions_workgroup(const TbCrcTables* tables, uint32_t sel, CrcPoly c, const uint32_t* tbw, uint32_t n,
                                                   uint32_t region0, uint32_t count, uint32_t factor, uint32_t* lds, uint32_t tid)
{
  uint32_t*       y32 = lds;
  uint32_t*       y1k = lds + 1024;
  uint32_t*       y8k = lds + 2048;
  uint32_t*       yz  = lds + 3072;
  uint32_t*       msg = lds + 4096; // two buffers of 256 partials, used in turn: one barrier per region

  // A region's words, big-endian (zero beyond the transport block, bytes beyond its end masked off): round i of thread t is
  // the 16 bytes at word 1024 i + 4 t of the region.  16-byte loads: with one word per lane and load the same bytes took
  // three times as long to arrive (8.6 against 2.6 us per region, the launch 85 against 54 us: profiles/r03_prologue_trace.txt).
  // A transport block that does not start on a 16-byte boundary, and the 16 bytes that hold its end, take single words.
  constexpr int   ROUNDS = TB_CRC_WPT / 4;
  static_assert(TB_CRC_WPT == 16, "the tables assume four rounds of four words (y1k, y8k)");
  const uint32_t  nwords  = (n + 3u) >> 2;
  const bool      aligned = (reinterpret_cast<uintptr_t>(tbw) & 15u) == 0; // workgroup-uniform
  auto load_region = [&](uint32_t region, uint32_t (&w)[TB_CRC_WPT]) {
#pragma unroll
    for (int i = 0; i != ROUNDS; ++i) {
      const uint32_t idx = region * TB_CRC_REGION_WORDS + ((uint32_t)i * TB_CRC_THREADS + tid) * 4u;
      if (aligned && idx + 4u <= nwords && !((n & 3u) != 0 && idx + 4u == nwords)) {
        const uint4 v = *reinterpret_cast<const uint4*>(tbw + idx);
        w[4 * i] = __builtin_bswap32(v.x), w[4 * i + 1] = __builtin_bswap32(v.y);
        w[4 * i + 2] = __builtin_bswap32(v.z), w[4 * i + 3] = __builtin_bswap32(v.w);
      } else {
#pragma unroll
        for (int k = 0; k != 4; ++k) {
          uint32_t x = (idx + k < nwords) ? be_word(tbw, idx + k) : 0u;
          if ((n & 3u) != 0 && idx + k + 1u == nwords) {
            x &= 0xFFFFFFFFu << (8u * (4u - (n & 3u)));
          }
          w[4 * i + k] = x;
        }
      }
    }
  };
  uint32_t w[TB_CRC_WPT], wn[TB_CRC_WPT];
  load_region(region0, w);
#pragma unroll
  for (int k = 0; k != 4; ++k) {
    y32[k * 256 + tid] = tables->y32[sel][k][tid];
    y1k[k * 256 + tid] = tables->y1k[sel][k][tid];
    y8k[k * 256 + tid] = tables->y8k[sel][k][tid];
    yz[k * 256 + tid]  = tables->yz[sel][k][tid];
  }
  __syncthreads();
#ifdef NRPHY_WG_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (trace builds: tables in LDS, the first region's words in registers)
  NRPHY_WG_TRACE_MARK(2);
#endif
  // The workgroup's regions one after the other.  Lane l of wave 0 carries acc = sum over the regions so far of (its four
  // partials of the region, folded with y8k) * yz^(regions that follow): one more table product per region and lane, and the
  // two multiplications without a table -- by the lane's constant and by the share factor -- once per workgroup.
  // (Two regions' loads ahead instead of one -- three register sets in rotation -- changed nothing: with the sequences no
  // longer written out the workgroup waits for its turn at the vector unit and the LDS, not for its loads.)
  uint32_t acc = 0;
  for (uint32_t j = 0; j != count; ++j) { // workgroup-uniform
    if (j + 1u != count) {
      load_region(region0 + j + 1u, wn); // in flight while this region is reduced
    }
    // four independent chains of three products (the words of a round), then three products across the rounds
    uint32_t v[ROUNDS];
#pragma unroll
    for (int i = 0; i != ROUNDS; ++i) {
      v[i] = w[4 * i];
    }
#pragma unroll
    for (int k = 1; k != 4; ++k) {
#pragma unroll
      for (int i = 0; i != ROUNDS; ++i) {
        v[i] = crc_advance(y32, v[i]) ^ w[4 * i + k];
      }
    }
    uint32_t reg = v[0];
#pragma unroll
    for (int i = 1; i != ROUNDS; ++i) {
      reg = crc_advance(y1k, reg) ^ v[i];
    }
    uint32_t* m = msg + (j & 1u) * TB_CRC_THREADS;
    m[tid]      = reg;
    lds_barrier();
    if (tid < WAVE) {
      uint32_t r = m[tid];
#pragma unroll
      for (int k = 1; k != 4; ++k) {
        r = crc_advance(y8k, r) ^ m[tid + WAVE * k];
      }
      acc = (j != 0 ? crc_advance(yz, acc) : 0u) ^ r;
    }
    if (j + 1u != count) {
#pragma unroll
      for (int i = 0; i != TB_CRC_WPT; ++i) {
        w[i] = wn[i];
      }
    }
  }
  NRPHY_WG_TRACE_MARK(1);
  uint32_t share = 0;
  if (tid < WAVE) {
    uint32_t r = crc_mulmod32(tables->lane[sel][tid], acc, c);
    r          = wave_xor(r);
    share      = crc_mulmod(r, factor, c);
  }
  return share;
}

// round-to-nearest-even float -> bf16 exactly as the reference stores the grid
// (to_bf16, R/include/srsran/adt/bf16.h:39-56); values here are finite.
__device__ __forceinline__ uint32_t to_bf16_bits(float v)
{
  uint32_t u = __float_as_uint(v);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}

} // namespace nrphy
