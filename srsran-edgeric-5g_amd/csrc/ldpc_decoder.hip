// LDPC decoder for gfx950 (MI355X): layered scaled min-sum on int8 log-likelihood ratios ("next" row, SURVEY.md
// section 8f-1, receive side).
//
// Replaces ldpc_decoder_impl::decode with the generic message kernels
// (R/lib/phy/upper/channel_coding/ldpc/ldpc_decoder_impl.cpp:60-318, ldpc_decoder_generic.cpp:30-128; LLR arithmetic
// R/lib/phy/upper/log_likelihood_ratio.cpp:37-87) bit for bit: same clamps, same saturating / promoting sums, same
// tie-breaking of the two minima, same rounding of the scaled magnitude, same early stop on the CRC.
//
// One workgroup per codeblock, thread j owns lifted check j of every layer (Zc <= 384 threads).  The soft bits live in
// LDS.  Check j of a layer reads the soft bit of each neighbour at its rotated position (j + shift) mod Zc, subtracts
// its own previous message, runs the min-sum rule and writes the new soft bit back to the same place: within a layer
// every (variable, position) pair belongs to exactly one check, so a layer needs no exchange buffer and one barrier.
// The check-to-variable messages are kept in the compressed form the min-sum rule allows -- per lifted check its two
// scaled magnitudes, the edge holding the minimum and one sign bit per edge: 8 bytes instead of up to 19 -- read and
// written only by the thread that owns the check (coalesced, L2 resident; the next layer's record is prefetched).
// This is the reference's arithmetic re-indexed by check instead of by variable position; the values are the same.
#include "bits_device.h"

#include <cstdlib>
#include <type_traits>

namespace nrphy {

constexpr int LLR_MAX_V = 120;
constexpr int LLR_INF_V = 121; // what this kernel keeps an infinite soft bit as (the reference's LLR_INFTY is 127), see below

// ---- LLR arithmetic -------------------------------------------------------------------------------------------------
// The check-to-variable magnitudes are finite by construction: the two minima start at LLR_MAX and only shrink
// (ldpc_decoder_generic.cpp:88-107), and the scaling factor is below one.  With a finite message c the rules of
// log_likelihood_ratio.cpp:37-87 reduce to: "a - c" clamps to +-LLR_MAX unless the soft bit a is infinite, which then
// stays; "c + v" promotes a sum beyond +-LLR_MAX to infinity, and an infinite v stays (the "a == -b gives 0" case is
// what the plain sum yields anyway).  Any value beyond +-LLR_MAX behaves the same in every rule and only hard bits leave
// the kernel, so an infinite soft bit is kept as +-(LLR_MAX + 1): promotion is then a plain clamp of the sum to
// +-(LLR_MAX + 1), and so is the normalisation on load.  Everything is written with median / min / max / bit-field
// operations: no compare-and-select pairs, which cost wait states on gfx950.
__device__ __forceinline__ int med3(int x, int lo, int hi)
{
  return max(lo, min(x, hi)); // v_med3_i32
}
__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c)
{
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// Variable-to-check message a - c.  An infinite soft bit yields a value far beyond the finite range with the sign of
// a (+-512 on top): never a minimum, the right sign, and llr_add_promote() turns it back into an infinite soft bit.
__device__ __forceinline__ int llr_sub(int a, int c)
{
  const int d   = med3(a - c, -LLR_MAX_V, LLR_MAX_V);
  const int big = a - med3(a, -LLR_MAX_V, LLR_MAX_V); // 0, or +-1 for an infinite soft bit
  return (big << 9) + d;
}
// New soft bit c + v, promoting.
__device__ __forceinline__ int llr_add_promote(int c, int v)
{
  return med3(c + v, -LLR_INF_V, LLR_INF_V);
}

// Check record: lo = min1 | min2 << 8 | index of the minimum << 16 (scaled magnitudes, index 0xFF: none), hi = one sign
// bit per edge, edge t of a degree-DEG check at bit DEG - 1 - t.
template <uint32_t DEG>
struct CheckMessages {
  int      m1, m2;
  uint32_t hot, signs;
  __device__ __forceinline__ explicit CheckMessages(uint2 rec) :
    m1((int)(rec.x & 0xFFu)), m2((int)((rec.x >> 8) & 0xFFu)), signs(rec.y)
  {
    const uint32_t idx = rec.x >> 16;
    hot                = idx < 32u ? 1u << idx : 0u;
  }
  __device__ __forceinline__ int operator()(uint32_t t) const
  {
    const uint32_t sel = (uint32_t)__builtin_amdgcn_sbfe((int)hot, t, 1);             // all ones on the minimum's edge
    const int      mag = (int)((sel & (uint32_t)m2) | (~sel & (uint32_t)m1));         // v_bfi_b32
    const int      sgn = __builtin_amdgcn_sbfe((int)signs, DEG - 1u - t, 1);          // 0 or -1
    return (mag ^ sgn) - sgn;
  }
};

// One lifted check of degree DEG: reads the soft bits of its neighbours, returns its new record, writes them back.
// jm = j - Zc (wraps): min(j + shift, jm + shift) = (j + shift) mod Zc.
// FIRST: the first iteration, in which the check has not sent a message yet (all of old_rec zero): a - 0 needs no message and no
// subtraction.
template <uint32_t DEG, bool FIRST>
__device__ __forceinline__ uint2 process_check(int8_t* soft, const uint8_t* scaled, const NRPHY_CONSTANT uint32_t* edge,
                                               uint32_t zc, uint32_t j, uint32_t jm, uint2 old_rec)
{
  uint32_t addr[DEG];
  int      v[DEG];
#pragma unroll
  for (uint32_t t = 0; t != DEG; ++t) {
    const uint32_t e = edge[t], shift = e & 0xFFFFu;
    addr[t]          = (e >> 16) + min(j + shift, jm + shift); // the graph holds node * Zc
  }
#pragma unroll
  for (uint32_t t = 0; t != DEG; ++t) {
    v[t] = soft[addr[t]];
  }
  const CheckMessages<DEG> old(old_rec);
  // two smallest magnitudes as keys (magnitude << 8 | edge): ties go to the earlier edge as in the reference's strict
  // comparison; an untouched first key (index 0xFF) means both minima are LLR_MAX and the owner does not matter
  uint32_t key1 = ((uint32_t)LLR_MAX_V << 8) | 0xFFu, key2 = key1, neg = 0;
#pragma unroll
  for (uint32_t t = 0; t != DEG; ++t) {
    const int x = FIRST ? llr_sub(v[t], 0) : llr_sub(v[t], old(t));
    v[t]        = x;
    const uint32_t key = ((uint32_t)max(x, -x) << 8) | t;
    key2               = med3_u32(key, key1, key2);
    key1               = min(key, key1);
    neg                = __builtin_amdgcn_alignbit(neg, (uint32_t)x, 31); // neg << 1 | sign
  }
  // scale_llr (ldpc_decoder_generic.cpp:69-79) through the table of round(m * scaling_factor)
  const uint32_t s1    = scaled[key1 >> 8], s2 = scaled[key2 >> 8];
  const uint32_t signs = (__popc(neg) & 1u) ? (neg ^ ((1u << DEG) - 1u)) : neg;
  const uint2    mine  = make_uint2(s1 | (s2 << 8) | ((key1 & 0xFFu) << 16), signs);
  const CheckMessages<DEG> now(mine);
#pragma unroll
  for (uint32_t t = 0; t != DEG; ++t) {
    soft[addr[t]] = (int8_t)llr_add_promote(now(t), v[t]);
  }
  return mine;
}

// The layer routine for every row degree of the two base graphs.
template <bool FIRST>
__device__ __forceinline__ uint2 process_layer(uint32_t deg, int8_t* soft, const uint8_t* scaled,
                                               const NRPHY_CONSTANT uint32_t* edge, uint32_t zc, uint32_t j, uint32_t jm, uint2 old)
{
  switch (deg) {
    case 3: return process_check<3, FIRST>(soft, scaled, edge, zc, j, jm, old);
    case 4: return process_check<4, FIRST>(soft, scaled, edge, zc, j, jm, old);
    case 5: return process_check<5, FIRST>(soft, scaled, edge, zc, j, jm, old);
    case 6: return process_check<6, FIRST>(soft, scaled, edge, zc, j, jm, old);
    case 7: return process_check<7, FIRST>(soft, scaled, edge, zc, j, jm, old);
    case 8: return process_check<8, FIRST>(soft, scaled, edge, zc, j, jm, old);
    case 9: return process_check<9, FIRST>(soft, scaled, edge, zc, j, jm, old);
    case 10: return process_check<10, FIRST>(soft, scaled, edge, zc, j, jm, old);
    default: return process_check<19, FIRST>(soft, scaled, edge, zc, j, jm, old);
  }
}

// ---- Two checks per lane ----------------------------------------------------------------------------------------------
// For an even lifting size thread j owns checks j and j + Zc / 2 of every layer and carries their values as the two 16-bit
// halves of one register: the arithmetic of the min-sum rule (subtractions, clamps, magnitudes, the two running minima as
// 16-bit keys, the promotion of the new soft bit) is one packed instruction for both checks, the sign and minimum-owner masks
// of an edge come out of the record with two packed shifts for both, and the second check's soft-bit address is the first's
// plus or minus Zc / 2.  Per check and edge that is about 22 vector instructions instead of 31 (PMC: 93.8 k instead of 114.7 k
// per config-3 codeblock and 8 iterations; the per-layer work of a lane -- record, table look-ups, parity -- does not shrink)
// -- the kernel is bound by vector issue -- for exactly the same values: every operation is the 16-bit image of the one in
// process_check (values stay within
// +-633, keys within 16 bits because magnitudes beyond 255 -- an infinite soft bit's -- are clamped to 255, which like them is
// above LLR_MAX and never a minimum).
//
// Record of a pair of checks (16 bytes: the size of two single records):
//   x = m1A | m2A << 8 | m1B << 16 | m2B << 24     scaled minima of check A (= j) and B (= j + Zc / 2)
//   y = idxA | idxB << 16                           edge holding the minimum (0xFF: none)
//   z = signs of edges 0 .. 15, A in the low half, B in the high half (bit t: message of edge t is negative)
//   w = the same for edges 16 .. (at most 19 edges per check)
typedef short          s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s16x2(uint32_t x)
{
  return __builtin_bit_cast(s16x2, x);
}
__device__ __forceinline__ u16x2 as_u16x2(uint32_t x)
{
  return __builtin_bit_cast(u16x2, x);
}
__device__ __forceinline__ uint32_t as_word(s16x2 x)
{
  return __builtin_bit_cast(uint32_t, x);
}
__device__ __forceinline__ uint32_t as_word(u16x2 x)
{
  return __builtin_bit_cast(uint32_t, x);
}
__device__ __forceinline__ s16x2 splat_s16(int v)
{
  return s16x2{(short)v, (short)v};
}
// 0xFFFF in every half of `word` whose bit `bit` (< 16) is set, else 0: two packed shifts.
__device__ __forceinline__ uint32_t half_masks(uint32_t word, uint32_t bit)
{
  return as_word((as_s16x2(word) << splat_s16((int)(15u - bit))) >> splat_s16(15));
}
__device__ __forceinline__ s16x2 clamp_s16x2(s16x2 x, int lim)
{
  return __builtin_elementwise_max(__builtin_elementwise_min(x, splat_s16(lim)), splat_s16(-lim));
}
// llr_sub for both halves (a: soft bits, c: old messages).
__device__ __forceinline__ s16x2 llr_sub_pair(s16x2 a, s16x2 c)
{
  const s16x2 d   = clamp_s16x2(a - c, LLR_MAX_V);
  const s16x2 big = a - clamp_s16x2(a, LLR_MAX_V); // 0, or +-1 for an infinite soft bit
  return (big << splat_s16(9)) + d;
}
__device__ __forceinline__ s16x2 llr_sub_pair_first(s16x2 a)
{
  const s16x2 d = clamp_s16x2(a, LLR_MAX_V);
  return ((a - d) << splat_s16(9)) + d;
}

// The messages of the pair on edge t from a record's fields: m1 / m2 = (m1A | m1B << 16) / (m2A | m2B << 16), hot0 / hot1 =
// one-hot minimum owners, signs0 / signs1 = the record's z / w.
template <uint32_t T>
__device__ __forceinline__ s16x2 pair_message(uint32_t m1, uint32_t m2, uint32_t hot0, uint32_t hot1, uint32_t signs0, uint32_t signs1)
{
  const uint32_t sel = half_masks(T < 16u ? hot0 : hot1, T & 15u);
  const uint32_t sg  = half_masks(T < 16u ? signs0 : signs1, T & 15u);
  const uint32_t mag = __builtin_amdgcn_bitop3_b32(sel, m2, m1, 0xCA);
  return as_s16x2(mag ^ sg) - as_s16x2(sg);
}
__device__ __forceinline__ void pair_one_hot(uint32_t y, uint32_t& hot0, uint32_t& hot1)
{
  const uint32_t ia = y & 0xFFu, ib = (y >> 16) & 0xFFu;
  const uint32_t ha = ia < 32u ? 1u << ia : 0u, hb = ib < 32u ? 1u << ib : 0u;
  hot0              = (ha & 0xFFFFu) | (hb << 16);
  hot1              = (ha >> 16) | (hb & 0xFFFF0000u);
}

template <uint32_t DEG, uint32_t T, bool FIRST>
struct PairEdges {
  // Pass 1 over edges T .. DEG - 1: v2c messages, running minima, sign bits.
  static __device__ __forceinline__ void forward(const uint32_t (&a)[DEG], uint32_t (&x)[DEG], uint32_t m1, uint32_t m2, uint32_t hot0,
                                                 uint32_t hot1, uint32_t z, uint32_t w, uint32_t& k1, uint32_t& k2, uint32_t& nz,
                                                 uint32_t& nw)
  {
    if constexpr (T < DEG) {
      const s16x2 v = FIRST ? llr_sub_pair_first(as_s16x2(a[T]))
                            : llr_sub_pair(as_s16x2(a[T]), pair_message<T>(m1, m2, hot0, hot1, z, w));
      x[T]          = as_word(v);
      const s16x2    mag = __builtin_elementwise_max(v, splat_s16(0) - v);
      const uint32_t cap = as_word(__builtin_elementwise_min(as_u16x2(as_word(mag)), u16x2{255, 255}));
      const uint32_t key = (cap << 8) | (T * 0x00010001u);
      k2 = as_word(__builtin_elementwise_min(__builtin_elementwise_max(as_u16x2(key), as_u16x2(k1)), as_u16x2(k2)));
      k1 = as_word(__builtin_elementwise_min(as_u16x2(key), as_u16x2(k1)));
      const uint32_t neg = as_word(as_u16x2(as_word(v)) >> u16x2{15, 15}); // 1 in a half whose value is negative
      if (T < 16u) {
        nz |= neg << (T & 15u);
      } else {
        nw |= neg << (T & 15u);
      }
      PairEdges<DEG, T + 1, FIRST>::forward(a, x, m1, m2, hot0, hot1, z, w, k1, k2, nz, nw);
    }
  }
  // Pass 2: new soft bits = new message + v2c message, promoted, back to where they came from.
  static __device__ __forceinline__ void backward(int8_t* soft, const uint32_t (&addr1)[DEG], const uint32_t (&addr2)[DEG],
                                                  const uint32_t (&x)[DEG], uint32_t m1, uint32_t m2, uint32_t hot0, uint32_t hot1,
                                                  uint32_t z, uint32_t w)
  {
    if constexpr (T < DEG) {
      const s16x2    sum = pair_message<T>(m1, m2, hot0, hot1, z, w) + as_s16x2(x[T]);
      const uint32_t out = as_word(clamp_s16x2(sum, LLR_INF_V));
      soft[addr1[T]]     = (int8_t)out;
      soft[addr2[T]]     = (int8_t)(out >> 16);
      PairEdges<DEG, T + 1, FIRST>::backward(soft, addr1, addr2, x, m1, m2, hot0, hot1, z, w);
    }
  }
};

// Checks j and j + half of one layer (degree DEG); j < half = Zc / 2; jm = j - Zc (wraps).
template <uint32_t DEG, bool FIRST>
__device__ __forceinline__ uint4 process_check_pair(int8_t* soft, const uint8_t* scaled, const NRPHY_CONSTANT uint32_t* edge,
                                                    uint32_t half, uint32_t minus_half, uint32_t j, uint32_t jm, uint4 old)
{
  uint32_t addr1[DEG], addr2[DEG], a[DEG], x[DEG];
#pragma unroll
  for (uint32_t t = 0; t != DEG; ++t) {
    const uint32_t e = edge[t], shift = e & 0xFFFFu;
    const uint32_t pos = min(j + shift, jm + shift);   // (j + shift) mod Zc
    addr1[t]           = (e >> 16) + pos;              // the graph holds node * Zc
    addr2[t]           = addr1[t] + (pos < half ? half : minus_half); // (j + Zc / 2 + shift) mod Zc
  }
#pragma unroll
  for (uint32_t t = 0; t != DEG; ++t) {
    const int va = soft[addr1[t]], vb = soft[addr2[t]];
    a[t]         = __builtin_amdgcn_perm((uint32_t)vb, (uint32_t)va, 0x05040100u); // low halves: A | B << 16
  }
  uint32_t m1 = 0, m2 = 0, hot0 = 0, hot1 = 0;
  if (!FIRST) {
    m1 = old.x & 0x00FF00FFu;
    m2 = (old.x >> 8) & 0x00FF00FFu;
    pair_one_hot(old.y, hot0, hot1);
  }
  uint32_t k1 = (((uint32_t)LLR_MAX_V << 8) | 0xFFu) * 0x00010001u, k2 = k1, nz = 0, nw = 0;
  PairEdges<DEG, 0, FIRST>::forward(a, x, m1, m2, hot0, hot1, old.z, old.w, k1, k2, nz, nw);
  // scale_llr of the four minima through the table; sign of a message = parity of the OTHER signs
  const uint32_t s1a = scaled[(k1 >> 8) & 0xFFu], s1b = scaled[k1 >> 24], s2a = scaled[(k2 >> 8) & 0xFFu], s2b = scaled[k2 >> 24];
  const uint32_t n1 = s1a | (s1b << 16), n2 = s2a | (s2b << 16);
  const uint32_t sa = (nz & 0xFFFFu) | (nw << 16), sb = (nz >> 16) | (nw & 0xFFFF0000u);
  const uint32_t flip = ((0u - (__popc(sa) & 1u)) & 0xFFFFu) | ((0u - (__popc(sb) & 1u)) << 16);
  constexpr uint32_t ZBITS = DEG >= 16u ? 0xFFFFu : (1u << DEG) - 1u, WBITS = DEG > 16u ? (1u << (DEG - 16u)) - 1u : 0u;
  nz ^= flip & (ZBITS * 0x00010001u);
  nw ^= flip & (WBITS * 0x00010001u);
  const uint4 mine = make_uint4(n1 | (n2 << 8), k1 & 0x00FF00FFu, nz, nw);
  uint32_t    nh0, nh1;
  pair_one_hot(mine.y, nh0, nh1);
  PairEdges<DEG, 0, FIRST>::backward(soft, addr1, addr2, x, n1, n2, nh0, nh1, nz, nw);
  return mine;
}

template <bool FIRST>
__device__ __forceinline__ uint4 process_layer_pair(uint32_t deg, int8_t* soft, const uint8_t* scaled,
                                                    const NRPHY_CONSTANT uint32_t* edge, uint32_t half, uint32_t minus_half, uint32_t j,
                                                    uint32_t jm, uint4 old)
{
  switch (deg) {
    case 3: return process_check_pair<3, FIRST>(soft, scaled, edge, half, minus_half, j, jm, old);
    case 4: return process_check_pair<4, FIRST>(soft, scaled, edge, half, minus_half, j, jm, old);
    case 5: return process_check_pair<5, FIRST>(soft, scaled, edge, half, minus_half, j, jm, old);
    case 6: return process_check_pair<6, FIRST>(soft, scaled, edge, half, minus_half, j, jm, old);
    case 7: return process_check_pair<7, FIRST>(soft, scaled, edge, half, minus_half, j, jm, old);
    case 8: return process_check_pair<8, FIRST>(soft, scaled, edge, half, minus_half, j, jm, old);
    case 9: return process_check_pair<9, FIRST>(soft, scaled, edge, half, minus_half, j, jm, old);
    case 10: return process_check_pair<10, FIRST>(soft, scaled, edge, half, minus_half, j, jm, old);
    default: return process_check_pair<19, FIRST>(soft, scaled, edge, half, minus_half, j, jm, old);
  }
}

// One soft bit as it enters the decoder (ldpc_decoder_impl.cpp:128-164): whole nodes are clamped to +-64, the tail
// is taken as is (infinities in this kernel's form, see above).
__device__ __forceinline__ int load_soft(int v, bool whole_node)
{
  return whole_node ? med3(v, -64, 64) : med3(v, -LLR_INF_V, LLR_INF_V);
}

// Hard bits [32 w, 32 w + 32) of the soft bits (MSB first); zero_seen: an undecided one among the first `limit`.
__device__ __forceinline__ uint32_t hard_word(const int8_t* soft, uint32_t w, uint32_t limit, bool& zero_seen)
{
  const uint4    lo = *reinterpret_cast<const uint4*>(soft + 32u * w);
  const uint4    hi = *reinterpret_cast<const uint4*>(soft + 32u * w + 16u);
  const uint32_t x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  uint32_t       word = 0, zeros = 0;
#pragma unroll
  for (uint32_t i = 0; i != 32; ++i) {
    const int v = (int)(int8_t)(x[i >> 2] >> (8u * (i & 3u)));
    word |= (v <= 0 ? 1u : 0u) << (31u - i);
    zeros |= (v == 0 ? 1u : 0u) << (31u - i);
  }
  const uint32_t valid = limit > 32u * w ? topmask(limit - 32u * w < 32u ? limit - 32u * w : 32u) : 0u;
  zero_seen |= (zeros & valid) != 0;
  return word;
}

// The check records of a codeblock live in a slot of the caller's scratch.  A batch larger than the pool shares it: a
// workgroup claims a free slot (one flag word per slot) when it starts and gives it back when it is done.  Workgroup b first
// tries slot b mod nof_slots -- free for the workgroups that start a launch, and usually given back by workgroup
// b - nof_slots by the time b starts -- and walks on from there.  The pool holds at least as many slots as workgroups fit the
// device at once, so a free one always exists and the search ends; should that bound ever be wrong the search gives up
// after a fixed number of probes and the codeblock is reported as not decoded (never a hang).
//
// Handing a slot from one workgroup to the next needs no cache maintenance: a workgroup never reads a record it has not
// written itself, every record store is a write-through (agent-scope) store, and the owner waits for its stores to complete
// before it clears the flag -- so no write of the old owner can land after one of the new owner.  (An agent-scope fence
// here instead, with plain stores, wrote the L2 back per workgroup and cost 1 ms per 6656 codeblocks.)
__device__ __forceinline__ uint32_t acquire_slot(uint32_t* flags, uint32_t nof_slots, uint32_t first)
{
  uint32_t s = first % nof_slots;
  for (uint32_t probes = 0; probes != (1u << 20); ++probes) {
    if (__hip_atomic_load(&flags[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 &&
        __hip_atomic_exchange(&flags[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      return s;
    }
    s = s + 1u == nof_slots ? 0u : s + 1u;
    if ((probes & 63u) == 63u) {
      __builtin_amdgcn_s_sleep(8);
    }
  }
  return 0xFFFFFFFFu;
}

__device__ __forceinline__ void store_record(uint2* rec, uint2 v)
{
  __hip_atomic_store(reinterpret_cast<uint64_t*>(rec), (uint64_t)v.x | ((uint64_t)v.y << 32), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_record(uint4* rec, uint4 v) // a pair of checks: two write-through stores
{
  uint64_t* q = reinterpret_cast<uint64_t*>(rec);
  __hip_atomic_store(q, (uint64_t)v.x | ((uint64_t)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, (uint64_t)v.z | ((uint64_t)v.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// PAIR: two checks per lane (even lifting sizes: Zc / 2 threads per codeblock), see process_check_pair.
template <bool PAIR>
__device__ __forceinline__ void ldpc_decode_body(const LdpcDecodeLaunch& p)
{
  extern __shared__ __attribute__((aligned(16))) int8_t dec_lds[];
  __shared__ uint32_t s_flag[4];
  __shared__ uint8_t  s_scaled[128];
  const uint32_t zc = p.zc, j = threadIdx.x;
  int8_t*        soft = dec_lds; // [nof_nodes][zc] (+ 32 bytes of slack for the word reads of the last hard bits)

  if (p.skip != nullptr && p.skip[blockIdx.x] != 0) { // workgroup-uniform: decoded in an earlier transmission
    return;
  }
  const auto*   graph  = to_constant(p.graph); // wave-uniform reads: scalar loads
  const int8_t* llr    = p.llr + (size_t)blockIdx.x * p.llr_stride;
  const uint32_t half = zc >> 1;
  const bool    active = PAIR ? j < half : j < zc;
  const bool    pooled = p.nof_slots < gridDim.x; // fewer slots than codeblocks: claim one
  if (j == 0) { // (read after the barriers below; the claim's latency hides behind the loads)
    s_flag[3] = pooled ? acquire_slot(p.slot_flags, p.nof_slots, blockIdx.x) : blockIdx.x;
  }

  // load_soft_bits (ldpc_decoder_impl.cpp:128-164): two punctured nodes, then the input.  The last non-zero soft bit
  // decides how many layers take part (:88-116).
  uint32_t       last_nz   = 0;
  const uint32_t T         = blockDim.x;
  const uint32_t clamp_end = (p.nof_llr / zc) * zc; // whole nodes
  const uint32_t lds_bytes = p.nof_nodes * zc + 48u;
  for (uint32_t i = j; i < 2u * zc; i += T) {
    soft[i] = 0;
  }
  for (uint32_t i = 2u * zc + p.nof_llr + j; i < lds_bytes; i += T) {
    soft[i] = 0;
  }
  if (((reinterpret_cast<uintptr_t>(llr) | zc) & 3u) == 0) {
    // four soft bits per lane and load
    const uint32_t  nd  = p.nof_llr >> 2;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(llr);
    uint32_t*       dst = reinterpret_cast<uint32_t*>(soft + 2u * zc);
#pragma unroll 4
    for (uint32_t d = j; d < nd; d += T) {
      const uint32_t x = src[d];
      if (x == 0) { // (at high code rates most of the buffer: nothing received there yet)
        dst[d] = 0;
        continue;
      }
      last_nz    = max(last_nz, 4u * d + 4u - ((uint32_t)__clz(x) >> 3));
      uint32_t y = 0;
#pragma unroll
      for (uint32_t b = 0; b != 4; ++b) {
        const int v = load_soft((int)(int8_t)(x >> (8u * b)), 4u * d + b < clamp_end);
        y |= ((uint32_t)v & 0xFFu) << (8u * b);
      }
      dst[d] = y;
    }
    const uint32_t i = 4u * nd + j;
    if (i < p.nof_llr) {
      const int v = llr[i];
      last_nz     = v != 0 ? max(last_nz, i + 1u) : last_nz;
      soft[2u * zc + i] = (int8_t)load_soft(v, i < clamp_end);
    }
  } else {
#pragma unroll 4
    for (uint32_t i = j; i < p.nof_llr; i += T) {
      const int v = llr[i];
      last_nz     = v != 0 ? max(last_nz, i + 1u) : last_nz;
      soft[2u * zc + i] = (int8_t)load_soft(v, i < clamp_end);
    }
  }
  for (uint32_t m = j; m <= (uint32_t)LLR_MAX_V; m += T) { // a workgroup may be a single wavefront (Zc <= 64)
    s_scaled[m] = (uint8_t)roundf((float)m * p.scaling_factor); // rounded half away from zero
  }
  if (j < 3) {
    s_flag[j] = 0;
  }
  __syncthreads();
  if (last_nz != 0) {
    atomicMax(&s_flag[0], last_nz);
  }
  __syncthreads();
  const uint32_t input_size = s_flag[0];
  const uint32_t K          = p.bg_k * zc;
  const uint32_t nw_k       = (K + 31u) >> 5;
  uint32_t       iterations = 0;

  // Early stop: the message is a multiple of the generator polynomial.  Thread w owns hard-bit word w; its weight
  // x^(bits after the word) mod g comes from the host.
  const CrcPoly  crc    = {p.crc_poly, p.crc_order};
  const uint32_t n_msg  = K - p.nof_filler;
  const uint32_t jm = j - zc;

  if (input_size != 0) { // workgroup-uniform
    const uint32_t slot = s_flag[3];
    uint2*         rec  = p.scratch + (size_t)(slot == 0xFFFFFFFFu ? 0u : slot) * p.nof_layers_max * zc;
    const uint32_t max_iterations = slot == 0xFFFFFFFFu ? 0u : p.max_iterations; // no slot: reported as not decoded
    uint32_t cb_len = input_size + 2u * zc;
    cb_len          = cb_len < K + 4u * zc ? K + 4u * zc : cb_len;
    cb_len          = ((cb_len + zc - 1u) / zc) * zc;
    const uint32_t nof_layers = cb_len / zc - p.bg_k;

    typedef typename std::conditional<PAIR, uint4, uint2>::type Record; // a pair of checks per lane has a record of twice the size
    Record*        recs        = reinterpret_cast<Record*>(rec);
    const uint32_t rec_stride  = PAIR ? half : zc;                       // records of one layer
    const uint32_t minus_half  = 0u - half;
    for (uint32_t it = 0; it != max_iterations && iterations == 0; ++it) {
      Record next = {};
      if (it != 0 && active) {
        next = recs[j];
      }
      for (uint32_t m = 0; m != nof_layers; ++m) {
        const uint32_t e0 = graph->row_ptr[m], deg = graph->row_ptr[m + 1u] - e0;
        const Record   old = next;
        if (it != 0 && active && m + 1u != nof_layers) {
          next = recs[(size_t)(m + 1u) * rec_stride + j];
        }
        if (active) {
          const auto* edge = graph->edge + e0;
          if constexpr (PAIR) {
            const uint4 mine = it == 0 ? process_layer_pair<true>(deg, soft, s_scaled, edge, half, minus_half, j, jm, old)
                                       : process_layer_pair<false>(deg, soft, s_scaled, edge, half, minus_half, j, jm, old);
            store_record(&recs[(size_t)m * rec_stride + j], mine);
          } else {
            const uint2 mine = it == 0 ? process_layer<true>(deg, soft, s_scaled, edge, zc, j, jm, old)
                                       : process_layer<false>(deg, soft, s_scaled, edge, zc, j, jm, old);
            store_record(&recs[(size_t)m * rec_stride + j], mine);
          }
        }
        lds_barrier();
      }
      // Early stop (ldpc_decoder_impl.cpp:118-126): every hard bit decided and the CRC of the significant bits zero.
      // crc_at_end (pusch_codeblock_decoder.cpp:59-68): no check until the last iteration, then the CRC alone decides.
      if (p.crc_order != 0 && (!p.crc_at_end || it + 1u == max_iterations)) {
        if (j < 2) {
          s_flag[1 + j] = 0;
        }
        lds_barrier();
        bool     zero_seen = false;
        uint32_t part      = 0;
        for (uint32_t w = j; w < nw_k; w += T) { // thread w owns hard-bit word w (one trip, two with two checks per lane)
          const uint32_t word = hard_word(soft, w, K, zero_seen);
          if (32u * w < n_msg) {
            const uint32_t w_bits = n_msg - 32u * w < 32u ? n_msg - 32u * w : 32u;
            part ^= crc_mulmod32(p.crc_weight[w], word >> (32u - w_bits), crc);
          }
        }
        for (int o = WAVE / 2; o != 0; o >>= 1) {
          part ^= __shfl_xor(part, o);
        }
        if ((j & (WAVE - 1)) == 0 && part != 0) {
          atomicXor(&s_flag[2], part);
        }
        if (zero_seen) {
          atomicOr(&s_flag[1], 1u);
        }
        lds_barrier();
        if ((s_flag[1] == 0 || p.crc_at_end) && s_flag[2] == 0) {
          iterations = it + 1u;
        }
        lds_barrier(); // the flags are cleared again at the top of the next check
      }
    }
  }
  if (pooled && s_flag[3] != 0xFFFFFFFFu) { // workgroup-uniform
    // Give the slot back once every record store of this workgroup has completed (see acquire_slot): every wave waits for
    // its own outstanding stores -- s_waitcnt vmcnt(0), written out because a workgroup-scope release fence only has to
    // order them with respect to this workgroup (it compiles to a wait for LDS traffic) -- then the barrier, then the flag.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (j == 0) {
      __hip_atomic_store(&p.slot_flags[s_flag[3]], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // Hard bits of the message, packed MSB first (all-zero input: soft <= 0 everywhere, every bit one as in
  // ldpc_decoder_impl.cpp:91-96).
  uint8_t* out = p.out + (size_t)blockIdx.x * p.out_stride;
  for (uint32_t w = j; w < nw_k; w += T) {
    bool           unused = false;
    const uint32_t word   = hard_word(soft, w, K, unused) & topmask(K - 32u * w < 32u ? K - 32u * w : 32u);
    const uint32_t nbytes = (K + 7u) / 8u;
#pragma unroll
    for (uint32_t b = 0; b != 4; ++b) {
      if (4u * w + b < nbytes) {
        out[4u * w + b] = (uint8_t)(word >> (24u - 8u * b));
      }
    }
  }
  if (j == 0 && p.iterations) {
    p.iterations[blockIdx.x] = iterations;
  }
  if (j == 0 && p.ok_flags && iterations != 0) {
    p.ok_flags[blockIdx.x] = 1;
  }
}

__global__ __launch_bounds__(384) __attribute__((amdgpu_waves_per_eu(8))) void ldpc_decode_kernel(LdpcDecodeLaunch p)
{
  ldpc_decode_body<false>(p);
}

// Two checks per lane: half the threads, registers for twice the values per lane (two address arrays, one packed value
// array of 19 edges each).  Measured on one box, BASELINE config 5 with 8 fixed iterations, nrphy_pusch_decode_batch per 256
// slots (profiles/r03_decoder_pairs.txt): one check per lane 6.14 ms; two per lane with the registers of 6 waves per SIMD
// (80 VGPRs, 26 spilled) 5.69, of 5 waves (96, 19 spilled) 5.35, of 4 waves (128, 6 spilled) 5.24, of 3 (153, none) 5.27.
#ifndef NRPHY_DECODER_PAIR_WAVES
#define NRPHY_DECODER_PAIR_WAVES 4
#endif
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(NRPHY_DECODER_PAIR_WAVES))) void ldpc_decode_pairs_kernel(LdpcDecodeLaunch p)
{
  ldpc_decode_body<true>(p);
}

size_t ldpc_decode_lds_bytes(const LdpcDecodeLaunch& p)
{
  return (((size_t)p.nof_nodes * p.zc + 15u) & ~(size_t)15u) + 48u;
}

hipError_t launch_ldpc_decode(const LdpcDecodeLaunch& p, uint32_t n_cb, hipStream_t stream)
{
  if (n_cb == 0) {
    return hipSuccess;
  }
  // NRPHY_DECODER_PAIRS=0: one check per lane whatever the lifting size (A/B runs; the results are identical).
  const char*        pairs_env = std::getenv("NRPHY_DECODER_PAIRS"); // (read per launch: the tests run both kernels in one process)
  const bool         pairs     = (p.zc & 1u) == 0 && p.zc >= 4u && !(pairs_env != nullptr && pairs_env[0] == '0');
  const uint32_t     checks    = pairs ? p.zc / 2u : p.zc;
  const uint32_t     threads   = ((checks + WAVE - 1) / WAVE) * WAVE;
  const size_t       lds       = ldpc_decode_lds_bytes(p);
  const void*        kernel    = pairs ? reinterpret_cast<const void*>(ldpc_decode_pairs_kernel) : reinterpret_cast<const void*>(ldpc_decode_kernel);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      return e;
    }
  }
  if (pairs) {
    hipLaunchKernelGGL(ldpc_decode_pairs_kernel, dim3(n_cb), dim3(threads), lds, stream, p);
  } else {
    hipLaunchKernelGGL(ldpc_decode_kernel, dim3(n_cb), dim3(threads), lds, stream, p);
  }
  return hipGetLastError();
}

} // namespace nrphy
