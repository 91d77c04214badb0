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
//
// Three forms of the kernel, same values in each: one check per lane (ldpc_decode_kernel: odd lifting sizes); two checks per
// lane in packed 16-bit arithmetic (ldpc_decode_pairs_kernel); and two checks per lane with the messages of every edge kept in
// LDS and the soft-bit addresses read from a table (ldpc_decode_pairs_lm_kernel: codeblocks that run few layers -- high code
// rates --, decided per codeblock; see "messages kept per edge in LDS" below).
#include "bits_device.h"

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

// LDS accesses of the pair kernels by LDS address (an integer): through a pointer into the kernel's LDS array every access
// carries an addition of the array's (link-time) address -- "v_add 0, x" once per soft bit read and written.
typedef __attribute__((address_space(3))) int8_t   lds_i8_t;
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
__device__ __forceinline__ int lds_load_i8(uint32_t addr)
{
  return *reinterpret_cast<const lds_i8_t*>((uintptr_t)addr);
}
__device__ __forceinline__ void lds_store_i8(uint32_t addr, uint32_t v)
{
  *reinterpret_cast<lds_i8_t*>((uintptr_t)addr) = (int8_t)v;
}
__device__ __forceinline__ uint32_t lds_load_u32(uint32_t addr)
{
  return *reinterpret_cast<const lds_u32_t*>((uintptr_t)addr);
}
__device__ __forceinline__ void lds_store_u32(uint32_t addr, uint32_t v)
{
  *reinterpret_cast<lds_u32_t*>((uintptr_t)addr) = v;
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

// The soft bits (or messages) of both checks on edges B .. B + 9 of a layer, read together and combined into packed pairs
// (low halves: A | B << 16).  The empty asm statement makes all twenty values be in their registers at one point: without it
// the compiler's scheduler sinks every pair of LDS reads to its use and reuses two temporaries -- a chain of DEG dependent LDS
// round trips per pass (read, wait for everything, combine, read ...) instead of twenty reads in flight.
template <uint32_t DEG, uint32_t B, typename AddrA, typename AddrB>
__device__ __forceinline__ void load_pairs10(uint32_t (&out)[DEG], AddrA addr_a, AddrB addr_b)
{
  if constexpr (B < DEG) {
    constexpr uint32_t N = DEG - B < 10u ? DEG - B : 10u;
    uint32_t           va[10], vb[10];
#pragma unroll
    for (uint32_t k = 0; k != 10; ++k) {
      const uint32_t t = k < N ? B + k : B + N - 1u;
      va[k]            = (uint32_t)lds_load_i8(addr_a(t));
      vb[k]            = (uint32_t)lds_load_i8(addr_b(t));
    }
    asm volatile("" ::"v"(va[0]), "v"(vb[0]), "v"(va[1]), "v"(vb[1]), "v"(va[2]), "v"(vb[2]), "v"(va[3]), "v"(vb[3]), "v"(va[4]), "v"(vb[4]),
                 "v"(va[5]), "v"(vb[5]), "v"(va[6]), "v"(vb[6]), "v"(va[7]), "v"(vb[7]), "v"(va[8]), "v"(vb[8]), "v"(va[9]), "v"(vb[9]));
#pragma unroll
    for (uint32_t k = 0; k != N; ++k) {
      out[B + k] = __builtin_amdgcn_perm(vb[k], va[k], 0x05040100u);
    }
  }
}

// (ds_read_i8_d16 / _d16_hi would place the two bytes in the halves of one register without the v_perm -- not on this device:
// with SRAM ECC enabled, as on MI300 / MI355X, a d16 load writes the whole register and clears the other half, which is why the
// compiler never selects them; tried in assembly in round 4, wrong results.)
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
  static __device__ __forceinline__ void backward(const uint32_t (&addr1)[DEG], const uint32_t (&addr2)[DEG],
                                                  const uint32_t (&x)[DEG], uint32_t m1, uint32_t m2, uint32_t hot0, uint32_t hot1,
                                                  uint32_t z, uint32_t w)
  {
    if constexpr (T < DEG) {
      const s16x2    sum = pair_message<T>(m1, m2, hot0, hot1, z, w) + as_s16x2(x[T]);
      const uint32_t out = as_word(clamp_s16x2(sum, LLR_INF_V));
      lds_store_i8(addr1[T], out);
      lds_store_i8(addr2[T], out >> 16);
      PairEdges<DEG, T + 1, FIRST>::backward(addr1, addr2, x, m1, m2, hot0, hot1, z, w);
    }
  }
};

// Checks j and j + half of one layer (degree DEG); j < half = Zc / 2; jm = j - Zc (wraps).
template <uint32_t DEG, bool FIRST>
__device__ __forceinline__ uint4 process_check_pair(uint32_t soft, const uint8_t* scaled, const NRPHY_CONSTANT uint32_t* edge,
                                                    uint32_t half, uint32_t minus_half, uint32_t j, uint32_t jm, uint4 old)
{
  uint32_t addr1[DEG], addr2[DEG], a[DEG], x[DEG];
#pragma unroll
  for (uint32_t t = 0; t != DEG; ++t) {
    const uint32_t e = edge[t], shift = e & 0xFFFFu;
    const uint32_t pos = min(j + shift, jm + shift);   // (j + shift) mod Zc
    addr1[t]           = (soft + (e >> 16)) + pos;     // LDS address (the graph holds node * Zc; the sum in brackets is scalar)
    addr2[t]           = addr1[t] + (pos < half ? half : minus_half); // (j + Zc / 2 + shift) mod Zc
  }
  load_pairs10<DEG, 0>(a, [&](uint32_t t) { return addr1[t]; }, [&](uint32_t t) { return addr2[t]; });
  load_pairs10<DEG, 10>(a, [&](uint32_t t) { return addr1[t]; }, [&](uint32_t t) { return addr2[t]; });
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
  PairEdges<DEG, 0, FIRST>::backward(addr1, addr2, x, n1, n2, nh0, nh1, nz, nw);
  return mine;
}

template <bool FIRST>
__device__ __forceinline__ uint4 process_layer_pair(uint32_t deg, uint32_t soft, const uint8_t* scaled,
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

// ---- Two checks per lane, check-to-variable messages kept per edge in LDS ------------------------------------------------
// When the layers a codeblock needs are few enough (high code rates: BASELINE config 5 runs 4 layers of degree 19), the
// messages of every edge fit the LDS next to the soft bits: one byte per (edge, check), the messages of two edges and the
// two checks of a lane in one word -- bytes (edge t: check j, check j + Zc / 2; edge t + 1: the same), word r = t / 2 of a
// layer at its row r, lane j.  The edge passes then read the old message instead of rebuilding it from a compressed record
// (7 vector instructions per pair of checks and edge in the forward pass), the sign of a new message comes from the sign of
// the variable-to-check value it answers (new = P * sign(x) * magnitude with P the parity of all signs, folded into the two
// magnitudes once per layer), no record is packed, stored or prefetched, and the scaling of the minima is arithmetic where the
// host has checked that it equals the table (a table look-up is one more LDS round trip per layer, and under this kernel's
// LDS load a round trip costs several hundred cycles).  Same values as process_check_pair, operation for operation.
__device__ __forceinline__ uint32_t pk_mad_i16(uint32_t a, uint32_t b, uint32_t c)
{
  uint32_t r;
  asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
  return r;
}
// llr_sub for both halves with the shift and the addition as one multiply-add.
__device__ __forceinline__ s16x2 llr_sub_pair_mad(s16x2 a, s16x2 c, uint32_t k512)
{
  const s16x2 d   = clamp_s16x2(a - c, LLR_MAX_V);
  const s16x2 big = a - clamp_s16x2(a, LLR_MAX_V); // 0, or +-1 for an infinite soft bit
  return as_s16x2(pk_mad_i16(as_word(big), k512, as_word(d)));
}
__device__ __forceinline__ s16x2 llr_sub_pair_first_mad(s16x2 a, uint32_t k512)
{
  const s16x2 d = clamp_s16x2(a, LLR_MAX_V);
  return as_s16x2(pk_mad_i16(as_word(a - d), k512, as_word(d)));
}

// scale_llr (ldpc_decoder_generic.cpp:69-79) of the magnitudes of a pair of keys (magnitude << 5 | edge in each half) ->
// A | B << 16.  Three rules, each used only where the host has checked that it gives round(m * scaling_factor) for every m the
// decoder can meet (0 .. LLR_MAX): packed 16-bit fixed point (m * F + 256) >> 9 -- three instructions for both halves --, float
// arithmetic, or the table in LDS (one more LDS round trip per layer).
struct ScaleRule {
  const uint8_t* table;      // round(m * scaling_factor), m = 0 .. LLR_MAX
  float          factor;
  uint32_t       mode;       // 2: fixed point with `fixed`, 1: float arithmetic, 0: the table
  uint32_t       fixed;      // F in both halves
  __device__ __forceinline__ uint32_t operator()(uint32_t keys) const
  {
    const u16x2 m = as_u16x2(keys) >> u16x2{5, 5};
    if (mode == 2u) {
      return as_word((m * as_u16x2(fixed) + u16x2{256, 256}) >> u16x2{9, 9});
    }
    const uint32_t ma = as_word(m) & 0xFFFFu, mb = as_word(m) >> 16;
    if (mode == 1u) {
      const uint32_t ra = (uint32_t)__fadd_rn(__fmul_rn((float)ma, factor), 0.5f), rb = (uint32_t)__fadd_rn(__fmul_rn((float)mb, factor), 0.5f);
      return ra | (rb << 16);
    }
    return (uint32_t)table[ma] | ((uint32_t)table[mb] << 16);
  }
};

// One-hot masks of the edges holding the minima from a pair of keys (edge index in the low five bits of each half; 31: none, a
// bit no edge tests).
template <uint32_t DEG>
__device__ __forceinline__ void key_one_hot(uint32_t keys, uint32_t& hot0, uint32_t& hot1)
{
  if constexpr (DEG <= 16u) { // (edge 31 -> 1 << 15 in a 16-bit shift: bit 15, beyond the edges of such a check)
    hot0 = as_word(u16x2{1, 1} << (as_u16x2(keys) & u16x2{15, 15}));
    hot1 = 0;
  } else {
    const uint32_t ia = keys & 31u, ib = (keys >> 16) & 31u;
    const uint32_t ha = 1u << ia, hb = 1u << ib;
    hot0              = (ha & 0xFFFFu) | (hb << 16);
    hot1              = (ha >> 16) | (hb & 0xFFFF0000u);
  }
}

// -DNRPHY_DEC_TRACE (profiles/probes/decoder_trace_run.py): one wave of one codeblock adds up the cycles between the marks TR(k) of
// a layer (s_memtime; ~300 cycles of its own per mark) and leaves the sums in the launch's scratch.
#ifdef NRPHY_DEC_TRACE
struct Trace { uint64_t acc[10]; uint64_t last; };
#define TR(k)                                                                                                        \
  do {                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    if (tr != nullptr) {                                                                                             \
      const uint64_t t_ = __builtin_readcyclecounter();                                                              \
      tr->acc[k] += t_ - tr->last;                                                                                   \
      tr->last = t_;                                                                                                 \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  } while (0)
#else
struct Trace;
#define TR(k)
#endif

#ifndef NRPHY_DEC_SELECT_BY_MASK
#define NRPHY_DEC_SELECT_BY_MASK 0 // 1: the message magnitude selected through a full-width mask (the form until late round 4), for A/B
#endif
template <uint32_t DEG, uint32_t T, bool FIRST>
struct LmEdges {
  static __device__ __forceinline__ void forward(const uint32_t (&c)[DEG], uint32_t (&x)[DEG], uint32_t k512, uint32_t& k1, uint32_t& k2,
                                                 uint32_t& par)
  {
    if constexpr (T < DEG) {
      const s16x2 v = FIRST ? llr_sub_pair_first_mad(as_s16x2(x[T]), k512) : llr_sub_pair_mad(as_s16x2(x[T]), as_s16x2(c[T]), k512);
      x[T]          = as_word(v);
      // key = magnitude * 32 + T in both halves: ten bits of magnitude (an infinite soft bit's value is below 640) over five of
      // edge index -- no clamp of the magnitude; the edge index as an inline constant (as (mag << 5) | literal it costs a move)
      const s16x2 mag = __builtin_elementwise_max(v, splat_s16(0) - v);
      uint32_t    key;
      asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(key) : "v"(as_word(mag)), "s"(k512 >> 4), "n"(T));
      k2 = as_word(__builtin_elementwise_min(__builtin_elementwise_max(as_u16x2(key), as_u16x2(k1)), as_u16x2(k2)));
      k1 = as_word(__builtin_elementwise_min(as_u16x2(key), as_u16x2(k1)));
      // bit 15 of a half of `par`: parity of the negative values so far -- two edges per instruction (a ^ b ^ c as one v_bitop3)
      if constexpr (T % 2u == 1u) {
        par = __builtin_amdgcn_bitop3_b32(par, x[T - 1u], x[T], 0x96);
      } else if constexpr (T + 1u == DEG) {
        par ^= x[T];
      }
      LmEdges<DEG, T + 1, FIRST>::forward(c, x, k512, k1, k2, par);
    }
  }
  // The new message of the pair on edge T and the soft bits it leads to; m1 / m2: the scaled minima of the pair with the
  // parity of the signs applied (negative when odd).
  static __device__ __forceinline__ uint32_t answer(const uint32_t (&addr1)[DEG], const uint32_t (&addr2)[DEG],
                                                    const uint32_t (&x)[DEG], uint32_t m1, uint32_t m2, uint32_t hot0, uint32_t hot1)
  {
    const uint32_t s   = as_word(as_s16x2(x[T]) >> splat_s16(15)) | 0x00010001u; // -1 / +1: the sign of the value answered
#if NRPHY_DEC_SELECT_BY_MASK
    const uint32_t sel = half_masks(T < 16u ? hot0 : hot1, T & 15u);
    const uint32_t mag = __builtin_amdgcn_bitop3_b32(sel, m2, m1, 0xCA);
#else
    // the second minimum for the edge that holds the first: m1 + bit * (m2 - m1) in every half, `m2` arriving here as the
    // difference (process_check_pair_lm) -- a shift and a mask (plain VOP2) and one packed multiply-add instead of two packed
    // shifts for a full-width mask and a select
    const uint32_t bit = ((T < 16u ? hot0 : hot1) >> (T & 15u)) & 0x00010001u;
    uint32_t       mag;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(mag) : "v"(bit), "v"(m2), "v"(m1));
#endif
    const s16x2    msg = as_s16x2(as_word(as_u16x2(mag) * as_u16x2(s)));
    const uint32_t out = as_word(clamp_s16x2(msg + as_s16x2(x[T]), LLR_INF_V));
    lds_store_i8(addr1[T], out);
    lds_store_i8(addr2[T], out >> 16);
    return as_word(msg);
  }
  // Edges T and T + 1 (T even): new soft bits, and the word of their messages (store(r, word): row r of the layer).
  template <typename Store>
  static __device__ __forceinline__ void backward(const Store& store, const uint32_t (&addr1)[DEG], const uint32_t (&addr2)[DEG],
                                                  const uint32_t (&x)[DEG], uint32_t m1, uint32_t m2, uint32_t hot0, uint32_t hot1)
  {
    if constexpr (T < DEG) {
      const uint32_t lo = answer(addr1, addr2, x, m1, m2, hot0, hot1);
      uint32_t       hi = 0;
      if constexpr (T + 1 < DEG) {
        hi = LmEdges<DEG, T + 1, FIRST>::answer(addr1, addr2, x, m1, m2, hot0, hot1);
      }
      store(T / 2u, __builtin_amdgcn_perm(hi, lo, 0x06040200u)); // bytes: A(T), B(T), A(T + 1), B(T + 1)
      LmEdges<DEG, T + 2, FIRST>::backward(store, addr1, addr2, x, m1, m2, hot0, hot1);
    }
  }
};

// Where the messages of a codeblock live.  MsgLds: behind its soft bits in LDS (row r of a layer at mrow + r * row_bytes, the
// lane's own word).  MsgSlot: in the codeblock's slot of the caller's scratch, [row][lane] words, when the LDS has no room for
// them -- every lane reads back only what it wrote itself, an iteration later; the words of a layer's first PF rows are
// requested a layer ahead by the caller (`pre`), the rest (layers of degree 19 only) at the start of the layer.  The stores are
// write-through at agent scope like the check records' (see acquire_slot: a slot passes from workgroup to workgroup).
struct MsgLds {
  uint32_t mrow, row_bytes;
  __device__ __forceinline__ uint32_t load(uint32_t r) const { return lds_load_u32(mrow + r * row_bytes); }
  __device__ __forceinline__ void     operator()(uint32_t r, uint32_t word) const { lds_store_u32(mrow + r * row_bytes, word); }
};
template <uint32_t PF>
struct MsgSlot {
  uint32_t* row0;      // the layer's first row (wave-uniform: the accesses take it as a scalar base, the lane as a 32-bit offset)
  uint32_t  row_words; // lanes per row
  uint32_t  lane;
  uint32_t  pre[PF != 0 ? PF : 1]; // rows 0 .. PF - 1 as requested a layer ahead
  __device__ __forceinline__ uint32_t load(uint32_t r) const { return r < PF ? pre[r] : (row0 + r * row_words)[lane]; }
  __device__ __forceinline__ void     operator()(uint32_t r, uint32_t word) const
  {
    __hip_atomic_store(&(row0 + r * row_words)[lane], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};

// msg: the layer's messages (MsgLds / MsgSlot).  aq: the lane's soft-bit addresses on the layer's edges from the graph's table
// (LdpcDecodeLaunch::pair_addr; the kernel's soft bits start at LDS address 0), four edges per element: check j in the low
// half, check j + Zc / 2 in the high half.
// ahead(): called between the two edge passes -- the caller requests the next layer's soft-bit addresses there: early enough
// for the backward pass to cover the trip to L2, late enough not to hold twenty more registers through the forward pass.
template <uint32_t DEG, bool FIRST, uint32_t NQ, typename Msg, typename Ahead>
__device__ __forceinline__ void process_check_pair_lm(const Msg& msg, const ScaleRule& scale, const uint4 (&aq)[NQ], uint32_t k512, Trace* tr,
                                                      const Ahead& ahead)
{
  static_assert(DEG <= 4u * NQ, "the address rows of the layer");
  constexpr uint32_t NP = (DEG + 1u) / 2u;
  uint32_t           addr1[DEG], addr2[DEG], x[DEG], c[DEG], w[NP];
  TR(0);
  if (!FIRST) { // the lane's own words of old messages: on their way while the addresses are computed
#pragma unroll
    for (uint32_t r = 0; r != NP; ++r) {
      w[r] = msg.load(r);
    }
  }
#pragma unroll
  for (uint32_t t = 0; t != DEG; ++t) {
    const uint4&   q = aq[t / 4u];
    const uint32_t a = (t % 4u) == 0 ? q.x : (t % 4u) == 1 ? q.y : (t % 4u) == 2 ? q.z : q.w;
    addr1[t]         = a & 0xFFFFu;
    addr2[t]         = a >> 16;
  }
  TR(1);
  load_pairs10<DEG, 0>(x, [&](uint32_t u) { return addr1[u]; }, [&](uint32_t u) { return addr2[u]; });
  load_pairs10<DEG, 10>(x, [&](uint32_t u) { return addr1[u]; }, [&](uint32_t u) { return addr2[u]; });
  TR(2);
  if (!FIRST) {
#define NRPHY_IX(k) ((k) < NP ? (k) : NP - 1u)
    asm volatile("" ::"v"(w[NRPHY_IX(0)]), "v"(w[NRPHY_IX(1)]), "v"(w[NRPHY_IX(2)]), "v"(w[NRPHY_IX(3)]), "v"(w[NRPHY_IX(4)]),
                 "v"(w[NRPHY_IX(5)]), "v"(w[NRPHY_IX(6)]), "v"(w[NRPHY_IX(7)]), "v"(w[NRPHY_IX(8)]), "v"(w[NRPHY_IX(9)]));
#undef NRPHY_IX
    // bytes to packed 16-bit pairs, sign-extended by the permute itself: it takes a sign from bytes 1, 3, 5, 7 of its two
    // sources, which the word and the word shifted by a byte place all four messages on
#pragma unroll
    for (uint32_t r = 0; r != NP; ++r) {
      const uint32_t ws = w[r] << 8;
      c[2u * r]         = __builtin_amdgcn_perm(w[r], ws, 0x0A050804u);
      if (2u * r + 1u < DEG) {
        c[2u * r + 1u] = __builtin_amdgcn_perm(w[r], ws, 0x0B070906u);
      }
    }
  } else {
#pragma unroll
    for (uint32_t t = 0; t != DEG; ++t) {
      c[t] = 0;
    }
  }
  TR(3);
  uint32_t k1 = (((uint32_t)LLR_MAX_V << 5) | 31u) * 0x00010001u, k2 = k1, par = 0; // (edge 31: none)
  LmEdges<DEG, 0, FIRST>::forward(c, x, k512, k1, k2, par);
  TR(4);
  // scale_llr of the four minima; the sign of a message = parity of the OTHER signs
  const uint32_t n1 = scale(k1), n2 = scale(k2);
  const uint32_t pm = as_word(as_s16x2(par) >> splat_s16(15)); // 0xFFFF in a half with an odd number of negative values
  const uint32_t m1 = as_word(as_s16x2(n1 ^ pm) - as_s16x2(pm)), m2 = as_word(as_s16x2(n2 ^ pm) - as_s16x2(pm));
  uint32_t       hot0, hot1;
  key_one_hot<DEG>(k1, hot0, hot1);
  ahead();
  TR(5);
#if NRPHY_DEC_SELECT_BY_MASK
  LmEdges<DEG, 0, FIRST>::backward(msg, addr1, addr2, x, m1, m2, hot0, hot1);
#else
  LmEdges<DEG, 0, FIRST>::backward(msg, addr1, addr2, x, m1, as_word(as_s16x2(m2) - as_s16x2(m1)), hot0, hot1);
#endif
  TR(6);
}

// MAXDEG: the largest row degree of the base graph the kernel was built for (19: base graph 1, 10: base graph 2).
template <bool FIRST, uint32_t MAXDEG, uint32_t NQ, typename Msg, typename Ahead>
__device__ __forceinline__ void process_layer_pair_lm(uint32_t deg, const Msg& msg, const ScaleRule& scale, const uint4 (&aq)[NQ],
                                                      uint32_t k512, Trace* tr, const Ahead& ahead)
{
  switch (deg) {
    case 3: return process_check_pair_lm<3, FIRST>(msg, scale, aq, k512, tr, ahead);
    case 4: return process_check_pair_lm<4, FIRST>(msg, scale, aq, k512, tr, ahead);
    case 5: return process_check_pair_lm<5, FIRST>(msg, scale, aq, k512, tr, ahead);
    case 6: return process_check_pair_lm<6, FIRST>(msg, scale, aq, k512, tr, ahead);
    case 7: return process_check_pair_lm<7, FIRST>(msg, scale, aq, k512, tr, ahead);
    case 8: return process_check_pair_lm<8, FIRST>(msg, scale, aq, k512, tr, ahead);
    case 9: return process_check_pair_lm<9, FIRST>(msg, scale, aq, k512, tr, ahead);
    case 10: return process_check_pair_lm<10, FIRST>(msg, scale, aq, k512, tr, ahead);
    default:
      if constexpr (MAXDEG > 10u) {
        return process_check_pair_lm<19, FIRST>(msg, scale, aq, k512, tr, ahead);
      }
      return;
  }
}

// One soft bit as it enters the decoder (ldpc_decoder_impl.cpp:128-164): whole nodes are clamped to +-64, the tail
// is taken as is (infinities in this kernel's form, see above).
__device__ __forceinline__ int load_soft(int v, bool whole_node)
{
  return whole_node ? med3(v, -64, 64) : med3(v, -LLR_INF_V, LLR_INF_V);
}

// Hard bits [32 w, 32 w + 32) of the soft bits (MSB first; a soft bit <= 0 decides for one); zero_seen: an undecided one among
// the first `limit`.  Four soft bits of a word at a time: bit 7 of a byte of ((x & 0x7F..) + 0x7F..) | x says "not zero", the
// sign says "negative", and a dot product with the weights 128 .. 1 gathers eight of them into a byte.
__device__ __forceinline__ uint32_t hard_word(const int8_t* soft, uint32_t w, uint32_t limit, bool& zero_seen)
{
  const uint4    lo = *reinterpret_cast<const uint4*>(soft + 32u * w);
  const uint4    hi = *reinterpret_cast<const uint4*>(soft + 32u * w + 16u);
  const uint32_t x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  uint32_t       word = 0, zeros = 0;
#pragma unroll
  for (uint32_t k = 0; k != 8; k += 2) {
    uint32_t one[2], zero[2];
#pragma unroll
    for (uint32_t h = 0; h != 2; ++h) {
      const uint32_t nz = ((x[k + h] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x[k + h]; // bit 7 of a byte: the soft bit is not zero
      zero[h]           = (~nz & 0x80808080u) >> 7;
      one[h]            = ((x[k + h] | ~nz) & 0x80808080u) >> 7;              // negative or zero
    }
    const uint32_t shift = 24u - 4u * k;
    word |= __builtin_amdgcn_udot4(one[0], 0x10204080u, __builtin_amdgcn_udot4(one[1], 0x01020408u, 0u, false), false) << shift;
    zeros |= __builtin_amdgcn_udot4(zero[0], 0x10204080u, __builtin_amdgcn_udot4(zero[1], 0x01020408u, 0u, false), false) << shift;
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
// LM: the check-to-variable messages kept per edge instead of as compressed records -- behind the soft bits in LDS when the
// layers a codeblock runs leave room for them there, else in the codeblock's slot of the caller's scratch (decided per
// codeblock; the record path is not part of such a kernel).
// MAXDEG (LM only): the largest row degree of the base graph -- 19 (base graph 1) or 10 (base graph 2).
// SLOT_ONLY (LM only): a launch without room for messages in LDS -- every codeblock keeps them in its slot, the first rows of a
// layer's old messages requested a layer ahead; otherwise the launch's LDS was sized for the messages of the expected layers and
// only a codeblock that runs more than those (stale soft bits in a HARQ buffer) takes its slot, without the requests ahead.
template <bool PAIR, bool LM = false, uint32_t MAXDEG = 19, bool SLOT_ONLY = false>
__device__ __forceinline__ void ldpc_decode_body(const LdpcDecodeLaunch& p)
{
  // All of the kernel's LDS is the launch's dynamic allocation, the soft bits at its start: with a static variable in front
  // of them every soft-bit address would carry the variable's size as an extra addition (two per edge and pass).
  extern __shared__ __attribute__((aligned(16))) int8_t dec_lds[];
  uint32_t* const s_flag   = reinterpret_cast<uint32_t*>(dec_lds + p.lds_tail_off); // [4]
  uint8_t* const  s_scaled = reinterpret_cast<uint8_t*>(dec_lds + p.lds_tail_off + 16u); // [128]
  const uint32_t zc = p.zc, j = threadIdx.x;
  int8_t*        soft = dec_lds; // [nof_nodes][zc] (+ 32 bytes of slack for the word reads of the last hard bits)
  const uint32_t soft_a = (uint32_t)(uintptr_t)(lds_i8_t*)dec_lds; // ... as an LDS address (see lds_load_i8)

  if (p.skip != nullptr && p.skip[blockIdx.x] != 0) { // workgroup-uniform: decoded in an earlier transmission
    return;
  }
  const auto*   graph  = to_constant(p.graph); // wave-uniform reads: scalar loads
  const int8_t* llr    = p.llr + (size_t)blockIdx.x * p.llr_stride;
  const uint32_t half = zc >> 1;
  const bool    active = PAIR ? j < half : j < zc;
  const bool    pooled = p.nof_slots < gridDim.x; // fewer slots than codeblocks: claim one
  // With two checks per lane the messages may live in LDS instead (decided below, once the layers are known): the claim
  // then waits until it is known to be needed; otherwise its latency hides behind the loads.
  const bool    claim_late = LM && p.lm_lds_bytes != 0;
  if (j == 0) { // (read after the barriers below)
    s_flag[3] = pooled ? (claim_late ? 0xFFFFFFFFu : acquire_slot(p.slot_flags, p.nof_slots, blockIdx.x)) : blockIdx.x;
  }

  // load_soft_bits (ldpc_decoder_impl.cpp:128-164): two punctured nodes, then the input.  The last non-zero soft bit
  // decides how many layers take part (:88-116).
  uint32_t       last_nz   = 0;
  const uint32_t T         = blockDim.x;
  const uint32_t clamp_end = (p.nof_llr / zc) * zc; // whole nodes
  const uint32_t lds_bytes = p.nof_nodes * zc + 48u;
  if ((zc & 3u) == 0) { // (lds_bytes is a multiple of four then)
    for (uint32_t i = 4u * j; i < 2u * zc; i += 4u * T) {
      *reinterpret_cast<uint32_t*>(soft + i) = 0;
    }
    const uint32_t end = 2u * zc + p.nof_llr, up = (end + 3u) & ~3u;
    if (j < up - end) {
      soft[end + j] = 0;
    }
    for (uint32_t i = up + 4u * j; i < lds_bytes; i += 4u * T) {
      *reinterpret_cast<uint32_t*>(soft + i) = 0;
    }
  } else {
    for (uint32_t i = j; i < 2u * zc; i += T) {
      soft[i] = 0;
    }
    for (uint32_t i = 2u * zc + p.nof_llr + j; i < lds_bytes; i += T) {
      soft[i] = 0;
    }
  }
  // One word of four soft bits (index `first` .. first + 3) as it goes into LDS.
  auto convert_word = [&](uint32_t x, uint32_t first) -> uint32_t {
    if (x == 0) { // (at high code rates most of the buffer: nothing received there yet)
      return 0;
    }
    last_nz    = max(last_nz, first + 4u - ((uint32_t)__clz(x) >> 3));
    uint32_t y = 0;
#pragma unroll
    for (uint32_t b = 0; b != 4; ++b) {
      const int v = load_soft((int)(int8_t)(x >> (8u * b)), first + b < clamp_end);
      y |= ((uint32_t)v & 0xFFu) << (8u * b);
    }
    return y;
  };
  // The same for a word that lies on one side of clamp_end (lifting sizes that are multiples of four), in packed 16-bit
  // arithmetic and without a branch: the sixteen-byte loads below run it on every word of a non-zero quadruple.
  auto convert_word_aligned = [&](uint32_t x, uint32_t first) -> uint32_t {
    last_nz = x != 0 ? max(last_nz, first + 4u - ((uint32_t)__clz(x) >> 3)) : last_nz;
    const int      lim = first < clamp_end ? 64 : LLR_INF_V;
    const uint32_t xs  = x << 8;
    const s16x2    lo  = clamp_s16x2(as_s16x2(__builtin_amdgcn_perm(x, xs, 0x0A050804u)), lim); // bytes 0, 1 sign-extended
    const s16x2    hi  = clamp_s16x2(as_s16x2(__builtin_amdgcn_perm(x, xs, 0x0B070906u)), lim); // bytes 2, 3
    return __builtin_amdgcn_perm(as_word(hi), as_word(lo), 0x06040200u);
  };
  if (((reinterpret_cast<uintptr_t>(llr) & 15u) | (zc & 7u)) == 0) {
    // sixteen soft bits per lane and load, up to twelve (four) loads of a lane in flight: the whole codeblock in one or two trips to
    // memory (four-byte loads unrolled by four took eight trips for a config-3 codeblock)
    constexpr uint32_t U   = PAIR ? 12 : 4; // (the one-check kernel runs at 64 registers)
    const uint32_t     nq  = p.nof_llr >> 4;
    const uint4*       src = reinterpret_cast<const uint4*>(llr);
    uint4*             dst = reinterpret_cast<uint4*>(soft + 2u * zc);
    for (uint32_t q0 = j; q0 < nq; q0 += U * T) {
      uint4 v[U];
#pragma unroll
      for (uint32_t u = 0; u != U; ++u) {
        const uint32_t q = q0 + u * T;
        v[u]             = q < nq ? src[q] : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (uint32_t u = 0; u != U; ++u) {
        const uint32_t q = q0 + u * T;
        if (q < nq) {
          uint4 y = make_uint4(0, 0, 0, 0);
          if ((v[u].x | v[u].y | v[u].z | v[u].w) != 0) {
            y = make_uint4(convert_word_aligned(v[u].x, 16u * q), convert_word_aligned(v[u].y, 16u * q + 4u),
                           convert_word_aligned(v[u].z, 16u * q + 8u), convert_word_aligned(v[u].w, 16u * q + 12u));
          }
          dst[q] = y;
        }
      }
    }
    const uint32_t i = 16u * nq + j; // (fewer than sixteen left)
    if (i < p.nof_llr) {
      const int v = llr[i];
      last_nz     = v != 0 ? max(last_nz, i + 1u) : last_nz;
      soft[2u * zc + i] = (int8_t)load_soft(v, i < clamp_end);
    }
  } else if (((reinterpret_cast<uintptr_t>(llr) | zc) & 3u) == 0) {
    // four soft bits per lane and load
    const uint32_t  nd  = p.nof_llr >> 2;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(llr);
    uint32_t*       dst = reinterpret_cast<uint32_t*>(soft + 2u * zc);
#pragma unroll 4
    for (uint32_t d = j; d < nd; d += T) {
      dst[d] = convert_word(src[d], 4u * d);
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

  // Early stop: the message is a multiple of the generator polynomial.  Thread w owns hard-bit word w; the tables of its
  // weight x^(bits after the word) mod g come from the host.
  const uint32_t n_msg  = K - p.nof_filler;
  const uint32_t jm = j - zc;

  if (input_size != 0) { // workgroup-uniform
    uint32_t cb_len = input_size + 2u * zc;
    cb_len          = cb_len < K + 4u * zc ? K + 4u * zc : cb_len;
    cb_len          = ((cb_len + zc - 1u) / zc) * zc;
    const uint32_t nof_layers = cb_len / zc - p.bg_k;
    // Messages per edge in LDS (process_check_pair_lm) when the layers this codeblock runs leave room for them behind the
    // soft bits those layers touch (the rows beyond are all zero and never read again): workgroup-uniform, decided per
    // codeblock from its own soft bits, so the result never depends on the launch's LDS budget.
    const uint32_t msg_off = (cb_len + 48u + 15u) & ~15u;
    // (the table of soft-bit addresses counts from LDS address 0, where this kernel's only LDS array starts)
    const bool     lm      = LM && !SLOT_ONLY && p.lm_lds_bytes != 0 && msg_off + graph->pair_ptr[nof_layers] * 2u * zc <= p.lm_lds_bytes;
    if (claim_late && !lm && pooled) {
      if (j == 0) {
        s_flag[3] = acquire_slot(p.slot_flags, p.nof_slots, blockIdx.x);
      }
      __syncthreads();
    }
    const uint32_t slot = lm ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)s_flag[3]); // (an LDS read: uniform, but not to the compiler)
    uint8_t* const slot_mem = reinterpret_cast<uint8_t*>(p.scratch) + (size_t)(slot == 0xFFFFFFFFu ? 0u : slot) * p.slot_bytes;
    // no slot: reported as not decoded (and likewise should the soft bits ever not start at LDS address 0, which the table of
    // soft-bit addresses of the message kernels assumes)
    const uint32_t max_iterations = (slot == 0xFFFFFFFFu || (LM && soft_a != 0)) ? 0u : p.max_iterations;
    const uint32_t  msgs  = soft_a + msg_off + 4u * j; // LDS address of the lane's word: [row of two edges][lane], four bytes
    const uint32_t  k512  = 0x02000200u;
    const ScaleRule scale = {s_scaled, p.scaling_factor, p.scale_arithmetic, p.scale_fixed * 0x00010001u};
#ifdef NRPHY_DEC_TRACE
    Trace  trace = {};
    Trace* tr    = (blockIdx.x == gridDim.x / 2u && j < 64u) ? &trace : nullptr;
    if (tr) trace.last = __builtin_readcyclecounter();
#else
    Trace* tr = nullptr;
#endif

    typedef typename std::conditional<PAIR, uint4, uint2>::type Record; // a pair of checks per lane has a record of twice the size
    Record*        recs        = reinterpret_cast<Record*>(slot_mem);
    const uint32_t rec_stride  = PAIR ? half : zc;                       // records of one layer
    const uint32_t minus_half  = 0u - half;
    constexpr uint32_t NQ = MAXDEG > 12u ? 5u : 3u; // rows of four soft-bit addresses a layer can have
    constexpr uint32_t PF  = SLOT_ONLY ? (MAXDEG > 12u ? 5u : 3u) : 0u; // rows of messages requested a layer ahead (messages in the slot)
    constexpr uint32_t PFA = PF != 0u ? PF : 1u;    // (array extent)
    // (messages kept per edge) the first layer's soft-bit addresses and, in the slot, its old messages: requested at the end of
    // the previous iteration
    uint4    wrap[NQ]    = {};
    uint32_t wrap_pre[PFA] = {};
    for (uint32_t it = 0; it != max_iterations && iterations == 0; ++it) {
      if constexpr (LM) {
        // The lane's soft-bit addresses of a layer: NQ rows of the table (sixteen bytes per lane and row: four edges),
        // requested a layer ahead.
        const uint32_t jj   = active ? j : half - 1u; // (idle lanes stay inside the tables)
        // (uniform base + 32-bit lane index: the loads take their base from scalar registers instead of a 64-bit pointer per
        // lane and table row held -- or spilled -- across the iteration)
        const uint4* const atab = reinterpret_cast<const uint4*>(p.pair_addr);
        uint32_t* const    gmsg = reinterpret_cast<uint32_t*>(slot_mem); // (messages in the slot) row 0
        uint4          cur[NQ];
        uint32_t       pre[PFA];
        if (it == 0) {
#pragma unroll
          for (uint32_t q = 0; q != NQ; ++q) {
            cur[q] = (atab + q * half)[jj];
          }
#pragma unroll
          for (uint32_t r = 0; r != PF; ++r) {
            pre[r] = 0;
          }
        } else {
#pragma unroll
          for (uint32_t q = 0; q != NQ; ++q) {
            cur[q] = wrap[q];
          }
#pragma unroll
          for (uint32_t r = 0; r != PF; ++r) {
            pre[r] = wrap_pre[r];
          }
        }
        uint32_t e0 = graph->row_ptr[0], e1 = graph->row_ptr[1], rows = 0, quads = 0;
        // One layer with the addresses in `now` (and, in the slot, the first rows of its old messages in `pre_now`); the
        // next layer's (after the last: the first's, for the next iteration) are requested into `next` / `pre_next` before
        // the layer starts.  Two calls per trip with the arrays swapped: no copies.
        auto layer = [&](uint32_t m, const uint4 (&now)[NQ], uint4 (&next)[NQ], const uint32_t (&pre_now)[PFA],
                         uint32_t (&pre_next)[PFA]) __attribute__((always_inline)) {
          const uint32_t e2  = graph->row_ptr[m + 2u]; // (the array has a spare element) a layer ahead: off the critical path
          const uint32_t deg = e1 - e0;
          const bool     last      = m + 1u == nof_layers;
          const uint32_t rows_next = last ? 0u : rows + ((deg + 1u) >> 1);
          quads                    = last ? 0u : quads + ((deg + 3u) >> 2);
          auto ahead = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (uint32_t q = 0; q != NQ; ++q) {
              next[q] = (atab + (quads + q) * half)[jj];
            }
          };
          if constexpr (PF != 0u) { // (the rows read past a layer's own lie inside the slot: it has five spare rows)
#pragma unroll
            for (uint32_t r = 0; r != PF; ++r) {
              pre_next[r] = (gmsg + (rows_next + r) * half)[jj];
            }
          }
          if (active) {
            bool in_lds = false;
            if constexpr (!SLOT_ONLY) {
              in_lds = lm; // workgroup-uniform
              if (lm) {
                const MsgLds msg = {msgs + rows * 2u * zc, 2u * zc};
                if (it == 0) {
                  process_layer_pair_lm<true, MAXDEG>(deg, msg, scale, now, k512, nullptr, ahead);
                } else {
                  process_layer_pair_lm<false, MAXDEG>(deg, msg, scale, now, k512, tr, ahead);
                }
              }
            }
            if (!in_lds) {
              MsgSlot<PF> msg;
              msg.row0      = gmsg + rows * half;
              msg.row_words = half;
              msg.lane      = jj;
#pragma unroll
              for (uint32_t r = 0; r != PF; ++r) {
                msg.pre[r] = pre_now[r];
              }
              if (it == 0) {
                process_layer_pair_lm<true, MAXDEG>(deg, msg, scale, now, k512, nullptr, ahead);
              } else {
                process_layer_pair_lm<false, MAXDEG>(deg, msg, scale, now, k512, tr, ahead);
              }
            }
          }
          rows += (deg + 1u) >> 1;
          lds_barrier();
#ifdef NRPHY_DEC_TRACE
          if (it != 0) { TR(7); } else if (tr) { tr->last = __builtin_readcyclecounter(); }
#endif
          e0 = e1;
          e1 = e2;
        };
        uint4    alt[NQ];
        uint32_t pre_alt[PFA];
        for (uint32_t m = 0; m < nof_layers; m += 2u) {
          layer(m, cur, alt, pre, pre_alt);
          if (m + 1u != nof_layers) {
            layer(m + 1u, alt, cur, pre_alt, pre);
          } else {
#pragma unroll
            for (uint32_t q = 0; q != NQ; ++q) {
              cur[q] = alt[q];
            }
#pragma unroll
            for (uint32_t r = 0; r != PF; ++r) {
              pre[r] = pre_alt[r];
            }
          }
        }
#pragma unroll
        for (uint32_t q = 0; q != NQ; ++q) {
          wrap[q] = cur[q];
        }
#pragma unroll
        for (uint32_t r = 0; r != PF; ++r) {
          wrap_pre[r] = pre[r];
        }
      } else {
        Record next = {};
        if (it != 0 && active) {
          next = recs[j];
        }
        uint32_t e0 = graph->row_ptr[0], e1 = graph->row_ptr[1];
        for (uint32_t m = 0; m != nof_layers; ++m) {
          const uint32_t e2  = graph->row_ptr[m + 2u]; // (the array has a spare element) a layer ahead: off the critical path
          const uint32_t deg = e1 - e0;
          const Record   old = next;
          if (it != 0 && active && m + 1u != nof_layers) {
            next = recs[(size_t)(m + 1u) * rec_stride + j];
          }
          if (active) {
            const auto* edge = graph->edge + e0;
            if constexpr (PAIR) {
              const uint4 mine = it == 0 ? process_layer_pair<true>(deg, soft_a, s_scaled, edge, half, minus_half, j, jm, old)
                                         : process_layer_pair<false>(deg, soft_a, s_scaled, edge, half, minus_half, j, jm, old);
              store_record(&recs[(size_t)m * rec_stride + j], mine);
            } else {
              const uint2 mine = it == 0 ? process_layer<true>(deg, soft, s_scaled, edge, zc, j, jm, old)
                                         : process_layer<false>(deg, soft, s_scaled, edge, zc, j, jm, old);
              store_record(&recs[(size_t)m * rec_stride + j], mine);
            }
          }
          lds_barrier();
          e0 = e1;
          e1 = e2;
        }
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
            // word * x^(bits after it) mod g: eight nibble tables of this word (L2), independent look-ups
            const uint32_t  v = word >> (32u - w_bits);
            const uint32_t* t = p.crc_weight + (size_t)w * DEC_CRC_TABLE_WORDS;
            part ^= t[v & 15u] ^ t[16u + ((v >> 4) & 15u)] ^ t[32u + ((v >> 8) & 15u)] ^ t[48u + ((v >> 12) & 15u)] ^
                    t[64u + ((v >> 16) & 15u)] ^ t[80u + ((v >> 20) & 15u)] ^ t[96u + ((v >> 24) & 15u)] ^ t[112u + (v >> 28)];
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
#ifdef NRPHY_DEC_TRACE
    if (tr && j == 0) {
      for (int k = 0; k != 10; ++k) {
        reinterpret_cast<uint64_t*>(p.scratch)[k] = trace.acc[k];
      }
    }
#endif
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

__global__ __launch_bounds__(384) __attribute__((amdgpu_waves_per_eu(5))) void ldpc_decode_kernel(LdpcDecodeLaunch p)
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
// ... with the messages kept per edge (LDS or slot), one kernel per base graph: base graph 1 has four rows of degree 19 --
// two address arrays, a value array and the words of old messages for nineteen edges -- and runs with the registers of three
// waves per SIMD (all that the LDS of a high-rate launch holds anyway: four workgroups of three waves per CU at BASELINE config
// 5; with 128 registers the compiler serialises the LDS reads of an edge pass through two temporaries); base graph 2 stops at
// degree 10 and fits the registers of four.
#ifndef NRPHY_DECODER_BG1_SLOT_WAVES
#define NRPHY_DECODER_BG1_SLOT_WAVES 3
#endif
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(3))) void ldpc_decode_msg_bg1_kernel(LdpcDecodeLaunch p)
{
  ldpc_decode_body<true, true, 19, false>(p);
}
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(NRPHY_DECODER_BG1_SLOT_WAVES))) void ldpc_decode_msg_bg1_slot_kernel(LdpcDecodeLaunch p)
{
  ldpc_decode_body<true, true, 19, true>(p);
}
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(4))) void ldpc_decode_msg_bg2_kernel(LdpcDecodeLaunch p)
{
  ldpc_decode_body<true, true, 10, false>(p);
}
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(4))) void ldpc_decode_msg_bg2_slot_kernel(LdpcDecodeLaunch p)
{
  ldpc_decode_body<true, true, 10, true>(p);
}

constexpr uint32_t LDS_TAIL_BYTES = 16u + 128u;

size_t ldpc_decode_lds_bytes(const LdpcDecodeLaunch& p)
{
  return (((size_t)p.nof_nodes * p.zc + 15u) & ~(size_t)15u) + 48u;
}

hipError_t launch_ldpc_decode(const LdpcDecodeLaunch& p_in, uint32_t n_cb, hipStream_t stream)
{
  if (n_cb == 0) {
    return hipSuccess;
  }
  LdpcDecodeLaunch p = p_in;
  // NRPHY_DECODER_PAIRS=0 (read when the context is created): one check per lane whatever the lifting size (A/B runs; the results are identical).
  const bool         pairs     = (p.zc & 1u) == 0 && p.zc >= 4u && p.knob_pairs != 0;
  const uint32_t     checks    = pairs ? p.zc / 2u : p.zc;
  const uint32_t     threads   = ((checks + WAVE - 1) / WAVE) * WAVE;
  size_t             lds       = ldpc_decode_lds_bytes(p);
  // Two checks per lane: messages per edge (NRPHY_DECODER_MSG=0: compressed records instead, the round-3 form, for A/B runs).
  // Behind the soft bits in LDS when the expected layers leave the CU at least twelve wavefronts (three per SIMD, where the
  // edge passes still hide their LDS round trips) -- NRPHY_DECODER_LDSMSG=2: whenever a workgroup's LDS can hold them at all
  // (tests), =0: never --, else in the codeblock's slot of the scratch.
  const bool         msg       = pairs && p.pair_addr != nullptr && (p.bg_k == 22u || p.bg_k == 10u) && p.knob_msg != 0;
  const uint32_t     waves     = threads / WAVE;
  const uint32_t     lm_cap    = p.knob_ldsmsg == 2 ? 160u * 1024u : ((160u * 1024u) / ((12u + waves - 1u) / waves)) & ~255u;
  if (msg && p.lm_lds_bytes != 0 && p.lm_lds_bytes + LDS_TAIL_BYTES <= lm_cap && p.knob_ldsmsg != 0) {
    lds            = lds > p.lm_lds_bytes ? lds : (size_t)p.lm_lds_bytes;
    p.lm_lds_bytes = (uint32_t)lds;
  } else {
    p.lm_lds_bytes = 0;
  }
  p.lds_tail_off = (uint32_t)lds; // the flags and the scaling table behind the soft bits (and messages)
  lds += LDS_TAIL_BYTES;
  typedef void (*kernel_t)(LdpcDecodeLaunch);
  const bool         in_lds    = p.lm_lds_bytes != 0;
  const kernel_t     kernel    = msg     ? (p.bg_k == 22u ? (in_lds ? ldpc_decode_msg_bg1_kernel : ldpc_decode_msg_bg1_slot_kernel)
                                                          : (in_lds ? ldpc_decode_msg_bg2_kernel : ldpc_decode_msg_bg2_slot_kernel))
                                 : pairs ? ldpc_decode_pairs_kernel
                                         : ldpc_decode_kernel;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      return e;
    }
  }
  hipLaunchKernelGGL(kernel, dim3(n_cb), dim3(threads), lds, stream, p);
  return hipGetLastError();
}

} // namespace nrphy
