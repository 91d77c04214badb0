// LDPC decoder for gfx950 (MI355X): layered scaled min-sum on int8 log-likelihood ratios ("next" row, SURVEY.md
// section 8f-1, receive side).
//
// Replaces ldpc_decoder_impl::decode with the generic message kernels
// (R/lib/phy/upper/channel_coding/ldpc/ldpc_decoder_impl.cpp:60-318, ldpc_decoder_generic.cpp:30-128; LLR arithmetic
// R/lib/phy/upper/log_likelihood_ratio.cpp:37-87) bit for bit: same clamps, same saturating / promoting sums, same
// tie-breaking of the two minima, same rounding of the scaled magnitude, same early stop on the CRC.
//
// One workgroup per codeblock, thread j owns lifted position j of every block (Zc <= 384 threads).  The soft bits and
// the variable-to-check messages of the layer in flight live in LDS.  The check-to-variable messages are kept in the
// compressed form the min-sum rule allows -- per lifted check its two scaled magnitudes, the edge holding the minimum
// and one sign bit per edge: 8 bytes instead of up to 20 -- in a global scratch area (mostly L2).  A layer is
//   1. v2c[edge][j]  = soft[var][j] - c2v(edge)[j]          (c2v rebuilt from the record of check (j - shift) mod Zc)
//   2. min-sum over the rotated v2c of check j               -> new record of check j
//   3. soft[var][j]  = c2v(edge)[j] (+) v2c[edge][j]          (promotion sum)
// with two workgroup barriers.
#include "bits_device.h"

namespace nrphy {

constexpr int LLR_MAX_V = 120;
constexpr int LLR_INF_V = 127;

__device__ __forceinline__ bool llr_isinf(int v)
{
  return v > LLR_MAX_V || v < -LLR_MAX_V;
}

// Saturating (PROMOTE = false) or promoting (PROMOTE = true) LLR sum.
template <bool PROMOTE>
__device__ __forceinline__ int llr_sum(int a, int b)
{
  if (a == -b) {
    return 0;
  }
  if (llr_isinf(a)) {
    return a;
  }
  if (llr_isinf(b)) {
    return b;
  }
  const int r   = a + b;
  const int top = PROMOTE ? LLR_INF_V : LLR_MAX_V;
  return r > LLR_MAX_V ? top : (r < -LLR_MAX_V ? -top : r);
}

// c2v message of edge t at a position served by check record `rec` (lo: min1 | min2 << 8 | idx << 16, hi: signs).
__device__ __forceinline__ int c2v_value(uint2 rec, uint32_t t)
{
  const int m1 = (int)(int8_t)(rec.x & 0xFFu), m2 = (int)(int8_t)((rec.x >> 8) & 0xFFu);
  const int v  = (t != ((rec.x >> 16) & 0xFFu)) ? m1 : m2;
  return ((rec.y >> t) & 1u) ? -v : v;
}

__global__ __launch_bounds__(384) void ldpc_decode_kernel(LdpcDecodeLaunch p)
{
  extern __shared__ __attribute__((aligned(16))) int8_t dec_lds[];
  const uint32_t zc = p.zc, j = threadIdx.x;
  const uint32_t n_hr = p.bg_k + 4u;
  int8_t*        soft = dec_lds;                                 // [n_nodes][zc]
  int8_t*        v2c  = dec_lds + (size_t)p.nof_nodes * zc;      // [n_hr + 1][zc]
  uint32_t*      crcw = reinterpret_cast<uint32_t*>(v2c + (size_t)(n_hr + 1u) * zc + 16u); // packed hard bits
  crcw                = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(crcw) + 15u) & ~(uintptr_t)15u);
  __shared__ uint32_t s_flag[4];

  const auto*   graph = to_constant(p.graph); // wave-uniform reads: scalar loads
  const int8_t* llr = p.llr + (size_t)blockIdx.x * p.llr_stride;
  uint2*        rec = p.scratch + (size_t)blockIdx.x * p.nof_layers_max * zc;
  const bool    active = j < zc;

  // load_soft_bits (ldpc_decoder_impl.cpp:128-164): two punctured nodes, whole nodes clamped to +-64, the tail as is.
  // The last non-zero soft bit decides how many layers take part (:88-116).
  uint32_t last_nz = 0;
  if (active) {
    const uint32_t full = p.nof_llr / zc;
    for (uint32_t n = 0; n != p.nof_nodes; ++n) {
      int v = 0;
      if (n >= 2u) {
        const uint32_t i = (n - 2u) * zc + j;
        if (i < p.nof_llr) {
          v = llr[i];
          if (v != 0) {
            last_nz = i + 1u;
          }
          if (n - 2u < full) {
            v = v > 64 ? 64 : (v < -64 ? -64 : v);
          }
        }
      }
      soft[n * zc + j] = (int8_t)v;
    }
  }
  if (j < 4) {
    s_flag[j] = 0;
  }
  __syncthreads();
  if (active && last_nz != 0) {
    atomicMax(&s_flag[0], last_nz);
  }
  __syncthreads();
  const uint32_t input_size = s_flag[0];
  const uint32_t K          = p.bg_k * zc;
  uint32_t       iterations = 0;
  if (input_size != 0) { // workgroup-uniform
    uint32_t cb_len = input_size + 2u * zc;
    cb_len          = cb_len < K + 4u * zc ? K + 4u * zc : cb_len;
    cb_len          = ((cb_len + zc - 1u) / zc) * zc;
    const uint32_t nof_layers = cb_len / zc - p.bg_k;

    for (uint32_t it = 0; it != p.max_iterations && iterations == 0; ++it) {
      for (uint32_t m = 0; m != nof_layers; ++m) {
        const uint32_t e0 = graph->row_ptr[m], deg = graph->row_ptr[m + 1u] - e0;
        uint2*         layer_rec = rec + (size_t)m * zc;
        // 1. variable-to-check messages
        if (active) {
          for (uint32_t t = 0; t != deg; ++t) {
            const uint32_t edge = graph->edge[e0 + t], var = edge >> 16, shift = edge & 0xFFFFu;
            const uint32_t slot = var < n_hr ? var : n_hr;
            int            s    = soft[var * zc + j];
            if (it != 0) {
              uint32_t k = j + zc - shift;
              k          = k >= zc ? k - zc : k;
              s          = llr_sum<false>(s, -c2v_value(layer_rec[k], t));
            }
            v2c[slot * zc + j] = (int8_t)s;
          }
        }
        __syncthreads();
        // 2. two smallest magnitudes, their owner, the sign product; the record of check j
        uint2 mine = make_uint2(0, 0);
        if (active) {
          int      min1 = LLR_MAX_V, min2 = LLR_MAX_V;
          uint32_t idx = 0, neg = 0;
          for (uint32_t t = 0; t != deg; ++t) {
            const uint32_t edge = graph->edge[e0 + t], var = edge >> 16, shift = edge & 0xFFFFu;
            const uint32_t slot = var < n_hr ? var : n_hr;
            uint32_t       k    = j + shift;
            k                   = k >= zc ? k - zc : k;
            const int  v        = v2c[slot * zc + k];
            const int  a        = v < 0 ? -v : v;
            const bool is_min   = a < min1;
            const int  new2     = is_min ? min1 : a;
            min2                = (a < min2) ? new2 : min2;
            idx                 = is_min ? t : idx;
            min1                = is_min ? a : min1;
            neg |= (v < 0 ? 1u : 0u) << t;
          }
          // scale_llr (ldpc_decoder_generic.cpp:69-79): infinities pass, the rest is rounded half away from zero
          const int s1 = llr_isinf(min1) ? min1 : (int)roundf((float)min1 * p.scaling_factor);
          const int s2 = llr_isinf(min2) ? min2 : (int)roundf((float)min2 * p.scaling_factor);
          // sign of the message on edge t = product of all signs x the sign of that edge's own input
          const uint32_t all = (deg >= 32u) ? 0xFFFFFFFFu : ((1u << deg) - 1u);
          const uint32_t signs = (__popc(neg) & 1u) ? (neg ^ all) : neg;
          mine = make_uint2((uint32_t)(s1 & 0xFF) | ((uint32_t)(s2 & 0xFF) << 8) | (idx << 16), signs);
          layer_rec[j] = mine;
        }
        __threadfence_block();
        __syncthreads();
        // 3. soft bits
        if (active) {
          for (uint32_t t = 0; t != deg; ++t) {
            const uint32_t edge = graph->edge[e0 + t], var = edge >> 16, shift = edge & 0xFFFFu;
            const uint32_t slot = var < n_hr ? var : n_hr;
            uint32_t       k    = j + zc - shift;
            k                   = k >= zc ? k - zc : k;
            const int c         = c2v_value(layer_rec[k], t);
            soft[var * zc + j]  = (int8_t)llr_sum<true>(c, (int)v2c[slot * zc + j]);
          }
        }
      }
      // Early stop (ldpc_decoder_impl.cpp:118-126): every hard bit decided and the CRC of the significant bits zero.
      if (p.crc_order != 0) {
        __syncthreads();
        if (j < 2) {
          s_flag[1 + j] = 0;
        }
        __syncthreads();
        const uint32_t n   = K - p.nof_filler;
        const uint32_t pad = (32u - (n & 31u)) & 31u, nw = (n + pad) >> 5;
        bool           zero_seen = false;
        for (uint32_t w = j; w < nw; w += blockDim.x) { // word w holds message bits [32 w - pad, 32 w - pad + 32)
          uint32_t word = 0;
          for (uint32_t b = 0; b != 32; ++b) {
            const int32_t i = (int32_t)(32u * w + b) - (int32_t)pad;
            if (i >= 0) {
              const int s = soft[i];
              zero_seen |= s == 0;
              word |= (s <= 0 ? 1u : 0u) << (31u - b);
            }
          }
          crcw[w] = word;
        }
        // zeros among the filler bits count too (get_hard_bits looks at all Kb * Zc soft bits)
        for (uint32_t i = n + j; i < K; i += blockDim.x) {
          zero_seen |= soft[i] == 0;
        }
        if (zero_seen) {
          atomicOr(&s_flag[1], 1u);
        }
        __syncthreads();
        if (j < WAVE) { // the first wavefront divides the packed message by the generator polynomial
          const CrcPoly c   = {p.crc_poly, p.crc_order};
          uint32_t      reg = 0;
          if (j == 0) {
            const uint32_t mask = (1u << c.order) - 1u, top = 1u << c.order;
            for (uint32_t w = 0; w != nw; ++w) {
              const uint32_t word = crcw[w];
              for (int b = 31; b >= 0; --b) {
                reg = (reg << 1) | ((word >> b) & 1u);
                if (reg & top) {
                  reg ^= c.poly;
                }
              }
            }
            reg &= mask;
            s_flag[2] = reg;
          }
        }
        __syncthreads();
        if (s_flag[1] == 0 && s_flag[2] == 0) {
          iterations = it + 1u;
        }
      }
    }
  } else if (p.crc_order == 0 && active) {
    // All-zero input and nobody to tell: every bit one (ldpc_decoder_impl.cpp:91-96); soft <= 0 yields exactly that.
  }
  __syncthreads();
  // Hard bits of the message, packed MSB first.
  uint8_t* out = p.out + (size_t)blockIdx.x * p.out_stride;
  for (uint32_t byte = j; byte < (K + 7u) / 8u; byte += blockDim.x) {
    uint32_t v = 0;
    for (uint32_t b = 0; b != 8; ++b) {
      const uint32_t i = 8u * byte + b;
      if (i < K) {
        v |= (soft[i] <= 0 ? 1u : 0u) << (7u - b);
      }
    }
    out[byte] = (uint8_t)v;
  }
  if (j == 0 && p.iterations) {
    p.iterations[blockIdx.x] = iterations;
  }
}

size_t ldpc_decode_lds_bytes(const LdpcDecodeLaunch& p)
{
  const size_t K = (size_t)p.bg_k * p.zc;
  return (size_t)p.nof_nodes * p.zc + (size_t)(p.bg_k + 5u) * p.zc + 32u + 4u * ((K + 31u) / 32u + 2u) + 16u;
}

hipError_t launch_ldpc_decode(const LdpcDecodeLaunch& p, uint32_t n_cb, hipStream_t stream)
{
  if (n_cb == 0) {
    return hipSuccess;
  }
  const uint32_t threads = ((p.zc + WAVE - 1) / WAVE) * WAVE;
  const size_t   lds     = ldpc_decode_lds_bytes(p);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ldpc_decode_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      return e;
    }
  }
  hipLaunchKernelGGL(ldpc_decode_kernel, dim3(n_cb), dim3(threads), lds, stream, p);
  return hipGetLastError();
}

} // namespace nrphy
