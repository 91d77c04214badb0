// Soft demodulator ("demodulation mapper") for gfx950.
//
// Replaces demodulation_mapper::demodulate_soft (R/include/srsran/phy/upper/channel_modulation/demodulation_mapper.h;
// R/lib/phy/upper/channel_modulation/demodulation_mapper_impl.cpp:33-106 and demodulation_mapper_{qpsk,qam16,qam64,qam256}.cpp).
// The reference's soft bits depend on where a symbol lies in the span handed over: the leading floor(n / B) * B symbols go
// through its AVX2 code (B = 16 QPSK, 8 16-QAM, 16 64-QAM, 4 256-QAM), the rest through its generic code, and the two round
// differently (mi355_nrphy.h, nrphy_demodulate_soft).  A thread takes two neighbouring symbols of one span and picks the
// arithmetic by their index, so every soft bit equals the reference's at the same position.
//
// HBM-bound elementwise work: 12 bytes in (symbol + noise variance) and Qm bytes out per symbol; tables (1 KB) in LDS.
#include "nrphy_internal.h"

#include <hip/hip_runtime.h>

namespace nrphy {
namespace {

constexpr float NEAR_ZERO = 1e-9f;
constexpr float LLR_MAXF  = 120.f;

// log_likelihood_ratio::quantize (R/lib/phy/upper/log_likelihood_ratio.cpp:89-98): round half away from zero.
__device__ __forceinline__ int quantize_generic(float value, float range)
{
  const float clipped = fabsf(value) > range ? copysignf(range, value) : value;
  return (int)roundf(__fmul_rn(__fdiv_rn(clipped, range), LLR_MAXF));
}
// mm256::quantize_ps (R/lib/phy/upper/channel_modulation/avx2_helpers.h:118-170): scale, clip, nearest even, NaN -> 0.
__device__ __forceinline__ int quantize_vector(float value, float scale /* 120 / range */)
{
  const float v = __fmul_rn(value, scale);
  const float c = __builtin_fmaxf(-LLR_MAXF, __builtin_fminf(v, LLR_MAXF)); // v_med3_f32; a NaN is replaced below
  return v != v ? 0 : (int)rintf(c);
}

__device__ __forceinline__ float safe_rcp(float noise)
{
  return noise > 0.f ? __fdiv_rn(1.0f, noise) : 0.f;
}

struct Tables {
  float2 line[DEMOD_MAX_PAIRS][16]; // (slope, intercept) of an interval: one 8-byte LDS read per soft bit
};

// The soft bits of a thread (up to 16) collected in registers: byte k of the thread's output.  Every index is a compile-time
// constant after unrolling (a byte array here ends up in LDS, one ds_write_b8 per soft bit).
struct LlrBytes {
  uint32_t w[4] = {0, 0, 0, 0};
  __device__ __forceinline__ void put(uint32_t k, int v) { w[k >> 2] |= ((uint32_t)v & 0xFFu) << (8u * (k & 3u)); }
  __device__ __forceinline__ uint8_t get(uint32_t k) const { return (uint8_t)(w[k >> 2] >> (8u * (k & 3u))); }
};

// One component (real or imaginary part) of a table-driven constellation, bit pair `pair`.  VECTOR: the reference's AVX2
// arithmetic (reciprocal width, nearest-even quantiser), else its generic one.
template <bool VECTOR>
__device__ __forceinline__ int interval_llr(const DemodLaunch& p, const Tables& t, uint32_t pair, float v, float rcp)
{
  const int    n    = (int)p.nof_intervals[pair];
  const float  pos  = VECTOR ? __fmul_rn(v, p.rcp_width[pair]) : __fdiv_rn(v, p.width[pair]);
  const int    idx  = max(0, min((int)floorf(pos) + n / 2, n - 1));
  const float2 line = t.line[pair][idx];
  const float  l    = __fmul_rn(__fmaf_rn(line.x, v, line.y), rcp);
  if (VECTOR) {
    return quantize_vector(fabsf(v) <= NEAR_ZERO ? 0.f : l, p.scale);
  }
  return quantize_generic(l, p.range);
}

// The soft bits of symbol i of its span into bytes [at, at + qm) of out.  The kernel is specialised per modulation (no run-time
// switch, the bit-pair loop unrolled) and per arithmetic: a thread whose symbols all lie in the span's vector part -- every
// thread but the last few of a span -- runs the VECTOR copy.
template <uint32_t MOD, bool VECTOR>
__device__ __forceinline__ void demodulate_symbol(const DemodLaunch& p, const Tables& t, uint32_t i, float re, float im, float noise,
                                                  LlrBytes& out, uint32_t at)
{
  constexpr float GAIN_PSK = 2.0f * 1.41421356237309504880f;
  if constexpr (MOD == NRPHY_MOD_BPSK || MOD == NRPHY_MOD_PI2_BPSK) {
    // pi/2-BPSK: odd symbols are rotated by -90 degrees first, (im, -re)
    const bool  rot = MOD == NRPHY_MOD_PI2_BPSK && (i & 1u);
    const float a = rot ? im : re, b = rot ? -re : im;
    out.put(at, !(noise > 0.f) ? 0 : quantize_generic(__fdiv_rn(__fmul_rn(GAIN_PSK, __fadd_rn(a, b)), noise), p.range));
  } else if constexpr (MOD == NRPHY_MOD_QPSK) {
#pragma unroll
    for (uint32_t c = 0; c != 2; ++c) {
      const float v = c ? im : re;
      if (VECTOR) {
        out.put(at + c, quantize_vector(__fmul_rn(__fmul_rn(GAIN_PSK, v), safe_rcp(noise)), p.scale));
      } else {
        out.put(at + c, !(noise > 0.f) ? 0 : quantize_generic(__fdiv_rn(__fmul_rn(GAIN_PSK, v), noise), p.range));
      }
    }
  } else if constexpr (MOD == NRPHY_MOD_QAM16) {
    const float g1 = p.qam16_gain, thr = p.qam16_threshold;
    const bool  blank = !VECTOR && __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im)) < NEAR_ZERO;
#pragma unroll
    for (uint32_t c = 0; c != 2; ++c) {
      const float v     = c ? im : re;
      const float first = __fmul_rn(g1, v);
      // 2 * first is exact, so the reference's contracted and uncontracted forms agree
      const float l01 = fabsf(v) > thr ? __fsub_rn(__fmul_rn(2.0f, first), copysignf(0.8f, v)) : first;
      if (VECTOR) {
        const float rcp  = safe_rcp(noise);
        const bool  zero = fabsf(v) <= NEAR_ZERO;
        out.put(at + c, quantize_vector(zero ? 0.f : __fmul_rn(l01, rcp), p.scale));
        out.put(at + 2u + c, quantize_vector(zero ? 0.f : __fmul_rn(__fsub_rn(0.8f, fabsf(first)), rcp), p.scale));
      } else if (!(blank || !(noise > 0.f))) {
        out.put(at + c, quantize_generic(__fdiv_rn(l01, noise), p.range));
        out.put(at + 2u + c, quantize_generic(__fdiv_rn(__fmaf_rn(-g1, fabsf(v), 0.8f), noise), p.range)); // contracted there
      }
    }
  } else { // 64-QAM, 256-QAM
    constexpr uint32_t pairs = MOD / 2u;
    const float        rcp   = safe_rcp(noise);
    const bool         blank = !VECTOR && __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im)) < NEAR_ZERO;
#pragma unroll
    for (uint32_t k = 0; k != pairs; ++k) {
      out.put(at + 2u * k, blank ? 0 : interval_llr<VECTOR>(p, t, k, re, rcp));
      out.put(at + 2u * k + 1u, blank ? 0 : interval_llr<VECTOR>(p, t, k, im, rcp));
    }
  }
}

template <uint32_t MOD>
__global__ __launch_bounds__(256) void demodulate_soft_kernel(DemodLaunch p, const float2* __restrict__ d_symbols,
                                                              const float* __restrict__ d_noise, int8_t* __restrict__ d_llr)
{
  __shared__ Tables t;
  if constexpr (MOD >= NRPHY_MOD_QAM64) {
    for (uint32_t k = threadIdx.x; k < DEMOD_MAX_PAIRS * 16u; k += blockDim.x) {
      t.line[k / 16u][k % 16u] = make_float2(p.slope[k / 16u][k % 16u], p.intercept[k / 16u][k % 16u]);
    }
    __syncthreads();
  }
  constexpr uint32_t qm    = MOD == NRPHY_MOD_PI2_BPSK ? 1u : MOD;
  const uint32_t     span  = blockIdx.y;
  const uint32_t     first = 2u * (blockIdx.x * blockDim.x + threadIdx.x); // the thread's first symbol
  if (first >= p.span_len) {
    return;
  }
  const uint32_t count = first + 1u < p.span_len ? 2u : 1u;
  const size_t   base  = (size_t)span * p.span_len + first;
  LlrBytes       out;
  if (count == 2u && first + 2u <= p.nof_vector) {
    // two symbols of the vector part: one 16-byte and one 8-byte load where the addresses allow
    float2 z0, z1;
    float  n0, n1;
    if (((reinterpret_cast<uintptr_t>(d_symbols + base) & 15u) | (reinterpret_cast<uintptr_t>(d_noise + base) & 7u)) == 0) {
      const float4 zz = *reinterpret_cast<const float4*>(d_symbols + base);
      const float2 nn = *reinterpret_cast<const float2*>(d_noise + base);
      z0 = make_float2(zz.x, zz.y), z1 = make_float2(zz.z, zz.w), n0 = nn.x, n1 = nn.y;
    } else {
      z0 = d_symbols[base], z1 = d_symbols[base + 1], n0 = d_noise[base], n1 = d_noise[base + 1];
    }
    demodulate_symbol<MOD, true>(p, t, first, z0.x, z0.y, n0, out, 0);
    demodulate_symbol<MOD, true>(p, t, first + 1u, z1.x, z1.y, n1, out, qm);
  } else {
#pragma unroll
    for (uint32_t s = 0; s != 2; ++s) {
      if (s < count) {
        const float2 z = d_symbols[base + s];
        if (first + s < p.nof_vector) {
          demodulate_symbol<MOD, true>(p, t, first + s, z.x, z.y, d_noise[base + s], out, s * qm);
        } else {
          demodulate_symbol<MOD, false>(p, t, first + s, z.x, z.y, d_noise[base + s], out, s * qm);
        }
      }
    }
  }
  int8_t*        dst    = d_llr + base * qm;
  const uint32_t nbytes = count * qm;
  if (nbytes == 16u && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
    *reinterpret_cast<uint4*>(dst) = make_uint4(out.w[0], out.w[1], out.w[2], out.w[3]);
  } else if ((nbytes & 3u) == 0 && (reinterpret_cast<uintptr_t>(dst) & 3u) == 0) {
#pragma unroll
    for (uint32_t k = 0; k != 4; ++k) {
      if (4u * k < nbytes) {
        reinterpret_cast<uint32_t*>(dst)[k] = out.w[k];
      }
    }
  } else {
#pragma unroll
    for (uint32_t k = 0; k != 16; ++k) {
      if (k < nbytes) {
        dst[k] = (int8_t)out.get(k);
      }
    }
  }
}

} // namespace

hipError_t launch_demodulate_soft(const DemodLaunch& p, uint32_t nof_spans, const float* d_symbols, const float* d_noise,
                                  int8_t* d_llr, hipStream_t stream)
{
  const uint32_t pairs  = (p.span_len + 1u) / 2u;
  const dim3     grid((pairs + 255u) / 256u, nof_spans), block(256);
  const float2*  sym = reinterpret_cast<const float2*>(d_symbols);
  switch (p.modulation) {
    case NRPHY_MOD_PI2_BPSK:
      hipLaunchKernelGGL(demodulate_soft_kernel<NRPHY_MOD_PI2_BPSK>, grid, block, 0, stream, p, sym, d_noise, d_llr);
      break;
    case NRPHY_MOD_BPSK:
      hipLaunchKernelGGL(demodulate_soft_kernel<NRPHY_MOD_BPSK>, grid, block, 0, stream, p, sym, d_noise, d_llr);
      break;
    case NRPHY_MOD_QPSK:
      hipLaunchKernelGGL(demodulate_soft_kernel<NRPHY_MOD_QPSK>, grid, block, 0, stream, p, sym, d_noise, d_llr);
      break;
    case NRPHY_MOD_QAM16:
      hipLaunchKernelGGL(demodulate_soft_kernel<NRPHY_MOD_QAM16>, grid, block, 0, stream, p, sym, d_noise, d_llr);
      break;
    case NRPHY_MOD_QAM64:
      hipLaunchKernelGGL(demodulate_soft_kernel<NRPHY_MOD_QAM64>, grid, block, 0, stream, p, sym, d_noise, d_llr);
      break;
    case NRPHY_MOD_QAM256:
      hipLaunchKernelGGL(demodulate_soft_kernel<NRPHY_MOD_QAM256>, grid, block, 0, stream, p, sym, d_noise, d_llr);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

} // namespace nrphy
