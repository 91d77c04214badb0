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
  float v = __fmul_rn(value, scale);
  v       = v > LLR_MAXF ? LLR_MAXF : v;
  v       = v < -LLR_MAXF ? -LLR_MAXF : v;
  return v != v ? 0 : (int)rintf(v);
}

__device__ __forceinline__ float safe_rcp(float noise)
{
  return noise > 0.f ? __fdiv_rn(1.0f, noise) : 0.f;
}

struct Tables {
  float slope[DEMOD_MAX_PAIRS][16], intercept[DEMOD_MAX_PAIRS][16];
};

// One component (real or imaginary part) of a table-driven constellation, bit pair p.
__device__ __forceinline__ int interval_llr(const DemodLaunch& p, const Tables& t, uint32_t pair, float v, float rcp, bool vector)
{
  const int   n   = (int)p.nof_intervals[pair];
  const float pos = vector ? __fmul_rn(v, p.rcp_width[pair]) : __fdiv_rn(v, p.width[pair]);
  int         idx = (int)floorf(pos) + n / 2;
  idx             = idx < 0 ? 0 : (idx > n - 1 ? n - 1 : idx);
  const float l   = __fmul_rn(__fmaf_rn(t.slope[pair][idx], v, t.intercept[pair][idx]), rcp);
  if (vector) {
    return quantize_vector(fabsf(v) <= NEAR_ZERO ? 0.f : l, p.scale);
  }
  return quantize_generic(l, p.range);
}

// The soft bits of one symbol into out[0 .. qm).
__device__ __forceinline__ void demodulate_symbol(const DemodLaunch& p, const Tables& t, uint32_t i, float re, float im, float noise,
                                                  int8_t* out)
{
  const bool  vector = i < p.nof_vector;
  constexpr float GAIN_PSK = 2.0f * 1.41421356237309504880f;
  switch (p.modulation) {
    case NRPHY_MOD_BPSK:
    case NRPHY_MOD_PI2_BPSK: {
      // pi/2-BPSK: odd symbols are rotated by -90 degrees first, (im, -re)
      const bool  rot = p.modulation == NRPHY_MOD_PI2_BPSK && (i & 1u);
      const float a = rot ? im : re, b = rot ? -re : im;
      out[0] = !(noise > 0.f) ? 0 : (int8_t)quantize_generic(__fdiv_rn(__fmul_rn(GAIN_PSK, __fadd_rn(a, b)), noise), p.range);
      break;
    }
    case NRPHY_MOD_QPSK: {
#pragma unroll
      for (int c = 0; c != 2; ++c) {
        const float v = c ? im : re;
        if (vector) {
          out[c] = (int8_t)quantize_vector(__fmul_rn(__fmul_rn(GAIN_PSK, v), safe_rcp(noise)), p.scale);
        } else {
          out[c] = !(noise > 0.f) ? 0 : (int8_t)quantize_generic(__fdiv_rn(__fmul_rn(GAIN_PSK, v), noise), p.range);
        }
      }
      break;
    }
    case NRPHY_MOD_QAM16: {
      const float g1 = p.qam16_gain, thr = p.qam16_threshold;
      const bool  blank = !vector && __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im)) < NEAR_ZERO;
#pragma unroll
      for (int c = 0; c != 2; ++c) {
        const float v     = c ? im : re;
        const float first = __fmul_rn(g1, v);
        // 2 * first is exact, so the reference's contracted and uncontracted forms agree
        const float l01 = fabsf(v) > thr ? __fsub_rn(__fmul_rn(2.0f, first), copysignf(0.8f, v)) : first;
        if (vector) {
          const float rcp  = safe_rcp(noise);
          const bool  zero = fabsf(v) <= NEAR_ZERO;
          out[c]           = (int8_t)quantize_vector(zero ? 0.f : __fmul_rn(l01, rcp), p.scale);
          out[2 + c]       = (int8_t)quantize_vector(zero ? 0.f : __fmul_rn(__fsub_rn(0.8f, fabsf(first)), rcp), p.scale);
        } else if (blank || !(noise > 0.f)) {
          out[c] = out[2 + c] = 0;
        } else {
          out[c]     = (int8_t)quantize_generic(__fdiv_rn(l01, noise), p.range);
          out[2 + c] = (int8_t)quantize_generic(__fdiv_rn(__fmaf_rn(-g1, fabsf(v), 0.8f), noise), p.range); // contracted there
        }
      }
      break;
    }
    default: { // 64-QAM, 256-QAM
      const uint32_t pairs = p.modulation / 2u;
      const float    rcp   = safe_rcp(noise);
      const bool     blank = !vector && __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im)) < NEAR_ZERO;
      for (uint32_t k = 0; k != pairs; ++k) {
        out[2 * k]     = blank ? 0 : (int8_t)interval_llr(p, t, k, re, rcp, vector);
        out[2 * k + 1] = blank ? 0 : (int8_t)interval_llr(p, t, k, im, rcp, vector);
      }
    }
  }
}

__global__ __launch_bounds__(256) void demodulate_soft_kernel(DemodLaunch p, const float2* __restrict__ d_symbols,
                                                              const float* __restrict__ d_noise, int8_t* __restrict__ d_llr)
{
  __shared__ Tables t;
  for (uint32_t k = threadIdx.x; k < DEMOD_MAX_PAIRS * 16u; k += blockDim.x) {
    t.slope[k / 16u][k % 16u]     = p.slope[k / 16u][k % 16u];
    t.intercept[k / 16u][k % 16u] = p.intercept[k / 16u][k % 16u];
  }
  __syncthreads();
  const uint32_t span  = blockIdx.y;
  const uint32_t qm    = p.modulation == NRPHY_MOD_PI2_BPSK ? 1u : p.modulation;
  const uint32_t first = 2u * (blockIdx.x * blockDim.x + threadIdx.x); // the thread's first symbol
  if (first >= p.span_len) {
    return;
  }
  const uint32_t count = first + 1u < p.span_len ? 2u : 1u;
  const size_t   base  = (size_t)span * p.span_len + first;
  alignas(16) int8_t out[16];
#pragma unroll
  for (uint32_t s = 0; s != 2; ++s) {
    if (s < count) {
      const float2 z = d_symbols[base + s];
      demodulate_symbol(p, t, first + s, z.x, z.y, d_noise[base + s], out + s * qm);
    }
  }
  int8_t*        dst    = d_llr + base * qm;
  const uint32_t nbytes = count * qm;
  if (nbytes == 16u && ((uintptr_t)dst & 15u) == 0) {
    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(out);
  } else if ((nbytes & 3u) == 0 && ((uintptr_t)dst & 3u) == 0) {
    for (uint32_t k = 0; k != nbytes / 4u; ++k) {
      reinterpret_cast<uint32_t*>(dst)[k] = reinterpret_cast<const uint32_t*>(out)[k];
    }
  } else {
    for (uint32_t k = 0; k != nbytes; ++k) {
      dst[k] = out[k];
    }
  }
}

} // namespace

hipError_t launch_demodulate_soft(const DemodLaunch& p, uint32_t nof_spans, const float* d_symbols, const float* d_noise,
                                  int8_t* d_llr, hipStream_t stream)
{
  const uint32_t pairs  = (p.span_len + 1u) / 2u;
  hipLaunchKernelGGL(demodulate_soft_kernel, dim3((pairs + 255u) / 256u, nof_spans), dim3(256), 0, stream, p,
                     reinterpret_cast<const float2*>(d_symbols), d_noise, d_llr);
  return hipGetLastError();
}

} // namespace nrphy
