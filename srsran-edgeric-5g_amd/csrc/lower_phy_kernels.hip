// Lower-PHY tail for gfx950 (MI355X): what happens to the modulator's samples and to the grid
// on their way to the radio unit.
//
//   amplitude_kernel      amplitude_controller::process -- gain, power measurements, clipping of real and imaginary
//                         parts (R/lib/phy/lower/amplitude_controller/amplitude_controller_clipping_impl.cpp:31-68,
//                         amplitude_controller_scaling_impl.cpp:28-37, R/lib/srsvec/clip.cpp:28-56)
//   convert_ci16_kernel   complex float -> complex int16 with a scale (R/lib/srsvec/conversion.cpp:29-65): the radio's
//                         sample format
//   ofh_compress_kernel   Open Fronthaul compression of resource-grid PRBs: 16-bit quantisation, block floating point
//                         exponent per PRB, bit packing, in the user-plane's byte order
//                         (R/lib/ofh/compression/iq_compression_{none,bfp}_impl.cpp, quantizer.h,
//                         compressed_prb_packer.cpp, R/lib/ofh/serdes/ofh_uplane_message_builder_impl.cpp:137-144)
//
// All three are one pass over their data: HBM-bound streaming kernels (8 B in, 8 or 4 B out per sample; 48 B in,
// 3 w (+ 1) B out per PRB).  The first two also exist fused into the OFDM modulator's store (ofdm_kernels.hip).
#include "bits_device.h"

namespace nrphy {

// One value through the reference's conversion: its 16-lane vector loop rounds to nearest even and saturates
// (_mm256_cvtps_epi32 + _mm256_packs_epi32), the scalar tail of a call rounds half away from zero (std::round).
__device__ __forceinline__ int to_int16_ref(float v, bool vector_lane)
{
  if (vector_lane) {
    const int r = __float2int_rn(v);
    return r > 32767 ? 32767 : (r < -32768 ? -32768 : r);
  }
  return (int)(int16_t)(int)roundf(v);
}

constexpr uint32_t AMP_THREADS = 256;

__global__ __launch_bounds__(AMP_THREADS) void amplitude_kernel(AmplitudeLaunch p)
{
  __shared__ float    s_sum[AMP_THREADS / WAVE], s_peak[AMP_THREADS / WAVE];
  __shared__ uint32_t s_clip[AMP_THREADS / WAVE];
  const uint32_t      tid = threadIdx.x, buffer = blockIdx.y;
  const float2*       in  = reinterpret_cast<const float2*>(p.in) + (size_t)buffer * p.in_stride;
  float2*             out = reinterpret_cast<float2*>(p.out) + (size_t)buffer * p.out_stride;
  float               sum = 0.f, peak = 0.f;
  uint32_t            clipped = 0;
  for (uint32_t i = blockIdx.x * AMP_THREADS + tid; i < p.nof_samples; i += gridDim.x * AMP_THREADS) {
    float2 v = in[i];
    v.x      = __fmul_rn(v.x, p.gain);
    v.y      = __fmul_rn(v.y, p.gain);
    if (p.measure) {
      const float pw = __fadd_rn(__fmul_rn(v.x, v.x), __fmul_rn(v.y, v.y));
      sum += pw;
      peak = fmaxf(peak, pw);
    }
    if (p.clip) {
      if (v.x > p.ceiling) {
        v.x = p.ceiling, ++clipped;
      } else if (v.x < -p.ceiling) {
        v.x = -p.ceiling, ++clipped;
      }
      if (v.y > p.ceiling) {
        v.y = p.ceiling, ++clipped;
      } else if (v.y < -p.ceiling) {
        v.y = -p.ceiling, ++clipped;
      }
    }
    out[i] = v;
  }
  if (p.stats == nullptr || !p.measure) {
    return;
  }
  for (int o = WAVE / 2; o != 0; o >>= 1) {
    sum += __shfl_xor(sum, o, WAVE);
    peak = fmaxf(peak, __shfl_xor(peak, o, WAVE));
    clipped += __shfl_xor(clipped, o, WAVE);
  }
  if ((tid & (WAVE - 1)) == 0) {
    s_sum[tid / WAVE]  = sum;
    s_peak[tid / WAVE] = peak;
    s_clip[tid / WAVE] = clipped;
  }
  __syncthreads();
  if (tid == 0) {
    for (uint32_t w = 1; w != AMP_THREADS / WAVE; ++w) {
      sum += s_sum[w];
      peak = fmaxf(peak, s_peak[w]);
      clipped += s_clip[w];
    }
    nrphy_amplitude_stats_t* st = p.stats + buffer;
    atomicAdd(&st->sum_power, sum);
    atomicMax(reinterpret_cast<uint32_t*>(&st->peak_power), __float_as_uint(peak)); // non-negative floats order like integers
    atomicAdd(&st->nof_clipped, clipped);
    if (blockIdx.x == 0) {
      st->nof_samples = p.nof_samples;
    }
  }
}

hipError_t launch_amplitude(const AmplitudeLaunch& p, uint32_t n_buffers, hipStream_t stream)
{
  if (n_buffers == 0 || p.nof_samples == 0) {
    return hipSuccess;
  }
  uint32_t blocks = (p.nof_samples + AMP_THREADS * 4 - 1) / (AMP_THREADS * 4); // ~4 samples per thread
  blocks          = blocks > 1024 ? 1024 : blocks;
  hipLaunchKernelGGL(amplitude_kernel, dim3(blocks, n_buffers), dim3(AMP_THREADS), 0, stream, p);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void convert_ci16_kernel(const float* __restrict__ in, size_t in_stride, int16_t* __restrict__ out,
                                                           size_t out_stride, uint32_t nof_samples, float scale)
{
  const float2*  src   = reinterpret_cast<const float2*>(in) + (size_t)blockIdx.y * in_stride;
  uint32_t*      dst   = reinterpret_cast<uint32_t*>(out) + (size_t)blockIdx.y * out_stride;
  const uint32_t n_vec = ((2u * nof_samples) / 16u) * 16u; // values the reference converts in its vector loop
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nof_samples; i += gridDim.x * 256u) {
    const float2 v  = src[i];
    const int    re = to_int16_ref(__fmul_rn(v.x, scale), 2u * i < n_vec), im = to_int16_ref(__fmul_rn(v.y, scale), 2u * i + 1u < n_vec);
    dst[i]          = ((uint32_t)re & 0xFFFFu) | ((uint32_t)im << 16);
  }
}

hipError_t launch_convert_ci16(const float* in, size_t in_stride, int16_t* out, size_t out_stride, uint32_t n_buffers,
                               uint32_t nof_samples, float scale, hipStream_t stream)
{
  if (n_buffers == 0 || nof_samples == 0) {
    return hipSuccess;
  }
  uint32_t blocks = (nof_samples + 1023u) / 1024u;
  blocks          = blocks > 1024 ? 1024 : blocks;
  hipLaunchKernelGGL(convert_ci16_kernel, dim3(blocks, n_buffers), dim3(256), 0, stream, in, in_stride, out, out_stride,
                     nof_samples, scale);
  return hipGetLastError();
}

// ---- Open Fronthaul compression --------------------------------------------------------------------------------------
constexpr uint32_t OFH_PRBS_PER_WG = 64;
constexpr uint32_t OFH_MAX_RECORD  = 49; // 3 * 16 + 1

__global__ __launch_bounds__(OFH_PRBS_PER_WG) void ofh_compress_kernel(OfhCompressLaunch p)
{
  __shared__ __attribute__((aligned(4))) uint8_t s_out[OFH_PRBS_PER_WG * OFH_MAX_RECORD + 8];
  const uint32_t  tid = threadIdx.x, prb = blockIdx.x * OFH_PRBS_PER_WG + tid, row = blockIdx.y;
  const uint32_t  w = p.data_width, rec = 3u * w + (p.bfp ? 1u : 0u);
  const uint32_t* src = p.prbs + (size_t)row * p.row_stride + 12u * prb;
  if (prb < p.nof_prb) {
    int q[24];
    // Quantisation (quantizer::to_fixed_point through srsvec::convert, conversion.cpp:202-230): value * scale to int16.
    // The reference's AVX2 compressors convert all PRBs of a call in one go (the first 16 * floor(24 n / 16) values in
    // the vector loop) except for the widths their packer lacks, which go PRB by PRB (16 of 24 in the vector loop).
    const uint32_t n_vec = p.whole_span ? ((24u * p.nof_prb) / 16u) * 16u : 0u;
    int            vmax = -32768, vmin = 32767;
#pragma unroll
    for (uint32_t k = 0; k != 12; ++k) {
      const uint32_t word = src[k];
      const float    re = __uint_as_float(word << 16), im = __uint_as_float(word & 0xFFFF0000u);
      const uint32_t i = 2u * k;
      const bool     v0 = p.whole_span ? 24u * prb + i < n_vec : i < 16u, v1 = p.whole_span ? 24u * prb + i + 1u < n_vec : i + 1u < 16u;
      q[i]              = to_int16_ref(__fmul_rn(re, p.scale), v0);
      q[i + 1]          = to_int16_ref(__fmul_rn(im, p.scale), v1);
      vmax              = max(vmax, max(q[i], q[i + 1]));
      vmin              = min(vmin, min(q[i], q[i + 1]));
    }
    uint8_t* o = s_out + tid * rec;
    uint32_t exponent = 0;
    if (p.bfp) {
      // Block floating point (O-RAN.WG4.CUS Annex A.1.2; iq_compression_bfp_impl.cpp:50-75, .h:63-77): the exponent
      // that makes the largest magnitude of the PRB fit data_width bits, then an arithmetic shift.
      const int      a = abs(vmax), b = abs(vmin) - 1;
      const uint32_t max_abs = (uint32_t)(a > b ? a : b) & 0xFFFFu, max_shift = 16u - w;
      uint32_t       lz = max_shift;
      if (max_abs != 0 && max_shift != 0) {
        lz = (uint32_t)__clz((int)max_abs) - 17u;
      }
      const uint32_t raw = max_shift < lz ? max_shift : lz;
      exponent           = max_shift - raw;
      *o++               = (uint8_t)exponent;
    }
    // compressed_prb_packer::pack: data_width bits per value, most significant bit first.
    uint64_t       acc = 0;
    uint32_t       nbits = 0;
    const uint32_t mask = (1u << w) - 1u;
#pragma unroll
    for (uint32_t i = 0; i != 24; ++i) {
      acc = (acc << w) | (uint64_t)((uint32_t)(q[i] >> exponent) & mask);
      nbits += w;
      while (nbits >= 8u) {
        *o++ = (uint8_t)(acc >> (nbits - 8u));
        nbits -= 8u;
      }
    }
  }
  __syncthreads();
  // The workgroup's records are contiguous in the output: copy them out together, dwords where the alignment allows.
  const uint32_t first = blockIdx.x * OFH_PRBS_PER_WG;
  const uint32_t count = p.nof_prb - first < OFH_PRBS_PER_WG ? p.nof_prb - first : OFH_PRBS_PER_WG;
  const uint32_t bytes = count * rec;
  uint8_t*       dst   = p.out + (size_t)row * p.out_row_stride + (size_t)first * rec;
  if ((reinterpret_cast<uintptr_t>(dst) & 3u) == 0) {
    const uint32_t nd = bytes >> 2;
    for (uint32_t i = tid; i < nd; i += OFH_PRBS_PER_WG) {
      reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(s_out)[i];
    }
    for (uint32_t i = 4u * nd + tid; i < bytes; i += OFH_PRBS_PER_WG) {
      dst[i] = s_out[i];
    }
  } else {
    for (uint32_t i = tid; i < bytes; i += OFH_PRBS_PER_WG) {
      dst[i] = s_out[i];
    }
  }
}

hipError_t launch_ofh_compress(const OfhCompressLaunch& p, uint32_t n_rows, hipStream_t stream)
{
  if (n_rows == 0 || p.nof_prb == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(ofh_compress_kernel, dim3((p.nof_prb + OFH_PRBS_PER_WG - 1) / OFH_PRBS_PER_WG, n_rows), dim3(OFH_PRBS_PER_WG), 0,
                     stream, p);
  return hipGetLastError();
}

} // namespace nrphy
