// LDPC rate dematcher for gfx950 (MI355X) ("next" row, SURVEY.md section 8f-1, receive side).
//
// Replaces ldpc_rate_dematcher_impl::rate_dematch (R/lib/phy/upper/channel_coding/ldpc/ldpc_rate_dematcher_impl.cpp:43-256):
// bit deinterleaver, then the walk over the circular soft buffer from k0 that copies (new data, first visit) or adds
// with the saturating LLR sum (HARQ combining, later visits of a repetition), skips the filler bits and sets them to
// +infinity.
//
// The reference walks the buffer sequentially; which stretches it clears, fills, copies into, adds to or leaves alone
// depends only on the configuration, not on the data.  The host runs the same walk once per call and hands the kernel
// the resulting list of range operations (DematchOp, nrphy_host.cpp: build_dematch_ops); the kernel is a gather: one
// thread owns four consecutive soft bits of one codeblock, applies to them, in order, every operation that covers them
// and writes them back once.  Positions nothing covers keep their content, as in the reference.
#include "bits_device.h"

namespace nrphy {

// Element s of the deinterleaved input: row j = s / cols of the Qm x cols table is bit j of every symbol
// (ldpc_rate_dematcher_impl.cpp:202-213).
__device__ __forceinline__ int dematch_fetch(const int8_t* in, uint32_t s, uint32_t qm, uint32_t cols)
{
  if (qm == 1) {
    return in[s];
  }
  const uint32_t j = s / cols, i = s - j * cols;
  return in[i * qm + j];
}

// in + old as log_likelihood_ratio::operator+ evaluates it (log_likelihood_ratio.cpp:37-80): opposite values cancel,
// an infinite input wins over an infinite buffer value, the rest saturates at +-LLR_MAX.
__device__ __forceinline__ int dematch_sum(int in, int old)
{
  const bool in_inf = in > 120 || in < -120, old_inf = old > 120 || old < -120;
  const int  s      = max(-120, min(in + old, 120));
  const int  r      = in_inf ? in : (old_inf ? old : s);
  return in == -old ? 0 : r;
}

template <uint32_t VEC>
__global__ __launch_bounds__(256) void ldpc_dematch_kernel(DematchLaunch p)
{
  const uint32_t first = (blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (first >= p.block_length) {
    return;
  }
  const int8_t* in  = p.in + (size_t)blockIdx.y * p.in_stride;
  int8_t*       out = p.out + (size_t)blockIdx.y * p.out_stride + first;
  int           v[VEC];
  if (VEC == 4) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(out);
#pragma unroll
    for (uint32_t i = 0; i != VEC; ++i) {
      v[i] = (int)(int8_t)(w >> (8u * i));
    }
  } else {
    v[0] = out[0];
  }
  bool touched = false;
  for (uint32_t k = 0; k != p.n_ops; ++k) {
    const DematchOp op = p.ops[k];
    if (first + VEC <= op.begin || first >= op.begin + op.count) {
      continue;
    }
    touched = true;
#pragma unroll
    for (uint32_t i = 0; i != VEC; ++i) {
      const uint32_t q = first + i;
      if (q >= op.begin && q < op.begin + op.count) {
        switch (op.kind) {
          case DEMATCH_ZERO:
            v[i] = 0;
            break;
          case DEMATCH_FILL:
            v[i] = 127; // LLR_INFINITY: a filler bit is a certain zero
            break;
          case DEMATCH_COPY:
            v[i] = dematch_fetch(in, op.src + (q - op.begin), p.qm, p.cols);
            break;
          default:
            v[i] = dematch_sum(dematch_fetch(in, op.src + (q - op.begin), p.qm, p.cols), v[i]);
            break;
        }
      }
    }
  }
  if (!touched) {
    return;
  }
  if (VEC == 4) {
    uint32_t w = 0;
#pragma unroll
    for (uint32_t i = 0; i != VEC; ++i) {
      w |= ((uint32_t)v[i] & 0xFFu) << (8u * i);
    }
    *reinterpret_cast<uint32_t*>(out) = w;
  } else {
    out[0] = (int8_t)v[0];
  }
}

hipError_t launch_ldpc_dematch(const DematchLaunch& p, uint32_t n_cb, hipStream_t stream)
{
  if (n_cb == 0 || p.n_ops == 0) {
    return hipSuccess;
  }
  // four soft bits per thread when every codeblock row is dword aligned (block lengths are multiples of 4 only for even Zc)
  const bool vec4 = ((reinterpret_cast<uintptr_t>(p.out) | p.out_stride | p.block_length) & 3u) == 0;
  if (vec4) {
    const uint32_t blocks = (p.block_length / 4 + 255) / 256;
    hipLaunchKernelGGL(ldpc_dematch_kernel<4>, dim3(blocks, n_cb), dim3(256), 0, stream, p);
  } else {
    const uint32_t blocks = (p.block_length + 255) / 256;
    hipLaunchKernelGGL(ldpc_dematch_kernel<1>, dim3(blocks, n_cb), dim3(256), 0, stream, p);
  }
  return hipGetLastError();
}

} // namespace nrphy
