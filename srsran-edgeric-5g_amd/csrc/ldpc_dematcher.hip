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
// (ldpc_rate_dematcher_impl.cpp:202-213).  s / cols through the float reciprocal with a correction step (s < 2^24).
__device__ __forceinline__ int dematch_fetch(const int8_t* in, uint32_t s, uint32_t qm, uint32_t cols, float rcp_cols)
{
  if (qm == 1) {
    return in[s];
  }
  uint32_t j = (uint32_t)((float)s * rcp_cols);
  int32_t  i = (int32_t)(s - j * cols);
  if (i < 0) {
    i += (int32_t)cols;
    --j;
  } else if (i >= (int32_t)cols) {
    i -= (int32_t)cols;
    ++j;
  }
  return in[(uint32_t)i * qm + j];
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

// Applies, in order, every operation that covers soft bits [first, first + VEC) of one codeblock.  `in` is the
// codeblock's input, in global memory or staged in LDS.
// old: what the buffer holds there (a dword of four soft bits, or one soft bit).  EXT: the operation list is in global
// memory (p.ops_ext) instead of the kernel argument.
template <uint32_t VEC, bool EXT>
__device__ __forceinline__ void dematch_positions(const DematchLaunch& p, const int8_t* in, int8_t* out_row, uint32_t first,
                                                  uint32_t old, float rcp_cols)
{
  int8_t* out = out_row + first;
  int     v[VEC];
#pragma unroll
  for (uint32_t i = 0; i != VEC; ++i) {
    v[i] = (int)(int8_t)(old >> (8u * i));
  }
  bool touched = false;
  for (uint32_t k = 0; k != p.n_ops; ++k) {
    const DematchOp op = EXT ? p.ops_ext[k] : p.ops[k];
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
            v[i] = dematch_fetch(in, op.src + (q - op.begin), p.qm, p.cols, rcp_cols);
            break;
          default:
            v[i] = dematch_sum(dematch_fetch(in, op.src + (q - op.begin), p.qm, p.cols, rcp_cols), v[i]);
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

// General form: the input is gathered from global memory (any alignment, any length).
template <uint32_t VEC, bool EXT>
__global__ __launch_bounds__(256) void ldpc_dematch_kernel(DematchLaunch p)
{
  const uint32_t first = (blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (first >= p.block_length) {
    return;
  }
  int8_t*        out_row = p.out + (size_t)blockIdx.z * p.out_stride_outer + (size_t)blockIdx.y * p.out_stride;
  const uint32_t old     = VEC == 4 ? *reinterpret_cast<const uint32_t*>(out_row + first) : (uint32_t)(uint8_t)out_row[first];
  dematch_positions<VEC, EXT>(p, p.in + (size_t)blockIdx.z * p.in_stride_outer + (size_t)blockIdx.y * p.in_stride, out_row,
                              first, old, 1.0f / (float)p.cols);
}

// One workgroup per codeblock: the input is read once, coalesced, into LDS; the strided reads of the deinterleaver
// then stay on chip.
constexpr uint32_t DEMATCH_LDS_THREADS = 1024; // 512: the same; 256: 7 % slower on the config-5 chain (A/B on one box)

// One workgroup per codeblock, everything on chip: the input and the soft buffer are read once, coalesced, into LDS;
// the operations run one after the other over their own ranges (no per-position range tests, the strided reads of the
// deinterleaver stay in LDS); the soft buffer is written back once.  Soft bits no operation covers are written back
// unchanged.
template <bool EXT>
__global__ __launch_bounds__(DEMATCH_LDS_THREADS) void ldpc_dematch_lds_kernel(DematchLaunch p)
{
  extern __shared__ __attribute__((aligned(16))) int8_t dematch_lds[];
  const uint32_t e      = p.cols * p.qm;
  int8_t*        staged = dematch_lds;                       // [e]
  int8_t*        soft   = dematch_lds + ((e + 15u) & ~15u);  // [block_length]
  const int8_t*  in      = p.in + (size_t)blockIdx.y * p.in_stride_outer + (size_t)blockIdx.x * p.in_stride;
  int8_t*        out_row = p.out + (size_t)blockIdx.y * p.out_stride_outer + (size_t)blockIdx.x * p.out_stride;
  const uint32_t T = blockDim.x, tid = threadIdx.x;
  if (!p.skip_load) { // (a first transmission overwrites the whole block: 25 KB per codeblock less to read)
    const uint32_t* src = reinterpret_cast<const uint32_t*>(out_row); // dword aligned (checked by the launcher)
    uint32_t*       dst = reinterpret_cast<uint32_t*>(soft);
    for (uint32_t i = tid; i < p.block_length / 4u; i += T) {
      dst[i] = src[i];
    }
  }
  if ((reinterpret_cast<uintptr_t>(in) & 3u) == 0) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(in);
    uint32_t*       dst = reinterpret_cast<uint32_t*>(staged);
    for (uint32_t i = tid; i < e / 4u; i += T) {
      dst[i] = src[i];
    }
    for (uint32_t i = (e & ~3u) + tid; i < e; i += T) {
      staged[i] = in[i];
    }
  } else {
    for (uint32_t i = tid; i < e; i += T) {
      staged[i] = in[i];
    }
  }
  __syncthreads();
  const float rcp_cols = 1.0f / (float)p.cols;
  for (uint32_t k = 0; k != p.n_ops; ++k) { // workgroup-uniform
    const DematchOp op  = EXT ? p.ops_ext[k] : p.ops[k];
    int8_t*         dst = soft + op.begin;
    switch (op.kind) {
      case DEMATCH_ZERO:
      case DEMATCH_FILL: {
        const int8_t   value = op.kind == DEMATCH_ZERO ? 0 : 127; // LLR_INFINITY: a filler bit is a certain zero
        const uint32_t head  = min(op.count, (4u - (op.begin & 3u)) & 3u); // bytes up to the first aligned dword
        const uint32_t words = (op.count - head) >> 2;
        if (tid < head) {
          dst[tid] = value;
        }
        uint32_t* dst32 = reinterpret_cast<uint32_t*>(dst + head);
        for (uint32_t q = tid; q < words; q += T) {
          dst32[q] = 0x01010101u * (uint32_t)(uint8_t)value;
        }
        for (uint32_t q = head + 4u * words + tid; q < op.count; q += T) {
          dst[q] = value;
        }
        break;
      }
      case DEMATCH_COPY:
        for (uint32_t q = tid; q < op.count; q += T) {
          dst[q] = (int8_t)dematch_fetch(staged, op.src + q, p.qm, p.cols, rcp_cols);
        }
        break;
      default:
        for (uint32_t q = tid; q < op.count; q += T) {
          dst[q] = (int8_t)dematch_sum(dematch_fetch(staged, op.src + q, p.qm, p.cols, rcp_cols), dst[q]);
        }
        break;
    }
    lds_barrier();
  }
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(soft);
    uint32_t*       dst = reinterpret_cast<uint32_t*>(out_row);
    for (uint32_t i = tid; i < p.block_length / 4u; i += T) {
      dst[i] = src[i];
    }
  }
}

// Direct form for operation lists whose destination ranges do not overlap (p.disjoint: no repetition that wraps onto itself -- every
// first transmission and every retransmission of a codeblock shorter than its buffer): nothing is staged.  A thread takes FOUR
// neighbouring received symbols -- the Qm soft bits of a symbol are row 0 .. Qm-1 of one column of the deinterleaver table,
// contiguous in the input -- transposes their bytes into one word per row (v_perm) and stores each word where the operation
// covering those four elements puts it: the lanes of a wavefront write 256 consecutive soft bits of every row with one store
// (byte-aligned dwords; one symbol per thread and a byte per store was a 64-byte store per wavefront and row).  Elements at the
// edge of an operation's range, or of the input, go one by one.  Clearing and filling run as dword stores.  Soft bits no
// operation covers are never touched, and the buffer is read only where an operation combines.
typedef uint32_t dematch_u32_bytewise __attribute__((aligned(1)));

// Byte b of the little-endian byte stream held in dw (compile-time index after unrolling).
template <uint32_t B>
__device__ __forceinline__ uint32_t dematch_byte(const uint32_t (&dw)[8])
{
  return (dw[B >> 2] >> (8u * (B & 3u))) & 0xFFu;
}
template <uint32_t QM, uint32_t J>
__device__ __forceinline__ void dematch_rows_from(const uint32_t (&dw)[8], uint32_t (&w)[8])
{
  if constexpr (J < QM) {
    w[J] = dematch_byte<J>(dw) | (dematch_byte<QM + J>(dw) << 8) | (dematch_byte<2 * QM + J>(dw) << 16) | (dematch_byte<3 * QM + J>(dw) << 24);
    dematch_rows_from<QM, J + 1>(dw, w);
  }
}
template <uint32_t QM>
__device__ __forceinline__ void dematch_rows(const uint32_t (&dw)[8], uint32_t (&w)[8])
{
  dematch_rows_from<QM, 0>(dw, w);
}

template <bool EXT>
__global__ __launch_bounds__(256) void ldpc_dematch_scatter_kernel(DematchLaunch p)
{
  const int8_t*  in      = p.in + (size_t)blockIdx.z * p.in_stride_outer + (size_t)blockIdx.y * p.in_stride;
  int8_t*        out_row = p.out + (size_t)blockIdx.z * p.out_stride_outer + (size_t)blockIdx.y * p.out_stride;
  const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, gsize = gridDim.x * blockDim.x;
  const uint32_t qm = p.qm, cols = p.cols;
  // the thread's symbols i0 .. i0 + nsym - 1: their 4 * Qm input bytes are contiguous (Qm dwords when all four exist)
  const uint32_t i0   = 4u * gtid;
  const uint32_t nsym = i0 < cols ? (cols - i0 < 4u ? cols - i0 : 4u) : 0u;
  uint32_t       dw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  {
    const int8_t* src = in + (size_t)i0 * qm;
    if (nsym == 4u && qm == 8 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
      const uint4 a = reinterpret_cast<const uint4*>(src)[0], b = reinterpret_cast<const uint4*>(src)[1];
      dw[0] = a.x, dw[1] = a.y, dw[2] = a.z, dw[3] = a.w, dw[4] = b.x, dw[5] = b.y, dw[6] = b.z, dw[7] = b.w;
    } else if (nsym == 4u && (reinterpret_cast<uintptr_t>(src) & 3u) == 0) {
#pragma unroll
      for (uint32_t k = 0; k != 8; ++k) {
        if (k < qm) {
          dw[k] = reinterpret_cast<const uint32_t*>(src)[k];
        }
      }
    } else {
#pragma unroll
      for (uint32_t b = 0; b != 32; ++b) { // (constant register indices: the last threads of a row, or an unaligned input)
        if (b < nsym * qm) {
          dw[b >> 2] |= (uint32_t)(uint8_t)src[b] << (8u * (b & 3u));
        }
      }
    }
  }
  // row words: w[j] = row j of the four symbols, first symbol in the low byte: byte k * Qm + j of the input is symbol k, row j
  uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  switch (qm) { // uniform
    case 8:
      dematch_rows<8>(dw, w);
      break;
    case 6:
      dematch_rows<6>(dw, w);
      break;
    case 4:
      dematch_rows<4>(dw, w);
      break;
    case 2:
      dematch_rows<2>(dw, w);
      break;
    default:
      dematch_rows<1>(dw, w);
      break;
  }
  for (uint32_t k = 0; k != p.n_ops; ++k) { // uniform
    const DematchOp op  = EXT ? p.ops_ext[k] : p.ops[k];
    int8_t*         dst = out_row + op.begin;
    if (op.kind == DEMATCH_ZERO || op.kind == DEMATCH_FILL) {
      const int8_t   value = op.kind == DEMATCH_ZERO ? 0 : 127; // LLR_INFINITY: a filler bit is a certain zero
      const uint32_t head  = min(op.count, (4u - (op.begin & 3u)) & 3u); // bytes up to the first aligned dword
      const uint32_t words = (op.count - head) >> 2;
      if (gtid < head) {
        dst[gtid] = value;
      }
      uint32_t* dst32 = reinterpret_cast<uint32_t*>(dst + head);
      for (uint32_t q = gtid; q < words; q += gsize) {
        dst32[q] = 0x01010101u * (uint32_t)(uint8_t)value;
      }
      const uint32_t q = head + 4u * words + gtid;
      if (q < op.count) {
        dst[q] = value;
      }
      continue;
    }
    if (nsym == 0u) {
      continue;
    }
    // elements j * cols + i0 .. + nsym - 1 of the deinterleaved input, for the rows the operation's source range reaches
#pragma unroll
    for (uint32_t j = 0; j != 8; ++j) {
      if (j >= qm) { // uniform
        break;
      }
      const uint32_t rel = j * cols + i0 - op.src; // (unsigned: huge when the first element lies before the range)
      if (nsym == 4u && rel < op.count && op.count - rel >= 4u) {
        int8_t* d = dst + rel;
        if (op.kind == DEMATCH_COPY) {
          *reinterpret_cast<dematch_u32_bytewise*>(d) = w[j];
        } else {
          const uint32_t old = *reinterpret_cast<const dematch_u32_bytewise*>(d);
          uint32_t       r   = 0;
#pragma unroll
          for (uint32_t e = 0; e != 4; ++e) {
            const int v = (int)(int8_t)((w[j] >> (8u * e)) & 0xFFu), o = (int)(int8_t)((old >> (8u * e)) & 0xFFu);
            r |= ((uint32_t)dematch_sum(v, o) & 0xFFu) << (8u * e);
          }
          *reinterpret_cast<dematch_u32_bytewise*>(d) = r;
        }
      } else {
#pragma unroll
        for (uint32_t e = 0; e != 4; ++e) {
          if (e < nsym && rel + e < op.count) { // (rel + e wraps back into range exactly for the elements at or behind op.src)
            const int v = (int)(int8_t)((w[j] >> (8u * e)) & 0xFFu);
            int8_t*   d = dst + (rel + e);
            *d          = (int8_t)(op.kind == DEMATCH_COPY ? v : dematch_sum(v, (int)*d));
          }
        }
      }
    }
  }
}

template <bool EXT>
static void launch_dematch_variant(const DematchLaunch& p, uint32_t n_cb, uint32_t n_outer, hipStream_t stream)
{
  // four soft bits per thread when every codeblock row is dword aligned (block lengths are multiples of 4 only for even Zc)
  const bool     vec4 = ((reinterpret_cast<uintptr_t>(p.out) | p.out_stride | p.out_stride_outer | p.block_length) & 3u) == 0;
  const uint32_t e    = p.cols * p.qm;
  const uint32_t lds  = ((e + 15u) & ~15u) + ((p.block_length + 15u) & ~15u);
  if (vec4 && p.disjoint) {
    hipLaunchKernelGGL(ldpc_dematch_scatter_kernel<EXT>, dim3(((p.cols + 3u) / 4u + 255u) / 256u, n_cb, n_outer), dim3(256), 0, stream, p);
  } else if (vec4 && lds <= 64u * 1024u) {
    hipLaunchKernelGGL(ldpc_dematch_lds_kernel<EXT>, dim3(n_cb, n_outer), dim3(DEMATCH_LDS_THREADS), lds, stream, p);
  } else if (vec4) {
    const uint32_t blocks = (p.block_length / 4 + 255) / 256;
    hipLaunchKernelGGL((ldpc_dematch_kernel<4, EXT>), dim3(blocks, n_cb, n_outer), dim3(256), 0, stream, p);
  } else {
    const uint32_t blocks = (p.block_length + 255) / 256;
    hipLaunchKernelGGL((ldpc_dematch_kernel<1, EXT>), dim3(blocks, n_cb, n_outer), dim3(256), 0, stream, p);
  }
}

hipError_t launch_ldpc_dematch(const DematchLaunch& p, uint32_t n_cb, hipStream_t stream, uint32_t n_outer)
{
  if (n_cb == 0 || n_outer == 0 || p.n_ops == 0) {
    return hipSuccess;
  }
  if (p.ops_ext != nullptr) {
    launch_dematch_variant<true>(p, n_cb, n_outer, stream);
  } else {
    launch_dematch_variant<false>(p, n_cb, n_outer, stream);
  }
  return hipGetLastError();
}

} // namespace nrphy
