// Transport-block side of the PUSCH decoder for gfx950 (MI355X) ("next" row).
//
// Replaces the tail of pusch_decoder_impl (R/lib/phy/upper/channel_processors/pusch/pusch_decoder_impl.cpp:386-497):
// once every codeblock of a transport block has passed its CRC, concatenate_codeblocks() copies the data bits of the
// codeblocks into the transport block and the CRC24A of the block is compared with the checksum the last codeblock
// carries; if it fails every codeblock flag is cleared again (one of them was a false positive).  A single-codeblock
// transport block is its codeblock.  Statistics as pusch_decoder_result: codeblocks decoded, iterations (a failed
// decode counts with the maximum).
//
// Two kernels.  pusch_concat_kernel: one thread per four transport-block bytes, which it gathers from the decoded
// messages (neighbouring threads read and write neighbouring bytes; only blocks whose codeblocks all passed are
// written).  pusch_tb_crc_kernel: one workgroup per transport block; the CRC24A of the assembled block by 16 KiB regions
// with coalesced 16-byte loads and table products (tbcrc_regions_workgroup, shared with the transmit side's prologue) --
// or, for a block that is not word-aligned in the caller's buffer, byte-wise: every thread runs the CRC over contiguous runs
// with a byte table in LDS and shifts each remainder to the end of the block with a host-computed weight (CRC is linear);
// the workgroup compares with the checksum, writes the result record and clears the codeblock flags when the comparison fails.
#include "bits_device.h"

namespace nrphy {

constexpr uint32_t ASSEMBLE_THREADS = PUSCH_ASSEMBLE_THREADS;

// Eight message bits starting at bit s (MSB-first packed bytes; reads one byte beyond the one holding bit s).
__device__ __forceinline__ uint32_t message_bits8(const uint8_t* m, uint32_t s)
{
  const uint32_t w = ((uint32_t)m[s >> 3] << 8) | m[(s >> 3) + 1u];
  return (w >> (8u - (s & 7u))) & 0xFFu;
}

// Byte i of the concatenated data bits: codeblock r holds bits [r * info, (r + 1) * info) of the stream.
__device__ __forceinline__ uint32_t stream_byte(const uint8_t* msgs, uint32_t msg_stride, uint32_t info, uint32_t i)
{
  const uint32_t pos = 8u * i, r = pos / info, s = pos - r * info;
  const uint8_t* m   = msgs + (size_t)r * msg_stride;
  uint32_t       b   = message_bits8(m, s);
  if (s + 8u > info) { // the byte straddles two codeblocks
    const uint32_t n1 = info - s;
    b                 = (b & (0xFFu << (8u - n1)) & 0xFFu) | (message_bits8(m + msg_stride, 0) >> n1);
  }
  return b;
}

__global__ __launch_bounds__(256) void pusch_concat_kernel(PuschAssembleLaunch p)
{
  const uint32_t tb = blockIdx.y, C = p.C;
  const uint8_t* cb_ok = p.cb_ok + (size_t)tb * C;
  int            bad   = 0;
  for (uint32_t r = threadIdx.x; r < C; r += blockDim.x) {
    bad |= cb_ok[r] == 0;
  }
  if (__syncthreads_or(bad)) { // workgroup-uniform: a codeblock is still missing, nothing to assemble
    return;
  }
  const uint32_t first = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
  if (first >= p.tb_bytes) {
    return;
  }
  const uint8_t* msgs = p.cb_msg + (size_t)tb * C * p.msg_stride;
  uint8_t*       out  = p.tb + (size_t)tb * p.tb_stride + first;
  const uint32_t n    = min(4u, p.tb_bytes - first);
  uint32_t       b[4] = {0, 0, 0, 0};
  bool           done = false;
  if (n == 4 && C != 1) {
    // The common case -- the four bytes come from ONE codeblock's message -- as one unaligned 4-byte load plus the byte
    // behind it and a shift (message_bits8 reads that byte too), one division per thread instead of four.
    const uint32_t pos = 8u * first, r = pos / p.cb_info_bits, s = pos - r * p.cb_info_bits;
    if (s + 32u <= p.cb_info_bits) {
      typedef uint32_t word_bytewise __attribute__((aligned(1)));
      const uint8_t* m   = msgs + (size_t)r * p.msg_stride + (s >> 3);
      const uint32_t hi  = __builtin_bswap32(*reinterpret_cast<const word_bytewise*>(m));
      const uint32_t sft = s & 7u;
      const uint32_t v   = sft != 0 ? (hi << sft) | ((uint32_t)m[4] >> (8u - sft)) : hi; // 32 stream bits, first in the MSB
      b[0] = v >> 24, b[1] = (v >> 16) & 0xFFu, b[2] = (v >> 8) & 0xFFu, b[3] = v & 0xFFu;
      done = true;
    }
  }
  for (uint32_t k = 0; k != (done ? 0u : n); ++k) {
    b[k] = (C == 1) ? msgs[first + k] : stream_byte(msgs, p.msg_stride, p.cb_info_bits, first + k);
  }
  if (n == 4 && (reinterpret_cast<uintptr_t>(out) & 3u) == 0) {
    *reinterpret_cast<uint32_t*>(out) = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
  } else {
    for (uint32_t k = 0; k != n; ++k) {
      out[k] = (uint8_t)b[k];
    }
  }
}

__global__ __launch_bounds__(TB_CRC_THREADS) void pusch_tb_crc_kernel(PuschAssembleLaunch p)
{
  __shared__ __attribute__((aligned(16))) uint32_t lds[TB_CRC_LDS_WORDS]; // the regions' tables and partials; the byte-wise form: its byte table
  __shared__ uint32_t s_acc[4]; // codeblocks ok, iteration sum, iteration max, CRC remainder
  const uint32_t tb = blockIdx.x, tid = threadIdx.x, C = p.C;
  uint8_t*       cb_ok = p.cb_ok + (size_t)tb * C;
  const uint8_t* msgs  = p.cb_msg + (size_t)tb * C * p.msg_stride;
  const CrcPoly  g     = crc24a();
  if (tid < 4) {
    s_acc[tid] = 0;
  }
  __syncthreads();
  for (uint32_t r = tid; r < C; r += TB_CRC_THREADS) {
    const bool ok = cb_ok[r] != 0;
    if (ok) {
      atomicAdd(&s_acc[0], 1u);
    }
    if (p.skipped[(size_t)tb * C + r] == 0) { // the decoder ran on this codeblock in this call
      const uint32_t it = p.cb_iter[(size_t)tb * C + r];
      const uint32_t n  = (ok && it != 0) ? it : p.max_iterations;
      atomicAdd(&s_acc[1], n);
      atomicMax(&s_acc[2], n);
    }
  }
  __syncthreads();
  const uint32_t n_ok  = s_acc[0];
  bool           tb_ok = false;
  if (n_ok == C) { // workgroup-uniform
    if (C == 1) {
      tb_ok = true; // the codeblock's CRC is the transport block's
    } else {
      const uint8_t* data = p.tb + (size_t)tb * p.tb_stride;
      if (((reinterpret_cast<uintptr_t>(data) | p.tb_stride) & 3u) == 0) { // workgroup-uniform
        // By 16 KiB regions with coalesced 16-byte loads and table products (the transmit side's TB-CRC role): the whole
        // block's words lie inside the caller's buffer (a stride that is a multiple of four covers the last word).  The
        // byte-wise form below -- every thread a contiguous run, a byte load per step -- took 0.14 ms per 1024 config-3 blocks.
        const uint32_t regions = (p.tb_bytes + TB_CRC_REGION_BYTES - 1u) / TB_CRC_REGION_BYTES;
        const uint32_t share   = tbcrc_regions_workgroup(p.tbcrc, 0u, g, reinterpret_cast<const uint32_t*>(data), p.tb_bytes, 0u,
                                                         regions, p.crc_factor, lds, tid);
        if (tid == 0) {
          s_acc[3] = share;
        }
      } else {
        uint32_t* s_table = lds;
        { // CRC24A byte table: remainder of b(x) * x^24
          uint32_t r = tid << 16;
          for (int k = 0; k != 8; ++k) {
            r = (r & 0x800000u) ? ((r << 1) ^ g.poly) : (r << 1);
          }
          s_table[tid] = r & 0xFFFFFFu;
        }
        __syncthreads();
        const uint32_t piece = (p.tb_bytes + ASSEMBLE_THREADS - 1) / ASSEMBLE_THREADS;
        for (uint32_t run = tid; run < ASSEMBLE_THREADS; run += TB_CRC_THREADS) { // (the weights are per run of this length)
          const uint32_t begin = min(run * piece, p.tb_bytes), end = min(begin + piece, p.tb_bytes);
          uint32_t       crc   = 0;
#pragma unroll 8
          for (uint32_t i = begin; i < end; ++i) {
            crc = ((crc << 8) ^ s_table[((crc >> 16) ^ data[i]) & 0xFFu]) & 0xFFFFFFu;
          }
          if (crc != 0) {
            atomicXor(&s_acc[3], crc_mulmod(crc, p.crc_weight[run], g));
          }
        }
      }
      __syncthreads();
      uint32_t checksum = 0;
      for (uint32_t k = 0; k != 3; ++k) {
        checksum = (checksum << 8) | stream_byte(msgs, p.msg_stride, p.cb_info_bits, p.tb_bytes + k);
      }
      tb_ok = s_acc[3] == checksum;
      if (!tb_ok) { // a codeblock CRC was a false positive: every codeblock is decoded again next time
        for (uint32_t r = tid; r < C; r += TB_CRC_THREADS) {
          cb_ok[r] = 0;
        }
      }
    }
  }
  if (tid == 0) {
    uint32_t* res = p.result + 4u * (size_t)tb;
    res[0]        = tb_ok ? 1u : 0u;
    res[1]        = n_ok;
    res[2]        = s_acc[1];
    res[3]        = s_acc[2];
  }
}

// ================================================================================================================
// Descrambling of soft bits: the sign of every log-likelihood ratio whose scrambling bit c(n) is one flips
// (pseudo_random_generator_impl::apply_xor on log_likelihood_ratio spans, pseudo_random_generator_impl.cpp:423-523;
// revert_scrambling in pusch_demodulator_impl.cpp:38-100 after the demodulation mapper).  blockIdx.x = chunk of DESCRAMBLE_CHUNK_WORDS x 32
// soft bits, blockIdx.y = codeword.  The workgroup generates its share of c(n) straight into LDS (jump to the chunk's
// offset, then the lifted recurrence), then every lane takes 16 soft bits per step: one 16-byte load, the 16 sequence
// bits spread to byte masks, bytewise two's-complement negation (-128 stays -128 as in the reference's 16-wide path),
// one 16-byte store.  HBM-bound: one byte read and one written per soft bit.
// ================================================================================================================
constexpr uint32_t DESCRAMBLE_CHUNK_WORDS = 2048;
constexpr uint32_t DESCRAMBLE_THREADS     = 256;

// The four most significant bits of `nibble << 28` spread to byte masks: bit 31 -> byte 0 (lowest address) ... bit 28
// -> byte 3.
__device__ __forceinline__ uint32_t byte_masks_msb_first(uint32_t bits4)
{
  // bits4 holds the four bits in its low nibble, first soft bit in bit 3.
  const uint32_t spread = ((bits4 >> 3) & 1u) | (((bits4 >> 2) & 1u) << 8) | (((bits4 >> 1) & 1u) << 16) | ((bits4 & 1u) << 24);
  return spread * 0xFFu;
}

// Per byte: m ? -x : x  (m = 0xFF or 0x00 per byte).
__device__ __forceinline__ uint32_t negate_bytes(uint32_t x, uint32_t m)
{
  const uint32_t a = x ^ m, b = m & 0x01010101u;
  return ((a & 0x7F7F7F7Fu) + b) ^ (a & 0x80808080u); // b < 0x80 per byte: its top bit never takes part
}

__global__ __launch_bounds__(DESCRAMBLE_THREADS) void llr_descramble_kernel(const GoldTables* gold, const uint32_t* x1_words,
                                                                            const uint32_t* __restrict__ c_init,
                                                                            const int8_t* __restrict__ in, size_t in_stride,
                                                                            int8_t* __restrict__ out, size_t out_stride,
                                                                            uint32_t length)
{
  __shared__ uint32_t ring[GOLD_RING_WORDS];
  __shared__ uint32_t seq[DESCRAMBLE_CHUNK_WORDS];
  const uint32_t tid        = threadIdx.x;
  const uint32_t first_word = blockIdx.x * DESCRAMBLE_CHUNK_WORDS;
  const uint32_t first      = first_word * 32u;
  const uint32_t count      = min(length - first, DESCRAMBLE_CHUNK_WORDS * 32u); // soft bits of this chunk (> 0 by the grid)
  const uint32_t nwords     = (count + 31u) / 32u;
  gold_sequence_workgroup<DESCRAMBLE_THREADS>(gold, x1_words, to_constant(c_init)[blockIdx.y], first_word, nwords, seq, ring, tid);
  __syncthreads();
  const int8_t* src = in + (size_t)blockIdx.y * in_stride + first;
  int8_t*       dst = out + (size_t)blockIdx.y * out_stride + first;
  // 16-byte steps need both rows aligned; the chunk offset is a multiple of 64 KiB, so that is a property of the
  // caller's pointers and strides (workgroup-uniform).
  const bool     wide   = ((((uintptr_t)src) | ((uintptr_t)dst)) & 15u) == 0;
  const uint32_t groups = wide ? count / 16u : 0u;
  for (uint32_t g = tid; g < groups; g += DESCRAMBLE_THREADS) {
    const uint32_t bits16 = (seq[g >> 1] >> ((g & 1u) ? 0u : 16u)) & 0xFFFFu;
    const uint4    v      = reinterpret_cast<const uint4*>(src)[g];
    uint4          r;
    r.x = negate_bytes(v.x, byte_masks_msb_first(bits16 >> 12));
    r.y = negate_bytes(v.y, byte_masks_msb_first(bits16 >> 8));
    r.z = negate_bytes(v.z, byte_masks_msb_first(bits16 >> 4));
    r.w = negate_bytes(v.w, byte_masks_msb_first(bits16));
    reinterpret_cast<uint4*>(dst)[g] = r;
  }
  for (uint32_t i = groups * 16u + tid; i < count; i += DESCRAMBLE_THREADS) { // ragged end, or unaligned rows
    const bool flip = ((seq[i >> 5] >> (31u - (i & 31u))) & 1u) != 0;
    const int8_t x  = src[i];
    dst[i]          = flip ? (int8_t)(uint8_t)(0u - (uint8_t)x) : x;
  }
}

hipError_t launch_llr_descramble(const GoldTables* gold, const uint32_t* x1_words, const uint32_t* d_c_init, uint32_t n_cw,
                                 uint32_t length, const int8_t* d_in, size_t in_stride, int8_t* d_out, size_t out_stride,
                                 hipStream_t stream)
{
  if (n_cw == 0 || length == 0) {
    return hipSuccess;
  }
  const uint32_t chunks = (length + DESCRAMBLE_CHUNK_WORDS * 32u - 1u) / (DESCRAMBLE_CHUNK_WORDS * 32u);
  hipLaunchKernelGGL(llr_descramble_kernel, dim3(chunks, n_cw), dim3(DESCRAMBLE_THREADS), 0, stream, gold, x1_words, d_c_init,
                     d_in, in_stride, d_out, out_stride, length);
  return hipGetLastError();
}

hipError_t launch_pusch_assemble(const PuschAssembleLaunch& p, uint32_t n_tb, hipStream_t stream)
{
  if (n_tb == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(pusch_concat_kernel, dim3((p.tb_bytes / 4 + 256) / 256, n_tb), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(pusch_tb_crc_kernel, dim3(n_tb), dim3(TB_CRC_THREADS), 0, stream, p);
  return hipGetLastError();
}

} // namespace nrphy
