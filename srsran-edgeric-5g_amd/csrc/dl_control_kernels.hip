// Downlink control channels for gfx950 (MI355X) (the other downlink grid writers, so that a
// whole slot's resource grid is produced in HBM).
//
//   pdcch_kernel   one wavefront per DCI: CRC24C with the RNTI mask, polar interleaving + sub-channel allocation (one
//                  host-built gather table per (K, E)), polar transform on packed words (five in-word stages, the rest
//                  across lanes), sub-block interleaving + bit selection, Gold scrambling, QPSK, precoding, RE mapping,
//                  and the PDCCH DM-RS.  Replaces pdcch_processor_impl::process
//                  (R/lib/phy/upper/channel_processors/pdcch_processor_impl.cpp:66-118) with pdcch_encoder_impl,
//                  pdcch_modulator_impl, dmrs_pdcch_processor_impl and the polar blocks behind them
//                  (R/lib/phy/upper/channel_coding/polar/).
//   ssb_kernel     one wavefront per SS/PBCH block: PBCH payload scrambling, CRC24C, polar coding (K = 56, E = 864),
//                  scrambling, QPSK, PBCH DM-RS, PSS and SSS, written on every port of the block.  Replaces
//                  ssb_processor_impl::process (R/lib/phy/upper/channel_processors/ssb_processor_impl.cpp:29-107) with
//                  pbch_encoder_impl, pbch_modulator_impl, dmrs_pbch_processor_impl, pss_processor_impl and
//                  sss_processor_impl.
//
// These channels are a few hundred resource elements per slot: the kernels are latency-, not bandwidth-bound, and
// exist so that the grid never has to visit the host.  A batch of slots gives one wave per DCI / block.
#include "bits_device.h"

namespace nrphy {

// TS 38.212 Table 5.4.1.1-1: sub-block interleaver pattern P(i).
__constant__ uint8_t SUBBLOCK_P[32] = {0,  1,  2,  4,  3,  5,  6,  7,  8,  16, 9,  17, 10, 18, 11, 19,
                                       12, 20, 13, 21, 14, 22, 15, 23, 24, 25, 26, 28, 27, 29, 30, 31};

// J(n) = P(floor(32 n / N)) * (N / 32) + n mod (N / 32), N = 2^log_n >= 32.
__device__ __forceinline__ uint32_t subblock_j(uint32_t n, uint32_t log_n)
{
  const uint32_t q = log_n - 5u; // log2(N / 32)
  return ((uint32_t)SUBBLOCK_P[n >> q] << q) + (n & ((1u << q) - 1u));
}

__device__ __forceinline__ uint32_t bit_of(const uint32_t* words, uint32_t i) // MSB-first bit i
{
  return (words[i >> 5] >> (31u - (i & 31u))) & 1u;
}

// d = u G_N on packed words, word w of the N-bit block in lane w (N <= 512: 16 lanes; the other lanes hold zero).
// Stage s: u[i] ^= u[i + s] for every i with bit s clear (polar_encoder_impl.cpp:33-52) -- in-word for s < 32 (bit i
// sits at position 31 - i: the partner is s positions to the right), across lanes above.
__device__ __forceinline__ uint32_t polar_transform(uint32_t x, uint32_t nwords, uint32_t lane)
{
  x ^= (x << 1) & 0xAAAAAAAAu;
  x ^= (x << 2) & 0xCCCCCCCCu;
  x ^= (x << 4) & 0xF0F0F0F0u;
  x ^= (x << 8) & 0xFF00FF00u;
  x ^= (x << 16) & 0xFFFF0000u;
  for (uint32_t s = 1; s < nwords; s <<= 1) { // wave-uniform trip count
    const uint32_t other = __shfl_down(x, s, WAVE);
    if (!(lane & s)) {
      x ^= other;
    }
  }
  return x;
}

typedef __bf16 dl_bf16x2 __attribute__((ext_vector_type(2)));
typedef float  dl_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t dl_cbf16(float re, float im) // round to nearest even, as to_bf16 of the reference
{
  const dl_f32x2  v = {re, im};
  const dl_bf16x2 b = __builtin_convertvector(v, dl_bf16x2);
  return *reinterpret_cast<const uint32_t*>(&b);
}

// x * w as the reference's precoder evaluates it (channel_precoder_avx2.cpp:51-56), to the grid on every port.
__device__ __forceinline__ void put_precoded(uint32_t* row /* port 0 */, size_t port_stride, uint32_t subc, float xr, float xi,
                                             const float* w, uint32_t nof_ports)
{
  for (uint32_t port = 0; port != nof_ports; ++port) {
    const float wr = w[2u * port], wi = w[2u * port + 1u];
    const float t0 = __fmul_rn(xi, wi), t1 = __fmul_rn(xr, wi);
    row[port * port_stride + subc] = dl_cbf16(__fmaf_rn(xr, wr, -t0), __fmaf_rn(xi, wr, t1));
  }
}

constexpr uint32_t DL_SEQ_WORDS = 224; // the longest sequence a wave holds: PBCH scrambling, 7 * 864 + 864 bits

__global__ __launch_bounds__(WAVE) void pdcch_kernel(DlControlLaunch p)
{
  __shared__ uint32_t s_seq[DL_SEQ_WORDS];
  __shared__ uint32_t s_scratch[DL_SEQ_WORDS];
  __shared__ uint32_t s_d[16];
  __shared__ uint8_t  s_payload[NRPHY_PDCCH_MAX_PAYLOAD];
  const uint32_t lane = threadIdx.x;
  const auto*    wk   = to_constant(&p.pdcch[blockIdx.x]);
  const uint32_t A = wk->A, N = wk->N, E = wk->E, nwords = N >> 5, log_n = 31u - (uint32_t)__clz(N);

  // 1. CRC attachment (TS 38.212 Section 7.3.2): CRC24C over 24 ones + payload = the constant share of the ones plus
  //    one host-computed weight x^(bits behind) mod g per set payload bit; the last 16 parity bits masked with the RNTI.
  uint32_t part = 0;
  for (uint32_t i = lane; i < A; i += WAVE) {
    const uint32_t b = p.bytes[wk->payload_offset + i] & 1u;
    s_payload[i]     = (uint8_t)b;
    part ^= b ? p.words[wk->crcw_offset + i] : 0u;
  }
  const uint32_t crc = (wave_xor(part) ^ wk->crc_const) ^ (wk->rnti & 0xFFFFu);
  wave_sync();
  // 2. Interleaving + sub-channel allocation: polar input bit i comes from message bit src[i] (0xFFFF: frozen).
  uint32_t x = 0;
  if (lane < nwords) {
    const uint16_t* src = p.tab16 + wk->src_offset + 32u * lane;
    for (uint32_t t = 0; t != 32; ++t) {
      const uint32_t s = src[t];
      uint32_t       b = 0;
      if (s != 0xFFFFu) {
        b = s < A ? (uint32_t)s_payload[s] : (crc >> (23u - (s - A))) & 1u;
      }
      x |= b << (31u - t);
    }
  }
  // 3. Polar transform.
  x = polar_transform(x, nwords, lane);
  if (lane < 16) {
    s_d[lane] = x;
  }
  // Scrambling sequence of the E rate-matched bits (TS 38.211 Section 7.3.2.3).
  gold_sequence_wave(p.gold, p.x1_words, wk->c_init_data, (E + 31u) >> 5, s_seq, s_scratch, lane);
  wave_sync();

  // 4. Sub-block interleaving + bit selection (TS 38.212 Sections 5.4.1.1, 5.4.1.2), scrambling, QPSK, precoding, mapping:
  //    symbol m of the candidate is RE m of (OFDM symbol, PRB ascending, subcarriers 0 2 3 4 6 7 8 10 11).
  const uint32_t  n_sym = E >> 1, per_symbol = 9u * wk->n_prb;
  const uint16_t* prbs  = p.tab16 + wk->prb_offset;
  const size_t    port_stride = (size_t)NRPHY_NSYMB * p.grid_nof_subc;
  uint32_t*       grid0 = p.grid ? p.grid + (size_t)wk->grid_index * p.grid_nof_ports * port_stride : nullptr;
  const float*    w     = p.weights + wk->weights_offset;
  for (uint32_t m = lane; m < n_sym; m += WAVE) {
    uint32_t bits[2];
#pragma unroll
    for (uint32_t b = 0; b != 2; ++b) {
      const uint32_t k   = 2u * m + b;
      uint32_t       idx = k;
      if (wk->mode == 0) {
        idx = k & (N - 1u); // repetition
      } else if (wk->mode == 1) {
        idx = k + (N - E); // puncturing: the first N - E bits are not sent
      }
      bits[b] = bit_of(s_d, subblock_j(idx, log_n));
      if (p.enc != nullptr) {
        p.enc[wk->enc_offset + k] = (uint8_t)bits[b];
      }
      bits[b] ^= bit_of(s_seq, k);
    }
    if (grid0 == nullptr) {
      continue;
    }
    const uint32_t l = wk->start_symbol + m / per_symbol, r = m % per_symbol;
    const uint32_t i_prb = r / 9u, i_re = r - 9u * i_prb;
    const uint32_t subc  = 12u * prbs[i_prb] + i_re + (i_re + 2u) / 3u; // 0 2 3 4 6 7 8 10 11
    const uint32_t prg   = subc / wk->prg_size_subc;
    put_precoded(grid0 + (size_t)l * p.grid_nof_subc, port_stride, subc, bits[0] ? -wk->data_amp : wk->data_amp,
                 bits[1] ? -wk->data_amp : wk->data_amp, w + 2u * prg * wk->nof_ports, wk->nof_ports);
  }
  if (grid0 == nullptr) {
    return;
  }
  // 5. DM-RS (TS 38.211 Section 7.4.1.3): pilot n = 3 (prb - reference) + k' on subcarrier 4 k' + 1 of the PRB.
  const uint32_t dmrs_words = (6u * (wk->top_prb - wk->ref_rb) + 31u) >> 5;
  for (uint32_t s = 0; s != wk->duration; ++s) { // wave-uniform
    wave_sync();
    gold_sequence_wave(p.gold, p.x1_words, wk->dmrs_c_init[s], dmrs_words, s_seq, s_scratch, lane);
    wave_sync();
    for (uint32_t i = lane; i < 3u * wk->n_prb; i += WAVE) {
      const uint32_t i_prb = i / 3u, kp = i - 3u * i_prb, prb = prbs[i_prb];
      const uint32_t n     = 3u * (prb - wk->ref_rb) + kp;
      const uint32_t subc  = 12u * prb + 4u * kp + 1u;
      const uint32_t prg   = subc / wk->prg_size_subc;
      put_precoded(grid0 + (size_t)(wk->start_symbol + s) * p.grid_nof_subc, port_stride, subc,
                   bit_of(s_seq, 2u * n) ? -wk->dmrs_amp : wk->dmrs_amp, bit_of(s_seq, 2u * n + 1u) ? -wk->dmrs_amp : wk->dmrs_amp,
                   w + 2u * prg * wk->nof_ports, wk->nof_ports);
    }
  }
}

// ---- SS/PBCH block ---------------------------------------------------------------------------------------------------
// The three length-127 m-sequences of TS 38.211 Sections 7.4.2.2 / 7.4.2.3 as bit masks (bit i of word i / 32, LSB
// first), built at compile time: x(i + 7) = (x(i + 4) + x(i)) mod 2 for PSS and the first SSS sequence,
// x(i + 7) = (x(i + 1) + x(i)) mod 2 for the second.
struct MSequence {
  uint32_t w[4];
};
constexpr MSequence m_sequence(uint32_t init /* x(6) .. x(0), x(0) in bit 0 */, uint32_t tap)
{
  MSequence s = {{0, 0, 0, 0}};
  uint32_t  x[134] = {};
  for (uint32_t i = 0; i != 7; ++i) {
    x[i] = (init >> i) & 1u;
  }
  for (uint32_t i = 0; i != 127; ++i) {
    x[i + 7] = (x[i + tap] + x[i]) & 1u;
  }
  for (uint32_t i = 0; i != 127; ++i) {
    s.w[i >> 5] |= x[i] << (i & 31u);
  }
  return s;
}
__constant__ MSequence M_PSS  = m_sequence(0x76u, 4); // x(6..0) = 1 1 1 0 1 1 0
__constant__ MSequence M_SSS0 = m_sequence(0x01u, 4);
__constant__ MSequence M_SSS1 = m_sequence(0x01u, 1);

__device__ __forceinline__ uint32_t mseq_bit(const MSequence& s, uint32_t i)
{
  return (s.w[i >> 5] >> (i & 31u)) & 1u;
}

__device__ __forceinline__ void put_all_ports(uint32_t* grid0, size_t port_stride, const SsbWork NRPHY_CONSTANT* wk, uint32_t l,
                                              uint32_t subc, uint32_t value, uint32_t nof_subc)
{
  for (uint32_t i = 0; i != wk->nof_ports; ++i) {
    grid0[(size_t)wk->ports[i] * port_stride + (size_t)l * nof_subc + subc] = value;
  }
}

__global__ __launch_bounds__(WAVE) void ssb_kernel(DlControlLaunch p)
{
  __shared__ uint32_t s_seq[DL_SEQ_WORDS];
  __shared__ uint32_t s_scratch[DL_SEQ_WORDS];
  __shared__ uint32_t s_d[16];
  __shared__ uint32_t s_dmrs[12];
  const uint32_t lane = threadIdx.x;
  const auto*    wk   = to_constant(&p.ssb[blockIdx.x]);

  // 1. PBCH payload scrambling (TS 38.212 Section 7.1.2): the j-th bit of c(M v + ...) goes to the j-th position that
  //    is scrambled; then CRC24C over the 32 bits (weights from the host).
  gold_sequence_wave(p.gold, p.x1_words, wk->pci, 4, s_seq, s_scratch, lane);
  wave_sync();
  uint32_t bit = 0, part = 0;
  if (lane < 32) {
    bit = (wk->a_bits >> (31u - lane)) & 1u;
    if ((wk->scr_mask >> (31u - lane)) & 1u) {
      const uint32_t rank = lane == 0 ? 0u : (uint32_t)__popc(wk->scr_mask >> (32u - lane));
      bit ^= bit_of(s_seq, wk->scr_adv + rank);
    }
    part = bit ? p.words[wk->crcw_offset + lane] : 0u;
  }
  const uint32_t crc     = wave_xor(part);
  const uint32_t a_prime = (uint32_t)__ballot(bit != 0); // lane i = payload bit i
  // 2. Interleaving + allocation + polar transform (N = 512: 16 words) -- message bit s < 32 from a', else CRC bit s - 32.
  uint32_t x = 0;
  if (lane < 16) {
    const uint16_t* src = p.tab16 + wk->src_offset + 32u * lane;
    for (uint32_t t = 0; t != 32; ++t) {
      const uint32_t s = src[t];
      uint32_t       b = 0;
      if (s != 0xFFFFu) {
        b = s < 32u ? (a_prime >> s) & 1u : (crc >> (23u - (s - 32u))) & 1u;
      }
      x |= b << (31u - t);
    }
  }
  x = polar_transform(x, 16, lane);
  if (lane < 16) {
    s_d[lane] = x;
  }
  wave_sync();
  // 3. The PBCH scrambling sequence (TS 38.211 Section 7.3.3.1), 864 bits from offset ssb_adv, and the DM-RS sequence.
  gold_sequence_wave(p.gold, p.x1_words, wk->pci, (wk->ssb_adv + 864u + 31u) >> 5, s_seq, s_scratch, lane);
  wave_sync();
  if (p.enc != nullptr) { // pbch_encoder::encode alone
    for (uint32_t k = lane; k < 864u; k += WAVE) {
      p.enc[wk->enc_offset + k] = (uint8_t)bit_of(s_d, subblock_j(k & 511u, 9));
    }
  }
  if (p.grid == nullptr) {
    return;
  }
  const size_t port_stride = (size_t)NRPHY_NSYMB * p.grid_nof_subc;
  uint32_t*    grid0       = p.grid + (size_t)wk->grid_index * p.grid_nof_ports * port_stride;
  const float  q           = 0.70710678118654752440f; // (float)M_SQRT1_2
  const uint32_t v         = wk->pci & 3u;
  // PBCH: 432 symbols over (l0 + 1: 180, l0 + 2: 36 + 36, l0 + 3: 180) data subcarriers, three of four per group of four.
  for (uint32_t m = lane; m < 432u; m += WAVE) {
    uint32_t s, j; // OFDM symbol of the block, data index within it
    if (m < 180u) {
      s = 1, j = m;
    } else if (m < 252u) {
      s = 2, j = m - 180u;
    } else {
      s = 3, j = m - 252u;
    }
    // the j-th subcarrier of the symbol that is not a DM-RS position: group of four g, position t in {0..3} \ {v}
    uint32_t g = j / 3u, t = j - 3u * g;
    t += t >= v ? 1u : 0u;
    uint32_t k = 4u * g + t;
    if (s == 2 && j >= 36u) {
      k += 144u; // the upper 48 subcarriers start at 192
    }
    const uint32_t b0 = bit_of(s_d, subblock_j((2u * m) & 511u, 9)) ^ bit_of(s_seq, wk->ssb_adv + 2u * m);
    const uint32_t b1 = bit_of(s_d, subblock_j((2u * m + 1u) & 511u, 9)) ^ bit_of(s_seq, wk->ssb_adv + 2u * m + 1u);
    put_all_ports(grid0, port_stride, wk, wk->l0 + s, wk->k0 + k, dl_cbf16(b0 ? -q : q, b1 ? -q : q), p.grid_nof_subc);
  }
  wave_sync();
  // DM-RS for PBCH (TS 38.211 Section 7.4.1.4): 144 pilots on subcarriers v, v + 4, ... of the same three symbols.
  gold_sequence_wave(p.gold, p.x1_words, wk->dmrs_c_init, 9, s_dmrs, s_scratch, lane);
  wave_sync();
  for (uint32_t d = lane; d < 144u; d += WAVE) {
    uint32_t s, j;
    if (d < 60u) {
      s = 1, j = d;
    } else if (d < 84u) {
      s = 2, j = d - 60u;
    } else {
      s = 3, j = d - 84u;
    }
    uint32_t k = 4u * j + v;
    if (s == 2 && j >= 12u) {
      k += 144u;
    }
    put_all_ports(grid0, port_stride, wk, wk->l0 + s, wk->k0 + k,
                  dl_cbf16(bit_of(s_dmrs, 2u * d) ? -q : q, bit_of(s_dmrs, 2u * d + 1u) ? -q : q), p.grid_nof_subc);
  }
  // PSS on symbol l0, SSS on l0 + 2, subcarriers 56 .. 182 (TS 38.211 Sections 7.4.2.2, 7.4.2.3).
  for (uint32_t n = lane; n < 127u; n += WAVE) {
    const float pss = mseq_bit(M_PSS, (n + wk->m_pss) % 127u) ? -wk->pss_amp : wk->pss_amp;
    const bool  n0 = mseq_bit(M_SSS0, (n + wk->m0) % 127u) != 0, n1 = mseq_bit(M_SSS1, (n + wk->m1) % 127u) != 0;
    // The reference multiplies the two SSS sequences as complex numbers: the imaginary part is -0 when both are -1.
    const float sss = (n0 != n1) ? -1.0f : 1.0f, sss_im = (n0 && n1) ? -0.0f : 0.0f;
    put_all_ports(grid0, port_stride, wk, wk->l0, wk->k0 + 56u + n, dl_cbf16(pss, 0.0f), p.grid_nof_subc);
    put_all_ports(grid0, port_stride, wk, wk->l0 + 2u, wk->k0 + 56u + n, dl_cbf16(sss, sss_im), p.grid_nof_subc);
  }
}

hipError_t launch_pdcch(const DlControlLaunch& p, uint32_t n, hipStream_t stream)
{
  if (n == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(pdcch_kernel, dim3(n), dim3(WAVE), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_ssb(const DlControlLaunch& p, uint32_t n, hipStream_t stream)
{
  if (n == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(ssb_kernel, dim3(n), dim3(WAVE), 0, stream, p);
  return hipGetLastError();
}

} // namespace nrphy
