// OFDM modulator and DFT kernels for gfx950 (MI355X).
//
// One workgroup modulates the OFDM symbols of one (grid, port) one after the other.  A symbol's resource-grid row
// goes from HBM straight into the registers of the first butterfly stage (bin placement and guard zeros are index
// arithmetic); while the transform of symbol l runs in LDS (Stockham autosort, radix-16/8/4/2 register butterflies)
// the row of symbol l+1 is already in flight.  Twiddles are powers of one table value per thread and stage, built in
// registers, so the only global traffic is the algorithmic one: grid in, IQ out.  The last stage applies phase
// compensation x scale and writes the useful part plus the cyclic prefix.
// Replaces ofdm_symbol_modulator_impl::modulate (R/lib/phy/lower/modulation/ofdm_modulator_impl.cpp:56-100) and
// dft_processor_generic_impl::run (R/lib/phy/generic_functions/dft_processor_generic_impl.cpp:14-218).
#include "nrphy_internal.h"

namespace nrphy {

__device__ __forceinline__ float2 cadd(float2 a, float2 b)
{
  return make_float2(a.x + b.x, a.y + b.y);
}
__device__ __forceinline__ float2 csub(float2 a, float2 b)
{
  return make_float2(a.x - b.x, a.y - b.y);
}
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// Multiplication by sign * j (sign = +1: inverse transform, -1: direct).
template <int SIGN>
__device__ __forceinline__ float2 mulj(float2 a)
{
  return SIGN > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

// ---- register butterflies: a[k] <- sum_n a[n] * exp(SIGN * 2 pi i n k / R), natural order in and out ---------
template <int SIGN>
__device__ __forceinline__ void dft2(float2& a0, float2& a1)
{
  float2 t = a0;
  a0       = cadd(t, a1);
  a1       = csub(t, a1);
}

template <int SIGN>
__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3)
{
  float2 p0 = cadd(a0, a2), q0 = csub(a0, a2);
  float2 p1 = cadd(a1, a3), q1 = mulj<SIGN>(csub(a1, a3));
  a0        = cadd(p0, p1);
  a1        = cadd(q0, q1);
  a2        = csub(p0, p1);
  a3        = csub(q0, q1);
}

template <int SIGN, int R>
struct Butterfly;

template <int SIGN>
struct Butterfly<SIGN, 2> {
  static __device__ __forceinline__ void run(float2 (&a)[2]) { dft2<SIGN>(a[0], a[1]); }
};
template <int SIGN>
struct Butterfly<SIGN, 4> {
  static __device__ __forceinline__ void run(float2 (&a)[4]) { dft4<SIGN>(a[0], a[1], a[2], a[3]); }
};
template <int SIGN>
struct Butterfly<SIGN, 8> {
  // 8 = 2 x 4: X[k1 + 2 k2] = sum_{n2<4} W8^(n2 k1) W4^(n2 k2) [ sum_{n1<2} x[4 n1 + n2] W2^(n1 k1) ].
  static __device__ __forceinline__ void run(float2 (&a)[8])
  {
    constexpr float h = 0.70710678118654752440f;
    dft2<SIGN>(a[0], a[4]);
    dft2<SIGN>(a[1], a[5]);
    dft2<SIGN>(a[2], a[6]);
    dft2<SIGN>(a[3], a[7]);
    // k1 = 1 row: multiply by W8^n2, n2 = 1, 2, 3.
    a[5] = cmul(a[5], make_float2(h, SIGN * h));
    a[6] = mulj<SIGN>(a[6]);
    a[7] = cmul(a[7], make_float2(-h, SIGN * h));
    dft4<SIGN>(a[0], a[1], a[2], a[3]); // k1 = 0: X[0], X[2], X[4], X[6]
    dft4<SIGN>(a[4], a[5], a[6], a[7]); // k1 = 1: X[1], X[3], X[5], X[7]
    float2 x1 = a[4], x2 = a[1], x3 = a[5], x4 = a[2], x5 = a[6], x6 = a[3];
    a[1] = x1;
    a[2] = x2;
    a[3] = x3;
    a[4] = x4;
    a[5] = x5;
    a[6] = x6;
  }
};
template <int SIGN>
struct Butterfly<SIGN, 16> {
  // 16 = 4 x 4: X[k1 + 4 k2] = sum_{n2<4} W16^(n2 k1) W4^(n2 k2) [ sum_{n1<4} x[4 n1 + n2] W4^(n1 k1) ].
  static __device__ __forceinline__ void run(float2 (&a)[16])
  {
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    // Inner transforms over n1 (stride 4) for each n2; result index k1 replaces n1.
    dft4<SIGN>(a[0], a[4], a[8], a[12]);
    dft4<SIGN>(a[1], a[5], a[9], a[13]);
    dft4<SIGN>(a[2], a[6], a[10], a[14]);
    dft4<SIGN>(a[3], a[7], a[11], a[15]);
    // Twiddles W16^(n2 k1) on element a[4 k1 + n2].
    a[5]  = cmul(a[5], make_float2(c1, SIGN * s1));    // 1*1
    a[6]  = cmul(a[6], make_float2(h, SIGN * h));      // 2*1
    a[7]  = cmul(a[7], make_float2(s1, SIGN * c1));    // 3*1
    a[9]  = cmul(a[9], make_float2(h, SIGN * h));      // 1*2
    a[10] = mulj<SIGN>(a[10]);                         // 2*2
    a[11] = cmul(a[11], make_float2(-h, SIGN * h));    // 3*2
    a[13] = cmul(a[13], make_float2(s1, SIGN * c1));   // 1*3
    a[14] = cmul(a[14], make_float2(-h, SIGN * h));    // 2*3
    a[15] = cmul(a[15], make_float2(-c1, -SIGN * s1)); // 3*3 = 9 -> W16^9
    // Outer transforms over n2 for each k1; result a[4 k1 + k2] = X[k1 + 4 k2].
    dft4<SIGN>(a[0], a[1], a[2], a[3]);
    dft4<SIGN>(a[4], a[5], a[6], a[7]);
    dft4<SIGN>(a[8], a[9], a[10], a[11]);
    dft4<SIGN>(a[12], a[13], a[14], a[15]);
    // Transpose 4x4 to natural order: X[k1 + 4 k2] currently at a[4 k1 + k2].
    float2 t;
    t = a[1];  a[1] = a[4];   a[4] = t;
    t = a[2];  a[2] = a[8];   a[8] = t;
    t = a[3];  a[3] = a[12];  a[12] = t;
    t = a[6];  a[6] = a[9];   a[9] = t;
    t = a[7];  a[7] = a[13];  a[13] = t;
    t = a[11]; a[11] = a[14]; a[14] = t;
  }
};

// a[j] *= b^j for j = 1..R-1.  Powers are built from b^2, b^4, b^8 (squarings) so that every power is at most
// three multiplications deep (a few ulp), and applied at once to keep few values live.
template <int R>
__device__ __forceinline__ void apply_twiddle_powers(float2 b, float2 (&a)[R])
{
  a[1] = cmul(a[1], b);
  if (R > 2) {
    const float2 b2 = cmul(b, b);
    a[2]            = cmul(a[2], b2);
    a[3]            = cmul(a[3], cmul(b2, b));
    if (R > 4) {
      const float2 b4 = cmul(b2, b2);
      a[4]            = cmul(a[4], b4);
      a[5]            = cmul(a[5], cmul(b4, b));
      a[6]            = cmul(a[6], cmul(b4, b2));
      a[7]            = cmul(a[7], cmul(b4, cmul(b2, b)));
      if (R > 8) {
        const float2 b8 = cmul(b4, b4);
        a[8]            = cmul(a[8], b8);
        a[9]            = cmul(a[9], cmul(b8, b));
        a[10]           = cmul(a[10], cmul(b8, b2));
        a[11]           = cmul(a[11], cmul(b8, cmul(b2, b)));
        const float2 b12 = cmul(b8, b4);
        a[12]            = cmul(a[12], b12);
        a[13]            = cmul(a[13], cmul(b12, b));
        a[14]            = cmul(a[14], cmul(b12, b2));
        a[15]            = cmul(a[15], cmul(b12, cmul(b2, b)));
      }
    }
  }
}

// LDS index padding: one extra element every 16 keeps the stride-16 stores of the first stage off a single bank.
__device__ __forceinline__ uint32_t pad(uint32_t i)
{
  return i + (i >> 4);
}

// Radix plans: N = R0 * R1 * R2 (R2 = 1 when two stages suffice), T = threads per transform = N / 16.
template <int N>
struct Plan;
template <> struct Plan<4096> { static constexpr int R0 = 16, R1 = 16, R2 = 16, T = 256; };
template <> struct Plan<2048> { static constexpr int R0 = 16, R1 = 16, R2 = 8, T = 128; };
template <> struct Plan<1024> { static constexpr int R0 = 16, R1 = 16, R2 = 4, T = 64; };
template <> struct Plan<512>  { static constexpr int R0 = 16, R1 = 16, R2 = 2, T = 64; };
template <> struct Plan<256>  { static constexpr int R0 = 16, R1 = 16, R2 = 1, T = 64; };
template <> struct Plan<128>  { static constexpr int R0 = 16, R1 = 8, R2 = 1, T = 64; };

// Input index k-th element of the first-stage butterfly of thread `tid`: x[tid + k * N / R0].
template <int N>
__device__ __forceinline__ uint32_t first_stage_index(uint32_t tid, int k)
{
  return tid + k * (N / Plan<N>::R0);
}

// Twiddle bases of a thread: stage s multiplies output j of its butterfly by (w_n^p)^j with w_n^p = tw[p * S].
template <int N>
struct TwiddleBase {
  float2 b0, b1; // first and second stage (the last stage of a plan has n1 = 1: no twiddles)
};

template <int SIGN, int N>
__device__ __forceinline__ TwiddleBase<N> load_twiddle_base(const float2* __restrict__ tw, uint32_t tid)
{
  using P = Plan<N>;
  TwiddleBase<N> t;
  // Stage 0: S = 1, p = butterfly index = tid (threads beyond N/R0 butterflies are idle in that stage).
  t.b0 = tw[tid & (N - 1)];
  // Stage 1: S = R0, p = b / R0 for butterfly b = tid (+ it*T); only the first iteration's base is kept here, the
  // others are derived in the stage (see stage_lds).
  t.b1 = tw[((tid / P::R0) * P::R0) & (N - 1)];
  if (SIGN < 0) {
    t.b0.y = -t.b0.y;
    t.b1.y = -t.b1.y;
  }
  return t;
}

// First Stockham stage on registers a[k] = x[tid + k N/R0]: y[R0 p + j] = DFT(a)[j] * w^(j p), p = tid, S = 1.
template <int SIGN, int N>
__device__ __forceinline__ void stage_first(float2 (&a)[Plan<N>::R0], float2 base, float2* lds, uint32_t tid)
{
  constexpr int R  = Plan<N>::R0;
  constexpr int NB = N / R;
  if (NB >= Plan<N>::T || tid < NB) {
    Butterfly<SIGN, R>::run(a);
    apply_twiddle_powers<R>(base, a);
#pragma unroll
    for (int j = 0; j != R; ++j) {
      lds[pad(R * tid + j)] = a[j];
    }
  }
  __syncthreads();
}

// A later Stockham stage (decimation in frequency, autosort).  n = N / S is the current transform length.
//   a[k] = x[q + S (p + k n/R)],   y[q + S (R p + j)] = DFT_R(a)[j] * w_n^(j p),   p < n/R, q < S.
template <int SIGN, int N, int T, int R, int S, bool LAST, typename Store>
__device__ __forceinline__ void stage_lds(float2* lds, const float2* __restrict__ tw, float2 base, uint32_t tid,
                                          Store store)
{
  constexpr int NB    = N / R;
  constexpr int ITERS = (NB + T - 1) / T;
  constexpr int n1    = N / S / R;
  float2        a[ITERS][R];
#pragma unroll
  for (int it = 0; it != ITERS; ++it) {
    uint32_t b = tid + it * T;
    if (NB % T == 0 || b < NB) {
      uint32_t p = b / S, q = b % S;
#pragma unroll
      for (int k = 0; k != R; ++k) {
        a[it][k] = lds[pad(q + S * (p + k * n1))];
      }
    }
  }
  __syncthreads(); // every read of this stage is done before anyone overwrites
#pragma unroll
  for (int it = 0; it != ITERS; ++it) {
    uint32_t b = tid + it * T;
    if (NB % T == 0 || b < NB) {
      uint32_t p = b / S, q = b % S;
      Butterfly<SIGN, R>::run(a[it]);
      if (n1 > 1) {
        float2 bs = base;
        if (it > 0) { // p differs per iteration: fetch this iteration's base (rare plans only)
          bs = tw[(p * S) & (N - 1)];
          if (SIGN < 0) {
            bs.y = -bs.y;
          }
        }
        apply_twiddle_powers<R>(bs, a[it]);
      }
#pragma unroll
      for (int j = 0; j != R; ++j) {
        float2   v   = a[it][j];
        uint32_t idx = q + S * (R * p + j);
        if (LAST) {
          store(idx, v);
        } else {
          lds[pad(idx)] = v;
        }
      }
    }
  }
  if (!LAST) {
    __syncthreads();
  }
}

template <int SIGN, int N, typename Store>
__device__ __forceinline__ void fft_from_registers(float2 (&a)[Plan<N>::R0], const TwiddleBase<N>& tb, float2* lds,
                                                   const float2* __restrict__ tw, uint32_t tid, Store store)
{
  using P = Plan<N>;
  constexpr int T = P::T;
  auto no_store   = [](uint32_t, float2) {};
  stage_first<SIGN, N>(a, tb.b0, lds, tid);
  if constexpr (P::R2 == 1) {
    stage_lds<SIGN, N, T, P::R1, P::R0, true>(lds, tw, tb.b1, tid, store);
  } else {
    stage_lds<SIGN, N, T, P::R1, P::R0, false>(lds, tw, tb.b1, tid, no_store);
    stage_lds<SIGN, N, T, P::R2, P::R0 * P::R1, true>(lds, tw, tb.b1, tid, store);
  }
  __syncthreads(); // the LDS buffer is reused by the next transform of this workgroup
}

// ================================================================================================================
// OFDM slot modulator.  blockIdx.x = (grid * nof_ports + port) * groups + group; a workgroup modulates SPW
// consecutive symbols of its (grid, port).  With SPW > 1 the next symbol's row is fetched while the current one is
// transformed.
// ================================================================================================================
template <int N>
__device__ __forceinline__ void load_symbol_row(uint32_t (&a)[Plan<N>::R0], const uint32_t* __restrict__ row,
                                                uint32_t half, uint32_t tid)
{
  // Bin placement (ofdm_modulator_impl.cpp:83-87): lower grid half -> top bins, upper half -> bins from DC, guard
  // bins zero.  Branch-free: guard bins read element 0 and discard it.
#pragma unroll
  for (int k = 0; k != Plan<N>::R0; ++k) {
    const uint32_t i   = first_stage_index<N>(tid, k);
    const bool     lo  = i < half;
    const bool     hi  = (i >= N - half) && (i < N);
    const uint32_t idx = lo ? i + half : (hi ? i - (N - half) : 0u);
    const uint32_t v   = row[idx];
    a[k]               = (lo || hi) ? v : 0u; // cbf16: real in the low half, imaginary in the high half
  }
}

template <int N, int SPW>
__global__ __launch_bounds__(Plan<N>::T) void ofdm_kernel(OfdmLaunch p, const uint32_t* __restrict__ d_grid,
                                                          const uint32_t* __restrict__ d_slot_index,
                                                          float2* __restrict__ d_iq)
{
  __shared__ float2 lds[N + N / 16 + 16];
  const uint32_t    tid_in = threadIdx.x;
  const uint32_t    groups = (p.nsymb + SPW - 1) / SPW;
  const uint32_t    gp     = blockIdx.x / groups; // grid * nof_ports + port
  const uint32_t    l0     = (blockIdx.x % groups) * SPW;
  const uint32_t    l1     = (l0 + SPW < p.nsymb) ? l0 + SPW : p.nsymb;
  const uint32_t    g      = gp / p.nof_ports;
  const uint32_t    slot   = d_slot_index ? d_slot_index[g] : 0u;
  const uint32_t    half   = p.rg_size >> 1;
  const uint32_t*   rows   = d_grid + (size_t)gp * NRPHY_NSYMB * p.rg_size;
  float2*           iq     = d_iq + (size_t)gp * p.slot_stride;

  // NOTE: a symbol loop with a register prefetch of the next row was tried (SPW = 2..14): hipcc hoists the ~100
  // loop-invariant per-thread LDS/global addresses of the three stages out of the loop, the kernel needs 226+
  // VGPRs and one workgroup per CU stays resident; one symbol per workgroup at 127 VGPRs keeps four.
  static_assert(SPW == 1, "one OFDM symbol per workgroup");
  (void)l1;
  const uint32_t       tid = tid_in;
  const uint32_t       l   = l0;
  const TwiddleBase<N> tb  = load_twiddle_base<+1, N>(p.twiddle, tid);
  uint32_t             raw[Plan<N>::R0];
  load_symbol_row<N>(raw, rows + (size_t)l * p.rg_size, half, tid);
  float2 cur[Plan<N>::R0];
#pragma unroll
  for (int k = 0; k != Plan<N>::R0; ++k) {
    cur[k] = make_float2(__uint_as_float(raw[k] << 16), __uint_as_float(raw[k] & 0xFFFF0000u));
  }
  const uint32_t sym = slot * p.nsymb + l; // symbol index within the subframe
  const uint32_t cp  = p.cp_len[sym];
  const float2   ph  = p.phase[sym];
  float2*        out = iq + p.sym_offset[sym];
  // Phase compensation x scale, then the cyclic prefix is the tail of the symbol (ofdm_modulator_impl.cpp:92-99).
  auto store = [&](uint32_t i, float2 v) {
    float2 y    = cmul(v, ph);
    out[cp + i] = y;
    if (i >= N - cp) {
      out[i - (N - cp)] = y;
    }
  };
  fft_from_registers<+1, N>(cur, tb, lds, p.twiddle, tid, store);
}

constexpr int OFDM_SYMBOLS_PER_WG = 1;

template <int N>
static hipError_t launch_ofdm_n(const OfdmLaunch& p, uint32_t nof_grids, const uint32_t* d_grid,
                                const uint32_t* d_slot_index, float2* d_iq, hipStream_t stream)
{
  constexpr int SPW    = OFDM_SYMBOLS_PER_WG;
  uint32_t      blocks = nof_grids * p.nof_ports * ((p.nsymb + SPW - 1) / SPW);
  hipLaunchKernelGGL((ofdm_kernel<N, SPW>), dim3(blocks), dim3(Plan<N>::T), 0, stream, p, d_grid, d_slot_index, d_iq);
  return hipGetLastError();
}

hipError_t launch_ofdm(const OfdmLaunch& p, uint32_t nof_grids, const uint32_t* d_grid, const uint32_t* d_slot_index,
                       float2* d_iq, hipStream_t stream)
{
  if (nof_grids == 0) {
    return hipSuccess;
  }
  switch (p.dft_size) {
    case 4096:
      return launch_ofdm_n<4096>(p, nof_grids, d_grid, d_slot_index, d_iq, stream);
    case 2048:
      return launch_ofdm_n<2048>(p, nof_grids, d_grid, d_slot_index, d_iq, stream);
    case 1024:
      return launch_ofdm_n<1024>(p, nof_grids, d_grid, d_slot_index, d_iq, stream);
    case 512:
      return launch_ofdm_n<512>(p, nof_grids, d_grid, d_slot_index, d_iq, stream);
    case 256:
      return launch_ofdm_n<256>(p, nof_grids, d_grid, d_slot_index, d_iq, stream);
    case 128:
      return launch_ofdm_n<128>(p, nof_grids, d_grid, d_slot_index, d_iq, stream);
    default:
      return hipErrorInvalidValue;
  }
}

// ================================================================================================================
// Plain batched DFT (dft_processor): one workgroup per transform.
// ================================================================================================================
template <int SIGN, int N>
__global__ __launch_bounds__(Plan<N>::T) void dft_kernel(const float2* __restrict__ tw, const float2* __restrict__ d_in,
                                                         float2* __restrict__ d_out)
{
  __shared__ float2 lds[N + N / 16 + 16];
  const uint32_t    tid = threadIdx.x;
  const float2*     in  = d_in + (size_t)blockIdx.x * N;
  float2*           out = d_out + (size_t)blockIdx.x * N;
  const TwiddleBase<N> tb = load_twiddle_base<SIGN, N>(tw, tid);
  float2            a[Plan<N>::R0];
#pragma unroll
  for (int k = 0; k != Plan<N>::R0; ++k) {
    uint32_t i = first_stage_index<N>(tid, k);
    a[k]       = (i < N) ? in[i] : make_float2(0.f, 0.f);
  }
  auto store = [&](uint32_t i, float2 v) { out[i] = v; };
  fft_from_registers<SIGN, N>(a, tb, lds, tw, tid, store);
}

template <int N>
static hipError_t launch_dft_n(int inverse, uint32_t batch, const float2* tw, const float2* d_in, float2* d_out,
                               hipStream_t stream)
{
  if (inverse) {
    hipLaunchKernelGGL((dft_kernel<+1, N>), dim3(batch), dim3(Plan<N>::T), 0, stream, tw, d_in, d_out);
  } else {
    hipLaunchKernelGGL((dft_kernel<-1, N>), dim3(batch), dim3(Plan<N>::T), 0, stream, tw, d_in, d_out);
  }
  return hipGetLastError();
}

bool dft_size_supported(uint32_t size)
{
  return size == 128 || size == 256 || size == 512 || size == 1024 || size == 2048 || size == 4096;
}

hipError_t launch_dft(uint32_t size, int inverse, uint32_t batch, const float2* tw, const float2* d_in, float2* d_out,
                      hipStream_t stream)
{
  if (batch == 0) {
    return hipSuccess;
  }
  switch (size) {
    case 4096:
      return launch_dft_n<4096>(inverse, batch, tw, d_in, d_out, stream);
    case 2048:
      return launch_dft_n<2048>(inverse, batch, tw, d_in, d_out, stream);
    case 1024:
      return launch_dft_n<1024>(inverse, batch, tw, d_in, d_out, stream);
    case 512:
      return launch_dft_n<512>(inverse, batch, tw, d_in, d_out, stream);
    case 256:
      return launch_dft_n<256>(inverse, batch, tw, d_in, d_out, stream);
    case 128:
      return launch_dft_n<128>(inverse, batch, tw, d_in, d_out, stream);
    default:
      return hipErrorInvalidValue;
  }
}

} // namespace nrphy
