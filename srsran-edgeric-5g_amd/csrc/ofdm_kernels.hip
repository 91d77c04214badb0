// OFDM modulator and DFT kernels for gfx950 (MI355X).
//
// One workgroup transforms one OFDM symbol of one (grid, port).  The symbol's resource-grid row goes from HBM
// straight into the registers of the first butterfly stage (bin placement is index arithmetic, the guard zeros come
// from the buffer range check); the transform runs in LDS (Stockham autosort, radix-16/8/4/2 register butterflies
// written with packed-FP32 instructions).  Twiddles are powers of one table value per thread and stage, built in
// registers, so the only global traffic is the algorithmic one: grid in, IQ out.  The last stage applies phase
// compensation x scale and writes the useful part plus the cyclic prefix.
// Replaces ofdm_symbol_modulator_impl::modulate (R/lib/phy/lower/modulation/ofdm_modulator_impl.cpp:56-100) and
// dft_processor_generic_impl::run (R/lib/phy/generic_functions/dft_processor_generic_impl.cpp:14-218).
#include "bits_device.h"

#include <type_traits>

// Probes of the profiling variants (-DNRPHY_PROBES, see pdsch_kernels.hip): bit 0 drops the IQ stores, bit 1 the grid loads
// (buffer ranges of zero bytes), bit 2 takes the grids first to last.  Not in the product library.
#ifdef NRPHY_PROBES
#define NRPHY_PROBE(p) ((p).probe)
#else
#define NRPHY_PROBE(p) 0u
#endif

namespace nrphy {

template <uint32_t V>
using Const = std::integral_constant<uint32_t, V>;

// f(Const<0>{}), ..., f(Const<COUNT - 1>{}): a loop whose index is a compile-time constant inside the body.
template <class F, uint32_t... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<uint32_t, I...>)
{
  (f(Const<I>{}), ...);
}
template <uint32_t COUNT, class F>
__device__ __forceinline__ void static_for(F&& f)
{
  static_for_impl(f, std::make_integer_sequence<uint32_t, COUNT>{});
}

// ---- complex arithmetic on packed FP32 ------------------------------------------------------------------------
// A complex number is one 64-bit VGPR pair (re, im) and every operation below is one or two v_pk_*_f32
// instructions.  hipcc pairs scalar float code into packed instructions on its own, but it cannot negate or swap one
// half of an operand (it emits both variants and v_mov's the halves back together), so the operations that need
// op_sel / neg_lo / neg_hi are written out: the register butterflies are VALU-bound, not memory-bound.
typedef float cf __attribute__((ext_vector_type(2)));

__device__ __forceinline__ cf cadd(cf a, cf b)
{
  return a + b;
}
__device__ __forceinline__ cf csub(cf a, cf b)
{
  return a - b;
}
// a * b:  t = (a.im b.im, a.re b.im);  result = (a.re b.re - t.lo, a.im b.re + t.hi).
__device__ __forceinline__ cf cmul(cf a, cf b)
{
  cf t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
  return r;
}
// The same with a wave-uniform b held in an SGPR pair (constants, the per-symbol phase).
__device__ __forceinline__ cf cmul_uniform(cf a, cf b)
{
  cf t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "s"(b), "v"(t));
  return r;
}
// a + j b = (a.re - b.im, a.im + b.re) and a - j b = (a.re + b.im, a.im - b.re).
__device__ __forceinline__ cf add_jb(cf a, cf b)
{
  cf r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ cf sub_jb(cf a, cf b)
{
  cf r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// a + SIGN j b, a - SIGN j b (SIGN = +1: inverse transform, -1: direct).
template <int SIGN>
__device__ __forceinline__ cf add_sjb(cf a, cf b)
{
  return SIGN > 0 ? add_jb(a, b) : sub_jb(a, b);
}
template <int SIGN>
__device__ __forceinline__ cf sub_sjb(cf a, cf b)
{
  return SIGN > 0 ? sub_jb(a, b) : add_jb(a, b);
}
__device__ __forceinline__ cf make_cf(float re, float im)
{
  cf r = {re, im};
  return r;
}

// ---- register butterflies: a[k] <- sum_n a[n] * exp(SIGN * 2 pi i n k / R), natural order in and out ---------
template <int SIGN>
__device__ __forceinline__ void dft2(cf& a0, cf& a1)
{
  cf t = a0;
  a0   = cadd(t, a1);
  a1   = csub(t, a1);
}

// ROT2: input a2 still has to be multiplied by SIGN j (a twiddle of the enclosing transform, folded in for free).
template <int SIGN, bool ROT2 = false>
__device__ __forceinline__ void dft4(cf& a0, cf& a1, cf& a2, cf& a3)
{
  cf p0 = ROT2 ? add_sjb<SIGN>(a0, a2) : cadd(a0, a2);
  cf q0 = ROT2 ? sub_sjb<SIGN>(a0, a2) : csub(a0, a2);
  cf p1 = cadd(a1, a3), d = csub(a1, a3);
  a0    = cadd(p0, p1);
  a1    = add_sjb<SIGN>(q0, d);
  a2    = csub(p0, p1);
  a3    = sub_sjb<SIGN>(q0, d);
}

template <int SIGN, int R>
struct Butterfly;

template <int SIGN>
struct Butterfly<SIGN, 2> {
  static __device__ __forceinline__ void run(cf (&a)[2]) { dft2<SIGN>(a[0], a[1]); }
};
template <int SIGN>
struct Butterfly<SIGN, 4> {
  static __device__ __forceinline__ void run(cf (&a)[4]) { dft4<SIGN>(a[0], a[1], a[2], a[3]); }
};
template <int SIGN>
struct Butterfly<SIGN, 8> {
  // 8 = 2 x 4: X[k1 + 2 k2] = sum_{n2<4} W8^(n2 k1) W4^(n2 k2) [ sum_{n1<2} x[4 n1 + n2] W2^(n1 k1) ].
  static __device__ __forceinline__ void run(cf (&a)[8])
  {
    constexpr float h = 0.70710678118654752440f;
    dft2<SIGN>(a[0], a[4]);
    dft2<SIGN>(a[1], a[5]);
    dft2<SIGN>(a[2], a[6]);
    dft2<SIGN>(a[3], a[7]);
    // k1 = 1 row: multiply by W8^n2, n2 = 1, 2, 3 (n2 = 2 is SIGN j, folded into the butterfly).
    a[5] = cmul_uniform(a[5], make_cf(h, SIGN * h));
    a[7] = cmul_uniform(a[7], make_cf(-h, SIGN * h));
    dft4<SIGN>(a[0], a[1], a[2], a[3]);       // k1 = 0: X[0], X[2], X[4], X[6]
    dft4<SIGN, true>(a[4], a[5], a[6], a[7]); // k1 = 1: X[1], X[3], X[5], X[7]
    cf x1 = a[4], x2 = a[1], x3 = a[5], x4 = a[2], x5 = a[6], x6 = a[3];
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
  static __device__ __forceinline__ void run(cf (&a)[16])
  {
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    // Inner transforms over n1 (stride 4) for each n2; result index k1 replaces n1.
    dft4<SIGN>(a[0], a[4], a[8], a[12]);
    dft4<SIGN>(a[1], a[5], a[9], a[13]);
    dft4<SIGN>(a[2], a[6], a[10], a[14]);
    dft4<SIGN>(a[3], a[7], a[11], a[15]);
    // Twiddles W16^(n2 k1) on element a[4 k1 + n2]; W16^4 = SIGN j on a[10] is folded into its butterfly.
    a[5]  = cmul_uniform(a[5], make_cf(c1, SIGN * s1));    // 1*1
    a[6]  = cmul_uniform(a[6], make_cf(h, SIGN * h));      // 2*1
    a[7]  = cmul_uniform(a[7], make_cf(s1, SIGN * c1));    // 3*1
    a[9]  = cmul_uniform(a[9], make_cf(h, SIGN * h));      // 1*2
    a[11] = cmul_uniform(a[11], make_cf(-h, SIGN * h));    // 3*2
    a[13] = cmul_uniform(a[13], make_cf(s1, SIGN * c1));   // 1*3
    a[14] = cmul_uniform(a[14], make_cf(-h, SIGN * h));    // 2*3
    a[15] = cmul_uniform(a[15], make_cf(-c1, -SIGN * s1)); // 3*3 = 9 -> W16^9
    // Outer transforms over n2 for each k1; result a[4 k1 + k2] = X[k1 + 4 k2].
    dft4<SIGN>(a[0], a[1], a[2], a[3]);
    dft4<SIGN>(a[4], a[5], a[6], a[7]);
    dft4<SIGN, true>(a[8], a[9], a[10], a[11]);
    dft4<SIGN>(a[12], a[13], a[14], a[15]);
    // Transpose 4x4 to natural order: X[k1 + 4 k2] currently at a[4 k1 + k2].
    cf t;
    t = a[1];  a[1] = a[4];   a[4] = t;
    t = a[2];  a[2] = a[8];   a[8] = t;
    t = a[3];  a[3] = a[12];  a[12] = t;
    t = a[6];  a[6] = a[9];   a[9] = t;
    t = a[7];  a[7] = a[13];  a[13] = t;
    t = a[11]; a[11] = a[14]; a[14] = t;
  }
};

// Radix 3: y0 = a0 + (a1 + a2), y1,2 = a0 - (a1 + a2) / 2 +- SIGN j (sqrt(3) / 2) (a1 - a2).
template <int SIGN>
__device__ __forceinline__ void dft3(cf& a0, cf& a1, cf& a2)
{
  constexpr float s = 0.86602540378443864676f;
  const cf        t = cadd(a1, a2), d = csub(a1, a2);
  const cf        u = a0 - 0.5f * t, v = s * d;
  a0                = cadd(a0, t);
  a1                = add_sjb<SIGN>(u, v);
  a2                = sub_sjb<SIGN>(u, v);
}
template <int SIGN>
struct Butterfly<SIGN, 3> {
  static __device__ __forceinline__ void run(cf (&a)[3]) { dft3<SIGN>(a[0], a[1], a[2]); }
};
template <int SIGN>
struct Butterfly<SIGN, 6> {
  // 6 = 2 x 3: X[k1 + 2 k2] = sum_{n2<3} W6^(n2 k1) W3^(n2 k2) [ sum_{n1<2} x[3 n1 + n2] W2^(n1 k1) ].
  static __device__ __forceinline__ void run(cf (&a)[6])
  {
    constexpr float s = 0.86602540378443864676f;
    dft2<SIGN>(a[0], a[3]);
    dft2<SIGN>(a[1], a[4]);
    dft2<SIGN>(a[2], a[5]);
    a[4] = cmul_uniform(a[4], make_cf(0.5f, SIGN * s));  // W6^1
    a[5] = cmul_uniform(a[5], make_cf(-0.5f, SIGN * s)); // W6^2
    dft3<SIGN>(a[0], a[1], a[2]); // k1 = 0: X[0], X[2], X[4]
    dft3<SIGN>(a[3], a[4], a[5]); // k1 = 1: X[1], X[3], X[5]
    const cf x1 = a[3], x2 = a[1], x3 = a[4], x4 = a[2];
    a[1] = x1;
    a[2] = x2;
    a[3] = x3;
    a[4] = x4;
  }
};
template <int SIGN>
struct Butterfly<SIGN, 12> {
  // 12 = 4 x 3: X[k1 + 4 k2] = sum_{n2<3} W12^(n2 k1) W3^(n2 k2) [ sum_{n1<4} x[3 n1 + n2] W4^(n1 k1) ].
  static __device__ __forceinline__ void run(cf (&a)[12])
  {
    constexpr float s = 0.86602540378443864676f;
    dft4<SIGN>(a[0], a[3], a[6], a[9]);
    dft4<SIGN>(a[1], a[4], a[7], a[10]);
    dft4<SIGN>(a[2], a[5], a[8], a[11]);
    // Twiddles W12^(n2 k1) on a[3 k1 + n2].
    a[4]  = cmul_uniform(a[4], make_cf(s, SIGN * 0.5f));    // 1*1
    a[5]  = cmul_uniform(a[5], make_cf(0.5f, SIGN * s));    // 2*1
    a[7]  = cmul_uniform(a[7], make_cf(0.5f, SIGN * s));    // 1*2
    a[8]  = cmul_uniform(a[8], make_cf(-0.5f, SIGN * s));   // 2*2
    a[10] = cmul_uniform(a[10], make_cf(0.f, (float)SIGN)); // 1*3: SIGN j
    a[11] = make_cf(-a[11].x, -a[11].y);                    // 2*3: -1
    dft3<SIGN>(a[0], a[1], a[2]);   // k1 = 0: X[0], X[4], X[8]
    dft3<SIGN>(a[3], a[4], a[5]);   // k1 = 1: X[1], X[5], X[9]
    dft3<SIGN>(a[6], a[7], a[8]);   // k1 = 2: X[2], X[6], X[10]
    dft3<SIGN>(a[9], a[10], a[11]); // k1 = 3: X[3], X[7], X[11]
    cf x[12];
#pragma unroll
    for (int k1 = 0; k1 != 4; ++k1) {
#pragma unroll
      for (int k2 = 0; k2 != 3; ++k2) {
        x[k1 + 4 * k2] = a[3 * k1 + k2];
      }
    }
#pragma unroll
    for (int k = 0; k != 12; ++k) {
      a[k] = x[k];
    }
  }
};

// a[j] *= b^j for j = 1..R-1.  Powers are built from b^2, b^4, b^8 (squarings) so that every power is at most
// three multiplications deep (a few ulp), and applied at once to keep few values live.
template <int R>
__device__ __forceinline__ void apply_twiddle_powers(cf b, cf (&a)[R])
{
  a[1] = cmul(a[1], b);
  if constexpr (R > 2) {
    const cf b2 = cmul(b, b);
    a[2]        = cmul(a[2], b2);
    if constexpr (R > 3) {
      a[3] = cmul(a[3], cmul(b2, b));
    }
    if constexpr (R > 4) {
      const cf b4 = cmul(b2, b2);
      a[4]        = cmul(a[4], b4);
      if constexpr (R > 5) {
        a[5] = cmul(a[5], cmul(b4, b));
      }
      if constexpr (R > 6) {
        a[6] = cmul(a[6], cmul(b4, b2));
        a[7] = cmul(a[7], cmul(b4, cmul(b2, b)));
      }
      if constexpr (R > 8) {
        const cf b8 = cmul(b4, b4);
        a[8]        = cmul(a[8], b8);
        a[9]        = cmul(a[9], cmul(b8, b));
        a[10]       = cmul(a[10], cmul(b8, b2));
        a[11]       = cmul(a[11], cmul(b8, cmul(b2, b)));
        if constexpr (R > 12) {
          const cf b12 = cmul(b8, b4);
          a[12]        = cmul(a[12], b12);
          a[13]        = cmul(a[13], cmul(b12, b));
          a[14]        = cmul(a[14], cmul(b12, b2));
          a[15]        = cmul(a[15], cmul(b12, cmul(b2, b)));
        }
      }
    }
  }
}

// LDS index padding: one extra element every 16 keeps the stride-16 stores of the first stage off a single bank.
__device__ __forceinline__ uint32_t pad(uint32_t i)
{
  return i + (i >> 4);
}

// pad(base + k * C) for k = 0, 1, ... from pb = pad(base): where the step is a multiple of 16 elements the padding is affine in k --
// (base + k C) >> 4 = (base >> 4) + k C / 16 -- so ONE address register serves all k and the rest is the instruction's immediate
// offset.  Written as pad(base + k * C) the compiler does not see that and keeps a register per address: sixteen per stage and
// direction, alive across the whole kernel -- 64 of the modulator's 160 vector registers.
template <int C>
__device__ __forceinline__ uint32_t pad_step(uint32_t base, uint32_t pb, int k)
{
  if constexpr (C % 16 == 0) {
    return pb + (uint32_t)k * (uint32_t)(C + C / 16);
  } else {
    return pad(base + (uint32_t)k * (uint32_t)C);
  }
}

// Radix plans: N = R0 * R1 * R2 * R3 (R2 = 1 when two stages suffice, R3 = 1 when three do), T = threads per
// transform = N / 16.
struct ThreeStages {
  static constexpr int R3 = 1;
};
template <int N>
struct Plan;
template <> struct Plan<4096> : ThreeStages { static constexpr int R0 = 16, R1 = 16, R2 = 16, T = 256; };
template <> struct Plan<2048> : ThreeStages { static constexpr int R0 = 16, R1 = 16, R2 = 8, T = 128; };
template <> struct Plan<1024> : ThreeStages { static constexpr int R0 = 16, R1 = 16, R2 = 4, T = 64; };
template <> struct Plan<512>  : ThreeStages { static constexpr int R0 = 16, R1 = 16, R2 = 2, T = 64; };
template <> struct Plan<256>  : ThreeStages { static constexpr int R0 = 16, R1 = 16, R2 = 1, T = 64; };
template <> struct Plan<128>  : ThreeStages { static constexpr int R0 = 16, R1 = 8, R2 = 1, T = 64; };
// 3 * 2^k (the 23.04 MHz family of sampling rates): the factor 3 (x 1, 2, 4) is the last, twiddle-free stage.
template <> struct Plan<3072> : ThreeStages { static constexpr int R0 = 16, R1 = 16, R2 = 12, T = 256; };
template <> struct Plan<1536> : ThreeStages { static constexpr int R0 = 16, R1 = 16, R2 = 6, T = 128; };
template <> struct Plan<768>  : ThreeStages { static constexpr int R0 = 16, R1 = 16, R2 = 3, T = 64; };
template <> struct Plan<384>  : ThreeStages { static constexpr int R0 = 16, R1 = 8, R2 = 3, T = 64; };
// 6144 = 3 * 2^11 and 4608 = 9 * 2^9 (the next sizes of the reference's generic DFT, dft_processor_generic_impl.cpp:201-202;
// 6144 is the 15 kHz transform of a 92.16 MHz sampling rate): four stages, still one transform per workgroup in LDS
// (52 KB / 39 KB); the last, twiddle-free stage is the radix 3.
template <> struct Plan<6144> { static constexpr int R0 = 16, R1 = 16, R2 = 8, R3 = 3, T = 384; };
template <> struct Plan<4608> { static constexpr int R0 = 16, R1 = 16, R2 = 6, R3 = 3, T = 288; };

// Input index k-th element of the first-stage butterfly of thread `tid`: x[tid + k * N / R0].
template <int N>
__device__ __forceinline__ uint32_t first_stage_index(uint32_t tid, int k)
{
  return tid + k * (N / Plan<N>::R0);
}

// Twiddle bases of a thread: stage s multiplies output j of its butterfly by (w_n^p)^j with w_n^p = tw[p * S].
template <int N>
struct TwiddleBase {
  cf b0, b1, b2; // first, second and (four-stage plans) third stage; the last stage of a plan has n1 = 1: no twiddles
};

template <int SIGN, int N>
__device__ __forceinline__ TwiddleBase<N> load_twiddle_base(const float2* __restrict__ tw, uint32_t tid)
{
  using P = Plan<N>;
  TwiddleBase<N> t;
  // Stage 0: S = 1, p = butterfly index = tid (threads beyond N/R0 butterflies are idle in that stage).
  const float2 w0 = tw[tid % N];
  // Stage 1: S = R0, p = b / R0 for butterfly b = tid (+ it*T); only the first iteration's base is kept here, the
  // others are derived in the stage (see stage_lds).
  const float2 w1 = tw[((tid / P::R0) * P::R0) % N];
  t.b0            = make_cf(w0.x, SIGN < 0 ? -w0.y : w0.y);
  t.b1            = make_cf(w1.x, SIGN < 0 ? -w1.y : w1.y);
  t.b2            = t.b1;
  if constexpr (P::R3 != 1) { // stage 2: S = R0 R1
    const float2 w2 = tw[((tid / (P::R0 * P::R1)) * (P::R0 * P::R1)) % N];
    t.b2            = make_cf(w2.x, SIGN < 0 ? -w2.y : w2.y);
  }
  return t;
}

// First Stockham stage on registers a[k] = x[tid + k N/R0]: y[R0 p + j] = DFT(a)[j] * w^(j p), p = tid, S = 1.
template <int SIGN, int N>
__device__ __forceinline__ void stage_first(cf (&a)[Plan<N>::R0], cf base, cf* lds, uint32_t tid)
{
  constexpr int R  = Plan<N>::R0;
  constexpr int NB = N / R;
  if (NB >= Plan<N>::T || tid < NB) {
    Butterfly<SIGN, R>::run(a);
    apply_twiddle_powers<R>(base, a);
    const uint32_t pb = pad(R * tid);
#pragma unroll
    for (int j = 0; j != R; ++j) {
      lds[R == 16 ? pb + j : pad(R * tid + j)] = a[j]; // (R = 16: the sixteen outputs share one padding step)
    }
  }
  __syncthreads();
}

// A sink that wants the R outputs of a last-stage thread in one call says so with a member constant `all_outputs`.
// (std::is_invocable on the sink's call operator is no test for it: the trait is evaluated inside host-side library templates,
// where a __device__ operator is never viable -- it answered "no" for every sink, and until round 4 both modulator kernels
// silently took the one-output-at-a-time form.)
template <typename Store, typename = void>
struct takes_all_outputs : std::false_type {};
template <typename Store>
struct takes_all_outputs<Store, std::void_t<decltype(Store::all_outputs)>> : std::bool_constant<Store::all_outputs> {};

// A later Stockham stage (decimation in frequency, autosort).  n = N / S is the current transform length.
//   a[k] = x[q + S (p + k n/R)],   y[q + S (R p + j)] = DFT_R(a)[j] * w_n^(j p),   p < n/R, q < S.
template <int SIGN, int N, int T, int R, int S, bool LAST, typename Store>
__device__ __forceinline__ void stage_lds(cf* lds, const float2* __restrict__ tw, cf base, uint32_t tid,
                                          Store store)
{
  constexpr int NB    = N / R;
  constexpr int ITERS = (NB + T - 1) / T;
  constexpr int n1    = N / S / R;
  cf            a[ITERS][R];
#pragma unroll
  for (int it = 0; it != ITERS; ++it) {
    uint32_t b = tid + it * T;
    if (NB % T == 0 || b < NB) {
      uint32_t p = b / S, q = b % S;
      const uint32_t rb = q + S * p, prb = pad(rb);
#pragma unroll
      for (int k = 0; k != R; ++k) {
        a[it][k] = lds[pad_step<S * n1>(rb, prb, k)];
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
      if constexpr (n1 > 1) {
        cf bs = base;
        if (it > 0) { // p differs per iteration: fetch this iteration's base (rare plans only)
          const float2 w = tw[(p * S) % N];
          bs             = make_cf(w.x, SIGN < 0 ? -w.y : w.y);
        }
        apply_twiddle_powers<R>(bs, a[it]);
      }
      if constexpr (LAST) {
        // The last stage has S * R = N, hence p = 0: output index = q + S * j, a per-thread part and a constant.
        static_assert(S * R == N, "last stage");
        if constexpr (takes_all_outputs<Store>::value) {
          store(q, Const<S>{}, a[it]); // the sink takes the thread's R outputs together (it may pair them up)
        } else {
          static_for<R>([&](auto J) { store(q, Const<S * decltype(J)::value>{}, Const<S>{}, a[it][decltype(J)::value]); });
        }
      } else {
        const uint32_t wb = q + S * R * p, pwb = pad(wb);
#pragma unroll
        for (int j = 0; j != R; ++j) {
          lds[pad_step<S>(wb, pwb, j)] = a[it][j];
        }
      }
    }
  }
  if (!LAST) {
    __syncthreads();
  }
}

template <int SIGN, int N, typename Store>
__device__ __forceinline__ void fft_from_registers(cf (&a)[Plan<N>::R0], const TwiddleBase<N>& tb, cf* lds,
                                                   const float2* __restrict__ tw, uint32_t tid, Store store)
{
  using P = Plan<N>;
  constexpr int T = P::T;
  auto no_store   = [](uint32_t, auto, auto, cf) {};
  stage_first<SIGN, N>(a, tb.b0, lds, tid);
  if constexpr (P::R2 == 1) {
    stage_lds<SIGN, N, T, P::R1, P::R0, true>(lds, tw, tb.b1, tid, store);
  } else if constexpr (P::R3 == 1) {
    stage_lds<SIGN, N, T, P::R1, P::R0, false>(lds, tw, tb.b1, tid, no_store);
    stage_lds<SIGN, N, T, P::R2, P::R0 * P::R1, true>(lds, tw, tb.b1, tid, store);
  } else {
    stage_lds<SIGN, N, T, P::R1, P::R0, false>(lds, tw, tb.b1, tid, no_store);
    stage_lds<SIGN, N, T, P::R2, P::R0 * P::R1, false>(lds, tw, tb.b2, tid, no_store);
    stage_lds<SIGN, N, T, P::R3, P::R0 * P::R1 * P::R2, true>(lds, tw, tb.b2, tid, store);
  }
  __syncthreads(); // the LDS buffer is reused by the next transform of this workgroup
}

// ================================================================================================================
// OFDM slot modulator.  blockIdx.x = (grid * nof_ports + port) * groups + group; a workgroup modulates SPW
// consecutive symbols of its (grid, port).  With SPW > 1 the next symbol's row is fetched while the current one is
// transformed.
// ================================================================================================================
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

// Cache policy of the IQ stores: non-temporal + sc1 (buffer aux bits 1 and 4).  The samples are not read again on the
// device.  Measured over the whole step (A/B on one box, 60 steps, 16-byte stores): default policy 0.52 ms, nt 0.47-0.50,
// nt + sc1 0.45 ms (6.1 TB/s); with nt the PDSCH launches that follow also run 4 % faster because 2 GB of IQ no longer
// pass through the caches.  nt + sc0 and nt + sc0 + sc1 were in between, sc0 + sc1 without nt slower.
#ifndef NRPHY_IQ_STORE_AUX
#define NRPHY_IQ_STORE_AUX 18 // (0 / 2 / 16 for A/B builds)
#endif
constexpr int AUX_NT = NRPHY_IQ_STORE_AUX;

// blockIdx = (symbol group, port, grid): no index arithmetic to undo, every per-symbol quantity is wave-uniform and
// lives in SGPRs.  A workgroup modulates SPW consecutive symbols of its (grid, port); the row of the next symbol is
// fetched into registers before the current one is transformed, so reads stay in flight during the butterflies.
// The row is read through a buffer descriptor sized to the row: bin placement (ofdm_modulator_impl.cpp:83-87: lower
// grid half -> top bins, upper half -> bins from DC, guard bins zero) is j = (i + rg_size / 2) mod N and the
// hardware range check supplies the zeros of the guard bins.  The output goes through a second descriptor with the
// constant part of every address in the scalar offset.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// Output side of ofdm_kernel: phase compensation x scale, then the symbol and its cyclic prefix (the tail of the
// symbol, ofdm_modulator_impl.cpp:92-99) through the buffer descriptor `rsrc` whose record 0 is the first sample of
// the prefix.  A thread of the last stage holds outputs q + S j; neighbouring lanes hold neighbouring samples, so every store
// instruction of a wave writes 512 contiguous bytes.  The prefix never exceeds N / 4 (extended CP), so only the outputs of the
// last quarter test for it.
// (A form that swapped half of the outputs between lane pairs to write 16-byte pairs of consecutive samples existed from
// round 2 on and was never selected -- see takes_all_outputs.  Selected at last in round 4 it was slower, 0.58 against 0.46 ms
// per 1024 config-3 slots: 80 more vector instructions per symbol for the swaps outweigh the wider stores; removed,
// profiles/r04_ofdm_sinks.txt.)
#ifndef NRPHY_SINK_ALL_WIRE
#define NRPHY_SINK_ALL_WIRE 1 // 0: the wire-format sink one output per call, always through the exact path (the form that ran until round 4)
#endif
template <int N>
struct IqSink {
  __amdgpu_buffer_rsrc_t rsrc;
  cf                     ph;
  uint32_t               cp;

  __device__ __forceinline__ void prefix_copy(uint32_t i, u32x2_t d) const
  {
    if (i >= N - cp) {
      __builtin_amdgcn_raw_buffer_store_b64(d, rsrc, (int)((i - (N - cp)) * 8u), 0, AUX_NT);
    }
  }

  // One output: idx = q + B.
  template <uint32_t B, uint32_t S>
  __device__ __forceinline__ void operator()(uint32_t q, Const<B>, Const<S>, cf v) const
  {
    const cf      y = cmul_uniform(v, ph);
    const u32x2_t d = {__float_as_uint(y.x), __float_as_uint(y.y)};
    __builtin_amdgcn_raw_buffer_store_b64(d, rsrc, (int)(q * 8u), (int)((cp + B) * 8u), AUX_NT);
    if constexpr (B + S > N - N / 4) {
      prefix_copy(q + B, d);
    }
  }

};

// The same output side with the samples leaving as complex int16: every sample goes through
// the reference's chain behind the modulator in its order of roundings -- amplitude controller (gain, optional
// clipping of real and imaginary parts: amplitude_controller_clipping_impl.cpp:31-68) and the radio's sample conversion
// (value * scale, round to nearest even, saturate: R/lib/srsvec/conversion.cpp:29-65) -- and 4 bytes instead of 8 go to
// HBM.  The controller's measurements (sum and maximum of |x|^2 after the gain, clipped parts) accumulate per thread;
// a sample inside the cyclic prefix counts twice, as in the buffer the reference measures.
// Wave-wide reductions without LDS traffic: butterflies inside a row of 16 lanes by DPP (lane ^ 1, lane ^ 2, mirrored halves,
// mirrored row), then the four row results through scalar registers.  Every lane returns the result.  Zero must be the
// operation's identity (sums and maxima of non-negative values here): a disabled lane contributes zero inside a row (bound_ctrl),
// and a row without active lanes -- the second half of the last wavefront of a 288-thread workgroup (N = 4608) -- is left out
// (its registers hold whatever an earlier wavefront left there).  Rows are active or inactive as a whole for every workgroup size
// in use (multiples of 32).
template <typename Op>
__device__ __forceinline__ uint32_t wave_reduce_bits(uint32_t v, Op op)
{
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));  // quad_perm [1, 0, 3, 2]
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));  // quad_perm [2, 3, 0, 1]
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true)); // row_half_mirror
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true)); // row_mirror
  const uint64_t active = __builtin_amdgcn_read_exec();
  const uint32_t r0 = (active >> 0) & 1u ? (uint32_t)__builtin_amdgcn_readlane((int)v, 0) : 0u;
  const uint32_t r1 = (active >> 16) & 1u ? (uint32_t)__builtin_amdgcn_readlane((int)v, 16) : 0u;
  const uint32_t r2 = (active >> 32) & 1u ? (uint32_t)__builtin_amdgcn_readlane((int)v, 32) : 0u;
  const uint32_t r3 = (active >> 48) & 1u ? (uint32_t)__builtin_amdgcn_readlane((int)v, 48) : 0u;
  return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ float wave_sum(float v)
{
  return __uint_as_float(wave_reduce_bits(__float_as_uint(v), [](uint32_t a, uint32_t b) { return __float_as_uint(__uint_as_float(a) + __uint_as_float(b)); }));
}
__device__ __forceinline__ float wave_max(float v)
{
  return __uint_as_float(wave_reduce_bits(__float_as_uint(v), [](uint32_t a, uint32_t b) { return __float_as_uint(fmaxf(__uint_as_float(a), __uint_as_float(b))); }));
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
  return wave_reduce_bits(v, [](uint32_t a, uint32_t b) { return a + b; });
}

#ifndef NRPHY_WIRE_EXP
#define NRPHY_WIRE_EXP 0 // profiling experiments (profiles/make_variant.sh): 1 = no power measurements, 2 = no exact path, 3 = no atomics
#endif
template <int N>
struct IqSinkCi16 {
  static constexpr bool  all_outputs = NRPHY_SINK_ALL_WIRE != 0;
  __amdgpu_buffer_rsrc_t rsrc;
  cf                     ph;
  uint32_t               cp;
  float                  gain, ceiling, scale, limit_sq;
  float*                 sum;
  float*                 peak;
  uint32_t*              clipped;

  // A sample through the amplitude controller and the int16 conversion, in three pieces.  PREFIX: the index may lie in the
  // part of the symbol that is also written as the cyclic prefix (those samples count twice in the measurements).
  //   measure: phase x scale, gain, power sum and peak (per lane).
  //   pack_exact: clipping with its count, then round-to-nearest-even of v * scale saturated to int16 (the reference's
  //     vector path) as a clamp to [-32768, 32767] followed by the addition of 1.5 * 2^23, whose low 16 result bits are the
  //     rounded integer.
  //   pack_plain: the same when no sample of the wave has a power above `limit_sq` = min(ceiling, largest magnitude that
  //     cannot saturate)^2, so that no component reaches the limit: both clamps are identities and nothing is counted.
  template <bool PREFIX>
  __device__ __forceinline__ cf measure(cf v, uint32_t idx, float& reach) const
  {
#pragma clang fp contract(off) // every product is rounded on its own, as the reference's separate steps are (the squares are made
    // values of their own as well: see pack_plain)
    const cf    g  = cmul_uniform(v, ph) * gain; // packed: (re, im) x gain
#if NRPHY_WIRE_EXP != 1
    cf          sq = g * g;
    asm("" : "+v"(sq));
    const float pw = sq.x + sq.y;
    *sum += pw;
    reach = __builtin_fmaxf(reach, pw);
    if constexpr (PREFIX) {
      *sum += idx >= N - cp ? pw : 0.f;
    }
#endif
    return g;
  }

  static constexpr float MAGIC = 12582912.f;

  template <bool PREFIX>
  __device__ __forceinline__ uint32_t pack_exact(cf g, uint32_t idx) const
  {
#pragma clang fp contract(off)
    const cf cl = make_cf(__builtin_amdgcn_fmed3f(g.x, -ceiling, ceiling), __builtin_amdgcn_fmed3f(g.y, -ceiling, ceiling));
    uint32_t c  = (g.x != cl.x ? 1u : 0u) + (g.y != cl.y ? 1u : 0u);
    if constexpr (PREFIX) {
      c = idx >= N - cp ? 2u * c : c;
    }
    *clipped += c;
    const cf sc = cl * scale;
    const cf r  = make_cf(__builtin_amdgcn_fmed3f(sc.x, -32768.f, 32767.f), __builtin_amdgcn_fmed3f(sc.y, -32768.f, 32767.f)) + MAGIC;
    return __builtin_amdgcn_perm(__float_as_uint(r.y), __float_as_uint(r.x), 0x05040100u);
  }

  __device__ __forceinline__ uint32_t pack_plain(cf g) const
  {
#pragma clang fp contract(off)
    // The product must be rounded on its own before the addition (the reference rounds v * scale to float, then to the nearest
    // even integer).  When this file was still compiled with -ffp-contract=fast the backend fused the two whatever the pragma
    // above said: the shipped library had v_pk_fma_f32 here (86 instead of 82 in the N = 512 instance), and a fused multiply-add
    // breaks ties at x.5 by the unrounded product -- the seeded sweep found 1.5e-4 of the samples one LSB off, every one of them
    // such a tie (profiles/r04_fuzz_sweep_summary.txt).  The file is built without contraction now (build.py) AND the rounded
    // product is made a value of its own by the empty assembly statement.  (A `hipcc -c -save-temps` listing did NOT show the
    // fusion -- that pipeline happened not to fuse here; the code object inside the library did: profiles/disasm_lib.py.)
    cf sc = g * scale;
    asm("" : "+v"(sc));
    const cf r = sc + MAGIC;
    return __builtin_amdgcn_perm(__float_as_uint(r.y), __float_as_uint(r.x), 0x05040100u);
  }

  template <uint32_t B, uint32_t S>
  __device__ __forceinline__ void operator()(uint32_t q, Const<B>, Const<S>, cf v) const
  {
    constexpr bool PREFIX = B + S > N - N / 4;
    float          reach  = 0.f;
    const uint32_t d      = pack_exact<PREFIX>(measure<PREFIX>(v, q + B, reach), q + B);
    *peak                 = fmaxf(*peak, reach);
    __builtin_amdgcn_raw_buffer_store_b32(d, rsrc, (int)(q * 4u), (int)((cp + B) * 4u), AUX_NT);
    if constexpr (PREFIX) {
      if (q + B >= N - cp) {
        __builtin_amdgcn_raw_buffer_store_b32(d, rsrc, (int)((q + B - (N - cp)) * 4u), 0, AUX_NT);
      }
    }
  }

  // The thread's R outputs idx = q + S j: measured, packed (the plain way unless some lane of the wave has a sample above the
  // limit), and stored one 4-byte sample per lane: a wave instruction writes 256 contiguous bytes.  (Swapping halves between
  // lane pairs for 8-byte stores costs more vector instructions than the wider stores save: the kernel is bound by its
  // instruction stream, not by the stores -- profiles/r02_ofdm_wire_probes.txt.)
  template <uint32_t S, int R>
  __device__ __forceinline__ void operator()(uint32_t q, Const<S>, cf (&a)[R]) const
  {
    float reach = 0.f;
    cf    g[R];
    static_for<R>([&](auto J) {
      constexpr uint32_t j = decltype(J)::value;
      g[j]                 = measure<(S * (j + 1) > N - N / 4)>(a[j], q + S * j, reach);
    });
    *peak = fmaxf(*peak, reach);
    uint32_t w[R];
#if NRPHY_WIRE_EXP == 2
    if (false) {
#else
    if (__builtin_expect(__ballot(reach > limit_sq) != 0, 0)) {
#endif
      static_for<R>([&](auto J) {
        constexpr uint32_t j = decltype(J)::value;
        w[j]                 = pack_exact<(S * (j + 1) > N - N / 4)>(g[j], q + S * j);
      });
    } else {
      static_for<R>([&](auto J) { w[decltype(J)::value] = pack_plain(g[decltype(J)::value]); });
    }
    static_for<R>([&](auto J) {
      constexpr uint32_t j = decltype(J)::value;
      __builtin_amdgcn_raw_buffer_store_b32(w[j], rsrc, (int)(q * 4u), (int)((cp + S * j) * 4u), AUX_NT);
      if constexpr (S * (j + 1) > N - N / 4) {
        if (q + S * j >= N - cp) {
          __builtin_amdgcn_raw_buffer_store_b32(w[j], rsrc, (int)((q + S * j - (N - cp)) * 4u), 0, AUX_NT);
        }
      }
    });
  }
};

template <int N>
__device__ __forceinline__ void load_symbol_row(uint32_t (&raw)[Plan<N>::R0], const OfdmLaunch& p,
                                                const uint32_t* row, uint32_t t4)
{
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint32_t*>(row), 0, (int)((NRPHY_PROBE(p) & 2u) ? 0u : p.rg_size * 4u), 0x00020000);
#pragma unroll
  for (int k = 0; k != Plan<N>::R0; ++k) {
    uint32_t off = t4 + (uint32_t)k * (N / Plan<N>::R0) * 4u; // byte offset of element (i + rg / 2) mod N
    if ((N & (N - 1)) == 0) {
      off &= 4u * N - 1u;
    } else {
      off = off >= 4u * N ? off - 4u * N : off;
    }
    raw[k]             = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, 0, 0);
  }
}

// Workgroups per CU.  With one address register per stage (pad_step) the transforms need 88 (float) / 119 (wire format) vector
// registers instead of 160 / 165 and four workgroups fit a CU where three did.  The wire-format kernel wants the fourth -- its
// residents cover one another's barriers and waits: 0.41 -> 0.38 ms per 1024 config-3 slots --, the float kernel, which sits at the
// memory system's rate, does not: 0.47 -> 0.51 ms with four, so it keeps three (A/B on one box, three rounds,
// profiles/r04_ofdm_sinks.txt).  -DNRPHY_OFDM_WAVES=n: both kernels at n waves per SIMD (experiments).
#ifdef NRPHY_OFDM_WAVES
#define OFDM_OCCUPANCY __attribute__((amdgpu_waves_per_eu(NRPHY_OFDM_WAVES, NRPHY_OFDM_WAVES)))
#else
#define OFDM_OCCUPANCY __attribute__((amdgpu_waves_per_eu(WIRE ? 1 : 3, WIRE ? 4 : 3)))
#endif
template <int N, int SPW, bool WIRE>
__global__ __launch_bounds__(Plan<N>::T) OFDM_OCCUPANCY void ofdm_kernel(OfdmLaunch p, const uint32_t* __restrict__ d_grid,
                                                          const uint32_t* __restrict__ d_slot_index,
                                                          float2* __restrict__ d_iq)
{
  __shared__ cf     lds[N + N / 16 + 16];
  const uint32_t    tid  = threadIdx.x;
  const uint32_t    l0   = blockIdx.x * SPW;
  const uint32_t    l1   = (l0 + SPW < p.nsymb) ? l0 + SPW : p.nsymb;
  // The grids are taken last to first: the launch usually follows the one that wrote them first to last, and what the
  // memory-side cache still holds of them is the end (A/B on one box, three rounds: this launch 0.504 -> 0.496 ms, the whole
  // step +1.1 %; with non-temporal DM-RS / zero-fill stores 0.481 -> 0.467 ms; profiles/r03_codeblock_experiments.txt).
  // NRPHY_OFDM_PROBE bit 2 restores first to last.
  const uint32_t    gz   = (NRPHY_PROBE(p) & 4u) ? blockIdx.z : gridDim.z - 1u - blockIdx.z;
  const uint32_t    gp   = gz * p.nof_ports + blockIdx.y; // grid * nof_ports + port
  const uint32_t    slot = d_slot_index ? to_constant(d_slot_index)[gz] : 0u;
  const uint32_t    t4   = (tid + (p.rg_size >> 1)) * 4u;
  // (NRPHY_OFDM_PROBE bit 3: every workgroup reads the first (grid, port)'s rows -- the same loads from 183 KB that stay in the L2)
  const uint32_t*   rows = d_grid + ((NRPHY_PROBE(p) & 8u) ? (size_t)0 : (size_t)gp * NRPHY_NSYMB * p.rg_size);
  float2*           iq   = d_iq + (size_t)gp * p.slot_stride;
  float             w_sum = 0.f, w_peak = 0.f; // amplitude controller measurements of this thread (WIRE)
  uint32_t          w_clipped = 0;

  const TwiddleBase<N> tb = load_twiddle_base<+1, N>(p.twiddle, tid);
  uint32_t             raw[Plan<N>::R0];
  load_symbol_row<N>(raw, p, rows + (size_t)l0 * p.rg_size, t4);
  for (uint32_t l = l0; l < l1; ++l) {
    cf cur[Plan<N>::R0];
#pragma unroll
    for (int k = 0; k != Plan<N>::R0; ++k) {
      cur[k] = make_cf(__uint_as_float(raw[k] << 16), __uint_as_float(raw[k] & 0xFFFF0000u)); // cbf16: re low, im high
    }
    if (SPW > 1 && l + 1 < l1) {
      // The 16 load offsets are two instructions each; hidden from loop-invariant code motion they cost no
      // registers across the butterflies.
      uint32_t t4_now = t4;
      asm volatile("" : "+v"(t4_now));
      load_symbol_row<N>(raw, p, rows + (size_t)(l + 1) * p.rg_size, t4_now);
    }
    const uint32_t sym = slot * p.nsymb + l; // symbol index within the subframe
    const uint32_t cp  = to_constant(p.cp_len)[sym];
    const cf       ph  = make_cf(to_constant(p.phase)[sym].x, to_constant(p.phase)[sym].y);
    if constexpr (WIRE) {
      // complex int16 out: [grid][port][slot_stride] samples of 4 bytes
      uint32_t* iq16 = reinterpret_cast<uint32_t*>(d_iq) + (size_t)gp * p.slot_stride + to_constant(p.sym_offset)[sym];
      const __amdgpu_buffer_rsrc_t rsrc_out =
          __builtin_amdgcn_make_buffer_rsrc(iq16, 0, (int)((NRPHY_PROBE(p) & 1u) ? 0u : (N + cp) * 4u), 0x00020000);
      // no clipping = a ceiling nothing exceeds
      const IqSinkCi16<N> store = {rsrc_out, ph, cp, p.wire_gain, p.wire_clip != 0 ? p.wire_ceiling : __builtin_inff(), p.wire_scale,
                                   p.wire_limit, &w_sum, &w_peak, &w_clipped};
      fft_from_registers<+1, N>(cur, tb, lds, p.twiddle, tid, store);
    } else {
      const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(
          iq + to_constant(p.sym_offset)[sym], 0, (int)((NRPHY_PROBE(p) & 1u) ? 0u : (N + cp) * 8u), 0x00020000);
      const IqSink<N> store = {rsrc_out, ph, cp};
      fft_from_registers<+1, N>(cur, tb, lds, p.twiddle, tid, store);
    }
  }
  if constexpr (WIRE) {
    // The workgroup's measurements go to its own record (lane 0 of each wave through LDS, then one 16-byte store): atomics
    // on the [grid][port] record from every wave cost 0.13 ms per 1024 slots; wire_stats_kernel adds the records up, in a
    // fixed order.
    if (p.wire_stats != nullptr && NRPHY_WIRE_EXP != 3) { // wave-uniform
      w_sum     = wave_sum(w_sum);
      w_peak    = wave_max(w_peak);
      w_clipped = wave_sum(w_clipped);
      constexpr uint32_t NW  = (Plan<N>::T + WAVE - 1) / WAVE; // (288 threads at N = 4608: four wavefronts and a half)
      uint32_t*          red = reinterpret_cast<uint32_t*>(lds);
      __syncthreads(); // the last symbol's butterflies are done with the LDS
      if ((tid & (WAVE - 1)) == 0) {
        red[3 * (tid / WAVE) + 0] = __float_as_uint(w_sum);
        red[3 * (tid / WAVE) + 1] = __float_as_uint(w_peak);
        red[3 * (tid / WAVE) + 2] = w_clipped;
      }
      __syncthreads();
      if (tid == 0) {
        float    sum = 0.f, peak = 0.f;
        uint32_t clipped = 0;
        for (uint32_t w = 0; w != NW; ++w) {
          sum += __uint_as_float(red[3 * w]);
          peak = fmaxf(peak, __uint_as_float(red[3 * w + 1]));
          clipped += red[3 * w + 2];
        }
        p.wire_partials[(size_t)gp * gridDim.x + blockIdx.x] = make_uint4(__float_as_uint(sum), __float_as_uint(peak), clipped, 0u);
      }
    }
  }
}

// Adds the per-workgroup records of ofdm_kernel<.., WIRE> up into the caller's [grid][port] measurements.
__global__ void wire_stats_kernel(OfdmLaunch p, const uint32_t* __restrict__ d_slot_index, uint32_t nof_gp, uint32_t per_gp)
{
  const uint32_t gp = blockIdx.x * blockDim.x + threadIdx.x;
  if (gp >= nof_gp) {
    return;
  }
  float    sum = 0.f, peak = 0.f;
  uint32_t clipped = 0;
  for (uint32_t i = 0; i != per_gp; ++i) {
    const uint4 r = p.wire_partials[(size_t)gp * per_gp + i];
    sum += __uint_as_float(r.x);
    peak = fmaxf(peak, __uint_as_float(r.y));
    clipped += r.z;
  }
  const uint32_t slot = d_slot_index ? d_slot_index[gp / p.nof_ports] : 0u;
  const uint32_t last = slot * p.nsymb + p.nsymb - 1u;
  nrphy_amplitude_stats_t st;
  st.sum_power     = sum;
  st.peak_power    = peak;
  st.nof_clipped   = clipped;
  st.nof_samples   = p.sym_offset[last] + p.cp_len[last] + p.dft_size;
  p.wire_stats[gp] = st;
}

// Symbols per workgroup, the next symbol's grid row requested before the current symbol's butterflies.
// Complex float output: measured at 1024 slots in round 2, 1 -> 0.512 ms, 7 -> 0.519 ms; in round 4 (A/B on one box, two rounds,
// profiles/r04_ofdm_spw.txt) 1 -> 0.454 ms, 2 -> 0.51, 4 -> 0.53: the kernel sits at the memory system's rate for its access
// shape with one symbol per workgroup, and more only take parallelism away (1 also keeps small batches spread over the chip).
// Complex int16 output (half the store traffic, more arithmetic per sample): 1 -> 0.434-0.444 ms, 2 -> 0.419-0.420,
// 3 -> 0.411-0.419, 4 -> 0.447-0.452 -- there the row of symbol k + 1 arriving under the butterflies of symbol k pays.  With four
// workgroups per CU (late round 4) the steps flatten: 1 -> 0.392-0.399, 2 -> 0.3875 / 0.3876, 3 -> 0.392, 4 -> 0.42 (14 = 4 + 4 + 4 + 2),
// 5 -> 0.386-0.390: two it is -- seven equal units per slot and port, and a single slot spread over more workgroups.
#ifndef NRPHY_OFDM_SPW
#define NRPHY_OFDM_SPW 1
#endif
#ifndef NRPHY_OFDM_SPW_WIRE
#define NRPHY_OFDM_SPW_WIRE 2
#endif
constexpr int OFDM_SYMBOLS_PER_WG = NRPHY_OFDM_SPW, OFDM_SYMBOLS_PER_WG_WIRE = NRPHY_OFDM_SPW_WIRE;

template <int N>
static hipError_t launch_ofdm_n(const OfdmLaunch& p, uint32_t nof_grids, const uint32_t* d_grid,
                                const uint32_t* d_slot_index, float2* d_iq, hipStream_t stream)
{
  if (p.wire) {
    constexpr int SPW = OFDM_SYMBOLS_PER_WG_WIRE;
    hipLaunchKernelGGL((ofdm_kernel<N, SPW, true>), dim3((p.nsymb + SPW - 1) / SPW, p.nof_ports, nof_grids), dim3(Plan<N>::T), 0,
                       stream, p, d_grid, d_slot_index, d_iq);
    if (p.wire_stats != nullptr) {
      const uint32_t nof_gp = nof_grids * p.nof_ports;
      hipLaunchKernelGGL(wire_stats_kernel, dim3((nof_gp + 255) / 256), dim3(256), 0, stream, p, d_slot_index, nof_gp,
                         (p.nsymb + SPW - 1) / SPW);
    }
  } else {
    constexpr int SPW = OFDM_SYMBOLS_PER_WG;
    hipLaunchKernelGGL((ofdm_kernel<N, SPW, false>), dim3((p.nsymb + SPW - 1) / SPW, p.nof_ports, nof_grids), dim3(Plan<N>::T), 0,
                       stream, p, d_grid, d_slot_index, d_iq);
  }
  return hipGetLastError();
}

hipError_t launch_ofdm(const OfdmLaunch& p, uint32_t nof_grids, const uint32_t* d_grid, const uint32_t* d_slot_index,
                       float2* d_iq, hipStream_t stream)
{
  if (nof_grids == 0) {
    return hipSuccess;
  }
  switch (p.dft_size) {
    case 6144:
      return launch_ofdm_n<6144>(p, nof_grids, d_grid, d_slot_index, d_iq, stream);
    case 4608:
      return launch_ofdm_n<4608>(p, nof_grids, d_grid, d_slot_index, d_iq, stream);
    case 4096:
      return launch_ofdm_n<4096>(p, nof_grids, d_grid, d_slot_index, d_iq, stream);
    case 3072:
      return launch_ofdm_n<3072>(p, nof_grids, d_grid, d_slot_index, d_iq, stream);
    case 1536:
      return launch_ofdm_n<1536>(p, nof_grids, d_grid, d_slot_index, d_iq, stream);
    case 768:
      return launch_ofdm_n<768>(p, nof_grids, d_grid, d_slot_index, d_iq, stream);
    case 384:
      return launch_ofdm_n<384>(p, nof_grids, d_grid, d_slot_index, d_iq, stream);
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
// OFDM demodulator: the receive-side mirror of ofdm_kernel.  Replaces ofdm_symbol_demodulator_impl::demodulate and
// ofdm_slot_demodulator_impl::demodulate (R/lib/phy/lower/modulation/ofdm_demodulator_impl.cpp:98-171): skip the
// cyclic prefix (minus the window offset), direct DFT, phase compensation x scale (x the window-offset phase ramp),
// top bins -> lower half of the grid, bins from DC -> upper half, stored as cbf16 (round to nearest even, what
// resource_grid_writer::put does with complex floats).
// ================================================================================================================
template <int N>
__global__ __launch_bounds__(Plan<N>::T) void ofdm_demod_kernel(OfdmLaunch p, const float2* __restrict__ d_iq,
                                                                const uint32_t* __restrict__ d_slot_index,
                                                                uint32_t window_offset, uint32_t* __restrict__ d_grid)
{
  __shared__ cf  lds[N + N / 16 + 16];
  const uint32_t tid  = threadIdx.x;
  const uint32_t l    = blockIdx.x;
  const uint32_t gp   = blockIdx.z * p.nof_ports + blockIdx.y; // grid * nof_ports + port
  const uint32_t slot = d_slot_index ? to_constant(d_slot_index)[blockIdx.z] : 0u;
  const uint32_t sym  = slot * p.nsymb + l; // symbol index within the subframe
  const uint32_t cp   = to_constant(p.cp_len)[sym];
  const cf       ph   = make_cf(to_constant(p.phase)[sym].x, to_constant(p.phase)[sym].y);
  const uint32_t half = p.rg_size >> 1;

  // The symbol's N samples after the cyclic prefix, straight into the registers of the first stage.
  const float2* in = d_iq + (size_t)gp * p.slot_stride + to_constant(p.sym_offset)[sym] + (cp - window_offset);
  const __amdgpu_buffer_rsrc_t rsrc_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(in), 0, (int)(N * 8u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(
      d_grid + ((size_t)gp * NRPHY_NSYMB + l) * p.rg_size, 0, (int)(p.rg_size * 4u), 0x00020000);
  const TwiddleBase<N> tb = load_twiddle_base<-1, N>(p.twiddle, tid);
  cf                   cur[Plan<N>::R0];
#pragma unroll
  for (int k = 0; k != Plan<N>::R0; ++k) {
    const u32x2_t raw = __builtin_amdgcn_raw_buffer_load_b64(
        rsrc_in, (int)((tid + (uint32_t)k * (N / Plan<N>::R0)) * 8u), 0, 0);
    cur[k] = make_cf(__uint_as_float(raw.x), __uint_as_float(raw.y));
  }
  auto store = [&](uint32_t q, auto base, auto, cf v) {
    constexpr uint32_t B   = decltype(base)::value;
    const uint32_t     idx = q + B;
    cf                 y   = cmul_uniform(v, ph);
    if (window_offset != 0) { // wave-uniform: exp(+j 2 pi window_offset idx / N) as the reference rounds it (host table)
      const float2 w = p.window_phase[idx];
      y              = cmul(y, make_cf(w.x, w.y));
    }
    // Bin -> subcarrier; bins of the guard band get an offset the range check drops.
    uint32_t k = 0xFFFFFFu;
    if (idx < half) {
      k = idx + half;
    } else if (idx >= N - half) {
      k = idx - (N - half);
    }
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t b = __builtin_convertvector(y, bf16x2_t);
    __builtin_amdgcn_raw_buffer_store_b32(*reinterpret_cast<const uint32_t*>(&b), rsrc_out, (int)(k * 4u), 0, 0);
  };
  fft_from_registers<-1, N>(cur, tb, lds, p.twiddle, tid, store);
}

template <int N>
static hipError_t launch_ofdm_demod_n(const OfdmLaunch& p, uint32_t nof_grids, const float2* d_iq,
                                      const uint32_t* d_slot_index, uint32_t window_offset, uint32_t* d_grid,
                                      hipStream_t stream)
{
  hipLaunchKernelGGL((ofdm_demod_kernel<N>), dim3(p.nsymb, p.nof_ports, nof_grids), dim3(Plan<N>::T), 0, stream, p,
                     d_iq, d_slot_index, window_offset, d_grid);
  return hipGetLastError();
}

hipError_t launch_ofdm_demod(const OfdmLaunch& p, uint32_t nof_grids, const float2* d_iq, const uint32_t* d_slot_index,
                             uint32_t window_offset, uint32_t* d_grid, hipStream_t stream)
{
  if (nof_grids == 0) {
    return hipSuccess;
  }
  switch (p.dft_size) {
    case 6144:
      return launch_ofdm_demod_n<6144>(p, nof_grids, d_iq, d_slot_index, window_offset, d_grid, stream);
    case 4608:
      return launch_ofdm_demod_n<4608>(p, nof_grids, d_iq, d_slot_index, window_offset, d_grid, stream);
    case 4096:
      return launch_ofdm_demod_n<4096>(p, nof_grids, d_iq, d_slot_index, window_offset, d_grid, stream);
    case 3072:
      return launch_ofdm_demod_n<3072>(p, nof_grids, d_iq, d_slot_index, window_offset, d_grid, stream);
    case 2048:
      return launch_ofdm_demod_n<2048>(p, nof_grids, d_iq, d_slot_index, window_offset, d_grid, stream);
    case 1536:
      return launch_ofdm_demod_n<1536>(p, nof_grids, d_iq, d_slot_index, window_offset, d_grid, stream);
    case 1024:
      return launch_ofdm_demod_n<1024>(p, nof_grids, d_iq, d_slot_index, window_offset, d_grid, stream);
    case 768:
      return launch_ofdm_demod_n<768>(p, nof_grids, d_iq, d_slot_index, window_offset, d_grid, stream);
    case 512:
      return launch_ofdm_demod_n<512>(p, nof_grids, d_iq, d_slot_index, window_offset, d_grid, stream);
    case 384:
      return launch_ofdm_demod_n<384>(p, nof_grids, d_iq, d_slot_index, window_offset, d_grid, stream);
    case 256:
      return launch_ofdm_demod_n<256>(p, nof_grids, d_iq, d_slot_index, window_offset, d_grid, stream);
    case 128:
      return launch_ofdm_demod_n<128>(p, nof_grids, d_iq, d_slot_index, window_offset, d_grid, stream);
    default:
      return hipErrorInvalidValue;
  }
}

// ================================================================================================================
// Plain batched DFT (dft_processor): one workgroup per transform.  Output element k of transform t goes to
// d_out[(t / out_group) * out_group * N + (t % out_group) + k * out_group]: out_group = 1 is the plain layout, the
// second pass of a large transform interleaves the out_group = N1 sub-transforms of one transform (see below).
// ================================================================================================================
template <int SIGN, int N>
__global__ __launch_bounds__(Plan<N>::T) void dft_kernel(const float2* __restrict__ tw, const float2* __restrict__ d_in,
                                                         float2* __restrict__ d_out, uint32_t out_group)
{
  __shared__ cf     lds[N + N / 16 + 16];
  const uint32_t    tid = threadIdx.x;
  const uint32_t    big = blockIdx.x / out_group, sub = blockIdx.x - big * out_group;
  const float2*     in  = d_in + (size_t)blockIdx.x * N;
  float2*           out = d_out + (size_t)big * out_group * N + sub;
  const TwiddleBase<N> tb = load_twiddle_base<SIGN, N>(tw, tid);
  cf                a[Plan<N>::R0];
#pragma unroll
  for (int k = 0; k != Plan<N>::R0; ++k) {
    const uint32_t i = first_stage_index<N>(tid, k);
    const float2   v = (i < N) ? in[i] : make_float2(0.f, 0.f);
    a[k]             = make_cf(v.x, v.y);
  }
  auto store = [&](uint32_t q, auto base, auto, cf v) {
    out[(size_t)(q + decltype(base)::value) * out_group] = make_float2(v.x, v.y);
  };
  fft_from_registers<SIGN, N>(a, tb, lds, tw, tid, store);
}

template <int N>
static hipError_t launch_dft_n(int inverse, uint32_t batch, const float2* tw, const float2* d_in, float2* d_out,
                               uint32_t out_group, hipStream_t stream)
{
  if (inverse) {
    hipLaunchKernelGGL((dft_kernel<+1, N>), dim3(batch), dim3(Plan<N>::T), 0, stream, tw, d_in, d_out, out_group);
  } else {
    hipLaunchKernelGGL((dft_kernel<-1, N>), dim3(batch), dim3(Plan<N>::T), 0, stream, tw, d_in, d_out, out_group);
  }
  return hipGetLastError();
}

static hipError_t launch_dft_lds(uint32_t size, int inverse, uint32_t batch, const float2* tw, const float2* d_in,
                                 float2* d_out, uint32_t out_group, hipStream_t stream)
{
  switch (size) {
    case 6144:
      return launch_dft_n<6144>(inverse, batch, tw, d_in, d_out, out_group, stream);
    case 4608:
      return launch_dft_n<4608>(inverse, batch, tw, d_in, d_out, out_group, stream);
    case 4096:
      return launch_dft_n<4096>(inverse, batch, tw, d_in, d_out, out_group, stream);
    case 3072:
      return launch_dft_n<3072>(inverse, batch, tw, d_in, d_out, out_group, stream);
    case 1536:
      return launch_dft_n<1536>(inverse, batch, tw, d_in, d_out, out_group, stream);
    case 768:
      return launch_dft_n<768>(inverse, batch, tw, d_in, d_out, out_group, stream);
    case 384:
      return launch_dft_n<384>(inverse, batch, tw, d_in, d_out, out_group, stream);
    case 2048:
      return launch_dft_n<2048>(inverse, batch, tw, d_in, d_out, out_group, stream);
    case 1024:
      return launch_dft_n<1024>(inverse, batch, tw, d_in, d_out, out_group, stream);
    case 512:
      return launch_dft_n<512>(inverse, batch, tw, d_in, d_out, out_group, stream);
    case 256:
      return launch_dft_n<256>(inverse, batch, tw, d_in, d_out, out_group, stream);
    case 128:
      return launch_dft_n<128>(inverse, batch, tw, d_in, d_out, out_group, stream);
    default:
      return hipErrorInvalidValue;
  }
}

// ---- transforms that do not fit LDS: 9216 ... 49152 (the PRACH sizes of the reference's generic DFT,
// dft_processor_generic_impl.cpp:203-208) -----------------------------------------------------------------------------
// N = N1 * N2 with N1 in {3, 6, 12} and N2 = 3072 or 4096, decimation in time over n = N2 n1 + n2, k = k1 + N1 k2:
//   X[k1 + N1 k2] = sum_{n2} W_N2^(n2 k2) { W_N^(n2 k1) sum_{n1} x[N2 n1 + n2] W_N1^(n1 k1) }.
// Pass 1 (this kernel): one thread per n2 takes the N1 inputs N2 apart (coalesced across the wave), runs the radix-N1
// butterfly in registers, applies W_N^(n2 k1) from the size-N table and writes tmp[k1][n2] (coalesced).  Pass 2: the LDS
// transform of size N2 over every row of tmp, its output k2 stored at k1 + N1 k2 (dft_kernel with out_group = N1).
template <int SIGN, int N1>
__global__ __launch_bounds__(256) void dft_columns_kernel(const float2* __restrict__ tw_n, uint32_t n2_size,
                                                          const float2* __restrict__ d_in, float2* __restrict__ d_tmp)
{
  const uint32_t n2 = blockIdx.x * 256u + threadIdx.x;
  if (n2 >= n2_size) {
    return;
  }
  const size_t  n   = (size_t)N1 * n2_size;
  const float2* in  = d_in + (size_t)blockIdx.y * n;
  float2*       tmp = d_tmp + (size_t)blockIdx.y * n;
  cf            a[N1];
#pragma unroll
  for (int k = 0; k != N1; ++k) {
    const float2 v = in[n2 + (size_t)k * n2_size];
    a[k]           = make_cf(v.x, v.y);
  }
  Butterfly<SIGN, N1>::run(a);
#pragma unroll
  for (int k = 0; k != N1; ++k) {
    cf v = a[k];
    if (k != 0) {
      const float2 w = tw_n[(uint32_t)k * n2]; // k n2 < N: exp(+j 2 pi k n2 / N) from the table, conjugated when direct
      v              = cmul(v, make_cf(w.x, SIGN < 0 ? -w.y : w.y));
    }
    tmp[n2 + (size_t)k * n2_size] = make_float2(v.x, v.y);
  }
}

template <int N1>
static hipError_t launch_dft_columns(int inverse, uint32_t batch, uint32_t n2_size, const float2* tw_n, const float2* d_in,
                                     float2* d_tmp, hipStream_t stream)
{
  const dim3 grid((n2_size + 255u) / 256u, batch);
  if (inverse) {
    hipLaunchKernelGGL((dft_columns_kernel<+1, N1>), grid, dim3(256), 0, stream, tw_n, n2_size, d_in, d_tmp);
  } else {
    hipLaunchKernelGGL((dft_columns_kernel<-1, N1>), grid, dim3(256), 0, stream, tw_n, n2_size, d_in, d_tmp);
  }
  return hipGetLastError();
}

// The split of a size: n1 = 1 for the sizes one workgroup transforms in LDS.
bool dft_split(uint32_t size, uint32_t* n1, uint32_t* n2)
{
  uint32_t a = 1, b = size;
  switch (size) {
    case 128: case 256: case 384: case 512: case 768: case 1024: case 1536: case 2048: case 3072: case 4096: case 4608:
    case 6144:
      break;
    case 9216:
      a = 3, b = 3072;
      break;
    case 12288:
      a = 3, b = 4096;
      break;
    case 18432:
      a = 6, b = 3072;
      break;
    case 24576:
      a = 6, b = 4096;
      break;
    case 36864:
      a = 12, b = 3072;
      break;
    case 49152:
      a = 12, b = 4096;
      break;
    default:
      return false;
  }
  if (n1) {
    *n1 = a;
  }
  if (n2) {
    *n2 = b;
  }
  return true;
}

bool dft_size_supported(uint32_t size)
{
  return dft_split(size, nullptr, nullptr);
}

// tw: the size's own twiddle table; for a split size also tw_n2 (the table of its LDS factor) and d_tmp, `batch`
// transforms of scratch.
hipError_t launch_dft(uint32_t size, int inverse, uint32_t batch, const float2* tw, const float2* tw_n2, float2* d_tmp,
                      const float2* d_in, float2* d_out, hipStream_t stream)
{
  uint32_t n1 = 1, n2 = size;
  if (!dft_split(size, &n1, &n2)) {
    return hipErrorInvalidValue;
  }
  if (batch == 0) {
    return hipSuccess;
  }
  if (n1 == 1) {
    return launch_dft_lds(size, inverse, batch, tw, d_in, d_out, 1, stream);
  }
  if (tw_n2 == nullptr || d_tmp == nullptr) {
    return hipErrorInvalidValue;
  }
  hipError_t e = hipErrorInvalidValue;
  switch (n1) {
    case 3:
      e = launch_dft_columns<3>(inverse, batch, n2, tw, d_in, d_tmp, stream);
      break;
    case 6:
      e = launch_dft_columns<6>(inverse, batch, n2, tw, d_in, d_tmp, stream);
      break;
    case 12:
      e = launch_dft_columns<12>(inverse, batch, n2, tw, d_in, d_tmp, stream);
      break;
    default:
      break;
  }
  if (e != hipSuccess) {
    return e;
  }
  return launch_dft_lds(n2, inverse, batch * n1, tw_n2, d_tmp, d_out, n1, stream);
}

} // namespace nrphy
