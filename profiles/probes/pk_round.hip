// Does a packed FP32 multiply followed by a packed FP32 add round twice?  (The wire-format sink's plain path relies on it:
// rint(v * scale) as (v * scale) + 1.5 * 2^23.)  hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast pk_round.hip -o pk_round && ./pk_round
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float cf __attribute__((ext_vector_type(2)));
struct Sink {
  float scale;
  __device__ __forceinline__ unsigned plain(cf g) const
  {
#pragma clang fp contract(off)
    const cf r = g * scale + 12582912.f;
    return __builtin_amdgcn_perm(__float_as_uint(r.y), __float_as_uint(r.x), 0x05040100u);
  }
};
__global__ void k(const float* in, float scale, unsigned* out, float* outf)
{
  const cf g = {in[2 * threadIdx.x], in[2 * threadIdx.x + 1]};
  Sink s{scale};
  out[4 * threadIdx.x + 0] = s.plain(g);
  cf p, r;
  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n s_nop 4" : "=v"(p) : "v"(g), "v"(cf{scale, scale}));
  asm volatile("v_pk_add_f32 %0, %1, %2\n s_nop 4" : "=v"(r) : "v"(p), "v"(cf{12582912.f, 12582912.f}));
  out[4 * threadIdx.x + 1] = __builtin_amdgcn_perm(__float_as_uint(r.y), __float_as_uint(r.x), 0x05040100u);
  const float a = __fadd_rn(__fmul_rn(g.x, scale), 12582912.f), b = __fadd_rn(__fmul_rn(g.y, scale), 12582912.f);
  out[4 * threadIdx.x + 2] = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x05040100u);
  out[4 * threadIdx.x + 3] = __builtin_amdgcn_perm(__float_as_uint(fmaf(g.y, scale, 12582912.f)), __float_as_uint(fmaf(g.x, scale, 12582912.f)), 0x05040100u);
  outf[2 * threadIdx.x] = p.x; outf[2 * threadIdx.x + 1] = p.y;
}
int main()
{
  const float h[8] = {0.44550502f, -0.8352250456809998f, 0.0906125009059906f, -0.1524749994277954f, 0.08207499980926514f, 0.25f, -0.0906125009059906f, 0.5f};
  const float scales[2] = {20000.f, 40000.f};
  float* d; unsigned* o; float* of;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64); hipMalloc(&of, 32);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (float sc : scales) {
    k<<<1, 4>>>(d, sc, o, of);
    unsigned r[16]; float p[8];
    hipMemcpy(r, o, 64, hipMemcpyDeviceToHost); hipMemcpy(p, of, 32, hipMemcpyDeviceToHost);
    for (int t = 0; t < 4; ++t)
      printf("scale %.0f in (%.9g, %.9g) product (%.6f, %.6f): C++ contract-off (%d, %d)  pk asm (%d, %d)  scalar rn (%d, %d)  fma (%d, %d)\n", sc, h[2 * t], h[2 * t + 1], p[2 * t], p[2 * t + 1],
             (short)(r[4 * t] & 0xFFFF), (short)(r[4 * t] >> 16), (short)(r[4 * t + 1] & 0xFFFF), (short)(r[4 * t + 1] >> 16), (short)(r[4 * t + 2] & 0xFFFF), (short)(r[4 * t + 2] >> 16),
             (short)(r[4 * t + 3] & 0xFFFF), (short)(r[4 * t + 3] >> 16));
  }
  return 0;
}
