// Follow-up to pk_round2.hip: packed FP32 multiply -> add with SCALAR-register operands broadcast by op_sel_hi (what the compiler
// emits for "vector * uniform + uniform"), ties in BOTH lanes.   two roundings: -16704; fused: -16705
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float cf __attribute__((ext_vector_type(2)));
__device__ __noinline__ cf both_sgpr(cf g, float s, float m, int junk)
{
  cf p, r;
  asm volatile("v_pk_mul_f32 %0, %3, %2 op_sel_hi:[0,1]\n v_pk_add_f32 %1, %0, %4 op_sel_hi:[1,0]\n s_nop 7" : "=&v"(p), "=&v"(r) : "v"(g), "s"(cf{s, __int_as_float(junk)}), "s"(cf{m, __int_as_float(junk)}));
  return r;
}
__device__ __noinline__ cf mul_sgpr(cf g, float s, cf m, int junk)
{
  cf p, r;
  asm volatile("v_pk_mul_f32 %0, %3, %2 op_sel_hi:[0,1]\n v_pk_add_f32 %1, %0, %4\n s_nop 7" : "=&v"(p), "=&v"(r) : "v"(g), "s"(cf{s, __int_as_float(junk)}), "v"(m));
  return r;
}
__device__ __noinline__ cf add_sgpr(cf g, cf s, float m, int junk)
{
  cf p, r;
  asm volatile("v_pk_mul_f32 %0, %2, %3\n v_pk_add_f32 %1, %0, %4 op_sel_hi:[1,0]\n s_nop 7" : "=&v"(p), "=&v"(r) : "v"(g), "v"(s), "s"(cf{m, __int_as_float(junk)}));
  return r;
}
__device__ __noinline__ cf mul_only_sgpr(cf g, float s, int junk)
{
  cf p;
  asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel_hi:[0,1]\n s_nop 7" : "=&v"(p) : "v"(g), "s"(cf{s, __int_as_float(junk)}));
  return p;
}
__global__ void k(float* out, int junk)
{
  const cf g = {-0.8352250456809998f, -0.8352250456809998f}, s = {20000.f, 20000.f}, m = {12582912.f, 12582912.f};
  const cf a = both_sgpr(g, 20000.f, 12582912.f, junk), b = mul_sgpr(g, 20000.f, m, junk), c = add_sgpr(g, s, 12582912.f, junk), d = mul_only_sgpr(g, 20000.f, junk);
  out[0] = a.x - 12582912.f; out[1] = a.y - 12582912.f; out[2] = b.x - 12582912.f; out[3] = b.y - 12582912.f;
  out[4] = c.x - 12582912.f; out[5] = c.y - 12582912.f; out[6] = d.x; out[7] = d.y;
}
int main()
{
  float* d; (void)hipMalloc(&d, 64);
  k<<<1, 1>>>(d, 0x5040100);
  float h[8]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("both operands scalar        (%.0f, %.0f)\nmultiplier scalar           (%.0f, %.0f)\naddend scalar               (%.0f, %.0f)\nproduct alone, scalar multiplier (%.6f, %.6f)\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  return 0;
}
