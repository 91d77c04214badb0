// Third follow-up: the forms the COMPILER emitted where ties came out as by a fused multiply-add -- the add writes the register it
// reads (v_pk_add_f32 v[a:b], v[a:b], s[..]) right behind the multiply that produced it.   two roundings: -16704; fused: -16705
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float cf __attribute__((ext_vector_type(2)));
#define INPLACE(name, between, mulsrc, addsrc, mulmod, addmod)                                                           \
  __device__ __noinline__ cf name(cf g, float s, float m, int junk)                                                     \
  {                                                                                                                      \
    cf r = g;                                                                                                            \
    asm volatile("v_pk_mul_f32 %0, %1, %0 " mulmod "\n" between "v_pk_add_f32 %0, %0, %2 " addmod "\n s_nop 7"           \
                 : "+v"(r) : mulsrc(cf{s, __int_as_float(junk)}), addsrc(cf{m, __int_as_float(junk)}));                    \
    return r;                                                                                                            \
  }
INPLACE(sgpr_b2b, "", "s", "s", "op_sel_hi:[0,1]", "op_sel_hi:[1,0]")
INPLACE(sgpr_nop0, "s_nop 0\n", "s", "s", "op_sel_hi:[0,1]", "op_sel_hi:[1,0]")
INPLACE(sgpr_nop3, "s_nop 3\n", "s", "s", "op_sel_hi:[0,1]", "op_sel_hi:[1,0]")
__device__ __noinline__ cf vgpr_b2b(cf g, cf s, cf m)
{
  cf r = g;
  asm volatile("v_pk_mul_f32 %0, %1, %0\n v_pk_add_f32 %0, %0, %2\n s_nop 7" : "+v"(r) : "v"(s), "v"(m));
  return r;
}
__global__ void k(float* out, int junk)
{
  const cf g = {-0.8352250456809998f, -0.8352250456809998f};
  const cf a = sgpr_b2b(g, 20000.f, 12582912.f, junk), b = sgpr_nop0(g, 20000.f, 12582912.f, junk), c = sgpr_nop3(g, 20000.f, 12582912.f, junk),
           d = vgpr_b2b(g, cf{20000.f, 20000.f}, cf{12582912.f, 12582912.f});
  const cf v[] = {a, b, c, d};
  for (int i = 0; i != 4; ++i) {
    out[2 * i] = v[i].x - 12582912.f;
    out[2 * i + 1] = v[i].y - 12582912.f;
  }
}
int main()
{
  float* d; (void)hipMalloc(&d, 64);
  k<<<1, 1>>>(d, 0x5040100);
  float h[8]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"in place, scalar operands, back to back", "in place, scalar operands, s_nop 0", "in place, scalar operands, s_nop 3", "in place, vector operands, back to back"};
  for (int i = 0; i != 4; ++i) printf("%-42s (%.0f, %.0f)\n", names[i], h[2 * i], h[2 * i + 1]);
  return 0;
}
