// Dependent packed FP32 multiply -> add on MI355X: how far apart must they be for the add to see the ROUNDED product?
// Input chosen so that the float product is an exact tie (x.5): two roundings give the even integer, a fused multiply-add the odd one.
// hipcc --offload-arch=gfx950 -O3 pk_round2.hip -o pk_round2 && ./pk_round2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float cf __attribute__((ext_vector_type(2)));
#define SEQ(name, between)                                                                                              \
  __device__ __noinline__ cf name(cf g, cf s, cf m)                                                                     \
  {                                                                                                                      \
    cf p, r;                                                                                                             \
    asm volatile("v_pk_mul_f32 %0, %2, %3\n" between "v_pk_add_f32 %1, %0, %4\n s_nop 7" : "=&v"(p), "=&v"(r) : "v"(g), "v"(s), "v"(m)); \
    return r;                                                                                                            \
  }
SEQ(gap0, "")
SEQ(gap1, "s_nop 0\n")
SEQ(gap2, "s_nop 1\n")
SEQ(gap3, "s_nop 2\n")
SEQ(gap4, "s_nop 3\n")
SEQ(gap8, "s_nop 7\n")
SEQ(valu1, "v_mov_b32 v40, v41\n")
SEQ(valu2, "v_mov_b32 v40, v41\n v_mov_b32 v42, v41\n")
__device__ __noinline__ cf sgpr_form(cf g, float s, float m)
{
  cf p, r;
  asm volatile("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[1,0]\n v_pk_add_f32 %1, %0, %4 op_sel_hi:[1,0]\n s_nop 7" : "=&v"(p), "=&v"(r) : "v"(g), "s"(cf{s, s}), "s"(cf{m, m}));
  return r;
}
__device__ __noinline__ float scalar_form(float g, float s, float m)
{
  float p, r;
  asm volatile("v_mul_f32 %0, %2, %3\n v_add_f32 %1, %0, %4\n s_nop 7" : "=&v"(p), "=&v"(r) : "v"(g), "v"(s), "v"(m));
  return r;
}
__global__ void k(float* out)
{
  const cf g = {-0.8352250456809998f, 0.0906125009059906f}, s = {20000.f, 40000.f}, m = {12582912.f, 12582912.f};
  const cf r[] = {gap0(g, s, m), gap1(g, s, m), gap2(g, s, m), gap3(g, s, m), gap4(g, s, m), gap8(g, s, m), valu1(g, s, m), valu2(g, s, m),
                  sgpr_form(g, 20000.f, 12582912.f)};
  for (int i = 0; i != 9; ++i) {
    out[2 * i] = r[i].x - 12582912.f;
    out[2 * i + 1] = r[i].y - 12582912.f;
  }
  out[18] = scalar_form(g.x, 20000.f, 12582912.f) - 12582912.f;
}
int main()
{
  float* d; hipMalloc(&d, 128);
  k<<<1, 1>>>(d);
  float h[19]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"back to back", "s_nop 0", "s_nop 1", "s_nop 2", "s_nop 3", "s_nop 7", "one v_mov between", "two v_mov between", "SGPR operands, back to back"};
  printf("two roundings: (-16704, 3624); fused: (-16705, 3625)\n");
  for (int i = 0; i != 9; ++i) printf("%-30s (%.0f, %.0f)\n", names[i], h[2 * i], h[2 * i + 1]);
  printf("%-30s %.0f\n", "v_mul_f32 -> v_add_f32", h[18]);
  return 0;
}
