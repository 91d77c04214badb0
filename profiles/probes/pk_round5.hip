// Does the UNSELECTED half of a scalar-register pair influence a packed FP32 operation?  The compiler broadcasts a uniform float
// to both lanes as  v_pk_add_f32 v[a:b], v[a:b], s[n:n+1] op_sel_hi:[1,0]  and leaves s[n+1] to whatever is there.
// p = -16704.5 (a tie once 1.5 * 2^23 is added): the even result is -16704.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float cf __attribute__((ext_vector_type(2)));
__device__ __noinline__ cf add_with_high(cf p, float m, unsigned high)
{
  cf r;
  asm volatile("s_mov_b32 s6, %2\n s_mov_b32 s7, %3\n s_nop 4\n v_pk_add_f32 %0, %1, s[6:7] op_sel_hi:[1,0]\n s_nop 7" : "=&v"(r) : "v"(p), "s"(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)))), "s"((unsigned)__builtin_amdgcn_readfirstlane((int)high)) : "s6", "s7");
  return r;
}
__device__ __noinline__ cf mul_with_high(cf g, float s, unsigned high)
{
  cf r;
  asm volatile("s_mov_b32 s4, %2\n s_mov_b32 s5, %3\n s_nop 4\n v_pk_mul_f32 %0, s[4:5], %1 op_sel_hi:[0,1]\n s_nop 7" : "=&v"(r) : "v"(g), "s"(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s)))), "s"((unsigned)__builtin_amdgcn_readfirstlane((int)high)) : "s4", "s5");
  return r;
}
__global__ void k(float* out, const unsigned* highs, int n)
{
  const cf p = {-16704.5f, -16704.5f}, g = {-0.8352250456809998f, -0.8352250456809998f};
  for (int i = 0; i != n; ++i) {
    const cf a = add_with_high(p, 12582912.f, highs[i]);
    const cf b = mul_with_high(g, 20000.f, highs[i]);
    const cf c = add_with_high(b, 12582912.f, highs[i]);
    out[6 * i + 0] = a.x - 12582912.f; out[6 * i + 1] = a.y - 12582912.f;
    out[6 * i + 2] = b.x; out[6 * i + 3] = b.y;
    out[6 * i + 4] = c.x - 12582912.f; out[6 * i + 5] = c.y - 12582912.f;
  }
}
int main()
{
  const unsigned highs[] = {0u, 0x05040100u, 0x3F800000u, 0xBF800000u, 0x7FC00000u, 0xFFFFFFFFu, 0x80000001u, 0x00000001u, 0x4B400000u, 0xCB400000u, 0x7F800000u, 0xFF800000u, 0x46828100u, 0xC6828100u};
  const int n = sizeof(highs) / sizeof(highs[0]);
  float* d; unsigned* dh;
  (void)hipMalloc(&d, 6 * n * 4); (void)hipMalloc(&dh, sizeof(highs));
  (void)hipMemcpy(dh, highs, sizeof(highs), hipMemcpyHostToDevice);
  k<<<1, 1>>>(d, dh, n);
  float h[6 * 16]; (void)hipMemcpy(h, d, 6 * n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i != n; ++i)
    printf("high half %08x: add (%.0f, %.0f)   multiply (%.4f, %.4f)   multiply then add (%.0f, %.0f)\n", highs[i], h[6 * i], h[6 * i + 1], h[6 * i + 2], h[6 * i + 3], h[6 * i + 4], h[6 * i + 5]);
  return 0;
}
