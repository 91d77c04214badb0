// The compiler's exact pair of pk_round.hip, register for register, written in assembly -- with and without the global load and
// s_waitcnt in front, with and without a wait state between the two.   two roundings: -16704; fused: -16705 (in the y lane)
#include <hip/hip_runtime.h>
#include <cstdio>
#define PAIR(name, front, between)                                                                                                \
  __global__ void name(const float* in, float scale, float* out)                                                                  \
  {                                                                                                                               \
    float lo, hi;                                                                                                                 \
    asm volatile(front "v_pk_mul_f32 v[6:7], s[4:5], v[2:3] op_sel_hi:[0,1]\n" between                                              \
                 "v_pk_add_f32 v[6:7], v[6:7], s[6:7] op_sel_hi:[1,0]\n s_nop 7\n v_mov_b32 %0, v6\n v_mov_b32 %1, v7"              \
                 : "=v"(lo), "=v"(hi) : "s"(in), "s"(scale) : "s4", "s5", "s6", "s7", "v2", "v3", "v6", "v7", "v8", "memory");       \
    out[0] = lo - 12582912.f;                                                                                                     \
    out[1] = hi - 12582912.f;                                                                                                     \
  }
#define LOADED "s_mov_b32 s4, %3\n s_mov_b32 s5, 0x5040100\n s_mov_b32 s6, 0x4b400000\n v_mov_b32 v8, 0\n global_load_dwordx2 v[2:3], v8, %2\n s_waitcnt vmcnt(0)\n"
#define MOVED "s_mov_b32 s4, %3\n s_mov_b32 s5, 0x5040100\n s_mov_b32 s6, 0x4b400000\n v_mov_b32 v8, 0\n global_load_dwordx2 v[2:3], v8, %2\n s_waitcnt vmcnt(0)\n s_nop 7\n s_nop 7\n"
PAIR(loaded_b2b, LOADED, "")
PAIR(loaded_nop0, LOADED, "s_nop 0\n")
PAIR(loaded_nop1, LOADED, "s_nop 1\n")
PAIR(settled_b2b, MOVED, "")
int main()
{
  const float h[2] = {0.44550502f, -0.8352250456809998f};
  float *d, *o; (void)hipMalloc(&d, 8); (void)hipMalloc(&o, 8);
  (void)hipMemcpy(d, h, 8, hipMemcpyHostToDevice);
  void (*ks[])(const float*, float, float*) = {loaded_b2b, loaded_nop0, loaded_nop1, settled_b2b};
  const char* names[] = {"load, wait, multiply, add", "load, wait, multiply, s_nop 0, add", "load, wait, multiply, s_nop 1, add", "load, wait, 16 idle cycles, multiply, add"};
  for (int i = 0; i != 4; ++i) {
    for (int rep = 0; rep != 3; ++rep) {
      ks[i]<<<1, 64>>>(d, 20000.f, o);
      float r[2]; (void)hipMemcpy(r, o, 8, hipMemcpyDeviceToHost);
      printf("%-44s run %d: (%.0f, %.0f)\n", names[i], rep, r[0], r[1]);
    }
  }
  return 0;
}
