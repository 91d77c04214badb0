// VALU issue-rate probe for gfx950: how many cycles does a SIMD spend on one wave64 instruction of each kind?
// Every wave runs ITER iterations of 8 independent chains of one instruction; 4 waves per SIMD hide the latency.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 32768;

template <int KIND>
__global__ __launch_bounds__(256) void rate(float* out, float seed)
{
  f2 a[8];
#pragma unroll
  for (int i = 0; i != 8; ++i) {
    a[i] = f2{seed + i + threadIdx.x, seed - i};
  }
  const f2 m = f2{1.0001f, 0.9999f}, c = f2{seed, -seed};
  unsigned long long mask = 0x5555555555555555ull;
  for (int it = 0; it != ITER; ++it) {
#pragma unroll
    for (int i = 0; i != 8; ++i) {
      if (KIND == 0) { // v_fma_f32
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
      } else if (KIND == 1) { // v_pk_fma_f32
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      } else if (KIND == 2) { // v_add_u32 (integer, VOP2)
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
      } else if (KIND == 3) { // v_med3_i32 (VOP3)
        asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
      } else if (KIND == 4) { // v_pk_add_f32
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      } else if (KIND == 5) { // v_pk_mul_f32 with op_sel (the complex-product form)
        asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,1] op_sel_hi:[0,1]" : "+v"(a[i]) : "v"(m));
      } else if (KIND == 6) { // v_cndmask_b32 (reads VCC)
        asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(m.x));
      } else if (KIND == 7) { // v_bfe_u32
        asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(a[i].x));
      } else if (KIND == 8) { // v_pk_fma_f32 with an SGPR operand
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(m), "v"(c));
      } else if (KIND == 9) { // v_cndmask_b32 with an SGPR-pair condition (VOP3)
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "s"(mask));
      } else if (KIND == 10) { // v_cmp writing VCC followed by v_cndmask reading it
        asm volatile("v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(m.x) : "vcc");
      } else if (KIND == 11) { // v_lshl_or_b32
        asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i].x) : "v"(m.x));
      } else if (KIND == 12) { // v_alignbit_b32
        asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i].x) : "v"(m.x));
      } else if (KIND == 13) { // v_mul_lo_u32
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
      } else if (KIND == 14) { // v_mad_u32_u24
        asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
      } else if (KIND == 15) { // v_xor_b32 (VOP2)
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
      } else if (KIND == 16) { // v_max_i32 (VOP2)
        asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
      } else if (KIND == 17) { // v_cvt_pk_bf16_f32
        asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
      } else if (KIND == 18) { // v_bitop3_b32 (xor of three)
        asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
      } else if (KIND == 19) { // v_med3_i32 with two inline constants
        asm volatile("v_med3_i32 %0, %0, -16, 64" : "+v"(a[i].x));
      } else if (KIND == 20) { // v_add3_u32
        asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
      } else if (KIND == 21) { // v_fma_f32 with an SGPR operand
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "s"(m.x), "v"(c.x));
      } else if (KIND == 22) { // v_cmp alone (writes an SGPR pair)
        asm volatile("v_cmp_lt_i32_e64 %1, %0, %2" : "+v"(a[i].x), "=s"(mask) : "v"(m.x));
      } else if (KIND == 23) { // v_and_b32 SDWA byte select
        asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[i].x) : "v"(m.x));
      } else if (KIND == 24) { // v_mov_b32 DPP row_shr
        asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i].x) : "v"(m.x));
      } else if (KIND == 25) { // round 4: the packed 16-bit instructions of the LDPC decoder -- v_pk_add_i16
        asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
      } else if (KIND == 26) { // v_pk_min_i16
        asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
      } else if (KIND == 27) { // v_pk_mad_i16
        asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
      } else if (KIND == 28) { // v_pk_lshlrev_b16
        asm volatile("v_pk_lshlrev_b16 %0, 3, %0" : "+v"(a[i].x));
      } else if (KIND == 29) { // v_perm_b32
        asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
      } else if (KIND == 30) { // v_pk_mul_lo_u16
        asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
      } else if (KIND == 31) { // v_lshrrev_b32 (VOP2)
        asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i].x));
      } else if (KIND == 32) { // v_lshl_add_u32
        asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i].x) : "v"(m.x));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i != 8; ++i) {
    s += a[i].x + a[i].y;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, float* out, int cus, double mhz)
{
  const int blocks = cus * 4; // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  hipEventRecord(a);
  hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  // per SIMD: 4 waves * ITER * 8 instructions
  const double instr = 4.0 * ITER * 8;
  printf("%-28s %8.3f ms  %6.2f cycles per wave64 instruction at %.0f MHz\n", name, ms, ms * 1e-3 * mhz * 1e6 / instr, mhz);
}

int main()
{
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int    cus = prop.multiProcessorCount;
  const double mhz = prop.clockRate / 1000.0;
  float*       out;
  hipMalloc(&out, (size_t)cus * 4 * 256 * 4);
  printf("%s: %d CUs, %.0f MHz\n", prop.name, cus, mhz);
  run<0>("v_fma_f32", out, cus, mhz);
  run<1>("v_pk_fma_f32", out, cus, mhz);
  run<8>("v_pk_fma_f32 (SGPR operand)", out, cus, mhz);
  run<4>("v_pk_add_f32", out, cus, mhz);
  run<5>("v_pk_mul_f32 op_sel", out, cus, mhz);
  run<2>("v_add_u32", out, cus, mhz);
  run<3>("v_med3_i32", out, cus, mhz);
  run<6>("v_cndmask_b32 vcc", out, cus, mhz);
  run<7>("v_bfe_u32", out, cus, mhz);
  run<9>("v_cndmask_b32 sgpr pair", out, cus, mhz);
  run<10>("v_cmp + v_cndmask (2 instr)", out, cus, mhz);
  run<22>("v_cmp -> sgpr pair", out, cus, mhz);
  run<11>("v_lshl_or_b32", out, cus, mhz);
  run<12>("v_alignbit_b32", out, cus, mhz);
  run<13>("v_mul_lo_u32", out, cus, mhz);
  run<14>("v_mad_u32_u24", out, cus, mhz);
  run<15>("v_xor_b32", out, cus, mhz);
  run<16>("v_max_i32", out, cus, mhz);
  run<17>("v_cvt_pk_bf16_f32", out, cus, mhz);
  run<18>("v_bitop3_b32", out, cus, mhz);
  run<19>("v_med3_i32 inline consts", out, cus, mhz);
  run<20>("v_add3_u32", out, cus, mhz);
  run<21>("v_fma_f32 sgpr operand", out, cus, mhz);
  run<23>("v_and_b32 sdwa", out, cus, mhz);
  run<24>("v_mov_b32 dpp row_shr", out, cus, mhz);
  run<25>("v_pk_add_i16", out, cus, mhz);
  run<26>("v_pk_min_i16", out, cus, mhz);
  run<27>("v_pk_mad_i16", out, cus, mhz);
  run<28>("v_pk_lshlrev_b16", out, cus, mhz);
  run<29>("v_perm_b32", out, cus, mhz);
  run<30>("v_pk_mul_lo_u16", out, cus, mhz);
  run<31>("v_lshrrev_b32", out, cus, mhz);
  run<32>("v_lshl_add_u32", out, cus, mhz);
  hipFree(out);
  return 0;
}
