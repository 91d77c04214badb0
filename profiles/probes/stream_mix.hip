// Streaming probe with the OFDM kernel's traffic shape and no arithmetic: every 256-thread workgroup reads one
// 3276-word grid row (13104 B) and writes one 4384-sample IQ symbol (35072 B).  Prints the achieved HBM rate: the
// practical ceiling for ofdm_kernel<4096> on this machine.   hipcc --offload-arch=gfx950 -O3 stream_mix.hip -o stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void mix(const uint32_t* __restrict__ in, float2* __restrict__ out, uint32_t row_words,
                                           uint32_t sym_samples)
{
  const uint32_t* row = in + (size_t)blockIdx.x * row_words;
  float2*         o   = out + (size_t)blockIdx.x * sym_samples;
  uint32_t        v[16];
#pragma unroll
  for (int k = 0; k != 16; ++k) {
    uint32_t i = threadIdx.x + 256 * k;
    v[k]       = i < row_words ? row[i] : 0u;
  }
#pragma unroll
  for (int k = 0; k != 16; ++k) {
    uint32_t i = threadIdx.x + 256 * k;
    o[i]       = make_float2(__uint_as_float(v[k] << 16), __uint_as_float(v[k] & 0xFFFF0000u));
  }
  for (uint32_t i = 4096 + threadIdx.x; i < sym_samples; i += 256) {
    o[i] = make_float2(__uint_as_float(v[0]), 0.f);
  }
}

// Variant: 16-byte stores (two samples per lane) and optionally non-temporal stores.
template <bool NT>
__global__ __launch_bounds__(256) void mix16(const uint32_t* __restrict__ in, float4* __restrict__ out,
                                             uint32_t row_words, uint32_t sym_samples)
{
  const uint32_t* row = in + (size_t)blockIdx.x * row_words;
  float4*         o   = out + (size_t)blockIdx.x * (sym_samples / 2);
  uint32_t        v[16];
#pragma unroll
  for (int k = 0; k != 16; ++k) {
    uint32_t i = threadIdx.x + 256 * k;
    v[k]       = i < row_words ? row[i] : 0u;
  }
#pragma unroll
  for (int k = 0; k != 8; ++k) {
    uint32_t i = threadIdx.x + 256 * k;
    float4   y = make_float4(__uint_as_float(v[2 * k] << 16), __uint_as_float(v[2 * k] & 0xFFFF0000u),
                             __uint_as_float(v[2 * k + 1] << 16), __uint_as_float(v[2 * k + 1] & 0xFFFF0000u));
    if (NT) {
      __builtin_nontemporal_store(y.x, &o[i].x);
      __builtin_nontemporal_store(y.y, &o[i].y);
      __builtin_nontemporal_store(y.z, &o[i].z);
      __builtin_nontemporal_store(y.w, &o[i].w);
    } else {
      o[i] = y;
    }
  }
  for (uint32_t i = 2048 + threadIdx.x; i < sym_samples / 2; i += 256) {
    o[i] = make_float4(__uint_as_float(v[0]), 0.f, 0.f, 0.f);
  }
}

int main()
{
  const uint32_t n_sym = 1024 * 14 * 4, row_words = 3276, sym_samples = 4384;
  uint32_t* in;
  float2*   out;
  hipMalloc(&in, (size_t)n_sym * row_words * 4);
  hipMalloc(&out, (size_t)n_sym * sym_samples * 8);
  hipMemset(in, 1, (size_t)n_sym * row_words * 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int rep = 0; rep != 3; ++rep) {
    hipEventRecord(a);
    for (int it = 0; it != 10; ++it) {
      hipLaunchKernelGGL(mix, dim3(n_sym), dim3(256), 0, 0, in, out, row_words, sym_samples);
    }
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double bytes = 10.0 * n_sym * (row_words * 4.0 + sym_samples * 8.0);
    printf("stream_mix: %.4f ms per launch, %.1f GB/s\n", ms / 10, bytes / (ms * 1e-3) / 1e9);
  }
  for (int rep = 0; rep != 3; ++rep) {
    hipEventRecord(a);
    for (int it = 0; it != 10; ++it) {
      hipLaunchKernelGGL(mix16<false>, dim3(n_sym), dim3(256), 0, 0, in, (float4*)out, row_words, sym_samples);
    }
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double bytes = 10.0 * n_sym * (row_words * 4.0 + sym_samples * 8.0);
    printf("stream_mix16: %.4f ms per launch, %.1f GB/s\n", ms / 10, bytes / (ms * 1e-3) / 1e9);
  }
  for (int rep = 0; rep != 3; ++rep) {
    hipEventRecord(a);
    for (int it = 0; it != 10; ++it) {
      hipLaunchKernelGGL(mix16<true>, dim3(n_sym), dim3(256), 0, 0, in, (float4*)out, row_words, sym_samples);
    }
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double bytes = 10.0 * n_sym * (row_words * 4.0 + sym_samples * 8.0);
    printf("stream_mix16 nt: %.4f ms per launch, %.1f GB/s\n", ms / 10, bytes / (ms * 1e-3) / 1e9);
  }
  return 0;
}
