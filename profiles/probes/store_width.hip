// Store-width probe for the resource-grid write pattern of the codeblock kernel: every wavefront writes runs of
// 64 * W bytes to four port planes (183 KB apart), 5 runs per wave, no arithmetic.  W = bytes per lane per store.
//   hipcc --offload-arch=gfx950 -O3 store_width.hip -o store_width
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WORDS>
__global__ __launch_bounds__(64) void grid_store(uint32_t* __restrict__ out, uint32_t plane_words, uint32_t runs)
{
  // block b owns `runs` consecutive runs of 64 * WORDS words in each of the four planes of its slot
  const uint32_t per_slot = 104;
  const uint32_t slot = blockIdx.x / per_slot, cb = blockIdx.x % per_slot;
  uint32_t*      base = out + (size_t)slot * 4 * plane_words + (size_t)cb * runs * 64 * WORDS;
  for (uint32_t r = 0; r != runs; ++r) {
#pragma unroll
    for (int port = 0; port != 4; ++port) {
      uint32_t* p = base + (size_t)port * plane_words + (r * 64 + threadIdx.x) * WORDS;
      if (WORDS == 1) {
        p[0] = r + port;
      } else if (WORDS == 2) {
        *reinterpret_cast<uint2*>(p) = make_uint2(r, port);
      } else {
        *reinterpret_cast<uint4*>(p) = make_uint4(r, port, r, port);
      }
    }
  }
}

template <int WORDS>
static void run(uint32_t* out, uint32_t plane_words, uint32_t slots)
{
  const uint32_t runs = 280 / (64 * WORDS) + 1; // ~279 RE per codeblock
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int rep = 0; rep != 2; ++rep) {
    hipEventRecord(a);
    for (int it = 0; it != 10; ++it) {
      hipLaunchKernelGGL(grid_store<WORDS>, dim3(slots * 104), dim3(64), 0, 0, out, plane_words, runs);
    }
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double bytes = 10.0 * slots * 104 * runs * 64.0 * WORDS * 4 * 4;
    printf("%2d B/lane: %.4f ms per launch, %.1f GB/s (%u runs per wave)\n", 4 * WORDS, ms / 10, bytes / (ms * 1e-3) / 1e9,
           runs);
  }
}

int main()
{
  const uint32_t slots = 1024, plane_words = 14 * 3276;
  uint32_t*      out;
  if (hipMalloc(&out, (size_t)slots * 4 * plane_words * 4) != hipSuccess) {
    return 1;
  }
  run<1>(out, plane_words, slots);
  run<2>(out, plane_words, slots);
  run<4>(out, plane_words, slots);
  return 0;
}
