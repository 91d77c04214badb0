// NZP-CSI-RS generator for gfx950 (MI355X) ("next" row, SURVEY.md section 8f-2: the other downlink grid writers).
//
// Replaces nzp_csi_rs_generator_impl::map (R/lib/phy/upper/signal_processors/nzp_csi_rs_generator_impl.cpp:96-352;
// RE patterns R/lib/ran/csi_rs/csi_rs_pattern.cpp rows 1-5; generic branch of resource_grid_mapper_impl::map,
// resource_grid_mapper_impl.cpp:150-277) for a batch of signals into device-resident grids.  A signal is tiny (at most
// 3 x 275 resource elements per port): one wavefront per (signal, CDM group) generates the group's Gold sequence into
// LDS (the wave-level generator the DM-RS pilots use), then each lane takes sequence elements m = lane, lane + 64, ...:
// QPSK value, FD-CDM2 sign of the group's second port, the group's layers times the precoding weights (the products
// evaluated as the reference's precoder does), bf16, one store per port.  Every CDM group writes all ports, as the
// reference does (zeros where the weights are zero).
#include "bits_device.h"

namespace nrphy {

__device__ __forceinline__ void csi_cmul_ref(float xr, float xi, float wr, float wi, float& outr, float& outi)
{
  // channel_precoder_avx2.cpp:51-56: fmaddsub(x, w.re, swap(x) * w.im)
  const float t0 = __fmul_rn(xi, wi), t1 = __fmul_rn(xr, wi);
  outr           = __fmaf_rn(xr, wr, -t0);
  outi           = __fmaf_rn(xi, wr, t1);
}

typedef __bf16 csi_bf16x2 __attribute__((ext_vector_type(2)));
typedef float  csi_f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(WAVE) void csi_rs_kernel(CsiRsLaunch p)
{
  __shared__ uint32_t s_seq[CSI_RS_MAX_SEQ_WORDS];
  __shared__ uint32_t s_scratch[CSI_RS_MAX_SEQ_WORDS];
  const uint32_t      lane = threadIdx.x;
  const auto*         wk   = to_constant(&p.work[blockIdx.x]);
  const uint32_t      first_bit = 2u * wk->advance, nbits = 2u * wk->seq_len;
  if (nbits == 0) {
    return;
  }
  const uint32_t nwords = (first_bit + nbits + 31u) >> 5;
  gold_sequence_wave(p.gold, p.x1_words, wk->c_init, nwords, s_seq, s_scratch, lane);
  wave_lds_fence();
  const uint32_t n_re_prb = wk->n_re_prb; // resource elements per occupied PRB
  const float*   w        = p.weights + wk->weights_offset;
  uint32_t*      grid     = p.grid + (size_t)wk->grid_index * p.grid_nof_ports * NRPHY_NSYMB * p.grid_nof_subc +
                   (size_t)wk->symbol * p.grid_nof_subc;
  for (uint32_t m = lane; m < wk->seq_len; m += WAVE) {
    const uint32_t bit = first_bit + 2u * m;
    const uint32_t two = (s_seq[bit >> 5] >> (30u - (bit & 31u))) & 3u; // c(2m') c(2m'+1), MSB first; bit is even
    float          xr = (two & 2u) ? -wk->amplitude : wk->amplitude, xi = (two & 1u) ? -wk->amplitude : wk->amplitude;
    // subcarrier: element m is the (m mod n_re_prb)-th set bit of the PRB's RE mask in PRB rb_begin + stride * (m / n_re_prb)
    const uint32_t i_prb = m / n_re_prb, i_re = m - i_prb * n_re_prb;
    uint32_t       mask  = wk->re_mask;
    for (uint32_t k = 0; k != i_re; ++k) {
      mask &= mask - 1u;
    }
    const uint32_t subc = 12u * (wk->rb_begin + wk->rb_stride * i_prb) + (uint32_t)__builtin_ctz(mask);
    for (uint32_t port = 0; port != wk->nof_ports; ++port) {
      float accr, acci;
      csi_cmul_ref(xr, xi, w[2u * (port * wk->nof_ports + wk->first_layer)],
                   w[2u * (port * wk->nof_ports + wk->first_layer) + 1u], accr, acci);
      if (wk->group_size == 2u) {
        // FD-CDM2 (fd_cdm2_table): the second port of the group is the first with every other element negated
        const float sr = (m & 1u) ? -xr : xr, si = (m & 1u) ? -xi : xi;
        float       pr, pi;
        csi_cmul_ref(sr, si, w[2u * (port * wk->nof_ports + wk->first_layer + 1u)],
                     w[2u * (port * wk->nof_ports + wk->first_layer + 1u) + 1u], pr, pi);
        accr = __fadd_rn(accr, pr);
        acci = __fadd_rn(acci, pi);
      }
      const csi_f32x2  v = {accr, acci};
      const csi_bf16x2 b = __builtin_convertvector(v, csi_bf16x2);
      grid[(size_t)port * NRPHY_NSYMB * p.grid_nof_subc + subc] = *reinterpret_cast<const uint32_t*>(&b);
    }
  }
}

// Sparse host writes into a device grid (nrphy_grid_put): entry i goes to word (port * 14 + symbol) * nof_subc + subc.
// Entries are applied in order: when two name the same resource element the later one must win, so every thread looks
// whether a later entry overwrites its own (the lists are a few hundred entries; duplicates are rare and short-lived).
__global__ __launch_bounds__(256) void grid_put_kernel(const uint32_t* __restrict__ index, const uint32_t* __restrict__ value,
                                                       uint32_t n, uint32_t* __restrict__ grid)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) {
    return;
  }
  const uint32_t mine = index[i];
  for (uint32_t k = i + 1; k < n; ++k) {
    if (index[k] == mine) {
      return; // a later entry owns this resource element
    }
  }
  grid[mine] = value[i];
}

hipError_t launch_grid_put(const uint32_t* d_index, const uint32_t* d_value, uint32_t n, uint32_t* d_grid, hipStream_t stream)
{
  if (n == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(grid_put_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_index, d_value, n, d_grid);
  return hipGetLastError();
}

hipError_t launch_csi_rs(const CsiRsLaunch& p, uint32_t n_work, hipStream_t stream)
{
  if (n_work == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(csi_rs_kernel, dim3(n_work), dim3(WAVE), 0, stream, p);
  return hipGetLastError();
}

} // namespace nrphy
