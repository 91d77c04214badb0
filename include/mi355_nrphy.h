/*
 * mi355_nrphy.h -- C ABI of the MI355X-native 5G NR downlink PHY hot path.
 *
 * This is the drop-in boundary for the PDSCH processor + OFDM modulator of the reference
 * (ushasigh/srsran-edgeric-5g, srsRAN-5G-ER/lib/phy).  Every entry point names the reference
 * interface it replaces (R/ = srsRAN-5G-ER/).  Plain C: POD structs, raw pointers and sizes, int
 * status codes, no exceptions, no C++ or torch types.
 *
 * Conventions
 *  - Bit buffers are MSB-first packed bytes (bit i lives in byte i/8, mask 0x80 >> (i%8)), the layout
 *    of the reference's bit_buffer (R/include/srsran/adt/bit_buffer.h:98-190).
 *  - A resource grid is an array [port][symbol(14)][subcarrier] of cbf16 (two bf16, real then
 *    imaginary, 4 bytes), subcarrier fastest -- the layout of resource_grid_impl
 *    (R/lib/phy/support/resource_grid_impl.cpp:29-57).  Batches add a leading [grid] dimension.
 *  - IQ output is complex float32 (real, imag), [grid][port][sample] with the slot's symbols back
 *    to back (cyclic prefix first), as ofdm_slot_modulator produces it
 *    (R/lib/phy/lower/modulation/ofdm_modulator_impl.cpp:115-139).
 *  - Pointers named d_* are device (HBM) pointers valid on the context's device; `stream` is a
 *    hipStream_t passed as void* (NULL = the context's own stream).  Calls that take a stream are
 *    asynchronous with respect to the host.  The plan-based calls (nrphy_pdsch_run, nrphy_ofdm_run,
 *    nrphy_ofdm_demod_run, nrphy_demodulate_soft, nrphy_llr_descramble, and nrphy_dft_run for the sizes up to 6144) neither
 *    allocate nor touch host memory and can be captured in a hipGraph, any number of them in any order -- a size's twiddle
 *    table is uploaded by the first call that uses it (plan creation, or one call outside the capture); nrphy_dft_run at
 *    the sizes above 6144 allocates its scratch in stream order (hipMallocAsync), which a capture records as memory nodes
 *    of the graph; the others say what they do at the call.  The grid writers
 *    that take host descriptors (nrphy_csi_rs_map, nrphy_pdcch_process, nrphy_ssb_process, nrphy_grid_put) copy them from
 *    host memory when called, like a plan creation: asynchronous, but not for capture (a replay would read host memory the
 *    caller has long released).
 *  - A PDSCH plan owns device scratch that every run rewrites before reading it (sequences, CRC shares): runs of
 *    ONE plan must be ordered (one stream, or events between streams); different plans may run concurrently.
 *    Every run is self-contained -- no state is carried from one run of a plan to the next.
 *  - Host-span entry points (*_host) are blocking and serialised per context (one lock for the whole call);
 *    they may be called from several threads.
 */
#ifndef MI355_NRPHY_H
#define MI355_NRPHY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRPHY_MAX_RB 275
#define NRPHY_NRE 12
#define NRPHY_NSYMB 14
#define NRPHY_MAX_PORTS 4
#define NRPHY_MAX_LAYERS 4
#define NRPHY_PRB_WORDS 5      /* 5 x 64 bits >= 275 PRB */
#define NRPHY_MAX_RESERVED 4   /* re_pattern_list::MAX_RE_PATTERN, R/include/srsran/phy/support/re_pattern.h:142 */
#define NRPHY_MAX_CODEBLOCKS 162 /* MAX_NOF_SEGMENTS, R/include/srsran/ran/sch/sch_constants.h:38 */
#define NRPHY_MAX_PRG ((NRPHY_MAX_RB + 3) / 4) /* precoding_constants::MAX_NOF_PRG: what the reference's precoding_configuration holds */
#define NRPHY_MAX_TB_BYTES (NRPHY_MAX_CODEBLOCKS * 8448 / 8) /* a transport block never has more bits than its codeblocks hold */

/* Status codes.  The reference aborts (srsran_assert) on argument errors; this ABI returns a code. */
enum {
  NRPHY_OK               = 0,
  NRPHY_ERR_INVALID_PDU  = 1, /* pdsch_pdu_validator::is_valid() == false */
  NRPHY_ERR_ARGUMENT     = 2, /* size mismatch, null pointer, unsupported size */
  NRPHY_ERR_DEVICE       = 3, /* HIP runtime error (no GPU, launch failure) */
  NRPHY_ERR_CAPACITY     = 4, /* batch exceeds the plan/context limits */
  NRPHY_ERR_NOT_READY    = 5  /* nrphy_dl_slot_poll: the slot's IQ has not reached the host yet */
};

/* RE pattern: {prb_mask, re_mask, symbols} of R/include/srsran/phy/support/re_pattern.h:40-77.
 * prb_mask bit p = PRB p of the grid (CRB index), re_mask bit k = subcarrier k of the PRB,
 * symbol_mask bit l = OFDM symbol l of the slot. */
typedef struct nrphy_re_pattern {
  uint64_t prb_mask[NRPHY_PRB_WORDS];
  uint16_t re_mask;
  uint16_t symbol_mask;
  uint32_t reserved_;
} nrphy_re_pattern_t;

/* POD mirror of pdsch_processor::pdu_t (R/include/srsran/phy/upper/channel_processors/pdsch_processor.h:58-155). */
typedef struct nrphy_pdsch_pdu {
  uint32_t slot_index;       /* slot_point::slot_index(): slot within the radio frame (DM-RS c_init) */
  uint32_t rnti;
  uint32_t bwp_start_rb;
  uint32_t bwp_size_rb;
  uint32_t cp;               /* 0 = normal (14 symbols per slot), 1 = extended (12; grid rows 12 and 13 stay zero) */
  uint32_t qm;               /* bits per symbol of codeword 0: 2 QPSK, 4 16QAM, 6 64QAM, 8 256QAM */
  uint32_t rv;               /* redundancy version 0..3 */
  uint32_t nof_codewords;    /* must be 1 (validator) */
  uint32_t n_id;
  uint32_t ref_point;        /* 0 = CRB0, 1 = PRB0 */
  uint32_t dmrs_symbol_mask; /* bit l = symbol l carries DM-RS */
  uint32_t dmrs_type;        /* 1 or 2; only 1 is valid */
  uint32_t scrambling_id;
  uint32_t n_scid;
  uint32_t nof_cdm_groups_without_data;
  uint32_t start_symbol_index;
  uint32_t nof_symbols;
  uint32_t ldpc_base_graph;  /* 1 or 2 */
  uint32_t tbs_lbrm_bytes;
  uint32_t vrb_contiguous;   /* rb_allocation::is_contiguous() of the VRB mask (validator rule) */
  uint64_t prb_mask[NRPHY_PRB_WORDS]; /* freq_alloc.get_prb_mask(bwp_start_rb, bwp_size_rb): allocated PRBs, grid-indexed */
  uint32_t nof_reserved;
  uint32_t tb_size_bytes;    /* data[0].size() */
  nrphy_re_pattern_t reserved[NRPHY_MAX_RESERVED];
  float    ratio_pdsch_dmrs_to_sss_dB;
  float    ratio_pdsch_data_to_sss_dB;
  /* precoding_configuration (R/include/srsran/phy/support/precoding_configuration.h) */
  uint32_t nof_layers;
  uint32_t nof_ports;
  uint32_t prg_size_rb;
  uint32_t nof_prg;          /* 1 .. NRPHY_MAX_PRG (prg_size_rb: 1 .. NRPHY_MAX_RB, NRPHY_MAX_RB = wideband) */
  const float* precoding;    /* host pointer: [nof_prg][nof_ports][nof_layers] complex (re, im) */
} nrphy_pdsch_pdu_t;

/* Scalars the reference derives per PDU (pdsch_processor_impl.cpp:75-136, ldpc_segmenter_impl.cpp:90-160,
 * ldpc.h:128-228).  Filled by nrphy_pdsch_derive(); useful to size buffers. */
typedef struct nrphy_pdsch_derived {
  uint32_t nof_re;            /* data RE per layer */
  uint32_t nof_codeblocks;    /* C */
  uint32_t lifting_size;      /* Zc */
  uint32_t segment_length;    /* K = Kb*Zc */
  uint32_t cb_info_bits;      /* K' - L_cb */
  uint32_t nof_filler_bits;   /* F */
  uint32_t nof_tb_crc_bits;   /* 16 or 24 */
  uint32_t nof_cb_crc_bits;   /* 0 or 24 */
  uint32_t zero_pad;          /* zero bits appended to the last CB */
  uint32_t full_length;       /* N = 66*Zc or 50*Zc */
  uint32_t n_ref;             /* Nref */
  uint32_t n_cb;              /* Ncb = min(N, Nref) */
  uint32_t k0;                /* rate-matching start */
  uint32_t nof_short_segments;
  uint32_t rm_length_short;   /* E of the short segments */
  uint32_t rm_length_long;    /* E of the others */
  uint32_t codeword_bits;     /* G */
} nrphy_pdsch_derived_t;

/* ofdm_modulator_configuration, R/include/srsran/phy/lower/modulation/ofdm_modulator.h:34-47. */
typedef struct nrphy_ofdm_config {
  uint32_t numerology;
  uint32_t bw_rb;
  uint32_t dft_size;
  uint32_t cp;        /* 0 normal, 1 extended (12 symbols per slot, each with a cyclic prefix of dft_size / 4) */
  float    scale;
  double   center_freq_hz;
} nrphy_ofdm_config_t;

typedef struct nrphy_ctx nrphy_ctx_t;
typedef struct nrphy_pdsch_plan nrphy_pdsch_plan_t;
typedef struct nrphy_ofdm_plan nrphy_ofdm_plan_t;

/* ---- library ------------------------------------------------------------------------------- */
const char* nrphy_version(void);
const char* nrphy_strerror(int status);
/* Trace ranges: the entry points bracket their work in rocTX ranges named like the reference's trace points -- "process_pdsch"
 * (R/lib/phy/upper/downlink_processor_single_executor_impl.cpp:116-125), "CB batch", "process_dmrs"
 * (R/lib/phy/upper/channel_processors/pdsch_processor_concurrent_impl.cpp:265,317,342,367), "process_pdcch", "process_ssb",
 * "process_nzp_csi_rs", "process_pusch", "cb_decode", "downlink_baseband" -- when the process has the rocTX library loaded
 * (a profiler: rocprofv3 --marker-trace) or NRPHY_TRACE=1 is set; NRPHY_TRACE=0 switches them off.  1 = ranges are live. */
int nrphy_trace_enabled(void);

/* Creates the device context (streams, constant tables).  Replaces the factory chain
 * create_downlink_processor_factory_sw / _hw (R/lib/phy/upper/upper_phy_factories.cpp:659-919).
 * Fails with NRPHY_ERR_DEVICE when no HIP device is present: there is no CPU fallback. */
int nrphy_create(nrphy_ctx_t** ctx, int device_id);
int nrphy_destroy(nrphy_ctx_t* ctx);
int nrphy_synchronize(nrphy_ctx_t* ctx, void* stream);

/* ---- host-only helpers (no device work) ------------------------------------------------------ */
/* pdsch_pdu_validator::is_valid (R/lib/phy/upper/channel_processors/pdsch_processor_validator_impl.cpp:99-181).
 * Returns NRPHY_OK or NRPHY_ERR_INVALID_PDU. */
int nrphy_pdsch_validate(const nrphy_pdsch_pdu_t* pdu);
/* Per-PDU derived scalars (see nrphy_pdsch_derived_t). */
int nrphy_pdsch_derive(const nrphy_pdsch_pdu_t* pdu, nrphy_pdsch_derived_t* out);
/* tbs_calculator_calculate (R/lib/ran/sch/tbs_calculator.cpp:124-144); returns TBS in bits. */
uint32_t nrphy_tbs_calculate(uint32_t nof_symb_sh, uint32_t nof_dmrs_prb, uint32_t nof_oh_prb, uint32_t qm,
                             float target_code_rate, uint32_t nof_layers, uint32_t n_prb);
/* ofdm_symbol_modulator::get_symbol_size (R/lib/phy/lower/modulation/ofdm_modulator_impl.h:68-71),
 * symbol_index counted within the subframe; and ofdm_slot_modulator::get_slot_size. */
uint32_t nrphy_ofdm_symbol_size(const nrphy_ofdm_config_t* cfg, uint32_t symbol_index);
uint32_t nrphy_ofdm_slot_size(const nrphy_ofdm_config_t* cfg, uint32_t slot_index);

/* ---- seam A: pdsch_processor::process, batched ------------------------------------------------
 * Replaces pdsch_processor::process (pdsch_processor.h:167-170; impl pdsch_processor_impl.cpp:30-73,
 * per-codeblock form pdsch_processor_concurrent_impl.cpp:55-338).
 *
 * A plan holds n_pdu PDUs.  PDU i reads its transport block at d_tb + tb_offset[i] and writes grid
 * number grid_index[i] of a batch of grids with nof_ports x 14 x nof_subc cbf16 each.  Creating a plan
 * validates every PDU (NRPHY_ERR_INVALID_PDU names none; use nrphy_pdsch_validate to find it), derives
 * the per-PDU and per-codeblock descriptors and uploads them; it may be run any number of times (runs of one plan
 * ordered with respect to each other, see the conventions above). */
int nrphy_pdsch_plan_create(nrphy_ctx_t* ctx, uint32_t n_pdu, const nrphy_pdsch_pdu_t* pdus,
                            const uint64_t* tb_offset, const uint32_t* grid_index, uint32_t nof_grids,
                            uint32_t grid_nof_ports, uint32_t grid_nof_subc, nrphy_pdsch_plan_t** plan);
int nrphy_pdsch_plan_destroy(nrphy_pdsch_plan_t* plan);
/* Total number of codeblocks / rate-matched codeword bits of the plan, and PDU i's offset (in bits,
 * a multiple of 32) into the codeword tap buffers. */
uint32_t nrphy_pdsch_plan_nof_codeblocks(const nrphy_pdsch_plan_t* plan);
uint64_t nrphy_pdsch_plan_codeword_bits(const nrphy_pdsch_plan_t* plan);
uint64_t nrphy_pdsch_plan_codeword_offset(const nrphy_pdsch_plan_t* plan, uint32_t pdu);

/* Runs the whole PDSCH path of every PDU of the plan: TB CRC, segmentation, CB CRC, LDPC encoding,
 * rate matching, bit interleaving, scrambling, modulation, layer mapping, precoding, RE mapping and
 * DM-RS generation.  d_grid may be NULL (encode only, seam B semantics).  Optional taps, either may be
 * NULL: d_cw_rm receives the rate-matched + interleaved codeword of pdsch_encoder::encode
 * (pdsch_encoder_impl.cpp:28-72) packed MSB-first, d_cw_scrambled the same after scrambling
 * (pdsch_modulator_impl.cpp:30-44).  The caller zeroes the grids (resource_grid::set_all_zero) or
 * passes zero_grids != 0 to have the run clear them first. */
int nrphy_pdsch_run(nrphy_pdsch_plan_t* plan, const uint8_t* d_tb, void* d_grid, uint8_t* d_cw_rm,
                    uint8_t* d_cw_scrambled, int zero_grids, void* stream);

/* Per-kernel device timing for the benchmark (the counterpart of logging_pdsch_processor_decorator,
 * R/lib/phy/upper/channel_processors/channel_processor_factories.cpp:1164-1240, which times process()).
 * After enabling with room for max_runs runs, every nrphy_pdsch_run records HIP events around its kernels on the
 * stream it launches on.  nrphy_pdsch_plan_kernel_times synchronises those events and returns the average
 * duration in milliseconds of {tb_crc, codeblock, dmrs, whole run} over the recorded runs (count in *nof_runs),
 * then resets the recording. */
int nrphy_pdsch_plan_enable_timing(nrphy_pdsch_plan_t* plan, uint32_t max_runs);
int nrphy_pdsch_plan_kernel_times(nrphy_pdsch_plan_t* plan, float avg_ms[4], uint32_t* nof_runs);
/* Only every stride-th run records its events (default 1: every run): an event between two launches costs the stream a few
 * microseconds -- 0.02 ms of a 0.88 ms step of the benchmark with six of them per step. */
int nrphy_pdsch_plan_timing_stride(nrphy_pdsch_plan_t* plan, uint32_t stride);

/* Host-span convenience with the reference's single-PDU semantics: copies the TB in, runs, copies
 * the grid out (blocking).  grid points to nof_ports x 14 x nof_subc cbf16 in host memory and is
 * overwritten only on the REs the PDU maps (like resource_grid_mapper), unless it is NULL.
 * cw_rm / cw_scrambled: optional host taps of codeword_bits bits (packed). */
int nrphy_pdsch_process_host(nrphy_ctx_t* ctx, const nrphy_pdsch_pdu_t* pdu, const uint8_t* tb, void* grid,
                             uint32_t grid_nof_ports, uint32_t grid_nof_subc, uint8_t* cw_rm,
                             uint8_t* cw_scrambled);
/* All PDSCH PDUs of one slot from host spans, blocking: one plan and one launch for the n_pdu transport blocks (tbs[i]:
 * pdus[i].tb_size_bytes bytes), all mapped into the one host grid, which is read first (what other channels wrote stays)
 * and written back.  What a FAPI DL_TTI.request carries for a slot (R/lib/fapi_adaptor/phy/fapi_to_phy_translator.cpp:
 * every dl_pdsch_pdu goes to its own pdsch_processor::process there) in one call; the PDUs' allocations are disjoint, as
 * the scheduler guarantees.  Statuses as nrphy_pdsch_plan_create. */
int nrphy_pdsch_process_slot_host(nrphy_ctx_t* ctx, uint32_t n_pdu, const nrphy_pdsch_pdu_t* pdus, const uint8_t* const* tbs,
                                  void* grid, uint32_t grid_nof_ports, uint32_t grid_nof_subc);

/* Asynchronous host-span form: pdsch_processor::process "may return before completion, the notifier fires from any
 * thread exactly once" (pdsch_processor.h:157-170; the reference's own asynchronous pool:
 * R/lib/phy/upper/channel_processors/pdsch_processor_asynchronous_pool.h:39-143).  A queue keeps up to `depth` PDUs in
 * flight, each on a stream of its own with pinned staging.  Every submit derives the PDU's state anew, like the reference
 * (pdsch_processor_concurrent_impl.cpp:55-207) -- live traffic brings a new pdu_t per slot -- straight into the operation's
 * staging: no device allocation, no blocking copy, one host-to-device copy for the tables and the transport block; only what
 * depends on the SHAPE of the PDU (RE mapping tables, zero-fill lists) is kept from earlier submits.  Submit copies the
 * transport block (the caller's span is free on return) and returns at once;
 * `done(user, status, grid)` runs on a thread of the HIP runtime when the PDU's grid has reached the host: `grid`
 * points at [grid_nof_ports][14][grid_nof_subc] cbf16 (zeros + the PDU's resource elements, DM-RS included), valid until
 * `done` returns -- the handler merges the PDU's RE into the caller's grid and signals its notifier; `status` is
 * NRPHY_ERR_DEVICE when the operation's stream reported an error.  NRPHY_ERR_CAPACITY: `depth` PDUs in flight
 * (nrphy_pdsch_async_wait_slot, or retry).  The handler must not call HIP or this library.
 * Tunable, read when the queue is created: environment variable NRPHY_ASYNC_ZERO_COPY = 1 lets the kernels read the
 * transport block from the operation's pinned staging instead of copying it to the device first, = 3 also lets them write
 * the grid into the pinned buffer `done` receives (pays with many operations in flight). */
typedef struct nrphy_pdsch_async nrphy_pdsch_async_t;
typedef void (*nrphy_pdsch_done_fn)(void* user, int status, const void* grid);
int nrphy_pdsch_async_create(nrphy_ctx_t* ctx, uint32_t depth, uint32_t grid_nof_ports, uint32_t grid_nof_subc,
                             uint32_t max_tb_bytes, nrphy_pdsch_async_t** queue);
int nrphy_pdsch_async_submit(nrphy_pdsch_async_t* queue, const nrphy_pdsch_pdu_t* pdu, const uint8_t* tb,
                             nrphy_pdsch_done_fn done, void* user);
/* All PDSCH PDUs of one slot as ONE operation (one plan, one launch, one grid, one completion): what a FAPI DL_TTI.request
 * carries for a slot.  The transport blocks together (each rounded up to a multiple of 4, plus 4) must fit max_tb_bytes of
 * nrphy_pdsch_async_create; the PDUs' allocations are disjoint.  Statuses as nrphy_pdsch_async_submit. */
int nrphy_pdsch_async_submit_slot(nrphy_pdsch_async_t* queue, uint32_t n_pdu, const nrphy_pdsch_pdu_t* pdus,
                                  const uint8_t* const* tbs, nrphy_pdsch_done_fn done, void* user);
int nrphy_pdsch_async_wait(nrphy_pdsch_async_t* queue);    /* until nothing is in flight */
int nrphy_pdsch_async_wait_slot(nrphy_pdsch_async_t* queue); /* until fewer than `depth` operations are in flight */
int nrphy_pdsch_async_destroy(nrphy_pdsch_async_t* queue); /* waits, then frees */
/* A completion handler that counts: `user` points at a uint64_t incremented atomically per successful PDU. */
void nrphy_pdsch_async_count_done(void* user, int status, const void* grid);

/* ---- seam B: pdsch_encoder::encode / hal::hw_accelerator_pdsch_enc in transport-block mode --------
 * Replaces pdsch_encoder::encode (R/include/srsran/phy/upper/channel_processors/pdsch_encoder.h;
 * impl R/lib/phy/upper/channel_processors/pdsch_encoder_impl.cpp:28-77) and what
 * pdsch_encoder_hw_impl::encode (pdsch_encoder_hw_impl.cpp:34-180) asks of a hardware accelerator:
 * TB CRC + segmentation + CB CRC + LDPC encoding + rate matching + bit interleaving of one transport
 * block, no scrambling.  The fields are pdsch_encoder::configuration (base_graph 1|2, rv, qm = bits per
 * symbol of `mod`, Nref with 0 = unlimited, nof_layers, nof_ch_symbols = RE x layers) + the TB size.
 * codeword_bits (nof_ch_symbols * qm bytes, one bit per byte: the reference's codeword span) and
 * codeword_packed (the same bits MSB-first) may each be NULL.  Host spans, blocking. */
typedef struct nrphy_pdsch_encoder_cfg {
  uint32_t base_graph;
  uint32_t rv;
  uint32_t qm;
  uint32_t nref;
  uint32_t nof_layers;
  uint32_t nof_ch_symbols;
  uint32_t tb_size_bytes;
} nrphy_pdsch_encoder_cfg_t;
int nrphy_pdsch_encode_host(nrphy_ctx_t* ctx, const nrphy_pdsch_encoder_cfg_t* cfg, const uint8_t* tb,
                            uint8_t* codeword_bits, uint8_t* codeword_packed);

/* ---- seam B pieces: ldpc_encoder::encode, batched ----------------------------------------------
 * Replaces ldpc_encoder::encode (R/lib/phy/upper/channel_coding/ldpc/ldpc_encoder_impl.cpp:44-81) for
 * n_cb codeblocks that share (base graph, lifting size).  d_msg: n_cb messages of Kb*Zc bits, each
 * starting on a multiple of msg_stride_bytes; filler bits are zeros.  d_out: n_cb outputs of
 * out_bits bits (the codeblock without its first 2*Zc bits, out_bits <= (N_full-2)*Zc), each starting
 * on a multiple of out_stride_bytes. */
int nrphy_ldpc_encode(nrphy_ctx_t* ctx, uint32_t base_graph, uint32_t lifting_size, uint32_t n_cb,
                      const uint8_t* d_msg, uint32_t msg_stride_bytes, uint32_t out_bits, uint8_t* d_out,
                      uint32_t out_stride_bytes, void* stream);

/* ---- seam C: ofdm_slot_modulator / ofdm_symbol_modulator, batched -------------------------------
 * Replaces ofdm_symbol_modulator::modulate and ofdm_slot_modulator::modulate
 * (R/include/srsran/phy/lower/modulation/ofdm_modulator.h:54-101; impl ofdm_modulator_impl.cpp:56-139).
 * Grid layout as above with nof_subc = 12*bw_rb.  Grid g is modulated as slot slot_index[g] of the
 * subframe (NULL: all slot 0) into d_iq + g * nof_ports * slot_size_max, port after port, where
 * slot_size_max = nrphy_ofdm_plan_slot_stride() (the size of slot 0, the largest). */
int nrphy_ofdm_plan_create(nrphy_ctx_t* ctx, const nrphy_ofdm_config_t* cfg, uint32_t nof_ports,
                           nrphy_ofdm_plan_t** plan);
int nrphy_ofdm_plan_destroy(nrphy_ofdm_plan_t* plan);
uint32_t nrphy_ofdm_plan_slot_stride(const nrphy_ofdm_plan_t* plan);
int nrphy_ofdm_run(nrphy_ofdm_plan_t* plan, uint32_t nof_grids, const void* d_grid, const uint32_t* slot_index,
                   float* d_iq, void* stream);
/* Same for the OFDM kernel: average milliseconds per nrphy_ofdm_run launch. */
int nrphy_ofdm_plan_enable_timing(nrphy_ofdm_plan_t* plan, uint32_t max_runs);
int nrphy_ofdm_plan_kernel_time(nrphy_ofdm_plan_t* plan, float* avg_ms, uint32_t* nof_runs);
int nrphy_ofdm_plan_timing_stride(nrphy_ofdm_plan_t* plan, uint32_t stride); /* as nrphy_pdsch_plan_timing_stride */
/* Host-span single-symbol form of ofdm_symbol_modulator::modulate: grid is one grid in host memory. */
int nrphy_ofdm_modulate_symbol_host(nrphy_ofdm_plan_t* plan, const void* grid, uint32_t port_index,
                                    uint32_t symbol_index, float* output, uint32_t output_size);

/* Host-span whole-slot form of ofdm_slot_modulator::modulate for every port of one grid: iq receives
 * nof_ports x nrphy_ofdm_slot_size(cfg, slot_index) complex samples, port after port (blocking). */
int nrphy_ofdm_modulate_slot_host(nrphy_ofdm_plan_t* plan, const void* grid, uint32_t slot_index, float* iq);

/* ---- device-resident resource grid: sparse writes from the host --------------------------------------
 * The channels this library does not generate (PRS, PT-RS, ...) stay on the CPU; their resource elements
 * -- a few hundred per slot -- are merged into the grid in HBM with one call per slot instead of moving the
 * grid.  Counterpart of resource_grid_writer::put(port, l, k_init, mask, symbols)
 * (R/include/srsran/phy/support/resource_grid_writer.h) for a grid that lives on the device: later entries
 * win over earlier ones, everything else in the grid is left alone. */
typedef struct nrphy_grid_re {
  uint16_t port;
  uint16_t symbol;
  uint32_t subc;
  uint32_t value;  /* cbf16: bf16 real part in the low half, imaginary part in the high half */
} nrphy_grid_re_t;
/* d_grid: ONE grid [nof_ports][14][nof_subc] in device memory; entries: host array, copied at the call into a
 * staging buffer of the call's own (stream-ordered allocation: hipMallocAsync / hipFreeAsync on `stream`).
 * Asynchronous on `stream` afterwards. */
int nrphy_grid_put(nrphy_ctx_t* ctx, void* d_grid, uint32_t nof_ports, uint32_t nof_subc, uint32_t n,
                   const nrphy_grid_re_t* entries, void* stream);

/* ---- soft-bit descrambling ("next" row, SURVEY.md section 8f-1: the step between the demodulation mapper and the
 * UL-SCH decoder) -------------------------------------------------------------------------------------------
 * Replaces pseudo_random_generator::apply_xor(span<log_likelihood_ratio>, span<const log_likelihood_ratio>)
 * (R/include/srsran/phy/upper/sequence_generators/pseudo_random_generator.h; implementation
 * R/lib/phy/upper/sequence_generators/pseudo_random_generator_impl.cpp:423-523) after init(c_init), and the same
 * operation written out in pusch_demodulator_impl (revert_scrambling on the generated sequence,
 * R/lib/phy/upper/channel_processors/pusch/pusch_demodulator_impl.cpp:38-100, 254-259, one OFDM symbol at a time):
 * out[i] = c(i) ? -in[i] : in[i] in 8-bit two's complement (-128 stays -128), c = the Gold sequence of TS 38.211
 * Section 5.2.1 from its first bit.  The UCI placeholder handling of that demodulator is not part of this call.
 * n_cw codewords of `length` soft bits each, row r at d_in + r * in_stride / d_out + r * out_stride (bytes; in place
 * is allowed: d_out == d_in), d_c_init: n_cw values IN DEVICE MEMORY (for PUSCH (rnti << 15) + n_id, TS 38.211
 * Section 6.3.1.1).  length <= 2^21; n_cw <= 65535.  16-byte aligned rows take the wide path.  Asynchronous on
 * `stream` (NULL: the context's stream); no host memory is touched, so the call can be captured in a hipGraph. */
int nrphy_llr_descramble(nrphy_ctx_t* ctx, uint32_t n_cw, const uint32_t* d_c_init, uint32_t length, const int8_t* d_in,
                         size_t in_stride, int8_t* d_out, size_t out_stride, void* stream);
/* One codeword from and to host memory (blocking; for tests and small cases). */
int nrphy_llr_descramble_host(nrphy_ctx_t* ctx, uint32_t c_init, uint32_t length, const int8_t* in, int8_t* out);

/* ---- receive side ("next" row): soft demodulator ("demodulation mapper") -------------------
 * Replaces demodulation_mapper::demodulate_soft (R/include/srsran/phy/upper/channel_modulation/demodulation_mapper.h:
 * 38-62; R/lib/phy/upper/channel_modulation/demodulation_mapper_impl.cpp:33-106 and demodulation_mapper_{qpsk,qam16,
 * qam64,qam256}.cpp): equalised symbols and their noise variances -> 8-bit log-likelihood ratios in [-120, 120],
 * bits-per-symbol values per symbol in the bit order of TS 38.211 Section 5.1.
 * The reference's value depends on the symbol's position in the span it is handed: with AVX2 (the build this library is
 * pinned to) the first floor(n / B) * B symbols (B = 16 QPSK, 8 16-QAM, 16 64-QAM, 4 256-QAM; none for the BPSKs) use
 * 1 / noise_var (0 when the variance is not > 0), floor(v * (1 / width)) for the interval, round-to-nearest-even and
 * blank a COMPONENT with |v| <= 1e-9; the remaining symbols divide by the variance (QPSK, 16-QAM), use floor(v / width),
 * round half away from zero and blank a SYMBOL with |z|^2 < 1e-9.  This call reproduces both, per position, bit for bit;
 * so one call must cover exactly one reference call: nof_spans spans of span_len symbols each, span r at
 * d_symbols + 2 * r * span_len floats (real, imaginary), d_noise_vars + r * span_len, d_llr + r * span_len * Qm
 * (nof_spans <= 65535).
 * A variance that is zero, negative or NaN gives zeros (as in the reference).  Asynchronous on `stream`; no host memory is
 * touched, so the call can be captured in a hipGraph. */
#define NRPHY_MOD_PI2_BPSK 0u
#define NRPHY_MOD_BPSK 1u
#define NRPHY_MOD_QPSK 2u
#define NRPHY_MOD_QAM16 4u
#define NRPHY_MOD_QAM64 6u
#define NRPHY_MOD_QAM256 8u
int nrphy_demodulate_soft(nrphy_ctx_t* ctx, uint32_t modulation, uint32_t nof_spans, uint32_t span_len, const float* d_symbols,
                          const float* d_noise_vars, int8_t* d_llr, void* stream);
/* One span from and to host memory (blocking; for the adaptor, tests and small cases). */
int nrphy_demodulate_soft_host(nrphy_ctx_t* ctx, uint32_t modulation, uint32_t nof_symbols, const float* symbols,
                               const float* noise_vars, int8_t* llr);

/* ---- other downlink grid writers ("next" row): NZP-CSI-RS generator -----------
 * Replaces nzp_csi_rs_generator::map (R/include/srsran/phy/upper/signal_processors/nzp_csi_rs_generator.h:
 * 39-90; impl R/lib/phy/upper/signal_processors/nzp_csi_rs_generator_impl.cpp:96-352 with the RE patterns of
 * R/lib/ran/csi_rs/csi_rs_pattern.cpp and resource_grid_mapper_impl::map, resource_grid_mapper_impl.cpp:
 * 150-277).  The fields are nzp_csi_rs_generator::config_t.  Rows 1 to 5 of TS 38.211 Table 7.4.1.5.3-1 (1, 1,
 * 2, 4, 4 ports: what a grid of NRPHY_MAX_PORTS ports can carry); density and CDM type must be the ones the
 * row allows.  Every CDM group writes its resource elements on ALL ports of the precoding (zeros where the
 * weights are zero), as the reference does. */
typedef struct nrphy_csi_rs_cfg {
  uint32_t slot_index;       /* slot within the frame */
  uint32_t cp;               /* 0 normal, 1 extended */
  uint32_t start_rb;
  uint32_t nof_rb;
  uint32_t row;              /* csi_rs_mapping_table_row, 1..5 */
  uint32_t nof_k_ref;
  uint32_t k_ref[6];         /* freq_allocation_ref_idx */
  uint32_t symbol_l0;
  uint32_t symbol_l1;        /* unused by rows 1..5 */
  uint32_t cdm;              /* csi_rs_cdm_type: 0 no_CDM, 1 fd_CDM2 */
  uint32_t density;          /* csi_rs_freq_density_type: 0 dot5_even_RB, 1 dot5_odd_RB, 2 one, 3 three */
  uint32_t scrambling_id;
  float    amplitude;
  uint32_t nof_ports;        /* ports of the row = ports and layers of the precoding */
  uint32_t prg_size_rb;
  uint32_t nof_prg;          /* must be 1: the reference's generator only handles wideband precoding */
  const float* precoding;    /* [nof_ports][nof_ports] complex: coefficient(layer, port) at [port][layer] */
} nrphy_csi_rs_cfg_t;
/* NRPHY_OK when the configuration is one this library maps (the reference's validator accepts everything). */
int nrphy_csi_rs_validate(const nrphy_csi_rs_cfg_t* cfg);
/* n signals into device grids ([grid][port][14][subc] cbf16): signal i into grid grid_index[i].  The
 * configurations are copied at the call (host pointers, like PDUs at plan creation) into a staging buffer of
 * the call's own (stream-ordered allocation on `stream`); the kernel itself is asynchronous. */
int nrphy_csi_rs_map(nrphy_ctx_t* ctx, uint32_t n, const nrphy_csi_rs_cfg_t* cfgs, const uint32_t* grid_index,
                     void* d_grid, uint32_t grid_nof_ports, uint32_t grid_nof_subc, void* stream);
/* One signal into a host grid [nof_ports][14][nof_subc] cbf16 (read and written; blocking). */
int nrphy_csi_rs_map_host(nrphy_ctx_t* ctx, const nrphy_csi_rs_cfg_t* cfg, void* grid, uint32_t nof_ports,
                          uint32_t nof_subc);

/* ---- other downlink grid writers: PDCCH processor ------------------------------------
 * Replaces pdcch_processor::process (R/include/srsran/phy/upper/channel_processors/pdcch_processor.h:47-151;
 * impl R/lib/phy/upper/channel_processors/pdcch_processor_impl.cpp:66-118) with everything behind it: the
 * CCE-to-REG mapping (R/lib/ran/pdcch/cce_to_prb_mapping.cpp), pdcch_encoder_impl (CRC24C with the RNTI mask, polar
 * interleaver / allocator / encoder / rate matcher: pdcch_encoder_impl.cpp:33-98, channel_coding/polar/), pdcch_modulator_impl
 * (scrambling, QPSK, precoding, mapping: pdcch_modulator_impl.cpp:30-90) and dmrs_pdcch_processor_impl
 * (R/lib/phy/upper/signal_processors/dmrs_pdcch_processor_impl.cpp:32-102).  The fields are pdcch_processor::pdu_t:
 * coreset_description + dci_description.  Precoding has one layer: [nof_prg][nof_ports] complex weights. */
#define NRPHY_PDCCH_MAX_PAYLOAD 128 /* pdcch_constants::MAX_DCI_PAYLOAD_SIZE */
typedef struct nrphy_pdcch_pdu {
  uint32_t slot_index;         /* slot within the radio frame (DM-RS c_init) */
  uint32_t cp;                 /* 0 normal, 1 extended */
  /* coreset_description */
  uint32_t bwp_size_rb;
  uint32_t bwp_start_rb;
  uint32_t start_symbol_index;
  uint32_t duration;           /* 1..3 */
  uint64_t frequency_resources; /* bit i = PRBs [6i, 6i + 6) of the BWP belong to the CORESET (45 bits) */
  uint32_t cce_to_reg_mapping; /* 0 CORESET0, 1 non-interleaved, 2 interleaved */
  uint32_t reg_bundle_size;    /* L, interleaved only */
  uint32_t interleaver_size;   /* R, interleaved only */
  uint32_t shift_index;        /* n_shift (interleaved), physical cell id (CORESET0) */
  /* dci_description */
  uint32_t rnti;
  uint32_t n_id_pdcch_dmrs;
  uint32_t n_id_pdcch_data;
  uint32_t n_rnti;
  uint32_t cce_index;
  uint32_t aggregation_level;  /* 1, 2, 4, 8, 16 */
  float    dmrs_power_offset_dB;
  float    data_power_offset_dB;
  uint32_t payload_size;       /* DCI bits: 12 .. 128 (polar K = payload + 24 in 36 .. 164) */
  uint8_t  payload[NRPHY_PDCCH_MAX_PAYLOAD]; /* one bit per byte, as the reference passes it */
  /* precoding_configuration, one layer */
  uint32_t nof_ports;
  uint32_t prg_size_rb;
  uint32_t nof_prg;
  const float* precoding;      /* host pointer: [nof_prg][nof_ports] complex (re, im) */
} nrphy_pdcch_pdu_t;
/* NRPHY_OK when the PDU is one the reference processes without running into its assertions (its validator accepts
 * everything): duration, aggregation level, payload size, a CCE range inside the CORESET, REG bundles that tile it,
 * PRGs that cover the allocation exactly. */
int nrphy_pdcch_validate(const nrphy_pdcch_pdu_t* pdu);
/* n PDUs into device grids ([grid][port][14][subc] cbf16): PDU i into grid grid_index[i] (NULL: all into grid 0).
 * Descriptors are copied at the call into a staging buffer of the call's own (stream-ordered allocation); the kernel
 * is asynchronous on `stream`.  Only the resource elements of the PDCCH and its DM-RS are written. */
int nrphy_pdcch_process(nrphy_ctx_t* ctx, uint32_t n, const nrphy_pdcch_pdu_t* pdus, const uint32_t* grid_index,
                        void* d_grid, uint32_t grid_nof_ports, uint32_t grid_nof_subc, void* stream);
/* One PDU into a host grid [nof_ports][14][nof_subc] cbf16 (read and written; blocking). */
int nrphy_pdcch_process_host(nrphy_ctx_t* ctx, const nrphy_pdcch_pdu_t* pdu, void* grid, uint32_t nof_ports,
                             uint32_t nof_subc);
/* pdcch_encoder::encode (R/include/srsran/phy/upper/channel_processors/pdcch_encoder.h:33-57) alone: payload bits
 * (one per byte) -> E = rm_length rate-matched bits (one per byte); host spans, blocking. */
int nrphy_pdcch_encode_host(nrphy_ctx_t* ctx, const uint8_t* payload, uint32_t payload_size, uint32_t rnti,
                            uint32_t rm_length, uint8_t* encoded);

/* ---- other downlink grid writers: SS/PBCH block processor -----------------------------
 * Replaces ssb_processor::process (R/include/srsran/phy/upper/channel_processors/ssb_processor.h:35-92; impl
 * R/lib/phy/upper/channel_processors/ssb_processor_impl.cpp:29-107) with pbch_encoder_impl (payload generation,
 * scrambling, CRC24C, polar coding, rate matching: pbch_encoder_impl.cpp:38-186), pbch_modulator_impl
 * (pbch_modulator_impl.cpp:29-109), dmrs_pbch_processor_impl, pss_processor_impl and sss_processor_impl
 * (R/lib/phy/upper/signal_processors/).  The fields are ssb_processor::pdu_t; the slot is given as numerology, system
 * frame number and slot within the frame (slot_point). */
typedef struct nrphy_ssb_pdu {
  uint32_t numerology;       /* of the slot: 0 = 15 kHz ... */
  uint32_t sfn;
  uint32_t slot_index;       /* slot within the radio frame */
  uint32_t phys_cell_id;     /* 0..1007 */
  float    beta_pss_dB;      /* PSS power relative to SSS */
  uint32_t ssb_idx;
  uint32_t L_max;            /* 4, 8 or 64 */
  uint32_t common_scs;       /* subCarrierSpacingCommon as a numerology: 0 = 15 kHz, 1 = 30, 2 = 60, 3 = 120, 4 = 240 */
  uint32_t subcarrier_offset; /* k_SSB */
  uint32_t offset_to_pointA;
  uint32_t pattern_case;     /* 0..4 = case A..E */
  uint8_t  bch_payload[32];  /* one bit per byte: 24 MIB bits (+ 8 the encoder regenerates) */
  uint32_t nof_ports;
  uint8_t  ports[NRPHY_MAX_PORTS]; /* grid ports that carry the block */
} nrphy_ssb_pdu_t;
/* Also refused: a block whose four symbols would run past symbol 13 of the slot (pattern case E, blocks starting at symbol 12). */
int nrphy_ssb_validate(const nrphy_ssb_pdu_t* pdu);
/* n blocks into device grids, as nrphy_pdcch_process. */
int nrphy_ssb_process(nrphy_ctx_t* ctx, uint32_t n, const nrphy_ssb_pdu_t* pdus, const uint32_t* grid_index, void* d_grid,
                      uint32_t grid_nof_ports, uint32_t grid_nof_subc, void* stream);
int nrphy_ssb_process_host(nrphy_ctx_t* ctx, const nrphy_ssb_pdu_t* pdu, void* grid, uint32_t nof_ports,
                           uint32_t nof_subc);
/* pbch_encoder::encode alone: the 864 rate-matched bits (one per byte) of the block's PBCH; host span, blocking. */
int nrphy_pbch_encode_host(nrphy_ctx_t* ctx, const nrphy_ssb_pdu_t* pdu, uint8_t* encoded);

/* ---- receive side ("next" row): LDPC rate dematcher --------------------------
 * Replaces ldpc_rate_dematcher::rate_dematch (R/include/srsran/phy/upper/channel_coding/ldpc/
 * ldpc_rate_dematcher.h; impl R/lib/phy/upper/channel_coding/ldpc/ldpc_rate_dematcher_impl.cpp:43-256):
 * rm_length soft bits as received -> the codeblock's soft buffer of (66 or 50) * Zc int8 LLRs (the
 * codeblock without its first 2*Zc bits), which the LDPC decoder reads.  new_data != 0: first
 * transmission, the buffer is rebuilt (filler bits +infinity = 127, bits not received 0); otherwise the
 * soft bits are added to what the buffer holds (HARQ combining, saturating LLR sum).  The fields are
 * those of codeblock_metadata the reference reads: tb_common {base_graph, lifting_size, rv, mod (as bits
 * per symbol, 1 = BPSK), Nref} and cb_specific.nof_filler_bits, plus the input length.  Results are
 * those of the reference's generic implementation bit for bit, including which soft bits it leaves
 * untouched (so the buffer is an in/out argument in both modes); its AVX2 implementation treats an
 * infinite soft bit like a finite one when combining. */
typedef struct nrphy_ldpc_rate_dematcher_cfg {
  uint32_t base_graph;
  uint32_t lifting_size;
  uint32_t rv;
  uint32_t qm;
  uint32_t nref;
  uint32_t nof_filler_bits;
  uint32_t rm_length;
} nrphy_ldpc_rate_dematcher_cfg_t;
/* n_cb codeblocks that share the configuration: input i at d_in + i * in_stride_bytes, soft buffer i at
 * d_soft + i * soft_stride_bytes.  Asynchronous on `stream`; capturable except for extreme repetition
 * (rm_length of dozens of buffer lengths), where a longer operation list goes through a stream-ordered
 * allocation of the call's own. */
int nrphy_ldpc_rate_dematch(nrphy_ctx_t* ctx, const nrphy_ldpc_rate_dematcher_cfg_t* cfg, uint32_t n_cb,
                            const int8_t* d_in, uint32_t in_stride_bytes, int8_t* d_soft, uint32_t soft_stride_bytes,
                            int new_data, void* stream);
/* Host-span form for one codeblock (blocking); soft_buffer is read and written. */
int nrphy_ldpc_rate_dematch_host(nrphy_ctx_t* ctx, const nrphy_ldpc_rate_dematcher_cfg_t* cfg, const int8_t* in,
                                 int8_t* soft_buffer, int new_data);

/* ---- receive side ("next" row): LDPC decoder ------------------------------
 * Replaces ldpc_decoder::decode (R/include/srsran/phy/upper/channel_coding/ldpc/ldpc_decoder.h;
 * impl R/lib/phy/upper/channel_coding/ldpc/ldpc_decoder_impl.cpp:60-126 with the message kernels of
 * ldpc_decoder_generic.cpp:30-128): layered scaled min-sum on int8 log-likelihood ratios (finite range
 * +-120, +-127 = certain), early stop when the hard bits pass the CRC.  The fields are
 * ldpc_decoder::configuration: codeblock metadata (base graph, lifting size, filler bits, CRC) and
 * algorithm details (max_iterations, scaling_factor in (0, 1)).  crc_poly: 0 = no early stop, 16 =
 * CRC16, 0x24A = CRC24A, 0x24B = CRC24B.  nof_llr soft bits per codeblock: the codeblock without its
 * first 2*Zc (punctured) bits, as the rate dematcher delivers it, between (Kb + 2) * Zc and
 * (N_full - 2) * Zc of them.  Results are bit-identical to the reference's generic implementation; its
 * AVX2 implementation uses other intermediate arithmetic and agrees only once both have converged. */
typedef struct nrphy_ldpc_decoder_cfg {
  uint32_t base_graph;
  uint32_t lifting_size;
  uint32_t nof_filler_bits;
  uint32_t crc_poly;
  uint32_t nof_llr;
  uint32_t max_iterations;
  float    scaling_factor;
} nrphy_ldpc_decoder_cfg_t;
/* n_cb codeblocks that share the configuration.  d_llr: codeblock i at i * llr_stride_bytes.  d_out:
 * the Kb*Zc hard bits of codeblock i, packed MSB first, at i * out_stride_bytes.  d_iterations (may be
 * NULL): per codeblock the iteration after which the CRC passed, 0 when it did not (or no CRC given).
 * d_scratch: nrphy_ldpc_decoder_scratch_bytes() bytes of device memory the CALLER owns for the duration of the call
 * (the decoder's check-to-variable records: a pool of slots shared by the workgroups resident at once, so its size
 * stops growing with the batch -- 0.2 GB at most); calls in flight at the same time need a scratch each.
 * After nrphy_ldpc_decoder_prepare() for the configuration (it uploads the decoder's graph and CRC weights; a call
 * without it does so itself, once, with an allocation and a blocking copy) the call neither allocates nor
 * synchronises and can be captured in a hipGraph. */
int nrphy_ldpc_decoder_scratch_bytes(nrphy_ctx_t* ctx, const nrphy_ldpc_decoder_cfg_t* cfg, uint32_t n_cb, uint64_t* bytes);
int nrphy_ldpc_decoder_prepare(nrphy_ctx_t* ctx, const nrphy_ldpc_decoder_cfg_t* cfg);
int nrphy_ldpc_decode(nrphy_ctx_t* ctx, const nrphy_ldpc_decoder_cfg_t* cfg, uint32_t n_cb, const int8_t* d_llr,
                      uint32_t llr_stride_bytes, uint8_t* d_out, uint32_t out_stride_bytes, uint32_t* d_iterations,
                      void* d_scratch, void* stream);
/* Host-span form for one codeblock (blocking). */
int nrphy_ldpc_decode_host(nrphy_ctx_t* ctx, const nrphy_ldpc_decoder_cfg_t* cfg, const int8_t* llr,
                           uint8_t* message_packed, uint32_t* iterations);

/* One codeblock through rate dematcher and decoder, host spans, one round trip: the operation of the
 * per-codeblock accelerator seam hal::hw_accelerator_pusch_dec::{configure,enqueue,dequeue}_operation +
 * read_operation_outputs (R/include/srsran/hal/phy/upper/channel_processors/pusch/hw_accelerator_pusch_dec.h:
 * 36-110; caller R/lib/phy/upper/channel_processors/pusch/pusch_decoder_hw_impl.cpp:180-345).  llr: the
 * dm->rm_length soft bits of the codeblock; soft_buffer: its HARQ buffer of (66 or 50) * Zc LLRs, read
 * and written; message_packed: Kb * Zc hard bits; *iterations: as nrphy_ldpc_decode.  crc_poly 0 runs
 * max_iterations without a check. */
int nrphy_pusch_decode_codeblock_host(nrphy_ctx_t* ctx, const nrphy_ldpc_rate_dematcher_cfg_t* dm, uint32_t crc_poly,
                                      uint32_t max_iterations, float scaling_factor, const int8_t* llr,
                                      int8_t* soft_buffer, int new_data, uint8_t* message_packed, uint32_t* iterations);

/* ---- receive side ("next" row): PUSCH (UL-SCH) decoder, transport-block level ----
 * Replaces pusch_decoder::new_data ... on_end_softbits (R/include/srsran/phy/upper/channel_processors/pusch/
 * pusch_decoder.h; impl R/lib/phy/upper/channel_processors/pusch/pusch_decoder_impl.cpp:94-497 with
 * ldpc_segmenter_rx and pusch_codeblock_decoder.cpp:28-71) for a batch of transport blocks that share one
 * configuration, everything resident in HBM: segmentation of the codeword LLRs, rate dematching of every
 * codeblock into its HARQ soft buffer, LDPC decoding of the codeblocks whose CRC has not passed yet,
 * concatenation, transport-block CRC.  The fields are pusch_decoder::configuration (base_graph, rv, mod,
 * Nref, nof_layers, nof_ldpc_iterations, use_early_stop, new_data) plus the transport-block size and the
 * number of channel symbols (codeword soft bits / bits per symbol). */
typedef struct nrphy_pusch_decoder_cfg {
  uint32_t base_graph;
  uint32_t qm;
  uint32_t rv;
  uint32_t nof_layers;
  uint32_t nref;            /* limited-buffer size N_ref in bits, 0 = none */
  uint32_t tb_size_bytes;
  uint32_t nof_ch_symbols;  /* G / qm */
  uint32_t max_iterations;
  uint32_t use_early_stop;
  uint32_t new_data;
} nrphy_pusch_decoder_cfg_t;
/* HARQ state the caller keeps per batch between transmissions (sizes from nrphy_pusch_decoder_sizes):
 * d_soft  = n_tb * soft_bytes_per_tb  int8 soft buffers, [tb][codeblock][(66 or 50) * Zc];
 * d_state = n_tb-dependent codeblock state (CRC flags, decoded messages, iteration counts), state_bytes(n_tb).
 * Neither needs initialising before a new_data call.  d_scratch = scratch_bytes of device memory owned by the caller
 * for the duration of one call (the LDPC decoder's records, see nrphy_ldpc_decode); it carries nothing between calls. */
int nrphy_pusch_decoder_sizes(nrphy_ctx_t* ctx, const nrphy_pusch_decoder_cfg_t* cfg, uint32_t n_tb, uint64_t* soft_bytes_per_tb,
                              uint64_t* state_bytes, uint64_t* scratch_bytes, uint32_t* nof_codeblocks);
/* Uploads what the configuration needs once (decoder graph, CRC weights): afterwards nrphy_pusch_decode_batch neither
 * allocates nor synchronises (capturable). */
int nrphy_pusch_decoder_prepare(nrphy_ctx_t* ctx, const nrphy_pusch_decoder_cfg_t* cfg);
/* d_llr: codeword LLRs of transport block i at i * llr_stride_bytes (nof_ch_symbols * qm of them, in the
 * order the demodulator delivers them).  d_tb: transport block i at i * tb_stride_bytes (written whenever
 * all its codeblock CRCs pass; valid when its tb_crc_ok is 1).  d_result: 4 words per transport block --
 * tb_crc_ok, codeblocks whose CRC passed, sum and maximum of the LDPC iterations of the codeblocks decoded
 * in this call (a failed decode counts max_iterations).  Asynchronous on `stream`. */
int nrphy_pusch_decode_batch(nrphy_ctx_t* ctx, const nrphy_pusch_decoder_cfg_t* cfg, uint32_t n_tb, const int8_t* d_llr,
                             uint64_t llr_stride_bytes, int8_t* d_soft, uint8_t* d_state, void* d_scratch, uint8_t* d_tb,
                             uint32_t tb_stride_bytes, uint32_t* d_result, void* stream);

/* ---- receive side of seam C ("next" row): OFDM demodulator ---------------
 * Replaces ofdm_symbol_demodulator::demodulate / ofdm_slot_demodulator::demodulate
 * (R/include/srsran/phy/lower/modulation/ofdm_demodulator.h; impl
 * R/lib/phy/lower/modulation/ofdm_demodulator_impl.cpp:98-171) for every port of nof_grids slots:
 * d_iq [grid][port][slot_stride] complex float (the layout nrphy_ofdm_run writes) -> d_grid
 * [grid][port][14][12*bw_rb] cbf16.  The plan's configuration doubles as ofdm_demodulator_configuration
 * (numerology, bw_rb, dft_size, cp, scale, center_freq_hz); window_offset is its
 * nof_samples_window_offset (must be below the shortest cyclic prefix).  slot_index as in nrphy_ofdm_run. */
int nrphy_ofdm_demod_run(nrphy_ofdm_plan_t* plan, uint32_t nof_grids, const float* d_iq, const uint32_t* slot_index,
                         uint32_t window_offset, void* d_grid, void* stream);
/* Host-span whole-slot form: iq holds nof_ports x nrphy_ofdm_slot_size(cfg, slot_index) complex samples,
 * port after port; grid receives [nof_ports][14][12*bw_rb] cbf16 (blocking). */
int nrphy_ofdm_demodulate_slot_host(nrphy_ofdm_plan_t* plan, const float* iq, uint32_t slot_index,
                                    uint32_t window_offset, void* grid);
/* Host-span form of ofdm_symbol_demodulator::demodulate for one symbol of one port: input = the symbol's
 * cyclic prefix + dft_size samples, symbol_index counted within the subframe; grid_row receives the
 * 12*bw_rb cbf16 values of that OFDM symbol (blocking). */
int nrphy_ofdm_demodulate_symbol_host(nrphy_ofdm_plan_t* plan, const float* input, uint32_t input_size,
                                      uint32_t symbol_index, uint32_t window_offset, void* grid_row);

/* ---- lower-PHY tail: amplitude controller, radio sample format, fronthaul compression ----
 * Amplitude controller: replaces amplitude_controller::process (R/include/srsran/phy/lower/amplitude_controller/
 * amplitude_controller.h:52-66; impl amplitude_controller_clipping_impl.cpp:31-68 and _scaling_impl.cpp:28-37) for
 * n_buffers baseband buffers of nof_samples complex floats (what the lower PHY hands over per port):
 * out = in * 10^(input_gain_dB / 20), then, with clipping enabled, real and imaginary parts limited to
 * +-full_scale_lin * 10^(ceiling_dBFS / 20).  kind 1 is the scaling implementation (gain only, no measurements).
 * The device leaves the raw measurements per buffer in d_stats (may be NULL): sum of |x|^2 and largest |x|^2 after
 * the gain and before clipping, number of clipped real / imaginary parts; nrphy_amplitude_metrics() turns them into
 * amplitude_controller_metrics on the host, carrying the running counters of the reference's object.  The reference
 * skips clipping when the measured power is not a normal number (all-zero, NaN): for such buffers clipping changes
 * nothing unless the ceiling itself is denormal, which is refused. */
typedef struct nrphy_amplitude_cfg {
  uint32_t kind;            /* 0 amplitude_controller_clipping_impl, 1 amplitude_controller_scaling_impl */
  uint32_t enable_clipping;
  float    input_gain_dB;
  float    full_scale_lin;
  float    ceiling_dBFS;
} nrphy_amplitude_cfg_t;
typedef struct nrphy_amplitude_stats { /* device side, one per buffer */
  float    sum_power;
  float    peak_power;
  uint32_t nof_clipped;
  uint32_t nof_samples;
} nrphy_amplitude_stats_t;
typedef struct nrphy_amplitude_metrics { /* amplitude_controller_metrics */
  float    avg_power_fs;
  float    peak_power_fs;
  float    papr_lin;
  float    gain_dB;
  uint64_t nof_processed_samples; /* running totals: pass the same struct to every call for one controller */
  uint64_t nof_clipped_samples;
  double   clipping_probability;
  uint32_t clipping_enabled;
  uint32_t reserved_;
} nrphy_amplitude_metrics_t;
/* Buffer i at d_in + i * in_stride / d_out + i * out_stride (strides in complex samples; in place allowed).
 * Asynchronous on `stream`, capturable (d_stats is cleared by the call itself with a memset node). */
int nrphy_amplitude_control(nrphy_ctx_t* ctx, const nrphy_amplitude_cfg_t* cfg, uint32_t n_buffers, uint32_t nof_samples,
                            const float* d_in, size_t in_stride, float* d_out, size_t out_stride,
                            nrphy_amplitude_stats_t* d_stats, void* stream);
/* Host arithmetic of amplitude_controller_clipping_impl::process on the measurements of ONE buffer; `metrics` is read
 * (running counters) and written. */
int nrphy_amplitude_metrics(const nrphy_amplitude_cfg_t* cfg, const nrphy_amplitude_stats_t* stats,
                            nrphy_amplitude_metrics_t* metrics);
/* One buffer from and to host memory (blocking); metrics may be NULL. */
int nrphy_amplitude_control_host(nrphy_ctx_t* ctx, const nrphy_amplitude_cfg_t* cfg, const float* in, uint32_t nof_samples,
                                 float* out, nrphy_amplitude_metrics_t* metrics);

/* Radio sample format: complex float -> complex int16 (I, Q interleaved), out = round_to_nearest_even(in * scale)
 * saturated to int16 -- srsvec::convert(span<const cf_t>, float, span<int16_t>) (R/lib/srsvec/conversion.cpp:29-65,
 * 323-328), which the radio layers call on every transmit buffer.  The reference's vector loop converts 16 values at
 * a time this way; the last (2 * nof_samples) mod 16 values of a buffer go through std::round (ties away from zero),
 * and so do they here.  Strides in complex samples. */
int nrphy_iq_convert_ci16(nrphy_ctx_t* ctx, uint32_t n_buffers, uint32_t nof_samples, const float* d_in, size_t in_stride,
                          float scale, int16_t* d_out, size_t out_stride, void* stream);
int nrphy_iq_convert_ci16_host(nrphy_ctx_t* ctx, const float* in, uint32_t nof_samples, float scale, int16_t* out);

/* The two above fused into the OFDM modulator's store: nrphy_ofdm_run with the slot leaving the device as complex
 * int16 (4 bytes per sample instead of 8: the IQ write is 71 % of the modulator's traffic).  Every sample takes the
 * path modulator -> amplitude controller (gain, clipping per buffer = per (grid, port) slot) -> conversion, in the
 * reference's order of roundings.  d_iq: [grid][port][slot_stride] complex int16; d_stats: [grid][port] or NULL.
 * Every sample is converted as the reference's vector loop does it (round to nearest even, saturate).  The reference's scalar tail
 * -- the last (2 n mod 16) floats of ONE conversion call round half away from zero and wrap instead of saturating -- has no
 * counterpart here: where it falls depends on how the caller of the reference cuts its buffers, not on the slot
 * (nrphy_iq_convert_ci16 reproduces it per call).
 * With d_stats the plan keeps one 16-byte record per workgroup in a buffer of its own that grows with the largest
 * nof_grids seen (a synchronous reallocation on growth only): run the largest batch once before capturing the call in a
 * hipGraph, and keep the runs of one plan ordered (Conventions). */
typedef struct nrphy_iq_wire_cfg {
  nrphy_amplitude_cfg_t amplitude;
  float                 ci16_scale;
} nrphy_iq_wire_cfg_t;
int nrphy_ofdm_run_ci16(nrphy_ofdm_plan_t* plan, uint32_t nof_grids, const void* d_grid, const uint32_t* slot_index,
                        const nrphy_iq_wire_cfg_t* cfg, int16_t* d_iq, nrphy_amplitude_stats_t* d_stats, void* stream);

/* Open Fronthaul IQ compression of resource-grid PRBs (split 7.2: the grid, not the time-domain signal, leaves the
 * DU).  Replaces iq_compressor::compress (R/include/srsran/ofh/compression/iq_compressor.h; impl
 * R/lib/ofh/compression/iq_compression_none_impl.cpp:31-55 and iq_compression_bfp_impl.cpp:31-98 with quantizer.h and
 * compressed_prb_packer.cpp) and the serialisation of the result in ofh_uplane_message_builder_impl.cpp:137-144: per
 * PRB [udCompParam: the BFP exponent, one byte, BFP only] + 24 samples of data_width bits packed MSB first.  One
 * row = one compress() call = the PRBs of one OFDM symbol of one port; results are those of the reference built for
 * AVX2 (its 16-lane conversion loop rounds to nearest even and saturates, the tail of a call rounds half away). */
typedef struct nrphy_ofh_compression_cfg {
  uint32_t type;        /* 0 none, 1 BFP */
  uint32_t data_width;  /* 1..16 */
  float    iq_scaling;
} nrphy_ofh_compression_cfg_t;
/* Bytes per compressed PRB: 3 * data_width (+ 1 for BFP). */
uint32_t nrphy_ofh_compressed_prb_bytes(const nrphy_ofh_compression_cfg_t* cfg);
/* n_rows rows of nof_prb PRBs: row r reads 12 * nof_prb cbf16 at d_prbs + r * row_stride (in cbf16 words) and writes
 * nof_prb records at d_out + r * out_row_stride bytes.  A batch of whole grids is n_rows = grids * ports * 14,
 * row_stride = 12 * nof_prb.  Asynchronous on `stream`, capturable. */
int nrphy_ofh_compress(nrphy_ctx_t* ctx, const nrphy_ofh_compression_cfg_t* cfg, uint32_t n_rows, uint32_t nof_prb,
                       const void* d_prbs, size_t row_stride, uint8_t* d_out, size_t out_row_stride, void* stream);
int nrphy_ofh_compress_host(nrphy_ctx_t* ctx, const nrphy_ofh_compression_cfg_t* cfg, uint32_t nof_prb, const void* prbs,
                            uint8_t* out);

/* dft_processor::run (R/include/srsran/phy/generic_functions/dft_processor.h:34-73; generic impl
 * dft_processor_generic_impl.cpp:14-218).  Unnormalised DFT of `size` complex floats, `batch` of them
 * back to back.  inverse != 0 uses exp(+j...).  Sizes: every size of the reference's generic implementation
 * (dft_processor_generic_impl.cpp:190-208) -- 128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 4608, 6144
 * (one workgroup per transform, in LDS; these are also the sizes nrphy_ofdm_plan_create accepts) and 9216, 12288,
 * 18432, 24576, 36864, 49152 (the PRACH sizes: a radix-3/6/12 column pass through a scratch copy of the batch,
 * allocated and released in stream order by the call, then LDS transforms of 3072 or 4096 points). */
int nrphy_dft_run(nrphy_ctx_t* ctx, uint32_t size, int inverse, uint32_t batch, const float* d_in, float* d_out,
                  void* stream);
/* Host-span form of dft_processor::run for one transform (blocking). */
int nrphy_dft_run_host(nrphy_ctx_t* ctx, uint32_t size, int inverse, const float* in, float* out);

/* ---- seams A and C on ONE device-resident grid: the downlink slot pipeline ----------------------------------------
 * What the reference does per slot -- the upper PHY's channel processors fill a resource grid
 * (R/lib/phy/upper/downlink_processor_single_executor_impl.cpp:52-215: configure_resource_grid, process_pdcch / _pdsch /
 * _ssb / _nzp_csi_rs, finish_processing_pdus -> send_resource_grid), the grid is handed to the lower PHY
 * (pdxch_processor_request_handler::handle_request, R/lib/phy/lower/processors/downlink/pdxch/pdxch_processor_impl.cpp:
 * 97-112) and the real-time thread asks for it one OFDM symbol at a time (pdxch_processor_baseband::process_symbol,
 * pdxch_processor_impl.cpp:47-95, calling ofdm_symbol_modulator::modulate per port) -- with the grid staying in HBM
 * from the first channel written to the last sample modulated: per slot the transport blocks go down and the IQ comes up,
 * the grid crosses PCIe only if the host asks for it (nrphy_dl_slot_read_grid).
 *
 * A pool owns `depth` slots, each with a stream, a device grid, pinned staging for transport blocks + plan tables and a
 * pinned IQ buffer.  Everything a slot is asked to do is enqueued on its stream in call order; no call allocates device
 * memory or waits for the device unless it says so (the control-channel writers and the sparse put stage their descriptors
 * through a stream-ordered allocation, as their stand-alone forms do).  Calls on ONE slot are serialised by the library (a lock per slot);
 * different slots may be driven from different threads.
 *
 *   nrphy_dl_slot_open      resource_grid::set_all_zero + a free slot: NRPHY_ERR_CAPACITY when all `depth` slots are open
 *   nrphy_dl_slot_pdsch     seam A for n PDUs of the slot (one plan, one launch; may be called more than once per slot --
 *                           pdsch_processor::process arrives PDU by PDU): the transport blocks are copied at the call
 *   nrphy_dl_slot_pdcch / _ssb / _csi_rs / _put    the other grid writers, as nrphy_pdcch_process ... on the slot's grid
 *   nrphy_dl_slot_load_grid a grid computed elsewhere (host span) replaces the slot's grid: seam C alone
 *   nrphy_dl_slot_modulate  seam C, submitted at grid hand-over: every symbol of every port of the slot is modulated
 *                           (nrphy_ofdm_run, or nrphy_ofdm_run_ci16 for a pool created with iq_format 1), ONE
 *                           device-to-host copy brings the IQ into the slot's pinned buffer, then `done(user, status,
 *                           slot_id)` runs on a thread of the HIP runtime (it must not call HIP or this library except
 *                           nrphy_dl_slot_iq / _poll).  Once per open.
 *   nrphy_dl_slot_poll      NRPHY_OK: the IQ is on the host; NRPHY_ERR_NOT_READY: not yet (or no modulate submitted);
 *                           an error code: the slot's stream failed.  An atomic load: what process_symbol calls.
 *   nrphy_dl_slot_wait      blocks until the modulate submitted for the slot has completed, returns its status
 *   nrphy_dl_slot_iq        pinned host samples of `port`: nof_samples = nrphy_ofdm_slot_size(cfg, slot index) complex
 *                           values (float32 pairs, or int16 pairs), symbols back to back, cyclic prefix first; valid from
 *                           completion until the slot is closed.  Pointer arithmetic only.
 *   nrphy_dl_slot_read_grid blocking: everything enqueued so far, then the grid [nof_ports][14][12 * bw_rb] cbf16 to the
 *                           host (resource_grid::get_reader() on a device-mirrored grid)
 *   nrphy_dl_slot_close     gives the slot back; waits for what it still has in flight
 * Statuses: NRPHY_ERR_ARGUMENT for a slot id that is not open or a call out of order (modulate twice, a writer after
 * modulate), NRPHY_ERR_INVALID_PDU as nrphy_pdsch_plan_create, NRPHY_ERR_CAPACITY when the slot's staging cannot take the
 * transport blocks (max_tb_bytes is per slot, all PDSCH calls together). */
typedef struct nrphy_dl_slots nrphy_dl_slots_t;
typedef void (*nrphy_dl_slot_done_fn)(void* user, int status, uint32_t slot_id);
typedef struct nrphy_dl_slots_cfg {
  nrphy_ofdm_config_t ofdm;         /* the lower PHY's modulator configuration (pdxch_processor_factories.cpp:41-48) */
  uint32_t            nof_ports;    /* grid ports = transmit ports */
  uint32_t            depth;        /* slots that can be open at once, 1..64 (the reference's request pool holds 16) */
  uint32_t            max_tb_bytes; /* transport-block bytes per slot, all PDUs together */
  uint32_t            iq_format;    /* 0 = complex float32 (baseband_gateway_buffer), 1 = complex int16 after the amplitude controller */
  nrphy_iq_wire_cfg_t wire;         /* iq_format 1 only */
} nrphy_dl_slots_cfg_t;
int nrphy_dl_slots_create(nrphy_ctx_t* ctx, const nrphy_dl_slots_cfg_t* cfg, nrphy_dl_slots_t** pool);
int nrphy_dl_slots_destroy(nrphy_dl_slots_t* pool); /* waits for everything in flight, then frees */
int nrphy_dl_slots_wait_free(nrphy_dl_slots_t* pool); /* blocks while all `depth` slots are open */
int nrphy_dl_slot_open(nrphy_dl_slots_t* pool, uint32_t* slot_id);
int nrphy_dl_slot_close(nrphy_dl_slots_t* pool, uint32_t slot_id);
int nrphy_dl_slot_pdsch(nrphy_dl_slots_t* pool, uint32_t slot_id, uint32_t n_pdu, const nrphy_pdsch_pdu_t* pdus,
                        const uint8_t* const* tbs);
int nrphy_dl_slot_put(nrphy_dl_slots_t* pool, uint32_t slot_id, uint32_t n, const nrphy_grid_re_t* entries);
int nrphy_dl_slot_load_grid(nrphy_dl_slots_t* pool, uint32_t slot_id, const void* grid);
int nrphy_dl_slot_modulate(nrphy_dl_slots_t* pool, uint32_t slot_id, uint32_t subframe_slot_index, nrphy_dl_slot_done_fn done,
                           void* user);
int nrphy_dl_slot_poll(nrphy_dl_slots_t* pool, uint32_t slot_id);
int nrphy_dl_slot_wait(nrphy_dl_slots_t* pool, uint32_t slot_id);
const void* nrphy_dl_slot_iq(nrphy_dl_slots_t* pool, uint32_t slot_id, uint32_t port, uint32_t* nof_samples);
/* Wire-format pools (iq_format 1): the amplitude controller's raw measurements of the slot's buffer of `port` (sum and peak of
 * |x|^2 after the gain and before clipping, clipped parts, samples) in pinned host memory, valid like nrphy_dl_slot_iq;
 * nrphy_amplitude_metrics() turns them into amplitude_controller_metrics, what downlink_processor_baseband_impl reports per
 * buffer (R/lib/phy/lower/processors/downlink/downlink_processor_baseband_impl.cpp:238-264).  NULL for a float pool. */
const nrphy_amplitude_stats_t* nrphy_dl_slot_amplitude_stats(nrphy_dl_slots_t* pool, uint32_t slot_id, uint32_t port);
int nrphy_dl_slot_read_grid(nrphy_dl_slots_t* pool, uint32_t slot_id, void* grid);
/* For callers that keep more of the chain on the device: the slot's grid in HBM and the stream its work is ordered on. */
void* nrphy_dl_slot_device_grid(nrphy_dl_slots_t* pool, uint32_t slot_id);
void* nrphy_dl_slot_stream(nrphy_dl_slots_t* pool, uint32_t slot_id);
int nrphy_dl_slot_pdcch(nrphy_dl_slots_t* pool, uint32_t slot_id, uint32_t n, const nrphy_pdcch_pdu_t* pdus);
int nrphy_dl_slot_ssb(nrphy_dl_slots_t* pool, uint32_t slot_id, uint32_t n, const nrphy_ssb_pdu_t* pdus);
int nrphy_dl_slot_csi_rs(nrphy_dl_slots_t* pool, uint32_t slot_id, uint32_t n, const nrphy_csi_rs_cfg_t* cfgs);


#ifdef __cplusplus
}
#endif
#endif /* MI355_NRPHY_H */
