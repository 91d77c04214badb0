// Internal declarations shared by the host side (nrphy_host.cpp) and the HIP kernels of libmi355nrphy.so.
// Not part of the ABI (that is include/mi355_nrphy.h).
#pragma once

#include "mi355_nrphy.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nrphy {

// ---- LDPC base graphs -------------------------------------------------------------------------------------
// One lifted graph per (base graph, lifting size): 2 x 51 graphs, built once per context from the
// 3GPP TS 38.212 Tables 5.3.2-2/-3 edge list (nr_ldpc_bg.inc).  The role of the reference's
// ldpc_graph_impl array (R/lib/phy/upper/channel_coding/ldpc/ldpc_graph_impl.cpp:29-65), laid out for the GPU:
// per check row a contiguous run of packed edges (column * Zc << 16 | shift), the identity column of extension rows
// dropped, plus the three numbers that describe the dual-diagonal core (see ldpc_device.h).
constexpr int NOF_LIFTING_SIZES = 51;
constexpr int NOF_GRAPHS        = 2 * NOF_LIFTING_SIZES;
constexpr int MAX_BG_ROWS       = 46;
constexpr int MAX_BG_EDGES      = 316;

struct LiftedGraph {
  uint16_t row_ptr[MAX_BG_ROWS + 2]; // edges of row m: [row_ptr[m], row_ptr[m+1])
  uint16_t core_b;                   // P^b p0 = sum(aux): the odd shift out of column Kb
  uint16_t core_s0;                  // shift of (row 0, column Kb)
  uint16_t core_s3;                  // shift of (row 3, column Kb)
  uint16_t core_mid;                 // 1: row 1 has the third edge of column Kb (BG1); 2: row 2 (BG2)
  uint32_t edge[MAX_BG_EDGES];       // (column * Zc) << 16 | lifted shift; core rows: systematic columns only
};

// ---- Gold sequence tables -----------------------------------------------------------------------------------
// x2 jump matrices: row r of (M2)^(2^k) as a 31-bit mask, k = 0..GOLD_JUMP_BITS-1 (state bit j = x2(n + j)).
constexpr int GOLD_JUMP_BITS  = 24;
constexpr int GOLD_X1_WORDS   = 1 << 16; // x1(n + 1600) for n < 2^21 bits, MSB-first words

constexpr uint32_t SCR_PARTS = 4; // workgroups sharing the scrambling sequence of one PDU (prologue)

constexpr int CRC_POW_WORDS = 288;     // >= 8448 / 32 + 1

struct GoldTables {
  uint32_t x2_jump[GOLD_JUMP_BITS][32];
  // x2_head[t][w]: mask over the 31-bit state giving bit t of word w of the next 31 words (w < 31): the first 992
  // sequence bits are linear in the state, so 31 lanes produce the 31 seed words of the word recurrence in parallel.
  uint32_t x2_head[32][32];
  uint32_t crc24b_pow32[CRC_POW_WORDS]; // x^(32 m) mod g_CRC24B(x): places a lane's partial CB-CRC
  uint32_t crc24b_table[256];           // byte table of CRC24B: (b x^24) mod g
  uint32_t crc24b_slice[4][256];        // (b x^(8k) x^24) mod g, k = 0..3: a 32-bit word per CRC step (16-byte aligned)
  // crc24b_mul[m][n][v] = (v x^(4n)) x^(32 m) mod g: a lane's partial CB-CRC times x^(32 m), one look-up per nibble
  // of the partial instead of a 24-step shift-and-add multiplication.
  uint32_t crc24b_mul[CRC_POW_WORDS][6][16];
  // Modulation tables (TS 38.211 Section 5.1): qam_lut[Qm/2 - 1][index of Qm bits, first bit in the MSB] = the
  // un-normalised odd-integer symbol of the reference's ci8 table, as floats (re, im).
  float2 qam_lut[4][256];
};

// ---- PDSCH plan ---------------------------------------------------------------------------------------------
constexpr int RE_CHUNK = 512; // data RE handled by one wavefront
// Codeblock CRC of the codeblock kernel: 4 = a 32-bit word per step through four byte tables (4 KB of LDS per wave),
// 3 = 24 bits through three tables + one byte step (3 KB).
#ifndef NRPHY_CRC_SLICES
#define NRPHY_CRC_SLICES 3
#endif
// Scratch region of a codeblock wave during LDPC encoding: doubled systematic blocks (2 * 22 * 12 words), the core rows'
// scratch (80 words), then the graph rows (pdsch_kernels.hip, CbShared).
#define NRPHY_CB_U_GRAPH_OFFSET (2 * 22 * 12 + 80)

enum SymKind : uint32_t { SYM_NONE = 0, SYM_CONTIGUOUS = 1, SYM_TABLE = 2 };

// Everything the device needs to know about one PDU (derived on the host at plan creation).
struct PduDev {
  uint64_t tb_offset;      // byte offset of the transport block in d_tb (multiple of 4)
  uint64_t cw_bit_offset;  // bit offset of the codeword in the tap buffers (multiple of 32)
  uint32_t tb_bytes;
  uint32_t grid_index;
  uint32_t graph;          // index into the lifted graph array
  uint32_t zc;
  uint32_t kb;             // 22 or 10
  uint32_t K;              // Kb * Zc
  uint32_t info_bits;      // K' - L_cb
  uint32_t filler;         // F
  uint32_t tb_crc_bits;    // 16 or 24
  uint32_t cb_crc_bits;    // 0 or 24
  uint32_t zero_pad;
  uint32_t C;
  uint32_t n_short;
  uint32_t e_short;
  uint32_t e_long;
  uint32_t n_cb;
  uint32_t k0;
  uint32_t nof_rows;       // parity rows to compute (>= 4): enough for every bit rate matching reads
  uint32_t qm;
  uint32_t nof_layers;
  uint32_t nof_ports;
  uint32_t c_init;         // scrambling sequence initialisation
  uint32_t item_first;     // index of the PDU's first work item in the plan's (bucket-sorted) work list; its items follow in codeblock / chunk order
  uint32_t scr_words;      // words of the scrambling sequence the prologue walks: ceil(G / 32) + read-ahead + a seed's length
  uint32_t nof_re;
  uint32_t weights_offset; // floats: data weights (scaled) [nof_prg][P][L][2] in the plan's weight array
  uint32_t dmrs_weights_offset; // floats: unscaled weights, same shape
  uint32_t nof_prg;
  uint32_t prg_size_subc;
  // RE mapping: data RE r of OFDM symbol l sits on subcarrier
  //   SYM_CONTIGUOUS: sym_arg[l] + r,   SYM_TABLE: re_table[sym_arg[l] + r]
  uint32_t sym_re_start[NRPHY_NSYMB + 1]; // prefix count of data RE before symbol l
  uint32_t sym_kind[NRPHY_NSYMB];
  uint32_t sym_arg[NRPHY_NSYMB];
  // DM-RS
  uint32_t dmrs_symbol_mask;
  uint32_t dmrs_zero_other_group; // 1: CDM group 1 is reserved but unused by this PDU -> the DM-RS waves write its zeros
  uint32_t dmrs_c_init[NRPHY_NSYMB];
  uint32_t dmrs_ref_rb;
  uint32_t dmrs_seq_offset; // word offset of the DM-RS sequences (one per DM-RS symbol, in symbol order)
  uint32_t dmrs_seq_words;  // words per DM-RS symbol: sequence bits [0, 12 * (end_prb - dmrs_ref_rb))
  float    dmrs_amplitude;
  uint32_t prb_mask[2 * NRPHY_PRB_WORDS];
  uint32_t first_prb;
  uint32_t end_prb;
  uint32_t crc_first;       // the PDU's transport-block CRC shares: tb_crc_part[crc_first .. crc_first + crc_count)
  uint32_t crc_count;
};

// One wavefront of the codeblock kernel: RE [re_begin, re_begin + re_count) of codeblock cb of PDU pdu.
struct CbWork {
  uint32_t pdu;
  uint32_t cb;
  uint32_t re_begin; // first RE of the chunk, counted within the codeblock
  uint32_t re_count;
};

// One workgroup of the DM-RS kernel: OFDM symbol `symbol` of PDU `pdu`.
constexpr int DMRS_PRB_CHUNK = 32; // PRBs per DM-RS wavefront
// Transport-block CRC: one 256-thread workgroup reduces a 16 KiB region of the transport block.
#ifndef NRPHY_CRC_WORDS_PER_THREAD
#define NRPHY_CRC_WORDS_PER_THREAD 16
#endif
constexpr uint32_t TB_CRC_REGION_WORDS = 256 * NRPHY_CRC_WORDS_PER_THREAD;
constexpr uint32_t TB_CRC_REGION_BYTES = 4 * TB_CRC_REGION_WORDS;

// Tables of the transport-block CRC, per polynomial ([0] CRC24A, [1] CRC16).  A workgroup takes a 16 KiB region in four
// rounds of 16-byte loads: thread t owns the words 1024 i + 4 t + j (round i, j = 0..3).  It folds the four words of a round
// with Horner's rule in x^32 (table y32), the four rounds with y1k = x^(32 * 1024); the 256 partials, 128 bits apart, are
// folded by 64 lanes with y8k = x^(128 * 64), and a workgroup that walks several regions steps from one to the next with
// yz = x^(8 * TB_CRC_REGION_BYTES).  y[k][b] = (b x^(8k)) y mod g, so that a 32-bit partial advances with four independent
// look-ups.
struct TbCrcTables {
  uint32_t y32[2][4][256];
  uint32_t y1k[2][4][256];
  uint32_t y8k[2][4][256];
  uint32_t yz[2][4][256];
  uint32_t lane[2][64]; // x^(128 (63 - l)) mod g
};

struct DmrsWork {
  uint32_t pdu;
  uint32_t symbol;
  uint32_t prb_begin;
  uint32_t prb_end;
};

// One 256-thread workgroup of the sequence role: wave 0 walks words [first, first + count) of the PDU's scrambling sequence
// and stores the seeds of the work items that start there; the PDU's first workgroup also generates its DM-RS sequences
// (waves 1-3).  Long sequences are split over up to SCR_PARTS workgroups in a small batch.
struct ScrWork {
  uint32_t pdu;
  uint32_t first;
  uint32_t count;
  uint32_t with_dmrs;
};

// One 256-thread workgroup of the TB-CRC role: regions [region, region + count) (16 KiB each) of the PDU's transport
// block, one after the other with the next region's words requested while the current one is reduced.  A region is
// reduced as if the transport block were zero-extended to the region's end; `factor` = x^(order + 8 (bytes - end of the
// LAST region)) mod g (a negative exponent for the last region of the block: x is invertible mod g) turns the workgroup's
// running remainder into its share of the CRC.
struct CrcWork {
  uint32_t pdu;
  uint32_t region;
  uint32_t factor;
  uint32_t count;
};
constexpr uint32_t TB_CRC_MAX_REGIONS_PER_WORK = 8;
constexpr uint32_t TB_CRC_TARGET_WORK          = 1024; // workgroups of the TB-CRC role a big batch aims at (4 per compute unit)

// Grid words no PDU of the plan maps must read as zero (resource_grid::set_all_zero in the reference).  Instead of
// clearing whole grids and overwriting most of them, the plan lists the uncovered runs and a few waves of the
// codeblock launch write zeros there: every grid word is written exactly once per run.
struct ZeroSeg {
  uint16_t symbol;
  uint16_t k0;
  uint16_t count;
  uint16_t pad_;
};
constexpr uint32_t ZERO_LONG_RUN = 32; // runs at least this long are cleared by the whole wave, shorter ones by one lane
struct ZeroWork {
  uint32_t grid;
  uint32_t port;
  uint32_t seg_begin;
  uint32_t seg_count; // segments [seg_begin, seg_begin + seg_count): the first seg_long of them are long runs
  uint32_t seg_long;
};

struct PdschLaunch {
  const ZeroWork*    zero_work;
  const ZeroSeg*     zero_segs;
  uint32_t           n_zero_work;     // zero-fill waves appended to the codeblock launch (0: caller cleared the grids)
  uint32_t           n_dmrs_in_launch; // DM-RS waves appended to the codeblock launch (0: separate launch)
  uint32_t*          scr;             // DM-RS sequences c(n) of every PDU, MSB-first words (prologue -> DM-RS waves)
  uint32_t*          scr_seed;        // [n_work][32]: the first 31 words of the x2 part of the scrambling sequence of every work item
                                      // (prologue -> codeblock waves, which expand them: gold_expand_seed_wave)
  const PduDev*      pdus;
  const CbWork*      work;
  const DmrsWork*    dmrs_work;
  const CrcWork*     crc_work;
  const ScrWork*     scr_work;
  uint32_t           n_scr_work;
  const TbCrcTables* tbcrc;
  uint32_t           n_crc_work;
  const float*       weights;
  const uint16_t*    re_table;
  const LiftedGraph* graphs;
  const GoldTables*  gold;
  const uint32_t*    x1_words;
  uint32_t*          tb_crc_part; // [n_crc_work] share of every 16 KiB region in its PDU's CRC, rewritten by every run
  uint32_t           n_pdu;
  uint32_t           n_work;
  uint32_t           work_base;      // index of work[0] in the plan's whole work list (a bucket launch starts inside it): the seeds' index
  uint32_t           n_dmrs_work;
  uint32_t           grid_nof_ports;
  uint32_t           grid_nof_subc;
  uint32_t           lds_lin_words;  // dynamic LDS carve of the codeblock kernel (words, multiples of 4): the codeblock ...
  uint32_t           lds_u_words;    // ... and the scratch region its stages share (pdsch_kernels.hip, CbShared)
  uint32_t           profile_stage; // 0 = run everything; n > 0 = codeblock waves stop after stage n (NRPHY_PROFILE_STAGE)
  uint32_t           extras_nt;     // 1: the stores of the DM-RS / zero-fill waves are non-temporal (NRPHY_EXTRAS_NT)
  uint32_t           prologue_order; // 0: sequence workgroups first; 1: spread among the CRC workgroups (NRPHY_PROLOGUE_ORDER)
  // 1: this run clears what it does not map (zero_grids): the DM-RS waves then also clear the resource elements of a CDM group
  // that is reserved (no data) but carries no pilots of the PDU; 0: like the reference's mapper, they leave those alone --
  // whatever another writer of the slot put there stays.
  uint32_t           zero_fill;
};

// Kernel launchers (defined in the .hip files).
hipError_t launch_prologue(const PdschLaunch& p, const uint8_t* d_tb, hipStream_t stream);
// Work items are sorted by bucket = (modulation order, layers): one output-stage specialisation each.
constexpr uint32_t CB_BUCKETS        = 16;
constexpr uint32_t CB_MIXED_MAX_WORK = 4096; // a mixed plan below this many work items takes the one-launch mixed kernel
inline uint32_t cb_bucket(uint32_t qm, uint32_t nof_layers)
{
  return (qm / 2u - 1u) * 4u + (nof_layers - 1u);
}
bool       codeblocks_take_bucket_launches(const PdschLaunch& p, const uint32_t* bucket_begin, int dispatch, uint32_t* nof_buckets);
hipError_t launch_codeblocks(const PdschLaunch& p, const uint32_t* bucket_begin, int dispatch, const uint8_t* d_tb,
                             uint32_t* d_grid, uint32_t* d_cw_rm, uint32_t* d_cw_scr, const hipStream_t* streams, uint32_t n_streams);
hipError_t launch_dmrs(const PdschLaunch& p, uint32_t* d_grid, hipStream_t stream);
hipError_t launch_ldpc_encode(const LiftedGraph* graphs, uint32_t graph, uint32_t kb, uint32_t zc, uint32_t n_cb,
                              const uint8_t* d_msg, uint32_t msg_stride, uint32_t out_bits, uint8_t* d_out,
                              uint32_t out_stride, hipStream_t stream);

// ---- LDPC decoder ("next" row, receive side) -------------------------------------------------------------------
// The whole lifted graph of one (base graph, lifting size) for the decoder: every edge of every check row in
// adjacency order (ascending variable index, the order that breaks ties between equal minima).
struct DecoderGraph {
  uint32_t row_ptr[MAX_BG_ROWS + 2]; // 32-bit: read with scalar loads (a 16-bit element goes through a vector load and a round trip to L2)
  uint32_t pair_ptr[MAX_BG_ROWS + 2]; // rows of two edges before row m: sum of ceil(degree / 2) (messages per edge in LDS)
  uint32_t quad_ptr[MAX_BG_ROWS + 2]; // rows of four edges before row m: sum of ceil(degree / 4) (the table of soft-bit addresses)
  uint32_t edge[MAX_BG_EDGES]; // (variable node * Zc) << 16 | lifted shift
};

constexpr uint32_t DEC_CRC_TABLE_WORDS = 8 * 16; // early-stop CRC: eight nibble tables per message word

struct LdpcDecodeLaunch {
  const DecoderGraph* graph;
  const uint32_t*     pair_addr;  // even lifting sizes: soft-bit addresses of (edge, pair of checks), [row of four edges][lane][4]; else null
  const int8_t*       llr;        // per codeblock: nof_llr soft bits (the codeblock without its first 2 Zc bits)
  uint2*              scratch;    // nof_slots slots of slot_bytes each: a codeblock's check records (8 bytes per check and layer) or its messages per edge
  uint32_t*           slot_flags;  // one word per slot (0 = free), all clear at launch; unused when every codeblock has a slot of its own
  uint32_t            nof_slots;  // >= the workgroups of this kernel the device can hold at once, <= codeblocks
  uint32_t            slot_bytes; // of one slot
  uint8_t*            out;        // per codeblock: Kb * Zc hard bits, packed MSB first
  uint32_t*           iterations; // per codeblock: iterations until the CRC passed, 0 = it did not (may be null)
  const uint32_t*     crc_weight; // per 32-bit word of the message DEC_CRC_TABLE_WORDS words: [nibble k][value v] = (v x^(4k)) x^(bits after the word) mod the CRC polynomial
  const uint8_t*      skip;       // per codeblock (may be null): non-zero = leave it alone (decoded earlier)
  uint8_t*            ok_flags;   // per codeblock (may be null): set to 1 when the CRC passed
  uint32_t            crc_at_end; // 1: check the CRC once, after max_iterations (no early stop)
  uint32_t            zc, bg_k, nof_nodes, nof_layers_max;
  uint32_t            nof_llr, llr_stride, out_stride, nof_filler;
  uint32_t            crc_poly, crc_order; // order 0: no early stop
  uint32_t            max_iterations;
  float               scaling_factor;
  // Two checks per lane with the messages per edge in LDS: in = the LDS bytes that takes for the layers the caller expects
  // (0: not wanted); launch_ldpc_decode() turns it into the launch's LDS size, or 0 when the kernel keeps to its records.
  // Each codeblock decides by the layers its own soft bits ask for whether it fits.
  uint32_t            lm_lds_bytes;
  uint32_t            lds_tail_off;     // set by launch_ldpc_decode(): where the kernel's flags and scaling table sit in its LDS
  // How the message kernels scale a minimum m = 0 .. 120 to round(m * scaling_factor): 2 = (m * scale_fixed + 256) >> 9 in
  // 16-bit arithmetic, 1 = (unsigned)(m * scaling_factor + 0.5f), 0 = the table in LDS -- the cheapest the host has verified.
  uint32_t            scale_arithmetic;
  uint32_t            scale_fixed;
  // The context's A/B knobs (Tunables): -1 = not set.  pairs 0: one check per lane; msg 0: check records instead of messages;
  // ldsmsg 0: messages never in LDS, 2: wherever a workgroup's LDS can hold them.
  int                 knob_pairs, knob_msg, knob_ldsmsg;
};
hipError_t launch_ldpc_decode(const LdpcDecodeLaunch& p, uint32_t n_cb, hipStream_t stream);

// ---- LDPC rate dematcher ("next" row, receive side) ---------------------------------------------------------------
enum DematchKind : uint32_t { DEMATCH_ZERO = 0, DEMATCH_FILL, DEMATCH_COPY, DEMATCH_COMBINE };
// One step of the reference's walk over the soft buffer: positions [begin, begin + count) are cleared, set to
// +infinity, or take / add elements [src, src + count) of the deinterleaved input.
struct DematchOp {
  uint32_t kind, begin, count, src;
};
constexpr uint32_t MAX_DEMATCH_OPS = 60;
struct DematchLaunch {
  const int8_t* in;  // per codeblock: rm_length soft bits as received
  int8_t*       out; // per codeblock: the soft buffer, block_length soft bits
  uint32_t      in_stride, out_stride, block_length, qm, cols, n_ops;
  uint32_t      in_stride_outer, out_stride_outer; // a second batch dimension (transport blocks of codeblocks)
  uint32_t      skip_load; // the operations write every soft bit of the block: the old contents need not be read
  uint32_t      disjoint;  // no two operations touch the same soft bit: they can run in any order, element by element
  const DematchOp* ops_ext; // the list in device memory when it has more than MAX_DEMATCH_OPS entries, else null
  DematchOp        ops[MAX_DEMATCH_OPS];
};
hipError_t launch_ldpc_dematch(const DematchLaunch& p, uint32_t n_cb, hipStream_t stream, uint32_t n_outer = 1);

// ---- PUSCH decoder: transport-block assembly ("next" row, receive side) --------------------------------------------
struct PuschAssembleLaunch {
  const uint8_t* cb_msg;      // [n_tb][C][msg_stride] decoded messages, packed MSB first
  uint8_t*       cb_ok;       // [n_tb][C] codeblock CRC flags (cleared when the transport-block CRC fails)
  const uint32_t* cb_iter;    // [n_tb][C] iterations of this call's decodes (0: failed or skipped)
  const uint8_t* skipped;     // [n_tb][C] 1 where this call did not run the decoder (CRC ok since an earlier transmission)
  uint8_t*       tb;          // [n_tb][tb_stride] transport blocks out
  uint32_t*      result;      // [n_tb][4]: tb_crc_ok, codeblocks with CRC ok, sum and max of iterations over decoded codeblocks
  const uint32_t* crc_weight; // [PUSCH_ASSEMBLE_THREADS] x^(8 * bytes behind run t of the block) mod CRC24A (the byte-wise form: unaligned blocks)
  const TbCrcTables* tbcrc;   // the context's TB-CRC tables and ...
  uint32_t       crc_factor;  // ... x^(24 + 8 (tb_bytes - end of the last 16 KiB region)) mod CRC24A (the form by regions)
  uint32_t       C, msg_stride, tb_stride, tb_bytes, cb_info_bits, max_iterations;
};
constexpr uint32_t PUSCH_ASSEMBLE_THREADS = 1024;
hipError_t launch_pusch_assemble(const PuschAssembleLaunch& p, uint32_t n_tb, hipStream_t stream);

// ---- NZP-CSI-RS ("next" row: other downlink grid writers) -------------------------------------------------------------
constexpr uint32_t CSI_RS_MAX_SEQ_WORDS = 128; // 2 * (3 * 275 skipped + 3 * 275 used) bits and some
// One CDM group of one signal (rows 1-5 have one OFDM symbol per group).
struct CsiRsWork {
  uint32_t grid_index, symbol, c_init;
  uint32_t advance, seq_len;          // sequence elements skipped / used
  uint32_t rb_begin, rb_stride, re_mask, n_re_prb;
  uint32_t nof_ports, first_layer, group_size;
  uint32_t weights_offset;            // floats: [nof_ports][nof_ports] complex of the signal
  float    amplitude;                 // config amplitude / sqrt(2)
};
struct CsiRsLaunch {
  const CsiRsWork*  work;
  const float*      weights;
  const GoldTables* gold;
  const uint32_t*   x1_words;
  uint32_t*         grid;
  uint32_t          grid_nof_ports, grid_nof_subc;
};
hipError_t launch_csi_rs(const CsiRsLaunch& p, uint32_t n_work, hipStream_t stream);
hipError_t launch_llr_descramble(const GoldTables* gold, const uint32_t* x1_words, const uint32_t* d_c_init, uint32_t n_cw,
                                 uint32_t length, const int8_t* d_in, size_t in_stride, int8_t* d_out, size_t out_stride,
                                 hipStream_t stream);
// ---- soft demodulator ("next" row, receive side) ---------------------------------------------------------------------
constexpr uint32_t DEMOD_MAX_PAIRS = 4;
struct DemodLaunch {
  uint32_t modulation; // NRPHY_MOD_*
  uint32_t span_len;   // symbols per span
  uint32_t nof_vector; // leading symbols of a span that take the reference's vector arithmetic
  float    range, scale; // quantisation range limit and 120 / range
  float    qam16_gain, qam16_threshold; // 4 / sqrt(10), 2 / sqrt(10)
  uint32_t nof_intervals[DEMOD_MAX_PAIRS];
  float    width[DEMOD_MAX_PAIRS], rcp_width[DEMOD_MAX_PAIRS];
  float    slope[DEMOD_MAX_PAIRS][16], intercept[DEMOD_MAX_PAIRS][16];
};
hipError_t launch_demodulate_soft(const DemodLaunch& p, uint32_t nof_spans, const float* d_symbols, const float* d_noise,
                                  int8_t* d_llr, hipStream_t stream);
hipError_t launch_grid_put(const uint32_t* d_index, const uint32_t* d_value, uint32_t n, uint32_t* d_grid, hipStream_t stream);

// ---- PDCCH and SS/PBCH block ("next" row: other downlink grid writers) --------------------------------------------------
// One wavefront per DCI.  Offsets index the launch's shared tables: tab16 (gather table of the polar input, PRB list),
// bytes (payload bits, one per byte), words (CRC weights), weights (floats).
struct PdcchWork {
  uint32_t grid_index;
  uint32_t A, E, N;        // payload bits, rate-matched bits, polar code length
  uint32_t mode;           // bit selection: 0 repetition, 1 puncturing, 2 shortening
  uint32_t rnti, crc_const; // crc_const: the share of the 24 leading ones in the CRC
  uint32_t c_init_data;
  uint32_t start_symbol, duration, n_prb, ref_rb, top_prb; // top_prb: highest allocated PRB + 1
  uint32_t nof_ports, prg_size_subc;
  uint32_t dmrs_c_init[3];
  float    data_amp, dmrs_amp;
  uint32_t weights_offset, src_offset, prb_offset, payload_offset, crcw_offset, enc_offset;
};
// One wavefront per SS/PBCH block.
struct SsbWork {
  uint32_t grid_index, l0, k0, pci;
  uint32_t a_bits;     // PBCH payload a_0 .. a_31 after the interleaver G, a_0 in the MSB
  uint32_t scr_mask;   // positions of a that are scrambled, position 0 in the MSB
  uint32_t scr_adv;    // M v: offset of the payload scrambling sequence
  uint32_t ssb_adv;    // (ssb_idx mod 8) * 864: offset of the PBCH scrambling sequence
  uint32_t dmrs_c_init;
  uint32_t m_pss, m0, m1;
  float    pss_amp;
  uint32_t nof_ports;
  uint8_t  ports[NRPHY_MAX_PORTS];
  uint32_t src_offset, crcw_offset, enc_offset;
};
struct DlControlLaunch {
  const PdcchWork*  pdcch;
  const SsbWork*    ssb;
  const float*      weights;
  const uint16_t*   tab16;
  const uint8_t*    bytes;
  const uint32_t*   words;
  const GoldTables* gold;
  const uint32_t*   x1_words;
  uint32_t*         grid; // null: encode only
  uint8_t*          enc;  // null, or the rate-matched bits (one per byte) of every work item at its enc_offset
  uint32_t          grid_nof_ports, grid_nof_subc;
};
hipError_t launch_pdcch(const DlControlLaunch& p, uint32_t n, hipStream_t stream);
hipError_t launch_ssb(const DlControlLaunch& p, uint32_t n, hipStream_t stream);

// ---- OFDM ---------------------------------------------------------------------------------------------------
struct OfdmLaunch {
  uint32_t       dft_size;
  uint32_t       rg_size;     // 12 * bw_rb
  uint32_t       nof_ports;
  uint32_t       nsymb;       // symbols per slot
  uint32_t       slot_stride; // samples per (grid, port) in the output
  const float2*  twiddle;     // exp(+j 2 pi k / N), k < N
  const float2*  phase;       // [symbols per subframe] phase compensation * scale
  const float2*  window_phase; // demodulator with a window offset: [dft_size] exp(+j 2 pi offset i / N), else unused
  const uint32_t* cp_len;     // [symbols per subframe]
  const uint32_t* sym_offset; // [symbols per subframe] start of the symbol within its slot (samples)
  uint32_t       probe;       // NRPHY_OFDM_PROBE: timing-only runs with the loads and/or stores range-checked away
  // Wire-format output (nrphy_ofdm_run_ci16): amplitude controller + int16 conversion fused into the store; d_iq then
  // points at complex int16 samples.
  uint32_t       wire = 0, wire_clip = 0;
  float          wire_gain = 1.f, wire_ceiling = 0.f, wire_scale = 1.f;
  float          wire_limit = 0.f; // sample powers up to this neither clip nor saturate: min(ceiling, largest x with x * scale <= 32767)^2
  nrphy_amplitude_stats_t* wire_stats = nullptr; // [grid][port], may be null
  uint4*         wire_partials = nullptr; // [grid][port][workgroup]: {sum power, peak power, clipped, -} of one workgroup
};

// ---- lower-PHY tail ("next" row: amplitude controller, radio sample format, fronthaul compression) ----------------------
struct AmplitudeLaunch {
  const float* in;
  float*       out;
  size_t       in_stride, out_stride; // complex samples between buffers
  uint32_t     nof_samples, measure, clip;
  float        gain, ceiling;
  nrphy_amplitude_stats_t* stats; // per buffer, cleared by the caller; may be null
};
hipError_t launch_amplitude(const AmplitudeLaunch& p, uint32_t n_buffers, hipStream_t stream);
hipError_t launch_convert_ci16(const float* in, size_t in_stride, int16_t* out, size_t out_stride, uint32_t n_buffers,
                               uint32_t nof_samples, float scale, hipStream_t stream);
struct OfhCompressLaunch {
  const uint32_t* prbs;      // cbf16 words
  uint8_t*        out;
  size_t          row_stride, out_row_stride; // words / bytes between rows
  uint32_t        nof_prb, data_width, bfp, whole_span;
  float           scale;     // quantiser gain * iq_scaling
};
hipError_t launch_ofh_compress(const OfhCompressLaunch& p, uint32_t n_rows, hipStream_t stream);
hipError_t launch_ofdm(const OfdmLaunch& p, uint32_t nof_grids, const uint32_t* d_grid, const uint32_t* d_slot_index,
                       float2* d_iq, hipStream_t stream);
// OFDM demodulation (the receive-side mirror of launch_ofdm): `p.phase` is the receive table (conjugate phase x scale),
// window_offset = nof_samples_window_offset of the reference's demodulator configuration.
hipError_t launch_ofdm_demod(const OfdmLaunch& p, uint32_t nof_grids, const float2* d_iq, const uint32_t* d_slot_index,
                             uint32_t window_offset, uint32_t* d_grid, hipStream_t stream);
// size = n1 * n2: n1 = 1 for the sizes one workgroup transforms in LDS (128 ... 6144, what the OFDM kernels take), else
// a radix-n1 column pass through global memory in front of n1 LDS transforms of size n2 (9216 ... 49152).
bool       dft_split(uint32_t size, uint32_t* n1, uint32_t* n2);
bool       dft_size_supported(uint32_t size);
inline bool dft_size_in_lds(uint32_t size)
{
  uint32_t n1 = 0, n2 = 0;
  return dft_split(size, &n1, &n2) && n1 == 1;
}
// twiddle: exp(+j 2 pi k / size); split sizes also need the table of n2 and `batch` transforms of scratch.
hipError_t launch_dft(uint32_t size, int inverse, uint32_t batch, const float2* twiddle, const float2* twiddle_n2,
                      float2* d_tmp, const float2* d_in, float2* d_out, hipStream_t stream);

} // namespace nrphy
