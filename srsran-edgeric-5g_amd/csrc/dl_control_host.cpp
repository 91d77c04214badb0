// Host side of the downlink control channels: PDCCH and SS/PBCH block processors.
//
// What the reference computes per PDU before its loops -- the polar code of (K, E), the CCE-to-REG mapping, sequence
// initialisations, amplitudes -- is derived here and handed to one wavefront per DCI / block
// (dl_control_kernels.hip).  No channel bit or resource element is produced on the host.
#include "nrphy_host_internal.h"
#include "nrphy_trace.h"

#include <cmath>

namespace {

#include "nr_polar_tables.inc"

const CrcField CRC24C_FIELD = {0x1B2B117U, 24};

// TS 38.212 Table 5.4.1.1-1.
const uint8_t SUBBLOCK_PATTERN[32] = {0,  1,  2,  4,  3,  5,  6,  7,  8,  16, 9,  17, 10, 18, 11, 19,
                                      12, 20, 13, 21, 14, 22, 15, 23, 24, 25, 26, 28, 27, 29, 30, 31};

// The polar code of a downlink channel (n_max = 9, no parity-check bits): code length, how the rate matcher selects
// bits, and for every polar input position the message bit it carries (after the interleaver of TS 38.212 Section
// 5.3.1.1) or 0xFFFF when frozen.  Replaces polar_code_impl::set (R/lib/phy/upper/channel_coding/polar/
// polar_code_impl.cpp:300-470), polar_interleaver_impl::interleave and polar_allocator_impl::allocate.
struct PolarCode {
  uint32_t              N = 0, mode = 0; // mode: 0 repetition, 1 puncturing, 2 shortening
  std::vector<uint16_t> src;
};

bool build_polar_code(uint32_t K, uint32_t E, PolarCode& code)
{
  if (K < 36 || K > 164 || K >= E || E > 8192) {
    return false;
  }
  // Code length (TS 38.212 Section 5.3.1): n = max(min(n1, n2, n_max), 5).
  uint32_t e = 1, k = 0;
  while ((1U << e) < E) {
    ++e;
  }
  while ((1U << k) < K) {
    ++k;
  }
  const uint32_t n1 = (8 * E <= 9 * (1U << (e - 1)) && 16 * K < 9 * E) ? e - 1 : e;
  const uint32_t n  = std::max<uint32_t>(5, std::min<uint32_t>(std::min(n1, k + 3), 9));
  const uint32_t N  = 1U << n;
  if (K >= N) {
    return false;
  }
  auto J = [N](uint32_t i) { return SUBBLOCK_PATTERN[(32 * i) / N] * (N / 32) + i % (N / 32); };
  // Positions that can carry nothing because rate matching drops their coded bits, and the low indices excluded
  // with them (the reference's bound: every index <= T, see polar_code_impl.cpp:395-414).
  std::vector<uint8_t> barred(N, 0);
  code.mode = 0;
  if (N > E) {
    uint32_t T = 0;
    if (16 * K <= 7 * E) {
      code.mode = 1;
      T         = (E >= 3 * N / 4) ? 3 * N / 4 - (E >> 1) - 1 : 9 * N / 16 - (E >> 2);
      for (uint32_t i = 0; i != N - E; ++i) {
        barred[J(i)] = 1;
      }
    } else {
      code.mode = 2;
      for (uint32_t i = E; i != N; ++i) {
        barred[J(i)] = 1;
      }
    }
    for (uint32_t i = 0; i <= T; ++i) {
      barred[i] = 1;
    }
  }
  // The K most reliable of the remaining positions: walk the polar sequence from its reliable end.
  std::vector<uint8_t> info(N, 0);
  uint32_t             found = 0;
  for (int i = 1023; i >= 0 && found != K; --i) {
    const uint32_t q = NR_POLAR_RELIABILITY[i];
    if (q < N && !barred[q]) {
      info[q] = 1;
      ++found;
    }
  }
  if (found != K) {
    return false;
  }
  // Interleaver: c'_k = c_{Pi(k)}; allocation: the k-th information position (ascending) carries c'_k.
  std::vector<uint16_t> pi;
  for (uint32_t m = 0; m != 164; ++m) {
    if (NR_POLAR_IL_MAX[m] >= 164 - K) {
      pi.push_back((uint16_t)(NR_POLAR_IL_MAX[m] - (164 - K)));
    }
  }
  code.N = N;
  code.src.assign(N, 0xFFFF);
  uint32_t next = 0;
  for (uint32_t i = 0; i != N; ++i) {
    if (info[i]) {
      code.src[i] = pi[next++];
    }
  }
  return true;
}

// The PRBs of a PDCCH candidate in ascending order (TS 38.211 Section 7.3.2.2; replaces pdcch_processor_impl::
// compute_rb_mask, pdcch_processor_impl.cpp:30-64, and R/lib/ran/pdcch/cce_to_prb_mapping.cpp).
bool pdcch_prb_list(const nrphy_pdcch_pdu_t& p, std::vector<uint16_t>& prbs)
{
  prbs.clear();
  const uint32_t al = p.aggregation_level, dur = p.duration;
  if (dur < 1 || dur > 3 || !(al == 1 || al == 2 || al == 4 || al == 8 || al == 16) || p.cce_to_reg_mapping > 2) {
    return false;
  }
  // The CORESET's PRBs in REG order (REGs are numbered time first: REG r sits on PRB coreset_prb[r / dur]).
  std::vector<uint16_t> coreset_prb;
  if (p.cce_to_reg_mapping == 0) {
    for (uint32_t i = 0; i != p.bwp_size_rb; ++i) {
      coreset_prb.push_back((uint16_t)(p.bwp_start_rb + i));
    }
  } else {
    for (uint32_t f = 0; f != 45; ++f) {
      if ((p.frequency_resources >> f) & 1U) {
        for (uint32_t i = 0; i != 6; ++i) {
          coreset_prb.push_back((uint16_t)(p.bwp_start_rb + 6 * f + i));
        }
      }
    }
  }
  const uint32_t n_reg = (uint32_t)coreset_prb.size() * dur;
  if (n_reg == 0 || 6 * ((uint64_t)p.cce_index + al) > n_reg) { // (64-bit: a huge CCE index must not wrap into range)
    return false;
  }
  std::vector<uint32_t> bundles; // (first REG, REG count) of every REG bundle of the candidate
  uint32_t              L = 6;
  if (p.cce_to_reg_mapping == 1) {
    bundles.push_back(6 * p.cce_index);
    L = 6 * al;
  } else {
    L                = p.cce_to_reg_mapping == 0 ? 6 : p.reg_bundle_size;
    const uint32_t R = p.cce_to_reg_mapping == 0 ? 2 : p.interleaver_size;
    // L in {2, 3, 6}, R in {2, 3, 6} (TS 38.211 Section 7.3.2.2); bounded before L * R is formed
    if (L == 0 || R == 0 || L > 6 || R > 6 || 6 % L != 0 || n_reg % (L * R) != 0 || L % dur != 0) {
      return false;
    }
    const uint32_t C = n_reg / (L * R);
    for (uint32_t b = p.cce_index * (6 / L); b != (p.cce_index + al) * (6 / L); ++b) {
      bundles.push_back(((b % R) * C + b / R + p.shift_index) % (n_reg / L) * L);
    }
  }
  if (L % dur != 0) {
    return false;
  }
  for (uint32_t first : bundles) {
    for (uint32_t r = first; r != first + L; r += dur) {
      if (r / dur >= coreset_prb.size()) {
        return false;
      }
      prbs.push_back(coreset_prb[r / dur]);
    }
  }
  std::sort(prbs.begin(), prbs.end());
  return prbs.size() * dur == 6 * al;
}

float db_to_amplitude(float db) // convert_dB_to_amplitude, R/include/srsran/support/math_utils.h:116-119
{
  return std::pow(10.0F, db / 20.0F);
}

// Everything a launch reads besides the grid, packed into one staging block (one allocation, one copy).
struct ControlStaging {
  std::vector<PdcchWork> pdcch;
  std::vector<SsbWork>   ssb;
  std::vector<float>     weights;
  std::vector<uint16_t>  tab16;
  std::vector<uint8_t>   bytes;
  std::vector<uint32_t>  words;

  int launch(nrphy_ctx* ctx, void* d_grid, uint32_t grid_nof_ports, uint32_t grid_nof_subc, uint8_t* d_enc, hipStream_t s)
  {
    auto           pad    = [](size_t b) { return (b + 63) & ~(size_t)63; };
    const size_t   o_work = 0, n_work = pdcch.size() * sizeof(PdcchWork) + ssb.size() * sizeof(SsbWork);
    const size_t   o_w = pad(n_work), o_t = o_w + pad(weights.size() * sizeof(float));
    const size_t   o_b = o_t + pad(tab16.size() * sizeof(uint16_t)), o_x = o_b + pad(bytes.size());
    const size_t   total = o_x + pad(words.size() * sizeof(uint32_t));
    std::vector<uint8_t> blob(total, 0);
    if (!pdcch.empty()) {
      std::memcpy(&blob[o_work], pdcch.data(), pdcch.size() * sizeof(PdcchWork));
    }
    if (!ssb.empty()) {
      std::memcpy(&blob[o_work], ssb.data(), ssb.size() * sizeof(SsbWork));
    }
    auto put = [&blob](size_t off, const void* src, size_t bytes_) {
      if (bytes_ != 0) {
        std::memcpy(&blob[off], src, bytes_);
      }
    };
    put(o_w, weights.data(), weights.size() * sizeof(float));
    put(o_t, tab16.data(), tab16.size() * sizeof(uint16_t));
    put(o_b, bytes.data(), bytes.size());
    put(o_x, words.data(), words.size() * sizeof(uint32_t));
    StreamStaging staging(s);
    uint8_t*      base = (uint8_t*)staging.alloc(total);
    if (base == nullptr) {
      return NRPHY_ERR_DEVICE;
    }
    HIP_TRY(hipMemcpyAsync(base, blob.data(), total, hipMemcpyHostToDevice, s));
    DlControlLaunch p;
    p.pdcch          = (const PdcchWork*)(base + o_work);
    p.ssb            = (const SsbWork*)(base + o_work);
    p.weights        = (const float*)(base + o_w);
    p.tab16          = (const uint16_t*)(base + o_t);
    p.bytes          = base + o_b;
    p.words          = (const uint32_t*)(base + o_x);
    p.gold           = ctx->d_gold;
    p.x1_words       = ctx->d_x1;
    p.grid           = (uint32_t*)d_grid;
    p.enc            = d_enc;
    p.grid_nof_ports = grid_nof_ports;
    p.grid_nof_subc  = grid_nof_subc;
    HIP_TRY(launch_pdcch(p, (uint32_t)pdcch.size(), s));
    HIP_TRY(launch_ssb(p, (uint32_t)ssb.size(), s));
    return NRPHY_OK;
  }
};

// Gather table of a polar code appended to tab16 once per distinct (K, E) of a call.
uint32_t polar_table_offset(ControlStaging& st, std::map<uint64_t, std::pair<uint32_t, PolarCode>>& cache, uint32_t K,
                            uint32_t E, const PolarCode** code)
{
  const uint64_t key = ((uint64_t)K << 32) | E;
  auto           it  = cache.find(key);
  if (it == cache.end()) {
    PolarCode c;
    if (!build_polar_code(K, E, c)) {
      *code = nullptr;
      return 0;
    }
    const uint32_t off = (uint32_t)st.tab16.size();
    st.tab16.insert(st.tab16.end(), c.src.begin(), c.src.end());
    it = cache.emplace(key, std::make_pair(off, std::move(c))).first;
  }
  *code = &it->second.second;
  return it->second.first;
}

// CRC24C weights of an A-bit message followed by the 24 parity bits: bit i weighs x^(A - 1 - i + 24) mod g.
uint32_t crc_weight_offset(ControlStaging& st, std::map<uint32_t, uint32_t>& cache, uint32_t A)
{
  auto it = cache.find(A);
  if (it == cache.end()) {
    const uint32_t off = (uint32_t)st.words.size();
    for (uint32_t i = 0; i != A; ++i) {
      st.words.push_back(CRC24C_FIELD.xpow((int64_t)(A - 1 - i) + 24));
    }
    it = cache.emplace(A, off).first;
  }
  return it->second;
}

bool add_pdcch(ControlStaging& st, std::map<uint64_t, std::pair<uint32_t, PolarCode>>& codes, std::map<uint32_t, uint32_t>& crcs,
               const nrphy_pdcch_pdu_t& p, uint32_t grid_index, uint32_t enc_offset, std::vector<uint16_t>& prbs)
{
  if (nrphy_pdcch_validate(&p) != NRPHY_OK || !pdcch_prb_list(p, prbs)) {
    return false;
  }
  PdcchWork w;
  std::memset(&w, 0, sizeof(w));
  const PolarCode* code = nullptr;
  w.grid_index     = grid_index;
  w.A              = p.payload_size;
  w.E              = 108 * p.aggregation_level; // aggregation_level * NOF_REG_PER_CCE * NOF_RE_PDCCH_PER_RB * 2
  w.src_offset     = polar_table_offset(st, codes, w.A + 24, w.E, &code);
  if (code == nullptr) {
    return false;
  }
  w.N    = code->N;
  w.mode = code->mode;
  w.rnti = p.rnti;
  // the 24 ones in front of the payload: sum of x^(A + 23 - j + 24), j < 24
  for (uint32_t j = 0; j != 24; ++j) {
    w.crc_const ^= CRC24C_FIELD.xpow((int64_t)(w.A + 23 - j) + 24);
  }
  w.crcw_offset    = crc_weight_offset(st, crcs, w.A);
  w.c_init_data    = (uint32_t)((((uint64_t)p.n_rnti << 16) + p.n_id_pdcch_data) & 0x7FFFFFFFULL);
  w.start_symbol   = p.start_symbol_index;
  w.duration       = p.duration;
  w.n_prb          = (uint32_t)prbs.size();
  w.ref_rb         = p.cce_to_reg_mapping == 0 ? p.bwp_start_rb : 0;
  w.top_prb        = prbs.back() + 1U;
  w.nof_ports      = p.nof_ports;
  w.prg_size_subc  = 12 * p.prg_size_rb;
  const uint32_t nsymb = p.cp ? 12 : 14;
  for (uint32_t s = 0; s != p.duration; ++s) {
    // dmrs_pdcch_processor_impl::c_init (dmrs_pdcch_processor_impl.cpp:32-38)
    const uint64_t a = (uint64_t)(nsymb * p.slot_index + p.start_symbol_index + s + 1) * (2 * p.n_id_pdcch_dmrs + 1);
    w.dmrs_c_init[s] = (uint32_t)(((a << 17) + 2 * p.n_id_pdcch_dmrs) & 0x7FFFFFFFULL);
  }
  // pdcch_modulator_impl::modulate: QPSK amplitude times the power scaling when that is a normal number
  const float scaling = db_to_amplitude(p.data_power_offset_dB);
  w.data_amp          = (float)M_SQRT1_2;
  if (std::isnormal(scaling)) {
    w.data_amp = w.data_amp * scaling;
  }
  w.dmrs_amp       = (float)(M_SQRT1_2 * (double)db_to_amplitude(p.dmrs_power_offset_dB));
  w.weights_offset = (uint32_t)st.weights.size();
  st.weights.insert(st.weights.end(), p.precoding, p.precoding + 2 * (size_t)p.nof_prg * p.nof_ports);
  w.prb_offset = (uint32_t)st.tab16.size();
  st.tab16.insert(st.tab16.end(), prbs.begin(), prbs.end());
  w.payload_offset = (uint32_t)st.bytes.size();
  st.bytes.insert(st.bytes.end(), p.payload, p.payload + p.payload_size);
  w.enc_offset = enc_offset;
  st.pdcch.push_back(w);
  return true;
}

// ---- SS/PBCH block ---------------------------------------------------------------------------------------------------
// TS 38.212 Table 7.1.1-1: PBCH payload interleaver pattern G(j).
const uint8_t PBCH_G[32] = {16, 23, 18, 17, 8,  30, 10, 6,  24, 7,  0,  5,  3,  2,  1,  4,
                            9,  11, 12, 13, 14, 15, 19, 20, 21, 22, 25, 26, 27, 28, 29, 31};

// First OFDM symbol of a candidate block within the half frame (TS 38.213 Section 4.1), -1 when out of range.
int ssb_first_symbol(uint32_t pattern_case, uint32_t idx)
{
  static const uint32_t group16[16] = {0, 1, 2, 3, 5, 6, 7, 8, 10, 11, 12, 13, 15, 16, 17, 18};
  switch (pattern_case) {
    case 0: // A
    case 2: // C
      return (int)((idx % 2 ? 8 : 2) + 14 * (idx / 2));
    case 1: { // B
      static const uint32_t s[4] = {4, 8, 16, 20};
      return (int)(s[idx % 4] + 28 * (idx / 4));
    }
    case 3: { // D
      static const uint32_t s[4] = {4, 8, 16, 20};
      return idx < 64 ? (int)(s[idx % 4] + 28 * group16[idx / 4]) : -1;
    }
    case 4: { // E
      static const uint32_t s[8] = {8, 12, 16, 20, 32, 36, 40, 44};
      return idx < 128 ? (int)(s[idx % 8] + 56 * group16[idx / 8]) : -1;
    }
    default:
      return -1;
  }
}

// First subcarrier of the block in the grid (ssb_get_k_first, R/include/srsran/ran/ssb_mapping.h:116-167), -1 when the
// combination is one the reference refuses.
int ssb_first_subcarrier(const nrphy_ssb_pdu_t& p)
{
  static const uint32_t block_scs_khz[5] = {15, 30, 30, 120, 240};
  if (p.pattern_case > 4 || p.common_scs > 4 || p.offset_to_pointA > 2199) { // is_scs_valid: up to 240 kHz
    return -1;
  }
  const bool     fr2        = p.pattern_case >= 3;
  const uint32_t common_khz = 15U << p.common_scs, scs = block_scs_khz[p.pattern_case];
  if ((!fr2 && common_khz > 60) || (fr2 && common_khz < 60) || p.subcarrier_offset > (fr2 ? 11U : 23U)) {
    return -1;
  }
  // subcarriers of 15 kHz between point A and the block: offsetToPointA counts PRBs of 15 (FR1) / 60 kHz (FR2), k_SSB
  // subcarriers of 15 kHz (FR1) / of the common spacing (FR2)
  const uint32_t k15 = (p.offset_to_pointA * 12 * (fr2 ? 60U : 15U) + p.subcarrier_offset * (fr2 ? common_khz : 15U)) / 15;
  if ((k15 * 15) % scs != 0) {
    return -1;
  }
  return (int)((k15 * 15) / scs);
}

bool add_ssb(ControlStaging& st, uint32_t src_offset, uint32_t crcw_offset, const nrphy_ssb_pdu_t& p, uint32_t grid_index,
             uint32_t grid_nof_ports, uint32_t grid_nof_subc, bool has_grid, uint32_t enc_offset)
{
  if (nrphy_ssb_validate(&p) != NRPHY_OK) {
    return false;
  }
  SsbWork w;
  std::memset(&w, 0, sizeof(w));
  w.grid_index = grid_index;
  w.l0         = (uint32_t)ssb_first_symbol(p.pattern_case, p.ssb_idx) % 14;
  w.k0         = (uint32_t)ssb_first_subcarrier(p);
  if (has_grid && (w.l0 + 4 > 14 || w.k0 + 240 > grid_nof_subc)) {
    return false;
  }
  for (uint32_t i = 0; i != p.nof_ports; ++i) {
    if (has_grid && p.ports[i] >= grid_nof_ports) {
      return false;
    }
    w.ports[i] = p.ports[i];
  }
  w.nof_ports = p.nof_ports;
  w.pci       = p.phys_cell_id;
  // PBCH payload generation (TS 38.212 Section 7.1.1; pbch_encoder_impl.cpp:38-86): descriptor fields into a_0 .. a_31.
  const uint32_t hrf = (p.slot_index / (1U << p.numerology)) >= 5 ? 1 : 0; // slot_point::is_odd_hrf
  uint8_t        a[32];
  std::memset(a, 0, sizeof(a));
  uint32_t j_sfn = 0, j_other = 14;
  for (uint32_t i = 0; i != 24; ++i) {
    a[PBCH_G[(i >= 1 && i < 7) ? j_sfn++ : j_other++]] = p.bch_payload[i] & 1U;
  }
  for (int b = 3; b >= 0; --b) {
    a[PBCH_G[j_sfn++]] = (uint8_t)((p.sfn >> b) & 1U);
  }
  a[PBCH_G[10]] = (uint8_t)hrf;
  if (p.L_max == 64) {
    a[PBCH_G[11]] = (uint8_t)((p.ssb_idx >> 5) & 1U);
    a[PBCH_G[12]] = (uint8_t)((p.ssb_idx >> 4) & 1U);
    a[PBCH_G[13]] = (uint8_t)((p.ssb_idx >> 3) & 1U);
  } else {
    a[PBCH_G[11]] = (uint8_t)((p.subcarrier_offset >> 4) & 1U);
  }
  // Scrambling (Section 7.1.2): not the half-frame bit, not the 2nd / 3rd LSB of the SFN, not the block index bits.
  w.scr_mask = 0xFFFFFFFFU;
  auto clear = [&w](uint32_t pos) { w.scr_mask &= ~(0x80000000U >> pos); };
  clear(PBCH_G[10]);
  clear(PBCH_G[7]);
  clear(PBCH_G[8]);
  if (p.L_max == 64) {
    clear(PBCH_G[11]);
    clear(PBCH_G[12]);
    clear(PBCH_G[13]);
  }
  for (uint32_t i = 0; i != 32; ++i) {
    w.a_bits |= (uint32_t)a[i] << (31 - i);
  }
  const uint32_t M = (p.L_max == 64) ? 32 - 6 : 32 - 3;
  w.scr_adv        = M * (2U * a[PBCH_G[7]] + a[PBCH_G[8]]);
  w.ssb_adv        = (p.ssb_idx & 7U) * 864U; // pbch_modulator_impl.cpp:35: three LSBs whatever L_max is
  // dmrs_pbch_processor_impl::c_init (dmrs_pbch_processor_impl.cpp:29-40)
  uint64_t i_ssb = (p.ssb_idx & 3U) + 4ULL * hrf;
  if (p.L_max == 8 || p.L_max == 64) {
    i_ssb = p.ssb_idx & 7U;
  }
  w.dmrs_c_init = (uint32_t)((((i_ssb + 1) * ((p.phys_cell_id / 4) + 1)) << 11) + ((i_ssb + 1) << 6) + (p.phys_cell_id % 4));
  const uint32_t nid1 = p.phys_cell_id / 3, nid2 = p.phys_cell_id % 3;
  w.m_pss       = 43 * nid2;
  w.m0          = 15 * (nid1 / 112) + 5 * nid2;
  w.m1          = nid1 % 112;
  w.pss_amp     = db_to_amplitude(p.beta_pss_dB);
  w.src_offset  = src_offset;
  w.crcw_offset = crcw_offset;
  w.enc_offset  = enc_offset;
  st.ssb.push_back(w);
  return true;
}

// The fixed PBCH code: K = 32 + 24, E = 864.
bool add_pbch_tables(ControlStaging& st, uint32_t* src_offset, uint32_t* crcw_offset)
{
  PolarCode code;
  if (!build_polar_code(56, 864, code) || code.N != 512) {
    return false;
  }
  *src_offset = (uint32_t)st.tab16.size();
  st.tab16.insert(st.tab16.end(), code.src.begin(), code.src.end());
  *crcw_offset = (uint32_t)st.words.size();
  for (uint32_t i = 0; i != 32; ++i) {
    st.words.push_back(CRC24C_FIELD.xpow((int64_t)(31 - i) + 24));
  }
  return true;
}

} // namespace

// ================================================================================================================
// PDCCH
// ================================================================================================================
extern "C" int nrphy_pdcch_validate(const nrphy_pdcch_pdu_t* p)
{
  if (p == nullptr || p->payload_size < 12 || p->payload_size > NRPHY_PDCCH_MAX_PAYLOAD || p->cp > 1 ||
      p->precoding == nullptr || p->nof_ports == 0 || p->nof_ports > NRPHY_MAX_PORTS || p->nof_prg == 0 || p->nof_prg > NRPHY_MAX_PRG ||
      p->prg_size_rb == 0 || p->prg_size_rb > NRPHY_MAX_RB || p->bwp_size_rb == 0 || p->bwp_start_rb + p->bwp_size_rb > NRPHY_MAX_RB ||
      p->start_symbol_index + p->duration > (p->cp ? 12U : 14U) || p->rnti > 65535 || p->n_rnti > 65535 ||
      p->n_id_pdcch_data > 65535 || p->n_id_pdcch_dmrs > 65535 || (p->frequency_resources >> 45) != 0) {
    return NRPHY_ERR_INVALID_PDU;
  }
  std::vector<uint16_t> prbs;
  if (!pdcch_prb_list(*p, prbs) || prbs.empty()) {
    return NRPHY_ERR_INVALID_PDU;
  }
  // polar_code_impl::set_code_params: K < E
  if (p->payload_size + 24 >= 108 * p->aggregation_level) {
    return NRPHY_ERR_INVALID_PDU;
  }
  // inside the BWP (the RB mask of pdcch_processor_impl has bwp_start + bwp_size bits) ...
  if (prbs.back() >= p->bwp_start_rb + p->bwp_size_rb) {
    return NRPHY_ERR_INVALID_PDU;
  }
  // ... and PRGs that cover the allocation exactly: resource_grid_mapper_impl.cpp:233-262 walks nof_prg slices of
  // prg_size over a mask that ends with the highest allocated PRB.
  const uint32_t top = prbs.back() + 1U;
  if ((uint64_t)(p->nof_prg - 1) * p->prg_size_rb >= top || (uint64_t)p->nof_prg * p->prg_size_rb < top) {
    return NRPHY_ERR_INVALID_PDU;
  }
  return NRPHY_OK;
}

extern "C" int nrphy_pdcch_process(nrphy_ctx_t* ctx, uint32_t n, const nrphy_pdcch_pdu_t* pdus, const uint32_t* grid_index,
                                   void* d_grid, uint32_t grid_nof_ports, uint32_t grid_nof_subc, void* stream)
{
  const TraceRange trace("process_pdcch");
  if (ctx == nullptr || (n != 0 && (pdus == nullptr || d_grid == nullptr)) || grid_nof_ports == 0 ||
      grid_nof_ports > NRPHY_MAX_PORTS || grid_nof_subc == 0) {
    return NRPHY_ERR_ARGUMENT;
  }
  if (n == 0) {
    return NRPHY_OK;
  }
  ControlStaging                                      st;
  std::map<uint64_t, std::pair<uint32_t, PolarCode>> codes;
  std::map<uint32_t, uint32_t>                        crcs;
  std::vector<uint16_t>                               prbs;
  for (uint32_t i = 0; i != n; ++i) {
    if (nrphy_pdcch_validate(&pdus[i]) != NRPHY_OK) {
      return NRPHY_ERR_INVALID_PDU;
    }
    if (!add_pdcch(st, codes, crcs, pdus[i], grid_index ? grid_index[i] : 0, 0, prbs)) {
      return NRPHY_ERR_INVALID_PDU;
    }
    if (pdus[i].nof_ports > grid_nof_ports || 12U * (prbs.back() + 1U) > grid_nof_subc) {
      return NRPHY_ERR_ARGUMENT;
    }
  }
  HIP_TRY(hipSetDevice(ctx->device));
  return st.launch(ctx, d_grid, grid_nof_ports, grid_nof_subc, nullptr, stream ? (hipStream_t)stream : ctx->stream);
}

namespace {

// Host grid in, one launch, host grid out: the shape of every *_process_host call below.
template <class Launch>
int with_host_grid(nrphy_ctx_t* ctx, void* grid, uint32_t nof_ports, uint32_t nof_subc, Launch launch)
{
  std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t bytes  = (size_t)nof_ports * NRPHY_NSYMB * nof_subc * 4;
  void*        d_grid = ctx_scratch(ctx, SCRATCH_GRID, bytes);
  if (d_grid == nullptr) {
    return NRPHY_ERR_DEVICE;
  }
  HIP_TRY(hipMemcpyAsync(d_grid, grid, bytes, hipMemcpyHostToDevice, ctx->stream));
  const int rc = launch(d_grid);
  if (rc != NRPHY_OK) {
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
  }
  HIP_TRY(hipMemcpyAsync(grid, d_grid, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return NRPHY_OK;
}

} // namespace

extern "C" int nrphy_pdcch_process_host(nrphy_ctx_t* ctx, const nrphy_pdcch_pdu_t* pdu, void* grid, uint32_t nof_ports,
                                        uint32_t nof_subc)
{
  if (ctx == nullptr || pdu == nullptr || grid == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  return with_host_grid(ctx, grid, nof_ports, nof_subc, [&](void* d_grid) {
    return nrphy_pdcch_process(ctx, 1, pdu, nullptr, d_grid, nof_ports, nof_subc, ctx->stream);
  });
}

extern "C" int nrphy_pdcch_encode_host(nrphy_ctx_t* ctx, const uint8_t* payload, uint32_t payload_size, uint32_t rnti,
                                       uint32_t rm_length, uint8_t* encoded)
{
  if (ctx == nullptr || payload == nullptr || encoded == nullptr || payload_size < 12 ||
      payload_size > NRPHY_PDCCH_MAX_PAYLOAD || rm_length == 0 || rm_length > 1728 || (rm_length & 1U) != 0) {
    return NRPHY_ERR_ARGUMENT;
  }
  ControlStaging                                      st;
  std::map<uint64_t, std::pair<uint32_t, PolarCode>> codes;
  std::map<uint32_t, uint32_t>                        crcs;
  const PolarCode*                                    code = nullptr;
  PdcchWork                                           w;
  std::memset(&w, 0, sizeof(w));
  w.A          = payload_size;
  w.E          = rm_length;
  w.src_offset = polar_table_offset(st, codes, payload_size + 24, rm_length, &code);
  if (code == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  w.N    = code->N;
  w.mode = code->mode;
  w.rnti = rnti;
  for (uint32_t j = 0; j != 24; ++j) {
    w.crc_const ^= CRC24C_FIELD.xpow((int64_t)(w.A + 23 - j) + 24);
  }
  w.crcw_offset    = crc_weight_offset(st, crcs, w.A);
  w.n_prb          = 1; // unused without a grid
  w.prg_size_subc  = 12;
  w.payload_offset = 0;
  st.bytes.assign(payload, payload + payload_size);
  st.pdcch.push_back(w);
  std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
  HIP_TRY(hipSetDevice(ctx->device));
  uint8_t* d_enc = (uint8_t*)ctx_scratch(ctx, SCRATCH_SMALL, 2048);
  if (d_enc == nullptr) {
    return NRPHY_ERR_DEVICE;
  }
  const int rc = st.launch(ctx, nullptr, 1, 12, d_enc, ctx->stream);
  if (rc != NRPHY_OK) {
    return rc;
  }
  HIP_TRY(hipMemcpyAsync(encoded, d_enc, rm_length, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return NRPHY_OK;
}

// ================================================================================================================
// SS/PBCH block
// ================================================================================================================
extern "C" int nrphy_ssb_validate(const nrphy_ssb_pdu_t* p)
{
  if (p == nullptr || p->numerology > 4 || p->sfn > 1023 || p->slot_index >= (10U << p->numerology) ||
      p->phys_cell_id > 1007 || (p->L_max != 4 && p->L_max != 8 && p->L_max != 64) || p->nof_ports == 0 ||
      p->nof_ports > NRPHY_MAX_PORTS || p->ssb_idx >= 64) {
    return NRPHY_ERR_INVALID_PDU;
  }
  const int l = ssb_first_symbol(p->pattern_case, p->ssb_idx);
  if (l < 0 || ssb_first_subcarrier(*p) < 0) {
    return NRPHY_ERR_INVALID_PDU;
  }
  // A PSS amplitude that is not a number (the reference would write 0 x inf = NaN imaginary parts): refused.
  if (!std::isfinite(db_to_amplitude(p->beta_pss_dB))) {
    return NRPHY_ERR_INVALID_PDU;
  }
  // ssb_processor_impl.cpp:41-44: the slot is the one of its half frame that holds the block
  if ((uint32_t)l / 14 != p->slot_index % ((10U << p->numerology) / 2)) {
    return NRPHY_ERR_INVALID_PDU;
  }
  // A case-E block that starts at symbol 12 of its slot would run past the 14 symbols of the slot grid (the reference
  // writes outside its grid there): refused.
  if ((uint32_t)l % 14 + 4 > 14) {
    return NRPHY_ERR_INVALID_PDU;
  }
  for (uint32_t i = 0; i != p->nof_ports; ++i) {
    if (p->ports[i] >= NRPHY_MAX_PORTS) {
      return NRPHY_ERR_INVALID_PDU;
    }
  }
  return NRPHY_OK;
}

extern "C" int nrphy_ssb_process(nrphy_ctx_t* ctx, uint32_t n, const nrphy_ssb_pdu_t* pdus, const uint32_t* grid_index,
                                 void* d_grid, uint32_t grid_nof_ports, uint32_t grid_nof_subc, void* stream)
{
  const TraceRange trace("process_ssb");
  if (ctx == nullptr || (n != 0 && (pdus == nullptr || d_grid == nullptr)) || grid_nof_ports == 0 ||
      grid_nof_ports > NRPHY_MAX_PORTS || grid_nof_subc == 0) {
    return NRPHY_ERR_ARGUMENT;
  }
  if (n == 0) {
    return NRPHY_OK;
  }
  ControlStaging st;
  uint32_t       src_offset = 0, crcw_offset = 0;
  if (!add_pbch_tables(st, &src_offset, &crcw_offset)) {
    return NRPHY_ERR_DEVICE;
  }
  for (uint32_t i = 0; i != n; ++i) {
    if (nrphy_ssb_validate(&pdus[i]) != NRPHY_OK) {
      return NRPHY_ERR_INVALID_PDU;
    }
    if (!add_ssb(st, src_offset, crcw_offset, pdus[i], grid_index ? grid_index[i] : 0, grid_nof_ports, grid_nof_subc, true, 0)) {
      return NRPHY_ERR_ARGUMENT; // the block does not fit the grid
    }
  }
  HIP_TRY(hipSetDevice(ctx->device));
  return st.launch(ctx, d_grid, grid_nof_ports, grid_nof_subc, nullptr, stream ? (hipStream_t)stream : ctx->stream);
}

extern "C" int nrphy_ssb_process_host(nrphy_ctx_t* ctx, const nrphy_ssb_pdu_t* pdu, void* grid, uint32_t nof_ports,
                                      uint32_t nof_subc)
{
  if (ctx == nullptr || pdu == nullptr || grid == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  return with_host_grid(ctx, grid, nof_ports, nof_subc, [&](void* d_grid) {
    return nrphy_ssb_process(ctx, 1, pdu, nullptr, d_grid, nof_ports, nof_subc, ctx->stream);
  });
}

extern "C" int nrphy_pbch_encode_host(nrphy_ctx_t* ctx, const nrphy_ssb_pdu_t* pdu, uint8_t* encoded)
{
  if (ctx == nullptr || pdu == nullptr || encoded == nullptr) {
    return NRPHY_ERR_ARGUMENT;
  }
  if (nrphy_ssb_validate(pdu) != NRPHY_OK) {
    return NRPHY_ERR_INVALID_PDU;
  }
  ControlStaging st;
  uint32_t       src_offset = 0, crcw_offset = 0;
  if (!add_pbch_tables(st, &src_offset, &crcw_offset) || !add_ssb(st, src_offset, crcw_offset, *pdu, 0, 0, 0, false, 0)) {
    return NRPHY_ERR_DEVICE;
  }
  std::lock_guard<std::recursive_mutex> lock(ctx->host_mutex);
  HIP_TRY(hipSetDevice(ctx->device));
  uint8_t* d_enc = (uint8_t*)ctx_scratch(ctx, SCRATCH_SMALL, 2048);
  if (d_enc == nullptr) {
    return NRPHY_ERR_DEVICE;
  }
  const int rc = st.launch(ctx, nullptr, 1, 12, d_enc, ctx->stream);
  if (rc != NRPHY_OK) {
    return rc;
  }
  HIP_TRY(hipMemcpyAsync(encoded, d_enc, 864, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return NRPHY_OK;
}
