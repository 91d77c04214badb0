/* TEST INFRASTRUCTURE -- the CPU oracle of the PDSCH + OFDM hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (srsran-edgeric-5g_amd/) never includes, links or calls anything under oracle/.
 *
 * Parity is PINNED: every function below is checked against the compiled reference
 * (oracle/_ref/libsrsref.so, built from /root/reference by oracle/Makefile) in tests/test_oracle_vs_ref.py
 * and against the golden vectors generated from it under tests/golden/ (tests/golden/generate.py).
 */
#ifndef NRPHY_ORACLE_H
#define NRPHY_ORACLE_H

#include "mi355_nrphy.h"

#ifdef __cplusplus
extern "C" {
#endif

int      oracle_pdsch_validate(const nrphy_pdsch_pdu_t* pdu);
int      oracle_pdsch_derive(const nrphy_pdsch_pdu_t* pdu, nrphy_pdsch_derived_t* out);
uint32_t oracle_tbs_calculate(uint32_t nof_symb_sh, uint32_t nof_dmrs_prb, uint32_t nof_oh_prb, uint32_t qm,
                              float target_code_rate, uint32_t nof_layers, uint32_t n_prb);
/* poly: 16 (CRC16), 0x24A (CRC24A), 0x24B (CRC24B). */
uint32_t oracle_crc(uint32_t poly, const uint8_t* data, uint32_t nbytes);
uint32_t oracle_crc_bits(uint32_t poly, const uint8_t* data, uint32_t nbits);
/* Segments a TB: C segments of K bits packed at stride_bytes; meta[5*i..] = {rm_length, cw_offset, filler,
 * full_length, nof_crc_bits}; returns C. */
int oracle_ldpc_segment(uint32_t bg, uint32_t rv, uint32_t qm, uint32_t nref, uint32_t nof_layers,
                        uint32_t nof_ch_symbols, const uint8_t* tb, uint32_t tb_bytes, uint8_t* segments,
                        uint32_t stride_bytes, uint32_t* meta, uint32_t* lifting_size);
int oracle_ldpc_encode(uint32_t bg, uint32_t zc, const uint8_t* msg, uint32_t out_bits, uint8_t* out);
int oracle_ldpc_rate_match(uint32_t bg, uint32_t zc, uint32_t rv, uint32_t qm, uint32_t nref, uint32_t nof_filler,
                           const uint8_t* in, uint32_t in_bits, uint8_t* out, uint32_t rm_length);
void  oracle_prg_apply_xor(uint32_t c_init, uint32_t offset, uint8_t* data, uint32_t nbits);
void  oracle_prg_apply_xor_llr(uint32_t c_init, uint32_t offset, const int8_t* in, int8_t* out, uint32_t n);
void  oracle_prg_generate_float(uint32_t c_init, uint32_t offset, float value, float* out, uint32_t n);
float oracle_modulate_ci8(uint32_t qm, const uint8_t* bits, uint32_t nsym, int8_t* out);
/* pdsch_encoder::encode -> packed rate-matched codeword (codeword_bits bits). */
int oracle_pdsch_encode(const nrphy_pdsch_pdu_t* pdu, const uint8_t* tb, uint8_t* cw_rm);
/* pdsch_processor::process into grid [nof_ports][14][nof_subc] cbf16 (only mapped REs are written).
 * cw_rm / cw_scrambled may be NULL. */
int oracle_pdsch_process(const nrphy_pdsch_pdu_t* pdu, const uint8_t* tb, uint16_t* grid, uint32_t nof_ports,
                         uint32_t nof_subc, uint8_t* cw_rm, uint8_t* cw_scrambled);
int      oracle_dft(uint32_t n, int inverse, const float* in, float* out);
/* demodulation_mapper::demodulate_soft (nrphy_oracle_rx.c); modulation = NRPHY_MOD_*. */
int  oracle_demodulate_soft(uint32_t modulation, size_t n, const float* symbols, const float* noise_vars, int8_t* llr);
void oracle_demod_tables(unsigned qm, unsigned pair, float* width, unsigned* n, float* slope, float* intercept);
/* LDPC decoder ("next" row, receive side): returns the iteration count (>= 1) when the CRC passed, 0 otherwise. */
int oracle_ldpc_decode(uint32_t bg, uint32_t zc, uint32_t nof_filler, uint32_t crc_poly_id, uint32_t max_iterations,
                       float scaling_factor, const int8_t* llr, uint32_t nof_llr, uint8_t* message_bits);
/* NZP-CSI-RS generator ("next" row, section 8f-2): writes the signal's RE into grid [nof_ports][14][nof_subc] cbf16. */
int oracle_csi_rs_validate(const nrphy_csi_rs_cfg_t* cfg);
int oracle_csi_rs_map(const nrphy_csi_rs_cfg_t* cfg, uint16_t* grid, uint32_t nof_ports, uint32_t nof_subc);
/* Downlink control side (section 8f-2), oracle/nrphy_oracle_dl.c: polar code construction (returns N, fills the mask of
 * the K information positions), PDCCH encoder / processor, PBCH encoder / SS/PBCH block processor.  Bit arrays are one
 * bit per byte; grids [nof_ports][14][nof_subc] cbf16, only the channel's RE are written. */
int oracle_polar_code(uint32_t K, uint32_t E, uint32_t n_max, uint8_t* k_set_mask);
int oracle_pdcch_encode(const uint8_t* payload, uint32_t payload_size, uint32_t rnti, uint32_t rm_length, uint8_t* encoded);
int oracle_pdcch_validate(const nrphy_pdcch_pdu_t* pdu);
int oracle_pdcch_process(const nrphy_pdcch_pdu_t* pdu, uint16_t* grid, uint32_t nof_ports, uint32_t nof_subc);
int oracle_ssb_validate(const nrphy_ssb_pdu_t* pdu);
int oracle_pbch_encode(const nrphy_ssb_pdu_t* pdu, uint8_t* encoded);
int oracle_ssb_process(const nrphy_ssb_pdu_t* pdu, uint16_t* grid, uint32_t nof_ports, uint32_t nof_subc);
/* Lower-PHY tail (section 8f-3), oracle/nrphy_oracle_lower.c. */
int oracle_amplitude_control(const nrphy_amplitude_cfg_t* cfg, const float* in, uint32_t nof_samples, float* out,
                             nrphy_amplitude_stats_t* stats);
int oracle_amplitude_metrics(const nrphy_amplitude_cfg_t* cfg, const nrphy_amplitude_stats_t* stats,
                             nrphy_amplitude_metrics_t* metrics);
int oracle_iq_convert_ci16(const float* in, uint32_t nof_samples, float scale, int16_t* out);
uint32_t oracle_ofh_compressed_prb_bytes(const nrphy_ofh_compression_cfg_t* cfg);
int oracle_ofh_compress(const nrphy_ofh_compression_cfg_t* cfg, const uint16_t* prbs, uint32_t nof_prb, uint8_t* out);
/* LDPC rate dematcher ("next" row): out = soft buffer of (66 or 50) * Zc LLRs, read and written. */
int oracle_ldpc_rate_dematch(uint32_t bg, uint32_t zc, uint32_t rv, uint32_t qm, uint32_t nref, uint32_t nof_filler,
                             int new_data, const int8_t* in, uint32_t e, int8_t* out);
uint32_t oracle_ofdm_symbol_size(const nrphy_ofdm_config_t* cfg, uint32_t symbol_index);
uint32_t oracle_ofdm_slot_size(const nrphy_ofdm_config_t* cfg, uint32_t slot_index);
int oracle_ofdm_demodulate_slot(const nrphy_ofdm_config_t* cfg, const float* iq, uint32_t nof_ports,
                                uint32_t slot_index, uint32_t window_offset, uint16_t* grid);
int oracle_ofdm_modulate_slot(const nrphy_ofdm_config_t* cfg, const uint16_t* grid, uint32_t nof_ports,
                              uint32_t slot_index, float* iq);
/* CPU baseline: threads workers x reps slots (PDSCH, + OFDM when ofdm != NULL); returns seconds. */
double oracle_bench(const nrphy_pdsch_pdu_t* pdu, const uint8_t* tb, uint32_t nof_ports, uint32_t nof_subc,
                    const nrphy_ofdm_config_t* ofdm, uint32_t threads, uint32_t reps);

#ifdef __cplusplus
}
#endif
#endif
