#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the COMPILED REFERENCE (oracle/_ref/libsrsref.so, built by
`make -C oracle ref` from /root/reference).  Run in the build container only; the outputs (small .npz files: inputs +
expected outputs, never reference source) are committed and travel to the GPU box.

    python tests/golden/generate.py [section ...]     (no argument: every section)

Sections: base (the round-1 files: codebooks, ldpc_encoder, pdsch_processor, ofdm_modulator, ofdm_demodulator),
ofdm_sizes (DFT sizes 4608 / 6144: ofdm_sizes.npz), dl_control (PDCCH and SS/PBCH block processors: dl_control.npz).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import backends  # noqa: E402

abi = backends.abi
r = backends.ref()
assert r is not None, "build oracle/_ref first: make -C oracle ref"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


SECTIONS = sys.argv[1:] or ["base", "ofdm_sizes", "dl_control"]


def section_ofdm_sizes():
    """OFDM modulator / demodulator at the two DFT sizes added in round 2 (15 kHz SCS: 6144 = 92.16 Msps, the size the
    reference's ofdm_modulator_unittest sweeps; 4608 = 69.12 Msps)."""
    rng = np.random.default_rng(6144)
    og = {}
    for name, (mu, bw, n, fc, slot) in {"n6144": (0, 273, 6144, 3.5e9, 0), "n4608": (0, 270, 4608, 2.4e9, 0)}.items():
        cfg = abi.OfdmConfig(mu, bw, n, 0, 1.0 / np.sqrt(n), fc)
        grid = (rng.standard_normal((1, 14, bw * 12, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        og[name + "_grid"] = grid
        og[name + "_iq"] = r.ofdm_slot(cfg, grid, slot)
        og[name + "_cfg"] = np.array([mu, bw, n, fc, slot], dtype=np.float64)
    for name, (mu, bw, n, fc, slot, wo) in {"d6144": (0, 273, 6144, 3.5e9, 0, 0), "d4608w": (0, 270, 4608, 2.4e9, 0, 11)}.items():
        cfg = abi.OfdmConfig(mu, bw, n, 0, 1.0 / np.sqrt(n), fc)
        size = backends.pkg.lib.slot_size(cfg, slot)
        iq = (rng.standard_normal((1, size)) + 1j * rng.standard_normal((1, size))).astype(np.complex64)
        og[name + "_iq"] = iq
        og[name + "_grid"] = r.ofdm_demod_slot(cfg, iq, slot, wo)
        og[name + "_cfg"] = np.array([mu, bw, n, fc, slot, wo], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "ofdm_sizes.npz"), **og)


def section_dl_control():
    """PDCCH and SS/PBCH block processors of the compiled reference on seeded PDUs (tests/cases.py generators; the PDUs
    are stored field by field, so the fixture does not depend on the generators staying the same): encoder outputs and
    the grid words the processor wrote into an all-zero grid."""
    import cases
    rng = np.random.default_rng(38213)
    g = {}
    n_pdcch, n_ssb = 24, 12
    for i in range(n_pdcch):
        pdu = cases.random_pdcch(rng)
        grid = r.pdcch_process(pdu, np.zeros((4, 14, 52 * 12, 2), np.uint16), simd=1)
        assert np.array_equal(grid, r.pdcch_process(pdu, np.zeros((4, 14, 52 * 12, 2), np.uint16), simd=0))
        g.update(cases.pdu_to_arrays(pdu, "pdcch%02d_" % i))
        payload = np.array(list(pdu.payload)[: pdu.payload_size], np.uint8)
        g["pdcch%02d_encoded" % i] = np.packbits(r.pdcch_encode(payload, pdu.rnti, 108 * pdu.aggregation_level))
        w = grid.view(np.uint32).reshape(-1)
        idx = np.flatnonzero(w)
        g["pdcch%02d_idx" % i] = idx.astype(np.uint32)
        g["pdcch%02d_val" % i] = w[idx]
    for i in range(n_ssb):
        pdu = cases.random_ssb(rng, nof_ports=3)
        grid = r.ssb_process(pdu, np.zeros((3, 14, 52 * 12, 2), np.uint16))
        g.update(cases.pdu_to_arrays(pdu, "ssb%02d_" % i))
        g["ssb%02d_encoded" % i] = np.packbits(r.pbch_encode(pdu))
        w = grid.view(np.uint32).reshape(-1)
        idx = np.flatnonzero(w)   # the SSS writes -0 imaginary parts: non-zero words, kept
        g["ssb%02d_idx" % i] = idx.astype(np.uint32)
        g["ssb%02d_val" % i] = w[idx]
    g["n_pdcch"], g["n_ssb"] = np.array(n_pdcch), np.array(n_ssb)
    np.savez_compressed(os.path.join(HERE, "dl_control.npz"), **g)


if "ofdm_sizes" in SECTIONS:
    section_ofdm_sizes()
if "dl_control" in SECTIONS:
    section_dl_control()
if "base" not in SECTIONS:
    print("golden vectors written to", HERE, SECTIONS)
    sys.exit(0)


# 1. Precoding codebooks the BASELINE configs use.
np.savez(os.path.join(HERE, "codebooks.npz"),
         single_port=r.codebook(1), two_layer_two_ports_0=r.codebook(3, 0),
         four_layer_four_ports_0_0=r.codebook(7, 0, 0), one_layer_two_ports_1=r.codebook(2, 1),
         three_layer_four_ports_1_0=r.codebook(6, 1, 0), two_layer_four_ports_1_0_1=r.codebook(5, 1, 0, 1))

import cases  # noqa: E402  (needs codebooks.npz)

# 2. LDPC encoder: every (base graph, lifting size), one random message, full-length output
#    (the configurations of ldpc_enc_dec_test.cpp: 51 lifting sizes x 2 base graphs).
rng = np.random.default_rng(38212)
msgs, outs, keys = [], [], []
for bg in (1, 2):
    kb = 22 if bg == 1 else 10
    for zc in [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 26, 28, 30, 32, 36, 40, 44, 48, 52,
               56, 60, 64, 72, 80, 88, 96, 104, 112, 120, 128, 144, 160, 176, 192, 208, 224, 240, 256, 288, 320, 352,
               384]:
        msg = np.packbits(rng.integers(0, 2, kb * zc, dtype=np.uint8))
        n = (66 if bg == 1 else 50) * zc
        out = r.ldpc_encode(bg, zc, msg, n, simd=1)
        assert np.array_equal(out, r.ldpc_encode(bg, zc, msg, n, simd=0))
        keys.append((bg, zc))
        msgs.append(msg)
        outs.append(out)
np.savez_compressed(os.path.join(HERE, "ldpc_encoder.npz"), keys=np.array(keys),
                    **{"msg_%d_%d" % k: m for k, m in zip(keys, msgs)},
                    **{"out_%d_%d" % k: o for k, o in zip(keys, outs)})

# 3. PDSCH processor: BASELINE configs 1-3 + unit-test-like PDUs.  Stored: seed for the TB, SHA-256 of the codeword
#    and of the grid, and the first/last 64 grid words of every port (small files; the TB is regenerated from the seed).
def pdu_fields(pdu):
    return {n: np.array(getattr(pdu, n)) for n, _ in abi.PdschPdu._fields_
            if n not in ("prb_mask", "reserved", "precoding")}


golden = {}
items = [("cfg%d" % c,) + cases.baseline_config(c)[:3] for c in (1, 2, 3)]
for i, pdu in enumerate(cases.unit_test_like_pdus(np.random.default_rng(2024))):
    items.append(("unit%02d" % i, pdu, 4, 26 * 12))
o = backends.oracle()
for name, pdu, nof_ports, nof_subc in items:
    tb = np.random.default_rng(abs(hash(name)) % 2**32 if False else sum(map(ord, name))).integers(
        0, 256, pdu.tb_size_bytes, dtype=np.uint8)
    d = o.derive(pdu)
    grid = r.pdsch_process(pdu, tb, nof_ports, nof_subc, simd=1)
    cw = r.pdsch_encode(pdu, tb, d)[: (d["codeword_bits"] + 7) // 8]
    g32 = grid.view(np.uint32).reshape(nof_ports, -1)
    nz = [np.flatnonzero(g32[p]) for p in range(nof_ports)]
    golden[name + "_tb_seed"] = np.array(sum(map(ord, name)))
    golden[name + "_cw_sha"] = np.array(sha(cw))
    golden[name + "_grid_sha"] = np.array(sha(grid))
    golden[name + "_cw_head"] = cw[:64].copy()
    for p in range(nof_ports):
        idx = nz[p][:64] if len(nz[p]) else np.zeros(0, np.int64)
        golden["%s_p%d_idx" % (name, p)] = idx
        golden["%s_p%d_val" % (name, p)] = g32[p][idx]
np.savez_compressed(os.path.join(HERE, "pdsch_processor.npz"), **golden)

# 4. OFDM modulator: random bf16 grid rows -> reference IQ (generic DFT; FFTW is absent here), sub-sampled.
rng = np.random.default_rng(38211)
og = {}
for name, (mu, bw, n, fc, slot) in {"n4096": (1, 273, 4096, 3.5e9, 1), "n2048": (0, 106, 2048, 2.4e9, 0),
                                    "n1024": (0, 52, 1024, 2.4e9, 0),
                                    # the 3 * 2^k sizes of the 23.04 MHz family of sampling rates
                                    "n1536": (0, 106, 1536, 2.4e9, 0), "n3072": (1, 217, 3072, 3.5e9, 0),
                                    "n768": (0, 52, 768, 2.4e9, 0), "n384": (0, 25, 384, 2.4e9, 0)}.items():
    cfg = abi.OfdmConfig(mu, bw, n, 0, 1.0 / np.sqrt(n), fc)
    grid = (rng.standard_normal((1, 14, bw * 12, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    iq = r.ofdm_slot(cfg, grid, slot)
    og[name + "_grid"] = grid
    og[name + "_iq"] = iq
    og[name + "_cfg"] = np.array([mu, bw, n, fc, slot], dtype=np.float64)
# extended cyclic prefix (12 symbols per slot, 60 kHz SCS: the numerology-2 rows of ofdm_modulator_test_data.h); own
# generator so that the vectors above do not change
rng = np.random.default_rng(38213)
for name, (mu, bw, n, fc, slot) in {"x512": (2, 24, 512, 3.5e9, 3), "x2048": (2, 96, 2048, 28e9, 0)}.items():
    cfg = abi.OfdmConfig(mu, bw, n, 1, 1.0 / np.sqrt(n), fc)
    grid = (rng.standard_normal((1, 14, bw * 12, 2)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    og[name + "_grid"] = grid
    og[name + "_iq"] = r.ofdm_slot(cfg, grid, slot)
    og[name + "_cfg"] = np.array([mu, bw, n, fc, slot, 1], dtype=np.float64)
np.savez_compressed(os.path.join(HERE, "ofdm_modulator.npz"), **og)

# 5. OFDM demodulator ("next" row, PUSCH receive side): random IQ -> reference grid (cbf16), with and without a DFT
#    window offset.
rng = np.random.default_rng(38212)
dg = {}
for name, (mu, bw, n, fc, slot, wo) in {"d4096": (1, 273, 4096, 3.5e9, 1, 0), "d2048w": (0, 106, 2048, 2.4e9, 0, 9),
                                        "d1536": (0, 106, 1536, 2.4e9, 0, 0), "d512w": (0, 25, 512, 2.4e9, 0, 3)}.items():
    cfg = abi.OfdmConfig(mu, bw, n, 0, 1.0 / np.sqrt(n), fc)
    size = backends.pkg.lib.slot_size(cfg, slot)
    iq = (rng.standard_normal((2, size)) + 1j * rng.standard_normal((2, size))).astype(np.complex64)
    dg[name + "_iq"] = iq
    dg[name + "_grid"] = r.ofdm_demod_slot(cfg, iq, slot, wo)
    dg[name + "_cfg"] = np.array([mu, bw, n, fc, slot, wo], dtype=np.float64)
rng = np.random.default_rng(38214)
for name, (mu, bw, n, fc, slot, wo) in {"dx1024w": (2, 48, 1024, 3.5e9, 1, 7)}.items():
    cfg = abi.OfdmConfig(mu, bw, n, 1, 1.0 / np.sqrt(n), fc)
    size = backends.pkg.lib.slot_size(cfg, slot)
    iq = (rng.standard_normal((2, size)) + 1j * rng.standard_normal((2, size))).astype(np.complex64)
    dg[name + "_iq"] = iq
    dg[name + "_grid"] = r.ofdm_demod_slot(cfg, iq, slot, wo)
    dg[name + "_cfg"] = np.array([mu, bw, n, fc, slot, wo, 1], dtype=np.float64)
np.savez_compressed(os.path.join(HERE, "ofdm_demodulator.npz"), **dg)
print("golden vectors written to", HERE)
