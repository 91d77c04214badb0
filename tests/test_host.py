"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol, host-side logic (validator,
derivation, TBS, OFDM sizes) agrees with the oracle, the product fails loudly without a GPU, and the multi-GPU
sharding/aggregation path works over gloo with world_size 2."""
import ctypes as C
import os
import sys
import re
import socket

import numpy as np
import pytest

import backends
import cases

abi = backends.abi
lib = backends.pkg.lib


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(backends.ROOT, "include", "mi355_nrphy.h")).read()
    declared = set(re.findall(r"\b(nrphy_[a-z0-9_]+)\s*\(", header))
    assert declared == set(abi.ABI_SYMBOLS), declared ^ set(abi.ABI_SYMBOLS)
    handle = lib.load()
    for name in declared:
        assert hasattr(handle, name), name


def test_pod_layout_matches_header():
    """ctypes mirror vs the C compiler's layout of the PODs (sizes computed by a tiny C program)."""
    import subprocess
    import tempfile
    src = r'''#include "mi355_nrphy.h"
#include <stdio.h>
#include <stddef.h>
int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(nrphy_pdsch_pdu_t),
 offsetof(nrphy_pdsch_pdu_t, prb_mask), offsetof(nrphy_pdsch_pdu_t, reserved), offsetof(nrphy_pdsch_pdu_t, precoding),
 sizeof(nrphy_re_pattern_t), sizeof(nrphy_pdsch_derived_t), sizeof(nrphy_ofdm_config_t), sizeof(nrphy_pdsch_encoder_cfg_t),
 sizeof(nrphy_ldpc_decoder_cfg_t), sizeof(nrphy_ldpc_rate_dematcher_cfg_t), sizeof(nrphy_pusch_decoder_cfg_t),
 sizeof(nrphy_csi_rs_cfg_t), offsetof(nrphy_csi_rs_cfg_t, amplitude), offsetof(nrphy_csi_rs_cfg_t, precoding),
 sizeof(nrphy_grid_re_t));return 0;}'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(backends.ROOT, "include"), os.path.join(d, "t.c"), "-o",
                        os.path.join(d, "t")], check=True, timeout=120)
        out = subprocess.run([os.path.join(d, "t")], check=True, capture_output=True, timeout=60).stdout.split()
    want = [C.sizeof(abi.PdschPdu), abi.PdschPdu.prb_mask.offset, abi.PdschPdu.reserved.offset,
            abi.PdschPdu.precoding.offset, C.sizeof(abi.RePattern), C.sizeof(abi.PdschDerived), C.sizeof(abi.OfdmConfig),
            C.sizeof(abi.PdschEncoderCfg), C.sizeof(abi.LdpcDecoderCfg), C.sizeof(abi.LdpcRateDematcherCfg),
            C.sizeof(abi.PuschDecoderCfg), C.sizeof(abi.CsiRsCfg), abi.CsiRsCfg.amplitude.offset,
            abi.CsiRsCfg.precoding.offset, C.sizeof(abi.GridRe)]
    assert [int(x) for x in out] == want


def test_dl_control_pod_layout_and_validators(oracle):
    """PDCCH / SS/PBCH PODs: ctypes mirror vs the C compiler's layout; the library's validators (host only, no device)
    against the oracle's on random and on broken PDUs."""
    import subprocess
    import tempfile
    src = r'''#include "mi355_nrphy.h"
#include <stdio.h>
#include <stddef.h>
int main(void){printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(nrphy_pdcch_pdu_t), offsetof(nrphy_pdcch_pdu_t, frequency_resources),
 offsetof(nrphy_pdcch_pdu_t, payload), offsetof(nrphy_pdcch_pdu_t, precoding), sizeof(nrphy_ssb_pdu_t),
 offsetof(nrphy_ssb_pdu_t, bch_payload), offsetof(nrphy_ssb_pdu_t, ports));return 0;}'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(backends.ROOT, "include"), os.path.join(d, "t.c"), "-o",
                        os.path.join(d, "t")], check=True, timeout=120)
        out = subprocess.run([os.path.join(d, "t")], check=True, capture_output=True, timeout=60).stdout.split()
    want = [C.sizeof(abi.PdcchPdu), abi.PdcchPdu.frequency_resources.offset, abi.PdcchPdu.payload.offset,
            abi.PdcchPdu.precoding.offset, C.sizeof(abi.SsbPdu), abi.SsbPdu.bch_payload.offset, abi.SsbPdu.ports.offset]
    assert [int(x) for x in out] == want
    h = lib.load()
    rng = np.random.default_rng(11)
    for _ in range(200):
        pdu = cases.random_pdcch(rng)
        field = str(rng.choice(["none", "cce_index", "aggregation_level", "duration", "payload_size", "prg_size_rb", "nof_prg",
                                "reg_bundle_size", "interleaver_size", "start_symbol_index", "frequency_resources"]))
        if field != "none":
            setattr(pdu, field, int(rng.integers(0, 20)))
        assert h.nrphy_pdcch_validate(C.byref(pdu)) == oracle.pdcch_validate(pdu), field
        ssb = cases.random_ssb(rng, nof_ports=4)
        field = str(rng.choice(["none", "slot_index", "ssb_idx", "L_max", "pattern_case", "subcarrier_offset", "common_scs",
                                "numerology"]))
        if field != "none":
            setattr(ssb, field, int(rng.integers(0, 9)))
        assert h.nrphy_ssb_validate(C.byref(ssb)) == oracle.ssb_validate(ssb), field


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lib.NrphyError) as e:
        lib.Context(0)
    assert e.value.status == abi.ERR_DEVICE


def test_host_derivation_matches_oracle(oracle):
    import test_gpu_parity
    rng = np.random.default_rng(1)
    pdus = [cases.baseline_config(c)[0] for c in (1, 2, 3)] + cases.unit_test_like_pdus(rng) + cases.mixed_cell(1)[0]
    pdus += [p for _, p, _, _ in test_gpu_parity.edge_case_pdus()]
    for pdu in pdus:
        assert lib.validate(pdu) == oracle.validate(pdu) == 0
        assert lib.derive(pdu) == oracle.derive(pdu)
    for args in [(12, 36, 0, 8, 948, 4, 270), (12, 36, 0, 2, 120, 1, 52), (14, 12, 0, 4, 378, 3, 17), (2, 6, 6, 2, 30, 1, 1),
                 (13, 24, 12, 6, 666, 2, 273)]:
        assert lib.tbs_calculate(*args) == oracle.tbs(*args)
    for mu, n in ((0, 1024), (0, 2048), (1, 4096), (2, 512), (3, 256)):
        cfg = abi.OfdmConfig(mu, 10, n, 0, 1.0, 0.0)
        for sym in range(14 << mu):
            assert lib.symbol_size(cfg, sym) == oracle._f("ofdm_symbol_size")(cfg, sym)
        for slot in range(1 << mu):
            assert lib.slot_size(cfg, slot) == oracle._f("ofdm_slot_size")(cfg, slot)
    # 30 kHz, N = 4096: CP 352 on symbols 0 and 14 of the subframe, 288 elsewhere; 61440 samples per slot.
    cfg = abi.OfdmConfig(1, 273, 4096, 0, 1.0, 0.0)
    assert [lib.symbol_size(cfg, s) - 4096 for s in (0, 1, 13, 14, 15)] == [352, 288, 288, 352, 288]
    assert lib.slot_size(cfg, 0) == lib.slot_size(cfg, 1) == 61440


def test_validator_rules():
    """Each rule of pdsch_processor_validator_impl::is_valid (pdsch_processor_validator_test.cpp)."""
    base = dict(bwp_start_rb=1, bwp_size_rb=25, qm=4, dmrs_symbols=(2, 7), prb_start=3, prb_count=10, start_symbol=2,
                nof_symbols=10, tb_size_bytes=100, precoding=abi.identity_precoding(2))
    assert lib.validate(abi.make_pdu(**base)) == abi.OK
    bad = [dict(dmrs_symbols=(1,)), dict(dmrs_symbols=(12,)), dict(dmrs_type=2), dict(start_symbol=6, dmrs_symbols=(7,)),
           dict(tbs_lbrm_bytes=0), dict(prb_start=20), dict(nof_cdm_groups_without_data=3), dict(nof_codewords=2),
           dict(vrb_contiguous=0), dict(precoding=np.zeros((1, 1, 2, 2), np.float32)),  # more layers than ports
           dict(reserved=[(range(0, 26), [1] * 12, [0, 0, 1] + [0] * 11)]), dict(qm=3), dict(rv=4), dict(base_graph=3),
           dict(tb_size_bytes=0)]
    for change in bad:
        assert lib.validate(abi.make_pdu(**dict(base, **change))) == abi.ERR_INVALID_PDU, change


# ---- multi-GPU path over gloo ------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total_slots, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sharding = backends.load_package().sharding
    first, n = sharding.shard_slots(total_slots, rank, world)
    dist.barrier()
    slots, samples, seconds = sharding.aggregate(dist, torch.device("cpu"), n, n * 245760, 0.5 + 0.25 * rank)
    out.put((rank, first, n, slots, samples, seconds))
    dist.barrier()
    dist.destroy_process_group()


def test_slot_sharding_and_aggregation_gloo_world2():
    import torch.multiprocessing as mp
    sharding = backends.load_package().sharding
    for total, world in ((10, 3), (7, 8), (256, 4)):
        parts = [sharding.shard_slots(total, r, world) for r in range(world)]
        assert sum(n for _, n in parts) == total
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(world - 1))
    assert sorted({sharding.cell_affine_rank(c, s, 8, 4) for c in range(4) for s in range(2)}) == list(range(8))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 257, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[1:3] for r in res] == [(0, 129), (129, 128)]
    for r in res:
        assert r[3] == 257 and r[4] == 257 * 245760 and abs(r[5] - 0.75) < 1e-9


def _bench_line(argv, env_extra=None, timeout=300):
    """Runs bench.py with `argv` as a child process; returns (exit code, the lines of stdout that are JSON objects, stderr)."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(backends.ROOT, "bench.py")] + argv, cwd=backends.ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    return p.returncode, [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")], p.stderr


def test_bench_gpus_n_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` with no launcher around it starts the two ranks itself (a child torch.distributed.run; the
    dry mode runs the process group, sharding, totals and per-rank gather over gloo without device work) and prints ONE line
    that says n_gpus 2 -- the obvious multi-GPU command must not silently measure one rank."""
    rc, lines, err = _bench_line(["--gpus", "2", "--dry-run", "--steps", "3", "--slots", "100"])
    assert rc == 0, err[-3000:]
    assert len(lines) == 1, lines
    out = lines[0]
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["scaling"] == "weak"
    assert out["per_rank"]["slots_per_step"] == [100, 100] and out["total_slots"] == 2 * 100 * 3
    assert out["collective_backend"].startswith("gloo")
    # config 4 is placed by cell affinity; three ranks is one of the world sizes that leaves the shares uneven
    rc, lines, err = _bench_line(["--gpus", "3", "--dry-run", "--steps", "1", "--slots", "8", "--config", "4"])
    assert rc == 0, err[-3000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 3
    assert sum(lines[0]["per_rank"]["slots_per_step"]) == 24 and lines[0]["total_slots"] == 24


def test_bench_eight_ranks_config4_cell_affine_dry_run():
    """The driver's largest command, rehearsed without devices: `bench.py --gpus 8 --config 4` places BASELINE config 4's stream
    of cell-slots by cell affinity (cell c on ranks 2c, 2c + 1 alternating slots) -- every one of the eight ranks gets exactly
    its 1024 cell-slots per step, and rank 0's line adds them up."""
    rc, lines, err = _bench_line(["--gpus", "8", "--dry-run", "--config", "4", "--steps", "2"], timeout=600)
    assert rc == 0, err[-3000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 8 and lines[0]["dry_run"] is True
    assert lines[0]["per_rank"]["slots_per_step"] == [1024] * 8 and lines[0]["total_slots"] == 8 * 1024 * 2


def test_trace_ranges_follow_the_environment():
    """rocTX ranges named like the reference's trace points ("process_pdsch", "CB batch", ...: nrphy_trace.h) are live when
    NRPHY_TRACE=1 finds the rocTX library, off with NRPHY_TRACE=0 -- decided once per process, so each case is a child."""
    import subprocess
    code = ("import ctypes, sys; lib = ctypes.CDLL(sys.argv[1]); lib.nrphy_trace_enabled.restype = ctypes.c_int; "
            "print(lib.nrphy_trace_enabled())")
    for value, want in (("1", "1"), ("0", "0")):
        env = dict(os.environ, NRPHY_TRACE=value)
        r = subprocess.run([sys.executable, "-c", code, backends.pkg.lib.LIB_PATH], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and r.stdout.strip() == want, (value, r.stdout, r.stderr[-500:])


def test_bench_refuses_a_world_size_other_than_gpus():
    rc, lines, err = _bench_line(["--gpus", "2", "--dry-run"], {"RANK": "0", "WORLD_SIZE": "3", "LOCAL_RANK": "0"})
    assert rc == 2 and not lines and "WORLD_SIZE=3" in err
    rc, lines, err = _bench_line(["--gpus", "1", "--dry-run"])   # one rank: no launcher, no process group
    assert rc == 0 and len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["collective_backend"].startswith("none")


def test_adaptors_compile_against_reference_headers(tmp_path):
    """The srsRAN-side adaptors (seams A, B, C) must compile against the reference's own interface headers."""
    import subprocess
    ref = "/root/reference/srsRAN-5G-ER"
    if not os.path.isdir(ref):
        pytest.skip("reference headers not available on this machine")
    src = tmp_path / "check.cpp"
    src.write_text('#include "mi355_nrphy_srsran.h"\nint main() { return 0; }\n')
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-w", "-I", os.path.join(backends.ROOT, "include"),
           "-I", os.path.join(backends.PKG_DIR, "adaptors"), "-I", ref + "/include", "-I", ref + "/external/fmt/include",
           "-I", ref + "/external", str(src)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


def test_reference_unit_test_configurations_host_side():
    """The product's validator and derivation (host logic, no GPU) on every PDU of tests/golden/ref_test_configs.npz: the 24
    pdsch_processor_test_data.h PDUs, the 36 pdsch_modulator_test_data.h and 96 type-1 dmrs_pdsch_processor_test_data.h
    configurations are accepted and derive like the oracle; the 96 type-2 ones are refused; the 11
    ldpc_segmenter_test_data.h known answers come out of nrphy_pdsch_derive."""
    import os
    g = np.load(os.path.join(cases.GOLDEN, "ref_test_configs.npz"))
    o = backends.oracle()
    lib = backends.pkg.lib
    keys = ["proc_%d" % i for i in range(24)] + ["mod_%d" % i for i in range(36)]
    for i in range(192):
        pdu = cases.pdsch_pdu_from_fixture(g, "dmrs_%d" % i)
        if g["dmrs_info"][i, 2]:
            keys.append("dmrs_%d" % i)
        else:
            assert lib.validate(pdu) != 0
    for key in keys:
        pdu = cases.pdsch_pdu_from_fixture(g, key)
        assert lib.validate(pdu) == 0, key
        assert lib.derive(pdu) == o.derive(pdu), key
    for tbs_bits, bg, nof_segments, segment_length in g["seg_cases"].tolist():
        d = lib.derive(backends.abi.make_pdu(base_graph=bg, tb_size_bytes=tbs_bits // 8, prb_count=52, qm=2))
        assert (d["nof_codeblocks"], d["segment_length"]) == (nof_segments, segment_length)


def test_reference_unit_test_configurations_round2_host_side():
    """The product's PDCCH / SS-PBCH / NZP-CSI-RS validators (host logic) on every configuration of
    tests/golden/ref_test_configs2.npz: accepted or refused exactly as the fixture records and as the oracle decides."""
    import ctypes as C
    import os
    g = np.load(os.path.join(cases.GOLDEN, "ref_test_configs2.npz"))
    o, abi = backends.oracle(), backends.abi
    h = backends.pkg.lib.load()
    for kind, cls, n, product, oracle in (("pdcch", abi.PdcchPdu, 114, h.nrphy_pdcch_validate, o.pdcch_validate),
                                          ("ssb", abi.SsbPdu, 240, h.nrphy_ssb_validate, o.ssb_validate),
                                          ("csi", abi.CsiRsCfg, 102, h.nrphy_csi_rs_validate, o.csi_rs_validate)):
        flags = g[kind + "_valid"]
        assert flags.shape == (n,)
        for i in range(n):
            obj = cases.struct_from_fixture(cls, g, "%s_%d" % (kind, i))
            assert (product(C.byref(obj)) == 0) == bool(flags[i]) == (oracle(obj) == 0), (kind, i)
    for i in range(232):
        pdu = cases.pbch_message_pdu(cases.struct_from_fixture(abi.SsbPdu, g, "pb_%d" % i))
        assert h.nrphy_ssb_validate(C.byref(pdu)) == 0, i


def test_validators_on_hostile_descriptors():
    """The validators are the gate in front of host code that sizes reads and loops by descriptor fields: they must refuse, never
    crash on, out-of-range values.  Regression cases found by profiles/fuzz_validators_cpu.py (a CCE interleaver size of 2^31 made
    L * R wrap to zero: division by zero in the product AND the oracle; a PRG count of 2^31 would have sized the read of the weight
    array; a hole in a "contiguous" allocation, which the reference's validator refuses), then a short run of that fuzz."""
    import ctypes as C
    import os
    import sys
    o, abi = backends.oracle(), backends.abi
    lib = backends.pkg.lib
    h = lib.load()
    rng = np.random.default_rng(31)
    pdcch = cases.random_pdcch(rng)
    assert h.nrphy_pdcch_validate(C.byref(pdcch)) == 0 and o.pdcch_validate(pdcch) == 0
    for field, value in (("interleaver_size", 1 << 31), ("reg_bundle_size", 1 << 31), ("cce_index", 1 << 31), ("cce_index", (1 << 32) - 1),
                         ("nof_prg", 1 << 31), ("prg_size_rb", 1 << 31), ("aggregation_level", 1 << 31)):
        bad = type(pdcch).from_buffer_copy(bytes(pdcch))
        bad.cce_to_reg_mapping = 2
        setattr(bad, field, value)
        assert h.nrphy_pdcch_validate(C.byref(bad)) != 0 and o.pdcch_validate(bad) != 0, field
    pdu = cases.baseline_config(2)[0]
    assert lib.validate(pdu) == 0 and o.validate(pdu) == 0
    for field, value in (("nof_prg", 1 << 31), ("nof_prg", 70), ("cp", 5), ("prg_size_rb", 0),
                         ("tb_size_bytes", (1 << 29) + 1000), ("tb_size_bytes", 162 * 1056 + 1)):   # (8 x the first wraps to 8000 bits)
        bad = type(pdu).from_buffer_copy(bytes(pdu))
        setattr(bad, field, value)
        assert lib.validate(bad) != 0 and o.validate(bad) != 0, field
    bad = type(pdu).from_buffer_copy(bytes(pdu))
    bad.prb_mask[0] &= ~(1 << 40)          # a hole in the allocation while vrb_contiguous says contiguous
    assert lib.validate(bad) != 0 and o.validate(bad) != 0
    r = backends.ref()
    if r is not None:
        assert r.validate(bad) != 0
    sys.path.insert(0, os.path.join(cases.ROOT if hasattr(cases, "ROOT") else os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles"))
    import fuzz_validators_cpu
    for kind, (n, refused, disagreements, ref_refuses) in fuzz_validators_cpu.run(300, base=12345, verbose=False).items():
        assert n >= 300 and refused > 0 and disagreements == 0 and ref_refuses == 0, (kind, n, refused, disagreements, ref_refuses)


def test_host_only_helpers_of_the_lower_phy_tail():
    """Entry points that are pure host arithmetic: nrphy_amplitude_metrics (amplitude_controller_clipping_impl's metrics from
    the device's raw measurements, running counters included) against the compiled reference where it is built and against the
    oracle; nrphy_ofh_compressed_prb_bytes; OFDM symbol / slot sizes against the oracle's for every numerology and prefix."""
    import ctypes as C
    lib, o, abi = backends.pkg.lib, backends.oracle(), backends.abi
    h = lib.load()
    r = backends.ref()
    rng = np.random.default_rng(99)
    running = abi.AmplitudeMetrics()
    total = clipped = 0
    for cfg in (abi.AmplitudeCfg(0, 1, -2.0, 1.0, -9.0), abi.AmplitudeCfg(0, 0, 3.0, 2.0, -3.0), abi.AmplitudeCfg(0, 1, -20.0, 1.0, -1.0)):
        x = ((rng.standard_normal(3000) + 1j * rng.standard_normal(3000)) * 0.4).astype(np.complex64)
        _, om = o.amplitude_control(cfg, x)
        m = abi.AmplitudeMetrics()
        assert h.nrphy_amplitude_metrics(C.byref(cfg), C.byref(om["stats"]), C.byref(m)) == 0
        for key in ("avg_power_fs", "peak_power_fs", "papr_lin", "gain_dB"):
            assert getattr(m, key) == om[key], key
        if r is not None:
            _, rm = r.amplitude_control(cfg, x)
            assert abs(m.avg_power_fs - rm["avg_power_fs"]) <= 1e-5 * abs(rm["avg_power_fs"])
            assert m.peak_power_fs == np.float32(rm["peak_power_fs"]) and m.gain_dB == np.float32(rm["gain_dB"])
            assert int(m.nof_clipped_samples) == rm["nof_clipped"] and int(m.nof_processed_samples) == rm["nof_processed"]
        # one controller object over several buffers: the counters accumulate -- while clipping is enabled, as in the
        # reference (amplitude_controller_clipping_impl.cpp:37-66 measures and counts inside `if (clipping_enabled)`)
        assert h.nrphy_amplitude_metrics(C.byref(cfg), C.byref(om["stats"]), C.byref(running)) == 0
        total += 3000 if cfg.enable_clipping else 0
        clipped += int(m.nof_clipped_samples)
        assert int(running.nof_processed_samples) == total and int(running.nof_clipped_samples) == clipped
    for typ, width in ((0, 8), (0, 9), (0, 16), (1, 8), (1, 9), (1, 14)):
        cfg = abi.OfhCompressionCfg(typ, width, 1.0)
        assert h.nrphy_ofh_compressed_prb_bytes(C.byref(cfg)) == o.lib.oracle_ofh_compressed_prb_bytes(C.byref(cfg)) == 3 * width + typ
    for mu in range(5):
        for cp in ((0, 1) if mu == 2 else (0,)):
            cfg = abi.OfdmConfig(mu, 24, 512, cp, 1.0, 3.5e9)
            nsym = 12 if cp else 14
            sizes = [lib.symbol_size(cfg, s) for s in range(nsym << mu)]
            assert all(s >= 512 for s in sizes)
            for slot in range(1 << mu):
                assert lib.slot_size(cfg, slot) == sum(sizes[slot * nsym:(slot + 1) * nsym])
            # one subframe is 1 ms: 15 kHz * 512 samples * 2^mu per ms
            assert sum(sizes) == 15 * 512 * (1 << mu)


def test_shipped_code_objects_spill_no_vector_registers_and_use_no_scratch():
    """What the library really contains (profiles/disasm_lib.py: .hip_fatbin -> bundle entries -> the code objects' metadata), not
    what a separate compile reports: every kernel without vector-register spills and without scratch memory; the headline kernels
    within the register budgets DESIGN.md states (eight codeblock waves per SIMD need <= 96 scalar and <= 64 vector registers)."""
    sys.path.insert(0, os.path.join(backends.ROOT, "profiles"))
    import disasm_lib
    rows = disasm_lib.resources(os.path.join(backends.ROOT, "srsran-edgeric-5g_amd", "csrc", "libmi355nrphy.so"))
    assert len(rows) > 100
    by_name = {r[0]: r for r in rows}
    for name, sgpr, vgpr, sgpr_spill, vgpr_spill, scratch, lds in rows:
        assert vgpr_spill == "0" and scratch == "0", (name, vgpr_spill, scratch)
    cb = by_name["void nrphy::codeblock_kernel_t<8, 4>"]
    assert int(cb[1]) <= 96 and int(cb[2]) <= 64 and cb[3] == "0"
    assert int(by_name["void nrphy::ofdm_kernel<4096, 1, false>"][2]) <= 168      # three workgroups per CU
    assert int(by_name["nrphy::ldpc_decode_msg_bg2_slot_kernel"][2]) <= 128        # four waves per SIMD


def test_no_source_is_built_with_floating_point_contraction():
    """Under -ffp-contract=fast the backend fuses across `#pragma clang fp contract(off)` (round 4: the wire-format sink's exact
    rounding): every source is built with contraction off, and where a kernel wants a fused multiply-add it says so."""
    build = open(os.path.join(backends.ROOT, "srsran-edgeric-5g_amd", "build.py")).read()
    assert "CONTRACT = {}" in build and '"-ffp-contract=off"' in build and "contract=fast\"" not in build
    for script in ("make_variant.sh", "kernel_resources.sh"):
        assert "contract=fast" not in open(os.path.join(backends.ROOT, "profiles", script)).read()
