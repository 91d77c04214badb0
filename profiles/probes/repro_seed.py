"""Repro aid for a seeded-sweep mismatch (profiles/fuzz_sweep.py leg `pdsch`, base 0): re-runs the leg's PDUs one by one on the
device, and for a PDU whose grid or codeword taps differ from the oracle's prints its shape and which codeblocks / resource
elements differ.  It found the seed walk bug of round 3 (an item whose 31-word scrambling seed ran past the end of a sequence
part hid the item behind it; tests/test_gpu_parity.py::test_pdsch_overlapping_scrambling_seeds).
Usage (GPU box, repository root): python3 profiles/probes/repro_seed.py"""
import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import backends, cases
abi, lib = backends.abi, backends.pkg.lib
o = backends.oracle()
ctx = lib.Context(0)
BASE = 0
for seed in range(6):
    rng = np.random.default_rng(BASE + 1000 + seed)
    for idx, (pdu, P, S) in enumerate(cases.random_pdus(o.tbs, rng, 80)):
        if o.validate(pdu) != 0 or o.derive(pdu)["nof_re"] == 0:
            continue
        tb = cases.random_tb(rng, pdu)
        d = o.derive(pdu)
        want, orm, oscr = o.pdsch_process(pdu, tb, P, S, taps=True, codeword_bits=d["codeword_bits"])
        rng.integers(0, 3)
        got, rm, scr = ctx.pdsch_process_host(pdu, tb, P, S, taps=True)
        if not (np.array_equal(got, want) and np.array_equal(rm, orm) and np.array_equal(scr, oscr)):
            lq = pdu.qm * pdu.nof_layers
            print("MISMATCH seed", seed, "idx", idx, "qm", pdu.qm, "layers", pdu.nof_layers, "ports", P, "tb", pdu.tb_size_bytes,
                  {k: d[k] for k in d if k in ("nof_codeblocks", "codeword_bits", "nof_short_segments", "rm_length_short", "rm_length_long", "lifting_size", "nof_re")})
            print("  grid equal", np.array_equal(got, want), "rm equal", np.array_equal(rm, orm), "scr equal", np.array_equal(scr, oscr))
            a = np.unpackbits(scr); b = np.unpackbits(oscr)
            bad = np.nonzero(a != b)[0]
            print("  scrambled bits differing:", len(bad), "first", bad[:5], "last", bad[-5:], "of", len(a))
            # per codeblock
            es, el, ns = d["rm_length_short"], d["rm_length_long"], d["nof_short_segments"]
            C = d["nof_codeblocks"]
            off = 0
            for cb in range(C):
                e = es if cb < ns else el
                nb = np.count_nonzero((bad >= off) & (bad < off + e))
                if nb:
                    inb = bad[(bad >= off) & (bad < off + e)] - off
                    print("   cb", cb, "E", e, "bit offset", off, "word", off // 32, "bad", nb, "first bad bit in cb", inb[0], "-> RE", inb[0] // lq, "last", inb[-1], "nre", e // lq)
                off += e
print("done")
