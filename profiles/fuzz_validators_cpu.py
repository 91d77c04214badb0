#!/usr/bin/env python3
"""Validator agreement on MUTATED descriptors (CPU only; the product's validators are host code): a random field of a valid random
PDU / configuration is overwritten with a random value, then  product == oracle  (all four descriptor types) and, for the PDSCH PDU,
oracle == the compiled reference's pdsch_processor_validator_impl.  Usage: python3 profiles/fuzz_validators_cpu.py [count]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import backends  # noqa: E402
import cases  # noqa: E402

BASE = int(os.environ.get("NRPHY_FUZZ_SEED", "0"))
abi = backends.abi
o, r = backends.oracle(), backends.ref()
h = backends.pkg.lib.load()


def scalar_fields(obj):
    out = []
    for name, typ in obj._fields_:
        if typ in (C.c_uint32, C.c_int32, C.c_uint16, C.c_uint8, C.c_uint64, C.c_float):
            out.append((name, typ, None))
        elif hasattr(typ, "_length_") and typ._type_ in (C.c_uint32, C.c_uint8, C.c_uint16, C.c_uint64):
            out.append((name, typ._type_, typ._length_))
    return out


def mutate(rng, obj):
    fields = scalar_fields(obj)
    name, typ, length = fields[int(rng.integers(0, len(fields)))]
    if typ is C.c_float:
        value = float(rng.choice([0.0, -1.0, 1e9, float("nan"), float(rng.uniform(-40, 40))]))
    else:
        cur = getattr(obj, name) if length is None else getattr(obj, name)[0]
        top = {C.c_uint8: 255, C.c_uint16: 65535}.get(typ, 1 << 31)
        value = int(rng.choice([0, 1, cur + 1, max(cur - 1, 0), 2 * cur + 3, int(rng.integers(0, 64)), int(rng.integers(0, 100000)), top]))
        value = min(value, top if typ is not C.c_uint64 else (1 << 63))
    if length is None:
        setattr(obj, name, value)
    else:
        getattr(obj, name)[int(rng.integers(0, length))] = value
    return name, (value if typ is not C.c_float else 0)


def run(count, base=BASE, verbose=True):
    """{kind: (descriptors, refused, product/oracle disagreements, accepted by the oracle but refused by the reference)}."""
    rng = np.random.default_rng(base + 777)
    stats = {}
    for kind in ("pdsch", "pdcch", "ssb", "csi"):
        n = bad = refused = bad_ref = 0
        while n < count:
            if kind == "pdsch":
                pool = [x[0] for x in cases.random_pdus(o.tbs, rng, 20)]
            elif kind == "pdcch":
                pool = [cases.random_pdcch(rng) for _ in range(20)]
            elif kind == "ssb":
                pool = [cases.random_ssb(rng, int(rng.integers(24, 107)), int(rng.integers(1, 5))) for _ in range(20)]
            else:
                pool = [c[1] for c in cases.csi_rs_cases(rng)]
            for obj in pool:
                mild = True
                for _ in range(int(rng.integers(1, 3))):
                    field, value = mutate(rng, obj)
                    mild = mild and value <= 1024
                if kind == "pdsch":
                    a, b = backends.pkg.lib.validate(obj), o.validate(obj)
                    # The reference's validator is asked only where the oracle accepts and every mutated value is small: it must accept
                    # everything the oracle does (it knows nothing of the limits this ABI adds), but its own descriptor types cannot hold
                    # most mutated values (a cell identity of 2^31 ...) -- the harness's conversion of such a PDU crashes inside the
                    # reference's containers.
                    if b == 0:   # the caller's side of the contract: a weight array as large as the (possibly overwritten) counts say
                        cases.attach_weights(obj, np.zeros((obj.nof_prg, obj.nof_ports, obj.nof_layers, 2), np.float32))
                    c = r.validate(obj) if (r is not None and b == 0 and mild) else b
                    if b == 0 and c != 0:
                        bad_ref += 1
                        if verbose:
                            print("PDSCH oracle accepts, reference refuses: field", field, flush=True)
                    if a == 0 and b == 0 and backends.pkg.lib.derive(obj) != o.derive(obj):   # host derivation on whatever passes
                        bad += 1
                        if verbose:
                            print("PDSCH DERIVATION DISAGREEMENT after field", field, flush=True)
                elif kind == "pdcch":
                    a, b = h.nrphy_pdcch_validate(C.byref(obj)), o.pdcch_validate(obj)
                elif kind == "ssb":
                    a, b = h.nrphy_ssb_validate(C.byref(obj)), o.ssb_validate(obj)
                else:
                    a, b = h.nrphy_csi_rs_validate(C.byref(obj)), o.csi_rs_validate(obj)
                n += 1
                refused += b != 0
                if (a == 0) != (b == 0):
                    bad += 1
                    if verbose:
                        print(kind.upper(), "VALIDATOR DISAGREEMENT field", field, "product", a, "oracle", b, flush=True)
        stats[kind] = (n, refused, bad, bad_ref)
        if verbose:
            print("%s: %d mutated descriptors, %d refused, %d product/oracle disagreements%s" % (
                kind, n, refused, bad, (", %d accepted by the oracle but refused by the reference" % bad_ref) if kind == "pdsch" else ""),
                  flush=True)
    return stats


if __name__ == "__main__":
    result = run(int(sys.argv[1]) if len(sys.argv) > 1 else 4000)
    sys.exit(1 if any(v[2] or v[3] for v in result.values()) else 0)
