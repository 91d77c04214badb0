#!/usr/bin/env python3
"""Probe: nrphy_demodulate_soft against the oracle on link-like inputs (constellation + noise, one variance per call)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import backends
lib = backends.pkg.lib
ctx = lib.Context(0)
o = backends.oracle()
rng = np.random.default_rng(5)
for mod, nsym, spans in ((6, 7488, 6), (8, 5616, 6), (4, 11232, 3), (2, 22464, 2), (6, 7489, 3)):
    sym = (rng.standard_normal((spans, nsym, 2)) * 0.7).astype(np.float32)
    nv = np.full((spans, nsym), 0.008, np.float32)
    qm = max(mod, 1)
    d_llr = torch.zeros(spans * nsym * qm + 32, dtype=torch.int8, device="cuda")
    ctx.demodulate_soft(mod, spans, nsym, torch.from_numpy(sym).cuda(), torch.from_numpy(nv).cuda(), d_llr)
    ctx.synchronize()
    got = d_llr.cpu().numpy()[: spans * nsym * qm].reshape(spans, nsym * qm)
    for r in range(spans):
        want = o.demodulate_soft(mod, sym[r].view(np.complex64).reshape(-1), nv[r])
        bad = np.flatnonzero(got[r] != want)
        print(mod, nsym, r, len(bad), bad[:8], got[r][bad[:8]], want[bad[:8]])
