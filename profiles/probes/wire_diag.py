import os, sys
sys.path.insert(0, "profiles"); sys.path.insert(0, "tests")
sys.argv = ["x", "--nothing"]
import numpy as np, torch
import backends
abi, lib = backends.abi, backends.pkg.lib
o = backends.oracle_module() if hasattr(backends, "oracle_module") else None
import importlib
fs = importlib.import_module("fuzz_sweep")
o, ctx = fs.o, fs.ctx
rng = np.random.default_rng(0 + 16180)
shown = 0
for t in range(40):
    size = int(rng.choice([256, 512, 1024, 1536, 2048, 4096, 4608, 6144]))
    mu, ext = int(rng.integers(0, 4)), int(rng.integers(0, 5) == 0)
    bw, ports, slots = int(rng.integers(1, min(275, (size - 1) // 12) + 1)), int(rng.integers(1, 4)), int(rng.integers(1, 4))
    ocfg = abi.OfdmConfig(mu, bw, size, ext, float(rng.uniform(0.5, 2.0)) / np.sqrt(size), float(rng.choice([0.0, 2.4e9, 3.5e9])))
    amp = abi.AmplitudeCfg(0, int(rng.integers(0, 2)), float(rng.uniform(-20, 6)), float(rng.choice([1.0, 2.0])), float(rng.uniform(-12, -0.5)))
    wire_cfg = abi.IqWireCfg(amp, float(rng.choice([32767.0, 20000.0, 40000.0])))
    grid = ((rng.standard_normal((slots, ports, 14, bw * 12, 2)) * rng.uniform(0.2, 1.0)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    pl = lib.OfdmPlan(ctx, ocfg, ports)
    d_grid = torch.from_numpy(grid.view(np.uint32).reshape(slots, ports, 14, bw * 12).view(np.int32)).cuda()
    d_slot = torch.from_numpy((np.arange(slots, dtype=np.uint32) % (1 << mu)).view(np.int32)).cuda()
    d_iq16 = torch.zeros((slots, ports, pl.slot_stride, 2), dtype=torch.int16, device="cuda")
    d_stats = torch.zeros((slots * ports, 4), dtype=torch.int32, device="cuda")
    d_iq = torch.zeros((slots, ports, pl.slot_stride, 2), dtype=torch.float32, device="cuda")
    pl.run_ci16(slots, d_grid, wire_cfg, d_iq16, d_slot_index=d_slot, d_stats=d_stats)
    pl.run(slots, d_grid, d_iq, d_slot_index=d_slot)
    ctx.synchronize()
    fused, stats = d_iq16.cpu().numpy(), d_stats.cpu().numpy()
    iq = d_iq.cpu().numpy().view(np.complex64).reshape(slots, ports, -1)
    for s_ in range(slots):
        ssz = lib.slot_size(ocfg, int(s_ % (1 << mu)))
        for p_ in range(ports):
            y, m = o.amplitude_control(wire_cfg.amplitude, iq[s_, p_, :ssz])
            want = o.iq_convert_ci16(np.concatenate([y, np.zeros(8, np.complex64)]), wire_cfg.ci16_scale).reshape(-1, 2)[:ssz]
            st = stats[s_ * ports + p_]
            d = np.abs(fused[s_, p_, :ssz].astype(np.int32) - want.astype(np.int32))
            c = [d.max() <= 1, np.mean(fused[s_, p_, :ssz] == want) > 0.9999, st[3] == ssz, st[2] == m["nof_clipped"],
                 st[1].view(np.float32) == np.float32(m["stats"].peak_power),
                 abs(st[0].view(np.float32) - m["stats"].sum_power) <= 1e-4 * max(m["stats"].sum_power, 1e-30)]
            if not all(c) and shown < 12:
                shown += 1
                print(t, "size", size, "mu", mu, "ext", ext, "bw", bw, "clip", amp.enable_clipping, "gain", round(amp.input_gain_dB, 2), "ceil", round(amp.ceiling_dBFS, 2), "scale", wire_cfg.ci16_scale,
                      "checks", c, "maxdiff", d.max(), "neq", int((fused[s_, p_, :ssz] != want).sum()), "of", ssz * 2, "clipped", st[2], m["nof_clipped"],
                      "peak", st[1].view(np.float32), m["stats"].peak_power, "sum", st[0].view(np.float32), m["stats"].sum_power, flush=True)
                bad = np.argwhere(d.max(axis=1) > 0)[:6, 0]
                print("   first differing samples", bad, fused[s_, p_, bad].tolist(), want[bad].tolist(), "y", y[bad].tolist())
    pl.close()
