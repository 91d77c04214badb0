import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, backends, cases
lib, abi = backends.pkg.lib, backends.pkg.abi
ctx = lib.Context(0)
pdu, ports, subc, ofdm = cases.baseline_config(3)
slots = 8
pdus = [cases.baseline_config(3, slot_index=i % 20)[0] for i in range(slots)]
stride = (pdus[0].tb_size_bytes + 255) & ~255
tb = torch.randint(0, 256, (slots * stride,), dtype=torch.uint8, device="cuda")
plan = lib.PdschPlan(ctx, pdus, [i * stride for i in range(slots)], list(range(slots)), slots, ports, subc)
oplan = lib.OfdmPlan(ctx, ofdm, ports)
d_grid = torch.zeros((slots, ports, 14, subc), dtype=torch.int32, device="cuda")
plan.run(tb, d_grid, zero_grids=True)
for gain in (-14.0, -17.0, -20.0):
    wire = abi.IqWireCfg(abi.AmplitudeCfg(0, 1, gain, 1.0, -1.0), 32767.0)
    d_iq = torch.zeros((slots, ports, oplan.slot_stride, 2), dtype=torch.int16, device="cuda")
    d_stats = torch.zeros((slots * ports, 4), dtype=torch.int32, device="cuda")
    d_slot = torch.tensor([i % 2 for i in range(slots)], dtype=torch.int32, device="cuda")
    oplan.run_ci16(slots, d_grid, wire, d_iq, d_slot_index=d_slot, d_stats=d_stats)
    ctx.synchronize()
    st = d_stats.cpu().numpy()
    n = st[:, 3].astype(np.float64)
    mean = st[:, 0].view(np.float32).astype(np.float64) / n
    print("gain %.0f dB: mean power %.4f (%.1f dBFS), peak power %.3f, clipped %d of %d samples; limit^2 = %.3f" % (
        gain, mean.mean(), 10 * np.log10(mean.mean()), st[:, 1].view(np.float32).max(), int(st[:, 2].sum()), int(n.sum()), 10 ** (-2 / 20)))
print("ofdm scale", ofdm.scale)
