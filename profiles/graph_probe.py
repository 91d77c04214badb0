import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
import backends, cases
abi, lib = backends.abi, backends.pkg.lib
ctx = lib.Context(0)
oracle = backends.oracle()
pdu, nof_ports, nof_subc, ocfg = cases.baseline_config(3)
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pdus = [cases.baseline_config(3, slot_index=i % 20)[0] for i in range(slots)]
tb_bytes = (pdu.tb_size_bytes + 3) & ~3
plan = lib.PdschPlan(ctx, pdus, [i * tb_bytes for i in range(slots)], list(range(slots)), slots, nof_ports, nof_subc)
oplan = lib.OfdmPlan(ctx, ocfg, nof_ports)
rng = np.random.default_rng(1)
d_tb = [torch.zeros(slots * tb_bytes, dtype=torch.uint8, device="cuda") for _ in range(2)]
d_grid = [torch.zeros((slots, nof_ports, 14, nof_subc), dtype=torch.int32, device="cuda") for _ in range(2)]
d_iq = [torch.zeros((slots, nof_ports, oplan.slot_stride, 2), dtype=torch.float32, device="cuda") for _ in range(2)]
d_slot = torch.tensor([i % 2 for i in range(slots)], dtype=torch.int32, device="cuda")
def fill():
    tbs = []
    for k in range(2):
        h = np.zeros(slots * tb_bytes, np.uint8)
        for i in range(slots):
            h[i * tb_bytes: i * tb_bytes + pdu.tb_size_bytes] = rng.integers(0, 256, pdu.tb_size_bytes, dtype=np.uint8)
        d_tb[k].copy_(torch.from_numpy(h)); tbs.append(h)
    return tbs
s = torch.cuda.Stream()
def two_steps(stream):
    for k in range(2):
        plan.run(d_tb[k], d_grid[k], zero_grids=True, stream=stream)
        oplan.run(slots, d_grid[k], d_iq[k], d_slot_index=d_slot, stream=stream)
fill()
with torch.cuda.stream(s):
    two_steps(s.cuda_stream)   # warm-up (allocations, epoch parity back to even)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    two_steps(torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
ok = True
for rep in range(3):
    tbs = fill()
    g.replay()
    torch.cuda.synchronize()
    for k in range(2):
        grid = d_grid[k][0].cpu().numpy().view(np.uint16).reshape(nof_ports, 14, nof_subc, 2)
        want = oracle.pdsch_process(pdus[0], tbs[k][:pdu.tb_size_bytes], nof_ports, nof_subc)
        ok &= bool(np.array_equal(grid, want))
print("graph replay parity:", ok)
def timeit(fn, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n / 2 * 1e6
def eager():
    with torch.cuda.stream(s): two_steps(s.cuda_stream)
print("slots/step %d: eager %.1f us/step, graph %.1f us/step" % (slots, timeit(eager), timeit(g.replay)))
