import sys, time
sys.path.insert(0, 'tests')
import backends, cases
lib = backends.pkg.lib
ctx = lib.Context(0)
for cfg in (3, 2):
    pdu, ports, subc, ofdm = cases.baseline_config(cfg)
    stride = (pdu.tb_size_bytes + 255) & ~255
    for n in (1, 64, 1024):
        pdus = [cases.baseline_config(cfg, slot_index=i % 10)[0] for i in range(n)]
        t0 = time.perf_counter()
        plan = lib.PdschPlan(ctx, pdus, [i * stride for i in range(n)], list(range(n)), n, ports, subc)
        t1 = time.perf_counter()
        print("config %d: plan of %4d PDUs: %.2f ms (%.3f ms/PDU)" % (cfg, n, 1e3 * (t1 - t0), 1e3 * (t1 - t0) / n), flush=True)
        del plan
cell = []
grid_of = []
for i in range(256):
    c, ports, subc = cases.mixed_cell(i % 4)
    cell += c; grid_of += [i] * 4
offs = []; o = 0
for q in cell:
    offs.append(o); o += (q.tb_size_bytes + 255) & ~255
t0 = time.perf_counter(); plan = lib.PdschPlan(ctx, cell, offs, grid_of, 256, ports, subc); t1 = time.perf_counter()
print("config 4: plan of 1024 PDUs on 256 grids: %.2f ms" % (1e3 * (t1 - t0)))
