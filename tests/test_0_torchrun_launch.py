"""bench.py under the launcher the driver uses for N > 1, at N = 1 on the one GPU of the test box: RCCL initialisation, the
barrier and sharding.aggregate's all-reduces run on a real device.  The file sorts first so the launcher starts before this
pytest process has touched the GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_under_torch_distributed_run_one_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    with socket.socket() as s:   # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--slots", "64", "--no-secondary", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["metric"] == "pdsch_slots_per_sec"
    assert out["collective_backend"].startswith("rccl")
    assert out["verified_vs_oracle"] is True
    assert out["value"] > 0 and out["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_bench_gpus_2_self_launch_rehearsal_on_one_device():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the two ranks itself (it never touches the GPU)
    and relays rank 0's line.  On this one-GPU box both ranks work on cuda:0 and the process group is gloo
    (--rehearse-one-device: RCCL refuses two ranks on one device), so this is the N-rank code path end to end -- sharding, both
    ranks' device work, totals, per-rank gather, the config-4 secondary placed by cell affinity -- not a scaling measurement."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-device", "--steps", "2", "--warmup", "1",
           "--slots", "64", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["verified_vs_oracle"] is True
    assert out["per_rank"]["slots_per_step"] == [64, 64] and len(out["per_rank"]["hbm_frac"]) == 2
    assert out["collective_backend"].startswith("gloo (rehearsal")
    c4 = out["secondary"]["config4"]
    assert c4["verified_vs_oracle"] is True and sum(c4["per_rank"]["slots_per_step"]) == 2048
    assert "cell_affine_rank" in c4["config"]["sharding"]
