"""bench.py's N-rank path rehearsed on ONE device: `--gpus 2 --share-gpu` starts two process ranks that both use device 0
and exchange through the host-staged transport over gloo (dist.td_host_transport) -- decomposition, ghost pruning, plan,
halo exchange, all-reduced dots, max-over-ranks timing and the one JSON line are those of the driver's multi-GPU run;
only the transport differs (RCCL forms no communicator between two ranks of one device)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_process_ranks_sharing_the_gpu():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--ncell", "20", "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--no-dropin", "--no-alt"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line, the record"
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 1 and rec["scaling"] == "weak" and rec["value"] > 0
    cfg = rec["config"]
    assert cfg["global_rows"] == 2 * 20 ** 3 and cfg["rows_per_gpu"] == 20 ** 3
    assert cfg["converged"] == 1 and cfg["rel_res"] < 1e-8
    assert "rehearsal" in cfg["parallelism"]
