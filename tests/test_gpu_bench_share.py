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
    assert cfg["library_row_order"] == "bricks" and cfg["atom_order"] == "lexicographic"
    mg = rec["multi_gpu"]
    assert mg["rccl_ranks"] == 2 and mg["transport"] == "host" and len(mg["per_rank"]) == 2
    assert mg["iterations_equal_on_all_ranks"] and mg["global_rel_residual"] < 1e-7
    assert mg["a_times_one_max_abs"] <= 1e-9 * mg["a_times_signs_max_abs"]
    for q in mg["per_rank"]:
        assert q["peers"] >= 1 and q["ghost_cols"] > 0 and q["halo_bytes_per_spmv"] > 0 and q["halo_profile"]["products"] > 0


@pytest.mark.gpu
def test_bench_four_process_ranks_validate_their_own_record():
    """the self-validating N > 1 record on four process ranks of 40^3 particles sharing the device (a
    GPU box admits at most six processes on its card): same iteration count on every rank, the communicator spans the
    job, the global residual from the distributed product, A 1 = 0 across every rank boundary, per-rank halo figures"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--share-gpu", "--ncell", "40", "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline", "--no-dropin", "--no-alt"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    mg = rec["multi_gpu"]
    assert rec["n_gpus"] == 4 and mg["rccl_ranks"] == 4 and len(mg["per_rank"]) == 4
    assert len({q["iterations"] for q in mg["per_rank"]}) == 1 and mg["per_rank"][0]["iterations"] == rec["config"]["iterations"]
    assert all(q["comm"]["ranks"] == 4 and q["comm"]["rank"] == q["rank"] for q in mg["per_rank"])
    assert all(q["peers"] == 3 for q in mg["per_rank"])                      # 2 x 2 x 1 periodic: three distinct neighbours
    assert mg["global_rel_residual"] < 1e-7 and mg["a_times_one_max_abs"] <= 1e-9 * mg["a_times_signs_max_abs"]
