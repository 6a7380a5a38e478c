"""bench.py end to end on the GPU box: the single-GPU line the driver records, and the N-rank launch
(`python -m torch.distributed.run ... bench.py --gpus N`) rehearsed with the ranks sharing the one GPU
(`--share-gpu`: gloo rendezvous instead of RCCL, everything else the same code path)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


def _last_json(out):
    lines = [ln for ln in out.strip().splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


@pytest.mark.gpu
def test_bench_single_gpu_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["steps"] == 20 and d["dtype"] == "f64"
    assert d["value"] > 1e8 and d["scaling"] == "weak" and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["avg_launch_us"] > 0
    assert abs(d["lm"]["rms_px"] - 0.13703750955) < 1e-9 and d["lm"]["iters_to_tol_1e-3"] == 4


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CCAL_P2P_TIMEOUT_MS="4000")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "20", "--warmup", "3", "--share-gpu", "--images", "120", "--timed-only"],
                       capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _last_json(p.stdout)
    assert REQUIRED <= set(d) and d["n_gpus"] == 2 and "cpu_baseline" not in d
    assert d["config"]["allreduce"].startswith("p2p")            # the direct exchange passed its self-test
    assert d["config"]["images_per_gpu"] == 120 and 0.12 < d["lm"]["rms_px"] < 0.15
    assert d["value"] == pytest.approx(2 * 120 * 54 * d["lm"]["sweeps_per_solve"] / (d["ms_per_step"] * 1e-3), rel=1e-6)
