"""bench.py end to end on the GPU box: the single-GPU line the driver records, and the N-rank launch -- through a
launcher (`python -m torch.distributed.run ... bench.py --gpus 2`) and without one (`python bench.py --gpus 4`: bench.py
starts its own ranks) -- rehearsed with the ranks sharing the one GPU (`--share-gpu`: gloo rendezvous instead of RCCL,
one exchange workgroup per rank instead of the fused exchange, everything else the same code path)."""
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
    assert d["value"] > 1e8 and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert "configs[2]" in d["config"]["workload"] and d["config"]["free_intrinsics"] == 8 and d["config"]["images_total"] == 1000
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["avg_launch_us"] > 0
    assert r["fp64_tflops"] > 0 and abs(r["fp64_frac"] - r["fp64_tflops"] / r["fp64_peak_tflops"]) < 1e-12
    assert "latency_floor_us" in r and "traffic_source" in r
    assert abs(d["lm"]["rms_px"] - 0.13703750955) < 1e-9 and d["lm"]["iters_to_tol_1e-3"] == 4
    # extra blocks: the saturated C5 shape on this one GPU and the map / metric entry points
    assert d["c5"]["config"]["images_total"] == 10000 and d["c5"]["config"]["free_intrinsics"] == 9 and d["c5"]["value"] > 1e9
    assert "configs[4]" in d["c5"]["config"]["workload"] and 0.12 < d["c5"]["lm"]["rms_px"] < 0.16
    m = d["maps"]
    assert m["points"] == 54000 and m["forward_points_per_s"] > 1e6 and m["inverse_points_per_s"] > 1e6
    # (the Julia maps carry ONE radial coefficient: C3's k2, p1, p2 are dropped there, so this is not the fit's 0.137 px)
    assert m["errors"]["reprojection"] < 5.0 and m["single_point_call_us"] < 500
    # corner detection on the reference's example images: its own bar (every corner within 1 px of corners.json)
    t = d["detect"]
    assert t["images"] == 8 and t["max_corner_error_px"] < 1.0 and t["images_per_s"] > 10


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal():
    env = dict(os.environ, CCAL_P2P_TIMEOUT_MS="4000")
    for attempt in range(3):          # (a port found free can be taken before the launcher listens on it: once more with another)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                            "--gpus", "2", "--steps", "20", "--warmup", "3", "--share-gpu", "--images", "120", "--timed-only"],
                           capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
        if p.returncode == 0 or not ("EADDRINUSE" in p.stderr or "address already in use" in p.stderr.lower()):
            break
    assert p.returncode == 0, p.stderr[-3000:]
    d = _last_json(p.stdout)
    assert REQUIRED <= set(d) and d["n_gpus"] == 2 and "cpu_baseline" not in d
    assert d["config"]["allreduce"].startswith("p2p")            # the direct exchange passed its self-test
    # default at N > 1: ONE problem sharded over the ranks (BASELINE.json configs[3]), strong scaling
    assert d["scaling"] == "strong" and "configs[3]" in d["config"]["workload"]
    assert d["config"]["images_total"] == 120 and d["config"]["images_rank0"] == 60 and 0.12 < d["lm"]["rms_px"] < 0.15
    assert d["value"] == pytest.approx(120 * 54 * d["lm"]["sweeps_per_solve"] / (d["ms_per_step"] * 1e-3), rel=1e-6)


@pytest.mark.gpu
def test_bench_four_ranks_full_c3_rehearsal():
    """The command the driver runs at N = 4 -- `python bench.py --gpus 4`, NO launcher: bench.py starts its own ranks --
    rehearsed with the ranks sharing the one GPU: the real 1000-image C3 set in four shards, plus the weak-scaling
    block at its named size (1000 images per rank) and the C5 block (cut to 2048 images to keep the rehearsal short).
    Ranks that share a GPU exchange through ONE workgroup each (ccal_p2p_set_exchange), so a waiting rank never keeps
    the sender off the chip: every block must come through the direct exchange, no fallback."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["CCAL_P2P_TIMEOUT_MS"] = "8000"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "10", "--warmup", "2",
                        "--share-gpu", "--c5-images", "2048"], capture_output=True, text=True, timeout=800, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and "configs[3]" in d["config"]["workload"]
    assert d["config"]["images_total"] == 1000 and d["config"]["images_rank0"] == 250
    assert abs(d["lm"]["rms_px"] - 0.13703750955) < 1e-9 and d["lm"]["iters_to_tol_1e-3"] == 4    # = the one-GPU fit
    w = d["weak_scaling"]
    assert w["config"]["images_total"] == 4000 and w["config"]["images_rank0"] == 1000 and 0.12 < w["lm"]["rms_px"] < 0.15
    c5 = d["c5"]
    assert c5["config"]["images_total"] == 2048 and c5["config"]["images_rank0"] == 512 and 0.12 < c5["lm"]["rms_px"] < 0.16
    for blk in (d, w, c5):
        assert blk["config"]["allreduce"].startswith("p2p") and "one exchange workgroup" in blk["config"]["allreduce"]
    # the same sharded fit over the torch.distributed hook (RCCL inside the library refuses ranks that share a device)
    t = d["transports"]
    assert set(t) == {"torch"} and t["torch"]["transport"].startswith("torch.distributed")
    assert abs(t["torch"]["rms_px"] - d["lm"]["rms_px"]) < 1e-9 and t["torch"]["iters"] == 4 and t["torch"]["ms_per_step"] > 0
    # independent calibrations, 32 per rank and round, no exchange: the block that scales with GPUs when the problems are many and small
    f = d["independent_fits"]
    assert f["ranks"] == 4 and f["scaling"] == "weak" and f["fits_per_s"] > 1000 and f["corner_residuals_per_s"] > 0
