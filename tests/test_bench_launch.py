"""`python bench.py --gpus N` with no launcher in the environment must start its own N ranks (the driver's command
line is exactly that): the launch plumbing of bench.py -- child processes with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_*, started before anything touches a GPU, rank 0's JSON line relayed, a non-zero exit when a rank dies --
checked here without a GPU through `--dry-run` (gloo rendezvous only).  The same launch with real solves is the
`-m gpu` rehearsal in tests/test_bench_rehearsal.py."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    return env


def test_bench_starts_its_own_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT, env=_env())
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                         # one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d == {"dry_run": True, "n_gpus": 2, "sum_of_rank_numbers": 3, "launcher": "own child processes"}


def test_bench_starts_eight_ranks():
    """the driver's largest command, `python3 bench.py --gpus 8 ...`: eight child ranks rendezvous and one line comes back"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT, env=_env())
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["sum_of_rank_numbers"] == 36 and json.loads(lines[0])["n_gpus"] == 8


def test_bench_keeps_an_external_launcher():
    """With WORLD_SIZE in the environment (torch.distributed.run, the driver's N > 1 form) the process IS a rank."""
    env = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip().splitlines()[-1])["launcher"] == "external"


def test_bench_exits_nonzero_when_a_rank_dies():
    env = dict(_env(), CCAL_BENCH_FAIL_RANK="1", CCAL_BENCH_GRACE_S="3")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT, env=env)
    assert p.returncode != 0 and time.time() - t0 < 120
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_deadline_ends_a_stuck_optional_block_with_the_line_it_has():
    """bench.Deadline: rank 0 prints what has been measured (marked incomplete) and the process ends with a status of its
    own -- the line is out, but a hung transport must not read as a clean run"""
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "with bench.Deadline(0.5, 0, lambda: {'value': 1}):\n"
            "    time.sleep(30)\n"
            "print('not reached')\n") % ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, cwd=ROOT, env=_env())
    import json
    import bench
    assert p.returncode == bench.DEADLINE_EXIT != 0 and "abandoned" in p.stderr
    assert json.loads(p.stdout.strip()) == {"value": 1, "incomplete": True}
    code = ("import sys; sys.path.insert(0, %r); import bench\n"
            "with bench.Deadline(30.0, 0, lambda: {'value': 1}):\n"
            "    pass\n"
            "print('done')\n") % ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, cwd=ROOT, env=_env())
    assert p.returncode == 0 and p.stdout.strip() == "done"
