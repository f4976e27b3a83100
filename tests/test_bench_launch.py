"""bench.py as the driver runs it: `python bench.py --gpus N` must be a real N-rank run.  --dry-run
replaces the GPU step by a stand-in, everything else -- the launcher, the TCP rendezvous, barriers,
max over ranks, the tiled gather schedule (gloo) and the JSON line -- is the code of the real run."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(cmd, env=None, timeout=180):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR",
                                                          "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    e.update(env or {})
    return subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


def json_line(out):
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (out.stdout[-2000:], out.stderr[-2000:])     # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_gpus_n_starts_n_ranks(n):
    out = run([sys.executable, BENCH, "--gpus", str(n), "--dry-run", "--steps", "3", "--warmup", "1",
               "--gather", "none"])
    assert out.returncode == 0, out.stderr[-2000:]
    line = json_line(out)
    assert line["n_gpus"] == n and line["steps"] == 3 and line["warmup"] == 1
    assert line["config"]["launched_by"] == "bench.py" and line["scaling"] == "weak"
    assert line["value"] > 0 and line["ms_per_step"] > 0


def test_gather_phase_on_two_gloo_ranks():
    out = run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1", "--comm", "torch",
               "--gather-tiles", "5"])
    assert out.returncode == 0, out.stderr[-2000:]
    line = json_line(out)
    g = line["with_gather"]
    assert line["n_gpus"] == 2 and g["ranks"] == 2 and g["tiles"] == 5 and g["ring_slots"] == 2
    assert g["gathered_bytes_per_step"] == 64 * 96 * 8          # (world - 1) shards of 64 x 96 doubles
    assert g["ms_per_step"] > 0


def test_world_size_must_match_gpus():
    out = run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr
    out = run([sys.executable, BENCH, "--gpus", "1", "--dry-run"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr


def test_a_failing_rank_fails_the_run():
    out = run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--gather", "none"],
              env={"SMM_BENCH_TEST_FAIL_RANK": "1"}, timeout=120)
    assert out.returncode != 0
    assert "rank 1 exited with 7" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]    # no line from a broken run


def test_under_torch_distributed_run():
    """The driver's other launch form: one rank per GPU started by torch.distributed.run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--dry-run",
               "--steps", "2", "--warmup", "1", "--gather", "none"], timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json_line(out)
    assert line["n_gpus"] == 2 and line["config"]["launched_by"] == "external launcher"


def test_cpu_threads_respects_affinity_and_quota(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    share, usable = bench.cpu_threads()
    assert 1 <= share <= 16 and share <= usable <= len(os.sched_getaffinity(0))
    monkeypatch.setattr(bench, "cgroup_cpu_quota", lambda: 3.0)
    assert bench.cpu_threads() == (min(3, len(os.sched_getaffinity(0))),) * 2
    monkeypatch.setattr(bench, "cgroup_cpu_quota", lambda: 0.4)          # never below one thread
    assert bench.cpu_threads() == (1, 1)
