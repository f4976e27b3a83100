"""The world-size > 1 DATA path executed on the one GPU of the box.

RCCL refuses two ranks on one device, so until an 8-GPU node runs the driver's scaling bench the only
way to execute `smm_comm_gather / allgather` -> `Comm.gather_rows` -> `TiledRingGather` on device memory
with more than one rank is a stand-in collective library: tests/cpp/fake_rccl.cpp (blocking, staged
through POSIX shared memory), bound through the same `SMM_RCCL_LIB` hook the product uses for librccl.
The ranks are fresh child processes sharing device 0.  Nothing here says anything about xGMI or
scaling: it proves the code above the collective (event hand-over, ring-slot views, short last tile,
byte accounting, bench.py's `with_gather` phase) moves the right rows of the right rank.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "fake_rccl.cpp")
RANK = os.path.join(ROOT, "tests", "cpp", "stand_in_rank.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    subprocess.check_call(["g++", "-O1", "-Wall", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           SRC, "-o", so, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-Wl,-rpath,/opt/rocm/lib"])
    return so


def _env(fake, **extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR",
                                                            "MASTER_PORT", "LOCAL_WORLD_SIZE", "SMM_RDV_PORT")}
    env.update(SMM_RCCL_LIB=fake, SMM_FAKE_RCCL_TIMEOUT_S="90", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra)
    return env


@pytest.mark.parametrize("world,slot_bytes", [(2, None), (3, None), (4, None), (2, "4096")])
def test_gather_allgather_and_the_tiled_ring_on_device_memory(hip, fake_rccl, tmp_path, world, slot_bytes):
    """2, 3 and 4 rank processes on device 0: Comm.gather (root = last rank) / allgather in f64 and f32,
    regrid_sharded root / all / none with a short last shard, TiledRingGather with 4 tiles of 2, 2, 2, 1 rows over
    2 slots for 2 steps -- root's assembled Y bit-equal to oracle.apply_c per rank and tile, `gathered_bytes` as
    tests/test_distributed_cpu.py computes it.  The third case shrinks the stand-in's staging slot to 4 KiB so
    that every collective takes several chunks."""
    port = _free_port()
    extra = {"SMM_FAKE_RCCL_SLOT_BYTES": slot_bytes} if slot_bytes else {}
    outs = [str(tmp_path / f"rank{r}.json") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, RANK, str(r), str(world), str(port), outs[r]], cwd=ROOT,
                              env=_env(fake_rccl, **extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=300)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} exited with {p.returncode}:\n{logs[r][-3000:]}"
    seen = [json.load(open(o)) for o in outs]
    n_dst = 36 * 18
    assert seen[0]["ring"] == {"delivered": [0, 1, 2, 3] * 2, "gathered_bytes": 2 * (world - 1) * 7 * n_dst * 8,
                               "tiles": 4}
    for r in range(1, world):
        assert seen[r]["ring"] == {"delivered": [], "gathered_bytes": 0, "tiles": 4}
        assert seen[r]["regrid_sharded"] == "ok" and seen[r]["gather_allgather"] == "ok"


def test_a_rank_that_never_arrives_ends_the_others_with_a_status(hip, fake_rccl, tmp_path):
    """The stand-in's barrier has a deadline: with one of two ranks missing, smm_comm_create does not hang the box --
    the waiting rank prints a line and exits 86."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from smmregrid_amd.comm import Comm, unique_id\n"
            "from smmregrid_amd.device import set_device\n"
            "set_device(0)\n"
            "Comm(0, 2, comm_id=unique_id())\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=_env(fake_rccl, SMM_FAKE_RCCL_TIMEOUT_S="2"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 86, (out.returncode, out.stderr[-1500:])
    assert "not every rank called ncclCommInitRank in time" in out.stderr


def test_bench_two_ranks_with_gather_through_the_stand_in(hip, fake_rccl):
    """`bench.py --gpus 2 --gather root` with both ranks on device 0: launcher, rendezvous, compute loop, communicator
    set-up, the compute + tiled-gather loop (`gather_phase`) and the JSON line -- a `with_gather` block without
    `error`, bytes as the schedule predicts."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--gather", "root", "--steps", "2",
                          "--warmup", "1", "--batch", "96", "--gather-tiles", "5", "--no-cpu-baseline"], cwd=ROOT,
                         env=_env(fake_rccl, SMM_BENCH_SHARE_GPUS="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert "2 ranks share 1 device" in line["config"]["rehearsal"] and "fake_rccl" in line["config"]["rehearsal"]
    g = line["with_gather"]
    assert "error" not in g, g
    assert g["ranks"] == 2 and g["tiles"] == 5 and g["value"] > 0 and g["ms_per_step"] > 0
    assert g["gathered_bytes_per_step"] == 96 * 64800 * 8         # (world - 1) shards of 96 x 64 800 doubles
    assert line["spot_check"]["bit_equal_to_oracle"] is True


def test_bench_two_ranks_run_the_baseline_configs_with_their_gathers(hip, fake_rccl):
    """The rest of the N > 1 bench that had never executed on a GPU: after the headline (config 2 at full size on both
    ranks, its compute + gather loop) every rank runs its share of BASELINE configs 4 and 5 -- here 8 rows each
    (--config-batch) -- with the HBM guard decided by all ranks together, the per-rank kernel times collected over the
    rendezvous and the compute + tiled-gather loop of `baseline_config_block`, all through the stand-in library."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--gather", "root", "--steps", "2",
                          "--warmup", "1", "--config-batch", "8", "--config-steps", "2", "--config-warmup", "1",
                          "--config-gather-steps", "1", "--gather-tiles", "3"], cwd=ROOT,
                         env=_env(fake_rccl, SMM_BENCH_SHARE_GPUS="1"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and "error" not in line["with_gather"] and line["with_gather"]["ranks"] == 2
    assert line["with_gather"]["gathered_bytes_per_step"] == 3600 * 64800 * 8
    for name, d_cells, y_item in (("cfg4", 12 * 1024 * 1024, 8), ("cfg5", 720 * 360, 8)):
        blk = line["baseline_configs"][name]
        assert blk["n_gpus"] == 2 and blk["rows_per_gpu"] == 8 and blk["spot_check"] is True and blk["value"] > 0
        assert 0 < blk["kernel_ms_min"] <= blk["kernel_ms_max"]
        g = blk["with_gather"]
        assert "error" not in g, g
        assert g["ranks"] == 2 and g["gathered_bytes_per_step"] == 8 * d_cells * y_item and g["tiles"] == 3
