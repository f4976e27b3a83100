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


def test_n_rank_line_carries_baseline_configs_4_and_5():
    """At N > 1 every rank also runs its share of BASELINE configs 4 and 5 (dry run: stand-in steps), compute
    only and with the tiled gather; rank 0 reports cells/s, the spread of the per-rank kernel times, the ranks
    the communicator saw and the gathered bytes.  The last line is the compact one; the bulky details come
    first on a line that does not start with '{'."""
    out = run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1", "--comm", "torch",
               "--config-steps", "3"])
    assert out.returncode == 0, out.stderr[-2000:]
    line = json_line(out)
    assert len(json.dumps(line)) < 4096 and out.stdout.rstrip().splitlines()[-1].startswith("{")
    assert max(len(v) for v in _strings(line)) <= 100
    for name in ("cfg4", "cfg5"):
        blk = line["baseline_configs"][name]
        assert blk["n_gpus"] == 2 and blk["value"] > 0 and blk["rows_per_gpu"] == 64
        assert 0 < blk["kernel_ms_min"] <= blk["kernel_ms_max"]
        g = blk["with_gather"]
        assert g["ranks"] == 2 and g["gathered_bytes_per_step"] == 64 * 96 * 8 and g["ms_per_step"] > 0
    r = line["roofline"]
    assert 0 < r["kernel_ms_min_rank"] <= r["kernel_ms_max_rank"]
    details = [ln for ln in out.stdout.splitlines() if ln.startswith("details: ")]
    assert len(details) == 1 and "baseline_configs" in json.loads(details[0][len("details: "):])


def test_eight_rank_dry_run_prints_the_hbm_need_and_skips_what_does_not_fit():
    """before a BASELINE config allocates, every rank holds its need (X + Y shards, operator,
    on rank 0 the gather's ring: config 4 = 57 + 110 + 3 + 35 GB) against the free device memory; the ranks decide
    together, and a config that does not fit is reported as skipped instead of dying in hipMalloc.  The dry-run line
    at --gpus 8 carries `hbm_needed_gb` per config."""
    sys.path.insert(0, ROOT)
    import argparse
    import bench
    a = argparse.Namespace(gather_tiles=8)
    n4 = bench.hbm_need(a, "cfg4", 1095, 8, True)
    assert round(n4["x"] / 1e9, 1) == 57.4 and round(n4["y"] / 1e9, 1) == 110.2 and 30e9 < n4["ring"] < 2.1 * bench.RING_SLOT_BYTES
    assert n4["total"] < 288e9 and bench.hbm_need(a, "cfg4", 1095, 8, False)["ring"] == 0
    n5 = bench.hbm_need(a, "cfg5", 12741, 8, True)
    assert round(n5["x"] / 1e9, 1) == 105.8 and round(n5["y"] / 1e9, 1) == 26.4 and n5["total"] < 288e9
    out = run([sys.executable, BENCH, "--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1", "--comm", "torch",
               "--config-steps", "2"], timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json_line(out)
    assert line["n_gpus"] == 8 and len(json.dumps(line)) < 4096
    assert line["baseline_configs"]["cfg4"]["hbm_needed_gb"] == round(n4["total"] / 1e9, 1)
    assert line["baseline_configs"]["cfg5"]["hbm_needed_gb"] == round(n5["total"] / 1e9, 1)
    # a GPU with 180 GB free: config 4 (206 GB on rank 0, 171 GB elsewhere) is skipped by ALL ranks, config 5 runs
    out = run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1", "--comm", "torch",
               "--config-steps", "2"], env={"SMM_BENCH_TEST_FREE_GB": "180"}, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json_line(out)
    c4 = line["baseline_configs"]["cfg4"]
    assert "needs" in c4["skipped"] and "value" not in c4 and c4["hbm_needed_gb"] > 170
    assert line["baseline_configs"]["cfg5"]["value"] > 0 and line["value"] > 0


def test_reference_sized_cases_of_the_bench_are_built_from_the_committed_fixtures():
    """tools/user_path_bench.py (blocks `reference_sized` / `host_to_host` of the bench line) needs a GPU to run; what
    can break without one -- the fixtures, their shapes, the record counts of speed-evaluation.ipynb cell 5 -- is
    checked here."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import user_path_bench as u
    cases = u._cases()
    assert {n: f.shape for n, (_, f, _, _) in cases.items()} == {
        "2t_era5": (12, 73, 144), "tas_healpix2": (12, 12288), "tas_ecearth": (12, 256, 512),
        "temp3d_fesom": (12, 3, 3140), "ua_ipsl_nan": (2, 19, 143, 144)}
    assert cases["ua_ipsl_nan"][2] == "r90x45" and cases["ua_ipsl_nan"][3] == {"check_nan": True}     # basic_test.py:95-102
    assert all(t == "r360x180" for n, (_, _, t, _) in cases.items() if n != "ua_ipsl_nan")
    ds = cases["temp3d_fesom"][0]
    assert ds["lon_bnds"].shape == (3140, 16) and ds["temp"].coords["lon"].attrs["bounds"] == "lon_bnds"
    import numpy as np
    a = u._tile_time(np.arange(6, dtype=np.float32).reshape(2, 3), 5)
    assert a.shape == (5, 3) and np.array_equal(a[2], a[0] + 1) and np.array_equal(a[4], a[0] + 2)


def _strings(obj):
    if isinstance(obj, str):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _strings(v)
    elif isinstance(obj, list):
        for v in obj:
            yield from _strings(v)


def test_a_hanging_gather_ends_non_zero_with_the_line_still_printed():
    """The gather watchdog: a rank that never reaches the collective costs the gather figures and the exit
    status (a hung RCCL must not look like a clean run), never the compute value."""
    out = run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1", "--comm", "torch",
               "--gather-timeout", "4"], env={"SMM_BENCH_TEST_HANG_GATHER": "1"}, timeout=120)
    assert out.returncode != 0
    line = json_line(out)
    assert line["value"] > 0 and "error" in line["with_gather"]


def test_secondary_entries_keep_their_units():
    """the roofline block of a secondary workload is nested, it no longer overwrites the
    entry's cells/s unit with GB/s; the layout summary of the final line is built from it."""
    sys.path.insert(0, ROOT)
    import bench
    entry = {"value": 1.4e11, "unit": "cells/s", "create_ms": 12.34,
             "roofline": {"kernel_ms": 1.7, "frac": 0.68, "traffic": 9.35e9, "algorithmic_bytes": 9.33e9, "unit": "GB/s"},
             "spot_check": {"bit_equal_to_oracle": True}}
    assert entry["unit"] == "cells/s"
    lay = bench.layout_summary(entry)
    assert lay == {"ms": 1.7, "frac": 0.68, "traffic_ratio": 1.002, "spot_check": True}
    out = {"roofline": {"kernel_ms": 2.75, "frac": 0.42, "traffic": 17.05e9, "algorithmic_bytes": 9.33e9},
           "spot_check": {"bit_equal_to_oracle": True}, "config": {"workload": "x" * 300, "plan": {"a": 1}}}
    cfg3 = {"value": 6e10, "unit": "cells/s", "roofline": {"kernel_ms": 11.4, "frac": 0.47, "traffic": 68.8e9, "algorithmic_bytes": 43.0e9,
                                                             "layout_floor_ratio": 1.31, "refetch_ratio": 1.221},
            "spot_check": {"bit_equal_to_oracle": True}}
    c3h = {"time_steps": 24, "input_GB": 21.2, "same_bits": True, "spot_check": True,
           "packed": {"seconds": 0.1912345, "cells_per_s": 6.1e8, "host_GBs": 110.0, "chunks": 56, "stage_ms": {}},
           "whole_rows": {"seconds": 0.412, "cells_per_s": 2.83e8, "host_GBs": 51.0, "chunks": 24, "stage_ms": {}}}
    details = {"others": {"cfg2sb": entry, "cfg3": cfg3, "cfg3c": {"error": "boom"}, "cfg3_host": c3h},
               "baseline_configs": {"cfg4": {"value": 3.4e11, "kernel_ms": 39.6, "frac": 0.5172, "workload": "w", "steps": 3,
                                             "algorithmic_bytes": 163.68e9,
                                             "f32_out": {"kernel_ms": 28.123456, "algorithmic_bytes": 108.6e9, "frac": 0.48271,
                                                         "spot_check": True, "value": 4.9e11}}},
               "reference_sized": {"2t_era5": {"in": "12x73x144", "init_ms": 3.2, "regrid_ms": 0.41, "cpu_scipy_ms": 5.5,
                                               "cpu_c1_ms": 2.2, "bit_equal": True, "cells": 777600, "calls": 25},
                                   "tas_ecearth": {"error": "x" * 300}},
               "host_to_host": {"rows": 512, "reps": 9, "ceilings": {"pcie_h2d_GBs": 57.5, "host_copy_GBs": 110.4},
                                "pinned_packed": {"cells_per_s": 9.87654e8, "cells_per_s_best": 1.0123e9, "host_GBs": 123.456, "ms": 33.0,
                                                  "pcie_frac": 0.731, "host_mem_frac": 0.812,
                                                  "stage_ms": {"stage_in": 15.76, "h2d": 23.22, "kernel": 0.29, "d2h": 4.78,
                                                               "copy_out": 0.0, "wait": 9.67, "total": 29.69}},
                                "pcie_bytes_per_row": {"packed": 2592000, "whole_rows": 8824320}, "spot_check": True,
                                "staging_threads": 16, "bound": "packed: PCIe", "cpu_cells_per_s": 8.5e8}}
    line = bench.final_line(out, details)
    r = line["roofline"]
    # the record keeps the first two dozen scalars of a nested block and the tail of the output: scalars first,
    # nested summaries last, the user-path and per-config blocks at the end of the line; the per-workload fractions are
    # not repeated as scalars and the line stays well inside its 4-kB budget
    keys = list(r)
    assert keys.index("batch_fastest_frac") < keys.index("layouts") < keys.index("configs") == len(keys) - 1
    assert not [k for k in keys if k.startswith("cfg")] and all(not isinstance(r[k], dict) for k in keys[:-2])
    assert len(json.dumps(line)) < 3500
    assert list(line)[-3:] == ["reference_sized", "baseline_configs", "host_to_host"]
    assert line["reference_sized"]["2t_era5"] == ["12x73x144", 3.2, 0.41, 5.5, 2.2, True] and "regrid() host->host" in line["reference_sized"]["cols"]
    assert len(line["reference_sized"]["tas_ecearth"]["error"]) <= 60
    h = line["host_to_host"]
    assert h["pinned_packed"] == [988, 1012, 0.73, 0.81, 15.8, 23.2, 0.3, 4.8, 0.0, 9.7, 29.7] and "Mcells/s med, best" in h["cols"]
    assert h["spot_check"] is True and h["ceil_GBs"] == {"h2d": 57.5, "host_copy": 110.4} and h["staging_threads"] == 16
    assert h["reps"] == 9 and h["cpu_Mcells_per_s"] == 850
    assert h["cfg3_levels"] == {"steps": 24, "in_GB": 21.2, "packed": [610, 0.191], "whole_rows": [283, 0.412], "ok": True}
    assert "cfg3_host" not in r["configs"]
    c4 = line["baseline_configs"]["cfg4"]
    assert c4["frac"] == 0.5172 and c4["alg_GB"] == 163.68
    assert c4["f32_out"] == {"ms": 28.123, "frac": 0.4827, "alg_GB": 108.6, "spot_check": True}
    assert r["configs"]["cfg3"] == {"ms": 11.4, "frac": 0.47, "floor": 1.31, "refetch": 1.221}
    assert r["spot_checks_bit_equal"] == "3 of 3" and "spot_check" not in r["layouts"]["batch_fastest"]
    assert r["layouts"]["batch_fastest"]["frac"] == 0.68 and r["batch_fastest_frac"] == 0.68
    assert r["traffic_ratio"] == 1.827 and r["batch_fastest_traffic_ratio"] == 1.002 and "traffic_ratio" not in r["layouts"]["native"]
    assert r["configs"]["cfg3c"] == {"error": "boom"}
    assert len(line["config"]["workload"]) <= 100 and "plan" not in line["config"]


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
    for attempt in range(2):        # the probed port (and port + 23 of the rendezvous) can be taken between probe and bind: one retry
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        out = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--dry-run",
                   "--steps", "2", "--warmup", "1", "--gather", "none"], timeout=300)
        if out.returncode == 0 or "Address already in use" not in out.stderr:
            break
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
