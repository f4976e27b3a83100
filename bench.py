#!/usr/bin/env python3
"""Benchmark of the regrid apply path (BASELINE.json metric: regridded cells/s).

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (smm_apply) over the whole batch of the
workload: config 2 of BASELINE.json by default -- r1440x721 -> r360x180
bilinear, 3600 time steps, f64 -- with X and Y resident in HBM.  For N > 1
(launched by torch.distributed.run, one rank per GPU) every rank regrids its
own 3600-step shard of the time axis (weak scaling; batch rows are independent,
SURVEY 8e) and the Y shards are gathered to rank 0 with RCCL unless
--gather none.

Rank 0 prints ONE JSON line with the driver's contract plus `roofline`
(HBM bound, algorithmic bytes of SURVEY 8d / live HIP-event kernel time) and
`cpu_baseline` (the CPU oracle -- a port of the reference's step sequence --
timed on this host's cores over a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (method, source grid, target grid, batch, x dtype)
    "cfg2": ("bil", "r1440x721", "r360x180", 3600, "f64"),
    "cfg5tile": ("con", "r1440x721", "r720x360", 1024, "f64"),   # config-5 geometry, one GPU's worth of rows
    "cfg1": ("bil", "r180x90", "r90x45", 1, "f64"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="override the batch rows per GPU")
    ap.add_argument("--kernel", default="auto", choices=["auto", "sell", "tile"])
    ap.add_argument("--variant", type=int, default=0, help="kernel variant knob (0 = default)")
    ap.add_argument("--jpb", type=int, default=0, help="tile kernel: batch rows per workgroup")
    ap.add_argument("--gather", default="root", choices=["root", "none"],
                    help="N>1: RCCL gather of the Y shards to rank 0 inside the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"),
                    help="PMC-derived HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                         "passes, corrected per MI355X_MICROARCH.md), keyed by workload/batch/kernel")
    return ap.parse_args()


def algorithmic_bytes(op, n_batch, sx, sy):
    """B*(U*sx + D*sy) + nnz*(8+4) + (D+1)*4."""
    return n_batch * (op.n_used_src * sx + op.n_dst * sy) + op.nnz * 12 + (op.n_dst + 1) * 4


def cpu_baseline(weights, n_batch_full, x_dtype, budget_s):
    """Oracle (C port of regrid.py:545-570, OpenMP over batch rows) on a bounded sample:
    a fixed block of distinct batch rows of the same workload, passed repeatedly until
    about `budget_s` seconds of CPU work have been timed."""
    from oracle import oracle
    n_src, n_dst = weights.sizes["src_grid_size"], weights.sizes["dst_grid_size"]
    csr = oracle.coo_to_csr_c(n_src, n_dst, weights["src_address"].values,
                              weights["dst_address"].values, weights["remap_matrix"].values)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    threads = int(os.environ.get("SMM_CPU_THREADS", min(avail, 16)))  # one GPU's CPU share
    rng = np.random.default_rng(20260723)
    dt = np.float64 if x_dtype == "f64" else np.float32
    rows = int(min(n_batch_full, max(threads * 8, 128)))
    x = np.empty((rows, n_src), dtype=dt)
    for r in range(rows):
        x[r] = 250.0 + 30.0 * rng.standard_normal(n_src, dtype=np.float32 if dt == np.float32 else np.float64)
    frac = weights["dst_grid_frac"].values
    oracle.apply_c(csr, x[:threads], False, None, frac, 0.5, threads=threads)   # warm the team
    passes, spent = 0, 0.0
    y = None
    while spent < budget_s and passes < 1000:
        t0 = time.perf_counter()
        y = oracle.apply_c(csr, x, False, None, frac, 0.5, threads=threads)
        spent += time.perf_counter() - t0
        passes += 1
    return {"value": passes * rows * n_dst / spent, "unit": "cells/s", "cores": threads,
            "kind": "port",
            "sample": f"{rows} of {n_batch_full} batch rows x {passes} passes, oracle/oracle.c "
                      f"(OpenMP over rows, {threads} threads of {avail} visible), {spent:.1f} s"}, y


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from smmregrid_amd import SparseOperator, _lib, gridgen
    from smmregrid_amd.device import DeviceArray, Event, set_device, synchronize, device_name

    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    set_device(local_rank)

    method, sgrid, tgrid, n_batch, x_dtype = WORKLOADS[args.workload]
    if args.batch:
        n_batch = args.batch
    weights = gridgen.generate_weights(sgrid, tgrid, method=method)
    n_src, n_dst = weights.sizes["src_grid_size"], weights.sizes["dst_grid_size"]
    op = SparseOperator(n_src, n_dst, weights["src_address"].values, weights["dst_address"].values,
                        weights["remap_matrix"].values, device=local_rank)
    op.set_epilogue(weights["dst_grid_imask"].values, weights["dst_grid_frac"].values)
    np_dt = np.float64 if x_dtype == "f64" else np.float32
    sx = np.dtype(np_dt).itemsize

    # device-resident fields; with N > 1 torch owns the Y buffer so RCCL can move it
    x = DeviceArray((n_batch, n_src), np_dt)
    x.fill_random(seed=20260723 + 1000003 * rank, mean=250.0, sigma=30.0)
    y_t = None
    if world > 1:
        y_t = torch.empty((n_batch, n_dst), dtype=torch.float64, device=f"cuda:{local_rank}")
        y = DeviceArray((n_batch, n_dst), np.float64, ptr=y_t.data_ptr())
        gathered = None
        if args.gather == "root" and rank == 0:
            gathered = [torch.empty_like(y_t) for _ in range(world)]
    else:
        y = DeviceArray((n_batch, n_dst), np.float64)

    flags = {"auto": 0, "sell": _lib.APPLY_KERNEL_SELL, "tile": _lib.APPLY_KERNEL_TILE}[args.kernel]
    flags |= (args.variant << 16) | (args.jpb << 20)
    area_min = 0.5

    def step():
        op.apply(x, y=y, masked=False, remap_area_min=area_min, flags=flags)
        if world > 1 and args.gather == "root":
            torch.cuda.current_stream().synchronize()   # kernel ran on the null stream: ordered already
            dist.gather(y_t, gathered, dst=0)

    def barrier():
        synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()

    # timed region: exactly K steps; per-launch kernel time from HIP events on the launch stream
    ev = [(Event(), Event()) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        op.apply(x, y=y, masked=False, remap_area_min=area_min, flags=flags)
        ev[k][1].record()
        if world > 1 and args.gather == "root":
            dist.gather(y_t, gathered, dst=0)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [a.elapsed_ms(b) for a, b in ev]

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        cells = float(n_dst) * n_batch * world * args.steps
        k_avg = float(np.mean(kernel_ms)) * 1e-3
        b_alg = algorithmic_bytes(op, n_batch, sx, 8)
        achieved = b_alg / k_avg / 1e9
        traffic = None
        if args.traffic_json and os.path.exists(args.traffic_json):
            key = f"{args.workload}/{n_batch}/{args.kernel}/{args.variant}"
            traffic = json.load(open(args.traffic_json)).get(key, {}).get("hbm_bytes_per_launch")
        out = {
            "metric": "regridded cells/sec (dst_pts x time x lev)",
            "value": cells / elapsed,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": x_dtype,
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {sgrid}->{tgrid} {method}, {n_batch} batch rows "
                                   f"per GPU, {x_dtype} in / f64 out, X and Y resident in HBM",
                       "S": n_src, "D": n_dst, "nnz": op.nnz, "U": op.n_used_src,
                       "kernel": args.kernel, "plan": op.plan_info(), "gather": args.gather if world > 1 else "n/a",
                       "device": device_name(local_rank)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": k_avg * 1e3, "algorithmic_bytes": b_alg,
                         "full_stream_bytes": n_batch * (n_src * sx + n_dst * 8) + op.nnz * 12},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"], _ = cpu_baseline(weights, n_batch, x_dtype, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
