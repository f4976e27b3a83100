#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants in ONE process (HIP events, median/min).

  python tools/tune.py --workload cfg2 --batch 3600 --rounds 7 \
      --variants sell tile:tile_walk=16 tile:tile_walk=16,tile_x_loads=1 tile:tile_staging=2,tile_rows_per_step=1

variant = kernel[:knob=value,...] with the named knobs of smm_debug_set_tuning (smmregrid_amd/_lib.py: TUNE_KNOBS).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench import WORKLOADS, algorithmic_bytes  # noqa: E402
from smmregrid_amd import SparseOperator, _lib, gridgen  # noqa: E402
from smmregrid_amd.device import DeviceArray, Event, synchronize  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--variants", nargs="+", default=["sell", "tile", "tile:tile_walk=16", "tile:tile_x_loads=1"])
    ap.add_argument("--check", action="store_true", help="compare every variant's Y with the first one")
    args = ap.parse_args()
    method, sgrid, tgrid, n_batch, x_dtype = WORKLOADS[args.workload][:5]      # 2-D workloads, native layout
    n_batch = args.batch or n_batch
    w = gridgen.generate_weights(sgrid, tgrid, method=method)
    n_src, n_dst = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
    op = SparseOperator(n_src, n_dst, w["src_address"].values, w["dst_address"].values,
                        w["remap_matrix"].values, device=0)
    op.set_epilogue(w["dst_grid_imask"].values, w["dst_grid_frac"].values)
    dt = np.float64 if x_dtype == "f64" else np.float32
    x = DeviceArray((n_batch, n_src), dt).fill_random(20260723, 250.0, 30.0)
    y = DeviceArray((n_batch, n_dst), np.float64)
    print("plan:", op.plan_info(), "nnz", op.nnz, "U", op.n_used_src, file=sys.stderr)

    def flags_of(v):
        return {"sell": _lib.APPLY_KERNEL_SELL, "tile": _lib.APPLY_KERNEL_TILE, "auto": 0}[v.split(":")[0]]

    def knobs_of(v):
        spec = v.split(":", 1)[1] if ":" in v else ""
        return _lib.tuning(**{k: int(n) for k, n in (kv.split("=") for kv in spec.split(",") if kv)})

    times = {v: [] for v in args.variants}
    ref = None
    for v in args.variants:   # warm-up + optional check
        with knobs_of(v):
            op.apply(x, y=y, remap_area_min=0.5, flags=flags_of(v))
        synchronize()
        if args.check:
            h = y.rows(0, min(n_batch, 8)).to_host()
            if ref is None:
                ref = h
            else:
                assert np.array_equal(h.view(np.uint64), ref.view(np.uint64)), f"{v} differs"
    for _ in range(args.rounds):
        for v in args.variants:
            a, b = Event(), Event()
            a.record()
            with knobs_of(v):
                op.apply(x, y=y, remap_area_min=0.5, flags=flags_of(v))
            b.record()
            b.synchronize()
            times[v].append(a.elapsed_ms(b))
    b_alg = algorithmic_bytes(op, n_batch, np.dtype(dt).itemsize, 8)
    out = {}
    for v, t in times.items():
        med, mn = float(np.median(t)), float(np.min(t))
        out[v] = {"median_ms": med, "min_ms": mn, "alg_GBs_median": b_alg / med / 1e6,
                  "cells_per_s": n_batch * n_dst / med * 1e3}
        print(f"{v:>14s}  median {med:8.3f} ms  min {mn:8.3f} ms  alg {b_alg / med / 1e6:8.1f} GB/s",
              file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
