#!/usr/bin/env python3
"""Benchmark of the regrid apply path (BASELINE.json metric: regridded cells/s).

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the whole batch of the workload.
Default workload = config 2 of BASELINE.json (the configuration the metric is
quoted on): r1440x721 -> r360x180 bilinear, 3600 time steps, f64, X and Y
resident in HBM.  Other workloads (--workload, see WORKLOADS): cfg3 / cfg3c / cfg3sb (masked ocean
levels, one grouped launch; rows on 128-B lines / packed / every level kept batch-fastest), cfg2sb /
cfg2sbk (config 2 with the field kept batch-fastest: the opt-in operand layout for device-resident
producers), cfg4s / cfg5tile (config-4 / config-5 geometry, reduced batch), cfg4 / cfg5 (one GPU's
share of BASELINE configs 4 / 5: with --gpus 8 these are the fixed-total-batch lines of BASELINE.json),
cfg1, upsample, conhi.

For N > 1 there is one rank per GPU: either the caller started them (torch.distributed.run /
any launcher that exports RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT), or -- when
`--gpus N` is given and WORLD_SIZE is not set -- this process starts N rank processes itself,
before touching any GPU, relays rank 0's JSON line and exits non-zero if any rank does.  Every rank
regrids its own full-size shard of the time axis (weak scaling; batch rows are independent,
SURVEY 8e) and the Y shards stay resident on their GPUs: `value` is that job (no data-path
collective; barriers and the max over ranks go through a host-side TCP rendezvous).  A second timed
loop in the same run adds the RCCL gather of the Y shards to rank 0 after every step and is
reported as `with_gather` beside it (xGMI-link bound; --gather none skips it).  The plumbing of that
gather is the library's own RCCL communicator (--comm native, default: no torch in the process) or
torch.distributed (--comm torch).  `n_gpus` is the number of ranks that really joined.

Rank 0 prints TWO stdout lines.  The LAST one is the compact JSON line (< 4 kB, every string < 100 characters):
the driver's contract plus `roofline` (HBM bound; algorithmic bytes of SURVEY 8d over the live HIP-event kernel time;
`roofline.layouts` = config 2 in the native and the batch-fastest layout, `roofline.configs` = the secondary
workloads' fractions and traffic ratios), `cpu_baseline` (the CPU oracle -- a port of the reference's step sequence --
timed on this host's cores over a bounded sample of the same workload) and `baseline_configs` (one rank's share of
BASELINE configs 4 and 5 per GPU; at N > 1 with the per-rank kernel times and the gather figures).  The line before it,
prefixed `details: `, holds the bulky per-workload entries (`others`: config 2 with the field kept batch-fastest,
BASELINE config 3 packed / on 128-B lines / batch-fastest, config-4 geometry, config-5 geometry), each timed in the
same run with kernel time, algorithmic bytes, roofline fraction, replayed PMC traffic and a bit-equality spot check
of its timed output against the CPU oracle.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (method, source grid, target grid, batch rows, x dtype[, layout])
    # the driver's line: config 2 (headline) ...
    "cfg2": ("bil", "r1440x721", "r360x180", 3600, "f64"),
    # ... config 2 with the operand kept batch-fastest by a device-resident producer (smm_apply_sb): "sb" = X (S, B),
    # "sbk" = the result kept batch-fastest too, Y (D, B): what a chain of regrids passes on (SMM_APPLY_SB_Y_SB)
    "cfg2sb": ("bil", "r1440x721", "r360x180", 3600, "f64", "sb"),
    "cfg2sbk": ("bil", "r1440x721", "r360x180", 3600, "f64", "sbk"),
    # ... BASELINE config 3, masked levels: (method, source nx x ny, target, (time steps, levels), x dtype, layout).
    # "pad": field rows start on 128-B lines (row pitch rounded up to 16 doubles: the pitch the library's own H2D
    # staging uses); "cfg3c": rows packed back to back, every row of the 1442x1021 grid then starts mid-line
    # (S * 8 B = 80 mod 128); "cfg3sb": the field kept batch-fastest per level, X (L, S, T) (smm_group_apply_sb)
    "cfg3": ("con3d", (1442, 1021), "r360x180", (120, 75), "f64", "pad"),
    "cfg3c": ("con3d", (1442, 1021), "r360x180", (120, 75), "f64"),
    "cfg3sb": ("con3d", (1442, 1021), "r360x180", (120, 75), "f64", "sb"),
    # ... config-4 geometry (regular Gaussian n1280 = 5120x2560 -> HEALPix nside 1024, f32 in) with a reduced batch and
    # one GPU's share of BASELINE configs 4 / 5 (8760 / 8 rows of f32; 137 x 744 / 8 rows of f64)
    "cfg4s": ("bil", "n1280", "hp1024", 128, "f32"),
    "cfg4": ("bil", "n1280", "hp1024", 1095, "f32"),
    "cfg5tile": ("con", "r1440x721", "r720x360", 1024, "f64"),
    "cfg5": ("con", "r1440x721", "r720x360", 12741, "f64"),
    # not in the driver's line: config 1 (plumbing), a quick level group, and the two kernel forms the BASELINE configs
    # do not reach -- coarse -> fine (4-KB tiles, two batch rows per LDS-DMA step; Y-write bound) and 0.1 -> 1 degree
    # conservative (~120 links per row: rows split over lane groups)
    "cfg1": ("bil", "r180x90", "r90x45", 1, "f64"),
    "cfg3s": ("con3d", (1442, 1021), "r360x180", (16, 8), "f64"),
    "upsample": ("bil", "r360x180", "r1440x721", 1024, "f64"),
    "conhi": ("con", "r3600x1800", "r360x180", 256, "f64"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="override the batch rows per GPU")
    ap.add_argument("--kernel", default="auto", choices=["auto", "sell", "tile"])
    ap.add_argument("--tune", default="", help="tuning knobs for A/B runs, e.g. tile_walk=64,xcd_run=8 "
                                               "(smm_debug_set_tuning; names in smmregrid_amd/_lib.py TUNE_KNOBS)")
    ap.add_argument("--gather", default="root", choices=["root", "none"],
                    help="N>1: also time the steps followed by the RCCL gather of the Y shards to rank 0")
    ap.add_argument("--gather-tiles", type=int, default=8,
                    help="row tiles of the overlapped gather (tile k travels while tile k+1 is computed)")
    ap.add_argument("--comm", default="native", choices=["native", "torch"],
                    help="N>1 gather plumbing: the library's smm_comm_* over RCCL (no torch in the process) "
                         "or torch.distributed (nccl == RCCL; gloo with --dry-run)")
    ap.add_argument("--gather-timeout", type=float, default=300.0,
                    help="seconds everything after the headline's compute loop may take at N > 1 (communicator set-up, "
                         "the gather loops, the BASELINE configs) before the line is printed as it stands and the run "
                         "ends with status 3")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU, no kernel: a stand-in step exercises launcher, rendezvous, barriers, "
                         "max over ranks and the JSON line (CPU tests of the N>1 path)")
    ap.add_argument("--others", default="default",
                    help="secondary workloads timed after the headline at N=1 (comma list, 'default' or 'none')")
    ap.add_argument("--others-steps", type=int, default=20)
    ap.add_argument("--others-budget", type=float, default=240.0,
                    help="seconds after which remaining secondary workloads are skipped")
    ap.add_argument("--configs", default="default",
                    help="BASELINE configs run after the headline on the same ranks, each rank its share of the "
                         "fixed total batch (comma list, 'default' = cfg4,cfg5, 'none')")
    ap.add_argument("--config-steps", type=int, default=10)
    ap.add_argument("--config-warmup", type=int, default=2)
    ap.add_argument("--config-gather-steps", type=int, default=2,
                    help="steps of the compute + gather loop of a BASELINE config (config 4 moves 110 GB per rank and step)")
    ap.add_argument("--config-batch", type=int, default=0, help="override the rows per GPU of --configs (tests)")
    ap.add_argument("--user-path", default="default", choices=["default", "none"],
                    help="N = 1: also time the user-visible paths host to host (tools/user_path_bench.py): smm_apply_host on "
                         "config-2 rows (block `host_to_host`) and Regridder.regrid on the reference's own fields with the CPU "
                         "oracle beside them (block `reference_sized`)")
    ap.add_argument("--host-rows", type=int, default=512, help="batch rows of the host_to_host block")
    ap.add_argument("--host-level-steps", type=int, default=40,
                    help="time steps of the config-3 field held in host memory (75 levels each; 40 = 35 GB; packing needs >= 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"),
                    help="PMC-derived HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                         "passes, corrected per MI355X_MICROARCH.md), keyed by workload/batch/kernel")
    return ap.parse_args()


def algorithmic_bytes(op, n_batch, sx, sy):
    """B*(U*sx + D*sy) + nnz*(8+4) + (D+1)*4."""
    return n_batch * (op.n_used_src * sx + op.n_dst * sy) + op.nnz * 12 + (op.n_dst + 1) * 4


def distinct_lines(used_cells, itemsize, phases=(0,)):
    """Distinct 128-B lines of one field row that hold a used source cell: the floor of what ANY kernel must fetch per
    batch row in the native (B, S) layout (a union over the operator, so lines shared by neighbouring destination blocks
    count once -- what a tile plan stages beyond it is halo re-fetch).  `phases` = byte offsets of the row's first cell
    inside its line; rows packed back to back cycle through several, the mean is returned."""
    cells = np.asarray(used_cells, dtype=np.int64) * itemsize
    return float(np.mean([np.unique((cells + int(ph)) // 128).size for ph in phases]))


class Problem2D:
    """One operator, X (B, S) -> Y (B, D)."""

    def __init__(self, name, device, rank, batch=None):
        from smmregrid_amd import SparseOperator, gridgen
        from smmregrid_amd.device import DeviceArray
        method, sgrid, tgrid, n_batch, self.x_dtype = WORKLOADS[name][:5]
        self.layout = WORKLOADS[name][5] if len(WORKLOADS[name]) > 5 else "bs"
        self.y_dt = np.float64     # the reference's result_type(x, f64), regrid.py:550
        self.n_batch = batch or n_batch
        self.weights = gridgen.generate_weights(sgrid, tgrid, method=method)
        w = self.weights
        self.n_src, self.n_dst = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
        self.op = SparseOperator(self.n_src, self.n_dst, w["src_address"].values,
                                 w["dst_address"].values, w["remap_matrix"].values, device=device,
                                 dst_dims=w["dst_grid_dims"].values)
        self.op.set_epilogue(w["dst_grid_imask"].values, w["dst_grid_frac"].values)
        self.create_ms = self.op.create_ms     # smm_operator_create: sort + duplicate sum + layouts + upload
        self.np_dt = np.float64 if self.x_dtype == "f64" else np.float32
        x_shape = {"bs": (self.n_batch, self.n_src), "sb": (self.n_src, self.n_batch),
                   "sbk": (self.n_src, self.n_batch)}[self.layout]
        self.x = DeviceArray(x_shape, self.np_dt)
        self.x.fill_random(seed=20260723 + 1000003 * rank, mean=250.0, sigma=30.0)
        if self.layout != "bs":
            self.op.prepare_sb()
        self.y_shape = (self.n_dst, self.n_batch) if self.layout == "sbk" else (self.n_batch, self.n_dst)
        lay = {"bs": "X (B, S) native layout", "sb": "X (S, B) batch-fastest", "sbk": "X (S, B) batch-fastest"}
        self.desc = (f"{name}: {sgrid}->{tgrid} {method}, {self.n_batch} batch rows per GPU, "
                     f"{self.x_dtype} in / {'f32' if self.y_dt == np.float32 else 'f64'} out, {lay[self.layout]}, "
                     f"{'Y (D, B) kept batch-fastest' if self.layout == 'sbk' else 'Y (B, D)'}, X and Y resident in HBM")
        self.meta = {"S": self.n_src, "D": self.n_dst, "nnz": self.op.nnz, "U": self.op.n_used_src,
                     "plan": self.op.plan_info()}

    def cells(self):
        return float(self.n_dst) * self.n_batch

    def alg_bytes(self):
        return algorithmic_bytes(self.op, self.n_batch, np.dtype(self.np_dt).itemsize, np.dtype(self.y_dt).itemsize)

    def full_stream_bytes(self):
        return (self.n_batch * (self.n_src * np.dtype(self.np_dt).itemsize + self.n_dst * np.dtype(self.y_dt).itemsize)
                + self.op.nnz * 12)

    def line_bytes(self):
        """Bytes of the whole 128-B source lines the links touch (what any kernel must move
        from HBM in the native (B, S) layout) + Y + the operator once."""
        if self.layout != "bs":
            return None                   # batch-fastest: every needed cell is a contiguous run
        staged = self.op.plan_info()["staged_src_elems"]
        if not staged:
            return None
        return (self.n_batch * (staged * np.dtype(self.np_dt).itemsize + self.n_dst * np.dtype(self.y_dt).itemsize)
                + self.op.nnz * 12)

    def distinct_line_bytes(self):
        """Layout floor: the union of distinct 128-B lines of X that carry a used cell, + Y + the operator once."""
        if self.layout != "bs":
            return None
        isz = np.dtype(self.np_dt).itemsize
        row = self.n_src * isz
        phases = sorted({(r * row) % 128 for r in range(32)})      # rows back to back: the start offsets they cycle through
        lines = distinct_lines(self.op.used_sources(), isz, phases)
        return self.n_batch * (lines * 128 + self.n_dst * np.dtype(self.y_dt).itemsize) + self.op.nnz * 12

    def run(self, y, flags):
        if self.layout == "bs":
            self.op.apply(self.x, y=y, masked=False, remap_area_min=0.5, flags=flags)
        else:
            self.op.apply_sb(self.x, y=y, masked=False, remap_area_min=0.5, flags=flags,
                             keep_batch_fastest=self.layout == "sbk")

    def spot_check(self, y, max_dst=65536):
        """One batch row of the timed output against oracle/oracle.c (bit equality; NaN positions
        identical).  The oracle builds its own CSR from the links (restricted to the first
        `max_dst` destination cells for the 50-M-link operator)."""
        from oracle import oracle
        from smmregrid_amd import _lib
        import ctypes
        r = self.n_batch // 2
        isz = np.dtype(self.np_dt).itemsize
        if self.layout == "bs":
            xrow = self.x.rows(r, r + 1).to_host().reshape(-1)[:self.n_src]
        else:                                   # column r of the (S or U, B) field
            n_rows = self.x.shape[0]
            col = np.empty(n_rows, dtype=self.np_dt)
            _lib.call("smm_memcpy2d_d2h", col.ctypes.data_as(ctypes.c_void_p), isz,
                      ctypes.c_void_p(self.x.ptr + r * isz), self.n_batch * isz, isz, n_rows, None)
            xrow = col
        if self.layout == "sbk":                # column r of the (D, B) result
            got = np.empty(self.n_dst, dtype=np.float64)
            _lib.call("smm_memcpy2d_d2h", got.ctypes.data_as(ctypes.c_void_p), 8, ctypes.c_void_p(y.ptr + r * 8),
                      self.n_batch * 8, 8, self.n_dst, None)
        else:
            got = y.rows(r, r + 1).to_host().reshape(-1)
        w = self.weights
        n_chk = int(min(self.n_dst, max_dst))
        dst = w["dst_address"].values
        keep = slice(None) if n_chk == self.n_dst else (dst <= n_chk)
        rm = w["remap_matrix"].values
        csr = oracle.coo_to_csr_c(self.n_src, n_chk, w["src_address"].values[keep], dst[keep],
                                  (rm[:, 0] if rm.ndim == 2 else rm)[keep])
        ref = oracle.apply_c(csr, xrow[None, :], False, None, w["dst_grid_frac"].values[:n_chk], 0.5)[0]
        got = got[:n_chk]
        ref = ref.astype(got.dtype)              # an f32 store is the rounded f64 result
        same = np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(got[~np.isnan(ref)], ref[~np.isnan(ref)])
        return {"batch_row": int(r), "cells": n_chk, "bit_equal_to_oracle": bool(same)}

    def free(self):
        self.x.free()
        self.op.close()

    def run_rows(self, y, flags, r0, r1):
        """The same product restricted to batch rows [r0, r1) (one tile of the overlapped gather)."""
        if self.layout == "bs":
            self.op.apply(self.x.rows(r0, r1), y=y.rows(r0, r1), masked=False, remap_area_min=0.5, flags=flags)
            return
        if self.layout == "sbk":
            raise SystemExit("the tiled gather moves (B, D) row tiles: use cfg2 / cfg2sb with --gpus N")
        import ctypes
        from smmregrid_amd import _lib
        from smmregrid_amd.device import dtype_code
        isz = np.dtype(self.np_dt).itemsize          # batch rows are a column range of the (S, B) field
        fl = flags
        _lib.call("smm_apply_sb", self.op.handle, ctypes.c_void_p(self.x.ptr + r0 * isz), dtype_code(self.x.dtype),
                  self.n_batch, ctypes.c_void_p(y.rows(r0, r1).ptr), dtype_code(y.dtype), self.n_dst, r1 - r0,
                  0.5, fl, None)

    def cpu_baseline(self, budget_s):
        """The reference's step sequence (regrid.py:545-570) on this host's cores, over a bounded
        sample of the same workload (a block of distinct batch rows passed repeatedly).  Three legs
        (SURVEY 8d): numpy + scipy.sparse on one core, oracle/oracle.c with OpenMP on this GPU's CPU
        share (16 threads) and on every core the process may really use (affinity capped by the
        cgroup CPU quota; one leg when the two coincide)."""
        from oracle import oracle
        w = self.weights
        csr = oracle.coo_to_csr_c(self.n_src, self.n_dst, w["src_address"].values,
                                  w["dst_address"].values, w["remap_matrix"].values)
        threads, avail = cpu_threads()
        rng = np.random.default_rng(20260723)
        rows = int(min(self.n_batch, max(threads * 8, avail, 128)))   # >= one row per thread of the widest leg
        x = np.empty((rows, self.n_src), dtype=self.np_dt)
        for r in range(rows):
            x[r] = 250.0 + 30.0 * rng.standard_normal(self.n_src, dtype=self.np_dt)
        frac = w["dst_grid_frac"].values

        def timed(fn, n_rows, budget, max_passes=1000):
            passes, spent = 0, 0.0
            while spent < budget and passes < max_passes:
                t0 = time.perf_counter()
                fn()
                spent += time.perf_counter() - t0
                passes += 1
            return passes * n_rows * self.n_dst / spent, passes, spent

        legs = []
        # (i) numpy + scipy.sparse, single thread: the closest stand-in for the reference's own
        # single-threaded per-block product
        srows = min(rows, 32)
        v, p, t = timed(lambda: oracle.apply(csr, x[:srows], False, None, frac, 0.5), srows, budget_s * 0.25, 50)
        legs.append({"value": v, "unit": "cells/s", "cores": 1, "kind": "port",
                     "impl": "oracle/oracle.py (numpy + scipy.sparse CSR product)",
                     "rows_sampled": srows, "passes": p, "seconds": round(t, 1),
                     "sample": f"{srows} of {self.n_batch} batch rows x {p} passes, {t:.1f} s"})
        # (ii) C port with OpenMP over batch rows: this GPU's CPU share, then every visible core
        for nt in sorted({threads, avail}):
            oracle.apply_c(csr, x[:min(rows, nt)], False, None, frac, 0.5, threads=nt)  # warm the team
            v, p, t = timed(lambda: oracle.apply_c(csr, x, False, None, frac, 0.5, threads=nt), rows,
                            budget_s * 0.375)
            # ("every core the process may use" = affinity capped by the cgroup CPU quota: cpu_threads())
            legs.append({"value": v, "unit": "cells/s", "cores": nt, "kind": "port",
                         "impl": "oracle/oracle.c (OpenMP over batch rows)",
                         "rows_sampled": rows, "passes": p, "seconds": round(t, 1),
                         "sample": f"{rows} of {self.n_batch} batch rows x {p} passes, {nt} threads of "
                                   f"{avail} usable (affinity capped by the cgroup quota; os.cpu_count() = "
                                   f"{os.cpu_count()}), {t:.1f} s"})
        return cpu_summary(legs, avail, f"of {self.n_batch} batch rows")


class ProblemLevels:
    """Config 3: per-level ocean masks, X (T, L, S) -> Y (T, L, D) in one grouped launch."""

    def __init__(self, name, device, rank, batch=None):
        from smmregrid_amd import OperatorGroup, gridgen
        from smmregrid_amd.device import DeviceArray
        from smmregrid_amd.weights import compute_weights_matrix3d
        _, (nx, ny), tgrid, (n_t, n_lev), self.x_dtype = WORKLOADS[name][:5]
        self.n_t, self.n_lev = batch or n_t, n_lev
        src = gridgen.regular_grid(nx, ny, name=f"tripolar-like {nx}x{ny}")
        masks = gridgen.synthetic_ocean_masks(nx, ny, n_lev)
        levels = np.arange(n_lev, dtype=np.float64)
        w3 = gridgen.ConservativeLevels(src, tgrid).stack(masks, levels)
        self.weights = w3
        self.n_src, self.n_dst = w3.sizes["src_grid_size"], w3.sizes["dst_grid_size"]
        t0 = time.perf_counter()
        self.ops = compute_weights_matrix3d(w3, "lev", device=device)
        self.create_ms = (time.perf_counter() - t0) * 1e3      # all levels' operators (created by a thread pool)
        self.create_ms_sum = float(sum(op.create_ms for op in self.ops))
        self.dst_imask = np.stack([op.mask_apply(masks[i]) for i, op in enumerate(self.ops)])
        frac = w3["dst_grid_frac"].values
        for i, op in enumerate(self.ops):
            op.set_epilogue(self.dst_imask[i], frac[i])
        self.group = OperatorGroup(self.ops)
        self.masked_levels = (~(self.dst_imask == 1).all(axis=1)).astype(np.uint8)
        self.level_index = np.arange(n_lev, dtype=np.int32)
        # one time slab on the host (NaN on land per level), replicated over time on the device
        rng = np.random.default_rng(20260723 + rank)
        slab = (10.0 + 5.0 * rng.standard_normal((n_lev, self.n_src), dtype=np.float32)).astype(np.float64)
        slab[masks == 0] = np.nan
        self.slab, self.masks = slab, masks
        self.layout = "sb" if len(WORKLOADS[name]) > 5 and WORKLOADS[name][5] == "sb" else "bs"
        self.nx, self.ny, self.tgrid = nx, ny, tgrid
        self.y_shape = (self.n_t, 1, n_lev, self.n_dst)
        self.np_dt = np.float64
        nnz = sum(op.nnz for op in self.ops)
        self.meta = {"S": self.n_src, "D": self.n_dst, "nnz_total": nnz, "levels": n_lev,
                     "max_row_nnz": max(op.max_row_nnz for op in self.ops),
                     "plan": self.ops[0].plan_info()}
        self.x = None
        self.set_layout(name, len(WORKLOADS[name]) > 5 and WORKLOADS[name][5] == "pad")

    def set_layout(self, name, padded):
        """(Re)allocate the field: rows on 128-B lines (`cfg3`) or packed back to back (`cfg3c`)."""
        from smmregrid_amd import _lib
        from smmregrid_amd.device import DeviceArray
        import ctypes
        if self.x is not None:
            self.x.free()
        self.name, self.padded = name, bool(padded)
        self.layout = "sb" if len(WORKLOADS[name]) > 5 and WORKLOADS[name][5] == "sb" else "bs"
        self.y_shape = (self.n_t, 1, self.n_lev, self.n_dst)
        n_lev, slab = self.n_lev, self.slab
        nx, ny, tgrid = self.nx, self.ny, self.tgrid
        ldx = -(-self.n_src // 16) * 16 if self.padded else self.n_src
        if self.layout == "sb":
            # X (L, S, pitch >= T): every time step carries the same slab, so a cell's T values are one constant
            self.ldt = -(-self.n_t // 16) * 16          # cells start on 128-B lines
            self.x = DeviceArray((n_lev, self.n_src, self.ldt), np.float64)
            for lv in range(n_lev):
                self.x.rows(lv, lv + 1).copy_from_host(np.repeat(slab[lv][:, None], self.ldt, axis=1)[None])
        else:
            self.x = DeviceArray((self.n_t, n_lev, 1, ldx), np.float64)
            first = np.zeros((1, n_lev, 1, ldx))
            first[0, :, 0, :self.n_src] = slab
            self.x.rows(0, 1).copy_from_host(first)
            for t in range(1, self.n_t):
                _lib.call("smm_memcpy_d2d", ctypes.c_void_p(self.x.rows(t, t + 1).ptr),
                          ctypes.c_void_p(self.x.ptr), first.nbytes, None)
        self.desc = (f"{name}: {nx}x{ny} tripolar-like -> {tgrid} conservative, {self.n_t} time steps x "
                     f"{n_lev} masked levels per GPU, f64, remap_area_min 0.5, grouped launch, "
                     + (f"X (L, S, pitch {getattr(self, 'ldt', 0)}) batch-fastest per level" if self.layout == "sb" else
                        "X (T, L, S) " + (f"with rows on 128-B lines (pitch {ldx})" if self.padded else "packed")))

    def spot_check(self, y):
        """One (time step, level) row of the timed output per probed level against oracle/oracle.c,
        the oracle building its own CSR from that level's links."""
        from oracle import oracle
        w3 = self.weights
        ll = w3["link_length"].values
        frac = w3["dst_grid_frac"].values
        t = self.n_t // 2
        yv = y.reshape(self.n_t, self.n_lev, self.n_dst)
        same, cells, levels = True, 0, []
        for lv in sorted({0, self.n_lev // 2, self.n_lev - 1}):
            n = int(ll[lv])
            rm = w3["remap_matrix"].values[lv, :n]
            csr = oracle.coo_to_csr_c(self.n_src, self.n_dst, w3["src_address"].values[lv, :n],
                                      w3["dst_address"].values[lv, :n], rm[:, 0] if rm.ndim == 2 else rm)
            ref = oracle.apply_c(csr, self.slab[lv][None, :], bool(self.masked_levels[lv]), self.dst_imask[lv],
                                 frac[lv], 0.5)[0]
            got = yv.rows(t, t + 1).to_host().reshape(self.n_lev, self.n_dst)[lv]
            ok = ~np.isnan(ref)
            same = same and np.array_equal(np.isnan(got), ~ok) and np.array_equal(got[ok], ref[ok])
            cells += self.n_dst
            levels.append(int(lv))
        return {"time_step": int(t), "levels": levels, "cells": cells, "bit_equal_to_oracle": bool(same)}

    def free(self):
        if self.x is not None:
            self.x.free()
        self.group.close()
        for op in self.ops:
            op.close()

    def cells(self):
        return float(self.n_dst) * self.n_t * self.n_lev

    def alg_bytes(self):
        return sum(algorithmic_bytes(op, self.n_t, 8, 8) for op in self.ops)

    def full_stream_bytes(self):
        return self.n_t * self.n_lev * (self.n_src + self.n_dst) * 8 + sum(op.nnz for op in self.ops) * 12

    def line_bytes(self):
        """What the members' own tile plans stage per time step (every block's distinct lines, halo lines shared by
        neighbouring blocks counted per block); None when a level has no plan of its own (the group's shared block
        shape then differs from it anyway)."""
        if self.layout == "sb":
            return None
        staged = [op.plan_info()["staged_src_elems"] for op in self.ops]
        if not all(op.plan_info()["tile_plan"] for op in self.ops):
            return None
        return self.n_t * (sum(staged) + self.n_lev * self.n_dst) * 8 + sum(op.nnz for op in self.ops) * 12

    def distinct_line_bytes(self):
        """Layout floor per launch: per level the UNION of distinct 128-B lines of a field row that carry a used cell
        (rows on 128-B lines: every row starts a line; packed: S * 8 B = 80 mod 128, the rows cycle through 8 start
        offsets), + Y + the operators once.  PMC traffic above it is halo re-fetch between blocks, below it cache hits."""
        if self.layout == "sb":
            return None
        row = self.n_src * 8
        phases = (0,) if self.padded else sorted({(r * row) % 128 for r in range(32)})
        lines = sum(distinct_lines(op.used_sources(), 8, phases) for op in self.ops)
        return self.n_t * (lines * 128 + self.n_lev * self.n_dst * 8) + sum(op.nnz for op in self.ops) * 12

    def run(self, y, flags):
        if self.layout == "sb":
            self.group.apply_sb(self.x, self.level_index, self.masked_levels, y=y.reshape(self.n_t, self.n_lev, self.n_dst),
                                masked=True, remap_area_min=0.5, transpose=True, flags=flags, n_batch=self.n_t)
            return
        self.group.apply(self.x, self.level_index, self.masked_levels, y=y, masked=True,
                         remap_area_min=0.5, transpose=True, flags=flags)

    def run_rows(self, y, flags, r0, r1):
        if self.layout == "sb":
            raise SystemExit("the tiled gather needs time-major rows: use cfg3 / cfg3c with --gpus N")
        self.group.apply(self.x.rows(r0, r1), self.level_index, self.masked_levels, y=y.rows(r0, r1),
                         masked=True, remap_area_min=0.5, transpose=True, flags=flags)

    def cpu_baseline(self, budget_s):
        """Same legs as Problem2D.cpu_baseline, level by level as regrid.py:387-418 loops."""
        from oracle import oracle
        threads, avail = cpu_threads()
        csrs = [op.export_csr() for op in self.ops]
        frac = self.weights["dst_grid_frac"].values

        def one_pass(t_rows, fn, **kw):
            x = np.broadcast_to(self.slab[None], (t_rows,) + self.slab.shape)
            for lv in range(self.n_lev):
                fn(csrs[lv], np.ascontiguousarray(x[:, lv]), bool(self.masked_levels[lv]),
                   self.dst_imask[lv], frac[lv], 0.5, **kw)

        def timed(t_rows, budget, max_passes, fn, **kw):
            passes, spent = 0, 0.0
            while spent < budget and passes < max_passes:
                t0 = time.perf_counter()
                one_pass(t_rows, fn, **kw)
                spent += time.perf_counter() - t0
                passes += 1
            return passes * t_rows * self.n_lev * self.n_dst / spent, passes, spent

        legs = []
        v, p, t = timed(2, budget_s * 0.25, 10, oracle.apply)
        legs.append({"value": v, "unit": "cells/s", "cores": 1, "kind": "port",
                     "impl": "oracle/oracle.py (numpy + scipy.sparse CSR product), level by level",
                     "rows_sampled": 2 * self.n_lev, "passes": p, "seconds": round(t, 1),
                     "sample": f"2 of {self.n_t} time steps x {self.n_lev} levels x {p} passes, {t:.1f} s"})
        for nt in sorted({threads, avail}):
            t_rows = max(nt, 16)
            v, p, t = timed(t_rows, budget_s * 0.375, 100, oracle.apply_c, threads=nt)
            legs.append({"value": v, "unit": "cells/s", "cores": nt, "kind": "port",
                         "impl": "oracle/oracle.c level by level (OpenMP over rows)",
                         "rows_sampled": t_rows * self.n_lev, "passes": p, "seconds": round(t, 1),
                         "sample": f"{t_rows} of {self.n_t} time steps x {self.n_lev} levels x {p} passes, "
                                   f"{nt} threads of {avail} usable (affinity capped by the cgroup quota; "
                                   f"os.cpu_count() = {os.cpu_count()}), {t:.1f} s"})
        return cpu_summary(legs, avail, f"of {self.n_t * self.n_lev} (time step, level) rows")


def cpu_summary(legs, usable, of_what):
    """`cpu_baseline` of the line: the best leg, its facts in keys of their own, one short sentence."""
    best = max(legs, key=lambda leg: leg["value"])
    one = [leg["value"] for leg in legs if leg["cores"] == 1 and "scipy" in leg["impl"]]
    return {"value": best["value"], "unit": "cells/s", "cores": best["cores"], "kind": "port",
            "sample": f"{best['rows_sampled']} {of_what} x {best['passes']} passes, {best['seconds']} s",
            "impl": best["impl"].split(" (")[0], "threads": best["cores"], "usable_cores": usable,
            "rows_sampled": best["rows_sampled"], "passes": best["passes"], "seconds": best["seconds"],
            "scipy_1core_value": one[0] if one else None, "legs": legs}


class DryProblem:
    """--dry-run: no GPU, no kernel.  A stand-in step (a small numpy product) so that the launcher,
    the rendezvous, the barriers, the max over ranks, the tiled gather schedule (gloo with --comm
    torch) and the JSON line of the N>1 path can be exercised on any host."""

    x_dtype = "f64"

    def __init__(self, name, rank):
        self.n_batch, self.n_dst, self.rank = 64, 96, rank
        self.w = np.random.default_rng(7).standard_normal((32, self.n_dst))
        self.x = np.random.default_rng(11 + rank).standard_normal((self.n_batch, 32))
        self.y_shape = (self.n_batch, self.n_dst)
        self.desc = f"dry run of {name}: stand-in step, no GPU"
        self.meta = {"dry_run": True}

    def cells(self):
        return float(self.n_batch * self.n_dst)

    def alg_bytes(self):
        return self.x.nbytes + self.n_batch * self.n_dst * 8

    def full_stream_bytes(self):
        return self.alg_bytes()

    def line_bytes(self):
        return None

    def run(self, y, flags):
        y[...] = self.x @ self.w

    def run_rows(self, y, flags, r0, r1):
        y[r0:r1] = self.x[r0:r1] @ self.w


class HostEvent:
    """Wall-clock stand-in for a HIP event (--dry-run)."""

    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_ms(self, stop):
        return (stop.t - self.t) * 1e3


def make_comm(args, rank, world, local_rank, rdv):
    """The data-plane communicator of the gather phase (created AFTER the compute measurement, so a
    failure here can never touch `value`)."""
    if args.comm == "native":
        if args.dry_run:
            raise RuntimeError("--dry-run has no RCCL: use --comm torch (gloo) or --gather none")
        from smmregrid_amd.comm import Comm
        return Comm(rank, world, rendezvous=rdv), None
    import torch
    import torch.distributed as dist
    from tools.torch_comm import TorchComm
    if args.dry_run:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    return TorchComm(), dist


def stream_copy_gbs(nbytes=2 << 30, reps=5):
    """Measured device-to-device copy rate of this GPU (read + write bytes per second): the
    practical HBM ceiling to hold the kernel's PMC traffic rate against."""
    import ctypes
    from smmregrid_amd import _lib
    from smmregrid_amd.device import DeviceArray, Event
    a = DeviceArray((nbytes // 8,), np.float64).fill_bytes(1)
    b = DeviceArray((nbytes // 8,), np.float64)
    _lib.call("smm_memcpy_d2d", ctypes.c_void_p(b.ptr), ctypes.c_void_p(a.ptr), nbytes, None)
    e0, e1 = Event(), Event()
    e0.record()
    for _ in range(reps):
        _lib.call("smm_memcpy_d2d", ctypes.c_void_p(b.ptr), ctypes.c_void_p(a.ptr), nbytes, None)
    e1.record()
    e1.synchronize()
    gbs = 2.0 * nbytes * reps / (e0.elapsed_ms(e1) * 1e-3) / 1e9
    a.free()
    b.free()
    return gbs


HOST_ONLY_SOURCES = ("smm_hostpool.cpp", "smm_comm.cpp")   # staging pool, RCCL binding: cannot change a kernel's traffic


def kernel_source_sha():
    """sha256 over the sources that decide what a kernel moves -- the kernels, their launchers and the plan builder; ties a
    replayed PMC figure to the code that produced it."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "smmregrid_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hpp", ".hip", ".cpp", ".h")) and name not in HOST_ONLY_SOURCES:
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def cgroup_cpu_quota():
    """CPUs the cgroup's CFS quota grants this process (None = unlimited / unknown): the tightest
    `cpu.max` (cgroup v2) or cfs_quota / cfs_period (v1) from the process's own cgroup up to the root."""
    best = None

    def take(q, period):
        nonlocal best
        try:
            q, period = float(q), float(period)
        except ValueError:
            return
        if q > 0 and period > 0:
            best = q / period if best is None else min(best, q / period)

    rel, rel_v1 = "", None
    try:
        for line in open("/proc/self/cgroup"):
            parts = line.strip().split(":", 2)
            if len(parts) == 3 and parts[0] == "0":
                rel = parts[2]                                   # cgroup v2
            elif len(parts) == 3 and "cpu" in parts[1].split(","):
                rel_v1 = parts[2]                                # cgroup v1 cpu controller
    except OSError:
        pass

    def up(root, rel_path):
        paths, cur = [root], rel_path.strip("/")
        while cur:
            paths.append(os.path.join(root, cur))
            cur = os.path.dirname(cur)
        return paths

    for base in up("/sys/fs/cgroup", rel):
        try:
            q, period = open(os.path.join(base, "cpu.max")).read().split()[:2]
            if q != "max":
                take(q, period)
        except (OSError, ValueError):
            pass
    for root in ("/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"):
        for base in up(root, rel_v1 or ""):
            try:
                take(open(os.path.join(base, "cpu.cfs_quota_us")).read().strip(),
                     open(os.path.join(base, "cpu.cfs_period_us")).read().strip())
            except OSError:
                pass
    return best


def cpu_threads():
    """(threads of the share leg, threads of the all-cores leg): the cores this process may REALLY
    use = scheduler affinity capped by the cgroup CPU quota (a 16-CPU share of a 256-thread host has
    affinity 256 and a quota of 16: 256 OpenMP threads there only oversubscribe)."""
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = cgroup_cpu_quota()
    usable = affinity if quota is None else max(1, min(affinity, int(math.floor(quota + 0.5))))
    return int(os.environ.get("SMM_CPU_THREADS", min(usable, 16))), usable  # 16 = one GPU's CPU share


def launch_ranks(n_ranks):
    """`--gpus N` without WORLD_SIZE: start N rank processes of this script (fresh interpreters, one
    per GPU; this parent never initialises HIP), relay rank 0's stdout, exit non-zero if any rank
    fails.  Ranks that are still running when one has failed are terminated by their own PIDs."""
    import socket
    # two distinct free ports: a torch store's (MASTER_PORT) and the control plane's TCP rendezvous
    socks = [socket.socket() for _ in range(2)]
    for sk in socks:
        sk.bind(("127.0.0.1", 0))
    ports = [sk.getsockname()[1] for sk in socks]
    for sk in socks:
        sk.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(ports[0]), SMM_RDV_PORT=str(ports[1]),
                   WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks), RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line)
            sys.stdout.flush()

    th = threading.Thread(target=relay, daemon=True)
    th.start()
    rc = 0
    alive = set(range(n_ranks))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in alive:
                    procs[q].terminate()
        time.sleep(0.05)
    th.join(timeout=5.0)
    return rc


def traffic_entry(args, workload, batch):
    """PMC traffic cannot be collected inside this run (rocprofv3 counter passes are separate
    processes): it is replayed from profiles/traffic.json, but only while the kernel sources
    are the ones the PMC pass ran (sha recorded by tools/summarize_pmc.py); else null."""
    if not (args.traffic_json and os.path.exists(args.traffic_json)):
        return None, None
    key = f"{workload}/{batch or 'default'}/{args.kernel}/{args.tune or 0}"
    entry = json.load(open(args.traffic_json)).get(key)
    if not entry:
        return None, None
    now = kernel_source_sha()
    fresh = entry.get("kernel_sha") == now
    return (entry.get("hbm_bytes_per_launch") if fresh else None,
            {"file": os.path.relpath(args.traffic_json, ROOT), "key": key, "summary": entry.get("source"),
             "pmc_kernel_sha": entry.get("kernel_sha"), "pmc_git_head": entry.get("git_head"),
             "current_kernel_sha": now, "fresh": fresh})


def roofline_block(args, prob, k_avg, workload, batch, with_copy_rate=True):
    b_alg = prob.alg_bytes()
    achieved = b_alg / k_avg / 1e9
    traffic, traffic_source = traffic_entry(args, workload, batch)
    line = prob.line_bytes()
    floor = prob.distinct_line_bytes() if hasattr(prob, "distinct_line_bytes") else None
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
           "kernel_ms": k_avg * 1e3, "algorithmic_bytes": b_alg,
           "full_stream_bytes": prob.full_stream_bytes(), "line_granular_bytes": line,
           "line_granular_frac": (line / k_avg / 1e9 / HBM_PEAK_GBS) if line else None,
           # the split of traffic / algorithmic: layout floor (distinct 128-B lines) x re-fetch between blocks
           "distinct_line_bytes": floor,
           "layout_floor_ratio": (floor / b_alg) if floor else None,
           "refetch_ratio": (traffic / floor) if (floor and traffic) else None,
           "traffic_GBs": (traffic / k_avg / 1e9) if traffic else None}
    if with_copy_rate:
        out["stream_copy_GBs"] = stream_copy_gbs()
    return out


OTHERS_DEFAULT = ["cfg2sb", "cfg2sbk", "cfg3", "cfg3c", "cfg3sb", "cfg4s", "cfg5tile"]
# BASELINE.json configs 4 and 5: every rank regrids its share of the fixed total batch (8760 / 8 = 1095 rows
# of f32; 137 x 744 / 8 = 12 741 rows of f64)
BASELINE_CONFIGS = ["cfg4", "cfg5"]
RING_SLOT_BYTES = 16 << 30      # root's receive ring: two slots of at most this size (all ranks' tile)


def short(text, n=96):
    """The driver's record cuts long strings mid-word: every string of the final line stays below ~100 chars."""
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def run_others(args, names, local_rank, flags, t_start):
    """Secondary workloads in the same run (N = 1): W warm-up launches, K launches timed with HIP
    events on the launch stream, roofline accounting as for the headline, and a bit-equality spot
    check of the timed output against the CPU oracle.  `cfg3` and `cfg3c` share their 75 operators
    (only the field's row pitch differs)."""
    from smmregrid_amd.device import DeviceArray, Event, synchronize
    out, shared = {}, {}
    for name in names:
        if time.perf_counter() - t_start > args.others_budget:
            out[name] = {"skipped": f"time budget of {args.others_budget:.0f} s used up"}
            continue
        t0 = time.perf_counter()
        try:
            levels = WORKLOADS[name][0] == "con3d"
            if levels and "levels" in shared:
                prob = shared["levels"]
                prob.set_layout(name, len(WORKLOADS[name]) > 5 and WORKLOADS[name][5] == "pad")
            else:
                prob = (ProblemLevels if levels else Problem2D)(name, local_rank, 0)
                if levels:
                    shared["levels"] = prob
            y = DeviceArray(prob.y_shape, getattr(prob, "y_dt", np.float64),
                            layout="sb" if getattr(prob, "layout", "") == "sbk" else "bs")
            for _ in range(3):
                prob.run(y, flags)
            synchronize()
            ev = [(Event(), Event()) for _ in range(args.others_steps)]
            w0 = time.perf_counter()
            for a, b in ev:
                a.record()
                prob.run(y, flags)
                b.record()
            synchronize()
            wall = time.perf_counter() - w0
            k_avg = float(np.mean([a.elapsed_ms(b) for a, b in ev])) * 1e-3
            entry = {"workload": prob.desc, "steps": args.others_steps, "warmup": 3,
                     "value": prob.cells() * args.others_steps / wall, "unit": "cells/s",
                     "ms_per_step": wall / args.others_steps * 1e3, "dtype": prob.x_dtype,
                     "create_ms": prob.create_ms,
                     "roofline": roofline_block(args, prob, k_avg, name, None, with_copy_rate=False)}
            entry["spot_check"] = prob.spot_check(y)
            if not entry["spot_check"]["bit_equal_to_oracle"]:
                raise SystemExit(f"bench.py: {name}: the timed output differs from the CPU oracle: "
                                 f"{entry['spot_check']}")
            y.free()
            if not levels:
                prob.free()
            entry["setup_and_run_s"] = time.perf_counter() - t0
            out[name] = entry
        except SystemExit:
            raise
        except Exception as exc:   # one secondary workload must not lose the headline
            out[name] = {"error": repr(exc)}
    if "levels" in shared:
        if args.user_path != "none" and time.perf_counter() - t_start <= args.others_budget:
            try:       # the same 75 operators from HOST memory (smm_group_apply_host): never `value`, PCIe-inclusive
                out["cfg3_host"] = levels_host_to_host(shared["levels"], args.host_level_steps)
            except Exception as exc:   # must not lose the line
                out["cfg3_host"] = {"error": short(repr(exc), 90)}
        shared["levels"].free()
    return out


def levels_host_to_host(prob, n_t, reps=3):
    """BASELINE config 3's operators on a field held in HOST memory, (n_t, 75, 1, S) f64 with NaN on land per level, through
    smm_group_apply_host: level-major packing of the used cells (the default) against whole rows, median of `reps` calls, the
    stage split of smm_debug_host_stats, one (time step, level) row checked against the oracle."""
    from oracle import oracle
    from smmregrid_amd import _lib
    from smmregrid_amd.device import result_cache
    x = np.ascontiguousarray(np.broadcast_to(prob.slab[None, :, None, :], (n_t, prob.n_lev, 1, prob.n_src)))
    x[:, :, 0, ::131] += np.arange(n_t, dtype=np.float64)[:, None, None]          # time steps differ (NaN stays NaN)
    used = sum(op.n_used_src for op in prob.ops)
    out = {"workload": f"cfg3 operators, {n_t} x {prob.n_lev} x {prob.n_src} f64 in host memory", "time_steps": int(n_t),
           "input_GB": round(x.nbytes / 1e9, 2), "pcie_GB": {"packed": round((used + prob.n_lev * prob.n_dst) * 8 * n_t / 1e9, 2),
                                                              "whole_rows": round((x.nbytes + n_t * prob.n_lev * prob.n_dst * 8) / 1e9, 2)}}
    ys = {}
    for mode, fl in (("packed", 0), ("whole_rows", _lib.APPLY_HOST_NO_PACK)):
        call = lambda: prob.group.apply_host(x, prob.level_index, prob.masked_levels, masked=True, remap_area_min=0.5, flags=fl)
        ys[mode] = call()                                                           # warm-up: staging buffers
        result_cache.wait()       # ... and the page-locked result block prepared in the background (steady state of a loop)
        _lib.host_stats(reset=True)
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            call()
            times.append(time.perf_counter() - t0)
        st = _lib.host_stats(reset=True)
        sec = float(np.median(times))
        out[mode] = {"seconds": sec, "cells_per_s": n_t * prob.n_lev * prob.n_dst / sec, "host_GBs": x.nbytes / sec / 1e9,
                     "chunks": int(st["chunks"] / max(st["calls"], 1)),
                     "stage_ms": {k[:-3]: round(st[k] / max(st["calls"], 1), 1) for k in
                                  ("stage_in_ms", "h2d_ms", "kernel_ms", "d2h_ms", "copy_out_ms", "wait_ms", "total_ms")}}
    out["same_bits"] = bool(np.array_equal(ys["packed"], ys["whole_rows"], equal_nan=True))
    t, lv = n_t // 2, prob.n_lev // 2
    w3 = prob.weights
    n = int(w3["link_length"].values[lv])
    rm = w3["remap_matrix"].values[lv, :n]
    csr = oracle.coo_to_csr_c(prob.n_src, prob.n_dst, w3["src_address"].values[lv, :n], w3["dst_address"].values[lv, :n],
                              rm[:, 0] if rm.ndim == 2 else rm)
    ref = oracle.apply_c(csr, x[t, lv], bool(prob.masked_levels[lv]), prob.dst_imask[lv], w3["dst_grid_frac"].values[lv], 0.5)[0]
    got = ys["packed"][t, 0, lv]
    out["spot_check"] = bool(np.array_equal(got, ref, equal_nan=True))
    return out


def layout_summary(entry):
    """{ms, frac, traffic_ratio, spot_check} of one timed workload for `roofline.layouts / configs` (the raw traffic
    bytes and creation times stay in the `details` line)."""
    if not entry or "roofline" not in entry:
        return {"error": short((entry or {}).get("error") or (entry or {}).get("skipped") or "not run", 80)}
    r = entry["roofline"]
    out = {"ms": round(r["kernel_ms"], 3), "frac": round(r["frac"], 4)}
    if r.get("traffic"):
        out["traffic_ratio"] = round(r["traffic"] / r["algorithmic_bytes"], 3)
    if r.get("layout_floor_ratio"):       # traffic_ratio = floor (distinct 128-B lines / algorithmic) x refetch (traffic / floor)
        out["floor"] = round(r["layout_floor_ratio"], 3)
        if r.get("refetch_ratio"):
            out["refetch"] = round(r["refetch_ratio"], 3)
            out.pop("traffic_ratio", None)            # = floor x refetch: the compact line carries the two factors
    if "spot_check" in entry:
        out["spot_check"] = bool(entry["spot_check"].get("bit_equal_to_oracle"))
    return out


class Runner:
    """The timing protocol shared by the headline, the gather phases and the BASELINE configs of one run:
    W warm-up steps, then exactly K steps between barriers (device synchronise + host rendezvous), the
    elapsed time as the MAX over ranks, the kernel time from HIP events recorded on the launch stream."""

    def __init__(self, rdv, new_event, synchronize):
        self.rdv, self.new_event, self.synchronize = rdv, new_event, synchronize

    def barrier(self):
        self.synchronize()
        if self.rdv:
            self.rdv.barrier()

    def timed(self, prob, y, flags, steps, warmup, ring=None):
        def step(events=None):
            if ring is not None:
                # tile k's gather (communication stream) overlaps tile k+1's kernel (null stream)
                for k, (r0, r1) in enumerate(ring.tiles):
                    prob.run_rows(y, flags, r0, r1)
                    ring.gather_tile(k)
                ring.finish()
                return
            if events:
                events[0].record()
            prob.run(y, flags)            # launches on the null stream
            if events:
                events[1].record()

        for _ in range(warmup):
            step()
        self.barrier()
        ev = [(self.new_event(), self.new_event()) for _ in range(steps)]
        t0 = time.perf_counter()
        for k in range(steps):
            step(ev[k])
        self.barrier()
        dt = time.perf_counter() - t0
        if self.rdv:
            dt = self.rdv.max(dt)
        return dt, ([] if ring is not None else [a.elapsed_ms(b) for a, b in ev])   # tiles are not timed one by one

    def kernel_ms_over_ranks(self, kernel_ms):
        """(mean kernel ms of this rank, [mean kernel ms of every rank])."""
        import struct
        mine = float(np.mean(kernel_ms)) if kernel_ms else 0.0
        if not self.rdv:
            return mine, [mine]
        return mine, [struct.unpack("<d", p)[0] for p in self.rdv.allgather(struct.pack("<d", mine))]


def _grid_cells(name):
    """Cells of a CDO grid name (r<nx>x<ny>, n<N> regular Gaussian, hp<nside>) without building the grid."""
    import re
    m = re.match(r"r(\d+)x(\d+)$", name)
    if m:
        return int(m.group(1)) * int(m.group(2))
    m = re.match(r"n(\d+)$", name)
    if m:
        return 8 * int(m.group(1)) ** 2
    m = re.match(r"hp(\d+)", name)
    if m:
        return 12 * int(m.group(1)) ** 2
    raise ValueError(f"no size rule for grid {name!r}")


def hbm_need(args, name, rows, world, with_ring):
    """Device bytes one rank needs for its share of a 2-D workload (SURVEY 8e capacity note): X + Y shards, the
    operator (SELL-64 + canonical CSR + tile plan, ~64 B per link with links per row estimated from the method) and --
    on the rank that receives the gather -- the ring of two tile slots holding every rank's tile."""
    method, sgrid, tgrid = WORKLOADS[name][:3]
    S, D = _grid_cells(sgrid), _grid_cells(tgrid)
    sx = 8 if WORKLOADS[name][4] == "f64" else 4
    sy = 8
    per_row = {"bil": 4, "nn": 1}.get(method, max(4, -(-S // D) + 5))
    need = {"x": rows * S * sx, "y": rows * D * sy, "operator": 64 * per_row * D, "ring": 0}
    if with_ring and world > 1:
        shard = rows * D * sy
        tiles = max(1, min(rows, max(args.gather_tiles, -(-world * shard // RING_SLOT_BYTES))))
        need["ring"] = 2 * world * (-(-rows // tiles)) * D * sy
    need["total"] = sum(need.values())
    return need


def gather_tiles_for(args, prob, world):
    """Row tiles of the overlapped gather: at least --gather-tiles, and enough of them that one ring slot on
    the root (every rank's tile) stays within RING_SLOT_BYTES (config 4: 8 x 110 GB of Y shards)."""
    rows = prob.y_shape[0]
    shard_bytes = int(np.prod(prob.y_shape, dtype=np.int64)) * np.dtype(getattr(prob, "y_dt", np.float64)).itemsize
    need = -(-world * shard_bytes // RING_SLOT_BYTES)
    return int(max(1, min(rows, max(args.gather_tiles, need))))


def gather_phase(args, runner, prob, y, flags, comm, world, n_ranks, steps, warmup):
    """The timed job followed by the RCCL gather of the Y shards to rank 0 after every step."""
    from smmregrid_amd.distributed import TiledRingGather
    ring = TiledRingGather(comm, y, root=0, tiles=gather_tiles_for(args, prob, world), slots=2)
    g_elapsed, _ = runner.timed(prob, y, flags, steps, warmup, ring=ring)
    res = {"value": prob.cells() * n_ranks * steps / g_elapsed, "unit": "cells/s", "steps": steps, "warmup": warmup,
           "ms_per_step": g_elapsed / steps * 1e3, "ranks": comm.world,
           "gathered_bytes_per_step": ring.gathered_bytes // max(steps + warmup, 1),
           "tiles": len(ring.tiles), "ring_slots": 2, "overlapped_with_compute": True}
    for slot in ring.ring or []:
        if hasattr(slot, "free"):
            slot.free()
    return res


def compact(obj, digits=6):
    """Nested blocks of the final line: floats to `digits` significant digits (the line stays small)."""
    if isinstance(obj, float):
        return float(f"{obj:.{digits}g}") if math.isfinite(obj) else obj
    if isinstance(obj, dict):
        return {k: compact(v, digits) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [compact(v, digits) for v in obj]
    return obj


ROOFLINE_SCALARS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "algorithmic_bytes",
                    "traffic_ratio", "traffic_fresh", "layout_floor_ratio", "refetch_ratio", "stream_copy_GBs")


def final_line(out, details):
    """The compact last line (driver contract + roofline + cpu_baseline + the per-config blocks): scalars in
    their own keys, strings short, nested blocks small.  Everything bulky went to the `details` line.
    The driver's record keeps the first two dozen scalar keys of a nested block and the last 2 kB of the output:
    `roofline` therefore lists its scalars first (the headline's, then every other workload's fraction) and its
    nested `layouts` / `configs` last, and the line ends with `reference_sized`, `baseline_configs`, `host_to_host`."""
    line = dict(out)
    if isinstance(line.get("value"), float):
        line["value"] = float(round(line["value"]))            # cells/s to the cell: the digits behind the point are noise
    if isinstance(line.get("ms_per_step"), float):
        line["ms_per_step"] = round(line["ms_per_step"], 6)
    if isinstance(line.get("spot_check"), dict):
        line["spot_check"] = {k: v for k, v in line["spot_check"].items() if k != "batch_row"}
    others = details.get("others") or {}
    full = line.get("roofline") or {}
    if full.get("traffic") and full.get("algorithmic_bytes"):
        full["traffic_ratio"] = round(full["traffic"] / full["algorithmic_bytes"], 3)
    src = full.get("traffic_source")
    if src:
        full["traffic_fresh"] = bool(src.get("fresh"))
    roof = {k: full[k] for k in ROOFLINE_SCALARS if k in full}
    for k in ("dry_run",):
        if k in full:
            roof[k] = full[k]
    nested = {}
    if others:
        native = {"ms": round(full["kernel_ms"], 3), "frac": round(full["frac"], 4)}
        if full.get("traffic_ratio"):
            native["traffic_ratio"] = full["traffic_ratio"]
        if "spot_check" in line:
            native["spot_check"] = bool(line["spot_check"].get("bit_equal_to_oracle"))
        sb = layout_summary(others.get("cfg2sb"))
        nested["layouts"] = {"native": native, "batch_fastest": sb}
        if "cfg2sbk" in others:
            nested["layouts"]["batch_fastest_y_kept"] = layout_summary(others.get("cfg2sbk"))
        # the same figures once more as plain scalars (a record that keeps only scalars still shows them); the traffic ratios
        # of the two layouts live in the scalars only (`traffic_ratio`, `batch_fastest_traffic_ratio`)
        for key in ("frac", "ms", "traffic_ratio"):
            if key in sb:
                roof[f"batch_fastest_{key}"] = sb[key]
        for e in nested["layouts"].values():
            e.pop("traffic_ratio", None)
        # (the per-workload fractions live in `roofline.configs` and `baseline_configs` only: round 5 repeated them as
        # scalars here and the line came within 0.4 kB of its budget)
        nested["configs"] = {n: layout_summary(e) for n, e in others.items() if not n.startswith("cfg2sb") and n != "cfg3_host"}
    for k in ("kernel_ms_min_rank", "kernel_ms_max_rank"):
        if k in full and line.get("n_gpus", 1) > 1:
            roof[k] = full[k]
    checks = [e.pop("spot_check") for blk in nested.values() for e in blk.values() if isinstance(e, dict) and "spot_check" in e]
    if checks:
        roof["spot_checks_bit_equal"] = f"{sum(bool(c) for c in checks)} of {len(checks)}"   # every timed output against the oracle
    roof.update(nested)
    if "roofline" in line:
        line["roofline"] = roof
    cpu = line.get("cpu_baseline")
    if cpu:
        for k in ("legs", "threads", "rows_sampled", "passes", "seconds", "usable_cores", "impl"):   # `sample` / `cores` / `kind` say it; the legs are in `details`
            cpu.pop(k, None)
        cpu["sample"] = short(cpu["sample"], 60)
    cfg = line.get("config") or {}
    cfg.pop("plan", None)
    for k in ("gather", "comm"):
        if cfg.get(k) == "n/a":
            cfg.pop(k)
    for k, v in list(cfg.items()):
        if isinstance(v, str):
            cfg[k] = short(v)
    # the tail of the line: one GPU's share of BASELINE configs 4 / 5, then the user-visible paths
    line.pop("baseline_configs", None)
    for name, blk in (details.get("baseline_configs") or {}).items():
        keep = {k: v for k, v in blk.items() if k not in ("workload", "steps", "warmup", "unit", "dtype", "algorithmic_bytes",
                                                          "setup_and_run_s", "create_ms")}
        if keep.get("n_gpus") == 1:          # one rank: the spread over ranks and the host-side step time say nothing new
            for k in ("kernel_ms_min", "kernel_ms_max", "ms_per_step"):
                keep.pop(k, None)
        if "with_gather" in keep and "value" in keep["with_gather"]:
            g = keep["with_gather"]
            keep["with_gather"] = {k: g[k] for k in ("value", "ms_per_step", "ranks", "gathered_bytes_per_step", "tiles", "steps")}
        if "f32_out" in keep and "frac" in keep["f32_out"]:      # config 4 with the opt-in f32 store (SURVEY 8d's f32 / f32 bytes)
            f = keep["f32_out"]
            keep["f32_out"] = {"ms": round(f["kernel_ms"], 3), "frac": round(f["frac"], 4), "alg_GB": round(f["algorithmic_bytes"] / 1e9, 2),
                               "spot_check": f["spot_check"]}
        if "algorithmic_bytes" in blk:
            keep["alg_GB"] = round(blk["algorithmic_bytes"] / 1e9, 2)
        line.setdefault("baseline_configs", {})[name] = keep
    ref = details.get("reference_sized")
    if ref:
        keys = ("in", "init_ms", "regrid_ms", "cpu_scipy_ms", "cpu_c1_ms", "bit_equal")
        line["reference_sized"] = {"cols": "shape; ms: init, regrid() host->host, 1-core scipy, 1-core C; bit_equal"}
        for n, e in ref.items():
            line["reference_sized"][n] = ([e.get(k) for k in keys] if "regrid_ms" in e else
                                          {"error": short(e.get("error") or e.get("skipped"), 60)})
    h2h = details.get("host_to_host")
    if h2h:
        # per mode: median and best cells/s of `reps` calls, the fraction of the measured PCIe (H2D) and host-memory
        # (2 x the staging pool's copy rate) ceilings it runs at, and the stage split of one call in ms
        blk = {"rows": h2h.get("rows"), "reps": h2h.get("reps"),
               "cols": "Mcells/s med, best; frac PCIe, host-mem; ms: in,h2d,kern,d2h,out,wait,total"}
        ceil = h2h.get("ceilings") or {}
        blk["ceil_GBs"] = {k.replace("pcie_", "").replace("_GBs", ""): v for k, v in ceil.items()}
        for k, e in h2h.items():
            if isinstance(e, dict) and "cells_per_s" in e:
                st = e.get("stage_ms") or {}
                blk[k] = [round(e["cells_per_s"] / 1e6), round(e.get("cells_per_s_best", 0) / 1e6),
                          round(e.get("pcie_frac", 0), 2), round(e.get("host_mem_frac", 0), 2)] + \
                         [round(st.get(n, 0), 1) for n in ("stage_in", "h2d", "kernel", "d2h", "copy_out", "wait", "total")]
        c3 = others.get("cfg3_host") or {}
        if "packed" in c3:      # config 3's 75 operators from host memory: Mcells/s and seconds, packed (level-major) / whole rows
            blk["cfg3_levels"] = {"steps": c3["time_steps"], "in_GB": c3["input_GB"],
                                  "packed": [round(c3["packed"]["cells_per_s"] / 1e6), round(c3["packed"]["seconds"], 3)],
                                  "whole_rows": [round(c3["whole_rows"]["cells_per_s"] / 1e6), round(c3["whole_rows"]["seconds"], 3)],
                                  "ok": bool(c3.get("same_bits") and c3.get("spot_check"))}
        elif "error" in c3:
            blk["cfg3_levels"] = {"error": c3["error"]}
        if h2h.get("cpu_cells_per_s"):
            blk["cpu_Mcells_per_s"] = round(h2h["cpu_cells_per_s"] / 1e6)
        for k in ("staging_threads", "spot_check", "bound", "error"):
            if k in h2h:
                blk[k] = h2h[k]
        line["host_to_host"] = blk
    # the record keeps the last 2 kB of the output whatever happens: the user-visible paths and the BASELINE configs go last
    for key in ("reference_sized", "baseline_configs", "host_to_host"):
        if key in line:
            line[key] = line.pop(key)
    for key in ("config", "roofline", "cpu_baseline", "with_gather", "baseline_configs", "spot_check", "host_to_host",
                "reference_sized"):
        if key in line:
            line[key] = compact(line[key], 5 if key in ("baseline_configs", "host_to_host") else 6)
    return line


def baseline_config_block(args, name, runner, local_rank, rank, world, n_ranks, flags, comm):
    """One of BASELINE configs 4 / 5 on this run's ranks: every rank regrids its share of the fixed total
    batch (compute only, then with the tiled gather to rank 0).  Returns rank 0's block (None elsewhere)."""
    from smmregrid_amd.device import DeviceArray
    steps, warmup = args.config_steps, args.config_warmup
    t0 = time.perf_counter()
    # Does this rank's share fit its GPU?  Decided by all ranks together BEFORE anything is allocated: a config that
    # does not fit is reported as skipped, it does not end the run in hipMalloc (rank 0 also holds the gather's ring).
    rows = args.config_batch or WORKLOADS[name][3]
    need = hbm_need(args, name, rows, world, with_ring=(comm is not None and args.gather == "root" and rank == 0))
    free = None
    if os.environ.get("SMM_BENCH_TEST_FREE_GB"):          # tests/test_bench_launch.py: a GPU with that much free memory
        free = int(float(os.environ["SMM_BENCH_TEST_FREE_GB"]) * 1e9)
        rows = WORKLOADS[name][3]
        need = hbm_need(args, name, rows, world, with_ring=(args.gather == "root" and rank == 0))
    elif not args.dry_run:
        from smmregrid_amd.device import mem_info
        free = mem_info()[0]
    fits = free is None or need["total"] * 1.03 + (1 << 30) <= free
    verdicts = [fits]
    if runner.rdv:
        verdicts = [v == b"1" for v in runner.rdv.allgather(b"1" if fits else b"0")]
    if not all(verdicts):
        if rank != 0:
            return None
        return {"skipped": short(f"needs {need['total'] / 1e9:.0f} GB per rank, {(free or 0) / 1e9:.0f} GB free on rank 0; "
                                 f"ranks that fit: {sum(verdicts)} of {len(verdicts)}"),
                "hbm_needed_gb": round(need["total"] / 1e9, 1)}
    if args.dry_run:
        prob = DryProblem(name, rank)
        y = np.zeros(prob.y_shape)
    else:
        prob = Problem2D(name, local_rank, rank, batch=args.config_batch or None)
        y = DeviceArray(prob.y_shape, prob.y_dt)
    elapsed, kernel_ms = runner.timed(prob, y, flags, steps, warmup)
    mine, per_rank = runner.kernel_ms_over_ranks(kernel_ms)
    blk = None
    if rank == 0:
        k_avg = mine * 1e-3
        blk = {"workload": short(prob.desc), "rows_per_gpu": int(getattr(prob, "n_batch", 0)), "dtype": prob.x_dtype,
               "steps": steps, "warmup": warmup, "n_gpus": n_ranks,
               "value": prob.cells() * n_ranks * steps / elapsed, "unit": "cells/s",
               "ms_per_step": elapsed / steps * 1e3, "kernel_ms": mine,
               "kernel_ms_min": min(per_rank), "kernel_ms_max": max(per_rank),
               "create_ms": getattr(prob, "create_ms", None),
               # what the full share (not the dry run's stand-in) asks of one GPU: X + Y + operator (+ ring on rank 0)
               "hbm_needed_gb": round(hbm_need(args, name, WORKLOADS[name][3], world,
                                               with_ring=args.gather == "root")["total"] / 1e9, 1)}
        if not args.dry_run:
            alg = prob.alg_bytes()
            blk.update(algorithmic_bytes=alg, frac=alg / k_avg / 1e9 / HBM_PEAK_GBS)
            chk = prob.spot_check(y)
            blk["spot_check"] = bool(chk["bit_equal_to_oracle"])
            if not blk["spot_check"]:
                raise SystemExit(f"bench.py: {name}: the timed output differs from the CPU oracle: {chk}")
    if name == "cfg4" and world == 1 and not args.dry_run:
        # SURVEY 8d quotes config 4's algorithmic bytes f32 in / f32 out; the default above is the reference's dtype rule
        # (result_type(x, f64) = f64, regrid.py:550).  The opt-in f32 store (y_dtype = SMM_F32: the rounded f64 result)
        # on the same operator and field, timed the same way, with its own algorithmic bytes and spot check.
        try:
            y.free()
            prob.y_dt = np.float32
            y = DeviceArray(prob.y_shape, np.float32)
            _, k32 = runner.timed(prob, y, flags, steps, warmup)
            k32_avg = float(np.mean(k32)) * 1e-3
            alg32 = prob.alg_bytes()
            chk32 = prob.spot_check(y)
            blk["f32_out"] = {"kernel_ms": k32_avg * 1e3, "algorithmic_bytes": alg32,
                              "frac": alg32 / k32_avg / 1e9 / HBM_PEAK_GBS, "spot_check": bool(chk32["bit_equal_to_oracle"]),
                              "value": prob.cells() / k32_avg}
            if not blk["f32_out"]["spot_check"]:
                raise SystemExit(f"bench.py: {name} f32 out: the timed output differs from the rounded oracle: {chk32}")
        except SystemExit:
            raise
        except Exception as exc:
            blk["f32_out"] = {"error": short(repr(exc), 80)}
    if comm is not None and args.gather == "root":
        try:
            g = gather_phase(args, runner, prob, y, flags, comm, world, n_ranks, args.config_gather_steps, 1)
            if rank == 0:
                blk["with_gather"] = g
        except Exception as exc:
            if rank == 0:
                blk["with_gather"] = {"error": short(repr(exc))}
    if hasattr(y, "free"):
        y.free()
    if hasattr(prob, "free"):
        prob.free()
    if rank == 0:
        blk["setup_and_run_s"] = round(time.perf_counter() - t0, 2)
    return blk


def main():
    t_start = time.perf_counter()
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))      # this parent process never touches a GPU
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU, the two must agree")
    use_dist = world > 1 or bool(os.environ.get("SMM_BENCH_FORCE_DIST"))  # rehearsal of the N>1 path on one GPU
    if args.dry_run and os.environ.get("SMM_BENCH_TEST_FAIL_RANK") == str(rank):
        raise SystemExit(7)                    # tests/test_bench_launch.py: a rank that dies early

    # control plane: host-side TCP rendezvous (barrier, max over ranks, RCCL id hand-over); no GPU involved
    rdv = None
    n_ranks = 1
    if use_dist:
        from smmregrid_amd.comm import HostRendezvous
        rdv = HostRendezvous(rank, world)
        n_ranks = len({int(p) for p in rdv.allgather(str(rank).encode())})   # ranks that really joined

    rehearsal = None
    if args.dry_run:
        prob = DryProblem(args.workload, rank)
        y = np.zeros(prob.y_shape)
        flags = 0
        new_event, synchronize, dev_name = HostEvent, (lambda: None), "none (dry run)"
    else:
        from smmregrid_amd import _lib
        from smmregrid_amd.device import DeviceArray, Event, device_name, set_device, synchronize
        n_dev = _lib.device_count()
        if n_dev <= 0:
            raise SystemExit("bench.py: no HIP device (there is no CPU fallback; --dry-run exercises the plumbing)")
        if local_rank >= n_dev:
            if not os.environ.get("SMM_BENCH_SHARE_GPUS"):
                raise SystemExit(f"bench.py: rank {rank} has no GPU of its own ({n_dev} visible): one rank per GPU "
                                 "(SMM_BENCH_SHARE_GPUS=1 lets ranks share devices for a rehearsal)")
            local_rank %= n_dev              # rehearsal on a smaller box: ranks share devices, RCCL will refuse
        if world > n_dev and os.environ.get("SMM_BENCH_SHARE_GPUS"):
            rehearsal = f"{world} ranks share {n_dev} device(s)"
            if os.environ.get("SMM_RCCL_LIB"):
                rehearsal += f"; collectives through {os.path.basename(os.environ['SMM_RCCL_LIB'])}"
        set_device(local_rank)
        cls = ProblemLevels if WORKLOADS[args.workload][0] == "con3d" else Problem2D
        prob = cls(args.workload, local_rank, rank, batch=args.batch)
        y = DeviceArray(prob.y_shape, getattr(prob, "y_dt", np.float64))
        flags = {"auto": 0, "sell": _lib.APPLY_KERNEL_SELL, "tile": _lib.APPLY_KERNEL_TILE}[args.kernel]
        for kv in [kv for kv in args.tune.split(",") if kv]:      # process-wide, for the whole run
            _lib.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
        new_event, dev_name = Event, device_name(local_rank)
    runner = Runner(rdv, new_event, synchronize)

    # the measured job: every rank regrids its shard, Y shards stay resident on their GPUs
    elapsed, kernel_ms = runner.timed(prob, y, flags, args.steps, args.warmup)
    k_mine, k_ranks = runner.kernel_ms_over_ranks(kernel_ms)

    out, details = None, {}
    if rank == 0:
        k_avg = k_mine * 1e-3
        cfg = {"workload": prob.desc, "kernel": args.kernel, "gather": args.gather if use_dist else "n/a",
               "comm": args.comm if use_dist else "n/a", "device": dev_name,
               "launched_by": "bench.py" if os.environ.get("LOCAL_WORLD_SIZE") and "TORCHELASTIC_RUN_ID" not in os.environ
               and world > 1 else ("external launcher" if world > 1 else "single process"),
               "rows_per_gpu": int(getattr(prob, "n_batch", getattr(prob, "n_t", 0))),
               "layout": "native (B, S)" if getattr(prob, "layout", "bs") == "bs" else "batch-fastest",
               "resident_in_hbm": True, "create_ms": getattr(prob, "create_ms", None)}
        cfg.update(prob.meta)
        if rehearsal:
            # not a multi-GPU measurement: the ranks ran on fewer devices than ranks (tests / a rehearsal of the N > 1 path)
            cfg["rehearsal"] = short(rehearsal + ": no scaling or xGMI meaning", 96)
        details["plan"] = cfg.get("plan")
        out = {
            "metric": "regridded cells/sec (dst_pts x time x lev)",
            "value": prob.cells() * n_ranks * args.steps / elapsed,
            "unit": "cells/s",
            "n_gpus": n_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": prob.x_dtype,
            "data": "synthetic",
            "config": cfg,
        }
        if args.dry_run:
            out["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                               "traffic": None, "kernel_ms": k_avg * 1e3, "dry_run": True}
        else:
            out["roofline"] = roofline_block(args, prob, k_avg, args.workload, args.batch)
            if hasattr(prob, "spot_check"):
                out["spot_check"] = prob.spot_check(y)
                if not out["spot_check"]["bit_equal_to_oracle"]:
                    if os.environ.get("SMM_BENCH_ABLATION") != "1":
                        raise SystemExit(f"bench.py: the timed output differs from the CPU oracle: {out['spot_check']}")
                    # a timing-only ablation build (tools/exp/build_exp.sh: -DSMM_EXP_*) computes wrong results by
                    # design; the explicit opt-in keeps the line but marks it as no measurement of the product
                    out["spot_check"]["experiment_library"] = os.environ.get("SMM_LIB_PATH", "in-tree")
                    out["invalid"] = "ablation build: results differ from the oracle (SMM_BENCH_ABLATION=1)"
        out["roofline"]["kernel_ms_min_rank"] = min(k_ranks)
        out["roofline"]["kernel_ms_max_rank"] = max(k_ranks)

    def emit(code=0):
        """Rank 0: the bulky details first (a line that does not start with '{'), the compact line last."""
        if rank == 0:
            out["wall_s"] = round(time.perf_counter() - t_start, 1)
            print("details: " + json.dumps(details), flush=True)
            print(json.dumps(final_line(out, details)), flush=True)
        if code:
            os._exit(code)

    # Everything that involves RCCL runs under a watchdog: the compute measurement above stands whatever
    # happens below; a phase that does not finish in time costs its own figures and a non-zero exit status.
    gather_wanted = use_dist and args.gather == "root"
    done = threading.Event()

    def give_up():
        """Timer thread.  The main thread may still be filling `out` / `details` (a slow gather rather than a hung
        one): the line is built from deep copies, any failure on the way falls back to the minimal contract line,
        and the exit happens whatever the printing did (an exception here used to leave the rank
        hanging with the launcher waiting for it)."""
        if done.is_set():
            return
        try:
            if rank == 0:
                minimal = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                                   "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
                minimal["with_gather"] = {"error": f"gather phases exceeded {args.gather_timeout:.0f} s"}
                try:
                    import copy
                    snap_out, snap_details = None, None
                    for _ in range(3):          # a dict changing size under the copy: try again
                        try:
                            snap_out, snap_details = copy.deepcopy(out), copy.deepcopy(details)
                            break
                        except RuntimeError:
                            time.sleep(0.05)
                    snap_out.setdefault("with_gather", minimal["with_gather"])
                    snap_out["wall_s"] = time.perf_counter() - t_start
                    text = "details: " + json.dumps(snap_details) + "\n" + json.dumps(final_line(snap_out, snap_details))
                except Exception:
                    text = json.dumps(minimal)
                print(text, flush=True)
        finally:
            os._exit(3)     # start nothing new: the line (with the compute value) is out, the status says what happened

    dog = None
    if gather_wanted:
        dog = threading.Timer(args.gather_timeout, give_up)
        dog.daemon = True
        dog.start()
        if args.dry_run and os.environ.get("SMM_BENCH_TEST_HANG_GATHER") == str(rank):
            time.sleep(args.gather_timeout + 30)      # tests/test_bench_launch.py: a gather that never returns

    comm = dist_mod = None
    if gather_wanted:
        # the same job followed by the RCCL gather of the Y shards to rank 0 (north star's exchange step),
        # reported beside it: xGMI-link bound, see DESIGN.md "Multi-GPU".  The communicator is created only now.
        try:
            setup_error = b""
            try:
                comm, dist_mod = make_comm(args, rank, world, local_rank, rdv)
            except Exception as exc:
                setup_error = repr(exc).encode()
            # all ranks agree before anyone enters the collective loop: one rank without a communicator
            # would leave the others waiting at the first barrier of the gather phase
            failures = [(r, e.decode()) for r, e in enumerate(rdv.allgather(setup_error)) if e]
            if failures:
                comm = None
                raise RuntimeError("communicator set-up failed on rank(s) "
                                   + "; ".join(f"{r}: {e}" for r, e in failures[:3]))
            g = gather_phase(args, runner, prob, y, flags, comm, world, n_ranks, args.steps, args.warmup)
            if rank == 0:
                out["with_gather"] = g
        except Exception as exc:  # report, never lose the compute measurement
            if rank == 0:
                out["with_gather"] = {"error": short(repr(exc), 200)}

    if rank == 0 and world == 1 and not args.dry_run:
        if not args.no_cpu_baseline:                 # CPU baseline: rank 0 at N = 1 only
            cpu = prob.cpu_baseline(args.cpu_seconds)
            details["cpu_baseline_legs"] = cpu.get("legs")
            out["cpu_baseline"] = cpu
    if hasattr(y, "free"):
        y.free()
    if hasattr(prob, "free"):
        prob.free()
    if rank == 0 and world == 1 and not args.dry_run and args.user_path != "none" and args.workload == "cfg2" \
            and args.batch is None:
        # the path a user of the drop-in sees: host buffers in, host buffers out (PCIe-inclusive; never `value`)
        from tools.user_path_bench import host_to_host, reference_sized
        for key, fn in (("reference_sized", lambda: reference_sized(local_rank)),
                        ("host_to_host", lambda: host_to_host(local_rank, rows=args.host_rows,
                                                              cpu_cells_per_s=(out.get("cpu_baseline") or {}).get("value")))):
            t0 = time.perf_counter()
            try:
                details[key] = fn()
            except Exception as exc:       # must not lose the headline
                details[key] = {"error": short(repr(exc), 90)}
            details.setdefault("user_path_seconds", {})[key] = round(time.perf_counter() - t0, 1)
    if rank == 0 and world == 1 and not args.dry_run:
        names = OTHERS_DEFAULT if args.others == "default" else [n for n in args.others.split(",") if n and n != "none"]
        if names and args.workload == "cfg2" and args.batch is None:
            unknown = [n for n in names if n not in WORKLOADS]
            if unknown:
                raise SystemExit(f"bench.py: unknown workload(s) in --others: {unknown}")
            details["others"] = run_others(args, names, local_rank, flags, t_start)

    # BASELINE configs 4 and 5 on the same ranks (each rank its share of the fixed total batch)
    names = [] if args.configs == "none" else (BASELINE_CONFIGS if args.configs == "default" else
                                               [n for n in args.configs.split(",") if n])
    if names and args.workload == "cfg2" and args.batch is None:
        unknown = [n for n in names if n not in WORKLOADS or WORKLOADS[n][0] == "con3d"]
        if unknown:
            raise SystemExit(f"bench.py: --configs takes 2-D workloads, not {unknown}")
        for name in names:
            late = time.perf_counter() - t_start > args.others_budget     # rank 0's clock decides for all
            if rdv:
                late = rdv.bcast(b"1" if late else b"0") == b"1"
            if late:
                if rank == 0:
                    details.setdefault("baseline_configs", {})[name] = {"skipped": "time budget used up"}
                continue
            blk = baseline_config_block(args, name, runner, local_rank, rank, world, n_ranks, flags, comm)
            if rank == 0:
                details.setdefault("baseline_configs", {})[name] = blk

    done.set()
    if dog is not None:
        dog.cancel()
    try:
        if dist_mod is not None:
            dist_mod.destroy_process_group()
        elif comm is not None:
            comm.close()
    except Exception:
        pass

    if rank == 0:
        emit()

    if rdv:
        try:
            rdv.barrier()
        except Exception:
            pass
        rdv.close()


if __name__ == "__main__":
    main()
