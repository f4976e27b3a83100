#!/usr/bin/env python3
"""Soak: many seeded random operators / fields through every kernel, compared bit for bit with the oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_fuzz import make_links
from tests.helpers import field
from oracle import oracle
from smmregrid_amd import SparseOperator, _lib, to_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = time.time(); bad = 0
for seed in range(n):
    rng = np.random.default_rng(900000 + seed)
    n_src = int(rng.integers(1, 20000)); n_dst = int(rng.integers(1, 5000))
    kind = ["random", "ragged", "banded", "longband"][seed % 4]
    if kind == "longband":
        n_src, n_dst = int(rng.integers(5000, 200000)), int(rng.integers(1, 900))
    src, dst, w = make_links(rng, kind, n_src, n_dst)
    if seed % 3 == 1 and src.size:                 # links in CDO's (dst, src) order: the builder's sort-free path
        o = np.lexsort((src, dst)); src, dst, w = src[o], dst[o], w[o]
    # round 4: the builder on 1 .. 6 host threads (0 = automatic), the same operator whatever the count
    _lib.call("smm_set_host_threads", int(rng.integers(0, 7)), None)
    op = SparseOperator(n_src, n_dst, src, dst, w, device=0, prune_zeros=(seed % 5 == 2))
    _lib.call("smm_set_host_threads", 0, None)
    if not (seed % 5 == 2):
        rp, cc, vv = oracle.coo_to_csr_c(n_src, n_dst, src, dst, w)
        got = op.export_csr()
        if not (np.array_equal(got[0], rp) and np.array_equal(got[1], cc) and np.array_equal(got[2].view(np.uint64), vv.view(np.uint64))):
            bad += 1; print("CSR MISMATCH seed", seed, flush=True)
    csr = op.export_csr()
    imask = (rng.random(n_dst) > 0.3).astype(np.int32); frac = rng.random(n_dst)
    op.set_epilogue(imask, frac)
    dtype = np.float32 if (seed // 4) % 2 else np.float64
    x = field(rng, int(rng.integers(1, 70)), n_src, dtype=dtype, nan_frac=0.03, inf_frac=0.003)
    amin = float(rng.choice([0.0, 0.5])); masked = bool((seed // 8) % 2)
    ref = oracle.apply_c(csr, x, masked, imask, frac, amin)
    ks = [(0, {}), (_lib.APPLY_KERNEL_SELL, {})]
    if op.plan_info()["tile_plan"]:
        # default tile kernel, other block orders, single-row steps, odd walk lengths (tails of multi-row steps),
        # LDS-DMA staging forced wherever the field is 16-B aligned, register staging forced
        t = _lib.APPLY_KERNEL_TILE
        ks += [(t, k) for k in ({}, dict(xcd_run=-1), dict(xcd_run=8), dict(xcd_run=128),
                                dict(tile_staging=1, tile_rows_per_step=1), dict(tile_split_rows=1),
                                dict(tile_staging=2, tile_rows_per_step=1), dict(tile_staging=1),
                                dict(tile_staging=2, tile_rows_per_step=1, tile_walk=3))]
        ks += [(t, dict(tile_walk=j)) for j in (1, 3, 5, 7)]
    dx = to_device(x)
    for fl, knobs in ks:
      with _lib.tuning(**knobs):
        for rep in range(3):
            # round 4: every third repetition with a launch-grid limit a few parts below the grid the batch needs
            if rep == 2:
                nb = op.launch_info(x.shape[0], dtype, flags=fl)["n_blocks"]
                per_row = op.launch_info(1, dtype, flags=fl)["n_blocks"]
                _lib.call("smm_debug_set_grid_limit", int(max(per_row, nb // int(rng.integers(2, 6)))))
            y = op.apply(dx, masked=masked, remap_area_min=amin, flags=fl).to_host()
            same = np.array_equal(np.isnan(y), np.isnan(ref)) and np.array_equal(y[~np.isnan(y)], ref[~np.isnan(ref)])
            _lib.call("smm_debug_set_grid_limit", 0)
            if not same:
                bad += 1; print("MISMATCH seed", seed, "flags", fl, knobs, "rep", rep, flush=True)
    # batch-fastest entry point (full and packed X) and the host pipeline (packing when the operator qualifies)
    xt = np.ascontiguousarray(x.T)
    for packed in (False, True):
        xx = np.ascontiguousarray(xt[op.used_sources()]) if packed else xt
        y = op.apply_sb(to_device(xx), masked=masked, remap_area_min=amin, packed=packed).to_host()
        if not (np.array_equal(np.isnan(y), np.isnan(ref)) and np.array_equal(y[~np.isnan(y)], ref[~np.isnan(ref)])):
            bad += 1; print("MISMATCH seed", seed, "apply_sb packed", packed, flush=True)
        yk = op.apply_sb(to_device(xx), masked=masked, remap_area_min=amin, packed=packed, keep_batch_fastest=True).to_host().T
        if not (np.array_equal(np.isnan(yk), np.isnan(ref)) and np.array_equal(yk[~np.isnan(yk)], ref[~np.isnan(ref)])):
            bad += 1; print("MISMATCH seed", seed, "apply_sb kept batch-fastest, packed", packed, flush=True)
    y = op.apply_host(x, masked=masked, remap_area_min=amin, chunk_rows=int(rng.integers(0, 40)))
    if not (np.array_equal(np.isnan(y), np.isnan(ref)) and np.array_equal(y[~np.isnan(y)], ref[~np.isnan(ref)])):
        bad += 1; print("MISMATCH seed", seed, "apply_host", flush=True)
    op.close()
    if seed % 50 == 49: print(f"{seed+1} cases, {bad} mismatches, {time.time()-t0:.0f}s", flush=True)
# level groups (grouped launch, both Y layouts, host pipeline): the fuzz test body on many more seeds
from tests.test_gpu_fuzz import test_fuzz_levels
n_groups = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, n // 4)      # soak.py <operators> [<level groups>]
for seed in range(100, 100 + n_groups):
    try:
        test_fuzz_levels(_lib, seed)
    except AssertionError as e:
        bad += 1; print("GROUP MISMATCH seed", seed, str(e)[:200], flush=True)
print("soak done:", n, "cases +", n_groups, "level groups,", bad, "mismatches")
sys.exit(1 if bad else 0)
