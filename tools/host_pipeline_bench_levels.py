#!/usr/bin/env python3
"""Host-to-host rate of smm_group_apply_host on config-3 shaped fields (75 masked ocean levels,
1442x1021 -> r360x180 conservative): level-major packing of the used cells (round 6) against whole rows, with the
stage split of smm_debug_host_stats.   python tools/host_pipeline_bench_levels.py [time steps, default 16]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smmregrid_amd import OperatorGroup, _lib, gridgen
from smmregrid_amd.device import result_cache
from smmregrid_amd.weights import compute_weights_matrix3d

n_t = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nx, ny, n_lev = 1442, 1021, 75
src = gridgen.regular_grid(nx, ny)
masks = gridgen.synthetic_ocean_masks(nx, ny, n_lev)
w3 = gridgen.ConservativeLevels(src, "r360x180").stack(masks, np.arange(n_lev, dtype=np.float64))
ops = compute_weights_matrix3d(w3, "lev", device=0)
imask = np.stack([op.mask_apply(masks[i]) for i, op in enumerate(ops)])
for i, op in enumerate(ops):
    op.set_epilogue(imask[i], w3["dst_grid_frac"].values[i])
grp = OperatorGroup(ops)
rng = np.random.default_rng(0)
slab = 10.0 + 5.0 * rng.standard_normal((n_lev, src.size))
slab[masks == 0] = np.nan
x = np.ascontiguousarray(np.broadcast_to(slab[None, :, None, :], (n_t, n_lev, 1, src.size)))
lev = np.arange(n_lev, dtype=np.int32)
ml = (~(imask == 1).all(axis=1)).astype(np.uint8)
out = {"time_steps": n_t, "levels": n_lev, "input_GB": x.nbytes / 1e9,
       "used_fraction": sum(op.n_used_src for op in ops) / (n_lev * src.size)}
ref = None
for mode, fl in (("packed", 0), ("whole_rows", _lib.APPLY_HOST_NO_PACK)):
    y = grp.apply_host(x, lev, ml, masked=True, remap_area_min=0.5, flags=fl)      # warm-up
    y = None
    result_cache.wait()                # the steady state of a loop: the page-locked result block prepared in the background is there
    _lib.host_stats(reset=True)
    t0 = time.perf_counter()
    y = grp.apply_host(x, lev, ml, masked=True, remap_area_min=0.5, flags=fl)
    dt = time.perf_counter() - t0
    st = _lib.host_stats(reset=True)
    out[mode] = {"seconds": dt, "cells_per_s": n_t * n_lev * 64800 / dt, "host_GBs": x.nbytes / dt / 1e9}
    out[mode]["stages"] = {k: round(v, 1) for k, v in st.items()}
    if ref is None:
        ref = y
    else:
        out["bit_identical"] = bool(np.array_equal(np.isnan(y), np.isnan(ref)) and
                                    np.array_equal(y[~np.isnan(y)], ref[~np.isnan(ref)]))
print(json.dumps(out))
