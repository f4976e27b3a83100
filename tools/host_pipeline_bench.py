#!/usr/bin/env python3
"""Host-to-host throughput of smm_apply_host (PCIe-inclusive) beside the device-resident kernel rate."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smmregrid_amd import SparseOperator, _lib, gridgen, pinned_empty

rows = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 512
w = gridgen.bilinear_weights("r1440x721", "r360x180")
S, D = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
prune = "--prune" in sys.argv      # drop the exact-zero links (half of them between these aligned grids)
op = SparseOperator(S, D, w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values, device=0,
                    prune_zeros=prune)
op.set_epilogue(w["dst_grid_imask"].values, w["dst_grid_frac"].values)
rng = np.random.default_rng(0)
out = {}
for kind in ("pageable", "pinned"):
    alloc = (lambda s, d: np.empty(s, d)) if kind == "pageable" else pinned_empty
    x = alloc((rows, S), np.float64)
    x[...] = 250 + 30 * rng.standard_normal((1, S))
    y = alloc((rows, D), np.float64)
    for mode, fl in (("packed", 0), ("whole_rows", _lib.APPLY_HOST_NO_PACK)):
        op.apply_host(x, out=y, remap_area_min=0.5, flags=fl)           # warm-up (allocations, page faults)
        t0 = time.perf_counter()
        op.apply_host(x, out=y, remap_area_min=0.5, flags=fl)
        dt = time.perf_counter() - t0
        out[kind + "/" + mode] = {"rows": rows, "seconds": dt, "cells_per_s": rows * D / dt,
                                  "host_GBs": (x.nbytes + y.nbytes) / dt / 1e9}
out["prune_zeros"] = prune
out["used_source_cells"] = op.n_used_src
print(json.dumps(out))
