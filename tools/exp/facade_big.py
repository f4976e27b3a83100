#!/usr/bin/env python3
"""What a user of the drop-in sees at config-2 size: Regridder(weights=w).regrid(DataArray) on a numpy-backed (rows, 721, 1440) f64 field
against the bare smm_apply_host call on the same rows.   python tools/exp/facade_big.py [rows]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smmregrid_amd import DataArray, Regridder, gridgen

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 512
w = gridgen.bilinear_weights("r1440x721", "r360x180")
g = gridgen.parse_grid("r1440x721")
rng = np.random.default_rng(1)
x = 250.0 + 30.0 * rng.standard_normal((rows, 721, 1440))
field = DataArray(x, dims=("time", "lat", "lon"), coords={"time": np.arange(rows), "lat": g.lat, "lon": g.lon}, name="t2m")
rg = Regridder(weights=w)
op = rg.grids[0].weights_matrix
out = {}
def med(fn, n=7):
    from smmregrid_amd.device import result_cache
    fn(); result_cache.wait(); fn(); result_cache.wait(); ts = []     # steady state of a loop: page-locked result blocks are there
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3
out["facade_regrid_ms"] = med(lambda: rg.regrid(field).values)
x2 = x.reshape(rows, -1)
out["apply_host_new_output_ms"] = med(lambda: op.apply_host(x2, remap_area_min=0.5))
y = np.empty((rows, op.n_dst))
out["apply_host_reused_output_ms"] = med(lambda: op.apply_host(x2, out=y, remap_area_min=0.5))
out["cells_per_s_facade"] = rows * op.n_dst / (out["facade_regrid_ms"] * 1e-3)
print(json.dumps(out))
