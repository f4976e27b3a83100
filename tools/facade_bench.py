#!/usr/bin/env python3
"""What a user of the drop-in sees: Regridder(weights).regrid(DataArray) on host (numpy) fields of
config-2 shape, wall time per call (PCIe-inclusive; never the bench's `value`)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smmregrid_amd import DataArray, Regridder, gridgen

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 512
src = gridgen.parse_grid("r1440x721")
w = gridgen.bilinear_weights("r1440x721", "r360x180")
rng = np.random.default_rng(0)
x = (250 + 30 * rng.standard_normal((1, 721, 1440))).repeat(rows, axis=0)
fld = DataArray(x, dims=("time", "lat", "lon"), coords={"time": np.arange(rows), "lat": src.lat, "lon": src.lon},
                name="t2m")
out = {}
for name, kw in (("default", {}), ("prune_zero_weights", {"prune_zero_weights": True})):
    t0 = time.perf_counter()
    rg = Regridder(weights=w, device=0, **kw)
    t_init = time.perf_counter() - t0
    rg.regrid(fld)                                  # warm-up: pinned staging buffers, page faults
    t0 = time.perf_counter()
    y = rg.regrid(fld)
    dt = time.perf_counter() - t0
    out[name] = {"rows": rows, "init_s": round(t_init, 3), "regrid_s": round(dt, 4),
                 "cells_per_s": rows * 64800 / dt, "input_GBs": x.nbytes / dt / 1e9}
print(json.dumps(out))
