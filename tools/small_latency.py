#!/usr/bin/env python3
"""Latency of the facade on reference-sized inputs (12 x 73 x 144 f32 -> r180x90), host in / host out."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smmregrid_amd import CdoGenerate, DataArray, Regridder, gridgen

src = gridgen.regular_grid_from_centers(np.arange(144) * 2.5, -90 + np.arange(73) * 2.5)
rng = np.random.default_rng(0)
x = (280 + 10 * rng.standard_normal((12, 73, 144))).astype(np.float32)
field = DataArray(x, dims=("time", "lat", "lon"), coords={"lat": src.lat, "lon": src.lon}, name="2t")
t0 = time.perf_counter(); w = CdoGenerate(field, "r180x90").weights(method="con"); t1 = time.perf_counter()
rg = Regridder(weights=w); t2 = time.perf_counter()
print(f"weights {1e3*(t1-t0):.1f} ms, Regridder init {1e3*(t2-t1):.1f} ms")
for i in range(5):
    t = time.perf_counter(); out = rg.regrid(field); dt = time.perf_counter() - t
    print(f"regrid call {i}: {1e3*dt:.3f} ms", out.shape)
op = rg.grids[0].weights_matrix
xx = x.reshape(12, -1)
for i in range(3):
    t = time.perf_counter(); y = op.apply_host(xx, remap_area_min=0.5); dt = time.perf_counter() - t
    print(f"apply_host {i}: {1e3*dt:.3f} ms")
