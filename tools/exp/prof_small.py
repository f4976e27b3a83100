import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np
import user_path_bench as u
from smmregrid_amd import Regridder
cases = u._cases()
for name in ("2t_era5", "temp3d_fesom", "ua_ipsl_nan"):
    source, field, target, kw = cases[name]
    gen = Regridder(source_grid=source, target_grid=target, method="con", device=0, **kw)
    rg = Regridder(weights=gen.grids[0].weights, device=0)
    for _ in range(5): rg.regrid(field)
    t0 = time.perf_counter()
    for _ in range(200): rg.regrid(field)
    print(name, "ms per call", (time.perf_counter() - t0) / 200 * 1e3)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): rg.regrid(field)
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4500])
