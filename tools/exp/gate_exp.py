import sys, time, json
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
from smmregrid_amd import _lib, pinned_empty
from smmregrid_amd.device import set_device
set_device(0)
prob = bench.ProblemLevels("cfg3", 0, 0)
out = {}
for lv in (0, 20, 40):
    op = prob.ops[lv]
    S, D, U = op.n_src, op.n_dst, op.n_used_src
    rows = 256
    for kind in ("pageable", "pinned"):
        alloc = (lambda s, d: np.empty(s, d)) if kind == "pageable" else pinned_empty
        x = alloc((rows, S), np.float64); x[...] = prob.slab[lv][None, :]
        y = alloc((rows, D), np.float64)
        res = {}
        for mode, fl in (("packed", 0), ("whole", _lib.APPLY_HOST_NO_PACK)):
            op.apply_host(x, out=y, masked=True, remap_area_min=0.5, flags=fl)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); op.apply_host(x, out=y, masked=True, remap_area_min=0.5, flags=fl); ts.append(time.perf_counter() - t0)
            res[mode] = round(float(np.median(ts)) * 1e3, 2)
            res[mode + "_chk"] = float(np.nansum(y[rows // 2]))
        out[f"level {lv} U/S={U / S:.2f} {kind}"] = res
        del x, y
print(json.dumps(out, indent=1))
