"""Kernel-time floor: operators with no / few links (Y fill only), B = 1024, D = 64800."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smmregrid_amd import SparseOperator, _lib
from smmregrid_amd.device import DeviceArray, Event, synchronize

S, D, B = 1472282, 64800, 1024
x = DeviceArray((B, S), np.float64).fill_random(1, 250.0, 30.0)
y = DeviceArray((B, D), np.float64)
rng = np.random.default_rng(0)
for name, nlinked in (("empty", 0), ("1 linked row", 1), ("64 linked rows (one block)", 64), ("5% rows", D // 20), ("all rows 1 link", D)):
    rows = np.sort(rng.choice(D, size=nlinked, replace=False)) if nlinked else np.zeros(0, np.int64)
    src = (rows * 20 % S + 1).astype(np.int32)
    dst = (rows + 1).astype(np.int32)
    op = SparseOperator(S, D, src, dst, np.ones(rows.size), device=0)
    for kern, fl in (("tile", _lib.APPLY_KERNEL_TILE), ("sell", _lib.APPLY_KERNEL_SELL)):
        if kern == "tile" and not op.plan_info()["tile_plan"]:
            continue
        ts = []
        for _ in range(6):
            a, b = Event(), Event()
            a.record(); op.apply(x, y=y, flags=fl); b.record(); b.synchronize()
            ts.append(a.elapsed_ms(b))
        print(f"{name:28s} {kern}: {np.median(ts[1:]):.3f} ms  (Y = {B*D*8/1e9:.2f} GB -> {B*D*8/np.median(ts[1:])/1e6:.0f} GB/s)")
