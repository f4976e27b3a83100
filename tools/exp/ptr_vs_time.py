"""Does the kernel time of config 2 follow the placement of X / Y?  One line per process."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smmregrid_amd import SparseOperator, gridgen
from smmregrid_amd.device import DeviceArray, Event
w = gridgen.generate_weights("r1440x721", "r360x180", method="bil")
op = SparseOperator(w.sizes["src_grid_size"], w.sizes["dst_grid_size"], w["src_address"].values,
                    w["dst_address"].values, w["remap_matrix"].values, device=0)
pad_mb = int(os.environ.get("PAD_MB", "0"))
pad = DeviceArray((pad_mb << 20,), np.uint8) if pad_mb else None
x = DeviceArray((3600, op.n_src), np.float64).fill_random(1, 250.0, 30.0)
y = DeviceArray((3600, op.n_dst), np.float64)
ts = []
for _ in range(12):
    a, b = Event(), Event()
    a.record(); op.apply(x, y=y); b.record(); b.synchronize()
    ts.append(a.elapsed_ms(b))
print(f"x {x.ptr:#x} (mod 1 GiB {x.ptr % (1 << 30):#x}, mod 2 MiB {x.ptr % (1 << 21):#x})  y {y.ptr:#x}  "
      f"median {np.median(ts[2:]):.3f} ms  min {np.min(ts):.3f}")
