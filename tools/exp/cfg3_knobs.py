#!/usr/bin/env python3
"""Round 6, config 3 in the native layout: does any launch-shape knob change how much of the latitude halo (1.20 x re-fetch,
DESIGN.md section 5) the L2 catches?  One process, the problem built once, knobs alternated `reps` times; kernel ms from HIP events.
    python tools/exp/cfg3_knobs.py [reps]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from smmregrid_amd import _lib  # noqa: E402
from smmregrid_amd.device import DeviceArray, Event, set_device, synchronize  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
set_device(0)
prob = bench.ProblemLevels("cfg3", 0, 0)
y = DeviceArray(prob.y_shape, np.float64)
variants = [{}, {"tile_x_loads": 1}, {"tile_x_loads": 2}, {"xcd_run": 6}, {"xcd_run": 12}, {"xcd_run": 64}, {"xcd_run": 128},
            {"xcd_run": -1}, {"tile_walk": 120}, {"tile_walk": 32}, {"tile_walk": 16}, {"tile_links": 1}]
if len(sys.argv) > 2 and sys.argv[2] == "occupancy":
    # experiment build only (smm_launch.hpp adds the sb_lds_pad knob to the tile kernel's dynamic LDS): fewer workgroups per CU
    variants = [{}] + [{"sb_lds_pad": pad} for pad in (2048, 4096, 6144, 8192, 12288, 18432, 26624, 40960)] + \
               [{"sb_lds_pad": 12288, "xcd_run": 64}, {"sb_lds_pad": 18432, "xcd_run": 64}]
res = {json.dumps(v): [] for v in variants}
for rep in range(reps):
    for v in variants:
        with _lib.tuning(**v):
            for _ in range(2):
                prob.run(y, 0)
            synchronize()
            ev = [(Event(), Event()) for _ in range(8)]
            for a, b in ev:
                a.record()
                prob.run(y, 0)
                b.record()
            synchronize()
            res[json.dumps(v)].append(float(np.mean([a.elapsed_ms(b) for a, b in ev])))
chk = prob.spot_check(y)
for k, v in res.items():
    print("%-28s %s" % (k, " ".join("%.3f" % t for t in v)))
print("spot check of the last output:", chk["bit_equal_to_oracle"])
