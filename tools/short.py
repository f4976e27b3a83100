#!/usr/bin/env python3
"""Compact one-line summary of a bench.py JSON line read from stdin (optionally prefixed by argv[1:])."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    print(" ".join(sys.argv[1:]), d["config"]["kernel"], f"ms/step {d['ms_per_step']:.3f}",
          f"kernel_ms {r['kernel_ms']:.3f}", f"cells/s {d['value']:.4g}", f"alg GB/s {r['achieved']:.0f}",
          f"frac {r['frac']:.3f}", "cpu", d.get("cpu_baseline", {}).get("value"))
