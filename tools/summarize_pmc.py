#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc counter_collection CSVs to per-launch HBM bytes.

  python tools/summarize_pmc.py --fetch <dir> --write <dir> [--rdreq <dir>] --key cfg2/default/auto/0 \
      --out profiles/r01_cfg2_pmc_summary.json --traffic profiles/traffic.json

Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of a wide
(16 B per lane) coalesced read stream, so it is doubled for the tile kernel (which stages with
16-B-per-lane loads); TCC_EA0_RDREQ x 128 B is the cross-check.  WRITE_SIZE is exact.
"""
import argparse
import collections
import csv
import glob
import json
import os


def mean_counter(d, kernel_substr):
    f = max(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if kernel_substr in r["Kernel_Name"]:
            per_kernel[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # the dominant kernel of the run: set-up launches (mask products of a level group) also match
    name = max(per_kernel, key=lambda k: sum(sum(v) for v in per_kernel[k].values()))
    return {k: (sum(v) / len(v), len(v)) for k, v in per_kernel[name].items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--rdreq")
    ap.add_argument("--kernel", default="_apply_")
    ap.add_argument("--wide-loads", type=int, default=1, help="1: double FETCH_SIZE (16 B/lane streams)")
    ap.add_argument("--key", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--traffic", required=True)
    ap.add_argument("--launches-per-step", type=int, default=1,
                    help="kernel launches that make up one bench step (cfg3sb: one per data level): the per-launch "
                         "means are multiplied by it, so `hbm_bytes_per_launch` stays the bytes of one step")
    ap.add_argument("--git-head", default=os.environ.get("SMM_GIT_HEAD"),
                    help="commit the profiled tree was built from (the GPU box has no .git)")
    a = ap.parse_args()
    fetch, nf = mean_counter(a.fetch, a.kernel)["FETCH_SIZE"]
    write, nw = mean_counter(a.write, a.kernel)["WRITE_SIZE"]
    fetch, write = fetch * a.launches_per_step, write * a.launches_per_step
    out = {"key": a.key, "launches_averaged": {"fetch": nf, "write": nw}, "launches_per_step": a.launches_per_step,
           "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
           "fetch_correction": 2 if a.wide_loads else 1,
           "fetch_bytes": fetch * 1024 * (2 if a.wide_loads else 1), "write_bytes": write * 1024}
    if a.rdreq:
        rd = mean_counter(a.rdreq, a.kernel)
        out["TCC_EA0_RDREQ_sum"] = rd.get("TCC_EA0_RDREQ_sum", (None,))[0]
        out["TCC_EA0_RDREQ_32B_sum"] = rd.get("TCC_EA0_RDREQ_32B_sum", (None,))[0]
        if out["TCC_EA0_RDREQ_sum"]:
            out["TCC_EA0_RDREQ_sum"] *= a.launches_per_step
            out["rdreq_x128B_bytes"] = out["TCC_EA0_RDREQ_sum"] * 128
    out["hbm_bytes_per_launch"] = out["fetch_bytes"] + out["write_bytes"]
    json.dump(out, open(a.out, "w"), indent=1)
    traffic = json.load(open(a.traffic)) if os.path.exists(a.traffic) else {}
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_sha
    out["kernel_sha"] = kernel_source_sha()
    out["git_head"] = a.git_head
    json.dump(out, open(a.out, "w"), indent=1)
    traffic[a.key] = {"hbm_bytes_per_launch": out["hbm_bytes_per_launch"], "source": os.path.basename(a.out),
                      "kernel_sha": out["kernel_sha"], "git_head": a.git_head}
    json.dump(traffic, open(a.traffic, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
