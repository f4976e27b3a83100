#!/usr/bin/env python3
"""The two user-visible paths of the drop-in, timed host to host with the CPU oracle beside them
("also report host-to-host time"; BASELINE.md section 1: the only sizes the reference
publishes anything on are its own test files).  Used by bench.py (blocks `host_to_host` and
`reference_sized` of the final line) and runnable on its own:

    python tools/user_path_bench.py [rows]

host_to_host     config 2 (r1440x721 -> r360x180 bilinear, f64), `rows` batch rows in host memory
                 through smm_apply_host -- what Regridder.apply_weights receives (regrid.py:537-541) --
                 from pageable and from pinned buffers, with the staging copy packing the used source
                 cells (default) and shipping whole rows (SMM_APPLY_HOST_NO_PACK).  PCIe-bound: never
                 the bench's `value`.
reference_sized  the reference's own fields (fixtures under tests/golden, time axis tiled to the record
                 counts of speed-evaluation.ipynb cell 5) through the call pattern of that cell --
                 weights pre-computed, `Regridder(weights=w)`, `regrid(DataArray)` with host input and host
                 output -- median of >= 20 calls, and the CPU oracle (numpy + scipy.sparse on one core,
                 oracle.c on one thread) on the same product.  These fields are 10^3 - 10^4 times smaller
                 than config 2: launch latency, PCIe and Python decide, not HBM.

The oracle is the checker and the CPU leg here, never part of what is timed as the GPU path.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _median_ms(fn, reps, budget_s=None):
    """Median / minimum wall time of fn() in ms over `reps` calls (fewer when `budget_s` runs out, never below 3)."""
    times, t_all = [], time.perf_counter()
    for i in range(reps):
        t0 = time.perf_counter()
        fn()
        times.append((time.perf_counter() - t0) * 1e3)
        if budget_s is not None and i >= 2 and time.perf_counter() - t_all > budget_s:
            break
    return float(np.median(times)), float(np.min(times)), len(times)


# ------------------------------------------------------------------------------------------ host to host

def _ceilings(rows_bytes=512 << 20):
    """The two rates the host pipeline cannot beat, measured in the same run: pinned hipMemcpy H2D / D2H / both
    directions at once (GB/s per direction), and the library's own multi-threaded host -> host copy into pinned
    memory (GB/s of bytes copied; the memory system moves at least twice that: one read, one write)."""
    import ctypes
    from smmregrid_amd import _lib, pinned_empty
    from smmregrid_amd.device import DeviceArray, Stream, synchronize
    n = rows_bytes // 8
    h1, h2 = pinned_empty((n,), np.float64), pinned_empty((n,), np.float64)
    h1[...] = 1.0
    h2[...] = 2.0
    d1, d2 = DeviceArray((n,), np.float64), DeviceArray((n,), np.float64)
    s1, s2 = Stream(), Stream()

    def h2d():
        d1.copy_from_host(h1, stream=s1)
        s1.synchronize()

    def d2h():
        d2.to_host(out=h2, stream=s2)

    def both():
        d1.copy_from_host(h1, stream=s1)
        _lib.call("smm_memcpy_d2h", h2.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d2.ptr), d2.nbytes, s2.handle)
        s1.synchronize()
        s2.synchronize()

    out = {}
    for name, fn in (("h2d", h2d), ("d2h", d2h), ("both", both)):
        fn()
        _, ms_min, _ = _median_ms(fn, 5)
        out[f"pcie_{name}_GBs"] = rows_bytes / (ms_min * 1e-3) / 1e9
    synchronize()
    page = np.full(n, 3.0)

    def hcopy():
        _lib.call("smm_host_memcpy", h1.ctypes.data_as(ctypes.c_void_p), page.ctypes.data_as(ctypes.c_void_p), page.nbytes)

    hcopy()
    _, ms_min, _ = _median_ms(hcopy, 7)
    out["host_copy_GBs"] = rows_bytes / (ms_min * 1e-3) / 1e9
    d1.free()
    d2.free()
    return out


def host_to_host(device=0, rows=512, cpu_cells_per_s=None, reps=9):
    """smm_apply_host on config-2 rows held in host memory; returns the compact block of the bench line: per mode the
    median and best of `reps` calls, where the time went (smm_debug_host_stats: pack / copy-in, H2D, kernel, D2H,
    copy-out, wait), and the fraction of the two measured ceilings -- PCIe and host memory -- the mode runs at."""
    from smmregrid_amd import SparseOperator, _lib, gridgen, pinned_empty
    w = gridgen.bilinear_weights("r1440x721", "r360x180")
    S, D = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
    op = SparseOperator(S, D, w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values, device=device)
    op.set_epilogue(w["dst_grid_imask"].values, w["dst_grid_frac"].values)
    U = op.n_used_src
    used_lines = int(np.unique(op.used_sources() // 8).size)          # 64-B lines of a source row the pack reads
    rng = np.random.default_rng(20260723)
    row = 250.0 + 30.0 * rng.standard_normal(S)
    ceil = _ceilings()
    out = {"workload": f"cfg2 rows in host memory, {rows} x {S} f64 -> {rows} x {D} f64", "rows": int(rows), "reps": int(reps),
           "pcie_bytes_per_row": {"packed": int((U + D) * 8), "whole_rows": int((S + D) * 8)},
           "ceilings": {k: round(v, 1) for k, v in ceil.items()}, "staging_threads": None}
    # host-memory bytes one row costs (reads + writes that reach DRAM; a plain store into a line not in cache reads it
    # first, a non-temporal one does not): pack = touched source lines + the packed block, the DMA engine's read of the
    # staging block and write of Y, and -- for pageable buffers -- the copies between the caller's arrays and staging
    nt = 1
    host_bytes = {
        ("pageable", "packed"): used_lines * 64 + U * 8 * nt + U * 8 + D * 8 + 3 * D * 8,
        ("pinned", "packed"): used_lines * 64 + U * 8 * nt + U * 8 + D * 8,
        ("pageable", "whole_rows"): 3 * S * 8 + S * 8 + D * 8 + 3 * D * 8,
        ("pinned", "whole_rows"): S * 8 + D * 8,
    }
    first = None
    for kind in ("pageable", "pinned"):
        alloc = (lambda s, d: np.empty(s, d)) if kind == "pageable" else pinned_empty
        x = alloc((rows, S), np.float64)
        x[...] = row[None, :]
        x[:, ::97] += np.arange(rows)[:, None]           # rows differ
        y = alloc((rows, D), np.float64)
        for mode, fl in (("packed", 0), ("whole_rows", _lib.APPLY_HOST_NO_PACK)):
            op.apply_host(x, out=y, remap_area_min=0.5, flags=fl)           # warm-up: staging buffers, page faults
            _lib.host_stats(reset=True)
            ms, ms_min, n = _median_ms(lambda: op.apply_host(x, out=y, remap_area_min=0.5, flags=fl), reps)
            st = _lib.host_stats(reset=True)
            calls = max(st["calls"], 1.0)
            e = {"cells_per_s": rows * D / (ms * 1e-3), "cells_per_s_best": rows * D / (ms_min * 1e-3), "ms": ms, "ms_min": ms_min,
                 "calls": n, "host_GBs": (x.nbytes + y.nbytes) / (ms * 1e-3) / 1e9,
                 "stage_ms": {k[:-3]: round(st[k] / calls, 2) for k in ("stage_in_ms", "h2d_ms", "kernel_ms", "d2h_ms",
                                                                        "copy_out_ms", "wait_ms", "total_ms")}}
            # X over PCIe against the one-direction H2D rate (Y, a fifth to a sixteenth of the bytes, travels the other way)
            x_b = (out["pcie_bytes_per_row"][mode] - D * 8) * rows
            e["pcie_frac"] = (x_b / (ms * 1e-3) / 1e9) / ceil["pcie_h2d_GBs"]
            e["host_mem_frac"] = (host_bytes[(kind, mode)] * rows / (ms * 1e-3) / 1e9) / (2.0 * ceil["host_copy_GBs"])
            out[f"{kind}_{mode}"] = e
            out["staging_threads"] = int(st["threads"])
            if first is None:
                first = y[rows // 2].copy(), x[rows // 2].copy()
            else:                                                           # every mode gives the same bits
                e["same_bits"] = bool(np.array_equal(y[rows // 2], first[0], equal_nan=True))
        del x, y
    # one row of the result against the oracle (its own CSR from the links)
    from oracle import oracle
    csr = oracle.coo_to_csr_c(S, D, w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values)
    ref = oracle.apply_c(csr, first[1][None, :], False, None, w["dst_grid_frac"].values, 0.5)[0]
    out["spot_check"] = bool(np.array_equal(ref, first[0], equal_nan=True))
    if cpu_cells_per_s:
        out["cpu_cells_per_s"] = float(cpu_cells_per_s)      # the same product on the host's cores (cpu_baseline)
    out["bound"] = _what_bounds(out)
    op.close()
    return out


def _what_bounds(out):
    """The binding ceiling per mode family, from the fractions above (a short string for the bench line)."""
    def worst(e):
        return "host memory" if e["host_mem_frac"] >= e["pcie_frac"] else "PCIe"
    return f"packed: {worst(out['pageable_packed'])}; whole rows: {worst(out['pinned_whole_rows'])}"


# ------------------------------------------------------------------------------------------ reference-sized

def _tile_time(a, n):
    """The fixture's time steps repeated up to the reference's record count (values shifted per copy)."""
    reps = -(-n // a.shape[0])
    return np.concatenate([a + np.asarray(k, a.dtype) for k in range(reps)], axis=0)[:n]


def _cases():
    """name -> (source for weight generation, field to regrid, target, Regridder keywords)."""
    from smmregrid_amd import DataArray, Dataset
    cases = {}
    z = np.load(os.path.join(GOLDEN, "2t_era5.npz"))
    f = DataArray(z["t2m"], dims=("time", "lat", "lon"), coords={"time": z["time"], "lat": z["lat"], "lon": z["lon"]},
                  name="2t", attrs={"units": "K"})
    cases["2t_era5"] = (f, f, "r360x180", {})
    z = np.load(os.path.join(GOLDEN, "tas_healpix2.npz"))
    coords = {"time": np.arange(12), "lat": DataArray(z["lat"], dims=("pix",), attrs={"units": "radian"}),
              "lon": DataArray(z["lon"], dims=("pix",), attrs={"units": "radian"})}
    f = DataArray(_tile_time(z["tas"], 12), dims=("time", "pix"), coords=coords, name="tas",
                  attrs={"CDI_grid_type": "unstructured"})
    cases["tas_healpix2"] = (f, f, "r360x180", {})
    z = np.load(os.path.join(GOLDEN, "tas_ecearth.npz"))
    coords = {"time": np.arange(12),
              "lat": DataArray(z["lat"], dims=("lat",), attrs={"units": "degrees_north", "bounds": "lat_bnds"}),
              "lon": DataArray(z["lon"], dims=("lon",), attrs={"units": "degrees_east", "bounds": "lon_bnds"})}
    ds = Dataset({"tas": DataArray(_tile_time(z["tas"], 12), dims=("time", "lat", "lon"), coords=coords, name="tas",
                                   attrs={"units": "K"})})
    ds["lat_bnds"] = (("lat", "bnds"), z["lat_bnds"])
    ds["lon_bnds"] = (("lon", "bnds"), z["lon_bnds"])
    cases["tas_ecearth"] = (ds, ds["tas"], "r360x180", {})
    z = np.load(os.path.join(GOLDEN, "temp3d_fesom.npz"))
    coords = {"time": np.arange(12), "nz1": z["nz1"],
              "lon": DataArray(z["lon"], dims=("nod2",), attrs={"units": "degrees_east", "bounds": "lon_bnds"}),
              "lat": DataArray(z["lat"], dims=("nod2",), attrs={"units": "degrees_north", "bounds": "lat_bnds"})}
    temp = np.stack([z["temp"] + np.float32(t) for t in range(12)])
    ds = Dataset({"temp": DataArray(temp, dims=("time", "nz1", "nod2"), coords=coords, name="temp",
                                    attrs={"units": "degC", "coordinates": "lat lon"})})
    ds["lon_bnds"] = (("nod2", "nv"), z["lon_bnds"].astype(np.float64))
    ds["lat_bnds"] = (("nod2", "nv"), z["lat_bnds"].astype(np.float64))
    cases["temp3d_fesom"] = (ds, ds["temp"], "r360x180", {})
    z = np.load(os.path.join(GOLDEN, "ua_ipsl_t0.npz"))
    f = DataArray(_tile_time(z["ua"][None], 2), dims=("time", "plev", "lat", "lon"),
                  coords={"time": [0, 1], "plev": z["plev"], "lat": z["lat"], "lon": z["lon"]}, name="ua")
    cases["ua_ipsl_nan"] = (Dataset({"ua": f}), f, "r90x45", {"check_nan": True})      # basic_test.py:95-102
    return cases


def _oracle_product(rg, field):
    """(fn_scipy, fn_c) computing the same product as rg.regrid(field) with the oracle on prebuilt CSRs, and the
    reference result of fn_c for the bit comparison."""
    from oracle import oracle
    g = rg.grids[0]
    w = g.weights
    x = np.asarray(field.values)
    rm = w["remap_matrix"].values
    if g.mask_dim:
        ll = w["link_length"].values
        n_lev = ll.size
        S, D = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
        csrs = [oracle.coo_to_csr_c(S, D, w["src_address"].values[i, :ll[i]], w["dst_address"].values[i, :ll[i]],
                                    rm[i, :ll[i], 0] if rm.ndim == 3 else rm[i, :ll[i]]) for i in range(n_lev)]
        lev_axis = field.dims.index(g.mask_dim)
        n_h = len(g.horizontal_dims) if getattr(g, "horizontal_dims", None) else x.ndim - lev_axis - 1
        xs = x.reshape(x.shape[:x.ndim - n_h] + (-1,))
        masked = np.atleast_1d(np.asarray(g.masked)).astype(bool)
        masked = np.broadcast_to(masked, (n_lev,))
        frac = w["dst_grid_frac"].values if "dst_grid_frac" in w else None
        args = (csrs, xs, lev_axis, np.arange(n_lev), masked, w["dst_grid_imask"].values, frac, rg.remap_area_min, True)
        return (lambda: oracle.apply_levels(*args, use_c=False)), (lambda: oracle.apply_levels(*args, use_c=True))
    S, D = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
    csr = oracle.coo_to_csr_c(S, D, w["src_address"].values, w["dst_address"].values, rm)
    xs = x.reshape(-1, S)
    frac = w["dst_grid_frac"].values if "dst_grid_frac" in w else None
    args = (csr, xs, bool(np.asarray(g.masked).any()), w["dst_grid_imask"].values, frac, rg.remap_area_min)
    return (lambda: oracle.apply(*args)), (lambda: oracle.apply_c(*args, threads=1))


def reference_sized(device=0, reps=25, budget_s=40.0, only=None):
    """Per reference field: Regridder(weights=w) init ms, regrid() ms host to host (median), the oracle's ms for the
    same product (scipy on one core / oracle.c on one thread), bit equality of the two results."""
    from smmregrid_amd import Regridder
    out = {}
    t_start = time.perf_counter()
    for name, (source, field, target, kw) in _cases().items():
        if only and name not in only:
            continue
        if time.perf_counter() - t_start > budget_s:
            out[name] = {"skipped": "time budget"}
            continue
        try:
            t0 = time.perf_counter()
            gen = Regridder(source_grid=source, target_grid=target, method="con", device=device, **kw)
            w = gen.grids[0].weights                      # native generator (no cdo here): outside what is timed below
            gen_ms = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            rg = Regridder(weights=w, device=device)      # speed-evaluation.ipynb cell 5: weights pre-computed
            init_ms = (time.perf_counter() - t0) * 1e3
            res = rg.regrid(field)                        # warm-up (staging buffers) + the result to compare
            ms, ms_min, n = _median_ms(lambda: rg.regrid(field).values, reps, budget_s=3.0)
            f_scipy, f_c = _oracle_product(rg, field)
            ref = f_c()
            cpu_ms, _, _ = _median_ms(f_scipy, 7, budget_s=3.0)
            c1_ms, _, _ = _median_ms(f_c, 7, budget_s=3.0)
            got = np.asarray(res.values).reshape(ref.shape)
            g = rg.grids[0]
            out[name] = {"in": "x".join(str(s) for s in field.shape), "to": target, "cells": int(ref.size),
                         "init_ms": round(init_ms, 2), "regrid_ms": round(ms, 3), "regrid_ms_min": round(ms_min, 3),
                         "calls": n, "cpu_scipy_ms": round(cpu_ms, 3), "cpu_c1_ms": round(c1_ms, 3),
                         "gpu_over_scipy": round(cpu_ms / ms, 2), "bit_equal": bool(np.array_equal(got, ref, equal_nan=True)),
                         "levels": int(len(g.weights_matrix)) if g.mask_dim else 0, "weights_ms": round(gen_ms - init_ms, 1)}
        except Exception as exc:      # one case must not lose the bench line
            out[name] = {"error": repr(exc)[:90]}
    return out


if __name__ == "__main__":
    # python tools/user_path_bench.py [rows] [--only-h2h] [knob=value ...]   (knobs: smm_debug_set_tuning names, A/B runs)
    n_rows = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 512
    from smmregrid_amd import _lib
    for kv in [a for a in sys.argv[1:] if "=" in a]:
        _lib.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
    res = {"host_to_host": host_to_host(rows=n_rows)}
    if "--only-h2h" not in sys.argv:
        res["reference_sized"] = reference_sized()
    print(json.dumps(res))
