"""Hypothesis table for the one reference-held number of the path:
/root/reference/tests/basic_test.py:95-102 expects 589 missing cells at level 1 of ua-ipsl.nc
regridded to r90x45 with `check_nan=True` (conservative weights made by `cdo gencon`, one run
per level, `dst_grid_frac < 0.5` -> NaN, regrid.py:562-565).  No `cdo` exists here, so the
weights come from the native generator, which gives 567.  This script recomputes the count under
every variation of geometry / mask / cut that was tried (round 4), from the committed fixture
`tests/golden/ua_ipsl_t0.npz` (and, when /root/reference is present, the other two time steps).

    python tools/known_answer_589.py            # prints the table of DESIGN.md section 2
"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from smmregrid_amd import gridgen                      # noqa: E402
from smmregrid_amd.gridgen import DEG, _overlap_1d     # noqa: E402


def mid_bounds(c, clamp=None):
    """CDO's grid_gen_corners (+ grid_check_lat_borders when `clamp`)."""
    b = np.empty(c.size + 1)
    b[1:-1] = 0.5 * (c[1:] + c[:-1])
    b[0] = 2 * c[0] - b[1]
    b[-1] = 2 * c[-1] - b[-2]
    return np.clip(b, *clamp) if clamp else b


def frac(src_lon_b, src_lat_b, dst_lon_b, dst_lat_b, mask2d, planar=False, by_count=False):
    """Unmasked share of every destination cell (separable lon x lat overlaps)."""
    ny, nx = mask2d.shape
    f = (lambda v: v * DEG) if planar else (lambda v: np.sin(v * DEG))
    ld, ls, lw = _overlap_1d(src_lon_b, dst_lon_b, periodic=360.0)
    td, ts, tw = _overlap_1d(f(src_lat_b), f(dst_lat_b))
    a = np.zeros((dst_lon_b.size - 1, nx))
    b = np.zeros((dst_lat_b.size - 1, ny))
    a[ld, ls] = 1.0 if by_count else lw
    b[td, ts] = 1.0 if by_count else tw
    tot = b @ np.ones((ny, nx)) @ a.T
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(tot > 0, (b @ mask2d.astype(float) @ a.T) / tot, 0.0), (a, b)


def main():
    z = np.load(os.path.join(ROOT, "tests", "golden", "ua_ipsl_t0.npz"))
    ua, lat, lon = z["ua"], z["lat"], z["lon"]
    dst = gridgen.parse_grid("r90x45")
    m = np.isfinite(ua[1])
    slonb, slatb = mid_bounds(lon), mid_bounds(lat, (-90, 90))
    rows = []

    def add(name, fr, cut=0.5, le=False):
        rows.append((name, int(((fr <= cut) if le else (fr < cut)).sum())))

    base, (a, b) = frac(slonb, slatb, dst.lon_b, dst.lat_b, m)
    add("baseline: bounds at mid-points, poles clamped, exact (lon, sin lat) areas, frac = unmasked / cell", base)
    add("  the same through gridgen.generate_weights (what the test runs)",
        gridgen.generate_weights(gridgen.regular_grid_from_centers(lon, lat), "r90x45", "con",
                                 src_mask=m.ravel())["dst_grid_frac"].values)
    add("cut `<=` instead of `<`", base, le=True)
    for cut in (0.45, 0.55, 0.5555, 0.56, 0.6, 0.9, 0.909):
        add(f"cut at {cut} (0.909: links to masked cells kept, 1e20 * w > 1e19)", base, cut)
    add("areas in (lon, lat) instead of (lon, sin lat)", frac(slonb, slatb, dst.lon_b, dst.lat_b, m, planar=True)[0])
    add("pole rows of the source not clamped at +-90",
        frac(slonb, mid_bounds(lat), dst.lon_b, np.r_[-91, dst.lat_b[1:-1], 91], m)[0])
    add("source latitudes as 143 equal cells of 180/143 degrees",
        frac(slonb, np.linspace(-90, 90, 144), dst.lon_b, dst.lat_b, m)[0])
    add("target longitude cells [0, 4] instead of [-2, 2]", frac(slonb, slatb, dst.lon_b + 2, dst.lat_b, m)[0])
    add("source longitude cells [0, 2.5] instead of [-1.25, 1.25]", frac(slonb + 1.25, slatb, dst.lon_b, dst.lat_b, m)[0])
    add("both longitudes as left edges", frac(slonb + 1.25, slatb, dst.lon_b + 2, dst.lat_b, m)[0])
    add("target r90x45 as 45 latitude CENTRES -90 .. 90 (not CDO's r-grid: tests/data/r360x180.nc is cell-centred)",
        frac(slonb, slatb, dst.lon_b, mid_bounds(np.linspace(-90, 90, 45), (-90, 90)), m)[0])
    add("target latitude cells shifted by half a cell north / south",
        frac(slonb, slatb, dst.lon_b, np.clip(dst.lat_b + 2, -90, 90), m)[0])
    add("  (south)", frac(slonb, slatb, dst.lon_b, np.clip(dst.lat_b - 2, -90, 90), m)[0])
    add("missing-value pattern shifted by one source row north / south", frac(slonb, slatb, dst.lon_b, dst.lat_b, np.roll(m, 1, axis=0))[0])
    add("  (south)", frac(slonb, slatb, dst.lon_b, dst.lat_b, np.roll(m, -1, axis=0))[0])
    add("missing-value pattern shifted by one source column east / west", frac(slonb, slatb, dst.lon_b, dst.lat_b, np.roll(m, 1, axis=1))[0])
    add("  (west)", frac(slonb, slatb, dst.lon_b, dst.lat_b, np.roll(m, -1, axis=1))[0])
    add("frac by NUMBER of overlapping source cells, `<`", frac(slonb, slatb, dst.lon_b, dst.lat_b, m, by_count=True)[0])
    add("frac by number of overlapping source cells, `<=`",
        frac(slonb, slatb, dst.lon_b, dst.lat_b, m, by_count=True)[0], le=True)
    jj = np.abs(dst.lat[:, None] - lat[None, :]).argmin(1)
    ii = np.abs(((dst.lon[:, None] - lon[None, :] + 180) % 360) - 180).argmin(1)
    rows.append(("nearest source cell is missing (not a conservative rule; for scale)", int((~m[jj][:, ii]).sum())))
    for lev in (0, 2):
        add(f"level {lev} instead of level 1 (off-by-one in -sellevidx)",
            frac(slonb, slatb, dst.lon_b, dst.lat_b, np.isfinite(ua[lev]))[0])

    ref_file = "/root/reference/tests/data/ua-ipsl.nc"
    if os.path.exists(ref_file):           # exploration only; nothing on the GPU box reads this
        from smmregrid_amd.io import open_dataset
        full = open_dataset(ref_file)["ua"].values
        tot = b @ np.ones(m.shape) @ a.T
        for t in (1, 2):
            mw = np.isfinite(full[t, 1])
            un = b @ mw.astype(float) @ a.T
            bad = b @ (mw & ~m).astype(float) @ a.T
            with np.errstate(invalid="ignore", divide="ignore"):
                wbad = np.where(un > 0, bad / un, 0.0)
            rows.append((f"weights from time step {t}, field of step 0 (frac cut or 1e20 * w > 1e19)",
                         int(((un / tot < 0.5) | (wbad > 0.1)).sum())))

    scan = []
    for dl, sl, planar in itertools.product((0, 2, -2, 1, -1), (0, 1.25, -1.25, 0.625, -0.625), (False, True)):
        scan.append(int((frac(slonb + sl, slatb, dst.lon_b + dl, dst.lat_b, m, planar=planar)[0] < 0.5).sum()))
    rows.append((f"50 half / quarter-cell longitude shifts of either grid: min / mean / max",
                 f"{min(scan)} / {np.mean(scan):.0f} / {max(scan)}"))

    width = max(len(r[0]) for r in rows)
    print(f"{'variation':{width}s}  missing cells at level 1 (reference expects 589)")
    for name, c in rows:
        print(f"{name:{width}s}  {c}")


if __name__ == "__main__":
    main()
