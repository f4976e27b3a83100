"""CPU check of the hypothesis table behind the reference's one stored number
(/root/reference/tests/basic_test.py:95-102 expects 589 missing cells)."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("known_answer_589", os.path.join(ROOT, "tools", "known_answer_589.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_native_conservative_count_and_its_sensitivity():
    """The native generator's dst_grid_frac (what Regridder(check_nan=True) cuts at 0.5) equals the separable
    overlap computation of the tool, gives 567 at level 1 and none at the top level, and a half-cell shift of
    the target longitudes alone moves the count to 583: only CDO's exact geometry could decide 567 vs 589."""
    t = _tool()
    from smmregrid_amd import gridgen
    z = np.load(os.path.join(ROOT, "tests", "golden", "ua_ipsl_t0.npz"))
    ua, lat, lon = z["ua"], z["lat"], z["lon"]
    dst = gridgen.parse_grid("r90x45")
    slonb, slatb = t.mid_bounds(lon), t.mid_bounds(lat, (-90, 90))
    m = np.isfinite(ua[1])
    base, _ = t.frac(slonb, slatb, dst.lon_b, dst.lat_b, m)
    w = gridgen.generate_weights(gridgen.regular_grid_from_centers(lon, lat), "r90x45", "con", src_mask=m.ravel())
    np.testing.assert_allclose(w["dst_grid_frac"].values, base.ravel(), rtol=0, atol=1e-12)
    assert int((base < 0.5).sum()) == 567 and int((base <= 0.5).sum()) == 567
    assert int((t.frac(slonb, slatb, dst.lon_b + 2, dst.lat_b, m)[0] < 0.5).sum()) == 583
    top, _ = t.frac(slonb, slatb, dst.lon_b, dst.lat_b, np.isfinite(ua[-1]))
    assert int((top < 0.5).sum()) == 0
