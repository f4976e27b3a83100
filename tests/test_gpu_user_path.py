"""The user-visible paths the bench line reports (tools/user_path_bench.py): the reference's own fields through
`Regridder(weights=w).regrid()` host to host, and `smm_apply_host` on config-2 rows -- here for their checks (every
result bit-equal to the oracle on the oracle's own CSR, every staging mode the same bits), not for their timings."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_reference_sized_fields_are_bit_equal_to_the_oracle(hip):
    import user_path_bench as u
    out = u.reference_sized(device=0, reps=3, budget_s=120.0)
    assert set(out) == {"2t_era5", "tas_healpix2", "tas_ecearth", "temp3d_fesom", "ua_ipsl_nan"}
    for name, e in out.items():
        assert "error" not in e and "skipped" not in e, (name, e)
        assert e["bit_equal"] is True and e["regrid_ms"] > 0 and e["cpu_scipy_ms"] > 0 and e["init_ms"] > 0, (name, e)
    assert out["temp3d_fesom"]["levels"] == 3 and out["ua_ipsl_nan"]["levels"] == 19 and out["2t_era5"]["levels"] == 0
    assert out["2t_era5"]["cells"] == 12 * 64800 and out["ua_ipsl_nan"]["cells"] == 2 * 19 * 4050


def test_host_to_host_modes_give_the_same_bits(hip):
    import user_path_bench as u
    out = u.host_to_host(device=0, rows=48, cpu_cells_per_s=1.0e9)
    assert out["spot_check"] is True and out["rows"] == 48
    assert out["pcie_bytes_per_row"] == {"packed": (259200 + 64800) * 8, "whole_rows": (1038240 + 64800) * 8}
    for key in ("pageable_whole_rows", "pinned_packed", "pinned_whole_rows"):
        assert out[key]["same_bits"] is True and out[key]["cells_per_s"] > 0
    assert out["pageable_packed"]["cells_per_s"] > 0 and out["cpu_cells_per_s"] == 1.0e9
