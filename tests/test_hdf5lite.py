"""The built-in HDF5 / NetCDF-4 reader (smmregrid_amd/hdf5lite.py) against what h5py
read from the same files (tests/golden/make_hdf5_fixtures.py, run once with an h5py interpreter):
four data files of the reference's own test suite plus synthetic files covering the other format
branches (v0 superblock / v1 headers / symbol-table groups; v3 superblock / v4 chunk indexes;
dense links and attributes; big-endian, integer, compact, fletcher32, unwritten chunks)."""
import glob
import json
import os

import numpy as np
import pytest

from smmregrid_amd import hdf5lite, io

REFDATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refdata")
FILES = sorted(f for f in glob.glob(os.path.join(REFDATA, "*")) if not f.endswith(".npz"))


def test_fixture_set_is_complete():
    names = {os.path.basename(f) for f in FILES}
    assert {"2t-era5.nc", "healpix_0.nc", "regional.nc", "r360x180.nc", "synthetic_earliest.h5",
            "synthetic_latest.h5"} <= names


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_values_attributes_and_dims_match_h5py(path):
    exp = np.load(path + ".expected.npz", allow_pickle=False)
    meta = json.loads(str(exp["__meta__"]))
    with hdf5lite.File(path) as f:
        by_addr = {f.root[k].addr: k for k in f.keys()}
        n = 0
        for key in exp.files:
            if key == "__meta__":
                continue
            name = key.replace("|", "/")
            node = f[name]
            got = node.read()
            want = exp[key]
            assert got.shape == want.shape and got.dtype.kind == want.dtype.kind and got.dtype.itemsize == want.dtype.itemsize
            assert np.array_equal(got, want, equal_nan=(want.dtype.kind == "f")), name
            m = meta[name]
            for a, v in m.items():
                if a == "__dims__":
                    refs = node.attrs["DIMENSION_LIST"]
                    assert [by_addr.get(int(r[0])) if len(r) else None for r in refs] == v
                    continue
                lite = node.attrs[a]
                if isinstance(v, str):
                    assert io._attr_value(lite) == v, (name, a)
                elif v is None:
                    assert lite is None
                elif isinstance(v, list) and v and isinstance(v[0], str):
                    assert [io._attr_value(x) for x in np.atleast_1d(lite)] == v
                else:
                    assert np.allclose(np.asarray(lite, dtype=float).ravel(), np.asarray(v, dtype=float), equal_nan=True), (name, a)
            n += 1
        assert n >= 4
        for a, v in meta["/"].items():
            if isinstance(v, str):
                assert io._attr_value(f.attrs[a]) == v


def test_netcdf4_file_opens_like_the_h5py_path():
    """2t-era5.nc through io.open_dataset == the fixture made earlier with h5py (2t_era5.npz)."""
    ds = io._open_netcdf4_lite(os.path.join(REFDATA, "2t-era5.nc"))
    ref = np.load(os.path.join(os.path.dirname(REFDATA), "2t_era5.npz"))
    assert ds["2t"].dims == ("time", "lat", "lon") and ds["time_bnds"].dims == ("time", "bnds")
    assert set(ds.coords) == {"time", "lat", "lon"} and "bnds" not in ds.variables
    assert np.array_equal(ds["2t"].values, ref["t2m"]) and ds["2t"].values.dtype == np.float32
    for k in ("lat", "lon", "time"):
        assert np.array_equal(ds.coords[k].values, ref[k])
    assert ds.coords["lat"].attrs["units"] == "degrees_north" and ds.attrs["Conventions"] == "CF-1.6"
    assert ds["2t"].attrs["code"] == 167                       # 1-element array attribute -> scalar


def test_healpix_and_levels_file():
    ds = io._open_netcdf4_lite(os.path.join(REFDATA, "healpix_0.nc"))
    assert ds["ta"].dims == ("time", "level_full", "x") and ds["ta"].values.shape == (2, 90, 12)
    assert ds["tas"].dims == ("time", "x")


def test_cf_decoding_of_packed_data():
    path = os.path.join(REFDATA, "synthetic_latest.h5")
    raw = io._open_netcdf4_lite(path, decode=False)["packed"]
    dec = io._open_netcdf4_lite(path, decode=True)["packed"]
    assert raw.values.dtype == np.int16 and dec.values.dtype == np.float32
    want = raw.values.astype(np.float32) * np.float32(0.01) + np.float32(273.15)
    assert np.array_equal(dec.values, want)
    assert "scale_factor" not in dec.attrs and "attr_07" in dec.attrs and dec.attrs["attr_07"] == 10.5


def test_unwritten_chunks_take_the_fill_value():
    with hdf5lite.File(os.path.join(REFDATA, "synthetic_earliest.h5")) as f:
        holes = f["holes"].read()
        assert np.all(holes[:5, :5] == 1.5) and np.all(holes[5:, :] == -999.0) and np.all(holes[:5, 5:] == -999.0)
        assert f["grp/inner"].read().tolist() == list(range(6))
        assert f["scalar"].read().shape == () and float(f["scalar"].read()) == 3.25
        assert f["contig_be"].read().dtype == np.dtype("float64")          # big-endian on disk, native in memory
        assert len([k for k in f.keys() if k.startswith("many_")]) == 30
        with pytest.raises(KeyError):
            f["nope"]


def test_open_weights_dispatches_on_magic(tmp_path):
    ds = io.open_weights(os.path.join(REFDATA, "regional.nc"))
    assert ds["pr"].dims == ("time", "lat", "lon") and ds["pr"].values.shape == (1, 90, 61)
    assert io.open_dataset is io.open_weights


def test_bad_files_fail_loudly(tmp_path):
    p = tmp_path / "not_hdf5.nc"
    p.write_bytes(b"GARBAGE" * 100)
    with pytest.raises(hdf5lite.H5Error, match="not an HDF5"):
        hdf5lite.File(str(p))
    good = open(os.path.join(REFDATA, "regional.nc"), "rb").read()
    q = tmp_path / "truncated.nc"
    q.write_bytes(good[:3000])
    with pytest.raises(hdf5lite.H5Error):
        with hdf5lite.File(str(q)) as f:
            for k in f.keys():
                node = f[k]
                if isinstance(node, hdf5lite.DatasetNode):
                    node.read()
    e = tmp_path / "empty.nc"
    e.write_bytes(b"")
    with pytest.raises(hdf5lite.H5Error):
        hdf5lite.File(str(e))


def test_h5py_reader_and_builtin_reader_decode_alike():
    """io.open_weights prefers h5py when it is importable: both NetCDF-4 readers must hand the same
    decoded field (CF _FillValue / missing_value -> NaN, scale_factor / add_offset) to the regridder.
    h5py lives only in the image's second interpreter, so the comparison runs there."""
    import shutil
    import subprocess
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py) or subprocess.run([py, "-c", "import h5py, numpy"], capture_output=True).returncode:
        pytest.skip("no interpreter with h5py on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, glob, numpy as np, h5py\n"
        "sys.path.insert(0, %r)\n"
        "from smmregrid_amd import io\n"
        "n = 0\n"
        "for f in sorted(glob.glob(%r)):\n"
        "    a, b = io._open_netcdf4_h5py(h5py, f), io._open_netcdf4_lite(f)\n"
        "    assert sorted(a._vars) == sorted(b._vars), f\n"
        "    for k in a._vars:\n"
        "        x, y = np.asarray(a._vars[k].values), np.asarray(b._vars[k].values)\n"
        "        assert x.dtype == y.dtype and x.shape == y.shape, (f, k, x.dtype, y.dtype)\n"
        "        assert np.array_equal(x, y, equal_nan=(x.dtype.kind == 'f')), (f, k)\n"
        "        assert a._vars[k].dims == b._vars[k].dims and set(a._vars[k].attrs) == set(b._vars[k].attrs), (f, k)\n"
        "        assert not ({'_FillValue', 'missing_value', 'scale_factor', 'add_offset'} & set(a._vars[k].attrs))\n"
        "        n += 1\n"
        "print('compared', n)\n" % (root, os.path.join(root, "tests", "golden", "refdata", "*.nc")))
    out = subprocess.run([py, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("compared") and int(out.stdout.split()[1]) >= 4
