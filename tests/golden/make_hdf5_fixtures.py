"""HDF5 / NetCDF-4 fixtures for smmregrid_amd/hdf5lite.py (data only, no code of the reference):

    /opt/conda/bin/python3.9 tests/golden/make_hdf5_fixtures.py

1. Cross-checks hdf5lite against h5py on EVERY HDF5 file of the reference's tests/data (values of
   all datasets, numeric and text attributes, dimension names) and prints one line per file.
2. Copies the four small data files the reference's tests use (2t-era5.nc, healpix_0.nc,
   regional.nc, r360x180.nc) to tests/golden/refdata/ and writes what h5py reads from them
   (raw, undecoded values) to tests/golden/refdata/<name>.expected.npz.
3. Writes synthetic HDF5 files with h5py that cover format branches those files do not (old-style
   groups / v1 object headers, contiguous + compact + big-endian + integer data, fletcher32,
   missing chunks with a fill value, scale/offset packing) to tests/golden/refdata/synthetic_*.h5.
"""
import glob
import json
import os
import shutil
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from smmregrid_amd import hdf5lite  # noqa: E402

OUT = os.path.join(HERE, "refdata")
REF = "/root/reference/tests/data"
SMALL = ["2t-era5.nc", "healpix_0.nc", "regional.nc", "r360x180.nc"]


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype.kind == "f":
        return np.array_equal(a, b.astype(a.dtype), equal_nan=True)
    return np.array_equal(a, b)


def norm_attr(v):
    if isinstance(v, h5py.Empty):
        return None                                   # null dataspace: hdf5lite gives None
    if isinstance(v, bytes):
        return v.decode()
    if isinstance(v, str):
        return v
    v = np.asarray(v)
    if v.dtype.kind in "SO":
        return [x.decode() if isinstance(x, bytes) else str(x) for x in v.ravel()]
    return v.ravel().tolist()


def norm_lite(v):
    if v is None:
        return None
    if isinstance(v, np.ndarray) and v.dtype.kind == "S":
        return [bytes(x).decode() for x in v.ravel()] if v.ndim else bytes(v.ravel()[0]).decode()
    if isinstance(v, np.ndarray) and v.dtype.kind == "O":
        return [str(x) for x in v.ravel()] if v.ndim else str(v.ravel()[0])
    return np.asarray(v).ravel().tolist()


def crosscheck(path):
    bad = []
    n = 0
    with h5py.File(path, "r") as h, hdf5lite.File(path) as l:
        def visit(hg, lg, prefix):
            nonlocal n
            assert sorted(hg.keys()) == sorted(lg.keys()), (prefix, sorted(hg.keys()), sorted(lg.keys()))
            for k in hg.keys():
                hv, lv = hg[k], lg[k]
                if isinstance(hv, h5py.Group):
                    visit(hv, lv, prefix + k + "/")
                    continue
                n += 1
                if hv.dtype.kind in "iuf":
                    if hv.shape is not None and not same(hv[...], lv.read()):
                        bad.append(prefix + k)
                for a in hv.attrs:
                    if a in ("REFERENCE_LIST",):
                        continue
                    if a == "DIMENSION_LIST":
                        names = [hv.dims[i][0].name.split("/")[-1] if len(hv.dims[i]) else None for i in range(hv.ndim)]
                        refs = lv.attrs[a]
                        by_addr = {lg[q].addr: q for q in lg.keys()}
                        got = [by_addr.get(int(r[0])) if len(r) else None for r in refs]
                        if names != got:
                            bad.append(prefix + k + "@dims")
                        continue
                    x, y = norm_attr(hv.attrs[a]), norm_lite(lv.attrs[a])
                    if isinstance(x, str) and isinstance(y, list) and len(y) == 1:
                        y = y[0]
                    if x != y and not (isinstance(x, list) and np.allclose(x, y, equal_nan=True)):
                        bad.append(prefix + k + "@" + a)
        visit(h, l.root, "/")
        for a in h.attrs:
            x, y = norm_attr(h.attrs[a]), norm_lite(l.attrs[a])
            if isinstance(x, str) and isinstance(y, list) and len(y) == 1:
                y = y[0]
            if x != y:
                bad.append("/@" + a)
    return n, bad


def synthetic():
    rng = np.random.default_rng(7)
    # (a) earliest format: v0 superblock, v1 object headers, symbol-table groups
    p = os.path.join(OUT, "synthetic_earliest.h5")
    with h5py.File(p, "w", libver="earliest") as f:
        f.create_dataset("contig_be", data=rng.standard_normal((5, 7)).astype(">f8"))
        f.create_dataset("ints", data=np.arange(-20, 20, dtype="<i2").reshape(4, 10))
        f.create_dataset("u8", data=np.arange(12, dtype="u1"))
        d = f.create_dataset("chunked", data=rng.standard_normal((33, 20)).astype("f4"), chunks=(8, 8),
                             compression="gzip", shuffle=True, fletcher32=True)
        d.attrs["units"] = np.bytes_("K")
        d.attrs["scale_factor"] = np.float64(0.5)
        h = f.create_dataset("holes", shape=(10, 10), dtype="f4", chunks=(5, 5), fillvalue=-999.0)
        h[0:5, 0:5] = 1.5                                 # three of four chunks never written
        f.create_dataset("scalar", data=np.float64(3.25))
        g = f.create_group("grp")
        g.create_dataset("inner", data=np.arange(6, dtype="<i8"))
        for i in range(30):                               # enough links for a multi-entry symbol node
            f.create_dataset(f"many_{i:02d}", data=np.full(3, i, dtype="<i4"))
        f.attrs["title"] = "synthetic earliest"
        f.attrs["vlen_text"] = "variable-length string"
    # (b) latest format: v3 superblock, v2 headers, v4 layouts (single chunk, implicit, fixed array)
    p = os.path.join(OUT, "synthetic_latest.h5")
    with h5py.File(p, "w", libver="latest") as f:
        f.create_dataset("single", data=rng.standard_normal((6, 6)), chunks=(6, 6), compression="gzip")
        f.create_dataset("fixed_array", data=rng.standard_normal((40, 12)).astype("f4"), chunks=(8, 4),
                         compression="gzip", shuffle=True)
        d = f.create_dataset("implicit", shape=(12, 8), dtype="<i4", chunks=(4, 4))
        d.id  # allocation at creation keeps the index implicit only with early allocation; data below
        d[...] = np.arange(96, dtype="<i4").reshape(12, 8)
        f.create_dataset("compact", data=np.arange(5, dtype="<f4"))
        v = f.create_dataset("packed", data=(rng.integers(-100, 100, (4, 9))).astype("<i2"))
        v.attrs["scale_factor"] = np.float32(0.01)
        v.attrs["add_offset"] = np.float32(273.15)
        v.attrs["_FillValue"] = np.int16(-32768)
        for i in range(40):                               # dense attribute storage (fractal heap + v2 B-tree)
            v.attrs[f"attr_{i:02d}"] = np.float64(i) * 1.5
        for i in range(20):                               # dense link storage in the root group
            f.create_dataset(f"link_{i:02d}", data=np.full(2, i, dtype="<u2"))
    return [os.path.join(OUT, "synthetic_earliest.h5"), os.path.join(OUT, "synthetic_latest.h5")]


def dump_expected(path):
    out = {}
    meta = {}
    with h5py.File(path, "r") as h:
        def visit(g, prefix):
            for k, v in g.items():
                if isinstance(v, h5py.Group):
                    visit(v, prefix + k + "/")
                elif v.dtype.kind in "iuf" and v.shape is not None:
                    out[(prefix + k).replace("/", "|")] = np.asarray(v[...])
                    meta[prefix + k] = {a: norm_attr(v.attrs[a]) for a in v.attrs
                                        if a not in ("DIMENSION_LIST", "REFERENCE_LIST")}
                    if "DIMENSION_LIST" in v.attrs:
                        meta[prefix + k]["__dims__"] = [v.dims[i][0].name.split("/")[-1] if len(v.dims[i]) else None
                                                        for i in range(v.ndim)]
        visit(h, "")
        meta["/"] = {a: norm_attr(h.attrs[a]) for a in h.attrs}
    out["__meta__"] = np.array(json.dumps(meta))
    np.savez_compressed(path + ".expected.npz", **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for f in sorted(glob.glob(os.path.join(REF, "*.nc"))):
        if open(f, "rb").read(4) != b"\x89HDF":
            continue
        n, bad = crosscheck(f)
        print(os.path.basename(f), "datasets", n, "mismatches", bad)
    for name in SMALL:
        shutil.copy(os.path.join(REF, name), os.path.join(OUT, name))
        os.chmod(os.path.join(OUT, name), 0o644)
        dump_expected(os.path.join(OUT, name))
    for p in synthetic():
        n, bad = crosscheck(p)
        print(os.path.basename(p), "datasets", n, "mismatches", bad)
        dump_expected(p)
