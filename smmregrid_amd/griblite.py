"""Minimal GRIB reader (editions 1 and 2): grid-point fields in simple packing on lon/lat, regular Gaussian and
reduced Gaussian grids, laid out as cfgrib lays them out for xarray.

The reference's own tests read one GRIB file, `tests/data/lsm-ifs.grb` (identity2d_test.py:22-27, util_test.py:57: an
IFS land-sea mask on the octahedral reduced Gaussian grid O96), through `xarray.open_mfdataset` + cfgrib / ecCodes, and hand the
file name to `cdo` for the weights.  Neither is on this image, so the field and its grid are decoded here from the
published WMO FM 92 GRIB edition-1 layout (sections 0-5; ECMWF local table 128 for the variable names):

    section 0  'GRIB', total length (3 octets), edition
    section 1  product definition: parameter, level, reference time, decimal scale factor D
    section 2  grid description: representation type (0 lon/lat, 4 Gaussian), Ni, Nj, first / last point, N,
               scanning mode, and for reduced grids the list PL of points per row
    section 3  optional bitmap of the points that carry a value
    section 4  binary data: binary scale factor E, reference value R (IBM hexadecimal float), bits per value, X
               value = (R + X * 2^E) / 10^D

Conventions of cfgrib kept: variable names (`lsm`, `t2m`, ...), dimensions (`latitude`, `longitude`) for rectangular
grids and (`values`,) with `latitude(values)` / `longitude(values)` for reduced ones, `time` / level dimensions only
when a file holds more than one of them, missing points as NaN, float32 fields.

Edition 2 (WMO FM 92 GRIB edition 2; what newer IFS / ICON output comes in) is decoded for the same family: sections
0 - 8, grid definition templates 3.0 (lon/lat) and 3.40 (Gaussian, regular or with the list of points per row),
product definition templates 4.0 / 4.1 / 4.8 / 4.11 (the leading octets they share), data representation template 5.0
(simple packing, IEEE reference value), bitmap section, several fields per message (repeated sections 2 - 7).

Anything else in a GRIB file (spherical harmonics, second-order / JPEG / PNG / CCSDS packing, rotated or projected
grids) raises GribUnsupported naming the feature.
"""
import numpy as np

from .xrlite import DataArray, Dataset


class GribUnsupported(NotImplementedError):
    pass


# ECMWF table 128 entries that turn up in climate work -> (cfgrib variable name, long name, units)
_TABLE_128 = {
    31: ("siconc", "Sea ice area fraction", "(0 - 1)"), 34: ("sst", "Sea surface temperature", "K"),
    129: ("z", "Geopotential", "m**2 s**-2"), 130: ("t", "Temperature", "K"),
    131: ("u", "U component of wind", "m s**-1"), 132: ("v", "V component of wind", "m s**-1"),
    133: ("q", "Specific humidity", "kg kg**-1"), 134: ("sp", "Surface pressure", "Pa"),
    151: ("msl", "Mean sea level pressure", "Pa"), 164: ("tcc", "Total cloud cover", "(0 - 1)"),
    165: ("u10", "10 metre U wind component", "m s**-1"), 166: ("v10", "10 metre V wind component", "m s**-1"),
    167: ("t2m", "2 metre temperature", "K"), 168: ("d2m", "2 metre dewpoint temperature", "K"),
    172: ("lsm", "Land-sea mask", "(0 - 1)"), 228: ("tp", "Total precipitation", "m"),
}
_LEVEL_DIMS = {100: "isobaricInhPa", 109: "hybrid", 105: "heightAboveGround", 111: "depthBelowLand",
               160: "depthBelowSea"}


# GRIB-2 (discipline, category, number) -> (cfgrib variable name, long name, units); WMO code table 4.2
_TABLE_G2 = {
    (0, 0, 0): ("t", "Temperature", "K"), (0, 0, 6): ("d", "Dew point temperature", "K"),
    (0, 1, 0): ("q", "Specific humidity", "kg kg**-1"), (0, 1, 8): ("tp", "Total precipitation", "kg m**-2"),
    (0, 2, 2): ("u", "U component of wind", "m s**-1"), (0, 2, 3): ("v", "V component of wind", "m s**-1"),
    (0, 3, 0): ("sp", "Pressure", "Pa"), (0, 3, 1): ("msl", "Pressure reduced to MSL", "Pa"),
    (0, 3, 4): ("z", "Geopotential", "m**2 s**-2"), (0, 3, 5): ("gh", "Geopotential height", "gpm"),
    (0, 6, 1): ("tcc", "Total cloud cover", "%"), (2, 0, 0): ("lsm", "Land-sea mask", "(0 - 1)"),
    (10, 3, 0): ("sst", "Sea surface temperature", "K"), (10, 2, 0): ("siconc", "Sea ice area fraction", "(0 - 1)"),
}
# near-surface fields get cfgrib's own names: (name above, height above ground in m) -> name
_G2_NEAR_SURFACE = {("t", 2.0): "t2m", ("d", 2.0): "d2m", ("u", 10.0): "u10", ("v", 10.0): "v10"}
_LEVEL_DIMS_G2 = {100: "isobaricInhPa", 105: "hybrid", 103: "heightAboveGround", 106: "depthBelowLandLayer",
                  160: "depthBelowSea"}


def _uint(b):
    return int.from_bytes(b, "big")


def _sint(b):
    """GRIB-1 signed integers are sign-and-magnitude."""
    v = _uint(b)
    top = 1 << (8 * len(b) - 1)
    return -(v & (top - 1)) if v & top else v


def _ibm_float(b):
    """IBM System/360 single precision: sign, 7-bit excess-64 exponent of 16, 24-bit fraction."""
    v = _uint(b)
    sign = -1.0 if v >> 31 else 1.0
    return sign * (v & 0xFFFFFF) / float(1 << 24) * 16.0 ** (((v >> 24) & 0x7F) - 64)


def gaussian_latitudes(n):
    """The 2N Gaussian latitudes, north to south: arcsin of the roots of the Legendre polynomial of degree 2N."""
    x, _ = np.polynomial.legendre.leggauss(2 * n)
    return np.degrees(np.arcsin(x))[::-1]


def _unpack_bits(raw, nbits, count):
    """`count` unsigned big-endian integers of `nbits` bits each from the byte string `raw`."""
    if nbits == 0:
        return np.zeros(count, dtype=np.float64)
    if nbits in (8, 16, 32):
        return np.frombuffer(raw, dtype={8: ">u1", 16: ">u2", 32: ">u4"}[nbits], count=count).astype(np.float64)
    if nbits == 24:
        b = np.frombuffer(raw, dtype=np.uint8, count=3 * count).reshape(count, 3).astype(np.uint32)
        return ((b[:, 0] << 16) | (b[:, 1] << 8) | b[:, 2]).astype(np.float64)
    bits = np.unpackbits(np.frombuffer(raw, dtype=np.uint8))[:nbits * count].reshape(count, nbits)
    return bits.astype(np.uint64) @ (np.uint64(1) << np.arange(nbits - 1, -1, -1, dtype=np.uint64))


class _Field:
    """Grid and coordinates shared by the fields of both editions."""

    def _set_grid(self, rep, ni, nj, la1, lo1, la2, lo2, n_gauss, scan, pl, tol):
        if scan & 0x20:
            raise GribUnsupported("scanning mode with consecutive points along j")
        if scan & 0x10:
            raise GribUnsupported("scanning mode with alternating row direction")
        self.rep, self.nj = rep, nj
        self.pl = pl
        if pl is not None and pl.size != nj:
            raise ValueError("reduced grid: the list of points per row is shorter than Nj")
        if rep == 4:
            lat = gaussian_latitudes(n_gauss)
            if lat.size != nj:                       # a band of a Gaussian grid: rows from La1 to La2
                first = int(np.abs(lat - max(la1, la2)).argmin())
                lat = lat[first:first + nj]
            if abs(lat[0] - max(la1, la2)) > tol:
                raise ValueError(f"Gaussian latitude {lat[0]:.4f} does not match the file's first row {max(la1, la2)}")
            if scan & 0x40:
                lat = lat[::-1]
        else:
            lat = np.linspace(la1, la2, nj)
        self.lat = lat
        if pl is None:
            if lo2 < lo1:
                lo2 += 360.0
            lon = np.linspace(lo1, lo2, ni)
            self.lon = lon[::-1] if scan & 0x80 else lon
            self.ni, self.npoints = ni, ni * nj
        else:
            if scan & 0x80:
                raise GribUnsupported("reduced grid scanned east to west")
            self.lon = None
            self.ni, self.npoints = None, int(pl.sum())

    def point_coordinates(self):
        """latitude / longitude of every point of a reduced grid, in the order of the values: PL[j] points per row,
        evenly spaced from 0 east (the full circle divided by PL[j])."""
        lat = np.repeat(self.lat, self.pl)
        lon = np.concatenate([np.arange(p) * (360.0 / p) for p in self.pl])
        return lat, lon

    @property
    def grid_key(self):
        return (self.rep, self.nj, self.ni, None if self.pl is None else self.pl.tobytes(), self.lat.tobytes())


class _Message(_Field):
    """One decoded GRIB-1 message: metadata + values in the file's scanning order (NaN where the bitmap says so)."""
    edition = 1

    def __init__(self, buf, start):
        if buf[start:start + 4] != b"GRIB":
            raise ValueError("not a GRIB message")
        if buf[start + 7] != 1:
            raise GribUnsupported(f"GRIB edition {buf[start + 7]}")
        self.length = _uint(buf[start + 4:start + 7])
        if buf[start + self.length - 4:start + self.length] != b"7777":
            raise ValueError("GRIB message does not end in 7777 (truncated file?)")
        pos = start + 8
        pds = buf[pos:pos + _uint(buf[pos:pos + 3])]
        pos += len(pds)
        self.table, self.centre, self.param = pds[3], pds[4], pds[8]
        self.level_type = pds[9]
        self.level = _uint(pds[10:12]) if self.level_type in (100, 105, 109, 111, 160, 103, 107, 113, 115, 117, 119, 125) \
            else pds[10]
        century = pds[24] if len(pds) > 24 else 21
        year = (century - 1) * 100 + pds[12]
        self.time = np.datetime64(f"{year:04d}-{max(pds[13], 1):02d}-{max(pds[14], 1):02d}T{pds[15]:02d}:{pds[16]:02d}")
        unit_hours = {0: 1 / 60.0, 1: 1.0, 2: 24.0, 10: 3.0, 11: 6.0, 12: 12.0, 254: 1 / 3600.0}.get(pds[17])
        tri = pds[20]
        p1, p2 = pds[18], pds[19]
        step = {0: p1, 1: 0, 10: (p1 << 8) | p2}.get(tri, p2)
        self.step_hours = float(step) * unit_hours if unit_hours is not None else 0.0
        self.decimal_scale = _sint(pds[26:28]) if len(pds) >= 28 else 0
        has_gds, has_bms = bool(pds[7] & 0x80), bool(pds[7] & 0x40)
        if not has_gds:
            raise GribUnsupported("GRIB-1 message without a grid description section (predefined grid numbers)")
        gds = buf[pos:pos + _uint(buf[pos:pos + 3])]
        pos += len(gds)
        self._grid(gds)
        bitmap = None
        if has_bms:
            bms = buf[pos:pos + _uint(buf[pos:pos + 3])]
            pos += len(bms)
            if _uint(bms[4:6]) != 0:
                raise GribUnsupported("predefined GRIB-1 bitmaps")
            bitmap = np.unpackbits(np.frombuffer(bms[6:], dtype=np.uint8))[:self.npoints].astype(bool)
        bds = buf[pos:pos + _uint(buf[pos:pos + 3])]
        flag = bds[3]
        if flag & 0x80:
            raise GribUnsupported("spherical harmonic coefficients")
        if flag & 0x40:
            raise GribUnsupported("second-order (complex) packing")
        if flag & 0x10:
            raise GribUnsupported("GRIB-1 binary data section with additional flags (matrix of values)")
        scale = 2.0 ** _sint(bds[4:6])
        ref = _ibm_float(bds[6:10])
        nbits = bds[10]
        count = self.npoints if bitmap is None else int(bitmap.sum())
        avail = ((len(bds) - 11) * 8 - (flag & 0x0F)) // nbits if nbits else count
        if avail < count:
            raise ValueError(f"GRIB data section holds {avail} values, the grid needs {count}")
        x = _unpack_bits(bds[11:], nbits, count)
        packed = (ref + x * scale) / 10.0 ** self.decimal_scale
        if bitmap is None:
            self.values = packed
        else:
            self.values = np.full(self.npoints, np.nan)
            self.values[bitmap] = packed

    def _grid(self, gds):
        nv, pvpl, rep = gds[3], gds[4], gds[5]
        if rep not in (0, 4):
            raise GribUnsupported(f"GRIB-1 data representation type {rep} (lon/lat = 0 and Gaussian = 4 are decoded)")
        ni, nj = _uint(gds[6:8]), _uint(gds[8:10])
        la1, lo1 = _sint(gds[10:13]) / 1000.0, _sint(gds[13:16]) / 1000.0
        la2, lo2 = _sint(gds[17:20]) / 1000.0, _sint(gds[20:23]) / 1000.0
        pl = None
        if pvpl != 255 and ni == 0xFFFF:
            off = pvpl - 1 + 4 * nv
            pl = np.frombuffer(bytes(gds[off:off + 2 * nj]), dtype=">u2").astype(np.int64)
        self._set_grid(rep, ni, nj, la1, lo1, la2, lo2, _uint(gds[25:27]), gds[27], pl, tol=2e-3)


class _Field2(_Field):
    """One field of a GRIB-2 message (a message may repeat sections 2 - 7 / 3 - 7 / 4 - 7 for further fields)."""
    edition = 2

    def __init__(self, discipline, sec1, sec3, sec4, sec5, sec6, sec7, prev_bitmap):
        self.centre = _uint(sec1[5:7])
        self.time = np.datetime64(f"{_uint(sec1[12:14]):04d}-{max(sec1[14], 1):02d}-{max(sec1[15], 1):02d}"
                                  f"T{sec1[16]:02d}:{sec1[17]:02d}:{sec1[18]:02d}")
        # --- section 3: grid
        if sec3[5] != 0:
            raise GribUnsupported("GRIB-2 grid defined by reference to a predefined grid")
        n_points, n_oct, template = _uint(sec3[6:10]), sec3[10], _uint(sec3[12:14])
        if template not in (0, 40):
            raise GribUnsupported(f"GRIB-2 grid definition template 3.{template} (3.0 lon/lat and 3.40 Gaussian are decoded)")
        ni, nj = _uint(sec3[30:34]), _uint(sec3[34:38])
        basic, sub = _uint(sec3[38:42]), _uint(sec3[42:46])
        unit = 1e-6 if basic in (0, 0xFFFFFFFF) or sub in (0, 0xFFFFFFFF) else basic / float(sub)
        la1, lo1 = _sint(sec3[46:50]) * unit, _sint(sec3[50:54]) * unit
        la2, lo2 = _sint(sec3[55:59]) * unit, _sint(sec3[59:63]) * unit
        pl = None
        if n_oct:
            pl = np.array([_uint(sec3[72 + i * n_oct:72 + (i + 1) * n_oct]) for i in range(nj)], dtype=np.int64)
            ni = None
        elif ni == 0xFFFFFFFF:
            raise ValueError("GRIB-2 reduced grid without its list of points per row")
        self._set_grid(4 if template == 40 else 0, ni, nj, la1, lo1, la2, lo2, _uint(sec3[67:71]), sec3[71], pl, tol=1e-4)
        if self.npoints != n_points:
            raise ValueError(f"GRIB-2 grid of {self.npoints} points in a section that announces {n_points}")
        # --- section 4: product
        pdt = _uint(sec4[7:9])
        if pdt not in (0, 1, 8, 11):
            raise GribUnsupported(f"GRIB-2 product definition template 4.{pdt}")
        self.table = f"g2:{discipline}"
        self.param = (discipline, sec4[9], sec4[10])
        unit_hours = {0: 1 / 60.0, 1: 1.0, 2: 24.0, 10: 3.0, 11: 6.0, 12: 12.0, 13: 1 / 3600.0}.get(sec4[17])
        self.step_hours = float(_uint(sec4[18:22])) * unit_hours if unit_hours is not None else 0.0
        self.level_type = sec4[22]
        factor, scaled = sec4[23], _uint(sec4[24:28])
        level = 0.0 if scaled == 0xFFFFFFFF or factor == 0xFF else _sint(sec4[24:28]) * 10.0 ** -_sint(sec4[23:24])
        self.level = level / 100.0 if self.level_type == 100 else level          # Pa -> hPa, as cfgrib's isobaricInhPa
        # --- section 5 / 6 / 7: packed values
        n_coded, drt = _uint(sec5[5:9]), _uint(sec5[9:11])
        if drt != 0:
            names = {2: "complex packing", 3: "complex packing with spatial differencing", 40: "JPEG 2000 packing",
                     41: "PNG packing", 42: "CCSDS packing", 50: "spherical harmonics", 51: "spherical harmonics"}
            raise GribUnsupported(f"GRIB-2 data representation template 5.{drt} ({names.get(drt, 'not simple packing')})")
        ref = float(np.frombuffer(bytes(sec5[11:15]), dtype=">f4")[0])
        scale, decimal, nbits = 2.0 ** _sint(sec5[15:17]), _sint(sec5[17:19]), sec5[19]
        indicator = sec6[5] if sec6 is not None else 255
        if indicator == 0:
            self.bitmap = np.unpackbits(np.frombuffer(bytes(sec6[6:]), dtype=np.uint8))[:self.npoints].astype(bool)
        elif indicator == 254:
            if prev_bitmap is None:
                raise ValueError("GRIB-2 field refers to a previous bitmap that does not exist")
            self.bitmap = prev_bitmap
        elif indicator == 255:
            self.bitmap = None
        else:
            raise GribUnsupported("predefined GRIB-2 bitmaps")
        count = self.npoints if self.bitmap is None else int(self.bitmap.sum())
        if n_coded != count:
            raise ValueError(f"GRIB-2 data section codes {n_coded} values, grid and bitmap need {count}")
        if nbits and (len(sec7) - 5) * 8 < count * nbits:
            raise ValueError("GRIB-2 data section is shorter than its values")
        packed = (ref + _unpack_bits(bytes(sec7[5:]), nbits, count) * scale) / 10.0 ** decimal
        if self.bitmap is None:
            self.values = packed
        else:
            self.values = np.full(self.npoints, np.nan)
            self.values[self.bitmap] = packed


def _fields_of_message2(buf, start):
    """The fields of the GRIB-2 message at `start`, and the message's length."""
    discipline, total = buf[start + 6], _uint(buf[start + 8:start + 16])
    if buf[start + total - 4:start + total] != b"7777":
        raise ValueError("GRIB message does not end in 7777 (truncated file?)")
    pos, end = start + 16, start + total - 4
    sec = {}
    fields, bitmap = [], None
    while pos < end:
        length, number = _uint(buf[pos:pos + 4]), buf[pos + 4]
        if length < 5 or pos + length > end:
            raise ValueError("GRIB-2 section runs past the end of its message")
        sec[number] = buf[pos:pos + length]
        if number == 3:                       # a new grid: sections 4 - 7 follow again
            sec.pop(6, None)
        if number == 7:
            for need in (1, 3, 4, 5):
                if need not in sec:
                    raise ValueError(f"GRIB-2 message without section {need}")
            f = _Field2(discipline, sec[1], sec[3], sec[4], sec[5], sec.get(6), sec[7], bitmap)
            bitmap = f.bitmap
            fields.append(f)
        pos += length
    return fields, total


def read_messages(path):
    with open(path, "rb") as f:
        buf = f.read()
    out, pos = [], 0
    while True:
        pos = buf.find(b"GRIB", pos)
        if pos < 0:
            break
        edition = buf[pos + 7] if pos + 8 <= len(buf) else 0
        if edition == 2:
            fields, length = _fields_of_message2(buf, pos)
            out.extend(fields)
            pos += length
            continue
        if edition != 1:
            raise GribUnsupported(f"GRIB edition {edition}")
        m = _Message(buf, pos)
        out.append(m)
        pos += m.length
    if not out:
        raise ValueError(f"{path} holds no GRIB message")
    return out


def open_grib(path):
    """Dataset of the fields of a GRIB file, one variable per parameter.  All messages must share one grid."""
    msgs = read_messages(path)
    if len({m.grid_key for m in msgs}) != 1:
        raise GribUnsupported("GRIB file with fields on several grids")
    g = msgs[0]
    ds = Dataset(attrs={"GRIB_edition": int(g.edition), "GRIB_centre": {98: "ecmf"}.get(g.centre, str(g.centre)),
                        "Conventions": "CF-1.7", "institution": "European Centre for Medium-Range Weather Forecasts"
                        if g.centre == 98 else str(g.centre)})
    if g.pl is None:
        hdims, hshape = ("latitude", "longitude"), (g.nj, g.ni)
        coords = {"latitude": DataArray(g.lat, dims=("latitude",), attrs={"units": "degrees_north",
                                                                         "standard_name": "latitude"}),
                  "longitude": DataArray(g.lon, dims=("longitude",), attrs={"units": "degrees_east",
                                                                           "standard_name": "longitude"})}
        grid_type = "regular_ll" if g.rep == 0 else "regular_gg"
    else:
        lat, lon = g.point_coordinates()
        hdims, hshape = ("values",), (g.npoints,)
        coords = {"latitude": DataArray(lat, dims=("values",), attrs={"units": "degrees_north",
                                                                     "standard_name": "latitude"}),
                  "longitude": DataArray(lon, dims=("values",), attrs={"units": "degrees_east",
                                                                      "standard_name": "longitude"})}
        grid_type = "reduced_gg" if g.rep == 4 else "reduced_ll"
    by_param = {}
    for m in msgs:
        by_param.setdefault((m.table, m.param, m.level_type), []).append(m)
    for (table, param, level_type), group in by_param.items():
        if isinstance(param, tuple):             # edition 2: (discipline, category, number)
            tag = "p" + "_".join(str(v) for v in param)
            name, long_name, units = _TABLE_G2.get(param, (tag, f"parameter {param}", "unknown"))
            heights = {m.level for m in group}
            if level_type == 103 and len(heights) == 1:
                name = _G2_NEAR_SURFACE.get((name, float(next(iter(heights)))), name)
            param_id = param[0] * 1000000 + param[1] * 1000 + param[2]
        else:
            name, long_name, units = _TABLE_128.get(param, (f"p{param}", f"parameter {param}", "unknown")) \
                if table == 128 else (f"p{param}", f"parameter {param} of table {table}", "unknown")
            param_id = int(param)
        if name in ds.data_vars:
            name = f"{name}_{level_type}"
        times = sorted({m.time + np.timedelta64(int(round(m.step_hours * 3600)), "s") for m in group})
        levels = sorted({m.level for m in group})
        arr = np.full((len(times), len(levels)) + hshape, np.nan, dtype=np.float32)
        for m in group:
            t = times.index(m.time + np.timedelta64(int(round(m.step_hours * 3600)), "s"))
            arr[t, levels.index(m.level)] = m.values.reshape(hshape)
        dims, vcoords = [], dict(coords)
        level_dim = (_LEVEL_DIMS_G2 if isinstance(param, tuple) else _LEVEL_DIMS).get(level_type, "level")
        if len(times) > 1:
            dims.append("time")
            vcoords["time"] = DataArray(np.array(times, dtype="datetime64[s]").astype(np.float64), dims=("time",),
                                        attrs={"units": "seconds since 1970-01-01", "standard_name": "time"})
        else:
            arr = arr[0:1]
        if len(levels) > 1:
            dims.append(level_dim)
            vcoords[level_dim] = DataArray(np.array(levels, dtype=np.float64), dims=(level_dim,))
        arr = arr.reshape(tuple(n for n, keep in ((len(times), len(times) > 1), (len(levels), len(levels) > 1)) if keep)
                          + hshape)
        ds[name] = DataArray(arr, dims=tuple(dims) + hdims, coords=vcoords, name=name,
                             attrs={"long_name": long_name, "units": units, "GRIB_paramId": param_id,
                                    "GRIB_gridType": grid_type, "GRIB_shortName": name})
    for k, c in coords.items():
        ds.coords[k] = c
    return ds
