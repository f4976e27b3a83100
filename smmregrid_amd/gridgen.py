"""Native generation of SCRIP-format remap weights for structured grids.

The reference obtains weights by shelling out to ``cdo gen<method>``
(cdogenerate.py:234-303).  No ``cdo`` binary exists where this package is
built or benchmarked, so the weights the hot path consumes are produced here
with the same file layout CDO writes (variables read at weights.py:31-34,
regrid.py:500-515, cdogenerate.py:320-343): 1-based ``src_address`` /
``dst_address``, ``remap_matrix[num_links, num_wgts]``, grid sizes, dims,
masks, ``dst_grid_frac``, centre coordinates in radians.

Supported: global regular lon/lat grids ``r<NX>x<NY>`` / ``global_<res>``, regular
Gaussian grids ``F<N>`` / ``n<N>`` (CDO naming, cdogrid.py:11-22), explicit regular grids, HEALPix targets ``hp<NSIDE>[_nested|_ring]``
(cdogrid.py:18-19); methods ``bil`` (4-point bilinear from a regular or a HEALPix source), ``nn`` (nearest
neighbour) and ``con`` (first-order conservative, `fracarea` / `destarea` normalisation: exact overlap
areas regular <-> regular; regular <-> HEALPix through the pixels' equal-area nested sub-pixels).
"""
import re

import numpy as np

from .xrlite import DataArray, Dataset

DEG = np.pi / 180.0


# --------------------------------------------------------------------------- grids

class Grid:
    """Horizontal grid description.

    kind 'regular': 1-D ``lon``/``lat`` centres (degrees), cell bounds
    ``lon_b``/``lat_b`` (nx+1 / ny+1), cell index = j * nx + i (lon fastest,
    the CDO/SCRIP ordering).  kind 'points': 1-D cell list with centres only.
    """

    def __init__(self, kind, lon, lat, lon_b=None, lat_b=None, name=None, cdo_type="lonlat"):
        self.kind = kind
        self.lon = np.asarray(lon, dtype=np.float64)
        self.lat = np.asarray(lat, dtype=np.float64)
        self.lon_b = None if lon_b is None else np.asarray(lon_b, dtype=np.float64)
        self.lat_b = None if lat_b is None else np.asarray(lat_b, dtype=np.float64)
        self.name = name
        self.cdo_type = cdo_type
        # regular grids are held south-to-north; a file that stores latitudes north-to-south (ERA5
        # and friends) sets this flag and generate_weights renumbers cells to the file's order
        self.lat_descending = False
        # HEALPix grids made by parse_grid know their resolution and pixel order (conservative weights need them)
        self.nside = None
        self.nested = None
        self.shape2d = None        # (nx, ny) of a curvilinear grid given by 2-D centre coordinates
        self.vertices = None       # (lon_v, lat_v), each (cells, V): cell polygons of an unstructured / curvilinear grid

    @property
    def dims(self):
        """SCRIP ``*_grid_dims`` (fastest first)."""
        if self.kind == "regular":
            return np.array([self.lon.size, self.lat.size], dtype=np.int32)
        if getattr(self, "shape2d", None):                       # curvilinear centres: (nx, ny) of the 2-D coordinates
            return np.array(self.shape2d, dtype=np.int32)
        return np.array([self.lon.size], dtype=np.int32)

    @property
    def size(self):
        return int(np.prod(self.dims))

    def centers(self):
        """(lon, lat) of every cell in address order, degrees."""
        if self.kind == "regular":
            lon2, lat2 = np.meshgrid(self.lon, self.lat)
            return lon2.ravel(), lat2.ravel()
        return self.lon, self.lat


def regular_grid(nx, ny, name=None):
    """CDO global regular grid r<NX>x<NY>: lon_i = i*360/NX, cell-centred lats
    from -90+90/NY northwards."""
    dlon, dlat = 360.0 / nx, 180.0 / ny
    lon = np.arange(nx) * dlon
    lat = -90.0 + dlat * (np.arange(ny) + 0.5)
    lon_b = (np.arange(nx + 1) - 0.5) * dlon
    lat_b = -90.0 + dlat * np.arange(ny + 1)
    return Grid("regular", lon, lat, lon_b, lat_b, name=name or f"r{nx}x{ny}")


def regular_grid_from_centers(lon, lat, name=None, lon_b=None, lat_b=None):
    """Regular grid from 1-D centre coordinates.  Cell bounds: the given edges (`lon_b` nx + 1, `lat_b` ny + 1, in
    the order of the centres -- what a file's lon_bnds / lat_bnds hold, e.g. the true Gaussian cell edges), else
    mid-points with the poles clipped (what CDO generates for a file without bounds)."""
    lon = np.asarray(lon, dtype=np.float64)
    lat = np.asarray(lat, dtype=np.float64)

    def bounds(c, lo=None, hi=None):
        b = np.empty(c.size + 1)
        b[1:-1] = 0.5 * (c[1:] + c[:-1])
        b[0] = c[0] - 0.5 * (c[1] - c[0])
        b[-1] = c[-1] + 0.5 * (c[-1] - c[-2])
        if lo is not None:
            b = np.clip(b, lo, hi)
        return b

    descending = lat.size > 1 and lat[0] > lat[-1]
    if descending:
        lat = lat[::-1]
        if lat_b is not None:
            lat_b = np.asarray(lat_b, dtype=np.float64)[::-1]
    lon_edges = bounds(lon) if lon_b is None else np.asarray(lon_b, dtype=np.float64)
    lat_edges = bounds(lat, -90.0, 90.0) if lat_b is None else np.clip(np.asarray(lat_b, dtype=np.float64), -90.0, 90.0)
    if lon_edges.size != lon.size + 1 or lat_edges.size != lat.size + 1 or np.any(np.diff(lat_edges) <= 0) \
            or np.any(np.diff(lon_edges) <= 0):
        raise ValueError("cell bounds must be nx + 1 / ny + 1 strictly increasing edges")
    g = Grid("regular", lon, lat, lon_edges, lat_edges, name=name or "lonlat")
    g.lat_descending = bool(descending)
    return g


def healpix_centers(nside, nested=True):
    """Pixel-centre (lon, lat) in degrees of all 12*nside^2 HEALPix pixels."""
    npix = 12 * nside * nside
    pix = np.arange(npix, dtype=np.int64)
    if nested:
        # nested -> (face, ix, iy) by bit de-interleaving, then to ring coordinates
        npface = nside * nside
        face = pix // npface
        p = pix % npface

        def compact(v):
            v = v & 0x5555555555555555
            v = (v | (v >> 1)) & 0x3333333333333333
            v = (v | (v >> 2)) & 0x0F0F0F0F0F0F0F0F
            v = (v | (v >> 4)) & 0x00FF00FF00FF00FF
            v = (v | (v >> 8)) & 0x0000FFFF0000FFFF
            v = (v | (v >> 16)) & 0x00000000FFFFFFFF
            return v

        ix = compact(p)
        iy = compact(p >> 1)
        jrll = np.array([2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4], dtype=np.int64)
        jpll = np.array([1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7], dtype=np.int64)
        jr = jrll[face] * nside - ix - iy - 1
        nr = np.where(jr < nside, jr, np.where(jr > 3 * nside, 4 * nside - jr, nside))
        z = np.where(jr < nside, 1.0 - nr * nr / (3.0 * nside * nside),
                     np.where(jr > 3 * nside, -1.0 + nr * nr / (3.0 * nside * nside),
                              (2 * nside - jr) * 2.0 / (3.0 * nside)))
        kshift = np.where((jr < nside) | (jr > 3 * nside), 0, (jr - nside) & 1)
        jp = (jpll[face] * nr + ix - iy + 1 + kshift) // 2
        jp = np.where(jp > 4 * nr, jp - 4 * nr, jp)
        jp = np.where(jp < 1, jp + 4 * nr, jp)
        phi = (jp - (kshift + 1) * 0.5) * (np.pi / 2.0) / nr
    else:
        ncap = 2 * nside * (nside - 1)
        z = np.empty(npix)
        phi = np.empty(npix)
        north = pix < ncap
        ph = (pix[north] + 1) / 2.0
        iring = np.floor(np.sqrt(ph - np.sqrt(np.floor(ph)))).astype(np.int64) + 1
        iphi = pix[north] + 1 - 2 * iring * (iring - 1)
        z[north] = 1.0 - iring * iring / (3.0 * nside * nside)
        phi[north] = (iphi - 0.5) * np.pi / (2.0 * iring)
        eq = (pix >= ncap) & (pix < npix - ncap)
        ip = pix[eq] - ncap
        iring = ip // (4 * nside) + nside
        iphi = ip % (4 * nside) + 1
        fodd = 0.5 * (1 + ((iring + nside) & 1))
        z[eq] = (2 * nside - iring) * 2.0 / (3.0 * nside)
        phi[eq] = (iphi - fodd) * np.pi / (2.0 * nside)
        south = pix >= npix - ncap
        ip = npix - pix[south]
        ph = ip / 2.0
        iring = np.floor(np.sqrt(ph - np.sqrt(np.floor(ph)))).astype(np.int64) + 1
        iphi = 4 * iring + 1 - (ip - 2 * iring * (iring - 1))
        z[south] = -1.0 + iring * iring / (3.0 * nside * nside)
        phi[south] = (iphi - 0.5) * np.pi / (2.0 * iring)
    lat = np.degrees(np.arcsin(np.clip(z, -1.0, 1.0)))
    lon = np.degrees(phi) % 360.0
    return lon, lat


def healpix_grid_of_centers(lon, lat, tol=1e-6):
    """The HEALPix grid (nested or ring) whose pixel centres the given cell-centre list is, or None.  A file that
    holds a HEALPix field with explicit coordinates (cdo setgrid,hp<N>; tests/data/tas-healpix2.nc of the
    reference) is recognised by them, which opens the methods that need the pixels themselves (con, bil)."""
    lon = np.asarray(lon, dtype=np.float64).ravel()
    lat = np.asarray(lat, dtype=np.float64).ravel()
    n = lon.size
    if n < 12 or n % 12 or lat.size != n:
        return None
    nside = int(round(np.sqrt(n / 12.0)))
    if 12 * nside * nside != n or nside & (nside - 1):
        return None
    for nested in (True, False):
        hlon, hlat = healpix_centers(nside, nested=nested)
        dlon = np.abs(((hlon - lon + 180.0) % 360.0) - 180.0)
        pole = np.abs(np.abs(hlat) - 90.0) < 1e-9
        if np.all(np.abs(hlat - lat) <= tol) and np.all((dlon <= tol) | pole):
            g = Grid("points", hlon, hlat, name=f"hp{nside}_{'nested' if nested else 'ring'}", cdo_type="healpix")
            g.nside, g.nested = nside, nested
            return g
    return None


def reduced_grid_vertices(lon, lat, tol=1e-6):
    """Cell polygons (cells, V) of a REDUCED grid given as a list of centres: rows of constant latitude, each with its
    own number of evenly spaced points around the full circle (reduced / octahedral Gaussian grids of IFS, as GRIB
    files carry them).  A cell spans +-180 / n_row degrees around its centre and the latitude band of its row; the band
    edges are the Gaussian ones (sin(edge) = 1 - cumulative Gaussian weights) when the rows are the Gaussian latitudes,
    else the mid-points between rows, with the poles closing the first and last band.  None if the list is no such
    grid."""
    lon, lat = np.asarray(lon, dtype=np.float64), np.asarray(lat, dtype=np.float64)
    if lon.ndim != 1 or lon.size != lat.size or lon.size < 8:
        return None
    start = np.flatnonzero(np.r_[True, np.abs(np.diff(lat)) > tol])
    count = np.diff(np.r_[start, lat.size])
    rows = lat[start]
    if rows.size < 2 or rows.size == lat.size or not (np.all(np.diff(rows) < 0) or np.all(np.diff(rows) > 0)):
        return None
    step = 360.0 / np.repeat(count, count)
    k = np.arange(lat.size) - np.repeat(start, count)
    if np.abs(((lon - np.repeat(lon[start], count) - k * step + 180.0) % 360.0) - 180.0).max() > 1e-4:
        return None
    nrow = rows.size
    edges = None
    if nrow % 2 == 0:
        x, w = np.polynomial.legendre.leggauss(nrow)
        glat = np.degrees(np.arcsin(x))[::-1]                                   # north to south
        north_first = rows[0] > rows[-1]
        if np.abs((rows if north_first else rows[::-1]) - glat).max() < 2e-3:
            e = np.degrees(np.arcsin(np.clip(1.0 - np.r_[0.0, np.cumsum(w[::-1])], -1.0, 1.0)))
            edges = e if north_first else e[::-1]
    if edges is None:
        mid = 0.5 * (rows[1:] + rows[:-1])
        edges = np.r_[90.0 if rows[0] > rows[-1] else -90.0, mid, -90.0 if rows[0] > rows[-1] else 90.0]
    e0, e1 = np.repeat(edges[:-1], count), np.repeat(edges[1:], count)
    south, north = np.minimum(e0, e1), np.maximum(e0, e1)
    # the polygon generator joins vertices by great circles, a cell's north and south edges are parallels: put a vertex
    # every <= 2 degrees of longitude along them (the 20 cells next to the pole of an octahedral grid are 18 degrees wide)
    q = int(np.clip(np.ceil(step.max() / 2.0), 1, 16))
    t = np.arange(q + 1) / q
    along = (lon - step / 2.0)[:, None] + step[:, None] * t[None, :]                     # west -> east
    lon_v = np.concatenate([along, along[:, ::-1]], axis=1) % 360.0
    lat_v = np.concatenate([np.repeat(south[:, None], q + 1, axis=1), np.repeat(north[:, None], q + 1, axis=1)], axis=1)
    return lon_v, lat_v


def gaussian_grid(n, name=None):
    """Regular Gaussian grid F<N> / n<N>: 4N longitudes from 0, 2N Gauss-Legendre latitudes
    (south to north here; cell bounds at mid-points, clipped at the poles)."""
    x, _ = np.polynomial.legendre.leggauss(2 * n)
    lat = np.degrees(np.arcsin(x))
    lon = np.arange(4 * n) * (360.0 / (4 * n))
    g = regular_grid_from_centers(lon, lat, name=name or f"F{n}")
    g.lat_b[0], g.lat_b[-1] = -90.0, 90.0       # the first/last cells reach the poles
    g.cdo_type = "gaussian"
    return g


_R_GRID = re.compile(r"^r(\d+)x(\d+)$")
_HP_GRID = re.compile(r"^hp(\d+)(_nested|_ring)?$")
_HPZ_GRID = re.compile(r"^hpz(\d+)$")
_GAUSS_GRID = re.compile(r"^[Fn](\d+)$")
_GLOBAL_GRID = re.compile(r"^global_(\d+(?:\.\d+)?)$")
_ZONAL_GRID = re.compile(r"^zonal_(\d+(?:\.\d+)?)$")
_POINT_GRID = re.compile(r"^lon=(-?\d+(?:\.\d+)?)/lat=(-?\d+(?:\.\d+)?)$")


def parse_grid(spec):
    """CDO-style grid name -> Grid (subset of cdogrid.py:11-22)."""
    if isinstance(spec, Grid):
        return spec
    m = _R_GRID.match(spec)
    if m:
        return regular_grid(int(m.group(1)), int(m.group(2)), name=spec)
    m = _HP_GRID.match(spec)
    if m:
        nside = int(m.group(1))
        lon, lat = healpix_centers(nside, nested=(m.group(2) != "_ring"))
        g = Grid("points", lon, lat, name=spec, cdo_type="healpix")
        g.nside, g.nested = nside, m.group(2) != "_ring"
        return g
    m = _HPZ_GRID.match(spec)
    if m:
        lon, lat = healpix_centers(2 ** int(m.group(1)), nested=True)
        g = Grid("points", lon, lat, name=spec, cdo_type="healpix")
        g.nside, g.nested = 2 ** int(m.group(1)), True
        return g
    m = _GAUSS_GRID.match(spec)
    if m:
        return gaussian_grid(int(m.group(1)), name=spec)
    m = _GLOBAL_GRID.match(spec)
    if m:
        res = float(m.group(1))
        nx, ny = int(round(360.0 / res)), int(round(180.0 / res))
        lon = -180.0 + res * (np.arange(nx) + 0.5)          # CDO global_<res>: cell-centred from -180
        lat = -90.0 + res * (np.arange(ny) + 0.5)
        return regular_grid_from_centers(lon, lat, name=spec)
    m = _ZONAL_GRID.match(spec)
    if m:                                                   # CDO zonal_<dy>: one cell around the globe per latitude band
        res = float(m.group(1))
        ny = int(round(180.0 / res))
        return regular_grid_from_centers(np.array([0.0]), -90.0 + res * (np.arange(ny) + 0.5), name=spec,
                                         lon_b=np.array([-180.0, 180.0]))
    m = _POINT_GRID.match(spec)
    if m:                                                   # CDO lon=<x>/lat=<y>: one grid point (no cell: nn / bil / dis)
        return Grid("points", np.array([float(m.group(1)) % 360.0]), np.array([float(m.group(2))]), name=spec,
                    cdo_type="lonlat")
    raise ValueError(f"grid '{spec}' is not supported by the native weight generator (r<NX>x<NY>, global_<res>, "
                     "zonal_<res>, lon=<x>/lat=<y>, F<N> / n<N>, hp<N>[_nested|_ring], hpz<zoom>; dcw: regions "
                     "and gme grids need the cdo binary)")


# --------------------------------------------------------------------------- weights

def _scrip_dataset(src, dst, src_addr, dst_addr, w, method, src_imask=None, dst_frac=None,
                   dst_area=None, norm="fracarea"):
    n_links = int(src_addr.size)
    dlon, dlat = dst.centers()
    slon, slat = src.centers()
    ds = Dataset(attrs={
        "title": "smmregrid_amd native weights",
        "normalization": norm,
        "map_method": {"bil": "Bilinear remapping", "nn": "Nearest neighbor remapping",
                       "con": "Conservative remapping", "dis": "Distance weighted avg of nearest neighbors",
                       "laf": "Largest area fraction", "bic": "Bicubic remapping"}[method],
        "conventions": "SCRIP",
        "source_grid": src.cdo_type,
        "dest_grid": dst.cdo_type,
    })
    ds["src_grid_dims"] = (("src_grid_rank",), src.dims)
    ds["dst_grid_dims"] = (("dst_grid_rank",), dst.dims)
    ds["src_grid_center_lat"] = (("src_grid_size",), slat * DEG, {"units": "radians"})
    ds["dst_grid_center_lat"] = (("dst_grid_size",), dlat * DEG, {"units": "radians"})
    ds["src_grid_center_lon"] = (("src_grid_size",), slon * DEG, {"units": "radians"})
    ds["dst_grid_center_lon"] = (("dst_grid_size",), dlon * DEG, {"units": "radians"})
    ds["src_grid_imask"] = (("src_grid_size",),
                            np.ones(src.size, np.int32) if src_imask is None
                            else np.asarray(src_imask, np.int32).ravel())
    ds["dst_grid_imask"] = (("dst_grid_size",), np.ones(dst.size, np.int32))
    if dst_area is not None:
        ds["dst_grid_area"] = (("dst_grid_size",), np.asarray(dst_area, np.float64))
    ds["dst_grid_frac"] = (("dst_grid_size",),
                           np.ones(dst.size) if dst_frac is None else np.asarray(dst_frac, np.float64))
    ds["src_address"] = (("num_links",), np.asarray(src_addr, np.int32))
    ds["dst_address"] = (("num_links",), np.asarray(dst_addr, np.int32))
    w = np.asarray(w, np.float64)
    ds["remap_matrix"] = (("num_links", "num_wgts"), w.reshape(n_links, -1) if w.ndim == 2 else w.reshape(n_links, 1))
    return ds


def _sort_links(src_addr, dst_addr, w):
    order = np.lexsort((src_addr, dst_addr))  # CDO stores links sorted by (dst, src)
    return src_addr[order], dst_addr[order], w[order]


def _unit_vectors(lon, lat):
    lam, phi = np.radians(lon), np.radians(lat)
    return np.stack([np.cos(phi) * np.cos(lam), np.cos(phi) * np.sin(lam), np.sin(phi)], axis=1)


def _is_cyclic(grid):
    """A regular grid whose longitudes close the circle (last centre + one step = first centre + 360)."""
    nx = grid.lon.size
    if nx < 2:
        return True
    step = (grid.lon[-1] - grid.lon[0]) / (nx - 1)
    return abs(grid.lon[-1] + step - grid.lon[0] - 360.0) < 1e-6 * max(1.0, abs(step)) * nx


def _lon_corners(src, lon):
    """(i0, i1, fx) of the two source columns either side of every longitude.  Global (cyclic) grids wrap and are
    taken as equally spaced, as before; a REGIONAL grid does not wrap: points beyond its first / last column take
    that column (fx clamped), the columns need not be equally spaced."""
    nx = src.lon.size
    if _is_cyclic(src):
        dlon = 360.0 / nx if nx > 1 else 360.0
        u = ((lon - src.lon[0]) % 360.0) / dlon
        i0 = np.floor(u).astype(np.int64)
        fx = u - i0
        i0 = i0 % nx
        return i0, (i0 + 1) % nx, fx
    rel = ((np.asarray(lon) - src.lon[0] + 180.0) % 360.0) - 180.0 + src.lon[0]     # nearest image of the point
    rel = np.where(rel < src.lon[0] - 180.0, rel + 360.0, rel)
    i1 = np.clip(np.searchsorted(src.lon, rel, side="right"), 1, nx - 1)
    i0 = i1 - 1
    fx = np.clip((rel - src.lon[i0]) / (src.lon[i1] - src.lon[i0]), 0.0, 1.0)
    return i0, i1, fx


def bilinear_weights(src, dst, src_mask=None, extrapolate=True):
    """4-point bilinear from a regular lon/lat source (periodic in longitude,
    clamped at the first/last latitude row) to the destination cell centres.
    Without a mask: four links per destination cell, zero weights kept (as CDO's genbil does).
    With ``src_mask`` (0 = masked source cell, e.g. land): masked corners are dropped and the
    weights of the remaining corners renormalised to sum to 1; a destination cell whose four
    corners are all masked gets no link (the apply path then yields NaN for it through
    ``dst_grid_imask``, weights.py:47-52).  CDO's genbil likewise never links a masked source cell."""
    src, dst = parse_grid(src), parse_grid(dst)
    if src.cdo_type == "healpix" and src.nside is not None:
        return _healpix_bilinear(src, dst, src_mask)
    if src.kind != "regular" and src.shape2d is not None:
        return _curvilinear_bilinear(src, dst, src_mask, extrapolate=extrapolate)
    if src.kind != "regular":
        raise ValueError("bilinear generation needs a regular, curvilinear (2-D centres) or HEALPix (hp<N>) source grid")
    nx, ny = src.lon.size, src.lat.size
    lon, lat = dst.centers()
    i0, i1, fx = _lon_corners(src, lon)
    j1 = np.clip(np.searchsorted(src.lat, lat, side="right"), 1, ny - 1)
    j0 = j1 - 1
    fy = np.clip((lat - src.lat[j0]) / (src.lat[j1] - src.lat[j0]), 0.0, 1.0)
    src4 = np.stack([j0 * nx + i0, j0 * nx + i1, j1 * nx + i0, j1 * nx + i1], axis=1)
    w4 = np.stack([(1 - fx) * (1 - fy), fx * (1 - fy), (1 - fx) * fy, fx * fy], axis=1)
    order = np.argsort(src4, axis=1, kind="stable")      # links sorted by (dst, src) as CDO stores them
    src4 = np.take_along_axis(src4, order, axis=1)
    w4 = np.take_along_axis(w4, order, axis=1)
    dst4 = np.repeat(np.arange(1, lon.size + 1, dtype=np.int32)[:, None], 4, axis=1)
    if src_mask is None:
        return _scrip_dataset(src, dst, (src4.ravel() + 1).astype(np.int32), dst4.ravel(), w4.ravel(), "bil")
    imask = (np.asarray(src_mask).ravel() != 0).astype(np.int32)
    if imask.size != src.size:
        raise ValueError(f"src_mask has {imask.size} cells, the source grid {src.size}")
    valid = imask[src4] != 0
    wv = np.where(valid, w4, 0.0)
    tot = wv.sum(axis=1)
    # valid corners that all carry weight 0 (the point sits on a masked node): share equally
    flat = (tot == 0.0) & valid.any(axis=1)
    wv[flat] = valid[flat] / valid[flat].sum(axis=1, keepdims=True)
    tot[flat] = 1.0
    with np.errstate(invalid="ignore", divide="ignore"):
        wv = wv / tot[:, None]
    return _scrip_dataset(src, dst, (src4[valid] + 1).astype(np.int32), dst4[valid], wv[valid], "bil",
                          src_imask=imask)


def bicubic_weights(src, dst, src_mask=None):
    """SCRIP bicubic from a regular lon/lat source: per corner of the enclosing source box four weights -- for the
    value and for its derivatives along i, along j and the cross derivative (cubic Hermite basis), `num_wgts` = 4 as
    CDO's genbic writes them.  The reference applies column 0 only (weights.py:33: `remap_matrix[:, 0]`), i.e. the
    value basis h00(x) h00(y) with h00(t) = 1 - 3 t^2 + 2 t^3; columns 1 - 3 are there for the file's sake.
    Masked corners (src_mask == 0) drop out and column 0 is renormalised, as for bilinear weights."""
    src, dst = parse_grid(src), parse_grid(dst)
    if src.kind != "regular":
        raise ValueError("bicubic generation needs a regular source grid")
    nx, ny = src.lon.size, src.lat.size
    lon, lat = dst.centers()
    i0, i1, fx = _lon_corners(src, lon)
    j1 = np.clip(np.searchsorted(src.lat, lat, side="right"), 1, ny - 1)
    j0 = j1 - 1
    fy = np.clip((lat - src.lat[j0]) / (src.lat[j1] - src.lat[j0]), 0.0, 1.0)

    def basis(t):       # value and slope basis of the cubic Hermite interpolant at the near / far end
        return (1 - 3 * t ** 2 + 2 * t ** 3, 3 * t ** 2 - 2 * t ** 3, t - 2 * t ** 2 + t ** 3, t ** 3 - t ** 2)

    ax0, ax1, bx0, bx1 = basis(fx)
    ay0, ay1, by0, by1 = basis(fy)
    src4 = np.stack([j0 * nx + i0, j0 * nx + i1, j1 * nx + i0, j1 * nx + i1], axis=1)
    vx = np.stack([ax0, ax1, ax0, ax1], axis=1)
    vy = np.stack([ay0, ay0, ay1, ay1], axis=1)
    sx = np.stack([bx0, bx1, bx0, bx1], axis=1)
    sy = np.stack([by0, by0, by1, by1], axis=1)
    w4 = np.stack([vx * vy, sx * vy, vx * sy, sx * sy], axis=2)           # (n, corner, 4 weights)
    order = np.argsort(src4, axis=1, kind="stable")
    src4 = np.take_along_axis(src4, order, axis=1)
    w4 = np.take_along_axis(w4, order[:, :, None], axis=1)
    dst4 = np.repeat(np.arange(1, lon.size + 1, dtype=np.int32)[:, None], 4, axis=1)
    if src_mask is None:
        return _scrip_dataset(src, dst, (src4.ravel() + 1).astype(np.int32), dst4.ravel(), w4.reshape(-1, 4), "bic")
    imask = (np.asarray(src_mask).ravel() != 0).astype(np.int32)
    if imask.size != src.size:
        raise ValueError(f"src_mask has {imask.size} cells, the source grid {src.size}")
    valid = imask[src4] != 0
    w0 = np.where(valid, w4[:, :, 0], 0.0)
    tot = w0.sum(axis=1)
    flat = (tot == 0.0) & valid.any(axis=1)
    w0[flat] = valid[flat] / valid[flat].sum(axis=1, keepdims=True)
    tot[flat] = 1.0
    with np.errstate(invalid="ignore", divide="ignore"):
        w4[:, :, 0] = w0 / tot[:, None]
    return _scrip_dataset(src, dst, (src4[valid] + 1).astype(np.int32), dst4[valid], w4[valid], "bic", src_imask=imask)


def _healpix_bilinear(src, dst, src_mask=None):
    """4-point interpolation from a HEALPix source: the two pixels either side of the point on the ring above and
    on the ring below it, linear in longitude along each ring and in colatitude between the rings (the standard
    HEALPix interpolation scheme); beyond the first / last ring the four pixels of that ring share the polar part.
    Weights are >= 0 and sum to 1.  With `src_mask`, masked pixels are dropped and the rest renormalised."""
    nside = src.nside
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    lon, lat = dst.centers()
    n = lon.size
    theta = np.radians(90.0 - np.asarray(lat, dtype=np.float64))
    phi = np.radians(np.asarray(lon, dtype=np.float64) % 360.0)
    z = np.cos(theta)
    az = np.abs(z)
    ir_cap = np.floor(nside * np.sqrt(3.0 * (1.0 - az))).astype(np.int64)
    ir1 = np.where(az <= 2.0 / 3.0, np.floor(nside * (2.0 - 1.5 * z)).astype(np.int64),
                   np.where(z > 0, ir_cap, 4 * nside - ir_cap - 1))            # ring above the point (0 = none)
    ir2 = ir1 + 1

    def ring_info(ir):
        """(first pixel, pixels, colatitude, shifted) of ring ir (1 .. 4 nside - 1), RING order."""
        ir = np.clip(ir, 1, 4 * nside - 1)
        north = ir < nside
        south = ir > 3 * nside
        irs = 4 * nside - ir
        nr = np.where(north, 4 * ir, np.where(south, 4 * irs, 4 * nside))
        sp = np.where(north, 2 * ir * (ir - 1), np.where(south, npix - 2 * irs * (irs + 1), ncap + (ir - nside) * 4 * nside))
        zz = np.where(north, 1.0 - ir * ir / (3.0 * nside * nside),
                      np.where(south, -(1.0 - irs * irs / (3.0 * nside * nside)), (2 * nside - ir) * 2.0 / (3.0 * nside)))
        shifted = np.where(north | south, 1, 1 - ((ir - nside) & 1))
        return sp, nr, np.arccos(np.clip(zz, -1.0, 1.0)), shifted

    def along(ir):
        sp, nr, th, sh = ring_info(ir)
        dphi = 2.0 * np.pi / nr
        tmp = phi / dphi - 0.5 * sh
        i1 = np.floor(tmp).astype(np.int64)
        w1 = tmp - i1
        i1 = np.where(i1 < 0, i1 + nr, i1)
        i2 = np.where(i1 + 1 >= nr, 0, i1 + 1)
        return sp + i1, sp + i2, 1.0 - w1, w1, th

    a1, a2, wa1, wa2, th1 = along(ir1)
    b1, b2, wb1, wb2, th2 = along(ir2)
    pix = np.stack([a1, a2, b1, b2], axis=1)
    wgt = np.stack([wa1, wa2, wb1, wb2], axis=1)
    north = ir1 == 0
    south = ir2 == 4 * nside
    mid = ~(north | south)
    with np.errstate(invalid="ignore", divide="ignore"):
        wt = np.where(mid, (theta - th1) / np.where(th2 != th1, th2 - th1, 1.0), 0.0)
    wgt[mid, 0:2] *= (1.0 - wt[mid])[:, None]
    wgt[mid, 2:4] *= wt[mid][:, None]
    if north.any():                        # above the first ring: its four pixels
        wn = theta[north] / th2[north]
        fac = (1.0 - wn) * 0.25
        pix[north, 0] = (pix[north, 2] + 2) & 3
        pix[north, 1] = (pix[north, 3] + 2) & 3
        wgt[north, 0] = fac
        wgt[north, 1] = fac
        wgt[north, 2] = wgt[north, 2] * wn + fac
        wgt[north, 3] = wgt[north, 3] * wn + fac
    if south.any():                        # below the last ring
        ws = (theta[south] - th1[south]) / (np.pi - th1[south])
        fac = ws * 0.25
        wgt[south, 0] = wgt[south, 0] * (1.0 - ws) + fac
        wgt[south, 1] = wgt[south, 1] * (1.0 - ws) + fac
        pix[south, 2] = ((pix[south, 0] + 2) & 3) + npix - 4
        pix[south, 3] = ((pix[south, 1] + 2) & 3) + npix - 4
        wgt[south, 2] = fac
        wgt[south, 3] = fac
    if src.nested:                          # the source field is stored in nested order
        nlon, nlat = healpix_centers(nside, nested=True)
        ring2nest = np.argsort(healpix_ring_index(nside, nlon, nlat))
        pix = ring2nest[pix]
    order = np.argsort(pix, axis=1, kind="stable")
    pix = np.take_along_axis(pix, order, axis=1)
    wgt = np.take_along_axis(wgt, order, axis=1)
    dst4 = np.repeat(np.arange(1, n + 1, dtype=np.int32)[:, None], 4, axis=1)
    if src_mask is None:
        return _scrip_dataset(src, dst, (pix.ravel() + 1).astype(np.int32), dst4.ravel(), wgt.ravel(), "bil")
    imask = (np.asarray(src_mask).ravel() != 0).astype(np.int32)
    if imask.size != src.size:
        raise ValueError(f"src_mask has {imask.size} cells, the source grid {src.size}")
    valid = imask[pix] != 0
    wv = np.where(valid, wgt, 0.0)
    tot = wv.sum(axis=1)
    flat = (tot == 0.0) & valid.any(axis=1)
    wv[flat] = valid[flat] / valid[flat].sum(axis=1, keepdims=True)
    tot[flat] = 1.0
    with np.errstate(invalid="ignore", divide="ignore"):
        wv = wv / tot[:, None]
    return _scrip_dataset(src, dst, (pix[valid] + 1).astype(np.int32), dst4[valid], wv[valid], "bil", src_imask=imask)


def _curvilinear_bilinear(src, dst, src_mask=None, candidates=6, iterations=30, extrapolate=True):
    """Bilinear weights from a curvilinear source (2-D lon / lat of the cell centres, e.g. an ORCA ocean grid): SCRIP's
    scheme, as CDO's genbil uses it.  The four neighbouring centres (j, i), (j, i+1), (j+1, i+1), (j+1, i) span a
    quadrilateral in the (lon, lat) plane -- longitudes taken relative to the target point, so the date line is no
    seam; the quadrilateral that holds the target point is looked up among those whose centroids are nearest, the
    point's (a, b) in [0, 1]^2 come from Newton's iteration on the bilinear map, and the weights are
    (1-a)(1-b), a(1-b), ab, (1-a)b.  Points that plane cannot place (it tears at the geographic poles) are looked up
    once more in the tangent plane at the point itself.  A grid that closes in longitude without repeating columns gets the wrap-around
    quadrilaterals too.  Masked corners are dropped and the rest renormalised; a point in no quadrilateral (beyond the
    edge of a regional grid, inside the hole of a tripolar one) or with four masked corners takes the nearest unmasked
    centre with weight 1 -- CDO's REMAP_EXTRAPOLATE=on, which is what the reference's CdoGenerate sets by default; with
    `extrapolate=False` such a point gets no link."""
    from scipy.spatial import cKDTree
    nx, ny = src.shape2d
    slon, slat = (np.asarray(a, dtype=np.float64).reshape(ny, nx) for a in src.centers())
    imask = None
    if src_mask is not None:
        imask = (np.asarray(src_mask).ravel() != 0).astype(np.int32)
        if imask.size != src.size:
            raise ValueError(f"src_mask has {imask.size} cells, the source grid {src.size}")
    if nx < 2 or ny < 2:
        return nearest_weights(src, dst, src_mask=src_mask)
    unit = _unit_vectors(slon.ravel(), slat.ravel()).reshape(ny, nx, 3)
    step = np.linalg.norm(unit[:, 1:] - unit[:, :-1], axis=2)
    seam = np.linalg.norm(unit[:, 0] - unit[:, -1], axis=1)
    wrap = bool(np.median(seam) < 2.0 * np.median(step) and np.median(seam) > 0.25 * np.median(step))
    ii = np.arange(nx if wrap else nx - 1)
    jj = np.arange(ny - 1)
    J, I = np.meshgrid(jj, ii, indexing="ij")
    I1 = (I + 1) % nx
    corner = np.stack([J * nx + I, J * nx + I1, (J + 1) * nx + I1, (J + 1) * nx + I], axis=-1).reshape(-1, 4)
    flat_unit = unit.reshape(-1, 3)
    centroid = flat_unit[corner].sum(axis=1)
    norm = np.linalg.norm(centroid, axis=1)
    good = norm > 1e-9
    corner, centroid = corner[good], centroid[good] / norm[good, None]
    tree = cKDTree(centroid)
    tlon, tlat = dst.centers()
    n = tlon.size
    k = int(min(candidates, corner.shape[0]))
    tunit = _unit_vectors(tlon, tlat)
    _, cand = tree.query(tunit, k=k)
    cand = cand.reshape(n, k)
    flon, flat_ = slon.ravel(), slat.ravel()
    east = np.stack([-np.sin(np.radians(tlon)), np.cos(np.radians(tlon)), np.zeros(n)], axis=1)
    north = np.cross(tunit, east)

    def plane(points, q, gnomonic):
        """Corners `q` (P, 4) as seen from target points `points`: (lon, lat) offsets, or the tangent plane at the
        point (great circles straight, no pole, no date line) for the places the first cannot handle."""
        if not gnomonic:
            return ((flon[q] - tlon[points, None] + 180.0) % 360.0) - 180.0, flat_[q] - tlat[points, None], \
                np.ones(q.shape, dtype=bool)
        v = flat_unit[q]                                                       # (P, 4, 3)
        dp = np.einsum("pck,pk->pc", v, tunit[points])
        front = dp > 0.2
        dp = np.where(front, dp, 1.0)
        return np.einsum("pck,pk->pc", v, east[points]) / dp, np.einsum("pck,pk->pc", v, north[points]) / dp, front

    def locate(points, gnomonic):
        """Quadrilateral, (a, b) and success flag of every point in `points`."""
        quad = np.full(points.size, -1, dtype=np.int64)
        todo = np.arange(points.size)
        for c in range(k):
            if todo.size == 0:
                break
            x, y, front = plane(points[todo], corner[cand[points[todo], c]], gnomonic)
            x2, y2 = np.roll(x, -1, axis=1), np.roll(y, -1, axis=1)
            cr = x * y2 - x2 * y                                               # side of every edge the point lies on
            inside = ((cr >= 0).all(axis=1) | (cr <= 0).all(axis=1)) & (np.abs(cr).max(axis=1) > 0) & front.all(axis=1)
            if not gnomonic:
                inside &= np.abs(x).max(axis=1) < 90.0
            quad[todo[inside]] = cand[points[todo[inside]], c]
            todo = todo[~inside]
        hit = np.flatnonzero(quad >= 0)
        q = corner[quad[hit]]
        x, y, _ = plane(points[hit], q, gnomonic)
        a = np.full(hit.size, 0.5)
        b = np.full(hit.size, 0.5)
        dx1, dx2, dx3 = x[:, 1] - x[:, 0], x[:, 3] - x[:, 0], x[:, 0] - x[:, 1] + x[:, 2] - x[:, 3]
        dy1, dy2, dy3 = y[:, 1] - y[:, 0], y[:, 3] - y[:, 0], y[:, 0] - y[:, 1] + y[:, 2] - y[:, 3]
        for _ in range(iterations):                                            # Newton on x(a, b) = 0, y(a, b) = 0
            fx = x[:, 0] + dx1 * a + dx2 * b + dx3 * a * b
            fy = y[:, 0] + dy1 * a + dy2 * b + dy3 * a * b
            xa, xb = dx1 + dx3 * b, dx2 + dx3 * a
            ya, yb = dy1 + dy3 * b, dy2 + dy3 * a
            det = xa * yb - xb * ya
            det = np.where(np.abs(det) > 1e-300, det, 1e-300)
            da, db = (-fx * yb + fy * xb) / det, (-fy * xa + fx * ya) / det
            a, b = a + da, b + db
            if max(np.abs(da).max(initial=0.0), np.abs(db).max(initial=0.0)) < 1e-13:
                break
        ok = np.isfinite(a) & np.isfinite(b) & (a > -1e-6) & (a < 1 + 1e-6) & (b > -1e-6) & (b < 1 + 1e-6)
        return points[hit[ok]], q[ok], np.clip(a[ok], 0.0, 1.0), np.clip(b[ok], 0.0, 1.0)

    found, q, a, b = locate(np.arange(n), gnomonic=False)
    left = np.setdiff1d(np.arange(n), found, assume_unique=True)
    if left.size:                     # around the geographic poles the (lon, lat) plane tears: look again on the sphere
        f2, q2, a2, b2 = locate(left, gnomonic=True)
        found, q, a, b = np.concatenate([found, f2]), np.concatenate([q, q2]), np.concatenate([a, a2]), \
            np.concatenate([b, b2])
    w4 = np.stack([(1 - a) * (1 - b), a * (1 - b), a * b, (1 - a) * b], axis=1)
    valid = np.ones_like(q, dtype=bool) if imask is None else imask[q] != 0
    # the overlap columns of a tripolar grid name the same place twice: two corners on one spot are fine, weights add up
    wv = np.where(valid, w4, 0.0)
    tot = wv.sum(axis=1)
    flat0 = (tot == 0.0) & valid.any(axis=1)
    wv[flat0] = valid[flat0] / valid[flat0].sum(axis=1, keepdims=True)
    tot[flat0] = 1.0
    usable = valid.any(axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        wv = wv / tot[:, None]
    keep = valid & usable[:, None]
    d4 = np.repeat(found[:, None], 4, axis=1)
    src_addr, dst_addr, w = q[keep], d4[keep], wv[keep]
    rest = np.setdiff1d(np.arange(n), found[usable], assume_unique=True)    # extrapolation: nearest unmasked centre
    cells = np.arange(src.size, dtype=np.int64) if imask is None else np.flatnonzero(imask)
    if rest.size and cells.size and extrapolate:
        _, idx = cKDTree(flat_unit[cells]).query(_unit_vectors(tlon[rest], tlat[rest]), k=1)
        src_addr = np.concatenate([src_addr, cells[idx]])
        dst_addr = np.concatenate([dst_addr, rest])
        w = np.concatenate([w, np.ones(rest.size)])
    src_addr, dst_addr, w = _sort_links(src_addr + 1, dst_addr + 1, w)
    return _scrip_dataset(src, dst, src_addr.astype(np.int32), dst_addr.astype(np.int32), w, "bil", src_imask=imask)


def nearest_weights(src, dst, src_mask=None):
    """Nearest source cell, one link of weight 1 per destination.  With ``src_mask`` the nearest
    UNMASKED cell is taken (CDO's gennn searches unmasked cells only)."""
    src, dst = parse_grid(src), parse_grid(dst)
    imask = None
    if src_mask is not None:
        imask = (np.asarray(src_mask).ravel() != 0).astype(np.int32)
        if imask.size != src.size:
            raise ValueError(f"src_mask has {imask.size} cells, the source grid {src.size}")
    if src.kind != "regular" or imask is not None or not _is_cyclic(src):
        # cell-centre list (HEALPix, unstructured), masked source or regional lon/lat grid: nearest by great-circle
        # distance == nearest by chord length between unit vectors
        from scipy.spatial import cKDTree
        slon, slat = src.centers()
        lon, lat = dst.centers()
        cells = np.arange(src.size, dtype=np.int64) if imask is None else np.flatnonzero(imask)
        d = np.arange(lon.size, dtype=np.int64)
        if cells.size == 0:
            return _scrip_dataset(src, dst, np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0), "nn",
                                  src_imask=imask)
        _, idx = cKDTree(_unit_vectors(slon[cells], slat[cells])).query(_unit_vectors(lon, lat), k=1)
        return _scrip_dataset(src, dst, cells[idx] + 1, d + 1, np.ones(lon.size), "nn", src_imask=imask)
    nx, ny = src.lon.size, src.lat.size
    lon, lat = dst.centers()
    dlon = 360.0 / nx
    i = np.floor(((lon - src.lon[0]) % 360.0) / dlon + 0.5).astype(np.int64) % nx
    jb = np.clip(np.searchsorted(src.lat_b, lat, side="right") - 1, 0, ny - 1)
    d = np.arange(lon.size, dtype=np.int64)
    return _scrip_dataset(src, dst, jb * nx + i + 1, d + 1, np.ones(lon.size), "nn")


def distance_weights(src, dst, src_mask=None, neighbours=4):
    """Inverse-distance weighted average of the `neighbours` nearest (unmasked) source cell centres (CDO's
    gendis: four neighbours, weights 1 / great-circle distance, normalised; a coinciding centre takes it all)."""
    from scipy.spatial import cKDTree
    src, dst = parse_grid(src), parse_grid(dst)
    imask = None
    if src_mask is not None:
        imask = (np.asarray(src_mask).ravel() != 0).astype(np.int32)
        if imask.size != src.size:
            raise ValueError(f"src_mask has {imask.size} cells, the source grid {src.size}")
    slon, slat = src.centers()
    lon, lat = dst.centers()
    cells = np.arange(src.size, dtype=np.int64) if imask is None else np.flatnonzero(imask)
    n = lon.size
    if cells.size == 0:
        return _scrip_dataset(src, dst, np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0), "dis", src_imask=imask)
    k = int(min(neighbours, cells.size))
    chord, idx = cKDTree(_unit_vectors(slon[cells], slat[cells])).query(_unit_vectors(lon, lat), k=k)
    chord, idx = chord.reshape(n, k), idx.reshape(n, k)
    ang = 2.0 * np.arcsin(np.clip(chord / 2.0, 0.0, 1.0))                 # great-circle angle
    hit = ang <= 1e-14
    with np.errstate(divide="ignore"):
        w = np.where(hit.any(axis=1, keepdims=True), hit.astype(np.float64), 1.0 / np.where(hit, 1.0, ang))
    w = w / w.sum(axis=1, keepdims=True)
    src_idx = cells[idx]
    order = np.argsort(src_idx, axis=1, kind="stable")
    src_idx = np.take_along_axis(src_idx, order, axis=1)
    w = np.take_along_axis(w, order, axis=1)
    dst_idx = np.repeat(np.arange(1, n + 1, dtype=np.int64)[:, None], k, axis=1)
    return _scrip_dataset(src, dst, src_idx.ravel() + 1, dst_idx.ravel(), w.ravel(), "dis", src_imask=imask)


def _overlap_1d(sb, db, periodic=None):
    """Overlap lengths between source intervals sb[k]..sb[k+1] and destination
    intervals db[m]..db[m+1].  Returns (dst_idx, src_idx, length)."""
    ns, nd = sb.size - 1, db.size - 1
    di, si, ln = [], [], []
    shifts = (0.0,) if periodic is None else (-periodic, 0.0, periodic)
    for sh in shifts:
        lo = np.maximum(db[:-1, None], sb[None, :-1] + sh)
        hi = np.minimum(db[1:, None], sb[None, 1:] + sh)
        ov = hi - lo
        m, k = np.nonzero(ov > 1e-12 * max(1.0, float(np.abs(db).max())))
        di.append(m)
        si.append(k)
        ln.append(ov[m, k])
    di, si, ln = np.concatenate(di), np.concatenate(si), np.concatenate(ln)
    # merge pieces of the same (dst, src) pair produced by different periodic images
    key = di * ns + si
    order = np.argsort(key, kind="stable")
    key, di, si, ln = key[order], di[order], si[order], ln[order]
    first = np.concatenate(([True], key[1:] != key[:-1]))
    seg = np.cumsum(first) - 1
    out_len = np.zeros(int(seg[-1]) + 1 if seg.size else 0)
    np.add.at(out_len, seg, ln)
    return di[first], si[first], out_len


def conservative_weights(src, dst, src_mask=None, norm="fracarea"):
    """First-order conservative weights between regular lon/lat grids.
    Overlap areas are exact in (lon, sin lat).  Masked source cells
    (src_mask == 0) contribute no links; ``dst_grid_frac`` is the unmasked
    fraction of each destination cell and weights are normalised per
    `fracarea` (rows with frac > 0 sum to 1) or `destarea`."""
    src, dst = parse_grid(src), parse_grid(dst)
    if src.kind != "regular" or dst.kind != "regular":
        raise ValueError("conservative generation needs cells: regular / Gaussian / HEALPix grids, or a list of cell "
                         "centres that comes with its cell vertices (lon_bnds / lat_bnds in the source Dataset) and "
                         "a regular destination grid; a bare list of centres has no areas to conserve")
    nx, mx = src.lon.size, dst.lon.size
    ld, ls, lw = _overlap_1d(src.lon_b, dst.lon_b, periodic=360.0)
    td, ts, tw = _overlap_1d(np.sin(src.lat_b * DEG), np.sin(dst.lat_b * DEG))
    dst_addr = (td[:, None] * mx + ld[None, :]).ravel()
    src_addr = (ts[:, None] * nx + ls[None, :]).ravel()
    area = (tw[:, None] * (lw[None, :] * DEG)).ravel()
    dst_area = (np.diff(np.sin(dst.lat_b * DEG))[:, None] * (np.diff(dst.lon_b) * DEG)[None, :]).ravel()
    imask = None
    if src_mask is not None:
        imask = (np.asarray(src_mask).ravel() != 0).astype(np.int32)
        keep = imask[src_addr] != 0
        dst_addr, src_addr, area = dst_addr[keep], src_addr[keep], area[keep]
    covered = np.zeros(dst.size)
    np.add.at(covered, dst_addr, area)
    frac = covered / dst_area
    if norm == "fracarea":
        w = area / covered[dst_addr]
    elif norm == "destarea":
        w = area / dst_area[dst_addr]
    else:
        raise ValueError("norm must be 'fracarea' or 'destarea'")
    src_addr, dst_addr, w = _sort_links(src_addr + 1, dst_addr + 1, w)
    return _scrip_dataset(src, dst, src_addr, dst_addr, w, "con", src_imask=imask, dst_frac=frac,
                          dst_area=dst_area, norm=norm)


def healpix_ring_index(nside, lon, lat):
    """RING-order pixel index of the directions (lon, lat) in degrees (the standard ang2pix_ring)."""
    z = np.sin(np.radians(np.asarray(lat, dtype=np.float64)))
    za = np.abs(z)
    tt = (np.asarray(lon, dtype=np.float64) % 360.0) / 90.0          # in [0, 4)
    nl4 = 4 * nside
    pix = np.empty(z.shape, dtype=np.int64)
    eq = za <= 2.0 / 3.0
    # equatorial belt
    t1 = nside * (0.5 + tt[eq])
    t2 = nside * z[eq] * 0.75
    jp = np.floor(t1 - t2).astype(np.int64)                            # ascending edge line
    jm = np.floor(t1 + t2).astype(np.int64)                            # descending edge line
    ir = nside + 1 + jp - jm                                           # ring counted from z = 2/3, 1 .. 2 nside + 1
    kshift = 1 - (ir & 1)
    ip = ((jp + jm - nside + kshift + 1) // 2) % nl4
    pix[eq] = 2 * nside * (nside - 1) + (ir - 1) * nl4 + ip
    # polar caps
    cap = ~eq
    tp = tt[cap] - np.floor(tt[cap])
    tmp = nside * np.sqrt(3.0 * (1.0 - za[cap]))
    jp = np.floor(tp * tmp).astype(np.int64)
    jm = np.floor((1.0 - tp) * tmp).astype(np.int64)
    ir = jp + jm + 1                                                   # ring counted from the closest pole
    ip = np.minimum(np.floor(tt[cap] * ir).astype(np.int64), 4 * ir - 1)
    north = z[cap] > 0
    pix[cap] = np.where(north, 2 * ir * (ir - 1) + ip, 12 * nside * nside - 2 * ir * (ir + 1) + ip)
    return pix


def _regular_cell_index(grid, lon, lat):
    """Address (lon fastest) of the regular grid's cell that holds each direction; -1 for a direction the grid does
    not cover (a regional grid: beyond its first / last cell edge in longitude or latitude)."""
    nx, ny = grid.lon.size, grid.lat.size
    span = grid.lon_b[-1] - grid.lon_b[0]
    lat = np.asarray(lat)
    u = (np.asarray(lon) - grid.lon_b[0]) % 360.0
    outside = (lat < grid.lat_b[0] - 1e-12) | (lat > grid.lat_b[-1] + 1e-12)
    if abs(span - 360.0) < 1e-9 and np.allclose(np.diff(grid.lon_b), span / nx):
        i = np.minimum((u / (span / nx)).astype(np.int64), nx - 1)
    else:
        if span < 360.0 - 1e-9:
            outside |= u > span + 1e-12
        i = np.clip(np.searchsorted(grid.lon_b - grid.lon_b[0], u, side="right") - 1, 0, nx - 1)
    j = np.clip(np.searchsorted(grid.lat_b, lat, side="right") - 1, 0, ny - 1)
    return np.where(outside, -1, j * nx + i)


def sampled_conservative_weights(src, dst, src_mask=None, norm="fracarea", samples=None, chunk=1 << 22):
    """First-order conservative weights when one side is a HEALPix grid and the other a regular lon/lat grid.

    HEALPix pixels have curved edges but a gift: a pixel at resolution nside is exactly the union of the 4^k
    EQUAL-AREA pixels of its nested subdivision at nside * 2^k.  The overlap area of a HEALPix pixel with a
    lon/lat cell is therefore (sub-pixels whose centre falls into the cell) / 4^k of the pixel's area, with an
    error that shrinks with the sub-pixel size along the cell edges only.  `samples` = 4^k sub-pixels per pixel
    (default: at least 64, and enough that a sub-pixel is at most a quarter of the other grid's cell width).
    CDO's own gencon approximates HEALPix pixels too (quadrilaterals with great-circle edges).

    Masked source cells (src_mask == 0) take no part; `dst_grid_frac` = unmasked share of the destination cell,
    weights normalised per `fracarea` (rows with a link sum to 1) or `destarea`, as conservative_weights does."""
    src, dst = parse_grid(src), parse_grid(dst)
    hp_is_dst = dst.cdo_type == "healpix" and dst.nside is not None
    hp, reg = (dst, src) if hp_is_dst else (src, dst)
    if hp.cdo_type != "healpix" or hp.nside is None or reg.kind != "regular":
        raise ValueError("sampled conservative weights need one HEALPix grid (hp<N>[_nested|_ring]) and one regular grid")
    nside, npix = hp.nside, 12 * hp.nside * hp.nside
    if samples is None:
        cell = min(float(np.min(np.diff(reg.lon_b))), float(np.min(np.diff(reg.lat_b)[1:-1])) if reg.lat.size > 2 else 180.0)
        pix_deg = np.degrees(np.sqrt(4.0 * np.pi / npix))
        k = 3
        while pix_deg / 2 ** k > cell / 4.0 and k < 8:
            k += 1
    else:
        k = max(0, int(round(np.log(max(int(samples), 1)) / np.log(4.0))))
    fine = nside << k
    per = 4 ** k
    imask = None
    if src_mask is not None:
        imask = (np.asarray(src_mask).ravel() != 0).astype(np.int32)
        if imask.size != src.size:
            raise ValueError(f"src_mask has {imask.size} cells, the source grid {src.size}")
    # pairs (hp pixel, regular cell) with their sub-pixel counts, accumulated over chunks of fine pixels
    keys, counts = [], []
    n_reg = reg.size
    for lo in range(0, 12 * fine * fine, chunk):
        hi = min(12 * fine * fine, lo + chunk)
        flon, flat = _healpix_centers_range(fine, lo, hi)
        parent = np.arange(lo, hi, dtype=np.int64) >> (2 * k)                   # nested index of the coarse pixel
        cell_idx = _regular_cell_index(reg, flon, flat)
        inside = cell_idx >= 0                     # a regional lon/lat grid: sub-pixels beyond its edges belong to no cell
        parent, cell_idx = parent[inside], cell_idx[inside]
        key = parent * n_reg + cell_idx
        uk, uc = np.unique(key, return_counts=True)
        keys.append(uk)
        counts.append(uc)
    key = np.concatenate(keys)
    cnt = np.concatenate(counts).astype(np.float64)
    uk, inv = np.unique(key, return_inverse=True)                               # chunks may split a pixel's sub-pixels
    cnt = np.bincount(inv, weights=cnt)
    hp_idx, reg_idx = uk // n_reg, uk % n_reg
    if not hp_is_dst:
        # a lon/lat target cell too small to catch a sub-pixel centre (the last rows before a pole; a fine regional
        # grid under few `samples`) takes the pixel that holds its own centre -- HEALPix covers the sphere, so the
        # centre of every cell of a regional grid lies in some pixel too
        empty = np.flatnonzero(np.bincount(reg_idx, minlength=n_reg) == 0)
        if empty.size:
            clon, clat = reg.centers()
            nlon, nlat = healpix_centers(nside, nested=True)
            ring2nest = np.argsort(healpix_ring_index(nside, nlon, nlat))
            hp_idx = np.concatenate([hp_idx, ring2nest[healpix_ring_index(nside, clon[empty], clat[empty])]])
            reg_idx = np.concatenate([reg_idx, empty])
            cnt = np.concatenate([cnt, np.full(empty.size, 1e-6)])               # a token area: one link, weight 1
    if not hp.nested:                                                           # ring-ordered grid: renumber the pixels
        plon, plat = healpix_centers(nside, nested=True)
        hp_idx = healpix_ring_index(nside, plon, plat)[hp_idx]
    pix_area = 4.0 * np.pi / npix
    area = cnt / per * pix_area                                                 # overlap area on the unit sphere
    if hp_is_dst:
        dst_addr, src_addr = hp_idx, reg_idx
        dst_area = np.full(npix, pix_area)
    else:
        dst_addr, src_addr = reg_idx, hp_idx
        dst_area = (np.diff(np.sin(reg.lat_b * DEG))[:, None] * (np.diff(reg.lon_b) * DEG)[None, :]).ravel()
    if imask is not None:
        keep = imask[src_addr] != 0
        dst_addr, src_addr, area = dst_addr[keep], src_addr[keep], area[keep]
    covered = np.bincount(dst_addr, weights=area, minlength=dst.size)
    if hp_is_dst:
        frac = covered / dst_area
    else:
        # a lon/lat cell's sampled area (all sub-pixels that fell into it) stands for its area: the fraction is
        # the unmasked share of the samples, free of the sampling error of the cell's own outline
        total = np.bincount(reg_idx, weights=cnt / per * pix_area, minlength=dst.size)
        with np.errstate(invalid="ignore", divide="ignore"):
            frac = np.where(total > 0, covered / total, 0.0)
    if norm == "fracarea":
        w = area / covered[dst_addr]
    elif norm == "destarea":
        w = area / (dst_area[dst_addr] if hp_is_dst else np.maximum(np.bincount(reg_idx, weights=cnt / per * pix_area,
                                                                        minlength=dst.size), 1e-300)[dst_addr])
    else:
        raise ValueError("norm must be 'fracarea' or 'destarea'")
    src_addr, dst_addr, w = _sort_links(src_addr + 1, dst_addr + 1, w)
    return _scrip_dataset(src, dst, src_addr, dst_addr, w, "con", src_imask=imask, dst_frac=np.clip(frac, 0.0, 1.0),
                          dst_area=dst_area, norm=norm)


def polygon_areas(lon_v, lat_v):
    """Areas on the unit sphere of cells given by their vertices (cells, V) in degrees, great-circle edges: the signed
    spherical excesses of the triangles (vertex 0, vertex k, vertex k + 1), tan(E / 2) = a . (b x c) / (1 + a.b + b.c
    + c.a).  Vertices repeated to pad a short polygon add triangles of no area; either orientation."""
    lon_v, lat_v = np.asarray(lon_v, dtype=np.float64), np.asarray(lat_v, dtype=np.float64)
    n, V = lon_v.shape
    v = _unit_vectors(lon_v.ravel(), lat_v.ravel()).reshape(n, V, 3)
    a, b, c = v[:, :1], v[:, 1:-1], v[:, 2:]
    triple = np.einsum("nvk,nvk->nv", np.broadcast_to(a, b.shape), np.cross(b, c))
    denom = 1.0 + np.einsum("nvk,nvk->nv", np.broadcast_to(a, b.shape), b) + np.einsum("nvk,nvk->nv", b, c) + \
        np.einsum("nvk,nvk->nv", c, np.broadcast_to(a, c.shape))
    return np.abs(2.0 * np.arctan2(triple, denom).sum(axis=1))


class _PolygonLocator:
    """Which cell of a polygon grid (cell vertices (cells, V) in degrees, great-circle edges) holds a direction?
    Candidates are the cells with the nearest centres (KD-tree on unit vectors), tested nearest first in the
    candidate's own gnomonic plane, where great circles are straight lines (crossing-number test)."""

    def __init__(self, grid, neighbours=8):
        from scipy.spatial import cKDTree
        lon_v, lat_v = (np.asarray(a, dtype=np.float64) for a in grid.vertices)
        n, V = lon_v.shape
        if n != grid.size:
            raise ValueError("cell vertices do not match the number of cells")
        clon, clat = grid.centers()
        self.n = n
        self.centre = _unit_vectors(clon, clat)
        self.tree = cKDTree(self.centre)
        self.k = int(min(neighbours, n))
        self.east = np.stack([-np.sin(np.radians(clon)), np.cos(np.radians(clon)), np.zeros(n)], axis=1)
        self.north = np.cross(self.centre, self.east)
        vert = _unit_vectors(lon_v.ravel(), lat_v.ravel()).reshape(n, V, 3)
        dotc = np.einsum("nvk,nk->nv", vert, self.centre)
        self.usable = (dotc > 1e-6).all(axis=1)                               # a vertex beyond the horizon: not a cell
        dotc = np.where(dotc > 1e-6, dotc, 1.0)
        self.vx = np.einsum("nvk,nk->nv", vert, self.east) / dotc
        self.vy = np.einsum("nvk,nk->nv", vert, self.north) / dotc
        self.vx2, self.vy2 = np.roll(self.vx, -1, axis=1), np.roll(self.vy, -1, axis=1)

    def locate(self, p):
        """Cell index per unit vector of `p` (P, 3); -1 where no candidate cell holds it.  A point that lies exactly on
        an edge two cells share can fall out of both (each cell is tested in its own projection): points nobody claims
        are tried once more a hair (1e-7 rad) to the north-east, which decides for one side and changes no area."""
        owner = self._locate(p)
        lost = np.flatnonzero(owner < 0)
        if lost.size:
            q = p[lost]
            east = np.stack([-q[:, 1], q[:, 0], np.zeros(lost.size)], axis=1)
            norm = np.linalg.norm(east, axis=1, keepdims=True)
            east = np.where(norm > 1e-12, east / np.where(norm > 1e-12, norm, 1.0), np.array([[1.0, 0.0, 0.0]]))
            north = np.cross(q, east)
            q = q + 1e-7 * (0.6 * east + 0.8 * north)
            owner[lost] = self._locate(q / np.linalg.norm(q, axis=1, keepdims=True))
        return owner

    def _locate(self, p):
        _, cand = self.tree.query(p, k=self.k)
        cand = cand.reshape(-1, self.k)
        owner = np.full(p.shape[0], -1, dtype=np.int64)
        todo = np.arange(p.shape[0])
        for c in range(self.k):               # nearest centre first: most points lie in that cell, few reach the later ones
            if todo.size == 0:
                break
            cc = cand[todo, c]
            q = p[todo]
            dp = np.einsum("pk,pk->p", q, self.centre[cc])
            ok = (dp > 1e-6) & self.usable[cc]
            dp = np.where(ok, dp, 1.0)
            px = (np.einsum("pk,pk->p", q, self.east[cc]) / dp)[:, None]
            py = (np.einsum("pk,pk->p", q, self.north[cc]) / dp)[:, None]
            x1, y1, x2, y2 = self.vx[cc], self.vy[cc], self.vx2[cc], self.vy2[cc]
            with np.errstate(invalid="ignore", divide="ignore"):
                cross = ((y1 > py) != (y2 > py)) & (px < (x2 - x1) * (py - y1) / (y2 - y1) + x1)
            inside = ok & (np.sum(cross, axis=1) % 2 == 1)
            owner[todo[inside]] = cc[inside]
            todo = todo[~inside]
        return owner


def _regular_cell_areas(grid):
    return (np.diff(np.sin(grid.lat_b * DEG))[:, None] * (np.diff(grid.lon_b) * DEG)[None, :]).ravel()


def _grid_cell_areas(grid):
    """Cell areas on the unit sphere of any grid that has cells."""
    if grid.kind == "regular":
        return _regular_cell_areas(grid)
    if grid.vertices is not None:
        return polygon_areas(*grid.vertices)
    if grid.cdo_type == "healpix" and grid.nside:
        return np.full(grid.size, 4.0 * np.pi / grid.size)
    raise ValueError("the grid has no cells (a list of centres)")


def _typical_cell_degrees(grid):
    if grid.kind == "regular":
        return min(float(np.min(np.diff(grid.lon_b))), float(np.min(np.diff(grid.lat_b))))
    return float(np.degrees(np.sqrt(np.median(_grid_cell_areas(grid)))))


def _sample_lattice(grid, m, chunk):
    """Equal-area sample points carried by the cells of a regular or HEALPix grid: yields (unit vectors, owning cell,
    area of one sample) in chunks.  Regular: m x m sub-cells per cell, uniform in longitude and in sin latitude;
    HEALPix: the 4^k pixels of the nested subdivision with 4^k >= m^2 (the parent pixel is their exact union)."""
    if grid.kind == "regular":
        mx = grid.lon.size
        fr = (np.arange(m) + 0.5) / m
        sub_lon = (grid.lon_b[:-1, None] + np.diff(grid.lon_b)[:, None] * fr[None, :]).ravel()
        sb = np.sin(grid.lat_b * DEG)
        sub_lat = np.degrees(np.arcsin(np.clip((sb[:-1, None] + np.diff(sb)[:, None] * fr[None, :]).ravel(), -1.0, 1.0)))
        area = _regular_cell_areas(grid) / (m * m)
        n_sub = sub_lon.size * sub_lat.size
        for lo in range(0, n_sub, chunk):
            idx = np.arange(lo, min(n_sub, lo + chunk), dtype=np.int64)
            jj, ii = idx // sub_lon.size, idx % sub_lon.size
            cell = (jj // m) * mx + (ii // m)
            yield _unit_vectors(sub_lon[ii], sub_lat[jj]), cell, area[cell]
        return
    k = max(0, int(np.ceil(np.log2(max(m, 1)))))
    fine = grid.nside << k
    to_file_order = None
    if not grid.nested:                                   # the field is stored in ring order
        nlon, nlat = healpix_centers(grid.nside, nested=True)
        to_file_order = healpix_ring_index(grid.nside, nlon, nlat)
    area = 4.0 * np.pi / (12.0 * fine * fine)
    for lo in range(0, 12 * fine * fine, chunk):
        hi = min(12 * fine * fine, lo + chunk)
        flon, flat = _healpix_centers_range(fine, lo, hi)
        parent = np.arange(lo, hi, dtype=np.int64) >> (2 * k)
        if to_file_order is not None:
            parent = to_file_order[parent]
        yield _unit_vectors(flon, flat), parent, np.full(hi - lo, area)


def _polygon_overlap_areas(src, dst, m, chunk=200000):
    """(dst cell, src cell, overlap area on the unit sphere) of every pair of cells of two grids of which at least one
    is a grid of POLYGONS: see polygon_conservative_weights.  Kept on the source grid per target, so the levels of a
    3-D field (same cells, another mask) pay for the geometry once."""
    cache = src.__dict__.setdefault("_overlap_cache", {})
    key = (dst.kind, dst.size, dst.lon.tobytes()[:4096], dst.lat.tobytes()[:4096],
           None if dst.kind != "regular" else (dst.lon_b.tobytes(), dst.lat_b.tobytes()),
           None if dst.vertices is None else tuple(np.asarray(v).tobytes()[:4096] for v in dst.vertices),
           getattr(dst, "nested", None), m)
    if key in cache:
        return cache[key]
    has_lattice = lambda g: g.kind == "regular" or (g.cdo_type == "healpix" and g.nside is not None)   # noqa: E731
    if has_lattice(dst):                       # samples carried by the target cells, looked up in the source polygons
        carrier, other, carrier_is_dst = dst, _PolygonLocator(src), True
    elif has_lattice(src):                     # ... or carried by the source cells, looked up in the target polygons
        carrier, other, carrier_is_dst = src, _PolygonLocator(dst), False
    else:                                      # polygons on both sides: a fine HEALPix lattice looked up in both
        cell = min(_typical_cell_degrees(src), _typical_cell_degrees(dst))
        # pixels half as wide as cell / m (a HEALPix pixel of nside n is 58.6 / n degrees across)
        nside = int(np.clip(2 ** int(np.ceil(np.log2(max(58.6 * 2 * m / max(cell, 1e-6), 1.0)))), 8, 1024))
        carrier = Grid("points", np.zeros(0), np.zeros(0), name=f"hp{nside}", cdo_type="healpix")
        carrier.nside, carrier.nested = nside, True
        carrier.lon = carrier.lat = np.zeros(12 * nside * nside)          # sizes only; the centres come in chunks
        loc_s, loc_d = _PolygonLocator(src), _PolygonLocator(dst)
    n_src = src.size
    keys, sums = [], []
    both = not (has_lattice(dst) or has_lattice(src))
    # the area of a polygon target cell AS THE SAMPLES SEE IT (so that a fully covered cell has fraction 1 exactly);
    # only when the samples reach everywhere -- a regional carrier leaves part of a target cell unsampled
    whole_sphere = both or carrier.kind != "regular" or \
        (abs(carrier.lon_b[-1] - carrier.lon_b[0] - 360.0) < 1e-6 and carrier.lat_b[0] <= -90.0 + 1e-6
         and carrier.lat_b[-1] >= 90.0 - 1e-6)
    seen = np.zeros(dst.size) if (dst.vertices is not None and whole_sphere) else None
    for p, cell, area in _sample_lattice(carrier, 1 if both else m, chunk):
        if both:
            d_own, s_own = loc_d.locate(p), loc_s.locate(p)
        elif carrier_is_dst:
            d_own, s_own = cell, other.locate(p)
        else:
            d_own, s_own = other.locate(p), cell
        if seen is not None:
            seen += np.bincount(d_own[d_own >= 0], weights=area[d_own >= 0], minlength=dst.size)
        hit = (d_own >= 0) & (s_own >= 0)
        uk, inv = np.unique(d_own[hit] * n_src + s_own[hit], return_inverse=True)
        keys.append(uk)
        sums.append(np.bincount(inv, weights=area[hit], minlength=uk.size))
    key_all = np.concatenate(keys) if keys else np.zeros(0, np.int64)
    tot = np.concatenate(sums) if sums else np.zeros(0)
    uk, inv = np.unique(key_all, return_inverse=True)
    tot = np.bincount(inv, weights=tot) if uk.size else tot
    cache.clear()                                                             # one target at a time is enough
    cache[key] = (uk // n_src, uk % n_src, tot, _grid_cell_areas(dst), seen)
    return cache[key]


def polygon_conservative_weights(src, dst, src_mask=None, norm="fracarea", samples=None):
    """First-order conservative weights when at least one side is a grid of POLYGON cells (unstructured meshes,
    curvilinear grids: the cell vertices a file carries as lon_bnds / lat_bnds, edges taken as great circles as CDO
    does); the other side may be a regular lon/lat grid, a HEALPix grid or polygons again.

    The overlap areas are counted on a lattice of equal-area sample points: the m x m sub-cells (uniform in longitude
    and in sin latitude) of the regular side's cells, or the nested sub-pixels of the HEALPix side's pixels, or -- with
    polygons on both sides -- the pixels of a fine HEALPix grid.  A sample belongs to the polygon that holds it, searched
    among the nearest cell centres and tested in the candidate's gnomonic plane, where great circles are straight
    lines.  Areas are exact up to the samples cut by a polygon edge; `samples` = m (default: samples a third of the
    smaller cells wide, 3 <= m <= 10); a polygon target cell's area is counted on the same samples, so a fully covered
    cell has fraction 1 exactly (`dst_grid_area` holds the exact polygon area).  Target area no source cell covers (outside a regional mesh, land of an ocean
    mesh) counts as uncovered: `dst_grid_frac` = unmasked covered share, as for lon/lat sources."""
    src, dst = parse_grid(src), parse_grid(dst)
    if src.vertices is None and dst.vertices is None:
        raise ValueError("polygon conservative weights need cell vertices on at least one side")
    for g, side in ((src, "source"), (dst, "destination")):
        if g.vertices is None and g.kind != "regular" and not (g.cdo_type == "healpix" and g.nside):
            raise ValueError(f"polygon conservative weights: the {side} grid has no cells (regular, HEALPix or cell "
                             "vertices are needed)")
    n = src.size
    imask = None
    if src_mask is not None:
        imask = (np.asarray(src_mask).ravel() != 0).astype(np.int32)
        if imask.size != n:
            raise ValueError(f"src_mask has {imask.size} cells, the source grid {n}")
    if samples is None:
        carrier, other = (dst, src) if dst.vertices is None else (src, dst)
        ratio = _typical_cell_degrees(carrier) / max(_typical_cell_degrees(other), 1e-9) if carrier.vertices is None \
            else 1.0
        m = int(np.clip(np.ceil(3.0 * ratio), 3, 10))
    else:
        m = max(1, int(samples))
    dst_addr, src_addr, area, dst_area, seen = _polygon_overlap_areas(src, dst, m)
    if imask is not None:
        keep = imask[src_addr] != 0
        dst_addr, src_addr, area = dst_addr[keep], src_addr[keep], area[keep]
    covered = np.bincount(dst_addr, weights=area, minlength=dst.size)
    counted = dst_area if seen is None else seen                    # the target cell's area in the overlaps' own measure
    with np.errstate(invalid="ignore", divide="ignore"):
        frac = np.where(counted > 0, covered / counted, 0.0)
    if norm == "fracarea":
        w = area / covered[dst_addr]
    elif norm == "destarea":
        w = area / counted[dst_addr]
    else:
        raise ValueError("norm must be 'fracarea' or 'destarea'")
    src_addr, dst_addr, w = _sort_links(src_addr + 1, dst_addr + 1, w)
    return _scrip_dataset(src, dst, src_addr, dst_addr, w, "con", src_imask=imask, dst_frac=np.clip(frac, 0.0, 1.0),
                          dst_area=dst_area, norm=norm)


def healpix_hierarchy_weights(src, dst, src_mask=None, norm="fracarea"):
    """First-order conservative weights between two HEALPix grids: exact, because a coarse pixel is the union of
    the 4^k pixels below it in the nested hierarchy.  Coarsening averages the (unmasked) children, refining copies
    the parent.  Either grid may be in ring order."""
    src, dst = parse_grid(src), parse_grid(dst)

    def nest_of(g):       # nested index of every pixel in the grid's own order
        if g.nested:
            return np.arange(g.size, dtype=np.int64)
        nlon, nlat = healpix_centers(g.nside, nested=True)
        ring_of_nest = healpix_ring_index(g.nside, nlon, nlat)
        out = np.empty(g.size, dtype=np.int64)
        out[ring_of_nest] = np.arange(g.size, dtype=np.int64)
        return out

    def order_of(g):      # the grid's own index of every nested pixel
        inv = np.empty(g.size, dtype=np.int64)
        inv[nest_of(g)] = np.arange(g.size, dtype=np.int64)
        return inv

    ks, kd = int(round(np.log2(src.nside))), int(round(np.log2(dst.nside)))
    if 2 ** ks != src.nside or 2 ** kd != dst.nside:
        raise ValueError("HEALPix hierarchy weights need nside to be powers of two")
    if ks >= kd:          # coarsening (or the same resolution): every source pixel has one parent
        parent_nest = nest_of(src) >> (2 * (ks - kd))
        dst_addr = order_of(dst)[parent_nest]
        src_addr = np.arange(src.size, dtype=np.int64)
    else:                 # refining: every destination pixel has one parent
        parent_nest = nest_of(dst) >> (2 * (kd - ks))
        src_addr = order_of(src)[parent_nest]
        dst_addr = np.arange(dst.size, dtype=np.int64)
    area = np.full(src_addr.size, 4.0 * np.pi / max(src.size, dst.size))      # the finer pixel's area per link
    dst_area = np.full(dst.size, 4.0 * np.pi / dst.size)
    imask = None
    if src_mask is not None:
        imask = (np.asarray(src_mask).ravel() != 0).astype(np.int32)
        if imask.size != src.size:
            raise ValueError(f"src_mask has {imask.size} cells, the source grid {src.size}")
        keep = imask[src_addr] != 0
        dst_addr, src_addr, area = dst_addr[keep], src_addr[keep], area[keep]
    covered = np.bincount(dst_addr, weights=area, minlength=dst.size)
    frac = covered / dst_area
    if norm == "fracarea":
        w = area / covered[dst_addr]
    elif norm == "destarea":
        w = area / dst_area[dst_addr]
    else:
        raise ValueError("norm must be 'fracarea' or 'destarea'")
    src_addr, dst_addr, w = _sort_links(src_addr + 1, dst_addr + 1, w)
    return _scrip_dataset(src, dst, src_addr, dst_addr, w, "con", src_imask=imask, dst_frac=np.clip(frac, 0.0, 1.0),
                          dst_area=dst_area, norm=norm)


def _healpix_centers_range(nside, lo, hi):
    """Centres of the NESTED pixels lo .. hi - 1 at resolution nside (degrees)."""
    pix = np.arange(lo, hi, dtype=np.int64)
    npface = nside * nside
    face = pix // npface
    p = pix % npface

    def compact(v):
        v = v & 0x5555555555555555
        v = (v | (v >> 1)) & 0x3333333333333333
        v = (v | (v >> 2)) & 0x0F0F0F0F0F0F0F0F
        v = (v | (v >> 4)) & 0x00FF00FF00FF00FF
        v = (v | (v >> 8)) & 0x0000FFFF0000FFFF
        v = (v | (v >> 16)) & 0x00000000FFFFFFFF
        return v

    ix = compact(p)
    iy = compact(p >> 1)
    jrll = np.array([2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4], dtype=np.int64)
    jpll = np.array([1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7], dtype=np.int64)
    jr = jrll[face] * nside - ix - iy - 1
    nr = np.where(jr < nside, jr, np.where(jr > 3 * nside, 4 * nside - jr, nside))
    z = np.where(jr < nside, 1.0 - nr * nr / (3.0 * nside * nside),
                 np.where(jr > 3 * nside, -1.0 + nr * nr / (3.0 * nside * nside),
                          (2 * nside - jr) * 2.0 / (3.0 * nside)))
    kshift = np.where((jr < nside) | (jr > 3 * nside), 0, (jr - nside) & 1)
    jp = (jpll[face] * nr + ix - iy + 1 + kshift) // 2
    jp = np.where(jp > 4 * nr, jp - 4 * nr, jp)
    jp = np.where(jp < 1, jp + 4 * nr, jp)
    phi = (jp - (kshift + 1) * 0.5) * (np.pi / 2.0) / nr
    return np.degrees(phi) % 360.0, np.degrees(np.arcsin(np.clip(z, -1.0, 1.0)))


def conservative2_weights(src, dst, src_mask=None, norm="fracarea"):
    """SCRIP second-order conservative weights between regular lon/lat grids (Jones 1999, eqs. 4 - 6), `num_wgts` = 3:
    per overlap A_nk of source cell n with destination cell k

        w1 = |A_nk|,   w2 = int_A (lat - lat_n) dA,   w3 = int_A cos(lat) (lon - lon_n) dA

    with (lat_n, lon_n) the area centroid of the source cell, all three normalised like the first-order weight.
    A destination value is then sum_n f_n w1 + (df/dlat)_n w2 + (1/cos(lat) df/dlon)_n w3.  The reference applies
    column 0 only (weights.py:33), which is the first-order weight.  Exact for lon/lat boxes."""
    src, dst = parse_grid(src), parse_grid(dst)
    if src.kind != "regular" or dst.kind != "regular":
        raise ValueError("second-order conservative generation needs regular source and destination grids")
    first = conservative_weights(src, dst, src_mask=src_mask, norm=norm)
    s_addr = first["src_address"].values.astype(np.int64) - 1
    d_addr = first["dst_address"].values.astype(np.int64) - 1
    nx, mx = src.lon.size, dst.lon.size
    js, is_ = s_addr // nx, s_addr % nx
    jd, id_ = d_addr // mx, d_addr % mx
    sb, db = src.lat_b * DEG, dst.lat_b * DEG
    t1 = np.maximum(sb[js], db[jd])                       # latitude range of the overlap
    t2 = np.minimum(sb[js + 1], db[jd + 1])
    # longitude range of the overlap, measured from the source cell's centre (periodic)
    sl1, sl2 = src.lon_b[is_], src.lon_b[is_ + 1]
    mid = 0.5 * (sl1 + sl2)
    dl1 = ((dst.lon_b[id_] - mid + 180.0) % 360.0) - 180.0
    dl2 = dl1 + (dst.lon_b[id_ + 1] - dst.lon_b[id_])
    half = 0.5 * (sl2 - sl1)
    # a destination cell wider than 360 - source width could meet the source cell twice; the first-order pass
    # merged those pieces, here the piece around the source centre is the one integrated (global grids of
    # ordinary resolution have a single piece)
    p1 = np.maximum(dl1, -half) * DEG
    p2 = np.minimum(dl2, half) * DEG
    wrap = p2 <= p1                                        # the overlap lies on the other side of the date line
    p1 = np.where(wrap, np.maximum(dl1 + 360.0, -half) * DEG, p1)
    p2 = np.where(wrap, np.minimum(dl2 + 360.0, half) * DEG, p2)

    def i_cos(a, b):          # int cos(t) dt
        return np.sin(b) - np.sin(a)

    def i_tcos(a, b):         # int t cos(t) dt
        return (b * np.sin(b) + np.cos(b)) - (a * np.sin(a) + np.cos(a))

    def i_cos2(a, b):         # int cos^2(t) dt
        return 0.5 * (b - a) + 0.25 * (np.sin(2 * b) - np.sin(2 * a))

    lat_c = i_tcos(sb[:-1], sb[1:]) / i_cos(sb[:-1], sb[1:])             # area centroid latitude of every source row
    dphi = p2 - p1
    w1 = i_cos(t1, t2) * dphi
    w2 = (i_tcos(t1, t2) - lat_c[js] * i_cos(t1, t2)) * dphi
    w3 = i_cos2(t1, t2) * 0.5 * (p2 * p2 - p1 * p1)                       # lon measured from the source centre
    scale = first["remap_matrix"].values[:, 0] / np.where(w1 != 0.0, w1, 1.0)   # the first-order normalisation
    w = np.stack([first["remap_matrix"].values[:, 0], w2 * scale, w3 * scale], axis=1)
    ds = first
    ds["remap_matrix"] = (("num_links", "num_wgts"), w)
    ds.attrs["map_method"] = "Conservative remapping, second order"
    return ds


def _unflipped(grid):
    """The same grid without the north-to-south flag (weights in its internal south-to-north order)."""
    if grid.kind != "regular" or not grid.lat_descending:
        return grid
    g = Grid("regular", grid.lon, grid.lat, grid.lon_b, grid.lat_b, name=grid.name, cdo_type=grid.cdo_type)
    return g


def _flip_rows(values, nx):
    """Reverse the latitude rows of a per-cell vector stored lon-fastest."""
    v = np.asarray(values)
    return v.reshape(-1, nx)[::-1].ravel()


def _flip_address(addr, nx, ny):
    a = np.asarray(addr, dtype=np.int64) - 1
    return ((ny - 1 - a // nx) * nx + a % nx + 1).astype(np.int32)


def _inside_source(src, dst, method):
    """Target points the source grid reaches WITHOUT extrapolation (CDO's REMAP_EXTRAPOLATE=off): for the point-wise
    interpolations (bil, bic) the hull of the source cell CENTRES -- a point beyond the first / last row or column of
    centres has no enclosing cell --, for nn / dis the area the source CELLS cover.  A bare list of centres has no
    extent: everything counts as inside."""
    lon, lat = dst.centers()
    inside = np.ones(lon.size, dtype=bool)
    if src.kind == "regular":
        centres = method in ("bil", "bic")
        la = src.lat if centres else src.lat_b
        inside &= (lat >= la[0] - 1e-12) & (lat <= la[-1] + 1e-12)
        if not _is_cyclic(src):
            lo = src.lon if centres else src.lon_b
            u = (lon - lo[0]) % 360.0
            inside &= u <= (lo[-1] - lo[0]) + 1e-12
    elif src.vertices is not None and method not in ("bil", "bic"):
        inside &= _PolygonLocator(src).locate(_unit_vectors(lon, lat)) >= 0
    return inside


def generate_weights(src, dst, method="con", src_mask=None, norm="fracarea", extrapolate=True):
    """Dispatch on CDO method names (cdogenerate.py:73): con/ycon -> conservative,
    bil -> bilinear, nn -> nearest, dis -> inverse-distance average of four neighbours.  Grids whose latitude axis runs north-to-south are
    computed south-to-north and renumbered to the file's cell order afterwards."""
    src, dst = parse_grid(src), parse_grid(dst)
    flip_s = src.kind == "regular" and src.lat_descending
    flip_d = dst.kind == "regular" and dst.lat_descending
    if flip_s and src_mask is not None:
        src_mask = _flip_rows(np.asarray(src_mask).ravel(), src.lon.size)
    if method in ("con", "ycon"):
        if src.cdo_type == "healpix" and dst.cdo_type == "healpix" and src.nside and dst.nside:
            ds = healpix_hierarchy_weights(src, dst, src_mask=src_mask, norm=norm)
        elif src.vertices is not None or dst.vertices is not None:
            ds = polygon_conservative_weights(src, dst, src_mask=src_mask, norm=norm)
        elif "healpix" in (src.cdo_type, dst.cdo_type) and (src.kind == "regular" or dst.kind == "regular"):
            ds = sampled_conservative_weights(src, dst, src_mask=src_mask, norm=norm)
        else:
            ds = conservative_weights(src, dst, src_mask=src_mask, norm=norm)
    elif method == "bil":
        ds = bilinear_weights(src, dst, src_mask=src_mask, extrapolate=extrapolate)
    elif method == "nn":
        ds = nearest_weights(src, dst, src_mask=src_mask)
    elif method == "dis":
        ds = distance_weights(src, dst, src_mask=src_mask)
    elif method == "con2":
        ds = conservative2_weights(src, dst, src_mask=src_mask, norm=norm)
    elif method == "bic":
        ds = bicubic_weights(src, dst, src_mask=src_mask)
    elif method == "laf":
        # largest area fraction: the (unmasked) source cell with the largest overlap, weight 1 -- read off the
        # conservative weights of the same pair (ties: the lowest source address, the order links are stored in)
        con = generate_weights(_unflipped(src), _unflipped(dst), method="con",
                               src_mask=None if src_mask is None else np.asarray(src_mask).ravel(), norm="fracarea")
        d_all, s_all, w_all = con["dst_address"].values, con["src_address"].values, con["remap_matrix"].values[:, 0]
        order = np.lexsort((s_all, -w_all, d_all))                       # per destination: heaviest link first
        first = np.concatenate(([True], d_all[order][1:] != d_all[order][:-1]))
        pick = order[first]
        ds = _scrip_dataset(src, dst, s_all[pick], d_all[pick], np.ones(pick.size), "laf",
                            src_imask=None if src_mask is None else (np.asarray(src_mask).ravel() != 0).astype(np.int32),
                            dst_frac=con["dst_grid_frac"].values)
    else:
        raise ValueError(f"method '{method}' is not available without the cdo binary "
                         "(native generator: con, ycon, con2, bil, bic, nn, dis, laf)")
    if not extrapolate and method in ("bil", "bic", "nn", "dis"):
        # REMAP_EXTRAPOLATE=off: target points outside the source grid get no link (missing after the regrid)
        keep = _inside_source(_unflipped(src), _unflipped(dst), method)[ds["dst_address"].values - 1]
        if not keep.all():
            for name in ("src_address", "dst_address"):
                ds[name] = (("num_links",), ds[name].values[keep])
            ds["remap_matrix"] = (("num_links", "num_wgts"), ds["remap_matrix"].values[keep])
            # a destination that lost all its links is not covered: its fraction is 0 (and the destination mask says
            # so), as a consumer of `dst_grid_frac` / remap_area_min expects -- not the extrapolating case's 1
            linked = np.bincount(ds["dst_address"].values - 1, minlength=dst.size) > 0
            for name, zero in (("dst_grid_frac", 0.0), ("dst_grid_imask", 0)):
                if name in ds:
                    v = ds[name].values.copy()
                    v[~linked] = zero
                    ds[name] = (ds[name].dims, v, ds[name].attrs)
    if not (flip_s or flip_d):
        return ds
    src_addr, dst_addr = ds["src_address"].values, ds["dst_address"].values
    w = ds["remap_matrix"].values
    if flip_s:
        nx, ny = src.lon.size, src.lat.size
        src_addr = _flip_address(src_addr, nx, ny)
        for name in ("src_grid_imask", "src_grid_center_lat", "src_grid_center_lon"):
            ds[name] = (ds[name].dims, _flip_rows(ds[name].values, nx), ds[name].attrs)
    if flip_d:
        mx, my = dst.lon.size, dst.lat.size
        dst_addr = _flip_address(dst_addr, mx, my)
        for name in ("dst_grid_imask", "dst_grid_frac", "dst_grid_area", "dst_grid_center_lat",
                     "dst_grid_center_lon"):
            if name in ds:
                ds[name] = (ds[name].dims, _flip_rows(ds[name].values, mx), ds[name].attrs)
    order = np.lexsort((src_addr, dst_addr))
    ds["src_address"] = (("num_links",), src_addr[order])
    ds["dst_address"] = (("num_links",), dst_addr[order])
    ds["remap_matrix"] = (("num_links", "num_wgts"), w[order])
    return ds


def stack_level_weights(weights_list, levels, mask_dim="lev", method="con"):
    """Combine per-level 2-D weights into the 3-D layout of
    cdogenerate.py:310-343: link arrays zero-padded to the longest level plus
    ``link_length``; masks (and frac/area for conservative methods) per level."""
    nl = [int(w["src_address"].shape[0]) for w in weights_list]
    nl_max = max(nl) if nl else 0
    L = len(weights_list)
    first = weights_list[0]
    ds = Dataset(attrs=dict(first.attrs))
    ds.coords[mask_dim] = DataArray(np.asarray(levels), dims=(mask_dim,), name=mask_dim)
    ds["link_length"] = ((mask_dim,), np.asarray(nl, dtype=np.int64))
    per_level = ["src_address", "dst_address", "remap_matrix", "src_grid_imask", "dst_grid_imask"]
    if method in ("ycon", "con2", "con"):
        per_level += ["dst_grid_area", "dst_grid_frac"]
    for name, var in first.data_vars.items():
        if name in per_level:
            continue
        ds[name] = var
    for name in per_level:
        if name not in first:
            continue
        v0 = first[name]
        if "num_links" in v0.dims or "numLinks" in v0.dims:   # CDO writes either name (cdogenerate.py:315)
            shape = (L, nl_max) + tuple(v0.shape[1:])
            out = np.zeros(shape, dtype=v0.values.dtype)
            for i, w in enumerate(weights_list):
                out[i, :nl[i]] = w[name].values
        else:
            out = np.stack([w[name].values for w in weights_list], axis=0)
        ds[name] = ((mask_dim,) + tuple(v0.dims), out)
    return ds


# --------------------------------------------------------------------------- masked levels

class ConservativeLevels:
    """Conservative weights for many land/sea masks of one grid pair: the
    overlap geometry is computed once, each level only filters and normalises
    (what the reference gets from one ``cdo -sellevidx,k`` run per level,
    cdogenerate.py:179-228)."""

    def __init__(self, src, dst):
        self.src, self.dst = parse_grid(src), parse_grid(dst)
        if self.src.kind != "regular" or self.dst.kind != "regular":
            raise ValueError("conservative generation needs regular source and destination grids")
        nx, mx = self.src.lon.size, self.dst.lon.size
        ld, ls, lw = _overlap_1d(self.src.lon_b, self.dst.lon_b, periodic=360.0)
        td, ts, tw = _overlap_1d(np.sin(self.src.lat_b * DEG), np.sin(self.dst.lat_b * DEG))
        dst_addr = (td[:, None] * mx + ld[None, :]).ravel()
        src_addr = (ts[:, None] * nx + ls[None, :]).ravel()
        area = (tw[:, None] * (lw[None, :] * DEG)).ravel()
        order = np.lexsort((src_addr, dst_addr))
        self.dst_addr, self.src_addr, self.area = dst_addr[order], src_addr[order], area[order]
        self.dst_area = (np.diff(np.sin(self.dst.lat_b * DEG))[:, None] *
                         (np.diff(self.dst.lon_b) * DEG)[None, :]).ravel()

    def level(self, src_mask=None, norm="fracarea"):
        dst_addr, src_addr, area = self.dst_addr, self.src_addr, self.area
        imask = None
        if src_mask is not None:
            imask = (np.asarray(src_mask).ravel() != 0).astype(np.int32)
            keep = imask[src_addr] != 0
            dst_addr, src_addr, area = dst_addr[keep], src_addr[keep], area[keep]
        covered = np.bincount(dst_addr, weights=area, minlength=self.dst.size)
        frac = covered / self.dst_area
        w = area / (covered[dst_addr] if norm == "fracarea" else self.dst_area[dst_addr])
        return _scrip_dataset(self.src, self.dst, src_addr + 1, dst_addr + 1, w, "con",
                              src_imask=imask, dst_frac=frac, dst_area=self.dst_area, norm=norm)

    def stack(self, masks, levels, mask_dim="lev"):
        return stack_level_weights([self.level(m) for m in masks], levels, mask_dim=mask_dim,
                                   method="con")


def synthetic_ocean_masks(nx, ny, n_lev, seed=20260723, top=0.66, bottom=0.05):
    """Depth-dependent land/sea masks (n_lev, ny*nx) for benchmarks: a smooth random
    bathymetry thresholded so the ocean fraction falls monotonically from `top`
    at level 0 to `bottom` at the deepest level, deeper oceans nested in shallower ones."""
    rng = np.random.default_rng(seed)
    cy, cx = max(ny // 24, 4), max(nx // 24, 4)
    coarse = rng.standard_normal((cy, cx))
    yi = np.linspace(0, cy - 1, ny)
    xi = np.linspace(0, cx - 1, nx)
    y0 = np.clip(np.floor(yi).astype(int), 0, cy - 2)
    x0 = np.clip(np.floor(xi).astype(int), 0, cx - 2)
    fy = (yi - y0)[:, None]
    fx = (xi - x0)[None, :]
    b = ((1 - fy) * (1 - fx) * coarse[y0][:, x0] + (1 - fy) * fx * coarse[y0][:, x0 + 1] +
         fy * (1 - fx) * coarse[y0 + 1][:, x0] + fy * fx * coarse[y0 + 1][:, x0 + 1])
    b += 0.05 * rng.standard_normal((ny, nx))
    fracs = np.linspace(top, bottom, n_lev) if n_lev > 1 else np.array([top])
    thr = np.quantile(b, 1.0 - fracs)
    return (b.ravel()[None, :] > thr[:, None]).astype(np.int32)
