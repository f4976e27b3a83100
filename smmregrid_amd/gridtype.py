"""Name-based classification of dimensions (reference: gridtype.py:6-12, :130-176).

Only what the apply path needs: which dims are horizontal, which one is the
masked (vertical) dimension, which are time / other.  Grid *kind* detection,
bounds handling and multi-variable bookkeeping of the reference's
GridInspector are metadata and out of this round's scope.
"""

DEFAULT_DIMS = {
    "horizontal": ["i", "j", "x", "y", "lon", "lat", "longitude", "latitude",
                   "cell", "cells", "ncells", "values", "value", "nod2", "pix", "elem",
                   "nav_lon", "nav_lat", "rgrid"],
    "mask": ["lev", "nz1", "nz", "depth", "depth_full", "depth_half"],
    "time": ["time", "time_counter", "valid_time", "forecast_time"],
}


def tolist(value):
    """util.py:8-20 of the reference: None -> None, str -> [str], iterable -> list."""
    if value is None:
        return None
    if isinstance(value, str):
        return [value]
    return list(value)


class GridType:
    """Dimension roles of one grid plus the operator state the Regridder attaches."""

    def __init__(self, dims, extra_dims=None, override=False, weights=None):
        if extra_dims is not None and not isinstance(extra_dims, dict):
            raise TypeError("extra_dims must be a dictionary or None.")
        table = self._table(extra_dims, override)
        dims = list(dims)
        # keep the data's own order (the reference goes through a set; order is irrelevant there)
        self.horizontal_dims = [d for d in dims if d in table.get("horizontal", [])] or None
        mask = [d for d in dims if d in table.get("mask", [])]
        if len(mask) > 1:
            raise ValueError(f"Only one masked dimension can be processed at the time: check {mask}")
        self.mask_dim = mask[0] if mask else None
        self.dims = (self.horizontal_dims or []) + ([self.mask_dim] if self.mask_dim else [])
        self.time_dims = [d for d in dims if d in table.get("time", [])] or None
        used = set(self.horizontal_dims or []) | set(self.time_dims or [])
        if self.mask_dim:
            used.add(self.mask_dim)
        self.other_dims = [d for d in dims if d not in used]
        self.variables = {}
        self.bounds = []
        self.kind = None
        self.masked = None
        self.weights = weights
        self.weights_matrix = None
        self.group = None          # OperatorGroup for masked-level weights

    @staticmethod
    def _table(extra_dims, override):
        if extra_dims is None:
            return DEFAULT_DIMS
        if override:
            return extra_dims
        table = {k: list(v) for k, v in DEFAULT_DIMS.items()}
        for key, extra in extra_dims.items():
            if extra:
                table[key] = list(dict.fromkeys(table.get(key, []) + list(extra)))
        return table

    def __eq__(self, other):
        return isinstance(other, GridType) and set(self.dims) == set(other.dims)

    def __hash__(self):
        return hash(tuple(sorted(self.dims)))

    def __repr__(self):
        return (f"GridType(horizontal_dims={self.horizontal_dims}, mask_dim={self.mask_dim}, "
                f"time_dims={self.time_dims}, other_dims={self.other_dims})")
