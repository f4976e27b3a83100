"""Weights -> device operators.  Mirrors smmregrid/weights.py of the reference
(same function names and argument meaning); the sparse.COO / dask objects are
replaced by `SparseOperator` handles whose arithmetic runs in HIP kernels."""
import numpy as np

from .operator import SparseOperator
from .xrlite import DataArray, Dataset, from_xarray


def _links_dim(weights):
    # CDO 2.2.0 fix (weights.py:12-13)
    return "numLinks" if "numLinks" in weights.dims else "num_links"


def _level_slice(var, mask_dim, i):
    """values of `var` at level i when it carries mask_dim, else the values as they are."""
    v = var.values
    if mask_dim in var.dims:
        return np.take(v, i, axis=var.dims.index(mask_dim))
    return v


def _dst_dims(weights):
    """dst_grid_dims of the weights file (fastest dimension first, regrid.py:572-579), or None."""
    if "dst_grid_dims" not in weights:
        return None
    return np.asarray(weights["dst_grid_dims"].values).astype(np.int32).ravel()


def compute_weights_matrix(weights, device=None, prune_zeros=False):
    """CDO weights -> one operator of shape (S, D)   (weights.py:25-44)."""
    weights = from_xarray(weights)
    src_address = weights["src_address"].values
    dst_address = weights["dst_address"].values
    remap_matrix = weights["remap_matrix"].values
    if remap_matrix.ndim == 2:
        remap_matrix = remap_matrix[:, 0]
    n_src = weights.sizes["src_grid_size"]
    n_dst = weights.sizes["dst_grid_size"]
    return SparseOperator(n_src, n_dst, src_address, dst_address, remap_matrix, device=device,
                          dst_dims=_dst_dims(weights), prune_zeros=prune_zeros)


def compute_weights_matrix3d(weights, mask_dim="lev", device=None, prune_zeros=False, workers=None):
    """Per-level operators; links truncated to link_length[level]   (weights.py:7-23).
    The levels are independent: they are created by up to `workers` host threads at a time (default:
    min(8, levels); the library shares its builder threads between the creations in flight), in level
    order in the returned list."""
    weights = from_xarray(weights)
    link_length = np.asarray(weights["link_length"].values).astype(np.int64)
    n_src = weights.sizes["src_grid_size"]
    n_dst = weights.sizes["dst_grid_size"]
    dims = _dst_dims(weights)
    if device is None:
        from .device import current_device
        device = current_device()          # resolved here: worker threads start on device 0

    def one(i):
        nl = int(link_length[i])
        src = _level_slice(weights["src_address"], mask_dim, i)[:nl]
        dst = _level_slice(weights["dst_address"], mask_dim, i)[:nl]
        rm = _level_slice(weights["remap_matrix"], mask_dim, i)[:nl]
        if rm.ndim == 2:
            rm = rm[:, 0]
        return SparseOperator(n_src, n_dst, src, dst, rm, device=device, dst_dims=dims, prune_zeros=prune_zeros)

    n_lev = len(link_length)
    workers = min(8, n_lev) if workers is None else max(1, min(int(workers), n_lev))
    if workers <= 1:
        return [one(i) for i in range(n_lev)]
    from concurrent.futures import ThreadPoolExecutor
    done = []
    try:
        with ThreadPoolExecutor(max_workers=workers) as pool:
            futures = [pool.submit(one, i) for i in range(n_lev)]
            for f in futures:
                done.append(f.result())
    except BaseException:
        for f in futures:                   # a failing level: release the operators the others made
            if f.done() and not f.cancelled() and f.exception() is None:
                f.result().close()
        raise
    return done


def mask_tensordot(src_mask, weights_matrix):
    """Destination mask from the source mask   (weights.py:47-52)."""
    return weights_matrix.mask_apply(src_mask)


def mask_weights(weights, weights_matrix, mask_dim=None):
    """Pre-compute dst_grid_imask (weights.py:55-84); returns a new weights Dataset."""
    weights = from_xarray(weights)
    src_mask = weights["src_grid_imask"]
    old = weights["dst_grid_imask"]
    if mask_dim is not None:
        levels = [mask_tensordot(_level_slice(src_mask, mask_dim, i), weights_matrix[i])
                  for i in range(weights.sizes[mask_dim])]
        dst_mask = np.stack(levels, axis=0)
        other = [d for d in old.dims if d != mask_dim]
        dims = [mask_dim] + other
    else:
        dst_mask = mask_tensordot(src_mask.values, weights_matrix)
        dims = list(old.dims)
    out = Dataset(attrs=weights.attrs, coords=weights.coords)
    for k, v in weights.data_vars.items():
        out[k] = v
    out["dst_grid_imask"] = DataArray(dst_mask, dims=dims, attrs=old.attrs, name="dst_grid_imask")
    return out


def check_mask(weights, mask_dim=None):
    """True where the destination mask is not all ones (weights.py:103-120):
    scalar bool, or a bool vector over mask_dim."""
    weights = from_xarray(weights)
    wdst = weights["dst_grid_imask"]
    v = wdst.values
    if mask_dim is not None:
        ax = wdst.dims.index(mask_dim)
        red = tuple(i for i in range(v.ndim) if i != ax)
        return ~(v == 1).all(axis=red)
    return bool(~(v == 1).all())
