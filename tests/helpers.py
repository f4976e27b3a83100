"""Shared builders for the parity tests (seeded synthetic operators and fields)."""
import numpy as np


def random_links(rng, n_src, n_dst, nnz, dup_frac=0.05, zero_frac=0.02):
    """Random SCRIP links (1-based) with duplicates, explicit zeros and empty rows."""
    src = rng.integers(1, n_src + 1, size=nnz).astype(np.int32)
    dst = rng.integers(1, n_dst + 1, size=nnz).astype(np.int32)
    w = rng.uniform(-0.5, 1.5, size=nnz)
    ndup = int(nnz * dup_frac)
    if ndup:
        pick = rng.integers(0, nnz, size=ndup)
        tgt = rng.integers(0, nnz, size=ndup)
        src[tgt] = src[pick]
        dst[tgt] = dst[pick]
    nzero = int(nnz * zero_frac)
    if nzero:
        w[rng.integers(0, nnz, size=nzero)] = 0.0
    return src, dst, w


def ragged_links(rng, n_src, n_dst, max_len=40):
    """Ocean-like ragged rows: row length 0..max_len, clustered source columns."""
    src, dst, w = [], [], []
    for d in range(n_dst):
        ln = int(rng.integers(0, max_len + 1)) if rng.random() > 0.15 else 0
        if ln == 0:
            continue
        c0 = int(rng.integers(0, n_src))
        cols = np.unique((c0 + rng.integers(0, 3 * max_len, size=ln)) % n_src)
        ww = rng.random(cols.size)
        ww /= ww.sum()
        src.append(cols + 1)
        dst.append(np.full(cols.size, d + 1))
        w.append(ww)
    if not src:                            # every row drew "no links" (tiny n_dst): an empty operator
        return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0)
    src = np.concatenate(src).astype(np.int32)
    dst = np.concatenate(dst).astype(np.int32)
    w = np.concatenate(w)
    perm = rng.permutation(src.size)       # the builder must not rely on sorted links
    return src[perm], dst[perm], w[perm]


def field(rng, n_batch, n_src, dtype=np.float64, nan_frac=0.0, inf_frac=0.0):
    x = (250.0 + 30.0 * rng.standard_normal((n_batch, n_src))).astype(dtype)
    if nan_frac:
        x[rng.random(x.shape) < nan_frac] = np.nan
    if inf_frac:
        m = rng.random(x.shape) < inf_frac
        x[m] = np.where(rng.random(m.sum()) < 0.5, np.inf, -np.inf)
    return x


def assert_same(y, ref, rtol=1e-6, exact=False):
    """checker.py:69 protocol tightened: identical NaN positions, values within
    rtol (north star: 1e-6 relative); exact=True demands bit equality."""
    y = np.asarray(y)
    ref = np.asarray(ref)
    assert y.shape == ref.shape
    ny, nr = np.isnan(y), np.isnan(ref)
    assert np.array_equal(ny, nr), f"NaN pattern differs at {np.argwhere(ny != nr)[:5]}"
    if exact:
        assert np.array_equal(y[~ny].view(np.uint64) if y.dtype == np.float64 else y[~ny],
                              ref[~nr].view(np.uint64) if ref.dtype == np.float64 else ref[~nr]), \
            "values are not bit identical"
    else:
        np.testing.assert_allclose(y[~ny], ref[~nr], rtol=rtol, atol=0.0)


def reorder_links(rng, src, dst, w, split_frac=0.3):
    """The same weights matrix written as a different link list: a random share of the links is
    split into two or three duplicate (dst, src) entries whose weights add up to the original,
    then all links are shuffled.  sparse.COO sums duplicates and sorts coordinates
    (weights.py:37-39), so every such list describes one matrix; only the order in which the
    duplicates are summed, and possibly the product's summation order, may differ."""
    src, dst, w = np.asarray(src), np.asarray(dst), np.asarray(w, dtype=np.float64)
    pick = rng.random(src.size) < split_frac
    parts = rng.integers(2, 4, size=src.size)
    s_out, d_out, w_out = [src[~pick]], [dst[~pick]], [w[~pick]]
    for k in (2, 3):
        m = pick & (parts == k)
        cuts = np.sort(rng.random((int(m.sum()), k - 1)), axis=1)
        edges = np.concatenate([np.zeros((cuts.shape[0], 1)), cuts, np.ones((cuts.shape[0], 1))], axis=1)
        share = np.diff(edges, axis=1) * w[m][:, None]          # k pieces summing to w (to rounding)
        s_out.append(np.repeat(src[m], k))
        d_out.append(np.repeat(dst[m], k))
        w_out.append(share.ravel())
    s2, d2, w2 = np.concatenate(s_out), np.concatenate(d_out), np.concatenate(w_out)
    perm = rng.permutation(s2.size)
    return s2[perm].astype(np.int32), d2[perm].astype(np.int32), w2[perm]


def max_rel_spread(outputs):
    """NaN patterns of all outputs must agree; returns the largest relative difference between
    any output and the first over the finite cells."""
    base = np.asarray(outputs[0])
    nan0 = np.isnan(base)
    worst = 0.0
    for y in outputs[1:]:
        y = np.asarray(y)
        assert np.array_equal(np.isnan(y), nan0), "NaN pattern depends on the link order"
        ok = ~nan0
        denom = np.maximum(np.abs(base[ok]), np.finfo(np.float64).tiny)
        if ok.any():
            worst = max(worst, float(np.max(np.abs(y[ok] - base[ok]) / denom)))
    return worst


def kernel_forms(*tile_knobs, sell_knobs=({"sell_batch_rows": 8}, {"sell_batch_rows": 2})):
    """(apply flags, tuning knobs) pairs a parity loop walks: the library's own choice, the forced tile kernel plain
    and under each of `tile_knobs` (dicts of smm_debug_set_tuning knobs), the forced SELL kernel plain and under
    `sell_knobs`.  ABI v5 moved the launch-shape variants out of the apply flags (unknown flag bits are refused
    since v6), so every form is reached through `_lib.tuning(**knobs)`."""
    from smmregrid_amd import _lib
    t, s = _lib.APPLY_KERNEL_TILE, _lib.APPLY_KERNEL_SELL
    return [(0, {}), (t, {})] + [(t, dict(k)) for k in tile_knobs] + [(s, {})] + [(s, dict(k)) for k in sell_knobs]
