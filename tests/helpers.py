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
