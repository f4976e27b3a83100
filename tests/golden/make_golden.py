"""Generates the committed golden fixtures (inputs + expected outputs).

Run from the repo root:  python tests/golden/make_golden.py
The expected outputs come from the CPU oracle's sequential C restatement
(oracle/oracle.c) and are cross-checked here against the independent
numpy/scipy restatement (oracle/oracle.py) before being written.  They are NOT
outputs of the reference itself, which cannot be imported in this environment
(SURVEY 8c): parity is unpinned by reference artefacts.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402
from smmregrid_amd import gridgen  # noqa: E402
from tests.helpers import field, ragged_links  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20260723)


def check(a, b):
    assert np.array_equal(np.isnan(a), np.isnan(b))
    np.testing.assert_allclose(a[~np.isnan(a)], b[~np.isnan(b)], rtol=1e-12)


def case_bilinear():
    # BASELINE config 1: 2D r180x90 -> r90x45 bilinear, single timestep (plus two more rows with NaNs)
    w = gridgen.bilinear_weights("r180x90", "r90x45")
    n_src, n_dst = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
    src, dst, rm = w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values[:, 0]
    csr = oracle.coo_to_csr_c(n_src, n_dst, src, dst, rm)
    g = gridgen.parse_grid("r180x90")
    lon, lat = g.centers()
    x = np.stack([2.0 * lon + 3.0 * lat + rng.standard_normal(n_src),
                  field(rng, 1, n_src, nan_frac=0.01)[0],
                  field(rng, 1, n_src)[0]])
    imask, frac = w["dst_grid_imask"].values, w["dst_grid_frac"].values
    y = oracle.apply_c(csr, x, False, imask, frac, 0.5)
    check(y, oracle.apply(csr, x, False, imask, frac, 0.5))
    np.savez_compressed(os.path.join(HERE, "bil_r180x90_r90x45.npz"), n_src=n_src, n_dst=n_dst,
                        src_address=src, dst_address=dst, remap_matrix=rm, rowptr=csr[0], col=csr[1],
                        val=csr[2], x=x, y=y, masked=False, dst_imask=imask, dst_frac=frac,
                        area_min=0.5)


def case_masked_levels():
    # config-3-like: conservative weights with a per-level source mask, 4 levels x 3 steps
    src_g, dst_g = gridgen.parse_grid("r48x24"), gridgen.parse_grid("r12x6")
    L, T = 4, 3
    base = rng.random(src_g.size)
    per = []
    for l in range(L):
        mask = (base > 0.25 + 0.2 * l).astype(np.int32)      # ocean shrinks with depth
        per.append(gridgen.conservative_weights(src_g, dst_g, src_mask=mask))
    w3 = gridgen.stack_level_weights(per, np.array([0.5, 10.0, 100.0, 1000.0]))
    ll = w3["link_length"].values
    csrs = [oracle.coo_to_csr_c(src_g.size, dst_g.size, w3["src_address"].values[i, :ll[i]],
                                w3["dst_address"].values[i, :ll[i]],
                                w3["remap_matrix"].values[i, :ll[i], 0]) for i in range(L)]
    simask = w3["src_grid_imask"].values
    dimask = np.stack([oracle.mask_apply_c(csrs[i], simask[i]) for i in range(L)])
    masked_levels = oracle.check_mask(dimask)
    x = field(rng, T * L, src_g.size).reshape(T, L, src_g.size)
    for l in range(L):
        x[:, l, simask[l] == 0] = np.nan
    level_index = np.array([0, 1, 2, 3], np.int32)
    frac = w3["dst_grid_frac"].values
    y = oracle.apply_levels(csrs, x, 1, level_index, masked_levels, dimask, frac, 0.5, True)
    check(y, oracle.apply_levels(csrs, x, 1, level_index, masked_levels, dimask, frac, 0.5, True,
                                 use_c=False))
    np.savez_compressed(os.path.join(HERE, "con_masked_levels.npz"), n_src=src_g.size,
                        n_dst=dst_g.size, link_length=ll, src_address=w3["src_address"].values,
                        dst_address=w3["dst_address"].values,
                        remap_matrix=w3["remap_matrix"].values[:, :, 0], src_imask=simask,
                        dst_imask=dimask, dst_frac=frac, masked_levels=masked_levels,
                        level_index=level_index, x=x, y=y, area_min=0.5,
                        levels=np.array([0.5, 10.0, 100.0, 1000.0]))


def case_ragged():
    n_src, n_dst = 700, 333
    src, dst, rm = ragged_links(rng, n_src, n_dst, max_len=40)
    csr = oracle.coo_to_csr_c(n_src, n_dst, src, dst, rm)
    x = field(rng, 5, n_src, dtype=np.float32, nan_frac=0.03, inf_frac=0.01)
    imask = (rng.random(n_dst) > 0.1).astype(np.int32)
    frac = rng.random(n_dst)
    y = oracle.apply_c(csr, x, True, imask, frac, 0.3)
    check(y, oracle.apply(csr, x, True, imask, frac, 0.3))
    np.savez_compressed(os.path.join(HERE, "ragged_random.npz"), n_src=n_src, n_dst=n_dst,
                        src_address=src, dst_address=dst, remap_matrix=rm, rowptr=csr[0], col=csr[1],
                        val=csr[2], x=x, y=y, masked=True, dst_imask=imask, dst_frac=frac,
                        area_min=0.3)


if __name__ == "__main__":
    case_bilinear()
    case_masked_levels()
    case_ragged()
    print("golden fixtures written to", HERE)
