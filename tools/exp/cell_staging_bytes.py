"""Bytes a cell-staging batch-fastest kernel would move on config 3 (per block size) against the native tile plan
and the algorithmic bytes: host-only arithmetic on the bench operators."""
import numpy as np, sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from smmregrid_amd import gridgen
nx, ny, n_lev = 1442, 1021, 75
src = gridgen.regular_grid(nx, ny)
masks = gridgen.synthetic_ocean_masks(nx, ny, n_lev)
cl = gridgen.ConservativeLevels(src, "r360x180")
S, D, T = nx*ny, 64800, 120
tot = {4:0, 8:0, 16:0, 64:0}
lines64 = 0; U = 0; nnz = 0; lines_distinct = 0
for lv in range(n_lev):
    w = cl.level(masks[lv])
    s = w["src_address"].values.astype(np.int64)-1; d = w["dst_address"].values.astype(np.int64)-1
    nnz += s.size; U += np.unique(s).size
    for R in tot:
        tot[R] += np.unique((d//R)*S + s).size
    # native: distinct 128-B lines (16 f64) per 64-row block, rows padded to 128-B lines
    lines64 += np.unique((d//64)*(S//16+1) + s//16).size
    lines_distinct += np.unique(s//16).size
alg = T*(U + n_lev*D)*8 + nnz*12
print("algorithmic GB", alg/1e9, "U", U, "nnz", nnz)
print("native staged lines x128B x T:", lines64*128*T/1e9, "GB; distinct lines", lines_distinct*128*T/1e9)
for R,c in tot.items():
    print(f"cell staging, {R}-row blocks: cells {c}  X bytes (T=120 exact) {c*T*8/1e9:.1f} GB, with 128-entry lines {c*128*8/1e9:.1f} GB; ratio to U {c/U:.3f}")
print("Y GB", T*n_lev*D*8/1e9)
