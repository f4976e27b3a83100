"""Operator creation time by workload (host builder + uploads)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import WORKLOADS
from smmregrid_amd import SparseOperator, gridgen
for name in sys.argv[1:]:
    method, sgrid, tgrid, _, _ = WORKLOADS[name][:5]
    t = time.time(); w = gridgen.generate_weights(sgrid, tgrid, method=method); tg = time.time() - t
    t = time.time()
    op = SparseOperator(w.sizes["src_grid_size"], w.sizes["dst_grid_size"], w["src_address"].values,
                        w["dst_address"].values, w["remap_matrix"].values, device=0)
    print(f"{name}: weights {tg:.2f} s, operator {time.time() - t:.2f} s, nnz {op.nnz}, {op.plan_info()}")
