"""N>1 path on CPU: world_size-2 gloo process group, batch-row sharding + gather.
The per-rank compute is the CPU oracle here (tests may use it as a stand-in);
on the GPU box the same code runs with the HIP operator (tests/test_gpu_facade.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, gather, n_rows, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import oracle
    from smmregrid_amd import gridgen
    from smmregrid_amd.distributed import regrid_sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = gridgen.conservative_weights("r48x24", "r16x8")
        n_src, n_dst = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
        csr = oracle.coo_to_csr_c(n_src, n_dst, w["src_address"].values, w["dst_address"].values,
                                  w["remap_matrix"].values)
        rng = np.random.default_rng(20260723)
        x = 250.0 + 30.0 * rng.standard_normal((n_rows, n_src))
        x[n_rows // 2, :50] = np.nan
        seen = []

        def apply_fn(rows):
            seen.append(rows.shape[0])
            return oracle.apply_c(csr, rows)

        out = regrid_sharded(x, apply_fn, n_dst, gather=gather)
        ref = oracle.apply_c(csr, x)
        ok = True
        if gather == "none":
            from smmregrid_amd.distributed import shard_bounds
            lo, hi = shard_bounds(n_rows, world, rank)
            ok = np.array_equal(out, ref[lo:hi], equal_nan=True)
        elif gather == "all" or rank == 0:
            ok = out is not None and np.array_equal(out, ref, equal_nan=True)
        else:
            ok = out is None
        ret[rank] = (bool(ok), seen[0])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("gather", ["root", "all", "none"])
@pytest.mark.parametrize("n_rows", [7, 8, 1])
def test_world2_gloo_sharded_regrid(gather, n_rows):
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        procs = [mp.get_context("spawn").Process(target=_worker,
                                                 args=(r, world, port, gather, n_rows, ret))
                 for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        per = -(-n_rows // world)
        assert ret[0] == (True, min(per, n_rows))
        assert ret[1] == (True, max(0, min(per, n_rows - per)))


def test_shard_bounds_cover_rows_exactly():
    from smmregrid_amd.distributed import shard_bounds, shard_sizes
    for n in (0, 1, 7, 8, 3600, 101928):
        for world in (1, 2, 4, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert sum(shard_sizes(n, world)) == n
    assert shard_sizes(8760, 8) == [1095] * 8           # BASELINE config 4
    assert shard_sizes(101928, 8) == [12741] * 8        # BASELINE config 5
