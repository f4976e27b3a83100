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


def _ring_worker(rank, world, port, n_rows, tiles, slots, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from smmregrid_amd.distributed import TiledRingGather, tile_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D = 13
        shard = torch.zeros((n_rows, D), dtype=torch.float64)
        seen, errors = [], []

        def value(r, step, rows):          # what rank r's shard holds in these rows at this step
            base = torch.arange(rows[0], rows[1], dtype=torch.float64)[:, None] * 100.0
            return base + torch.arange(D, dtype=torch.float64)[None, :] + 1e6 * r + 1e4 * step

        step_now = [0]

        def on_tile(k, parts):
            r0, r1 = bounds[k]
            seen.append(k)
            if len(parts) != world:
                errors.append(f"tile {k}: {len(parts)} parts")
            for r, part in enumerate(parts):
                if part.shape != (r1 - r0, D) or not torch.equal(part, value(r, step_now[0], (r0, r1))):
                    errors.append(f"step {step_now[0]} tile {k} rank {r}: wrong rows in the ring slot")

        bounds = tile_bounds(n_rows, tiles)
        ring = TiledRingGather(dist, torch, shard, root=0, tiles=tiles, slots=slots, on_tile=on_tile)
        assert ring.tiles == bounds
        for step in range(2):              # the ring is reused across steps as bench.py does
            step_now[0] = step
            for k, (r0, r1) in enumerate(bounds):
                shard[r0:r1] = value(rank, step, (r0, r1))    # stands for the kernel of tile k
                ring.gather_tile(k)
                # a slot is never handed out while its previous tile is undelivered
                assert len(ring.pending) <= slots
            ring.finish()
            assert not ring.pending
        ret[rank] = (errors, seen, ring.gathered_bytes, ring.delivered, len(bounds),
                     None if ring.ring is None else (len(ring.ring), ring.ring[0][0].shape[0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rows,tiles,slots", [(37, 5, 2), (8, 8, 2), (5, 8, 2), (64, 4, 3), (1, 8, 2)])
def test_world2_gloo_tiled_ring_gather(n_rows, tiles, slots):
    """The gather schedule bench.py times as `with_gather`, on 2 gloo ranks: tiles that do not divide
    the rows, ring slots reused within and across steps, every tile delivered once and in order with
    the right rows of the right rank, and the byte accounting bench.py reports."""
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        procs = [mp.get_context("spawn").Process(target=_ring_worker, args=(r, world, port, n_rows, tiles, slots, ret))
                 for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        from smmregrid_amd.distributed import tile_bounds
        bounds = tile_bounds(n_rows, tiles)
        assert bounds[0][0] == 0 and bounds[-1][1] == n_rows and all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
        assert len(bounds) <= tiles
        errors, seen, gathered, delivered, n_tiles, ring_shape = ret[0]
        assert errors == []
        assert seen == list(range(len(bounds))) * 2                    # each tile once per step, in order
        # with_gather accounting: the root receives (world - 1) shards of n_rows x D doubles per step
        assert gathered == 2 * (world - 1) * n_rows * 13 * 8
        assert delivered == 2 * len(bounds) and n_tiles == len(bounds)
        assert ring_shape == (slots, max(b - a for a, b in bounds))   # ring of tile-sized slots, not the full Y
        errors1, seen1, gathered1, delivered1, _, ring1 = ret[1]
        assert errors1 == [] and seen1 == [] and gathered1 == 0 and ring1 is None and delivered1 == 2 * len(bounds)
