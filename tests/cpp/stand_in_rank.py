"""TEST INFRASTRUCTURE: one rank process of tests/test_gpu_multirank_stand_in.py.

    python stand_in_rank.py RANK WORLD PORT OUT.json

Started as a fresh child (one per rank, all on device 0), with SMM_RCCL_LIB naming the stand-in
collective library built from tests/cpp/fake_rccl.cpp.  Runs the product's world-size > 1 data path --
smm_comm_* through `Comm.gather / allgather / gather_rows`, `distributed.regrid_sharded` and
`distributed.TiledRingGather` on device memory with the HIP operator as the per-rank compute -- and
checks every assembled result bit for bit against the CPU oracle.  Writes what it saw to OUT.json.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    from oracle import oracle
    from smmregrid_amd import SparseOperator, gridgen, to_device
    from smmregrid_amd.comm import Comm, HostRendezvous
    from smmregrid_amd.device import DeviceArray, set_device, synchronize
    from smmregrid_amd.distributed import TiledRingGather, regrid_sharded, shard_bounds, tile_bounds

    assert "fake_rccl" in os.environ.get("SMM_RCCL_LIB", ""), "this script is for the stand-in library only"
    set_device(0)                                   # every rank shares the one GPU of the box
    rdv = HostRendezvous(rank, world, addr="127.0.0.1", port=port, timeout=120.0)
    comm = Comm(rank, world, rendezvous=rdv)        # smm_comm_unique_id on rank 0, smm_comm_create everywhere
    seen = {"rank": rank, "world": world}

    # ---- Comm.gather / Comm.allgather: rank r's block lands at index r ------------------------------------------
    def block(r, dtype):
        return (np.arange(5 * 301, dtype=np.float64).reshape(5, 301) * 0.25 + 1000.0 * r).astype(dtype)

    for dtype in (np.float64, np.float32):
        shard = to_device(block(rank, dtype))
        got = comm.gather(shard, root=world - 1)    # a root other than 0
        if rank == world - 1:
            host = got.to_host()
            assert host.shape == (world, 5, 301)
            for r in range(world):
                assert np.array_equal(host[r], block(r, dtype)), ("gather", dtype, r)
        else:
            assert got is None
        every = comm.allgather(shard).to_host()
        for r in range(world):
            assert np.array_equal(every[r], block(r, dtype)), ("allgather", dtype, r)
    seen["gather_allgather"] = "ok"

    # ---- regrid_sharded: the HIP operator per rank, shards in HBM, assembled result against the oracle ----------
    rng = np.random.default_rng(20260723)
    w = gridgen.conservative_weights("r96x48", "r36x18")
    n_src, n_dst = 96 * 48, 36 * 18
    op = SparseOperator(n_src, n_dst, w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values,
                        device=0)
    csr = oracle.coo_to_csr_c(n_src, n_dst, w["src_address"].values, w["dst_address"].values,
                              w["remap_matrix"].values)
    n_rows = 10                                      # 2 ranks: 5 + 5; 3 ranks: 4 + 4 + 2; 4 ranks: 3 + 3 + 3 + 1 (a short last shard)
    x = 250.0 + 30.0 * rng.standard_normal((n_rows, n_src))
    x[3, :40] = np.nan
    ref = oracle.apply_c(csr, x)
    kinds = []

    def apply_fn(rows, out):
        kinds.append(type(out).__name__)
        op.apply(to_device(rows), y=out)

    got = regrid_sharded(x, apply_fn, n_dst, comm, gather="root")
    assert (got is None) == (rank != 0)
    if rank == 0:
        assert np.array_equal(got, ref, equal_nan=True), "regrid_sharded(gather='root') differs from the oracle"
    got = regrid_sharded(x, apply_fn, n_dst, comm, gather="all")
    assert np.array_equal(got, ref, equal_nan=True), "regrid_sharded(gather='all') differs from the oracle"
    mine = regrid_sharded(x, apply_fn, n_dst, comm, gather="none")
    lo, hi = shard_bounds(n_rows, world, rank)
    assert isinstance(mine, DeviceArray) and np.array_equal(mine.to_host(), ref[lo:hi], equal_nan=True)
    assert set(kinds) == {"DeviceArray"}
    seen["regrid_sharded"] = "ok"

    # ---- TiledRingGather: tiles that do not divide the rows, 2 slots reused within and across steps --------------
    rows, tiles, slots, steps = 7, 5, 2, 2
    bounds = tile_bounds(rows, tiles)                # ceil(7 / 5) = 2 rows per tile: 4 tiles, the last one short
    assert [b - a for a, b in bounds] == [2, 2, 2, 1]

    def x_of(r, step):                               # rank r's field at a step: every rank can rebuild every other's
        g = np.random.default_rng(1000 * step + r)
        return 250.0 + 30.0 * g.standard_normal((rows, n_src))

    shard = DeviceArray((rows, n_dst), np.float64)
    delivered, errors = [], []
    step_now = [0]

    def on_tile(k, parts):
        r0, r1 = bounds[k]
        delivered.append(k)
        if len(parts) != world:
            errors.append(f"tile {k}: {len(parts)} parts")
        for r, part in enumerate(parts):
            want = oracle.apply_c(csr, x_of(r, step_now[0])[r0:r1])
            if part.shape != (r1 - r0, n_dst) or not np.array_equal(part.to_host(), want):
                errors.append(f"step {step_now[0]} tile {k} rank {r}: wrong rows in the ring slot")

    ring = TiledRingGather(comm, shard, root=0, tiles=tiles, slots=slots, on_tile=on_tile)
    assert ring.tiles == bounds
    for step in range(steps):
        step_now[0] = step
        xd = to_device(x_of(rank, step))
        for k, (r0, r1) in enumerate(bounds):
            op.apply(xd.rows(r0, r1), y=shard.rows(r0, r1))     # tile k's kernel on the null stream ...
            ring.gather_tile(k)                                  # ... its gather behind it on the communication stream
            assert len(ring.pending) <= slots
        ring.finish()
        assert not ring.pending
        synchronize()
        xd.free()
    assert errors == [], errors
    assert ring.delivered == steps * len(bounds)
    if rank == 0:
        assert delivered == list(range(len(bounds))) * steps
        assert ring.gathered_bytes == steps * (world - 1) * rows * n_dst * 8
        assert len(ring.ring) == slots and ring.ring[0].shape == (world, 2, n_dst)    # tile-sized slots, not the full Y
    else:
        assert delivered == [] and ring.gathered_bytes == 0 and ring.ring is None
    seen["ring"] = {"delivered": delivered, "gathered_bytes": ring.gathered_bytes, "tiles": len(bounds)}

    rdv.barrier()
    comm.close()
    rdv.close()
    with open(out_path, "w") as f:
        json.dump(seen, f)
    print(f"stand-in rank {rank} of {world}: ok", flush=True)


if __name__ == "__main__":
    main()
