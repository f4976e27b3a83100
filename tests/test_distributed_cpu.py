"""N>1 path on CPU: world_size-2 process groups, batch-row sharding + gather.
The communicator is injected (smmregrid_amd.distributed imports neither torch nor the HIP library):
here a gloo adapter (tools/torch_comm.py) with the CPU oracle as the per-rank compute; on the GPU
box the same code runs with the RCCL communicator and the HIP operator (tests/test_gpu_facade.py).
The host-side control plane (smmregrid_amd.comm.HostRendezvous: barrier, max over ranks, hand-over
of the RCCL unique id) is plain TCP and runs here as it does on the GPU box."""
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
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from smmregrid_amd import gridgen
    from smmregrid_amd.distributed import regrid_sharded
    from tools.torch_comm import TorchComm
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = TorchComm()
        w = gridgen.conservative_weights("r48x24", "r16x8")
        n_src, n_dst = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
        csr = oracle.coo_to_csr_c(n_src, n_dst, w["src_address"].values, w["dst_address"].values,
                                  w["remap_matrix"].values)
        rng = np.random.default_rng(20260723)
        x = 250.0 + 30.0 * rng.standard_normal((n_rows, n_src))
        x[n_rows // 2, :50] = np.nan
        seen = []

        def apply_fn(rows, out):
            seen.append(rows.shape[0])
            assert tuple(out.shape) == (rows.shape[0], n_dst)
            out.copy_(torch.from_numpy(oracle.apply_c(csr, rows)))

        out = regrid_sharded(x, apply_fn, n_dst, comm, gather=gather)
        if gather == "none":
            out = out.numpy()
        seen = seen or [0]
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
    from tools.torch_comm import TorchComm
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
        ring = TiledRingGather(TorchComm(), shard, root=0, tiles=tiles, slots=slots, on_tile=on_tile)
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
                     None if ring.ring is None else (len(ring.ring), ring.ring[0].shape[1]))
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


# ---------------------------------------------------------------- host-side control plane (no torch)

def _rdv_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    from smmregrid_amd import comm as smm_comm
    # the RCCL id itself needs librccl + a GPU: stub the one call that creates it
    smm_comm.unique_id = lambda: bytes((7 * i + 3) % 256 for i in range(smm_comm.ID_BYTES))
    rdv = smm_comm.HostRendezvous(rank, world, addr="127.0.0.1", port=port, timeout=60.0)
    try:
        cid = smm_comm.exchange_id(rank, world, rendezvous=rdv)
        rdv.barrier()
        top = rdv.max(10.0 + rank)
        parts = rdv.allgather(f"rank{rank}".encode() * (rank + 1))
        ret[rank] = (cid, top, parts, rdv.n_connected)
    finally:
        rdv.close()


@pytest.mark.parametrize("world", [2, 3])
def test_host_rendezvous_and_id_exchange(world):
    """comm.exchange_id over TCP with smm_comm_unique_id stubbed: every rank ends up with the same
    128 bytes; barrier / max / allgather of the control plane with 2 and 3 processes."""
    import multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_rdv_worker, args=(r, world, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        ids = {ret[r][0] for r in range(world)}
        assert len(ids) == 1 and len(next(iter(ids))) == 128
        for r in range(world):
            assert ret[r][1] == 10.0 + world - 1
            assert ret[r][2] == [f"rank{q}".encode() * (q + 1) for q in range(world)]
        assert ret[0][3] == world


def test_exchange_id_opens_its_own_rendezvous():
    """Without a rendezvous object exchange_id opens one on MASTER_ADDR / MASTER_PORT + 23."""
    import multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_own_rdv_worker, args=(r, 2, port, ret)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert ret[0] == ret[1] and len(ret[0]) == 128


def _own_rdv_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from smmregrid_amd import comm as smm_comm
    smm_comm.unique_id = lambda: bytes(range(128))
    ret[rank] = smm_comm.exchange_id(rank, world, timeout=60.0)


def test_rendezvous_missing_rank_times_out():
    from smmregrid_amd.comm import HostRendezvous
    with pytest.raises(TimeoutError):
        HostRendezvous(0, 2, addr="127.0.0.1", port=_free_port(), timeout=0.5)


def test_product_package_does_not_import_torch():
    """smmregrid_amd (distributed.py and comm.py included) never imports torch: the multi-process
    plumbing of the product is the C ABI's RCCL communicator plus plain sockets."""
    import subprocess
    code = ("import sys; import smmregrid_amd, smmregrid_amd.distributed, smmregrid_amd.comm; "
            "assert 'torch' not in sys.modules, 'torch was imported'; print('ok')")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
    import re
    pkg = os.path.join(ROOT, "smmregrid_amd")
    for name in os.listdir(pkg):
        if name.endswith(".py"):
            text = open(os.path.join(pkg, name)).read()
            assert not re.search(r"^\s*(import torch|from torch)", text, re.M), name
