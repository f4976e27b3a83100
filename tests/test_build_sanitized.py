"""The host-side operator builder (csrc/smm_build.cpp) and the host pipelines' staging pool (csrc/smm_hostpool.cpp)
under AddressSanitizer + UBSan -- and once under ThreadSanitizer -- on the CPU: canonical CSR bit-exact against
scipy, SELL-64 and tile-plan invariants, error paths, injected thread-start and task failures."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle
from tests.helpers import ragged_links, random_links

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "build_harness_asan")
SOURCES = [os.path.join(ROOT, "tests", "cpp", "build_harness.cpp"),
           os.path.join(ROOT, "smmregrid_amd", "csrc", "smm_build.cpp"),
           os.path.join(ROOT, "smmregrid_amd", "csrc", "smm_hostpool.cpp")]


@pytest.fixture(scope="module")
def harness():
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-o", EXE] + SOURCES)
    yield EXE
    os.remove(EXE)


def run(exe, n_src, n_dst, src, dst, w, threads=1):
    text = f"{n_src} {n_dst} {len(src)}\n" + "".join(f"{int(s)} {int(d)} {float(v)!r}\n" for s, d, v in zip(src, dst, w))
    out = subprocess.run([exe, str(threads)], input=text, capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout.splitlines()


@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("case", ["random", "ragged", "dups", "cdo_order", "empty", "one_long_row", "tail_cols",
                                  "few_wide_blocks"])
def test_builder_under_sanitizers(harness, rng, case, threads):
    """`threads` = host threads of the builders (smm_set_host_threads): the threaded sort / duplicate sum, SELL
    fill and tile-plan construction give the same bits as one thread; "cdo_order" = links already ordered by
    (dst, src) with their duplicates adjacent (what `cdo gen*` writes: the builder's sort-free path)."""
    if case == "random":
        n_src, n_dst = 5000, 1300
        src, dst, w = random_links(rng, n_src, n_dst, 9000)
    elif case == "ragged":
        n_src, n_dst = 3000, 777
        src, dst, w = ragged_links(rng, n_src, n_dst, max_len=60)
    elif case == "dups":
        n_src, n_dst = 40, 130
        src, dst, w = random_links(rng, n_src, n_dst, 4000, dup_frac=0.5)
    elif case == "cdo_order":
        n_src, n_dst = 700, 400
        src, dst, w = random_links(rng, n_src, n_dst, 6000, dup_frac=0.3)
        order = np.lexsort((src, dst))
        src, dst, w = src[order], dst[order], w[order]
    elif case == "empty":
        n_src, n_dst = 10, 5
        src, dst, w = np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0)
    elif case == "one_long_row":
        n_src, n_dst = 20000, 70
        src = np.concatenate([rng.permutation(n_src)[:9000] + 1, [1, 2]]).astype(np.int32)
        dst = np.concatenate([np.full(9000, 65), [1, 70]]).astype(np.int32)
        w = rng.random(src.size)
    elif case == "few_wide_blocks":
        # narrow banded blocks plus one block spread over ~300 chunks: tighten_tile_plan demotes it
        n_dst, n_src = 256 * 120, 256 * 120 * 2 + 8
        d = np.repeat(np.arange(n_dst), 4)
        s_ = d * 2 + np.tile(np.arange(4), n_dst)
        wide = (d >= 256 * 50) & (d < 256 * 51)
        s_[wide] = 256 * 50 * 2 + rng.integers(0, 4800, size=int(wide.sum()))
        src, dst, w = (s_ + 1).astype(np.int32), (d + 1).astype(np.int32), rng.random(d.size)
    else:
        n_src, n_dst = 1006, 64
        src = np.concatenate([np.full(64, 1006), np.full(64, 1005), np.arange(1, 65)]).astype(np.int32)
        dst = np.tile(np.arange(1, 65), 3).astype(np.int32)
        w = rng.random(src.size)
    lines = run(harness, n_src, n_dst, src, dst, w, threads)
    assert lines[0].startswith("CSR")
    nnz, n_used, max_row, n_slots = (int(v) for v in lines[0].split()[1:])
    rowptr, col, val = oracle.coo_to_csr(n_src, n_dst, src, dst, w)
    assert nnz == col.size and n_used == np.unique(col).size
    assert max_row == (np.diff(rowptr).max() if n_dst else 0) and n_slots % 64 == 0
    assert np.array_equal(np.array(lines[1].split(), dtype=np.int64), rowptr)
    assert np.array_equal(np.array(lines[2].split(), dtype=np.int64), col)
    ref_c = oracle.coo_to_csr_c(n_src, n_dst, src, dst, w)[2]
    got = np.array([float(v) for v in lines[3].split()])
    assert np.array_equal(got.view(np.uint64), ref_c.view(np.uint64))
    plans = [ln.split() for ln in lines if ln.startswith("PLAN")]
    assert len(plans) == 3 and all(p[-1] == "0" for p in plans)          # every LDS index resolves
    tight = [ln.split() for ln in lines if ln.startswith("TIGHT")]
    assert len(tight) == 3 and all(t[-1] == "0" for t in tight)          # ... also after tighten_tile_plan
    assert lines[-1] == "SELLBAD 0"
    if case == "few_wide_blocks":
        t4 = [t for t in tight if t[1] == "4"][0]
        assert int(t4[2]) == 64 and int(t4[3]) <= 34 and 1000 <= int(t4[4]) <= 1024   # budget 512 -> 64, one block direct
    assert "ADOPTBAD 0" in lines            # smm_operator_create_csr's validator (adopt_csr)
    assert "SPLITBAD 0" in lines            # launch grids beyond the limit are cut into parts (smm::split_batch)
    fault = [ln.split() for ln in lines if ln.startswith("FAULTBAD")][0]
    # a throwing worker task surfaces on the caller (no std::terminate), unstartable threads only cost parallelism
    assert fault[1] == "0" and (int(fault[2]) > 0 or threads == 1)
    assert "CHUNKBAD 0" in lines            # host pipelines size their chunks from X AND Y bytes (U << D)
    prune = [ln.split() for ln in lines if ln.startswith("PRUNEBAD")][0]
    assert prune[1] == "0" and int(prune[2]) == int((val == 0.0).sum())     # exact-zero links dropped, rest intact
    check_pool_line(lines)


def check_pool_line(lines):
    """POOLBAD <mismatches> <statuses from injected task failures> <workers after the first big pack> <usable cpus>:
    the staging pool packs the same block whatever the thread count / store kind, an injected bad_alloc in a task
    comes back as status 1 (both with workers and with thread start refused) and never aborts, the workers
    persist between calls (at most threads - 1 of them)."""
    pool = [ln.split() for ln in lines if ln.startswith("POOLBAD")][0]
    assert pool[1] == "0", pool
    assert int(pool[2]) >= 2 and 0 <= int(pool[3]) <= 5 and int(pool[4]) >= 1


def test_staging_pool_under_thread_sanitizer(rng, tmp_path):
    """The same harness under -fsanitize=thread: the pool's job hand-over (generation counter, claim counter,
    completion wait) and the builders' worker pools are race-free."""
    exe = str(tmp_path / "build_harness_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=thread", "-o", exe] + SOURCES)
    n_src, n_dst = 900, 300
    src, dst, w = random_links(rng, n_src, n_dst, 2500)
    text = f"{n_src} {n_dst} {len(src)}\n" + "".join(f"{int(s)} {int(d)} {float(v)!r}\n" for s, d, v in zip(src, dst, w))
    out = subprocess.run([exe, "3"], input=text, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"))
    assert out.returncode == 0, out.stderr[-3000:]
    assert "ThreadSanitizer" not in out.stderr
    check_pool_line(out.stdout.splitlines())


def test_builder_rejects_bad_addresses(harness):
    lines = run(harness, 3, 2, [4], [1], [1.0])
    assert lines[0].startswith("ERROR src_address")
    lines = run(harness, 3, 2, [1], [0], [1.0])
    assert lines[0].startswith("ERROR dst_address")
