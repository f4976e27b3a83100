"""The C-ABI library builds, loads on a GPU-less host and exports every symbol
include/smmregrid_amd.h declares.  No compute calls here."""
import ctypes
import os
import re

import pytest

from smmregrid_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "smmregrid_amd.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smm_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_surface():
    syms = declared_symbols()
    for name in ("smm_operator_create", "smm_apply", "smm_group_apply", "smm_operator_mask_apply",
                 "smm_operator_export_csr", "smm_last_error"):
        assert name in syms


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_ctypes_table_matches_header():
    table = set(_lib.SIGNATURES) | set(_lib.SPECIAL)
    assert table == set(declared_symbols())


def test_tuning_knobs_match_the_header_and_are_not_apply_flags():
    """ABI v5: the tuning knobs are named entries of smm_debug_set_tuning's enum, in the order of _lib.TUNE_KNOBS;
    the apply flags carry no variant / walk-length bit fields any more."""
    text = open(HEADER).read()
    names = re.findall(r"\b(SMM_TUNE_[A-Z_]+)\b\s*(?:=\s*0)?\s*,", text)
    assert [n[len("SMM_TUNE_"):].lower() for n in names] == list(_lib.TUNE_KNOBS)
    assert "SMM_APPLY_VARIANT_SHIFT" not in text and "SMM_APPLY_JPB_SHIFT" not in text
    lib = _lib.load()
    prev = ctypes.c_int(-7)
    assert lib.smm_debug_set_tuning(_lib.TUNE["tile_walk"], 5, ctypes.byref(prev)) == 0 and prev.value == 0
    assert lib.smm_debug_set_tuning(_lib.TUNE["tile_walk"], 0, ctypes.byref(prev)) == 0 and prev.value == 5
    assert lib.smm_debug_set_tuning(len(_lib.TUNE_KNOBS), 1, None) == _lib.SMM_ERR_INVALID
    with pytest.raises(KeyError):
        _lib.tuning(no_such_knob=1)


def test_abi_version_and_error_string():
    lib = _lib.load()
    assert lib.smm_abi_version() == 6
    assert isinstance(lib.smm_last_error(), (bytes, type(None)))


def test_no_cpu_fallback_without_device():
    """On a host without a GPU every compute entry fails loudly with NO_DEVICE."""
    if _lib.device_count() > 0:
        pytest.skip("a device is present")
    import numpy as np
    from smmregrid_amd import SparseOperator
    with pytest.raises(_lib.SmmNoDeviceError):
        SparseOperator(4, 4, np.array([1], np.int32), np.array([1], np.int32), np.array([1.0]),
                       device=0)


def test_missing_rccl_is_unsupported_not_a_crash():
    """smm_comm_* with no loadable librccl (SMM_RCCL_LIB points nowhere) returns SMM_ERR_UNSUPPORTED
    with the loader's message (round 1 called dlerror() twice and dereferenced NULL)."""
    import subprocess
    import sys
    code = (
        "import ctypes, sys\n"
        "sys.path.insert(0, %r)\n"
        "from smmregrid_amd import _lib\n"
        "lib = _lib.load()\n"
        "buf = ctypes.create_string_buffer(128)\n"
        "rc = lib.smm_comm_unique_id(buf)\n"
        "h = ctypes.c_void_p()\n"
        "rc2 = lib.smm_comm_create(buf, 1, 0, ctypes.byref(h))\n"
        "print(rc, rc2, lib.smm_last_error().decode())\n" % ROOT)
    env = dict(os.environ, SMM_RCCL_LIB="/nonexistent/librccl.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rc, rc2, msg = out.stdout.strip().split(" ", 2)
    assert int(rc) == _lib.SMM_ERR_UNSUPPORTED and int(rc2) == _lib.SMM_ERR_UNSUPPORTED
    assert "cannot load librccl" in msg and "nonexistent" in msg


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: no module of the product package, and no native source of the
    library, may import, link or call anything under oracle/ (only tests/, smoke() and bench.py's
    cpu_baseline leg do)."""
    import ast
    pkg = os.path.join(ROOT, "smmregrid_amd")
    for dirpath, _, files in os.walk(pkg):
        for name in files:
            path = os.path.join(dirpath, name)
            if name.endswith(".py"):
                tree = ast.parse(open(path).read())
                for node in ast.walk(tree):
                    mods = []
                    if isinstance(node, ast.Import):
                        mods = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom):
                        mods = [node.module or ""]
                    assert not any(m == "oracle" or m.startswith("oracle.") for m in mods), path
            elif name.endswith((".hip", ".cpp", ".hpp", ".h")) or name == "Makefile":
                text = open(path).read()     # comments may name the oracle; includes and link lines may not
                assert not re.search(r'#\s*include[^\n]*oracle|liboracle|-loracle|oracle\.(c|so)\b', text), path
    lib = ctypes.CDLL(_lib.LIB_PATH) if os.path.exists(_lib.LIB_PATH) else None
    if lib is not None:
        assert not hasattr(lib, "oracle_apply")


def test_process_wide_setters_need_no_device():
    """smm_set_host_threads / smm_debug_set_grid_limit are host-side state: they answer without a GPU, hand back
    the previous setting, and refuse negative values with SMM_ERR_INVALID (ABI v4)."""
    prev = ctypes.c_int(-1)
    _lib.call("smm_set_host_threads", 5, ctypes.byref(prev))
    first = prev.value
    _lib.call("smm_set_host_threads", first, ctypes.byref(prev))
    assert prev.value == 5 and first >= 0
    _lib.call("smm_set_host_threads", first, None)          # the out pointer may be NULL
    with pytest.raises(_lib.SmmError) as err:
        _lib.call("smm_set_host_threads", -1, None)
    assert err.value.code == _lib.SMM_ERR_INVALID
    _lib.call("smm_debug_set_grid_limit", 12345)
    _lib.call("smm_debug_set_grid_limit", 0)
    with pytest.raises(_lib.SmmError):
        _lib.call("smm_debug_set_grid_limit", -3)


def test_host_memcpy_on_the_staging_pool_survives_a_fork():
    """smm_host_memcpy (no GPU involved): the parallel copy of the host pipelines' staging pool.  The pool is per
    process -- a child of fork() has none of the parent's worker threads and must start its own instead of waiting
    for threads that do not exist; both processes exit cleanly (the workers are joined at exit)."""
    import numpy as np
    lib = _lib.load()
    prev = ctypes.c_int(0)
    assert lib.smm_set_host_threads(4, ctypes.byref(prev)) == 0
    try:
        src = np.arange(12 << 20, dtype=np.float64)          # 96 MiB: above the single-thread threshold
        dst = np.zeros_like(src)
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        assert lib.smm_host_memcpy(ptr(dst), ptr(src), src.nbytes) == 0 and np.array_equal(dst, src)
        assert lib.smm_host_memcpy(ptr(dst), None, 8) == _lib.SMM_ERR_INVALID
        assert lib.smm_host_memcpy(None, None, 0) == 0
        pid = os.fork()
        if pid == 0:                                         # the child: a pool of its own
            code = 1
            try:
                dst[:] = 0.0
                ok = lib.smm_host_memcpy(ptr(dst), ptr(src), src.nbytes) == 0 and np.array_equal(dst, src)
                code = 0 if ok else 2
            finally:
                os._exit(code)
        _, status = os.waitpid(pid, 0)
        assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
        dst[:] = 0.0                                         # the parent's pool is untouched by the fork
        assert lib.smm_host_memcpy(ptr(dst), ptr(src), src.nbytes) == 0 and np.array_equal(dst, src)
    finally:
        lib.smm_set_host_threads(prev.value, None)
