"""INTEGRATION.md is the reference-side binding a maintainer of jhardenberg/smmregrid would add.  The document's own
code blocks are executed here as written: without a GPU the ctypes prototypes they declare are held against the
binding this package ships (`_lib.SIGNATURES`); on the GPU box the stub builds an operator from a weights object
(`weights.py:25-44`), runs the fill + product + three `where`s of `regrid.py:545-570` through `smm_apply_host`, and the
result is compared with the oracle bit for bit."""
import ctypes
import os
import re
import types

import numpy as np
import pytest

from smmregrid_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOC = os.path.join(ROOT, "INTEGRATION.md")


def python_blocks():
    text = open(DOC).read()
    return re.findall(r"```python\n(.*?)```", text, flags=re.S)


def stub_namespace():
    """The loader stub of section 1, executed with the in-tree library path."""
    blocks = python_blocks()
    assert len(blocks) >= 3 and "ctypes.CDLL" in blocks[0]
    ns = {}
    exec(blocks[0].replace('ctypes.CDLL("libsmmregrid_hip.so")', f"ctypes.CDLL({_lib.LIB_PATH!r})"), ns)
    return ns, blocks


def test_the_documented_prototypes_match_the_shipped_binding():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    ns, blocks = stub_namespace()
    lib = ns["_lib"]
    declared = re.findall(r"_lib\.(smm_[a-z0-9_]+)\.argtypes", blocks[0] + blocks[2])
    assert {"smm_operator_create", "smm_operator_set_epilogue", "smm_operator_mask_apply", "smm_apply", "smm_apply_host"} <= set(declared)
    exec(re.search(r"^_lib\.smm_apply_host\.argtypes = .*$", blocks[2], flags=re.M).group(0), ns)
    for name in declared:
        got = getattr(lib, name).argtypes
        want = _lib.SIGNATURES[name]
        assert len(got) == len(want), name
        for a, b in zip(got, want):
            # the document writes POINTER(c_void_p) where the binding has its alias; sizes must agree
            assert ctypes.sizeof(a) == ctypes.sizeof(b), (name, a, b)
    assert ns["_check"](0) is None
    with pytest.raises(RuntimeError):
        ns["_check"](1)


@pytest.mark.gpu
def test_the_documented_stub_regrids_like_the_oracle(hip, rng):
    from oracle import oracle
    from smmregrid_amd import gridgen
    ns, blocks = stub_namespace()
    exec(blocks[1], ns)                                           # def compute_weights_matrix(weights, device=0)
    w = gridgen.conservative_weights("r96x48", "r36x18")
    weights = types.SimpleNamespace(src_address=w["src_address"], dst_address=w["dst_address"], remap_matrix=w["remap_matrix"],
                                    sizes=w.sizes)
    h = ns["compute_weights_matrix"](weights)
    S, D = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
    imask = np.ascontiguousarray(w["dst_grid_imask"].values, dtype=np.int32)
    frac = np.ascontiguousarray(w["dst_grid_frac"].values, dtype=np.float64)
    ns["_check"](ns["_lib"].smm_operator_set_epilogue(h, ns["_ptr"](imask), ns["_ptr"](frac)))      # regrid.py:198-203
    x = (250.0 + 30.0 * rng.standard_normal((5, 3, 48, 96))).astype(np.float32)
    x[1, 2, 10:20, 30:60] = np.nan
    ns.update(h=h, S=S, D=D, masked=True, source_array=x, kept_shape=(5, 3), tgt_shape=(18, 36),
              self=types.SimpleNamespace(remap_area_min=0.5))
    exec(blocks[2], ns)                                           # the body of apply_weights around regrid.py:545-570
    target = ns["target"]
    assert target.shape == (5, 3, 18, 36) and target.dtype == np.float64
    csr = oracle.coo_to_csr_c(S, D, w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values)
    ref = oracle.apply_c(csr, x.reshape(-1, S), True, imask, frac, 0.5).reshape(target.shape)
    assert np.array_equal(np.isnan(target), np.isnan(ref)) and np.array_equal(target[~np.isnan(ref)], ref[~np.isnan(ref)])
    _lib.call("smm_operator_destroy", h)
