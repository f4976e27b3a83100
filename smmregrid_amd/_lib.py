"""ctypes binding of libsmmregrid_hip.so (C ABI in include/smmregrid_amd.h).

There is no CPU fallback: if the shared library is missing or no HIP device is
present every compute call raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMM_LIB_PATH") or os.path.join(_HERE, "libsmmregrid_hip.so")

SMM_OK = 0
SMM_ERR_INVALID = 1
SMM_ERR_NO_DEVICE = 2
SMM_ERR_HIP = 3
SMM_ERR_ALLOC = 4
SMM_ERR_UNSUPPORTED = 5
SMM_ERR_INTERNAL = 6

SMM_F32 = 0
SMM_F64 = 1

APPLY_MASKED = 1 << 0
APPLY_NO_FILL = 1 << 1
APPLY_SB_PACKED = 1 << 2
APPLY_HOST_NO_PACK = 1 << 3
APPLY_SB_Y_SB = 1 << 4
CREATE_PRUNE_ZEROS = 1 << 0
APPLY_KERNEL_SELL = 1 << 8
APPLY_KERNEL_TILE = 1 << 9

# smm_debug_set_tuning knobs (tests, tools, benchmarks; the results do not depend on them)
TUNE_KNOBS = ("sell_batch_rows", "tile_walk", "tile_staging", "tile_rows_per_step", "tile_x_loads",
              "tile_split_rows", "tile_links", "xcd_run", "sb_strip", "sb_loads", "sb_level_launches", "sb_lds_pad",
              "host_pack_stores", "host_chunk_kb")
TUNE = {name: i for i, name in enumerate(TUNE_KNOBS)}
STAGING_REGISTERS, STAGING_DMA = 1, 2


class SmmError(RuntimeError):
    """A libsmmregrid_hip call failed (status code in .code)."""

    def __init__(self, code, message):
        super().__init__(f"libsmmregrid_hip error {code}: {message}")
        self.code = code


class SmmNoDeviceError(SmmError):
    """No usable HIP device: the HIP path is mandatory, nothing falls back to the CPU."""


_i64 = ctypes.c_int64
_p = ctypes.c_void_p
_pp = ctypes.POINTER(ctypes.c_void_p)
_int = ctypes.c_int
_dbl = ctypes.c_double
_uint = ctypes.c_uint
_size = ctypes.c_size_t

# name -> argtypes; every entry point returns int status except the two noted
SIGNATURES = {
    "smm_device_count": [ctypes.POINTER(_int)],
    "smm_set_device": [_int],
    "smm_get_device": [ctypes.POINTER(_int)],
    "smm_device_name": [_int, ctypes.c_char_p, _size],
    "smm_mem_info": [ctypes.POINTER(_size), ctypes.POINTER(_size)],
    "smm_malloc": [_pp, _size],
    "smm_free": [_p],
    "smm_host_alloc": [_pp, _size],
    "smm_host_free": [_p],
    "smm_host_memcpy": [_p, _p, _size],
    "smm_memcpy_h2d": [_p, _p, _size, _p],
    "smm_memcpy_d2h": [_p, _p, _size, _p],
    "smm_memcpy_d2d": [_p, _p, _size, _p],
    "smm_memcpy2d_h2d": [_p, _size, _p, _size, _size, _size, _p],
    "smm_memcpy2d_d2h": [_p, _size, _p, _size, _size, _size, _p],
    "smm_memset": [_p, _int, _size, _p],
    "smm_stream_create": [_pp],
    "smm_stream_destroy": [_p],
    "smm_stream_sync": [_p],
    "smm_device_sync": [],
    "smm_event_create": [_pp],
    "smm_event_destroy": [_p],
    "smm_event_record": [_p, _p],
    "smm_event_sync": [_p],
    "smm_stream_wait_event": [_p, _p],
    "smm_event_elapsed_ms": [_p, _p, ctypes.POINTER(ctypes.c_float)],
    "smm_fill_random": [_p, _int, _i64, ctypes.c_uint64, _dbl, _dbl, _p],
    "smm_operator_create": [_i64, _i64, _i64, _p, _p, _p, _int, _pp],
    "smm_operator_create_csr": [_i64, _i64, _p, _p, _p, _int, _pp],
    "smm_operator_create_opt": [_i64, _i64, _i64, _p, _p, _p, _uint, _int, _pp],
    "smm_operator_destroy": [_p],
    "smm_operator_info": [_p] + [ctypes.POINTER(_i64)] * 5,
    "smm_operator_export_csr": [_p, _p, _p, _p],
    "smm_operator_set_epilogue": [_p, _p, _p],
    "smm_operator_mask_apply": [_p, _p, _p],
    "smm_operator_plan_info": [_p, ctypes.POINTER(_int), ctypes.POINTER(_i64), ctypes.POINTER(_i64)],
    "smm_operator_launch_info": [_p, _int, _i64, _uint, ctypes.POINTER(_int), ctypes.POINTER(_int),
                                 ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_i64),
                                 ctypes.POINTER(_i64), ctypes.POINTER(_int)],
    "smm_group_create": [_pp, _int, _pp],
    "smm_group_prepare": [_p, _i64, _p, _p],
    "smm_group_launch_info": [_p, _int, _i64, _i64, _i64, _uint, ctypes.POINTER(_int), ctypes.POINTER(_int),
                              ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_i64),
                              ctypes.POINTER(_i64), ctypes.POINTER(_int)],
    "smm_group_destroy": [_p],
    "smm_group_plan_info": [_p, ctypes.POINTER(_int), ctypes.POINTER(_int)],
    "smm_apply": [_p, _p, _int, _i64, _p, _int, _i64, _i64, _dbl, _uint, _p],
    "smm_operator_prepare_sb": [_p],
    "smm_operator_used_sources": [_p, _p],
    "smm_apply_sb": [_p, _p, _int, _i64, _p, _int, _i64, _i64, _dbl, _uint, _p],
    "smm_apply_host": [_p, _p, _int, _i64, _p, _int, _i64, _i64, _dbl, _uint, _i64],
    "smm_group_apply": [_p, _p, _int, _i64, _i64, _i64, _p, _int, _i64, _i64, _i64,
                        _i64, _i64, _i64, _p, _p, _dbl, _uint, _p],
    "smm_group_prepare_sb": [_p],
    "smm_group_apply_sb": [_p, _p, _int, _i64, _i64, _p, _int, _i64, _i64, _i64, _i64, _p, _p, _dbl, _uint, _p],
    "smm_group_apply_host": [_p, _p, _int, _p, _int, _i64, _i64, _i64, _int, _p, _p, _dbl, _uint, _i64],
    "smm_debug_fail_at_chunk": [_i64],
    "smm_debug_host_stats": [ctypes.POINTER(_dbl), _int, _int],
    "smm_debug_staging_faults": [_int, _i64],
    "smm_debug_set_grid_limit": [_i64],
    "smm_debug_set_tuning": [_int, _int, ctypes.POINTER(_int)],
    "smm_set_host_threads": [_int, ctypes.POINTER(_int)],
    "smm_comm_unique_id": [_p],
    "smm_comm_create": [_p, _int, _int, _pp],
    "smm_comm_destroy": [_p],
    "smm_comm_gather": [_p, _p, _p, _i64, _int, _int, _p],
    "smm_comm_allgather": [_p, _p, _p, _i64, _int, _p],
}
SPECIAL = {"smm_abi_version": (_int, []), "smm_last_error": (ctypes.c_char_p, [])}

_lib = None


def build(force=False):
    """Compile the shared library in-tree with hipcc for gfx950."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load():
    """Load the library (once) and declare every prototype.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SmmError(SMM_ERR_UNSUPPORTED,
                       f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "or `make -C smmregrid_amd/csrc`. There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            # timing experiments load older builds of the library side by side (tools/exp/ab_libs.sh)
            if os.environ.get("SMM_LIB_ALLOW_MISSING"):
                continue
            raise SmmError(SMM_ERR_UNSUPPORTED, f"{LIB_PATH} does not export {name}: stale build, run build()")
        fn.restype = _int
        fn.argtypes = argtypes
    for name, (restype, argtypes) in SPECIAL.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status):
    if status == SMM_OK:
        return
    msg = load().smm_last_error()
    msg = msg.decode("utf-8", "replace") if msg else ""
    if status == SMM_ERR_NO_DEVICE:
        raise SmmNoDeviceError(status, msg)
    raise SmmError(status, msg)


def call(name, *args):
    check(getattr(load(), name)(*args))


def device_count():
    """Number of HIP devices; 0 when none (never raises for 'no device')."""
    n = _int(0)
    status = load().smm_device_count(ctypes.byref(n))
    if status == SMM_ERR_NO_DEVICE:
        return 0
    check(status)
    return n.value


HOST_STATS = ("calls", "chunks", "stage_in_ms", "h2d_ms", "kernel_ms", "d2h_ms", "copy_out_ms", "wait_ms", "total_ms",
              "threads")


def host_stats(reset=False):
    """smm_debug_host_stats: where smm_apply_host / smm_group_apply_host spent their time since the last reset."""
    buf = (_dbl * len(HOST_STATS))()
    call("smm_debug_host_stats", buf, len(HOST_STATS), 1 if reset else 0)
    return dict(zip(HOST_STATS, (float(v) for v in buf)))


def set_tuning(knob, value):
    """smm_debug_set_tuning: one named knob (TUNE_KNOBS), 0 = the library's own choice.  Returns the former value."""
    prev = _int(0)
    call("smm_debug_set_tuning", TUNE[knob] if isinstance(knob, str) else int(knob), int(value), ctypes.byref(prev))
    return prev.value


class tuning:
    """`with tuning(tile_staging=STAGING_DMA, tile_rows_per_step=2): ...` -- knobs set for the block, restored after."""

    def __init__(self, **knobs):
        unknown = [k for k in knobs if k not in TUNE]
        if unknown:
            raise KeyError(f"unknown tuning knob(s) {unknown}; known: {TUNE_KNOBS}")
        self.knobs, self.prev = knobs, {}

    def __enter__(self):
        for k, v in self.knobs.items():
            self.prev[k] = set_tuning(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            set_tuning(k, v)
        return False
