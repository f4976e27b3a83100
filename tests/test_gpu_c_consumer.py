"""The C ABI used from plain C (no Python, no HIP headers): tests/cpp/abi_smoke.c."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_program_regrids_through_the_abi(hip, tmp_path):
    exe = os.path.join(tmp_path, "abi_smoke")
    lib = os.path.join(ROOT, "smmregrid_amd", "libsmmregrid_hip.so")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "abi_smoke.c"), "-o", exe, lib, "-lm",
                           "-Wl,-rpath," + os.path.dirname(lib)])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi-smoke mismatches=0" in out.stdout
    assert "src_address[0]=4097" in out.stdout          # the refused link is named in the message
