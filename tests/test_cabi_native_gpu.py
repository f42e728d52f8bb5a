"""The C ABI used from plain C (no Python, no torch in the process): tests/native/cabi_check.c is compiled against
include/dasv_hip.h, linked with libdasv_hip.so and the plain-C oracle, and run on the GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_plain_c_program_drives_the_library(tmp_path):
    gcc = shutil.which("gcc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if gcc is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("no C compiler / HIP headers on this box")
    lib_dir = os.path.join(ROOT, "doubleattentionspeakerverification_amd", "lib")
    assert os.path.exists(os.path.join(lib_dir, "libdasv_hip.so")), "run __graft_entry__.build() first"
    exe = str(tmp_path / "cabi_check")
    subprocess.check_call([gcc, "-std=c99", "-O2", os.path.join(ROOT, "tests", "native", "cabi_check.c"),
                           os.path.join(ROOT, "oracle", "dmha_oracle.c"), "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(rocm, "include"), "-D__HIP_PLATFORM_AMD__", "-L", lib_dir,
                           "-L", os.path.join(rocm, "lib"), "-ldasv_hip", "-lamdhip64", "-lm", "-Wl,-rpath," + lib_dir,
                           "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", exe])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(res.stdout, res.stderr)
    assert res.returncode == 0 and "CABI_NATIVE_OK" in res.stdout
