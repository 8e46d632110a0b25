"""The drop-in boundary is a plain C ABI: a C++ program with raw HIP buffers and no PyTorch, compiled against
include/het_amd.h and linked to het_amd/libhet_amd.so, runs the segment GEMM forward/backward and checks them against
host loops (tests/capi/standalone.cpp)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_without_torch(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "standalone")
    lib_dir = os.path.join(ROOT, "het_amd")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "capi", "standalone.cpp"), "-L", lib_dir, "-lhet_amd",
                    "-Wl,-rpath," + lib_dir, "-o", exe], check=True, timeout=600)
    out = subprocess.run([exe], check=False, capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "CAPI STANDALONE OK" in out.stdout
