"""The compiled registration object libtorch_hrt.so (csrc/torch_export.cpp): the reference's way of binding its kernels
(torch.ops.load_library, hrt/python/kernels/__init__.py:4-16) served without this package's Python."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "het_amd", "libtorch_hrt.so")
pytestmark = pytest.mark.gpu


def _need_lib():
    if not os.path.exists(LIB):  # (optional target: __graft_entry__.build() warns when it could not be built)
        pytest.skip("libtorch_hrt.so not built: make -C het_amd/csrc torch_hrt")


def test_reference_rgat_sequence_on_the_compiled_registration_alone():
    """A fresh interpreter loads ONLY libtorch_hrt.so (het_amd is never imported) and runs the reference's RGAT op sequence,
    forward and backward, against the fp64 oracle layer."""
    _need_lib()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "capi", "torch_hrt_sequence.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "TORCH_HRT_SEQUENCE_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_every_compiled_op_against_the_oracle():
    """The op-level parity tests of tests/test_gpu_ops.py with K = the compiled registration (HET_TORCH_HRT_LIB: het_amd.kernels then
    defines no op of its own): every reference-named op of libtorch_hrt.so against the fp64 oracle, grouped and atomics modes."""
    _need_lib()
    env = dict(os.environ, HET_TORCH_HRT_LIB=LIB)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_ops.py"), "-x", "-q", "-m", "gpu", "-k",
                        "not node_backward and not duplicate"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
