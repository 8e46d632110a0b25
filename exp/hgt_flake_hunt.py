"""Repeats tests/test_gpu_layers.py::_run_hgt_fused (the exact test body: CPU oracle, graph and layer moved to the GPU, one cold
forward + backward) over the parameter grid many times in one process; prints every assertion failure.  ROUNDS=<n>"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _pytest.monkeypatch import MonkeyPatch
from tests.test_gpu_layers import _run_hgt_fused
fails = 0
n = 0
for rnd in range(int(os.environ.get("ROUNDS", "12"))):
    for H in (8, 1, 4):
        for fused_attn, compact_dst in ((False, True), (True, True), (False, False)):
            mp = MonkeyPatch()
            n += 1
            try:
                _run_hgt_fused(fused_attn, compact_dst, H, 64, 64, mp)
            except AssertionError as ex:
                fails += 1
                print(f"round {rnd} H={H} fused_attn={fused_attn} compact_dst={compact_dst}: {str(ex)[:400]}", flush=True)
            finally:
                mp.undo()
print(f"{fails} failures in {n} runs")
