import sys, json, io, contextlib
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import het_amd.plan as plan
from het_amd import train
for en in (True, False, True, False):
    plan.enabled = en
    plan.clear()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        train.main(["-d", "mag", "--num_heads", "4", "--num_layers", "2", "-e", "3", "--batch_size", "1024", "--fanout", "25", "20"])
    d = json.loads(buf.getvalue().strip().splitlines()[-1])
    print("plan.enabled", en, "fwd %.3f bwd %.3f sample+layout %.3f" % (d["mean_forward_ms"], d["mean_backward_ms"], d["minibatch_sample_and_layout_ms"]), flush=True)
