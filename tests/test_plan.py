"""Host logic of het_amd/plan.py: the per-thread override of the groupings switch and its propagation from an autograd
node's forward to its backward (no GPU, no library call)."""
import threading

import torch

import het_amd.plan as plan


def test_forced_is_per_thread_and_restores():
    assert plan.is_enabled() == plan.enabled
    seen = {}
    with plan.forced(False):
        assert plan.is_enabled() is False
        t = threading.Thread(target=lambda: seen.setdefault("other", plan.is_enabled()))
        t.start()
        t.join()
        with plan.forced(None):  # no override inside: the process-wide default again
            assert plan.is_enabled() == plan.enabled
        assert plan.is_enabled() is False
    assert seen["other"] == plan.enabled  # the other thread never saw the override
    assert plan.is_enabled() == plan.enabled


def test_backward_runs_with_the_choice_of_its_forward():
    seen = []

    @plan.consistent
    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            seen.append(("fwd", plan.is_enabled()))
            return x * 2

        @staticmethod
        def backward(ctx, g):
            seen.append(("bwd", plan.is_enabled()))
            return g * 2

    x = torch.ones(3, requires_grad=True)
    with plan.forced(False):  # e.g. sampling.one_shot_graphs around the forward only
        y = Probe.apply(x)
    y.sum().backward()        # outside the block (and, on a GPU, on an autograd worker thread)
    z = Probe.apply(x)
    with plan.forced(False):
        z.sum().backward()    # a grouped forward keeps its grouped backward inside somebody else's one-shot block
    assert seen == [("fwd", False), ("bwd", False), ("fwd", plan.enabled), ("bwd", plan.enabled)]
    assert torch.equal(x.grad, torch.full((3,), 4.0))
