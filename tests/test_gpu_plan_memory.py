"""Where the groupings' device memory lives: with het_amd.kernels imported, the library allocates through
torch's caching allocator (include/het_amd.h het_set_allocator; het_amd/_lib.py use_torch_allocator), so the groupings are part
of torch.cuda.memory_allocated, eviction from the plan cache gives the memory back to torch's pool, and a stream of one-shot
graphs does not grow the footprint."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _graph_tensors(seed, n=4000, e=60000):
    g = torch.Generator().manual_seed(seed)
    col = torch.randint(0, n, (e,), generator=g).to(DEV)
    p0 = torch.randint(0, n, (e,), generator=g).to(DEV)
    return col, p0, n


def test_groupings_are_inside_torch_allocator_statistics():
    import het_amd.kernels  # noqa: F401  (installs the allocator)
    from het_amd import _lib, plan
    assert _lib.allocator_is_external()
    plan.clear()
    torch.cuda.synchronize()
    col, p0, n = _graph_tensors(1, n=20000, e=400000)
    before = torch.cuda.memory_allocated()
    g = plan.get_grouping(None, col, n, p0, None)
    torch.cuda.synchronize()
    held = torch.cuda.memory_allocated() - before
    assert g.nbytes > 400000 * 8  # perm + payload at least
    assert held >= 0.9 * g.nbytes, (held, g.nbytes)  # (torch rounds blocks up: never less than the grouping's own count)
    assert torch.cuda.max_memory_allocated() >= before + g.nbytes
    del g
    plan.clear()
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() <= before + 1024  # destroyed -> back in torch's pool


def test_stream_of_graphs_does_not_grow_the_footprint():
    """80 distinct graphs through a cache of 32 entries: after the cache is full, memory_allocated stays flat (every new grouping
    evicts the oldest, whose memory returns to torch's pool) -- with plain hipMalloc the groupings were outside these statistics
    and each eviction cost a hipFree."""
    import het_amd.kernels  # noqa: F401
    from het_amd import plan
    plan.clear()
    torch.cuda.synchronize()
    trace = []
    for i in range(80):
        col, p0, n = _graph_tensors(100 + i)
        plan.get_grouping(None, col, n, p0, None)
        del col, p0  # (the cache entry keeps its source tensors alive, and releases them with the grouping)
        trace.append(torch.cuda.memory_allocated())
    assert len(plan._cache) == plan._MAX_ENTRIES
    full = max(trace[33:40])
    assert max(trace[40:]) <= full * 1.02 + (1 << 20), (full, max(trace[40:]))
    assert trace[20] < full  # (it did grow while the cache was filling: the figure does see the groupings)
    plan.clear()


def test_c_abi_default_stays_hipmalloc():
    """The bare C ABI (no het_set_allocator call) allocates with hipMalloc: switch the allocator off, build a grouping -- torch's
    statistics do not move -- and free it after the allocator is back on (a pointer is released through the allocator it came from)."""
    import het_amd.kernels  # noqa: F401
    from het_amd import _lib, plan
    plan.clear()
    col, p0, n = _graph_tensors(7, n=20000, e=400000)
    torch.cuda.synchronize()
    _lib.use_torch_allocator(False)
    try:
        assert not _lib.allocator_is_external()
        before = torch.cuda.memory_allocated()
        g = plan.get_grouping(None, col, n, p0, None)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() - before < 1 << 16
    finally:
        _lib.use_torch_allocator(True)
    assert _lib.allocator_is_external()
    del g
    plan.clear()  # hipFree of the hipMalloc'ed buffers, with torch's allocator installed again
    g2 = plan.get_grouping(None, col, n, p0, None)
    assert g2.num_segments > 0
    plan.clear()


def test_grouping_released_after_its_use_on_another_stream():
    """A grouping built on one stream, read by a queue of launches on ANOTHER stream, dropped from the cache while those launches
    are pending, its memory re-used at once on the creation stream (torch's allocator hands a freed block straight back to the
    stream that owns it): het_grouping_destroy has to order the release after the other stream's use (het_grouping_note_stream,
    include/het_amd.h; the cache lookup on the other stream reports it) -- the grouped sums still come out right."""
    import het_amd.kernels as k
    from het_amd import _lib, plan
    assert _lib.allocator_is_external()
    plan.clear()
    n, e, X = 3000, 400000, 64
    gen = torch.Generator().manual_seed(5)
    key = torch.randint(0, n, (e,), generator=gen).to(DEV)
    rows = torch.randn(e, X, generator=gen).to(DEV)
    want = torch.zeros(n, X, dtype=torch.float64, device=DEV).index_add_(0, key, rows.double()).float()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for trial in range(4):
        first = torch.zeros(n, X, device=DEV)
        k.rows_scatter_add_(first, key, rows)  # builds the grouping of `key` (by destination row) on the current stream
        outs = []
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(12):  # a queue of launches on the side stream that read the grouping's arrays
                out = torch.zeros(n, X, device=DEV)
                k.rows_scatter_add_(out, key, rows)
                outs.append(out)
        plan.clear()  # destroyed while the side stream still has those launches queued
        junk = [torch.full((e + 64,), -1, dtype=torch.int32, device=DEV) for _ in range(24)]  # the freed blocks, re-used on the creation stream
        torch.cuda.synchronize()
        del junk
        for out in [first] + outs:
            torch.testing.assert_close(out, want, rtol=2e-4, atol=2e-4)
