"""Multi-GPU execution of one relational layer: destination-range partition with a
halo exchange of source features (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for the tests).

The reference is single-GPU (no collective anywhere in its tree); this is new design
(SURVEY.md section 8e).  Every reduction of the forward pass -- the softmax denominator and
the aggregation -- is keyed by the destination node, so giving each rank a contiguous
range of destination nodes together with ALL in-edges of those nodes keeps the
forward reductions local.  What crosses GPUs:

  forward   x[halo]: features of remote source nodes, one all-to-all (point-to-point
            xGMI links all busy at once) before the layer;
  backward  grad_x[halo]: the reverse all-to-all, added into the owners' gradients;
            weight gradients: all-reduce (a few hundred KiB).

Ranges are balanced on in-edge count, not node count.  Nodes WITHOUT in-edges (pure sources: the
authors of ogbn-mag) cost their owner nothing but the sends, so they are dealt out evenly instead of
falling into whichever range surrounds them (on mag that put all 1.13 M authors, and 2.2 M of the
4.5 M rows sent per step at 8 ranks, on rank 0).  Ownership is expressed by renumbering the nodes
(``DistPlan.node_order``: new id -> original id) so that every rank owns a contiguous range of new ids.
Local node numbering: owned nodes first (new id - lo), then halo nodes ordered by new id (hence grouped
by owning rank, so the received buffer is already in halo order).
"""
from __future__ import annotations

import dataclasses
import os
from typing import Callable, List

import torch
import torch.distributed as dist

from .graph import HetGraph
from .synth import IntegratedCOO


# Cost of owning a destination node, in edges: its self-loop / output / gradient rows are per-node work on top of the
# per-edge work of its in-edges (exp/dist_rank_share.py: with edges alone the ranks that own many low-degree
# destinations run 40 % longer than the ones that own few high-degree ones).
NODE_WEIGHT = float(os.environ.get("HET_DIST_NODE_WEIGHT", "12"))  # sweep 0..64 on ogbn-mag: best 8..16 at 2, 4 and 8 ranks


# Cost of a halo row (a remote source a rank has to receive, project and return a gradient for), in edges, and the number
# of refinement rounds of the boundaries against it (the halo of a range is only known once the ranges are: start from the
# edge + node balance, measure every rank's halo, move the boundaries towards equal edge + node + halo cost, repeat).
HALO_WEIGHT = float(os.environ.get("HET_DIST_HALO_WEIGHT", "8"))  # sweep 0..24 on ogbn-mag: 8 is best at 2, 4 and 8 ranks
REFINE_ROUNDS = int(os.environ.get("HET_DIST_REFINE", "4"))


def partition_bounds(col: torch.Tensor, num_nodes: int, world: int, node_weight: float = None, row: torch.Tensor = None,
                     halo_weight: float = None) -> torch.Tensor:
    """Node-id boundaries [world+1] of contiguous destination ranges with ~equal cost = in-edges + node_weight per
    destination (node_weight 0: equal in-edge counts) + halo_weight per remote source row of the range (needs ``row``)."""
    node_weight = NODE_WEIGHT if node_weight is None else node_weight
    halo_weight = HALO_WEIGHT if halo_weight is None else halo_weight
    dev = col.device
    indeg = torch.bincount(col, minlength=num_nodes)
    cost = indeg.to(torch.float64) + node_weight * (indeg > 0).to(torch.float64)
    csum = torch.cumsum(cost, 0)
    total = float(csum[-1]) if num_nodes else 0.0

    def bounds_for(shares):
        targets = torch.cumsum(shares, 0)[:-1] * total
        cuts = torch.searchsorted(csum, targets, right=False) + 1
        b = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), cuts.clamp(max=num_nodes),
                       torch.tensor([num_nodes], dtype=torch.int64, device=dev)])
        return torch.cummax(b, 0).values

    shares = torch.full((world,), 1.0 / world, dtype=torch.float64, device=dev)
    b = bounds_for(shares)
    if row is None or world < 2 or halo_weight <= 0 or col.numel() == 0:
        return b
    has_in = indeg > 0
    for _ in range(REFINE_ROUNDS):
        upper = b[1:].contiguous()
        own_dst = torch.searchsorted(upper, col, right=True).clamp(max=world - 1)
        own_src = torch.searchsorted(upper, row, right=True).clamp(max=world - 1)
        # (sources without in-edges are dealt out evenly afterwards, node_ownership: counted as remote here)
        remote = (own_dst != own_src) | ~has_in[row]
        halo = torch.bincount(torch.unique(own_dst[remote] * num_nodes + row[remote]) // num_nodes, minlength=world)
        base = csum[(b[1:] - 1).clamp(min=0)] * (b[1:] > 0)
        part = base - torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), base[:-1]])
        rank_cost = part + halo_weight * halo.to(torch.float64)
        shares = shares * (rank_cost.mean() / rank_cost.clamp(min=1.0))
        shares = shares / shares.sum()
        b = bounds_for(shares)
    return b


def node_ownership(col: torch.Tensor, num_nodes: int, world: int, row: torch.Tensor = None):
    """(node_order [N] new -> original id, new_id [N] original -> new, bounds [world+1] in new ids).
    Destinations: contiguous original-id ranges with ~equal in-edge counts.  Nodes without in-edges: equal shares."""
    dev = col.device
    indeg = torch.bincount(col, minlength=num_nodes)
    b = partition_bounds(col, num_nodes, world, row=row)
    ids = torch.arange(num_nodes, device=dev)
    owner = torch.searchsorted(b[1:].contiguous(), ids, right=True).clamp(max=world - 1)
    free = indeg == 0
    n_free = int(free.sum())
    if n_free:
        k = torch.cumsum(free.to(torch.int64), 0)[free] - 1          # rank of each free node among the free nodes
        owner[free] = (k * world) // n_free
    node_order = torch.sort(owner * num_nodes + ids).indices          # by (owner, original id)
    new_id = torch.empty_like(node_order)
    new_id[node_order] = ids
    counts = torch.bincount(owner, minlength=world)
    bounds = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(counts, 0)])
    return node_order, new_id, bounds


@dataclasses.dataclass
class DistPlan:
    rank: int
    world: int
    bounds: torch.Tensor          # [world+1] boundaries in the renumbered node ids
    node_order: torch.Tensor      # [N] original id of renumbered node i (rank r owns node_order[bounds[r]:bounds[r+1]])
    n_own: int
    n_halo: int
    local: IntegratedCOO          # this rank's edges in local node ids (relation-major, eids = arange)
    halo_global: torch.Tensor     # [n_halo] global ids of the halo nodes (sorted)
    send_idx: torch.Tensor        # [sum(send_counts)] owned-local ids to send, grouped by destination rank
    send_counts: List[int]
    recv_counts: List[int]
    num_global_edges: int
    edge_cut: int                 # edges whose source lives on another rank (all ranks)
    # The exchanges run in `chunks` pieces (HET_DIST_CHUNKS): piece c moves, from every peer, the c-th C-th of the rows that peer
    # sends -- all links busy in every piece -- so that the consumer of the halo rows (the projection of the (relation, source)
    # rows) starts on piece 0 while piece 1 is on the wire.  Halo nodes are numbered piece-major (piece, then owning rank, then
    # node id) and send_idx is ordered the same way: a piece is one contiguous range of both buffers.
    chunks: int = 1
    send_splits: List[List[int]] = None   # [chunks][world] rows of piece c that go to rank p
    recv_splits: List[List[int]] = None   # [chunks][world] rows of piece c that come from rank p
    halo_chunk_ptr: List[int] = None      # [chunks+1] halo-row range of piece c (add n_own for local node ids)
    send_chunk_ptr: List[int] = None      # [chunks+1] send_idx range of piece c

    @property
    def num_local_edges(self) -> int:
        return self.local.num_edges

    @property
    def num_local_nodes(self) -> int:
        return self.n_own + self.n_halo


CHUNKS = int(os.environ.get("HET_DIST_CHUNKS", "4"))  # pieces of each halo exchange (1: one monolithic all-to-all)


def _piece_major(counts: List[int], chunks: int):
    """For blocks of counts[p] consecutive rows (one block per peer p): the permutation that lists the rows piece-major -- piece c
    of block p is its rows [c * n // chunks, (c + 1) * n // chunks) -- and the [chunks][len(counts)] piece sizes.  Sender and
    receiver cut a block of n rows at the same places."""
    starts = [0]
    for n in counts:
        starts.append(starts[-1] + n)
    order, splits = [], []
    for c in range(chunks):
        row = []
        for p, n in enumerate(counts):
            a, b = c * n // chunks, (c + 1) * n // chunks
            row.append(b - a)
            if b > a:
                order.append(torch.arange(starts[p] + a, starts[p] + b, dtype=torch.int64))
        splits.append(row)
    perm = torch.cat(order) if order else torch.zeros(0, dtype=torch.int64)
    return perm, splits


def build_plan(coo: IntegratedCOO, rank: int, world: int, chunks: int = None) -> DistPlan:
    """Every rank holds the (seeded, identical) global edge list and derives its own share:
    no communication is needed to build the plan."""
    chunks = max(1, CHUNKS if chunks is None else int(chunks))
    N = coo.num_nodes
    node_order, new_id, bounds = node_ownership(coo.col, N, world, row=coo.row)
    row, col, rel = new_id[coo.row], new_id[coo.col], coo.rel
    owner_dst = torch.searchsorted(bounds[1:].contiguous(), col, right=True)
    owner_src = torch.searchsorted(bounds[1:].contiguous(), row, right=True)
    cut = owner_dst != owner_src
    # distinct (needing rank, remote source) pairs == halo memberships of every rank
    pairs = torch.unique(owner_dst[cut] * N + row[cut])
    need_rank, need_node = pairs // N, pairs % N
    node_owner = torch.searchsorted(bounds[1:].contiguous(), need_node, right=True)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    dev = row.device

    mine = need_rank == rank
    halo_sorted = need_node[mine]                    # sorted by global id => grouped by owner
    recv_counts = torch.bincount(node_owner[mine], minlength=world).tolist()
    send_sel = node_owner == rank                    # pairs sorted by (needing rank, node)
    send_sorted = (need_node[send_sel] - lo).contiguous()
    send_counts = torch.bincount(need_rank[send_sel], minlength=world).tolist()
    # piece-major order of both buffers (see DistPlan.chunks)
    rperm, recv_splits = _piece_major(recv_counts, chunks)
    sperm, send_splits = _piece_major(send_counts, chunks)
    rperm, sperm = rperm.to(dev), sperm.to(dev)
    halo_global = halo_sorted[rperm]
    send_idx = send_sorted[sperm].contiguous()
    halo_pos = torch.empty_like(rperm)               # sorted position -> local halo index
    halo_pos[rperm] = torch.arange(rperm.numel(), device=dev)
    cum = lambda rows: [0] + torch.cumsum(torch.tensor([sum(r) for r in rows]), 0).tolist()

    e_sel = owner_dst == rank
    l_row, l_col, l_rel = row[e_sel], col[e_sel] - lo, rel[e_sel]
    remote = (l_row < lo) | (l_row >= hi)
    n_own, n_halo = hi - lo, int(halo_global.numel())
    if n_halo:
        l_row_local = torch.where(remote, n_own + halo_pos[torch.searchsorted(halo_sorted, l_row).clamp(max=n_halo - 1)], l_row - lo)
    else:
        l_row_local = l_row - lo
    local = IntegratedCOO(num_nodes=n_own + n_halo, num_rels=coo.num_rels,
                          node_type_offsets=torch.tensor([0, n_own + n_halo], device=row.device),
                          row=l_row_local.contiguous(), col=l_col.contiguous(), rel=l_rel.contiguous(),
                          eids=torch.arange(int(l_row.numel()), dtype=torch.int64, device=row.device))
    return DistPlan(rank, world, bounds, node_order, n_own, n_halo, local, halo_global, send_idx, send_counts, recv_counts,
                    coo.num_edges, int(cut.sum()), chunks, send_splits, recv_splits, cum(recv_splits), cum(send_splits))


def _all_to_all(recv, send, recv_counts, send_counts, group):
    """all_to_all_single; with the gloo backend (CPU tests, or several test ranks sharing one GPU) device tensors are
    staged through the host, with nccl (= RCCL over xGMI) they go device to device."""
    if send.is_cuda and dist.get_backend(group) == "gloo":
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(r, send.cpu(), recv_counts, send_counts, group=group)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv, send, recv_counts, send_counts, group=group)


def _all_reduce(t, group):
    if t.is_cuda and dist.get_backend(group) == "gloo":
        c = t.cpu()
        dist.all_reduce(c, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, group=group)


def _pieces(plan: DistPlan, reverse: bool):
    """(send range, send splits, recv range, recv splits) of every piece of an exchange.  Forward: the packed rows (send_idx
    order) travel to the peers' halo rows; reverse: halo-row gradients travel back into a buffer in send_idx order."""
    out = []
    for c in range(plan.chunks):
        packed = (plan.send_chunk_ptr[c], plan.send_chunk_ptr[c + 1], plan.send_splits[c])
        halo = (plan.halo_chunk_ptr[c], plan.halo_chunk_ptr[c + 1], plan.recv_splits[c])
        out.append((halo, packed) if reverse else (packed, halo))
    return out


def _exchange(plan: DistPlan, recv, send, reverse: bool, group):
    """The whole exchange, piece after piece, synchronously (HaloExchange; HaloContext under gloo)."""
    for (sa, sb, ssp), (ra, rb, rsp) in _pieces(plan, reverse):
        _all_to_all(recv[ra:rb], send[sa:sb], rsp, ssp, group)


def _gather_rows(x, idx):
    if x.is_cuda:  # the product path: HIP kernels behind the C ABI (fails loudly without the library)
        from . import kernels
        return kernels.rows_gather(x.contiguous(), idx)
    return x.index_select(0, idx).contiguous()  # CPU (gloo tests with the oracle as the per-rank layer)


def _scatter_add_rows(out, idx, src):
    if out.is_cuda:
        from . import kernels
        return kernels.rows_scatter_add_(out, idx, src.contiguous())
    return out.index_add_(0, idx, src)


class HaloContext:
    """The halo exchange of one layer step as explicit start / finish calls, for a layer that overlaps it with its own
    work (het_amd/backend/rgat_fused_layer.py): with the nccl backend the all-to-all runs on RCCL's stream while the
    launch stream keeps going (``async_op=True``; ``wait()`` makes the launch stream wait for it).  With gloo (CPU tests, or
    test ranks sharing one GPU) the calls complete immediately.

      forward   x_local = start_push(x_own)   ... work that reads owned rows only ...   finish_push()
      backward  start_return(grad_local)      ... work that leaves the halo rows alone ...   finish_return(grad_own)
    """

    def __init__(self, plan: DistPlan, group):
        self.plan, self.group = plan, group
        self._works = None  # the pieces of the exchange in flight: [(work or None, piece index)], in issue order
        self._keep = self._back = None
        # exposed wait of the launch stream for the exchanges of a step (bench.py: `timing = []` switches it on):
        # (what, event recorded when the launch stream reaches the wait, event recorded when it resumes)
        self.timing = None

    @property
    def chunks(self) -> int:
        return self.plan.chunks

    def _a2a(self, recv, send, reverse):
        # one exchange in flight per context: an unpaired start_* (an exception between start and finish, a re-entered layer)
        # would silently drop the handles of collectives that are still writing `recv`
        if self._works is not None:
            raise RuntimeError("HaloContext: an exchange is still in flight (start_* without its finish_*)")
        if send.is_cuda and dist.get_backend(self.group) != "gloo":
            works = []
            for c, ((sa, sb, ssp), (ra, rb, rsp)) in enumerate(_pieces(self.plan, reverse)):
                works.append((dist.all_to_all_single(recv[ra:rb], send[sa:sb], rsp, ssp, group=self.group, async_op=True), c))
            self._works = works
            self._keep = (recv, send)  # alive until the collectives have finished
        else:
            _exchange(self.plan, recv, send, reverse, self.group)
            self._works = []

    def _wait(self, what="exchange", upto=None):
        """Make the launch stream wait for the pieces of the exchange in flight up to piece ``upto`` (None: all of them, and the
        exchange is over)."""
        if self._works is None:
            return
        while self._works and (upto is None or self._works[0][1] <= upto):
            work, c = self._works[0]
            ev = None
            if self.timing is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            # (a wait that raised leaves the collective possibly still writing its buffers: they stay referenced -- and the
            #  context unusable -- until abort() gets a wait through or the process ends)
            work.wait()
            self._works.pop(0)
            if ev is not None:
                ev[1].record()
                self.timing.append((f"{what}[{c}]" if self.plan.chunks > 1 else what, ev[0], ev[1]))
        if upto is None:
            self._works = self._keep = None

    def abort(self):
        """Wait for an exchange a failed step left in flight (exception between start_* and finish_*), so that its buffers may
        be freed and the context reused."""
        try:
            self._wait("abort")
        except BaseException:
            self._works = None  # the handles are dropped, the buffers (`_keep`) are deliberately leaked with the failed collective
            raise
        finally:
            self._back = None

    def start_push(self, x_own):
        p = self.plan
        send = _gather_rows(x_own, p.send_idx)
        x_local = x_own.new_empty((p.n_own + p.n_halo, x_own.shape[1]))
        x_local[: p.n_own].copy_(x_own)
        self._a2a(x_local[p.n_own:], send, False)  # straight into the halo rows
        return x_local

    def wait_push_piece(self, c: int):
        """The halo rows of pieces 0..c have arrived (local nodes n_own + halo_chunk_ptr[c] .. n_own + halo_chunk_ptr[c + 1])."""
        self._wait("push", upto=c)

    def finish_push(self):
        self._wait("push")

    def start_return(self, grad_local):
        p = self.plan
        self._back = grad_local.new_empty((int(p.send_idx.numel()), grad_local.shape[1]))
        self._a2a(self._back, grad_local[p.n_own:], True)

    def finish_return(self, grad_own):
        self._wait("return")
        _scatter_add_rows(grad_own, self.plan.send_idx, self._back)
        self._back = None
        return grad_own


class HaloExchange(torch.autograd.Function):
    """x_own [n_own, K] -> [n_own + n_halo, K]: owned rows followed by the halo rows received from
    their owners.  Backward sends the halo gradients home and adds them to the owners' rows."""

    @staticmethod
    def forward(ctx, x_own, plan: DistPlan, group):
        ctx.plan, ctx.group = plan, group
        send = _gather_rows(x_own, plan.send_idx)
        x_local = x_own.new_empty((plan.n_own + plan.n_halo, x_own.shape[1]))
        x_local[: plan.n_own].copy_(x_own)
        _exchange(plan, x_local[plan.n_own:], send, False, group)  # straight into the halo rows
        return x_local

    @staticmethod
    def backward(ctx, grad):
        plan = ctx.plan
        g_own = grad[: plan.n_own].clone()
        g_halo = grad[plan.n_own:].contiguous()
        back = grad.new_empty((int(plan.send_idx.numel()), grad.shape[1]))
        _exchange(plan, back, g_halo, True, ctx.group)
        _scatter_add_rows(g_own, plan.send_idx, back)
        return g_own, None, None


class DistLayer:
    """One layer sharded over the ranks of ``group``.  ``layer_fn(graph, x_local, n_own)`` computes the layer
    on the local graph (HET_RGATLayer on the GPU; the CPU oracle in the gloo tests) and returns
    [n_local, X] or only its first n_own rows; rows of owned nodes are kept.  ``params`` are replicated and their gradients
    all-reduced after backward."""

    def __init__(self, coo: IntegratedCOO, layer_fn: Callable, params, group=None, full_layouts: bool = False,
                 halo_layer_fn: Callable = None, plan: DistPlan = None):
        """``halo_layer_fn(graph, x_own, halo)`` (optional): a layer that runs the exchange itself through a HaloContext and
        overlaps it with its own work; it returns None when it cannot (then ``layer_fn`` runs behind HaloExchange).
        ``plan`` (optional): a plan built elsewhere (LocalRanks builds all ranks' plans from one pass)."""
        self.group = group
        if plan is None:
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
            plan = build_plan(coo, self.rank, self.world)
        self.rank, self.world = plan.rank, plan.world
        self.plan = plan
        self.graph = HetGraph.from_integrated_coo(self.plan.local, full=full_layouts)
        self.layer_fn = layer_fn
        self.halo_layer_fn = halo_layer_fn
        self.halo = HaloContext(self.plan, group)
        self.params = list(params)

    def forward(self, x_own: torch.Tensor) -> torch.Tensor:
        if self.halo_layer_fn is not None:
            try:
                out = self.halo_layer_fn(self.graph, x_own, self.halo)
            except BaseException:
                self.halo.abort()  # (no exchange may stay in flight behind a failed step)
                raise
            if out is not None:
                return out
        x_local = HaloExchange.apply(x_own, self.plan, self.group)
        return self.layer_fn(self.graph, x_local, self.plan.n_own)[: self.plan.n_own]

    def reduce_param_grads(self):
        grads = [p.grad for p in self.params if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        _all_reduce(flat, self.group)
        off = 0
        for g in grads:
            g.copy_(flat[off: off + g.numel()].view_as(g))
            off += g.numel()


def rccl_preflight(group=None, device=None, timeout_s: float = None) -> None:
    """First RCCL contact, made boring: one tiny all_to_all_single (every rank sends its rank number to every peer) and a
    barrier, under a watchdog.  A rank whose exchange has not completed within ``timeout_s`` (HET_DIST_PREFLIGHT_TIMEOUT,
    default 60 s) -- a peer that never joined, an xGMI / IPC set-up that hangs inside the library -- prints which rank it is and
    what it was waiting for and leaves with exit code 3 (torch.distributed.run then stops the others); a wrong payload raises.
    Exit, never re-exec: this process has a GPU context."""
    import sys
    import threading
    timeout_s = float(os.environ.get("HET_DIST_PREFLIGHT_TIMEOUT", "60")) if timeout_s is None else float(timeout_s)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    stage = ["communicator set-up"]

    def expired():
        sys.stderr.write(f"[het_amd.dist] RCCL preflight: rank {rank}/{world} still in '{stage[0]}' after {timeout_s:.0f} s "
                         f"(device {device}, MASTER_ADDR={os.environ.get('MASTER_ADDR')}, HSA_ENABLE_IPC_MODE_LEGACY="
                         f"{os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}): giving up\n")
        sys.stderr.flush()
        os._exit(3)

    dog = threading.Timer(timeout_s, expired)
    dog.daemon = True
    dog.start()
    try:
        send = torch.full((world,), float(rank), dtype=torch.float32, device=device)
        recv = torch.full((world,), -1.0, dtype=torch.float32, device=device)
        stage[0] = "all_to_all_single of one float per peer"
        dist.all_to_all_single(recv, send, group=group)
        if send.is_cuda:
            torch.cuda.synchronize(send.device)
        if not torch.equal(recv.cpu(), torch.arange(world, dtype=torch.float32)):
            raise RuntimeError(f"RCCL preflight: rank {rank} received {recv.tolist()} instead of the peers' rank numbers")
        stage[0] = "barrier"
        dist.barrier(group=group)
    finally:
        dog.cancel()


class _Loopback:
    """The wire of LocalRanks: what each logical rank put up for its peers, kept in this process."""

    def __init__(self):
        self.push = {}   # rank -> rows it sends in the forward exchange (grouped by receiving rank)
        self.back = {}   # rank -> gradients of its halo rows (grouped by owning rank), sent home in the backward exchange


class LoopbackHalo(HaloContext):
    """HaloContext of one logical rank of LocalRanks: same pack / unpack code, the all-to-all replaced by slicing the peers'
    buffers.  The forward rows of every rank are put up before any rank's layer runs; the returned gradient rows are added to
    their owners' gradients after every rank's backward has run (LocalRanks.backward) -- sums commute."""

    def __init__(self, plan: DistPlan, plans, wire: _Loopback):
        super().__init__(plan, None)
        self.plans, self.wire = plans, wire

    def start_push(self, x_own):
        p = self.plan
        x_local = x_own.new_empty((p.n_own + p.n_halo, x_own.shape[1]))
        x_local[: p.n_own].copy_(x_own)
        off = p.n_own
        for c in range(p.chunks):       # a piece's received rows are ordered by sending rank (all_to_all_single)
            for q in range(p.world):
                n = p.recv_splits[c][q]
                if n:
                    pq = self.plans[q]
                    assert pq.send_splits[c][p.rank] == n, "send / receive splits of the plans disagree"
                    a = pq.send_chunk_ptr[c] + sum(pq.send_splits[c][: p.rank])
                    x_local[off: off + n].copy_(self.wire.push[q][a: a + n])
                off += n
        return x_local

    def wait_push_piece(self, c: int):
        pass

    def finish_push(self):
        pass

    def start_return(self, grad_local):
        self.wire.back[self.plan.rank] = grad_local[self.plan.n_own:].clone()

    def finish_return(self, grad_own):
        return grad_own


class LocalRanks:
    """All ``world`` ranks of a partition as LOGICAL ranks of one process on one device: every rank's DistPlan, local graph
    and halo pack / unpack exactly as in a multi-process run, one replicated layer (its parameter gradients accumulate over
    the ranks' backward passes = the all-reduce), the exchange by slicing.  A rehearsal of an N-GPU step where N GPUs (or N
    processes on one GPU: the test boxes allow 6) are not to be had -- tests/test_gpu_dist.py runs BASELINE.json configs[4]
    (RGAT, feat 128, 8-way) through it against the single-process layer."""

    def __init__(self, coo: IntegratedCOO, world: int, layer, full_layouts: bool = False, overlap: bool = True):
        self.world, self.layer, self.overlap = world, layer, overlap
        self.plans = [build_plan(coo, r, world) for r in range(world)]
        self.graphs = [HetGraph.from_integrated_coo(p.local, full=full_layouts) for p in self.plans]
        self.wire = _Loopback()
        self.halos = [LoopbackHalo(p, self.plans, self.wire) for p in self.plans]
        self.took_halo_path = [False] * world
        self._x_local = [None] * world

    def owned_nodes(self, r):
        p = self.plans[r]
        return p.node_order[int(p.bounds[r]): int(p.bounds[r + 1])]

    def forward(self, x_own: List[torch.Tensor]) -> List[torch.Tensor]:
        for r, p in enumerate(self.plans):
            self.wire.push[r] = _gather_rows(x_own[r].detach(), p.send_idx)
        outs = []
        for r, p in enumerate(self.plans):
            out = self.layer.forward_with_halo(self.graphs[r], x_own[r], self.halos[r]) if self.overlap else None
            self.took_halo_path[r] = out is not None
            if out is None:  # exchange first, then the layer on the local graph (DistLayer.forward's other branch)
                xl = self.halos[r].start_push(x_own[r].detach()).requires_grad_(True)
                self._x_local[r] = xl
                out = self.layer(self.graphs[r], xl, num_dst=p.n_own)[: p.n_own]
            outs.append(out)
        return outs

    def backward(self, outs: List[torch.Tensor], gradouts: List[torch.Tensor], x_own: List[torch.Tensor]):
        """Runs every rank's backward; x_own[r].grad then holds the owned rows' gradient including the rows peers returned."""
        for r, p in enumerate(self.plans):
            outs[r].backward(gradouts[r])
            if not self.took_halo_path[r]:
                g = self._x_local[r].grad
                self.halos[r].start_return(g)
                x_own[r].grad = g[: p.n_own].clone()
                self._x_local[r] = None
        for r, p in enumerate(self.plans):  # the reverse all-to-all: rank r gets back the gradients of the rows it sent
            parts = []
            for c in range(p.chunks):
                for q in range(self.world):
                    n = p.send_splits[c][q]
                    if n:
                        pq = self.plans[q]
                        a = pq.halo_chunk_ptr[c] + sum(pq.recv_splits[c][:r])
                        parts.append(self.wire.back[q][a: a + n])
            if parts:
                _scatter_add_rows(x_own[r].grad, p.send_idx, torch.cat(parts))
        self.wire.back.clear()
        self.wire.push.clear()


class DistRGAT:
    """bench.py's multi-GPU step: a replicated HET_RGATLayer over the local shard of the graph."""

    def __init__(self, coo: IntegratedCOO, in_feat, out_feat, heads, device, **layer_flags):
        from .layers import HET_RGATLayer
        if dist.get_backend() == "nccl" and os.environ.get("HET_DIST_PREFLIGHT", "1") == "1":
            rccl_preflight(None, device)  # fails fast, naming the rank, before minutes of plan building behind a dead link
        for f in ("row", "col", "rel", "eids", "node_type_offsets"):
            setattr(coo, f, getattr(coo, f).to(device))
        torch.manual_seed(0)  # same weights on every rank
        self.layer = HET_RGATLayer(in_feat, out_feat, coo.num_rels, heads, self_loop=True, dropout=0.0,
                                   **layer_flags).to(device)
        overlap = os.environ.get("HET_DIST_OVERLAP", "1") == "1"
        self.dl = DistLayer(coo, lambda g, x, n_own: self.layer(g, x, num_dst=n_own), self.layer.parameters(),
                            full_layouts=bool(layer_flags.get("compact_as_of_node_flag")),
                            halo_layer_fn=(lambda g, x_own, halo: self.layer.forward_with_halo(g, x_own, halo)) if overlap else None)
        p = self.dl.plan
        self.embed = torch.nn.Parameter(torch.empty(p.n_own, in_feat, device=device))
        torch.nn.init.xavier_uniform_(self.embed)
        self.go = torch.randn(p.n_own, out_feat, device=device)
        self.num_local_edges, self.num_local_nodes = p.num_local_edges, p.num_local_nodes

    def step(self):
        for q in self.layer.parameters():
            q.grad = None
        self.embed.grad = None
        out = self.dl.forward(self.embed)
        out.backward(self.go)
        self.dl.reduce_param_grads()
