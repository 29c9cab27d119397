"""Stream ordering of the two-bucket gradient exchange against a collective with ProcessGroupNCCL's semantics.

The two-rank tests run over gloo, which synchronises with the host: an ordering error between the engine's two streams and
an ASYNCHRONOUS collective could not show there.  RCCL through torch.distributed works on its own stream: at enqueue the
collective's stream waits for an event recorded on the caller's current stream; at completion the caller's stream waits for
the collective's event; the host never blocks.  This test swaps `parallel.dist` for a stand-in with exactly that protocol
(one process standing for two identical replicas: the "sum" doubles the buffer) and runs the step the way bench.py and
run_graphcount run it — begin_step -> all_reduce_early -> end_step -> all_reduce_late -> Adam, SURVEY 8e — against the same
step with the collective executed synchronously on the caller's stream.  Parameters after several steps must be BITWISE equal.
"""
import pytest
import torch

from conftest import require_gpu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _AsyncDist(object):
    """torch.distributed stand-in: world of 2 identical replicas, all_reduce(SUM) = x2, run on a private stream behind
    ProcessGroupNCCL's event protocol (or, sync=True, in place on the caller's stream)."""

    class ReduceOp(object):
        SUM = "sum"

    def __init__(self, sync, spin=0):
        self.sync, self.calls = sync, 0
        self.stream = torch.cuda.Stream(device=DEV)
        self.spin = spin

    def is_available(self):
        return True

    def is_initialized(self):
        return True

    def get_world_size(self, group=None):
        return 2

    def get_rank(self, group=None):
        return 0

    def all_reduce(self, t, op=None, group=None, async_op=False):
        self.calls += 1
        if self.sync:
            t.mul_(2.0)
            return None
        cur = torch.cuda.current_stream(t.device)
        start = torch.cuda.Event()
        start.record(cur)                     # "inputs are ready" as of the caller's stream order at enqueue
        self.stream.wait_event(start)
        with torch.cuda.stream(self.stream):
            if self.spin:                     # a slow collective: what it overlaps with must not touch its buffer
                torch.cuda._sleep(self.spin)
            t.mul_(2.0)
            done = torch.cuda.Event()
            done.record(self.stream)
        cur.wait_event(done)                  # work.wait(): the caller's STREAM waits, the host does not
        return None


def _run(E, sync, steps=4):
    from esc_gnn_amd import parallel
    from esc_gnn_amd.datasets import build_count_dataset
    graphs = build_count_dataset(0, 48, h=2, use_rd=True, self_loop=True)
    gen = torch.Generator().manual_seed(3)
    for g in graphs:
        g.x = torch.randn(g.x.shape, generator=gen)
        g.y = torch.randn(g.x.size(0), generator=gen)
    store = E.DeviceGraphStore(graphs, DEV)
    torch.manual_seed(5)
    model = E.NestedGIN_eff(None, 3, 64, use_rd=True, graph_pred=False, dropout=0, edge_nest=True, use_cycle=True).to(DEV).train()
    opt = E.optim.FlatAdam(model.parameters(), lr=1e-2, late=E.parallel.edge_pipeline_parameters(model))
    eng = E.StepEngine(model)
    fake = _AsyncDist(sync, spin=0 if sync else 2_000_000)
    real = parallel.dist
    parallel.dist = fake
    try:
        ids = [torch.arange(i * 16, (i + 1) * 16) for i in range(3)]
        nxt = store.collate(ids[0])
        for i in range(steps):
            b = nxt
            eng.begin_step(b, loss_denom=1)
            nxt = store.collate(ids[(i + 1) % 3])          # the next collate sits between the two halves, as in bench.py
            opt.all_reduce_early()
            eng.end_step()
            opt.step(grad_denom=opt.all_reduce_late(b.x.size(0)))
        torch.cuda.synchronize()
    finally:
        parallel.dist = real
    assert fake.calls == 2 * steps
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu(), opt.flat_grad.detach().cpu().clone()


def test_two_bucket_exchange_is_ordered_against_an_asynchronous_collective():
    require_gpu()
    import esc_gnn_amd as E
    want_p, want_g = _run(E, sync=True)
    for _ in range(2):
        got_p, got_g = _run(E, sync=False)
        assert torch.equal(got_g, want_g)
        assert torch.equal(got_p, want_p)
