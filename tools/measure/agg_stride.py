"""Scatter-add (esc_gine_aggregate_fwd, C = 256) with the edge-term rows contiguous (ld 256) against rows that are a column
block of a [E, 768] matrix (ld 768) — what the batched edge-term GEMM hands it — cold operands (10 rotating buffer sets)."""
import os, sys, torch
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import esc_gnn_amd as E
from esc_gnn_amd import _native as nv
from esc_gnn_amd.datasets import build_count_dataset
dev = "cuda:0"
graphs = build_count_dataset(0, 256, h=3)
store = E.DeviceGraphStore(graphs, dev)
b = store.collate(torch.arange(128))
plan = E.plan_of(b)
N, Ee, H = plan.num_nodes, plan.num_edges, 256
s = nv.stream()
eps = torch.zeros(1, device=dev)
def run(ld, col):
    sets = [(torch.randn(N, H, device=dev), torch.randn(Ee, ld, device=dev), torch.empty(N, H, device=dev)) for _ in range(10)]
    fns = [(lambda a=a, e=e, c=c: nv.call("esc_gine_aggregate_fwd", nv.ptr(a), H, e.data_ptr() + 4 * col, ld, nv.ptr(plan.in_ptr), nv.ptr(plan.in_edge),
                                          nv.ptr(plan.in_src), nv.ptr(eps), N, H, nv.ptr(c), H, s)) for a, e, c in sets]
    for f in fns: f()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20):
        for f in fns: f()
    z.record(); torch.cuda.synchronize()
    return a.elapsed_time(z) / 200 * 1e3
for ld, col in ((256, 0), (768, 0), (768, 256), (256, 0), (768, 512), (1024, 256)):
    print("ld_e %4d col %3d: %.2f us" % (ld, col, run(ld, col)))
