"""MEASURE a replayed step graph: the counting model's training step — forward + L1 + backward on both
streams, gradient reductions, Adam — captured ONCE on a fixed batch into a hipGraph and replayed, against the same step
enqueued eagerly on the same batch.  A replay of one fixed batch is the upper bound of what any step graph can give
(a real loop would additionally have to patch N / E / Z-dependent grids and arguments per batch).

    python tools/measure/graph_replay.py            -> profiles/r03_step_graph.txt

Prints: eager ms/step (fixed batch), host enqueue time per eager step, graph replay ms/step, host time per replay.
"""
import os
import sys
import time

import torch

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import esc_gnn_amd as E
from esc_gnn_amd.datasets import build_count_dataset

DEV = "cuda:0"
bs = 128
graphs = build_count_dataset(0, 2 * bs, h=3, use_rd=True, self_loop=True)
y = torch.cat([g.y.view(-1) for g in graphs])
for g in graphs:
    g.y = (g.y.view(-1) - y.mean()) / y.std()
store = E.DeviceGraphStore(graphs, DEV)
torch.manual_seed(0)
m = E.NestedGIN_eff(None, 4, 256, use_rd=True, graph_pred=False, dropout=0, edge_nest=True, use_cycle=True).to(DEV)
opt = E.optim.FlatAdam(m.parameters(), lr=1e-3)
m.train()
eng = E.StepEngine(m)
batch = store.collate(torch.arange(bs))
STEPS = 200


def step():
    loss = eng.train_step(batch)
    opt.step()
    return loss


for _ in range(20):
    step()
torch.cuda.synchronize()

# ---- eager, fixed batch ------------------------------------------------------------------------------------------------
t0 = time.perf_counter()
for _ in range(STEPS):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_eager = time.perf_counter() - t0
print("eager, fixed batch      : %.3f ms/step (host enqueue %.3f ms/step)" % (t_eager / STEPS * 1e3, t_host / STEPS * 1e3), flush=True)

# ---- the same step captured into a graph -------------------------------------------------------------------------------
try:
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        loss = step()
    torch.cuda.synchronize()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        g.replay()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_graph = time.perf_counter() - t0
    print("graph replay, same batch: %.3f ms/step (host %.3f ms/replay), loss %.6f" % (t_graph / STEPS * 1e3, t_host / STEPS * 1e3, float(loss)), flush=True)
    print("ratio graph / eager     : %.3f" % (t_graph / t_eager))
except Exception as exc:       # a capture that the runtime refuses is a finding too
    print("graph capture failed: %s: %s" % (type(exc).__name__, str(exc)[:400]), flush=True)

# ---- what the bench loop measures (fresh batch per step, collate inside): for reference ---------------------------------
ids = [torch.arange(i * bs, (i + 1) * bs) for i in range(2)]
for i in range(10):
    b = store.collate(ids[i % 2]); eng.train_step(b); opt.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(STEPS):
    b = store.collate(ids[i % 2]); eng.train_step(b); opt.step()
torch.cuda.synchronize()
print("eager, collate per step : %.3f ms/step" % ((time.perf_counter() - t0) / STEPS * 1e3))
