"""How long does the HOST take to enqueue one step (no device sync)?  vs the device time."""
import sys, time, torch
import os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]      # the repo root
import esc_gnn_amd as E
from esc_gnn_amd.datasets import build_count_dataset
DEV = 'cuda:0'
bs = int(os.environ.get("ESC_BS", "128"))
graphs = build_count_dataset(0, 4 * bs, h=3, use_rd=True, self_loop=True)
y = torch.cat([g.y.view(-1) for g in graphs])
for g in graphs:
    g.y = (g.y.view(-1) - y.mean()) / y.std()
store = E.DeviceGraphStore(graphs, DEV)
ids = [torch.arange(i * bs, (i + 1) * bs) for i in range(4)]
m = E.NestedGIN_eff(None, 4, 256, use_rd=True, graph_pred=False, dropout=0, edge_nest=True, use_cycle=True).to(DEV)
opt = E.optim.FlatAdam(m.parameters(), lr=1e-3)
m.train()
eng = E.StepEngine(m)
nxt = store.collate(ids[0])
def step(i):
    global nxt
    b = nxt
    t0 = time.perf_counter()
    loss = eng.begin_step(b)
    t1 = time.perf_counter()
    nxt = store.collate(ids[(i + 1) % 4])
    t2 = time.perf_counter()
    eng.end_step()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    return (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
if os.environ.get("ESC_NODE_STREAM"):          # the whole loop on a non-default (non-blocking) stream
    _side = torch.cuda.Stream()
    _side.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(_side)
    nxt = store.collate(ids[0])
for i in range(10): step(i)
torch.cuda.synchronize()
# host-only: enqueue 30 steps, time the enqueue; then sync
t0 = time.perf_counter()
parts = [step(i) for i in range(30)]
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
import numpy as np
p = np.array(parts).mean(0) * 1e6
print(f"host enqueue per step {t_enq / 30 * 1e6:.0f} us  (begin_step {p[0]:.0f}, collate {p[1]:.0f}, end_step {p[2]:.0f}, adam {p[3]:.0f});  wall per step incl. device {t_all / 30 * 1e6:.0f} us")
import ctypes
from esc_gnn_amd import _native as nv
out = (ctypes.c_double * 6)()
n = nv.lib().esc_engine_phase_times(out, 12)
if n:
    print("phases over %d steps (us): start->edge fwd done %.0f | start->node fwd done %.0f | node bwd %.0f | node bwd done->edge bwd done %.0f | start->end %.0f | end->next start %.0f" % ((n,) + tuple(v * 1e3 for v in out)))
