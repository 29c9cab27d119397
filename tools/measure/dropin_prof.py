"""Where the HOST time of the reference's own loop goes (model(batch) + torch.nn.L1Loss + optimizer): per-segment host
timers (no device sync inside) next to the synchronised wall time, and a cProfile of 200 steps."""
import cProfile, pstats, sys, time, os, io
import torch
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import esc_gnn_amd as E
from esc_gnn_amd.datasets import build_count_dataset
DEV = 'cuda:0'
graphs = build_count_dataset(0, 1024, h=3, use_rd=True, self_loop=True)
store = E.DeviceGraphStore(graphs, DEV)
bs = 128
ids = [torch.arange(i * bs, (i + 1) * bs) for i in range(len(store) // bs)]
which = sys.argv[1] if len(sys.argv) > 1 else "flat"
torch.manual_seed(0)
model = E.NestedGIN_eff(None, 4, 256, use_rd=True, graph_pred=False, dropout=0, edge_nest=True, use_cycle=True).to(DEV).train()
opt = E.optim.FlatAdam(model.parameters(), lr=1e-3) if which == "flat" else torch.optim.Adam(model.parameters(), lr=1e-3)
crit = torch.nn.L1Loss()
T = [0.0] * 6
def step(i, timed=False):
    t0 = time.perf_counter()
    b = store.collate(ids[i % len(ids)])
    t1 = time.perf_counter()
    opt.zero_grad()
    t2 = time.perf_counter()
    out = model(b)
    t3 = time.perf_counter()
    loss = crit(out, b.y.view(-1, 1))
    t4 = time.perf_counter()
    loss.backward()
    t5 = time.perf_counter()
    opt.step()
    t6 = time.perf_counter()
    if timed:
        for k, (a, c) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6))):
            T[k] += c - a
for i in range(10): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 200
for i in range(N): step(i, True)
th = time.perf_counter() - t0
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%s: wall %.3f ms/step, host %.3f ms/step; host us: collate %.0f zero_grad %.0f model() %.0f loss %.0f backward %.0f opt.step %.0f" %
      (which, dt / N * 1e3, th / N * 1e3, *[t / N * 1e6 for t in T]), flush=True)
pr = cProfile.Profile(); pr.enable()
for i in range(N): step(i)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
