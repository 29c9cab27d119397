import sys, time, torch
import os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]      # the repo root
import esc_gnn_amd as E
from esc_gnn_amd.datasets import build_count_dataset
DEV = 'cuda:0'
graphs = build_count_dataset(0, 1024, h=3, use_rd=True, self_loop=True)
store = E.DeviceGraphStore(graphs, DEV)
bs = 128
ids = [torch.arange(i * bs, (i + 1) * bs) for i in range(len(store) // bs)]
def run(name, make_opt, steps=40):
    torch.manual_seed(0)
    model = E.NestedGIN_eff(None, 4, 256, use_rd=True, graph_pred=False, dropout=0, edge_nest=True, use_cycle=True).to(DEV).train()
    opt = make_opt(model)
    crit = torch.nn.L1Loss()
    def step(i):
        b = store.collate(ids[i % len(ids)])
        opt.zero_grad()
        loss = crit(model(b), b.y.view(-1, 1))
        loss.backward()
        opt.step()
    for i in range(5): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s: %.2f ms/step, %.0f graphs/s" % (name, dt / steps * 1e3, bs * steps / dt), flush=True)
run("reference loop: model(batch) + torch.nn.L1Loss + torch.optim.Adam", lambda m: torch.optim.Adam(m.parameters(), lr=1e-3))
run("model(batch) + torch.nn.L1Loss + esc FlatAdam (direct gradient writes)", lambda m: E.optim.FlatAdam(m.parameters(), lr=1e-3))
def off(m):
    o = E.optim.FlatAdam(m.parameters(), lr=1e-3); o.engine_direct = False; return o
run("model(batch) + torch.nn.L1Loss + esc FlatAdam (accumulate path)", off)
