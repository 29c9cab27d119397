"""The reference's own loop, loader included (run_graphcount.py:453-455,487-505): `for data in DataLoader(dataset, 128,
shuffle=True): data = data.to(device); ...` — host collate (device=None: what the reference's loader does) against the
device-pinned DataLoader, per step INCLUDING the loader.  -> profiles/r03_dropin_loader_times.txt"""
import os, sys, time, torch
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import esc_gnn_amd as E
from esc_gnn_amd.datasets import build_count_dataset
DEV = "cuda:0"
graphs = build_count_dataset(0, 1024, h=3, use_rd=True, self_loop=True)
bs = 128


def run(name, loader, make_opt, epochs=4):
    torch.manual_seed(0)
    model = E.NestedGIN_eff(None, 4, 256, use_rd=True, graph_pred=False, dropout=0, edge_nest=True, use_cycle=True).to(DEV).train()
    opt = make_opt(model)
    crit = torch.nn.L1Loss()

    def epoch():
        n = 0
        for data in loader:
            data = data.to(DEV)
            opt.zero_grad()
            loss = crit(model(data), data.y.view(-1, 1))
            loss.backward()
            opt.step()
            n += 1
        return n
    epoch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps = sum(epoch() for _ in range(epochs))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    # the loader alone (collate + .to): what the loop pays per batch before the model sees it
    torch.cuda.synchronize(); t1 = time.perf_counter()
    k = 0
    for _ in range(epochs):
        for data in loader:
            data = data.to(DEV); k += 1
    torch.cuda.synchronize(); dl = time.perf_counter() - t1
    print("%s: %.2f ms/step incl. loader (%.0f graphs/s); loader alone %.2f ms/batch" % (name, dt / steps * 1e3, bs * steps / dt, dl / k * 1e3), flush=True)


adam = lambda m: torch.optim.Adam(m.parameters(), lr=1e-3)
flat = lambda m: E.optim.FlatAdam(m.parameters(), lr=1e-3)
run("host collate (reference loader, device=None) + torch Adam", E.DataLoader(graphs, batch_size=bs, shuffle=True, device=None), adam)
run("device-pinned DataLoader + torch Adam", E.DataLoader(graphs, batch_size=bs, shuffle=True), adam)
run("device-pinned DataLoader + FlatAdam", E.DataLoader(graphs, batch_size=bs, shuffle=True), flat)
