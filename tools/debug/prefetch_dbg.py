import sys, os, torch, faulthandler, functools
faulthandler.dump_traceback_later(50, exit=True)
print = functools.partial(print, flush=True)
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import esc_gnn_amd as E
from esc_gnn_amd.datasets import build_feature_dataset, synthetic_ogbmol_graphs
from esc_gnn_amd.engine import OgbStepEngine
from esc_gnn_amd.ogb_mol_gnn import GNN
from esc_gnn_amd.harness import prefetched
from esc_gnn_amd import _native as nv
dev = torch.device("cuda:0")
graphs = build_feature_dataset(synthetic_ogbmol_graphs(0, 112), 2, use_rd=True, self_loop=True)
store = E.DeviceGraphStore(graphs, dev)
print("y", [float(g.y) for g in graphs[:32]])
def run(pref, two, drop=0.3, steps=14):
    nv.call("esc_engine_set_two_stream_min_edges", 0 if two else 12000)
    torch.manual_seed(5)
    model = GNN("ogbg-molhiv", 1, num_layer=2, emb_dim=32, gnn_type="gin_eff", virtual_node=True, residual=True,
                drop_ratio=drop, use_rd=True).to(dev).train()
    eng = OgbStepEngine(model)
    opt = E.optim.FlatAdam(model.parameters(), lr=1e-3)
    bs = 16
    ids = [torch.arange(i * bs, (i + 1) * bs) % len(store) for i in range(steps)]
    batches = (store.collate(i) for i in ids)
    if pref:
        batches = prefetched(batches, dev, eng.prepare)
    out = []
    for b in batches:
        loss, pred = eng.train_step(b, return_pred=True)
        opt.step()
        out.append((float(loss), float(pred.abs().max()), float(pred.mean())))
        print('  step', len(out), out[-1])
    torch.cuda.synchronize()
    return out
for two in (0, 1):
    for pref in (0, 1):
        for rep in range(2):
            r = run(pref, two)
            print("two", two, "pref", pref, " ".join("%.6f" % a[0] for a in r))
    print("pred absmax", " ".join("%.4f" % a[1] for a in r))
# engine vs per-op, no dropout
torch.manual_seed(5)
import copy
m1 = GNN("ogbg-molhiv", 1, num_layer=2, emb_dim=32, gnn_type="gin_eff", virtual_node=True, residual=True, drop_ratio=0.0, use_rd=True).to(dev).train()
m2 = copy.deepcopy(m1); m1.step_engine = False
b = store.collate(torch.arange(16))
p1 = m1(b); l1 = E.ops.bce_with_logits_loss(p1, b.y.view(-1, 1))
l2, p2 = OgbStepEngine(m2).train_step(store.collate(torch.arange(16)), return_pred=True)
print("per-op", float(l1), p1.view(-1)[:6].tolist()); print("engine", float(l2), p2.view(-1)[:6].tolist())
