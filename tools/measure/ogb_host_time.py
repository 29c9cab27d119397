"""host-side cost of one config-5 step, phase by phase (the device is drained before every step, so no call ever waits
for queue space): collate, embedding plans (_ogb_batch), the engine call (all launches of the step), FlatAdam."""
import sys, time, torch, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import esc_gnn_amd as E
from esc_gnn_amd.datasets import synthetic_ogbmol_graphs, build_feature_dataset
from esc_gnn_amd.ogb_mol_gnn import GNN
from esc_gnn_amd.engine import OgbStepEngine
DEV = 'cuda:0'
og = build_feature_dataset(synthetic_ogbmol_graphs(0, 1024), 4, use_rd=True, self_loop=True)
store = E.DeviceGraphStore(og, DEV)
bs = 256
model = GNN("ogbg-molhiv", 1, num_layer=6, emb_dim=300, gnn_type="gin_eff", virtual_node=True, residual=True, drop_ratio=0.65,
            use_rd=True).to(DEV).train()
opt = E.optim.FlatAdam(model.parameters(), lr=1e-3)
eng = OgbStepEngine(model)
ids = [torch.arange(i * bs, (i + 1) * bs) for i in range(len(store) // bs)]
T = dict(collate=0.0, prepare=0.0, step=0.0, adam=0.0, device=0.0)
n = 0
for i in range(25):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b = store.collate(ids[i % len(ids)]); t1 = time.perf_counter()
    eng.prepare(b); t2 = time.perf_counter()
    eng.train_step(b); t3 = time.perf_counter()
    opt.step(); t4 = time.perf_counter()
    torch.cuda.synchronize(); t5 = time.perf_counter()
    if i >= 5:
        n += 1
        for k, v in (("collate", t1 - t0), ("prepare", t2 - t1), ("step", t3 - t2), ("adam", t4 - t3), ("device", t5 - t0)):
            T[k] += v
print("host ms per step: " + ", ".join("%s %.3f" % (k, T[k] / n * 1e3) for k in ("collate", "prepare", "step", "adam")) +
      "; sum %.3f; enqueue-to-drained %.3f" % (sum(T[k] for k in ("collate", "prepare", "step", "adam")) / n * 1e3, T["device"] / n * 1e3))
