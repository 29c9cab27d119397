import sys, time, torch
import os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]      # the repo root
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
def step(i):
    b = store.collate(ids[i % len(ids)])
    eng.train_step(b)
    opt.step()
if os.environ.get("PREFETCH"):
    from esc_gnn_amd.harness import prefetched
    def loop(n):
        for b in prefetched((store.collate(ids[i % len(ids)]) for i in range(n)), DEV, eng.prepare):
            eng.train_step(b)
            opt.step()
    loop(3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loop(20)
else:
    for i in range(3): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): step(i)
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 20 * 1e3)
