import sys, time, torch
import os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]      # the repo root
import esc_gnn_amd as E
from esc_gnn_amd.datasets import synthetic_zinc_graphs, build_feature_dataset
from esc_gnn_amd.zinc_models import NestedGIN_eff as ZincModel
from esc_gnn_amd.engine import ZincStepEngine
DEV = 'cuda:0'
og = build_feature_dataset(synthetic_zinc_graphs(0, 1024), 3, use_rd=True, self_loop=False)
store = E.DeviceGraphStore(og, DEV)
bs = 128
model = ZincModel(None, num_layers=5).to(DEV).train()
opt = E.optim.FlatAdam(model.parameters(), lr=1e-3)
eng = ZincStepEngine(model)
ids = [torch.arange(i * bs, (i + 1) * bs) for i in range(len(store) // bs)]
def step(i):
    b = store.collate(ids[i % len(ids)])
    eng.train_step(b)
    opt.step()
for i in range(3): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20): step(i)
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 20 * 1e3)
