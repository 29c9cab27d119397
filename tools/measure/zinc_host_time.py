"""ZincStepEngine: host enqueue time per step against the wall time (is the config-4 loop host-bound?)"""
import sys, time, torch
import os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]      # the repo root
import esc_gnn_amd as E
from esc_gnn_amd.datasets import synthetic_zinc_graphs, build_feature_dataset
from esc_gnn_amd.zinc_models import NestedGIN_eff as ZincModel
from esc_gnn_amd.engine import ZincStepEngine
DEV = 'cuda:0'
if os.environ.get('ESC_TWO_MIN'):
    from esc_gnn_amd import _native as _nv
    _nv.call('esc_engine_set_two_stream_min_edges', int(os.environ['ESC_TWO_MIN']))
og = build_feature_dataset(synthetic_zinc_graphs(0, 1024), 3, use_rd=True, self_loop=False)
store = E.DeviceGraphStore(og, DEV)
bs = 128
model = ZincModel(None, num_layers=5).to(DEV).train()
opt = E.optim.FlatAdam(model.parameters(), lr=1e-3)
eng = ZincStepEngine(model)
ids = [torch.arange(i * bs, (i + 1) * bs) for i in range(len(store) // bs)]
T = [0.0, 0.0, 0.0]
def step(i, timed=False):
    t0 = time.perf_counter()
    b = store.collate(ids[i % len(ids)])
    t1 = time.perf_counter()
    eng.train_step(b)
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    if timed:
        T[0] += t1 - t0; T[1] += t2 - t1; T[2] += t3 - t2
for i in range(5): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 40
for i in range(N): step(i, True)
th = time.perf_counter() - t0
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("ZINC step: wall %.3f ms, host enqueue %.3f ms (collate %.0f us, train_step %.0f us, adam %.0f us)" % (dt / N * 1e3, th / N * 1e3, T[0] / N * 1e6, T[1] / N * 1e6, T[2] / N * 1e6))
