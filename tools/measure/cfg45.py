import sys, time, torch
import os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]      # the repo root
import esc_gnn_amd as E
from esc_gnn_amd.datasets import synthetic_zinc_graphs, synthetic_ogbmol_graphs, build_feature_dataset
from esc_gnn_amd.zinc_models import NestedGIN_eff as ZincModel
from esc_gnn_amd.ogb_mol_gnn import GNN
DEV = 'cuda:0'
import os
if os.environ.get('ESC_TWO_MIN'):
    from esc_gnn_amd import _native as _nv
    _nv.call('esc_engine_set_two_stream_min_edges', int(os.environ['ESC_TWO_MIN']))
def bench(name, model, store, bs, loss_fn, steps=20):
    opt = E.optim.FlatAdam(model.parameters(), lr=1e-3)
    model.train()
    ids = [torch.arange(i * bs, (i + 1) * bs) for i in range(len(store) // bs)]
    def step(i):
        b = store.collate(ids[i % len(ids)])
        opt.zero_grad()
        loss = loss_fn(model(b), b)
        loss.backward()
        opt.step()
        return b
    for i in range(5): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    edges = 0
    for i in range(steps): edges += step(i).edge_index.size(1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s: %.2f ms/step, %.0f graphs/s, %.0f edges/batch" % (name, dt / steps * 1e3, bs * steps / dt, edges / steps), flush=True)
t0 = time.time()
zg = build_feature_dataset(synthetic_zinc_graphs(0, 1024), 3, use_rd=True, self_loop=False)
print("zinc features %.2fs" % (time.time() - t0))
zs = E.DeviceGraphStore(zg, DEV)
bench("ZINC NestedGIN_eff L=5 bs=128 (model(batch): engine autograd node)", ZincModel(None, num_layers=5).to(DEV), zs, 128,
      lambda p, b: E.ops.l1_loss(p, b.y.view(-1, 1)))
def bench_zinc_engine(store, bs, steps=30):
    from esc_gnn_amd.engine import ZincStepEngine
    model = ZincModel(None, num_layers=5).to(DEV).train()
    opt = E.optim.FlatAdam(model.parameters(), lr=1e-3)
    eng = ZincStepEngine(model)
    ids = [torch.arange(i * bs, (i + 1) * bs) for i in range(len(store) // bs)]
    def step(i):
        b = store.collate(ids[i % len(ids)])
        eng.train_step(b)
        opt.step()
    for i in range(5): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("ZINC NestedGIN_eff L=5 bs=128 (ZincStepEngine.train_step): %.2f ms/step, %.0f graphs/s" % (dt / steps * 1e3, bs * steps / dt), flush=True)
    from esc_gnn_amd.harness import prefetched
    def loop(n):
        for b in prefetched((store.collate(ids[i % len(ids)]) for i in range(n)), DEV, eng.prepare):
            eng.train_step(b)
            opt.step()
    loop(5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loop(steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("ZINC NestedGIN_eff L=5 bs=128 (ZincStepEngine.train_step, next batch collated on a side stream): %.2f ms/step, %.0f graphs/s" % (dt / steps * 1e3, bs * steps / dt), flush=True)
bench_zinc_engine(zs, 128)
mz = ZincModel(None, num_layers=5).to(DEV); mz.step_engine = False
bench("ZINC NestedGIN_eff L=5 bs=128 (per-op path)", mz, zs, 128, lambda p, b: E.ops.l1_loss(p, b.y.view(-1, 1)))
t0 = time.time()
og = build_feature_dataset(synthetic_ogbmol_graphs(0, 1024), 4, use_rd=True, self_loop=True)
print("molhiv features %.2fs" % (time.time() - t0))
os_ = E.DeviceGraphStore(og, DEV)
m = GNN("ogbg-molhiv", 1, num_layer=6, emb_dim=300, gnn_type="gin_eff", virtual_node=True, residual=True, drop_ratio=0.65,
        use_rd=True).to(DEV)
def bench_ogb_engine(store, bs, steps=30):
    from esc_gnn_amd.engine import OgbStepEngine
    model = GNN("ogbg-molhiv", 1, num_layer=6, emb_dim=300, gnn_type="gin_eff", virtual_node=True, residual=True, drop_ratio=0.65,
                use_rd=True).to(DEV).train()
    opt = E.optim.FlatAdam(model.parameters(), lr=1e-3)
    eng = OgbStepEngine(model)
    ids = [torch.arange(i * bs, (i + 1) * bs) for i in range(len(store) // bs)]
    def step(i):
        b = store.collate(ids[i % len(ids)])
        eng.train_step(b)
        opt.step()
    for i in range(5): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("ogbg-molhiv gin_eff h=4 L=6 emb=300 bs=256 drop 0.65 (OgbStepEngine.train_step): %.2f ms/step, %.0f graphs/s" % (dt / steps * 1e3, bs * steps / dt), flush=True)
    from esc_gnn_amd.harness import prefetched
    def loop(n):
        for b in prefetched((store.collate(ids[i % len(ids)]) for i in range(n)), DEV, eng.prepare):
            eng.train_step(b)
            opt.step()
    loop(5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loop(steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("ogbg-molhiv gin_eff h=4 L=6 emb=300 bs=256 drop 0.65 (OgbStepEngine.train_step, next batch collated on a side stream): %.2f ms/step, %.0f graphs/s" % (dt / steps * 1e3, bs * steps / dt), flush=True)
bench_ogb_engine(os_, 256)
mo = GNN("ogbg-molhiv", 1, num_layer=6, emb_dim=300, gnn_type="gin_eff", virtual_node=True, residual=True, drop_ratio=0.65,
         use_rd=True).to(DEV); mo.step_engine = False
bench("ogbg-molhiv gin_eff h=4 L=6 emb=300 bs=256 (per-op path)", mo, os_, 256,
      lambda p, b: E.ops.bce_with_logits_loss(p, b.y.view(-1, 1)))
bench("ogbg-molhiv gin_eff h=4 L=6 emb=300 bs=256 (model(batch): engine autograd node)", m, os_, 256,
      lambda p, b: E.ops.bce_with_logits_loss(p, b.y.view(-1, 1)))
