import sys, os, torch, faulthandler, functools
faulthandler.dump_traceback_later(60, exit=True)
print = functools.partial(print, flush=True)
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import esc_gnn_amd as E
from esc_gnn_amd.datasets import build_feature_dataset, synthetic_ogbmol_graphs
from esc_gnn_amd.engine import OgbStepEngine, _ogb_batch
from esc_gnn_amd.ogb_mol_gnn import GNN, ATOM_FEATURE_DIMS, BOND_FEATURE_DIMS
dev = torch.device("cuda:0")
graphs = build_feature_dataset(synthetic_ogbmol_graphs(0, 112), 2, use_rd=True, self_loop=True)
store = E.DeviceGraphStore(graphs, dev)
print("int_ranges", getattr(store, "int_ranges", None))
torch.manual_seed(5)
model = GNN("ogbg-molhiv", 1, num_layer=2, emb_dim=32, gnn_type="gin_eff", virtual_node=True, residual=True, drop_ratio=0.3, use_rd=True).to(dev).train()
b = store.collate(torch.arange(16))
print("x", b.x.shape, b.x.dtype, b.x.min(0).values.tolist(), b.x.max(0).values.tolist())
print("ea", b.edge_attr.shape, b.edge_attr.min(0).values.tolist(), b.edge_attr.max(0).values.tolist())
bb, keep = _ogb_batch(model, b, True, 0)
y, plan, gptr, pa, pb, zero = keep
torch.cuda.synchronize()
for name, p, dims in (("atoms", pa, ATOM_FEATURE_DIMS), ("bonds", pb, BOND_FEATURE_DIMS)):
    cp, cc, cr = p["col_ptr"].cpu(), p["c_col"].cpu(), p["c_row"].cpu()
    n = p["entries"]
    print(name, "entries", n, "rows", p["rows"], "col_ptr len", cp.numel(), "last", int(cp[-1]), "monotone", bool((cp[1:] >= cp[:-1]).all()),
          "c_col sorted", bool((cc[1:] >= cc[:-1]).all()), "c_col range", int(cc.min()), int(cc.max()), "c_row range", int(cr.min()), int(cr.max()))
    cnt = torch.bincount(cc.long(), minlength=p["rows"])
    print("   counts match", bool((cnt == (cp[1:] - cp[:-1])).all()))
print("N E Z G", bb.N, bb.E, bb.Z, bb.G, "gptr", gptr.cpu().tolist())
eng = OgbStepEngine(model)
need = E._native.lib().esc_ogb_workspace_floats
print("workspace floats", eng._workspace(bb).numel())
loss, pred = eng.train_step(b, return_pred=True)
print("loss", float(loss), pred.view(-1).tolist())
