"""ORACLE / TEST INFRASTRUCTURE — golden-vector generator (runs ONLY in the build container).

Imports the *unmodified* reference modules /root/reference/utils_edge_efficient.py and
/root/reference/batch.py under the PyG stand-in in oracle/pyg_shim, runs them on
deterministic inputs and stores inputs + outputs as small .npz fixtures in tests/golden/.
Every case is simultaneously checked against oracle/ref_features.py (our restatement) so a
fixture is never written from a run where the two disagree.

    python oracle/make_golden.py            # regenerates tests/golden/*.npz

Only data (inputs / expected outputs) is committed — never reference source.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "pyg_shim"), REF, HERE]

import utils_edge_efficient as ref_feat  # noqa: E402  (the reference, imported in place)
from batch import Batch as RefBatch  # noqa: E402      (the reference's batch.py)
from torch_geometric.data import Data as ShimData  # noqa: E402

import graph_sources as gs  # noqa: E402
import ref_features as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def run_reference(n, s, t, h, use_rd, self_loop, edge_attr=None, x=None):
    d = ShimData(x=torch.ones(n, 1) if x is None else x,
                 edge_index=torch.tensor(np.stack([s, t])),
                 edge_attr=edge_attr, y=None)
    o = ref_feat.create_subgraphs(d, h, use_rd=use_rd, self_loop=self_loop)
    return o


def check_oracle(n, s, t, h, use_rd, self_loop, o):
    mine = orc.encode_graph(s, t, n, h, use_rd, self_loop)
    ok = (np.array_equal(mine["edge_src"], o.edge_index[0].numpy())
          and np.array_equal(mine["edge_dst"], o.edge_index[1].numpy())
          and np.array_equal(mine["pos_enc"], o.pos_enc.numpy())
          and np.array_equal(mine["pos_index"], o.pos_index.numpy())
          and np.array_equal(mine["pos_batch"], o.pos_batch.numpy()))
    return ok


class Pack(object):
    """Ragged cases flattened into a handful of int32 arrays."""

    def __init__(self):
        self.meta = []      # name, n, h, use_rd, self_loop
        self.cols = {k: [] for k in ("in_src", "in_dst", "out_src", "out_dst", "pos_enc", "pos_index", "pos_batch")}

    def add(self, name, n, s, t, h, use_rd, self_loop, o):
        self.meta.append((name, n, h, int(use_rd), int(self_loop)))
        self.cols["in_src"].append(s)
        self.cols["in_dst"].append(t)
        self.cols["out_src"].append(o.edge_index[0].numpy())
        self.cols["out_dst"].append(o.edge_index[1].numpy())
        self.cols["pos_enc"].append(o.pos_enc.numpy())
        self.cols["pos_index"].append(o.pos_index.numpy())
        self.cols["pos_batch"].append(o.pos_batch.numpy())

    def save(self, path):
        arrs = {"names": np.array([m[0] for m in self.meta]),
                "meta": np.array([m[1:] for m in self.meta], dtype=np.int32)}
        for k, parts in self.cols.items():
            arrs[k] = np.concatenate(parts).astype(np.int32) if parts else np.zeros(0, np.int32)
            arrs[k + "_ptr"] = np.cumsum([0] + [len(p) for p in parts]).astype(np.int64)
        np.savez_compressed(path, **arrs)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(self.meta), "cases")


CONFIGS = {"count": (3, True, True), "deep": (4, True, True), "zinc": (3, True, False),
           "sr": (3, False, True), "h1": (1, True, True), "h2": (2, True, True),
           "plain": (3, False, False)}



def collate_molhiv():
    """molhiv-like batch at the h of BASELINE config 5: x int64[n,9], edge_attr int64[m,3], self loops (attr filled with 1),
    h=4, resistance distance (run_ogb_mol.py pre-transform).  Four molecule-like graphs of 12-18 atoms."""
    sys.path.insert(0, HERE)
    import ref_model as rmod
    rng = np.random.RandomState(13)
    datas, store = [], {}
    for j, seed in enumerate((31, 32, 33, 34)):
        n, s, t = gs.molecule_like_graph(seed, 12, 18)
        xs = np.stack([rng.randint(0, d, size=n) for d in rmod.ATOM_FEATURE_DIMS], axis=1)
        ea = np.stack([rng.randint(0, d, size=s.shape[0]) for d in rmod.BOND_FEATURE_DIMS], axis=1)
        d = ShimData(x=torch.tensor(xs), edge_index=torch.tensor(np.stack([s, t])), edge_attr=torch.tensor(ea),
                     y=torch.tensor([[float(rng.rand() > 0.5)]]))
        o = ref_feat.create_subgraphs(d, 4, use_rd=True, self_loop=True)
        datas.append(o)
        for k in o.keys:
            store["g%d_%s" % (j, k)] = o[k].numpy()
    b = RefBatch.from_data_list(datas)
    store["keys"] = np.array(sorted(b.keys))
    for k in b.keys:
        store["batch_" + k] = b[k].numpy()
    store["num_graphs"] = np.int64(b.num_graphs)
    np.savez_compressed(os.path.join(OUT, "collate_molhiv4.npz"), **store)
    print("wrote collate_molhiv4.npz (h=4)", sorted(b.keys))


def main():
    os.makedirs(OUT, exist_ok=True)
    t0 = time.time()
    bad = []

    def do(pack, name, n, s, t, cfg):
        h, rd, sl = CONFIGS[cfg]
        o = run_reference(n, s, t, h, rd, sl)
        if not check_oracle(n, s, t, h, rd, sl, o):
            bad.append((name, cfg))
        pack.add("%s/%s" % (name, cfg), n, s, t, h, rd, sl, o)
        return o

    # 1. hand cases x every config
    p = Pack()
    for name, (n, s, t) in gs.hand_cases().items():
        for cfg in CONFIGS:
            do(p, name, n, s, t, cfg)
    p.save(os.path.join(OUT, "features_hand.npz"))

    # 2. count-shaped random regular graphs
    p = Pack()
    for g in range(8):
        n, s, t = gs.count_shape_graph(g)
        for cfg in ("count",) + (("deep", "h2") if g < 4 else ()):
            do(p, "rrg%d" % g, n, s, t, cfg)
    p.save(os.path.join(OUT, "features_count.npz"))

    # 3. molecule-like trees with rings
    p = Pack()
    for seed in range(6):
        n, s, t = gs.molecule_like_graph(seed)
        for cfg in ("zinc", "deep"):
            do(p, "mol%d" % seed, n, s, t, cfg)
    p.save(os.path.join(OUT, "features_mol.npz"))

    # 4. random directed graphs incl. pre-existing self loops
    p = Pack()
    for seed in range(4):
        n, s, t = gs.random_directed_graph(seed, 8 + 2 * seed, 20 + 6 * seed)
        for cfg in ("sr", "plain", "h2", "zinc"):
            do(p, "dir%d" % seed, n, s, t, cfg)
    p.save(os.path.join(OUT, "features_directed.npz"))

    # 5. the graphs the reference ships: SR25 (2 of 15) and EXP (first 3)
    p = Pack()
    sr = gs.read_g6(os.path.join(REF, "data/sr25/raw/sr251256.g6"))
    for i in (0, 7):
        n, s, t = sr[i]
        do(p, "sr25_%d" % i, n, s, t, "sr")
    n, s, t = sr[3]
    do(p, "sr25_3", n, s, t, "count")
    for i, (n, s, t) in enumerate(gs.read_exp_txt(os.path.join(REF, "data/EXP/GRAPHSAT.txt"), 3)):
        do(p, "exp%d" % i, n, s, t, "sr")
    p.save(os.path.join(OUT, "features_shipped.npz"))

    # 6. edge_attr pass-through (loops filled with 1) — 1-D and 2-D attributes
    n, s, t = gs.molecule_like_graph(11)
    rng = np.random.RandomState(5)
    ea1 = torch.tensor(rng.randint(0, 4, size=s.shape[0]))
    ea2 = torch.tensor(rng.randint(0, 5, size=(s.shape[0], 3)))
    o1 = run_reference(n, s, t, 2, True, True, edge_attr=ea1)
    o2 = run_reference(n, s, t, 2, True, True, edge_attr=ea2)
    np.savez_compressed(os.path.join(OUT, "features_edge_attr.npz"), n=n, src=s, dst=t,
                        ea1_in=ea1.numpy(), ea1_out=o1.edge_attr.numpy(),
                        ea2_in=ea2.numpy(), ea2_out=o2.edge_attr.numpy(),
                        out_src=o1.edge_index[0].numpy(), out_dst=o1.edge_index[1].numpy())

    # 7. collate goldens: reference batch.py on reference-produced Data objects
    for tag, ids, cfg in (("count3", (0, 1, 2), "count"), ("mixed4", (3, 4, 5, 6), "count")):
        h, rd, sl = CONFIGS[cfg]
        datas, store = [], {}
        for j, g in enumerate(ids):
            n, s, t = gs.count_shape_graph(g)
            x = torch.ones(n, 10)
            d = ShimData(x=x, edge_index=torch.tensor(np.stack([s, t])),
                         y=torch.tensor(gs.triangle_counts(n, s, t)))
            o = ref_feat.create_subgraphs(d, h, use_rd=rd, self_loop=sl)
            datas.append(o)
            for k in o.keys:
                store["g%d_%s" % (j, k)] = o[k].numpy()
        b = RefBatch.from_data_list(datas)
        store["keys"] = np.array(sorted(b.keys))
        for k in b.keys:
            store["batch_" + k] = b[k].numpy()
        store["num_graphs"] = np.int64(b.num_graphs)
        np.savez_compressed(os.path.join(OUT, "collate_%s.npz" % tag), **store)
        print("wrote collate_%s.npz" % tag, sorted(b.keys))

    # 8. ZINC-like batch: integer node types, integer edge types, graph-level y, no self loops (run_zinc.py:76)
    rng = np.random.RandomState(9)
    datas, store = [], {}
    for j, seed in enumerate((21, 22, 23)):
        n, s, t = gs.molecule_like_graph(seed)
        d = ShimData(x=torch.tensor(rng.randint(0, 28, size=n)), edge_index=torch.tensor(np.stack([s, t])),
                     edge_attr=torch.tensor(rng.randint(0, 4, size=s.shape[0])), y=torch.tensor([float(rng.randn())]))
        o = ref_feat.create_subgraphs(d, 3, use_rd=True, self_loop=False)
        datas.append(o)
        for k in o.keys:
            store["g%d_%s" % (j, k)] = o[k].numpy()
    b = RefBatch.from_data_list(datas)
    store["keys"] = np.array(sorted(b.keys))
    for k in b.keys:
        store["batch_" + k] = b[k].numpy()
    store["num_graphs"] = np.int64(b.num_graphs)
    np.savez_compressed(os.path.join(OUT, "collate_zinc3.npz"), **store)
    print("wrote collate_zinc3.npz", sorted(b.keys))

    # 9. molhiv-like batch (config 5)
    collate_molhiv()


    print("oracle disagreements:", bad, " elapsed %.1fs" % (time.time() - t0))
    if bad:
        raise SystemExit(1)


if __name__ == "__main__":
    if "--only-molhiv" in sys.argv:
        collate_molhiv()
    else:
        main()
