import sys, time, torch
import os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]      # the repo root
import esc_gnn_amd as E
from esc_gnn_amd import _native as nv
from esc_gnn_amd.datasets import synthetic_count_graphs, synthetic_ogbmol_graphs
from esc_gnn_amd.utils_edge_efficient import encode_edge_lists
def run(name, raw, h, sl):
    nodes = [int(d.num_nodes) for d in raw]; edges = [d.edge_index for d in raw]
    encode_edge_lists(nodes[:64], edges[:64], h, True, sl)
    nv.prof_enable("features", True); nv.prof_reset("features")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = encode_edge_lists(nodes, edges, h, True, sl)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms = nv.prof_read_all("features")
    nv.prof_enable("features", False)
    print("%s: %d graphs, %d out edges: wall %.1f ms, kernel launches (ms): %s  sum %.2f ms" % (
        name, len(raw), sum(o[0].size(1) for o in out), dt * 1e3, ["%.3f" % m for m in ms] if ms is not None else None,
        sum(ms) if ms is not None else -1), flush=True)
run("count h=3", synthetic_count_graphs(0, 1500), 3, True)
run("molhiv h=4", synthetic_ogbmol_graphs(0, 4096), 4, True)
