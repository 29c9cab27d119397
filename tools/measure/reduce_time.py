import sys, ctypes, torch
import os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]      # the repo root
import esc_gnn_amd as E
from esc_gnn_amd import _native as nv
from ctypes import c_void_p, c_int64, c_int32
class Job(ctypes.Structure):
    _fields_ = [("slabs", c_void_p), ("n", c_int64), ("splits", c_int32), ("cols", c_int64), ("dw", c_void_p), ("ld_dw", c_int64),
                ("db_part", c_void_p), ("rows", c_int64), ("db", c_void_p)]
dev = 'cuda:0'
lib = nv.lib()
lib.esc_slab_reduce_jobs.argtypes = [c_void_p, ctypes.c_int, c_void_p]
def bench(name, specs, reps=30, sets=6):
    # specs: list of (rows N_out, cols K_in, splits); `sets` rotating copies so the slabs are not cache-resident
    all_jobs, keep = [], []
    for _ in range(sets):
        arr = (Job * len(specs))()
        for j, (r, c, sp) in enumerate(specs):
            slabs = torch.randn(sp * (r * c + r), device=dev); dw = torch.empty(r, c, device=dev); db = torch.empty(r, device=dev)
            keep += [slabs, dw, db]
            arr[j].slabs, arr[j].n, arr[j].splits, arr[j].cols = slabs.data_ptr(), r * c, sp, c
            arr[j].dw, arr[j].ld_dw = dw.data_ptr(), c
            arr[j].db_part, arr[j].rows, arr[j].db = slabs.data_ptr() + 4 * sp * r * c, r, db.data_ptr()
        all_jobs.append(arr)
    vol = sum(sp * (r * c + r) * 4 + r * c * 4 for r, c, sp in specs)
    s = nv.stream()
    for a in all_jobs: lib.esc_slab_reduce_jobs(a, len(specs), s)
    torch.cuda.synchronize()
    nv.prof_reset("linear"); nv.prof_enable("linear", True)
    for i in range(reps): lib.esc_slab_reduce_jobs(all_jobs[i % sets], len(specs), s)
    torch.cuda.synchronize(); nv.prof_enable("linear", False)
    n, ms = nv.prof_read("linear")
    us = ms / reps * 1e3
    print("%-58s %7.1f us  %6.1f MB  %6.0f GB/s" % (name, us, vol / 1e6, vol / us / 1e3), flush=True)
bench("one edge job 256x256 x60 splits", [(256, 256, 60)])
bench("four edge jobs", [(256, 256, 60)] * 4)
bench("one node job 256x256 x15 splits", [(256, 256, 15)])
bench("eight node jobs", [(256, 256, 15)] * 8)
bench("node-side set (8 node + lin1 256x1280 x4 + small)", [(256, 256, 15)] * 8 + [(256, 1024, 4), (256, 256, 15), (256, 10, 75), (1, 256, 19)])
bench("edge-side final (zlin x60)", [(256, 256, 60)])
bench("edge + node all (r01-style single launch)", [(256, 256, 60)] * 4 + [(256, 256, 15)] * 8 + [(256, 1024, 4), (256, 256, 15)])
