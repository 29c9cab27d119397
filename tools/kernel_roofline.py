#!/usr/bin/env python3
"""Per-kernel roofline table at the BASELINE config-1 shapes (N=2400 nodes, E=15200 edges, Z≈5.5e5 histogram entries,
H=256), each kernel launched back to back on a real collated batch and timed with the library's dispatch-event hook.

    python tools/kernel_roofline.py > profiles/r01_kernel_roofline.txt          (on a GPU box)

Algorithmic bytes / flops are the SURVEY §8(d) figures.  Back-to-back launches on ONE set of buffers keep
operands warm in L2 (8 x 4 MB) / Infinity Cache (256 MB): those rows are labelled "cache-resident" — they are ceilings, not
HBM figures (a 36 MB working set at 7.1 TB/s is above the ~6.3 TB/s the HBM sustains).  The "cold" rows rotate through
enough buffer sets (> 320 MB in total) that every launch streams its operands from HBM.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import esc_gnn_amd as E  # noqa: E402
from esc_gnn_amd import _native as nv  # noqa: E402
from esc_gnn_amd.datasets import build_count_dataset  # noqa: E402

HBM, MFMA = 8000.0, 157.3
dev = "cuda:0"


def timed(kind, fn, n=60):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    nv.prof_reset(kind)
    nv.prof_enable(kind, True)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    nv.prof_enable(kind, False)
    launches, ms = nv.prof_read(kind)
    return ms / n * 1e3, launches / n


def timed_rotating(kind, fns, n=60):
    """like timed(), launch i uses buffer set i % len(fns)"""
    for f in fns:
        f()
    torch.cuda.synchronize()
    nv.prof_reset(kind)
    nv.prof_enable(kind, True)
    for i in range(n):
        fns[i % len(fns)]()
    torch.cuda.synchronize()
    nv.prof_enable(kind, False)
    launches, ms = nv.prof_read(kind)
    return ms / n * 1e3, launches / n


def timed_wall(fn, n=100):
    """stream time per call of n back-to-back calls (torch events around the whole run: launch gaps included) — the only
    clock that treats a library GEMM and ours alike"""
    for _ in range(10):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    graphs = build_count_dataset(0, 256, h=3)
    store = E.DeviceGraphStore(graphs, dev)
    b = store.collate(torch.arange(128))
    plan = E.plan_of(b)
    N, Ee, Z, H = plan.num_nodes, plan.num_edges, plan.nnz, 256
    s = nv.stream()
    rows = []

    def add(name, us, launches, nbytes=None, flops=None):
        if nbytes is not None:
            gbs = nbytes / us / 1e3
            rows.append((name, "%.1f" % us, "%d" % launches, "%.1f MB" % (nbytes / 1e6), "%.0f GB/s" % gbs, "%.0f %% of HBM" % (100 * gbs / HBM)))
        else:
            tf = flops / us / 1e6
            rows.append((name, "%.1f" % us, "%d" % launches, "%.2f GFLOP" % (flops / 1e9), "%.1f TFLOP/s" % tf, "%.0f %% of fp32 MFMA" % (100 * tf / MFMA)))

    x = torch.randn(N, H, device=dev)
    e = torch.randn(Ee, H, device=dev)
    eps = torch.zeros(1, device=dev)
    out = torch.empty(N, H, device=dev)
    us, k = timed("agg_fwd", lambda: nv.call("esc_gine_aggregate_fwd", nv.ptr(x), H, nv.ptr(e), H, nv.ptr(plan.in_ptr), nv.ptr(plan.in_edge), nv.ptr(plan.in_src), nv.ptr(eps), N, H, nv.ptr(out), H, s))
    add("aggregate forward (scatter-add), cache-resident", us, k, 2 * Ee * H * 4 + 2 * N * H * 4 + Ee * 8 + (N + 1) * 4)
    sets = [(torch.randn(N, H, device=dev), torch.randn(Ee, H, device=dev), torch.empty(N, H, device=dev)) for _ in range(10)]
    fns = [(lambda a=a, b_=b_, c=c: nv.call("esc_gine_aggregate_fwd", nv.ptr(a), H, nv.ptr(b_), H, nv.ptr(plan.in_ptr), nv.ptr(plan.in_edge),
                                            nv.ptr(plan.in_src), nv.ptr(eps), N, H, nv.ptr(c), H, s)) for a, b_, c in sets]
    us, k = timed_rotating("agg_fwd", fns)
    add("aggregate forward (scatter-add), cold: 10 buffer sets = 360 MB", us, k, 2 * Ee * H * 4 + 2 * N * H * 4 + Ee * 8 + (N + 1) * 4)
    del sets, fns
    g = torch.randn(N, H, device=dev)
    de = torch.empty(Ee, H, device=dev)
    dx = torch.zeros(N, H, device=dev)
    dp = torch.empty(2 * N, device=dev)
    us, k = timed("agg_bwd", lambda: nv.call("esc_gine_aggregate_bwd", nv.ptr(x), H, nv.ptr(e), H, nv.ptr(g), H, nv.ptr(plan.out_ptr), nv.ptr(plan.out_edge), nv.ptr(plan.out_dst), nv.ptr(eps), N, H, nv.ptr(de), H, nv.ptr(dx), H, 1, nv.ptr(dp), s))
    add("aggregate backward, cache-resident", us, k, 2 * Ee * H * 4 + 3 * N * H * 4 + Ee * 8)
    sets = [(torch.randn(N, H, device=dev), torch.randn(Ee, H, device=dev), torch.randn(N, H, device=dev), torch.empty(Ee, H, device=dev),
             torch.zeros(N, H, device=dev)) for _ in range(9)]
    fns = [(lambda a=a, b_=b_, c=c, d=d, f=f: nv.call("esc_gine_aggregate_bwd", nv.ptr(a), H, nv.ptr(b_), H, nv.ptr(c), H, nv.ptr(plan.out_ptr),
                                                      nv.ptr(plan.out_edge), nv.ptr(plan.out_dst), nv.ptr(eps), N, H, nv.ptr(d), H, nv.ptr(f), H, 1,
                                                      nv.ptr(dp), s)) for a, b_, c, d, f in sets]
    us, k = timed_rotating("agg_bwd", fns)
    add("aggregate backward, cold: 9 buffer sets = 350 MB", us, k, 2 * Ee * H * 4 + 3 * N * H * 4 + Ee * 8)
    del sets, fns
    table = torch.randn(1800, H, device=dev)
    zb = torch.empty(Ee, H, device=dev)
    us, k = timed("bag_fwd", lambda: nv.call("esc_bag_fwd", nv.ptr(table), H, nv.ptr(plan.row_ptr), nv.ptr(plan.bag_idx), nv.ptr(plan.bag_val), Ee, nv.ptr(zb), H, s))
    add("bag forward, cache-resident (HBM bytes)", us, k, Z * 8 + (Ee + 1) * 4 + 1800 * H * 4 + Ee * H * 4)
    add("bag forward (L2 row traffic Z*H*4)", us, k, Z * H * 4)
    us, k = timed("bag_fwd", lambda: nv.call("esc_bag_fwd_rows", nv.ptr(table), 1800, H, nv.ptr(plan.row_ptr), nv.ptr(plan.bag_idx), nv.ptr(plan.bag_val), Ee, nv.ptr(zb), H, 0, None, s))
    add("bag forward through esc_bag_fwd_rows (LDS-staged with ESC_BAG_TILED=1, else the kernel above)", us, k, Z * 8 + (Ee + 1) * 4 + 1800 * H * 4 + Ee * H * 4)
    if int(nv.lib().esc_bag_fwd_stats_block_rows(nv.ptr(table), 1800, H, nv.ptr(zb), H, Ee)):     # (only with ESC_BAG_TILED=1)
        st = torch.empty((Ee // 128 + 1) * H * 2, device=dev)
        us, k = timed("bag_fwd", lambda: nv.call("esc_bag_fwd_rows", nv.ptr(table), 1800, H, nv.ptr(plan.row_ptr), nv.ptr(plan.bag_idx), nv.ptr(plan.bag_val), Ee, nv.ptr(zb), H, 0, nv.ptr(st), s))
        add("bag forward, LDS-staged + BatchNorm partials epilogue", us, k, Z * 8 + (Ee + 1) * 4 + 1800 * H * 4 + Ee * H * 4)
    dt = torch.empty(1800, H, device=dev)
    scr = torch.empty(int(nv.lib().esc_bag_bwd_scratch(Z, H)), device=dev)
    us, k = timed("bag_bwd", lambda: nv.call("esc_bag_bwd_table_rows", nv.ptr(zb), H, H, nv.ptr(plan.col_ptr), nv.ptr(plan.col_row), nv.ptr(plan.col_val), nv.ptr(plan.col_col), Z, 1800, Ee, 0, nv.ptr(dt), nv.ptr(scr), s))
    add("bag table gradient, XCD-aware (HBM bytes)", us, k, Ee * H * 4 + Z * 8 + 1800 * H * 4)
    add("bag table gradient (L2 row traffic)", us, k, Z * H * 4)
    us, k = timed("bag_bwd", lambda: nv.call("esc_bag_bwd_table", nv.ptr(zb), H, H, nv.ptr(plan.col_ptr), nv.ptr(plan.col_row), nv.ptr(plan.col_val), nv.ptr(plan.col_col), Z, 1800, nv.ptr(dt), nv.ptr(scr), s))
    add("bag table gradient, chunks in linear order", us, k, Z * H * 4)
    ids = torch.arange(128)
    us, k = timed("collate", lambda: store.collate(ids), n=30)
    add("device collate (2 kernels)", us, k, 27e6)
    # BatchNorm passes, edge-sized
    M = Ee
    xx = torch.randn(M, H, device=dev); yy = torch.randn(M, H, device=dev); dy = torch.randn(M, H, device=dev); dxx = torch.empty(M, H, device=dev)
    mean = torch.zeros(H, device=dev); inv = torch.ones(H, device=dev); ga = torch.ones(H, device=dev); be = torch.zeros(H, device=dev)
    dg = torch.empty(H, device=dev); db = torch.empty(H, device=dev); sc = torch.ones(H, device=dev); sh = torch.zeros(H, device=dev)
    scratch = torch.empty(nv.lib().esc_bn_scratch(H), device=dev)
    us, k = timed("norm", lambda: nv.call("esc_bn_bwd", nv.ptr(xx), H, None, 0, nv.ptr(dy), H, M, H, nv.ptr(mean), nv.ptr(inv), nv.ptr(ga), nv.ptr(be), 1, nv.ptr(dxx), H, nv.ptr(dg), nv.ptr(db), nv.ptr(scratch), s))
    add("BatchNorm backward, edge rows (3 kernels), cache-resident", us, k, 5 * M * H * 4)
    us, k = timed("norm", lambda: nv.call("esc_affine_act", nv.ptr(xx), H, M, H, nv.ptr(sc), nv.ptr(sh), 1, nv.ptr(yy), H, s))
    add("affine + ReLU, edge rows, cache-resident", us, k, 2 * M * H * 4)
    # GEMMs
    w = torch.randn(H, H, device=dev); bias = torch.randn(H, device=dev)
    for name, rows_ in (("edge rows 15200x256x256", Ee), ("node rows 2400x256x256", N)):
        a = torch.randn(rows_, H, device=dev); c = torch.empty(rows_, H, device=dev)
        us, k = timed("linear", lambda: nv.call("esc_linear_fwd", nv.ptr(a), H, nv.ptr(w), H, nv.ptr(bias), None, None, rows_, H, H, nv.ptr(c), H, None, s))
        add("Linear forward, " + name, us, k, flops=2.0 * rows_ * H * H)
        dw = torch.empty(H, H, device=dev); dbb = torch.empty(H, device=dev); da = torch.empty(rows_, H, device=dev)
        slabs = torch.empty(int(nv.lib().esc_linear_bwd_weight_scratch(rows_, H, H)), device=dev)
        us, k = timed("linear", lambda: nv.call("esc_linear_bwd_both", nv.ptr(c), H, nv.ptr(a), H, None, None, nv.ptr(w), H, rows_, H, H, nv.ptr(da), H, 0, nv.ptr(dw), H, nv.ptr(dbb), nv.ptr(slabs), s))
        add("Linear backward dX+dW (+reduce), " + name, us, k, flops=4.0 * rows_ * H * H)
    # config-5 shapes: emb 300 / 600 (reduction lengths that are not multiples of the 32-wide K-step)
    for name, rows_, n_out, k_in in (("edge rows 20000x300x300", 20000, 300, 300), ("node rows 6500x600x300", 6500, 600, 300),
                                     ("node rows 6500x300x600", 6500, 300, 600)):
        a = torch.randn(rows_, k_in, device=dev); c = torch.empty(rows_, n_out, device=dev)
        w2 = torch.randn(n_out, k_in, device=dev); b2 = torch.randn(n_out, device=dev)
        us, k = timed("linear", lambda: nv.call("esc_linear_fwd", nv.ptr(a), k_in, nv.ptr(w2), k_in, nv.ptr(b2), None, None, rows_, n_out, k_in, nv.ptr(c), n_out, None, s))
        add("Linear forward, " + name, us, k, flops=2.0 * rows_ * n_out * k_in)
        dw = torch.empty(n_out, k_in, device=dev); dbb = torch.empty(n_out, device=dev); da = torch.empty(rows_, k_in, device=dev)
        slabs = torch.empty(int(nv.lib().esc_linear_bwd_weight_scratch(rows_, n_out, k_in)), device=dev)
        us, k = timed("linear", lambda: nv.call("esc_linear_bwd_both", nv.ptr(c), n_out, nv.ptr(a), k_in, None, None, nv.ptr(w2), k_in, rows_, n_out, k_in, nv.ptr(da), k_in, 0, nv.ptr(dw), k_in, nv.ptr(dbb), nv.ptr(slabs), s))
        add("Linear backward dX+dW (+reduce), " + name, us, k, flops=4.0 * rows_ * n_out * k_in)
    # the vendor library on the same shapes (torch.addmm -> rocBLAS / hipBLASLt, fp32, no TF32 on gfx950), same clock for both
    torch.backends.cuda.matmul.allow_tf32 = False
    for name, rows_, n_out, k_in in (("edge rows 15200x256x256", Ee, H, H), ("node rows 2400x256x256", N, H, H),
                                     ("edge rows 20000x300x300", 20000, 300, 300), ("node rows 6500x600x300", 6500, 600, 300)):
        a = torch.randn(rows_, k_in, device=dev); c = torch.empty(rows_, n_out, device=dev)
        w2 = torch.randn(n_out, k_in, device=dev); b2 = torch.randn(n_out, device=dev)
        us_lib = timed_wall(lambda: torch.addmm(b2, a, w2.t(), out=c))
        us_own = timed_wall(lambda: nv.call("esc_linear_fwd", nv.ptr(a), k_in, nv.ptr(w2), k_in, nv.ptr(b2), None, None, rows_, n_out, k_in, nv.ptr(c), n_out, None, s))
        add("Linear forward wall/call, " + name + ": esc_linear_fwd", us_own, 1, flops=2.0 * rows_ * n_out * k_in)
        add("Linear forward wall/call, " + name + ": torch.addmm (vendor library)", us_lib, 1, flops=2.0 * rows_ * n_out * k_in)
    print("MI355X, BASELINE config 1 shapes: N=%d E=%d Z=%d H=%d; peaks: HBM %.0f GB/s, fp32 MFMA %.1f TFLOP/s" % (N, Ee, Z, H, HBM, MFMA))
    head = ("kernel (back to back)", "us/call", "launches", "algorithmic", "achieved", "of peak")
    wid = [max(len(r[i]) for r in rows + [head]) for i in range(6)]
    for r in [head] + rows:
        print("  ".join(r[i].ljust(wid[i]) for i in range(6)))


if __name__ == "__main__":
    main()
