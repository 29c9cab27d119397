#!/usr/bin/env python3
"""bench.py — NestedGIN_eff training-step throughput on MI355X (the metric of BASELINE.json).

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch: device collate + forward + L1 loss + backward +
(gradient all-reduce when N>1) + Adam, on the synthetic count_cycle-shaped workload of SURVEY.md §8(d)
(random regular graphs, h=3, rd + self loops, bs=128 PER GPU, L=4, H=256).  The pre-processed
dataset (features from the HIP feature builder) is resident in HBM before the timed region.
Prints ONE JSON line on rank 0.

`roofline` = the scatter-add (esc::agg_fwd_wave): algorithmic bytes per launch / the launch's execution window, measured live in
the timed region on the device wall clock (first workgroup in -> last wave out, esc_prof_span_*); the hipExtLaunchKernelGGL event
pairs of all launches (= inter-kernel dispatch gap + kernel) and the committed rocprofv3 trace's average stand beside it
(`event_pairs`, `rocprofv3_avg_us`; DESIGN.md section 5 has the evidence for that choice of clock).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s measured-achievable
MFMA_F32_PEAK_TF = 157.3    # v_mfma_f32_32x32x2_f32 dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch_size", type=int, default=128)
    ap.add_argument("--graphs", type=int, default=1500, help="graphs in the (train) split kept in HBM per rank")
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--h", type=int, default=3)
    ap.add_argument("--lr", type=float, default=1e-2)
    ap.add_argument("--cpu_seconds", type=float, default=20.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no_breakdown", action="store_true")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): --batch_size graphs per GPU and step; strong: ONE global batch of --batch_size graphs "
                         "per step, sliced by graph over the ranks (SURVEY 8e)")
    ap.add_argument("--tune", action="append", default=[], metavar="KNOB=VALUE", help="esc_tune_set(knob, value), repeatable")
    ap.add_argument("--node_priority", type=int, default=0,
                    help="-1: run the step (the node pipeline) on a HIGH-priority HIP stream instead of torch's default stream")
    ap.add_argument("--streams", type=int, default=None,
                    help="esc_engine_set_side_stream mode (default: the library's; 0 = everything on one stream)")
    ap.add_argument("--path", choices=("engine", "autograd"), default="engine",
                    help="engine: one esc_engine_train_step call per step; autograd: per-op torch.autograd path")
    return ap.parse_args()


def linear_flops_per_step(N, E, H, L, in_dim=10):
    """2*M*K*N per GEMM forward, x3 for forward + input grad + weight grad."""
    f = 0
    f += 2 * E * H * H                                   # z_embedding.3
    f += 2 * E * H * in_dim + 2 * E * H * H * (L - 1)    # conv*.lin (edge term)
    f += 2 * N * in_dim * H + 2 * N * H * H              # conv1.nn
    f += (L - 1) * 2 * 2 * N * H * H                     # convs.nn
    f += 2 * N * in_dim * H + 2 * N * H * H              # x_embedding
    f += 2 * N * (L + 1) * H * H + 2 * N * H             # lin1, lin2
    return 3 * f


def aggregate_bytes(N, E, C):
    """algorithmic bytes of one aggregate-forward launch."""
    return 2 * E * C * 4 + 2 * N * C * 4 + E * 8 + (N + 1) * 4


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the hot path has no CPU fallback")
    local = local % torch.cuda.device_count()             # rehearsals may put several ranks on one card
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.node_priority < 0:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    if world > 1:                                         # RCCL over xGMI; ESC_DIST_BACKEND=gloo only for rehearsing on one GPU
        backend = os.environ.get("ESC_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    import esc_gnn_amd as E
    from esc_gnn_amd import _native as nv
    from esc_gnn_amd.datasets import build_count_dataset
    from esc_gnn_amd.store import DeviceGraphStore

    if args.streams is not None:
        nv.call("esc_engine_set_side_stream", args.streams)
    for kv in args.tune:
        k, v = kv.split("=")
        nv.call("esc_tune_set", int(k), int(v))
    torch.manual_seed(0)
    # ---- dataset: synthetic graphs -> HIP feature build -> HBM-resident store -----------------------
    t0 = time.time()
    # weak scaling: every rank owns its own split; strong scaling: every rank holds the SAME split and takes its slice of
    # each global batch
    first = 0 if args.scaling == "strong" else rank * args.graphs
    graphs = build_count_dataset(first, args.graphs, h=args.h, use_rd=True, self_loop=True)
    torch.cuda.synchronize()
    t_feat = time.time() - t0
    # y normalisation by mean/std of the split (run_graphcount.py:441-447)
    y_all = torch.cat([g.y.view(-1) for g in graphs])
    mean, std = y_all.mean(), y_all.std()
    for g in graphs:
        g.y = (g.y.view(-1) - mean) / std
    store = DeviceGraphStore(graphs, dev)
    nb = args.graphs // args.batch_size                     # full batches only, shuffle=False
    batch_ids = [torch.arange(i * args.batch_size, (i + 1) * args.batch_size) for i in range(nb)]   # host ids, like a sampler

    model = E.NestedGIN_eff(None, args.layers, args.hidden, use_rd=True, graph_pred=False, dropout=0,
                            edge_nest=True, use_cycle=True).to(dev)
    E.parallel.broadcast_parameters(model, 0)               # identical replicas
    # two gradient buckets (world > 1): the node pipeline's bucket is all-reduced while the edge pipeline's backward tail
    # is still running, the edge pipeline's bucket (+ the node-count slot) after the join
    opt = E.optim.FlatAdam(model.parameters(), lr=args.lr, late=E.parallel.edge_pipeline_parameters(model))
    model.train()
    engine = E.StepEngine(model) if args.path == "engine" else None
    if engine is None:
        model.engine_forward = False                       # --path autograd measures the per-op path

    stats = dict(graphs=0, nodes=0, edges=0, nnz=0)

    def tally(b):
        stats["graphs"] += b.num_graphs
        stats["nodes"] += b.x.size(0)
        stats["edges"] += b.edge_index.size(1)
        stats["nnz"] += b.pos_enc.numel()

    nxt = {"b": None}
    mode = {"scaling": args.scaling, "store": store}      # what step() runs: the headline mode, then (N > 1) the other one

    def next_ids(i):
        """weak scaling: every rank walks its own split in batches of --batch_size graphs (per-GPU work fixed).
        strong scaling: ONE global batch of --batch_size graphs per step, rank r collates its contiguous
        slice [r*B/W, (r+1)*B/W) of it (run_graphcount.py's data-parallel mode)."""
        ids = batch_ids[i % nb]
        if mode["scaling"] == "strong" and world > 1:         # contiguous slices of near-equal EDGE count (the step's cost is edge-sized work)
            ep = mode["store"].h_edge_ptr
            lo, hi = E.parallel.shard_slice_balanced((ep[ids + 1] - ep[ids]).tolist(), rank, world)
            return ids[lo:hi]
        return ids

    def step(i, count=False):
        # host ids: async pinned staging, no host/device sync.  Engine path: the NEXT batch is collated between the two
        # halves of the step, i.e. on the node stream while the edge pipeline finishes its backward
        store = mode["store"]
        b = nxt["b"] if nxt["b"] is not None else store.collate(next_ids(i))
        nxt["b"] = None
        if engine is not None:
            # world > 1: gradients of sum|err| (not the local mean): ONE RCCL all-reduce of grad ++ [n_local] gives the
            # global sums, and the division by the global node count rides on the Adam launch
            loss = engine.begin_step(b, loss_denom=1 if world > 1 else None)
            nxt["b"] = store.collate(next_ids(i + 1))
            if world > 1:
                opt.all_reduce_early()                      # node-pipeline bucket: overlaps the edge tail
            engine.end_step()
            if world > 1:
                opt.step(grad_denom=opt.all_reduce_late(b.x.size(0)))
            else:
                opt.step()
            if count:
                tally(b)
            return loss
        opt.zero_grad()
        pred = model(b)
        loss = E.ops.l1_loss(pred, b.y)
        loss.backward()
        if world > 1:                                       # ONE all-reduce: grad*n_local ++ [n_local]
            opt.all_reduce_weighted(b.x.size(0))
        opt.step()
        if count:
            tally(b)
        return loss

    for i in range(args.warmup):
        step(i)
    # dominant-kernel timing with HIP events on the launch stream, inside the timed region
    nv.prof_reset("agg_fwd")
    # + the kernel's own first-wave-in -> last-wave-out window, stamped by the launches of the first 10 timed steps
    try:
        nv.prof_span_arm("agg_fwd", min(args.steps, 10) * max(args.layers - 1, 1))
    except Exception as exc:                                    # the roofline then falls back to the event pairs (clock says so)
        print("bench: in-kernel stamps unavailable (%s)" % exc, file=sys.stderr)
    nv.prof_enable("agg_fwd", True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, count=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    nv.prof_enable("agg_fwd", False)
    n_agg, ms_agg = nv.prof_read("agg_fwd")
    agg_launch_ms = nv.prof_read_all("agg_fwd")
    try:
        agg_span_us = [u for u in nv.prof_span_read("agg_fwd") if 0 < u < 1e4]      # (a wrapped / unset stamp is dropped)
    except Exception:
        agg_span_us = []
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
        tot = torch.tensor([stats["graphs"], stats["edges"]], device=dev, dtype=torch.float64)
        dist.all_reduce(tot)
        total_graphs, total_edges = float(tot[0]), float(tot[1])
    else:
        total_graphs, total_edges = float(stats["graphs"]), float(stats["edges"])

    # ---- N > 1: the OTHER scaling mode, same K / W, barrier and max-over-ranks timing (SURVEY 8e slices ONE global batch of
    # --batch_size graphs over the ranks = "strong"; the headline default keeps --batch_size graphs per GPU = "weak") ------
    other = None
    if world > 1:
        other_mode = "strong" if args.scaling == "weak" else "weak"
        if other_mode == "strong" and rank != 0:           # strong: every rank slices the SAME split (rank 0's)
            g0 = build_count_dataset(0, args.graphs, h=args.h, use_rd=True, self_loop=True)
            y0 = torch.cat([g.y.view(-1) for g in g0])          # the shared split's own normalisation: identical on every rank
            for g in g0:
                g.y = (g.y.view(-1) - y0.mean()) / y0.std()
            mode["store"] = DeviceGraphStore(g0, dev)
        elif other_mode == "weak" and rank != 0:           # weak: every rank walks its own split
            gr = build_count_dataset(rank * args.graphs, args.graphs, h=args.h, use_rd=True, self_loop=True)
            yr = torch.cat([g.y.view(-1) for g in gr])
            for g in gr:
                g.y = (g.y.view(-1) - yr.mean()) / yr.std()
            mode["store"] = DeviceGraphStore(gr, dev)
        mode["scaling"] = other_mode
        nxt["b"] = None
        keep = dict(stats)
        for k in stats:
            stats[k] = 0
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i, count=True)
        torch.cuda.synchronize()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tot = torch.tensor([stats["graphs"], stats["edges"]], device=dev, dtype=torch.float64)
        dist.all_reduce(tot)
        other = dict(scaling=other_mode, value=round(float(tot[0]) / float(t), 1), unit="graphs/s",
                     ms_per_step=round(float(t) / args.steps * 1e3, 3),
                     edges_aggregated_per_s=round(float(tot[1]) * args.layers / float(t), 1),
                     global_batch=args.batch_size * (world if other_mode == "weak" else 1),
                     graphs_per_rank_and_step=round(float(tot[0]) / args.steps / world, 1))
        stats.update(keep)
        mode["scaling"], mode["store"] = args.scaling, store
        nxt["b"] = None

    # ---- per-family breakdown (separate instrumented steps, not part of `value`) -------------------
    breakdown = {}
    if not args.no_breakdown:
        fams = ["agg_fwd", "agg_bwd", "bag_fwd", "bag_bwd", "linear", "norm", "collate"]
        for f in fams:
            nv.prof_reset(f)
            nv.prof_enable(f, True)
        k_extra = 5
        for i in range(k_extra):
            step(i)
        torch.cuda.synchronize()
        for f in fams:
            nv.prof_enable(f, False)
            n, ms = nv.prof_read(f)
            breakdown[f] = dict(launches_per_step=n / k_extra, ms_per_step=ms / k_extra)

    # the same Linear kernels with the step on ONE stream: each launch has the CUs to itself, so this is what the GEMM
    # kernels do alone; the figure above is what they do while the other pipeline's kernels share the machine
    linear_one_stream_ms = None
    if not args.no_breakdown and args.path == "engine" and world == 1:
        nv.call("esc_engine_set_side_stream", 0)
        for i in range(2):
            step(i)
        nv.prof_reset("linear")
        nv.prof_enable("linear", True)
        for i in range(5):
            step(i)
        torch.cuda.synchronize()
        nv.prof_enable("linear", False)
        linear_one_stream_ms = nv.prof_read("linear")[1] / 5
        nv.call("esc_engine_set_side_stream", 2 if args.streams is None else args.streams)

    # the dominant MFMA kernel on its own: the 128x128x32 tile family that runs the edge-row Linear layers (z_embedding.3 and
    # conv*.lin, forward and the fused dX+dW launch) — event pairs per launch inside the normal two-stream step
    edge_gemm = None
    if not args.no_breakdown and args.path == "engine" and world == 1:
        nv.prof_reset("gemm_edge")
        nv.prof_enable("gemm_edge", True)
        for i in range(5):
            step(i)
        torch.cuda.synchronize()
        nv.prof_enable("gemm_edge", False)
        per = sorted(nv.prof_read_all("gemm_edge"))
        if per:
            edge_gemm = dict(launches_per_step=len(per) / 5, ms_per_step=sum(per) / 5, median_ms=per[len(per) // 2])

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    N_avg = stats["nodes"] / args.steps
    E_avg = stats["edges"] / args.steps
    ms_step = elapsed / args.steps * 1e3
    value = total_graphs / elapsed
    edges_agg_per_s = total_edges * args.layers / elapsed

    # roofline of the scatter-add: the row-per-wave aggregate-forward kernel, launched by the (L-1) layers of
    # width `hidden` (the 10-wide first layer runs the narrow element kernel and is not in this average)
    alg_bytes = aggregate_bytes(N_avg, E_avg, args.hidden)
    avg_ms = ms_agg / max(n_agg, 1)
    pair_achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if n_agg else 0.0
    # Two live clocks, both inside the timed region.  (a) in-kernel: every workgroup stores the device wall clock at its first
    # instruction, every wave at its last one (after its stores have been acknowledged); max - min = the launch's execution window.
    # (b) event pairs (hipExtLaunchKernelGGL start / stop): the start marker completes when the kernel in FRONT has drained, so a pair =
    # inter-kernel dispatch gap + kernel, and the gap moves with the queue's state (1.5 us with per-layer edge terms, 2.5 us
    # with batched ones, the same kernel).  (a) agrees with rocprofv3's dispatch begin -> end of the committed trace within a few percent
    # and is what `achieved` / `frac` use; (b) is kept beside it.
    span_sorted = sorted(agg_span_us)
    span_avg = sum(agg_span_us) / len(agg_span_us) if agg_span_us else 0.0
    dur_us = span_avg if agg_span_us else avg_ms * 1e3
    achieved = alg_bytes / (dur_us * 1e-6) / 1e9 if dur_us > 0 else 0.0
    per_step = max(args.layers - 1, 1)                      # launch i of a step belongs to layer 1 + i % (L-1)
    by_layer = [round(1e3 * sum(agg_launch_ms[k::per_step]) / max(len(agg_launch_ms[k::per_step]), 1), 2) for k in range(per_step)] if agg_launch_ms else []
    per_launch = sorted(agg_launch_ms)
    med_ms = per_launch[len(per_launch) // 2] if per_launch else 0.0
    min_ms = per_launch[0] if per_launch else 0.0
    # HBM bytes per launch: cannot be measured from inside this process (rocprofv3 --pmc, two separate passes).  When
    # tools/profile_round.sh has run those passes on THIS command, its result is committed under profiles/ and quoted
    # here with its provenance; `frac` never uses it (algorithmic bytes / live launch time, as BASELINE prescribes),
    # `hbm_frac` = measured traffic / live launch time / peak is what the HBM pins actually carried.
    traffic, tsrc = None, None
    import glob
    tpaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_agg_fwd.json")))
    if tpaths:
        tj = json.load(open(tpaths[-1]))
        traffic = tj.get("traffic_bytes_per_launch")
        tsrc = ("committed PMC passes of this command (profiles/%s: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate "
                "runs, FETCH x2 gfx950 correction, tools/parse_pmc.py) - NOT measured in this run" % os.path.basename(tpaths[-1]))
    # the same kernel's average in the committed rocprofv3 kernel trace of this command (a TRACED, host-bound step: quoted for
    # reconciliation with `avg_us`, never used in `frac`)
    rp_avg, rp_src = None, None
    spaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats.csv")))
    if spaths:
        try:
            import csv
            for row in csv.DictReader(open(spaths[-1])):
                if "agg_fwd_wave<2" in row.get("Name", "") or ("agg_fwd_wave<4, true" in row.get("Name", "") and rp_avg is None):
                    rp_avg = round(float(row["AverageNs"]) * 1e-3, 2)
                    rp_src = "profiles/%s (rocprofv3 --kernel-trace --stats of this command; not measured in this run)" % os.path.basename(spaths[-1])
                    if "agg_fwd_wave<2" in row["Name"]:
                        break
        except Exception:
            rp_avg, rp_src = None, None
    split = "2, true> (two waves per destination row" if (args.hidden == 256 and os.environ.get("ESC_AGG_SPLIT", "2") == "2") else "4, true> (one wave per destination row"
    fr = lambda us: round(alg_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if us and us > 0 else None
    pairs = dict(launches=n_agg, avg_us=round(avg_ms * 1e3, 2), median_us=round(med_ms * 1e3, 2), min_us=round(min_ms * 1e3, 2),
                 frac=fr(avg_ms * 1e3), frac_median=fr(med_ms * 1e3), frac_min_time=fr(min_ms * 1e3), by_layer_us=by_layer,
                 clock="hipExtLaunchKernelGGL start/stop event pair per launch on the launch stream, all launches of the timed region: "
                       "inter-kernel dispatch gap + kernel")
    roofline = dict(kernel="esc::agg_fwd_wave<%s; GINE aggregate forward = the scatter-add, C=%d; the gathered rows get the previous "
                           "layer's BatchNorm+ReLU applied as they are read)" % (split, args.hidden),
                    bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=None if traffic is None else int(traffic),
                    traffic_source=tsrc,
                    hbm_frac=None if (traffic is None or dur_us <= 0) else round(traffic / (dur_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                    launches=len(agg_span_us) if agg_span_us else n_agg, avg_us=round(dur_us, 2),
                    median_us=round(span_sorted[len(span_sorted) // 2], 2) if span_sorted else round(med_ms * 1e3, 2),
                    min_us=round(span_sorted[0], 2) if span_sorted else round(min_ms * 1e3, 2),
                    clock=("in-kernel execution window on the device wall clock (first workgroup in -> last wave out, stores acknowledged), "
                           "stamped by the launches of the first 10 timed steps; the event pairs of ALL launches are in `event_pairs`, the "
                           "committed rocprofv3 trace's average in `rocprofv3_avg_us`") if agg_span_us else pairs["clock"],
                    event_pairs=pairs, rocprofv3_avg_us=rp_avg, rocprofv3_source=rp_src, frac_rocprofv3=fr(rp_avg),
                    alg_bytes_per_launch=int(alg_bytes))
    extra = {}
    if "linear" in breakdown and breakdown["linear"]["ms_per_step"] > 0:
        fl = linear_flops_per_step(N_avg, E_avg, args.hidden, args.layers)
        tf = fl / (breakdown["linear"]["ms_per_step"] * 1e-3) / 1e12
        allin = dict(kernels="every Linear launch of the step (fwd + dX + dW): esc::dma::gemm_kernel / gemm_dual_kernel in all tile shapes + "
                             "the small-dimension kernels", achieved=round(tf, 2), frac=round(tf / MFMA_F32_PEAK_TF, 4), flops_per_step=int(fl),
                     note="durations summed inside the two-stream step (kernels of the other pipeline share the CUs)")
        if linear_one_stream_ms:
            tf1 = fl / (linear_one_stream_ms * 1e-3) / 1e12
            allin.update(achieved_one_stream=round(tf1, 2), frac_one_stream=round(tf1 / MFMA_F32_PEAK_TF, 4),
                         linear_ms_per_step_one_stream=round(linear_one_stream_ms, 4))
        if edge_gemm:
            # z_embedding.3 + (L-1) H-wide conv*.lin: forward 2*E*H*H each, the fused dX+dW launch 4*E*H*H each
            n_wide = args.layers                        # 1 + (L - 1)
            fl_e = 6.0 * E_avg * args.hidden * args.hidden * n_wide
            tfe = fl_e / (edge_gemm["ms_per_step"] * 1e-3) / 1e12
            extra["roofline_mfma"] = dict(
                kernel="esc::dma::gemm_kernel / gemm_dual_kernel <128,128,32, 4 compute + 4 loader waves>: the edge-row Linear launches "
                       "(z_embedding.3, conv*.lin; forward and fused dX+dW) = %.0f %% of the step's Linear flops" % (100 * fl_e / fl),
                bound="mfma", achieved=round(tfe, 2), peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=round(tfe / MFMA_F32_PEAK_TF, 4),
                flops_per_step=int(fl_e), launches_per_step=edge_gemm["launches_per_step"],
                avg_us=round(edge_gemm["ms_per_step"] / edge_gemm["launches_per_step"] * 1e3, 2),
                clock="event pair per launch inside the two-stream step, mean over 5 steps", all_linear=allin)
        else:
            extra["roofline_mfma"] = dict(kernel=allin["kernels"], bound="mfma", achieved=allin["achieved"], peak=MFMA_F32_PEAK_TF,
                                          unit="TFLOP/s", frac=allin["frac"], flops_per_step=allin["flops_per_step"], all_linear=allin)
    cpu = cpu_baseline(args, graphs) if (args.cpu_seconds > 0 and world == 1) else None   # rank 0, N=1 only

    out = {
        "metric": "graphs/sec + edges-aggregated/sec, NestedGIN_eff h=3 bs=128 @1/2/4/8 GPU",
        "value": round(value, 1), "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "count_cycle-shaped random regular graphs (n in 10/15/20/30), target=triangles, "
                               "NestedGIN_eff h=%d layers=%d hidden=%d, bs=%d per GPU (configs[1])"
                               % (args.h, args.layers, args.hidden, args.batch_size),
                   "global_batch": args.batch_size * (world if args.scaling == "weak" else 1),
                   "parallelism": "dp%d graph-sharded" % world, "step_path": args.path,
                   "rccl_ranks": (dist.get_world_size() if (world > 1 and dist.get_backend() == "nccl") else (1 if world == 1 else 0)),
                   "ranks": world,
                   "collective_backend": (dist.get_backend() if world > 1 else None),
                   "grad_allreduce": ("two buckets: node-pipeline gradients during the edge backward tail, edge-pipeline gradients "
                                      "+ node count after the join" if world > 1 else None),
                   "nodes_per_batch": round(N_avg, 1), "edges_per_batch": round(E_avg, 1),
                   "nnz_per_batch": round(stats["nnz"] / args.steps, 1)},
        "edges_aggregated_per_s": round(edges_agg_per_s, 1),
        "roofline": roofline,
        "cpu_baseline": cpu,
        "feature_build": {"graphs": args.graphs, "seconds": round(t_feat, 3),
                          "note": "graph generation (networkx) + HIP create_subgraphs_many + host copies"},
        "kernel_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in breakdown.items()},
    }
    if world > 1:                     # both scaling modes in one line: `value` is the --scaling one, the other rides along
        this = dict(scaling=args.scaling, value=out["value"], unit="graphs/s", ms_per_step=out["ms_per_step"],
                    edges_aggregated_per_s=out["edges_aggregated_per_s"], global_batch=out["config"]["global_batch"],
                    graphs_per_rank_and_step=round(total_graphs / args.steps / world, 1))
        out[args.scaling] = this
        out[other["scaling"]] = other
    out.update(extra)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(args, graphs):
    """Reference path on the host cores: the oracle's pure-PyTorch NestedGIN_eff (what PyG 2.0.4 dispatches to
    on CPU) + the host-side python collate, same batches, same hyper-parameters.  Bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_model as rm            # oracle: used here ONLY as the timed CPU baseline
    import esc_gnn_amd as E
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, int(os.environ.get("ESC_CPU_THREADS", "16"))))   # the box's CPU share per GPU is 16
    torch.manual_seed(0)
    model = rm.NestedGINEffRef(args.layers, args.hidden)
    opt = torch.optim.Adam(model.parameters(), lr=args.lr)
    model.train()
    bs = args.batch_size
    nb = len(graphs) // bs

    def one(i):
        t0 = time.perf_counter()
        b = E.Batch.from_data_list(graphs[(i % nb) * bs:(i % nb + 1) * bs])
        bd = dict(x=b.x, edge_index=b.edge_index, pos_enc=b.pos_enc, pos_index=b.pos_index,
                  pos_batch=b.pos_batch, batch=b.batch, y=b.y)
        rm.train_step(model, opt, bd)
        return time.perf_counter() - t0

    # the baseline gets the thread count it runs fastest with (torch's CPU kernels do not always scale to every core of
    # the box's share): two steps at each candidate, the best one runs the sample
    tried = {}
    for th in sorted({cores, max(1, cores // 2), max(1, cores // 4), 1}, reverse=True):
        torch.set_num_threads(th)
        one(0)
        tried[th] = one(1)
    best = min(tried, key=tried.get)
    torch.set_num_threads(best)
    times = []
    t_start = time.time()
    i = 2
    while True:
        times.append(one(i))
        i += 1
        if (time.time() - t_start > args.cpu_seconds and len(times) >= 3) or len(times) >= 30:
            break
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(bs / med, 1), "unit": "graphs/s", "cores": best, "kind": "port",
            "sample": "%d training steps (python collate + fwd + bwd + Adam) of the same bs=%d batches, median; "
                      "torch %s, %d threads (fastest of %s tried; %d available)"
                      % (len(times), bs, torch.__version__, best, sorted(tried), cores)}


if __name__ == "__main__":
    main()
