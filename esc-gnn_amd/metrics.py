"""Evaluation metrics of the OGB molecule benchmarks, standing in for `ogb.graphproppred.Evaluator`
(un-vendored `ogb==1.3.3`, requirements.txt:54; used at /root/reference/run_ogb_mol.py:366,146-147).
rocauc / ap are averaged over tasks that have both classes among their labelled (non-NaN) entries — the
published definition of that evaluator; checked against scikit-learn in tests/test_cli.py.  Host-side numpy: this
is bookkeeping at epoch end, not the hot path."""
import numpy as np


def _rankdata_average(a):
    order = np.argsort(a, kind="mergesort")
    ranks = np.empty(len(a), dtype=np.float64)
    sa = a[order]
    bounds = np.flatnonzero(np.concatenate(([True], sa[1:] != sa[:-1], [True])))
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        ranks[order[lo:hi]] = 0.5 * (lo + hi - 1) + 1.0
    return ranks


def roc_auc(y_true, y_score):
    y_true = np.asarray(y_true) > 0.5
    pos, neg = int(y_true.sum()), int((~y_true).sum())
    if pos == 0 or neg == 0:
        raise ValueError("roc_auc needs both classes")
    r = _rankdata_average(np.asarray(y_score, dtype=np.float64))
    return float((r[y_true].sum() - pos * (pos + 1) / 2.0) / (pos * neg))


def average_precision(y_true, y_score):
    y_true = np.asarray(y_true) > 0.5
    s = np.asarray(y_score, dtype=np.float64)
    order = np.argsort(-s, kind="mergesort")
    s, t = s[order], y_true[order]
    last = np.flatnonzero(np.concatenate((s[1:] != s[:-1], [True])))      # one threshold per distinct score
    tp = np.cumsum(t)[last].astype(np.float64)
    precision = tp / (last + 1.0)
    recall = tp / max(1, int(t.sum()))
    return float(np.sum(np.diff(np.concatenate(([0.0], recall))) * precision))


class Evaluator(object):
    """`Evaluator(name).eval({'y_true': [G,T], 'y_pred': [G,T]}) -> {metric: value}`."""

    METRIC = {"ogbg-molhiv": "rocauc", "ogbg-molpcba": "ap"}

    def __init__(self, name):
        self.name = name
        self.eval_metric = self.METRIC.get(name, "rocauc")

    def eval(self, input_dict):
        y_true, y_pred = np.asarray(input_dict["y_true"]), np.asarray(input_dict["y_pred"])
        if y_true.shape != y_pred.shape or y_true.ndim != 2:
            raise RuntimeError("Evaluator: y_true / y_pred must both be [num_graphs, num_tasks]")
        fn = roc_auc if self.eval_metric == "rocauc" else average_precision
        vals = []
        for t in range(y_true.shape[1]):
            lab = y_true[:, t] == y_true[:, t]
            yt = y_true[lab, t]
            if self.eval_metric == "rocauc" and not ((yt == 1).any() and (yt == 0).any()):
                continue
            if self.eval_metric == "ap" and not ((yt == 1).any() and (yt == 0).any()):
                continue
            vals.append(fn(yt, y_pred[lab, t]))
        if not vals:
            raise RuntimeError("No positively labeled data available. Cannot compute " + self.eval_metric)
        return {self.eval_metric: float(np.mean(vals))}
