import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class FeatureCases(object):
    """Reader of the ragged feature fixtures written by oracle/make_golden.py."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name))
        self.names = [str(s) for s in self.z["names"]]

    def __len__(self):
        return len(self.names)

    def col(self, key, i):
        p = self.z[key + "_ptr"]
        return self.z[key][p[i]:p[i + 1]].astype(np.int64)

    def case(self, i):
        n, h, rd, sl = map(int, self.z["meta"][i])
        d = dict(name=self.names[i], n=n, h=h, use_rd=bool(rd), self_loop=bool(sl))
        for k in ("in_src", "in_dst", "out_src", "out_dst", "pos_enc", "pos_index", "pos_batch"):
            d[k] = self.col(k, i)
        return d


FEATURE_FILES = ("features_hand.npz", "features_count.npz", "features_mol.npz",
                 "features_directed.npz", "features_shipped.npz")


@pytest.fixture(scope="session")
def feature_cases():
    out = []
    for f in FEATURE_FILES:
        fc = FeatureCases(f)
        out.extend(fc.case(i) for i in range(len(fc)))
    return out


def load_collate(tag):
    z = np.load(os.path.join(GOLDEN, "collate_%s.npz" % tag))
    graphs, j = [], 0
    while ("g%d_x" % j) in z.files:
        graphs.append({k[len("g%d_" % j):]: z[k] for k in z.files if k.startswith("g%d_" % j)})
        j += 1
    batch = {k[len("batch_"):]: z[k] for k in z.files if k.startswith("batch_")}
    return graphs, batch, int(z["num_graphs"])


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no HIP device is visible")
