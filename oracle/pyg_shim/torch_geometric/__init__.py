"""Minimal stand-in for the torch_geometric surface that the reference's
utils_edge_efficient.py / batch.py touch at import or call time.

TEST INFRASTRUCTURE ONLY.  This is our own code (not PyG, not the reference);
it exists so that oracle/make_golden.py can import the *unmodified* reference
modules from /root/reference in the build container and record their outputs
as golden vectors.  Nothing in the product imports it.
"""
from . import data, utils  # noqa: F401


def is_debug_enabled():
    return False
