"""The C-ABI library loads on a CPU-only box and exports every symbol include/escgnn_hip.h declares
(no compute calls here)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "escgnn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(esc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import esc_gnn_amd as E
    names = _declared()
    assert len(names) >= 15
    h = ctypes.CDLL(E._native.LIB_PATH)
    missing = [n for n in names if not hasattr(h, n)]
    assert not missing, missing
    assert E._native.lib().esc_abi_version() == E._native.ABI_VERSION


def test_binding_covers_header():
    import esc_gnn_amd as E
    assert sorted(E._native.SIGNATURES) == _declared()


def test_cpu_tensor_is_refused_loudly():
    import pytest
    import torch
    import esc_gnn_amd as E
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        E.ops.linear(torch.zeros(2, 4), torch.zeros(3, 4), None)


def test_host_index_tensor_is_refused_not_dereferenced():
    """an index tensor left on the host must raise (its address would fault on the GPU); checked without a GPU by
    handing the op a meta-device weight is not possible, so check the guard helper directly"""
    import pytest
    import torch
    from esc_gnn_amd import ops
    with pytest.raises(RuntimeError, match="move the batch to the device"):
        ops._on(torch.device("cuda:0"), torch.zeros(3, dtype=torch.long))
