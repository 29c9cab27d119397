"""esc_gnn_amd — MI355X-native hot path of ESC-GNN's NestedGIN_eff.

Public surface mirrors the reference's modules for this path:
    create_subgraphs (utils_edge_efficient.py), Batch (batch.py), DataLoader (dataloader.py),
    NestedGIN_eff (run_graphcount.py), GINEConv.
"""
from .data import Data  # noqa: F401
from .batch import Batch  # noqa: F401
from .dataloader import DataLoader  # noqa: F401
from .plan import BatchPlan, plan_of  # noqa: F401
from . import _native, ops, optim, parallel  # noqa: F401
from .nn import GINEConv, Linear, global_add_pool, global_mean_pool  # noqa: F401
from .run_graphcount import NestedGIN_eff  # noqa: F401
from .engine import StepEngine  # noqa: F401
from .store import DeviceGraphStore, DeviceLoader  # noqa: F401
from .utils_edge_efficient import create_subgraphs, create_subgraphs_many  # noqa: F401

__all__ = ["create_subgraphs", "create_subgraphs_many", "Data", "Batch", "DataLoader", "BatchPlan", "plan_of", "GINEConv", "Linear",
           "NestedGIN_eff", "global_add_pool", "global_mean_pool", "ops"]


def install_dropin():
    """Register this package's modules under the reference's top-level module names, so that an unmodified
    `from utils_edge_efficient import create_subgraphs`, `from batch import Batch`,
    `from dataloader import DataLoader`, `from kernel.gin import NestedGIN_eff`, `from zinc_models import *`,
    `from ogb_mol_gnn import GNN` or `from modules.gine_operations import GINEPLUS` resolves here."""
    import sys
    import types
    from . import (batch as _batch, dataloader as _dataloader, kernel_gin as _kernel_gin, ogb_mol_gnn as _ogb,
                   utils_edge_efficient as _uee, zinc_models as _zinc)
    from .modules import gine_operations as _gine_ops
    sys.modules.setdefault("utils_edge_efficient", _uee)
    sys.modules.setdefault("zinc_models", _zinc)
    sys.modules.setdefault("ogb_mol_gnn", _ogb)
    mods = sys.modules.setdefault("modules", types.ModuleType("modules"))
    mods.gine_operations = _gine_ops
    sys.modules.setdefault("modules.gine_operations", _gine_ops)
    sys.modules.setdefault("batch", _batch)
    sys.modules.setdefault("dataloader", _dataloader)
    pkg = sys.modules.setdefault("kernel", types.ModuleType("kernel"))
    pkg.gin = _kernel_gin
    sys.modules.setdefault("kernel.gin", _kernel_gin)
