"""MI355X-native DISGAT hot path (edge-disentangled multi-head attention + SSL losses).

Drop-in for the `--model=DISGAT` path of TianxiangZhao/EdgeDisentangle_SSL; see DESIGN.md.
"""
from . import _lib  # noqa: F401
from .graph import CSRGraph, graph_of  # noqa: F401
from .layers import DisGALayer, FuseLayer, GraphConvolution, SageConv, disga_heads  # noqa: F401
from .models import DISGAT, MLP  # noqa: F401
