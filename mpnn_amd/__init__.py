"""mpnn_amd: MI355X-native (gfx950 HIP) ProteinMPNN message-passing hot path.

Drop-in for the reference's `mpnn.models.ProteinMPNN` (FridrichMethod/MPNN,
src/mpnn/models/model_utils.py): same constructor, forward signature and
state_dict; compute runs in hand-written HIP kernels behind the C ABI declared
in include/mpnn_hip.h.
"""
from .functional import cat_neighbors_nodes, gather_edges, gather_nodes, gather_nodes_t
from .modules import (DecLayer, EncLayer, PositionalEncodings, PositionWiseFeedForward, ProteinFeatures,
                      ProteinMPNN)

__all__ = ["ProteinMPNN", "ProteinFeatures", "EncLayer", "DecLayer", "PositionWiseFeedForward",
           "PositionalEncodings", "gather_edges", "gather_nodes", "gather_nodes_t", "cat_neighbors_nodes"]
