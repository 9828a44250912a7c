"""state_dict layout of the ProteinMPNN hot path (reference: model_utils.py:366-425).

One table of (key, shape) pairs, used by the module classes, by the weight
packer that feeds the HIP kernels, and by tests that need a reference-shaped
state_dict without importing the reference.
"""
from __future__ import annotations

from collections import OrderedDict


def param_shapes(hidden_dim=128, num_encoder_layers=3, num_decoder_layers=3, num_letters=21,
                 vocab=21, use_virtual_center=False, occupancy_cutoff=None,
                 num_positional_embeddings=16, num_rbf=16, max_relative_feature=32):
    H = hidden_dim
    n_atoms = 6 if use_virtual_center else 5
    edge_in = num_positional_embeddings + num_rbf * n_atoms * n_atoms
    t = OrderedDict()
    t["features.embeddings.linear.weight"] = (num_positional_embeddings, 2 * max_relative_feature + 2)
    t["features.embeddings.linear.bias"] = (num_positional_embeddings,)
    t["features.edge_embedding.weight"] = (H, edge_in)
    t["features.norm_edges.weight"] = (H,)
    t["features.norm_edges.bias"] = (H,)
    t["W_e.weight"] = (H, H)
    t["W_e.bias"] = (H,)
    t["W_s.weight"] = (vocab, H)

    def lin(prefix, out_f, in_f):
        t[prefix + ".weight"] = (out_f, in_f)
        t[prefix + ".bias"] = (out_f,)

    def norm(prefix):
        t[prefix + ".weight"] = (H,)
        t[prefix + ".bias"] = (H,)

    for l in range(num_encoder_layers):
        p = f"encoder_layers.{l}"
        for n in ("norm1", "norm2", "norm3"):
            norm(f"{p}.{n}")
        lin(f"{p}.W1", H, 3 * H); lin(f"{p}.W2", H, H); lin(f"{p}.W3", H, H)
        lin(f"{p}.W11", H, 3 * H); lin(f"{p}.W12", H, H); lin(f"{p}.W13", H, H)
        lin(f"{p}.dense.W_in", 4 * H, H); lin(f"{p}.dense.W_out", H, 4 * H)
    for l in range(num_decoder_layers):
        p = f"decoder_layers.{l}"
        for n in ("norm1", "norm2"):
            norm(f"{p}.{n}")
        lin(f"{p}.W1", H, 4 * H); lin(f"{p}.W2", H, H); lin(f"{p}.W3", H, H)
        lin(f"{p}.dense.W_in", 4 * H, H); lin(f"{p}.dense.W_out", H, 4 * H)
    lin("W_out", num_letters, H)
    if occupancy_cutoff is not None:
        lin("occupancy_mlp.0", H, 1)
        lin("occupancy_mlp.2", H, H)
        norm("occupancy_mlp.3")
    return t
