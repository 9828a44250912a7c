"""MI355X-native ProteinMPNN modules: the reference's nn.Module surface on gfx950 HIP kernels.

Mirrors `src/mpnn/models/model_utils.py` of FridrichMethod/MPNN: same class names, constructor
keyword arguments (:366-381), forward signatures (:427-437, :309, :73, :118) and state_dict
keys/shapes, so reference checkpoints load and the reference's train / fine-tune / EnergyMPNN
callers work unchanged.  The modules own only parameters; all compute is done by the C-ABI
library (include/mpnn_hip.h) through `mpnn_amd.engine`.  There is no PyTorch or CPU fallback:
inputs must live on a ROCm device.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import engine


def _on_device_of(t):
    """Kernels launch on the CURRENT device's stream: make the tensor's device current for the call (a model on cuda:1
    called while cuda:0 is current would otherwise launch on the wrong GPU).  CPU tensors fall through to the
    engine's own error."""
    import contextlib
    return torch.cuda.device(t.device) if isinstance(t, torch.Tensor) and t.is_cuda else contextlib.nullcontext()


class PositionWiseFeedForward(nn.Module):
    """Parameters of the position-wise FFN (reference :141-155); evaluated inside mpnn_node_update."""

    def __init__(self, num_hidden, num_ff):
        super().__init__()
        self.W_in = nn.Linear(num_hidden, num_ff, bias=True)
        self.W_out = nn.Linear(num_ff, num_hidden, bias=True)

    def forward(self, h_V):
        """W_out(gelu(W_in(h_V)))   (reference :151-155).  Inside EncLayer / DecLayer the block is fused into
        mpnn_node_update; called on its own it is that kernel's FFN-only form (128 -> 4 x 128 -> 128 only)."""
        with _on_device_of(h_V), torch.autocast(device_type="cuda", enabled=False):
            return engine.ffn_standalone(self, h_V)


class PositionalEncodings(nn.Module):
    """Relative-position / chain embedding (reference :158-177); evaluated as a 66-row table lookup."""

    def __init__(self, num_embeddings, max_relative_feature=32):
        super().__init__()
        self.num_embeddings = num_embeddings
        self.max_relative_feature = max_relative_feature
        self.linear = nn.Linear(2 * max_relative_feature + 1 + 1, num_embeddings)

    def forward(self, offset, mask):
        """offset, mask: integer tensors of one shape -> [..., num_embeddings]   (reference :168-177): the one-hot
        times Linear is a lookup in the 66-row table mpnn_pos_table builds.  Inside ProteinFeatures the lookup is
        fused into mpnn_edge_features."""
        with _on_device_of(offset), torch.autocast(device_type="cuda", enabled=False):
            return engine.positional_encodings_standalone(self, offset, mask)


class EncLayer(nn.Module):
    """Encoder layer (reference :48-95)."""

    def __init__(self, num_hidden, num_in, dropout=0.1, num_heads=None, scale=30):
        super().__init__()
        self.num_hidden = num_hidden
        self.num_in = num_in
        self.scale = scale
        self.dropout_p = dropout
        self.norm1 = nn.LayerNorm(num_hidden)
        self.norm2 = nn.LayerNorm(num_hidden)
        self.norm3 = nn.LayerNorm(num_hidden)
        self.W1 = nn.Linear(num_hidden + num_in, num_hidden, bias=True)
        self.W2 = nn.Linear(num_hidden, num_hidden, bias=True)
        self.W3 = nn.Linear(num_hidden, num_hidden, bias=True)
        self.W11 = nn.Linear(num_hidden + num_in, num_hidden, bias=True)
        self.W12 = nn.Linear(num_hidden, num_hidden, bias=True)
        self.W13 = nn.Linear(num_hidden, num_hidden, bias=True)
        self.dense = PositionWiseFeedForward(num_hidden, num_hidden * 4)

    def forward(self, h_V, h_E, E_idx, mask_V=None, mask_attend=None):
        """h_V [B,L,128], h_E [B,L,K,128], E_idx [B,L,K] -> (h_V, h_E), same layouts as the reference."""
        with _on_device_of(h_V), torch.autocast(device_type="cuda", enabled=False):
            return engine.enc_layer_standalone(self, h_V, h_E, E_idx, mask_V, mask_attend)


class DecLayer(nn.Module):
    """Decoder layer (reference :98-138).  Inside ProteinMPNN.forward the decoder input h_ESV is never
    materialised (mpnn_dec_message assembles it on the fly)."""

    def __init__(self, num_hidden, num_in, dropout=0.1, num_heads=None, scale=30):
        super().__init__()
        self.num_hidden = num_hidden
        self.num_in = num_in
        self.scale = scale
        self.dropout_p = dropout
        self.norm1 = nn.LayerNorm(num_hidden)
        self.norm2 = nn.LayerNorm(num_hidden)
        self.W1 = nn.Linear(num_hidden + num_in, num_hidden, bias=True)
        self.W2 = nn.Linear(num_hidden, num_hidden, bias=True)
        self.W3 = nn.Linear(num_hidden, num_hidden, bias=True)
        self.dense = PositionWiseFeedForward(num_hidden, num_hidden * 4)

    def forward(self, h_V, h_E, mask_V=None, mask_attend=None):
        """h_V [B,L,128], h_E = h_ESV [B,L,K,384] -> h_V, as the reference (:118-138)."""
        with _on_device_of(h_V), torch.autocast(device_type="cuda", enabled=False):
            return engine.dec_layer_standalone(self, h_V, h_E, mask_V, mask_attend)


class ProteinFeatures(nn.Module):
    """Backbone featurizer (reference :180-360)."""

    def __init__(self, edge_features, node_features, num_positional_embeddings=16, num_rbf=16, top_k=30,
                 augment_eps=0.0, num_chain_embeddings=16, edge_cutoff=None, use_virtual_center=False,
                 occupancy_cutoff=None):
        super().__init__()
        self.edge_features = edge_features
        self.node_features = node_features
        self.top_k = top_k
        self.augment_eps = augment_eps
        self.num_rbf = num_rbf
        self.num_positional_embeddings = num_positional_embeddings
        self.edge_cutoff = edge_cutoff
        self.use_virtual_center = use_virtual_center
        self.occupancy_cutoff = occupancy_cutoff
        self.embeddings = PositionalEncodings(num_positional_embeddings)
        num_atoms = 6 if use_virtual_center else 5
        edge_in = num_positional_embeddings + num_rbf * (num_atoms**2)
        self.edge_embedding = nn.Linear(edge_in, edge_features, bias=False)
        self.norm_edges = nn.LayerNorm(edge_features)

    def forward(self, X, mask, residue_idx, chain_labels, backbone_noise=None):
        """-> (E [B,L,K,128], E_idx [B,L,K] int64, occupancy [B,L,1] | None, edge_mask [B,L,K] | None)."""
        with _on_device_of(X), torch.autocast(device_type="cuda", enabled=False):
            return engine.features_standalone(self, X, mask, residue_idx, chain_labels, backbone_noise)


class ProteinMPNN(nn.Module):
    """Drop-in for `mpnn.models.ProteinMPNN` (reference :363-529)."""

    def __init__(self, num_letters=21, node_features=128, edge_features=128, hidden_dim=128,
                 num_encoder_layers=3, num_decoder_layers=3, vocab=21, k_neighbors=32, augment_eps=0.1,
                 dropout=0.1, edge_cutoff=None, use_virtual_center=False, occupancy_cutoff=None):
        super().__init__()
        if not (node_features == edge_features == hidden_dim == 128):
            raise ValueError("mpnn_amd kernels are specialised for node/edge/hidden size 128 "
                             "(the only size the reference's drivers use)")
        if not 1 <= k_neighbors <= 64:
            raise ValueError("mpnn_amd supports 1 <= k_neighbors <= 64")
        if num_letters > 32:
            raise ValueError("mpnn_amd supports num_letters <= 32")
        self.node_features = node_features
        self.edge_features = edge_features
        self.hidden_dim = hidden_dim
        self.num_letters = num_letters
        self.dropout_p = dropout
        # NB the reference passes (node_features, edge_features) into parameters named
        # (edge_features, node_features) (:390-392); harmless at 128/128 and kept.
        self.features = ProteinFeatures(node_features, edge_features, top_k=k_neighbors, augment_eps=augment_eps,
                                        edge_cutoff=edge_cutoff, use_virtual_center=use_virtual_center,
                                        occupancy_cutoff=occupancy_cutoff)
        self.W_e = nn.Linear(edge_features, hidden_dim, bias=True)
        self.W_s = nn.Embedding(vocab, hidden_dim)
        self.encoder_layers = nn.ModuleList(
            [EncLayer(hidden_dim, hidden_dim * 2, dropout=dropout) for _ in range(num_encoder_layers)])
        self.decoder_layers = nn.ModuleList(
            [DecLayer(hidden_dim, hidden_dim * 3, dropout=dropout) for _ in range(num_decoder_layers)])
        self.W_out = nn.Linear(hidden_dim, num_letters, bias=True)
        self.occupancy_mlp = None
        if occupancy_cutoff is not None:
            self.occupancy_mlp = nn.Sequential(nn.Linear(1, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, hidden_dim),
                                               nn.LayerNorm(hidden_dim))
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self._packed = None  # engine.PackedWeights cache
        self.precision = "fp32"  # forward arithmetic: "fp32" (default) or "f16" (see set_precision)

    def set_precision(self, mode):
        """'fp32': every product fp32-class (three f16 MFMA products, fp32 accumulate).  'f16': one f16 product per
        fp32 product in inference forward / sampling (bf16-class accuracy, stated in include/mpnn_hip.h); training
        always runs fp32-class."""
        if mode not in ("fp32", "f16"):
            raise ValueError("precision must be 'fp32' or 'f16'")
        self.precision = mode
        return self

    def forward(self, X, S, mask, chain_M, residue_idx, chain_encoding_all, fix_order=None,
                fix_backbone_noise=None):
        """Graph-conditioned sequence model: log_probs [B,L,num_letters] (reference :427-529).

        Extension for scoring many sequences on few structures (the reference's EnergyMPNN repeats one structure B
        times, energy_mpnn.py:42-62): X / mask / residue_idx / chain_encoding_all may be given with n < B rows (row r of
        S belongs to structure r % n) or as `expand`ed stride-0 views; in eval mode featurizer and encoder then run
        once per distinct structure.  The result equals the call on the materialised batch."""
        # The kernels compute in fp32-class arithmetic whatever autocast says (>= the precision the reference's bf16
        # autocast asks for, train.py:380-384); the ATen glue around them must not be re-typed under them.
        with _on_device_of(X), torch.autocast(device_type="cuda", enabled=False):
            return self._forward(X, S, mask, chain_M, residue_idx, chain_encoding_all, fix_order, fix_backbone_noise)

    def _forward(self, X, S, mask, chain_M, residue_idx, chain_encoding_all, fix_order, fix_backbone_noise):
        # autograd semantics as for any nn.Module: with grad mode on and trainable parameters the differentiable path
        # runs (in eval mode too: no dropout, no backbone noise), inference callers use torch.no_grad()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from . import training
            B = S.shape[0]  # a structure side given once (engine.shared_structures) is materialised for training
            X, mask, residue_idx, chain_encoding_all, _ = engine.shared_structures(X, S, mask, residue_idx,
                                                                                   chain_encoding_all)
            X, mask, chain_M, residue_idx, chain_encoding_all, fix_order, fix_backbone_noise = (
                engine._tile_rows(t, B) for t in (X, mask, chain_M, residue_idx, chain_encoding_all, fix_order,
                                                  fix_backbone_noise))
            return training.forward_train(self, X, S, mask, chain_M, residue_idx, chain_encoding_all, fix_order,
                                          fix_backbone_noise)
        return engine.forward(self, X, S, mask, chain_M, residue_idx, chain_encoding_all, fix_order,
                              fix_backbone_noise)

    def refresh_weights(self):
        """Forget the packed copies of the parameters the inference kernels cache.  Needed only after parameters were
        changed behind autograd's back (`p.data` edits); optimizer steps and `load_state_dict` are covered."""
        engine.invalidate_packed(self)
        return self

    def check_numerics(self):
        """Wait for the calls issued so far and raise RuntimeError if any of them (named by its call number) produced
        inf / NaN log-probs (an operand left the f16-split range of the matrix-pipe kernels, include/mpnn_hip.h OPERAND
        RANGE).  Without this call a bad call is only WARNED about (engine.NumericsWarning, with its call number), one
        call late and without any host synchronisation, when the model is called next - that call runs as usual."""
        g = self.__dict__.get("_numerics")
        if g is not None:
            g.check()
        return self

    def sample(self, X, mask, chain_M, residue_idx, chain_encoding_all, S_true=None, n_samples=1, temperature=0.1,
               seed=0, fix_order=None, logit_bias=None):
        """Autoregressive sampling (no counterpart in the reference; see engine.sample)."""
        with _on_device_of(X), torch.autocast(device_type="cuda", enabled=False):
            return engine.sample(self, X, mask, chain_M, residue_idx, chain_encoding_all, S_true, n_samples,
                                 temperature, seed, fix_order, logit_bias)
