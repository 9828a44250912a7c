"""Host-side orchestration of the gfx950 kernels (one HIP stream, no host syncs).

Data layout in HBM (fp32; N = B*L residues, K' = min(k, L), KT = ceil(K'/16)):
  atoms   [N][6][4]        backbone + virtual atoms
  ca_soa  [B][4][L]        CA coordinates + mask, structure-of-arrays for the k-NN sweep
  E_idx32 [N][16 KT]       neighbour list (int32, padded with self)
  hE      [N][8][KT][64][4] edge features in MFMA fragment layout (never [B,L,K,128] on the hot path)
  hV      [N][128]         node features
  PQ      [N][128 nproj]   per-node projections consumed by the next edge kernel
  gsum    [N][128], cnt[N] masked neighbour sums of the message MLP's second layer
Nothing of size [B,L,L] or [B,L,K,384/512] is ever materialised.
"""
from __future__ import annotations

from typing import Optional

import os
import warnings

import torch

from . import _lib
from ._lib import call, ptr, stream_ptr

H = 128


def _require_device(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"mpnn_amd: `{name}` must be on a ROCm device (got {t.device}); "
                           "the HIP path has no CPU fallback")


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def _i64(t):
    return t.detach().to(torch.int64).contiguous()


# ---------------------------------------------------------------------------
# weight packing (cached on the module; rebuilt when any parameter changes)
# ---------------------------------------------------------------------------
class PackedWeights:
    def __init__(self, model):
        self.key = self.make_key(model)
        dev = model.W_out.weight.device
        self.dev = dev
        st = stream_ptr()
        self._keep = []

        def pack(W, col0=0, ncols=None, out=None):
            W = _f32(W)
            self._keep.append(W)
            out_f, ld = W.shape
            ncols = ld - col0 if ncols is None else ncols
            nto, nti = (out_f + 15) // 16, (ncols + 15) // 16
            if out is None:
                out = torch.empty(nto * nti * 256, device=dev, dtype=torch.float32)
            call("mpnn_pack_linear", ptr(W), out_f, ncols, ld, col0, ptr(out), st)
            return out

        def pack_split(W, col0=0, ncols=None, rot=0, out=None):
            """f16 hi/lo split packing for the matrix-pipe kernels: [out/16][in/32][2][64] x 16 B."""
            W = _f32(W)
            self._keep.append(W)
            out_f, ld = W.shape
            ncols = ld - col0 if ncols is None else ncols
            nto, nkk = (out_f + 15) // 16, (ncols + 31) // 32
            if out is None:
                out = torch.empty(nto * nkk * 2 * 64 * 4, device=dev, dtype=torch.float32)  # 16 B per lane
            call("mpnn_pack_linear_split", ptr(W), out_f, ncols, ld, col0, rot, ptr(out), st)
            return out

        def vec(b):
            return _f32(b)

        def proj_group(blocks):
            """blocks: list of (W, col0, bias|None) -> (n split-packed 128x128 matrices, bias [n*128])."""
            n = len(blocks)
            if n == 0:
                return None, None, 0
            Wp = torch.empty(n * 16384, device=dev, dtype=torch.float32)  # n split-packed 128x128 (64 KiB each)
            bp = torch.zeros(n * H, device=dev, dtype=torch.float32)
            for i, (W, col0, b) in enumerate(blocks):
                pack_split(W, col0, H, out=Wp[i * 16384:(i + 1) * 16384])
                if b is not None:
                    bp[i * H:(i + 1) * H] = _f32(b)
            return Wp, bp, n

        f = model.features
        self.n_atoms = 6 if f.use_virtual_center else 5
        self.pos_table = torch.empty(66 * 16, device=dev, dtype=torch.float32)
        Wl, bl = _f32(f.embeddings.linear.weight), _f32(f.embeddings.linear.bias)
        self._keep += [Wl, bl]
        call("mpnn_pos_table", ptr(Wl), ptr(bl), 66, 16, ptr(self.pos_table), st)
        self.rbf_mu = torch.linspace(2.0, 22.0, 16, device=dev, dtype=torch.float32)
        self.Wemb = pack_split(f.edge_embedding.weight, rot=16)  # feature order: RBF blocks, then positional
        self.ln_e_w, self.ln_e_b = vec(f.norm_edges.weight), vec(f.norm_edges.bias)
        self.We, self.be = pack_split(model.W_e.weight), vec(model.W_e.bias)

        enc, dec = list(model.encoder_layers), list(model.decoder_layers)
        self.enc, self.dec = [], []
        for lyr in enc:
            d = dict(W1b=pack_split(lyr.W1.weight, H, H), W2=pack_split(lyr.W2.weight), b2=vec(lyr.W2.bias),
                     W3=pack_split(lyr.W3.weight), b3=vec(lyr.W3.bias),
                     W11b=pack_split(lyr.W11.weight, H, H), W12=pack_split(lyr.W12.weight), b12=vec(lyr.W12.bias),
                     W13=pack_split(lyr.W13.weight), b13=vec(lyr.W13.bias),
                     ln1w=vec(lyr.norm1.weight), ln1b=vec(lyr.norm1.bias), ln2w=vec(lyr.norm2.weight),
                     ln2b=vec(lyr.norm2.bias), ln3w=vec(lyr.norm3.weight), ln3b=vec(lyr.norm3.bias),
                     Win=pack_split(lyr.dense.W_in.weight), bin=vec(lyr.dense.W_in.bias),
                     Wout=torch.cat([pack_split(lyr.dense.W_out.weight, H * q, H) for q in range(4)]),
                     bout=vec(lyr.dense.W_out.bias), scale=float(lyr.scale))
            self.enc.append(d)
        for lyr in dec:
            d = dict(W1b=pack_split(lyr.W1.weight, H, H), W2=pack_split(lyr.W2.weight), b2=vec(lyr.W2.bias),
                     W3=pack_split(lyr.W3.weight), b3=vec(lyr.W3.bias),
                     ln1w=vec(lyr.norm1.weight), ln1b=vec(lyr.norm1.bias), ln2w=vec(lyr.norm2.weight),
                     ln2b=vec(lyr.norm2.bias),
                     Win=pack_split(lyr.dense.W_in.weight), bin=vec(lyr.dense.W_in.bias),
                     Wout=torch.cat([pack_split(lyr.dense.W_out.weight, H * q, H) for q in range(4)]),
                     Wout8=torch.cat([pack_split(lyr.dense.W_out.weight, 64 * q, 64) for q in range(8)]),
                     bout=vec(lyr.dense.W_out.bias), scale=float(lyr.scale))
            self.dec.append(d)

        # projection groups: what the NEXT edge kernel needs from the node features
        def enc_msg_blocks(l):  # P = W1a h + b1 ; Q = W1c h
            return [(enc[l].W1.weight, 0, enc[l].W1.bias), (enc[l].W1.weight, 2 * H, None)]

        def enc_upd_blocks(l):  # P' = W11a h + b11 ; Q' = W11c h
            return [(enc[l].W11.weight, 0, enc[l].W11.bias), (enc[l].W11.weight, 2 * H, None)]

        def dec_first_blocks():  # Pa_0, A_0 (= W1d_0 h + TS_0[S], table added by the kernel), PVenc_0, PVenc_1, ...
            if not dec:
                return []
            blocks = [(dec[0].W1.weight, 0, dec[0].W1.bias), (dec[0].W1.weight, 3 * H, None)]
            blocks += [(d_.W1.weight, 3 * H, None) for d_ in dec]
            return blocks

        def dec_next_blocks(l):
            return [(dec[l].W1.weight, 0, dec[l].W1.bias), (dec[l].W1.weight, 3 * H, None)]

        self.g_init = proj_group(enc_msg_blocks(0) if enc else dec_first_blocks())
        self.g_enc = []
        for l in range(len(enc)):
            nxt = enc_msg_blocks(l + 1) if l + 1 < len(enc) else dec_first_blocks()
            self.g_enc.append(proj_group(enc_upd_blocks(l) + nxt))
        self.g_dec = [proj_group(dec_next_blocks(l + 1)) if l + 1 < len(dec) else (None, None, 0)
                      for l in range(len(dec))]
        # per-letter table TS = [W1c_0 | W1c_1 | ...] W_s^T : the sequence term of every decoder layer
        self.n_vocab = model.W_s.weight.shape[0]
        self.Ws = _f32(model.W_s.weight)
        self.TS = None
        if dec:
            g_ts = proj_group([(d_.W1.weight, 2 * H, None) for d_ in dec])
            self.TS = torch.empty(self.n_vocab, len(dec) * H, device=dev, dtype=torch.float32)
            call("mpnn_node_update", ptr(self.Ws), None, None, 0.0, None, None, None, None, None, None, None, None,
                 None, None, None, 1.0, 0, ptr(g_ts[0]), ptr(g_ts[1]), g_ts[2], None, 0, 0, None, 0, self.n_vocab,
                 None, ptr(self.TS), st)
            self._keep.append(g_ts)
        self.Wlogit, self.blogit = pack(model.W_out.weight), vec(model.W_out.bias)
        self.n_letters = model.W_out.weight.shape[0]
        self.occ = None
        if model.occupancy_mlp is not None:
            m = model.occupancy_mlp
            self.occ = dict(w0=_f32(m[0].weight).reshape(-1), b0=vec(m[0].bias), W2=pack(m[2].weight),
                            b2=vec(m[2].bias), lnw=vec(m[3].weight), lnb=vec(m[3].bias))

    @staticmethod
    def make_key(model):
        return tuple((p.data_ptr(), p._version, str(p.device)) for p in model.parameters())


F16_MAX = 65504.0


def check_weight_range(model):
    """Every product on the path runs on the f16 matrix pipe with operands split as f16 hi + lo (include/mpnn_hip.h,
    OPERAND RANGE): a weight of magnitude >= 65504 (or a non-finite one) cannot be represented.  Checked once per
    packing, i.e. once per weight change - the only host synchronisation of the inference path, and none per call."""
    ps = [p.detach() for p in model.parameters() if p.numel()]
    if not ps:
        return
    amax = float(torch.stack(torch._foreach_norm(ps, float("inf"))).max())
    if not (amax < F16_MAX):
        raise RuntimeError(f"mpnn_amd: a parameter has magnitude {amax:.4g} (or is not finite); the MI355X kernels need "
                           f"|w| < {F16_MAX:.0f} (operands are split into f16 hi + lo halves for the matrix pipe)")


class NumericsWarning(RuntimeWarning):
    """An EARLIER call of this model produced inf / NaN log-probs (found without blocking at the start of a later call)."""


class NumericsGuard:
    """Device-side overflow flags of one model.  After every forward / sampling call `mpnn_range_flag` sets the call's
    own flag slot when a log-prob is inf / NaN - what an activation beyond the f16-split range (|x| >= 131008) turns
    into.  The slot travels to pinned host memory asynchronously.  Calls are numbered (`serial`, from 1), so a report
    names the call it belongs to:
      * `poll()` (start of the model's next call, non-blocking) only WARNS (NumericsWarning) about finished bad calls
        and lets the current call run: one bad request must not abort the valid one that follows it;
      * `check()` (`model.check_numerics()`, blocking) raises RuntimeError naming every bad call since the last check.
    Nothing here synchronises the stream on the hot path, and nothing is done while a HIP graph is being captured."""

    SLOTS = 32
    WHY = ("an activation or weight left the range of the f16-split matrix-pipe arithmetic (|x| < 65504, "
           "include/mpnn_hip.h OPERAND RANGE), or the inputs were not finite")

    def __init__(self, dev):
        self.flags = torch.zeros(self.SLOTS, device=dev, dtype=torch.int32)
        self.host = torch.zeros(self.SLOTS, dtype=torch.int32).pin_memory()
        self.pending = []  # (event, slot, serial, description)
        self.serial = 0    # calls watched so far
        self.failed = []   # (serial, description) of bad calls not yet raised by check()

    def watch(self, t, what):
        """-> the call's serial number (0 when nothing was recorded)."""
        if torch.cuda.is_current_stream_capturing() or os.environ.get("MPNN_NO_NUMERICS_GUARD") == "1":
            return 0  # (the switch is for timing-only ablation builds of the library, whose outputs are garbage)
        if len(self.pending) >= self.SLOTS:
            self.failed += self._collect(block=True)
        self.serial += 1
        slot = self.serial % self.SLOTS
        call("mpnn_range_flag", ptr(t), t.numel(), float("inf"), self.flags.data_ptr() + 4 * slot, 1, stream_ptr())
        self.host[slot:slot + 1].copy_(self.flags[slot:slot + 1], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((ev, slot, self.serial, what))
        return self.serial

    def _collect(self, block):
        """Finished calls leave `pending`; -> [(serial, description)] of those that were bad."""
        bad, keep = [], []
        for ev, slot, serial, what in self.pending:
            if block:
                ev.synchronize()
            if ev.query():
                if int(self.host[slot]) != 0:
                    bad.append((serial, what))
                    self.flags[slot:slot + 1].zero_()  # the slot is reused SLOTS calls later
                    self.host[slot] = 0
            else:
                keep.append((ev, slot, serial, what))
        self.pending = keep
        return bad

    def poll(self):
        if not self.pending or torch.cuda.is_current_stream_capturing():
            return
        for serial, what in self._collect(block=False):
            self.failed.append((serial, what))
            warnings.warn(NumericsWarning(f"mpnn_amd: call #{serial} of this model ({what}) produced inf / NaN log-probs: "
                                          f"{self.WHY}; the current call (#{self.serial + 1}) runs as usual"), stacklevel=4)

    def check(self):
        if torch.cuda.is_current_stream_capturing():
            return
        self.failed += self._collect(block=True)
        if self.failed:
            failed, self.failed = self.failed[-self.SLOTS:], []
            calls = ", ".join(f"#{n} ({w})" for n, w in failed)
            raise RuntimeError(f"mpnn_amd: call {calls} of {self.serial} produced inf / NaN log-probs: {self.WHY}")


def numerics_guard(model, dev) -> NumericsGuard:
    g = model.__dict__.get("_numerics")
    if g is None or g.flags.device != dev:
        g = NumericsGuard(dev)
        model.__dict__["_numerics"] = g
    return g


def invalidate_packed(model):
    """Drop every cached packed-weight set under `model`.  The cache key is (data_ptr, _version) per parameter, and
    torch's FUSED optimizers update parameters without touching `_version` (measured: AdamW(fused=True) leaves it
    unchanged), as does any edit through `.data`; the differentiable forward therefore calls this, so that whatever
    optimizer follows it, the next inference call packs the new weights."""
    for m in model.modules():
        m.__dict__.pop("_packed", None)


def packed(model) -> PackedWeights:
    pw = getattr(model, "_packed", None)
    if pw is None or pw.key != PackedWeights.make_key(model):
        check_weight_range(model)
        saved = _lib.set_precision("fp32")  # derived tables (TS) are always built with fp32-class products
        try:
            pw = PackedWeights(model)
        finally:
            _lib.set_precision(saved)
        model._packed = pw
    return pw


# ---------------------------------------------------------------------------
# featurizer pieces
# ---------------------------------------------------------------------------
def _geometry(features_mod, X, mask, backbone_noise, training):
    """atoms, ca_soa, E_idx64, E_idx32, D_nb, edge_mask, K  (reference :311-332)."""
    B, L = mask.shape
    dev = X.device
    st = stream_ptr()
    X = _f32(X)
    noise = None
    if backbone_noise is not None:
        noise = _f32(backbone_noise)
    elif training and features_mod.augment_eps > 0:
        noise = features_mod.augment_eps * torch.randn_like(X)
    atoms = torch.empty(B * L * 24, device=dev, dtype=torch.float32)
    ca_soa = torch.empty(B * 4 * L, device=dev, dtype=torch.float32)
    call("mpnn_backbone_atoms", ptr(X), ptr(noise), ptr(mask), B, L, ptr(atoms), ptr(ca_soa), st)
    K = min(int(features_mod.top_k), L)
    Kp = 16 * ((K + 15) // 16)
    cutoff = features_mod.edge_cutoff
    E_idx64 = torch.empty(B, L, K, device=dev, dtype=torch.int64)
    E_idx32 = torch.empty(B * L * Kp, device=dev, dtype=torch.int32)
    D_nb = torch.empty(B, L, K, device=dev, dtype=torch.float32)
    edge_mask = torch.empty(B, L, K, device=dev, dtype=torch.float32) if cutoff is not None else None
    call("mpnn_knn_topk", ptr(ca_soa), B, L, K, float(cutoff) if cutoff is not None else 0.0, ptr(E_idx64),
         ptr(E_idx32), ptr(D_nb), ptr(edge_mask), st)
    return atoms, ca_soa, E_idx64, E_idx32, D_nb, edge_mask, K


def _edge_features(pw, atoms, E_idx32, residue_idx, chain_labels, B, L, K, want_E):
    dev = atoms.device
    KT = (K + 15) // 16
    hE = torch.empty(B * L * 8 * KT * 256, device=dev, dtype=torch.float32)
    E_nat = torch.empty(B, L, K, H, device=dev, dtype=torch.float32) if want_E else None
    call("mpnn_edge_features", ptr(atoms), ptr(E_idx32), ptr(residue_idx), ptr(chain_labels), ptr(pw.pos_table),
         ptr(pw.rbf_mu), ptr(pw.Wemb), ptr(pw.ln_e_w), ptr(pw.ln_e_b), ptr(pw.We), ptr(pw.be), B, L, K, pw.n_atoms,
         ptr(E_nat), ptr(hE), stream_ptr())
    return hE, E_nat


def _occupancy(features_mod, atoms, mask, B, L):
    if features_mod.occupancy_cutoff is None:
        return None
    occ = torch.empty(B, L, 1, device=atoms.device, dtype=torch.float32)
    call("mpnn_occupancy", ptr(atoms), ptr(mask), B, L, float(features_mod.occupancy_cutoff), ptr(occ), stream_ptr())
    return occ


def _find_owner(features_mod):
    owner = getattr(features_mod, "_owner", None)
    return owner() if owner is not None else None


def features_standalone(features_mod, X, mask, residue_idx, chain_labels, backbone_noise=None):
    """ProteinFeatures.forward surface (reference :309-360): returns the natural-layout E."""
    _require_device(X, "X")
    mask = _f32(mask)
    B, L = mask.shape
    pw = _FeaturePack.get(features_mod)
    atoms, _, E_idx64, E_idx32, _, edge_mask, K = _geometry(features_mod, X, mask, backbone_noise,
                                                            features_mod.training)
    occ = _occupancy(features_mod, atoms, mask, B, L)
    _, E_nat = _edge_features(pw, atoms, E_idx32, _i64(residue_idx), _i64(chain_labels), B, L, K, True)
    return E_nat, E_idx64, occ, edge_mask


class _FeaturePack:
    """Packed weights for a stand-alone ProteinFeatures call (W_e is not part of that module, so an
    identity-free dummy is used and only E is consumed)."""

    @staticmethod
    def get(fm):
        key = tuple((p.data_ptr(), p._version) for p in fm.parameters())
        c = getattr(fm, "_packed", None)
        if c is not None and c.key == key:
            return c
        c = _FeaturePack()
        c.key = key
        dev = fm.edge_embedding.weight.device
        st = stream_ptr()
        c.n_atoms = 6 if fm.use_virtual_center else 5
        c.pos_table = torch.empty(66 * 16, device=dev, dtype=torch.float32)
        Wl, bl = _f32(fm.embeddings.linear.weight), _f32(fm.embeddings.linear.bias)
        call("mpnn_pos_table", ptr(Wl), ptr(bl), 66, 16, ptr(c.pos_table), st)
        c.rbf_mu = torch.linspace(2.0, 22.0, 16, device=dev, dtype=torch.float32)
        W = _f32(fm.edge_embedding.weight)
        nkk = (W.shape[1] + 31) // 32
        c.Wemb = torch.empty(8 * nkk * 512, device=dev, dtype=torch.float32)
        call("mpnn_pack_linear_split", ptr(W), H, W.shape[1], W.shape[1], 0, 16, ptr(c.Wemb), st)
        c.ln_e_w, c.ln_e_b = _f32(fm.norm_edges.weight), _f32(fm.norm_edges.bias)
        c.We = torch.zeros(8 * 4 * 512, device=dev, dtype=torch.float32)
        c.be = torch.zeros(H, device=dev, dtype=torch.float32)
        c._keep = (Wl, bl, W)
        fm._packed = c
        return c


# ---------------------------------------------------------------------------
# node kernel wrapper
# ---------------------------------------------------------------------------
def _node_update(hV_in, n_nodes, lw, gsum, cnt, cnt_const, mask, group, hV_out, PQ, table=None):
    """table = (TS, ld, off, S, block): adds TS[S[n]][off:off+128] to projection block `block`."""
    Wp, bp, nproj = group
    tb = (ptr(table[0]), table[1], table[2], ptr(table[3]), table[4]) if table is not None else (None, 0, 0, None, 0)
    if lw is None:
        call("mpnn_node_update", ptr(hV_in), None, None, 0.0, None, None, None, None, None, None, None, None, None,
             None, None, 1.0, 0, ptr(Wp), ptr(bp), nproj, *tb, n_nodes, None, ptr(PQ), stream_ptr())
    else:
        call("mpnn_node_update", ptr(hV_in), ptr(gsum), ptr(cnt), float(cnt_const), ptr(mask), ptr(lw["W3"]),
             ptr(lw["b3"]), ptr(lw["ln1w"]), ptr(lw["ln1b"]), ptr(lw["Win"]), ptr(lw["bin"]), ptr(lw["Wout"]),
             ptr(lw["bout"]), ptr(lw["ln2w"]), ptr(lw["ln2b"]), lw["scale"], 1, ptr(Wp), ptr(bp), nproj, *tb,
             n_nodes, ptr(hV_out), ptr(PQ), stream_ptr())


# ---------------------------------------------------------------------------
# full forward
# ---------------------------------------------------------------------------
def forward(model, X, S, mask, chain_M, residue_idx, chain_encoding_all, fix_order=None, fix_backbone_noise=None,
            return_intermediates: bool = False):
    """ProteinMPNN.forward (reference :427-529) on the HIP kernels, inference form (no autograd graph; the
    differentiable form is training.forward_train, which ProteinMPNN.forward picks in training mode)."""
    _require_device(X, "X")
    guard = numerics_guard(model, X.device)
    guard.poll()
    _lib.set_precision(getattr(model, "precision", "fp32"))
    try:
        X, mask, residue_idx, chain_encoding_all, n_struct = shared_structures(X, S, mask, residue_idx,
                                                                               chain_encoding_all)
        B = S.shape[0]
        if n_struct != B:
            # per-ROW backbone noise (B rows on n < B structures) makes every row its own structure: the shared path
            # applies only when the noise is absent or given per structure (reference adds it per row, :311-312)
            noise_rows = None if fix_backbone_noise is None else fix_backbone_noise.shape[0]
            if noise_rows == B and B > 1 and fix_backbone_noise.stride(0) == 0:
                fix_backbone_noise, noise_rows = fix_backbone_noise[:1].expand(n_struct, *fix_backbone_noise.shape[1:]), n_struct
            if (not model.training and not return_intermediates and model.features.top_k <= 48
                    and len(model.decoder_layers) > 0 and noise_rows in (None, n_struct)):
                out = _forward_shared(model, X, S, mask, chain_M, residue_idx, chain_encoding_all, fix_order,
                                      fix_backbone_noise)
                guard.watch(out, "a forward call")
                return out
            X, mask, residue_idx, chain_encoding_all = (_tile_rows(t, B) for t in
                                                        (X, mask, residue_idx, chain_encoding_all))
            if fix_backbone_noise is not None:
                fix_backbone_noise = _tile_rows(fix_backbone_noise, B)
        out = _forward(model, X, S, mask, _tile_rows(chain_M, B), residue_idx, chain_encoding_all,
                       _tile_rows(fix_order, B), fix_backbone_noise, return_intermediates)
        guard.watch(out["log_probs"] if return_intermediates else out, "a forward call")
        return out
    finally:
        _lib.set_precision("fp32")


def _tile_rows(t, B):
    """[n, ...] -> [B, ...] with row r = t[r % n] (the row order of shared-structure batches); None passes through."""
    if t is None or t.shape[0] == B:
        return t
    return t.repeat(B // t.shape[0], *([1] * (t.dim() - 1)))


def shared_structures(X, S, mask, residue_idx, chain_encoding_all):
    """Batches that score many sequences on few structures (the reference's EnergyMPNN repeats one structure B times,
    energy_mpnn.py:42-62) may pass the structure side ONCE: X / mask / residue_idx / chain_encoding_all with n rows and
    S (chain_M, fix_order) with B = m * n rows, row r belonging to structure r % n -- or as stride-0 `expand`ed views
    of one row.  -> the structure-side tensors cut to their n distinct rows, and n."""
    B = S.shape[0]
    side = (X, mask, residue_idx, chain_encoding_all)
    n = X.shape[0]
    if any(t.shape[0] != n for t in side):
        raise ValueError("X, mask, residue_idx and chain_encoding_all must have the same number of rows")
    if n == B and B > 1 and all(t.stride(0) == 0 for t in side):
        return X[:1], mask[:1], residue_idx[:1], chain_encoding_all[:1], 1
    if n < 1 or B % n != 0:
        raise ValueError(f"S has {B} rows, the structure side {n}: B must be a multiple of the structures")
    return X, mask, residue_idx, chain_encoding_all, n


def _decoding_rank(chain_M, mask, fix_order, B, L):
    """decoding order [B, L] (given, or drawn as the reference does, :489-491) and its inverse permutation `rank`."""
    dev = mask.device
    if fix_order is not None:
        order = _i64(fix_order)
    else:  # the argsort of a [B, L] key is host-side plumbing
        chain_M = _f32(chain_M) * mask
        order = torch.argsort((chain_M + 0.0001) * torch.abs(torch.randn(chain_M.shape, device=dev)))
    rank = torch.zeros(B * L, device=dev, dtype=torch.int32)
    call("mpnn_order_rank", ptr(order), B, L, ptr(rank), stream_ptr())
    return order, rank


def _forward(model, X, S, mask, chain_M, residue_idx, chain_encoding_all, fix_order, fix_backbone_noise,
             return_intermediates):
    dev = X.device
    mask = _f32(mask)
    B, L = mask.shape
    N = B * L
    S = _i64(S)
    residue_idx, chain_labels = _i64(residue_idx), _i64(chain_encoding_all)
    pw = packed(model)
    fm = model.features
    inter = {}

    atoms, _, E_idx64, E_idx32, D_nb, edge_mask, K = _geometry(fm, X, mask, fix_backbone_noise, model.training)
    occ = _occupancy(fm, atoms, mask, B, L)
    order = rank = None
    if len(pw.dec):
        order, rank = _decoding_rank(chain_M, mask, fix_order, B, L)
    if return_intermediates:
        inter.update(E_idx=E_idx64, D_neighbors=D_nb, edge_mask=edge_mask, occupancy=occ, h_V_enc=[], h_V_dec=[],
                     h_E_enc=[])
        if rank is not None:
            bw = torch.empty(B, L, K, device=dev, dtype=torch.float32)
            fw = torch.empty_like(bw)
            call("mpnn_order_masks", ptr(rank), ptr(E_idx64), ptr(mask), ptr(edge_mask), B, L, K, ptr(bw), ptr(fw),
                 stream_ptr())
            inter.update(mask_bw=bw, mask_fw=fw, order=order)
    log_probs = _layers(pw, B, L, K, atoms, E_idx32, edge_mask, occ, mask, S, residue_idx, chain_labels, rank,
                        inter if return_intermediates else None)
    if return_intermediates:
        inter["log_probs"] = log_probs
        return inter
    return log_probs


def _layers(pw, B, L, K, atoms, E_idx32, edge_mask, occ, mask, S, residue_idx, chain_labels, rank, inter):
    """Edge features, encoder, decoder, head on [B, L] residues whose neighbour ids are local to their row."""
    dev = atoms.device
    N = B * L
    st = stream_ptr()
    hE, _ = _edge_features(pw, atoms, E_idx32, residue_idx, chain_labels, B, L, K, False)

    hV = torch.zeros(N, H, device=dev, dtype=torch.float32)
    has_occ = pw.occ is not None and occ is not None
    if has_occ:
        o = pw.occ
        call("mpnn_occupancy_mlp", ptr(occ), ptr(o["w0"]), ptr(o["b0"]), ptr(o["W2"]), ptr(o["b2"]), ptr(o["lnw"]),
             ptr(o["lnb"]), N, ptr(hV), st)
    if inter is not None:
        inter["h_E0"] = edges_to_natural(hE, N, K).view(B, L, K, H)

    gsum = torch.empty(N, H, device=dev, dtype=torch.float32)
    cnt = torch.empty(N, device=dev, dtype=torch.float32)

    def new_pq(group):
        return torch.empty(N, max(group[2], 1) * H, device=dev, dtype=torch.float32) if group[2] else None

    n_enc, n_dec = len(pw.enc), len(pw.dec)
    ldts = pw.TS.shape[1] if pw.TS is not None else 0

    def dec_table(layer, block):  # adds TS_layer[S] to projection block `block` (the decoder's A rows)
        return (pw.TS, ldts, H * layer, S, block) if n_dec else None

    if n_enc and not has_occ:
        # h_V = 0: every node's P | Q row of the first message kernel is the bias row (stride 0), no launch needed
        PQ, ld = pw.g_init[1], 0
    else:
        PQ = new_pq(pw.g_init)
        ld = PQ.shape[1] if PQ is not None else 0
        if pw.g_init[2]:
            _node_update(hV, N, None, None, None, 0.0, None, pw.g_init, None, PQ,
                         table=dec_table(0, 1) if (n_enc == 0 and n_dec) else None)

    for l, lw in enumerate(pw.enc):
        ld = ld if l == 0 else PQ.shape[1]
        offP, offQ = (0, H) if l == 0 else (2 * H, 3 * H)
        call("mpnn_enc_message", ptr(hE), ptr(PQ), ld, offP, offQ, ptr(E_idx32), ptr(mask), ptr(edge_mask),
             ptr(lw["W1b"]), ptr(lw["W2"]), ptr(lw["b2"]), B, L, K, ptr(gsum), ptr(cnt), st)
        grp = pw.g_enc[l]
        PQn = new_pq(grp)
        hVn = torch.empty_like(hV)
        last = l + 1 == n_enc
        _node_update(hV, N, lw, gsum, cnt, 0.0, mask, grp, hVn, PQn,
                     table=dec_table(0, 3) if (last and n_dec) else None)
        hV, PQ = hVn, PQn
        call("mpnn_enc_edge_update", ptr(hE), ptr(PQ), PQ.shape[1], 0, H, ptr(E_idx32), ptr(lw["W11b"]),
             ptr(lw["W12"]), ptr(lw["b12"]), ptr(lw["W13"]), ptr(lw["b13"]), ptr(lw["ln3w"]), ptr(lw["ln3b"]), B, L, K,
             st)
        if inter is not None:
            inter["h_V_enc"].append(hV.view(B, L, H))
            inter["h_E_enc"].append(edges_to_natural(hE, N, K).view(B, L, K, H))

    # ---- decoder
    if n_dec:
        # projection buffer after the last encoder stage (g_init when there is no encoder):
        #   [P' | Q' |] Pa_0 | A_0 | PVenc_0 | PVenc_1 | ...      (A_l = W1d_l h + W1c_l W_s[S])
        base = 2 * H if n_enc else 0
        PQenc = PQ
        ldenc = PQenc.shape[1]
        PQd, offPa, offA = PQenc, base, base + H
        for l, lw in enumerate(pw.dec):
            call("mpnn_dec_message", ptr(hE), ptr(PQd), PQd.shape[1], offPa, offA, ptr(PQenc), ldenc,
                 base + 2 * H + H * l, ptr(rank), ptr(E_idx32), ptr(mask), ptr(edge_mask), ptr(lw["W1b"]),
                 ptr(lw["W2"]), ptr(lw["b2"]), B, L, K, ptr(gsum), st)
            grp = pw.g_dec[l]
            PQn = new_pq(grp)
            hVn = torch.empty_like(hV)
            _node_update(hV, N, lw, gsum, None, float(K), mask, grp, hVn, PQn,
                         table=dec_table(l + 1, 1) if grp[2] else None)
            hV = hVn
            if PQn is not None:
                PQd, offPa, offA = PQn, 0, H
            if inter is not None:
                inter["h_V_dec"].append(hV.view(B, L, H))

    log_probs = torch.empty(B, L, pw.n_letters, device=dev, dtype=torch.float32)
    call("mpnn_logits_log_softmax", ptr(hV), ptr(pw.Wlogit), ptr(pw.blogit), pw.n_letters, N, ptr(log_probs), st)
    return log_probs


# ---------------------------------------------------------------------------
# shared-structure forward: featurizer + encoder once per DISTINCT structure, decoder per row
# ---------------------------------------------------------------------------
def _forward_shared(model, X, S, mask, chain_M, residue_idx, chain_encoding_all, fix_order, fix_backbone_noise):
    """Same result as _forward on the materialised batch (row r = structure r % n): in eval mode featurizer and encoder
    do not see S or the decoding order, so their outputs are computed once and every decoder kernel reads them at
    (residue % (n L)).  Per-row state: h_V, the projections Pa | A (A carries the sequence term), rank, gsum."""
    dev = X.device
    mask = _f32(mask)
    n, L = mask.shape
    S = _i64(S)
    B = S.shape[0]
    m = B // n
    Ns = B * L
    pw = packed(model)
    st = stream_ptr()
    with torch.no_grad():
        enc = _encode(model, X, mask, _i64(residue_idx), _i64(chain_encoding_all), fix_backbone_noise)
        K = enc["K"]
        mask_s = _tile_rows(mask, B).contiguous()
        cm = _f32(_tile_rows(chain_M, B)) * mask_s
        if fix_order is not None:
            order = _i64(_tile_rows(fix_order, B))
        else:  # reference :489-491, one draw per row
            order = torch.argsort((cm + 0.0001) * torch.abs(torch.randn(cm.shape, device=dev)))
        rank = torch.zeros(Ns, device=dev, dtype=torch.int32)
        call("mpnn_order_rank", ptr(order), B, L, ptr(rank), st)
        base = 2 * H if len(pw.enc) else 0
        PQenc = enc["PQ"]
        hV = enc["hV"].repeat(m, 1)
        PQd = PQenc[:, base:base + 2 * H].repeat(m, 1)           # Pa_0 | A_0 per row ...
        PQd[:, H:] += pw.TS[S.reshape(-1), :H]                    # ... + W1c_0 W_s[S]  (reference :478-483)
        offPa, offA = 0, H
        gsum = torch.empty(Ns, H, device=dev, dtype=torch.float32)
        ldts = pw.TS.shape[1]
        for l, lw in enumerate(pw.dec):
            call("mpnn_dec_message_shared", ptr(enc["hE"]), ptr(PQd), PQd.shape[1], offPa, offA, ptr(PQenc),
                 PQenc.shape[1], base + 2 * H + H * l, ptr(rank), ptr(enc["E_idx32"]), ptr(mask_s),
                 ptr(enc["edge_mask"]), ptr(lw["W1b"]), ptr(lw["W2"]), ptr(lw["b2"]), B, n, L, K, ptr(gsum), st)
            grp = pw.g_dec[l]
            PQn = torch.empty(Ns, grp[2] * H, device=dev, dtype=torch.float32) if grp[2] else None
            hVn = torch.empty_like(hV)
            _node_update(hV, Ns, lw, gsum, None, float(K), mask_s, grp, hVn, PQn,
                         table=(pw.TS, ldts, H * (l + 1), S, 1) if grp[2] else None)
            hV = hVn
            if PQn is not None:
                PQd, offPa, offA = PQn, 0, H
        log_probs = torch.empty(B, L, pw.n_letters, device=dev, dtype=torch.float32)
        call("mpnn_logits_log_softmax", ptr(hV), ptr(pw.Wlogit), ptr(pw.blogit), pw.n_letters, Ns, ptr(log_probs), st)
    return log_probs


# ---------------------------------------------------------------------------
# autoregressive sampling (BASELINE configs[4]; no reference function, contract in include/mpnn_hip.h)
# ---------------------------------------------------------------------------
def _encode(model, X, mask, residue_idx, chain_labels, backbone_noise, geometry=None):
    """Featurizer + encoder stack (reference :441-470) -> dict(hE, hV, PQ, K, E_idx64, E_idx32, edge_mask).
    PQ's trailing blocks are Pa_0 | A_0 (WITHOUT the sequence term) | PVenc_0 | PVenc_1 | ...
    geometry: the result of _geometry when the caller has already run the neighbour search (sampling does, to
    schedule its levels beside the encoder)."""
    dev = X.device
    B, L = mask.shape
    N = B * L
    pw = packed(model)
    st = stream_ptr()
    fm = model.features
    if geometry is None:
        geometry = _geometry(fm, X, mask, backbone_noise, model.training)
    atoms, _, E_idx64, E_idx32, _, edge_mask, K = geometry
    occ = _occupancy(fm, atoms, mask, B, L)
    hE, _ = _edge_features(pw, atoms, E_idx32, residue_idx, chain_labels, B, L, K, False)
    hV = torch.zeros(N, H, device=dev, dtype=torch.float32)
    if pw.occ is not None and occ is not None:
        o = pw.occ
        call("mpnn_occupancy_mlp", ptr(occ), ptr(o["w0"]), ptr(o["b0"]), ptr(o["W2"]), ptr(o["b2"]), ptr(o["lnw"]),
             ptr(o["lnb"]), N, ptr(hV), st)
    gsum = torch.empty(N, H, device=dev, dtype=torch.float32)
    cnt = torch.empty(N, device=dev, dtype=torch.float32)

    def new_pq(group):
        return torch.empty(N, max(group[2], 1) * H, device=dev, dtype=torch.float32) if group[2] else None

    if pw.enc and not (pw.occ is not None and occ is not None):
        PQ, ld = pw.g_init[1], 0   # h_V = 0: the bias row for every node (see _layers)
    else:
        PQ = new_pq(pw.g_init)
        ld = PQ.shape[1] if PQ is not None else 0
        if pw.g_init[2]:
            _node_update(hV, N, None, None, None, 0.0, None, pw.g_init, None, PQ)
    for l, lw in enumerate(pw.enc):
        ld = ld if l == 0 else PQ.shape[1]
        offP, offQ = (0, H) if l == 0 else (2 * H, 3 * H)
        call("mpnn_enc_message", ptr(hE), ptr(PQ), ld, offP, offQ, ptr(E_idx32), ptr(mask), ptr(edge_mask),
             ptr(lw["W1b"]), ptr(lw["W2"]), ptr(lw["b2"]), B, L, K, ptr(gsum), ptr(cnt), st)
        grp = pw.g_enc[l]
        PQn, hVn = new_pq(grp), torch.empty_like(hV)
        _node_update(hV, N, lw, gsum, cnt, 0.0, mask, grp, hVn, PQn)
        hV, PQ = hVn, PQn
        call("mpnn_enc_edge_update", ptr(hE), ptr(PQ), PQ.shape[1], 0, H, ptr(E_idx32), ptr(lw["W11b"]),
             ptr(lw["W12"]), ptr(lw["b12"]), ptr(lw["W13"]), ptr(lw["b13"]), ptr(lw["ln3w"]), ptr(lw["ln3b"]), B, L, K,
             st)
    return dict(hE=hE, hV=hV, PQ=PQ, K=K, E_idx64=E_idx64, E_idx32=E_idx32, edge_mask=edge_mask)


def decode_levels(order, rank, E_idx32, n_rows, enc_rows, L, K):
    """Dependency level of every residue of every sample (mpnn_decode_levels): residues of one level only read
    finished residues, so a level is one batched launch.
    -> (residue ids sorted by level [n_rows*L] int32, residues per level: list of int)."""
    perm, level = _decode_levels_async(order, rank, E_idx32, n_rows, enc_rows, L, K)
    return perm, torch.bincount(level)[1:].tolist()


def _decode_levels_async(order, rank, E_idx32, n_rows, enc_rows, L, K):
    """The device part of decode_levels (no host read): -> (perm, level)."""
    level = torch.empty(n_rows * L, device=order.device, dtype=torch.int32)
    call("mpnn_decode_levels", ptr(order), ptr(rank), ptr(E_idx32), n_rows, enc_rows, L, K, ptr(level), stream_ptr())
    perm = torch.argsort(level, stable=True).to(torch.int32)
    return perm, level


_SIDE_STREAMS = {}


def _side_stream(dev):
    s = _SIDE_STREAMS.get(dev)
    if s is None:
        s = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return s


def sample(model, X, mask, chain_M, residue_idx, chain_encoding_all, S_true=None, n_samples=1, temperature=0.1,
           seed=0, fix_order=None, logit_bias=None, backbone_noise=None):
    """Draw n_samples sequences per structure with a random (or given) decoding order.

    Returns dict(S [n_samples,B,L] int64, log_probs [n_samples,B,L,21] (the model's log-probs at every residue, as a
    teacher-forced forward on S with the same order returns them), order [n_samples,B,L]).  Residues with
    chain_M * mask == 0 are not designed: they keep S_true.  The featurizer and encoder run once per structure."""
    _require_device(X, "X")
    guard = numerics_guard(model, X.device)
    guard.poll()
    _lib.set_precision(getattr(model, "precision", "fp32"))
    try:
        out = _sample(model, X, mask, chain_M, residue_idx, chain_encoding_all, S_true, n_samples, temperature, seed,
                      fix_order, logit_bias, backbone_noise)
        guard.watch(out["log_probs"], "a sampling call")
        return out
    finally:
        _lib.set_precision("fp32")


def _sample(model, X, mask, chain_M, residue_idx, chain_encoding_all, S_true, n_samples, temperature, seed, fix_order,
            logit_bias, backbone_noise):
    dev = X.device
    mask = _f32(mask)
    B, L = mask.shape
    N = B * L
    pw = packed(model)
    n_dec = len(pw.dec)
    if n_dec == 0 or n_dec > 4:
        raise ValueError("mpnn_amd.sample needs 1..4 decoder layers")
    st = stream_ptr()
    with torch.no_grad():
        # The level schedule depends on the neighbour lists and the decoding orders only, and the host must read the
        # level sizes before it can issue the level loop.  So: neighbour search first; orders, ranks, levels and their
        # sort on a SIDE stream while the main stream runs edge features and the encoder; the one host read of the call
        # waits for the side stream only - the encoder is already enqueued and keeps the GPU busy meanwhile.
        geometry = _geometry(model.features, X, mask, backbone_noise, model.training)
        K = geometry[6]
        if K > 64:
            raise NotImplementedError("mpnn_amd.sample covers k_neighbors <= 64")
        main = torch.cuda.current_stream()
        side = _side_stream(dev)
        Ns = n_samples * N
        cm = (_f32(chain_M) * mask)
        if fix_order is not None:
            order = _i64(fix_order)
            if order.shape[-2:] != (B, L) or (order.dim() == 3 and order.shape[0] != n_samples):
                raise ValueError(f"fix_order must be [B, L] or [n_samples, B, L] = [{n_samples}, {B}, {L}], "
                                 f"got {tuple(order.shape)}")
            # the level scheduler indexes with the order: it must be a permutation of 0..L-1 per row
            if not bool((torch.sort(order, dim=-1).values == torch.arange(L, device=dev)).all()):
                raise ValueError("fix_order: every row must be a permutation of 0..L-1")
            order = order.expand(n_samples, B, L).contiguous() if order.dim() == 2 else order.contiguous()
        else:  # reference :489-491, one draw per sample
            g_ = torch.Generator(device=dev)
            g_.manual_seed(int(seed))
            noise = torch.randn(n_samples, B, L, device=dev, generator=g_)
            order = torch.argsort((cm.unsqueeze(0) + 0.0001) * torch.abs(noise), dim=-1)
        geo_done = torch.cuda.Event()
        geo_done.record(main)          # the orders and the neighbour lists were produced on the main stream
        side.wait_event(geo_done)
        with torch.cuda.stream(side):
            rank = torch.zeros(Ns, device=dev, dtype=torch.int32)
            call("mpnn_order_rank", ptr(order), n_samples * B, L, ptr(rank), stream_ptr())
            perm, level = _decode_levels_async(order, rank, geometry[3], n_samples * B, B, L, K)
        enc = _encode(model, X, mask, _i64(residue_idx), _i64(chain_encoding_all), backbone_noise, geometry=geometry)
        for t in (rank, perm, level):
            t.record_stream(main)      # allocated on the side stream, used on the main one from here on

        S_in = _i64(S_true).reshape(1, N).expand(n_samples, N).contiguous() if S_true is not None else \
            torch.zeros(n_samples, N, device=dev, dtype=torch.int64)
        S_out = torch.zeros_like(S_in)
        designable = cm.reshape(1, N).expand(n_samples, N).contiguous()
        mask_s = mask.reshape(1, N).expand(n_samples, N).contiguous()
        log_probs = torch.zeros(Ns, pw.n_letters, device=dev, dtype=torch.float32)
        # sample-side state: layer inputs hV_l, projections [Pa_l | A_l]; encoder rows PVenc_l stay shared
        base = 2 * H if len(pw.enc) else 0
        PQenc = enc["PQ"]
        hVs = [enc["hV"].repeat(n_samples, 1)] + [torch.empty(Ns, H, device=dev) for _ in range(n_dec)]
        PA = [PQenc[:, base:base + 2 * H].repeat(n_samples, 1)] + \
             [torch.empty(Ns, 2 * H, device=dev) for _ in range(n_dec - 1)]
        # "split" launches only: per-tile neighbour sums [residue][4][128], written by the message kernel, added in tile
        # order by the node kernel (sized below, once the mode is known)
        gsum = None
        import ctypes
        A_rows = (ctypes.c_void_p * n_dec)(*[t.data_ptr() for t in PA])
        A_ld = (ctypes.c_int * n_dec)(*([2 * H] * n_dec))
        A_off = (ctypes.c_int * n_dec)(*([H] * n_dec))
        lb = _f32(logit_bias).to(dev) if logit_bias is not None else None
        # ~7 launches per level and ~75 levels: argument lists are built once, only (list pointer, count) change
        lib = _lib.load()
        plans = []
        for l, lw in enumerate(pw.dec):
            grp = pw.g_dec[l]
            msg = [ptr(enc["hE"]), ptr(PA[l]), 2 * H, 0, ptr(PA[l]), 2 * H, H, ptr(PQenc), PQenc.shape[1],
                   base + 2 * H + H * l, ptr(rank), ptr(enc["E_idx32"]), ptr(mask_s), ptr(enc["edge_mask"]),
                   ptr(lw["W1b"]), ptr(lw["W2"]), ptr(lw["b2"]), None, 0, N, L, K, None, st]
            node = [ptr(hVs[l]), None, (K + 15) // 16, float(K), ptr(mask_s), ptr(lw["W3"]), ptr(lw["b3"]), ptr(lw["ln1w"]),
                    ptr(lw["ln1b"]), ptr(lw["Win"]), ptr(lw["bin"]), ptr(lw["Wout"]), ptr(lw["bout"]), ptr(lw["ln2w"]),
                    ptr(lw["ln2b"]), lw["scale"], ptr(grp[0]), ptr(grp[1]), grp[2], None, 0, ptr(hVs[l + 1]),
                    ptr(PA[l + 1]) if grp[2] else None, st]
            # both in one launch (mpnn_dec_layer_subset): the message arguments without gsum, then the node update's
            layer = msg[:22] + [node[0], node[3]] + node[5:19] + [node[21], node[22], st]
            layer[30] = ptr(lw["Wout8"])  # the fused kernel splits the FFN over eight waves: W_out in 64-column blocks
            plans.append((msg, node, layer))
        head = [ptr(hVs[n_dec]), ptr(pw.Wlogit), ptr(pw.blogit), ptr(lb), pw.n_letters, None, 0, float(temperature),
                int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(designable), ptr(S_in), ptr(S_out), ptr(log_probs), ptr(pw.TS),
                pw.TS.shape[1], n_dec, A_rows, A_ld, A_off, st]
        # MPNN_SAMPLE_LAUNCHES (the default is what ships; the others exist for tests and A/B timing):
        #   "run"        the whole level loop in ONE library call (mpnn_decode_run): per level one launch per decoder
        #                layer, the last one also drawing (mpnn_dec_layer_sample_subset)
        #   "run/head"   the same with the draw in a launch of its own (mpnn_sample_subset)
        #   "layer", "layer/head"   the same launches, issued from this Python loop (round 2's form: equal device time,
        #                more host time)
        #   "split"      message / node update / head launches (the only form for K > 48: four column tiles)
        mode = os.environ.get("MPNN_SAMPLE_LAUNCHES", "run")
        if mode not in ("run", "run/head", "layer", "layer/head", "split"):
            raise ValueError(f"MPNN_SAMPLE_LAUNCHES={mode!r}: expected run, run/head, layer, layer/head or split")
        run_in_c = mode.startswith("run")
        head_fused = mode in ("run", "layer") and plans[-1][2][37] == 0   # the last decoder layer has no projections
        if mode != "split":
            mode = "layer"
        if os.environ.get("MPNN_SAMPLE_UNFUSED") == "1" or K > 48:
            mode = "split"  # the fused level kernels give a workgroup four residues x three column tiles (K <= 48)
        if mode == "split":
            gsum = torch.empty(Ns, 4, H, device=dev, dtype=torch.float32)
            for msg, node, _ in plans:
                msg[22], node[1] = ptr(gsum), ptr(gsum)
        f_layer = getattr(lib, _lib.entry("mpnn_dec_layer_subset"))
        f_last = getattr(lib, _lib.entry("mpnn_dec_layer_sample_subset"))
        last = plans[-1][2][:35] + [plans[-1][2][38]] + head[1:5] + head[7:19] + [st]
        f_msg = getattr(lib, _lib.entry("mpnn_dec_message_subset"))
        f_node = getattr(lib, _lib.entry("mpnn_node_update_subset"))
        f_head = lib.mpnn_sample_subset
        if _lib._profile is not None:  # bench.py's per-entry-point timing goes through the generic wrapper
            f_layer = lambda *a: (call("mpnn_dec_layer_subset", *a), 0)[1]
            f_last = lambda *a: (call("mpnn_dec_layer_sample_subset", *a), 0)[1]
            f_msg = lambda *a: (call("mpnn_dec_message_subset", *a), 0)[1]
            f_node = lambda *a: (call("mpnn_node_update_subset", *a), 0)[1]
            f_head = lambda *a: (call("mpnn_sample_subset", *a), 0)[1]
        layers = (_lib.LevelLayer * n_dec)()
        for l, lw in enumerate(pw.dec):
            grp = pw.g_dec[l]
            y = layers[l]
            y.P, y.ldp, y.offP = ptr(PA[l]), 2 * H, 0
            y.RowA, y.lda, y.offA = ptr(PA[l]), 2 * H, H
            y.offEnc = base + 2 * H + H * l
            y.W1b_packed, y.W2_packed, y.b2 = ptr(lw["W1b"]), ptr(lw["W2"]), ptr(lw["b2"])
            y.hV_in, y.hV_out = ptr(hVs[l]), ptr(hVs[l + 1])
            y.W3_packed, y.b3, y.ln1_w, y.ln1_b = ptr(lw["W3"]), ptr(lw["b3"]), ptr(lw["ln1w"]), ptr(lw["ln1b"])
            y.Win_packed, y.b_in, y.Wout8_packed, y.b_out = ptr(lw["Win"]), ptr(lw["bin"]), ptr(lw["Wout8"]), ptr(lw["bout"])
            y.ln2_w, y.ln2_b, y.scale = ptr(lw["ln2w"]), ptr(lw["ln2b"]), lw["scale"]
            y.Wproj_packed, y.bproj, y.nproj = ptr(grp[0]), ptr(grp[1]), grp[2]
            y.PQ = ptr(PA[l + 1]) if grp[2] else None
        # everything above (state tensors, argument lists) was prepared without knowing the level sizes; now the call's
        # one host read, which waits for the side stream only
        with torch.cuda.stream(side):
            counts = torch.bincount(level)[1:].tolist()
        main.wait_stream(side)
        list_ptr = perm.data_ptr()
        start = 0
        rc = 0
        if run_in_c and mode == "layer" and plans[-1][2][37] == 0:
            c_counts = (ctypes.c_int * len(counts))(*counts)
            call("mpnn_decode_run", ptr(enc["hE"]), ptr(PQenc), PQenc.shape[1], ptr(rank), ptr(enc["E_idx32"]),
                 ptr(mask_s), ptr(enc["edge_mask"]), list_ptr, c_counts, len(counts), N, L, K, float(K), layers, n_dec,
                 ptr(pw.Wlogit), ptr(pw.blogit), ptr(lb), pw.n_letters, float(temperature),
                 int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(designable), ptr(S_in), ptr(S_out), ptr(log_probs), ptr(pw.TS),
                 pw.TS.shape[1], 1 if head_fused else 0, st)
            counts_loop = []
        else:
            counts_loop = counts
        for cnt_l in counts_loop:
            p_nodes = list_ptr + 4 * start
            start += cnt_l
            if mode == "layer" and head_fused:
                for msg, node, layer in plans[:-1]:
                    layer[17], layer[18] = p_nodes, cnt_l
                    rc = rc or f_layer(*layer)
                last[17], last[18] = p_nodes, cnt_l
                rc = rc or f_last(*last)
                continue
            for msg, node, layer in plans:
                if mode == "layer":
                    layer[17], layer[18] = p_nodes, cnt_l
                    rc = rc or f_layer(*layer)
                else:
                    msg[17], msg[18] = p_nodes, cnt_l
                    node[19], node[20] = p_nodes, cnt_l
                    rc = rc or f_msg(*msg) or f_node(*node)
            head[5], head[6] = p_nodes, cnt_l
            rc = rc or f_head(*head)
        if rc:
            raise RuntimeError(f"mpnn_amd.sample: kernel launch failed (code {rc}): "
                               f"{lib.mpnn_last_error().decode(errors='replace')}")
    return dict(S=S_out.view(n_samples, B, L), log_probs=log_probs.view(n_samples, B, L, -1),
                order=order.view(n_samples, B, L), n_levels=len(counts))


def edges_to_natural(hE, n_nodes, K):
    out = torch.empty(n_nodes, K, H, device=hE.device, dtype=torch.float32)
    call("mpnn_edges_to_natural", ptr(hE), n_nodes, K, ptr(out), stream_ptr())
    return out


def edges_from_natural(h_E_nat, n_nodes, K):
    KT = (K + 15) // 16
    out = torch.empty(n_nodes * 8 * KT * 256, device=h_E_nat.device, dtype=torch.float32)
    call("mpnn_edges_from_natural", ptr(h_E_nat), n_nodes, K, ptr(out), stream_ptr())
    return out


# ---------------------------------------------------------------------------
# stand-alone EncLayer surface (reference :73-95): natural layouts in and out
# ---------------------------------------------------------------------------
def enc_layer_standalone(layer, h_V, h_E, E_idx, mask_V=None, mask_attend=None):
    _require_device(h_V, "h_V")
    dev = h_V.device
    B, L, K = E_idx.shape
    N = B * L
    st = stream_ptr()
    key = tuple((p.data_ptr(), p._version) for p in layer.parameters())
    c = getattr(layer, "_packed", None)
    if c is None or c["key"] != key:
        def pack_split(W, col0=0, ncols=None):
            W = _f32(W)
            out_f, ld = W.shape
            ncols = ld - col0 if ncols is None else ncols
            out = torch.empty(((out_f + 15) // 16) * ((ncols + 31) // 32) * 512, device=dev, dtype=torch.float32)
            call("mpnn_pack_linear_split", ptr(W), out_f, ncols, ld, col0, 0, ptr(out), st)
            return out
        Wp = torch.cat([pack_split(layer.W1.weight, 0, H), pack_split(layer.W1.weight, 2 * H, H)])
        bp = torch.cat([_f32(layer.W1.bias), torch.zeros(H, device=dev)])
        Wp2 = torch.cat([pack_split(layer.W11.weight, 0, H), pack_split(layer.W11.weight, 2 * H, H)])
        bp2 = torch.cat([_f32(layer.W11.bias), torch.zeros(H, device=dev)])
        c = dict(key=key, g1=(Wp, bp, 2), g2=(Wp2, bp2, 2), W1b=pack_split(layer.W1.weight, H, H),
                 W2=pack_split(layer.W2.weight),
                 b2=_f32(layer.W2.bias), W3=pack_split(layer.W3.weight), b3=_f32(layer.W3.bias),
                 W11b=pack_split(layer.W11.weight, H, H), W12=pack_split(layer.W12.weight), b12=_f32(layer.W12.bias),
                 W13=pack_split(layer.W13.weight), b13=_f32(layer.W13.bias), ln1w=_f32(layer.norm1.weight),
                 ln1b=_f32(layer.norm1.bias), ln2w=_f32(layer.norm2.weight), ln2b=_f32(layer.norm2.bias),
                 ln3w=_f32(layer.norm3.weight), ln3b=_f32(layer.norm3.bias), Win=pack_split(layer.dense.W_in.weight),
                 bin=_f32(layer.dense.W_in.bias),
                 Wout=torch.cat([pack_split(layer.dense.W_out.weight, H * q, H) for q in range(4)]),
                 bout=_f32(layer.dense.W_out.bias), scale=float(layer.scale))
        layer._packed = c
    hV = _f32(h_V).view(N, H)
    hE = edges_from_natural(_f32(h_E), N, K)
    Kp = 16 * ((K + 15) // 16)
    idx32 = torch.arange(L, device=dev, dtype=torch.int32).view(1, L, 1).expand(B, L, Kp).contiguous()
    idx32[:, :, :K] = E_idx.to(torch.int32)
    ones = torch.ones(N, device=dev, dtype=torch.float32)
    m_att = _f32(mask_attend) if mask_attend is not None else None
    m_V = _f32(mask_V).reshape(N) if mask_V is not None else ones
    PQ = torch.empty(N, 2 * H, device=dev, dtype=torch.float32)
    _node_update(hV, N, None, None, None, 0.0, None, c["g1"], None, PQ)
    gsum = torch.empty(N, H, device=dev, dtype=torch.float32)
    cnt = torch.empty(N, device=dev, dtype=torch.float32)
    call("mpnn_enc_message", ptr(hE), ptr(PQ), 2 * H, 0, H, ptr(idx32), ptr(ones), ptr(m_att), ptr(c["W1b"]),
         ptr(c["W2"]), ptr(c["b2"]), B, L, K, ptr(gsum), ptr(cnt), st)
    hVn = torch.empty_like(hV)
    PQ2 = torch.empty(N, 2 * H, device=dev, dtype=torch.float32)
    _node_update(hV, N, c, gsum, cnt, 0.0, m_V, c["g2"], hVn, PQ2)
    call("mpnn_enc_edge_update", ptr(hE), ptr(PQ2), 2 * H, 0, H, ptr(idx32), ptr(c["W11b"]), ptr(c["W12"]),
         ptr(c["b12"]), ptr(c["W13"]), ptr(c["b13"]), ptr(c["ln3w"]), ptr(c["ln3b"]), B, L, K, st)
    return hVn.view(B, L, H), edges_to_natural(hE, N, K).view(B, L, K, H)


# ---------------------------------------------------------------------------
# stand-alone PositionWiseFeedForward / PositionalEncodings surfaces (reference :151-155, :168-177)
# ---------------------------------------------------------------------------
def _pack_split_new(W, dev, col0=0, ncols=None):
    W = _f32(W)
    out_f, ld = W.shape
    ncols = ld - col0 if ncols is None else ncols
    out = torch.empty(((out_f + 15) // 16) * ((ncols + 31) // 32) * 512, device=dev, dtype=torch.float32)
    call("mpnn_pack_linear_split", ptr(W), out_f, ncols, ld, col0, 0, ptr(out), stream_ptr())
    return out


def ffn_standalone(ffn, h_V):
    _require_device(h_V, "h_V")
    dev = h_V.device
    if tuple(ffn.W_in.weight.shape) != (4 * H, H) or tuple(ffn.W_out.weight.shape) != (H, 4 * H):
        raise ValueError("mpnn_amd.PositionWiseFeedForward runs 128 -> 512 -> 128 only (the reference's only use)")
    key = tuple((p.data_ptr(), p._version) for p in ffn.parameters())
    c = ffn.__dict__.get("_packed")
    if c is None or c["key"] != key:
        c = dict(key=key, Win=_pack_split_new(ffn.W_in.weight, dev), bin=_f32(ffn.W_in.bias),
                 Wout=torch.cat([_pack_split_new(ffn.W_out.weight, dev, H * q, H) for q in range(4)]),
                 bout=_f32(ffn.W_out.bias))
        ffn.__dict__["_packed"] = c
    x = _f32(h_V).reshape(-1, H)
    n = x.shape[0]
    out = torch.empty(n, H, device=dev, dtype=torch.float32)
    if n:
        call("mpnn_node_update", ptr(x), None, None, 0.0, None, None, None, None, None, ptr(c["Win"]), ptr(c["bin"]),
             ptr(c["Wout"]), ptr(c["bout"]), None, None, 1.0, 2, None, None, 0, None, 0, 0, None, 0, n, ptr(out), None,
             stream_ptr())
    return out.view(*h_V.shape[:-1], H)


def positional_encodings_standalone(pe, offset, mask):
    _require_device(offset, "offset")
    dev = offset.device
    nb, npos = 2 * pe.max_relative_feature + 2, pe.num_embeddings
    key = tuple((p.data_ptr(), p._version) for p in pe.parameters())
    c = pe.__dict__.get("_table")
    if c is None or c[0] != key:
        table = torch.empty(nb * npos, device=dev, dtype=torch.float32)
        Wl, bl = _f32(pe.linear.weight), _f32(pe.linear.bias)
        call("mpnn_pos_table", ptr(Wl), ptr(bl), nb, npos, ptr(table), stream_ptr())
        c = (key, table, (Wl, bl))
        pe.__dict__["_table"] = c
    off, m = _i64(offset), _i64(mask.expand_as(offset))
    out = torch.empty(*offset.shape, npos, device=dev, dtype=torch.float32)
    if off.numel():
        call("mpnn_positional_encodings", ptr(off), ptr(m), off.numel(), pe.max_relative_feature, ptr(c[1]), npos,
             ptr(out), stream_ptr())
    return out


# ---------------------------------------------------------------------------
# stand-alone DecLayer surface (reference :118-138): h_ESV materialised by the caller, natural layouts
# ---------------------------------------------------------------------------
def dec_layer_standalone(layer, h_V, h_ESV, mask_V=None, mask_attend=None):
    """The fused decoder inside ProteinMPNN.forward never materialises h_ESV; this entry serves callers that do.
    W1 [h_V_i | h_ESV_ik] = (W1a h_V_i + b1) + W1r h_ESV_ik: the per-edge term is mpnn_linear_rows (a Linear(384 -> 128)
    over the B L K rows on the matrix pipe); its result takes the place of the edge tensor in the message kernel
    (identity as the per-edge weight, no gathered row), then the shared node kernel finishes the layer."""
    _require_device(h_V, "h_V")
    dev = h_V.device
    B, L, K, C = h_ESV.shape
    N = B * L
    st = stream_ptr()
    if K > 64 or C != layer.W1.weight.shape[1] - H or C % H:
        raise ValueError("DecLayer: h_ESV must be [B, L, K <= 64, num_in], num_in a multiple of 128")
    key = tuple((p.data_ptr(), p._version) for p in layer.parameters())
    c = getattr(layer, "_packed", None)
    if c is None or c["key"] != key:
        def pack_split(W, col0=0, ncols=None):
            W = _f32(W)
            out_f, ld = W.shape
            ncols = ld - col0 if ncols is None else ncols
            out = torch.empty(((out_f + 15) // 16) * ((ncols + 31) // 32) * 512, device=dev, dtype=torch.float32)
            call("mpnn_pack_linear_split", ptr(W), out_f, ncols, ld, col0, 0, ptr(out), st)
            return out
        c = dict(key=key, g1=(pack_split(layer.W1.weight, 0, H), _f32(layer.W1.bias), 1),
                 eye=pack_split(torch.eye(H, device=dev)),
                 Wr=torch.cat([pack_split(layer.W1.weight, H * (1 + q), H) for q in range(C // H)]),
                 W2=pack_split(layer.W2.weight), b2=_f32(layer.W2.bias), W3=pack_split(layer.W3.weight),
                 b3=_f32(layer.W3.bias), ln1w=_f32(layer.norm1.weight), ln1b=_f32(layer.norm1.bias),
                 ln2w=_f32(layer.norm2.weight), ln2b=_f32(layer.norm2.bias), Win=pack_split(layer.dense.W_in.weight),
                 bin=_f32(layer.dense.W_in.bias),
                 Wout=torch.cat([pack_split(layer.dense.W_out.weight, H * q, H) for q in range(4)]),
                 bout=_f32(layer.dense.W_out.bias), scale=float(layer.scale))
        layer._packed = c
    hV = _f32(h_V).view(N, H)
    X = torch.empty(N * K, H, device=dev, dtype=torch.float32)           # per-edge W1r h_ESV
    call("mpnn_linear_rows", ptr(_f32(h_ESV).view(N * K, C)), C, C, ptr(c["Wr"]), None, N * K, ptr(X), st)
    hE = edges_from_natural(X.view(N, K, H), N, K)
    Kp = 16 * ((K + 15) // 16)
    idx32 = torch.arange(L, device=dev, dtype=torch.int32).view(1, L, 1).expand(B, L, Kp).contiguous()  # rows unused
    ones = torch.ones(N, device=dev, dtype=torch.float32)
    m_att = _f32(mask_attend) if mask_attend is not None else None
    m_V = _f32(mask_V).reshape(N) if mask_V is not None else ones
    PQ = torch.empty(N, 2 * H, device=dev, dtype=torch.float32)
    PQ[:, H:] = 0.0                                                      # the "gathered row" is zero here
    P = torch.empty(N, H, device=dev, dtype=torch.float32)
    _node_update(hV, N, None, None, None, 0.0, None, c["g1"], None, P)
    PQ[:, :H] = P
    gsum = torch.empty(N, H, device=dev, dtype=torch.float32)
    cnt = torch.empty(N, device=dev, dtype=torch.float32)
    call("mpnn_enc_message", ptr(hE), ptr(PQ), 2 * H, 0, H, ptr(idx32), ptr(ones), ptr(m_att), ptr(c["eye"]),
         ptr(c["W2"]), ptr(c["b2"]), B, L, K, ptr(gsum), ptr(cnt), st)
    hVn = torch.empty_like(hV)
    _node_update(hV, N, c, gsum, cnt, 0.0, m_V, (None, None, 0), hVn, None)
    return hVn.view(B, L, H)
