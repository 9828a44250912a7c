"""ctypes binding of libmpnn_hip.so (C ABI in include/mpnn_hip.h).

The library is built in-tree by `make` / `__graft_entry__.build()`.  There is no
CPU fallback: if the shared object is missing, or a call fails, a RuntimeError
is raised.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch  # noqa: F401  (must be imported first: the library binds to the HIP runtime torch loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPNN_HIP_LIB") or os.path.join(_HERE, "libmpnn_hip.so")  # override: kernel A/B builds

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float

_U64 = ctypes.c_uint64


class LevelLayer(ctypes.Structure):
    """mpnn_level_layer (include/mpnn_hip.h): one decoder layer of mpnn_decode_run."""
    _fields_ = [("P", _P), ("ldp", _I), ("offP", _I), ("RowA", _P), ("lda", _I), ("offA", _I), ("offEnc", _I),
                ("W1b_packed", _P), ("W2_packed", _P), ("b2", _P), ("hV_in", _P),
                ("W3_packed", _P), ("b3", _P), ("ln1_w", _P), ("ln1_b", _P), ("Win_packed", _P), ("b_in", _P),
                ("Wout8_packed", _P), ("b_out", _P), ("ln2_w", _P), ("ln2_b", _P), ("scale", _F),
                ("Wproj_packed", _P), ("bproj", _P), ("nproj", _I), ("hV_out", _P), ("PQ", _P)]


# name -> argtypes (all return int except where noted)
SIGNATURES = {
    "mpnn_pack_linear": [_P, _I, _I, _I, _I, _P, _P],
    "mpnn_pack_linear_split": [_P, _I, _I, _I, _I, _I, _P, _P],
    "mpnn_pos_table": [_P, _P, _I, _I, _P, _P],
    "mpnn_backbone_atoms": [_P, _P, _P, _I, _I, _P, _P, _P],
    "mpnn_knn_topk": [_P, _I, _I, _I, _F, _P, _P, _P, _P, _P],
    "mpnn_occupancy": [_P, _P, _I, _I, _F, _P, _P],
    "mpnn_edge_features": [_P] * 11 + [_I, _I, _I, _I, _P, _P, _P],
    "mpnn_edge_features_biased": [_P] * 12 + [_I, _I, _I, _I, _P, _P, _P],
    "mpnn_edges_to_natural": [_P, _I, _I, _P, _P],
    "mpnn_edges_from_natural": [_P, _I, _I, _P, _P],
    "mpnn_order_rank": [_P, _I, _I, _P, _P],
    "mpnn_order_masks": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "mpnn_enc_message": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "mpnn_enc_edge_update": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "mpnn_dec_message": [_P, _P, _I, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P],
    "mpnn_dec_message_valid": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P],
    "mpnn_dec_message_shared": [_P, _P, _I, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "mpnn_decode_run": [_P, _P, _I, _P, _P, _P, _P, _P, ctypes.POINTER(_I), _I, _I, _I, _I, _F,
                        ctypes.POINTER(LevelLayer), _I, _P, _P, _P, _I, _F, _U64, _P, _P, _P, _P, _P, _I, _I, _P],
    "mpnn_node_update": [_P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _P, _P, _I,
                         _P, _I, _I, _P, _I, _I, _P, _P, _P],
    "mpnn_occupancy_mlp": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P],
    "mpnn_logits_log_softmax": [_P, _P, _P, _I, _I, _P, _P],
    # training step (forward with saved pre-activations, backward, weight gradients)
    "mpnn_message_train": [_I] + [_P] * 11 + [_I] * 3 + [_P] * 4 + [_I, _P],
    "mpnn_enc_edge_update_train": [_P] * 11 + [_I] * 3 + [_P] * 4 + [_F, ctypes.c_uint32, _P, _P],
    "mpnn_message_bwd": [_I] + [_P] * 8 + [_I] * 3 + [_P] * 4 + [_I, _P, _I, _P],
    "mpnn_message_bwd_fused": [_I] + [_P] * 9 + [_I] * 3 + [_P] * 3 + [_I, _I] + [_P] * 5 + [_I, _P],
    "mpnn_enc_edge_update_bwd_fused": [_P] * 8 + [_I] * 3 + [_P] * 12 + [_F, ctypes.c_uint32, _P, _I, _P],
    "mpnn_enc_edge_update_bwd": [_P] * 8 + [_I] * 3 + [_P] * 7 + [_F, ctypes.c_uint32, _P, _I, _P],
    "mpnn_edge_wgrad": [_P, _I, _P, _P, _I, _I, _I, _P, _P, _P, _I, _P],
    "mpnn_gather_rows": [_P] * 6 + [_I, _I] + [_P] * 4,
    "mpnn_inverse_lists": [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "mpnn_inverse_sel": [_P, _P, _P, _I, _I, _P, _P],
    "mpnn_edge_features_train": [_P] * 12 + [_I, _I, _I, _I, _P, _P, _P, _P],
    "mpnn_edge_features_bwd": [_P] * 4 + [_I] * 3 + [_P] * 4 + [_I, _P],
    "mpnn_feature_chunks": [_P] * 5 + [_I] * 4 + [_P, _P],
    # autoregressive sampling
    "mpnn_decode_levels": [_P, _P, _P, _I, _I, _I, _I, _P, _P],
    "mpnn_dec_message_subset": [_P, _P, _I, _I, _P, _I, _I, _P, _I, _I] + [_P] * 8 + [_I] * 4 + [_P, _P],
    "mpnn_dec_layer_subset": [_P, _P, _I, _I, _P, _I, _I, _P, _I, _I] + [_P] * 8 + [_I] * 4 + [_P, _F] + [_P] * 10 +
                             [_F, _P, _P, _I, _P, _P, _P],
    "mpnn_dec_layer_sample_subset": [_P, _P, _I, _I, _P, _I, _I, _P, _I, _I] + [_P] * 8 + [_I] * 4 + [_P, _F] + [_P] * 10 +
                                    [_F, _P] + [_P, _P, _P, _I, _F, ctypes.c_uint64] + [_P] * 5 + [_I, _I, _P, _P, _P, _P],
    "mpnn_node_update_subset": [_P, _P, _I, _F, _P] + [_P] * 10 + [_F, _P, _P, _I, _P, _I, _P, _P, _P],
    "mpnn_sample_subset": [_P, _P, _P, _P, _I, _P, _I, _F, ctypes.c_uint64, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "mpnn_node_update_train": [_P, _P, _P, _F] + [_P] * 11 + [_F, _P, _P, _I, _I] + [_P] * 5 + [_F, ctypes.c_uint32, _P],
    "mpnn_node_update_bwd": [_P] * 6 + [_F] + [_P] * 4 + [_F, _P, _I, _P, _P, _P, _I] + [_P] * 8 +
                            [_F, ctypes.c_uint32, _P, _I, _P],
    "mpnn_node_update_bwd_each": [_P] * 6 + [_F] + [_P] * 4 + [_F, _P, _I, _P, _P, _P, _I] + [_P] * 8 +
                                 [_F, ctypes.c_uint32, _P, _I, _P],
    "mpnn_node_wgrad_batch": [_I] + [_P] * 9 + [_I, _P, _I, _P],
    "mpnn_pack_blocks": [_P, _P, _P, _I, _P, _P],
    "mpnn_feature_wgrad": [_P] * 7 + [_I] * 4 + [_P, _P, _P, _I, _P],
    "mpnn_seq_table": [_P, _P, _I, _I, _P, _P],
    "mpnn_add_table_rows": [_P, _P, _P, _I, _I, _P, _P],
    "mpnn_seq_table_bwd": [_P, _I, _P, _P, _I, _I, _P, _P, _P],
    "mpnn_logits_log_softmax_bwd": [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P],
    "mpnn_loss_smoothed": [_P, _P, _P, _I, _I, _F, _F, _P, _P, _P, _P, _I, _P],
    "mpnn_assemble_batch": [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "mpnn_assemble_batch_ordered": [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "mpnn_mask_self": [_P, _I, _I, _P, _P],
    "mpnn_node_wgrad": [_P, _I, _I, _P, _I, _I, _I, _P, _I, _P, _P, _I, _P],
    "mpnn_gather_indexed": [_P, _P, ctypes.c_longlong, _I, ctypes.c_longlong, _I, _P, _I, _I, _P],
    "mpnn_positional_encodings": [_P, _P, ctypes.c_longlong, _I, _P, _I, _P, _P],
    "mpnn_grad_sumsq": [_I, _P, _P, _P, _P, _I, _P],
    "mpnn_grad_scale": [_I, _P, _P, _P, _F, _P],
    "mpnn_adamw_step": [_I, _P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _I, _P, _F, _P],
    "mpnn_linear_rows": [_P, _I, _I, _P, _P, ctypes.c_longlong, _P, _P],
    "mpnn_probe_elementwise": [_P, _I, _I, _P, _P],
    "mpnn_range_flag": [_P, ctypes.c_longlong, _F, _P, _I, _P],
    "mpnn_node_wgrad_noalloc": [_P, _I, _I, _P, _I, _I, _I, _P, _I, _P, _P],
}

# reduced-precision twins (one f16 product per fp32 product) of the forward entry points: same signatures
F16_ENTRIES = ("mpnn_edge_features", "mpnn_enc_message", "mpnn_enc_edge_update", "mpnn_dec_message", "mpnn_node_update",
               "mpnn_dec_message_subset", "mpnn_node_update_subset", "mpnn_dec_message_shared", "mpnn_dec_layer_subset", "mpnn_dec_layer_sample_subset",
               "mpnn_linear_rows", "mpnn_edge_features_biased", "mpnn_dec_message_valid",
               "mpnn_decode_run")
for _n in F16_ENTRIES:
    SIGNATURES[_n + "_f16"] = SIGNATURES[_n]

_lib = None


class _Mode(threading.local):
    """Precision mode of the CALLING THREAD: ctypes releases the GIL around library calls, so two models with different
    precisions on two threads interleave; a module-global mode would let one retarget the other's entry points."""
    suffix = ""


_mode = _Mode()


def set_precision(mode: str):
    """'fp32' (default: 3 f16 products per fp32 product, fp32-class) or 'f16' (1 product) for the forward kernels the
    calling thread launches from here on.  -> the previous mode (restore it with another call)."""
    if mode not in ("fp32", "f16"):
        raise ValueError("precision must be 'fp32' or 'f16'")
    prev = "f16" if _mode.suffix else "fp32"
    _mode.suffix = "_f16" if mode == "f16" else ""
    return prev


def entry(name: str) -> str:
    """Entry-point name under the calling thread's precision mode."""
    return name + _mode.suffix if _mode.suffix and name in F16_ENTRIES else name


def load() -> ctypes.CDLL:
    """Load (once) and return the library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP extension first (`make` at the repo root or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). mpnn_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _I
    lib.mpnn_version.restype = _I
    lib.mpnn_version.argtypes = []
    lib.mpnn_workspace_bytes.restype = ctypes.c_longlong
    lib.mpnn_workspace_bytes.argtypes = [ctypes.c_char_p, _I, _I]
    lib.mpnn_last_error.restype = ctypes.c_char_p
    lib.mpnn_last_error.argtypes = []
    lib.mpnn_ablation_build.restype = _I
    lib.mpnn_ablation_build.argtypes = []
    if lib.mpnn_ablation_build() and os.environ.get("MPNN_ALLOW_ABLATION") != "1":
        raise RuntimeError(f"{LIB_PATH} was built with a timing-only ablation switch (csrc/mpnn_ablation.h): its results "
                           "are wrong by design.  Rebuild with plain `make`, or set MPNN_ALLOW_ABLATION=1 for an A/B "
                           "timing run")
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must be contiguous."""
    if t is None:
        return None
    assert t.is_contiguous(), "mpnn_amd kernels need contiguous tensors"
    return t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


# optional per-entry-point device timing (bench.py): name -> list of (start_event, end_event),
# recorded on the stream the kernels are launched on
_profile = None


def set_profiler(store):
    """store: dict to fill with name -> [(start, end) torch.cuda.Event pairs], or None to switch off."""
    global _profile
    _profile = store


def call(name: str, *args):
    lib = load()
    name = entry(name)
    if _profile is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args)
        e1.record()
        _profile.setdefault(name, []).append((e0, e1))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.mpnn_last_error().decode(errors="replace")
        raise RuntimeError(f"{name} failed (code {rc}): {msg}")
