"""The allocation-free entry point mpnn_node_wgrad_noalloc (built-in scratch array) against mpnn_node_wgrad (caller
workspace) through the C ABI: two calls in a row, the second on a smaller input; an input beyond the built-in scratch
is rejected.  Bar (fp32 sums of fp32 products; the two entry points run the same kernels on the same partition, only
the order of four atomic adds per element may differ): |d| <= 1e-6 * max|dW|, and 1e-5 relative L2 to an fp64 matmul."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(lib, n_rows, out_f, in_f, seed, ldy=None, ldx=None, bias=True):
    from mpnn_amd._lib import ptr, stream_ptr

    g = torch.Generator().manual_seed(seed)
    ldy, ldx = ldy or out_f, ldx or in_f
    dY = torch.randn(n_rows, ldy, generator=g).cuda()
    X = torch.randn(n_rows, ldx, generator=g).cuda()
    st = stream_ptr()
    res = []
    parts = 128 * 16512  # the size of the built-in scratch: both entry points then split the rows the same way
    ws = torch.empty(parts, device="cuda")
    for name in ("mpnn_node_wgrad", "mpnn_node_wgrad_noalloc"):
        dW = torch.zeros(out_f, in_f, device="cuda")
        db = torch.zeros(out_f, device="cuda") if bias else None
        args = [ptr(dY), ldy, out_f, ptr(X), ldx, in_f, n_rows, ptr(dW), in_f, ptr(db)]
        if name == "mpnn_node_wgrad":
            args += [ptr(ws), parts]
        rc = getattr(lib, name)(*args, st)
        assert rc == 0, lib.mpnn_last_error().decode()
        res.append((dW, db))
    torch.cuda.synchronize()
    ref = dY[:, :out_f].double().t() @ X[:, :in_f].double()
    (dW0, db0), (dW1, db1) = res
    scale = float(dW0.abs().max())
    assert float((dW1 - dW0).abs().max()) <= 1e-6 * scale, float((dW1 - dW0).abs().max())
    assert float((dW1.double() - ref).norm() / ref.norm()) <= 1e-5
    if bias:
        assert float((db1 - db0).abs().max()) <= 1e-6 * float(db0.abs().max())
        assert float((db1.double() - dY[:, :out_f].double().sum(0)).norm()) <= 1e-5 * float(dY.double().sum(0).norm())


def test_noalloc_matches_workspace_entry_point_twice_in_a_row():
    from mpnn_amd import _lib

    lib = _lib.load()
    _both(lib, n_rows=10240, out_f=512, in_f=128, seed=1)             # W_in of the FFN at a 10k-residue step
    _both(lib, n_rows=777, out_f=21, in_f=128, seed=2, ldy=24)        # smaller, ragged: stale partials must not leak
    _both(lib, n_rows=5, out_f=128, in_f=512, seed=3, ldx=520, bias=False)


def test_noalloc_rejects_sizes_beyond_its_scratch():
    from mpnn_amd import _lib
    from mpnn_amd._lib import ptr, stream_ptr

    lib = _lib.load()
    dY = torch.zeros(4, 2048, device="cuda")
    X = torch.zeros(4, 2048, device="cuda")
    dW = torch.zeros(2048, 2048, device="cuda")
    rc = lib.mpnn_node_wgrad_noalloc(ptr(dY), 2048, 2048, ptr(X), 2048, 2048, 4, ptr(dW), 2048, None, stream_ptr())
    assert rc == -1 and b"scratch" in lib.mpnn_last_error()
    before = torch.cuda.memory_allocated()
    rc = lib.mpnn_node_wgrad_noalloc(ptr(dY), 2048, 128, ptr(X), 2048, 128, 4, ptr(dW), 2048, None, stream_ptr())
    assert rc == 0 and torch.cuda.memory_allocated() == before
    torch.cuda.synchronize()
