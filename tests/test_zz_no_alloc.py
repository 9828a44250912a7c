"""The allocation-free entry point mpnn_node_wgrad_noalloc (built-in scratch array) against mpnn_node_wgrad (caller
workspace) through the C ABI: two calls in a row, the second on a smaller input; an input beyond the built-in scratch
is rejected.  Both entry points run the same kernels (k_node_wgrad_split: three split-f16 MFMA products per fp32
product, each 32-row batch of dY scaled by an exact power of two, fp32 accumulation, partial sums added in a fixed
order) on the same row partition, so their results are held to |d| <= 1e-6 * max|dW| of each other (they are in fact
bit-identical) and to 1e-5 relative L2 of an fp64 matmul (the split products' 2^-21 per product).
The first call on a device resolves the scratch symbol (hipGetSymbolAddress) and must happen outside graph capture;
operands must stay below the f16 range (|x| < 65504) like every split-product operand of this library."""
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
    # no allocation of ANY kind: torch's caching allocator sees only its own (memory_allocated), a hipMalloc inside the
    # library would show in the device's free memory (mem_get_info = hipMemGetInfo)
    torch.cuda.synchronize()
    before, free_before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    for _ in range(3):
        rc = lib.mpnn_node_wgrad_noalloc(ptr(dY), 2048, 128, ptr(X), 2048, 128, 4, ptr(dW), 2048, None, stream_ptr())
        assert rc == 0
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.cuda.mem_get_info()[0] >= free_before, "the allocation-free entry point took device memory"
