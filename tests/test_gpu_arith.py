"""Unit tests of the device arithmetic every MLP on the path is built from, through the C ABI
(mpnn_probe_elementwise): the exact-erf GELU (reference nn.GELU(), model_utils.py:70,115,149) against the fp64 erf
form, and the f16 hi/lo operand split that stands in for fp32 products on the matrix pipe."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def probe(x, mode):
    from mpnn_amd._lib import call, ptr, stream_ptr

    x = x.contiguous().cuda()
    y = torch.empty_like(x)
    call("mpnn_probe_elementwise", ptr(x), x.numel(), mode, ptr(y), stream_ptr())
    torch.cuda.synchronize()
    return y.cpu()


def test_gelu_within_3e7_of_the_fp64_erf_form():
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.linspace(-14, 14, 2_000_003), torch.linspace(-1e-3, 1e-3, 100_001),
                   torch.randn(1_000_000, generator=g) * 3.0,
                   torch.tensor([0.0, -0.0, 1e-30, -1e-30, 12.0, -12.0, 12.5, -40.0, 40.0, 1e4, -1e4, 6e4])])
    y = probe(x, 0).double()
    xd = x.double()
    ref = 0.5 * xd * (1.0 + torch.erf(xd / math.sqrt(2.0)))
    err = (y - ref).abs()
    bound = 3e-7 + 6e-8 * ref.abs()      # 3e-7 absolute; half an ulp of the fp32 result beyond |x| ~ 4
    worst = int(torch.argmax(err - bound))
    assert bool((err <= bound).all()), f"gelu({float(x[worst])}) = {float(y[worst])}, want {float(ref[worst])}"
    small = xd.abs() <= 6.0
    print(f"max |gelu - ref| on |x| <= 6: {float(err[small].max()):.3e}; overall {float(err.max()):.3e}")


def test_f16_split_keeps_22_bits():
    g = torch.Generator().manual_seed(1)
    mant = torch.rand(2_000_000, generator=g) + 1.0
    expo = torch.randint(-13, 15, (2_000_000,), generator=g).float()
    sign = torch.randint(0, 2, (2_000_000,), generator=g).float() * 2 - 1
    x = sign * mant * torch.exp2(expo)                      # the f16 normal range: 2^-13 .. 2^15
    y = probe(x, 1)
    # hi carries 11 bits, lo the next 11 - as long as lo is a NORMAL f16 (|x| >= 2^-3); below that lo runs into the f16
    # subnormals and the resolution becomes absolute: half a subnormal ulp = 2^-25  (mpnn_common.h)
    err = (y.double() - x.double()).abs()
    bound = torch.maximum(x.double().abs() * 2.0 ** -21, torch.tensor(2.0 ** -25, dtype=torch.float64))
    assert bool((err <= bound).all()), float((err / bound).max())
    big = x.abs() >= 0.125
    assert float((err[big] / x.double().abs()[big]).max()) <= 2.0 ** -21
    # exactly representable inputs (11 significant bits) come back exactly, zero stays zero
    e = torch.tensor([0.0, 1.0, -2.5, 1024.0, 65504.0, -65504.0, 2.0 ** -14, 3.0 * 2.0 ** -24])
    assert torch.equal(probe(e, 1), e)
    # below the f16 normal range the split keeps 2^-24 ABSOLUTE resolution (documented in mpnn_common.h)
    t = torch.rand(100_000, generator=g) * 2.0 ** -14
    assert float((probe(t, 1) - t).abs().max()) <= 2.0 ** -25 + 1e-12


def test_f16_mode_rounding_is_nearest_even_and_overflows_to_inf():
    """The *_f16 entry points' single operand rounding (mode 2: gfx950's v_cvt_pk_f16_f32): equal to torch's fp32 ->
    fp16 cast (round to nearest even), including +-inf from |x| >= 65520 on - an out-of-range operand poisons the
    result instead of saturating at 65504."""
    g = torch.Generator().manual_seed(2)
    x = torch.cat([torch.randn(1_000_000, generator=g) * 100.0, torch.randn(100_000, generator=g) * 1e-4,
                   torch.tensor([65503.9, 65504.0, 65519.9, 65520.0, 7e4, -65519.9, -65520.0, -1e9, 0.0, -0.0,
                                 2049.0, 2051.0, 1e-8])])
    if x.numel() % 8:
        x = torch.cat([x, torch.zeros(8 - x.numel() % 8)])
    y = probe(x, 2)
    want = x.to(torch.float16).float()
    assert torch.equal(y, want), (y[-16:], want[-16:])
    assert torch.isinf(y[x.abs() >= 65520.0]).all() and torch.isfinite(y[x.abs() < 65520.0]).all()
