"""Host-side pieces of the training path that need no GPU: the row-slab contraction (both precisions), the partial-vector unpacking."""
import torch

from neuraloc_amd import train


def test_contract_row_slabs_match_the_plain_product_in_both_precisions():
    g = torch.Generator().manual_seed(0)
    for dt, tol in ((torch.float64, 1e-12), (torch.float32, 2e-4)):
        for K in (4096 * 3, 4099):                      # a row count with a slab divisor / without one
            X = torch.randn(K, 24, generator=g).to(dt)
            Y = torch.randn(K, 9, generator=g).to(dt)
            if dt == torch.float32 and K == 4099:
                continue                                # (the small-output fp32 form is a HIP kernel: GPU tests)
            want = X.double().t() @ Y.double()
            got = train._contract(X, Y)
            assert got.dtype == dt and float((got.double() - want).abs().max()) <= tol * float(want.abs().max())
            acc = torch.ones(24, 9, dtype=dt)
            got2 = train._contract(X, Y, acc)
            assert float((got2.double() - want - 1.0).abs().max()) <= tol * float(want.abs().max())


def test_unpack_partials_layout():
    class _Net:
        A = torch.eye(3, 5)
    m, D1 = 4, 5
    P = m * D1 + m + m * m + m + m + D1 + 1 + D1 * D1
    part = torch.arange(2 * P, dtype=torch.float32).reshape(2, P)
    grads = train._unpack_partials(part, m, D1, _Net())
    gv = part.sum(0)
    assert torch.equal(grads["N.layers.0.weight"], gv[:m * D1].reshape(m, D1))
    assert torch.equal(grads["N.layers.0.bias"], gv[m * D1:m * D1 + m])
    o = m * D1 + m
    assert torch.equal(grads["N.layers.1.weight"], gv[o:o + m * m].reshape(m, m))
    o += m * m + m + m
    assert torch.equal(grads["c.weight"], gv[o:o + D1].reshape(1, D1))
    dM = gv[o + D1 + 1:o + D1 + 1 + D1 * D1].reshape(D1, D1)
    assert torch.allclose(grads["A"], _Net.A @ (dM + dM.t()))
