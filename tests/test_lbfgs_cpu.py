"""Host logic of pde_amd.lbfgs.LBFGS against torch.optim.LBFGS on the CPU (the device launches are
replaced by tests/lbfgs_cpu_kernels.py).  Same algorithm => same trajectory up to rounding: the two
optimizers are run on identical problems and compared iterate by iterate."""
import numpy as np
import pytest
import torch

from lbfgs_cpu_kernels import CpuLBFGSKernels
from pde_amd.lbfgs import LBFGS, _cubic_min


def _problem(kind, dtype):
    torch.manual_seed(3)
    if kind == "rosenbrock":
        p = [torch.nn.Parameter(torch.tensor([-1.2, 1.0, 0.7, -0.3], dtype=dtype)),
             torch.nn.Parameter(torch.tensor([[0.5, -0.5], [1.5, 0.2]], dtype=dtype))]

        def f():
            v = torch.cat([q.reshape(-1) for q in p])
            return (100.0 * (v[1:] - v[:-1] ** 2) ** 2 + (1.0 - v[:-1]) ** 2).sum()
    else:  # small tanh regression: smooth, non-quadratic, like the PINN data terms
        X = torch.randn(64, 3, dtype=dtype)
        y = torch.sin(X.sum(1, keepdim=True))
        lin1, lin2 = torch.nn.Linear(3, 8).to(dtype), torch.nn.Linear(8, 1).to(dtype)
        p = list(lin1.parameters()) + list(lin2.parameters())

        def f():
            return torch.mean((lin2(torch.tanh(lin1(X))) - y) ** 2)
    return p, f


def _run(opt_cls, kind, dtype, max_iter, history=7, **kw):
    p, f = _problem(kind, dtype)
    trace = []
    if opt_cls is LBFGS:
        n = sum(q.numel() for q in p)
        opt = LBFGS(p, lr=1.0, max_iter=max_iter, max_eval=10 * max_iter, history_size=history,
                    tolerance_grad=1e-12, tolerance_change=1e-15, line_search_fn="strong_wolfe",
                    kernels=CpuLBFGSKernels(n, history, dtype), **kw)
        # LBFGS keeps a float32 flat buffer; for the float64 comparison re-point it
        if dtype == torch.float64:
            pytest.skip("flat buffers are float32 by design")
    else:
        opt = torch.optim.LBFGS(p, lr=1.0, max_iter=max_iter, max_eval=10 * max_iter, history_size=history,
                                tolerance_grad=1e-12, tolerance_change=1e-15, line_search_fn="strong_wolfe")

    def closure():
        opt.zero_grad()
        loss = f()
        loss.backward()
        trace.append(float(loss.detach()))
        return loss
    opt.step(closure)
    return trace, torch.cat([q.detach().reshape(-1) for q in p]).double().numpy(), opt


@pytest.mark.parametrize("kind", ["rosenbrock", "tanhnet"])
def test_same_trajectory_as_torch_lbfgs(kind):
    n_it = 12
    tr_a, x_a, opt_a = _run(LBFGS, kind, torch.float32, n_it)
    tr_b, x_b, opt_b = _run(torch.optim.LBFGS, kind, torch.float32, n_it)
    # identical evaluation count (same line-search decisions) and matching losses at every evaluation
    assert len(tr_a) == len(tr_b), (len(tr_a), len(tr_b))
    assert np.allclose(tr_a, tr_b, rtol=2e-3, atol=1e-6), np.max(np.abs(np.array(tr_a) - np.array(tr_b)))
    assert np.allclose(x_a, x_b, rtol=5e-3, atol=5e-4)
    assert tr_a[-1] < 0.2 * tr_a[0]
    sa, sb = opt_a.state[opt_a._params[0]], opt_b.state[opt_b._params[0]]
    assert sa["func_evals"] == sb["func_evals"] and sa["n_iter"] == sb["n_iter"]
    # one blocking read per closure evaluation + one for the FIRST direction: from the second iteration on the
    # direction's statistics come back with the line search's first evaluation (LBFGS.speculate)
    assert opt_a.host_syncs == len(tr_a) + 1
    # the plain order (two reads per iteration) makes the same evaluations
    LBFGS.speculate = False
    try:
        tr_c, x_c, opt_c = _run(LBFGS, kind, torch.float32, n_it)
    finally:
        LBFGS.speculate = True
    assert tr_c == tr_a and np.array_equal(x_c, x_a)
    assert opt_c.host_syncs == len(tr_c) + sa["n_iter"]


def test_history_ring_drops_oldest_pair():
    tr_a, x_a, _ = _run(LBFGS, "tanhnet", torch.float32, 25, history=3)
    tr_b, x_b, _ = _run(torch.optim.LBFGS, "tanhnet", torch.float32, 25, history=3)
    # float32 rounding differences grow along an L-BFGS run (measured here: 1e-7 after 5 evaluations,
    # 1e-4 after 15, 1e-2 after 25), so the late part of the run is compared loosely
    assert len(tr_a) == len(tr_b)
    assert np.allclose(tr_a[:16], tr_b[:16], rtol=1e-3, atol=1e-6)
    assert np.allclose(tr_a, tr_b, rtol=0.2)


def test_flat_views_and_zero_grad():
    p, f = _problem("tanhnet", torch.float32)
    n = sum(q.numel() for q in p)
    opt = LBFGS(p, max_iter=2, line_search_fn="strong_wolfe", history_size=4, kernels=CpuLBFGSKernels(n, 4))
    base = opt._x.data_ptr()
    off = 0
    for q in p:
        assert q.data_ptr() == base + 4 * off and q.grad.data_ptr() == opt._g.data_ptr() + 4 * off
        off += q.numel()
    f().backward()
    assert float(opt._g.abs().sum()) > 0
    opt.zero_grad()
    assert float(opt._g.abs().sum()) == 0 and all(q.grad is not None for q in p)
    # a closure that drops .grad (torch's default zero_grad) is still gathered correctly
    for q in p:
        q.grad = None
    f().backward()
    want = torch.cat([q.grad.reshape(-1) for q in p]).clone()
    assert torch.equal(opt._gather(), want)
    with pytest.raises(ValueError):
        LBFGS(p, flat_params=torch.zeros(n), kernels=CpuLBFGSKernels(n, 4))


def test_stops_on_gradient_tolerance_and_max_eval():
    p = [torch.nn.Parameter(torch.zeros(3))]
    opt = LBFGS(p, tolerance_grad=1e-7, line_search_fn="strong_wolfe", kernels=CpuLBFGSKernels(3, 5), history_size=5)
    calls = []

    def closure():
        opt.zero_grad()
        loss = (p[0] ** 2).sum()
        loss.backward()
        calls.append(1)
        return loss
    opt.step(closure)
    assert len(calls) == 1                      # gradient already below tolerance: no iteration
    p2, f2 = _problem("rosenbrock", torch.float32)
    n = sum(q.numel() for q in p2)
    opt2 = LBFGS(p2, max_iter=1000, max_eval=9, line_search_fn="strong_wolfe", history_size=5,
                 kernels=CpuLBFGSKernels(n, 5))
    ev = []

    def c2():
        opt2.zero_grad()
        loss = f2()
        loss.backward()
        ev.append(1)
        return loss
    opt2.step(c2)
    assert 9 <= len(ev) <= 9 + 25               # stops at the first iteration boundary past max_eval


def test_cubic_min():
    # f(x) = (x-1)^2 through x=0 and x=3: exact minimiser of the interpolating cubic is 1
    assert abs(_cubic_min(0.0, 1.0, -2.0, 3.0, 4.0, 4.0) - 1.0) < 1e-12
    assert _cubic_min(0.0, 1.0, -2.0, 3.0, 4.0, 4.0, bounds=(1.5, 2.0)) == 1.5
    # no real minimiser -> midpoint
    assert _cubic_min(0.0, 0.0, 1.0, 1.0, 1.0, 2.0) == 0.5
    # degenerate denominator (tensor arithmetic would give inf/nan there) -> midpoint as well
    assert _cubic_min(0.0, 0.0, 1.0, 1.0, 1.0, 1.0) == 0.5


def test_train_uses_device_lbfgs_and_matches_torch_lbfgs():
    """PhysicsInformedNN.train(): the L-BFGS phase through pde_amd.lbfgs.LBFGS (CPU stand-in kernels,
    oracle engine) follows torch.optim.LBFGS on the same model."""
    import os
    from test_host_logic import GOLDEN, feed, make_model
    from pde_amd import scripts
    g = np.load(os.path.join(GOLDEN, "burgers_9x20_xavier.npz"))
    out = {}
    for impl in ("device", "torch"):
        m = make_model(g, scripts.burgers.PI.PhysicsInformedNN)
        type(m).lbfgs_impl = impl
        type(m).lbfgs_kernels_factory = staticmethod(lambda n, h: CpuLBFGSKernels(n, h))
        feed(m, g, "burgers")
        m.train(1, LBFGS_max_iter=6)
        out[impl] = (m.iter, m.dnn.flat_theta().clone(), type(m.optimizer_LBFGS).__module__)
    assert out["device"][2] == "pde_amd.lbfgs" and out["torch"][2].startswith("torch.optim")
    assert out["device"][0] == out["torch"][0] > 2            # same number of closure evaluations
    assert torch.allclose(out["device"][1], out["torch"][1], rtol=1e-3, atol=1e-4)
