"""GPU tests of the device L-BFGS (csrc/lbfgs.hip + pde_amd/lbfgs.py): the direction
kernel against a float64 restatement of the two-loop recursion (tests/lbfgs_cpu_kernels.py), and a
training run against torch.optim.LBFGS on the same model."""
import numpy as np
import pytest
import torch

from lbfgs_cpu_kernels import CpuLBFGSKernels

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


# history 50: one row per lane, three whole images in LDS; 70: five rows per lane, whole images; 100: one panel per loop;
# 150: two panels per loop
@pytest.mark.parametrize("n,hist", [(7851, 5), (30802, 50), (37, 3), (1024, 1), (12737, 150), (2000, 70), (3001, 100)])
def test_direction_kernel_matches_float64_two_loop(n, hist):
    """A synthetic quadratic-like sequence (g_{k+1} = g_k + A-ish * step) drives both implementations
    through more steps than the history holds (ring wrap) and through rejected pairs (y.s <= 1e-10).
    Tolerance: float32 dot products over n terms, amplified by up to 2*hist dependent steps -> 2e-4."""
    from pde_amd._native import NativeLBFGSKernels
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(n + hist)
    diag = torch.tensor(rng.uniform(0.5, 4.0, n))
    k_dev = NativeLBFGSKernels(n, hist, dev)
    k_ref = CpuLBFGSKernels(n, hist, torch.float64)
    g = torch.tensor(rng.standard_normal(n))
    d_dev = torch.zeros(n, dtype=torch.float32, device=dev)
    d_ref = torch.zeros(n, dtype=torch.float64)
    t = 1.0
    for it in range(2 * hist + 4):
        g32 = g.float().to(dev)
        st_dev = k_dev.direction(g32, d_dev, t, it == 0).cpu().double().numpy()
        st_ref = k_ref.direction(g, d_ref, t, it == 0).numpy()
        assert _rel(d_dev.cpu().double().numpy(), d_ref.numpy()) < 2e-4, it
        assert abs(st_dev[0] - st_ref[0]) <= 2e-4 * abs(st_ref[0]) + 1e-6        # g.d
        assert abs(st_dev[1] - st_ref[1]) <= 1e-5 * st_ref[1]                     # sum|g|
        assert abs(st_dev[2] - st_ref[2]) <= 2e-4 * st_ref[2]                     # max|d|
        assert abs(st_dev[3] - st_ref[3]) <= 1e-6 * st_ref[3]                     # max|g|
        assert st_dev[4] == st_ref[4]                                             # pairs held
        assert st_ref[0] < 0                                                      # a descent direction
        # keep both on the SAME iterate sequence (the reference's): the comparison is per step
        d_dev.copy_(d_ref.float())
        t = float(rng.uniform(0.3, 1.0))
        if it == hist + 1:
            g = g.clone()                  # zero step: y = 0 -> the pair must be rejected by both
            t = 0.0
        else:
            g = g + diag * (t * d_ref) + 1e-3 * torch.tensor(rng.standard_normal(n))
    out = k_dev.eval_stats(g.float().to(dev), d_dev, torch.tensor([1.25], device=dev)).cpu().numpy()
    assert out[0] == 1.25
    assert abs(out[1] - float(g.dot(d_ref))) <= 2e-4 * abs(float(g.dot(d_ref))) + 1e-6
    assert abs(out[2] - float(g.abs().max())) <= 1e-6 * float(g.abs().max())


def test_direction_advance_moves_the_weights_and_eval_stats_keeps_and_mirrors():
    """pinn_lbfgs_direction_advance with x: the same direction and statistics as without, x0 = x and x = x + t_next d;
    pinn_lbfgs_eval_stats_keep: the gradient copy, and the pinned-host mirror [stats8 | out4] that `fetch()` polls for
    equals the device buffers."""
    from pde_amd._native import NativeLBFGSKernels
    dev = torch.device("cuda", 0)
    n, hist = 7851, 6
    rng = np.random.default_rng(5)
    ka, kb = NativeLBFGSKernels(n, hist, dev), NativeLBFGSKernels(n, hist, dev)
    if ka.mirror_dev is None:       # (the library then reads the statistics with a copy + stream wait; never seen on this pool)
        pytest.skip("pinned host memory is not mapped into the device's address space on this box")
    diag = torch.tensor(rng.uniform(0.5, 4.0, n), dtype=torch.float32, device=dev)
    g = torch.tensor(rng.standard_normal(n), dtype=torch.float32, device=dev)
    da, db = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    x = torch.tensor(rng.standard_normal(n), dtype=torch.float32, device=dev)
    x0 = torch.full((n,), 7.0, device=dev)
    t = 1.0
    for it in range(2 * hist + 3):
        xb = x.clone()
        sa = ka.direction(g, da, t, it == 0, x=x, x0=x0, t_next=0.5).clone()
        sb = kb.direction(g, db, t, it == 0).clone()
        assert torch.equal(da, db) and torch.equal(sa, sb), it
        assert torch.equal(x0, xb), it
        want = xb.double() + 0.5 * da.double()
        assert float((x.double() - want).abs().max()) <= 1.2e-7 * float(want.abs().max()), it
        t = float(rng.uniform(0.3, 1.0))
        g = g + diag * (t * da) + 1e-3 * torch.tensor(rng.standard_normal(n), dtype=torch.float32, device=dev)
        keep = torch.empty_like(g)
        out = ka.eval_stats(g, da, torch.tensor([0.5 + it], device=dev), keep)
        stats8, out4 = ka.fetch()
        torch.cuda.synchronize()
        assert torch.equal(keep, g)
        assert out4 == out.cpu().tolist() and stats8 == ka.stats.cpu().tolist(), it
        assert out4[0] == 0.5 + it
        assert abs(out4[1] - float(g.double().dot(da.double()))) <= 2e-5 * float(g.double().abs().dot(da.double().abs()))


@pytest.mark.parametrize("P,n_terms,nf", [(7851, 2, 1), (30802, 4, 2), (37, 5, 3), (301, 1, 0)])
def test_closure_sum_matches_the_torch_expressions(P, n_terms, nf):
    """pinn_closure_sum = what the closure did with a divide and four adds: loss = loss0 + loss_b + loss_f + ... in
    float32 and this order, loss_f = sum(sums) / N, grad = g_data + g_f (written or accumulated)."""
    from pde_amd._native import closure_sum
    dev = torch.device("cuda", 0)
    torch.manual_seed(P)
    terms = torch.rand(8, device=dev)
    sums = torch.rand(nf, device=dev) * 1000.0
    ga, gb = torch.randn(P, device=dev), torch.randn(P, device=dev)
    N = 20000.0
    tc, sc = terms.cpu().numpy(), sums.cpu().numpy()
    lf = np.float32(0.0)
    if nf:
        lf = sc[0]
        for c in range(1, nf):
            lf = np.float32(lf + sc[c])
        lf = np.float32(lf / np.float32(N))
    want = tc[0]
    if n_terms > 1:
        want = np.float32(want + tc[1])
    want = np.float32(want + lf)
    for k in range(2, n_terms):
        want = np.float32(want + tc[k])
    for accumulate, clear in ((False, False), (True, False), (False, True)):
        grad = torch.full((P,), 3.0, device=dev)
        gb_in, sums_in = gb.clone(), sums.clone()
        loss2 = closure_sum(terms, n_terms, sums_in, N, ga, gb_in, grad, accumulate, clear)
        torch.cuda.synchronize()
        assert loss2.cpu().numpy().tolist() == [float(want), float(lf)]
        ref = ga + gb
        if accumulate:
            ref = torch.full((P,), 3.0, device=dev) + ref
        assert torch.equal(grad, ref)
        if clear:                   # (the captured closure's one collocation buffer is left zeroed for the next replay)
            assert not gb_in.any() and not sums_in.any()
        else:
            assert torch.equal(gb_in, gb) and torch.equal(sums_in, sums)


def test_lbfgs_entry_points_reject_bad_arguments():
    from pde_amd._native import load_library
    lib = load_library()
    assert lib.pinn_lbfgs_workspace_floats(0, 5) < 0
    assert lib.pinn_lbfgs_workspace_floats(100, 0) < 0
    assert lib.pinn_lbfgs_workspace_floats(100, 257) < 0
    small, big = lib.pinn_lbfgs_workspace_floats(100, 4), lib.pinn_lbfgs_workspace_floats(30802, 50)
    assert small > 2 * 5 * 100 and big > 2 * 51 * 30802 and big < 4 * 51 * 30802
    assert lib.pinn_lbfgs_direction(None, 100, 4, None, None, 1.0, 1, None, None) == -4
    assert lib.pinn_lbfgs_eval_stats(None, None, 100, None, None, None) == -4
    assert lib.pinn_lbfgs_direction_advance(None, 100, 4, None, None, 1.0, 1, None, None, None, 1.0, None) == -4
    assert lib.pinn_lbfgs_eval_stats_keep(None, None, 100, None, None, None, None, None, 0, None) == -4
    assert lib.pinn_closure_sum(None, 2, None, 1, 1.0, None, None, None, None, 10, 0, None) == -4
    assert lib.pinn_host_device_pointer(None, None) == -4


def _burgers_model(impl):
    import os
    from conftest import GOLDEN
    from test_host_logic import feed
    from test_gpu_parity import _load_theta
    from pde_amd import scripts
    g = np.load(os.path.join(GOLDEN, "burgers_9x20_xavier.npz"))
    cls = type("B", (scripts.burgers.PI.PhysicsInformedNN,), {"lbfgs_impl": impl})
    m = cls(g["layers"].tolist(), g["lb"], g["ub"])
    _load_theta(m, g["theta"])
    feed(m, g, "burgers")
    return m


def test_training_with_device_lbfgs_follows_torch_lbfgs():
    """Same model, same data, L-BFGS phase only: the device implementation makes the same line-search
    decisions as torch.optim.LBFGS for the first iterations (same closure count) and the weights agree
    to 1e-3 (float32 rounding grows along an L-BFGS run; see tests/test_lbfgs_cpu.py)."""
    res = {}
    for impl in ("device", "torch"):
        m = _burgers_model(impl)
        m.train(0, LBFGS_max_iter=8)
        res[impl] = (m.iter, m.dnn.flat_theta().detach().cpu().double().numpy().copy(), m)
    assert type(res["device"][2].optimizer_LBFGS).__module__ == "pde_amd.lbfgs"
    assert res["device"][0] == res["torch"][0] > 8
    assert _rel(res["device"][1], res["torch"][1]) < 1e-3
    opt = res["device"][2].optimizer_LBFGS
    st = opt.state[opt._params[0]]
    # one blocking read per closure evaluation + one for the first direction (LBFGS.speculate: from the second
    # iteration on the direction's statistics come back with the line search's first evaluation)
    assert opt.host_syncs == st["func_evals"] + 1


def test_long_device_lbfgs_run_converges():
    m = _burgers_model("device")
    m.iter = 1
    l0 = m.backward().item()
    m.train(0, LBFGS_max_iter=150)
    m.optimizer_LBFGS.zero_grad()
    l1 = m.backward().item()
    assert l1 < 0.2 * l0, (l0, l1)


def test_graph_replayed_closure_matches_eager_closure():
    """use_hip_graph: the L-BFGS closure replayed from a captured HIP graph runs the same kernels on
    the same buffers as the eager closure -> identical evaluation count, weights equal to rounding."""
    res = {}
    for graph in (False, True):
        m = _burgers_model("device")
        m.use_hip_graph = graph
        m.train(0, LBFGS_max_iter=10)
        res[graph] = (m.iter, m.dnn.flat_theta().detach().cpu().double().numpy().copy())
        assert hasattr(m, "_lbfgs_graph") == graph
    assert res[True][0] == res[False][0]
    assert _rel(res[True][1], res[False][1]) < 1e-3


def test_device_sampled_collocation_points_train():
    """drivers.lhs_device on the GPU -> set_training_data(Xf=tensor) -> a few training steps."""
    from pde_amd import drivers, scripts
    dev = torch.device("cuda", 0)
    lb, ub = np.array([-10.0, -2.0]), np.array([10.0, 2.0])
    H = drivers.lhs_device(2, 100000, dev, generator=torch.Generator(device=dev).manual_seed(5))
    assert H.is_cuda and sorted(torch.floor(H[:, 0] * 100000).long().tolist()) == list(range(100000))
    Xf = torch.tensor(lb, device=dev) + torch.tensor(ub - lb, device=dev) * H
    m = scripts.kdv5_1soliton.PI.PhysicsInformedNN([2, 50, 50, 50, 50, 1], lb, ub)
    x0 = np.linspace(-10, 10, 50)[:, None]
    m.set_training_data(x0, 0.5 / np.cosh(0.5 * x0) ** 2, np.linspace(-2, 2, 50)[:, None], Xf)
    assert m.xf.shape == (100000, 1) and m.xf.is_cuda
    l0 = m.backward().item()
    for p in m.dnn.parameters():
        p.grad = None
    m.train(20, LBFGS_max_iter=0)
    assert m.backward().item() < l0


@pytest.mark.parametrize("width", [50, 100])
def test_graph_replay_reproduces_eager_gradient(width):
    """The captured L-BFGS closure replayed several times gives the eager closure's loss and gradient
    every time.  Width 100 accumulates its weight-gradient tiles into a per-workgroup slab with
    atomics; the slab is cleared by a kernel (as a memset node of the graph the clear was not ordered
    before the kernel on later replays and the gradient came back accumulated / non-finite)."""
    from pde_amd import scripts
    from pde_amd.lbfgs import LBFGS
    torch.manual_seed(4)
    lb, ub = np.array([-1.0, 0.0]), np.array([1.0, 0.99])
    m = scripts.burgers.PI.PhysicsInformedNN([2, width, width, width, 1], lb, ub)
    rng = np.random.default_rng(1)
    x0 = rng.uniform(lb[0], ub[0], (50, 1)); tb = rng.uniform(lb[1], ub[1], (50, 1))
    m.set_training_data(x0, -np.sin(np.pi * x0), tb, lb + (ub - lb) * rng.random((5000, 2)), 0.01 / np.pi)
    opt = LBFGS(list(m.dnn.parameters()), flat_params=m.dnn.flat_theta(), history_size=5, line_search_fn="strong_wolfe")
    m.optimizer_LBFGS = opt
    opt.zero_grad()
    loss_e = float(m._loss_terms()["loss"].detach())
    torch.cuda.synchronize()
    g_e = opt.flat_grad().clone()
    closure = m._graph_closure()
    assert closure is not None
    m.iter = 1
    for _ in range(4):
        loss = closure()
        torch.cuda.synchronize()
        g = opt.flat_grad()
        assert torch.isfinite(g).all()
        assert abs(float(loss.detach()) - loss_e) <= 1e-6 * abs(loss_e)
        assert float((g - g_e).norm() / g_e.norm()) < 1e-5


def test_gnss_sliding_window_end_to_end(tmp_path):
    """Two-output NLS/GNSS net at the reference's width 100 through the sliding-window driver on the HIP
    path: captured Adam phase, device L-BFGS with the captured closure, window warm start, stitching of
    u, v and h (NLS/sw-GNSS-adamW.py:262-404)."""
    from pde_amd import drivers, scripts
    x = np.linspace(-5, 5, 48)[:, None]
    t = np.linspace(-2, 0, 11)[:, None]
    uu = (1.0 / np.cosh(x.T - 0.3 * t)) * np.exp(1j * (0.5 * x.T + 0.4 * t))
    cls = type("GSW", (scripts.gnss.sw.PhysicsInformedNN,), {"base_dir": str(tmp_path), "use_hip_graph": True})
    models, (fu, fv, fh), errs = drivers.sliding_window(cls, [2, 100, 100, 100, 100, 2], x, t, uu, N0=32, Nb=16,
                                                        Nf=3000, n_overlap=16, nIter=20, LBFGS_max_iter=5,
                                                        log_name=None, rng=np.random.default_rng(0))
    assert len(models) == 3 and fu.shape == fv.shape == fh.shape == (11 * 48, 1)
    assert all(np.isfinite(e) for e in errs) and np.isfinite(fh).all()
    assert all(type(m.optimizer_LBFGS).__module__ == "pde_amd.lbfgs" for m in models)
    assert all(hasattr(m, "_lbfgs_graph") for m in models)
    # the stitched field is what the reloaded window models predict
    _, (fu2, fv2, fh2), errs2 = drivers.sliding_window(cls, [2, 100, 100, 100, 100, 2], x, t, uu, train=False)
    assert np.allclose(fu2, fu, atol=1e-6) and np.allclose(fv2, fv, atol=1e-6) and np.allclose(errs2, errs, atol=1e-6)


@pytest.mark.parametrize("adamw", [False, True])
def test_fused_adam_follows_torch_adam(adamw):
    """pde_amd.optim.FusedAdam (one launch per step, pinn_adam_step) against torch.optim.Adam / AdamW on the same
    gradients: 60 steps on a flat 7 851-vector with a drifting synthetic gradient, parameters and both moments
    agree to float32 rounding (1e-6 relative), the step count lives on the device."""
    from pde_amd.optim import FusedAdam
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    n = 7851
    flat = torch.randn(n, device=dev) * 0.3
    shapes = [(50, 2), (50,), (50, 50), (50,), (50, 50), (50,), (50, 50), (50,), (1, 50), (1,)]
    def views(buf):
        out, off = [], 0
        for sh in shapes:
            k = int(np.prod(sh))
            out.append(torch.nn.Parameter(buf[off:off + k].view(sh)))
            off += k
        return out
    fa_buf, ta_buf = flat.clone(), flat.clone()
    pf, pt = views(fa_buf), views(ta_buf)
    kw = dict(weight_decay=1e-2) if adamw else {}
    fa = FusedAdam(pf, flat_params=fa_buf, decoupled=adamw, **kw)
    ta = (torch.optim.AdamW if adamw else torch.optim.Adam)(pt, **kw)
    for it in range(60):
        g = torch.sin(torch.arange(n, device=dev) * 0.01 + it) * (1.0 + 0.1 * it) + 0.05 * fa_buf.detach()
        fa.zero_grad()
        fa.flat_grad().add_(g)
        ta.zero_grad()
        off = 0
        for p in pt:
            k = p.numel()
            p.grad = g[off:off + k].view(p.shape).clone()
            off += k
        fa.step()
        ta.step()
    torch.cuda.synchronize()
    assert fa.step_count == 60
    ref = ta_buf.detach().double()
    assert float((fa_buf.detach().double() - ref).norm() / ref.norm()) < 1e-6
    m_ref = torch.cat([ta.state[p]["exp_avg"].reshape(-1) for p in pt]).double()
    v_ref = torch.cat([ta.state[p]["exp_avg_sq"].reshape(-1) for p in pt]).double()
    assert float((fa.exp_avg.double() - m_ref).norm() / m_ref.norm()) < 1e-6
    assert float((fa.exp_avg_sq.double() - v_ref).norm() / v_ref.norm()) < 1e-6
    # state_dict round trip into a fresh optimizer (dtypes preserved: the counters are float64 on the device),
    # then both take the same further steps
    fb_buf = fa_buf.detach().clone()
    fb = FusedAdam(views(fb_buf), flat_params=fb_buf, decoupled=adamw, **kw)
    fb.load_state_dict(fa.state_dict())
    assert fb.step_count == 60 and fb.device_state.dtype == torch.float64
    for it in range(3):
        g = torch.cos(torch.arange(n, device=dev) * 0.02 + it)
        for o in (fa, fb):
            o.zero_grad()
            o.flat_grad().add_(g)
            o.step()
    torch.cuda.synchronize()
    assert fb.step_count == 63 and torch.equal(fb_buf, fa_buf.detach())
