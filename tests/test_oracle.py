"""CPU tests of the oracle itself (no GPU): what pins it, given that the reference ships no tests.

  - Random123 known-answer vectors for the Philox core (tests/golden/philox_kat.json)
  - the deterministic log / sincos / Box-Muller against double-precision NumPy
  - analytic known answers derived from VBLinear.lua's formulas (SURVEY.md 8c KAT 1-8)
  - PyTorch-CPU autograd and finite differences for every gradient
  - an independent float64 NumPy derivation of the whole LRT step (oracle/ref_numpy.py)
  - the committed golden file (regression pin; tests/golden/make_golden.py)
"""
import json
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_philox_known_answer_vectors(oracle):
    kat = json.load(open(os.path.join(HERE, "golden", "philox_kat.json")))
    assert len(kat) == 3
    for v in kat:
        assert oracle.philox_raw(v["ctr"], v["key"]) == v["out"]


def test_det_log_and_sincos_accuracy(oracle):
    rng = np.random.default_rng(0)
    for k in list(rng.integers(1, 1 << 24, 2000)) + [1, 2, 3, (1 << 24) - 1, 1 << 24, 1 << 23, 11863283]:
        x = np.float32(k * 2.0 ** -24)
        got = oracle.det_logf(float(x))
        want = math.log(float(x))
        assert abs(got - want) <= 2e-7 * abs(want) + 1e-9, (k, got, want)
    for k in list(rng.integers(0, 1 << 24, 2000)) + [0, 1, 1 << 21, (1 << 21) - 1, 1 << 22, 3 << 22, (1 << 24) - 1]:
        c, s = oracle.det_sincos2pi(int(k))
        th = 2 * math.pi * int(k) / 2 ** 24
        assert abs(c - math.cos(th)) <= 2.5e-7 and abs(s - math.sin(th)) <= 2.5e-7, k


def test_normals_are_standard_normal(oracle):
    z = oracle.fill_normal(512, 2048, 3, oracle.STREAM_EPS, 0, 1).astype(np.float64).ravel()
    n = z.size
    assert abs(z.mean()) < 5 / math.sqrt(n)
    assert abs(z.var() - 1) < 5 * math.sqrt(2 / n)
    assert abs((z ** 3).mean()) < 5 * math.sqrt(15 / n)
    assert abs((z ** 4).mean() - 3) < 5 * math.sqrt(96 / n)
    # Kolmogorov-Smirnov against the normal CDF, alpha far below 1e-6
    from scipy import stats
    assert stats.kstest(z[:200000], "norm").statistic < 2.5 / math.sqrt(200000)
    # streams, layers, draws and rows are decorrelated
    a = oracle.fill_normal(64, 64, 3, oracle.STREAM_EPS, 0, 1).ravel()
    for other in (oracle.fill_normal(64, 64, 3, oracle.STREAM_ZETA, 0, 1), oracle.fill_normal(64, 64, 3, 1, 1, 1),
                  oracle.fill_normal(64, 64, 3, 1, 0, 2), oracle.fill_normal(64, 64, 4, 1, 0, 1)):
        assert abs(np.corrcoef(a, other.ravel())[0, 1]) < 0.08
    # addressing: a sub-block generated with a row offset equals the same rows of the full block
    full = oracle.fill_normal(10, 9, 3, 2, 5, 3, 100)
    part = oracle.fill_normal(4, 9, 3, 2, 5, 3, 103)
    assert np.array_equal(full[3:7], part)


OPT = dict(var_init=1e-3, mu_init=1, B=1e6, S=1, seed=3)


def _layer(oracle, mode, I, O, rng):
    om = oracle.OracleVBLinear(I, O, dict(OPT, mode=mode))
    om.means[:] = rng.normal(0, math.sqrt(2.0 / I), (O, I)).astype(np.float32)
    om.lvars[:] = rng.normal(math.log(1e-3), 0.3, (O, I)).astype(np.float32)
    om.bias[:] = rng.normal(0, 0.1, O).astype(np.float32)
    om.compute_prior()
    return om


def test_fresh_layer_known_answers(oracle):
    """KAT 1, 2: mu_init = 0 => var_hat == var_init, LC == 0, KL gradients == 0; msr_init => 2 / I."""
    om = oracle.OracleVBLinear(64, 48, dict(OPT, mu_init=0))
    assert abs(om.var_hat - 1e-3) < 1e-9
    assert abs(om.calc_lc(OPT)) < 1e-9
    _, mlc = om.compute_mugrads(OPT)
    _, vlc = om.compute_vargrads(OPT)
    assert np.abs(mlc).max() == 0 and np.abs(vlc).max() < 1e-12
    om2 = oracle.OracleVBLinear(50, 20, dict(OPT, mu_init=0, msr_init=True))
    assert abs(om2.var_init - 2 / 50) < 1e-12 and np.allclose(om2.lvars, math.log(2 / 50))


def test_wn_known_answers(oracle):
    """KAT 3, 4, 6: MAP forward == Linear(means); e == 0 => WN == MAP and gradSum stays 0;
    gradWeight == g^T x, gradSum == (g^T x) . e, gradSum unaffected by `scale`."""
    rng = np.random.default_rng(1)
    om = _layer(oracle, "wn", 12, 9, rng)
    x = rng.normal(0, 1, (5, 12)).astype(np.float32)
    g = rng.normal(0, 1, (5, 9)).astype(np.float32)
    om.clamp_to_map()
    y_map = om.updateOutput(x)
    np.testing.assert_allclose(y_map, x @ om.means.T + om.bias, rtol=1e-5, atol=1e-6)
    w = oracle.sample(om.means, om.stdv, np.zeros_like(om.means))
    assert np.array_equal(w, om.means)
    om._map = False
    om.sample()
    om.updateOutput(x)
    om.backward(x, g, 0.25)
    np.testing.assert_allclose(om.gradWeight, 0.25 * (g.T @ x), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(om.gradSum, (g.T @ x) * om.e, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(om.gradBias, 0.25 * g.sum(0), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(om.gradInput, g @ om.weight, rtol=1e-5, atol=1e-6)


def test_wn_lrt_link_single_row(oracle):
    """KAT 5: for N = 1 the WN output x (mu + sigma e)^T + b equals the LRT output with the induced
    z_o = sum_i x_i sigma_oi e_oi / sqrt(sum_i x_i^2 sigma_oi^2)."""
    rng = np.random.default_rng(2)
    om = _layer(oracle, "wn", 16, 8, rng)
    x = rng.normal(0, 1, (1, 16)).astype(np.float32)
    om.sample()
    y_wn = om.updateOutput(x)
    z = ((x * om.stdv * om.e).sum(1) / np.sqrt((x * x * om.vars).sum(1)))[None, :].astype(np.float32)
    y_lrt, _, _ = oracle.lrt_forward(x, om.means, om.lvars, om.bias, z)
    np.testing.assert_allclose(y_lrt, y_wn, rtol=1e-5, atol=1e-6)


def test_lrt_gradients_against_torch_autograd(oracle):
    """The LRT backward (gradInput, d/dmeans, d/dlvars through gradSum, gradBias) against PyTorch-CPU
    autograd of y = x mu^T + b + sqrt((x.x) exp(lv)^T) . z in float64."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(3)
    N, I, O = 6, 10, 7
    om = _layer(oracle, "lrt", I, O, rng)
    x = rng.normal(0, 1, (N, I)).astype(np.float32)
    g = rng.normal(0, 1, (N, O)).astype(np.float32)
    om.sample()
    y = om.updateOutput(x)
    gx = om.backward(x, g, 1.0)
    zeta = oracle.fill_normal(N, O, 3, oracle.STREAM_ZETA, 0, om.draw, 0)
    tx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    tm = torch.tensor(om.means, dtype=torch.float64, requires_grad=True)
    tl = torch.tensor(om.lvars, dtype=torch.float64, requires_grad=True)
    tb = torch.tensor(om.bias, dtype=torch.float64, requires_grad=True)
    ty = tx @ tm.T + tb + torch.sqrt((tx * tx) @ torch.exp(tl).T) * torch.tensor(zeta, dtype=torch.float64)
    np.testing.assert_allclose(y, ty.detach().numpy(), rtol=2e-5, atol=2e-6)
    (ty * torch.tensor(g, dtype=torch.float64)).sum().backward()
    np.testing.assert_allclose(gx, tx.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(om.gradWeight, tm.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(om.gradBias, tb.grad.numpy(), rtol=2e-4, atol=2e-5)
    # VBLinear.lua:97: gradSum / (2S) . stdv must be dL/dlvars
    vle, _ = om.compute_vargrads(dict(OPT, S=1))
    np.testing.assert_allclose(vle, tl.grad.numpy(), rtol=5e-4, atol=1e-5)


def test_kl_gradients_by_finite_differences(oracle):
    """KAT 7: compute_mugrads / compute_vargrads KL terms are d(LC)/dmeans, d(LC)/dlvars with var_hat held
    fixed -- and equal the total derivative, because var_hat = mean(sigma^2 + mu^2) is the stationary point."""
    rng = np.random.default_rng(4)
    om = _layer(oracle, "wn", 6, 5, rng)
    opt = dict(OPT, B=10.0)
    _, mlc = om.compute_mugrads(opt)
    _, vlc = om.compute_vargrads(opt)

    def lc(means, lvars, free_var_hat):
        v = np.exp(lvars.astype(np.float64)); q = means.astype(np.float64) ** 2
        vh = (v + q).mean() if free_var_hat else om.var_hat
        return ((0.5 * np.log(vh) - 0.5 * np.log(v)) + (q + v - vh) / (2 * vh)).sum() / opt["B"]

    eps = 1e-4
    for free in (False, True):
        for (o, i) in [(0, 0), (2, 3), (4, 5)]:
            d = np.zeros_like(om.means); d[o, i] = eps
            fd_mu = (lc(om.means + d, om.lvars, free) - lc(om.means - d, om.lvars, free)) / (2 * eps)
            fd_lv = (lc(om.means, om.lvars + d, free) - lc(om.means, om.lvars - d, free)) / (2 * eps)
            assert abs(fd_mu - mlc[o, i]) <= 2e-3 * abs(mlc[o, i]) + 1e-7
            assert abs(fd_lv - vlc[o, i]) <= 2e-3 * abs(vlc[o, i]) + 1e-7
    assert abs(om.calc_lc(opt) - lc(om.means, om.lvars, False)) <= 1e-5 * abs(om.calc_lc(opt))


def test_wn_and_lrt_agree_in_expectation(oracle):
    """KAT 8 (statistical, fixed seeds): over 600 draws the WN and LRT outputs have the same mean and
    variance per activation, and the two d/dlvars estimators have the same mean (5-sigma bounds)."""
    rng = np.random.default_rng(5)
    N, I, O, D = 4, 12, 6, 600
    wn, lrt = _layer(oracle, "wn", I, O, rng), None
    lrt = oracle.OracleVBLinear(I, O, dict(OPT, mode="lrt"))
    lrt.means[:] = wn.means; lrt.lvars[:] = wn.lvars; lrt.bias[:] = wn.bias; lrt.compute_prior()
    x = rng.normal(0, 1, (N, I)).astype(np.float32)
    g = rng.normal(0, 1, (N, O)).astype(np.float32)
    ys = {"wn": [], "lrt": []}
    glv = {"wn": [], "lrt": []}
    for name, m in (("wn", wn), ("lrt", lrt)):
        for _ in range(D):
            m.gradSum[:] = 0
            m.sample()
            ys[name].append(m.updateOutput(x).copy())
            m.backward(x, 2 * (m.output - 0.3) * g * g, 1.0)          # a loss with curvature: sum g^2 (y - .3)^2
            glv[name].append((m.gradSum / 2 * m.stdv).copy())
    mean_want = x @ wn.means.T + wn.bias
    var_want = (x * x) @ wn.vars.T
    for name in ys:
        y = np.stack(ys[name])
        assert np.abs(y.mean(0) - mean_want).max() < 5 * np.sqrt(var_want.max() / D)
        assert np.abs(y.var(0) / var_want - 1).max() < 5 * math.sqrt(2 / D) * 1.5
    a, b = np.stack(glv["wn"]), np.stack(glv["lrt"])
    se = np.sqrt(a.var(0) / D + b.var(0) / D)
    assert (np.abs(a.mean(0) - b.mean(0)) <= 5.5 * se + 1e-12).all()


def test_mlp_step_against_independent_float64_derivation(oracle):
    """OracleMLP (C kernels, fp32 fma chains, module-by-module backward) against oracle/ref_numpy's
    emulate_lrt_step (float64, derived independently, fused formulas): loss and every gradient."""
    from oracle.ref_numpy import emulate_lrt_step
    opt = dict(OPT, mode="lrt", input_size=20, hidden=[16, 12], n_classes=10, S=1)
    onet = oracle.OracleMLP(opt)
    for om in onet.vb:
        om.means[:] = om.weight
        om.compute_prior()
    N = 9
    x = oracle.fill_normal(N, 20, 3, oracle.STREAM_DATA, 0, 0)
    t = (np.arange(N) * 3 % 10).astype(np.int32)
    onet.resetGradients(); onet.sample()
    err, acc = onet.run(x, t)
    layers = [dict(means=om.means, lvars=om.lvars, bias=om.bias) for om in onet.vb]
    zetas = [oracle.fill_normal(N, om.O, 3, oracle.STREAM_ZETA, k, 1, 0).astype(np.float64) for k, om in enumerate(onet.vb)]
    loss, res, gw3, gb3 = emulate_lrt_step(layers, onet.last.weight, onet.last.bias, x, t, zetas, B=opt["B"])
    assert abs(loss - err) <= 1e-5 * abs(err)
    for k, om in enumerate(onet.vb):
        np.testing.assert_allclose(om.gradWeight, res[k]["gradWeight"], rtol=0, atol=2e-5 * np.abs(res[k]["gradWeight"]).max())
        np.testing.assert_allclose(om.gradSum, res[k]["gradSum"], rtol=0, atol=5e-5 * np.abs(res[k]["gradSum"]).max())
        np.testing.assert_allclose(om.gradBias, res[k]["gradBias"], rtol=0, atol=2e-5 * np.abs(res[k]["gradBias"]).max())
        mle, mlc = om.compute_mugrads(opt)
        vle, vlc = om.compute_vargrads(opt)
        np.testing.assert_allclose(mle + mlc, res[k]["grad_mu"], rtol=0, atol=2e-5 * np.abs(res[k]["grad_mu"]).max())
        np.testing.assert_allclose(vle + vlc, res[k]["grad_lv"], rtol=0, atol=5e-5 * np.abs(res[k]["grad_lv"]).max())
    np.testing.assert_allclose(onet.last.gradWeight, gw3, rtol=0, atol=2e-5 * np.abs(gw3).max())
    np.testing.assert_allclose(onet.last.gradBias, gb3, rtol=0, atol=2e-5 * np.abs(gb3).max())


def test_reference_cpu_port_matches_oracle_math(oracle):
    """The BLAS port timed as cpu_baseline (oracle/ref_numpy.ReferenceCpuMLP) computes the same WN
    quantities as the C oracle when both are given the same weights and noise."""
    from oracle.ref_numpy import ReferenceCpuMLP
    net = ReferenceCpuMLP(12, [10, 8], 5, seed=3)
    rng = np.random.default_rng(6)
    x = rng.normal(0, 1, (7, 12)).astype(np.float32)
    t = rng.integers(0, 5, 7).astype(np.int32)
    err, acc = net.step(x, t)
    lay = net.layers[0]
    y0 = oracle.linear_forward(x, lay["weight"], lay["bias"])
    assert np.isfinite(err) and 0 <= acc <= 100
    np.testing.assert_allclose(lay["weight"], lay["means"] + lay["stdv"] * lay["e"], rtol=1e-6)
    # gradSum == gradWeight . e   (VBLinear.lua:113-115 with scale = 1, one draw)
    np.testing.assert_allclose(lay["gradSum"], lay["gradWeight"] * lay["e"], rtol=1e-5, atol=1e-7)
    assert y0.shape == (7, 10)


def test_adam_and_sgd_restatement(oracle):
    """vbo_adam_step == the published Adam in float64, and agrees with torch.optim.Adam (which places eps
    after the bias correction: identical up to eps-sized terms when |g| >> eps); vbo_sgd_step == x - lr g."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(8)
    x0 = rng.normal(0, 1, 257).astype(np.float32)
    gs = [rng.normal(0, 1, 257).astype(np.float32) for _ in range(5)]
    x, m, v = x0.copy(), np.zeros_like(x0), np.zeros_like(x0)
    x64, m64, v64 = x0.astype(np.float64), np.zeros(257), np.zeros(257)
    tx = torch.tensor(x0, dtype=torch.float64, requires_grad=True)
    topt = torch.optim.Adam([tx], lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    for t, g in enumerate(gs, start=1):
        upd = oracle.adam_step(x, g, m, v, 1e-2, 0.9, 0.999, 1e-8, t)
        m64 = 0.9 * m64 + 0.1 * g
        v64 = 0.999 * v64 + 0.001 * g.astype(np.float64) ** 2
        step = 1e-2 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        x64 = x64 - step * m64 / (np.sqrt(v64) + 1e-8)
        tx.grad = torch.tensor(g, dtype=torch.float64)
        topt.step()
        np.testing.assert_allclose(x, x64, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(x, tx.detach().numpy(), rtol=1e-4, atol=1e-5)
        assert np.abs(upd).max() <= 1.0001e-2 * 3.2          # |update| <= lr / sqrt(1 - b2) bound, loosely
    y = x0.copy()
    oracle.sgd_step(y, gs[0], 0.25)
    np.testing.assert_allclose(y, x0 - 0.25 * gs[0], rtol=1e-6, atol=1e-7)


def test_committed_golden_vectors(oracle):
    """Regression pin: the oracle reproduces tests/golden/vblinear_small.npz (made by make_golden.py)."""
    gold = np.load(os.path.join(HERE, "golden", "vblinear_small.npz"))
    assert np.array_equal(gold["normal_eps_5x7"], oracle.fill_normal(5, 7, 3, oracle.STREAM_EPS, 1, 1))
    assert np.array_equal(gold["normal_zeta_3x5_row1000"], oracle.fill_normal(3, 5, 3, oracle.STREAM_ZETA, 2, 7, 1000))
    for mode in ("wn", "lrt"):
        for (N, I, O) in ((3, 7, 5), (32, 64, 48)):
            pre = f"{mode}_{N}x{I}x{O}_"
            opt = dict(var_init=1e-3, mu_init=1, B=1e6, S=2, mode=mode, seed=3)
            om = oracle.OracleVBLinear(I, O, opt, layer_id=1)
            om.means[:] = gold[pre + "in_means"]; om.lvars[:] = gold[pre + "in_lvars"]; om.bias[:] = gold[pre + "in_bias"]
            om.compute_prior()
            x, g = gold[pre + "in_x"], gold[pre + "in_g"]
            for d in range(2):
                om.sample()
                np.testing.assert_allclose(om.updateOutput(x), gold[pre + f"out_output{d}"], rtol=1e-6, atol=1e-7)
                np.testing.assert_allclose(om.backward(x, g, 0.5), gold[pre + f"out_gradInput{d}"], rtol=1e-6, atol=1e-8)
            np.testing.assert_allclose(om.gradWeight, gold[pre + "out_gradWeight"], rtol=1e-6, atol=1e-8)
            np.testing.assert_allclose(om.gradSum, gold[pre + "out_gradSum"], rtol=1e-6, atol=1e-8)
            assert abs(om.var_hat - float(gold[pre + "out_var_hat"])) <= 1e-12


def test_numpy_philox_helper_is_the_contracts_generator(oracle):
    """tests/_philox_np.py (the words behind a window of normals, for the hardware-form error model test): Random123's known answers,
    the oracle's vbo_philox_raw on random counters, and float64 Box-Muller on its words against the oracle's normals."""
    import json
    from tests import _philox_np as P
    kat = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "philox_kat.json")))
    for v in kat:
        assert [int(w) for w in P.philox4x32_10(*v["ctr"], *v["key"])] == v["out"]
    rng = np.random.default_rng(0)
    for _ in range(20):
        c = [int(v) for v in rng.integers(0, 2 ** 32, 4)]
        k = [int(v) for v in rng.integers(0, 2 ** 32, 2)]
        assert [int(v) for v in P.philox4x32_10(*c, *k)] == oracle.philox_raw(c, k)
    x0, x1 = P.window_words(8, 64, 3, 2, 1, 5, 4096)
    z0, z1, _ = P.ideal_box_muller(x0, x1)
    want = oracle.fill_normal(8, 64, 3, 2, 1, 5, 4096)
    assert np.abs(np.stack([z0, z1], -1).reshape(8, 64) - want).max() < 1e-6         # the contract's polynomial forms vs float64
