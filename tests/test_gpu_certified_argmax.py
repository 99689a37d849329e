"""The argmax-Q step of large batches (DDQNAgent.actGreedy for every patch of every episode, DDQNAgent.scala:22-23,
MLP.scala:15-22) screens the states with fused multiply-adds and evaluates in the reference's arithmetic only where the
screen cannot certify the first argmax.  Whatever the screen decides, the actions -- and everything downstream of
them -- must be those of the exact evaluation alone."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def epi():
    import epidemicsimulator_amd as E
    return E


def _with_env(value, fn):
    """Runs fn() with EPI_MLP_CERTIFY set to `value` (None = unset): the library reads it when a network is uploaded."""
    old = os.environ.get("EPI_MLP_CERTIFY")
    try:
        if value is None:
            os.environ.pop("EPI_MLP_CERTIFY", None)
        else:
            os.environ["EPI_MLP_CERTIFY"] = value
        return fn()
    finally:
        if old is None:
            os.environ.pop("EPI_MLP_CERTIFY", None)
        else:
            os.environ["EPI_MLP_CERTIFY"] = old


@pytest.mark.parametrize("name", ["c2", "c4"])
def test_sweep_is_the_same_with_and_without_the_screen(epi, name):
    """A slab of the calibration grid (the rollouts take argmax Q at act-call 114): default (screen + exact pass for the
    flagged states), screen off, and every state sent through the exact pass give the same losses, arrival times and
    peaks, bit for bit -- dense and sparse worlds."""
    from epidemicsimulator_amd import synth
    wl = synth.workload(name)
    T = wl["steps"]
    base = synth.base_state(T)

    def sweep():
        eng = epi.GridEngine(epi.HyperParams(), T, base, wl["seed_idx"], [c.pop for c in wl["countries"]], wl["conn"],
                             wl["profiles"], wl["betas"][:3], wl["travel"][:4], wl["t_arrive"], wl["t_peak"], True)
        out = eng.run(want_metrics=True)
        eng.close()
        return out

    ref = _with_env("0", sweep)
    for mode in (None, "all"):
        got = _with_env(mode, sweep)
        for k in ("loss", "arrival", "peak"):
            np.testing.assert_array_equal(got[k], ref[k], err_msg="%s, EPI_MLP_CERTIFY=%s" % (k, mode))
    assert np.isfinite(ref["loss"]).all()


def test_exact_ties_go_to_the_exact_pass_and_take_the_first_index(epi):
    """A network whose output rows are pairwise identical produces exactly tied Q-values: no margin can certify them, the
    exact pass decides, and the first of the tied maxima is the action (the reference's argmax scan keeps the first
    strict maximum).  512 toy worlds = 3,072 states on one network: the large-batch kernels."""
    rng = np.random.default_rng(5)
    H = 64
    W1 = rng.normal(size=(H, 7)) * 0.5
    W2 = rng.normal(size=(H, H)) * 0.2
    W3 = rng.normal(size=(8, H)) * 0.3
    W3[1::2] = W3[0::2]          # outputs (0,1), (2,3), (4,5), (6,7) tie exactly
    b3 = np.repeat(rng.normal(size=4) * 0.1, 2)

    def actions():
        w = epi.WorldToy(epi.HyperParams(hidden=H), epi.State(0, 0, 0, 0, 0, 10_000, 0, 120, 42), 0, epi.ToyConn.W,
                         validate=True, batch=512)
        w.set_mlp(W1, W2, W3, None, None, b3)
        out = []
        from epidemicsimulator_amd import _capi
        lib = _capi.lib()
        st = (_capi.CState * (w.batch * w.n))()
        r = np.zeros((w.batch, w.n)); a = np.zeros((w.batch, w.n), dtype=np.int32)
        for _ in range(4):
            rc = lib.epi_world_step_greedy(w._h, None, C.c_int32(120), st, r.ctypes.data_as(C.POINTER(C.c_double)),
                                           a.ctypes.data_as(C.POINTER(C.c_int32)))
            assert rc == 0
            out.append(a.copy())
        w.close()
        return np.stack(out)

    ref = _with_env("0", actions)
    got = _with_env(None, actions)
    np.testing.assert_array_equal(got, ref)
    assert set(np.unique(ref)) <= {0, 2, 4, 6}     # always the first of a tied pair


# ---------------------------------------------------------------- adversarial margins
def _adversarial_net(rng, H, k_lo, k_hi, with_bias):
    """A network whose eight output rows are one random row with a single weight each moved by k units in the last
    place, k log-uniform in [k_lo, k_hi]: the Q-values of a state differ from each other by k * ulp(w) * h2_j -- from
    one rounding of the final sum up to 1e-6 -- so that the rounding of the sums themselves (fused or not, on the
    matrix cores or not) decides which of them is the largest."""
    W1 = rng.normal(size=(H, 7)) * 0.5
    b1 = rng.normal(size=H) * 0.1
    W2 = rng.normal(size=(H, H)) * (1.5 / np.sqrt(H))
    b2 = rng.normal(size=H) * 0.1
    base = rng.normal(size=H) * 0.3
    W3 = np.tile(base, (8, 1))
    for o in range(1, 8):
        j = int(rng.integers(H))
        k = int(np.exp(rng.uniform(np.log(k_lo), np.log(k_hi))))
        W3[o, j] = W3[o, j] + k * np.spacing(W3[o, j]) * (1 if rng.random() < 0.5 else -1)
    b3 = np.zeros(8)
    if with_bias:
        b3[:] = rng.normal() * 0.05               # one bias for all rows: the margins stay the weights' doing
    return W1, b1, W2, b2, W3, b3


def _random_states(rng, epi, B):
    """[B][6] State records with random integer compartments (epi_world_set_states takes whole numbers)."""
    from epidemicsimulator_amd import WorldToy
    pops = np.array([c.pop for c in epi.ToyConn.C], dtype=np.float64)
    st = np.zeros((B, 6), dtype=WorldToy.STATE_DTYPE)
    frac = rng.dirichlet(np.ones(5) * 0.7, size=(B, 6))
    scale = np.floor(pops[None, :] * rng.uniform(0.001, 1.0, size=(B, 6)))
    for k, f in enumerate(("s", "i", "r", "d", "v")):
        st[f] = np.floor(frac[:, :, k] * scale)
    st["hosp_cap"] = np.exp(rng.uniform(np.log(1.0), np.log(1e7), size=(B, 6)))
    st["t"] = rng.integers(0, 120, size=(B, 6))
    st["t_max"], st["seed"] = 120, 42
    return st


@pytest.mark.parametrize("H", [64, 128])
def test_near_tied_q_values_from_one_ulp_to_1e_minus_6(epi, H):
    """98,304 random states on networks built so that the top Q-values lie between one unit in the last place and
    1e-6 apart, some of them exactly equal at the clamp: with the screen on (default), off (EPI_MLP_CERTIFY=0) and
    distrusted (=all) the actions must be identical, state by state; and the screen must have decided some states itself
    and passed others on -- otherwise the test would not exercise the boundary it is about."""
    from epidemicsimulator_amd import _capi
    lib = _capi.lib()
    B = 16384
    rng = np.random.default_rng(100 + H)
    states = _random_states(rng, epi, B)
    nets = [(_adversarial_net(rng, H, 1, 16, False), 10.0), (_adversarial_net(rng, H, 1, 2 ** 10, True), 10.0),
            (_adversarial_net(rng, H, 2 ** 10, 2 ** 30, False), 10.0), (_adversarial_net(rng, H, 2 ** 20, 2 ** 44, True), 10.0),
            (_adversarial_net(rng, H, 2 ** 30, 2 ** 46, False), 10.0),
            (_adversarial_net(rng, H, 2 ** 30, 2 ** 46, True), 0.2)]       # clamp inside the range of the values: exact ties at +-0.2

    def run(net, clip):
        W1, b1, W2, b2, W3, b3 = net
        w = epi.WorldToy(epi.HyperParams(hidden=H), epi.State(0, 0, 0, 0, 0, 10_000, 0, 120, 42), 0, epi.ToyConn.W,
                         validate=False, batch=B)
        assert lib.epi_world_set_states(w._h, C.cast(states.ctypes.data, C.POINTER(_capi.CState))) == 0, _capi.last_error()
        w.set_mlp(W1, W2, W3, b1, b2, b3, clip_q=clip)
        st = (_capi.CState * (B * 6))()
        r = np.zeros((B, 6)); a = np.full((B, 6), -1, dtype=np.int32)
        assert lib.epi_world_step_greedy(w._h, None, C.c_int32(120), st, r.ctypes.data_as(C.POINTER(C.c_double)),
                                         a.ctypes.data_as(C.POINTER(C.c_int32))) == 0, _capi.last_error()
        stats = w.screen_stats()
        w.close()
        return a, stats

    screened = flagged = 0
    per_net = []
    for net, clip in nets:
        ref, st0 = _with_env("0", lambda: run(net, clip))
        got, st1 = _with_env(None, lambda: run(net, clip))
        allp, st2 = _with_env("all", lambda: run(net, clip))
        assert st0 == (0, 0), "screen off: nothing screened"
        assert st1[0] == B * 6 and st2 == (B * 6, B * 6), (st1, st2)
        np.testing.assert_array_equal(got, ref)
        np.testing.assert_array_equal(allp, ref)
        assert ref.min() >= 0 and ref.max() <= 7 and len(np.unique(ref)) >= 2
        screened += st1[0]; flagged += st1[1]
        per_net.append(st1[1])
    assert 0 < flagged < screened, per_net
    assert per_net[0] == B * 6, "margins of a few units in the last place can never be certified: %s" % per_net
    assert per_net[4] < B * 6, "margins around 1e-6 are far above the bound for most states: %s" % per_net
    print("flagged per network:", per_net)
