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
