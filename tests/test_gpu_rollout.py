"""epi_world_rollout_greedy: the trainer's greedy evaluation loop (MutliTrainer.scala:128-161) kept on the device,
against the oracle's WorldToy.stepOneGreedy and against the engine's own single steps."""
import numpy as np
import pytest

from oracle import oracle_lib as O
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def epi():
    import epidemicsimulator_amd as E
    assert E.lib().epi_device_count() >= 1, "no HIP device: GPU tests must run on the MI355X box"
    return E


def _toy_pair(epi, hidden=128, validate=True, t_max=120):
    hp = epi.HyperParams(learn_every=2**31 - 1, hidden=hidden)
    base = epi.State(0, 0, 0, 0, 0, 10_000, 0, t_max, 42)
    mk = lambda: epi.EpidemicEnv(0.26, 0.12, 0.008, 0.02, hp.reward_clip)
    w = epi.WorldToy(hp, base, 0, epi.ToyConn.W, make_env=mk, validate=validate, learning=False)
    ohp = O.default_hyper(); ohp.learn_every = 2**31 - 1; ohp.hidden = hidden
    env = O.EnvParams(0.26, 0.12, 0.008, 0.02, 5.0, 0.0)
    dflt = util.prof_dict(epi.Geo.CountryProfiles.get_profile("Unknown"))
    ow = O.World([c.pop for c in epi.ToyConn.C], epi.ToyConn.W, 0, util.oracle_state(base), [env] * 6, [dflt] * 6, [42] * 6,
                 hp=ohp, validate=validate)
    return w, ow


@pytest.mark.parametrize("hidden", [128, 16, 200, 256])
def test_rollout_untrained_matches_oracle(epi, hidden):
    steps = 150
    w, ow = _toy_pair(epi, hidden)
    out = w.rollout_greedy(steps, 100)
    series = np.zeros((steps, 6))
    for k in range(steps):
        ow.step(100, True)
        series[k] = [x.i for x in ow.states()]
    assert ow.error()[0] == 0
    np.testing.assert_array_equal(out["i_series"][:, 0, :], series)
    want = ow.states()
    for f in ("s", "i", "r", "d", "v", "hosp_cap", "t"):
        np.testing.assert_array_equal(out["states"][0][f], [getattr(x, f) for x in want])
    for c in range(6):
        assert out["arrival"][0, c] == epi.Validation.arrival_time(series[:, c], 10.0)      # Validation.scala:15-19
        assert out["peak"][0, c] == epi.Validation.peak_i(series[:, c])
    # a second rollout continues from where the first stopped, its metrics count from its own first step
    out2 = w.rollout_greedy(30, 100, want_series=False)
    for k in range(30):
        ow.step(100, True)
    np.testing.assert_array_equal(out2["states"][0]["i"], [x.i for x in ow.states()])
    assert (out2["arrival"] <= 30).all()
    w.close(); ow.close()


@pytest.mark.parametrize("n,batch", [(40, 5), (300, 2), (1200, 2)])
def test_rollout_equals_single_greedy_steps(epi, n, batch):
    """Dense and sparse kernels, several instances started from different states: K steps in one call equal K
    calls of epi_world_step_greedy (same env streams, same network)."""
    from epidemicsimulator_amd import synth
    countries, xy = synth.make_countries(n, 50 + n)
    conn = synth.gravity_csr(countries, xy, 50 + n, mean_degree=8.0) if n > 1024 else synth.gravity_dense(countries, xy)
    rng = np.random.default_rng(n)
    base = epi.State(0, 0, 0, 0, 0, 10_000, 0, 90, 42)
    worlds = []
    for _ in range(2):
        it = iter([epi.EpidemicEnv(0.25 + 0.001 * k, 0.12, 0.008, 0.04, 5.0, 0.0, c.name, 100 + k, profile=c)
                   for k, c in enumerate(countries)])
        worlds.append(epi.WorldToy(epi.HyperParams(), base, 3, conn, make_env_for=lambda name, pop: next(it), validate=True,
                                   countries=countries, batch=batch, learning=False))
    wa, wb = worlds
    # separate the instances: a few steps with random actions
    for _ in range(3):
        acts = rng.integers(0, 8, (batch, n)); gauss = rng.normal(size=(batch, n))
        wa.step_arrays(acts, gauss, 90); wb.step_arrays(acts, gauss, 90)
    K = 9
    out = wa.rollout_greedy(K, 90)
    wb._ensure_untrained_mlp()
    import ctypes as C
    series = np.zeros((K, batch, n))
    st = np.empty((batch, n), dtype=epi.WorldToy.STATE_DTYPE)
    for k in range(K):
        epi.check(epi.lib().epi_world_step_greedy(wb._h, None, C.c_int32(90), C.cast(st.ctypes.data, C.POINTER(epi._capi.CState)),
                                                  None, None))
        series[k] = st["i"]
    np.testing.assert_array_equal(out["i_series"], series)
    np.testing.assert_array_equal(out["states"], st)
    fa, fb = wa.last_flows(), wb.last_flows()
    np.testing.assert_array_equal(fa[0], fb[0]); np.testing.assert_array_equal(fa[1], fb[1])
    wa.close(); wb.close()


@pytest.mark.parametrize("hidden", [64, 32])
def test_rollout_per_node_networks_on_a_large_batch(epi, hidden):
    """Every node asks its own network (WorldToy.scala:177), here through the C ABI with 130 instances per node.  With
    64 hidden units that evaluation takes the tiled network kernel, with one instance (or 32 units) the unit-per-thread
    one: instances that start alike and share the noise draws stay alike, and equal the single-instance world."""
    import ctypes as C
    from epidemicsimulator_amd import synth
    countries, xy = synth.make_countries(9, 77)
    n = len(countries)
    conn = synth.gravity_dense(countries, xy)
    base = epi.State(0, 0, 0, 0, 0, 10_000, 0, 60, 42)
    rng = np.random.default_rng(hidden)
    nets = [[w + 0.05 * rng.standard_normal(w.shape) for w in epi.mlp_init(hidden)] + [0.1 * rng.standard_normal(hidden),
            0.1 * rng.standard_normal(hidden), 0.1 * rng.standard_normal(8)] for _ in countries]
    hp = epi.HyperParams(hidden=hidden)

    def rollout(batch, steps=12):
        it = iter([epi.EpidemicEnv(0.25 + 0.01 * k, 0.12, 0.008, 0.04, 5.0, 0.0, c.name, 100 + k, profile=c)
                   for k, c in enumerate(countries)])
        w = epi.WorldToy(hp, base, 2, conn, make_env_for=lambda name, pop: next(it), validate=True, countries=countries,
                         batch=batch, learning=False)
        pool = epi.Agents([hp] * n)
        for k, (W1, W2, W3, b1, b2, b3) in enumerate(nets):
            pool.set_weights(k, [W1, b1, W2, b2, W3, b3])
        series = np.zeros((steps, batch, n)); arr = np.zeros((batch, n), dtype=np.int32); pk = np.zeros((batch, n))
        st = np.empty((batch, n), dtype=epi.WorldToy.STATE_DTYPE)
        epi.check(epi.lib().epi_world_rollout_greedy(w._h, pool._h, C.c_int32(steps), C.c_int32(60), series.ctypes.data,
                                                     arr.ctypes.data, pk.ctypes.data, st.ctypes.data))
        w.close()
        return series, st

    sa, sta = rollout(130)
    sb, stb = rollout(1)
    for e in range(130):
        np.testing.assert_array_equal(sa[:, e, :], sb[:, 0, :])
        np.testing.assert_array_equal(sta[e], stb[0])
    # the per-node networks are really the ones consulted: the untrained network takes other actions
    it = iter([epi.EpidemicEnv(0.25 + 0.01 * k, 0.12, 0.008, 0.04, 5.0, 0.0, c.name, 100 + k, profile=c)
               for k, c in enumerate(countries)])
    w0 = epi.WorldToy(hp, base, 2, conn, make_env_for=lambda name, pop: next(it), validate=True, countries=countries,
                      batch=1, learning=False)
    assert not np.array_equal(w0.rollout_greedy(12, 60)["i_series"], sb)
    w0.close()


def test_rollout_with_trained_agents_and_evaluation(epi):
    """Every node asks its own, trained, network (WorldToy.scala:177): a short training run on both sides, then the
    evaluation -- snapshot, reset, greedy steps, restore -- against the oracle's learners."""
    names = [c.name for c in epi.ToyConn.C]
    hp = epi.HyperParams(hidden=32, batch_size=3, learn_every=2, epsilon_decay_steps=30, epsilon_end=0.1, target_update_every=3,
                         replay_capacity=40, lr=3e-3)
    chp = {"India": epi.HyperParams(hidden=32, batch_size=2, learn_every=3, epsilon_decay_steps=20, epsilon_end=0.3,
                                    target_update_every=2, replay_capacity=30, lr=1e-2, clip_q=2.0)}
    t_max = 200
    base = epi.State(1e6, 0, 0, 0, 0, 10000.0, 0, t_max, 1)
    w = epi.WorldToy(hp, base, validate=True, country_hyper_params=chp)
    envs = w._envs
    ow = O.World([c.pop for c in w.countries], np.asarray(epi.ToyConn.W, dtype=np.float64), 0, util.oracle_state(base),
                 [O.EnvParams(e.base_beta, e.base_gamma, e.base_ifr, e.noise, e.reward_clip, e.risk_factor) for e in envs],
                 [util.prof_dict(e.country_profile) for e in envs], [e.seed for e in envs], validate=True)
    hps = [chp.get(nm, hp) for nm in names]
    ow.enable_learning([O.LearnHyper(h.gamma, h.lr, h.clip_q, h.grad_clip, h.ema_alpha, h.hidden, h.replay_capacity,
                                     h.target_update_every, h.batch_size * 4) for h in hps],
                       [O.Hyper(h.epsilon_start, h.epsilon_end, h.epsilon_decay_steps, h.hidden, h.clip_q, h.reward_clip,
                                h.batch_size, h.learn_every) for h in hps],
                       [h.learn_every for h in hps], [h.batch_size for h in hps], [2 if nm == "India" else 1 for nm in names])
    train = 45
    learned = 0
    for _ in range(train):
        w.step_one(t_max)
        ow.step_learn(t_max)
        learned += sum(l is not None for l in w.last_losses)
    assert learned > 20 and ow.error()[0] == 0
    before = w.snapshot_states()
    # engine: the whole evaluation on the device
    K = 60
    metrics = w.evaluate_greedy(K)
    after = w.snapshot_states()
    assert [(s.s, s.i, s.r, s.d, s.v, s.hosp_cap, s.t) for s in after] == [(s.s, s.i, s.r, s.d, s.v, s.hosp_cap, s.t) for s in before]
    # oracle: reset, K greedy steps with each node's own learner, the env streams continuing after the training draws
    ow.reset()
    gauss = [O.jrand_gaussians(e.seed, train + K) for e in envs]
    learners = [ow.learner(c) for c in range(6)]
    series = np.zeros((K, 6))
    for k in range(K):
        cur = ow.states()
        acts = [int(np.argmax(learners[c].q_values(cur[c]))) for c in range(6)]
        ow.step_given(acts, [gauss[c][train + k] for c in range(6)], K)
        series[k] = [x.i for x in ow.states()]
    for c, nm in enumerate(names):
        pop = float(w.nodes[c].pop)
        assert metrics["eval_arrival_%s" % nm] == epi.Validation.arrival_time(series[:, c], 10.0)
        assert metrics["eval_peakI_%s" % nm] == epi.Validation.peak_i(series[:, c])
        assert metrics["eval/arrivalN_%s" % nm] == epi.Validation.arrival_time(series[:, c] / max(1.0, pop), 1e-4)
        assert metrics["eval/peakI_norm_%s" % nm] == epi.Validation.peak_i(series[:, c] / max(1.0, pop))
    w.close(); ow.close()
