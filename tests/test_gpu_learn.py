"""GPU parity of the DDQN learners (epi_agents_*) with the CPU oracle: replay sampling, ddqnTarget,
MLP.trainBatch, target-network copies.  Bit-exact: weights, Q-values, losses are compared with ==."""
import numpy as np
import pytest

from oracle import oracle_lib as O
from tests import util

pytestmark = pytest.mark.gpu

NAMES = ["W1", "b1", "W2", "b2", "W3", "b3"]


@pytest.fixture(scope="module")
def epi():
    import epidemicsimulator_amd as E
    assert E.lib().epi_device_count() >= 1, "no HIP device: GPU tests must run on the MI355X box"
    return E


def _state_tuple(rng, t, t_max=60):
    pop = int(rng.integers(10_000, 50_000_000))
    i = int(rng.integers(0, max(2, pop // 50))); r = int(rng.integers(0, pop // 10)); d = int(rng.integers(0, pop // 500))
    v = int(rng.integers(0, pop // 20))
    return (float(pop - i - r - d), float(i), float(r), float(d), float(v), float(rng.integers(50, 100_000)), int(t), t_max, 1)


def _feed(epi, rng, agents, learners, count, ids=None):
    """`count` random transitions per agent, the same ones into the engine and into the oracle learners."""
    ids = list(range(len(learners))) if ids is None else ids
    for k in range(count):
        trs = []
        for a in ids:
            s = _state_tuple(rng, k % 60); s2 = _state_tuple(rng, k % 60 + 1)
            act = int(rng.integers(0, 8)); r = float(rng.normal() * (12.0 if k % 5 == 0 else 2.0)); done = bool(rng.random() < 0.15)
            learners[a].observe(O.State(*s, 0), act, r, O.State(*s2, 0), done)
            trs.append(epi.Transition(epi.State(*s), act, r, epi.State(*s2), done))
        agents.observe(trs, ids)


def _hps(epi, n, **kw):
    out = []
    for a in range(n):
        d = dict(gamma=0.99 - 0.05 * a, lr=1e-3 * (1 + a), clip_q=10.0 - 3.0 * a, grad_clip=1.0 if a != 1 else 0.05,
                 ema_alpha=0.1 * a, target_update_every=3 + a)
        d.update(kw)
        out.append(epi.HyperParams(**d))
    return out


def _oracle_hp(h, max_need):
    return O.LearnHyper(h.gamma, h.lr, h.clip_q, h.grad_clip, h.ema_alpha, h.hidden, h.replay_capacity,
                        h.target_update_every, max_need)


def _assert_same_networks(agents, learners):
    for a, L in enumerate(learners):
        for tgt in (False, True):
            got, want = agents.weights(a, tgt), L.weights(tgt)
            for name, g, w in zip(NAMES, got, want):
                assert np.array_equal(g, w), "agent %d %s %s: max diff %g" % (a, "target" if tgt else "online", name, np.abs(g - w).max())


@pytest.mark.parametrize("hidden,batch,accum,capacity,pushes", [
    (16, 4, 2, 40, 55),        # ring overwritten, tiny network
    (100, 8, 1, 500, 64),      # width off the 64 grid
    (128, 128, 4, 50000, 700),  # the reference's defaults: 512 samples per learn
    (256, 10, 3, 64, 64),      # widest supported, need not a multiple of 4
    (130, 6, 2, 40, 50),       # 192 padded units: twelve waves, three chunks of activations
])
def test_learn_matches_oracle(epi, hidden, batch, accum, capacity, pushes):
    rng = np.random.default_rng(hidden + batch)
    hps = _hps(epi, 3, hidden=hidden, batch_size=batch, replay_capacity=capacity)
    agents = epi.Agents(hps, max_grad_accum=accum)
    learners = [O.Learner(_oracle_hp(h, batch * accum)) for h in hps]
    _feed(epi, rng, agents, learners, pushes)
    assert list(agents.replay_size()) == [L.replay_size() for L in learners]
    _assert_same_networks(agents, learners)
    steps = [0, 0, 0]
    for it in range(5):
        steps = [s + int(rng.integers(1, 4)) for s in steps]
        ids = [0, 1, 2] if it != 2 else [2, 0]   # a subset, out of order
        for a in ids:
            learners[a].set_steps(steps[a])
        want = [learners[a].learn(batch, accum) for a in ids]
        loss, ema = agents.learn([steps[a] for a in ids], [batch] * len(ids), accum, ids)
        assert list(loss) == want, "learn call %d" % it
        assert list(ema) == [learners[a].ema_loss() for a in ids]
        assert all(w > 0 for w in want)
        _feed(epi, rng, agents, learners, 3)
    _assert_same_networks(agents, learners)
    probe = [epi.State(*_state_tuple(rng, 7)) for _ in range(3)]
    for tgt in (False, True):
        q, g = agents.q_values(probe, target=tgt)
        for a in range(3):
            want_q = learners[a].q_values(util.oracle_state(probe[a]), tgt)
            assert np.array_equal(q[a], want_q)
            assert g[a] == int(np.argmax(want_q))


def test_short_buffer_returns_zero_and_draws_nothing(epi):
    """DDQNAgent.scala:52-53: learn returns 0.0 before the shuffle while the buffer is short."""
    rng = np.random.default_rng(5)
    hps = _hps(epi, 2, hidden=16, batch_size=8, replay_capacity=100)
    agents = epi.Agents(hps, max_grad_accum=2)
    learners = [O.Learner(_oracle_hp(h, 16)) for h in hps]
    _feed(epi, rng, agents, learners, 15)            # 15 < 16
    _feed(epi, rng, agents, learners, 5, ids=[1])    # agent 1 reaches 20
    loss, _ = agents.learn([1, 1], [8, 8], 2)
    for L in learners:
        L.set_steps(1)
    want = [L.learn(8, 2) for L in learners]
    assert loss[0] == 0.0 and want[0] == 0.0 and loss[1] == want[1] > 0
    # agent 0 drew nothing: its next sample is the first of a fresh Random(123)
    assert list(agents.sample(0, 15)) == list(learners[0].sample(15))
    _assert_same_networks(agents, learners)
    _feed(epi, rng, agents, learners, 20)
    with pytest.raises(epi.EpiError):
        agents.learn([2, 2], [8, 8], 4)              # needs 32 samples, the handle was sized for 16


def test_replay_buffer_spec(epi):
    """ReplayBufferSpec.scala:13-35 restated: identical buffers give identical samples; the ring keeps the
    newest `capacity` transitions."""
    def mk(i):
        return epi.State(1000 - i, float(i), 0.0, 0.0, 0.0, 100.0, i, 999, 1)

    def tr(i):
        return epi.Transition(mk(i), i % 3, float(i), mk(i + 1), False)

    hp = epi.HyperParams(hidden=16, replay_capacity=16, batch_size=4)
    a = epi.Agents([hp, hp], max_grad_accum=3)
    for i in range(16):
        a.observe([tr(i), tr(i)])
    sa, sb = a.sample(0, 12), a.sample(1, 12)
    assert len(sa) == 12 and list(sa) == list(sb)
    ring = epi.Agents([epi.HyperParams(hidden=16, replay_capacity=4, batch_size=1)], max_grad_accum=4)
    for i in range(8):
        ring.observe([tr(i)])
    assert ring.replay_size()[0] == 4
    # positions 0..3 of the ring now hold transitions 4..7 (ReplayBuffer.scala:16-17)
    assert sorted(ring.sample(0, 4)) == [0, 1, 2, 3]
    L = O.Learner(O.default_learn_hyper(hidden=16, replay_capacity=4))
    for i in range(8):
        L.observe(util.oracle_state(mk(i)), i % 3, float(i), util.oracle_state(mk(i + 1)), False)
    fresh = epi.Agents([epi.HyperParams(hidden=16, replay_capacity=4, batch_size=1)], max_grad_accum=4)
    for i in range(8):
        fresh.observe([tr(i)])
    assert list(fresh.sample(0, 4)) == list(L.sample(4))
    # the stored transitions are the last four: one learn on all four of them matches the oracle's
    L.set_steps(1)
    want = L.learn(1, 4)
    loss, _ = fresh.learn([1], [1], 4)
    assert loss[0] == want and want > 0


def test_ddqn_agent_mirror(epi):
    """DDQNAgent's own API (act / observe / learn / epsilonNow / qValues) on one agent against the oracle parts."""
    from oracle import pyref as P
    rng = np.random.default_rng(11)
    hp = epi.HyperParams(hidden=32, batch_size=4, replay_capacity=64, epsilon_decay_steps=50, target_update_every=5)
    ag = epi.DDQNAgent(hp)
    L = O.Learner(_oracle_hp(hp, 16))
    rnd = P.JavaRandom(9)
    steps = 0
    for k in range(60):
        s = _state_tuple(rng, k % 60); s2 = _state_tuple(rng, k % 60 + 1)
        st = epi.State(*s)
        frac = min(1.0, steps / 50.0)               # epsilon is read before act() counts the step (:26-27,32)
        eps = 1.0 + (0.1 - 1.0) * frac
        assert ag.epsilon_now == eps
        a = ag.act(st)
        steps += 1
        if rnd.next_double() < eps:
            want_a = rnd.next_int(8)
        else:
            want_a = int(np.argmax(L.q_values(util.oracle_state(st))))
        assert a == want_a, "step %d" % k
        r = float(rng.normal()); done = bool(k % 9 == 0)
        ag.observe(epi.Transition(st, a, r, epi.State(*s2), done))
        L.observe(O.State(*s, 0), a, r, O.State(*s2, 0), done)
        if k % 4 == 3:
            L.set_steps(steps)
            assert ag.learn(4, 2) == L.learn(4, 2)
            assert ag.ema_loss_now == L.ema_loss()
    assert ag.replay_size == L.replay_size() == 60
    assert np.array_equal(ag.q_values(epi.State(*s)), L.q_values(O.State(*s, 0)))


def test_world_step_one_with_learning_agents(epi):
    """WorldToy.stepOne end to end (:83-170) with the agents learning: the toy world with small batches so that
    the schedule fires early, per-country hyper-parameters (countryHyperParams, :49) and India's doubled
    intensity (:157-162).  States, rewards, losses and every node's two networks against the oracle world."""
    base = epi.State(1e6, 0, 0, 0, 0, 10000.0, 0, 400, 1)
    hp = epi.HyperParams(hidden=32, batch_size=4, learn_every=3, epsilon_decay_steps=40, target_update_every=7,
                         replay_capacity=30)
    chp = {"India": epi.HyperParams(hidden=32, batch_size=3, learn_every=2, epsilon_decay_steps=25, lr=5e-3,
                                    target_update_every=4, replay_capacity=50, ema_alpha=0.2)}
    w = epi.WorldToy(hp, base, validate=True, country_hyper_params=chp)
    names = [c.name for c in w.countries]
    assert "India" in names
    envs = w._envs
    ow = O.World([c.pop for c in w.countries], np.asarray(epi.ToyConn.W, dtype=np.float64), 0, util.oracle_state(base),
                 [O.EnvParams(e.base_beta, e.base_gamma, e.base_ifr, e.noise, e.reward_clip, e.risk_factor) for e in envs],
                 [util.prof_dict(e.country_profile) for e in envs], [e.seed for e in envs], validate=True)
    hps = [chp.get(nm, hp) for nm in names]
    ow.enable_learning([_oracle_hp(h, h.batch_size * 4) for h in hps],
                       [O.Hyper(h.epsilon_start, h.epsilon_end, h.epsilon_decay_steps, h.hidden, h.clip_q, h.reward_clip,
                                h.batch_size, h.learn_every) for h in hps],
                       [h.learn_every for h in hps], [h.batch_size for h in hps],
                       [2 if nm == "India" else 1 for nm in names])
    learned = 0
    for step in range(90):
        out = w.step_one(400)
        want_r, want_loss = ow.step_learn(400)
        assert ow.error()[0] == 0
        want_s = ow.states()
        for k, (nm, s, r) in enumerate(out):
            assert (s.s, s.i, s.r, s.d, s.v, s.hosp_cap, s.t) == want_s[k].tup(), "step %d node %s" % (step, nm)
            assert r == want_r[k]
            got_loss = w.last_losses[k]
            assert (got_loss is None and want_loss[k] == -1.0) or got_loss == want_loss[k], "step %d node %s" % (step, nm)
            learned += got_loss is not None
    assert learned > 100
    for k in range(w.n):
        L = ow.learner(k)
        L.hp = _oracle_hp(hps[k], 1)
        for tgt in (False, True):
            for name, g, x in zip(NAMES, w._pool.weights(k, tgt), L.weights(tgt)):
                assert np.array_equal(g, x), "node %d %s" % (k, name)
    # greedy evaluation now uses the trained networks (:172-235)
    out = w.step_one_greedy(400)
    acts = [int(np.argmax(ow.learner(k).q_values(want_s[k]))) for k in range(w.n)]
    assert [nd.last_a for nd in w.nodes] == acts


def test_empty_calls_are_no_ops(epi):
    """No agents listed: observe, learn and q_values return without touching anything."""
    hp = epi.HyperParams(hidden=16, batch_size=2, replay_capacity=8)
    agents = epi.Agents([hp, hp], max_grad_accum=2)
    before = agents.weights(1)
    agents.observe([], [])
    loss, ema = agents.learn([], [], 2, [])
    assert len(loss) == 0 and len(ema) == 0
    q, g = agents.q_values([], [])
    assert q.shape == (0, 8) and len(g) == 0
    assert list(agents.replay_size()) == [0, 0]
    assert list(agents.sample(0, 3)) == []
    for x, y in zip(before, agents.weights(1)):
        assert np.array_equal(x, y)
    with pytest.raises(epi.EpiError):
        agents.observe([epi.Transition(epi.State(1, 0, 0, 0, 0, 1, 0, 1, 1), 9, 0.0, epi.State(1, 0, 0, 0, 0, 1, 1, 1, 1), False)], [0])
    with pytest.raises(epi.EpiError):
        agents.q_values([epi.State(1, 0, 0, 0, 0, 1, 0, 1, 1)] * 2, [1, 1])   # an agent listed twice


def test_transitions_filed_on_the_device(epi):
    """epi_agents_observe_world (the transitions of WorldToy.scala:97-100 built from the world's device planes)
    against epi_agents_observe fed with transitions rebuilt on the host: (1) the whole training loop of a single
    world, both ways, losses and networks equal; (2) a batch of three instances: every agent receives its country's
    transition of every instance, in instance order."""
    hp = epi.HyperParams(hidden=16, batch_size=3, learn_every=2, epsilon_decay_steps=25, epsilon_end=0.2, target_update_every=3,
                         replay_capacity=25, lr=5e-3)
    base = epi.State(1e6, 0, 0, 0, 0, 10000.0, 0, 40, 1)     # tMax 40: the clock saturates and `done` fires inside the run
    wa = epi.WorldToy(hp, base, validate=True, device_transitions=True)
    wb = epi.WorldToy(hp, base, validate=True, device_transitions=False)
    for step in range(50):
        oa, ob = wa.step_one(40), wb.step_one(40)
        assert [(s.s, s.i, s.r, s.d, s.v, s.t) for _, s, _ in oa] == [(s.s, s.i, s.r, s.d, s.v, s.t) for _, s, _ in ob]
        assert wa.last_losses == wb.last_losses, step
    for k in range(6):
        for tgt in (False, True):
            for x, y in zip(wa._pool.weights(k, tgt), wb._pool.weights(k, tgt)):
                np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(wa._pool.replay_size(), wb._pool.replay_size())
    wa.close(); wb.close()

    # (2) batch of three worlds
    import ctypes as C
    B, n = 3, 6
    w = epi.WorldToy(hp, base, validate=True, batch=B, learning=False)
    epi.check(epi.lib().epi_world_keep_transitions(w._h, C.c_int32(1)))
    pa = epi.Agents([hp] * n, max_grad_accum=2)
    pb = epi.Agents([hp] * n, max_grad_accum=2)
    rng = np.random.default_rng(8)
    prev = w._get_states()
    for step in range(9):
        acts = rng.integers(-1, 10, (B, n))                 # out-of-range ids are clamped before they are stored
        gauss = rng.normal(size=(B, n))
        states, rewards = w.step_actions(acts, gauss, 40)
        ms, mi = w.last_flows()
        pa.observe_world(w)
        for c in range(n):
            for e in range(B):
                s3 = states[e][c]
                s2 = s3.copy(s=s3.s - float(ms[e][c]), i=s3.i - float(mi[e][c]), t=prev[e][c].t + 1)
                nn = max(1.0, s2.n)
                done = (s2.t >= s2.t_max) or (s2.i / nn < 1e-5) or (s2.d / nn > 0.02)
                pb.observe([epi.Transition(prev[e][c], int(np.clip(acts[e][c], 0, 7)), float(rewards[e][c]), s2, done)], [c])
        prev = states
    np.testing.assert_array_equal(pa.replay_size(), pb.replay_size())
    assert (pa.replay_size() == 25).all()                   # 27 pushes into rings of 25: the ring wrapped
    la, _ = pa.learn([5] * n, None, 2)
    lb, _ = pb.learn([5] * n, None, 2)
    np.testing.assert_array_equal(la, lb)
    assert (la != 0).all()
    for k in range(n):
        for x, y in zip(pa.weights(k), pb.weights(k)):
            np.testing.assert_array_equal(x, y)
    with pytest.raises(epi.EpiError):                       # a world that keeps nothing has nothing to file
        w2 = epi.WorldToy(hp, base, batch=1, learning=False)
        pa.observe_world(w2)
    w.close(); pa.close(); pb.close()


# ---------------------------------------------------------------- calibration sweeps whose rollouts learn
def _toy_sweep(epi, hp, steps, betas, travel, heterogeneous=True):
    from epidemicsimulator_amd import synth
    wl = dict(synth.workload("c1"), steps=steps, betas=betas, travel=travel)
    names = [c.name for c in wl["countries"]]
    intensity = [epi.WorldToy.LEARNING_INTENSITY.get(nm, 1) for nm in names]
    assert intensity == [1, 1, 2, 1, 1, 1] or sorted(intensity) == [1, 1, 1, 1, 1, 2]   # India learns twice as hard
    base = synth.base_state(steps)
    got = util.engine_grid(wl, base, heterogeneous, hp=hp, names=names)
    want = util.oracle_grid_learn(wl, base, hp, intensity, heterogeneous)
    return wl, base, got, want


def test_toy_sweep_600_steps_small_batches_equals_the_learning_oracle(epi):
    """The toy set with rollouts twice the default length and batchSize 8: the agents learn from step 32 on, every 8
    steps, and the episodes take argmax Q at act-calls 114 and 456 (DDQNAgent.scala:26-32) -- on trained weights,
    one network per (grid point, country).  Integers bit-exact, losses within 1e-12."""
    hp = epi.HyperParams(batch_size=8)
    wl, base, got, want = _toy_sweep(epi, hp, 600, [0.24, 0.26, 0.27], [0.5, 0.8, 0.9])
    util.assert_grid_equal(got, want)
    # the public entry point, learningIntensity taken from the country names
    rows, (best, pair) = epi.Calibrate.grid_surface(hp, epi.TrainConfig(steps_per_epoch=600), base, 0,
                                                    epi.Calibrate.Targets({c.name: wl["t_arrive"][k] for k, c in enumerate(wl["countries"])},
                                                                          {c.name: wl["t_peak"][k] for k, c in enumerate(wl["countries"])}),
                                                    wl["betas"], wl["travel"])
    np.testing.assert_allclose([r[2] for r in rows], want["loss"], rtol=1e-12, atol=0)
    assert abs(best - want["best_loss"]) <= 1e-12 * want["best_loss"]


@pytest.mark.parametrize("hidden,hetero,bs,every,eps0,lr", [(128, True, 4, 2, 0.3, 1e-2), (64, False, 2, 1, 0.2, 1e-1),
                                                            (24, True, 8, 5, 0.6, 5e-3)])
def test_toy_sweep_with_frequent_argmax_steps_equals_the_learning_oracle(epi, hidden, hetero, bs, every, eps0, lr):
    """Epsilon decays from eps0 to 0.05 within 100 steps and learning starts after 4 * batchSize steps, so most of the
    220 steps ask networks that have been trained: the trajectories depend on every learn call.  The sweep must differ
    from the one with untrained networks (otherwise the test shows nothing) and equal the learning oracle."""
    hp = epi.HyperParams(batch_size=bs, learn_every=every, hidden=hidden, epsilon_start=eps0, epsilon_end=0.05,
                         epsilon_decay_steps=100, target_update_every=10, lr=lr, grad_clip=5.0)
    wl, base, got, want = _toy_sweep(epi, hp, 220, [0.25, 0.27], [0.6, 0.9], hetero)
    util.assert_grid_equal(got, want)
    ohp, _ = util.oracle_hypers(hp)
    n = 6
    ta = np.zeros(n); tp = np.zeros(n); ha = np.zeros(n, dtype=np.uint8); hk = np.zeros(n, dtype=np.uint8)
    for c, v in wl["t_arrive"].items():
        ta[c], ha[c] = v, 1
    for c, v in wl["t_peak"].items():
        tp[c], hk[c] = v, 1
    untrained = O.grid_surface([c.pop for c in wl["countries"]], wl["conn"], 0, util.oracle_state(base),
                               [util.prof_dict(p) for p in wl["profiles"]], 220, wl["betas"], wl["travel"], ta, ha, tp, hk, hetero, hp=ohp)
    assert untrained["err"] == 3 and (untrained["peak"] != want["peak"]).any(), "learning changes the trajectories"


def test_learning_sweep_refusals(epi):
    """Without the learn step's hyper-parameters in the descriptor, for sparse worlds, through the allocation-free
    entry point and beyond 16,384 agents per call a sweep that would consult trained weights is refused."""
    import ctypes as C
    from epidemicsimulator_amd import synth, _capi
    import epidemicsimulator_amd as E
    hp = epi.HyperParams(batch_size=8, epsilon_start=0.5, epsilon_end=0.5)
    wl = synth.small_workload(20, 2, 2, 60, seed=5)
    d, keep = E._grid_desc(hp, 60, synth.base_state(60), 0, [c.pop for c in wl["countries"]], wl["conn"], wl["profiles"],
                           E.f64(wl["betas"]), E.f64(wl["travel"]), wl["t_arrive"], wl["t_peak"], True, 0)
    d.learn = None
    assert not E.lib().epi_grid_create(C.byref(d))
    assert E.lib().epi_last_status() == _capi.EPI_ERR_UNSUPPORTED and "epi_grid_desc.learn" in _capi.last_error()
    countries, xy = synth.make_countries(1100, 3)
    sp = dict(wl, countries=countries, profiles=countries, conn=synth.gravity_csr(countries, xy, 3, mean_degree=5.0))
    sp["t_arrive"], sp["t_peak"] = synth.targets(1100, 60, 3)
    with pytest.raises(epi.EpiError) as ei:
        util.engine_grid(sp, synth.base_state(60), True, hp=hp)
    assert ei.value.status == _capi.EPI_ERR_UNSUPPORTED and "dense worlds" in ei.value.message
    big = synth.small_workload(20, 30, 30, 60, seed=5)             # 900 grid points x 20 countries = 18,000 agents
    eng = epi.GridEngine(hp, 60, synth.base_state(60), 0, [c.pop for c in big["countries"]], big["conn"], big["profiles"],
                         big["betas"], big["travel"], big["t_arrive"], big["t_peak"], True)
    with pytest.raises(epi.EpiError) as ei:
        eng.run(want_metrics=False)
    assert ei.value.status == _capi.EPI_ERR_UNSUPPORTED and "16384" in ei.value.message
    part = eng.run(0, 4)                                           # the same handle takes a smaller call
    assert np.isfinite(part["loss"]).all()
    loss = util.DeviceArray(4, np.float64, -7.0)
    with pytest.raises(epi.EpiError) as ei:
        eng.enqueue_noalloc(0, 4, loss.ptr)
    assert ei.value.status == _capi.EPI_ERR_UNSUPPORTED
    eng.close()
