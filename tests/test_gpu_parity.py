"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same seeded
inputs.  Bar: bit-exact for every integer (allocations, compartments, flows, arrival, peak, actions);
fp64 losses and rewards within 1e-12 relative (rewards are in fact compared bit for bit)."""
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


# ---------------------------------------------------------------- allocateInt (env.scala:241-254)
def test_alloc_int_known_answers(epi):
    A = epi.EpidemicEnv.allocate_int
    assert A(10, [.5, .3, .2]) == [5, 3, 2]
    assert A(10, [1, 1, 1]) == [4, 3, 3]
    assert A(7, [0, 2, 1, 1]) == [0, 3, 2, 2]
    assert A(1, [.25] * 4) == [1, 0, 0, 0]
    assert A(5, [0, 0, 0]) == [0, 0, 0]
    assert A(0, [1, 2, 3]) == [0, 0, 0]
    assert A(-3, [1, 2, 3]) == [0, 0, 0]


@pytest.mark.parametrize("n", [1, 3, 6, 63, 64, 65, 200, 511, 1000, 1024])
def test_alloc_int_random_rows(epi, n):
    rng = np.random.default_rng(1000 + n)
    B = 48
    w = rng.random((B, n)) ** 3
    w[rng.random((B, n)) < 0.1] = 0.0
    w[3] = 1.0                       # all fractions identical: index-order ties
    w[4] = 0.0                       # zero sum
    w[5] = np.round(w[5] * 4) / 4    # many exact ties
    w[6, : n // 2] = 0.0
    totals = rng.integers(0, 2_000_000_000, B).astype(np.int32)
    totals[:12] = [0, 1, 2, 3, 5, 7, n, n + 1, max(n - 1, 0), 2 * n + 1, 17, 1_999_999_999]
    totals[3], totals[5] = 5 * n + n // 3, 123457
    got = epi.EpidemicEnv.allocate_int_batch(totals, w)
    for b in range(B):
        np.testing.assert_array_equal(got[b], O.alloc_int(totals[b], w[b]), err_msg="row %d total %d" % (b, totals[b]))


def test_alloc_int_small_totals_heavy_tail(epi):
    rng = np.random.default_rng(7)
    n, B = 1000, 64
    w = np.exp(rng.normal(0, 3.0, (B, n)))
    totals = rng.integers(1, 40, B).astype(np.int32)
    got = epi.EpidemicEnv.allocate_int_batch(totals, w)
    for b in range(B):
        np.testing.assert_array_equal(got[b], O.alloc_int(totals[b], w[b]))
        assert got[b].sum() == totals[b]


def test_alloc_int_rejects_bad_weights(epi):
    with pytest.raises(epi.EpiError):
        epi.EpidemicEnv.allocate_int(5, [1.0, float("nan")])
    with pytest.raises(epi.EpiError):
        epi.EpidemicEnv.allocate_int(5, [1.0, -0.5])


# ---------------------------------------------------------------- EpidemicEnv.step (env.scala:58-234)
def test_env_step_batch_matches_oracle(epi):
    rng = np.random.default_rng(11)
    B = 512
    countries, _ = __import__("epidemicsimulator_amd.synth", fromlist=["x"]).make_countries(B, 99)
    envs, states, acts, gauss = [], [], [], []
    for b in range(B):
        c = countries[b]
        envs.append(epi.EpidemicEnv(0.15 + 0.4 * rng.random(), 0.12, 0.004 + 0.02 * rng.random(), 0.05 + 0.1 * rng.random(),
                                    5.0, rng.random() * (b % 3 == 0), c.name, 42, profile=c))
        pop = int(c.pop)
        i = int(rng.integers(0, max(2, pop // 50)))
        r = int(rng.integers(0, pop // 10)); d = int(rng.integers(0, pop // 100)); v = int(rng.integers(0, pop // 10))
        states.append(epi.State(pop - i, i, r, d, v, float(rng.choice([1.0, 50.0, 1e4, 1e6])), int(rng.integers(0, 50)), 50, 1))
        acts.append(int(rng.integers(-1, 10)))
        gauss.append(float(rng.normal()))
    states[0] = epi.State(0, 0, 0, 0, 0, 1.0, 0, 10, 1)
    states[1] = epi.State(10, 0, 0, 0, 0, 1.0, 9, 10, 1)
    states[2] = epi.State(0.4, 2.5, 3.49, 0.5, 1.5, 7.0, 3, 10, 1)   # Math.round on non-integers
    nxt, rew, done = epi.EpidemicEnv.step_batch(envs, states, acts, gauss)
    for b in range(B):
        o_next, o_r, o_done = O.env_step(O.EnvParams(*[getattr(envs[b]._c(), f[0]) for f in O.EnvParams._fields_]),
                                         O.profiles_array([util.prof_dict(envs[b].country_profile)])[0],
                                         util.oracle_state(states[b]), acts[b], gauss[b])
        g = nxt[b]
        assert (g.s, g.i, g.r, g.d, g.v, g.hosp_cap, g.t) == o_next.tup(), "row %d" % b
        assert rew[b] == o_r, "reward row %d: %r vs %r" % (b, rew[b], o_r)
        assert bool(done[b]) == o_done


def test_heterogeneity_spec(epi):
    """HeterogeneitySpec.scala:18-29: different multipliers give a different next I."""
    cn = next(c for c in epi.ToyConn.C if c.name == "China")
    jp = next(c for c in epi.ToyConn.C if c.name == "Japan")
    s0 = epi.State(1_000_000, 1_000, 0, 0, 0, 10_000, 0, 10, 1)
    mk = lambda c: epi.EpidemicEnv(0.26 * c.beta_mul, 0.12, 0.008 * c.ifr_mul, c.noise, 5.0)
    a, _, _ = mk(cn).step(s0, epi.Action.from_id(0))
    b, _, _ = mk(jp).step(s0, epi.Action.from_id(0))
    assert a.i != b.i


def test_gaussian_streams_on_the_device_equal_the_host_ones(epi):
    """The per-country noise streams of a heterogeneous sweep are drawn on the device, a thread per stream (polar
    method on StrictMath.log / sqrt): every draw equals the host generator's, which the oracle's own generator and
    the JDK's documented algorithm pin."""
    import ctypes as C
    streams, draws = 1500, 257
    tab = np.zeros((draws, streams))
    epi.check(epi.lib().epi_java_gaussian_table(C.c_int64(42), C.c_int32(streams), C.c_int32(draws),
                                                tab.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(0)))
    one = np.zeros(draws)
    for c in list(range(0, streams, 7)) + [streams - 1]:
        epi.check(epi.lib().epi_java_gaussians(C.c_int64(42 + c), C.c_int32(draws), one.ctypes.data_as(C.POINTER(C.c_double))))
        np.testing.assert_array_equal(tab[:, c], one)
    assert abs(tab.mean()) < 0.01 and abs(tab.std() - 1.0) < 0.01
    # negative and large seeds take the same path
    for seed0 in (-5, 2**40 + 3):
        t2 = np.zeros((9, 3))
        epi.check(epi.lib().epi_java_gaussian_table(C.c_int64(seed0), C.c_int32(3), C.c_int32(9),
                                                    t2.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(0)))
        for c in range(3):
            o = np.zeros(9)
            epi.check(epi.lib().epi_java_gaussians(C.c_int64(seed0 + c), C.c_int32(9), o.ctypes.data_as(C.POINTER(C.c_double))))
            np.testing.assert_array_equal(t2[:, c], o)


# ---------------------------------------------------------------- MLP.predict / actGreedy
def test_mlp_greedy_matches_oracle(epi):
    rng = np.random.default_rng(5)
    W = epi.mlp_init(128)
    oW = O.mlp_weights()
    for a, b in zip(W, oW):
        np.testing.assert_array_equal(a, b)
    states = []
    for _ in range(100):
        pop = int(rng.integers(1000, 10**9))
        i = int(rng.integers(0, pop // 3)); r = int(rng.integers(0, pop // 3))
        states.append(epi.State(pop - i - r, i, r, int(rng.integers(0, 1000)), int(rng.integers(0, 1000)),
                                float(rng.choice([1.0, 1e4, 3e5])), int(rng.integers(0, 300)), 300, 1))
    q, act = epi.mlp_greedy(states, W, 10.0)
    for k, st in enumerate(states):
        want = O.mlp_predict(np.array(st.features()), 10.0)
        np.testing.assert_array_equal(q[k], want)
        assert act[k] == int(np.argmax(want))


@pytest.mark.parametrize("hidden", [64, 128])
def test_mlp_large_batches_equal_small_ones(epi, hidden):
    """Batches of 128 states and more on one network of 64 or 128 hidden units take the tiled network kernel, smaller
    ones the unit-per-thread kernel: same Q-values and actions, bit for bit, a ragged last tile included; for the
    reference's width also against the oracle."""
    rng = np.random.default_rng(17 + hidden)
    W = epi.mlp_init(hidden)
    states = []
    for _ in range(333):
        pop = int(rng.integers(1000, 10**9))
        i = int(rng.integers(0, pop // 3)); r = int(rng.integers(0, pop // 3))
        states.append(epi.State(pop - i - r, i, r, int(rng.integers(0, 1000)), int(rng.integers(0, 1000)),
                                float(rng.choice([1.0, 1e4, 3e5])), int(rng.integers(0, 300)), 300, 1))
    q, act = epi.mlp_greedy(states, W, 10.0)
    for lo in range(0, len(states), 100):
        qs, acts = epi.mlp_greedy(states[lo:lo + 100], W, 10.0)
        np.testing.assert_array_equal(q[lo:lo + 100], qs)
        np.testing.assert_array_equal(act[lo:lo + 100], acts)
    assert np.abs(q).max() > 0
    # one to eight states take the latency-first kernel (weights requested at once, sums relayed through the parts)
    for cnt, lo in ((1, 0), (1, 77), (2, 10), (3, 20), (4, 30), (5, 40), (7, 50), (8, 60)):
        qs, acts = epi.mlp_greedy(states[lo:lo + cnt], W, 10.0)
        np.testing.assert_array_equal(q[lo:lo + cnt], qs)
        np.testing.assert_array_equal(act[lo:lo + cnt], acts)
    if hidden == 128:
        for k in range(0, len(states), 7):
            np.testing.assert_array_equal(q[k], O.mlp_predict(np.array(states[k].features()), 10.0))


def test_unit_ops_keep_their_staging(epi):
    """The per-call entry points reuse device staging between calls (growing, shrinking and growing sizes again) and
    the cached network of epi_mlp_greedy_batch is replaced when any weight changes."""
    rng = np.random.default_rng(5)
    assert epi.lib().epi_set_device(0) == 0
    assert epi.lib().epi_set_device(99) != 0
    for n, B in ((1000, 8), (6, 1), (300, 40), (6, 1)):
        w = rng.random((B, n)); totals = rng.integers(0, 10**9, B).astype(np.int32)
        got = epi.EpidemicEnv.allocate_int_batch(totals, w)
        for b in range(B):
            np.testing.assert_array_equal(got[b], O.alloc_int(totals[b], w[b]))
    W = [x.copy() for x in epi.mlp_init(128)]
    st = [epi.State(9e5, 1e3 + k, 50, 3, 0, 1e4, k, 300, 1) for k in range(5)]
    q1, a1 = epi.mlp_greedy(st, W, 10.0)
    q2, _ = epi.mlp_greedy(st[:2], W, 10.0)
    np.testing.assert_array_equal(q2, q1[:2])
    W[2][3, 17] += 0.25                                   # one output weight changes: the cached network must not be reused
    q3, _ = epi.mlp_greedy(st, W, 10.0)
    assert not np.array_equal(q3[:, 3], q1[:, 3]) and np.array_equal(q3[:, 2], q1[:, 2])
    q4, _ = epi.mlp_greedy(st, W, 0.01)                   # so does the clamp
    assert np.abs(q4).max() <= 0.01


# ---------------------------------------------------------------- WorldToy.stepOne (WorldToy.scala:83-145)
def _world_pair(epi, countries, conn, batch, seed=3):
    n = len(countries)
    rng = np.random.default_rng(seed)
    base = epi.State(0, 0, 0, 0, 0, 10_000, 0, 120, 42)
    envs = [epi.EpidemicEnv(0.2 + 0.2 * rng.random(), 0.12, 0.008, 0.03 + 0.05 * rng.random(), 5.0, 0.3 * (k % 2),
                            c.name, 100 + k, profile=c) for k, c in enumerate(countries)]
    it = iter(envs)
    w = epi.WorldToy(epi.HyperParams(), base, 0, conn, make_env_for=lambda name, pop: next(it), validate=True,
                     countries=countries, batch=batch)
    ow = O.World([c.pop for c in countries], util.dense_of(conn, n), 0, util.oracle_state(base),
                 [O.EnvParams(e.base_beta, e.base_gamma, e.base_ifr, e.noise, e.reward_clip, e.risk_factor) for e in envs],
                 [util.prof_dict(e.country_profile) for e in envs], [e.seed for e in envs], validate=True)
    return w, ow, rng


# 300 and 1000 countries run the 8- and 16-slot variants of the dense kernel: states and S/I flows of every step
@pytest.mark.parametrize("n", [6, 70, 200, 300, 1000])
def test_world_step_given_actions(epi, n):
    from epidemicsimulator_amd import synth
    if n == 6:
        countries, conn = list(epi.ToyConn.C), epi.ToyConn.W
    else:
        countries, xy = synth.make_countries(n, 31 + n)
        conn = synth.gravity_dense(countries, xy)
    w, ow, rng = _world_pair(epi, countries, conn, batch=1)
    for step in range(25 if n <= 300 else 14):
        acts = rng.integers(0, 8, n)
        gauss = rng.normal(size=n)
        states, rewards = w.step_actions([acts], [gauss], 120)
        o_r, o_ms, o_mi = ow.step_given(acts, gauss, 120)
        o_states = ow.states()
        assert ow.error()[0] == 0
        for c in range(n):
            g = states[0][c]
            assert (g.s, g.i, g.r, g.d, g.v, g.hosp_cap, g.t) == o_states[c].tup(), "step %d country %d" % (step, c)
        np.testing.assert_array_equal(rewards[0], o_r)
        ms, mi = w.last_flows()
        np.testing.assert_array_equal(ms[0], o_ms)
        np.testing.assert_array_equal(mi[0], o_mi)
        assert int(ms.sum()) + int(mi.sum()) == 0      # MobilitySpec.scala:61-62


def test_world_batch_instances_are_independent(epi):
    from epidemicsimulator_amd import synth
    countries, xy = synth.make_countries(40, 5)
    conn = synth.gravity_dense(countries, xy)
    B = 5
    w, ow, rng = _world_pair(epi, countries, conn, batch=B)
    acts = rng.integers(0, 8, (B, 40)); gauss = rng.normal(size=(B, 40))
    states, rewards = w.step_actions(acts, gauss, 120)
    for b in range(B):
        ow.reset()
        o_r, _, _ = ow.step_given(acts[b], gauss[b], 120)
        o_states = ow.states()
        for c in range(40):
            g = states[b][c]
            assert (g.s, g.i, g.r, g.d, g.v, g.hosp_cap, g.t) == o_states[c].tup()
        np.testing.assert_array_equal(rewards[b], o_r)


def test_world_result_arrays_equal_copied(epi):
    """epi_world_result_buffers: results read in place from the handle's own arrays are the ones the copying
    path returns."""
    from epidemicsimulator_amd import synth
    countries, xy = synth.make_countries(30, 8)
    conn = synth.gravity_dense(countries, xy)
    B = 7
    wa, _, rng = _world_pair(epi, countries, conn, batch=B)
    wb, _, _ = _world_pair(epi, countries, conn, batch=B)
    st_b, rw_b = wb.result_arrays()
    assert st_b.shape == (B, 30) and rw_b.shape == (B, 30)
    for step in range(5):
        acts = rng.integers(0, 8, (B, 30)); gauss = rng.normal(size=(B, 30))
        st_a, rw_a = wa.step_arrays(acts, gauss, 120)
        if step == 3:     # caller-owned arrays still work on the same world
            got_s, got_r = wb.step_arrays(acts, gauss, 120)
        else:
            got_s, got_r = wb.step_arrays(acts, gauss, 120, st_b, rw_b)
            assert got_s is st_b
        np.testing.assert_array_equal(got_s, st_a)
        np.testing.assert_array_equal(got_r, rw_a)
    del st_b, rw_b, got_s, got_r
    wa.close(); wb.close()


@pytest.mark.parametrize("kind", ["toy", "dense30", "sparse900"])
def test_world_packed_states_equal_the_full_records(epi, kind):
    """epi_world_step_packed: the 32-byte state records hold the same numbers as the 64-byte ones of epi_world_step,
    field by field and step by step, read in place or copied -- on the toy world (the kernel writes the records
    itself), on a dense world and on a sparse one (the unpack kernel writes them)."""
    from epidemicsimulator_amd import synth
    if kind == "toy":
        countries, conn, B = list(epi.ToyConn.C), epi.ToyConn.W, 37
    elif kind == "dense30":
        countries, xy = synth.make_countries(30, 8)
        conn, B = synth.gravity_dense(countries, xy), 5
    else:
        countries, xy = synth.make_countries(900, 21)
        conn, B = synth.gravity_csr(countries, xy, mean_degree=9.0), 3
    n = len(countries)
    wa, _, rng = _world_pair(epi, countries, conn, batch=B)
    wb, _, _ = _world_pair(epi, countries, conn, batch=B)
    pk = wb.packed_array()
    _, rw_b = wb.result_arrays()
    assert pk.shape == (B, n) and pk.dtype.itemsize == 32
    for step in range(6):
        acts = rng.integers(0, 8, (B, n)); gauss = rng.normal(size=(B, n))
        st_a, rw_a = wa.step_arrays(acts, gauss, 120)
        if step == 2:
            got_s, got_r = wb.step_arrays_packed(acts, gauss, 120)
        else:
            got_s, got_r = wb.step_arrays_packed(acts, gauss, 120, pk, rw_b)
            assert got_s is pk
        for f in ("s", "i", "r", "d", "v", "t", "hosp_cap"):
            np.testing.assert_array_equal(got_s[f], st_a[f], err_msg=f)
        np.testing.assert_array_equal(got_r, rw_a)
    assert int(st_a["i"].sum()) > 0
    del pk, rw_b, got_s, got_r
    wa.close(); wb.close()


@pytest.mark.parametrize("kind", ["toy", "dense30", "csr900"])
def test_world_step_without_outputs_leaves_the_same_states_on_the_device(epi, kind):
    """epi_world_step with NULL out_states / out_rewards (WorldToy.step_resident): nothing comes back per step, and the
    states the world holds afterwards (snapshotStates) and the flows of the last step are those of the same steps taken
    with outputs -- on the toy world (the kernel writes no records), a dense world and a sparse one (no unpack)."""
    from epidemicsimulator_amd import synth
    if kind == "toy":
        countries, conn, B = list(epi.ToyConn.C), epi.ToyConn.W, 37
    elif kind == "dense30":
        countries, xy = synth.make_countries(30, 8)
        conn, B = synth.gravity_dense(countries, xy), 5
    else:
        countries, xy = synth.make_countries(900, 21)
        conn, B = synth.gravity_csr(countries, xy, mean_degree=9.0), 3
    n = len(countries)
    wa, _, rng = _world_pair(epi, countries, conn, batch=B)
    wb, _, _ = _world_pair(epi, countries, conn, batch=B)
    for step in range(7):
        acts = rng.integers(0, 8, (B, n)); gauss = rng.normal(size=(B, n))
        st_a, _ = wa.step_arrays(acts, gauss, 120)
        assert wb.step_resident(acts, gauss, 120) is None
    snap_a, snap_b = wa.snapshot_states(), wb.snapshot_states()
    if B == 1:
        snap_a, snap_b = [snap_a], [snap_b]
    for b in range(B):
        for c in range(n):
            assert snap_a[b][c] == snap_b[b][c], (b, c)
            assert snap_b[b][c].i == st_a["i"][b, c]
    fa, fb = wa.last_flows(), wb.last_flows()
    np.testing.assert_array_equal(fa[0], fb[0])
    np.testing.assert_array_equal(fa[1], fb[1])
    assert int(st_a["i"].sum()) > 0
    wa.close(); wb.close()


def test_world_step_one_and_greedy_follow_reference_streams(epi):
    """stepOne / stepOneGreedy with the JVM-compatible agent and env streams (EvalDeterminismSpec.scala
    inputs: base tMax=120, beta 0.26, noise 0.02), checked against the oracle's WorldToy."""
    hp = epi.HyperParams(learn_every=2**31 - 1)
    base = epi.State(0, 0, 0, 0, 0, 10_000, 0, 120, 42)
    mk = lambda: epi.EpidemicEnv(0.26, 0.12, 0.008, 0.02, hp.reward_clip)
    n = 6
    env = O.EnvParams(0.26, 0.12, 0.008, 0.02, 5.0, 0.0)
    dflt = util.prof_dict(epi.Geo.CountryProfiles.get_profile("Unknown"))
    for greedy in (True, False):
        w = epi.WorldToy(hp, base, 0, epi.ToyConn.W, make_env=mk, validate=False)
        w2 = epi.WorldToy(hp, base, 0, epi.ToyConn.W, make_env=mk, validate=False)
        ohp = O.default_hyper(); ohp.learn_every = 2**31 - 1
        ow = O.World([c.pop for c in epi.ToyConn.C], epi.ToyConn.W, 0, util.oracle_state(base), [env] * n, [dflt] * n,
                     [42] * n, hp=ohp, validate=False)
        a_series, b_series = [], []
        for k in range(100):
            out = w.step_one_greedy(100) if greedy else w.step_one(100)
            out2 = w2.step_one_greedy(100) if greedy else w2.step_one(100)
            o_r = ow.step(100, greedy)
            o_states = ow.states()
            for c in range(n):
                g = out[c][1]
                assert (g.s, g.i, g.r, g.d, g.v, g.hosp_cap, g.t) == o_states[c].tup(), "step %d" % k
                assert out[c][2] == o_r[c]
            np.testing.assert_array_equal([nd.last_a for nd in w.nodes], ow.last_actions())
            a_series.append(out[0][1].i); b_series.append(out2[0][1].i)
        assert a_series == b_series       # EvalDeterminismSpec.scala:37-45


def oracle_best_beta():
    """Best beta of configuration 1's surface according to the oracle's own sweep."""
    from epidemicsimulator_amd import synth
    wl = synth.workload("c1")
    want = util.oracle_grid(wl, synth.base_state(wl["steps"]))
    return wl["betas"][want["best_idx"] // len(wl["travel"])]


def _c5_world(epi, batch, beta=None):
    """Configuration 5: `batch` instances of the six-country toy world with the heterogeneous env parameters of
    Calibrate.scala:34-51 at the calibrated beta -- the first minimum of configuration 1's surface, computed by the
    engine's sweep (bench.py's c5 workload takes it from the same place)."""
    from epidemicsimulator_amd import synth
    if beta is None:
        beta, _ = synth.calibrated_toy_point()
        assert beta == oracle_best_beta()
    countries = list(epi.ToyConn.C)
    names = [c.name for c in countries]

    def make_env_for(name, pop):
        p = epi.Geo.CountryProfiles.get_profile(name)
        fb = beta * (1.0 + (p.population_density / 1000.0) * 0.5) * (3.0 - p.healthcare_capacity * 2.0)
        fi = 0.008 * (1.0 + p.age_distribution_elderly * 2.0)
        fn = 0.05 * (1.0 + (1.0 - p.economic_resilience))
        return epi.EpidemicEnv(fb, 0.12, fi, fn, 5.0, 0.0, name, names.index(name) + 42)

    base = epi.State(1_000_000, 0, 0, 0, 0, 10_000, 0, 300, 1)
    w = epi.WorldToy(epi.HyperParams(), base, 0, epi.ToyConn.W, make_env_for=make_env_for, validate=True, batch=batch)
    envs = w._envs

    def oracle_world():
        return O.World([c.pop for c in countries], epi.ToyConn.W, 0, util.oracle_state(base),
                       [O.EnvParams(e.base_beta, e.base_gamma, e.base_ifr, e.noise, e.reward_clip, e.risk_factor) for e in envs],
                       [util.prof_dict(e.country_profile) for e in envs], [e.seed for e in envs], validate=True)
    return w, oracle_world, envs


@pytest.mark.parametrize("own_noise", [True, False])
def test_world_c5_batch_4096(epi, own_noise):
    """Configuration 5 at its workload: 4,096 instances, random int32[B][6] actions, 24 steps.  Mass conservation
    and net-zero flows for every instance each step; states, rewards and flows == the oracle's WorldToy for a sample
    of instances spread over the batch (first, last, and the four episodes of several waves).  own_noise: the
    engine's JVM-compatible env streams (gauss = NULL) against the same draws handed to the oracle."""
    B, n, steps = 4096, 6, 24
    w, oracle_world, envs = _c5_world(epi, B)
    rng = np.random.default_rng(4096 + own_noise)
    sample = [0, 1, 2, 3, 4, 7, 1021, 2048, 2049, 2050, 2051, 3333, 4092, 4093, 4094, 4095]
    worlds = {b: oracle_world() for b in sample}
    streams = [O.jrand_gaussians(e.seed, steps) for e in envs]
    pop_total = sum(c.pop for c in epi.ToyConn.C)
    for step in range(steps):
        acts = rng.integers(0, 8, (B, n)).astype(np.int32)
        if own_noise:
            gauss_in = None
            gauss = np.tile(np.array([streams[c][step] for c in range(n)]), (B, 1))
        else:
            gauss_in = gauss = rng.normal(size=(B, n))
        states, rewards = w.step_arrays(acts, gauss_in, 300)
        ms, mi = w.last_flows()
        total = states["s"] + states["i"] + states["r"] + states["d"] + states["v"]
        assert (total.sum(axis=1) == pop_total).all(), "step %d: world mass" % step
        assert ((ms.sum(axis=1) + mi.sum(axis=1)) == 0).all(), "step %d: net flow" % step
        assert (states["t"] == min(2 * (step + 1), 300)).all()
        for b, ow in worlds.items():
            o_r, o_ms, o_mi = ow.step_given(acts[b], gauss[b], 300)
            assert ow.error()[0] == 0
            want = ow.states()
            for f in ("s", "i", "r", "d", "v", "hosp_cap", "t"):
                np.testing.assert_array_equal(states[b][f], [getattr(x, f) for x in want], err_msg="step %d instance %d %s" % (step, b, f))
            np.testing.assert_array_equal(rewards[b], o_r)
            np.testing.assert_array_equal(ms[b], o_ms)
            np.testing.assert_array_equal(mi[b], o_mi)
    # instances fed the same inputs stay equal; different actions separate them
    assert not np.array_equal(states["i"][0], states["i"][1])
    w.close()
    for ow in worlds.values():
        ow.close()


def test_grid_learning_inside_rollout(epi):
    """WorldToy.stepOne lets the agents learn once gStep - lastLearnStep >= learnEvery and 4 * batchSize transitions
    are stored (WorldToy.scala:150-167), and Calibrate.rollout steps its world with exactly that stepOne.  A rollout
    in which an argmax-Q step follows the first learn call consults trained weights: the engine then runs one agent
    per (grid point, country) and must equal the oracle sweep whose worlds learn; one whose agents learn but never
    consult the network afterwards equals the plain oracle (which flags the learn step with err 3)."""
    from epidemicsimulator_amd import synth
    wl = synth.small_workload(20, 2, 2, 60, seed=5)
    base = synth.base_state(60)
    hp = epi.HyperParams(batch_size=8, learn_every=8, epsilon_start=0.5, epsilon_end=0.5)   # argmax Q every other step
    got = util.engine_grid(wl, base, True, hp=hp)
    want = util.oracle_grid_learn(wl, base, hp)
    util.assert_grid_equal(got, want)
    plain = util.oracle_grid(wl, base)   # default epsilon: no argmax-Q step within 60 steps, so a different trajectory
    assert (plain["peak"] != want["peak"]).any()
    hp = epi.HyperParams(batch_size=8, learn_every=8, epsilon_start=1.0, epsilon_end=1.0)   # explores every step
    got = util.engine_grid(wl, base, True, hp=hp)
    ohp = O.default_hyper(); ohp.batch_size = 8; ohp.learn_every = 8; ohp.epsilon_start = ohp.epsilon_end = 1.0
    n = 20
    ta = np.array([wl["t_arrive"][c] for c in range(n)]); tp = np.array([wl["t_peak"][c] for c in range(n)])
    ones = np.ones(n, dtype=np.uint8)
    want = O.grid_surface([c.pop for c in wl["countries"]], wl["conn"], 0, util.oracle_state(base),
                          [util.prof_dict(p) for p in wl["profiles"]], 60, wl["betas"], wl["travel"], ta, ones, tp, ones, True, hp=ohp)
    assert want["err"] == 3      # the oracle saw the learn trigger
    np.testing.assert_array_equal(got["arrival"], want["arrival"])
    np.testing.assert_array_equal(got["peak"], want["peak"])
    np.testing.assert_allclose(got["loss"], want["loss"], rtol=1e-12, atol=0)
    # the reference's defaults: 300 steps < 4 * 128 transitions, never refused
    util.engine_grid(dict(wl, steps=300), synth.base_state(300), True)


# ---------------------------------------------------------------- Calibrate.gridSurface
def test_grid_toy_default_config(epi):
    """Configuration 1: the toy country set, 4x5 grid, 300 steps, heterogeneous, fallback targets
    (App.scala:97-107) -- every arrival/peak bit-exact, losses within 1e-12."""
    from epidemicsimulator_amd import synth
    wl = synth.workload("c1")
    base = synth.base_state(300)
    got = util.engine_grid(wl, base)
    want = util.oracle_grid(wl, base)
    util.assert_grid_equal(got, want)


def test_grid_calibration_spec(epi):
    """CalibrationSpec.scala:6-16: empty targets, homogeneous, 60 steps: rows non-empty, loss exactly 0."""
    hp, cfg = epi.HyperParams(), epi.TrainConfig(epochs=1, steps_per_epoch=60)
    base = epi.State(1_000_000, 0, 0, 0, 0, 10_000, 0, 60, 3)
    rows, (best_loss, best) = epi.Calibrate.grid_surface(hp, cfg, base, 0, epi.Calibrate.Targets({}, {}), [0.24, 0.26],
                                                         [0.8, 1.0], validate=False, heterogeneous=False)
    assert len(rows) == 4 and all(r[2] == 0.0 for r in rows)
    assert best_loss == 0.0 and best == (0.24, 0.8)


@pytest.mark.parametrize("n,nb,nt,steps,het", [(70, 3, 4, 130, True), (200, 2, 3, 40, True), (200, 2, 2, 40, False),
                                               (130, 2, 2, 30, True), (1000, 1, 2, 12, True)])
def test_grid_synthetic_matches_oracle(epi, n, nb, nt, steps, het):
    from epidemicsimulator_amd import synth
    wl = synth.small_workload(n, nb, nt, steps, seed=77 + n)
    base = synth.base_state(steps)
    got = util.engine_grid(wl, base, het)
    want = util.oracle_grid(wl, base, het)
    util.assert_grid_equal(got, want)


def test_grid_slabs_equal_whole(epi):
    """A grid run in slabs (the multi-GPU partition) gives the same surface as one run."""
    from epidemicsimulator_amd import synth
    wl = synth.small_workload(64, 4, 4, 25, seed=9)
    base = synth.base_state(25)
    eng = epi.GridEngine(epi.HyperParams(), 25, base, 0, [c.pop for c in wl["countries"]], wl["conn"], wl["profiles"],
                         wl["betas"], wl["travel"], wl["t_arrive"], wl["t_peak"], True)
    whole = eng.run()
    parts = [eng.run(g0, 4) for g0 in range(0, 16, 4)]
    np.testing.assert_array_equal(np.concatenate([p["loss"] for p in parts]), whole["loss"])
    np.testing.assert_array_equal(np.concatenate([p["arrival"] for p in parts]), whole["arrival"])
    eng.close()


def test_grid_strided_shares_equal_whole(epi):
    """The interleaved split of the multi-GPU sweep: rank r's points r, r + ranks, ... give the same losses and
    metrics as the same points of one whole run, for a rank count that does not divide the grid."""
    from epidemicsimulator_amd import synth
    from epidemicsimulator_amd.dist import strided_bounds
    wl = synth.small_workload(40, 5, 7, 30, seed=21)
    eng = util.engine_for(wl, synth.base_state(30)) if hasattr(util, "engine_for") else None
    if eng is None:
        eng = epi.GridEngine(epi.HyperParams(), wl["steps"], synth.base_state(30), wl["seed_idx"], [c.pop for c in wl["countries"]],
                             wl["conn"], wl["profiles"], wl["betas"], wl["travel"], wl["t_arrive"], wl["t_peak"], True)
    whole = eng.run()
    G = 35
    for ranks in (3, 8):
        for r in range(ranks):
            g0, stride, cnt, _ = strided_bounds(G, ranks, r)
            part = eng.run_strided(g0, stride, cnt)
            idx = np.arange(g0, G, stride)
            np.testing.assert_array_equal(part["loss"], whole["loss"][idx])
            np.testing.assert_array_equal(part["arrival"], whole["arrival"][idx])
            np.testing.assert_array_equal(part["peak"], whole["peak"][idx])
    with pytest.raises(epi.EpiError):
        eng.run_strided(3, 8, 5)     # 3 + 4*8 = 35 is outside


def test_grid_csr_input_equals_dense(epi):
    from epidemicsimulator_amd import synth
    countries, xy = synth.make_countries(300, 21)
    csr = synth.gravity_csr(countries, xy, 21, mean_degree=6.0)
    wl = synth.small_workload(300, 2, 2, 30, seed=21)
    wl["countries"], wl["profiles"], wl["conn"] = countries, countries, csr
    base = synth.base_state(30)
    got = util.engine_grid(wl, base)
    want = util.oracle_grid(wl, base)
    util.assert_grid_equal(got, want)


def test_grid_assert_state_failure(epi):
    """Validation.assertState on the initial states (WorldToy.scala:53): base.r != 0 breaks the mass invariant."""
    from epidemicsimulator_amd import synth
    wl = synth.small_workload(8, 1, 1, 5)
    base = synth.base_state(5).copy(r=3.0)
    with pytest.raises(epi.IllegalArgumentException):
        util.engine_grid(wl, base)


def test_grid_full_size_properties(epi):
    """Configuration 2 at full size (200 countries, 32x32, T=200) through size-independent properties:
    arrival in [0, T], peak >= first-step I, finite losses, seed country arrives at step 0, and a sample
    of grid points re-run as single-point slabs reproduces the surface bit for bit."""
    from epidemicsimulator_amd import synth
    wl = synth.workload("c2")
    base = synth.base_state(wl["steps"])
    eng = epi.GridEngine(epi.HyperParams(), wl["steps"], base, 0, [c.pop for c in wl["countries"]], wl["conn"],
                         wl["profiles"], wl["betas"], wl["travel"], wl["t_arrive"], wl["t_peak"], True)
    out = eng.run()
    assert np.isfinite(out["loss"]).all() and (out["loss"] >= 0).all()
    assert (out["arrival"] >= 0).all() and (out["arrival"] <= wl["steps"]).all()
    assert (out["arrival"][:, 0] == 0).all()
    assert (out["peak"] >= 0).all()
    for g in (0, 517, 1023):
        one = eng.run(g, 1)
        assert one["loss"][0] == out["loss"][g]
        np.testing.assert_array_equal(one["arrival"][0], out["arrival"][g])
        np.testing.assert_array_equal(one["peak"][0], out["peak"][g])
    eng.close()
    # the whole full-size surface (all 1,024 grid points) against the oracle; the travel scales lie on either side of the
    # point (row sum x ts = 0.2) where the cap of ToyConn.scaleRow starts to change the seed row (ToyConn.scala:63-66)
    pts = [(b, t) for b in range(32) for t in range(32)]
    assert not util.seed_row_cap_binds(wl, 22) and util.seed_row_cap_binds(wl, 27) and util.seed_row_cap_binds(wl, 31)
    want = util.oracle_points(wl, base, pts)
    flat = [b * 32 + t for b, t in pts]
    np.testing.assert_array_equal(out["arrival"][flat], want["arrival"])
    np.testing.assert_array_equal(out["peak"][flat], want["peak"])
    np.testing.assert_allclose(out["loss"][flat], want["loss"], rtol=1e-12, atol=0)


def test_grid_full_size_c3(epi):
    """Configuration 3 at full size (1,000 countries, 64x64, T=200): the same size-independent properties, slabs
    of the flat grid as the 8-GPU run cuts them, and 64 grid points against the oracle."""
    from epidemicsimulator_amd import synth
    from epidemicsimulator_amd.dist import slab_bounds
    wl = synth.workload("c3")
    base = synth.base_state(wl["steps"])
    eng = epi.GridEngine(epi.HyperParams(), wl["steps"], base, 0, [c.pop for c in wl["countries"]], wl["conn"],
                         wl["profiles"], wl["betas"], wl["travel"], wl["t_arrive"], wl["t_peak"], True)
    out = eng.run(want_metrics=False)
    assert np.isfinite(out["loss"]).all() and (out["loss"] >= 0).all() and len(out["loss"]) == 4096
    g0, cnt, per = slab_bounds(4096, 8, 5)     # rank 5 of 8: grid points 2560..3071
    assert (g0, cnt, per) == (2560, 512, 512)
    slab = eng.run(g0, cnt)
    np.testing.assert_array_equal(slab["loss"], out["loss"][g0:g0 + cnt])
    assert (slab["arrival"] >= 0).all() and (slab["arrival"] <= wl["steps"]).all() and (slab["arrival"][:, 0] == 0).all()
    eng.close()
    # 64 grid points against the oracle (sixteen at a time, one thread each): eight beta rows, first and last among them,
    # x eight travel scales -- first, last, and scales below and above the point where ToyConn.scaleRow's 0.2 row-sum cap
    # binds (ToyConn.scala:63-66)
    pts = [(b, t) for b in (0, 9, 21, 30, 37, 46, 55, 63) for t in (0, 12, 27, 41, 50, 56, 60, 63)]
    assert not util.seed_row_cap_binds(wl, 50) and util.seed_row_cap_binds(wl, 56) and util.seed_row_cap_binds(wl, 63)
    want = util.oracle_points(wl, base, pts)
    eng = epi.GridEngine(epi.HyperParams(), wl["steps"], base, 0, [c.pop for c in wl["countries"]], wl["conn"],
                         wl["profiles"], wl["betas"], wl["travel"], wl["t_arrive"], wl["t_peak"], True)
    for k, (b, t) in enumerate(pts):
        got = eng.run(b * 64 + t, 1)
        np.testing.assert_array_equal(got["arrival"][0], want["arrival"][k])
        np.testing.assert_array_equal(got["peak"][0], want["peak"][k])
        np.testing.assert_allclose(got["loss"][0], want["loss"][k], rtol=1e-12, atol=0)
        assert got["loss"][0] == out["loss"][b * 64 + t]
    eng.close()


def test_grid_full_size_c4(epi):
    """Configuration 4 at full size (20,000 patches, CSR, 16x16, T=200): properties and slab invariance on the
    whole sweep, and 64 of the 256 grid points over all 200 steps against the oracle's CSR world."""
    from epidemicsimulator_amd import synth
    wl = synth.workload("c4")
    n = len(wl["countries"])
    base = synth.base_state(wl["steps"])
    eng = epi.GridEngine(epi.HyperParams(), wl["steps"], base, 0, [c.pop for c in wl["countries"]], wl["conn"],
                         wl["profiles"], wl["betas"], wl["travel"], wl["t_arrive"], wl["t_peak"], True)
    out = eng.run()
    assert np.isfinite(out["loss"]).all() and len(out["loss"]) == 256
    assert (out["arrival"] >= 0).all() and (out["arrival"] <= wl["steps"]).all() and (out["arrival"][:, 0] == 0).all()
    assert ((out["arrival"] < wl["steps"]).sum(axis=1) > n // 2).all(), "the infection reaches most patches in 200 steps"
    for g in (0, 100, 255):
        one = eng.run(g, 1)
        assert one["loss"][0] == out["loss"][g]
        np.testing.assert_array_equal(one["arrival"][0], out["arrival"][g])
        np.testing.assert_array_equal(one["peak"][0], out["peak"][g])
    eng.close()
    # a quarter of the grid (eight beta rows, first and last among them, x eight travel scales: 64 points) for all 200
    # steps against the oracle's CSR world
    pts = [(b, t) for b in (0, 2, 5, 7, 9, 11, 13, 15) for t in (0, 1, 3, 6, 8, 10, 13, 15)]
    want = util.oracle_points(wl, base, pts, sparse=True)
    flat = [b * 16 + t for b, t in pts]
    np.testing.assert_array_equal(out["arrival"][flat], want["arrival"])
    np.testing.assert_array_equal(out["peak"][flat], want["peak"])
    np.testing.assert_allclose(out["loss"][flat], want["loss"], rtol=1e-12, atol=0)
    # and one sweep seeded in the graph's longest row (more than 32 stored weights: a row the lane-per-row blocks do
    # not hold, so its travel-scale variants take the one-wave-per-row path), two travel scales, cap binding in one
    rp = wl["conn"][0]
    deg = np.diff(rp)
    hub = int(np.argmax(deg))
    assert deg[hub] > 32, "the 20,000-patch graph has a row beyond the 32 lane slots"
    wl3 = dict(wl, seed_idx=hub)
    pts3 = [(4, 2), (11, 14)]
    want3 = util.oracle_points(wl3, base, pts3, sparse=True)
    eng3 = epi.GridEngine(epi.HyperParams(), wl["steps"], base, hub, [c.pop for c in wl["countries"]], wl["conn"],
                          wl["profiles"], wl["betas"], wl["travel"], wl["t_arrive"], wl["t_peak"], True)
    for k, (b, t) in enumerate(pts3):
        got = eng3.run(b * 16 + t, 1)
        np.testing.assert_array_equal(got["arrival"][0], want3["arrival"][k])
        np.testing.assert_array_equal(got["peak"][0], want3["peak"][k])
        np.testing.assert_allclose(got["loss"][0], want3["loss"][k], rtol=1e-12, atol=0)
    eng3.close()


# ---------------------------------------------------------------- sparse (CSR) worlds, n > 1024
def _sparse_workload(n, deg, nb, nt, steps, seed):
    from epidemicsimulator_amd import synth
    countries, xy = synth.make_countries(n, seed)
    wl = synth.small_workload(8, nb, nt, steps, seed=seed)
    wl["countries"], wl["profiles"] = countries, countries
    wl["conn"] = synth.gravity_csr(countries, xy, seed, mean_degree=deg)
    wl["t_arrive"], wl["t_peak"] = synth.targets(n, steps, seed)
    return wl


@pytest.mark.parametrize("n,deg,nb,nt,steps,eps", [(1300, 6.0, 1, 2, 14, 1.0), (1100, 14.0, 2, 1, 10, 0.4), (2100, 3.0, 1, 1, 8, 1.0),
                                                   (1100, 30.0, 1, 2, 8, 1.0)])   # last: many rows beyond 32 stored weights
def test_grid_sparse_matches_oracle(epi, n, deg, nb, nt, steps, eps):
    """Worlds wider than the dense kernel run on the CSR kernel; eps < 1 makes argmax-Q steps frequent."""
    from epidemicsimulator_amd import synth
    wl = _sparse_workload(n, deg, nb, nt, steps, seed=300 + n)
    base = synth.base_state(steps)
    hp = epi.HyperParams(epsilon_start=eps, epsilon_end=eps)
    got = util.engine_grid(wl, base, True, hp=hp)
    ohp = O.default_hyper(); ohp.epsilon_start = eps; ohp.epsilon_end = eps
    nn = len(wl["countries"])
    ta = np.zeros(nn); tp = np.zeros(nn); ha = np.ones(nn, dtype=np.uint8)
    for c, v in wl["t_arrive"].items():
        ta[c] = v
    for c, v in wl["t_peak"].items():
        tp[c] = v
    want = O.grid_surface([c.pop for c in wl["countries"]], util.dense_of(wl["conn"], nn), 0, util.oracle_state(base),
                          [util.prof_dict(p) for p in wl["profiles"]], steps, wl["betas"], wl["travel"], ta, ha, tp, ha, True, hp=ohp)
    util.assert_grid_equal(got, want)


@pytest.mark.parametrize("deg,lo,hi,seed_idx", [(4.0, 1, 16, 0), (9.0, 17, 24, 0), (15.0, 25, 32, 0), (9.0, 17, 24, 777),
                                                (15.0, 25, 32, 1150), (26.0, 33, 64, 0)])
def test_grid_sparse_kernel_variants(epi, deg, lo, hi, seed_idx):
    """One case per compiled row width of the sparse kernel (16, 24 and 32 slots per lane) and for long rows,
    with the seed row in the first block and in a later one.  The longest row decides the variant; the
    graph is regenerated until it falls in the wanted range."""
    from epidemicsimulator_amd import synth
    n, steps = 1216, 12
    for seed in range(500, 540):
        wl = _sparse_workload(n, deg, 1, 2, steps, seed=seed)
        rp = wl["conn"][0]
        lane_max = int(np.max(np.where(np.diff(rp) <= 32, np.diff(rp), 0)))
        longest = int(np.max(np.diff(rp)))
        if lo <= (longest if hi > 32 else lane_max) <= hi and (hi > 32 or longest <= 32 or deg >= 15.0):
            break
    else:
        pytest.skip("no graph with the wanted longest row")
    if hi > 32:  # make the seed row itself one of the long rows
        seed_idx = int(np.argmax(np.diff(rp)))
    wl["seed_idx"] = seed_idx
    base = synth.base_state(steps)
    got = util.engine_grid(wl, base, True)
    ta = np.zeros(n); tp = np.zeros(n); ha = np.ones(n, dtype=np.uint8)
    for c, v in wl["t_arrive"].items():
        ta[c] = v
    for c, v in wl["t_peak"].items():
        tp[c] = v
    want = O.grid_surface([c.pop for c in wl["countries"]], util.dense_of(wl["conn"], n), seed_idx, util.oracle_state(base),
                          [util.prof_dict(p) for p in wl["profiles"]], steps, wl["betas"], wl["travel"], ta, ha, tp, ha, True)
    util.assert_grid_equal(got, want)
    assert (got["arrival"] < steps).sum() > 2 * 3, "the infection should have left the seed patch"


@pytest.mark.parametrize("kind", ["all_long", "isolated", "mixed_equal_weights", "hubs"])
def test_grid_sparse_odd_graphs(epi, kind):
    """Sparse graphs the blocked layout has to survive: every row longer than a lane holds (no blocks at all),
    rows without any stored weight, rows of identical weights (every fraction ties), and hub rows of up to 256
    stored weights, the seed among them."""
    from epidemicsimulator_amd import synth
    n, steps = 1100, 10
    countries, _ = synth.make_countries(n, 41)
    rng = np.random.default_rng(7)
    rowptr = [0]; col = []; val = []
    for i in range(n):
        if kind == "all_long":
            deg = 40
        elif kind == "hubs":   # a few rows of up to 256 stored weights (two and four register slots per lane)
            deg = {0: 250, 5: 100, 64: 65, 700: 129, 1099: 256}.get(i, int(rng.integers(1, 20)))
        elif kind == "isolated":
            deg = 0 if (i % 3 == 1 and i != 0) else int(rng.integers(1, 12))
        else:
            deg = int(rng.integers(1, 30))
        nb = np.sort(rng.choice(np.delete(np.arange(n), i), size=deg, replace=False))
        w = np.full(deg, 0.1 / max(deg, 1)) if kind == "mixed_equal_weights" else rng.random(deg) * (0.12 / max(deg, 1))
        col.extend(nb.tolist()); val.extend(w.tolist()); rowptr.append(len(col))
    wl = synth.small_workload(8, 1, 2, steps, seed=9)
    wl["countries"], wl["profiles"] = countries, countries
    wl["conn"] = (np.array(rowptr, dtype=np.int32), np.array(col, dtype=np.int32), np.array(val))
    wl["t_arrive"], wl["t_peak"] = synth.targets(n, steps, 9)
    base = synth.base_state(steps)
    got = util.engine_grid(wl, base, True)
    want = util.oracle_grid(wl, base)
    assert want["err"] == 0
    util.assert_grid_equal(got, want)


def test_world_sparse_step_given_actions(epi):
    from epidemicsimulator_amd import synth
    n = 1200
    countries, xy = synth.make_countries(n, 77)
    conn = synth.gravity_csr(countries, xy, 77, mean_degree=9.0)
    w, ow, rng = _world_pair(epi, countries, conn, batch=2)
    for step in range(6):
        acts = rng.integers(0, 8, (2, n)); gauss = rng.normal(size=(2, n))
        acts[1], gauss[1] = acts[0], gauss[0]
        states, rewards = w.step_actions(acts, gauss, 120)
        o_r, o_ms, o_mi = ow.step_given(acts[0], gauss[0], 120)
        o_states = ow.states()
        assert ow.error()[0] == 0
        for b in range(2):
            for c in range(n):
                g = states[b][c]
                assert (g.s, g.i, g.r, g.d, g.v, g.hosp_cap, g.t) == o_states[c].tup(), "step %d country %d" % (step, c)
            np.testing.assert_array_equal(rewards[b], o_r)
        ms, mi = w.last_flows()
        np.testing.assert_array_equal(ms[0], o_ms)
        np.testing.assert_array_equal(mi[1], o_mi)


@pytest.mark.parametrize("kind", ["gravity", "equal", "tiny_weights"])
def test_grid_dense_graph_above_the_dense_kernel(epi, kind):
    """A dense matrix for more than 1,024 countries: every row stores 1,099 weights and is streamed by one
    wave (seed row included, rescaled per travel value).  Gravity weights, identical weights (every
    fraction ties down to column order), and one heavy column among weights eight orders of magnitude smaller."""
    from epidemicsimulator_amd import synth
    n, steps = 1100, 6
    countries, xy = synth.make_countries(n, 5)
    if kind == "gravity":
        dense = synth.gravity_dense(countries, xy)
    elif kind == "equal":
        dense = np.full((n, n), 0.08 / (n - 1)); np.fill_diagonal(dense, 0.0)
    else:
        rng = np.random.default_rng(3)
        dense = rng.random((n, n)) * 1e-9; np.fill_diagonal(dense, 0.0)
        dense[:, 7] += 0.05 / n; dense[7, 7] = 0.0
    wl = synth.small_workload(8, 1, 2, steps, seed=12)
    wl["countries"], wl["profiles"], wl["conn"] = countries, countries, dense
    wl["seed_idx"] = 3
    wl["t_arrive"], wl["t_peak"] = synth.targets(n, steps, 12)
    base = synth.base_state(steps)
    got = util.engine_grid(wl, base, True)
    want = util.oracle_grid(wl, base)
    assert want["err"] == 0
    util.assert_grid_equal(got, want)


def test_grid_sparse_wide_hub_rows(epi):
    """CSR rows above 256 stored weights among ordinary rows, with lengths on and off multiples of 64."""
    from epidemicsimulator_amd import synth
    n, steps = 1500, 8
    countries, _ = synth.make_countries(n, 43)
    rng = np.random.default_rng(11)
    rowptr = [0]; col = []; val = []
    for i in range(n):
        deg = {0: 257, 9: 320, 640: 1499, 1499: 1000}.get(i, int(rng.integers(1, 20)))
        nb = np.sort(rng.choice(np.delete(np.arange(n), i), size=deg, replace=False))
        w = np.full(deg, 0.1 / deg) if i == 9 else rng.random(deg) * (0.12 / deg)
        col.extend(nb.tolist()); val.extend(w.tolist()); rowptr.append(len(col))
    wl = synth.small_workload(8, 1, 2, steps, seed=9)
    wl["countries"], wl["profiles"] = countries, countries
    wl["conn"] = (np.array(rowptr, dtype=np.int32), np.array(col, dtype=np.int32), np.array(val))
    wl["seed_idx"] = 640
    wl["t_arrive"], wl["t_peak"] = synth.targets(n, steps, 9)
    base = synth.base_state(steps)
    got = util.engine_grid(wl, base, True)
    want = util.oracle_grid(wl, base)
    assert want["err"] == 0
    util.assert_grid_equal(got, want)
    assert (got["arrival"] < steps).sum() > 2 * 3


# ---------------------------------------------------------------- edge cases
def test_grid_empty_and_degenerate_axes(epi):
    """Empty grids (gridSurface returns no rows, bestLoss = Double.MaxValue, bestPair = defaults,
    Calibrate.scala:96-98), zero-step episodes (arrival = series length 0, peakI(empty) = 0) and T = 1."""
    from epidemicsimulator_amd import synth
    hp = epi.HyperParams()
    base = epi.State(1_000_000, 0, 0, 0, 0, 10_000, 0, 60, 3)
    rows, (bl, best) = epi.Calibrate.grid_surface(hp, epi.TrainConfig(1, 60), base, 0, epi.Calibrate.Targets(), [], [0.8])
    assert rows == [] and bl == 1.7976931348623157e308 and best == (0.25, 0.8)
    rows, (bl, best) = epi.Calibrate.grid_surface(hp, epi.TrainConfig(1, 60), base, 0, epi.Calibrate.Targets(), [0.3], [])
    assert rows == [] and best == (0.3, 1.0)
    for steps in (0, 1, 2):
        wl = synth.small_workload(20, 2, 2, steps, seed=4)
        b = synth.base_state(max(steps, 1))
        got = util.engine_grid(wl, b)
        want = util.oracle_grid(wl, b)
        util.assert_grid_equal(got, want)
        if steps == 0:
            assert (got["arrival"] == 0).all() and (got["peak"] == 0).all()


@pytest.mark.parametrize("n,seed_idx", [(1, 0), (2, 1), (13, 12), (16, 15), (17, 3), (64, 63), (65, 64), (1024, 517)])
def test_grid_sizes_and_seed_positions(epi, n, seed_idx):
    """One country, wave-boundary sizes, the widest dense world (1,024) and seeds other than country 0."""
    from epidemicsimulator_amd import synth
    steps = 6 if n == 1024 else 25
    wl = synth.small_workload(n, 1, 2, steps, seed=1000 + n)
    if n == 1:
        wl["conn"] = np.zeros((1, 1))
    wl["seed_idx"] = seed_idx
    base = synth.base_state(steps)
    got = util.engine_grid(wl, base)
    want = util.oracle_grid(wl, base)
    util.assert_grid_equal(got, want)


def test_grid_isolated_and_uniform_rows(epi):
    """Rows with no outflow (all-zero weights), rows whose weights are all equal (every fraction ties:
    index order decides) and a seed row scaled to zero (travel = 0)."""
    from epidemicsimulator_amd import synth
    n, steps = 48, 30
    wl = synth.small_workload(n, 2, 3, steps, seed=61)
    W = np.array(wl["conn"])
    W[5, :] = 0.0                       # isolated origin
    W[:, 7] = 0.0                       # nobody travels to 7
    W[9, :] = 0.12 / (n - 1); W[9, 9] = 0.0   # identical weights
    W[11, :] = 0.0; W[11, 3] = 0.06; W[11, 40] = 0.06
    wl["conn"] = W
    wl["travel"] = [0.0, 1.0, 3.5]
    base = synth.base_state(steps)
    got = util.engine_grid(wl, base)
    want = util.oracle_grid(wl, base)
    util.assert_grid_equal(got, want)


def test_grid_other_hidden_sizes_and_epsilon(epi):
    """The argmax-Q branch with a smaller network and a low epsilon (many greedy steps)."""
    from epidemicsimulator_amd import synth
    wl = synth.small_workload(30, 2, 2, 40, seed=8)
    base = synth.base_state(40)
    for hidden, eps in ((64, 0.2), (16, 0.0), (128, 0.05)):
        hp = epi.HyperParams(hidden=hidden, epsilon_start=eps, epsilon_end=eps)
        got = util.engine_grid(wl, base, True, hp=hp)
        ohp = O.default_hyper(); ohp.hidden = hidden; ohp.epsilon_start = eps; ohp.epsilon_end = eps
        nn = 30
        ta = np.array([wl["t_arrive"][c] for c in range(nn)]); tp = np.array([wl["t_peak"][c] for c in range(nn)])
        ones = np.ones(nn, dtype=np.uint8)
        want = O.grid_surface([c.pop for c in wl["countries"]], wl["conn"], 0, util.oracle_state(base),
                              [util.prof_dict(p) for p in wl["profiles"]], 40, wl["betas"], wl["travel"], ta, ones, tp, ones, True, hp=ohp)
        util.assert_grid_equal(got, want)


def test_partial_targets(epi):
    """Targets present for some countries only (Map.getOrElse falls back to the model's own value)."""
    from epidemicsimulator_amd import synth
    wl = synth.small_workload(25, 2, 2, 30, seed=12)
    wl["t_arrive"] = {c: v for c, v in wl["t_arrive"].items() if c % 3 == 0}
    wl["t_peak"] = {c: v for c, v in wl["t_peak"].items() if c % 4 == 1}
    base = synth.base_state(30)
    util.assert_grid_equal(util.engine_grid(wl, base), util.oracle_grid(wl, base))


def test_int_range_is_reported(epi):
    """A patch pushed beyond Int.MaxValue by inflow: the reference's Math.round(x).toInt wraps and its mass
    `require` throws; the engine reports the overflow instead of continuing."""
    from epidemicsimulator_amd import synth
    countries, xy = synth.make_countries(3, 1)
    for c, p in zip(countries, (2_000_000_000, 2_000_000_000, 100_000_000)):
        c.pop = p
    W = np.array([[0, 0, 0.2], [0, 0, 0.2], [0, 0, 0]], dtype=np.float64)
    wl = synth.small_workload(3, 1, 1, 8, seed=1)
    wl["countries"], wl["profiles"], wl["conn"] = countries, countries, W
    with pytest.raises(epi.EpiError) as ei:
        util.engine_grid(wl, synth.base_state(8))
    assert ei.value.status in (3, 4, 6)
    assert util.oracle_grid(wl, synth.base_state(8))["err"] != 0


def test_int_range_in_a_sparse_sweep_names_the_callers_patch(epi):
    """A sweep numbers the patches of a sparse graph in its own order inside the engine; a failure is reported under
    the caller's patch number.  Patches 1500 and 1700 of a 2,000-patch ring pour 0.2 of two billion people each into
    patch 1600: the first failure any episode meets is there (or at one of the two sources)."""
    from epidemicsimulator_amd import synth
    n = 2000
    countries, xy = synth.make_countries(n, 5)
    rows = [[((i + 1) % n, 0.01), ((i + 7) % n, 0.01)] for i in range(n)]
    for src in (1500, 1700):
        countries[src].pop = 2_000_000_000
        rows[src] = [(1600, 0.2)]
    countries[1600].pop = 100_000_000
    rows = [sorted(r) for r in rows]
    rp = np.zeros(n + 1, dtype=np.int32); rp[1:] = np.cumsum([len(r) for r in rows])
    col = np.array([c for r in rows for c, _ in r], dtype=np.int32)
    val = np.array([v for r in rows for _, v in r], dtype=np.float64)
    wl = synth.small_workload(8, 1, 1, 6, seed=1)
    wl["countries"], wl["profiles"], wl["conn"] = countries, countries, (rp, col, val)
    wl["t_arrive"], wl["t_peak"] = synth.targets(n, 6, 1)
    wl["seed_idx"] = 3
    with pytest.raises(epi.EpiError) as ei:
        util.engine_grid(wl, synth.base_state(6))
    assert ei.value.status in (3, 4, 6)
    assert any("country %d," % c in ei.value.message for c in (1500, 1600, 1700)), ei.value.message


def test_bad_inputs_are_rejected(epi):
    from epidemicsimulator_amd import synth
    wl = synth.small_workload(6, 1, 1, 3)
    base = synth.base_state(3)
    bad = np.array(wl["conn"]); bad[1, 2] = -0.01
    wl2 = dict(wl, conn=bad)
    with pytest.raises(epi.EpiError):
        util.engine_grid(wl2, base)
    bad = np.array(wl["conn"]); bad[1, 2] = np.inf
    with pytest.raises(epi.EpiError):
        util.engine_grid(dict(wl, conn=bad), base)
    rp = np.array([0, 2, 2, 2, 2, 2, 2], dtype=np.int32); col = np.array([3, 1], dtype=np.int32); val = np.array([0.1, 0.1])
    with pytest.raises(epi.EpiError):
        util.engine_grid(dict(wl, conn=(rp, col, val)), base)       # columns not ascending
    w = epi.WorldToy(epi.HyperParams(), epi.State(0, 0, 0, 0, 0, 10_000, 0, 10, 1))
    with pytest.raises(epi.EpiError):
        w.restore_states([epi.State(1.5, 0, 0, 0, 0, 1.0, 0, 10, 1)] * 6)   # non-integer compartment


# ---------------------------------------------------------------- wire formats either side of the path
def test_snapshot_stream_records(epi, tmp_path):
    """Snapshot records (country, t, i, arrival, peakI, loss) of README.md:14 written as JSON lines; arrival and
    peak follow Validation.arrivalTime / peakI over the steps since the last reset."""
    import json
    hp = epi.HyperParams()
    w = epi.WorldToy(hp, epi.State(0, 0, 0, 0, 0, 10_000, 0, 120, 42), 0, epi.ToyConn.W, validate=True)
    series = {nd.name: [] for nd in w.nodes}
    for k in range(30):
        for name, s, _ in w.step_one(120):
            series[name].append(s.i)
    path = epi.Snapshots.write_epoch_snapshot(w, 30, 0.5, str(tmp_path / "stream_in" / "snapshots"))
    recs = [json.loads(line) for line in open(path)]
    assert [r["country"] for r in recs] == [c.name for c in epi.ToyConn.C]
    for r in recs:
        assert tuple(r.keys()) == epi.Snapshots.FIELDS
        want_arr = epi.Validation.arrival_time(series[r["country"]])
        assert r["arrival"] == (want_arr if want_arr < 30 else -1)
        assert r["peakI"] == epi.Validation.peak_i(series[r["country"]])
        assert r["i"] == series[r["country"]][-1] and r["t"] == 30 and r["loss"] == 0.5
    assert not [f for f in (tmp_path / "stream_in" / "snapshots").iterdir() if f.name.endswith(".tmp")]
    w.reset_to_initial()
    assert all(nd.arrival == -1 and nd.peak_i == 0.0 for nd in w.nodes)


# ---------------------------------------------------------------- graph construction on the device
@pytest.mark.parametrize("n", [6, 300, 4500])
def test_gravity_matrix_on_device(epi, n):
    """ToyConn.gravityMatrix on the GPU against the host evaluation: the device library's pow/sin/cos/atan2 may
    differ from the host libm in the last places (tolerance 1e-12 relative, stated here because no integer or
    pinned value exists for this function); MobilitySpec.scala:7-14's properties hold exactly."""
    from epidemicsimulator_amd import synth
    if n == 6:
        countries = list(epi.ToyConn.C)
        want = epi.ToyConn.W
    else:
        countries, _ = synth.make_countries(n, 11)
        pop = np.array([c.pop for c in countries], dtype=np.float64)
        lat_d = np.array([c.lat for c in countries]); lon_d = np.array([c.lon for c in countries])
        lat = np.radians(lat_d)   # Geo.scala:74-76: toRadians of the DIFFERENCES, and of each latitude
        h = np.sin(np.radians(lat_d[None, :] - lat_d[:, None]) / 2) ** 2 + np.cos(lat)[:, None] * np.cos(lat)[None, :] * np.sin(np.radians(lon_d[None, :] - lon_d[:, None]) / 2) ** 2
        d = np.maximum(300.0, 6371.0 * 2 * np.arctan2(np.sqrt(h), np.sqrt(1 - h)))
        want = 1e-6 * pop[:, None] ** 0.2 * pop[None, :] ** 1.0 / d ** 1.2
        np.fill_diagonal(want, 0.0)
        want *= np.minimum(1.0, 0.12 / want.sum(axis=1))[:, None]
    got = epi.ToyConn.gravity_matrix_device(countries)
    assert got.shape == (n, n)
    assert (np.diag(got) == 0.0).all()
    assert (got.sum(axis=1) <= 0.12 + 1e-12).all() and (got >= 0).all()
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=0)
    if n == 300:  # the device-built matrix is a valid engine input: sparse form of it runs and conserves mass
        rp, col, val = epi.ToyConn.to_csr(got, threshold=np.quantile(got, 0.9))
        assert rp[-1] == len(col) == len(val) and (np.diff(rp) > 0).all()


@pytest.mark.parametrize("n", [6, 300, 5000])
def test_gravity_csr_on_device(epi, n):
    """epi_gravity_csr -- the matrix formed and cut to a sparse graph on the device, rows compacted in column order --
    against the dense device matrix cut on the host: same structure, same doubles.  n = 5000 spans two row tiles."""
    from epidemicsimulator_amd import synth
    countries = list(epi.ToyConn.C) if n == 6 else synth.make_countries(n, 13)[0]
    dense = epi.ToyConn.gravity_matrix_device(countries)
    for thr in (0.0, float(np.quantile(dense, 0.97)), 1.0):
        want = epi.ToyConn.to_csr(dense, thr)
        got = epi.ToyConn.gravity_csr_device(countries, thr)
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g, w)
    # a sparse world built from it steps and conserves mass (rows that lost every entry export nothing)
    if n == 5000:
        rp, col, val = epi.ToyConn.gravity_csr_device(countries, float(np.quantile(dense, 0.997)))
        assert 5 < rp[-1] / n < 40
        base = epi.State(0, 0, 0, 0, 0, 10_000, 0, 50, 1)
        w = epi.WorldToy(epi.HyperParams(), base, 0, (rp, col, val), validate=True, countries=countries, learning=False)
        rng = np.random.default_rng(1)
        total = sum(c.pop for c in countries)
        for _ in range(4):
            st, _ = w.step_arrays(rng.integers(0, 8, (1, n)), rng.normal(size=(1, n)), 50)
            assert int((st["s"] + st["i"] + st["r"] + st["d"] + st["v"]).sum()) == total
        w.close()
