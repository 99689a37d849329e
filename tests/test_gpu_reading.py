"""The discriminating inputs of tests/test_reading_cpu.py (oracle/READING.md) through the engine's C ABI: on each of
them the HIP path must give the answer of the reading taken, not the alternative's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def epi():
    import epidemicsimulator_amd as E
    assert E.lib().epi_device_count() >= 1, "no HIP device: GPU tests must run on the MI355X box"
    return E


def _profile(epi):
    return epi.Geo.CountryProfiles.get_profile("Unknown")   # Geo.Country defaults (Geo.scala:4-21,67)


def _step(epi, env, st, act, gauss):
    e = epi.EpidemicEnv(*env, profile=_profile(epi))
    nxt, r, d = epi.EpidemicEnv.step_batch([e], [epi.State(*st)], [act], [gauss])
    s = nxt[0]
    return (s.s, s.i, s.r, s.d, s.v, s.hosp_cap, s.t), float(r[0]), bool(d[0])


ENV = (0.35, 0.12, 0.008, 0.05, 5.0, 0.0)


def test_allocate_int_readings(epi):
    A = epi.EpidemicEnv.allocate_int
    assert A(21, [0.38, 0.86, 0.62, 0.94, 0.77]) == [2, 5, 4, 6, 4]        # left-to-right sum (pairwise: [2, 5, 4, 5, 5])
    assert A(36, [6.0, 28.0, 26.0]) == [3, 17, 16]                         # divide then multiply (w*total/sumW: [4, 17, 15])
    assert A(1, [0.25] * 4) == [1, 0, 0, 0] and A(10, [1.0, 1.0, 1.0]) == [4, 3, 3]   # ties keep index order
    assert A(2147483647, [3.0, 1.0]) == [1610612735, 536870912]


def test_env_step_readings(epi):
    # math.round(x).toInt wraps (env.scala:81): 3e9 susceptibles become 0
    assert _step(epi, ENV, (3e9, 1000, 0, 0, 0, 1e4, 0, 100, 1), 0, 0.0)[0][:5] == (0.0, 872.0, 120.0, 8.0, 0.0)
    # i0 > cap compares with the unrounded capacity (env.scala:95)
    assert _step(epi, ENV, (0, 10000.0, 0, 0, 0, 9999.5, 0, 100, 1), 0, 0.0)[0][1:4] == (8320.0, 1200.0, 480.0)
    # (1.0 - pRec) - pDie (env.scala:112)
    assert _step(epi, (0.35, 0.15, 0.05, 0.05, 5.0, 0.0), (0, 8.0, 0, 0, 0, 1e4, 0, 100, 1), 0, 0.0)[0][1:4] == (6.0, 1.0, 1.0)
    # done divides by s + i + r + d, without v (env.scala:26,137)
    nxt, _, done = _step(epi, (0.35, 0.0, 0.0, 0.05, 5.0, 0.0), (0, 1.0, 0, 0, 1e6, 1e4, 0, 100, 1), 0, 0.0)
    assert nxt[1] == 1.0 and done is False


def test_mixing_pool_int_wrap_is_reported_or_reproduced(epi):
    """s0 + i0 + r0 beyond Int.MaxValue (env.scala:87): the reference's Int sum wraps, everybody is infected and the new
    I count wraps to a negative number, which its own mass check then rejects.  The engine keeps compartments in int32:
    it must either reproduce that state or report the row as leaving the Int range -- never the double-sum answer."""
    try:
        nxt, _, _ = _step(epi, ENV, (2147483000.0, 1000.0, 0, 0, 0, 1e4, 0, 100, 1), 0, 0.0)
    except epi.EpiError as e:
        assert e.status == 6, e
        return
    assert nxt[0] == 0.0 and nxt[1] == -2147483424.0


def _world(epi, pops, conn, seed_idx, hp=None):
    countries = [epi.Geo.Country("c%d" % k, p, 0.0, 0.0) for k, p in enumerate(pops)]
    base = epi.State(0, 0, 0, 0, 0, 10000.0, 0, 100, 1)
    prof = _profile(epi)
    return epi.WorldToy(hp or epi.HyperParams(), base, seed_idx, np.array(conn, dtype=float),
                        make_env=lambda: epi.EpidemicEnv(*ENV, profile=prof), validate=True, countries=countries, learning=False)


def test_world_readings(epi):
    w = _world(epi, [1999, 1000], [[0, 0], [0, 0]], 0)                      # (0.001 * pop).toInt truncates
    assert (w.nodes[0].s.s, w.nodes[0].s.i) == (1998.0, 1.0)
    w.close()
    w = _world(epi, [5, 100], [[0, 0.1999999998], [0, 0]], 1)               # floor(S * sumW + 1e-9): multiply, then add
    w.step_actions([[0, 0]], [[0.0, 0.0]], 100)
    ms, _ = w.last_flows()
    assert ms.tolist() == [[0, 0]]                                          # one fused multiply-add: [[-1, 1]]
    w.close()
    w = _world(epi, [1000, 1000], [[0.5, 0.1], [0, 0]], 1)                  # the diagonal is ignored
    w.step_actions([[0, 0]], [[0.0, 0.0]], 100)
    assert w.last_flows()[0].tolist() == [[-100, 100]]
    w.close()
    w = _world(epi, [1000, 1000], [[0, 0.1], [0.1, 0]], 0)                  # the clock: +2 per step, capped by the argument
    st, _ = w.step_actions([[0, 0]], [[0.0, 0.0]], 100)
    assert [s.t for s in st[0]] == [2, 2]
    st, _ = w.step_actions([[0, 0]], [[0.0, 0.0]], 3)
    assert [s.t for s in st[0]] == [3, 3]
    w.close()


def test_agent_readings(epi):
    hp = epi.HyperParams(epsilon_decay_steps=1, epsilon_start=1.0, epsilon_end=0.0)   # epsilon read before the step is counted
    w = _world(epi, [100000, 1000], [[0, 0.1], [0.1, 0]], 0, hp=hp)
    w.step_one(100)
    assert [nd.last_a for nd in w.nodes] == [6, 6]                          # counted first: the network's 4
    w.close()
    w = _world(epi, [100000, 1000], [[0, 0.1], [0.1, 0]], 0)
    w.step_one_greedy(100)
    assert [nd.last_a for nd in w.nodes] == [4, 4]
    w.close()
    w = _world(epi, [100000, 1000], [[0, 0.1], [0.1, 0]], 0, hp=epi.HyperParams(clip_q=0.0))   # equal maxima: the first
    w.step_one_greedy(100)
    assert [nd.last_a for nd in w.nodes] == [0, 0]                          # last maximum: 7
    w.close()
