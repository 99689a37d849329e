"""The reference's own ScalaTest specs restated against the C ABI directly (raw ctypes calls on libepi.so, no
mirror class in between), so that each reads like the Scala test it follows.  What the specs pin is properties,
not vectors; the oracle comparison of the same inputs is added where it is cheap."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import epidemicsimulator_amd as E
    lib = E.lib()
    assert lib.epi_device_count() >= 1, "no HIP device: GPU tests must run on the MI355X box"
    return lib


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def test_mobility_spec_net_zero(L):
    """MobilitySpec.scala:24-63: the WorldToy mobility loop on the spec's S and I vectors with ToyConn.W, every
    allocateInt call served by epi_alloc_int: integer-conserving mobility must net to zero."""
    from epidemicsimulator_amd import ToyConn
    W = np.ascontiguousarray(ToyConn.W)
    n = W.shape[0]
    S = np.array([1_000_000, 120_000, 1_000_000, 80_000, 60_000, 300_000], dtype=np.float64)
    I = np.array([1_000, 200, 900, 100, 80, 400], dtype=np.float64)
    moved_s = np.zeros(n, dtype=np.int64); moved_i = np.zeros(n, dtype=np.int64)
    for i in range(n):
        w_row = np.array([0.0 if j == i else W[i, j] for j in range(n)])
        sum_w = 0.0
        for x in w_row:
            sum_w = sum_w + x                                   # wRow.sum, left to right
        sum_w = max(0.0, sum_w)
        Si, Ii = max(0, int(np.floor(S[i] + 0.5))), max(0, int(np.floor(I[i] + 0.5)))
        export_s = int(min(Si, np.floor(Si * sum_w + 1e-9)))
        export_i = int(min(Ii, np.floor(Ii * sum_w + 1e-9)))
        alloc_s = np.zeros(n, dtype=np.int32); alloc_i = np.zeros(n, dtype=np.int32)
        assert L.epi_alloc_int(C.c_int32(export_s), _pd(w_row), C.c_int32(n), _pi(alloc_s)) == 0
        assert L.epi_alloc_int(C.c_int32(export_i), _pd(w_row), C.c_int32(n), _pi(alloc_i)) == 0
        np.testing.assert_array_equal(alloc_s, O.alloc_int(export_s, w_row))
        np.testing.assert_array_equal(alloc_i, O.alloc_int(export_i, w_row))
        assert alloc_s.sum() == export_s and alloc_i.sum() == export_i and alloc_s[i] == 0 == alloc_i[i]
        for j in range(n):
            if j != i:
                moved_s[j] += alloc_s[j]; moved_i[j] += alloc_i[j]
        moved_s[i] -= export_s; moved_i[i] -= export_i
    assert moved_s.sum() + moved_i.sum() == 0


def test_heterogeneity_spec_one_step(L):
    """HeterogeneitySpec.scala:18-29: one env.step from the same state with China's and Japan's multipliers
    (ToyConn.C) gives a different next I; both equal the oracle's env.step on the same Gaussian draw."""
    from epidemicsimulator_amd import ToyConn, Geo, _capi
    cn = next(c for c in ToyConn.C if c.name == "China"); jp = next(c for c in ToyConn.C if c.name == "Japan")
    envs = (_capi.EnvParams * 2)(*[_capi.EnvParams(0.26 * c.beta_mul, 0.12, 0.008 * c.ifr_mul, c.noise, 5.0, 0.0) for c in (cn, jp)])
    dflt = Geo.CountryProfiles.get_profile("Unknown")        # countryName defaults to "Unknown" (env.scala:39,44-48)
    profs = (_capi.Profile * 2)(dflt._c(), dflt._c())
    s0 = _capi.CState(1_000_000, 1_000, 0, 0, 0, 10_000, 0, 10, 1, 0)
    states = (_capi.CState * 2)(s0, s0)
    act = np.zeros(2, dtype=np.int32)
    g = np.zeros(1)
    assert L.epi_java_gaussians(C.c_int64(42), 1, _pd(g)) == 0   # both envs are seeded 42 (env.scala:40-42)
    gauss = np.array([g[0], g[0]])
    out = (_capi.CState * 2)(); reward = np.zeros(2); done = np.zeros(2, dtype=np.uint8)
    assert L.epi_env_step_batch(envs, profs, states, _pi(act), _pd(gauss), out, _pd(reward),
                                done.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int32(2)) == 0
    assert out[0].i != out[1].i
    for k, c in enumerate((cn, jp)):
        oe = O.EnvParams(0.26 * c.beta_mul, 0.12, 0.008 * c.ifr_mul, c.noise, 5.0, 0.0)
        op = O.profiles_array([dict(beta_mul=1.0, ifr_mul=1.0, noise=0.05, population_density=100.0, gdp_per_capita=30000.0,
                                    healthcare_capacity=0.7, age_elderly=0.15, connectivity=0.8, econ_resilience=0.7)])[0]
        nxt, r, d = O.env_step(oe, op, O.State(1_000_000, 1_000, 0, 0, 0, 10_000, 0, 10, 1, 0), 0, float(g[0]))
        assert (out[k].s, out[k].i, out[k].r, out[k].d, out[k].v, out[k].t) == (nxt.s, nxt.i, nxt.r, nxt.d, nxt.v, nxt.t)
        assert reward[k] == r and bool(done[k]) == d
        assert out[k].s + out[k].i + out[k].r + out[k].d + out[k].v == 1_001_000       # mass is conserved by env.step


def _toy_world(L, hp, base, env, validate):
    from epidemicsimulator_amd import ToyConn, Geo, _capi
    n = 6
    pops = np.array([c.pop for c in ToyConn.C], dtype=np.int32)
    W = np.ascontiguousarray(ToyConn.W)
    envs = (_capi.EnvParams * n)(*[env] * n)
    profs = (_capi.Profile * n)(*[Geo.CountryProfiles.get_profile("Unknown")._c()] * n)
    seeds = np.full(n, 42, dtype=np.int32)
    d = _capi.WorldDesc()
    d.n, d.batch, d.pop, d.conn = n, 1, _pi(pops), _pd(W)
    d.seed_idx, d.validate, d.base, d.env, d.prof, d.env_seeds, d.hp, d.device = 0, int(validate), base, envs, profs, _pi(seeds), hp, 0
    h = L.epi_world_create(C.byref(d))
    assert h, "epi_world_create failed"
    return C.c_void_p(h)


def test_eval_determinism_spec(L):
    """EvalDeterminismSpec.scala:7-45: two identically built worlds (learnEvery = Int.MaxValue, noise 0.02, beta 0.26)
    give the same USA I series over 100 stepOneGreedy steps."""
    from epidemicsimulator_amd import _capi
    import epidemicsimulator_amd as E
    hp = _capi.CHyper(1.0, 0.1, 100000, 128, 10.0, 5.0, 128, 2**31 - 1)
    base = _capi.CState(0, 0, 0, 0, 0, 10_000, 0, 120, 42, 0)
    env = _capi.EnvParams(0.26, 0.12, 0.008, 0.02, 5.0, 0.0)
    W1, W2, W3 = E.mlp_init(128)

    def run_eval_series(steps):
        w = _toy_world(L, hp, base, env, validate=False)
        assert L.epi_world_set_mlp(w, 128, _pd(W1), None, _pd(W2), None, _pd(W3), None, C.c_double(10.0)) == 0
        out = (_capi.CState * 6)()
        series = []
        for _ in range(steps):
            assert L.epi_world_step_greedy(w, None, C.c_int32(steps), out, None, None) == 0
            series.append(out[0].i)                              # USA is node 0 of ToyConn.C
        L.epi_world_destroy(w)
        return series

    a, b = run_eval_series(100), run_eval_series(100)
    assert len(a) == 100 and a == b
    assert a[0] > 0 and max(a) > a[0]                            # the seeded country's epidemic grows


def test_calibration_spec(L):
    """CalibrationSpec.scala:6-16 through epi_calibrate_grid: beta {0.24, 0.26} x travel {0.8, 1.0}, 60 steps,
    homogeneous, empty targets: rows non-empty, best loss finite (every difference is 0, so it is exactly 0)."""
    from epidemicsimulator_amd import ToyConn, _capi
    n = 6
    pops = np.array([c.pop for c in ToyConn.C], dtype=np.int32)
    W = np.ascontiguousarray(ToyConn.W)
    betas = np.array([0.24, 0.26]); travel = np.array([0.8, 1.0])
    zeros = np.zeros(n); none = np.zeros(n, dtype=np.uint8)
    d = _capi.GridDesc()
    d.n, d.pop, d.conn, d.seed_idx, d.heterogeneous = n, _pi(pops), _pd(W), 0, 0
    d.base = _capi.CState(1_000_000, 0, 0, 0, 0, 10_000, 0, 60, 3, 0)
    d.prof = None
    d.hp = _capi.CHyper(1.0, 0.1, 100000, 128, 10.0, 5.0, 128, 8)
    d.steps, d.n_beta, d.betas, d.n_travel, d.travel = 60, 2, _pd(betas), 2, _pd(travel)
    d.t_arrive, d.has_arrive, d.t_peak, d.has_peak = _pd(zeros), none.ctypes.data_as(C.POINTER(C.c_uint8)), _pd(zeros), none.ctypes.data_as(C.POINTER(C.c_uint8))
    d.device = 0
    loss = np.full(4, -1.0); bi = C.c_int32(-1); bl = C.c_double(-1.0)
    assert L.epi_calibrate_grid(C.byref(d), _pd(loss), None, None, C.byref(bi), C.byref(bl)) == 0
    assert len(loss) == 4 and np.isfinite(bl.value)
    assert (loss == 0.0).all() and bl.value == 0.0 and bi.value == 0
