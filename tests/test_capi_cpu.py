"""CPU tests of the C ABI: the library loads, exports every symbol include/epi.h declares, its host-side
JVM-compatible streams match the oracle, and every compute entry point refuses to run without a GPU
(there is no CPU fallback behind the ABI)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def epi():
    import epidemicsimulator_amd as E
    E.lib()
    return E


def header_functions():
    text = open(os.path.join(ROOT, "include", "epi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(epi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(epi):
    from epidemicsimulator_amd import _capi
    names = header_functions()
    assert len(names) >= 25
    L = epi.lib()
    for name in names:
        assert hasattr(L, name), "libepi.so does not export %s" % name
    assert sorted(_capi.EXPORTED_SYMBOLS) == names
    assert L.epi_version() >= 100


def test_struct_layouts_match_header(epi):
    from epidemicsimulator_amd import _capi
    assert C.sizeof(_capi.CState) == 64 and C.sizeof(_capi.Profile) == 72 and C.sizeof(_capi.EnvParams) == 48
    assert C.sizeof(_capi.CHyper) == 48 and C.sizeof(_capi.LearnHyper) == 56 == C.sizeof(O.LearnHyper)
    assert C.sizeof(O.State) == C.sizeof(_capi.CState)


def test_host_streams_match_oracle(epi, oracle):
    L = epi.lib()
    for seed in (0, 7, 9, 42, 1041):
        out = np.zeros(300)
        assert L.epi_java_gaussians(C.c_int64(seed), 300, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
        np.testing.assert_array_equal(out, O.jrand_gaussians(seed, 300))
    hp = epi.HyperParams()._c()
    acts = np.zeros(600, dtype=np.int32)
    assert L.epi_agent_actions(C.byref(hp), 600, acts.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    assert acts[:10].tolist() == [6, 6, 7, 2, 4, 0, 2, 3, 2, 6]
    assert np.flatnonzero(acts < 0)[:2].tolist() == [113, 455]
    # act(s) passes `epsilon` as an argument: it is read before the call counts the step (DDQNAgent.scala:26-27,32).
    # With a one-step decay from 1 to 0 the first call still explores, every later one takes argmax Q.
    from epidemicsimulator_amd import _capi
    hp1 = _capi.CHyper(1.0, 0.0, 1, 128, 10.0, 5.0, 128, 8)
    assert L.epi_agent_actions(C.byref(hp1), 5, acts.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    assert acts[0] >= 0 and (acts[1:5] < 0).all()
    from oracle import pyref as P
    ag = P.Agent(dict(P.DEFAULT_HP, epsilon_decay_steps=7, epsilon_end=0.2))
    hp7 = _capi.CHyper(1.0, 0.2, 7, 128, 10.0, 5.0, 128, 8)
    assert L.epi_agent_actions(C.byref(hp7), 40, acts.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    want = []
    for _ in range(40):
        eps = ag.epsilon(); ag.steps += 1
        want.append(ag.rnd.next_int(8) if ag.rnd.next_double() < eps else -1)
    assert acts[:40].tolist() == want
    W = epi.mlp_init(128)
    for a, b in zip(W, O.mlp_weights()):
        np.testing.assert_array_equal(a, b)


def test_scale_row_matches_oracle(epi, oracle):
    rng = np.random.default_rng(2)
    for n in (2, 6, 33):
        W = rng.random((n, n)) * 0.05
        np.fill_diagonal(W, 0.0)
        for row in (0, n - 1):
            for f in (0.0, 0.5, 1.0, 3.0, 40.0):
                np.testing.assert_array_equal(epi.ToyConn.scale_row(W, row, f), O.scale_row(W, row, f))
    A = epi.ToyConn.scale_row(epi.ToyConn.W, 0, 9.0)
    assert A[0, 0] == 0.0 and abs(A[0].sum() - 0.2) < 1e-12
    np.testing.assert_array_equal(A[1:], epi.ToyConn.W[1:])


def test_argmin_is_first_minimum(epi):
    from epidemicsimulator_amd.dist import first_min
    assert epi.grid_argmin([3.0, 1.0, 1.0, 2.0]) == (1, 1.0)
    assert epi.grid_argmin([]) == (0, 1.7976931348623157e308)
    assert epi.grid_argmin([float("inf"), float("nan")]) == (0, 1.7976931348623157e308)
    rng = np.random.default_rng(0)
    x = np.round(rng.random(500), 2)
    assert epi.grid_argmin(x) == first_min(x) == (int(np.argmin(x)), float(x.min()))


def test_validation_mirror(epi):
    V = epi.Validation
    V.assert_state(epi.State(9, 1, 0, 0, 0, 10, 0, 10, 1), V.Invariants(10, 10))
    with pytest.raises(epi.IllegalArgumentException):
        V.assert_state(epi.State(9, 2, 0, 0, 0, 10, 0, 10, 1), V.Invariants(10, 10))
    with pytest.raises(epi.IllegalArgumentException):
        V.assert_state(epi.State(9, 1, 0, 0, 0, 10, 11, 10, 1), V.Invariants(10, 10))
    assert V.arrival_time([1.0, 2.0, 3.0, 4.0, 5.0], 10.0) == 5
    assert V.peak_i([]) == 0.0 and abs(V.peak_i([0.1, 0.5, 0.3]) - 0.5) <= 1e-12
    assert epi.Action.from_id(-3) == epi.Action.NoneA and epi.Action.from_id(99) == epi.Action.MassVax


def test_toy_matrix_properties(epi):
    W = epi.ToyConn.W
    assert W.shape == (6, 6)
    for i in range(6):
        assert W[i, i] == 0.0 and W[i].sum() <= 0.12 + 1e-12
    assert epi.ToyConn.W2[5, 3] == W[5, 3] * 3.0


def test_compute_entry_points_need_a_gpu(epi):
    if epi.lib().epi_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(epi.EpiError) as ei:
        epi.EpidemicEnv.allocate_int(10, [1, 1, 1])
    assert ei.value.status == 2
    from epidemicsimulator_amd import synth
    wl = synth.small_workload(8, 1, 1, 3)
    with pytest.raises(epi.EpiError) as ei:
        epi.GridEngine(epi.HyperParams(), 3, synth.base_state(3), 0, [c.pop for c in wl["countries"]], wl["conn"],
                       wl["profiles"], wl["betas"], wl["travel"])
    assert "HIP" in str(ei.value)
    with pytest.raises(epi.EpiError):
        epi.WorldToy(epi.HyperParams(), synth.base_state(3))
    with pytest.raises(epi.EpiError) as ei:
        epi.Agents([epi.HyperParams()])
    assert "HIP" in str(ei.value)
    with pytest.raises(epi.EpiError) as ei:                # the multi-device sweep: no device, no engine, no fallback
        epi.MultiGridEngine(epi.HyperParams(), 3, synth.base_state(3), 0, [c.pop for c in wl["countries"]], wl["conn"],
                            wl["profiles"], wl["betas"], wl["travel"], devices=[0, 1])
    assert ei.value.status == 2 and "HIP" in str(ei.value)


def test_bad_arguments_are_rejected_before_the_device(epi):
    from epidemicsimulator_amd import synth
    with pytest.raises(epi.EpiError) as ei:                # one hidden width per handle, checked before any device call
        epi.Agents([epi.HyperParams(hidden=32), epi.HyperParams(hidden=64)])
    assert "hidden" in str(ei.value)
    with pytest.raises(epi.EpiError):
        epi.Agents([epi.HyperParams(hidden=512)])
    with pytest.raises(epi.EpiError):
        epi.Agents([epi.HyperParams(replay_capacity=0)])
    wl = synth.small_workload(8, 1, 1, 3)
    with pytest.raises(epi.IllegalArgumentException):      # Validation.assertState on the initial states
        epi.GridEngine(epi.HyperParams(), 3, synth.base_state(3).copy(r=2.0), 0, [c.pop for c in wl["countries"]], wl["conn"],
                       wl["profiles"], wl["betas"], wl["travel"])
    with pytest.raises(epi.EpiError):
        epi.GridEngine(epi.HyperParams(), 3, synth.base_state(3), 99, [c.pop for c in wl["countries"]], wl["conn"],
                       wl["profiles"], wl["betas"], wl["travel"])


def test_synthetic_inputs_are_deterministic(epi):
    from epidemicsimulator_amd import synth
    a = synth.workload("c2"); b = synth.workload("c2")
    np.testing.assert_array_equal(a["conn"], b["conn"])
    assert [c.pop for c in a["countries"]] == [c.pop for c in b["countries"]]
    W = a["conn"]
    assert W.shape == (200, 200) and (np.diag(W) == 0).all() and (W.sum(1) <= 0.12 + 1e-12).all() and (W >= 0).all()
    assert synth.splitmix64(0, 1)[0] == np.uint64(0xE220A8397B1DCDAF)
    rp, col, val = synth.gravity_csr(*synth.make_countries(300, 3), 3, 5.0)
    assert rp[-1] == len(col) and (np.diff(rp) >= 1).all()
    for i in range(300):
        c = col[rp[i]:rp[i + 1]]
        assert (np.diff(c) > 0).all() and i not in c


def test_surface_csv_is_what_the_heatmap_script_reads(epi, tmp_path):
    """calib_surface.csv: columns beta, travel, loss, pivoted exactly as the reference's plot script does
    (src/main/scala/resources/plot_histogram.py:12-13).  Doubles survive the file bit for bit."""
    import pandas as pd
    betas = [k / 100.0 for k in range(24, 28)]
    travel = [k / 10.0 for k in range(5, 10)]
    rng = np.random.default_rng(4)
    rows = [(b, ts, float(rng.random() * 10.0 ** int(rng.integers(-8, 8)))) for b in betas for ts in travel]
    path = tmp_path / "calib_surface.csv"
    epi.Calibrate.write_surface_csv(str(path), rows)
    assert open(path).readline() == "beta,travel,loss\n"
    assert epi.Calibrate.read_surface_csv(str(path)) == rows
    df = pd.read_csv(path)
    pivot = df.pivot(index="beta", columns="travel", values="loss")
    assert pivot.shape == (4, 5)
    assert list(pivot.index) == betas and list(pivot.columns) == travel
    for i, b in enumerate(betas):
        for k, ts in enumerate(travel):
            # pandas' default float parser drops digits past the 15th or so; the file itself is exact (above)
            assert abs(pivot.loc[b, ts] - rows[i * 5 + k][2]) <= 1e-12 * abs(rows[i * 5 + k][2])
    with pytest.raises(ValueError):
        (tmp_path / "bad.csv").write_text("a,b\n1,2\n")
        epi.Calibrate.read_surface_csv(str(tmp_path / "bad.csv"))


def test_profile_table_fills_the_same_descriptor(epi):
    """A ProfileTable (profiles marshalled once, kept between engines) must give byte for byte the descriptor arrays that
    the list of profile objects gives: populations, epi_profile records, targets."""
    import ctypes as C
    from epidemicsimulator_amd import synth
    import epidemicsimulator_amd as E
    wl = synth.small_workload(37, 2, 3, 20, seed=11)
    pops = [c.pop for c in wl["countries"]]
    args = (E.HyperParams(), 20, synth.base_state(20), 0)
    tail = (wl["conn"],)
    d1, keep1 = E._grid_desc(*args, pops, wl["conn"], wl["profiles"], E.f64(wl["betas"]), E.f64(wl["travel"]),
                             wl["t_arrive"], wl["t_peak"], True, 0)
    table = E.ProfileTable(wl["profiles"], pops)
    d2, keep2 = E._grid_desc(*args, table.pops, wl["conn"], table, E.f64(wl["betas"]), E.f64(wl["travel"]),
                             wl["t_arrive"], wl["t_peak"], True, 0)
    n = len(pops)
    assert d1.n == d2.n == n
    assert C.string_at(d1.prof, n * C.sizeof(E._capi.Profile)) == C.string_at(d2.prof, n * C.sizeof(E._capi.Profile))
    assert C.string_at(d1.pop, 4 * n) == C.string_at(d2.pop, 4 * n)
    for f, w in (("t_arrive", 8), ("t_peak", 8), ("has_arrive", 1), ("has_peak", 1)):
        assert C.string_at(getattr(d1, f), w * n) == C.string_at(getattr(d2, f), w * n), f
    assert sum(C.string_at(d1.has_arrive, n)) == len(wl["t_arrive"])
    with pytest.raises(ValueError):
        E._grid_desc(*args, pops[:-1], wl["conn"], table, E.f64(wl["betas"]), E.f64(wl["travel"]), {}, {}, True, 0)
