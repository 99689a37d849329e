"""The multi-device sweep inside the library (epi_grid_multi_*, RCCL all-gather of the loss pieces from C) on the
one GPU a test box has: one share, and several engines on device 0 -- the surface, the metrics and the first
minimum must equal the single-engine sweep bit for bit, whatever the split."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def epi():
    import epidemicsimulator_amd as E
    assert E.lib().epi_device_count() >= 1, "no HIP device: GPU tests must run on the MI355X box"
    return E


def _args(epi, wl, steps):
    from epidemicsimulator_amd import synth
    return (epi.HyperParams(), steps, synth.base_state(steps), wl["seed_idx"], [c.pop for c in wl["countries"]], wl["conn"],
            wl["profiles"], wl["betas"], wl["travel"], wl["t_arrive"], wl["t_peak"], True)


@pytest.mark.parametrize("devices,ranks", [([0], (1, 1)), ([0, 0], (1, 2)), ([0, 0, 0], (1, 3))])
def test_multi_equals_single_engine(epi, devices, ranks):
    from epidemicsimulator_amd import synth
    wl = synth.small_workload(40, 5, 7, 130, seed=21)        # 35 grid points: 3 shares of 12, 12, 11
    eng = epi.GridEngine(*_args(epi, wl, 130))
    whole = eng.run()
    eng.close()
    bi, bl = epi.grid_argmin(whole["loss"])
    m = epi.MultiGridEngine(*_args(epi, wl, 130), devices=devices)
    assert m.ranks() == ranks
    for rep in range(2):                                     # the handle is reusable
        out = m.run(want_metrics=(rep == 0))
        np.testing.assert_array_equal(out["loss"], whole["loss"])
        if rep == 0:
            np.testing.assert_array_equal(out["arrival"], whole["arrival"])
            np.testing.assert_array_equal(out["peak"], whole["peak"])
        assert (out["best_idx"], out["best_loss"]) == (bi, bl)
    m.close()


def test_multi_more_shares_than_points_and_one_shot(epi):
    from epidemicsimulator_amd import synth, _capi
    wl = synth.small_workload(12, 1, 3, 20, seed=3)          # 3 grid points, 4 shares: the last share is empty
    eng = epi.GridEngine(*_args(epi, wl, 20))
    whole = eng.run()
    eng.close()
    m = epi.MultiGridEngine(*_args(epi, wl, 20), devices=[0, 0, 0, 0])
    out = m.run()
    np.testing.assert_array_equal(out["loss"], whole["loss"])
    np.testing.assert_array_equal(out["arrival"], whole["arrival"])
    m.close()
    # epi_calibrate_grid_multi: create + run + destroy in one call, through Calibrate.grid_surface(devices=...)
    hp, cfg = epi.HyperParams(), epi.TrainConfig(epochs=1, steps_per_epoch=60)
    base = epi.State(1_000_000, 0, 0, 0, 0, 10_000, 0, 60, 3)
    tg = epi.Calibrate.Targets({"Japan": 20.0, "Italy": 31.0}, {"USA": 5e4})
    one = epi.Calibrate.grid_surface(hp, cfg, base, 0, tg, [0.24, 0.26, 0.3], [0.8, 1.0])
    two = epi.Calibrate.grid_surface(hp, cfg, base, 0, tg, [0.24, 0.26, 0.3], [0.8, 1.0], devices=[0, 0])
    assert one == two
    with pytest.raises(epi.EpiError) as ei:
        epi.MultiGridEngine(*_args(epi, wl, 20), devices=[0, 99])
    assert ei.value.status == _capi.EPI_ERR_BAD_ARGUMENT


def test_multi_full_size_c2(epi):
    """Configuration 2 at full size cut into eight shares (the 8-GPU split) on one device."""
    from epidemicsimulator_amd import synth
    wl = synth.workload("c2")
    eng = epi.GridEngine(*_args(epi, wl, wl["steps"]))
    whole = eng.run(want_metrics=False)
    eng.close()
    m = epi.MultiGridEngine(*_args(epi, wl, wl["steps"]), devices=[0] * 8)
    assert m.ranks() == (1, 8)
    out = m.run(want_metrics=False)
    m.close()
    np.testing.assert_array_equal(out["loss"], whole["loss"])
    assert (out["best_idx"], out["best_loss"]) == epi.grid_argmin(whole["loss"])


def test_a_share_that_fails_at_enqueue_leaves_a_reusable_handle(epi):
    """Share 2 of 3 fails on the host after shares 0 and 1 were enqueued: the call must wait for those two, collect
    their status, name the failing share -- and the very next run must give the whole surface again."""
    from epidemicsimulator_amd import synth, _capi
    wl = synth.small_workload(40, 5, 7, 130, seed=21)
    eng = epi.GridEngine(*_args(epi, wl, 130))
    whole = eng.run()
    eng.close()
    m = epi.MultiGridEngine(*_args(epi, wl, 130), devices=[0, 0, 0])
    for share in (2, 0, 1):
        m.inject_failure(share)
        with pytest.raises(epi.EpiError) as ei:
            m.run()
        assert ei.value.status == _capi.EPI_ERR_DEVICE and "share %d" % share in ei.value.message, ei.value.message
        out = m.run()
        np.testing.assert_array_equal(out["loss"], whole["loss"])
        np.testing.assert_array_equal(out["arrival"], whole["arrival"])
    m.close()


def test_a_share_that_fails_on_the_device_is_reported_and_the_handle_survives(epi):
    """The Int-range world: every episode overflows on the device.  Each run reports it with a clean status (never
    a stale 'earlier run' failure, never a hang), twice in a row, and destroy works."""
    from epidemicsimulator_amd import synth
    countries, xy = synth.make_countries(3, 1)
    for c, p in zip(countries, (2_000_000_000, 2_000_000_000, 100_000_000)):
        c.pop = p
    W = np.array([[0, 0, 0.2], [0, 0, 0.2], [0, 0, 0]], dtype=np.float64)
    wl = synth.small_workload(3, 2, 3, 8, seed=1)
    wl["countries"], wl["profiles"], wl["conn"] = countries, countries, W
    m = epi.MultiGridEngine(*_args(epi, wl, 8), devices=[0, 0, 0, 0])
    for _ in range(2):
        with pytest.raises(epi.EpiError) as ei:
            m.run()
        assert ei.value.status in (3, 4, 6), ei.value
        assert "instance 0," in ei.value.message, ei.value.message     # the first failing share's first grid point
    m.close()


def test_multi_timing_and_peer_table(epi):
    from epidemicsimulator_amd import synth
    wl = synth.small_workload(70, 4, 4, 60, seed=5)
    m = epi.MultiGridEngine(*_args(epi, wl, 60), devices=[0, 0])
    m.run(want_metrics=False)
    m.set_timing(True)
    for _ in range(3):
        m.run(want_metrics=False)
    t = m.times()
    assert t["runs"] == 3 and len(t["share_kernel_ms"]) == 2
    assert all(x > 0 for x in t["share_kernel_ms"]) and t["gather_ms"] > 0
    assert t["wall_ms"] >= max(t["share_kernel_ms"]) * 0.5
    assert m.peer_access().tolist() == [[1]]
    m.close()
