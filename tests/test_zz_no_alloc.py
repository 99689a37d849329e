"""epi_grid_run_noalloc -- the sweep for callers that cannot allocate device memory -- against
epi_grid_run_strided with out_on_device != 0, through the C ABI: two calls in a row on one handle, the second
with a smaller input (the fixed work area keeps the first call's contents), bit for bit; then the size limit."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def epi():
    import epidemicsimulator_amd as E
    assert E.lib().epi_device_count() >= 1, "no HIP device: GPU tests must run on the MI355X box"
    return E


def _engine(epi, wl, steps):
    from epidemicsimulator_amd import synth
    return epi.GridEngine(epi.HyperParams(), steps, synth.base_state(steps), wl["seed_idx"], [c.pop for c in wl["countries"]],
                          wl["conn"], wl["profiles"], wl["betas"], wl["travel"], wl["t_arrive"], wl["t_peak"], True)


def _both(eng, g_begin, g_stride, g_count, n):
    """(loss, arrival, peak) of the allocating entry point and of the allocation-free one, device buffers."""
    outs = []
    for fn in (eng.enqueue, eng.enqueue_noalloc):
        loss = util.DeviceArray(g_count, np.float64, -7.0)
        arr = util.DeviceArray((g_count, n), np.int32, -7)
        pk = util.DeviceArray((g_count, n), np.float64, -7.0)
        fn(g_begin, g_count, loss.ptr, 0, arr.ptr, pk.ptr, g_stride=g_stride)
        eng.status()
        outs.append((loss.download(), arr.download(), pk.download()))
        for d in (loss, arr, pk):
            d.free()
    return outs


@pytest.mark.parametrize("kind", ["dense", "toy", "sparse"])
def test_noalloc_equals_allocating_entry_point(epi, kind):
    from epidemicsimulator_amd import synth
    if kind == "dense":       # 130 steps: the argmax-Q step at index 113 splits the episode into three segments
        wl, steps = synth.small_workload(70, 4, 5, 130, seed=41), 130
    elif kind == "toy":
        wl, steps = synth.workload("c1"), 120
    else:
        countries, xy = synth.make_countries(1300, 43)
        wl, steps = synth.small_workload(8, 3, 3, 12, seed=43), 12
        wl["countries"], wl["profiles"] = countries, countries
        wl["conn"] = synth.gravity_csr(countries, xy, 43, mean_degree=7.0)
        wl["t_arrive"], wl["t_peak"] = synth.targets(1300, steps, 43)
    wl = dict(wl, steps=steps)
    n = len(wl["countries"])
    G = len(wl["betas"]) * len(wl["travel"])
    eng = _engine(epi, wl, steps)
    # first call: the whole grid; second call: a smaller, strided share of it
    for g_begin, g_stride, g_count in ((0, 1, G), (1, 3, (G - 1 + 2) // 3 - 1)):
        (l0, a0, p0), (l1, a1, p1) = _both(eng, g_begin, g_stride, g_count, n)
        assert (l0 >= 0).all() and (a0 >= 0).all(), "the allocating entry point filled its outputs"
        np.testing.assert_array_equal(l1, l0)
        np.testing.assert_array_equal(a1, a0)
        np.testing.assert_array_equal(p1, p0)
    eng.close()


def test_noalloc_rejects_what_does_not_fit(epi):
    from epidemicsimulator_amd import synth, _capi
    wl = synth.small_workload(200, 64, 64, 3, seed=9)          # 4,096 episodes of 256 padded countries: 43 MB
    eng = _engine(epi, dict(wl, steps=3), 3)
    loss = util.DeviceArray(4096, np.float64, -7.0)
    with pytest.raises(epi.EpiError) as ei:
        eng.enqueue_noalloc(0, 4096, loss.ptr)
    assert ei.value.status == _capi.EPI_ERR_UNSUPPORTED and "work area" in ei.value.message
    assert (loss.download() == -7.0).all(), "a refused call runs nothing"
    eng.enqueue_noalloc(0, 3000, loss.ptr)                      # 3,000 episodes fit
    eng.status()
    got = loss.download()
    want = eng.run(0, 3000, want_metrics=False)["loss"]
    np.testing.assert_array_equal(got[:3000], want)
    with pytest.raises(epi.EpiError):
        eng.enqueue_noalloc(0, 5, 0)                            # null loss pointer
    eng.close()
