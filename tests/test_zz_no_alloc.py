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


def _free_bytes():
    import ctypes as C
    H = util.hip_runtime()
    H.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert H.hipDeviceSynchronize() == 0
    assert H.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_noalloc_leaves_free_device_memory_unchanged(epi):
    """Free device memory before and after a series of calls (the first call on the device, which loads the kernels and
    looks the fixed array up, is made before the bracket).  The runtime hands out device memory in large granules, so
    the calls are sized such that their per-episode state (900 episodes x 128 padded countries x 41 B = 4.7 MB) would
    show: the allocating entry point, run afterwards at the same size, does move the figure."""
    from epidemicsimulator_amd import synth
    wl = synth.small_workload(70, 30, 30, 20, seed=41)
    eng = _engine(epi, dict(wl, steps=20), 20)
    loss = util.DeviceArray(900, np.float64, -7.0)
    arr = util.DeviceArray((900, 70), np.int32, -7)
    eng.enqueue_noalloc(0, 2, loss.ptr, 0, arr.ptr)
    eng.status()
    before = _free_bytes()
    for k in range(4):
        eng.enqueue_noalloc(k % 3, 900 - 100 * k, loss.ptr, 0, arr.ptr)
        eng.status()
    assert _free_bytes() == before
    eng.enqueue(0, 900, loss.ptr, 0, arr.ptr)
    eng.status()
    assert _free_bytes() < before, "the allocating entry point's work buffers are visible at this size"
    eng.close()


def test_noalloc_refuses_while_a_run_is_uncollected_and_never_waits(epi):
    from epidemicsimulator_amd import synth, _capi
    wl = synth.small_workload(70, 2, 2, 40, seed=4)
    eng = _engine(epi, dict(wl, steps=40), 40)
    loss = util.DeviceArray(4, np.float64, -7.0)
    eng.enqueue_noalloc(0, 4, loss.ptr)
    with pytest.raises(epi.EpiError) as ei:
        eng.enqueue_noalloc(0, 4, loss.ptr)
    assert ei.value.status == _capi.EPI_ERR_BAD_ARGUMENT and "epi_grid_status" in ei.value.message
    eng.status()
    want = eng.run(0, 4, want_metrics=False)["loss"]
    np.testing.assert_array_equal(loss.download(), want)
    eng.close()


def test_noalloc_status_belongs_to_the_handle(epi):
    """Two handles enqueue on one stream, one after the other, and both use the GPU's one work area: the first one's
    episodes fail on the device (Int range), the second one's do not.  Each handle must report its own outcome."""
    from epidemicsimulator_amd import synth
    countries, xy = synth.make_countries(3, 1)
    for c, p in zip(countries, (2_000_000_000, 2_000_000_000, 100_000_000)):
        c.pop = p
    bad = synth.small_workload(3, 1, 2, 8, seed=1)
    bad["countries"], bad["profiles"] = countries, countries
    bad["conn"] = np.array([[0, 0, 0.2], [0, 0, 0.2], [0, 0, 0]], dtype=np.float64)
    good = synth.small_workload(3, 1, 2, 8, seed=2)
    A = _engine(epi, dict(bad, steps=8), 8)
    B = _engine(epi, dict(good, steps=8), 8)
    la, lb = util.DeviceArray(2, np.float64, -7.0), util.DeviceArray(2, np.float64, -7.0)
    A.enqueue_noalloc(0, 2, la.ptr)
    B.enqueue_noalloc(0, 2, lb.ptr)          # same (null) stream: ordered after A, overwrites the shared area
    with pytest.raises(epi.EpiError) as ei:
        A.status()
    assert ei.value.status in (3, 4, 6), ei.value
    B.status()
    np.testing.assert_array_equal(lb.download(), B.run(0, 2, want_metrics=False)["loss"])
    A.close(); B.close()


def test_noalloc_can_be_recorded_under_stream_capture(epi):
    """The call makes no allocation, no device-wide wait, no attribute change: it may be recorded into a HIP graph
    (capture mode `global`, under which such calls fail) and replayed; the replays give the sweep's results."""
    import ctypes as C
    from epidemicsimulator_amd import synth
    H = util.hip_runtime()
    wl = synth.small_workload(70, 3, 4, 130, seed=17)          # 130 steps: three episode segments and the network step
    eng = _engine(epi, dict(wl, steps=130), 130)
    want = eng.run(want_metrics=True)
    loss = util.DeviceArray(12, np.float64, -7.0)
    arr = util.DeviceArray((12, 70), np.int32, -7)
    eng.enqueue_noalloc(0, 12, loss.ptr, 0, arr.ptr)           # kernels loaded, fixed array looked up
    eng.status()
    stream, graph, gexec = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert H.hipStreamCreate(C.byref(stream)) == 0
    assert H.hipStreamBeginCapture(stream, 0) == 0             # hipStreamCaptureModeGlobal
    try:
        eng.enqueue_noalloc(0, 12, loss.ptr, stream.value, arr.ptr)
    finally:
        rc = H.hipStreamEndCapture(stream, C.byref(graph))
    assert rc == 0 and graph.value, "the capture was invalidated (rc %d)" % rc
    H.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    assert H.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
    for rep in range(2):
        loss.upload(np.full(12, -7.0)); arr.upload(np.full((12, 70), -7, dtype=np.int32))
        assert H.hipGraphLaunch(gexec, stream) == 0
        assert H.hipStreamSynchronize(stream) == 0
        np.testing.assert_array_equal(loss.download(), want["loss"])
        np.testing.assert_array_equal(arr.download(), want["arrival"])
    eng.status()                                               # the status of the replays, from the handle's own table
    assert H.hipGraphExecDestroy(gexec) == 0 and H.hipGraphDestroy(graph) == 0 and H.hipStreamDestroy(stream) == 0
    eng.close()
