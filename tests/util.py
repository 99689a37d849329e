"""Shared helpers for the parity tests: the same seeded inputs go to the oracle and to the engine."""
import numpy as np

from oracle import oracle_lib as O


def prof_dict(c):
    """Geo.Country -> the oracle's profile dict."""
    return dict(beta_mul=c.beta_mul, ifr_mul=c.ifr_mul, noise=c.noise, population_density=c.population_density,
                gdp_per_capita=c.gdp_per_capita, healthcare_capacity=c.healthcare_capacity,
                age_elderly=c.age_distribution_elderly, connectivity=c.connectivity_strength,
                econ_resilience=c.economic_resilience)


def oracle_state(st):
    return O.State(st.s, st.i, st.r, st.d, st.v, st.hosp_cap, st.t, st.t_max, st.seed, 0)


def oracle_grid(wl, base, heterogeneous=True, sparse=False):
    """Oracle gridSurface on a workload dict from epidemicsimulator_amd.synth.  sparse: hand the oracle the CSR
    triple itself (O(stored entries) per step) instead of the densified matrix."""
    n = len(wl["countries"])
    ta = np.zeros(n); tp = np.zeros(n); ha = np.zeros(n, dtype=np.uint8); hk = np.zeros(n, dtype=np.uint8)
    for c, v in wl["t_arrive"].items():
        ta[c], ha[c] = v, 1
    for c, v in wl["t_peak"].items():
        tp[c], hk[c] = v, 1
    conn = wl["conn"] if sparse and isinstance(wl["conn"], tuple) else dense_of(wl["conn"], n)
    return O.grid_surface([c.pop for c in wl["countries"]], conn, wl["seed_idx"], oracle_state(base),
                          [prof_dict(p) for p in wl["profiles"]], wl["steps"], wl["betas"], wl["travel"], ta, ha, tp, hk,
                          heterogeneous)


def oracle_points(wl, base, points, threads=16, sparse=False, heterogeneous=True):
    """Oracle results of the grid points `points` = [(i_beta, i_travel), ...] of a workload, one single-point sweep
    per point, `threads` of them at a time (the C oracle releases the interpreter lock): dict(loss[P], arrival[P][n],
    peak[P][n])."""
    from concurrent.futures import ThreadPoolExecutor

    def one(pt):
        w = dict(wl)
        w["betas"], w["travel"] = [wl["betas"][pt[0]]], [wl["travel"][pt[1]]]
        out = oracle_grid(w, base, heterogeneous, sparse)
        assert out["err"] == 0, (pt, out["err"])
        return out

    with ThreadPoolExecutor(max_workers=max(1, min(threads, len(points)))) as ex:
        outs = list(ex.map(one, points))
    return dict(loss=np.array([o["loss"][0] for o in outs]), arrival=np.stack([o["arrival"][0] for o in outs]),
                peak=np.stack([o["peak"][0] for o in outs]))


def seed_row_cap_binds(wl, i_travel, cap=0.2):
    """Whether ToyConn.scaleRow's row-sum cap (ToyConn.scala:63-66) changes the seed row at this travel scale."""
    n = len(wl["countries"])
    i = wl["seed_idx"]
    if isinstance(wl["conn"], tuple):
        rp, col, val = wl["conn"]
        row = [v for c, v in zip(col[rp[i]:rp[i + 1]], val[rp[i]:rp[i + 1]]) if c != i]
    else:
        row = [wl["conn"][i][j] for j in range(n) if j != i]
    s = 0.0
    for v in row:
        s += v * wl["travel"][i_travel]
    return s > cap


def oracle_hypers(hp):
    """epidemicsimulator_amd.HyperParams -> the oracle's (Hyper, LearnHyper) pair."""
    ohp = O.Hyper(hp.epsilon_start, hp.epsilon_end, hp.epsilon_decay_steps, hp.hidden, hp.clip_q, hp.reward_clip,
                  hp.batch_size, hp.learn_every)
    lhp = O.LearnHyper(hp.gamma, hp.lr, hp.clip_q, hp.grad_clip, hp.ema_alpha, hp.hidden, hp.replay_capacity,
                       hp.target_update_every, hp.batch_size * 4)
    return ohp, lhp


def oracle_grid_learn(wl, base, hp, intensity=None, heterogeneous=True):
    """Oracle gridSurface whose rollouts step their worlds with the agents learning (WorldToy.scala:83-170 as written)."""
    n = len(wl["countries"])
    ta = np.zeros(n); tp = np.zeros(n); ha = np.zeros(n, dtype=np.uint8); hk = np.zeros(n, dtype=np.uint8)
    for c, v in wl["t_arrive"].items():
        ta[c], ha[c] = v, 1
    for c, v in wl["t_peak"].items():
        tp[c], hk[c] = v, 1
    ohp, lhp = oracle_hypers(hp)
    return O.grid_surface_learn([c.pop for c in wl["countries"]], dense_of(wl["conn"], n), wl["seed_idx"], oracle_state(base),
                                [prof_dict(p) for p in wl["profiles"]], wl["steps"], wl["betas"], wl["travel"], ta, ha, tp, hk,
                                lhp, intensity, heterogeneous, ohp)


def dense_of(conn, n):
    if isinstance(conn, tuple):
        rp, col, val = conn
        W = np.zeros((n, n))
        for i in range(n):
            W[i, col[rp[i]:rp[i + 1]]] = val[rp[i]:rp[i + 1]]
        return W
    return np.asarray(conn, dtype=np.float64)


def engine_grid(wl, base, heterogeneous=True, hp=None, names=None):
    from epidemicsimulator_amd import GridEngine, HyperParams, grid_argmin
    eng = GridEngine(hp or HyperParams(), wl["steps"], base, wl["seed_idx"], [c.pop for c in wl["countries"]],
                     wl["conn"], wl["profiles"], wl["betas"], wl["travel"], wl["t_arrive"], wl["t_peak"], heterogeneous,
                     names=names)
    try:
        out = eng.run()
    finally:
        eng.close()
    out["best_idx"], out["best_loss"] = grid_argmin(out["loss"])
    return out


def assert_grid_equal(got, want, rtol=1e-12):
    """Integer metrics bit-exact; fp64 losses within 1e-12 relative (the bar the north star states)."""
    assert want["err"] == 0
    np.testing.assert_array_equal(got["arrival"], want["arrival"])
    np.testing.assert_array_equal(got["peak"], want["peak"])
    np.testing.assert_allclose(got["loss"], want["loss"], rtol=rtol, atol=0.0)
    assert got["best_idx"] == want["best_idx"]
    assert abs(got["best_loss"] - want["best_loss"]) <= rtol * abs(want["best_loss"])


# ------------------------------------------------------------------------------------------------
# Device memory for tests that hand device pointers to the C ABI.  It comes from the HIP runtime the engine
# library itself is linked against (found among this process's mapped files once the library is loaded): a second
# runtime in the same process -- torch bundles its own -- would not see the device.
# ------------------------------------------------------------------------------------------------
_hip = None


def hip_runtime():
    global _hip
    if _hip is None:
        import ctypes as C
        from epidemicsimulator_amd import lib
        lib()
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64.so" in line:
                    path = line.split()[-1]
                    break
        assert path, "libepi.so is loaded but no libamdhip64.so is mapped"
        H = C.CDLL(path)
        H.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        H.hipFree.argtypes = [C.c_void_p]
        H.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        H.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip = H
    return _hip


class DeviceArray:
    """A device allocation with the shape and dtype of a numpy array."""

    def __init__(self, shape, dtype, fill=None):
        import ctypes as C
        self.shape, self.dtype = tuple(np.atleast_1d(shape)), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        assert hip_runtime().hipMalloc(C.byref(p), max(self.nbytes, 16)) == 0
        self.ptr = p.value
        if fill is not None:
            self.upload(np.full(self.shape, fill, dtype=self.dtype))

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes
        assert hip_runtime().hipMemcpy(self.ptr, a.ctypes.data, self.nbytes, 1) == 0   # hipMemcpyHostToDevice

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype)
        assert hip_runtime().hipDeviceSynchronize() == 0
        assert hip_runtime().hipMemcpy(out.ctypes.data, self.ptr, self.nbytes, 2) == 0   # hipMemcpyDeviceToHost
        return out

    def free(self):
        if self.ptr:
            hip_runtime().hipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
