"""ctypes binding of oracle/libepi_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
the engine package never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libepi_oracle.so")


class Profile(C.Structure):
    _fields_ = [(k, C.c_double) for k in (
        "beta_mul", "ifr_mul", "noise", "population_density", "gdp_per_capita",
        "healthcare_capacity", "age_elderly", "connectivity", "econ_resilience")]


class EnvParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("base_beta", "base_gamma", "base_ifr", "noise", "reward_clip", "risk_factor")]


class State(C.Structure):
    _fields_ = [("s", C.c_double), ("i", C.c_double), ("r", C.c_double), ("d", C.c_double), ("v", C.c_double),
                ("hosp_cap", C.c_double), ("t", C.c_int32), ("t_max", C.c_int32), ("seed", C.c_int32),
                ("pad_", C.c_int32)]

    def tup(self):
        return (self.s, self.i, self.r, self.d, self.v, self.hosp_cap, self.t)


class Hyper(C.Structure):
    _fields_ = [("epsilon_start", C.c_double), ("epsilon_end", C.c_double), ("epsilon_decay_steps", C.c_int32),
                ("hidden", C.c_int32), ("clip_q", C.c_double), ("reward_clip", C.c_double),
                ("batch_size", C.c_int32), ("learn_every", C.c_int32)]


def default_hyper():
    return Hyper(1.0, 0.1, 100000, 128, 10.0, 5.0, 128, 8)


PROFILE_KEYS = [f[0] for f in Profile._fields_]


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "epi_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_strict_log.restype = C.c_double
        L.orc_strict_log.argtypes = [C.c_double]
        L.orc_world_create.restype = C.c_void_p
        L.orc_world_create_csr.restype = C.c_void_p
        L.orc_peak_i.restype = C.c_double
        L.orc_bench_points.restype = C.c_int64
        L.orc_bench_points_csr.restype = C.c_int64
        L.orc_learner_create.restype = C.c_void_p
        L.orc_world_learner.restype = C.c_void_p
        L.orc_learner_learn.restype = C.c_double
        L.orc_learner_ema_loss.restype = C.c_double
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def profiles_array(profiles):
    arr = (Profile * len(profiles))()
    for k, p in enumerate(profiles):
        for key in PROFILE_KEYS:
            setattr(arr[k], key, float(p[key]))
    return arr


def jrand_gaussians(seed, n):
    out = np.zeros(n)
    lib().orc_jrand_gaussians(C.c_int64(seed), C.c_int32(n), _p(out, C.c_double))
    return out


def jrand_doubles(seed, n):
    out = np.zeros(n)
    lib().orc_jrand_doubles(C.c_int64(seed), C.c_int32(n), _p(out, C.c_double))
    return out


def jrand_ints(seed, n, bound=None):
    out = np.zeros(n, dtype=np.int32)
    if bound is None:
        lib().orc_jrand_ints(C.c_int64(seed), C.c_int32(n), _p(out, C.c_int32))
    else:
        lib().orc_jrand_ints_bound(C.c_int64(seed), C.c_int32(bound), C.c_int32(n), _p(out, C.c_int32))
    return out


def alloc_int(total, weights):
    w = np.ascontiguousarray(weights, dtype=np.float64)
    out = np.zeros(len(w), dtype=np.int32)
    lib().orc_alloc_int(C.c_int32(int(total)), _p(w, C.c_double), C.c_int32(len(w)), _p(out, C.c_int32))
    return out


def alloc_int_sparse(total, vals, cols, n, extra_cap=64):
    v = np.ascontiguousarray(vals, dtype=np.float64)
    c = np.ascontiguousarray(cols, dtype=np.int32)
    out = np.zeros(len(v), dtype=np.int32)
    extra = np.zeros(extra_cap, dtype=np.int32)
    ne = C.c_int32(0)
    lib().orc_alloc_int_sparse(C.c_int32(int(total)), _p(v, C.c_double), _p(c, C.c_int32), C.c_int32(len(v)),
                               C.c_int32(n), _p(out, C.c_int32), _p(extra, C.c_int32), C.c_int32(extra_cap), C.byref(ne))
    return out, extra[:min(ne.value, extra_cap)], ne.value


def env_step(env, prof, state, action, gauss):
    nxt = State()
    r = C.c_double(0)
    d = C.c_int32(0)
    lib().orc_env_step(C.byref(env), C.byref(prof), C.byref(state), C.c_int32(action), C.c_double(gauss),
                       C.byref(nxt), C.byref(r), C.byref(d))
    return nxt, r.value, bool(d.value)


def mlp_weights(n_in=7, h=128, n_out=8):
    W1 = np.zeros((h, n_in)); W2 = np.zeros((h, h)); W3 = np.zeros((n_out, h))
    lib().orc_mlp_weights(n_in, h, n_out, _p(W1, C.c_double), _p(W2, C.c_double), _p(W3, C.c_double))
    return W1, W2, W3


def mlp_predict(x, clamp=10.0, n_in=7, h=128, n_out=8):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(n_out)
    lib().orc_mlp_predict(n_in, h, n_out, _p(x, C.c_double), C.c_double(clamp), _p(y, C.c_double))
    return y


class LearnHyper(C.Structure):
    """HyperParams fields DDQNAgent.learn reads (config.scala:3-21)."""
    _fields_ = [("gamma", C.c_double), ("lr", C.c_double), ("clip_q", C.c_double), ("grad_clip", C.c_double),
                ("ema_alpha", C.c_double), ("hidden", C.c_int32), ("replay_capacity", C.c_int32),
                ("target_update_every", C.c_int32), ("max_need", C.c_int32)]


def default_learn_hyper(**kw):
    hp = LearnHyper(0.99, 1e-3, 10.0, 1.0, 0.0, 128, 50000, 1500, 512)
    for k, v in kw.items():
        setattr(hp, k, v)
    return hp


class Learner:
    """One DDQNAgent's replay buffer, two networks and learn step (DDQNAgent.scala:6-65)."""

    def __init__(self, hp):
        self.hp = hp
        self.h = C.c_void_p(lib().orc_learner_create(C.byref(hp)))

    def close(self):
        if self.h:
            lib().orc_learner_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def observe(self, s, a, r, s2, done):
        lib().orc_learner_observe(self.h, C.byref(s), C.c_int32(int(a)), C.c_double(r), C.byref(s2), C.c_int32(int(done)))

    def replay_size(self):
        return lib().orc_learner_replay_size(self.h)

    def set_steps(self, steps):
        lib().orc_learner_set_steps(self.h, C.c_int64(steps))

    def ema_loss(self):
        return lib().orc_learner_ema_loss(self.h)

    def sample(self, n):
        out = np.zeros(max(n, 1), dtype=np.int32)
        k = lib().orc_learner_sample(self.h, C.c_int32(n), _p(out, C.c_int32))
        return out[:k]

    def learn(self, batch, grad_accum=4):
        return lib().orc_learner_learn(self.h, C.c_int32(batch), C.c_int32(grad_accum))

    def weights(self, target=False):
        h = self.hp.hidden
        W = [np.zeros((h, 7)), np.zeros(h), np.zeros((h, h)), np.zeros(h), np.zeros((8, h)), np.zeros(8)]
        lib().orc_learner_get_weights(self.h, C.c_int32(1 if target else 0), *[_p(w, C.c_double) for w in W])
        return W

    def q_values(self, s, target=False):
        q = np.zeros(8)
        lib().orc_learner_q_values(self.h, C.c_int32(1 if target else 0), C.byref(s), _p(q, C.c_double))
        return q


def mlp_train_batch(weights, xs, tgts, grad_clip, clamp, lr):
    """MLP.trainBatch on copies of `weights` = [W1, b1, W2, b2, W3, b3]; returns (new weights, loss)."""
    W = [np.array(w, dtype=np.float64, order="C", copy=True) for w in weights]
    xs = np.ascontiguousarray(xs, dtype=np.float64); tgts = np.ascontiguousarray(tgts, dtype=np.float64)
    loss = C.c_double(0.0)
    lib().orc_mlp_train_batch(C.c_int32(W[1].shape[0]), *[_p(w, C.c_double) for w in W], _p(xs, C.c_double),
                              _p(tgts, C.c_double), C.c_int32(xs.shape[0]), C.c_double(grad_clip), C.c_double(clamp),
                              C.c_double(lr), C.byref(loss))
    return W, loss.value


def scale_row(W, row, factor):
    A = np.array(W, dtype=np.float64, order="C", copy=True)
    lib().orc_scale_row(_p(A, C.c_double), C.c_int32(A.shape[0]), C.c_int32(row), C.c_double(factor))
    return A


class World:
    """Handle on the oracle's WorldToy restatement."""

    def __init__(self, pops, conn, seed_idx, base, envs, profiles, env_seeds, hp=None, validate=True):
        self.n = len(pops)
        pops = np.ascontiguousarray(pops, dtype=np.int32)
        seeds = np.ascontiguousarray(env_seeds, dtype=np.int32)
        env_arr = (EnvParams * self.n)(*envs)
        self.hp = hp or default_hyper()
        if isinstance(conn, tuple):   # (rowptr, col, val): the mobility loop walks the stored entries only
            rp = np.ascontiguousarray(conn[0], dtype=np.int32); col = np.ascontiguousarray(conn[1], dtype=np.int32)
            val = np.ascontiguousarray(conn[2], dtype=np.float64)
            self.h = C.c_void_p(lib().orc_world_create_csr(
                C.c_int32(self.n), _p(pops, C.c_int32), _p(rp, C.c_int32), _p(col, C.c_int32), _p(val, C.c_double),
                C.c_int32(seed_idx), C.byref(base), env_arr, profiles_array(profiles), _p(seeds, C.c_int32),
                C.byref(self.hp), C.c_int32(int(validate))))
            return
        conn = np.ascontiguousarray(conn, dtype=np.float64)
        self.h = C.c_void_p(lib().orc_world_create(
            C.c_int32(self.n), _p(pops, C.c_int32), _p(conn, C.c_double), C.c_int32(seed_idx), C.byref(base),
            env_arr, profiles_array(profiles), _p(seeds, C.c_int32), C.byref(self.hp), C.c_int32(int(validate))))

    def close(self):
        if self.h:
            lib().orc_world_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def states(self):
        arr = (State * self.n)()
        lib().orc_world_get_states(self.h, arr)
        return arr

    def set_states(self, arr):
        lib().orc_world_set_states(self.h, arr)

    def reset(self):
        lib().orc_world_reset(self.h)

    def error(self):
        node = C.c_int32(0); t = C.c_int32(0)
        return lib().orc_world_error(self.h, C.byref(node), C.byref(t)), node.value, t.value

    def last_actions(self):
        out = np.zeros(self.n, dtype=np.int32)
        lib().orc_world_last_actions(self.h, _p(out, C.c_int32))
        return out

    def step(self, t_max, greedy=False):
        r = np.zeros(self.n)
        lib().orc_world_step(self.h, C.c_int32(t_max), C.c_int32(int(greedy)), _p(r, C.c_double))
        return r

    def enable_learning(self, learn_hps, hypers, learn_every, batch_size, intensity):
        """Per-node DDQN learners (WorldToy.scala:46-65,147-167): LearnHyper and Hyper records per node."""
        n = self.n
        self._hypers = (Hyper * n)(*hypers)
        lh = (LearnHyper * n)(*learn_hps)
        le = np.ascontiguousarray(learn_every, dtype=np.int32); bs = np.ascontiguousarray(batch_size, dtype=np.int32)
        it = np.ascontiguousarray(intensity, dtype=np.int32)
        lib().orc_world_enable_learning(self.h, lh, _p(le, C.c_int32), _p(bs, C.c_int32), _p(it, C.c_int32))

    def step_learn(self, t_max):
        """WorldToy.stepOne with learning: (rewards[n], losses[n]; -1 where the node did not learn)."""
        r = np.zeros(self.n); loss = np.zeros(self.n)
        lib().orc_world_step_learn(self.h, self._hypers, C.c_int32(t_max), _p(r, C.c_double), _p(loss, C.c_double))
        return r, loss

    def learner(self, c):
        """Borrowed view of node c's learner (weights / q_values); owned by the world."""
        L = Learner.__new__(Learner)
        L.h = C.c_void_p(lib().orc_world_learner(self.h, C.c_int32(c)))
        L.hp = None
        L.close = lambda: None
        return L

    def step_given(self, actions, gauss, t_max):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        g = np.ascontiguousarray(gauss, dtype=np.float64)
        r = np.zeros(self.n); ms = np.zeros(self.n, dtype=np.int32); mi = np.zeros(self.n, dtype=np.int32)
        lib().orc_world_step_given(self.h, _p(a, C.c_int32), _p(g, C.c_double), C.c_int32(t_max), _p(r, C.c_double),
                                   _p(ms, C.c_int32), _p(mi, C.c_int32))
        return r, ms, mi


def rollout(pops, W, seed_idx, base, profiles, steps, beta, ts, heterogeneous, hp=None, want_series=False):
    n = len(pops)
    pops = np.ascontiguousarray(pops, dtype=np.int32)
    W = np.ascontiguousarray(W, dtype=np.float64)
    hp = hp or default_hyper()
    arr = np.zeros(n, dtype=np.int32); pk = np.zeros(n); fin = np.zeros(n)
    series = np.zeros((n, steps)) if want_series else None
    err = C.c_int32(0)
    lib().orc_rollout(C.c_int32(n), _p(pops, C.c_int32), _p(W, C.c_double), C.c_int32(seed_idx), C.byref(base),
                      profiles_array(profiles), C.byref(hp), C.c_int32(steps), C.c_double(beta), C.c_double(ts),
                      C.c_int32(int(heterogeneous)), _p(arr, C.c_int32), _p(pk, C.c_double), _p(fin, C.c_double),
                      _p(series, C.c_double) if want_series else None, C.byref(err))
    return arr, pk, fin, series, err.value


def grid_surface(pops, W, seed_idx, base, profiles, steps, betas, tss, t_arrive, has_arr, t_peak, has_pk,
                 heterogeneous=True, hp=None):
    n = len(pops)
    pops = np.ascontiguousarray(pops, dtype=np.int32)
    betas = np.ascontiguousarray(betas, dtype=np.float64); tss = np.ascontiguousarray(tss, dtype=np.float64)
    ta = np.ascontiguousarray(t_arrive, dtype=np.float64); tp = np.ascontiguousarray(t_peak, dtype=np.float64)
    ha = np.ascontiguousarray(has_arr, dtype=np.uint8); hk = np.ascontiguousarray(has_pk, dtype=np.uint8)
    hp = hp or default_hyper()
    G = len(betas) * len(tss)
    loss = np.zeros(G); arr = np.zeros((G, n), dtype=np.int32); pk = np.zeros((G, n))
    bi = C.c_int32(0); bl = C.c_double(0)
    if isinstance(W, tuple):   # CSR graph (rowptr, col, val): O(stored entries) per world step
        rp = np.ascontiguousarray(W[0], dtype=np.int32); col = np.ascontiguousarray(W[1], dtype=np.int32)
        val = np.ascontiguousarray(W[2], dtype=np.float64)
        err = lib().orc_grid_surface_csr(C.c_int32(n), _p(pops, C.c_int32), _p(rp, C.c_int32), _p(col, C.c_int32), _p(val, C.c_double),
                                         C.c_int32(seed_idx), C.byref(base), profiles_array(profiles), C.byref(hp), C.c_int32(steps),
                                         _p(betas, C.c_double), C.c_int32(len(betas)), _p(tss, C.c_double), C.c_int32(len(tss)),
                                         _p(ta, C.c_double), _p(ha, C.c_uint8), _p(tp, C.c_double), _p(hk, C.c_uint8),
                                         C.c_int32(int(heterogeneous)), _p(loss, C.c_double), _p(arr, C.c_int32), _p(pk, C.c_double),
                                         C.byref(bi), C.byref(bl))
        return dict(loss=loss, arrival=arr, peak=pk, best_idx=bi.value, best_loss=bl.value, err=err)
    W = np.ascontiguousarray(W, dtype=np.float64)
    err = lib().orc_grid_surface(C.c_int32(n), _p(pops, C.c_int32), _p(W, C.c_double), C.c_int32(seed_idx), C.byref(base),
                                 profiles_array(profiles), C.byref(hp), C.c_int32(steps), _p(betas, C.c_double),
                                 C.c_int32(len(betas)), _p(tss, C.c_double), C.c_int32(len(tss)), _p(ta, C.c_double),
                                 _p(ha, C.c_uint8), _p(tp, C.c_double), _p(hk, C.c_uint8), C.c_int32(int(heterogeneous)),
                                 _p(loss, C.c_double), _p(arr, C.c_int32), _p(pk, C.c_double), C.byref(bi), C.byref(bl))
    return dict(loss=loss, arrival=arr, peak=pk, best_idx=bi.value, best_loss=bl.value, err=err)


def grid_surface_learn(pops, W, seed_idx, base, profiles, steps, betas, tss, t_arrive, has_arr, t_peak, has_pk,
                       learn_hp, intensity=None, heterogeneous=True, hp=None):
    """grid_surface with the rollouts' agents learning (WorldToy.scala:147-167): learn_hp is a LearnHyper,
    intensity[c] the gradAccumSteps of country c (1 when None)."""
    n = len(pops)
    pops = np.ascontiguousarray(pops, dtype=np.int32)
    betas = np.ascontiguousarray(betas, dtype=np.float64); tss = np.ascontiguousarray(tss, dtype=np.float64)
    ta = np.ascontiguousarray(t_arrive, dtype=np.float64); tp = np.ascontiguousarray(t_peak, dtype=np.float64)
    ha = np.ascontiguousarray(has_arr, dtype=np.uint8); hk = np.ascontiguousarray(has_pk, dtype=np.uint8)
    hp = hp or default_hyper()
    it = np.ascontiguousarray(intensity if intensity is not None else np.ones(n), dtype=np.int32)
    G = len(betas) * len(tss)
    loss = np.zeros(G); arr = np.zeros((G, n), dtype=np.int32); pk = np.zeros((G, n))
    bi = C.c_int32(0); bl = C.c_double(0)
    W = np.ascontiguousarray(W, dtype=np.float64)
    err = lib().orc_grid_surface_learn(C.c_int32(n), _p(pops, C.c_int32), _p(W, C.c_double), C.c_int32(seed_idx), C.byref(base),
                                       profiles_array(profiles), C.byref(hp), C.c_int32(steps), _p(betas, C.c_double),
                                       C.c_int32(len(betas)), _p(tss, C.c_double), C.c_int32(len(tss)), _p(ta, C.c_double),
                                       _p(ha, C.c_uint8), _p(tp, C.c_double), _p(hk, C.c_uint8), C.c_int32(int(heterogeneous)),
                                       _p(loss, C.c_double), _p(arr, C.c_int32), _p(pk, C.c_double), C.byref(bi), C.byref(bl),
                                       C.byref(learn_hp), _p(it, C.c_int32))
    return dict(loss=loss, arrival=arr, peak=pk, best_idx=bi.value, best_loss=bl.value, err=err)


def bench_points(pops, W, seed_idx, base, profiles, steps, betas, tss, heterogeneous, g0, n_points, hp=None):
    n = len(pops)
    pops = np.ascontiguousarray(pops, dtype=np.int32)
    betas = np.ascontiguousarray(betas, dtype=np.float64); tss = np.ascontiguousarray(tss, dtype=np.float64)
    hp = hp or default_hyper()
    cs = C.c_double(0)
    if isinstance(W, tuple):
        rp = np.ascontiguousarray(W[0], dtype=np.int32); col = np.ascontiguousarray(W[1], dtype=np.int32)
        val = np.ascontiguousarray(W[2], dtype=np.float64)
        done = lib().orc_bench_points_csr(C.c_int32(n), _p(pops, C.c_int32), _p(rp, C.c_int32), _p(col, C.c_int32), _p(val, C.c_double),
                                          C.c_int32(seed_idx), C.byref(base), profiles_array(profiles), C.byref(hp), C.c_int32(steps),
                                          _p(betas, C.c_double), C.c_int32(len(betas)), _p(tss, C.c_double), C.c_int32(len(tss)),
                                          C.c_int32(int(heterogeneous)), C.c_int32(g0), C.c_int32(n_points), C.byref(cs))
        return int(done), cs.value
    W = np.ascontiguousarray(W, dtype=np.float64)
    done = lib().orc_bench_points(C.c_int32(n), _p(pops, C.c_int32), _p(W, C.c_double), C.c_int32(seed_idx), C.byref(base),
                                  profiles_array(profiles), C.byref(hp), C.c_int32(steps), _p(betas, C.c_double),
                                  C.c_int32(len(betas)), _p(tss, C.c_double), C.c_int32(len(tss)),
                                  C.c_int32(int(heterogeneous)), C.c_int32(g0), C.c_int32(n_points), C.byref(cs))
    return int(done), cs.value
