"""epidemicsimulator_amd -- MI355X engine for the metapopulation step and the beta x travel
calibration sweep, behind the reference's own API names.

Python here is only the host-side mirror of the reference interface (same class and method
names, argument meaning and error behaviour as the Scala sources under
src/main/scala/epidemic/); all arithmetic of the path runs in libepi.so on the GPU.
"""
import ctypes as C
import enum
import math
from dataclasses import dataclass, field, replace
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _capi
from ._capi import (EpiError, IllegalArgumentException, EPI_OK, check, f64, i32, lib, ptr)

__all__ = ["Action", "State", "HyperParams", "TrainConfig", "Geo", "ToyConn", "EpidemicEnv", "WorldToy",
           "Calibrate", "Validation", "GridEngine", "MultiGridEngine", "Snapshots", "EpiError", "IllegalArgumentException"]


# ------------------------------------------------------------------------------------------------
# env.scala:6-30
# ------------------------------------------------------------------------------------------------
class Action(enum.IntEnum):
    NoneA = 0
    Distancing = 1
    TravelBan = 2
    Lockdown = 3
    NormalCare = 4
    SurgeCare = 5
    TargetedVax = 6
    MassVax = 7

    @staticmethod
    def from_id(i: int) -> "Action":  # env.scala:18-20
        return Action(max(0, min(7, int(i))))


@dataclass
class State:  # env.scala:22-30
    s: float
    i: float
    r: float
    d: float
    v: float
    hosp_cap: float
    t: int
    t_max: int
    seed: int = 0

    @property
    def n(self) -> float:
        return self.s + self.i + self.r + self.d

    @property
    def util(self) -> float:
        return self.i / max(1.0, self.hosp_cap)

    def features(self) -> List[float]:
        N = max(1.0, self.n)
        return [self.s / N, self.i / N, self.r / N, self.d / N, self.v / N, self.util,
                float(self.t) / max(1, self.t_max)]

    def copy(self, **kw) -> "State":
        return replace(self, **kw)

    def _c(self) -> _capi.CState:
        return _capi.CState(float(self.s), float(self.i), float(self.r), float(self.d), float(self.v),
                            float(self.hosp_cap), int(self.t), int(self.t_max), int(self.seed), 0)

    @staticmethod
    def _from_c(c) -> "State":
        return State(c.s, c.i, c.r, c.d, c.v, c.hosp_cap, c.t, c.t_max, c.seed)


@dataclass
class HyperParams:  # config.scala:3-21
    gamma: float = 0.99
    lr: float = 1e-3
    epsilon_start: float = 1.0
    epsilon_end: float = 0.1
    epsilon_decay_steps: int = 100000
    target_update_every: int = 1500
    replay_capacity: int = 50000
    batch_size: int = 128
    hidden: int = 128
    clip_q: float = 10.0
    reward_clip: float = 5.0
    grad_clip: float = 1.0
    log_interval: int = 500
    ema_alpha: float = 0.0
    checkpoint: bool = True
    learn_every: int = 8

    def _c(self) -> _capi.CHyper:
        return _capi.CHyper(self.epsilon_start, self.epsilon_end, self.epsilon_decay_steps, self.hidden, self.clip_q,
                            self.reward_clip, self.batch_size, self.learn_every)


@dataclass
class TrainConfig:  # config.scala:23-27
    epochs: int = 1000
    steps_per_epoch: int = 500
    eval_every: int = 10


# ------------------------------------------------------------------------------------------------
# Geo.scala
# ------------------------------------------------------------------------------------------------
class Geo:
    @dataclass
    class Country:  # Geo.scala:4-21
        name: str
        pop: int
        lat: float
        lon: float
        hosp_per_1k: float = 3.0
        beta_mul: float = 1.0
        ifr_mul: float = 1.0
        noise: float = 0.05
        population_density: float = 100.0
        gdp_per_capita: float = 30000.0
        healthcare_capacity: float = 0.7
        mobility_index: float = 0.8
        age_distribution_elderly: float = 0.15
        climate_factor: float = 0.6
        connectivity_strength: float = 0.8
        economic_resilience: float = 0.7

        def _c(self) -> _capi.Profile:
            return _capi.Profile(self.beta_mul, self.ifr_mul, self.noise, self.population_density,
                                 self.gdp_per_capita, self.healthcare_capacity, self.age_distribution_elderly,
                                 self.connectivity_strength, self.economic_resilience)

    class CountryProfiles:  # Geo.scala:23-69
        _TABLE = {
            "USA": ("USA", 331000000, 39.8, -98.5, 2.9, 1.0, 1.0, 0.05, 36.0, 65000.0, 0.85, 0.9, 0.20, 0.7, 0.95, 0.9),
            "Japan": ("Japan", 125000000, 36.2, 138.2, 13.1, 0.8, 0.9, 0.03, 347.0, 42000.0, 0.95, 0.6, 0.30, 0.5, 0.8, 0.8),
            "India": ("India", 1380000000, 20.6, 78.9, 0.5, 1.3, 1.2, 0.08, 464.0, 2100.0, 0.3, 0.4, 0.07, 0.8, 0.4, 0.3),
            "Italy": ("Italy", 60000000, 41.9, 12.6, 3.2, 1.1, 1.4, 0.04, 206.0, 35000.0, 0.8, 0.7, 0.28, 0.6, 0.85, 0.7),
            "Germany": ("Germany", 83000000, 51.2, 10.5, 8.0, 0.9, 0.8, 0.03, 240.0, 46000.0, 0.9, 0.8, 0.25, 0.4, 0.9, 0.85),
            "China": ("China", 1400000000, 35.9, 104.2, 4.3, 1.0, 1.0, 0.06, 153.0, 10500.0, 0.6, 0.5, 0.18, 0.7, 0.7, 0.6),
        }

        @staticmethod
        def get_profile(name: str) -> "Geo.Country":
            row = Geo.CountryProfiles._TABLE.get(name)
            if row is None:
                return Geo.Country(name, 50000000, 0.0, 0.0)  # default fallback, Geo.scala:67
            return Geo.Country(*row)

    @staticmethod
    def haversine_km(lat1, lon1, lat2, lon2) -> float:  # Geo.scala:72-79
        R = 6371.0
        d_lat = math.radians(lat2 - lat1)
        d_lon = math.radians(lon2 - lon1)
        a = math.pow(math.sin(d_lat / 2), 2) + math.cos(math.radians(lat1)) * math.cos(math.radians(lat2)) * math.pow(
            math.sin(d_lon / 2), 2)
        c = 2 * math.atan2(math.sqrt(a), math.sqrt(1 - a))
        return R * c


# ------------------------------------------------------------------------------------------------
# ToyConn.scala
# ------------------------------------------------------------------------------------------------
class ToyConn:
    C: List[Geo.Country] = [  # ToyConn.scala:7-14 (positional: hospPer1k, betaMul, ifrMul, noise)
        Geo.Country("USA", 330_000_000, 38.0, -97.0, 2.9, 1.00, 1.00, 0.05),
        Geo.Country("Japan", 125_000_000, 36.0, 138.0, 7.8, 0.92, 0.90, 0.04),
        Geo.Country("Italy", 60_000_000, 42.5, 12.5, 3.2, 0.98, 1.05, 0.05),
        Geo.Country("India", 1_420_000_000, 22.0, 79.0, 0.7, 1.05, 0.95, 0.06),
        Geo.Country("Germany", 83_000_000, 51.0, 10.0, 8.0, 0.96, 0.95, 0.04),
        Geo.Country("China", 1_410_000_000, 35.0, 103.0, 4.3, 1.03, 1.00, 0.05),
    ]
    flight_mul: Dict[Tuple[str, str], float] = {  # ToyConn.scala:16-20
        ("China", "India"): 1.8, ("India", "China"): 1.2, ("USA", "Japan"): 1.6, ("Japan", "USA"): 1.6,
        ("Germany", "Italy"): 1.3, ("Italy", "Germany"): 1.2}

    @staticmethod
    def gravity_matrix(countries, a=0.2, b=1.0, g=1.2, k=1e-6, cap_row=0.12) -> np.ndarray:
        """ToyConn.scala:22-47.  Built with the host libm: the JVM's Math.pow/sin/cos may differ in the
        last ulp, so the matrix is an engine INPUT rather than something the engine reproduces."""
        n = len(countries)
        W = np.zeros((n, n))
        for i, ci in enumerate(countries):
            row_sum = 0.0
            for j, cj in enumerate(countries):
                if i != j:
                    d = max(300.0, Geo.haversine_km(ci.lat, ci.lon, cj.lat, cj.lon))
                    f = ToyConn.flight_mul.get((ci.name, cj.name), 1.0)
                    w = k * math.pow(float(ci.pop), a) * math.pow(float(cj.pop), b) * f / math.pow(d, g)
                    W[i, j] = w
                    row_sum += w
            if row_sum > 0:
                scale = min(1.0, cap_row / row_sum)
                for j in range(n):
                    if i != j:
                        W[i, j] *= scale
            W[i, i] = 0.0
        return W

    @staticmethod
    def gravity_matrix_device(countries, a=0.2, b=1.0, g=1.2, k=1e-6, cap_row=0.12, device: int = 0) -> np.ndarray:
        """ToyConn.scala:22-47 computed on the GPU (for large synthetic graphs).  Entries may differ from
        gravity_matrix() in the last place: the device library's pow/sin/cos/atan2 are not the host's."""
        n = len(countries)
        pop = i32([c.pop for c in countries]); lat = f64([c.lat for c in countries]); lon = f64([c.lon for c in countries])
        fl = None
        names = [c.name for c in countries]
        pairs = [(names.index(x), names.index(y), v) for (x, y), v in ToyConn.flight_mul.items() if x in names and y in names]
        if pairs:
            fl = np.ones((n, n))
            for x, y, v in pairs:
                fl[x, y] = v
        W = np.zeros((n, n))
        check(lib().epi_gravity_matrix(n, ptr(pop, C.c_int32), ptr(lat, C.c_double), ptr(lon, C.c_double),
                                       ptr(fl, C.c_double) if fl is not None else None, a, b, g, k, cap_row,
                                       ptr(W, C.c_double), device))
        return W

    @staticmethod
    def gravity_csr_device(countries, threshold: float, a=0.2, b=1.0, g=1.2, k=1e-6, cap_row=0.12, device: int = 0):
        """gravity_matrix_device followed by to_csr, all on the GPU (epi_gravity_csr): the dense matrix is never
        stored, only (rowptr, col, val) comes back.  For large synthetic graphs."""
        n = len(countries)
        pop = i32([c.pop for c in countries]); lat = f64([c.lat for c in countries]); lon = f64([c.lon for c in countries])
        fl = None
        names = [c.name for c in countries]
        pairs = [(names.index(x), names.index(y), v) for (x, y), v in ToyConn.flight_mul.items() if x in names and y in names]
        if pairs:
            fl = np.ones((n, n))
            for x, y, v in pairs:
                fl[x, y] = v
        flp = ptr(fl, C.c_double) if fl is not None else None
        rowptr = np.zeros(n + 1, dtype=np.int32)
        args = (n, ptr(pop, C.c_int32), ptr(lat, C.c_double), ptr(lon, C.c_double), flp, a, b, g, k, cap_row, threshold,
                ptr(rowptr, C.c_int32))
        check(lib().epi_gravity_csr(*args, None, None, device))
        col = np.zeros(max(1, int(rowptr[-1])), dtype=np.int32); val = np.zeros(max(1, int(rowptr[-1])))
        check(lib().epi_gravity_csr(*args, ptr(col, C.c_int32), ptr(val, C.c_double), device))
        return rowptr, col[:rowptr[-1]], val[:rowptr[-1]]

    @staticmethod
    def to_csr(W, threshold: float = 0.0):
        """Dense -> (rowptr, col, val) keeping the off-diagonal entries above `threshold`, columns ascending."""
        W = np.asarray(W, dtype=np.float64)
        keep = W > threshold
        np.fill_diagonal(keep, False)
        rowptr = np.zeros(W.shape[0] + 1, dtype=np.int32)
        rowptr[1:] = np.cumsum(keep.sum(axis=1))
        rows, cols = np.nonzero(keep)
        return rowptr, cols.astype(np.int32), W[rows, cols]

    @staticmethod
    def scale_edge(W, src: int, dst: int, factor: float) -> np.ndarray:  # ToyConn.scala:51-55
        A = np.array(W, dtype=np.float64, copy=True)
        A[src, dst] *= factor
        return A

    @staticmethod
    def scale_row(W, row: int, factor: float) -> np.ndarray:  # ToyConn.scala:58-70
        A = f64(W)
        out = np.empty_like(A)
        check(lib().epi_scale_row(ptr(A, C.c_double), C.c_int32(A.shape[0]), C.c_int32(row), C.c_double(factor),
                                  ptr(out, C.c_double)))
        return out


ToyConn.W = ToyConn.gravity_matrix(ToyConn.C)
ToyConn.W2 = ToyConn.scale_edge(ToyConn.W, 5, 3, 3.0)  # ToyConn.scala:56


# ------------------------------------------------------------------------------------------------
# Validation.scala
# ------------------------------------------------------------------------------------------------
class Validation:
    @dataclass
    class Invariants:
        pop: int
        t_max: int

    @dataclass
    class Metrics:
        arrival: int
        peak_i: float
        final_i: float

    @staticmethod
    def _require(cond: bool, msg: str):
        if not cond:
            raise IllegalArgumentException(_capi.EPI_ERR_ASSERT_STATE, "requirement failed: " + msg)

    @staticmethod
    def assert_state(s: State, inv: "Validation.Invariants") -> None:  # Validation.scala:6-11
        Validation._require(s.s >= 0 and s.i >= 0 and s.r >= 0 and s.d >= 0 and s.v >= 0, "neg compartment: %s" % (s,))
        total = s.s + s.i + s.r + s.d + s.v
        Validation._require(int(math.floor(total + 0.5)) == inv.pop, "mass not conserved: %s != %s" % (total, inv.pop))
        Validation._require(0 <= s.t <= inv.t_max, "t out of bounds: %d not in [0, %d]" % (s.t, inv.t_max))

    @staticmethod
    def arrival_time(i_series: Sequence[float], thresh: float = 10.0) -> int:  # Validation.scala:15-19
        for k, v in enumerate(i_series):
            if v >= thresh:
                return k
        return len(i_series)

    @staticmethod
    def peak_i(i_series: Sequence[float]) -> float:  # Validation.scala:21-22
        return 0.0 if len(i_series) == 0 else max(i_series)


# ------------------------------------------------------------------------------------------------
# JVM-compatible streams held on the host side of the boundary
# ------------------------------------------------------------------------------------------------
class _GaussStream:
    """scala.util.Random(seed).nextGaussian() sequence (env.scala:42,91), extended on demand."""

    def __init__(self, seed: int):
        self.seed, self.pos, self.buf = int(seed), 0, np.zeros(0)

    def next(self) -> float:
        if self.pos >= len(self.buf):
            n = max(64, 2 * len(self.buf))
            out = np.zeros(n)
            check(lib().epi_java_gaussians(C.c_int64(self.seed), C.c_int32(n), ptr(out, C.c_double)))
            self.buf = out
        v = float(self.buf[self.pos])
        self.pos += 1
        return v


class _AgentStream:
    """Action choice of a never-trained DDQNAgent (DDQNAgent.scala:12-32); -1 marks the argmax-Q branch."""

    def __init__(self, hp: HyperParams):
        self.hp, self.pos, self.buf = hp, 0, np.zeros(0, dtype=np.int32)

    def next(self) -> int:
        if self.pos >= len(self.buf):
            n = max(256, 2 * len(self.buf))
            out = np.zeros(n, dtype=np.int32)
            chp = self.hp._c()
            check(lib().epi_agent_actions(C.byref(chp), C.c_int32(n), ptr(out, C.c_int32)))
            self.buf = out
        v = int(self.buf[self.pos])
        self.pos += 1
        return v


def mlp_init(hidden: int = 128):
    """Untrained network of MLP.scala:5-9."""
    W1 = np.zeros((hidden, 7)); W2 = np.zeros((hidden, hidden)); W3 = np.zeros((8, hidden))
    check(lib().epi_mlp_init(7, hidden, 8, ptr(W1, C.c_double), ptr(W2, C.c_double), ptr(W3, C.c_double)))
    return W1, W2, W3


def mlp_greedy(states: Sequence[State], weights, clip_q: float = 10.0):
    """DDQNAgent.actGreedy for a batch of states: (q[B][8], action[B])."""
    W1, W2, W3 = (f64(w) for w in weights)
    arr = (_capi.CState * len(states))(*[s._c() for s in states])
    q = np.zeros((len(states), 8)); act = np.zeros(len(states), dtype=np.int32)
    check(lib().epi_mlp_greedy_batch(arr, C.c_int32(len(states)), C.c_int32(W1.shape[0]), ptr(W1, C.c_double), None,
                                     ptr(W2, C.c_double), None, ptr(W3, C.c_double), None, C.c_double(clip_q),
                                     ptr(q, C.c_double), ptr(act, C.c_int32)))
    return q, act


# ------------------------------------------------------------------------------------------------
# DDQNAgent.scala / MLP.scala / ReplayBuffer.scala: learners on the device
# ------------------------------------------------------------------------------------------------
@dataclass
class Transition:  # ReplayBuffer.scala:8
    s: State
    a: int
    r: float
    s2: State
    done: bool


class Agents:
    """A set of independent DDQN learners (one per country, WorldToy.scala:46-65) held by one device handle:
    replay buffers, online and target networks, DDQNAgent.learn for any subset of them in one set of launches."""

    def __init__(self, hps: Sequence[HyperParams], max_grad_accum: int = 4, device: int = 0):
        self.hps = list(hps)
        self.n = len(self.hps)
        arr = (_capi.LearnHyper * self.n)(*[
            _capi.LearnHyper(h.gamma, h.lr, h.clip_q, h.grad_clip, h.ema_alpha, h.hidden, h.replay_capacity,
                             h.target_update_every, h.batch_size * max(1, max_grad_accum)) for h in self.hps])
        h = lib().epi_agents_create(C.c_int32(self.n), arr, C.c_int32(device))
        if not h:
            _capi.raise_last()
        self._h = C.c_void_p(h)

    def close(self):
        if getattr(self, "_h", None):
            lib().epi_agents_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _ids(ids, count):
        return (None, count) if ids is None else (ptr(i32(ids), C.c_int32), len(ids))

    def observe(self, transitions: Sequence[Transition], ids: Optional[Sequence[int]] = None) -> None:
        """DDQNAgent.observe (:34) for one transition per listed agent (default: agents 0..len-1)."""
        k = len(transitions)
        idp, _ = Agents._ids(ids, k)
        s = (_capi.CState * k)(*[t.s._c() for t in transitions])
        s2 = (_capi.CState * k)(*[t.s2._c() for t in transitions])
        a = i32([t.a for t in transitions]); r = f64([t.r for t in transitions])
        d = np.ascontiguousarray([1 if t.done else 0 for t in transitions], dtype=np.uint8)
        check(lib().epi_agents_observe(self._h, idp, C.c_int32(k), s, ptr(a, C.c_int32), ptr(r, C.c_double), s2,
                                       ptr(d, C.c_uint8)))

    def observe_world(self, world: "WorldToy") -> None:
        """DDQNAgent.observe for the world's last step, filed on the device (epi_agents_observe_world): agent c gets
        the transition of country c of every instance; no state crosses the bus."""
        check(lib().epi_agents_observe_world(self._h, world._h))

    def replay_size(self) -> np.ndarray:
        out = np.zeros(self.n, dtype=np.int32)
        check(lib().epi_agents_replay_size(self._h, ptr(out, C.c_int32)))
        return out

    def sample(self, agent: int, n: int) -> np.ndarray:
        """Buffer positions ReplayBuffer.sample(n) returns (ReplayBuffer.scala:21-22); advances the shuffle stream."""
        out = np.zeros(max(n, 1), dtype=np.int32); k = C.c_int32(0)
        check(lib().epi_agents_sample(self._h, C.c_int32(agent), C.c_int32(n), ptr(out, C.c_int32), C.byref(k)))
        return out[:k.value]

    def learn(self, steps: Sequence[int], batch: Optional[Sequence[int]] = None, grad_accum=4,
              ids: Optional[Sequence[int]] = None) -> Tuple[np.ndarray, np.ndarray]:
        """DDQNAgent.learn(batch, gradAccumSteps) (:51-65) for the listed agents; steps[k] is each agent's act()
        counter.  Returns (loss, emaLossNow) per listed agent."""
        idp, k = Agents._ids(ids, len(steps))
        who = list(range(k)) if ids is None else list(ids)
        b = i32([self.hps[i].batch_size for i in who] if batch is None else batch)
        ga = i32([grad_accum] * k if np.isscalar(grad_accum) else grad_accum)
        st = np.ascontiguousarray(steps, dtype=np.int64)
        loss = np.zeros(k); ema = np.zeros(k)
        check(lib().epi_agents_learn(self._h, idp, C.c_int32(k), ptr(b, C.c_int32), ptr(ga, C.c_int32),
                                     st.ctypes.data_as(C.POINTER(C.c_int64)), ptr(loss, C.c_double), ptr(ema, C.c_double)))
        return loss, ema

    def q_values(self, states: Sequence[State], ids: Optional[Sequence[int]] = None, target: bool = False):
        """(q[k][8], first argmax[k]) of one state per listed agent (DDQNAgent.qValues / actGreedy, :22-23,70)."""
        k = len(states)
        idp, _ = Agents._ids(ids, k)
        arr = (_capi.CState * k)(*[s._c() for s in states])
        q = np.zeros((k, 8)); g = np.zeros(k, dtype=np.int32)
        check(lib().epi_agents_q_values(self._h, idp, C.c_int32(k), arr, C.c_int32(int(target)), ptr(q, C.c_double),
                                        ptr(g, C.c_int32)))
        return q, g

    def weights(self, agent: int, target: bool = False):
        """[W1[h][7], b1[h], W2[h][h], b2[h], W3[8][h], b3[8]] of one agent's online (or target) network."""
        h = self.hps[agent].hidden
        W = [np.zeros((h, 7)), np.zeros(h), np.zeros((h, h)), np.zeros(h), np.zeros((8, h)), np.zeros(8)]
        check(lib().epi_agents_get_weights(self._h, C.c_int32(agent), C.c_int32(int(target)), *[ptr(w, C.c_double) for w in W]))
        return W

    def set_weights(self, agent: int, W, target: bool = False) -> None:
        W = [f64(w) for w in W]
        check(lib().epi_agents_set_weights(self._h, C.c_int32(agent), C.c_int32(int(target)), *[ptr(w, C.c_double) for w in W]))

    def set_timing(self, on: bool):
        check(lib().epi_agents_set_timing(self._h, C.c_int32(int(on))))

    def kernel_time(self):
        """(device ms of the learn launches, learn calls, samples trained on) since timing was switched on."""
        ms = C.c_double(0); k = C.c_int64(0); n = C.c_int64(0)
        check(lib().epi_agents_kernel_time(self._h, C.byref(ms), C.byref(k), C.byref(n)))
        return ms.value, k.value, n.value


class DDQNAgent:
    """DDQNAgent.scala:6-70 for one agent: its own epsilon stream (Random(9)) on the host, networks and learn
    step on the device.  `pool`/`index` place it in a shared Agents handle (WorldToy does that for its nodes)."""

    def __init__(self, hp: HyperParams, state_size: int = 7, action_size: int = 8, pool: Optional[Agents] = None,
                 index: int = 0, device: int = 0):
        assert state_size == 7 and action_size == 8, "State.features has 7 entries and Action 8 values"
        self.hp = hp
        self._pool = pool if pool is not None else Agents([hp], device=device)
        self._k = index if pool is not None else 0
        self.steps = 0
        self._rnd = _JavaRandom(9)

    @property
    def epsilon_now(self) -> float:  # :16-19,67
        frac = min(1.0, float(self.steps) / max(1, self.hp.epsilon_decay_steps))
        return self.hp.epsilon_start + (self.hp.epsilon_end - self.hp.epsilon_start) * frac

    def act_greedy(self, s: State) -> int:  # :22-23
        return int(self._pool.q_values([s], [self._k])[1][0])

    def act_with_epsilon(self, s: State, eps: float, count_step: bool = False) -> int:  # :26-30
        if count_step:
            self.steps += 1
        if self._rnd.next_double() < eps:
            return self._rnd.next_int(8)
        return self.act_greedy(s)

    def explores(self, eps: float, count_step: bool = False) -> int:
        """The random branch of act_with_epsilon without the network: 0..7, or -1 where argmax Q is taken."""
        if count_step:
            self.steps += 1
        return self._rnd.next_int(8) if self._rnd.next_double() < eps else -1

    def act(self, s: State) -> int:  # :32 -- epsilon is an argument: read before the call counts the step
        return self.act_with_epsilon(s, self.epsilon_now, count_step=True)

    def observe(self, tr: Transition) -> None:  # :34
        self._pool.observe([tr], [self._k])

    def learn(self, batch: int, grad_accum_steps: int = 4) -> float:  # :51-65
        loss, ema = self._pool.learn([self.steps], [batch], grad_accum_steps, [self._k])
        self._ema = float(ema[0])
        return float(loss[0])

    @property
    def ema_loss_now(self) -> float:  # :68
        return getattr(self, "_ema", 0.0)

    @property
    def replay_size(self) -> int:  # :69
        return int(self._pool.replay_size()[self._k])

    def q_values(self, s: State) -> np.ndarray:  # :70
        return self._pool.q_values([s], [self._k])[0][0]


class _JavaRandom:
    """java.util.Random on the host side of the boundary (48-bit LCG, JDK javadoc), for the agent's own draws."""

    def __init__(self, seed: int):
        self.state = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1)

    def _next(self, bits: int) -> int:
        self.state = (self.state * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        v = self.state >> (48 - bits)
        return v - (1 << 32) if v >= (1 << 31) else v  # the (int) cast

    def next_int(self, bound: int) -> int:
        r = self._next(31)
        m = bound - 1
        if bound & m == 0:
            return (bound * r) >> 31
        u = r
        while True:
            r = u % bound
            if u - r + m < (1 << 31):
                return r
            u = self._next(31)

    def next_double(self) -> float:
        return ((self._next(26) << 27) + self._next(27)) * (1.0 / (1 << 53))


# ------------------------------------------------------------------------------------------------
# env.scala:32-255
# ------------------------------------------------------------------------------------------------
class EpidemicEnv:
    def __init__(self, base_beta=0.35, base_gamma=0.12, base_ifr=0.008, noise=0.05, reward_clip=5.0, risk_factor=0.0,
                 country_name="Unknown", seed=42, profile: Optional[Geo.Country] = None):
        self.base_beta, self.base_gamma, self.base_ifr, self.noise = base_beta, base_gamma, base_ifr, noise
        self.reward_clip, self.risk_factor, self.country_name, self.seed = reward_clip, risk_factor, country_name, seed
        self.country_profile = profile if profile is not None else Geo.CountryProfiles.get_profile(country_name)
        self._rnd = _GaussStream(seed)

    def reset(self, s0: State) -> State:
        return s0

    def _c(self) -> _capi.EnvParams:
        return _capi.EnvParams(self.base_beta, self.base_gamma, self.base_ifr, self.noise, self.reward_clip,
                               self.risk_factor)

    @dataclass
    class EnvParams:
        beta: float
        gamma: float
        ifr: float
        noise: float

    @property
    def params(self) -> "EpidemicEnv.EnvParams":  # env.scala:236-237
        return EpidemicEnv.EnvParams(self.base_beta, self.base_gamma, self.base_ifr, self.noise)

    def step(self, st: State, act) -> Tuple[State, float, bool]:  # env.scala:58-140
        nxt, r, d = EpidemicEnv.step_batch([self], [st], [int(act)], [self._rnd.next()])
        return nxt[0], float(r[0]), bool(d[0])

    @staticmethod
    def step_batch(envs, states, actions, gauss):
        """B independent EpidemicEnv.step calls in one launch (epi_env_step_batch)."""
        B = len(envs)
        env_arr = (_capi.EnvParams * B)(*[e._c() for e in envs])
        prof_arr = (_capi.Profile * B)(*[e.country_profile._c() for e in envs])
        st_arr = (_capi.CState * B)(*[s._c() for s in states])
        out = (_capi.CState * B)()
        a = i32(actions); g = f64(gauss)
        r = np.zeros(B); d = np.zeros(B, dtype=np.uint8)
        check(lib().epi_env_step_batch(env_arr, prof_arr, st_arr, ptr(a, C.c_int32), ptr(g, C.c_double), out,
                                       ptr(r, C.c_double), ptr(d, C.c_uint8), C.c_int32(B)))
        return [State._from_c(o) for o in out], r, d

    @staticmethod
    def allocate_int(total: int, weights: Sequence[float]) -> List[int]:  # env.scala:241-254
        w = f64(weights)
        out = np.zeros(len(w), dtype=np.int32)
        check(lib().epi_alloc_int(C.c_int32(int(total)), ptr(w, C.c_double), C.c_int32(len(w)), ptr(out, C.c_int32)))
        return out.tolist()

    @staticmethod
    def allocate_int_batch(totals, weights) -> np.ndarray:
        w = f64(weights); t = i32(totals)
        out = np.zeros(w.shape, dtype=np.int32)
        check(lib().epi_alloc_int_batch(ptr(t, C.c_int32), ptr(w, C.c_double), C.c_int32(w.shape[1]),
                                        C.c_int32(w.shape[0]), ptr(out, C.c_int32)))
        return out


# ------------------------------------------------------------------------------------------------
# WorldToy.scala
# ------------------------------------------------------------------------------------------------
def _csr_or_dense(conn):
    """Accepts a dense 2-D array or a (rowptr, col, val) triple."""
    if isinstance(conn, tuple):
        rp, col, val = i32(conn[0]), i32(conn[1]), f64(conn[2])
        return None, rp, col, val
    return f64(conn), None, None, None


class WorldToy:
    """WorldToy.scala:11-235 with the country list as an input (default ToyConn.C) and an optional
    `batch` of identical, independently evolving instances stepped in one launch."""

    @dataclass
    class Node:
        idx: int
        name: str
        pop: int
        env: EpidemicEnv
        hyper_params: HyperParams
        s: State
        last_a: int = 0
        arrival: int = -1     # first step index with I >= 10 since the last reset (Validation.arrivalTime), -1 = not yet
        peak_i: float = 0.0   # running Validation.peakI
        agent: Optional["DDQNAgent"] = None   # WorldToy.scala:50; present when the world learns
        last_learn_step: int = 0

    def __init__(self, hp: HyperParams, base: State, seed_idx: int = 0, conn=None,
                 make_env: Callable[[], EpidemicEnv] = lambda: EpidemicEnv(),
                 make_env_for: Optional[Callable[[str, int], EpidemicEnv]] = None, validate: bool = False,
                 countries: Optional[Sequence[Geo.Country]] = None, batch: int = 1, device: int = 0,
                 country_hyper_params: Optional[dict] = None, learning: bool = True, device_transitions: bool = True):
        self.hp, self.base, self.seed_idx, self.validate, self.batch = hp, base, seed_idx, validate, batch
        self.device_transitions = device_transitions   # file the agents' transitions from the device planes (no round trip)
        self._device = device
        self.learning = bool(learning) and batch == 1
        self.g_step = 0
        self.last_losses: List[Optional[float]] = []
        self.countries = list(countries) if countries is not None else list(ToyConn.C)
        n = self.n = len(self.countries)
        conn = ToyConn.W if conn is None else conn
        envs = [make_env_for(c.name, c.pop) if make_env_for else make_env() for c in self.countries]
        self._envs = envs
        pops = i32([c.pop for c in self.countries])
        dense, rp, col, val = _csr_or_dense(conn)
        env_arr = (_capi.EnvParams * n)(*[e._c() for e in envs])
        prof_arr = (_capi.Profile * n)(*[e.country_profile._c() for e in envs])
        seeds = i32([e.seed for e in envs])
        d = _capi.WorldDesc()
        d.n, d.batch, d.pop = n, batch, ptr(pops, C.c_int32)
        d.conn = ptr(dense, C.c_double) if dense is not None else None
        d.csr_rowptr = ptr(rp, C.c_int32) if rp is not None else None
        d.csr_col = ptr(col, C.c_int32) if col is not None else None
        d.csr_val = ptr(val, C.c_double) if val is not None else None
        d.seed_idx, d.validate, d.base = seed_idx, int(validate), base._c()
        d.env, d.prof, d.env_seeds, d.hp, d.device = env_arr, prof_arr, ptr(seeds, C.c_int32), hp._c(), device
        h = lib().epi_world_create(C.byref(d))
        if not h:
            _capi.raise_last()
        self._h = C.c_void_p(h)
        self._agents = [_AgentStream(hp) for _ in range(n)]
        self._mlp = None
        self._steps_done = 0
        st0 = self._get_states()
        chp = country_hyper_params or {}
        self.nodes = [WorldToy.Node(i, c.name, c.pop, envs[i], chp.get(c.name, hp), st0[0][i])
                      for i, c in enumerate(self.countries)]
        self._pool = None

    def close(self):
        if getattr(self, "_h", None):
            lib().epi_world_destroy(self._h)
            self._h = None

    def result_arrays(self):
        """(states, rewards): numpy views of the handle's own page-locked result arrays.  Passed as out_states /
        out_rewards to step_arrays they are filled by the device without a further copy; they are overwritten by
        the next step and die with the world."""
        ps = C.c_void_p(); pr = C.c_void_p()
        check(lib().epi_world_result_buffers(self._h, C.byref(ps), C.byref(pr), None))
        cells = self.batch * self.n
        st = np.ctypeslib.as_array(C.cast(ps, C.POINTER(C.c_uint8)), shape=(cells * WorldToy.STATE_DTYPE.itemsize,))
        rw = np.ctypeslib.as_array(C.cast(pr, C.POINTER(C.c_double)), shape=(cells,))
        return st.view(WorldToy.STATE_DTYPE).reshape(self.batch, self.n), rw.reshape(self.batch, self.n)

    def packed_array(self):
        """numpy view of the handle's page-locked array of 32-byte state records (PACKED_DTYPE [batch][n]); passed as
        out_states to step_arrays_packed it is filled by the device without a further copy."""
        ps = C.c_void_p()
        check(lib().epi_world_packed_buffer(self._h, C.byref(ps)))
        cells = self.batch * self.n
        st = np.ctypeslib.as_array(C.cast(ps, C.POINTER(C.c_uint8)), shape=(cells * WorldToy.PACKED_DTYPE.itemsize,))
        return st.view(WorldToy.PACKED_DTYPE).reshape(self.batch, self.n)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state access -------------------------------------------------------------------------
    def _get_states(self) -> List[List[State]]:
        arr = (_capi.CState * (self.batch * self.n))()
        check(lib().epi_world_get_states(self._h, arr))
        return [[State._from_c(arr[b * self.n + c]) for c in range(self.n)] for b in range(self.batch)]

    def snapshot_states(self) -> List[State]:  # WorldToy.scala:75
        return self._get_states()[0] if self.batch == 1 else self._get_states()

    def restore_states(self, ss) -> None:  # :76
        flat = ss if self.batch == 1 else [s for inst in ss for s in inst]
        arr = (_capi.CState * (self.batch * self.n))(*[s._c() for s in flat])
        check(lib().epi_world_set_states(self._h, arr))
        self._sync_nodes()

    def reset_to_initial(self) -> None:  # :77
        check(lib().epi_world_reset(self._h))
        self._sync_nodes()
        self._steps_done = 0
        for node in self.nodes:
            node.arrival, node.peak_i = -1, 0.0

    def _track(self, states):
        """Arrival / peak bookkeeping of instance 0 for the snapshot stream (Validation.scala:15-22)."""
        for node, s in zip(self.nodes, states):
            if node.arrival < 0 and s.i >= 10.0:
                node.arrival = self._steps_done
            node.peak_i = max(node.peak_i, s.i)
        self._steps_done += 1

    def _sync_nodes(self):
        arr = (_capi.CState * (self.batch * self.n))()
        check(lib().epi_world_get_states(self._h, arr))
        for c, node in enumerate(self.nodes):
            node.s = State._from_c(arr[c])

    def last_flows(self):
        ms = np.zeros((self.batch, self.n), dtype=np.int32); mi = np.zeros((self.batch, self.n), dtype=np.int32)
        check(lib().epi_world_last_flows(self._h, ptr(ms, C.c_int32), ptr(mi, C.c_int32)))
        return ms, mi

    # -- stepping -----------------------------------------------------------------------------
    def step_actions(self, actions, gauss=None, t_max: Optional[int] = None):
        """One stepOne with caller-chosen actions[batch][n]; returns (states, rewards) arrays."""
        a = i32(np.asarray(actions).reshape(self.batch, self.n))
        g = None if gauss is None else f64(np.asarray(gauss).reshape(self.batch, self.n))
        out = (_capi.CState * (self.batch * self.n))()
        r = np.zeros((self.batch, self.n))
        check(lib().epi_world_step(self._h, ptr(a, C.c_int32), ptr(g, C.c_double) if g is not None else None,
                                   C.c_int32(self.base.t_max if t_max is None else t_max), out, ptr(r, C.c_double)))
        states = [[State._from_c(out[b * self.n + c]) for c in range(self.n)] for b in range(self.batch)]
        for node, s, act in zip(self.nodes, states[0], a[0]):
            node.s, node.last_a = s, int(act)
        self._track(states[0])
        return states, r

    # numpy image of one State record as the C ABI lays it out (epi_state, 64 bytes)
    STATE_DTYPE = np.dtype([("s", "f8"), ("i", "f8"), ("r", "f8"), ("d", "f8"), ("v", "f8"), ("hosp_cap", "f8"),
                            ("t", "i4"), ("t_max", "i4"), ("seed", "i4"), ("pad_", "i4")])

    def step_arrays(self, actions, gauss=None, t_max: Optional[int] = None, out_states=None, out_rewards=None):
        """step_actions without per-State Python objects, for batched rollouts: actions int32 [batch][n],
        gauss float64 [batch][n] or None; fills and returns (out_states: STATE_DTYPE [batch][n],
        out_rewards: float64 [batch][n]).  `nodes` is not refreshed."""
        a = i32(np.asarray(actions).reshape(self.batch, self.n))
        g = None if gauss is None else f64(np.asarray(gauss).reshape(self.batch, self.n))
        if out_states is None:
            out_states = np.empty((self.batch, self.n), dtype=WorldToy.STATE_DTYPE)
        if out_rewards is None:
            out_rewards = np.empty((self.batch, self.n))
        assert out_states.dtype == WorldToy.STATE_DTYPE and out_states.flags.c_contiguous
        assert out_rewards.dtype == np.float64 and out_rewards.flags.c_contiguous
        check(lib().epi_world_step(self._h, ptr(a, C.c_int32), ptr(g, C.c_double) if g is not None else None,
                                   C.c_int32(self.base.t_max if t_max is None else t_max),
                                   C.cast(out_states.ctypes.data, C.POINTER(_capi.CState)), ptr(out_rewards, C.c_double)))
        return out_states, out_rewards

    def step_resident(self, actions, gauss=None, t_max: Optional[int] = None):
        """step_arrays with nothing returned: epi_world_step with NULL out_states / out_rewards.  The states stay on
        the device (read them with snapshot_states / epi_world_get_states, or let the learners file the transitions
        with epi_agents_observe_world); only the actions -- and `gauss`, if given -- cross the bus."""
        a = i32(np.asarray(actions).reshape(self.batch, self.n))
        g = None if gauss is None else f64(np.asarray(gauss).reshape(self.batch, self.n))
        check(lib().epi_world_step(self._h, ptr(a, C.c_int32), ptr(g, C.c_double) if g is not None else None,
                                   C.c_int32(self.base.t_max if t_max is None else t_max), None, None))

    # the states of step_arrays in half the bytes: whole-number compartments as int32, t_max and seed left out (they
    # are the base state's for every instance)
    PACKED_DTYPE = np.dtype([("s", "i4"), ("i", "i4"), ("r", "i4"), ("d", "i4"), ("v", "i4"), ("t", "i4"),
                             ("hosp_cap", "f8")])

    def step_arrays_packed(self, actions, gauss=None, t_max: Optional[int] = None, out_states=None, out_rewards=None):
        """step_arrays with the states as PACKED_DTYPE records (32 bytes instead of 64 per instance and country)."""
        a = i32(np.asarray(actions).reshape(self.batch, self.n))
        g = None if gauss is None else f64(np.asarray(gauss).reshape(self.batch, self.n))
        if out_states is None:
            out_states = np.empty((self.batch, self.n), dtype=WorldToy.PACKED_DTYPE)
        if out_rewards is None:
            out_rewards = np.empty((self.batch, self.n))
        assert out_states.dtype == WorldToy.PACKED_DTYPE and out_states.flags.c_contiguous
        assert out_rewards.dtype == np.float64 and out_rewards.flags.c_contiguous
        check(lib().epi_world_step_packed(self._h, ptr(a, C.c_int32), ptr(g, C.c_double) if g is not None else None,
                                          C.c_int32(self.base.t_max if t_max is None else t_max),
                                          C.c_void_p(out_states.ctypes.data), ptr(out_rewards, C.c_double)))
        return out_states, out_rewards

    def set_timing(self, on: bool):
        check(lib().epi_world_set_timing(self._h, C.c_int32(int(on))))

    def kernel_time(self):
        """(summed episode-kernel ms, launches, country-steps) since timing was switched on."""
        ms = C.c_double(0); k = C.c_int64(0); cs = C.c_int64(0)
        check(lib().epi_world_kernel_time(self._h, C.byref(ms), C.byref(k), C.byref(cs)))
        return ms.value, k.value, cs.value

    def screen_stats(self) -> Tuple[int, int]:
        """(states the argmax screen looked at, states it sent to the exact evaluation) since set_mlp."""
        a = C.c_int64(0); b = C.c_int64(0)
        check(lib().epi_world_screen_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_mlp(self, W1, W2, W3, b1=None, b2=None, b3=None, clip_q=None):
        W1, W2, W3 = f64(W1), f64(W2), f64(W3)
        bs = [None if b is None else f64(b) for b in (b1, b2, b3)]
        check(lib().epi_world_set_mlp(self._h, C.c_int32(W1.shape[0]), ptr(W1, C.c_double),
                                      ptr(bs[0], C.c_double) if bs[0] is not None else None, ptr(W2, C.c_double),
                                      ptr(bs[1], C.c_double) if bs[1] is not None else None, ptr(W3, C.c_double),
                                      ptr(bs[2], C.c_double) if bs[2] is not None else None,
                                      C.c_double(self.hp.clip_q if clip_q is None else clip_q)))
        self._mlp = (W1, W2, W3)

    def _ensure_untrained_mlp(self):
        if self._mlp is None:
            self.set_mlp(*mlp_init(self.hp.hidden))

    def _ensure_agents(self):
        """One DDQNAgent per node (WorldToy.scala:49-50) in a shared device handle."""
        if self._pool is None:
            self._pool = Agents([nd.hyper_params for nd in self.nodes], max_grad_accum=4, device=self._device)
            for nd in self.nodes:
                nd.agent = DDQNAgent(nd.hyper_params, pool=self._pool, index=nd.idx)
            if self.device_transitions:
                check(lib().epi_world_keep_transitions(self._h, C.c_int32(1)))

    # WorldToy.scala:157-162
    LEARNING_INTENSITY = {"India": 2, "Japan": 1, "Germany": 1}

    def step_one(self, t_max: int) -> List[Tuple[str, State, float]]:
        """WorldToy.stepOne (:83-170).  With `learning` (the default for a single world) every node acts through
        its own DDQNAgent, observes its transition and learns on its own schedule (:147-167), all network
        arithmetic on the device.  Without it the agents stay untrained and only their epsilon-greedy streams
        are followed."""
        assert self.batch == 1, "step_one mirrors the single-world API; use step_actions for batches"
        if not self.learning:
            acts = [ag.next() for ag in self._agents]
            if any(a < 0 for a in acts):
                self._ensure_untrained_mlp()
                _, greedy = mlp_greedy([nd.s for nd in self.nodes], self._mlp, self.hp.clip_q)
                acts = [int(g) if a < 0 else a for a, g in zip(acts, greedy)]
            states, r = self.step_actions([acts], None, t_max)
            return [(nd.name, s, float(rv)) for nd, s, rv in zip(self.nodes, states[0], r[0])]
        self._ensure_agents()
        before = [nd.s for nd in self.nodes]
        acts = [nd.agent.explores(nd.agent.epsilon_now, count_step=True) for nd in self.nodes]   # :88
        greedy_ids = [k for k, a in enumerate(acts) if a < 0]
        if greedy_ids:
            _, g = self._pool.q_values([before[k] for k in greedy_ids], greedy_ids)
            for k, a in zip(greedy_ids, g):
                acts[k] = int(a)
        states, r = self.step_actions([acts], None, t_max)
        if self.device_transitions:
            self._pool.observe_world(self)     # :99, from the planes the world kept on the device
        else:
            self._observe_from_host(before, acts, states, r)
        self.g_step += 1
        return self._learn_due(states, r)

    def _observe_from_host(self, before, acts, states, r):
        """agent.observe (:99) with the transitions rebuilt on the host from the returned states and flows."""
        ms, mi = self.last_flows()
        trs = []
        for k, nd in enumerate(self.nodes):
            # the post-env.step state the transition holds (:90,99): the returned state without the mobility
            # flows of :132-136 and with env.step's own clock
            s3 = states[0][k]
            s2 = s3.copy(s=s3.s - float(ms[0][k]), i=s3.i - float(mi[0][k]), t=before[k].t + 1)
            nn = max(1.0, s2.n)
            done = (s2.t >= s2.t_max) or (s2.i / nn < 1e-5) or (s2.d / nn > 0.02)   # env.scala:137
            trs.append(Transition(before[k], acts[k], float(r[0][k]), s2, done))
        self._pool.observe(trs)

    def _learn_due(self, states, r):
        """The per-country learning schedule of WorldToy.scala:147-167."""
        sizes = self._pool.replay_size()
        due = [nd for nd in self.nodes
               if self.g_step - nd.last_learn_step >= nd.hyper_params.learn_every
               and sizes[nd.idx] >= nd.hyper_params.batch_size * 4]
        self.last_losses = [None] * self.n
        if due:
            ids = [nd.idx for nd in due]
            loss, ema = self._pool.learn([nd.agent.steps for nd in due], [nd.hyper_params.batch_size for nd in due],
                                         [WorldToy.LEARNING_INTENSITY.get(nd.name, 1) for nd in due], ids)
            for nd, l, e in zip(due, loss, ema):
                nd.last_learn_step = self.g_step
                nd.agent._ema = float(e)
                self.last_losses[nd.idx] = float(l)
        return [(nd.name, s, float(rv)) for nd, s, rv in zip(self.nodes, states[0], r[0])]

    def rollout_greedy(self, steps: int, t_max: int, want_series: bool = True, want_states: bool = True):
        """`steps` stepOneGreedy calls kept on the device (epi_world_rollout_greedy): every node asks its own agent's
        network when the world learns (WorldToy.scala:177), the untrained network otherwise.  Returns a dict with
        i_series [steps][batch][n] (optional), arrival [batch][n], peak [batch][n], states (STATE_DTYPE [batch][n])."""
        agents = None
        if self.learning and self._pool is not None:
            agents = self._pool._h
        else:
            self._ensure_untrained_mlp()
        B, n = self.batch, self.n
        series = np.zeros((steps, B, n)) if want_series else None
        arr = np.zeros((B, n), dtype=np.int32); pk = np.zeros((B, n))
        st = np.empty((B, n), dtype=WorldToy.STATE_DTYPE) if want_states else None
        check(lib().epi_world_rollout_greedy(self._h, agents, C.c_int32(steps), C.c_int32(t_max),
                                             series.ctypes.data if want_series else None, arr.ctypes.data, pk.ctypes.data,
                                             st.ctypes.data if want_states else None))
        self._sync_nodes()
        return dict(i_series=series, arrival=arr, peak=pk, states=st)

    def evaluate_greedy(self, max_eval_steps: int = 800) -> Dict[str, float]:
        """The periodic greedy evaluation of the trainer (MutliTrainer.scala:128-161): snapshot, reset, `max_eval_steps`
        greedy steps on the device, metrics of the I series per country, restore."""
        assert self.batch == 1
        saved = self.snapshot_states()
        self.reset_to_initial()
        out = self.rollout_greedy(max_eval_steps, max_eval_steps, want_series=True, want_states=False)
        metrics: Dict[str, float] = {}
        for k, nd in enumerate(self.nodes):
            series = out["i_series"][:, 0, k]
            norm = series / max(1.0, float(nd.pop))
            metrics["eval_arrival_%s" % nd.name] = float(Validation.arrival_time(series, 10.0))
            metrics["eval_peakI_%s" % nd.name] = float(Validation.peak_i(series))
            metrics["eval/arrivalN_%s" % nd.name] = float(Validation.arrival_time(norm, 1e-4))
            metrics["eval/peakI_norm_%s" % nd.name] = float(Validation.peak_i(norm))
            assert metrics["eval_arrival_%s" % nd.name] == out["arrival"][0, k]
            assert metrics["eval_peakI_%s" % nd.name] == out["peak"][0, k]
        self.restore_states(saved)
        return metrics

    def step_one_greedy(self, t_max: int) -> List[Tuple[str, State, float]]:  # :172-235
        if self.learning and self._pool is not None:   # every node asks its own, possibly trained, network (:177)
            _, g = self._pool.q_values([nd.s for nd in self.nodes])
            states, r = self.step_actions([[int(a) for a in g]], None, t_max)
            return [(nd.name, s, float(rv)) for nd, s, rv in zip(self.nodes, states[0], r[0])]
        self._ensure_untrained_mlp()
        out = (_capi.CState * (self.batch * self.n))()
        r = np.zeros((self.batch, self.n)); a = np.zeros((self.batch, self.n), dtype=np.int32)
        check(lib().epi_world_step_greedy(self._h, None, C.c_int32(t_max), out, ptr(r, C.c_double), ptr(a, C.c_int32)))
        states = [State._from_c(out[c]) for c in range(self.n)]
        for node, s, act in zip(self.nodes, states, a[0]):
            node.s, node.last_a = s, int(act)
        self._track(states)
        return [(nd.name, s, float(rv)) for nd, s, rv in zip(self.nodes, states, r[0])]


# ------------------------------------------------------------------------------------------------
# Calibrate.scala
# ------------------------------------------------------------------------------------------------
class ProfileTable:
    """Country profiles already marshalled into the C ABI's array of epi_profile records (and the populations into an
    int32 array): built once, it can be handed to any number of engines in place of the list of profile objects --
    marshalling 20,000 Python objects costs several times what epi_grid_create itself does."""

    def __init__(self, profiles, pops=None):
        self.n = len(profiles)
        self.array = (_capi.Profile * self.n)(*[p._c() if hasattr(p, "_c") else _capi.Profile(*p) for p in profiles])
        self.pops = i32(pops) if pops is not None else None

    def __len__(self):
        return self.n


def _grid_desc(hp, steps, base, seed_idx, pops, conn, profiles, betas, travel, t_arrive, t_peak, heterogeneous, device,
               names=None):
    """Fills an epi_grid_desc; returns it with the arrays it points into (keep them alive while it is used).
    `names` (country names) decide the learningIntensity of WorldToy.scala:157-162 for rollouts that learn."""
    n = len(pops)
    pops = profiles.pops if isinstance(profiles, ProfileTable) and profiles.pops is not None and pops is profiles.pops else i32(pops)
    dense, rp, col, val = _csr_or_dense(conn)
    if isinstance(profiles, ProfileTable):
        if len(profiles) != n:
            raise ValueError("ProfileTable of %d profiles for %d countries" % (len(profiles), n))
        prof_arr = profiles.array
    else:
        prof_arr = (_capi.Profile * n)(*[p._c() if hasattr(p, "_c") else _capi.Profile(*p) for p in profiles])
    ta = np.zeros(n); tp = np.zeros(n)
    ha = np.zeros(n, dtype=np.uint8); hk = np.zeros(n, dtype=np.uint8)
    if t_arrive:
        idx = np.fromiter(t_arrive.keys(), dtype=np.int64, count=len(t_arrive))
        ta[idx] = np.fromiter(t_arrive.values(), dtype=np.float64, count=len(t_arrive)); ha[idx] = 1
    if t_peak:
        idx = np.fromiter(t_peak.keys(), dtype=np.int64, count=len(t_peak))
        tp[idx] = np.fromiter(t_peak.values(), dtype=np.float64, count=len(t_peak)); hk[idx] = 1
    d = _capi.GridDesc()
    d.n, d.pop = n, ptr(pops, C.c_int32)
    d.conn = ptr(dense, C.c_double) if dense is not None else None
    d.csr_rowptr = ptr(rp, C.c_int32) if rp is not None else None
    d.csr_col = ptr(col, C.c_int32) if col is not None else None
    d.csr_val = ptr(val, C.c_double) if val is not None else None
    d.seed_idx, d.heterogeneous, d.base, d.prof, d.hp, d.steps = seed_idx, int(heterogeneous), base._c(), prof_arr, hp._c(), steps
    d.n_beta, d.betas, d.n_travel, d.travel = len(betas), ptr(betas, C.c_double), len(travel), ptr(travel, C.c_double)
    d.t_arrive, d.has_arrive, d.t_peak, d.has_peak = ptr(ta, C.c_double), ptr(ha, C.c_uint8), ptr(tp, C.c_double), ptr(hk, C.c_uint8)
    d.device = device
    # rollouts long enough for their agents to learn (WorldToy.scala:147-167)
    intensity = i32([WorldToy.LEARNING_INTENSITY.get(nm, 1) for nm in names] if names is not None else [1] * n)
    learn = _capi.LearnHyper(hp.gamma, hp.lr, hp.clip_q, hp.grad_clip, hp.ema_alpha, hp.hidden, hp.replay_capacity,
                             hp.target_update_every, hp.batch_size * int(max(intensity.max(), 1)))
    d.learn, d.learn_intensity = C.pointer(learn), ptr(intensity, C.c_int32)
    return d, (pops, dense, rp, col, val, prof_arr, ta, tp, ha, hk, betas, travel, learn, intensity)


class GridEngine:
    """Device-resident calibration sweep (epi_grid_*): build once, run slabs of grid points."""

    def __init__(self, hp: HyperParams, steps: int, base: State, seed_idx: int, pops, conn, profiles, betas, travel,
                 t_arrive=None, t_peak=None, heterogeneous=True, device: int = 0, names: Optional[Sequence[str]] = None):
        self.n = len(pops)
        self.betas, self.travel = f64(betas), f64(travel)
        self.G = len(self.betas) * len(self.travel)
        d, _keep = _grid_desc(hp, steps, base, seed_idx, pops, conn, profiles, self.betas, self.travel, t_arrive, t_peak,
                              heterogeneous, device, names)
        h = lib().epi_grid_create(C.byref(d))
        if not h:
            _capi.raise_last()
        self._h = C.c_void_p(h)

    def close(self):
        if getattr(self, "_h", None):
            lib().epi_grid_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, g_begin: int = 0, g_count: Optional[int] = None, want_metrics: bool = True):
        """Synchronous run of a slab; returns dict(loss, arrival, peak) as host arrays."""
        g_count = self.G - g_begin if g_count is None else g_count
        loss = np.zeros(g_count)
        arr = np.zeros((g_count, self.n), dtype=np.int32) if want_metrics else None
        pk = np.zeros((g_count, self.n)) if want_metrics else None
        check(lib().epi_grid_run(self._h, g_begin, g_count, loss.ctypes.data, arr.ctypes.data if want_metrics else None,
                                 pk.ctypes.data if want_metrics else None, 0, None))
        return dict(loss=loss, arrival=arr, peak=pk)

    def run_strided(self, g_begin: int, g_stride: int, g_count: int, want_metrics: bool = True):
        """Synchronous run of grid points g_begin + k * g_stride, k < g_count (an interleaved share of the grid)."""
        loss = np.zeros(g_count)
        arr = np.zeros((g_count, self.n), dtype=np.int32) if want_metrics else None
        pk = np.zeros((g_count, self.n)) if want_metrics else None
        check(lib().epi_grid_run_strided(self._h, g_begin, g_stride, g_count, loss.ctypes.data,
                                         arr.ctypes.data if want_metrics else None, pk.ctypes.data if want_metrics else None, 0, None))
        return dict(loss=loss, arrival=arr, peak=pk)

    def enqueue(self, g_begin: int, g_count: int, loss_dev_ptr: int, stream: int = 0, arrival_dev_ptr: int = 0,
                peak_dev_ptr: int = 0, g_stride: int = 1):
        """Asynchronous run into device buffers (raw addresses) on `stream` (a hipStream_t address)."""
        check(lib().epi_grid_run_strided(self._h, g_begin, g_stride, g_count, loss_dev_ptr or None, arrival_dev_ptr or None,
                                         peak_dev_ptr or None, 1, stream or None))

    def enqueue_noalloc(self, g_begin: int, g_count: int, loss_dev_ptr: int, stream: int = 0, arrival_dev_ptr: int = 0,
                        peak_dev_ptr: int = 0, g_stride: int = 1):
        """enqueue() through epi_grid_run_noalloc: per-episode state in the library's fixed work area, no
        allocation of any kind; limited to what that area holds (see include/epi.h)."""
        check(lib().epi_grid_run_noalloc(self._h, g_begin, g_stride, g_count, loss_dev_ptr or None, arrival_dev_ptr or None,
                                         peak_dev_ptr or None, stream or None))

    def status(self):
        check(lib().epi_grid_status(self._h))

    def screen_stats(self) -> Tuple[int, int]:
        """(states the argmax screen looked at, states it sent to the exact evaluation) by this engine so far."""
        a = C.c_int64(0); b = C.c_int64(0)
        check(lib().epi_grid_screen_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_timing(self, on: bool):
        check(lib().epi_grid_set_timing(self._h, int(on)))

    def kernel_time(self):
        """(total ms, launches, country-steps) of the episode kernel since set_timing(True)."""
        ms = C.c_double(0); k = C.c_int64(0); cs = C.c_int64(0)
        check(lib().epi_grid_kernel_time(self._h, C.byref(ms), C.byref(k), C.byref(cs)))
        return ms.value, k.value, cs.value


class MultiGridEngine:
    """The sweep over several GPUs from one process (epi_grid_multi_*): share k of the interleaved split runs on
    devices[k]; the loss pieces are all-gathered with RCCL; run() returns the whole surface and its first minimum."""

    def __init__(self, hp: HyperParams, steps: int, base: State, seed_idx: int, pops, conn, profiles, betas, travel,
                 t_arrive=None, t_peak=None, heterogeneous=True, devices: Sequence[int] = (0,),
                 names: Optional[Sequence[str]] = None):
        self.n = len(pops)
        self.betas, self.travel = f64(betas), f64(travel)
        self.G = len(self.betas) * len(self.travel)
        d, _keep = _grid_desc(hp, steps, base, seed_idx, pops, conn, profiles, self.betas, self.travel, t_arrive, t_peak,
                              heterogeneous, 0, names)
        dev = i32(list(devices))
        self.n_shares = len(dev)
        h = lib().epi_grid_multi_create(C.byref(d), ptr(dev, C.c_int32), C.c_int32(len(dev)))
        if not h:
            _capi.raise_last(_capi.EPI_ERR_DEVICE)
        self._h = C.c_void_p(h)

    def close(self):
        if getattr(self, "_h", None):
            lib().epi_grid_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ranks(self) -> Tuple[int, int]:
        """(communicator ranks = distinct devices, share slots per rank)."""
        r = C.c_int32(0); sl = C.c_int32(0)
        check(lib().epi_grid_multi_ranks(self._h, C.byref(r), C.byref(sl)))
        return r.value, sl.value

    def run(self, want_metrics: bool = True):
        loss = np.zeros(max(self.G, 1))
        arr = np.zeros((self.G, self.n), dtype=np.int32) if want_metrics else None
        pk = np.zeros((self.G, self.n)) if want_metrics else None
        bi = C.c_int32(0); bl = C.c_double(0)
        check(lib().epi_grid_multi_run(self._h, loss.ctypes.data, arr.ctypes.data if want_metrics else None,
                                       pk.ctypes.data if want_metrics else None, C.byref(bi), C.byref(bl)))
        return dict(loss=loss[:self.G], arrival=arr, peak=pk, best_idx=bi.value, best_loss=bl.value)

    def set_timing(self, on: bool):
        check(lib().epi_grid_multi_set_timing(self._h, int(on)))

    def times(self):
        """dict(share_kernel_ms[n_shares], gather_ms, wall_ms, runs) since set_timing(True)."""
        sk = np.zeros(self.n_shares); g = C.c_double(0); w = C.c_double(0); r = C.c_int64(0)
        check(lib().epi_grid_multi_times(self._h, ptr(sk, C.c_double), C.byref(g), C.byref(w), C.byref(r)))
        return dict(share_kernel_ms=sk, gather_ms=g.value, wall_ms=w.value, runs=r.value)

    def peer_access(self) -> np.ndarray:
        r, _ = self.ranks()
        m = np.zeros((r, r), dtype=np.uint8)
        check(lib().epi_grid_multi_peer_access(self._h, ptr(m, C.c_uint8)))
        return m

    def inject_failure(self, share: int):
        """Test hook: the next run() fails at this share's enqueue."""
        check(lib().epi_grid_multi_inject_failure(self._h, int(share)))


def grid_argmin(loss) -> Tuple[int, float]:
    loss = f64(loss)
    bi = C.c_int32(0); bl = C.c_double(0)
    check(lib().epi_grid_argmin(ptr(loss, C.c_double), C.c_int32(len(loss)), C.byref(bi), C.byref(bl)))
    return bi.value, bl.value


class Calibrate:
    @dataclass
    class Targets:  # Calibrate.scala:8-12
        t_arrive: Dict[str, float] = field(default_factory=dict)
        peak_i: Dict[str, float] = field(default_factory=dict)
        thresh_i: float = 10.0

    @staticmethod
    def grid_surface(hp: HyperParams, cfg: TrainConfig, base: State, seed_idx: int, targets: "Calibrate.Targets",
                     beta_grid: Sequence[float], travel_scales: Sequence[float], validate: bool = True,
                     heterogeneous: bool = True, countries: Optional[Sequence[Geo.Country]] = None, conn=None,
                     device: int = 0, devices: Optional[Sequence[int]] = None):
        """Calibrate.gridSurface (:84-116): (rows [(beta, ts, loss)], (bestLoss, (beta, ts)))."""
        countries = list(countries) if countries is not None else list(ToyConn.C)
        conn = ToyConn.W if conn is None else conn
        names = [c.name for c in countries]
        profiles = [Geo.CountryProfiles.get_profile(c.name) if c.name in Geo.CountryProfiles._TABLE else c
                    for c in countries]
        ta = {names.index(k): v for k, v in targets.t_arrive.items() if k in names}
        tp = {names.index(k): v for k, v in targets.peak_i.items() if k in names}
        if len(beta_grid) == 0 or len(travel_scales) == 0:
            best = (beta_grid[0] if len(beta_grid) else 0.25, travel_scales[0] if len(travel_scales) else 1.0)
            return [], (1.7976931348623157e308, best)
        if devices is not None:   # several GPUs of the node from this process, loss pieces all-gathered with RCCL
            eng = MultiGridEngine(hp, cfg.steps_per_epoch, base, seed_idx, [c.pop for c in countries], conn, profiles,
                                  beta_grid, travel_scales, ta, tp, heterogeneous, devices, names=names)
        else:
            eng = GridEngine(hp, cfg.steps_per_epoch, base, seed_idx, [c.pop for c in countries], conn, profiles,
                             beta_grid, travel_scales, ta, tp, heterogeneous, device, names=names)
        try:
            out = eng.run(want_metrics=False)
        finally:
            eng.close()
        loss = out["loss"]
        rows = [(float(b), float(ts), float(loss[i * len(travel_scales) + k]))
                for i, b in enumerate(beta_grid) for k, ts in enumerate(travel_scales)]
        bi, bl = grid_argmin(loss)
        return rows, (bl, (rows[bi][0], rows[bi][1]))

    # -- calibration-surface file (the CSV the reference's heatmap script reads: columns beta, travel, loss;
    #    src/main/scala/resources/plot_histogram.py:12-13).  The reference tree holds no writer for it, so the
    #    number format is this engine's choice: the shortest decimal that parses back to the same double.
    @staticmethod
    def write_surface_csv(path: str, rows: Sequence[Tuple[float, float, float]]) -> None:
        with open(path, "w", newline="\n") as f:
            f.write("beta,travel,loss\n")
            for b, ts, loss in rows:
                f.write("%s,%s,%s\n" % (repr(float(b)), repr(float(ts)), repr(float(loss))))

    @staticmethod
    def read_surface_csv(path: str) -> List[Tuple[float, float, float]]:
        with open(path) as f:
            head = f.readline().strip()
            if head != "beta,travel,loss":
                raise ValueError("%s: not a calibration surface (header %r)" % (path, head))
            return [tuple(float(x) for x in line.split(",")) for line in f if line.strip()]


class Snapshots:
    """Snapshot stream of the trainer (README.md:14: records (country, t, i, arrival, peakI, loss) under
    stream_in/snapshots; call sites MutliTrainer.scala:113,125).  The reference tree does not contain the
    writer itself, so the file layout is this engine's choice: one JSON-lines file per call, one record per
    country, which a Spark `readStream.schema(...).json(dir)` source picks up as it appears."""

    FIELDS = ("country", "t", "i", "arrival", "peakI", "loss")

    @staticmethod
    def records(world: "WorldToy", step: int, loss: float) -> List[dict]:
        return [{"country": nd.name, "t": int(step), "i": float(nd.s.i), "arrival": int(nd.arrival),
                 "peakI": float(nd.peak_i), "loss": float(loss)} for nd in world.nodes]

    @staticmethod
    def write_epoch_snapshot(world: "WorldToy", step: int, loss: float, out_dir: str = "stream_in/snapshots") -> str:
        import json
        import os
        os.makedirs(out_dir, exist_ok=True)
        final = os.path.join(out_dir, "snapshot-%010d.json" % int(step))
        tmp = final + ".tmp"   # written under another name and renamed: a streaming reader never sees half a file
        with open(tmp, "w", newline="\n") as f:
            for rec in Snapshots.records(world, step, loss):
                f.write(json.dumps(rec, allow_nan=False) + "\n")
        os.replace(tmp, final)
        return final
