"""pyref.py -- TEST INFRASTRUCTURE ONLY: second, independently written CPU
restatement of the reference path (pure Python floats = IEEE binary64, one
rounding per operation, no FMA).  It exists to cross-check oracle/epi_oracle.c
bit for bit on small cases and to generate the fixtures under tests/golden/.
Nothing in the shipped engine imports it.

Parity status: like the C oracle it is pinned by JDK known answers, the
reference's ValidationSpec answers and its conservation properties; numeric
trajectories are otherwise "parity unpinned" against a live Scala run (no JVM
is available to run the reference).

Citations are to the reference tree, src/main/scala/epidemic/.
"""
import math
import struct

MASK48 = (1 << 48) - 1
MULT = 0x5DEECE66D


def _i32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


class JavaRandom:
    """java.util.Random (behind scala.util.Random: env.scala:42, DDQNAgent.scala:14, MLP.scala:5)."""

    def __init__(self, seed):
        self.seed = (seed ^ MULT) & MASK48
        self.cached = None

    def next(self, bits):
        self.seed = (self.seed * MULT + 0xB) & MASK48
        return _i32(self.seed >> (48 - bits))

    def next_int(self, bound=None):
        if bound is None:
            return self.next(32)
        r = self.next(31)
        m = bound - 1
        if bound & m == 0:
            return _i32((bound * r) >> 31)
        u = r
        while True:
            r = u % bound
            if _i32(u - r + m) >= 0:
                return r
            u = self.next(31)

    def next_double(self):
        return ((self.next(26) << 27) + self.next(27)) * (1.0 / (1 << 53))

    def next_gaussian(self):
        if self.cached is not None:
            v, self.cached = self.cached, None
            return v
        while True:
            v1 = 2 * self.next_double() - 1
            v2 = 2 * self.next_double() - 1
            s = v1 * v1 + v2 * v2
            if s < 1 and s != 0:
                break
        mul = math.sqrt(-2 * strict_log(s) / s)
        self.cached = v2 * mul
        return v1 * mul


def _hi_lo(x):
    b = struct.unpack("<Q", struct.pack("<d", x))[0]
    return _i32(b >> 32), b & 0xFFFFFFFF


def _with_hi(x, hi):
    b = struct.unpack("<Q", struct.pack("<d", x))[0]
    return struct.unpack("<d", struct.pack("<Q", ((hi & 0xFFFFFFFF) << 32) | (b & 0xFFFFFFFF)))[0]


def strict_log(x):
    """StrictMath.log: fdlibm e_log.c algorithm (positive finite normal arguments and the
    documented special cases)."""
    ln2_hi, ln2_lo = 6.93147180369123816490e-01, 1.90821492927058770002e-10
    L = (6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01,
         2.222219843214978396e-01, 1.818357216161805012e-01, 1.531383769920937332e-01,
         1.479819860511658591e-01)
    hx, lx = _hi_lo(x)
    k = 0
    if hx < 0x00100000:
        if ((hx & 0x7FFFFFFF) | lx) == 0:
            return -math.inf
        if hx < 0:
            return math.nan
        k -= 54
        x *= 1.80143985094819840000e16
        hx, lx = _hi_lo(x)
    if hx >= 0x7FF00000:
        return x + x
    k += (hx >> 20) - 1023
    hx &= 0x000FFFFF
    i = (hx + 0x95F64) & 0x100000
    x = _with_hi(x, hx | (i ^ 0x3FF00000))
    k += i >> 20
    f = x - 1.0
    if (0x000FFFFF & (2 + hx)) < 3:
        if f == 0.0:
            if k == 0:
                return 0.0
            dk = float(k)
            return dk * ln2_hi + dk * ln2_lo
        R = f * f * (0.5 - 0.33333333333333333 * f)
        if k == 0:
            return f - R
        dk = float(k)
        return dk * ln2_hi - ((R - dk * ln2_lo) - f)
    s = f / (2.0 + f)
    dk = float(k)
    z = s * s
    i = hx - 0x6147A
    w = z * z
    j = 0x6B851 - hx
    t1 = w * (L[1] + w * (L[3] + w * L[5]))
    t2 = z * (L[0] + w * (L[2] + w * (L[4] + w * L[6])))
    i |= j
    R = t2 + t1
    if i > 0:
        hfsq = 0.5 * f * f
        if k == 0:
            return f - (hfsq - s * (hfsq + R))
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f)
    if k == 0:
        return f - s * (f - R)
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f)


# ---------------------------------------------------------------- JVM conversions
def d2i(x):
    if x != x:
        return 0
    if x >= 2147483647.0:
        return 2147483647
    if x <= -2147483648.0:
        return -2147483648
    return int(x)


def jround(x):  # Math.round(double) -> long
    if x != x:
        return 0
    fl = math.floor(x)
    if x - fl >= 0.5:
        fl += 1
    return max(-(1 << 63), min((1 << 63) - 1, fl))


def jround_i(x):  # math.round(x).toInt
    return _i32(jround(x))


def clamp01(x):
    return max(0.0, min(1.0, x))


def seq_sum(xs):
    it = iter(xs)
    try:
        acc = next(it)
    except StopIteration:
        return 0.0
    for v in it:
        acc = acc + v
    return acc


# ---------------------------------------------------------------- env.scala:241-254
def allocate_int(total, weights):
    n = len(weights)
    sum_w = seq_sum(weights)
    if total <= 0 or sum_w <= 0.0:
        return [0] * n
    raw = [(w / sum_w) * total for w in weights]
    base = [d2i(math.floor(r)) for r in raw]
    rem = _i32(total - _i32(sum(base)))
    order = sorted(range(n), key=lambda j: raw[j] - math.floor(raw[j]), reverse=True)  # stable
    k = 0
    while rem > 0 and k < n:
        base[order[k]] = _i32(base[order[k]] + 1)
        rem -= 1
        k += 1
    return base


# ---------------------------------------------------------------- Geo.scala:4-21 defaults
DEFAULT_PROFILE = dict(beta_mul=1.0, ifr_mul=1.0, noise=0.05, population_density=100.0,
                       gdp_per_capita=30000.0, healthcare_capacity=0.7, age_elderly=0.15,
                       connectivity=0.8, econ_resilience=0.7)


class State:
    __slots__ = ("s", "i", "r", "d", "v", "hosp_cap", "t", "t_max", "seed")

    def __init__(self, s, i, r, d, v, hosp_cap, t, t_max, seed=0):
        self.s, self.i, self.r, self.d, self.v = float(s), float(i), float(r), float(d), float(v)
        self.hosp_cap, self.t, self.t_max, self.seed = float(hosp_cap), t, t_max, seed

    def copy(self, **kw):
        st = State(self.s, self.i, self.r, self.d, self.v, self.hosp_cap, self.t, self.t_max, self.seed)
        for k, v in kw.items():
            setattr(st, k, v)
        return st

    @property
    def n(self):
        return self.s + self.i + self.r + self.d

    @property
    def util(self):
        return self.i / max(1.0, self.hosp_cap)

    def features(self):  # env.scala:28-30
        N = max(1.0, self.n)
        return [self.s / N, self.i / N, self.r / N, self.d / N, self.v / N, self.util,
                float(self.t) / max(1, self.t_max)]

    def tup(self):
        return (self.s, self.i, self.r, self.d, self.v, self.hosp_cap, self.t)


class EpidemicEnv:
    """env.scala:32-238."""

    def __init__(self, base_beta=0.35, base_gamma=0.12, base_ifr=0.008, noise=0.05, reward_clip=5.0,
                 risk_factor=0.0, profile=None, seed=42):
        self.base_beta, self.base_gamma, self.base_ifr = base_beta, base_gamma, base_ifr
        self.noise, self.reward_clip, self.risk_factor = noise, reward_clip, risk_factor
        self.p = dict(DEFAULT_PROFILE if profile is None else profile)
        self.rnd = JavaRandom(seed)

    def step(self, st, act, gauss=None):
        act = max(0, min(7, act))
        p = self.p
        beta = self.base_beta * p["beta_mul"]
        gamma = self.base_gamma
        ifr = self.base_ifr * p["ifr_mul"]
        cap = st.hosp_cap
        if act == 1:
            beta *= p["econ_resilience"] * 0.75 + 0.5
        elif act == 2:
            beta *= (1.0 - p["connectivity"]) * 0.4 + 0.4
        elif act == 3:
            beta *= p["econ_resilience"] * 0.5 + 0.2
        elif act == 5:
            cap *= 1.0 + p["healthcare_capacity"]
            gamma *= 1.3
        s0 = max(0, jround_i(st.s)); i0 = max(0, jround_i(st.i)); r0 = max(0, jround_i(st.r))
        d0 = max(0, jround_i(st.d)); v0 = max(0, jround_i(st.v))
        pop0 = _i32(s0 + i0 + r0 + d0 + v0)
        mix = max(1.0, float(_i32(s0 + i0 + r0)))
        country_noise = self.noise * (0.2 + p["noise"] * 1.5)
        z = self.rnd.next_gaussian() if gauss is None else gauss
        stoch = 1.0 + country_noise * z
        eff_beta = clamp01(beta * stoch)
        eff_gamma = clamp01(gamma * stoch)
        eff_ifr = clamp01((ifr * 6.0 if i0 > cap else ifr) * stoch)
        vax = 0.02 * p["healthcare_capacity"] if act == 6 else (0.05 * p["healthcare_capacity"] if act == 7 else 0.0)
        p_inf = clamp01(eff_beta * (float(i0) / mix if mix > 0 else 0.0))
        p_vax = clamp01(vax)
        s_al = allocate_int(s0, [max(0.0, 1.0 - p_inf - p_vax), p_inf, p_vax])
        p_rec = clamp01(eff_gamma); p_die = clamp01(eff_ifr)
        i_al = allocate_int(i0, [max(0.0, 1.0 - p_rec - p_die), p_rec, p_die])
        sN = _i32(s0 - s_al[1] - s_al[2]); iN = _i32(i_al[0] + s_al[1]); rN = _i32(r0 + i_al[1])
        dN = _i32(d0 + i_al[2]); vN = _i32(v0 + s_al[2])
        resid = _i32(pop0 - _i32(sN + iN + rN + dN + vN))
        if resid != 0:
            sN = max(0, _i32(sN + resid))
        nxt = st.copy(s=float(sN), i=float(iN), r=float(rN), d=float(dN), v=float(vN), hosp_cap=cap, t=_i32(st.t + 1))
        reward = self._reward(st, nxt, act)
        nn = max(1.0, nxt.n)
        done = (nxt.t >= nxt.t_max) or (nxt.i / nn < 1e-5) or (nxt.d / nn > 0.02)
        return nxt, reward, done

    def _reward(self, st, nx, act):  # env.scala:142-234
        p = self.p
        Nn = max(1.0, nx.n)
        di = (nx.i - st.i) / Nn
        dd = (nx.d - st.d) / Nn
        overflow = max(0.0, nx.util - 1.0)
        hc = p["healthcare_capacity"]
        hw = 0.2 if hc > 0.8 else (0.6 if hc > 0.5 else 0.9)
        ew = 1.0 - hw
        g = p["gdp_per_capita"]
        gf = 0.3 if g > 50000 else (1.0 if g > 20000 else (2.5 if g > 5000 else 5.0))
        hf = 0.2 if hc > 0.8 else (1.0 if hc > 0.5 else 8.0)
        dn = p["population_density"]
        df = 3.0 if dn > 400 else (2.0 if dn > 200 else (1.0 if dn > 100 else 0.5))
        af = 1.0 + p["age_elderly"] * 8.0
        inf_pen = di * 200.0 * df * hw * gf
        mort_pen = dd * 2000.0 * af * hf * hw
        hosp_pen = overflow * 1000.0 * hw * hf
        base_cost = (0.0, 0.05, 0.08, 0.15, 0.01, 0.06, 0.03, 0.05)[act]
        em = 0.3 if g > 50000 else (1.0 if g > 20000 else (3.0 if g > 5000 else 8.0))
        mm = 1.0 + (1.0 - p["econ_resilience"]) * 2.0 if act in (2, 3) else 1.0
        cost = base_cost * em * mm * 30.0 * ew * gf
        base_reward = -(inf_pen + mort_pen + hosp_pen + cost)
        bonus = {3: 0.05 * p["econ_resilience"], 2: 0.03 * p["connectivity"], 1: 0.02,
                 6: 0.02 * hc, 7: 0.04 * hc}.get(act, 0.0)
        shaped = base_reward * (1.0 + 0.75 * self.risk_factor) + (self.risk_factor * bonus)
        return max(-self.reward_clip * 3.0, min(self.reward_clip, shaped))


class MLP:  # MLP.scala:4-22
    def __init__(self, n_in=7, h=128, n_out=8):
        rnd = JavaRandom(7)
        mk = lambda r, c: [[rnd.next_gaussian() * 0.05 for _ in range(c)] for _ in range(r)]
        self.W1, self.W2, self.W3 = mk(h, n_in), mk(h, h), mk(n_out, h)
        self.b1, self.b2, self.b3 = [0.0] * h, [0.0] * h, [0.0] * n_out

    @staticmethod
    def _mv(W, x):
        return [seq_sum(w * xi for w, xi in zip(row, x)) for row in W]

    def predict(self, x, clamp=math.inf):
        relu = lambda v: v if v > 0 else 0.0
        h1 = [relu(s + b) for s, b in zip(self._mv(self.W1, x), self.b1)]
        h2 = [relu(s + b) for s, b in zip(self._mv(self.W2, h1), self.b2)]
        y = [s + b for s, b in zip(self._mv(self.W3, h2), self.b3)]
        if math.isfinite(clamp):
            y = [max(-clamp, min(clamp, v)) for v in y]
        return y


    def train_batch(self, xs, tgts, grad_clip, clamp):  # MLP.scala:29-99
        relu = lambda v: v if v > 0 else 0.0
        clip = lambda v, c: max(-c, min(c, v)) if math.isfinite(c) else v
        h, n_in, n_out = len(self.W2), len(self.W1[0]), len(self.W3)
        gW1 = [[0.0] * n_in for _ in range(h)]; gb1 = [0.0] * h
        gW2 = [[0.0] * h for _ in range(h)]; gb2 = [0.0] * h
        gW3 = [[0.0] * h for _ in range(n_out)]; gb3 = [0.0] * n_out
        loss_sum = 0.0
        for x0, tgt in zip(xs, tgts):
            z1 = [s + b for s, b in zip(self._mv(self.W1, x0), self.b1)]; h1 = [relu(z) for z in z1]
            z2 = [s + b for s, b in zip(self._mv(self.W2, h1), self.b2)]; h2 = [relu(z) for z in z2]
            y = [s + b for s, b in zip(self._mv(self.W3, h2), self.b3)]
            yc = [clip(v, clamp) for v in y]
            err = [p - t for p, t in zip(yc, tgt)]
            loss_sum += seq_sum(0.5 * e * e for e in err)
            dh2 = [seq_sum(self.W3[i][j] * err[i] for i in range(n_out)) for j in range(h)]
            dz2 = [g * relu(z) for g, z in zip(dh2, z2)]
            dh1 = [seq_sum(self.W2[i][j] * dz2[i] for i in range(h)) for j in range(h)]
            dz1 = [g * relu(z) for g, z in zip(dh1, z1)]
            for i in range(n_out):
                for j in range(h):
                    gW3[i][j] += err[i] * h2[j]
                gb3[i] += err[i]
            for i in range(h):
                for j in range(h):
                    gW2[i][j] += dz2[i] * h1[j]
                gb2[i] += dz2[i]
                for j in range(n_in):
                    gW1[i][j] += dz1[i] * x0[j]
                gb1[i] += dz1[i]
        scale = 1.0 / max(1, len(xs))
        lr = self.lr
        for W, g in ((self.W3, gW3), (self.W2, gW2), (self.W1, gW1)):
            for i in range(len(W)):
                for j in range(len(W[i])):
                    W[i][j] -= lr * clip(g[i][j] * scale, grad_clip)
        for b, g in ((self.b3, gb3), (self.b2, gb2), (self.b1, gb1)):
            for i in range(len(b)):
                b[i] -= lr * clip(g[i] * scale, grad_clip)
        return loss_sum / max(1, len(xs))

    def copy_from(self, src):  # MLP.scala:101-109
        for name in ("W1", "W2", "W3"):
            setattr(self, name, [list(r) for r in getattr(src, name)])
        for name in ("b1", "b2", "b3"):
            setattr(self, name, list(getattr(src, name)))


class Learner:
    """DDQNAgent.observe / ddqnTarget / learn with its ReplayBuffer (DDQNAgent.scala:34-65, ReplayBuffer.scala:10-22).
    hp: gamma, lr, clip_q, grad_clip, ema_alpha, hidden, replay_capacity, target_update_every."""

    def __init__(self, hp):
        self.hp = hp
        self.online = MLP(7, hp["hidden"], 8); self.target = MLP(7, hp["hidden"], 8)
        self.online.lr = self.target.lr = hp["lr"]
        self.buf = []; self.idx = 0
        self.rnd = JavaRandom(123)
        self.steps = 0; self.ema = 0.0

    def observe(self, s, a, r, s2, done):
        tr = (s.features(), a, r, s2.features(), done)
        if len(self.buf) < self.hp["replay_capacity"]:
            self.buf.append(tr)
        else:
            self.buf[self.idx] = tr
            self.idx = (self.idx + 1) % self.hp["replay_capacity"]

    def sample(self, n):  # rnd.shuffle(buf).take(n): scala.util.Random.shuffle, Scala 2.13 library
        b = list(range(len(self.buf)))
        for m in range(len(b), 1, -1):
            k = self.rnd.next_int(m)
            b[m - 1], b[k] = b[k], b[m - 1]
        return b[:n]

    def _clip_q(self, x):
        return max(-self.hp["clip_q"], min(self.hp["clip_q"], x))

    def _target(self, tr):
        fs, a, r, fs2, done = tr
        q = list(self.online.predict(fs, self.hp["clip_q"]))
        if done:
            q[a] = self._clip_q(r)
        else:
            qn = self.online.predict(fs2, self.hp["clip_q"])
            best = 0
            for k in range(1, len(qn)):
                if qn[k] > qn[best]:
                    best = k
            qt = self.target.predict(fs2, self.hp["clip_q"])
            q[a] = self._clip_q(r + self.hp["gamma"] * qt[best])
        return q

    def learn(self, batch, grad_accum=4):
        need = max(batch * grad_accum, batch)
        if len(self.buf) < need:
            return 0.0
        micros = [self.buf[k] for k in self.sample(need)]
        xs = [m[0] for m in micros]
        tgts = [self._target(m) for m in micros]
        loss = self.online.train_batch(xs, tgts, self.hp["grad_clip"], self.hp["clip_q"])
        if self.steps % self.hp["target_update_every"] == 0:
            self.target.copy_from(self.online)
        a = self.hp["ema_alpha"]
        self.ema = loss if self.ema == 0.0 else a * loss + (1 - a) * self.ema
        return loss


_SHARED_MLP = {}


class Agent:  # DDQNAgent.scala:6-32 (action choice only)
    def __init__(self, hp):
        self.hp = hp
        key = hp["hidden"]
        if key not in _SHARED_MLP:  # identical for every agent: fixed init seed
            _SHARED_MLP[key] = MLP(7, key, 8)
        self.net = _SHARED_MLP[key]
        self.steps = 0
        self.rnd = JavaRandom(9)
        self.replay = 0

    def epsilon(self):
        frac = min(1.0, float(self.steps) / max(1, self.hp["epsilon_decay_steps"]))
        return self.hp["epsilon_start"] + (self.hp["epsilon_end"] - self.hp["epsilon_start"]) * frac

    def act_greedy(self, s):
        q = self.net.predict(s.features(), self.hp["clip_q"])
        best = 0
        for k in range(1, len(q)):
            if q[k] > q[best]:
                best = k
        return best

    def act(self, s):
        eps = self.epsilon()  # argument of actWithEpsilon: evaluated before countStep raises the counter
        self.steps += 1
        if self.rnd.next_double() < eps:
            return self.rnd.next_int(8)
        return self.act_greedy(s)


DEFAULT_HP = dict(epsilon_start=1.0, epsilon_end=0.1, epsilon_decay_steps=100000, hidden=128, clip_q=10.0,
                  reward_clip=5.0, batch_size=128, learn_every=8)  # config.scala:3-21


def scale_row(W, row, factor):  # ToyConn.scala:58-70
    A = [list(r) for r in W]
    n = len(A)
    s = 0.0
    for j in range(n):
        if j != row:
            A[row][j] *= factor
            s += A[row][j]
    cap = min(0.2, s)
    if s > cap and s > 0:
        sc = cap / s
        for j in range(n):
            if j != row:
                A[row][j] *= sc
    A[row][row] = 0.0
    return A


def arrival_time(series, thresh=10.0):  # Validation.scala:15-19
    for k, v in enumerate(series):
        if v >= thresh:
            return k
    return len(series)


def peak_i(series):  # Validation.scala:21-22
    return 0.0 if len(series) == 0 else max(series)


def sum_state(s):  # WorldToy.scala:79-80
    return jround(s.s + s.i + s.r + s.d + s.v)


class MassError(Exception):
    pass


class World:
    """WorldToy.scala:11-235 over N input countries."""

    def __init__(self, hp, base, pops, conn, envs, seed_idx=0, validate=False):
        self.hp, self.pops, self.conn, self.envs, self.validate = hp, pops, conn, envs, validate
        self.n = len(pops)
        self.init = []
        for c, pop in enumerate(pops):  # :28-32
            hc = max(base.hosp_cap * (float(pop) / 1_000_000.0), 1.0)
            if c == seed_idx:
                i0 = d2i(0.001 * pop)
                st = base.copy(i=float(i0), s=float(pop - i0), hosp_cap=hc)
            else:
                st = base.copy(i=0.0, s=float(pop), hosp_cap=hc)
            self.init.append(st)
        self.cur = [s.copy() for s in self.init]
        self.agents = [Agent(hp) for _ in pops]
        self.g_step = 0
        self.last_a = [0] * self.n
        self.flows = None

    def _move(self, local, t_max):  # :103-145
        n = self.n
        total_local = sum(sum_state(s) for s in local) if self.validate else 0
        mS, mI = [0] * n, [0] * n
        for i in range(n):
            Si = max(0, jround_i(local[i].s)); Ii = max(0, jround_i(local[i].i))
            w_row = [0.0 if j == i else self.conn[i][j] for j in range(n)]
            sum_w = max(0.0, seq_sum(w_row))
            eS = _i32(min(Si, int(math.floor(Si * sum_w + 1e-9))))
            eI = _i32(min(Ii, int(math.floor(Ii * sum_w + 1e-9))))
            aS = allocate_int(eS, w_row); aI = allocate_int(eI, w_row)
            for j in range(n):
                if j != i:
                    mS[j] = _i32(mS[j] + aS[j]); mI[j] = _i32(mI[j] + aI[j])
            mS[i] = _i32(mS[i] - eS); mI[i] = _i32(mI[i] - eI)
        self.flows = (mS, mI)
        for c in range(n):
            s2 = local[c]
            self.cur[c] = s2.copy(s=max(0.0, s2.s + mS[c]), i=max(0.0, s2.i + mI[c]), t=min(_i32(s2.t + 1), t_max))
        if self.validate:
            if sum(sum_state(s) for s in self.cur) != total_local:
                raise MassError("global mass changed during mobility")

    def step_given(self, actions, gauss, t_max):
        local, rewards = [], []
        for c in range(self.n):
            tb = sum_state(self.cur[c])
            s2, r, _ = self.envs[c].step(self.cur[c], actions[c], gauss[c])
            self.last_a[c] = actions[c]
            if self.validate and sum_state(s2) != tb:
                raise MassError("mass not conserved (local)")
            local.append(s2); rewards.append(r)
        self._move(local, t_max)
        return rewards

    def step_one(self, t_max, greedy=False):  # :83-170 / :172-235
        local, rewards = [], []
        for c in range(self.n):
            ag = self.agents[c]
            a = ag.act_greedy(self.cur[c]) if greedy else ag.act(self.cur[c])
            tb = sum_state(self.cur[c])
            s2, r, _ = self.envs[c].step(self.cur[c], a)
            self.last_a[c] = a
            if self.validate and sum_state(s2) != tb:
                raise MassError("mass not conserved (local)")
            if not greedy:
                ag.replay += 1
            local.append(s2); rewards.append(r)
        self._move(local, t_max)
        if not greedy:
            self.g_step += 1
            for ag in self.agents:
                if self.g_step >= self.hp["learn_every"] and ag.replay >= self.hp["batch_size"] * 4:
                    raise NotImplementedError("DDQN learning would start here (out of scope)")
        return rewards


def make_calibration_envs(beta, profiles, hp, heterogeneous):  # Calibrate.scala:31-51
    envs = []
    for c, p in enumerate(profiles):
        if heterogeneous:
            fb = beta * (1.0 + (p["population_density"] / 1000.0) * 0.5) * (3.0 - p["healthcare_capacity"] * 2.0)
            fi = 0.008 * (1.0 + p["age_elderly"] * 2.0)
            fn = 0.05 * (1.0 + (1.0 - p["econ_resilience"]))
            envs.append(EpidemicEnv(fb, 0.12, fi, fn, hp["reward_clip"], 0.0, p, c + 42))
        else:
            envs.append(EpidemicEnv(beta, 0.12, 0.008, 0.05, hp["reward_clip"], 0.0, None, 7))
    return envs


def rollout(hp, steps, base, pops, W, profiles, seed_idx, beta, ts, heterogeneous, keep_series=False):  # :19-82
    Ws = scale_row(W, seed_idx, ts)
    world = World(hp, base, pops, Ws, make_calibration_envs(beta, profiles, hp, heterogeneous), seed_idx, validate=True)
    series = [[] for _ in pops]
    for _ in range(steps):
        world.step_one(steps)
        for c in range(len(pops)):
            series[c].append(world.cur[c].i)
    arr = [arrival_time(s, 10.0) for s in series]
    pk = [peak_i(s) for s in series]
    fin = [s[-1] if s else 0.0 for s in series]
    return (arr, pk, fin, series) if keep_series else (arr, pk, fin)


def rmse(xs):  # :14-17
    return math.sqrt(seq_sum(v * v for v in xs) / max(1.0, float(len(xs))))


def grid_surface(hp, steps, base, pops, W, profiles, seed_idx, t_arrive, t_peak, betas, tss, heterogeneous=True):
    """Calibrate.gridSurface :84-116.  t_arrive/t_peak: dict country index -> target."""
    best_loss, best = 1.7976931348623157e308, (betas[0] if betas else 0.25, tss[0] if tss else 1.0)
    rows = []
    for beta in betas:
        for ts in tss:
            arr, pk, _ = rollout(hp, steps, base, pops, W, profiles, seed_idx, beta, ts, heterogeneous)
            n = len(pops)
            a_loss = rmse([float(arr[c]) - t_arrive.get(c, float(arr[c])) for c in range(n)])
            p_loss = rmse([pk[c] - t_peak.get(c, pk[c]) for c in range(n)])
            loss = 0.5 * a_loss + 0.5 * p_loss
            rows.append((beta, ts, loss))
            if loss < best_loss:
                best_loss, best = loss, (beta, ts)
    return rows, (best_loss, best)
