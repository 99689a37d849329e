"""Seeded synthetic country graphs for tests and benchmarks (inputs only; no engine arithmetic).

All draws come from SplitMix64 so that every process, rank and machine builds identical inputs.
The workloads follow the configurations the calibration metric is quoted on:
  c1  toy: ToyConn.C / ToyConn.W, beta 0.24..0.27, travel 0.5..0.9, T=300 (App.scala:97-107)
  c2  200 countries dense, 32x32 grid, T=200
  c3  1,000 countries dense, 64x64 grid, T=200
  c4  20,000 patches CSR, 16x16 grid, T=200
"""
import numpy as np

from . import Geo, State, ToyConn

MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
DEFAULT_SEED = 20251226


def splitmix64(seed: int, count: int) -> np.ndarray:
    """`count` 64-bit outputs of SplitMix64 started at `seed`."""
    with np.errstate(over="ignore"):
        k = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(seed: int, count: int) -> np.ndarray:
    """Doubles in [0, 1) with 53 random bits."""
    return (splitmix64(seed, count) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def make_countries(n: int, seed: int = DEFAULT_SEED, pop_lo: float = 1e5, pop_hi: float = 5e8):
    """n synthetic countries: log-uniform populations, positions on a 20,000 km square, and
    profile fields spread over the ranges of Geo.CountryProfiles.  The population cap keeps every
    country inside the 32-bit Int range the reference computes in under 12 %/day mobility."""
    u = uniform(seed, 12 * n).reshape(12, n)
    pops = np.floor(np.exp(np.log(pop_lo) + u[0] * (np.log(pop_hi) - np.log(pop_lo)))).astype(np.int64)
    xs, ys = u[1] * 20000.0, u[2] * 20000.0
    countries = []
    for c in range(n):
        countries.append(Geo.Country(
            name="S%05d" % c, pop=int(pops[c]), lat=float(ys[c]), lon=float(xs[c]),
            beta_mul=0.8 + 0.5 * u[3][c], ifr_mul=0.8 + 0.6 * u[4][c], noise=0.03 + 0.05 * u[5][c],
            population_density=30.0 + 450.0 * u[6][c], gdp_per_capita=2000.0 + 63000.0 * u[7][c],
            healthcare_capacity=0.3 + 0.65 * u[8][c], age_distribution_elderly=0.07 + 0.23 * u[9][c],
            connectivity_strength=0.4 + 0.55 * u[10][c], economic_resilience=0.3 + 0.6 * u[11][c]))
    return countries, np.stack([xs, ys], axis=1)


def gravity_dense(countries, xy, cap_row: float = 0.12, a: float = 0.2, b: float = 1.0, g: float = 1.2) -> np.ndarray:
    """Dense gravity-like matrix in the shape of ToyConn.gravityMatrix: positive off-diagonal
    weights pop_i^a pop_j^b / max(300, d_ij)^g, zero diagonal, each row scaled to sum to cap_row."""
    pop = np.array([c.pop for c in countries], dtype=np.float64)
    d = np.sqrt(((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1))
    d = np.maximum(300.0, d)
    W = (pop[:, None] ** a) * (pop[None, :] ** b) / d ** g
    np.fill_diagonal(W, 0.0)
    W *= (cap_row / W.sum(axis=1))[:, None]
    # keep every row sum at or below the cap after rounding
    W *= np.minimum(1.0, cap_row / W.sum(axis=1))[:, None]
    return np.ascontiguousarray(W)


def gravity_csr(countries, xy, seed: int = DEFAULT_SEED, mean_degree: float = 16.0, cap_row: float = 0.12):
    """Sparse mobility network: each patch links to deg ~ Poisson(mean_degree) (min 1) distinct other
    patches, weights gravity-like, rows scaled to sum to cap_row.  Returns (rowptr, col, val)."""
    n = len(countries)
    pop = np.array([c.pop for c in countries], dtype=np.float64)
    rng = np.random.Generator(np.random.PCG64(int(splitmix64(seed ^ 0xC5A, 1)[0] >> np.uint64(1))))
    deg = np.maximum(1, rng.poisson(mean_degree, n))
    deg = np.minimum(deg, n - 1)
    rowptr = np.zeros(n + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(deg)
    col = np.zeros(rowptr[-1], dtype=np.int32)
    val = np.zeros(rowptr[-1], dtype=np.float64)
    for i in range(n):
        nb = rng.choice(n - 1, size=deg[i], replace=False)
        nb = np.sort(np.where(nb >= i, nb + 1, nb))
        dist = np.maximum(300.0, np.sqrt(((xy[nb] - xy[i]) ** 2).sum(-1)))
        w = pop[nb] / dist ** 1.2
        w *= cap_row / w.sum()
        w *= min(1.0, cap_row / w.sum())
        col[rowptr[i]:rowptr[i + 1]] = nb
        val[rowptr[i]:rowptr[i + 1]] = w
    return rowptr, col, val


def targets(n: int, steps: int, seed: int = DEFAULT_SEED):
    """Synthetic Targets: arrival uniform in {0..steps}, peak log-uniform in [1e3, 1e7], for every country."""
    u = uniform(seed ^ 0x7A26, 2 * n).reshape(2, n)
    arrive = np.floor(u[0] * (steps + 1))
    peak = np.floor(np.exp(np.log(1e3) + u[1] * (np.log(1e7) - np.log(1e3))))
    return {c: float(arrive[c]) for c in range(n)}, {c: float(peak[c]) for c in range(n)}


def base_state(steps: int) -> State:
    """App.scala:25-28 with tMax = steps."""
    return State(s=1_000_000, i=0, r=0, d=0, v=0, hosp_cap=10_000, t=0, t_max=steps, seed=1)


def workload(name: str, seed: int = DEFAULT_SEED):
    """Returns a dict describing one of the named configurations."""
    name = name.lower()
    if name == "c1":
        names = [c.name for c in ToyConn.C]
        ta = {"Japan": 40.0, "India": 55.0, "Italy": 60.0, "Germany": 58.0, "USA": 52.0, "China": 0.0}
        tp = {"Japan": 10000.0, "India": 18000.0, "Italy": 8000.0, "Germany": 9000.0, "USA": 12000.0, "China": 35000.0}
        return dict(name="c1", countries=list(ToyConn.C), profiles=[Geo.CountryProfiles.get_profile(n) for n in names],
                    conn=ToyConn.W, steps=300, betas=[k / 100.0 for k in range(24, 28)],
                    travel=[k / 10.0 for k in range(5, 10)], t_arrive={names.index(k): v for k, v in ta.items()},
                    t_peak={names.index(k): v for k, v in tp.items()}, seed_idx=0)
    sizes = {"c2": (200, 32, 200), "c3": (1000, 64, 200), "c4": (20000, 16, 200)}
    if name not in sizes:
        raise ValueError("unknown workload %r" % name)
    n, gdim, steps = sizes[name]
    countries, xy = make_countries(n, seed)
    conn = gravity_csr(countries, xy, seed) if name == "c4" else gravity_dense(countries, xy)
    ta, tp = targets(n, steps, seed)
    return dict(name=name, countries=countries, profiles=countries, conn=conn, steps=steps,
                betas=np.linspace(0.20, 0.40, gdim).tolist(), travel=np.linspace(0.25, 2.0, gdim).tolist(),
                t_arrive=ta, t_peak=tp, seed_idx=0)


def small_workload(n: int, n_beta: int, n_travel: int, steps: int, seed: int = DEFAULT_SEED):
    """A reduced dense configuration of the same family (parity tests)."""
    countries, xy = make_countries(n, seed)
    ta, tp = targets(n, steps, seed)
    return dict(name="small", countries=countries, profiles=countries, conn=gravity_dense(countries, xy), steps=steps,
                betas=np.linspace(0.20, 0.40, n_beta).tolist(), travel=np.linspace(0.25, 2.0, n_travel).tolist(),
                t_arrive=ta, t_peak=tp, seed_idx=0)


def calibrated_toy_point(device: int = 0):
    """(beta, travel scale) the default calibration picks for the toy world: the first minimum of configuration 1's
    4 x 5 surface (App.scala:97-109), computed by the engine's own sweep.  Configuration 5 -- the batched DDQN
    rollout "of the calibrated config" -- takes its beta from here instead of a constant."""
    import epidemicsimulator_amd as E
    wl = workload("c1")
    eng = E.GridEngine(E.HyperParams(), wl["steps"], base_state(wl["steps"]), wl["seed_idx"], [c.pop for c in wl["countries"]],
                       wl["conn"], wl["profiles"], wl["betas"], wl["travel"], wl["t_arrive"], wl["t_peak"], True, device=device)
    try:
        loss = eng.run(want_metrics=False)["loss"]
    finally:
        eng.close()
    best, _ = E.grid_argmin(loss)
    nt = len(wl["travel"])
    return float(wl["betas"][best // nt]), float(wl["travel"][best % nt])
