#!/usr/bin/env python3
"""Randomised differential soak: random world sizes, graphs, seeds and policies, engine against the oracle.
   python tools/soak.py [seconds]    (needs a GPU; prints one line per case and a summary)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import epidemicsimulator_amd as E
from epidemicsimulator_amd import synth
from oracle import oracle_lib as O
from tests import util


def run(budget=120.0, seed=2024, verbose=True):
    """Runs random cases for `budget` seconds; returns the number of cases (every one compared with ==)."""
    rng = np.random.default_rng(seed)
    O.build()
    t0 = time.time()
    cases = 0
    while time.time() - t0 < budget:
        sparse = rng.random() < 0.4
        n = int(rng.integers(1025, 2600)) if sparse else int(rng.choice([1, 2, 5, 6, 9, 16, 17, 33, 64, 65, 130, 200, 257, 600, 1000, 1024]))
        steps = int(rng.integers(3, 12)) if sparse or n > 300 else int(rng.integers(5, 60))
        seed = int(rng.integers(1, 1 << 30))
        countries, xy = synth.make_countries(n, seed)
        wl = synth.small_workload(8, 1, 2, steps, seed=seed)
        wl["countries"], wl["profiles"] = countries, countries
        if sparse:
            rp, col, val = synth.gravity_csr(countries, xy, seed, mean_degree=float(rng.choice([2.0, 5.0, 9.0, 14.0, 20.0, 31.0])))
            if rng.random() < 0.4:   # a few hub rows: mostly up to 256 stored weights, some streamed ones above that
                rows = [list(zip(col[rp[i]:rp[i + 1]], val[rp[i]:rp[i + 1]])) for i in range(n)]
                for i in rng.choice(n, size=int(rng.integers(1, 6)), replace=False):
                    deg = int(rng.integers(33, 257)) if rng.random() < 0.8 else int(rng.integers(257, n))
                    nb = np.sort(rng.choice(np.delete(np.arange(n), i), size=deg, replace=False))
                    w = rng.random(deg) + 0.01
                    w *= 0.12 / w.sum()
                    rows[i] = list(zip(nb.tolist(), w.tolist()))
                rp = np.zeros(n + 1, dtype=np.int32)
                rp[1:] = np.cumsum([len(r) for r in rows])
                col = np.array([c for r in rows for c, _ in r], dtype=np.int32)
                val = np.array([v for r in rows for _, v in r], dtype=np.float64)
            wl["conn"] = (rp, col, val)
        elif n > 1:
            wl["conn"] = synth.gravity_dense(countries, xy)
        else:
            wl["conn"] = np.zeros((1, 1))
        wl["t_arrive"], wl["t_peak"] = synth.targets(n, steps, seed)
        wl["seed_idx"] = int(rng.integers(0, n))
        eps = float(rng.choice([1.0, 1.0, 0.5, 0.0]))
        het = bool(rng.random() < 0.7)
        base = synth.base_state(steps)
        hp = E.HyperParams(epsilon_start=eps, epsilon_end=eps)
        ohp = O.default_hyper(); ohp.epsilon_start = eps; ohp.epsilon_end = eps
        got = util.engine_grid(wl, base, het, hp=hp)
        ta = np.zeros(n); tp = np.zeros(n); ha = np.ones(n, dtype=np.uint8)
        for c, v in wl["t_arrive"].items():
            ta[c] = v
        for c, v in wl["t_peak"].items():
            tp[c] = v
        want = O.grid_surface([c.pop for c in countries], util.dense_of(wl["conn"], n), wl["seed_idx"], util.oracle_state(base),
                              [util.prof_dict(p) for p in wl["profiles"]], steps, wl["betas"], wl["travel"], ta, ha, tp, ha, het, hp=ohp)
        util.assert_grid_equal(got, want)
        cases += 1
        if verbose:
            print("ok n=%d %s steps=%d seed_idx=%d eps=%.1f het=%d" % (n, "csr" if sparse else "dense", steps, wl["seed_idx"], eps, het), flush=True)
    if verbose:
        print("soak: %d cases, all equal" % cases)
    return cases


def main():
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 2024)


if __name__ == "__main__":
    main()
