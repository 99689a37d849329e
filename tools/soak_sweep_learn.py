#!/usr/bin/env python3
"""Randomised differential soak of calibration sweeps whose rollouts learn (Calibrate.scala:19-82 calling WorldToy.stepOne with
its DDQN agents, WorldToy.scala:83-170): random dense worlds, grids, schedules and learner hyper-parameters, engine against
the oracle.
   python tools/soak_sweep_learn.py [seconds] [seed]    (needs a GPU; prints one line per case and a summary)"""
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


def run(budget=120.0, seed=31, verbose=True):
    """Runs random cases for `budget` seconds; returns (cases, cases whose sweep differs from the untrained one)."""
    rng = np.random.default_rng(seed)
    O.build()
    t0 = time.time()
    cases = 0
    while time.time() - t0 < budget:
        n = int(rng.choice([1, 2, 3, 6, 6, 9, 16, 17, 20, 33, 64, 70]))
        steps = int(rng.integers(30, 160)) if n <= 20 else int(rng.integers(20, 60))
        wseed = int(rng.integers(1, 1 << 30))
        nb, nt = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        if n == 6 and rng.random() < 0.5:
            wl = dict(synth.workload("c1"), steps=steps)
            wl["betas"] = sorted(rng.choice([0.22, 0.24, 0.25, 0.26, 0.27, 0.3], size=nb, replace=False).tolist())
            wl["travel"] = sorted(rng.choice([0.5, 0.6, 0.8, 0.9, 1.4], size=nt, replace=False).tolist())
        else:
            wl = synth.small_workload(max(n, 2), nb, nt, steps, seed=wseed)
            if n == 1:
                countries, xy = synth.make_countries(1, wseed)
                wl["countries"], wl["profiles"], wl["conn"] = countries, countries, np.zeros((1, 1))
                wl["t_arrive"], wl["t_peak"] = synth.targets(1, steps, wseed)
            wl["seed_idx"] = int(rng.integers(0, n))
        bs = int(rng.choice([1, 2, 4, 8]))
        cap = int(rng.choice([4 * bs, 4 * bs + 3, 100, 100000]))
        hp = E.HyperParams(batch_size=bs, learn_every=int(rng.integers(1, 7)), hidden=int(rng.choice([8, 24, 64, 128, 200])),
                           epsilon_start=float(rng.choice([0.1, 0.3, 0.6, 1.0])), epsilon_end=0.05,
                           epsilon_decay_steps=int(rng.integers(20, 100)), target_update_every=int(rng.integers(3, 30)),
                           lr=float(rng.choice([1e-3, 1e-2, 1e-1])), grad_clip=float(rng.choice([0.5, 5.0, 100.0])),
                           replay_capacity=cap, gamma=float(rng.choice([0.9, 0.99])))
        names = ["India" if rng.random() < 0.3 else "P%d" % k for k in range(n)]
        intensity = [E.WorldToy.LEARNING_INTENSITY.get(nm, 1) for nm in names]
        het = bool(rng.random() < 0.6)
        base = synth.base_state(steps)
        got = util.engine_grid(wl, base, het, hp=hp, names=names)
        want = util.oracle_grid_learn(wl, base, hp, intensity, het)
        util.assert_grid_equal(got, want)
        cases += 1
        if verbose:
            print("ok n=%d grid=%dx%d steps=%d bs=%d every=%d hidden=%d cap=%d eps0=%.1f het=%d" %
                  (n, nb, nt, steps, bs, hp.learn_every, hp.hidden, cap, hp.epsilon_start, het), flush=True)
    if verbose:
        print("soak_sweep_learn: %d cases, all equal" % cases)
    return cases


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 31)
