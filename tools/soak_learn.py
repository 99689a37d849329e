#!/usr/bin/env python3
"""Randomised differential soak of the DDQN learners: random widths, batch sizes, capacities, hyper-parameters
and learn schedules, engine against the C oracle, everything compared with ==.
   python tools/soak_learn.py [seconds] [seed]    (needs a GPU)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import epidemicsimulator_amd as E
from oracle import oracle_lib as O


def state(rng, t):
    pop = int(rng.integers(1, 50_000_000))
    i = int(rng.integers(0, max(1, pop // 20) + 1)); r = int(rng.integers(0, pop // 10 + 1)); d = int(rng.integers(0, pop // 300 + 1))
    return (float(max(0, pop - i - r - d)), float(i), float(r), float(d), float(rng.integers(0, pop // 20 + 1)),
            float(rng.integers(0, 100_000)), int(t), int(rng.integers(1, 400)), 1)


def run(budget=120.0, seed=99, verbose=True):
    """Runs random cases for `budget` seconds; returns the number of cases (every one compared with ==)."""
    rng = np.random.default_rng(seed)
    O.build()
    t0 = time.time()
    cases = 0
    while time.time() - t0 < budget:
        hidden = int(rng.choice([1, 3, 16, 17, 48, 64, 65, 100, 128, 129, 200, 256]))
        batch = int(rng.integers(1, 40)); accum = int(rng.integers(1, 5))
        n_agents = int(rng.integers(1, 5))
        cap = int(rng.integers(batch * accum, 4 * batch * accum + 2))
        hps = [E.HyperParams(hidden=hidden, batch_size=batch, replay_capacity=cap, gamma=float(rng.random()),
                             lr=float(10 ** rng.uniform(-4, -1)), clip_q=float(rng.choice([0.5, 3.0, 10.0, 1e9])),
                             grad_clip=float(rng.choice([1e-3, 0.05, 1.0, 1e9])), ema_alpha=float(rng.random()),
                             target_update_every=int(rng.integers(1, 6))) for _ in range(n_agents)]
        agents = E.Agents(hps, max_grad_accum=accum)
        learners = [O.Learner(O.LearnHyper(h.gamma, h.lr, h.clip_q, h.grad_clip, h.ema_alpha, h.hidden, h.replay_capacity,
                                           h.target_update_every, batch * accum)) for h in hps]
        steps = [0] * n_agents
        for rnd in range(int(rng.integers(2, 6))):
            for k in range(int(rng.integers(1, cap + 10))):
                trs = []
                for a in range(n_agents):
                    s, s2 = state(rng, k), state(rng, k + 1)
                    act = int(rng.integers(0, 8)); r = float(rng.normal() * rng.choice([0.1, 2.0, 30.0])); done = bool(rng.random() < 0.2)
                    learners[a].observe(O.State(*s, 0), act, r, O.State(*s2, 0), done)
                    trs.append(E.Transition(E.State(*s), act, r, E.State(*s2), done))
                agents.observe(trs)
            ids = [a for a in range(n_agents) if rng.random() < 0.8] or [0]
            for a in ids:
                steps[a] += int(rng.integers(1, 5))
                learners[a].set_steps(steps[a])
            want = [learners[a].learn(batch, accum) for a in ids]
            loss, ema = agents.learn([steps[a] for a in ids], [batch] * len(ids), accum, ids)
            assert list(loss) == want, (hidden, batch, accum, cap, list(loss), want)
            assert list(ema) == [learners[a].ema_loss() for a in ids]
        for a, L in enumerate(learners):
            for tgt in (False, True):
                for g, w in zip(agents.weights(a, tgt), L.weights(tgt)):
                    assert np.array_equal(g, w), (hidden, batch, accum, cap, a, tgt)
        agents.close()
        cases += 1
        if verbose:
            print("ok hidden=%d batch=%d accum=%d cap=%d agents=%d" % (hidden, batch, accum, cap, n_agents), flush=True)
    if verbose:
        print("soak_learn: %d cases, all equal" % cases)
    return cases


def main():
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 99)


if __name__ == "__main__":
    main()
