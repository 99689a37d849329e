#!/usr/bin/env python3
"""Randomised differential soak of the training loop: WorldToy.stepOne with learning agents (act, world step,
observe, per-country learn schedule) on the engine against the oracle world that learns, random hyper-parameters
per country.  States, rewards, losses and all networks compared with ==.
   python tools/soak_train.py [seconds] [seed]    (needs a GPU)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import epidemicsimulator_amd as E
from oracle import oracle_lib as O
from tests import util


def run(budget=120.0, seed=3, verbose=True):
    """Runs random cases for `budget` seconds; returns the number of cases (every one compared with ==)."""
    rng = np.random.default_rng(seed)
    O.build()
    t0 = time.time()
    cases = 0
    names = [c.name for c in E.ToyConn.C]
    while time.time() - t0 < budget:
        hidden = int(rng.choice([8, 16, 32, 64]))

        def hp():
            return E.HyperParams(hidden=hidden, batch_size=int(rng.integers(1, 6)), learn_every=int(rng.integers(1, 6)),
                                 epsilon_decay_steps=int(rng.integers(5, 80)), epsilon_end=float(rng.choice([0.0, 0.1, 0.5])),
                                 target_update_every=int(rng.integers(1, 9)), replay_capacity=int(rng.integers(20, 60)),
                                 lr=float(10 ** rng.uniform(-4, -2)), gamma=float(rng.uniform(0.5, 0.99)),
                                 grad_clip=float(rng.choice([0.01, 1.0])), ema_alpha=float(rng.random()))
        base_hp = hp()
        chp = {nm: hp() for nm in names if rng.random() < 0.5}
        t_max = int(rng.integers(30, 300))
        base = E.State(1e6, 0, 0, 0, 0, 10000.0, 0, t_max, 1)
        w = E.WorldToy(base_hp, base, validate=True, country_hyper_params=chp)
        envs = w._envs
        ow = O.World([c.pop for c in w.countries], np.asarray(E.ToyConn.W, dtype=np.float64), 0, util.oracle_state(base),
                     [O.EnvParams(e.base_beta, e.base_gamma, e.base_ifr, e.noise, e.reward_clip, e.risk_factor) for e in envs],
                     [util.prof_dict(e.country_profile) for e in envs], [e.seed for e in envs], validate=True)
        hps = [chp.get(nm, base_hp) for nm in names]
        ow.enable_learning([O.LearnHyper(h.gamma, h.lr, h.clip_q, h.grad_clip, h.ema_alpha, h.hidden, h.replay_capacity,
                                         h.target_update_every, h.batch_size * 4) for h in hps],
                           [O.Hyper(h.epsilon_start, h.epsilon_end, h.epsilon_decay_steps, h.hidden, h.clip_q, h.reward_clip,
                                    h.batch_size, h.learn_every) for h in hps],
                           [h.learn_every for h in hps], [h.batch_size for h in hps], [2 if nm == "India" else 1 for nm in names])
        steps = int(rng.integers(30, 90))
        learned = 0
        for step in range(steps):
            out = w.step_one(t_max)
            want_r, want_loss = ow.step_learn(t_max)
            assert ow.error()[0] == 0
            want_s = ow.states()
            for k, (nm, s, r) in enumerate(out):
                assert (s.s, s.i, s.r, s.d, s.v, s.hosp_cap, s.t) == want_s[k].tup(), (cases, step, nm)
                assert r == want_r[k], (cases, step, nm)
                got = w.last_losses[k]
                assert (got is None and want_loss[k] == -1.0) or got == want_loss[k], (cases, step, nm, got, want_loss[k])
                learned += got is not None
        for k in range(w.n):
            L = ow.learner(k)
            L.hp = O.LearnHyper(0, 0, 0, 0, 0, hidden, 1, 1, 1)
            for tgt in (False, True):
                for g, x in zip(w._pool.weights(k, tgt), L.weights(tgt)):
                    assert np.array_equal(g, x), (cases, k, tgt)
        w.close(); ow.close()
        cases += 1
        if verbose:
            print("ok hidden=%d steps=%d learn calls=%d" % (hidden, steps, learned), flush=True)
    if verbose:
        print("soak_train: %d cases, all equal" % cases)
    return cases


def main():
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 3)


if __name__ == "__main__":
    main()
