#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from oracle/pyref.py (the pure-Python restatement of
the reference path).  The reference itself (Scala on the JVM) cannot run here, so these vectors are the
second restatement's outputs; tests check the C oracle and the GPU engine against them.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyref as P  # noqa: E402
from epidemicsimulator_amd import Geo, ToyConn  # noqa: E402
from tests.util import prof_dict  # noqa: E402


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)
        f.write("\n")


def main():
    # --- JVM random streams
    r = P.JavaRandom(42)
    ints = [r.next_int() for _ in range(4)]
    r = P.JavaRandom(42)
    dbls = [r.next_double() for _ in range(4)]
    gauss = {}
    for seed in (0, 7, 9, 42, 43, 47):
        r = P.JavaRandom(seed)
        gauss[str(seed)] = [r.next_gaussian().hex() for _ in range(8)]
    ag = P.Agent(P.DEFAULT_HP)
    acts = []
    for _ in range(520):
        eps = ag.epsilon()   # read before the step is counted (DDQNAgent.scala:26-27,32)
        ag.steps += 1
        acts.append(ag.rnd.next_int(8) if ag.rnd.next_double() < eps else -1)
    dump("jvm_streams.json", dict(next_int_seed42=ints, next_double_seed42=[d.hex() for d in dbls], gaussians=gauss,
                                  agent_actions=acts))

    # --- allocateInt
    cases = [(10, [.5, .3, .2]), (10, [1, 1, 1]), (7, [0, 2, 1, 1]), (1, [.25] * 4), (5, [0, 0, 0]), (0, [1, 2]),
             (-4, [1, 2]), (100, [0.1] * 7), (12345, [3.0, 1.5, 0.25, 0.0, 9.75]), (3, [1e-9, 1.0, 1e-9]),
             (2_000_000_000, [0.3, 0.3, 0.4]), (17, [1.0])]
    dump("alloc_int.json", [dict(total=t, weights=w, out=P.allocate_int(t, [float(x) for x in w])) for t, w in cases])

    # --- MobilitySpec.scala:29-30 inputs through the WorldToy mobility loop
    W = ToyConn.W.tolist()
    S = [1_000_000, 120_000, 1_000_000, 80_000, 60_000, 300_000]
    I = [1_000, 200, 900, 100, 80, 400]
    n = 6
    mS, mI = [0] * n, [0] * n
    for i in range(n):
        w_row = [0.0 if j == i else W[i][j] for j in range(n)]
        sw = max(0.0, P.seq_sum(w_row))
        eS = min(S[i], int(__import__("math").floor(S[i] * sw + 1e-9)))
        eI = min(I[i], int(__import__("math").floor(I[i] * sw + 1e-9)))
        aS, aI = P.allocate_int(eS, w_row), P.allocate_int(eI, w_row)
        for j in range(n):
            if j != i:
                mS[j] += aS[j]; mI[j] += aI[j]
        mS[i] -= eS; mI[i] -= eI
    dump("mobility_spec.json", dict(W=[[x.hex() for x in row] for row in W], S=S, I=I, moved_s=mS, moved_i=mI))

    # --- toy world: 6 countries, heterogeneous calibration envs, 120 steps (covers the greedy step 113)
    names = [c.name for c in ToyConn.C]
    pops = [c.pop for c in ToyConn.C]
    profs = [prof_dict(Geo.CountryProfiles.get_profile(nm)) for nm in names]
    steps = 120
    base = P.State(1e6, 0, 0, 0, 0, 10000.0, 0, steps, 1)
    arr, pk, fin, series = P.rollout(P.DEFAULT_HP, steps, base, pops, W, profs, 0, 0.25, 0.7, True, keep_series=True)
    dump("toy_rollout_hetero.json", dict(beta=0.25, travel=0.7, steps=steps, arrival=arr, peak=pk, final=fin,
                                         i_series=[[int(v) for v in s] for s in series]))
    # full state trajectory for 40 steps (homogeneous, beta 0.26, travel 1.0)
    Ws = P.scale_row(W, 0, 1.0)
    world = P.World(P.DEFAULT_HP, P.State(1e6, 0, 0, 0, 0, 10000.0, 0, 40, 1), pops, Ws,
                    P.make_calibration_envs(0.26, profs, P.DEFAULT_HP, False), 0, validate=True)
    traj = []
    for _ in range(40):
        rew = world.step_one(40)
        traj.append(dict(states=[[int(s.s), int(s.i), int(s.r), int(s.d), int(s.v), s.hosp_cap.hex(), s.t] for s in world.cur],
                         rewards=[x.hex() for x in rew], actions=list(world.last_a), flows=[list(world.flows[0]), list(world.flows[1])]))
    dump("toy_world_homog.json", dict(beta=0.26, travel=1.0, steps=40, trajectory=traj))

    # --- toy calibration surface: 2 x 3 grid, 60 steps, fallback targets of App.scala:99-102
    ta = {"Japan": 40.0, "India": 55.0, "Italy": 60.0, "Germany": 58.0, "USA": 52.0, "China": 0.0}
    tp = {"Japan": 10000.0, "India": 18000.0, "Italy": 8000.0, "Germany": 9000.0, "USA": 12000.0, "China": 35000.0}
    betas, tss = [0.24, 0.27], [0.5, 0.7, 0.9]
    rows, (best_loss, best) = P.grid_surface(P.DEFAULT_HP, 60, P.State(1e6, 0, 0, 0, 0, 10000.0, 0, 60, 1), pops, W, profs, 0,
                                             {names.index(k): v for k, v in ta.items()}, {names.index(k): v for k, v in tp.items()},
                                             betas, tss, True)
    dump("toy_grid.json", dict(betas=betas, travel=tss, steps=60, t_arrive=ta, t_peak=tp,
                               rows=[[b, t, l.hex()] for b, t, l in rows], best_loss=best_loss.hex(), best=list(best)))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
