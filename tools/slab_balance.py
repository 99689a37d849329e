#!/usr/bin/env python3
"""How evenly the contiguous beta-major slabs of the C3 grid split the work: device time of each of the 8 slabs
of 512 grid points, run one after the other on one GPU, against a strided split of the same grid.
   python tools/slab_balance.py [workload] [ranks]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import epidemicsimulator_amd as E
from epidemicsimulator_amd import synth


def main():
    wl = synth.workload(sys.argv[1] if len(sys.argv) > 1 else "c3")
    ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    T = wl["steps"]
    G = len(wl["betas"]) * len(wl["travel"])
    eng = E.GridEngine(E.HyperParams(), T, synth.base_state(T), wl["seed_idx"], [c.pop for c in wl["countries"]], wl["conn"],
                       wl["profiles"], wl["betas"], wl["travel"], wl["t_arrive"], wl["t_peak"], True)
    per = G // ranks
    eng.run(0, per, want_metrics=False)   # warm-up
    times = []
    for r in range(ranks):
        eng.set_timing(True)
        eng.run(r * per, per, want_metrics=False)
        ms, launches, cs = eng.kernel_time()
        eng.set_timing(False)
        times.append(ms)
    t = np.array(times)
    print("contiguous slabs: ms per slab", np.round(t, 1).tolist())
    print("max %.1f ms, mean %.1f ms, max/mean %.3f -> strong-scaling efficiency bound %.3f" % (t.max(), t.mean(), t.max() / t.mean(), t.mean() / t.max()))
    if hasattr(eng, "run_strided"):
        times = []
        for r in range(ranks):
            eng.set_timing(True)
            eng.run_strided(r, ranks, per, want_metrics=False)
            ms, launches, cs = eng.kernel_time()
            eng.set_timing(False)
            times.append(ms)
        t = np.array(times)
        print("strided split:    ms per rank", np.round(t, 1).tolist())
        print("max %.1f ms, mean %.1f ms, max/mean %.3f -> strong-scaling efficiency bound %.3f" % (t.max(), t.mean(), t.max() / t.mean(), t.mean() / t.max()))


if __name__ == "__main__":
    main()
