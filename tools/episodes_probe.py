"""Timing probe: a sweep (default: the sparse one, 20,000 patches, 200 steps; or c3 / c2) with 16 grid points per argument.  One episode occupies one
compute unit, so up to 256 grid points the sweep time is the time of one episode -- unless the episodes get in each other's
way in the memory system, which is what this probe shows."""
import sys
import time

import numpy as np
import torch

import epidemicsimulator_amd as E
from epidemicsimulator_amd import synth


def timed(wl, reps=3):
    T = wl["steps"]
    G = len(wl["betas"]) * len(wl["travel"])
    eng = E.GridEngine(E.HyperParams(), T, synth.base_state(T), wl["seed_idx"], [c.pop for c in wl["countries"]], wl["conn"],
                       wl["profiles"], wl["betas"], wl["travel"], wl["t_arrive"], wl["t_peak"], True, device=0)
    loss = torch.zeros(G, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    eng.enqueue(0, G, loss.data_ptr(), stream)
    torch.cuda.synchronize()
    eng.status()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.enqueue(0, G, loss.data_ptr(), stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    eng.status()
    return dt


if __name__ == "__main__":
    args = sys.argv[1:]
    name = args.pop(0) if args and not args[0].isdigit() else "c4"
    base = synth.workload(name)
    base["travel"] = np.linspace(0.25, 2.0, 16).tolist()
    for nb in [int(a) for a in args] or [4, 8, 16, 32]:
        wl = dict(base, betas=np.linspace(0.20, 0.40, nb).tolist())
        dt = timed(wl)
        print("%4d grid points: sweep %.2f ms" % (nb * len(wl["travel"]), dt * 1e3), flush=True)
