"""Timing probe: the sparse sweep (16 x 16 grid, 200 steps, mean degree 16) at several world sizes.  The engine's tables
grow with the world (12 bytes per stored weight), the work per patch does not: a rate that falls with the size shows what
the memory system costs the kernel."""
import sys
import time

import torch

import epidemicsimulator_amd as E
from epidemicsimulator_amd import synth


def sparse_workload(n, seed=synth.DEFAULT_SEED):
    wl = synth.workload("c4", seed)
    if n == len(wl["countries"]):
        return wl
    countries, xy = synth.make_countries(n, seed)
    ta, tp = synth.targets(n, wl["steps"], seed)
    out = dict(wl)
    out.update(countries=countries, profiles=countries, conn=synth.gravity_csr(countries, xy, seed), t_arrive=ta, t_peak=tp)
    return out


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
    return len(wl["countries"]) * T * G / dt


if __name__ == "__main__":
    for n in [int(a) for a in sys.argv[1:]] or [2500, 5000, 10000, 20000]:
        print("%6d patches: %.4g country-steps/s" % (n, timed(sparse_workload(n))), flush=True)
