"""Timing probe: the sparse sweep (16 x 16 grid, 200 steps, mean degree 16) at several world sizes.  The engine's tables
grow with the world (12 bytes per stored weight), the work per patch does not: a rate that falls with the size shows what
the memory system costs the kernel.  deg=<mean degree> arguments vary the row lengths instead (deg=1: nearly all
of the time is the local / apply pass)."""
import sys
import time

import torch

import epidemicsimulator_amd as E
from epidemicsimulator_amd import synth


def sparse_workload(n, seed=synth.DEFAULT_SEED, mean_degree=16.0):
    wl = synth.workload("c4", seed)
    if n == len(wl["countries"]) and mean_degree == 16.0:
        return wl
    countries, xy = synth.make_countries(n, seed)
    ta, tp = synth.targets(n, wl["steps"], seed)
    out = dict(wl)
    out.update(countries=countries, profiles=countries, conn=synth.gravity_csr(countries, xy, seed, mean_degree=mean_degree), t_arrive=ta, t_peak=tp)
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
    args = [a for a in sys.argv[1:] if not a.startswith("deg=")]
    deg = [float(a[4:]) for a in sys.argv[1:] if a.startswith("deg=")] or [16.0]
    for n in [int(a) for a in args] or [2500, 5000, 10000, 20000]:
        for d in deg:
            print("%6d patches, mean degree %4.1f: %.4g country-steps/s" % (n, d, timed(sparse_workload(n, mean_degree=d))), flush=True)
