"""Timing probe: the C4 sweep on the synthetic sparse graph as generated, and on the same graph with its patches
renumbered by falling row length (so that the engine's length-sorted row blocks coincide with the patch order).
Not a parity check: renumbering changes the column order that breaks ties."""
import sys
import time

import numpy as np
import torch

import epidemicsimulator_amd as E
from epidemicsimulator_amd import synth


def relabel(wl):
    rowptr, col, val = wl["conn"]
    n = len(wl["countries"])
    deg = np.diff(rowptr)
    order = np.argsort(-deg, kind="stable")          # new id k holds old patch order[k]
    new_of = np.empty(n, dtype=np.int64)
    new_of[order] = np.arange(n)
    rp = np.zeros(n + 1, dtype=np.int32)
    rp[1:] = np.cumsum(deg[order])
    c2 = np.zeros_like(col)
    v2 = np.zeros_like(val)
    for k, old in enumerate(order):
        a, b = rowptr[old], rowptr[old + 1]
        cc = new_of[col[a:b]]
        o = np.argsort(cc, kind="stable")
        c2[rp[k]:rp[k + 1]] = cc[o]
        v2[rp[k]:rp[k + 1]] = val[a:b][o]
    out = dict(wl)
    out["countries"] = [wl["countries"][o] for o in order]
    out["profiles"] = out["countries"]
    out["conn"] = (rp, c2, v2)
    out["t_arrive"] = {int(new_of[k]): v for k, v in wl["t_arrive"].items()}
    out["t_peak"] = {int(new_of[k]): v for k, v in wl["t_peak"].items()}
    out["seed_idx"] = int(new_of[wl["seed_idx"]])
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
    return len(wl["countries"]) * T * G / dt, float(loss.min().item())


if __name__ == "__main__":
    wl = synth.workload("c4")
    a = timed(wl)
    print("as generated        %.4g country-steps/s (best loss %.6g)" % a, flush=True)
    b = timed(relabel(wl))
    print("sorted by row length %.4g country-steps/s (best loss %.6g)" % b, flush=True)
