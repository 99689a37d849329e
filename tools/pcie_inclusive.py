#!/usr/bin/env python3
"""End-to-end time of one gridSurface call that starts from host arrays (engine creation, table build, upload,
sweep, download, argmin), for the PCIe-inclusive figure quoted beside the resident-input benchmark."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import epidemicsimulator_amd as E
from epidemicsimulator_amd import synth

for name in sys.argv[1:] or ["c3"]:
    wl = synth.workload(name)
    n, T = len(wl["countries"]), wl["steps"]
    G = len(wl["betas"]) * len(wl["travel"])
    base = synth.base_state(T)
    pops = [c.pop for c in wl["countries"]]
    table = E.ProfileTable(wl["profiles"], pops)   # what a caller holds between engines (a JVM shim: one MemorySegment)
    for rep in range(3):
        t0 = time.perf_counter()
        eng = E.GridEngine(E.HyperParams(), T, base, wl["seed_idx"], table.pops if rep else pops, wl["conn"], table if rep else wl["profiles"],
                           wl["betas"], wl["travel"], wl["t_arrive"], wl["t_peak"], True)
        t1 = time.perf_counter()
        out = eng.run(want_metrics=False)
        best = E.grid_argmin(out["loss"])
        t2 = time.perf_counter()
        eng.close()
        print("%s rep %d (%s): create %.1f ms, sweep+download+argmin %.1f ms, end to end %.4g country-steps/s" %
              (name, rep, "profile table" if rep else "profile objects", (t1 - t0) * 1e3, (t2 - t1) * 1e3, G * n * T / (t2 - t0)))
