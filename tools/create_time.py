"""Wall time of epi_grid_create for the 20,000-patch sparse workload (first call loads the kernels)."""
import sys, time
sys.path.insert(0, '.')
import epidemicsimulator_amd as E
from epidemicsimulator_amd import synth
wl = synth.workload("c4")
T = wl["steps"]
args = (E.HyperParams(), T, synth.base_state(T), wl["seed_idx"], [c.pop for c in wl["countries"]], wl["conn"], wl["profiles"], E.f64(wl["betas"]), E.f64(wl["travel"]), wl["t_arrive"], wl["t_peak"], True)
d, keep = E._grid_desc(*args, 0)
import ctypes as C
L = E.lib()
for rep in range(4):
    t = time.perf_counter()
    h = L.epi_grid_create(C.byref(d))
    dt = time.perf_counter() - t
    assert h
    L.epi_grid_destroy(C.c_void_p(h))
    print("epi_grid_create %d: %.3f ms" % (rep, dt * 1e3))
