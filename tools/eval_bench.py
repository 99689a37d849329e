#!/usr/bin/env python3
"""Timing of the two per-step paths a trainer drives, on the six-country toy world:
  (1) the batched world step (4,096 instances, actions in, states and rewards out every step), ms per step;
  (2) the greedy evaluation of MutliTrainer.scala:128-161 (800 stepOneGreedy calls): one C-ABI call per step
      against one epi_world_rollout_greedy call.
   python tools/eval_bench.py    (needs a GPU; prints one JSON line)"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import epidemicsimulator_amd as E


def main():
    out = {}
    base = E.State(1e6, 0, 0, 0, 0, 10000.0, 0, 800, 1)
    # (2) evaluation, one world
    w = E.WorldToy(E.HyperParams(), base, validate=True, learning=False)
    w._ensure_untrained_mlp()
    st = np.empty((1, 6), dtype=E.WorldToy.STATE_DTYPE)
    w.reset_to_initial()
    t0 = time.perf_counter()
    for _ in range(800):
        E.check(E.lib().epi_world_step_greedy(w._h, None, C.c_int32(800), C.cast(st.ctypes.data, C.POINTER(E._capi.CState)), None, None))
    out["eval800_one_call_per_step_ms"] = (time.perf_counter() - t0) * 1e3
    final_a = st.copy()
    for rep in range(2):
        w.reset_to_initial()
        w._envs  # the env streams simply continue, as in the reference
        t0 = time.perf_counter()
        r = w.rollout_greedy(800, 800)
        out["eval800_rollout_ms"] = (time.perf_counter() - t0) * 1e3
    w.close()
    # the same with per-node agents (trained networks live in the agents handle)
    w = E.WorldToy(E.HyperParams(), base, validate=True, learning=True)
    w._ensure_agents()
    w.reset_to_initial()
    t0 = time.perf_counter()
    for _ in range(800):
        w.step_one_greedy(800)
    out["eval800_agents_python_mirror_per_step_ms"] = (time.perf_counter() - t0) * 1e3
    for rep in range(2):
        t0 = time.perf_counter()
        w.evaluate_greedy(800)
        out["eval800_agents_evaluate_greedy_ms"] = (time.perf_counter() - t0) * 1e3
    w.close()
    # (1) batched step
    B = 4096
    w = E.WorldToy(E.HyperParams(), base, validate=True, batch=B, learning=False)
    rng = np.random.default_rng(1)
    acts = rng.integers(0, 8, (16, B, 6), dtype=np.int32)
    states, rewards = w.result_arrays()
    for k in range(20):
        w.step_arrays(acts[k % 16], None, 800, states, rewards)
    t0 = time.perf_counter()
    for k in range(300):
        w.step_arrays(acts[k % 16], None, 800, states, rewards)
    out["c5_ms_per_step_4096x6"] = (time.perf_counter() - t0) / 300 * 1e3
    w.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
