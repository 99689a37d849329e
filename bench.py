#!/usr/bin/env python3
"""Calibration-sweep benchmark: country-steps/s over the whole beta x travel grid.

  python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c1|c4|c5]

One "step" is one pass of the hot path over the whole grid of the workload: every (beta, travel)
point is one episode of T world steps over n countries, so a step processes G*n*T country-steps.
Inputs (graph, tables, targets) are resident in HBM before the timed region starts.

With N > 1 (launched by torch.distributed.run, one rank per GPU) the flat grid is cut into N
interleaved shares (rank r runs points r, r + N, ...); after its share every rank takes part in one RCCL all-gather of the
fp64 loss surface and then finds the first minimum.  Total work is fixed: "scaling": "strong".

Rank 0 prints one JSON line.  `roofline` prices the episode kernel by its ALGORITHMIC bytes
(8*n + 68 per country-step for dense graphs) over the kernel time measured with HIP events on the
launch stream; `cpu_baseline` is the CPU oracle (a C restatement of the reference's Scala path) timed on
a bounded sample of the same workload on the host cores.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def algorithmic_bytes_per_country_step(n, conn):
    if isinstance(conn, tuple):  # CSR: 12 B per stored edge + 8 B row-pointer pair + 68 B of state/noise/action
        rp = conn[0]
        return 12.0 * float(rp[-1]) / n + 8.0 + 68.0
    return 8.0 * n + 68.0


VALU_PEAK_GINST = 1024 * 2.4 / 4.0   # G wave-instructions/s: 256 CUs x 4 SIMDs, 2.4 GHz, a wave64 instruction takes 4 cycles


def profile_summary(workload):
    """The newest committed rocprofv3 summary of this workload under profiles/ (PMC counters are collected in
    rocprofv3 runs of their own, see profiles/collect.sh and profiles/summarize.py; they cannot be read inside
    this process).  ({}, None) when none exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_%s.json" % workload)))
    if not files:
        return {}, None
    with open(files[-1]) as f:
        return json.load(f), os.path.relpath(files[-1], ROOT)


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/."""
    d, src = profile_summary(workload)
    return d.get("traffic_bytes_per_launch"), src


def valu_roofline(workload, country_steps, kernel_ms, traffic):
    """The resource that limits the episode kernels in practice: VALU issue.  Wave instructions per country-step come
    from the committed SQ_INSTS_VALU pass, the time from this run's HIP events; next to it the HBM rate actually
    measured (counter bytes over this run's kernel time)."""
    d, src = profile_summary(workload)
    per_cs = d.get("valu_insts_per_country_step")
    if per_cs is None and d.get("counters_one_sweep", {}).get("SQ_INSTS_VALU") and d.get("bench_line"):
        per_cs = d["counters_one_sweep"]["SQ_INSTS_VALU"] / d["bench_line"]["config"]["country_steps_per_step"]
    if not per_cs or kernel_ms <= 0:
        return None
    achieved = per_cs * country_steps / (kernel_ms * 1e-3) / 1e9
    out = {"bound": "valu-issue", "achieved": achieved, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s",
           "frac": achieved / VALU_PEAK_GINST, "valu_wave_insts_per_country_step": per_cs,
           "salu_wave_insts_per_country_step": d.get("salu_insts_per_country_step"),
           "source": "%s: SQ_INSTS_VALU of a separate rocprofv3 --pmc pass of this command; time = this run's HIP events; "
                     "peak = 1,024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction (fp64 instructions measure 4.8 cycles, "
                     "tools/op_rate.hip, so the real ceiling is lower)" % src}
    return out


def host_cores():
    """Host cores the cpu_baseline leg may use: the cores this process is allowed on, capped by the share the
    machine's scheduler hands a job (OMP_NUM_THREADS on a shared GPU box, where os.cpu_count() shows every core of
    the machine, not the job's share)."""
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    cap = os.environ.get("OMP_NUM_THREADS", "")
    if cap.isdigit() and int(cap) > 0:
        allowed = min(allowed, int(cap))
    return max(1, allowed)


def cpu_baseline(wl, base, threads, seconds_target=15.0):
    """Oracle timed on a bounded sample: `threads` grid points (one per thread), the first `steps`
    steps of each, sized from a short probe so the leg takes about seconds_target."""
    import numpy as np
    from oracle import oracle_lib as O
    from tests import util
    O.build()
    pops = [c.pop for c in wl["countries"]]
    profs = [util.prof_dict(p) for p in wl["profiles"]]
    W = wl["conn"] if isinstance(wl["conn"], tuple) else util.dense_of(wl["conn"], len(pops))   # CSR graphs stay CSR
    nt = len(wl["travel"])
    G = len(wl["betas"]) * nt

    def run(g0, steps, out, k, min_seconds=0.0):
        b = util.oracle_state(base)
        b.t_max = steps
        t = time.perf_counter()
        done, reps = 0, 0
        while True:  # small workloads: repeat the point until the leg is long enough to time
            d, _ = O.bench_points(pops, W, wl["seed_idx"], b, profs, steps, wl["betas"], wl["travel"], True,
                                  (g0 + reps) % G, 1)
            done += d
            reps += 1
            if time.perf_counter() - t >= min_seconds:
                break
        out[k] = (done, time.perf_counter() - t, reps)

    probe = [None]
    probe_steps = max(1, min(wl["steps"], 4))
    run(0, probe_steps, probe, 0)
    per_step = probe[0][1] / probe_steps
    steps = int(max(1, min(wl["steps"], seconds_target / max(per_step, 1e-9))))
    threads = max(1, min(threads, G))
    res = [None] * threads
    floor_s = min(seconds_target, 10.0) if steps == wl["steps"] else 0.0
    ths = [threading.Thread(target=run, args=((k * (G // threads)) % G, steps, res, k, floor_s)) for k in range(threads)]
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    wall = time.perf_counter() - t0
    total = sum(r[0] for r in res)
    points = sum(r[2] for r in res)
    return {"value": total / wall, "unit": "country-steps/s", "cores": threads, "kind": "port",
            "host_cpu_count": os.cpu_count(), "cores_note": "cores = threads used = min(cores this process may run on, "
            "OMP_NUM_THREADS); host_cpu_count = os.cpu_count() of the whole machine",
            "sample": "%d grid points x first %d of %d steps on %d threads (one point at a time per thread), %.1f s wall; "
                      "single-thread probe %.0f country-steps/s" % (points, steps, wl["steps"], threads, wall,
                                                                     probe[0][0] / probe[0][1])}


def run_c5(args, rank, local_rank, world, use_dist):
    """Batched DDQN rollout leg: 4,096 instances of the calibrated toy world (best beta of the toy grid,
    heterogeneous env parameters as Calibrate.scala:34-51), one epi_world_step per iteration with caller-supplied
    actions int32[B][n]; the states and rewards of every instance come back to the host each step, as the
    trainer consumes them.  More GPUs are independent replicas (weak scaling), no collective on the data path."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import epidemicsimulator_amd as E
    from epidemicsimulator_amd import synth

    B = 4096
    beta, _travel = synth.calibrated_toy_point(local_rank)   # "the calibrated config": the C1 sweep's first minimum, computed here
    countries = list(E.ToyConn.C)
    n = len(countries)

    def make_env_for(name, pop):  # Calibrate.scala:34-51
        p = E.Geo.CountryProfiles.get_profile(name)
        fb = beta * (1.0 + (p.population_density / 1000.0) * 0.5) * (3.0 - p.healthcare_capacity * 2.0)
        fi = 0.008 * (1.0 + p.age_distribution_elderly * 2.0)
        fn = 0.05 * (1.0 + (1.0 - p.economic_resilience))
        return E.EpidemicEnv(fb, 0.12, fi, fn, 5.0, 0.0, name, [c.name for c in countries].index(name) + 42)

    T = 300
    base = synth.base_state(T)
    w = E.WorldToy(E.HyperParams(), base, 0, E.ToyConn.W, make_env_for=make_env_for, validate=True, batch=B,
                   device=local_rank)
    rng = np.random.Generator(np.random.PCG64(20251226 + rank))
    acts = rng.integers(0, 8, size=(64, B, n), dtype=np.int32)  # a ring of pre-drawn action batches
    states, rewards = w.result_arrays()   # the handle's page-locked result arrays, read in place by the trainer
    packed = args.c5_records == "packed"
    resident = args.c5_records == "none"
    if packed:
        states = w.packed_array()
    step = w.step_arrays_packed if packed else w.step_arrays
    if resident:
        def step(a, g, t_max, _states, _rewards):
            w.step_resident(a, g, t_max)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    k = 0
    for _ in range(args.warmup):
        step(acts[k % 64], None, T, states, rewards); k += 1
    fence()
    w.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if k % T == 0:
            w.reset_to_initial()  # episodes of T steps, as the trainer's epochs
        step(acts[k % 64], None, T, states, rewards); k += 1
    fence()
    elapsed = time.perf_counter() - t0
    k_ms, k_launches, k_cs = w.kernel_time()
    w.set_timing(False)
    tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    bpcs = 8.0 * n + 68.0
    achieved = (k_cs * bpcs / (k_ms * 1e-3)) / 1e9 if k_ms > 0 else 0.0
    traffic, traffic_src = measured_traffic("c5") if world == 1 else (None, None)
    out = {
        "metric": "country-steps/sec (batched world step, whole node)",
        "value": float(world) * B * n * args.steps / elapsed, "unit": "country-steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / max(1, args.steps) * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("c5: %d instances of the %d-country toy world at the calibrated beta %.2f, one world step per "
                                "iteration, actions in, states%s and rewards out over PCIe every step"
                                % (B, n, beta, " (32-byte records)" if packed else ""))
                               if not resident else
                               ("c5: %d instances of the %d-country toy world at the calibrated beta %.2f, one world step per "
                                "iteration, actions in, states and rewards left on the device (NULL outputs)" % (B, n, beta)),
                   "calibrated_beta": beta,
                   "instances_per_gpu": B, "country_steps_per_step": world * B * n,
                   "parallelism": "independent replicas" if world > 1 else "single GPU"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": traffic_src, "kernel": "episode_tiny_kernel",
                     "launches": int(k_launches), "avg_launch_ms": k_ms / max(1, k_launches),
                     "algorithmic_bytes_per_country_step": bpcs,
                     "algorithmic_bytes_per_launch": B * n * bpcs,
                     "note": "one launch per step of 4,096 six-country worlds; the kernel reads the actions from and writes "
                             "states and rewards to page-locked host memory itself, so avg_launch_ms is mostly the PCIe "
                             "transfer (%.1f MB per step) and the HBM roofline does not describe it"
                             % (B * n * (4.0 if resident else (52.0 if packed else 84.0)) / 1e6)},
    }
    # the bus this workload is defined to cross: result bytes per step over the kernel time (the kernel writes them to
    # page-locked host memory itself), against the raw rate of a PCIe 5.0 x16 link (32 GT/s x 16 lanes x 128/130)
    out_bytes = B * n * (0.0 if resident else (32.0 + 8.0 if packed else 64.0 + 8.0)) + B * 16.0
    out["roofline_pcie"] = {"bound": "pcie 5.0 x16, device to host", "achieved": out_bytes * k_launches / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0,
                            "peak": 63.0, "unit": "GB/s", "result_bytes_per_step": out_bytes,
                            "note": "kernel time includes a fixed 0.015 ms (the step with nothing returned, --c5-records none); "
                                    "the returned bytes beyond that travel at about 56 GB/s (full vs packed vs none records)"}
    out["roofline_pcie"]["frac"] = out["roofline_pcie"]["achieved"] / 63.0
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            c1 = synth.workload("c1")
            cb = cpu_baseline(c1, synth.base_state(c1["steps"]), args.cpu_threads)
            cb["sample"] = "toy-world rollouts of the oracle (same 6-country world and arithmetic, actions from the " \
                           "agent stream): " + cb["sample"]
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    w.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def run_learn(args, rank, local_rank, world, use_dist):
    """DDQN learn leg (outside the headline metric): `A` independent agents with the reference's default
    hyper-parameters (hidden 128, batch 128, gradAccumSteps 4 -> 512 samples per learn) each take one
    DDQNAgent.learn per iteration, all agents in one set of launches.  More GPUs hold more agents (weak)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import epidemicsimulator_amd as E

    A, fill = 192, 2048
    hp = E.HyperParams()
    agents = E.Agents([hp] * A, max_grad_accum=4, device=local_rank)
    rng = np.random.Generator(np.random.PCG64(77 + rank))

    def state():
        pop = int(rng.integers(10_000, 50_000_000)); i = int(rng.integers(0, pop // 50)); r = int(rng.integers(0, pop // 10))
        return E.State(float(pop - i - r), float(i), float(r), float(rng.integers(0, pop // 500)), 0.0,
                       float(rng.integers(50, 100_000)), int(rng.integers(0, 60)), 60, 1)

    pool = [state() for _ in range(256)]
    for k in range(fill):
        agents.observe([E.Transition(pool[(k + a) % 256], int(rng.integers(0, 8)), float(rng.normal() * 2), pool[(k + a + 1) % 256],
                                     bool(k % 11 == 0)) for a in range(A)])
    need = hp.batch_size * 4
    flop_per_sample = 2.0 * (3 * (7 * 128 + 128 * 128 + 128 * 8) + 2 * 128 * 128 + 7 * 128 + 3 * 128)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    step = 1
    for _ in range(args.warmup):
        agents.learn([step] * A, None, 4); step += 1
    fence()
    agents.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        agents.learn([step] * A, None, 4); step += 1
    fence()
    elapsed = time.perf_counter() - t0
    k_ms, k_calls, k_samples = agents.kernel_time()
    tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    achieved = k_samples * flop_per_sample / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    peak = 78.6 / 2.0   # vector fp64 peak counts an FMA as two flops; this path may not fuse, so one flop per issue slot
    out = {
        "metric": "samples/sec (DDQN learn step, whole node)",
        "value": float(world) * A * need * args.steps / elapsed, "unit": "samples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / max(1, args.steps) * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "learn: %d agents, hidden 128, %d samples per learn (batch 128 x gradAccumSteps 4), replay of "
                               "%d transitions each, shuffle on the host" % (A, need, fill),
                   "agents_per_gpu": A, "parallelism": "independent agents" if world > 1 else "single GPU"},
        "roofline": {"bound": "valu-fp64 (unfused, ordered sums: neither hbm nor mfma bounds this path)",
                     "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                     "kernel": "agents_forward_kernel + agents_backward_kernel + agents_grad_w2_kernel + agents_grad_small_kernel",
                     "launches": int(k_calls), "avg_launch_ms": k_ms / max(1, k_calls),
                     "flop_per_sample": flop_per_sample},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle_lib as O
            L = O.Learner(O.default_learn_hyper())
            for k in range(fill):
                a, b = pool[k % 256], pool[(k + 1) % 256]
                L.observe(O.State(a.s, a.i, a.r, a.d, a.v, a.hosp_cap, a.t, a.t_max, a.seed, 0), int(k % 8), 0.5,
                          O.State(b.s, b.i, b.r, b.d, b.v, b.hosp_cap, b.t, b.t_max, b.seed, 0), k % 11 == 0)
            L.set_steps(1)
            t1 = time.perf_counter(); reps = 0
            while time.perf_counter() - t1 < 10.0:
                L.learn(hp.batch_size, 4); reps += 1
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": reps * need / dt, "unit": "samples/s", "cores": 1, "kind": "port",
                                   "sample": "%d learn calls of one agent (512 samples each) in the C oracle" % reps}
        print(json.dumps(out), flush=True)
    agents.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def run_multi_in_process(args):
    """--multi-in-process: the same sweep driven from ONE process over several GPUs through the library's own
    multi-device path (epi_grid_multi_*: an engine and a stream per share, RCCL all-gather called from C), printing the
    same JSON line, so that it can be put on the scaling curve next to the one-process-per-GPU path.  --multi-devices
    names the device of every share (default 0 .. N-1; a device may be named more than once)."""
    import epidemicsimulator_amd as E
    from epidemicsimulator_amd import synth
    devices = [int(x) for x in args.multi_devices.split(",")] if args.multi_devices else list(range(args.gpus))
    seen = E.lib().epi_device_count()
    if seen < 1 or max(devices) >= seen:
        raise SystemExit("bench.py: --multi-in-process over devices %s needs %d HIP devices, %d visible"
                         % (devices, max(devices) + 1, seen))
    wl = synth.workload(args.workload)
    if args.workload.lower() in ("c5", "learn"):
        raise SystemExit("bench.py: --multi-in-process drives the calibration sweep (c1..c4)")
    n, T = len(wl["countries"]), wl["steps"]
    nb, nt = len(wl["betas"]), len(wl["travel"])
    G = nb * nt
    base = synth.base_state(T)
    m = E.MultiGridEngine(E.HyperParams(), T, base, wl["seed_idx"], [c.pop for c in wl["countries"]], wl["conn"],
                          wl["profiles"], wl["betas"], wl["travel"], wl["t_arrive"], wl["t_peak"], True, devices=devices)
    ranks, slots = m.ranks()
    for _ in range(args.warmup):
        m.run(want_metrics=False)
    m.set_timing(True)
    t0 = time.perf_counter()
    out = None
    for _ in range(args.steps):
        out = m.run(want_metrics=False)          # synchronous: returns with the surface on the host and the first minimum
    elapsed = time.perf_counter() - t0
    tm = m.times()
    peer = m.peer_access().tolist()
    m.close()
    total_cs = float(G) * n * T * args.steps
    bpcs = algorithmic_bytes_per_country_step(n, wl["conn"])
    share_ms = [float(x) / max(1, args.steps) for x in tm["share_kernel_ms"]]
    slowest = max(share_ms) if share_ms else 0.0
    per_share_cs = [float(len(range(k, G, len(devices)))) * n * T for k in range(len(devices))]
    k_slow = share_ms.index(slowest) if share_ms else 0
    achieved = per_share_cs[k_slow] * bpcs / (slowest * 1e-3) / 1e9 if slowest > 0 else 0.0
    line = {
        "metric": "country-steps/sec (calibration grid, whole node)",
        "value": total_cs / elapsed, "unit": "country-steps/s", "n_gpus": ranks, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / max(1, args.steps) * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %d countries, %dx%d beta x travel grid, %d-step episodes, %s" % (
            wl["name"], n, nb, nt, T, "CSR graph" if isinstance(wl["conn"], tuple) else "dense graph"),
            "grid_points": G, "country_steps_per_step": G * n * T,
            "parallelism": "one process, %d interleaved shares on devices %s (%d distinct), RCCL all-gather from C "
                           "(epi_grid_multi_run)" % (len(devices), devices, ranks),
            "best_idx": int(out["best_idx"]), "best_loss": float(out["best_loss"]), "rccl_ranks": ranks,
            "share_slots_per_rank": slots, "kernel_ms_per_step_by_share": [round(x, 3) for x in share_ms],
            "all_gather_ms_per_step": tm["gather_ms"] / max(1, args.steps),
            "host_ms_per_step_inside_the_library": tm["wall_ms"] / max(1, args.steps), "peer_access": peer},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel": "episode kernels of the slowest share (%d)" % k_slow,
                     "algorithmic_bytes_per_country_step": bpcs,
                     "note": "shares that sit on one device run concurrently, so a share's kernel time there includes "
                             "waiting for the others; with one share per device it is that device's kernel time"},
    }
    print(json.dumps(line), flush=True)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpu_count():
    """GPUs this machine offers, counted WITHOUT loading the HIP runtime: the kernel driver lists its nodes under
    /sys/class/kfd/kfd/topology/nodes, a GPU being a node with a non-zero gfx_target_version (CPUs have 0); a
    *_VISIBLE_DEVICES list in the environment caps the figure.  An upper bound only -- every rank still checks the
    device it is given -- but enough to refuse `--gpus N` on a machine that cannot hold N ranks before any starts."""
    import glob
    seen = 0
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(path) as f:
                for line in f:
                    key, _, val = line.partition(" ")
                    if key == "gfx_target_version" and val.strip() not in ("", "0"):
                        seen += 1
        except OSError:
            pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        lst = os.environ.get(var)
        if lst is not None:
            seen = min(seen, len([x for x in lst.split(",") if x.strip() != ""]))
    return seen


def launch_ranks(n_ranks, argv, poll_s=0.05, grace_s=5.0):
    """Starts `n_ranks` copies of this script, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, exactly what torch.distributed.run would set) and supervises them: all children are polled together;
    when one ends with a non-zero status the others -- which would otherwise sit in the rendezvous or in a collective
    until its timeout -- are terminated (SIGTERM, SIGKILL after `grace_s`), the failing rank and its status are named
    on stderr and that status is returned.  Returns 0 when every rank ended with 0.  The parent imports neither torch
    nor the engine and makes no HIP call: the children are plain child processes of a process that never held a GPU."""
    import signal
    import subprocess
    port = _free_port()
    procs = []
    try:
        for r in range(n_ranks):
            env = dict(os.environ)
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
        failed = None
        live = set(range(n_ranks))
        while live and failed is None:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0 and failed is None:
                    failed = (r, rc)
            if live and failed is None:
                time.sleep(poll_s)
        if failed is None:
            return 0
        r, rc = failed
        why = ("killed by signal %d" % -rc) if rc < 0 else ("exit status %d" % rc)
        sys.stderr.write("bench.py: rank %d of %d failed (%s); stopping the other %d rank(s)\n"
                         % (r, n_ranks, why, len(live)))
        sys.stderr.flush()
        return abs(rc) if abs(rc) < 256 else 1
    finally:
        # whatever way this function is left (a failed rank, an exception, a signal turned into KeyboardInterrupt):
        # no child survives the launcher
        rest = [p for p in procs if p.poll() is None]
        for p in rest:
            try:
                p.send_signal(signal.SIGTERM)
            except OSError:
                pass
        deadline = time.monotonic() + grace_s
        for p in rest:
            try:
                p.wait(timeout=max(0.0, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()


def run_rehearsal(args, rank, world):
    """--rehearse-cpu: the launcher, the process group (gloo), the interleaved split, the all-gather of the loss
    pieces and the first-minimum scan, with NO engine behind them (the engine needs a GPU): every rank fills its
    share with a fixed function of the grid index.  A test of the multi-rank plumbing, not a benchmark line."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import datetime
    from epidemicsimulator_amd.dist import first_min, gather_surface_strided, strided_bounds
    if args.fail_rank == rank and args.fail_at == "start":
        raise SystemExit(7)            # a rank that dies before the rendezvous (bad device, failed RCCL start-up)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.rendezvous_timeout))
    if args.fail_rank == rank and args.fail_at == "collective":
        os._exit(9)                    # a rank that dies with its peers already inside a collective
    if dist.get_world_size() != args.gpus:
        raise SystemExit("bench.py: process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))
    G = 4096
    surface = ((np.arange(G, dtype=np.float64) * 37.0 + 55.0) % 101.0) + 1.0   # repeated minima: "first" matters
    g0, stride, cnt, per = strided_bounds(G, world, rank)
    local = torch.full((per,), -1.0, dtype=torch.float64)
    local[:cnt] = torch.from_numpy(surface[g0::stride].copy())
    full = gather_surface_strided(local, G, world, force=True)
    idx, best = first_min(full)
    ok = bool(np.array_equal(full, surface))
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([cnt], dtype=torch.int64))
    if rank == 0:
        print(json.dumps({"rehearsal": True, "backend": "gloo", "n_gpus": dist.get_world_size(), "grid_points": G,
                          "points_per_rank": [int(c.item()) for c in counts], "surface_ok": ok,
                          "best_idx": int(idx), "best_loss": float(best)}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--c5-records", choices=("full", "packed", "none"), default="full",
                    help="c5 only: states come back as 64-byte records (epi_world_step), as 32-byte ones "
                         "(epi_world_step_packed), or not at all (NULL outputs: the states stay on the device for "
                         "device-side consumers, only the actions cross the bus)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the cpu_baseline leg; 0 = the host cores this process may use")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="multi-rank plumbing only, over gloo, no engine (see run_rehearsal)")
    ap.add_argument("--multi-in-process", action="store_true",
                    help="one process drives --gpus N devices through epi_grid_multi_* (RCCL from C) instead of one "
                         "rank per GPU")
    ap.add_argument("--multi-devices", default="",
                    help="with --multi-in-process: comma-separated device of every share (default 0..N-1)")
    ap.add_argument("--fail-rank", type=int, default=-1,
                    help="rehearsal only: this rank dies on purpose (launcher supervision test)")
    ap.add_argument("--fail-at", choices=("start", "collective"), default="start",
                    help="rehearsal only: where --fail-rank dies: before the rendezvous or after it, with its peers in "
                         "the all-gather")
    ap.add_argument("--rendezvous-timeout", type=float, default=180.0,
                    help="seconds a rank waits in init_process_group / a collective for its peers before giving up")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")

    # ---- N ranks, one per GPU.  Launched by torch.distributed.run the environment already names this process's
    # rank; launched plainly with --gpus N > 1 this process only starts the N ranks and waits for them.
    if args.multi_in_process:
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("bench.py: --multi-in-process is one process for all devices; do not launch it as several ranks")
        return run_multi_in_process(args)

    # The launcher imports neither torch nor the engine: it must never bring up the HIP runtime, because the ranks
    # are started from it and a process that holds a GPU may not start others on this kind of machine.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if not args.rehearse_cpu:
            seen = visible_gpu_count()
            if seen < args.gpus:
                raise SystemExit("bench.py: --gpus %d needs %d HIP devices, %d visible on this machine: refusing to "
                                 "run fewer ranks than asked for" % (args.gpus, args.gpus, seen))
        import signal
        signal.signal(signal.SIGTERM, lambda *_: sys.exit(143))   # so that launch_ranks' clean-up runs
        rc = launch_ranks(args.gpus, sys.argv[1:])
        if os.environ.get("EPI_BENCH_LAUNCHER_REPORT") == "1":
            heavy = sorted(m for m in ("torch", "numpy", "epidemicsimulator_amd") if m in sys.modules)
            sys.stderr.write("bench.py launcher: modules loaded in the parent: %s\n" % (heavy or "none of torch/numpy/engine"))
        raise SystemExit(rc)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: launched with WORLD_SIZE=%d but --gpus %d: the two must agree (one rank per GPU)"
                         % (world, args.gpus))
    if args.rehearse_cpu:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        return run_rehearsal(args, rank, world)

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d wants device %d but only %d are visible" % (rank, local_rank,
                                                                                         torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    # EPI_BENCH_FORCE_DIST=1 runs the RCCL exchange even with a single rank (rehearsal on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("EPI_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        import datetime
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                timeout=datetime.timedelta(seconds=args.rendezvous_timeout))
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))
        world = dist.get_world_size()
        # RCCL builds its rings on the first collective: do that here, outside every timed region, whatever --warmup says
        warm = torch.zeros(max(1, world), dtype=torch.float64, device="cuda")
        dist.all_reduce(warm)
        dist.all_gather_into_tensor(torch.empty(world * world, dtype=torch.float64, device="cuda"), warm)
        torch.cuda.synchronize()
    if args.cpu_threads <= 0:
        args.cpu_threads = host_cores()

    if args.workload.lower() == "c5":
        return run_c5(args, rank, local_rank, world, use_dist)
    if args.workload.lower() == "learn":
        return run_learn(args, rank, local_rank, world, use_dist)

    import epidemicsimulator_amd as E
    from epidemicsimulator_amd import synth

    wl = synth.workload(args.workload)
    n, T = len(wl["countries"]), wl["steps"]
    nb, nt = len(wl["betas"]), len(wl["travel"])
    G = nb * nt
    base = synth.base_state(T)
    eng = E.GridEngine(E.HyperParams(), T, base, wl["seed_idx"], [c.pop for c in wl["countries"]], wl["conn"],
                       wl["profiles"], wl["betas"], wl["travel"], wl["t_arrive"], wl["t_peak"], True, device=local_rank)

    # contiguous beta-major slabs, padded so that the all-gather pieces have equal size
    # interleaved split: rank r runs grid points r, r + world, ...; shares padded so that the all-gather pieces have equal size
    from epidemicsimulator_amd.dist import gather_surface_strided, strided_bounds
    g0, gstride, cnt, per = strided_bounds(G, world, rank)
    loss_local = torch.zeros(per, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def one_step():
        if cnt > 0:
            eng.enqueue(g0, cnt, loss_local.data_ptr(), stream, g_stride=gstride)
        host = gather_surface_strided(loss_local, G, world, force=use_dist)  # 8*G bytes back to the host
        return E.grid_argmin(host)                   # Calibrate.scala:112's strict '<' scan

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    eng.status()
    eng.set_timing(True)
    fence()
    t0 = time.perf_counter()
    best = None
    for _ in range(args.steps):
        best = one_step()
    fence()
    elapsed = time.perf_counter() - t0
    eng.status()
    k_ms, k_launches, k_cs = eng.kernel_time()
    eng.set_timing(False)

    tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    per_rank = torch.tensor([k_ms / max(1, args.steps), float(cnt)], dtype=torch.float64, device="cuda")
    per_rank_all = per_rank.unsqueeze(0)
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        per_rank_all = torch.empty(world, 2, dtype=torch.float64, device="cuda")
        dist.all_gather_into_tensor(per_rank_all, per_rank)
    elapsed = float(tt.item())
    per_rank_all = per_rank_all.cpu().tolist()

    total_cs = float(G) * n * T * args.steps
    value = total_cs / elapsed
    bpcs = algorithmic_bytes_per_country_step(n, wl["conn"])
    achieved = (k_cs * bpcs / (k_ms * 1e-3)) / 1e9 if k_ms > 0 else 0.0
    traffic, traffic_src = measured_traffic(wl["name"]) if world == 1 else (None, None)
    out = {
        "metric": "country-steps/sec (calibration grid, whole node)",
        "value": value, "unit": "country-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / max(1, args.steps) * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %d countries, %dx%d beta x travel grid, %d-step episodes, %s" % (
            wl["name"], n, nb, nt, T, "CSR graph" if isinstance(wl["conn"], tuple) else "dense graph"),
            "grid_points": G, "country_steps_per_step": G * n * T,
            "parallelism": "interleaved shares of %d grid points per GPU + RCCL all-gather of the fp64 loss surface" % per if world > 1
            else "single GPU", "best_idx": int(best[0]), "best_loss": float(best[1]),
            "rccl_ranks": dist.get_world_size() if use_dist else 0,
            "kernel_ms_per_step_by_rank": [round(r[0], 3) for r in per_rank_all],
            "grid_points_by_rank": [int(r[1]) for r in per_rank_all]},
        "roofline_valu": valu_roofline(wl["name"], k_cs, k_ms, traffic) if world == 1 else None,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "hbm_measured_gbs": (traffic * k_launches / (k_ms * 1e-3) / 1e9) if traffic and k_ms > 0 else None,
                     "traffic_source": ("%s: per-launch (FETCH_SIZE*2 + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes "
                                        "of this command" % traffic_src) if traffic_src else None,
                     "kernel": "episode_csr_kernel" if isinstance(wl["conn"], tuple) else (
                         "episode_tiny_kernel" if n <= 16 else "episode_dense_kernel"),
                     "launches": int(k_launches),
                     "avg_launch_ms": k_ms / max(1, k_launches),
                     "algorithmic_bytes_per_country_step": bpcs,
                     "algorithmic_bytes_per_launch": k_cs * bpcs / max(1, k_launches),
                     "note": "rank-0 kernel time from HIP events on the launch stream; algorithmic bytes per country-step: "
                             "8n+68 dense, 12*deg+76 sparse"},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, base, args.cpu_threads)
        print(json.dumps(out), flush=True)
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
