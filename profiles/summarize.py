#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of profiles/collect.sh into the committed summaries:
   python profiles/summarize.py gpurun_out/r1_c3 profiles/round1_c3

Writes <out>.json (read by bench.py for the `traffic` field) and <out>.md, and copies the kernel-stats CSV.
HBM traffic follows the gfx950 rule for FETCH_SIZE (reported in KiB and exactly half of the bytes of a
coalesced read; confirmed for this kernel's 8 B/lane pattern with tools/calib_fetch.hip: a 4 GiB read reports
2,097,155 KiB); WRITE_SIZE (KiB) is taken as is (the same tool: 4 GiB written with 4 B or 8 B per lane reports
4,194,304 KiB)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def newest(files):
    return [max(files, key=os.path.getmtime)] if files else []


def counters(path, kernel_sub):
    agg = collections.defaultdict(float)
    launches = collections.Counter()
    # gpurun merges results into an existing directory: only the newest run of each pass counts
    for f in newest(glob.glob(os.path.join(path, "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if kernel_sub in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                launches[r["Counter_Name"]] += 1
    return dict(agg), dict(launches)


def several_kernels(src, out, bench, names, stats_csv):
    """A step made of several kernels (the DDQN learn step): per-kernel times and VALU counters, no HBM roofline."""
    rows = {n: [r for r in csv.DictReader(open(stats_csv)) if n in r["Name"]][0] for n in names}
    total_ms = sum(float(r["AverageNs"]) for r in rows.values()) / 1e6
    sq = {n: counters(os.path.join(src, "sq"), n)[0] for n in names}
    summary = {"workload": bench["config"]["workload"], "bench_line": bench,
               "rocprof_avg_ms_per_kernel": {n: float(r["AverageNs"]) / 1e6 for n, r in rows.items()},
               "rocprof_avg_step_ms": total_ms, "bench_avg_launch_ms": bench["roofline"]["avg_launch_ms"],
               "sq_counters_one_step": sq}
    with open(out + ".json", "w") as f:
        json.dump(summary, f, indent=1)
        f.write("\n")
    with open(out + ".md", "w") as f:
        f.write("# rocprofv3 summary: %s\n\n" % summary["workload"])
        f.write("| quantity | value |\n|---|---|\n")
        f.write("| bench value | %.4g %s |\n" % (bench["value"], bench["unit"]))
        f.write("| achieved | %.2f %s = %.3f of %.1f (%s) |\n" % (bench["roofline"]["achieved"], bench["roofline"]["unit"],
                                                                  bench["roofline"]["frac"], bench["roofline"]["peak"], bench["roofline"]["bound"]))
        for n, r in rows.items():
            f.write("| `%s` avg, rocprofv3 --kernel-trace --stats | %.3f ms over %s calls |\n" % (n, float(r["AverageNs"]) / 1e6, r["Calls"]))
        f.write("| sum of the kernels, rocprofv3 | %.3f ms |\n" % total_ms)
        f.write("| device time of one learn call, HIP events in bench.py | %.3f ms |\n" % summary["bench_avg_launch_ms"])
        for n in names:
            for k in sorted(sq[n]):
                f.write("| %s: %s | %.4g |\n" % (n, k, sq[n][k]))
        if "cpu_baseline" in bench:
            f.write("| cpu_baseline | %.4g %s on %d threads (%s) |\n" % (bench["cpu_baseline"]["value"], bench["cpu_baseline"]["unit"],
                                                                        bench["cpu_baseline"]["cores"], bench["cpu_baseline"]["sample"]))
    print("wrote", out + ".json", out + ".md")


def main():
    src, out = sys.argv[1], sys.argv[2]
    bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
    kernel_sub = bench["roofline"]["kernel"]
    stats_csv = newest(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")))[0]
    shutil.copy(stats_csv, out + "_kernel_stats.csv")
    if " + " in kernel_sub:
        return several_kernels(src, out, bench, kernel_sub.split(" + "), stats_csv)
    kstats = [r for r in csv.DictReader(open(stats_csv)) if kernel_sub in r["Name"]][0]
    c = {}
    launches = 0
    for d in ("fetch", "write", "sq", "tcc"):
        vals, ln = counters(os.path.join(src, d), kernel_sub)
        c.update(vals)
        launches = max(launches, max(ln.values()) if ln else 0)
    fetch_bytes = c.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
    write_bytes = c.get("WRITE_SIZE", 0.0) * 1024.0
    summary = {
        "workload": bench["config"]["workload"],
        "bench_line": bench,
        "rocprof_kernel": kstats["Name"],
        "rocprof_calls": int(kstats["Calls"]),
        "rocprof_avg_launch_ms": float(kstats["AverageNs"]) / 1e6,
        "bench_avg_launch_ms": bench["roofline"]["avg_launch_ms"],
        "pmc_launches_per_pass": launches,
        "hbm_read_bytes_per_launch": fetch_bytes / max(1, launches),
        "hbm_write_bytes_per_launch": write_bytes / max(1, launches),
        "traffic_bytes_per_launch": (fetch_bytes + write_bytes) / max(1, launches),
        "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
        "l2_hit_rate": c.get("TCC_HIT_sum", 0.0) / max(1.0, c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0)),
        "valu_insts_per_launch": c.get("SQ_INSTS_VALU", 0.0) / max(1, launches),
        "salu_insts_per_launch": c.get("SQ_INSTS_SALU", 0.0) / max(1, launches),
        "counters_one_sweep": c,
    }
    cs_per_launch = summary["algorithmic_bytes_per_launch"] / bench["roofline"]["algorithmic_bytes_per_country_step"]
    summary["valu_insts_per_country_step"] = summary["valu_insts_per_launch"] / max(1.0, cs_per_launch)
    summary["salu_insts_per_country_step"] = summary["salu_insts_per_launch"] / max(1.0, cs_per_launch)
    with open(out + ".json", "w") as f:
        json.dump(summary, f, indent=1)
        f.write("\n")
    with open(out + ".md", "w") as f:
        f.write("# rocprofv3 summary: %s\n\n" % summary["workload"])
        f.write("| quantity | value |\n|---|---|\n")
        f.write("| bench value | %.4g %s |\n" % (bench["value"], bench["unit"]))
        f.write("| roofline.achieved (algorithmic) | %.1f GB/s = %.3f of 8 TB/s |\n" % (bench["roofline"]["achieved"], bench["roofline"]["frac"]))
        f.write("| kernel | `%s` |\n" % kstats["Name"])
        f.write("| avg launch, rocprofv3 --kernel-trace --stats | %.2f ms over %s calls |\n" % (summary["rocprof_avg_launch_ms"], kstats["Calls"]))
        f.write("| avg launch, HIP events in bench.py | %.2f ms |\n" % summary["bench_avg_launch_ms"])
        f.write("| HBM read per launch (FETCH_SIZE x 1024 x 2) | %.3e B |\n" % summary["hbm_read_bytes_per_launch"])
        f.write("| HBM write per launch (WRITE_SIZE x 1024) | %.3e B |\n" % summary["hbm_write_bytes_per_launch"])
        f.write("| algorithmic bytes per launch | %.3e B |\n" % summary["algorithmic_bytes_per_launch"])
        f.write("| L2 hit rate (TCC_HIT / (HIT+MISS)) | %.3f |\n" % summary["l2_hit_rate"])
        f.write("| measured HBM rate (read + write) / rocprofv3 avg launch | %.1f GB/s |\n" % (
            summary["traffic_bytes_per_launch"] / (summary["rocprof_avg_launch_ms"] * 1e-3) / 1e9))
        f.write("| VALU / SALU wave instructions per country-step | %.0f / %.0f |\n" % (
            summary["valu_insts_per_country_step"], summary["salu_insts_per_country_step"]))
        f.write("| VALU issue rate: wave instructions / s against 1,024 SIMDs x 2.4 GHz / 4 cycles | %.3g / 6.14e11 = %.2f |\n" % (
            summary["valu_insts_per_launch"] / (summary["rocprof_avg_launch_ms"] * 1e-3),
            summary["valu_insts_per_launch"] / (summary["rocprof_avg_launch_ms"] * 1e-3) / 6.144e11))
        for k in sorted(c):
            f.write("| %s (one sweep) | %.4g |\n" % (k, c[k]))
        if "cpu_baseline" in bench:
            f.write("| cpu_baseline | %.4g %s on %d threads (%s) |\n" % (bench["cpu_baseline"]["value"], bench["cpu_baseline"]["unit"],
                                                                        bench["cpu_baseline"]["cores"], bench["cpu_baseline"]["sample"]))
    print("wrote", out + ".json", out + ".md")


if __name__ == "__main__":
    main()
