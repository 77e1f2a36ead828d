#!/usr/bin/env python3
"""Round-3 archiver: copy the rocprofv3 summaries of gpurun_out/prof_<tag> (tools/prof3.sh) into profiles/ and derive
profiles/<round>_hbm_traffic.json / <round>_valu_issue.json as tools/archive_profile.py does, plus

  profiles/<round>_<tag>_timed_region.json   the per-launch durations of the timed kernel taken from the kernel TRACE of the
        driver's exact command (python3 bench.py --gpus 1 --steps 20 --warmup 5): launches 1-5 are the warm-up, 6-25 the timed
        region, the rest belong to the passes bench.py runs after it (host-buffer pass: partial frames; census; ...).  The
        average over launches 6-25 must reproduce the line's ms_per_step (the --stats average mixes all of them).

usage: tools/archive_profile3.py <tag> <round> <workload> [kernel-name-substring]"""
import csv, glob, json, os, subprocess, sys
tag, rnd, workload = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "c3")
want = sys.argv[4] if len(sys.argv) > 4 else None
src = f"gpurun_out/prof_{tag}"
subprocess.run([sys.executable, "tools/archive_profile.py", tag, rnd, workload] + ([want] if want else []), check=True)
trace = sorted(glob.glob(f"{src}/trace/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)   # newest run (see archive_profile.py)
rows = list(csv.DictReader(open(trace[-1])))
names = {}
for r in rows:
    if "rt_tile_kernel" in r["Kernel_Name"] and (want is None or want in r["Kernel_Name"]):
        names[r["Kernel_Name"]] = names.get(r["Kernel_Name"], 0) + 1
kern = max(names, key=names.get)
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if r["Kernel_Name"] == kern]
line = None
for ln in open(f"{src}/trace.log"):
    if ln.startswith("{") and '"metric"' in ln:
        line = json.loads(ln)
cmd = open(f"{src}/cmd.txt").read().strip()
out = {"command": cmd, "kernel": kern, "launches_in_trace": len(d), "durations_ms": [round(x, 4) for x in d],
       "warmup_launches": d[:5], "timed_launches_6_to_25_avg_ms": sum(d[5:25]) / max(len(d[5:25]), 1),
       "bench_line_under_the_profiler": None if line is None else
       {"ms_per_step": line["ms_per_step"], "value": line["value"], "roofline_avg_launch_ms": line["roofline"]["avg_launch_ms"]},
       "stats_average_all_launches_ms": sum(d) / len(d)}
if line is not None:
    out["trace_vs_line_ratio"] = out["timed_launches_6_to_25_avg_ms"] / line["ms_per_step"]
json.dump(out, open(f"profiles/{rnd}_{tag}_timed_region.json", "w"), indent=1)
for name in (f"profiles/{rnd}_hbm_traffic.json", f"profiles/{rnd}_valu_issue.json"):
    if os.path.exists(name):
        dd = json.load(open(name))
        if isinstance(dd.get(workload), dict):
            dd[workload]["avg_launch_ms_timed_region"] = out["timed_launches_6_to_25_avg_ms"]
            if "avg_launch_ms_in_trace" in dd[workload]:
                # the --stats average runs over EVERY launch of that kernel name in the driver's command (the host-buffer pass
                # launches partial frames, other_workloads renders c4 with the same kernel): not a per-frame figure
                dd[workload]["stats_average_all_launches_of_that_kernel_ms"] = dd[workload].pop("avg_launch_ms_in_trace")
            dd[workload]["timed_region"] = f"profiles/{rnd}_{tag}_timed_region.json"
            json.dump(dd, open(name, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "durations_ms"}, indent=1))
