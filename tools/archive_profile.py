#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of gpurun_out/prof_<tag> into profiles/ (tracked) and derive, per workload,
profiles/<round>_hbm_traffic.json and profiles/<round>_valu_issue.json, which bench.py reads into its roofline object
(marked there as committed profiles, not live measurements).

  HBM traffic   FETCH_SIZE x 2 + WRITE_SIZE (KiB counters; gfx950 tallies 128-byte read requests at 64 B:
                MI355X_MICROARCH.md, HBM / rocprofv3 section), per launch of the timed tile kernel
  VALU issue    SQ_INSTS_VALU x 2 cycles (a wave64 instruction occupies a SIMD-32 for 2 cycles, MI355X_MICROARCH.md:53-54)
                over SIMDs x kernel cycles (SQ_BUSY_CYCLES / 32 shader engines); lanes = SQ_THREAD_CYCLES_VALU /
                SQ_INSTS_VALU; lane-weighted = issue x lanes / 64

usage: tools/archive_profile.py <tag> <round> <workload> [kernel-name-substring]"""
import csv, glob, json, os, shutil, subprocess, sys
tag, rnd, workload = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "c3")
want = sys.argv[4] if len(sys.argv) > 4 else None
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
# gpurun MERGES a call's gpurun_out/ into the local one: files of earlier profile runs of the same tag (other process ids in
# their names) stay beside the new ones.  Only the newest run counts: everything more than an hour older than the newest
# file of the directory is ignored (round 3 nearly archived a mixture).
_all = [f for f in glob.glob(f"{src}/**/*.csv", recursive=True)]
_newest = max(os.path.getmtime(f) for f in _all)
_stale = {f for f in _all if os.path.getmtime(f) < _newest - 3600}
_glob = glob.glob
glob.glob = lambda pat, **kw: [f for f in _glob(pat, **kw) if f not in _stale]
stats = None
for f in glob.glob(f"{src}/trace/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, f"profiles/{rnd}_{tag}_kernel_stats.csv")
    stats = f
txt = subprocess.run([sys.executable, "tools/pmc_summary.py", src], capture_output=True, text=True).stdout
cmd = open(f"{src}/cmd.txt").read().strip() if os.path.exists(f"{src}/cmd.txt") else "python3 bench.py --steps 2 --warmup 1 ..."
open(f"profiles/{rnd}_{tag}_rocprof_summary.txt", "w").write(
    f"# rocprofv3 --kernel-trace --stats and --pmc passes of: {cmd}\n"
    f"# (separate passes per counter group, tools/prof2.sh; SQ_* cycle counters are quad-cycles)\n" + txt)


def timed_kernel():
    """Name of the timed tile kernel = the rt_tile_kernel instantiation with the most calls (the counting twin and
    the linear-engine launches of bench.py run once or thrice)."""
    best, calls = None, -1
    for row in csv.DictReader(open(stats)):
        n = row["Name"]
        if "rt_tile_kernel" in n and (want is None or want in n) and int(row["Calls"]) > calls:
            best, calls = n, int(row["Calls"])
    return best


kern = timed_kernel()
avg_ns = None
for row in csv.DictReader(open(stats)):
    if row["Name"] == kern:
        avg_ns = float(row["AverageNs"])


def per_dispatch(name):
    tot, n = 0.0, 0
    for f in glob.glob(f"{src}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == name and row["Kernel_Name"] == kern:
                tot += float(row["Counter_Value"]); n += 1
    return tot / n if n else None


fetch, write = per_dispatch("FETCH_SIZE"), per_dispatch("WRITE_SIZE")
p = f"profiles/{rnd}_hbm_traffic.json"
d = json.load(open(p)) if os.path.exists(p) else {}
if fetch is not None and write is not None:
    d[workload] = {"bytes_per_launch": (2 * fetch + write) * 1024, "fetch_size_kib": fetch, "write_size_kib": write,
                   "kernel": kern, "avg_launch_ms_in_trace": avg_ns / 1e6 if avg_ns else None,
                   "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE as reported",
                   "profile": f"profiles/{rnd}_{tag}_rocprof_summary.txt"}
json.dump(d, open(p, "w"), indent=1)
insts, busy = per_dispatch("SQ_INSTS_VALU"), per_dispatch("SQ_BUSY_CYCLES")
thr, wait, wcyc = per_dispatch("SQ_THREAD_CYCLES_VALU"), per_dispatch("SQ_WAIT_ANY"), per_dispatch("SQ_WAVE_CYCLES")
lds_act, lds_conf = per_dispatch("SQ_LDS_IDX_ACTIVE"), per_dispatch("SQ_LDS_BANK_CONFLICT")
if insts and busy:
    pi = f"profiles/{rnd}_valu_issue.json"
    di = json.load(open(pi)) if os.path.exists(pi) else {}
    cycles = busy / 32.0
    lanes = (thr / insts) if thr else None
    issue = insts * 2.0 / (1024.0 * cycles)
    di[workload] = {"kernel": kern, "valu_insts_per_launch": insts, "kernel_cycles": cycles, "simds": 1024,
                    "issue_frac_2cycle": issue, "active_lanes_per_valu_inst": lanes,
                    "lane_weighted_frac": issue * lanes / 64.0 if lanes else None,
                    "wait_any_over_wave_cycles": (wait / wcyc) if wait and wcyc else None,
                    "lds_busy_frac": (lds_act / 256.0 / cycles) if lds_act else None,
                    "lds_conflict_share_of_lds_cycles": (lds_conf / lds_act) if lds_act and lds_conf else None,
                    "note": "issue_frac_2cycle = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x SQ_BUSY_CYCLES / 32 SEs): every wave64 VALU "
                            "instruction priced at the SIMD-32's 2 cycles (MI355X_MICROARCH.md:53-54) — a LOWER bound on pipe "
                            "occupancy: tools/ubench/valu_classes measures 2 cycles only for fma / mul / add / mov / logic and "
                            "about twice that for min / max / cmp / cndmask / shifts / cvt / integer multiply.  lane_weighted_frac "
                            "= issue x active lanes / 64.",
                    "profile": f"profiles/{rnd}_{tag}_rocprof_summary.txt"}
    # round 4: the profiler's own derived busy / utilisation metrics (their own --pmc passes, tools/prof4.sh), where collected
    derived = {k: per_dispatch(k) for k in ("VALUBusy", "VALUUtilization", "SALUBusy", "LDSBankConflict", "MemUnitStalled")}
    if any(v is not None for v in derived.values()):
        di[workload]["derived_metrics_percent"] = derived
        di[workload]["derived_metrics_note"] = (
            "rocprofv3 derived metrics (gfx94x formulas: ROCm 7.2 ships no gfx950 section).  VALUBusy = 100 x SQ_ACTIVE_INST_VALU "
            "[quad-cycles a wave spends on VALU instructions, summed over waves] / CUs / cycles, i.e. per SIMD the sum of its waves' "
            "VALU windows over the elapsed time: it exceeds 100 when the windows of the four waves of a SIMD overlap — there is no "
            "idle vector-pipe time to find.  VALUUtilization = active lanes / 64 of the VALU instructions.  SALUBusy likewise for "
            "scalar instructions; LDSBankConflict = conflict cycles / elapsed cycles per CU; MemUnitStalled = TA data stall share.")
    json.dump(di, open(pi, "w"), indent=1)
    print(json.dumps(di[workload], indent=1))
print(json.dumps(d.get(workload), indent=1))
