#!/usr/bin/env python3
"""Condenses the raw rocprofv3 output of profiles/collect_pmc.sh (gpurun_out/<tag>/) into
profiles/<out>.json (read by bench.py, keyed by the sha256 of csrc/trace_kernel.hpp the passes ran on)
and profiles/<out>_kernel_stats.csv (the --stats summary).

    python profiles/make_pmc_json.py gpurun_out/pmc_r3_bench r03_pmc_trace
"""
import collections, csv, glob, hashlib, json, os, subprocess, sys

src, out = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha_box = open(os.path.join(src, "trace_kernel.sha256")).read().strip()
sha_here = hashlib.sha256(open(os.path.join(ROOT, "grace-devel_amd", "csrc", "trace_kernel.hpp"), "rb").read()).hexdigest()
if sha_box != sha_here:
    print("WARNING: csrc/trace_kernel.hpp changed since the passes ran (%s vs %s)" % (sha_box[:12], sha_here[:12]))


def short(name):
    # 'void (anonymous namespace)::trace_kernel<1, false, true>((anonymous namespace)::TraceArgs)'
    i = name.find("trace_kernel<")
    return name[i: name.find(">", i) + 1] if i >= 0 else None


kernels = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
    per_dispatch = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            per_dispatch[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), cs in per_dispatch.items():
        for c, v in cs.items():
            kernels[k][c].append(v)
durations = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            durations[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
res = {}
for k, cs in kernels.items():
    e = {c: sum(v) / len(v) for c, v in cs.items()}
    e["dispatches_per_pass"] = len(next(iter(cs.values())))
    if "FETCH_SIZE" in e: e["FETCH_SIZE_KB"] = e.pop("FETCH_SIZE")
    if "WRITE_SIZE" in e: e["WRITE_SIZE_KB"] = e.pop("WRITE_SIZE")
    if durations.get(k):
        d = sorted(durations[k])
        e["kernel_trace_ms_avg"] = sum(d) / len(d); e["kernel_trace_ms_min"] = d[0]; e["kernel_trace_calls"] = len(d)
    res[k] = e
try:
    head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    head = None
doc = {"what": "rocprofv3 PMC passes of `python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline` on one MI355X, "
               "per dispatch averages; recipe profiles/collect_pmc.sh; units: SQ_*CYCLES / SQ_ACTIVE_* / SQ_WAIT_* in "
               "quad-cycles summed over waves or SIMDs, GRBM_GUI_ACTIVE summed over the 8 XCDs, *_SIZE_KB in KiB "
               "(FETCH_SIZE is x2 low on gfx950 for wide loads: corrected by the reader, not here)",
       "trace_kernel_sha256": sha_box, "collected_at_commit": head, "kernels": res}
json.dump(doc, open(os.path.join(ROOT, "profiles", out + ".json"), "w"), indent=1, sort_keys=True)
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    open(os.path.join(ROOT, "profiles", out + "_kernel_stats.csv"), "w").write(open(f).read())
for k, e in sorted(res.items()):
    print(k, {c: ("%.4g" % v if isinstance(v, float) else v) for c, v in sorted(e.items())})
