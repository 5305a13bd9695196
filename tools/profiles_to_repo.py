"""Turns the scratch outputs of tools/collect_profiles.sh (gpurun_out/<round>/cfg<config>/) into the
committed evidence under profiles/: the bench line, rocprofv3 kernel stats, per-kernel PMC means and
pmc_<round>_cfg<config>.json with the HBM traffic per launch (FETCH_SIZE doubled: the gfx950
correction of MI355X_MICROARCH.md, 'HBM' section; FETCH/WRITE collected in separate passes).

    python tools/profiles_to_repo.py <round> <config>
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
config = sys.argv[2] if len(sys.argv) > 2 else "3"
src = os.path.join(ROOT, "gpurun_out", rnd, "cfg" + config)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
tag = "%s_cfg%s" % (rnd, config)


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]  # gpurun merges into gpurun_out/: older runs' files may still be there


def counters(path):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": len(next(iter(cs.values())))} for k, cs in acc.items()}


if os.path.exists(os.path.join(src, "bench.json")) and os.path.getsize(os.path.join(src, "bench.json")):
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "%s_bench.json" % tag))
avg_ms = {}
for f in newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, "%s_rocprofv3_kernel_stats.csv" % tag))
    for r in csv.DictReader(open(f)):
        avg_ms[r["Name"].split("(")[0]] = float(r["AverageNs"]) / 1e6
for f in newest(os.path.join(src, "stats", "*", "*_kernel_trace.csv")):
    with open(f) as fh, open(os.path.join(dst, "%s_rocprofv3_kernel_trace_head.csv" % tag), "w") as out:
        for i, line in enumerate(fh):
            if i < 8:
                out.write(line)

summary = {}
valu = newest(os.path.join(src, "pmc_valu", "*", "*_counter_collection.csv"))
if valu:
    for k, m in counters(valu[0]).items():
        if "sdfr::" not in k:
            continue
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        summary[k] = {
            "dispatches": m["dispatches"], "SQ_INSTS_VALU": m["SQ_INSTS_VALU"], "SQ_INSTS_SALU": m["SQ_INSTS_SALU"], "SQ_WAVES": m["SQ_WAVES"],
            "gpu_cycles_per_xcd": cyc, "ms_at_2.4GHz": cyc / 2.4e6,
            "valu_lane_utilization": m["SQ_THREAD_CYCLES_VALU"] / (m["SQ_ACTIVE_INST_VALU"] * 64.0),
            "avg_waves_per_simd": m["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * cyc),
            "cycles_per_valu_inst_per_simd": cyc * 1024.0 / m["SQ_INSTS_VALU"],
            "avg_ms": avg_ms.get(k),  # rocprofv3 --kernel-trace --stats average of the same command (no counters)
        }
fetch = newest(os.path.join(src, "pmc_fetch", "*", "*_counter_collection.csv"))
write = newest(os.path.join(src, "pmc_write", "*", "*_counter_collection.csv"))
traffic = {}
if fetch and write:
    fm, wm = counters(fetch[0]), counters(write[0])
    for k in fm:
        if "sdfr::" in k and k in wm:
            f_kb, w_kb = fm[k]["FETCH_SIZE"], wm[k]["WRITE_SIZE"]
            traffic[k] = {"FETCH_SIZE_KB_raw": f_kb, "WRITE_SIZE_KB": w_kb, "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0}
if summary:
    import bench

    cfg = bench.CONFIGS[config]
    main = [k for k in traffic if "k_pixel" in k]
    try:
        commit = open(os.path.join(src, "commit.txt")).read().strip()
    except Exception:
        commit = None
    out = {"round": rnd, "commit": commit,
           "workload": {"config": config, "key": cfg["key"], "width": cfg["width"], "height": cfg["height"], "schedule": "pixel"},
           "command": "bench.py --config %s --steps 16 --warmup 2 --no-cpu-baseline --no-second-pass --no-extra-passes" % config,
           "kernels": summary, "traffic": traffic,
           "hbm_bytes_per_launch": traffic[main[0]]["hbm_bytes_per_launch"] if main else None,
           "note": "PMC means per dispatch; traffic = (2*FETCH_SIZE + WRITE_SIZE) KB (separate passes, gfx950 FETCH_SIZE correction)"}
    with open(os.path.join(dst, "pmc_%s.json" % tag), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1)[:3000])
