"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter
per dispatch, plus derived VALU utilisation figures."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-60:]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    n = len(next(iter(cs.values())))
    print("%s  [%d dispatches]" % (name, n))
    for k, v in sorted(m.items()):
        print("   %-26s %.6g" % (k, v))
    if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m and m["SQ_ACTIVE_INST_VALU"]:
        print("   %-26s %.2f %%  (avg active lanes per VALU instruction / 64)" % ("VALUUtilization", 100 * m["SQ_THREAD_CYCLES_VALU"] / (m["SQ_ACTIVE_INST_VALU"] * 64)))
    if "GRBM_GUI_ACTIVE" in m and "SQ_ACTIVE_INST_VALU" in m and m["GRBM_GUI_ACTIVE"]:
        print("   %-26s %.2f %%  (SQ_ACTIVE_INST_VALU / CUs / GRBM_GUI_ACTIVE, gfx94x formula)" % ("VALUBusy", 100 * m["SQ_ACTIVE_INST_VALU"] / 256 / m["GRBM_GUI_ACTIVE"]))
