"""per-step table from a rocprofv3 kernel_stats.csv of a bench.py run: usage kstats.py FILE STEPS [TOP]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print(f"kernel time {tot / 1e6 / steps:.2f} ms/step, {calls / steps:.0f} launches/step")
import re
for r in rows[:top]:
    n = re.sub(r"\(anonymous namespace\)::|_GLOBAL__N_1|void |at::native::", "", r['Name'])
    print(f"{n[:86]:86s} {int(r['Calls']) / steps:7.1f} {float(r['TotalDurationNs']) / 1e6 / steps:7.3f} ms {float(r['AverageNs']) / 1e3:8.1f} us")
