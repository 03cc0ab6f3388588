"""Per-step kernel table from a rocprofv3 kernel_stats.csv of bench.py: calls per step, average time, time per step.
usage: python tools/kernel_table.py <kernel_stats.csv> [steps profiled]      (default: the number of k_sgd_step launches)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else float(sum(int(r["Calls"]) for r in rows if "k_sgd_step" in r["Name"]) or 1)
tot = 0.0
out = []
for r in rows:
    calls, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e3
    if calls < steps * 0.5:
        continue
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    out.append((calls * avg / steps, name[:64], calls / steps, avg))
    tot += calls * avg / steps
for per, name, cps, avg in sorted(out, reverse=True):
    print("%-64s %5.1f/step  avg %7.2f us  %7.1f us/step" % (name, cps, avg, per))
print("sum of kernel time per step: %.1f us" % tot)
