"""Summarise a rocprofv3 kernel trace: per-step kernel time, idle gaps, top kernels.
usage: python tools/trace_step.py <dir with *_kernel_trace.csv> [n_steps_in_run]"""
import csv, glob, re, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are delimited by the optimizer's kernel (last kernel of a step): use the k_site_insert / subm plan as the start marker
starts = [i for i, r in enumerate(rows) if "k_load_batch" in r["Kernel_Name"]]
if len(starts) < 4:
    starts = [i for i, r in enumerate(rows) if "k_site_insert" in r["Kernel_Name"]]
if len(starts) < 4:
    print("not enough steps"); sys.exit()
a, b = starts[-3], starts[-2]            # one full steady-state step
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"]); t1 = int(rows[b]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
print("step wall %.1f us, kernel busy %.1f us, idle %.1f us, %d kernels" % ((t1 - t0) / 1e3, busy / 1e3, (t1 - t0 - busy) / 1e3, len(step)))
prev_end = t0
agg = {}
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"])[:58]
    gap = (s - prev_end) / 1e3
    if "-v" in sys.argv:
        print("%8.1f  +gap %6.1f  dur %6.1f  end %7.1f  q%s  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, (e - t0) / 1e3,
                                                                  r.get("Queue_Id", "?"), name))
    k = agg.setdefault(name, [0, 0.0]); k[0] += 1; k[1] += (e - s) / 1e3
    prev_end = max(prev_end, e)
for name, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print("%7.1f us  x%-3d %s" % (us, n, name))
