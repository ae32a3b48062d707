"""Per-step GPU timeline of the one-rank protocol run from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace -d DIR -o t -- python3 tools/multi_one_rank.py morton
    python tools/multi_trace.py DIR
Prints, for the steady-state steps of the LAST configuration run, the kernels of one step in start order with
their durations and the gap to the previous kernel's end (any stream)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70], r.get("Stream_Id", "")))
rows.sort()
# one k_splitters launch per step
starts = [i for i, r in enumerate(rows) if "k_splitters" in r[2]]
if len(starts) < 4:
    print("kernels seen:", sorted(set(r[2] for r in rows)))
    sys.exit(1)
a, b = starts[-3], starts[-2]
t0 = rows[a][0]
prev_end = t0
print("step wall (k_splitters to the next k_splitters): %.1f us" % ((rows[b][0] - t0) / 1e3))
busy = 0
for s, e, name, stream in rows[a:b]:
    print("%8.1f  +%6.1f gap  %6.1f us  s%-3s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, stream, name))
    prev_end = max(prev_end, e)
    busy += e - s
print("sum of kernel durations %.1f us" % (busy / 1e3))
