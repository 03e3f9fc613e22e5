"""Analyse a rocprofv3 --kernel-trace CSV of bench.py: print the timeline (start offset, duration, queue, kernel) of one
training step (delimited by the Gibbs launches) with its idle time and time under >= 2 concurrent kernels.
    python tools/trace_step.py <kernel_trace.csv> [step index]"""
import csv, sys, glob, collections
path = sys.argv[1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
# step boundary: gibbs kernel launches (one per step)
gib = [i for i, r in enumerate(rows) if "gibbs" in r[2]]
print("kernels", len(rows), "gibbs launches", len(gib))
k = int(sys.argv[2]) if len(sys.argv) > 2 else -3
a, b = gib[k], gib[k + 1]
step = rows[a:b]
t0 = step[0][0]
print("step span us", (step[-1][1] - t0) / 1e3, "kernels", len(step))
busy = 0
cur_end = t0
events = []
for s, e, n, q, st in step:
    events.append((s, 1)); events.append((e, -1))
events.sort()
depth = 0; last = t0; idle = 0; par = 0
for t, d in events:
    if depth == 0: idle += t - last
    elif depth >= 2: par += t - last
    last = t; depth += d
print("idle us", idle / 1e3, "time with >=2 kernels us", par / 1e3)
short = lambda n: n.split("(")[0].replace("void dvg::", "").replace("dvg::", "")[:60]
for s, e, n, q, st in step:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q} {short(n)}")
