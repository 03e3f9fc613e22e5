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
# the convolution GEMM kernels (conv_igemm / conv_wino8 / conv_wino_wgrad8 / conv_wgrad*): the SUM of their durations (what
# bench.py's conv_all divides the executed FLOPs by: two kernels side by side count twice) and the UNION of their
# intervals (wall time during which at least one runs: executed FLOPs / this = what the chip delivers while they run)
gemm = sorted((s_, e_) for s_, e_, n_, q_, st_ in step if "conv_igemm" in n_ or "conv_wino" in n_ or "conv_wgrad" in n_)
tot = sum(e_ - s_ for s_, e_ in gemm); uni = 0; cur_s = cur_e = None
for s_, e_ in gemm:
    if cur_e is None or s_ > cur_e:
        if cur_e is not None: uni += cur_e - cur_s
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
if cur_e is not None: uni += cur_e - cur_s
print("convolution GEMM kernels:", len(gemm), "launches, sum of durations us", tot / 1e3, "union of their intervals us", uni / 1e3)
short = lambda n: n.split("(")[0].replace("void dvg::", "").replace("dvg::", "")[:60]
for s, e, n, q, st in step:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q} {short(n)}")
