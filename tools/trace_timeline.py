"""Timeline of steady-state steps from a rocprofv3 --kernel-trace csv: per kernel start offset / duration / queue,
per-queue busy time and the time no kernel at all was running.
usage: trace_timeline.py <kernel_trace.csv> [marker-substring=token_weights_kernel] [steps=2] [skip_from_end=2]"""
import csv, sys, collections

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "token_weights_kernel"
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 2
rows = []
if path.endswith(".db"):            # rocprofv3's default output (rocpd SQLite database)
    import sqlite3
    cur = sqlite3.connect(path).cursor()
    for r in cur.execute("select start, end, queue_id, name, grid_x * grid_y * grid_z, workgroup_x * workgroup_y * "
                         "workgroup_z, vgpr_count, lds_size from kernels"):
        rows.append(tuple(r))
else:
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"],
                     int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]),
                     int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]),
                     int(r["VGPR_Count"]), int(r["LDS_Block_Size"])))
rows.sort()
marks = [s for s, e, q, k, g, wg, vg, lds in rows if marker in k]
if len(marks) < nsteps + skip + 1:
    sys.exit(f"only {len(marks)} marker kernels")
t0, t1 = marks[-(nsteps + skip + 1)], marks[-(skip + 1)]
print(f"window: {nsteps} steps, {(t1 - t0) / 1e3 / nsteps:.1f} us per step")
sel = [r for r in rows if t0 <= r[0] < t1]


def short(name):
    name = name.replace("void ", "").replace("basd::", "")
    cut = name.find("(")
    return (name[:cut] if cut > 0 else name)[:60]


queues = sorted({r[2] for r in sel})
qname = {q: i for i, q in enumerate(queues)}
for s, e, q, k, g, wg, vg, lds in sel:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{qname[q]} wgs={g // max(1, wg):5d}x{wg:4d} v{vg:3d} lds{lds // 1024:3d}K {short(k)}")
busy = collections.defaultdict(int)
for s, e, q, k, g, wg, vg, lds in sel:
    busy[q] += e - s
ev = sorted([(s, 1) for s, e, *_ in sel] + [(e, -1) for s, e, *_ in sel])
depth, last, idle, conc = 0, t0, 0, 0
for t, dlt in ev:
    if depth == 0:
        idle += t - last
    conc += depth * (t - last)
    depth += dlt
    last = t
print("per-queue kernel time per step (us):", {f"q{qname[q]}": round(v / 1e3 / nsteps, 1) for q, v in busy.items()})
print(f"no kernel running: {idle / 1e3 / nsteps:.1f} us per step; mean kernels in flight {conc / max(1, t1 - t0):.2f}")
agg = collections.defaultdict(lambda: [0, 0])
for s, e, q, k, g, wg, vg, lds in sel:
    agg[short(k)][0] += e - s
    agg[short(k)][1] += 1
print("kernel time per step (us), top 25:")
for k, (v, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {v / 1e3 / nsteps:8.1f}  x{c / nsteps:4.1f}  {k}")
