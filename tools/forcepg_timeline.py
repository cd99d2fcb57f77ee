#!/usr/bin/env python3
"""Timeline of the collective step on one rank (bench.py --force-pg under rocprofv3 --kernel-trace): for the last steps of
the run, when the render kernels of step i + 1 and the RCCL kernel that carries step i's gather ran, per HIP queue - does
the gather travel beside the next render?      python3 tools/forcepg_timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys

path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
if not path:
    sys.exit("no kernel trace under " + sys.argv[1])
rows = list(csv.DictReader(open(path[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fir = [i for i, r in enumerate(rows) if "bas_render_f" in r["Kernel_Name"]]
if len(fir) < 12:
    sys.exit("too few FIR kernels in the trace")
# (the run ends with `steps` eager render-only steps that time the FIR kernel with HIP events: the timed, overlapped steps
#  lie in front of those - eight of them from the middle of the timed region)
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
first, last = fir[-n_steps - n_steps // 2 - 9], fir[-n_steps - n_steps // 2 - 1]
t0 = int(rows[first]["Start_Timestamp"])
print("# kernels of eight timed steps of bench.py --sources 32 --force-pg (one rank, RCCL communicator of size 1), us from the")
print("# first FIR kernel shown; queue = the HIP stream's hardware queue.  The gather of step i is launched behind the render of")
print("# step i on RCCL's stream; the root sums it (bas_mix_finish_kernel) behind the render of step i + 1.")
print(f"{'start':>9s} {'end':>9s} {'dur':>7s}  queue  kernel")
overlap_ns, gather_ns = 0, 0
firs = [(int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])) for i in fir]
for r in rows[first:last + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
    q = r.get("Queue_Id", r.get("Queue_ID", "?"))
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {q:>5s}  {name}")
    if "nccl" in r["Kernel_Name"].lower() or "rccl" in r["Kernel_Name"].lower():
        gather_ns += e - s
        for fs, fe in firs:
            overlap_ns += max(0, min(e, fe) - max(s, fs))
if gather_ns:
    print(f"# RCCL kernels: {gather_ns / 1e3:.1f} us in these steps, {overlap_ns / 1e3:.1f} us of it ({100.0 * overlap_ns / gather_ns:.0f} %) while a FIR kernel was running")
else:
    print("# no RCCL kernel in the trace (a communicator of size 1 may copy with a blit kernel or the DMA engines)")
