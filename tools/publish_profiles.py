#!/usr/bin/env python3
"""Copies the measurements tools/collect_profiles.sh left under gpurun_out/r02p/ into profiles/ (tracked):
summaries as they are, counter CSVs reduced to this library's kernels, and profiles/fir_hbm_traffic.json
recomputed from the FETCH_SIZE / WRITE_SIZE passes (what bench.py replays as roofline.traffic)."""
import collections, csv, io, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r02p")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[2] if len(sys.argv) > 2 else "r02_"


def copy(rel, name):
    p = os.path.join(SRC, rel)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copyfile(p, os.path.join(DST, TAG + name))
        return True
    print("missing:", rel)
    return False


def filter_counters(rel, name, write=True):
    p = os.path.join(SRC, rel)
    if not os.path.exists(p):
        print("missing:", rel)
        return {}
    rows = list(csv.DictReader(open(p)))
    keep = [r for r in rows if "bas_" in r["Kernel_Name"]]
    if write:                                                 # (the 25 k-line SQ counter CSVs are summarised, not kept)
        with open(os.path.join(DST, TAG + name), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(keep)
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in keep:
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return per


for rel, name in [("bench_default.json", "bench.json"), ("bench_300.json", "bench_300steps.json"),
                  ("bench_profiled.json", "bench_profiled.json"), ("prof/bench_kernel_stats.csv", "kernel_stats.csv"),
                  ("bench_unfused.json", "bench_unfused.json"), ("prof_unfused/bench_kernel_stats.csv", "kernel_stats_unfused.csv"),
                  ("single_source.json", "single_source.json"), ("prof_single/bench_kernel_stats.csv", "kernel_stats_single_source.csv"),
                  ("single_source_tile8192.json", "single_source_tile8192.json"), ("single_source_latency.txt", "single_source_latency.txt"),
                  ("stream_1024src_48k.json", "stream_1024src_48k.json"), ("stream_1024src_48k_regen.json", "stream_1024src_48k_regen.json"),
                  ("stream_host_time.txt", "stream_host_time.txt"), ("ubench_fir_pattern.txt", "ubench_fir_pattern.txt"),
                  ("ubench_fma_forms.txt", "ubench_fma_forms.txt"), ("ubench_fir_steps.txt", "ubench_fir_steps.txt"),
                  ("stamps_fz_256.txt", "stamps_fz_256sources.txt"), ("warmup_series.txt", "warmup_series.txt"), ("bench_2ranks_one_device.json", "bench_2ranks_one_device.json"),
                  ("stream_hour.json", "stream_hour.json"), ("stream_hour_regen.json", "stream_hour_regen.json"),
                  ("prof_stream/bench_kernel_stats.csv", "kernel_stats_stream.csv"), ("stream_forcepg.json", "stream_forcepg_nccl_world1.json"),
                  ("prof_share32/bench_kernel_stats.csv", "kernel_stats_32sources.csv"),
                  ("ab_noeval.txt", "ab_no_chunk_ir_evaluation.txt"), ("ab_ir_sharing.txt", "ab_ir_sharing.txt"),
                  ("ab_split_roles.txt", "ab_split_roles.txt"), ("stamps_fs_256.txt", "stamps_fs_256sources.txt"),
                  ("ubench_lone_wave.txt", "ubench_lone_wave.txt")]:
    copy(rel, name)

# per-rank shares: one table
rows = []
for n in (32, 64, 128, 256):
    row = [f"{n:4d} sources"]
    for kind in ("plain", "graph", "forcepg"):
        pth = os.path.join(SRC, f"share_{n}_{kind}.json")
        try:
            d = json.loads([ln for ln in open(pth) if ln.lstrip().startswith("{")][0])
            row.append(f"{kind} {d['ms_per_step'] * 1e3:7.1f} us/step (FIR kernel {d['roofline']['kernel_ms'] * 1e3:6.1f} us)")
        except Exception as e:                                # noqa: BLE001
            row.append(f"{kind} missing ({e})")
    rows.append(" | ".join(row))
if rows:
    with open(os.path.join(DST, TAG + "per_rank_shares.txt"), "w") as f:
        f.write("# bench.py --sources N --steps 200 --warmup 10 on ONE GPU: the share of the 256-source scene a rank renders at\n"
                "# 8 / 4 / 2 / 1 GPUs.  plain = eager launches incl. the peak rule; graph = the same step replayed as one hipGraph;\n"
                "# forcepg = the N > 1 code path on one rank with a real RCCL communicator (world_size 1): graph-replayed render, async\n"
                "# gather of the 3.5 MB partial mix to itself, fixed-order sum + peak rule on the root, steps overlapped.\n")
        f.write("\n".join(rows) + "\n")

fetch = filter_counters("pmc_FETCH_SIZE/pmc_counter_collection.csv", "pmc_FETCH_SIZE.csv")
write = filter_counters("pmc_WRITE_SIZE/pmc_counter_collection.csv", "pmc_WRITE_SIZE.csv")

for sq in ("pmc_SQ", "pmc_SQ2"):
    p = os.path.join(SRC, sq, "pmc_counter_collection.csv")
    if os.path.exists(p):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sq_summary.py"), p, "bas_render_f"],     # (fz: two workgroups per CU; fs: split roles)
                             capture_output=True, text=True).stdout
        open(os.path.join(DST, TAG + sq + "_fz_kernel_summary.txt"), "w").write(out)

def is_fir(k):
    return "bas_render_fz_kernel" in k or "bas_render_fs_kernel" in k


fz = [k for k in fetch if is_fir(k)]
if fz and any(is_fir(k) for k in write):
    kf = fz[0]
    kw = [k for k in write if is_fir(k)][0]
    f_kb = sum(fetch[kf]["FETCH_SIZE"]) / len(fetch[kf]["FETCH_SIZE"])
    w_kb = sum(write[kw]["WRITE_SIZE"]) / len(write[kw]["WRITE_SIZE"])
    rec = {"workload": "256x441000@K512S32L128", "fused": True, "kernel": kf.split("(")[0].replace("void ", ""),
           "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/" + TAG + "pmc_FETCH_SIZE.csv, "
                     + TAG + "pmc_WRITE_SIZE.csv), mean over the dispatches of python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline",
           "FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb,
           "correction": "MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced "
                         "streaming read (the x window, 459 MB, is read 16 B/lane) -> doubled; WRITE_SIZE is exact for 16-B/lane "
                         "stores.  The table gathers (unaligned 16-B buffer loads, mostly L2 hits) and the plan reads are "
                         "uncalibrated: doubling them too makes this an upper bound.",
           "bytes_per_launch": int(2 * f_kb * 1024 + w_kb * 1024),
           "lower_bound_bytes_if_only_x_is_doubled": int((f_kb * 1024 + 229.5e6) + w_kb * 1024)}
    json.dump(rec, open(os.path.join(DST, "fir_hbm_traffic.json"), "w"), indent=1)
    print("traffic MB:", rec["bytes_per_launch"] / 1e6)
print(sorted(os.listdir(DST)))
