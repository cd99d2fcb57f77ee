#!/usr/bin/env python3
"""Copies the measurements tools/collect_profiles_r04.sh left under gpurun_out/r04p/ into profiles/ (tracked) as r04_*:
summaries as they are, counter CSVs reduced to this library's kernels, profiles/fir_hbm_traffic.json recomputed from the
FETCH_SIZE / WRITE_SIZE passes with the traffic of every kernel of the step beside the FIR kernel's and its parts named."""
import collections, csv, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r04p")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[2] if len(sys.argv) > 2 else "r04_"


def copy(rel, name):
    p = os.path.join(SRC, rel)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copyfile(p, os.path.join(DST, TAG + name))
        return True
    print("missing:", rel)
    return False


def counters(rel, name):
    p = os.path.join(SRC, rel)
    if not os.path.exists(p):
        print("missing:", rel)
        return {}
    rows = list(csv.DictReader(open(p)))
    keep = [r for r in rows if "bas_" in r["Kernel_Name"]]
    with open(os.path.join(DST, TAG + name), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in keep:
        per[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return per


for rel, name in [("bench_default.json", "bench.json"), ("bench_300.json", "bench_300steps.json"), ("bench_taps100.json", "bench_taps100.json"),
                  ("bench_s16.json", "bench_subchunk16.json"), ("bench_s16_unfused.json", "bench_subchunk16_unfused.json"),
                  ("bench_s8.json", "bench_subchunk8.json"), ("bench_s8_unfused.json", "bench_subchunk8_unfused.json"),
                  ("bench_profiled.json", "bench_profiled.json"), ("prof/bench_kernel_stats.csv", "kernel_stats.csv"),
                  ("bench_unfused.json", "bench_unfused.json"), ("prof_unfused/bench_kernel_stats.csv", "kernel_stats_unfused.csv"),
                  ("single_source.json", "single_source.json"), ("prof_single/bench_kernel_stats.csv", "kernel_stats_single_source.csv"),
                  ("single_source_latency.txt", "single_source_latency.txt"),
                  ("stream_hour.json", "stream_hour.json"), ("stream_hour_regen.json", "stream_hour_regen.json"),
                  ("prof_stream/bench_kernel_stats.csv", "kernel_stats_stream.csv"), ("stream_forcepg.json", "stream_forcepg_nccl_world1.json"),
                  ("stream_host_time.txt", "stream_host_time.txt"), ("prof_rt/rt_kernel_stats.csv", "kernel_stats_realtime_block.csv"), ("prof_share32/bench_kernel_stats.csv", "kernel_stats_32sources.csv"),
                  ("forcepg_timeline.txt", "forcepg_timeline.txt"), ("ab_rolling_x.txt", "ab_rolling_x.txt"),
                  ("ab_sensitivity.txt", "ab_sensitivity.txt"), ("ubench_unit_block.txt", "ubench_unit_block.txt"),
                  ("stamps_fs_256.txt", "stamps_fs_256sources.txt"), ("bench_2ranks_one_device.json", "bench_2ranks_one_device.json"),
                  ("stress_fused.txt", "stress_fused.txt"), ("stress_fused_split.txt", "stress_fused_split.txt")]:
    copy(rel, name)


def line(pth):
    return json.loads([ln for ln in open(pth) if ln.lstrip().startswith("{")][0])


rows = []
for n in (27, 32, 33, 64, 128, 256):
    row = [f"{n:4d} sources"]
    for kind in ("plain", "graph", "forcepg"):
        pth = os.path.join(SRC, f"share_{n}_{kind}.json")
        try:
            d = line(pth)
            row.append(f"{kind} {d['ms_per_step'] * 1e3:7.1f} us/step (FIR kernel {d['roofline']['kernel_ms'] * 1e3:6.1f} us)")
        except Exception as e:                                # noqa: BLE001
            row.append(f"{kind} -")
    rows.append(" | ".join(row))
ch = []
for c in (1, 2, 4, 8):
    try:
        ch.append(f"NCCL_MAX_NCHANNELS={c}: {line(os.path.join(SRC, f'share_32_forcepg_ch{c}.json'))['ms_per_step'] * 1e3:.1f} us/step")
    except Exception:                                         # noqa: BLE001
        pass
if any("us/step" in r for r in rows):
    with open(os.path.join(DST, TAG + "per_rank_shares.txt"), "w") as f:
        f.write("# bench.py --sources N --steps 200 --warmup 10 on ONE GPU: the share of the 256-source scene a rank renders at\n"
                "# 8 / 4 / 2 / 1 GPUs (32 / 64 / 128 / 256), and the root-weighted split at 8 GPUs (27 on rank 0, 33 on the others).\n"
                "# plain = eager launches, the peak rule in the reduce kernel's tail; graph = the same step replayed as one hipGraph;\n"
                "# forcepg = the N > 1 code path on one rank with a real RCCL communicator (world_size 1): graph-replayed render, async\n"
                "# gather of the 3.5 MB partial mix to itself, then ONE launch on the root: fixed-order sum + max|y| + peak rule\n"
                "# (bas_mix_finish_f32; round 3: memset + sum + scale), steps overlapped.\n")
        f.write("\n".join(rows) + "\n")
        if ch:
            f.write("# the 32-source share through --force-pg by RCCL channel count: " + "; ".join(ch) + "\n")

fetch = counters("pmc_FETCH_SIZE/pmc_counter_collection.csv", "pmc_FETCH_SIZE.csv")
write = counters("pmc_WRITE_SIZE/pmc_counter_collection.csv", "pmc_WRITE_SIZE.csv")
for sq in ("pmc_SQ", "pmc_SQ2"):
    p = os.path.join(SRC, sq, "pmc_counter_collection.csv")
    if os.path.exists(p):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sq_summary.py"), p, "bas_render_f"], capture_output=True, text=True).stdout
        open(os.path.join(DST, TAG + sq + "_fs_kernel_summary.txt"), "w").write(out)

fir = [k for k in fetch if "bas_render_fs_kernel" in k or "bas_render_fz_kernel" in k]
if fir and fir[0] in write:
    k = fir[0]
    mean = lambda d, c: sum(d[c]) / len(d[c])
    f_kb, w_kb = mean(fetch[k], "FETCH_SIZE"), mean(write[k], "WRITE_SIZE")
    n_src, t_in, t_out, K, L = 256, 441344, 441471, 512, 128
    n_tiles = -(-t_out // 8192)
    parts = {"x_windows_read": 4 * n_src * n_tiles * (8192 + 128), "read_plans_read": 288 * n_src * n_tiles * (8192 // K + 2),
             "table_once_per_xcd_l2": 8 * 4 * 2 * 187 * 8 * (L + 4), "slab_parts_written": int(w_kb * 1024)}
    step = {kn: {"FETCH_SIZE_KB": round(mean(fetch[kn], "FETCH_SIZE"), 1), "WRITE_SIZE_KB": round(mean(write[kn], "WRITE_SIZE"), 1) if kn in write else None}
            for kn in fetch}
    rec = {"workload": "256x441000@K512S32L128", "fused": True, "kernel": k,
           "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/" + TAG + "pmc_FETCH_SIZE.csv, "
                     + TAG + "pmc_WRITE_SIZE.csv), mean over the dispatches of python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline; "
                     "bench.py measures the same two counters itself in every default run (roofline.traffic) and replays this "
                     "file only where rocprofv3 is not available",
           "FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb,
           "correction": "MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced "
                         "streaming read -> doubled for the whole counter (an upper bound: the table gathers are L2 hits)",
           "bytes_per_launch": int(2 * f_kb * 1024 + w_kb * 1024),
           "parts_expected_bytes": parts,
           "expected_reads_total": parts["x_windows_read"] + parts["read_plans_read"] + parts["table_once_per_xcd_l2"],
           "fetch_size_doubled_bytes": int(2 * f_kb * 1024),
           "every_kernel_of_the_step_KB": step}
    json.dump(rec, open(os.path.join(DST, "fir_hbm_traffic.json"), "w"), indent=1)
    print("FIR traffic MB:", rec["bytes_per_launch"] / 1e6, "expected reads MB:", rec["expected_reads_total"] / 1e6)
print(sorted(x for x in os.listdir(DST) if x.startswith(TAG)))
