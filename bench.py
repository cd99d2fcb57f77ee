#!/usr/bin/env python3
"""Benchmark of the moving-source render path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config 4 of BASELINE.json, per GPU): 256 concurrent moving sources x 10 s of
44.1 kHz mono, chunk K=512, subchunk S=32, IR L=128 taps (samples_to_keep=128, U=8),
synthetic table + seeded noise + per-source spiral/circle trajectories,
mixed to one stereo pair.  One "step" = one full pass of the hot path over that batch
with inputs resident in HBM: bas_interp2d_f32 (all chunk IRs) -> bas_render_mix_f32 (time-varying FIR + overlap-add + mix +
fused peak) -> [N>1: one RCCL gather of the
partial mixes to rank 0 + fixed-order sum] -> peak rule.

Scaling is WEAK: every GPU renders its own 256 sources (the scene has 256*N sources)
and the per-GPU partial mixes meet in ONE gather.  `value` = stereo output samples per
second of 256-source scene equivalents = N * T_out * steps / wall; at N=1 this is
exactly BASELINE.json's "stereo samples/sec for 256 concurrent moving sources".

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`
(dominant kernel = the FIR kernel, timed live with HIP events) and `cpu_baseline`
(the oracle's numpy port of the reference loop, timed on this box's host cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FS = 44100
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VALU_PEAK_TF = 157.3      # MI355X_MICROARCH.md: peak FP32 vector


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sources", type=int, default=256, help="sources per GPU (weak) / in the whole scene (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every GPU renders --sources sources; strong: BASELINE config 4 read "
                         "literally, ONE --sources-source scene sharded over the GPUs")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--chunk", type=int, default=512)
    ap.add_argument("--subchunk", type=int, default=32)
    ap.add_argument("--taps", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["scene", "stream"], default="scene",
                    help="scene = the contract's benchmark (default); stream = BASELINE config 5 shape: long stream "
                         "rendered block by block with carried state, inputs and trajectories generated on the device")
    ap.add_argument("--fs", type=int, default=FS)
    ap.add_argument("--block", type=int, default=262144, help="stream mode: input samples per block")
    ap.add_argument("--regen", action="store_true", help="stream mode: draw a fresh random block inside every timed step")
    ap.add_argument("--cpu-sources-per-core", type=int, default=16)
    return ap.parse_args()


# ---------------------------------------------------------------------------
# CPU baseline (rank 0, N=1): the oracle's port of the reference loop, one process per core
# ---------------------------------------------------------------------------
def _cpu_worker(job):
    """Render `n_src` sources of `n` samples with the numpy port of apply_hrtf.py:431-464."""
    import numpy as np
    from oracle import bas_oracle as orc
    import importlib
    synth = importlib.import_module("binaural-audio-synthesis_amd.synth")
    first, n_src, n, k, s, l, total_src = job
    tbl = synth.make_table("consistent", 0).truncated(l)
    in_length, _ = orc.render_lengths(n, k, l)
    t0 = time.perf_counter()
    for i in range(first, first + n_src):
        x = synth.integer_noise(1000 + i, n, 1.0 / total_src)
        traj = source_trajectory(synth, i, total_src, n)
        irs = orc.chunk_irs(tbl, k, in_length, traj)
        orc.render_from_irs(x, k, s, irs, normalize=False)
    return time.perf_counter() - t0


def source_trajectory(synth, i, total_src, n):
    """spiral / circle per source, phase 2*pi*i/total, period 2 + i/64 s."""
    import numpy as np
    phase = 2 * np.pi * i / total_src
    period = 2.0 + (i % 256) / 64.0
    if i % 2 == 0:
        return synth.trajectory("spiral", fs=FS, length_s=n / FS, turns=5.0, phase=phase)
    return synth.trajectory("circle_askew", fs=FS, period_s=period, phase=phase)


def host_cores():
    """Host cores this process may really use: affinity, capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(cores, int(os.environ.get("BAS_BENCH_MAX_CORES", "64"))))


def cpu_baseline(args, n, t_out):
    import multiprocessing as mp
    cores = host_cores()
    per = max(1, min(args.cpu_sources_per_core, -(-args.sources // cores)))   # at most the whole scene
    jobs = [(c * per, per, n, args.chunk, args.subchunk, args.taps, args.sources) for c in range(cores)]
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        busy = pool.map(_cpu_worker, jobs)                 # seconds each core spent rendering its sources
    wall = time.perf_counter() - t0
    n_rendered = cores * per
    # sources are independent and cost is linear in their number: extrapolate to the full scene.  The all-core
    # figure uses the slowest worker's render time (process start and table construction left out, which
    # favours the CPU); the single-core figure is one worker's rate (the reference itself is single-threaded).
    scene_seconds = max(busy) * args.sources / n_rendered
    one_core_scene_seconds = sorted(busy)[len(busy) // 2] * args.sources / per
    return {"value": t_out / scene_seconds, "unit": "stereo samples/s", "cores": cores, "kind": "port",
            "sample": f"{n_rendered} of {args.sources} sources x {n} samples ({per} per core, {cores} processes, "
                      f"slowest worker {max(busy):.1f} s of rendering, {wall:.1f} s wall incl. process start), "
                      f"extrapolated linearly to {args.sources} sources",
            "x_realtime": (n / FS) / scene_seconds,
            "single_core": {"value": t_out / one_core_scene_seconds, "unit": "stereo samples/s",
                            "x_realtime": (n / FS) / one_core_scene_seconds,
                            "x_realtime_one_source": (n / FS) / (one_core_scene_seconds / args.sources)}}


# ---------------------------------------------------------------------------
class HipEvents:
    """Raw hipEvent_t pairs via libamdhip64 (the library records them on the launch stream)."""

    def __init__(self, n):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.pairs = []
        for _ in range(n):
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            assert self.hip.hipEventCreate(ctypes.byref(a)) == 0 and self.hip.hipEventCreate(ctypes.byref(b)) == 0
            self.pairs.append((a, b))

    def elapsed_ms(self, i):
        ms = ctypes.c_float()
        rc = self.hip.hipEventElapsedTime(ctypes.byref(ms), self.pairs[i][0], self.pairs[i][1])
        assert rc == 0, f"hipEventElapsedTime rc={rc}"
        return ms.value


def stream_mode(args):
    """BASELINE config 5 (1024 sources, 48 kHz, hours of audio) in miniature: `steps` blocks of `block`
    samples through StreamRenderer; nothing but one block of inputs, chunk IRs and outputs is ever resident,
    and the trajectory -> parameter step runs on the device.  With --gpus N (torch.distributed.run) the
    1024 sources are sharded over the ranks and every block ends in one gather of the partial stereo block
    (distributed.ShardedStreamRenderer).  One JSON line, not the contract's metric."""
    import math
    import torch
    import torch.distributed as dist
    import binaural_audio_synthesis_amd as bas
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if os.environ.get("BAS_BENCH_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        backend = os.environ.get("BAS_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    n_total, k, s, l, fs, B = args.sources, args.chunk, args.subchunk, args.taps, args.fs, args.block
    host = bas.synth.make_table("consistent", 0).truncated(l)
    tbl = bas.irs_and_delaydiffs(host.upsampling, host.diffs_left, host.diffs_right, host.irs_left, host.irs_right,
                                 device=dev)
    st = bas.distributed.ShardedStreamRenderer(tbl, n_total, k, s)
    n_src = len(st.sources)
    src = torch.arange(st.sources.start, st.sources.stop, dtype=torch.float64, device=dev)[:, None]
    phase = 2 * math.pi * src / n_total
    period = (2.0 + (src % 256) / 64.0) * fs
    gen = torch.Generator(device=dev).manual_seed(5 + rank)
    gloo = world > 1 and dist.get_backend() == "gloo"

    # the input block lives in the renderer's own input buffer (what a decoder / H2D copy would fill in place):
    # resident in HBM when a timed step starts, as the bench contract asks; --regen draws a fresh block per step
    xin = st.local.input_view(B)
    xin.copy_((torch.rand((n_src, B), generator=gen, device=dev) * 2 - 1) * (1.0 / n_total))

    def block(i):
        t = (torch.arange(B // k + 1, dtype=torch.float64, device=dev)[None, :] * k + i * B)
        elev = (math.pi / 4) * torch.cos(2 * math.pi * t / period + phase)          # askew circles, per-source period
        azim = 2 * math.pi * t / period + phase
        if args.regen:
            xin.copy_((torch.rand((n_src, B), generator=gen, device=dev) * 2 - 1) * (1.0 / n_total))
        return xin, elev, azim

    def step(i):
        return st.process(*block(i))          # under gloo (rehearsal) gather_mix stages the gather through the host

    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        y = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cpu" if gloo else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    audio_s = args.steps * B / fs
    if rank == 0:
        print(json.dumps({"metric": "streaming render, x real-time", "value": audio_s / el, "unit": "x real-time",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "strong", "dtype": "f32",
                          "data": "synthetic; input block resident in the renderer's input buffer" + (" (redrawn every step, included in time)" if args.regen else "") + ", trajectories computed per block on the device (included in time)",
                          "config": {"workload": f"BASELINE config 5 shape: {n_total} sources @ {fs} Hz streamed in blocks of {B} "
                                                 f"samples, chunk {k}, subchunk {s}, {l} taps; sources sharded over {world} GPU(s), "
                                                 f"one gather per block", "block": B},
                          "source_samples_per_s": n_total * B * args.steps / el,
                          "hour_of_audio_seconds": 3600.0 / (audio_s / el), "peak": st.peak,
                          "out_block_shape": list(y.shape)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.mode == "stream":
        return stream_mode(args)
    import numpy as np
    import torch
    import torch.distributed as dist
    import binaural_audio_synthesis_amd as bas
    from binaural_audio_synthesis_amd import _hip

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    # rehearsal knobs (a 1-GPU box): BAS_BENCH_ONE_DEVICE=1 puts every rank on cuda:0,
    # BAS_BENCH_BACKEND=gloo stages the gather through host memory.  The driver's runs use neither.
    if os.environ.get("BAS_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("BAS_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    n = int(round(args.seconds * FS))
    k, s, l, n_src = args.chunk, args.subchunk, args.taps, args.sources
    if args.scaling == "strong":                             # one scene, sources split over the ranks
        n_src = len(bas.distributed.shard_sources(args.sources, world, rank))
    in_length = -(-n // k) * k
    t_out = in_length + l - 1
    n_q = in_length // k + 1

    # ---- inputs, resident in HBM before the timed region ---------------------------------
    host = bas.synth.make_table("consistent", 0).truncated(l)
    tbl = bas.irs_and_delaydiffs(host.upsampling, host.diffs_left, host.diffs_right, host.irs_left,
                                 host.irs_right, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    total_src = n_src * world if args.scaling == "weak" else args.sources
    first_src = rank * n_src if args.scaling == "weak" else bas.distributed.shard_sources(args.sources, world, rank).start
    x = torch.zeros((n_src, in_length), dtype=torch.float32, device=dev)
    x[:, :n] = (torch.rand((n_src, n), generator=gen, device=dev) * 2 - 1) * (1.0 / total_src)
    tq = np.arange(0, in_length + 1, k, dtype=np.float64)
    elev = np.empty((n_src, n_q))
    azim = np.empty((n_src, n_q))
    for i in range(n_src):
        elev[i], azim[i] = source_trajectory(bas.synth, first_src + i, total_src, n)(tq)
    idx_h, w_h = bas.sphere.interpolation_params_batch(elev, azim)
    idx = torch.from_numpy(idx_h.reshape(-1, 4)).to(dev)
    w = torch.from_numpy(w_h.reshape(-1, 3)).to(dev)
    y = torch.empty((2, t_out), dtype=torch.float32, device=dev)
    ws = torch.empty((_hip.lib().bas_render_workspace_bytes(n_src, in_length, k, s, l),), dtype=torch.uint8,
                     device=dev)
    ws_i = torch.empty((_hip.lib().bas_interp2d_workspace_bytes(n_src * n_q),), dtype=torch.uint8, device=dev)
    parts = torch.empty((world, 2, t_out), dtype=torch.float32, device=dev) if (world > 1 and rank == 0) else None
    y_final = torch.empty((2, t_out), dtype=torch.float32, device=dev)
    peak = torch.empty((1,), dtype=torch.float32, device=dev)
    ev = HipEvents(args.steps)

    fused = os.environ.get("BAS_BENCH_FUSED", "0") == "1"   # chunk IRs inside the FIR kernel (bas_render_mix_fused_f32)

    def render_into(y_buf, i_event):
        events = None if i_event is None else ev.pairs[i_event]
        return bas.apply_hrtf.render_params_device(x, k, s, tbl, idx, w, normalize="none", out=y_buf, events=events,
                                                   ws=ws, ws_plans=ws_i, fused=fused)[1]

    def mix_on_root(parts_buf):
        stream = _hip.current_stream(dev)
        _hip.call("bas_mix_partials_f32", _hip.ptr(parts_buf), world, 2 * t_out, 2 * t_out, _hip.ptr(y_final),
                  _hip.ptr(peak), stream)
        _hip.call("bas_scale_by_peak_f32", _hip.ptr(y_final), 2 * t_out, _hip.ptr(peak), stream)

    def step_sync(i_event=None):
        pk = render_into(y, i_event)
        if world > 1 and backend != "nccl":              # rehearsal only: gather through host memory
            y_host = y.cpu()
            if rank == 0:
                host_parts = [torch.empty_like(y_host) for _ in range(world)]
                dist.gather(y_host, gather_list=host_parts, dst=0)
                parts.copy_(torch.stack(host_parts))
            else:
                dist.gather(y_host, gather_list=None, dst=0)
        elif world > 1:
            dist.gather(y, gather_list=list(parts.unbind(0)) if rank == 0 else None, dst=0)
        if world > 1:
            if rank == 0:
                mix_on_root(parts)
        else:
            _hip.call("bas_scale_by_peak_f32", _hip.ptr(y), 2 * t_out, _hip.ptr(pk), _hip.current_stream(dev))

    # N > 1: the gather of step i travels (RCCL stream, xGMI) while step i+1 renders; y and the root's receive
    # buffer are double-buffered, the root sums step i right after it has launched step i+1's gather.  Every
    # collective is issued by all ranks in step order; drain() inside the timed region completes the last one.
    # BAS_BENCH_SYNC_GATHER=1 restores gather-then-continue.
    overlap = world > 1 and os.environ.get("BAS_BENCH_SYNC_GATHER", "0") != "1"
    ys = [y, torch.empty_like(y)] if overlap else [y]
    parts2 = [parts, torch.empty_like(parts)] if (overlap and rank == 0) else [parts]
    inflight = [None, None]
    counter = [0]

    def launch_gather(b):
        if backend == "nccl":
            work = dist.gather(ys[b], gather_list=list(parts2[b].unbind(0)) if rank == 0 else None, dst=0,
                               async_op=True)
            return (work, None, None)
        y_host = ys[b].cpu()                              # rehearsal only: gloo moves host memory
        host_parts = [torch.empty_like(y_host) for _ in range(world)] if rank == 0 else None
        return (dist.gather(y_host, gather_list=host_parts, dst=0, async_op=True), host_parts, y_host)

    def finish_gather(b):
        if inflight[b] is None:
            return
        work, host_parts, _ = inflight[b]
        work.wait()                                       # nccl: the current stream waits, the host does not
        if rank == 0:
            if host_parts is not None:
                parts2[b].copy_(torch.stack(host_parts))
            mix_on_root(parts2[b])
        inflight[b] = None

    def step_overlapped(i_event=None):
        b = counter[0] & 1
        counter[0] += 1
        finish_gather(b)                                  # the collective that read ys[b] two steps ago is done
        render_into(ys[b], i_event)
        inflight[b] = launch_gather(b)
        if rank == 0:
            finish_gather(b ^ 1)                          # previous step: its gather ran beside this render

    def drain():
        if overlap:
            b = counter[0] & 1
            finish_gather(b)                              # older one first
            finish_gather(b ^ 1)

    step = step_overlapped if overlap else step_sync

    def fence():
        drain()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if overlap and os.environ.get("BAS_BENCH_CHECK") == "1":   # rehearsal: the pipelined mix equals the synchronous one
        pipelined = y_final.clone() if rank == 0 else None
        step_sync()
        torch.cuda.synchronize(dev)
        if rank == 0:
            assert torch.equal(pipelined, y_final), "overlapped gather changed the mix"
            print("check: pipelined mix == synchronous mix", file=sys.stderr, flush=True)

    fir_ms = [ev.elapsed_ms(i) for i in range(args.steps)]
    fir_avg_s = sum(fir_ms) / len(fir_ms) / 1e3

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        # weak: 256-source-scene equivalents per second; strong: the one scene's stereo samples per second
        scenes = world if args.scaling == "weak" else 1
        value = scenes * t_out * args.steps / elapsed
        # algorithmic bytes of one FIR launch: inputs once, stereo mix once,
        # table once, 28 B of parameters per chunk IR
        m_cols = l * host.upsampling
        algo_bytes = 4 * n_src * in_length + 8 * t_out + 4 * (2 * 187 * m_cols + 2 * 187 * 187) + 28 * n_src * n_q
        algo_flops = 4 * l * n_src * in_length            # 2 ears x L FMA per source-sample
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "fir_hbm_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                rec = json.load(f)
            if rec.get("workload") == f"{n_src}x{n}@K{k}S{s}L{l}":
                traffic = rec.get("bytes_per_launch")
        out = {
            "metric": "stereo samples/sec, 256 concurrent moving sources per GPU @44.1kHz (x real-time in x_realtime)",
            "value": value, "unit": "stereo samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config 4 per GPU: {n_src} moving sources x {args.seconds:g} s @ {FS} Hz mono "
                                   f"-> 1 stereo mix; chunk {k}, subchunk {s}, {l}-tap HRIRs (U=8, 187 directions), "
                                   f"spiral/askew-circle trajectories; scene = {total_src} sources on {world} GPU(s)",
                       "sources_per_gpu": n_src, "samples_per_source": n, "chunk": k, "subchunk": s, "taps": l,
                       "out_samples": t_out, "parallelism": f"sources sharded over {world} GPU(s), 1 gather per step" +
                                                      (", travelling beside the next step's render" if overlap else "")},
            "x_realtime": (n / FS) * scenes / (elapsed / args.steps),
            "source_samples_per_s": total_src * in_length * args.steps / elapsed,
            "roofline": {"bound": "hbm", "achieved": algo_bytes / fir_avg_s / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": algo_bytes / fir_avg_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": _hip.lib().bas_render_kernel_name(n_src, in_length, k, s, l).decode(), "kernel_ms": fir_avg_s * 1e3,
                         "algorithmic_bytes": algo_bytes},
            "valu": {"achieved": algo_flops / fir_avg_s / 1e12, "peak": FP32_VALU_PEAK_TF, "unit": "TFLOP/s",
                     "frac": algo_flops / fir_avg_s / 1e12 / FP32_VALU_PEAK_TF,
                     "note": "the FIR is 128 flop/B: fp32-VALU bound, HBM fraction tops out near 15 %; peak is nominal "
                             "(2.4 GHz) - a bare v_pk_fma_f32 stream on random operands sustains 97 TFLOP/s here "
                             "(power-limited clock, tools/ubench_fir_pattern.hip, DESIGN.md 4)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, n, t_out)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
