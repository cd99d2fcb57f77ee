"""GPU parity tests added in round 2: the paths VERDICT r01 found untested.

  * BASELINE config 5 at its shape: 1024 sources, one 262 144-sample block + finish() through
    StreamRenderer with DEVICE float64 trajectories (the exact `bench.py --mode stream` path), windows
    checked against the oracle's float64 definition.
  * the product's own load_irs_and_delaydiffs (apply_hrtf.py:23-46) on a .mat written in the layout the
    reference reads, and a golden render through it; the CLI's --table path.
  * distributed.render_time_sharded and a two-rank ShardedStreamRenderer with the HIP renderer (two ranks on
    this box's one GPU under gloo).
  * tables with other upsampling factors (the reference accepts any; U < 4 takes the plain kernel).
  * the device-table cache of foreign table structs.

Tolerance: REL = 1e-5 norm-relative (max|got - want| / max|want|), as everywhere.
"""
import json
import os

import numpy as np
import pytest

from conftest import golden, rel_err
from oracle import bas_oracle as orc
import binaural_audio_synthesis_amd as bas

pytestmark = pytest.mark.gpu
REL = 1e-5


def _device_table(h):
    return bas.irs_and_delaydiffs(h.upsampling, h.diffs_left, h.diffs_right, h.irs_left, h.irs_right)


# ---------------------------------------------------------------------------
# a1: the product's loader
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["spiral_512_32_128", "spiral_512_32_100", "askew_128_16_100"])
def test_load_irs_and_delaydiffs_from_mat(tmp_path, tables, case):
    """bas.load_irs_and_delaydiffs (apply_hrtf.py:23-46): struct indexing, upsampling as int, truncation to
    samples_to_keep * upsampling columns, then a reference golden rendered through the loaded table."""
    g = golden(f"render_{case}.npz")
    meta = json.loads(str(g["meta"]))
    keep, full = meta["L"], tables[meta["table"]]
    path = str(tmp_path / "irs_and_delaydiffs_synth.mat")
    bas.synth.save_table_mat(path, full)
    d = bas.load_irs_and_delaydiffs(path, samples_to_keep=keep)
    assert isinstance(d.upsampling, int) and d.upsampling == 8
    assert tuple(d.irs_left.shape) == (187, keep * 8) and tuple(d.diffs_left.shape) == (187, 187)
    want = full.truncated(keep)
    assert np.array_equal(d.irs_left.cpu().numpy(), want.irs_left.astype(np.float32))
    assert np.array_equal(d.irs_right.cpu().numpy(), want.irs_right.astype(np.float32))
    assert np.array_equal(d.diffs_left.cpu().numpy(), full.diffs_left)
    assert np.array_equal(d.diffs_right.cpu().numpy(), full.diffs_right)
    traj = bas.synth.trajectory(meta["traj"], fs=meta["fs"], **meta["traj_kw"])
    got = bas.make_signal_move_2d(g["x"], meta["K"], meta["S"], traj, d)
    assert got.shape == g["y"].shape and rel_err(got, g["y"]) <= REL
    pts = golden("interp2d.npz")
    one = bas.interpolate_2d(d, np.float64(pts["points"][3, 0]), np.float64(pts["points"][3, 1]))
    assert rel_err(one, pts[f"{meta['table']}_{keep}"][3]) <= REL


def test_cli_with_table_file(tmp_path, tables):
    """cli.main --table: WAV in, the .mat loader, render, WAV out (apply_hrtf.py:570-646) vs the oracle."""
    import scipy.io.wavfile as wavfile
    from binaural_audio_synthesis_amd import cli
    fs, n = 44100, 6000
    pcm = (bas.synth.integer_noise(91, n, 0.3) * 32767).astype(np.int16)
    src = str(tmp_path / "mono.wav")
    wavfile.write(src, fs, pcm)
    mat = str(tmp_path / "table.mat")
    bas.synth.save_table_mat(mat, tables["consistent"])
    out = cli.main([src, "--table", mat, "--trajectory", "spiral"])
    _, got = wavfile.read(out)
    y = pcm.astype(np.float32) / pcm.max()
    want = orc.render(y, 512, 32, cli.presets(fs)["spiral"], tables["consistent"].truncated(100))
    assert got.shape == want.shape and rel_err(got, want) <= REL


# ---------------------------------------------------------------------------
# config 5: 1024 sources, device trajectories, streamed
# ---------------------------------------------------------------------------
def test_stream_config5_shape_device_trajectories(tables):
    """BASELINE config 5 shape on one GPU: 1024 sources @ 48 kHz, K=512, S=32, L=128, one 262 144-sample
    block + finish() through StreamRenderer, trajectories handed over as DEVICE float64 tensors (so a3 runs in
    bas_traj_params_f64) and the block written into the renderer's own input buffer - the path
    `bench.py --mode stream` times.  Windows of the mix (block start, a chunk boundary, a FIR-tile boundary, the
    block / finish seam, the L-1 tail) against oracle.render_window; every source contributes to every window."""
    import math
    import torch
    h = tables["consistent"].truncated(128)
    d = _device_table(h)
    n_src, k, s, l, fs, B = 1024, 512, 32, 128, 48000, 262144
    st = bas.StreamRenderer(d, n_src, k, s)
    gen = torch.Generator(device="cuda").manual_seed(55)
    xin = st.input_view(B)
    xin.copy_((torch.rand((n_src, B), generator=gen, device="cuda") * 2 - 1) * (1.0 / n_src))
    src = torch.arange(n_src, dtype=torch.float64, device="cuda")[:, None]
    phase = 2 * math.pi * src / n_src
    period = (2.0 + (src % 256) / 64.0) * fs
    t = torch.arange(B // k + 1, dtype=torch.float64, device="cuda")[None, :] * k
    elev = (math.pi / 4) * torch.cos(2 * math.pi * t / period + phase)
    azim = 2 * math.pi * t / period + phase
    y = torch.cat([st.process(xin, elev, azim), st.finish()], dim=0)
    assert y.shape == (B + l - 1, 2)
    scale = float(y.abs().max())
    assert abs(st.peak - scale) <= 1e-6 * scale
    x_host = xin.double().cpu().numpy()
    e_host, a_host = elev.cpu().numpy(), azim.cpu().numpy()
    halo = st.halo
    windows = [(0, 48), (300 * k - 24, 300 * k + 24), (20 * 8192 - halo - 24, 20 * 8192 - halo + 24),
               (B - 24, B + 24), (B + l - 1 - 48, B + l - 1)]
    worst = 0.0
    for n0, n1 in windows:
        m0, m1 = max(n0 - l + 1, 0), min(n1, B)
        want = np.zeros((2, n1 - n0))
        for i in range(n_src):
            cache = {}

            def ir_of(c, i=i, cache=cache):
                c = min(c, B // k)                       # past the stream's end only silence is filtered
                if c not in cache:
                    cache[c] = orc.interp2d(h, e_host[i, c], a_host[i, c])
                return cache[c]
            want += orc.render_window(x_host[i, m0:m1], m0, k, s, ir_of, l, n0, n1)
        got = y[n0:n1].double().cpu().numpy().T
        worst = max(worst, float(np.abs(got - want).max()) / scale)
    assert worst <= REL, worst


def test_stream_host_and_device_trajectories_agree(tables):
    """Same blocks once with host (numpy) trajectories, once with device tensors: identical audio."""
    import torch
    h = tables["consistent"].truncated(128)
    d = _device_table(h)
    n_src, k, s, blocks = 5, 512, 32, (2048, 512, 4096)
    n = sum(blocks)
    sigs = torch.from_numpy(np.stack([bas.synth.integer_noise(70 + i, n, 0.1) for i in range(n_src)])).cuda()
    t = np.arange(0, n + 1, k, dtype=np.float64)
    elev = np.stack([bas.synth.trajectory("circle_askew", period_s=0.05 + 0.02 * i, phase=i)(t)[0] for i in range(n_src)])
    azim = np.stack([bas.synth.trajectory("circle_askew", period_s=0.05 + 0.02 * i, phase=i)(t)[1] for i in range(n_src)])
    a, b = bas.StreamRenderer(d, n_src, k, s), bas.StreamRenderer(d, n_src, k, s)
    pos = 0
    for blk in blocks:
        c0, c1 = pos // k, (pos + blk) // k
        ya = a.process(sigs[:, pos:pos + blk], elev[:, c0:c1 + 1], azim[:, c0:c1 + 1])
        yb = b.process(sigs[:, pos:pos + blk], torch.from_numpy(elev[:, c0:c1 + 1].copy()).cuda(),
                       torch.from_numpy(azim[:, c0:c1 + 1].copy()).cuda())
        assert torch.equal(ya, yb)
        pos += blk
    assert torch.equal(a.finish(), b.finish())


# ---------------------------------------------------------------------------
# multi-rank paths with the HIP renderer (two ranks on this box's one GPU, gloo)
# ---------------------------------------------------------------------------
def _scene(n_src, n, k):
    h = bas.synth.make_table("consistent", 0).truncated(128)
    sigs = np.stack([bas.synth.integer_noise(500 + i, n, 0.2 / n_src) for i in range(n_src)])
    in_length = -(-n // k) * k
    t = np.arange(0, in_length + 1, k, dtype=np.float64)
    elev = np.empty((n_src, t.size))
    azim = np.empty((n_src, t.size))
    for i in range(n_src):
        name = ("spiral", "circle_askew", "passing")[i % 3]
        elev[i], azim[i] = bas.synth.trajectory(name, period_s=0.06 + 0.01 * i, length_s=n / 44100, turns=2.0 + i,
                                                phase=0.3 * i)(t)
    return h, sigs, elev, azim


def _two_rank_hip_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    n_src, n, k, s = 5, 20000, 512, 32
    h, sigs, elev, azim = _scene(n_src, n, k)
    d = _device_table(h)
    # (1) by time: every rank renders all sources over its own chunk range with the HIP kernels
    y = bas.distributed.render_time_sharded(torch.from_numpy(sigs).cuda(), k, s, elev, azim, d, 128, normalize="none")
    if rank == 0:
        np.save(os.path.join(out_dir, "time.npy"), y.cpu().numpy())
    else:
        assert y is None
    # (2) streams: sources sharded, one gather per block, carried state per rank
    st = bas.distributed.ShardedStreamRenderer(d, n_src, k, s)
    mine = slice(st.sources.start, st.sources.stop)
    outs, pos = [], 0
    blocks = (4096, 1024, 8192, 7168)                      # = in_length 20480 (n = 20000 padded to chunks)
    for blk in blocks:
        c0, c1 = pos // k, (pos + blk) // k
        xb = np.zeros((n_src, blk), dtype=np.float32)
        avail = max(0, min(n - pos, blk))
        xb[:, :avail] = sigs[:, pos:pos + avail]
        out = st.process(xb[mine], elev[mine, c0:c1 + 1], azim[mine, c0:c1 + 1])
        if rank == 0:
            outs.append(out.cpu().numpy())
        else:
            assert out is None
        pos += blk
    tail = st.finish()
    if rank == 0:
        outs.append(tail.cpu().numpy())
        np.save(os.path.join(out_dir, "stream.npy"), np.concatenate(outs, axis=0))
        np.save(os.path.join(out_dir, "peak.npy"), np.array([st.peak]))
    dist.barrier()
    dist.destroy_process_group()


def test_time_sharding_and_sharded_stream_two_ranks_hip(tmp_path):
    """distributed.render_time_sharded and a two-rank ShardedStreamRenderer driven by the real kernels:
    both must reproduce the one-process render of the whole scene (and the oracle)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_two_rank_hip_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    n_src, n, k, s = 5, 20000, 512, 32
    h, sigs, elev, azim = _scene(n_src, n, k)
    assert -(-n // k) * k == 4096 + 1024 + 8192 + 7168
    d = _device_table(h)
    one = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none").cpu().numpy()
    irs = [np.stack([orc.interp2d(h, elev[i, c], azim[i, c]) for c in range(elev.shape[1])]) for i in range(n_src)]
    want = orc.render_mix(sigs, k, s, irs, normalize=False)
    assert rel_err(one, want) <= REL
    by_time = np.load(str(tmp_path / "time.npy"))
    stream = np.load(str(tmp_path / "stream.npy"))
    assert by_time.shape == one.shape and rel_err(by_time, one) <= 2e-6 and rel_err(by_time, want) <= REL
    assert stream.shape == one.shape and rel_err(stream, one) <= 2e-6 and rel_err(stream, want) <= REL
    peak = float(np.load(str(tmp_path / "peak.npy"))[0])
    assert abs(peak - np.abs(one).max()) <= 2e-6 * np.abs(one).max()


# ---------------------------------------------------------------------------
# other upsampling factors
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("u", [1, 2, 3, 4, 5, 16])
def test_other_upsampling_factors(u):
    """The reference takes the factor from the table (apply_hrtf.py:38) and works for any; here U >= 4 runs the
    planned evaluation (and the fused FIR), U < 4 bas_interp2d_f32's plain kernel.  interpolate_2d at grid nodes,
    clamps and the pole plus a moving-source render, against the oracle on a U-upsampled synthetic table."""
    full = bas.synth.make_table("adversarial", 3, upsampling=u)
    h = full.truncated(64)
    d = _device_table(h)
    rng = np.random.default_rng(100 + u)
    elev = np.concatenate([rng.uniform(-1.0, 1.7, 40), np.deg2rad([-45.0, 0.0, 60.0, 75.0, 90.0, 89.9999])])
    azim = np.concatenate([rng.uniform(-7, 7, 40), np.deg2rad([15.0, 359.99, 30.0, 0.0, 10.0, 200.0])])
    got = bas.interpolate_2d_batch(d, elev, azim).cpu().numpy()
    want = np.stack([orc.interp2d(h, e, a) for e, a in zip(elev, azim)])
    assert got.shape == want.shape
    assert np.abs(got - want).max() / np.abs(want).max() <= REL
    n_src, n, k, s = 2, 6000, 512, 32
    sigs = np.stack([bas.synth.integer_noise(900 + u + i, n, 0.05) for i in range(n_src)])
    t = np.arange(0, -(-n // k) * k + 1, k, dtype=np.float64)
    te = rng.uniform(-1.0, 1.7, size=(n_src, t.size))
    ta = rng.uniform(-7, 7, size=(n_src, t.size))
    irs = [np.stack([orc.interp2d(h, te[i, c], ta[i, c]) for c in range(t.size)]) for i in range(n_src)]
    y = bas.render_sources(sigs, k, s, te, ta, d, normalize="none").cpu().numpy()
    assert rel_err(y, orc.render_mix(sigs, k, s, irs, normalize=False)) <= REL
    if u < 4:                                             # the planned / fused entry points refuse, loudly
        import torch
        idx_t = torch.zeros((1, 4), dtype=torch.int32, device="cuda")
        w_t = torch.zeros((1, 3), dtype=torch.float64, device="cuda")
        plans = torch.empty((bas._hip.lib().bas_interp2d_workspace_bytes(1),), dtype=torch.uint8, device="cuda")
        with pytest.raises(bas._hip.BasError) as err:
            bas._hip.call("bas_interp2d_plan_f32", bas._hip.ptr(d.diffs), bas._hip.ptr(idx_t), bas._hip.ptr(w_t), 1, 187,
                          64, u, bas._hip.ptr(plans), plans.numel(), None)
        assert err.value.code == -2 and "upsampling" in str(err.value)


# ---------------------------------------------------------------------------
# device-table cache of foreign structs
# ---------------------------------------------------------------------------
def test_device_table_cache_follows_rebinding(tables):
    """as_device_table caches the device copy of a foreign struct (e.g. the reference's class-as-struct) but
    must notice re-bound fields: re-truncating irs_left/irs_right gives a new table, not stale IRs."""
    full = tables["consistent"]

    class Struct:                                         # what apply_hrtf.py:36-44 builds
        pass
    t = Struct()
    h128 = full.truncated(128)
    t.upsampling, t.diffs_left, t.diffs_right = h128.upsampling, h128.diffs_left, h128.diffs_right
    t.irs_left, t.irs_right = h128.irs_left, h128.irs_right
    a = bas.apply_hrtf.as_device_table(t)
    assert bas.apply_hrtf.as_device_table(t) is a and a.L == 128
    h100 = full.truncated(100)
    t.irs_left, t.irs_right = h100.irs_left, h100.irs_right
    b = bas.apply_hrtf.as_device_table(t)
    assert b is not a and b.L == 100
    pts = golden("interp2d.npz")
    one = bas.interpolate_2d(t, np.float64(pts["points"][7, 0]), np.float64(pts["points"][7, 1]))
    assert one.shape == (2, 100) and rel_err(one, pts["consistent_100"][7]) <= REL


# ---------------------------------------------------------------------------
# streaming: the hipGraph fast path
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("k,s,B", [(512, 32, 512), (512, 32, 2048), (128, 32, 128), (1024, 64, 1024)])
def test_stream_graph_replay_equals_plain_launches(tables, k, s, B):
    """From the second block of a size on StreamRenderer replays one captured hipGraph per block (a3, read plans,
    fused FIR, carry copies, running peak).  Many equal blocks through the graph path == the same blocks through
    plain launches (graph=False) == the whole-signal render; producers may write audio and trajectories straight
    into the renderer's buffers (input_view / trajectory_views) and read the result in place (copy_out=False)."""
    import torch
    h = tables["consistent"].truncated(128)
    d = _device_table(h)
    n_src, n_blocks = 6, 9
    n = B * n_blocks
    sigs = torch.from_numpy(np.stack([bas.synth.integer_noise(300 + i, n, 0.1) for i in range(n_src)])).cuda()
    t = np.arange(0, n + 1, k, dtype=np.float64)
    elev = np.stack([bas.synth.trajectory("circle_askew", period_s=0.04 + 0.013 * i, phase=0.7 * i)(t)[0] for i in range(n_src)])
    azim = np.stack([bas.synth.trajectory("circle_askew", period_s=0.04 + 0.013 * i, phase=0.7 * i)(t)[1] for i in range(n_src)])
    whole = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none")
    g = bas.StreamRenderer(d, n_src, k, s, graph=True, copy_out=False)
    e = bas.StreamRenderer(d, n_src, k, s, graph=False)
    outs_g, outs_e = [], []
    for b in range(n_blocks):
        c0, c1 = b * B // k, (b + 1) * B // k
        xin = g.input_view(B)
        ev, av = g.trajectory_views(B)
        xin.copy_(sigs[:, b * B:(b + 1) * B])
        ev.copy_(torch.from_numpy(elev[:, c0:c1 + 1].copy()))
        av.copy_(torch.from_numpy(azim[:, c0:c1 + 1].copy()))
        y = g.process(xin, ev, av)
        assert y.data_ptr() == g._y.data_ptr() + 4 * g.halo                # a view of the renderer's own output
        outs_g.append(y.clone())
        outs_e.append(e.process(sigs[:, b * B:(b + 1) * B], elev[:, c0:c1 + 1], azim[:, c0:c1 + 1]))
        assert torch.equal(outs_g[-1], outs_e[-1]), b
    assert g._graph is not None and e._graph is None
    outs_g.append(g.finish())
    outs_e.append(e.finish())
    got = torch.cat(outs_g, dim=0)
    assert torch.equal(got, torch.cat(outs_e, dim=0))
    assert got.shape == whole.shape and rel_err(got.cpu().numpy(), whole.cpu().numpy()) <= 1e-6
    assert g.peak == e.peak and abs(g.peak - float(whole.abs().max())) <= 1e-6 * g.peak


# ---------------------------------------------------------------------------
# fused entry point: accumulate, direct output, long IRs
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n_src,n,l", [(1, 30000, 128), (1, 9000, 300), (6, 20000, 128), (40, 150000, 100), (40, 6000, 128)])
def test_fused_accumulate_and_direct_output(tables, n_src, n, l):
    """bas_render_mix_fused_f32 called directly: rendering the sources in two calls, the second with accumulate = 1,
    equals one call (and the oracle), and the reported peak is max|y| of the sum.  One source takes the direct-output
    form (no slabs; also with three tap segments at L = 300), several sources the slab + reduce form (40 x 150 000:
    tiles of 8192; 40 x 6 000: 41 parts per tile, summed by the wide reduce kernel)."""
    import torch
    from binaural_audio_synthesis_amd import _hip
    h = tables["consistent"].truncated(l)
    d = _device_table(h)
    k, s = 512, 32
    sigs = np.stack([bas.synth.integer_noise(1200 + i, n, 0.3 / n_src) for i in range(n_src)])
    in_length = -(-n // k) * k
    t = np.arange(0, in_length + 1, k, dtype=np.float64)
    elev = np.stack([bas.synth.trajectory("spiral", length_s=n / 44100, turns=1.0 + i % 5, phase=0.4 * i)(t)[0] for i in range(n_src)])
    azim = np.stack([bas.synth.trajectory("spiral", length_s=n / 44100, turns=1.0 + i % 5, phase=0.4 * i)(t)[1] for i in range(n_src)])
    lib = _hip.lib()
    x = torch.zeros((n_src, in_length), dtype=torch.float32, device="cuda")
    x[:, :n] = torch.from_numpy(sigs).cuda()
    idx, w = bas.sphere.interpolation_params_device(torch.from_numpy(elev).cuda(), torch.from_numpy(azim).cuda())
    idx, w = idx.reshape(-1, 4), w.reshape(-1, 3)
    n_q = in_length // k + 1
    t_out = in_length + l - 1
    stream = _hip.current_stream(x.device)

    def fused(x_part, idx_part, w_part, y, accumulate):
        ns = x_part.shape[0]
        assert lib.bas_render_fused_supported(ns, in_length, k, s, l) == 1
        plans = torch.empty((lib.bas_interp2d_workspace_bytes(idx_part.shape[0]),), dtype=torch.uint8, device="cuda")
        ws = _hip.new_workspace(lib.bas_render_fused_workspace_bytes(ns, in_length, k, s, l), "cuda")
        peak = torch.zeros(1, dtype=torch.float32, device="cuda")
        _hip.call("bas_interp2d_plan_f32", _hip.ptr(d.diffs), _hip.ptr(idx_part), _hip.ptr(w_part), idx_part.shape[0], d.ndir, l,
                  d.upsampling, _hip.ptr(plans), plans.numel(), stream)
        _hip.call("bas_render_mix_fused_f32", _hip.ptr(x_part), x_part.stride(0), _hip.ptr(d.packed), _hip.ptr(plans), ns, in_length,
                  k, s, l, d.upsampling, d.ndir, _hip.ptr(y), accumulate, _hip.ptr(peak), 0, _hip.ptr(ws), ws.numel(), stream)
        return float(peak)

    y_one = torch.full((2, t_out), 7.0, dtype=torch.float32, device="cuda")            # must be overwritten
    p_one = fused(x, idx.contiguous(), w.contiguous(), y_one, 0)
    irs = [np.stack([orc.interp2d(h, elev[i, c], azim[i, c]) for c in range(n_q)]) for i in range(n_src)]
    want = orc.render_mix(sigs, k, s, irs, normalize=False)
    assert rel_err(y_one.t().cpu().numpy(), want) <= REL
    assert abs(p_one - float(y_one.abs().max())) <= 1e-6 * p_one
    # the same scene in two calls: first part overwrites, second accumulates (a single source: itself twice)
    cut = max(n_src // 2, 1)
    y_two = torch.full((2, t_out), -3.0, dtype=torch.float32, device="cuda")
    fused(x[:cut].contiguous(), idx[:cut * n_q].contiguous(), w[:cut * n_q].contiguous(), y_two, 0)
    rest = slice(cut, n_src) if n_src > 1 else slice(0, 1)
    r0, r1 = rest.start * n_q, rest.stop * n_q
    p_two = fused(x[rest].contiguous(), idx[r0:r1].contiguous(), w[r0:r1].contiguous(), y_two, 1)
    expect = y_one if n_src > 1 else 2 * y_one
    scale = float(expect.abs().max())
    assert float((y_two - expect).abs().max()) <= 2e-6 * scale
    assert abs(p_two - float(y_two.abs().max())) <= 1e-6 * scale


def test_fused_path_random_shapes():
    """tools/stress_fused.py: random IR lengths (1 .. 300), chunk sizes 448 .. 4096, subchunks 32 .. 256, 1 .. 47 sources,
    signals up to 160 000 samples, random trajectories on the adversarial table - whichever fused kernel the plan picks
    (split roles, four waves per tile, h-only rows), direct output, slab and wide reduce, one to three tap segments -
    against the oracle's whole-signal render."""
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_fused.py"), "24", "3"], capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    # (the summary counts the cases by the kernel the plan picked for them)
    assert last.startswith("worst") and "fs_kernel" in last and "fq_kernel" in last and "fz_kernel<4,1>" in last, last


@pytest.mark.parametrize("k,s,l", [(512, 32, 128), (512, 32, 100), (256, 32, 128), (1024, 64, 128)])
def test_fast_fir_rounding_margin(tables, k, s, l):
    """The row step of both FIR kernels is a 2-parallel fast FIR (bas_fir.h: y_odd = P - A - B): its rounding must stay
    far inside the 1e-5 bar.  Low-pass noise is the unfriendly case (x_even ~ x_odd: P ~ 4 A); the direct float32 sum
    of the oracle sits at ~3e-7 of the float64 result itself, so 2e-6 is asked for here."""
    rng = np.random.default_rng(k + s + l)
    h = tables["consistent"].truncated(l)
    d = _device_table(h)
    n_src, n = 3, 20000
    white = rng.standard_normal((n_src, n + 64))
    low = np.stack([np.convolve(w, np.ones(64) / 64, mode="valid")[:n] for w in white])       # strongly low-pass
    sigs = np.ascontiguousarray(np.concatenate([low[:2], white[2:, :n] * 0.1]), dtype=np.float32)
    in_length, _ = orc.render_lengths(n, k, l)
    t = np.arange(0, in_length + 1, k, dtype=np.float64)
    elev = rng.uniform(-0.7, 1.4, size=(n_src, t.size))
    azim = rng.uniform(-3, 3, size=(n_src, t.size))
    irs = [np.stack([orc.interp2d(h, elev[i, c], azim[i, c]) for c in range(t.size)]) for i in range(n_src)]
    want = orc.render_mix(sigs, k, s, irs, normalize=False)
    for fused in (True, False):
        got = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none", fused=fused).cpu().numpy()
        assert got.shape == want.shape
        assert rel_err(got, want) <= 2e-6, (fused, rel_err(got, want))


@pytest.mark.parametrize("k,s,l", [(256, 32, 128), (320, 64, 100), (384, 128, 128), (416, 32, 200)])
def test_fused_small_chunks_h_only_rows(tables, k, s, l):
    """Chunk sizes 256 .. 447 with enough (tile, source) units run the fused kernel with h-only LDS rows
    (bas_render_fz_kernel<4, true>): against the oracle and against the stored-IR path."""
    rng = np.random.default_rng(k * 7 + s + l)
    h = tables["adversarial"].truncated(l)
    d = _device_table(h)
    n_src, n = 36, 130000
    in_length, _ = orc.render_lengths(n, k, l)
    assert bas._hip.lib().bas_render_fused_supported(n_src, in_length, k, s, l) == 1
    assert bas._hip.lib().bas_render_fused_supported(2, in_length, k, s, l) == 0          # few sources: stored-IR path
    sigs = np.stack([bas.synth.integer_noise(int(rng.integers(1e6)), n, 0.5 / n_src) for _ in range(n_src)])
    t = np.arange(0, in_length + 1, k, dtype=np.float64)
    elev = rng.uniform(-1.0, 1.7, size=(n_src, t.size))
    azim = rng.uniform(-7, 7, size=(n_src, t.size))
    fz = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none", fused=True).cpu().numpy()
    st = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none", fused=False).cpu().numpy()
    assert rel_err(fz, st) <= 2e-6
    # oracle on three sources of the scene (the whole scene would take minutes on the CPU)
    sub = [0, 17, 35]
    irs = [np.stack([orc.interp2d(h, elev[i, c], azim[i, c]) for c in range(t.size)]) for i in sub]
    want = orc.render_mix(sigs[sub], k, s, irs, normalize=False)
    got = bas.render_sources(sigs[sub], k, s, elev[sub], azim[sub], d, normalize="none").cpu().numpy()
    assert rel_err(got, want) <= REL
    # the fused render of the whole scene is linear in its sources: remove the other 33 through the stored-IR path
    rest = [i for i in range(n_src) if i not in sub]
    others = bas.render_sources(sigs[rest], k, s, elev[rest], azim[rest], d, normalize="none", fused=False).cpu().numpy()
    assert np.abs((fz - others) - want).max() <= 4e-6 * np.abs(fz).max()


def test_edge_shapes_no_sources_and_tiny_signals(tables):
    """Edges the reference handles by construction (apply_hrtf.py:405-414 pads to whole chunks): no sources at all, one sample,
    exactly one chunk, one sample more than a chunk - through both render paths."""
    import torch
    h = tables["consistent"].truncated(128)
    d = _device_table(h)
    k, s, l = 512, 32, 128
    # no sources: silence of the padded length, peak 0
    in_length, out_length = orc.render_lengths(1000, k, l)
    y = bas.render_sources(np.zeros((0, 1000), dtype=np.float32), k, s, np.zeros((0, in_length // k + 1)),
                           np.zeros((0, in_length // k + 1)), d, normalize="mix")
    assert tuple(y.shape) == (out_length, 2) or tuple(y.shape) == (2, out_length)
    assert float(torch.as_tensor(y).abs().max()) == 0.0
    rng = np.random.default_rng(5)
    for n in (1, k - 1, k, k + 1):
        in_length, _ = orc.render_lengths(n, k, l)
        t = np.arange(0, in_length + 1, k, dtype=np.float64)
        sigs = rng.standard_normal((2, n)).astype(np.float32)
        elev = rng.uniform(-0.7, 1.4, size=(2, t.size))
        azim = rng.uniform(-3, 3, size=(2, t.size))
        irs = [np.stack([orc.interp2d(h, elev[i, c], azim[i, c]) for c in range(t.size)]) for i in range(2)]
        want = orc.render_mix(sigs, k, s, irs, normalize=False)
        for fused in (True, False):
            got = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none", fused=fused).cpu().numpy()
            assert got.shape == want.shape and rel_err(got, want) <= REL, (n, fused)
