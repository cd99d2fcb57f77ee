"""GPU tests added in round 4: the kernel tails (max|y| and the peak rule apply_hrtf.py:462-464 inside the last kernel of a
render, bas_tail.h), the one-launch multi-GPU combine (bas_mix_finish_f32), the two halves of the fused render as entry
points of their own, the device-side error record (bas_render_status) with a fault-injection build, the table builder's
output through the product loader and interpolate_2d."""
import ctypes
import os

import numpy as np
import pytest

from conftest import rel_err
from oracle import bas_oracle as orc
import binaural_audio_synthesis_amd as bas

pytestmark = pytest.mark.gpu
REL = 1e-5


def _scene(l, n_src, n, k, gain, seed=0):
    h = bas.synth.make_table("consistent", 0).truncated(l)
    sigs = np.stack([bas.synth.integer_noise(seed + 400 + i, n, gain) for i in range(n_src)])
    in_length = -(-n // k) * k
    t = np.arange(0, in_length + 1, k, dtype=np.float64)
    elev = np.empty((n_src, t.size))
    azim = np.empty((n_src, t.size))
    for i in range(n_src):
        name = ("spiral", "circle_askew", "passing")[i % 3]
        elev[i], azim[i] = bas.synth.trajectory(name, period_s=0.07 + 0.013 * i, length_s=n / 44100, turns=1.0 + i % 5,
                                                phase=0.41 * i)(t)
    return h, sigs, elev, azim, in_length


def _device_inputs(h, sigs, elev, azim, in_length):
    import torch
    d = bas.irs_and_delaydiffs(h.upsampling, h.diffs_left, h.diffs_right, h.irs_left, h.irs_right)
    n_src, n = sigs.shape
    x = torch.zeros((n_src, in_length), dtype=torch.float32, device="cuda")
    x[:, :n] = torch.from_numpy(sigs).cuda()
    idx, w = bas.sphere.interpolation_params_device(torch.from_numpy(elev).cuda(), torch.from_numpy(azim).cuda())
    return d, x, idx.reshape(-1, 4).contiguous(), w.reshape(-1, 3).contiguous()


@pytest.mark.parametrize("n_src,n,k,s,l,kernel", [
    (1, 60000, 512, 32, 128, "bas_render_fq_kernel"),         # one source: the FIR kernel writes y itself and ends in the tail
    (2, 441000, 512, 32, 128, "bas_render_fq_kernel"),        # slabs + reduce, 216 x 2 units
    (1, 3300000, 512, 32, 100, "bas_render_fs_kernel<104>"),  # one long source on the split-role kernel: direct output
    (40, 200000, 512, 32, 128, "bas_render_fs_kernel<128>"),  # slabs + the plain reduce kernel
    (300, 2048, 512, 32, 128, None),                          # many sources on a short block: the wide reduce kernel
    (3, 9000, 1024, 64, 100, None),
])
@pytest.mark.parametrize("loud", [True, False])
def test_peak_rule_in_the_kernel_tail(n_src, n, k, s, l, kernel, loud):
    """normalize = 1 (the rule applied by the late workgroups of the last kernel) against normalize = 0 followed by
    bas_scale_by_peak_f32 (round 3's separate launch): same bytes, same peak - with the rule firing (peak 2.5) and not
    firing (0.4); three calls on ONE workspace (the control block resets itself); the peak against torch's own max."""
    import torch
    _hip = bas._hip
    lib = _hip.lib()
    h, sigs, elev, azim, in_length = _scene(l, n_src, n, k, 1.0 / max(n_src, 1) ** 0.5)
    d, x, idx, w = _device_inputs(h, sigs, elev, azim, in_length)
    assert lib.bas_render_fused_supported(n_src, in_length, k, s, l) == 1
    if kernel is not None:
        assert lib.bas_render_fused_kernel_name(n_src, in_length, k, s, l).decode() == kernel
    ws = _hip.new_workspace(lib.bas_render_fused_workspace_bytes(n_src, in_length, k, s, l), "cuda")
    probe, _ = bas.apply_hrtf.render_params_device(x, k, s, d, idx, w, normalize="none", ws=ws)
    x *= (2.5 if loud else 0.4) / float(probe.abs().max())    # the render is linear in x
    plain, pk_plain = bas.apply_hrtf.render_params_device(x, k, s, d, idx, w, normalize="none", ws=ws)
    plain = plain.clone()
    want_peak = float(plain.abs().max())
    assert float(pk_plain) == want_peak
    assert (want_peak > 1.0) == loud, want_peak               # the case is what it claims to be
    want = plain.clone()
    _hip.call("bas_scale_by_peak_f32", _hip.ptr(want), want.numel(), _hip.ptr(pk_plain), _hip.current_stream(x.device))
    for _ in range(3):
        got, pk = bas.apply_hrtf.render_params_device(x, k, s, d, idx, w, normalize="mix", ws=ws)
        assert float(pk) == want_peak
        assert torch.equal(got, want)
    assert not ws[:8].any()                                   # ticket pair back at zero
    _hip.check_status(ws, x.device)
    if loud:
        assert float(got.abs().max()) <= 1.0 + 1e-6
    # unaligned y: the rule falls back to the scale launch, same result
    buf = torch.empty((2 * (in_length + l - 1) + 1,), dtype=torch.float32, device="cuda")
    y_odd = buf[1:].view(2, -1)
    got2, pk2 = bas.apply_hrtf.render_params_device(x, k, s, d, idx, w, normalize="mix", ws=ws, out=y_odd)
    assert float(pk2) == want_peak and torch.equal(got2, want)


def test_tail_under_graph_replay():
    """The render with the rule in its tail captured into a hipGraph and replayed: the control block must be back at zero
    after every replay."""
    import torch
    _hip = bas._hip
    n_src, n, k, s, l = 6, 50000, 512, 32, 128
    h, sigs, elev, azim, in_length = _scene(l, n_src, n, k, 1.5)
    d, x, idx, w = _device_inputs(h, sigs, elev, azim, in_length)
    lib = _hip.lib()
    ws = _hip.new_workspace(lib.bas_render_fused_workspace_bytes(n_src, in_length, k, s, l), "cuda")
    wsp = torch.empty((lib.bas_interp2d_workspace_bytes(idx.shape[0]),), dtype=torch.uint8, device="cuda")
    y = torch.empty((2, in_length + l - 1), dtype=torch.float32, device="cuda")
    want, pk = bas.apply_hrtf.render_params_device(x, k, s, d, idx, w, normalize="mix", ws=ws, ws_plans=wsp)
    want = want.clone()
    assert float(pk) > 1.0
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        bas.apply_hrtf.render_params_device(x, k, s, d, idx, w, normalize="mix", ws=ws, ws_plans=wsp, out=y)
    for _ in range(4):
        y.fill_(9.0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, want)
        assert not ws[:8].any()


@pytest.mark.parametrize("n_parts,n,loud", [(8, 2 * 441471, True), (8, 2 * 441471, False), (3, 10007, True), (1, 5, True)])
def test_mix_finish_equals_sum_then_scale(n_parts, n, loud):
    """bas_mix_finish_f32 (fixed-order sum + max|y| + peak rule, one launch) against bas_mix_partials_f32 +
    bas_scale_by_peak_f32, bit for bit; unaligned part strides take the scalar path."""
    import torch
    _hip = bas._hip
    gen = torch.Generator(device="cuda").manual_seed(n_parts + n)
    stride = n + (3 if n % 4 else 0)
    parts = (torch.rand((n_parts, stride), generator=gen, device="cuda") - 0.5) * (1.0 if loud else 0.1)
    st = _hip.current_stream(parts.device)
    want = torch.empty((n,), dtype=torch.float32, device="cuda")
    pk0 = torch.empty((1,), dtype=torch.float32, device="cuda")
    _hip.call("bas_mix_partials_f32", _hip.ptr(parts), n_parts, stride, n, _hip.ptr(want), _hip.ptr(pk0), st)
    raw = want.clone()
    _hip.call("bas_scale_by_peak_f32", _hip.ptr(want), n, _hip.ptr(pk0), st)
    ws = _hip.new_workspace(_hip.lib().bas_mix_workspace_bytes(), "cuda")
    for normalize, ref in ((1, want), (0, raw), (1, want)):
        got = torch.full((n,), 5.0, dtype=torch.float32, device="cuda")
        pk = torch.empty((1,), dtype=torch.float32, device="cuda")
        _hip.call("bas_mix_finish_f32", _hip.ptr(parts), n_parts, stride, n, _hip.ptr(got), _hip.ptr(pk), normalize,
                  _hip.ptr(ws), ws.numel(), st)
        assert float(pk) == float(pk0) and torch.equal(got, ref)
    assert not ws[:8].any()


@pytest.mark.parametrize("n_src,n", [(12, 120000), (1, 60000)])
def test_fir_and_reduce_as_separate_entry_points(n_src, n):
    """bas_render_fused_fir_f32 + bas_render_fused_reduce_f32 back to back == bas_render_mix_fused_f32 (bit for bit), with
    other work of the caller enqueued between the two halves."""
    import torch
    _hip = bas._hip
    lib = _hip.lib()
    k, s, l = 512, 32, 128
    h, sigs, elev, azim, in_length = _scene(l, n_src, n, k, 2.0 / n_src)
    d, x, idx, w = _device_inputs(h, sigs, elev, azim, in_length)
    t_out = in_length + l - 1
    ws = _hip.new_workspace(lib.bas_render_fused_workspace_bytes(n_src, in_length, k, s, l), "cuda")
    plans = torch.empty((lib.bas_interp2d_workspace_bytes(idx.shape[0]),), dtype=torch.uint8, device="cuda")
    st = _hip.current_stream(x.device)
    _hip.call("bas_interp2d_plan_f32", _hip.ptr(d.diffs), _hip.ptr(idx), _hip.ptr(w), idx.shape[0], d.ndir, l, d.upsampling,
              _hip.ptr(plans), plans.numel(), st)

    def args(y, peak):
        return (_hip.ptr(x), x.stride(0), _hip.ptr(d.packed), _hip.ptr(plans), n_src, in_length, k, s, l, d.upsampling, d.ndir,
                _hip.ptr(y), 0, _hip.ptr(peak), 1, _hip.ptr(ws), ws.numel(), st)
    y1 = torch.empty((2, t_out), dtype=torch.float32, device="cuda")
    p1 = torch.empty((1,), dtype=torch.float32, device="cuda")
    _hip.call("bas_render_mix_fused_f32", *args(y1, p1))
    y2 = torch.full((2, t_out), 3.0, dtype=torch.float32, device="cuda")
    p2 = torch.empty((1,), dtype=torch.float32, device="cuda")
    _hip.call("bas_render_fused_fir_f32", *args(y2, p2))
    other = torch.arange(1000, device="cuda").sum()           # the caller's own work between the halves
    _hip.call("bas_render_fused_reduce_f32", *args(y2, p2))
    assert int(other) == 499500
    assert torch.equal(y1, y2) and float(p1) == float(p2)
    _hip.check_status(ws, x.device)


@pytest.mark.parametrize("n_src,n,nw,split", [(3, 30000, "4", "1"), (3, 30000, "4", "0"), (1, 9000, None, None)])
def test_handover_timeout_is_reported_not_rendered(monkeypatch, n_src, n, nw, split):
    """Fault injection (diagnostic build, BAS_DEBUG_FLAGS=256: the stager waves never publish their boundary chunk IR):
    the neighbours' bounded wait runs out, the affected outputs are NaN instead of plausible audio and
    bas_render_status returns a positive code with a text - for the split-role, the two-per-CU and the four-wave kernel.
    The same scene without the injection is clean."""
    import torch
    _hip = bas._hip
    k, s, l = 512, 32, 128
    h, sigs, elev, azim, in_length = _scene(l, n_src, n, k, 0.1)
    if nw is not None:
        monkeypatch.setenv("BAS_FZ_NW", nw)
        monkeypatch.setenv("BAS_FZ_SPLIT", split)
    with _hip.use_library(_hip.DIAG_LIB_PATH) as lib:
        d, x, idx, w = _device_inputs(h, sigs, elev, azim, in_length)
        ws = _hip.new_workspace(lib.bas_render_fused_workspace_bytes(n_src, in_length, k, s, l), "cuda")
        good, _ = bas.apply_hrtf.render_params_device(x, k, s, d, idx, w, normalize="none", ws=ws, fused=True)
        good = good.clone()
        _hip.check_status(ws, x.device)                       # nothing to report
        monkeypatch.setenv("BAS_DEBUG_FLAGS", "256")
        bad, _ = bas.apply_hrtf.render_params_device(x, k, s, d, idx, w, normalize="none", ws=ws, fused=True)
        torch.cuda.synchronize()
        assert torch.isnan(bad).any() and not torch.isnan(good).any()
        with pytest.raises(_hip.BasError) as err:
            _hip.check_status(ws, x.device)
        assert err.value.code > 0 and "hand-over" in str(err.value)
        _hip.check_status(ws, x.device)                       # the record is cleared by reading it
        monkeypatch.delenv("BAS_DEBUG_FLAGS")
        again, _ = bas.apply_hrtf.render_params_device(x, k, s, d, idx, w, normalize="none", ws=ws, fused=True)
        assert torch.equal(again, good)                       # the workspace is usable again


def test_built_table_through_the_product_loader(tmp_path):
    """SURVEY 8f-2 (parity unpinned: nothing in the reference can pin upsample_irs.m:15-54 here): a 187-direction table
    built by upsample_irs.py from synthetic pulse IRs round-trips through the .mat layout of upsample_irs.m:46-53 into
    load_irs_and_delaydiffs -> interpolate_2d on the device, and agrees with the oracle fed the same file; the delay
    differences are antisymmetric exactly (upsample_irs.m:31-32) and obey the mesh rule d[i,j] + d[j,k] = d[i,k] that
    apply_hrtf.py:246-252 relies on to the accuracy of the sub-sample peak estimate."""
    import scipy.io
    from binaural_audio_synthesis_amd import upsample_irs as up
    rng = np.random.default_rng(31)
    n_dir, n_taps, u, keep = 187, 64, 8, 48
    pos = 18 + rng.uniform(-5, 5, size=(2, n_dir))

    def pulse(p, width):
        t = np.arange(n_taps)
        return np.exp(-0.5 * ((t - p) / width) ** 2) * (1.0 + 0.2 * np.cos(0.9 * (t - p)))
    hl = np.stack([pulse(p, 2.5) for p in pos[0]])
    hr = np.stack([pulse(p, 3.0) for p in pos[1]])
    t = up.upsample_irs(hl, hr, u)
    path = str(tmp_path / "built.mat")
    up.save(path, t)
    rec = scipy.io.loadmat(path)["irs_and_delaydiffs"][0][0]
    for name in ("diffs_left", "diffs_right"):
        dd = rec[name]
        assert np.abs(dd + dd.T).max() <= 1e-12 and not np.diag(dd).any()
        i, j, kk = rng.integers(0, n_dir, size=(3, 2000))
        assert np.abs(dd[i, j] + dd[j, kk] - dd[i, kk]).max() <= 1e-3
    dev_tbl = bas.load_irs_and_delaydiffs(path, samples_to_keep=keep)
    assert dev_tbl.upsampling == u and dev_tbl.L == keep and dev_tbl.ndir == n_dir

    class HostTable:                                          # what the reference's loader returns (apply_hrtf.py:36-44)
        upsampling = int(rec["upsampling"][0][0])
        diffs_left, diffs_right = rec["diffs_left"], rec["diffs_right"]
        irs_left, irs_right = rec["irs_left"][:, :keep * u], rec["irs_right"][:, :keep * u]
    worst = 0.0
    for elev, azim in [(0.0, 0.3), (0.5, 2.0), (-0.7, 5.5), (1.2, 1.0), (1.5707, 0.2), (0.2617993877991494, 3.0), (-2.0, -1.0)]:
        want = orc.interp2d(HostTable, np.float64(elev), np.float64(azim))
        got = bas.interpolate_2d(dev_tbl, np.float64(elev), np.float64(azim))
        worst = max(worst, rel_err(got, want))
    assert worst <= REL, worst


def _oracle_mix(h, sigs, elev, azim, k, s):
    irs = [np.stack([orc.interp2d(h, elev[i, c], azim[i, c]) for c in range(elev.shape[1])]) for i in range(sigs.shape[0])]
    return orc.render_mix(sigs, k, s, irs, normalize=False)


@pytest.mark.parametrize("s", [16, 8])
@pytest.mark.parametrize("n_src,n,k,l", [
    (3, 30000, 512, 128),          # one unit per workgroup
    (40, 60000, 512, 128),         # two units per workgroup through the two LDS buffers
    (5, 40000, 512, 100),          # the unit block of 104 taps (the reference's default IR length)
    (2, 50000, 1024, 128),         # longer chunks: 64 / 128 subchunks per chunk
    (4, 30000, 576, 121),          # chunk size not a power of two, L rounded up to 128
    (1, 60000, 512, 128),          # one source: direct output
])
def test_fused_kernel_with_subchunks_of_16_and_8_vs_oracle(monkeypatch, n_src, n, k, l, s):
    """Subchunks of 16 / 8 samples (the reference accepts any divisor of the chunk size, apply_hrtf.py:401-402; :442-443)
    inside the split-role kernel: a row of 32 inputs meets two / four crossfaded tap sets (ffa_unit2_asm, ffa_unit4_asm).
    Small scenes through the diagnostic build (which gives the kernel to any scene on request) against the oracle and
    against the stored-IR path."""
    h, sigs, elev, azim, in_length = _scene(l, n_src, n, k, 0.3 / n_src)
    want = _oracle_mix(h, sigs, elev, azim, k, s)
    monkeypatch.setenv("BAS_FZ_NW", "4")
    monkeypatch.setenv("BAS_FZ_SPLIT", "1")
    with bas._hip.use_library(bas._hip.DIAG_LIB_PATH) as lib:
        d = bas.irs_and_delaydiffs(h.upsampling, h.diffs_left, h.diffs_right, h.irs_left, h.irs_right)
        assert lib.bas_render_fused_supported(n_src, in_length, k, s, l) == 1
        assert lib.bas_render_fused_kernel_name(n_src, in_length, k, s, l).decode() == \
            f"bas_render_fs_kernel<{128 if l > 104 else 104},{32 // s}>"
        got = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none", fused=True).cpu().numpy()
        stored = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none", fused=False).cpu().numpy()
    assert got.shape == want.shape and rel_err(got, want) <= REL, rel_err(got, want)
    assert rel_err(got, stored) <= 5e-6, rel_err(got, stored)


@pytest.mark.parametrize("s", [16, 8])
def test_shipped_library_fuses_small_subchunks_for_big_scenes(s):
    """BASELINE config 4's shape with subchunksize 16 / 8: served by the fused path of the SHIPPED library (round 3: stored
    chunk IRs, 0.75 ms per step); a 48-source scene rendered through it against the stored-IR path and oracle windows."""
    import torch
    lib = bas._hip.lib()
    assert os.path.basename(lib._name) == "libbas_hip.so"
    assert lib.bas_render_fused_supported(256, 441344, 512, s, 128) == 1
    assert lib.bas_render_fused_kernel_name(256, 441344, 512, s, 128).decode() == f"bas_render_fs_kernel<128,{32 // s}>"
    n_src, n, k, l = 48, 140000, 512, 128
    h, sigs, elev, azim, in_length = _scene(l, n_src, n, k, 0.5 / n_src)
    assert lib.bas_render_fused_kernel_name(n_src, in_length, k, s, l).decode() == f"bas_render_fs_kernel<128,{32 // s}>"
    d = bas.irs_and_delaydiffs(h.upsampling, h.diffs_left, h.diffs_right, h.irs_left, h.irs_right)
    got = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none").cpu().numpy()
    stored = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none", fused=False).cpu().numpy()
    assert rel_err(got, stored) <= 5e-6, rel_err(got, stored)
    scale = np.abs(got).max()
    worst = 0.0
    for n0, n1 in [(0, 40), (8192 - 20, 8192 + 20), (100 * k - 8, 100 * k + 24), (in_length + l - 1 - 40, in_length + l - 1)]:
        m0, m1 = max(n0 - l + 1, 0), min(n1, n)
        want = np.zeros((2, n1 - n0))
        for i in range(n_src):
            cache = {}

            def ir_of(c, i=i, cache=cache):
                if c not in cache:
                    cache[c] = orc.interp2d(h, elev[i, c], azim[i, c])
                return cache[c]
            want += orc.render_window(sigs[i, m0:m1].astype(np.float64), m0, k, s, ir_of, l, n0, n1)
        worst = max(worst, float(np.abs(got[n0:n1].T - want).max()) / scale)
    assert worst <= REL, worst


def test_plans_on_another_stream_then_the_fir_half():
    """plan_angles_device on a second stream + render_angles_device(plans_ready=True) behind an event (what
    bench.py --overlap-plans on does per step) == the one-stream render, bit for bit; the errors of the two halves."""
    import torch
    h, sigs, elev, azim, in_length = _scene(128, 12, 120000, 512, 0.3)
    d, x, _, _ = _device_inputs(h, sigs, elev, azim, in_length)
    e, z = torch.from_numpy(elev).cuda(), torch.from_numpy(azim).cuda()
    want, peak = bas.apply_hrtf.render_angles_device(x, 512, 32, d, e, z, normalize="mix")
    plans = torch.empty((bas._hip.lib().bas_interp2d_workspace_bytes(e.numel()),), dtype=torch.uint8, device="cuda")
    side, done = torch.cuda.Stream(), torch.cuda.Event()
    with torch.cuda.stream(side):
        bas.apply_hrtf.plan_angles_device(d, e, z, plans)
        done.record(side)
    torch.cuda.current_stream().wait_event(done)
    got, peak2 = bas.apply_hrtf.render_angles_device(x, 512, 32, d, e, z, normalize="mix", ws_plans=plans, plans_ready=True)
    torch.cuda.synchronize()
    assert torch.equal(got, want) and torch.equal(peak, peak2)
    with pytest.raises(ValueError, match="smaller than"):
        bas.apply_hrtf.plan_angles_device(d, e, z, plans[:64])
    with pytest.raises(ValueError, match="float64"):
        bas.apply_hrtf.plan_angles_device(d, e.float(), z.float(), plans)
    with pytest.raises(ValueError, match="plans_ready"):       # K = 96: stored chunk IRs, nothing reads plans
        bas.apply_hrtf.render_angles_device(x[:, :96 * 100], 96, 32, d, e[:, :101].contiguous(), z[:, :101].contiguous(),
                                            ws_plans=plans, plans_ready=True)


@pytest.mark.parametrize("n_src,k,blocks", [
    (256, 512, (512, 512, 1024, 512)),        # many sources on short blocks: the wide reduce kernel carries the state
    (40, 512, (16384, 8192, 16384)),          # the plain reduce kernel
    (1, 512, (4096, 2048)),                   # one source: the FIR kernel writes y itself, the epilogue kernel is launched
    (5, 128, (128, 256, 128)),                # chunk size the fused kernels do not serve: render + epilogue as before
])
def test_stream_block_in_one_call_equals_render_plus_epilogue(n_src, k, blocks):
    """bas_render_stream_block_f32 (the carried state and the running peak in the reduce kernel's tail) against the
    two-call form (render, then bas_stream_epilogue_f32): the same emitted samples, bit for bit, block after block (every
    block depends on the state the one before carried over), the same running peak and the same tail from finish(); and
    against the whole-signal render."""
    import torch
    l, s = 128, 32
    n = sum(blocks)
    h = bas.synth.make_table("consistent", 0).truncated(l)
    d = bas.irs_and_delaydiffs(h.upsampling, h.diffs_left, h.diffs_right, h.irs_left, h.irs_right)
    sigs = np.stack([bas.synth.integer_noise(900 + i, n, 0.5 / max(n_src, 1) ** 0.5) for i in range(n_src)])
    t = np.arange(0, n + 1, k, dtype=np.float64)
    elev = np.empty((n_src, t.size))
    azim = np.empty((n_src, t.size))
    for i in range(n_src):
        elev[i], azim[i] = bas.synth.trajectory(("spiral", "circle_askew", "passing")[i % 3], period_s=0.05 + 0.003 * i,
                                                length_s=n / 44100, turns=2.0, phase=0.3 * i)(t)
    outs = {}
    for one_call in (True, False):
        st = bas.StreamRenderer(d, n_src, k, s, graph=False)
        st.one_call = one_call
        got, pos = [], 0
        for b in blocks:
            c0, c1 = pos // k, (pos + b) // k
            got.append(st.process(sigs[:, pos:pos + b], elev[:, c0:c1 + 1], azim[:, c0:c1 + 1]))
            pos += b
        got.append(st.finish())
        outs[one_call] = (torch.cat(got, dim=0), st.peak)
    assert torch.equal(outs[True][0], outs[False][0])
    assert outs[True][1] == outs[False][1]
    whole = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none")
    assert rel_err(outs[True][0].cpu().numpy(), whole.cpu().numpy()) <= 1e-6
    assert abs(outs[True][1] - float(whole.abs().max())) <= 1e-6 * outs[True][1]


def test_stream_block_under_graph_replay_carries_state():
    """The one-call block captured into the stream's hipGraph (prepare()): replays concatenate to the whole-signal render."""
    import torch
    n_src, k, s, l, b, nblk = 64, 512, 32, 128, 1024, 6
    n = b * nblk
    h = bas.synth.make_table("consistent", 0).truncated(l)
    d = bas.irs_and_delaydiffs(h.upsampling, h.diffs_left, h.diffs_right, h.irs_left, h.irs_right)
    sigs = np.stack([bas.synth.integer_noise(950 + i, n, 0.05) for i in range(n_src)])
    t = np.arange(0, n + 1, k, dtype=np.float64)
    elev = np.empty((n_src, t.size))
    azim = np.empty((n_src, t.size))
    for i in range(n_src):
        elev[i], azim[i] = bas.synth.trajectory("spiral", period_s=0.05 + 0.003 * i, length_s=n / 44100, turns=2.0, phase=0.3 * i)(t)
    st = bas.StreamRenderer(d, n_src, k, s, graph=True)
    st.prepare(b)
    assert st._graph is not None
    got = []
    for i in range(nblk):
        c0, c1 = i * b // k, (i + 1) * b // k
        got.append(st.process(sigs[:, i * b:(i + 1) * b], elev[:, c0:c1 + 1], azim[:, c0:c1 + 1]))
    got.append(st.finish())
    whole = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none")
    assert rel_err(torch.cat(got, dim=0).cpu().numpy(), whole.cpu().numpy()) <= 1e-6
    assert abs(st.peak - float(whole.abs().max())) <= 1e-6 * st.peak


@pytest.mark.parametrize("branch", ["f64", "pyfloat"])
def test_ring_search_at_and_around_every_node(branch):
    """a3's search for the last node <= azim (sphere.py:103) starts from the azimuth's position on an evenly spaced ring and
    walks to the exact node: azimuths AT every node of every ring (the float32 table value as a double), one float64 and one
    float32 step to either side of it, just below 2 pi, negative and beyond 2 pi - device == host, bit for bit, both
    branches of the comparison (float64 azimuth against the float32 node / float32 against float32)."""
    import torch
    sp = bas.sphere
    tab = np.asarray(sp.index_elev_azim)
    elevs, azims = [], []
    for ring, (start, count) in enumerate(zip(sp.RING_START, sp.RING_COUNTS)):
        e_ring = float(tab[start, 1])
        for e in (e_ring, e_ring - 0.03, e_ring + 0.02):
            for v in tab[start:start + count, 2]:
                v64 = float(v)
                cands = [v64, np.nextafter(v64, -1.0), np.nextafter(v64, 10.0), float(np.nextafter(v, np.float32(-1))),
                         float(np.nextafter(v, np.float32(10))), v64 + 2 * np.pi, v64 - 2 * np.pi, v64 + 1e-9, v64 - 1e-9]
                elevs += [e] * len(cands)
                azims += cands
            for z in (2 * np.pi, np.nextafter(2 * np.pi, 0.0), float(np.float32(2 * np.pi)), -1e-12, 0.0, -0.0, 4 * np.pi - 1e-9):
                elevs.append(e)
                azims.append(z)
    e = np.array(elevs, dtype=np.float64)
    z = np.array(azims, dtype=np.float64)
    idx_h, w_h = sp.interpolation_params_batch(e, z, branch=branch)
    idx_d, w_d = sp.interpolation_params_device(torch.from_numpy(e).cuda(), torch.from_numpy(z).cuda(), branch=branch)
    assert np.array_equal(idx_d.cpu().numpy().reshape(idx_h.shape), idx_h)
    assert np.array_equal(w_d.cpu().numpy().reshape(w_h.shape), w_h)
    assert e.size > 5000


@pytest.mark.parametrize("l,s", [(256, 32), (250, 32), (384, 32), (512, 32), (256, 16), (505, 8)])
def test_unit_blocks_for_irs_of_several_segments(l, s):
    """IR lengths of several whole 128-tap segments (L = 249..256, 377..384, 505..512 - 512 is the default samples_to_keep
    of the reference's loader, apply_hrtf.py:23): every (unit, segment) pass of the split-role kernel is one unit block
    (round 4 until late: the per-step blocks, 4-8 % slower).  The shipped library on a big scene against the stored-IR path
    and oracle windows; subchunks of 16 / 8 (two / four tap sets per row) likewise."""
    lib = bas._hip.lib()
    assert os.path.basename(lib._name) == "libbas_hip.so"
    n_src, n, k = 40, 100000, 512
    h, sigs, elev, azim, in_length = _scene(l, n_src, n, k, 0.5 / n_src)
    want_name = "bas_render_fs_kernel<128>" if s == 32 else f"bas_render_fs_kernel<128,{32 // s}>"
    assert lib.bas_render_fused_kernel_name(n_src, in_length, k, s, l).decode() == want_name
    assert lib.bas_render_fused_kernel_name(256, 441344, 512, s, l).decode() == want_name
    assert lib.bas_render_fused_kernel_name(n_src, in_length, k, 32, l - 16).decode() == "bas_render_fs_kernel<0>"   # (a short last segment)
    assert lib.bas_render_fused_supported(n_src, in_length, k, 16, l - 16) == 0                                      # (and no second tap set there)
    d = bas.irs_and_delaydiffs(h.upsampling, h.diffs_left, h.diffs_right, h.irs_left, h.irs_right)
    got = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none").cpu().numpy()
    stored = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none", fused=False).cpu().numpy()
    assert rel_err(got, stored) <= 5e-6, rel_err(got, stored)
    scale = np.abs(got).max()
    worst = 0.0
    for n0, n1 in [(0, 40), (8192 - 20, 8192 + 20), (100 * k - 8, 100 * k + 24), (in_length + l - 1 - 40, in_length + l - 1)]:
        m0, m1 = max(n0 - l + 1, 0), min(n1, n)
        want = np.zeros((2, n1 - n0))
        for i in range(n_src):
            cache = {}

            def ir_of(c, i=i, cache=cache):
                if c not in cache:
                    cache[c] = orc.interp2d(h, elev[i, c], azim[i, c])
                return cache[c]
            want += orc.render_window(sigs[i, m0:m1].astype(np.float64), m0, k, s, ir_of, l, n0, n1)
        worst = max(worst, float(np.abs(got[n0:n1].T - want).max()) / scale)
    assert worst <= REL, worst


def test_mix_finish_of_unaligned_parts_many_times():
    """Eight gathered parts of 2 x 441 471 floats (every second one 8 bytes off a 16-byte boundary: the headline scene's
    mix) summed with 16-byte loads and stored by the inline-assembly sc1 store, 50 launches, rule firing and not: every one
    equals sum + scale bit for bit.  (The store reads its data registers up to two wait states after it issues; without the
    s_nop behind it the |y| of the peak search, computed in place, reached memory for a few lanes in one launch of seven.)"""
    import torch
    _hip = bas._hip
    n_parts, n = 8, 2 * 441471
    gen = torch.Generator(device="cuda").manual_seed(77)
    stride = n + 3
    parts = torch.rand((n_parts, stride), generator=gen, device="cuda") - 0.5
    st = _hip.current_stream(parts.device)
    want = torch.empty((n,), dtype=torch.float32, device="cuda")
    pk0 = torch.empty((1,), dtype=torch.float32, device="cuda")
    _hip.call("bas_mix_partials_f32", _hip.ptr(parts), n_parts, stride, n, _hip.ptr(want), _hip.ptr(pk0), st)
    raw = want.clone()
    _hip.call("bas_scale_by_peak_f32", _hip.ptr(want), n, _hip.ptr(pk0), st)
    assert float(pk0) > 1.0
    ws = _hip.new_workspace(_hip.lib().bas_mix_workspace_bytes(), "cuda")
    got = torch.empty((n,), dtype=torch.float32, device="cuda")
    pk = torch.empty((1,), dtype=torch.float32, device="cuda")
    for it in range(50):
        normalize = it & 1
        got.fill_(5.0)
        _hip.call("bas_mix_finish_f32", _hip.ptr(parts), n_parts, stride, n, _hip.ptr(got), _hip.ptr(pk), normalize,
                  _hip.ptr(ws), ws.numel(), st)
        assert torch.equal(got, want if normalize else raw), (it, int((got != (want if normalize else raw)).sum()))
        assert float(pk) == float(pk0)
