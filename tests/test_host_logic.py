"""CPU tests (no GPU): host-side geometry against the reference-generated goldens, the
C-ABI library's exports, and argument checking that happens before any launch."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import golden, ROOT
import binaural_audio_synthesis_amd as bas

sphere = bas.sphere


def test_index_table_layout():
    t = sphere.index_elev_azim
    assert t.shape == (187, 3) and t.dtype == np.float32
    assert np.array_equal(t[:, 0], np.arange(187, dtype=np.float32))
    assert t[73, 2] == np.float32(15) * np.float32(2 * np.pi / 360)
    assert t[168, 1] == np.float32(60) * np.float32(2 * np.pi / 360) and t[186, 1] == np.float32(90) * np.float32(2 * np.pi / 360)


@pytest.mark.parametrize("kind", ["pyfloat", "f64"])
def test_scalar_params_match_reference(kind):
    g = golden("azim_params.npz")
    conv = float if kind == "pyfloat" else np.float64
    for i, (e, z) in enumerate(zip(g["elev"], g["azim"])):
        b, a, af = sphere.azim_to_interpolation_params(np.float64(e), conv(z))
        assert (b, af) == (g[f"before_{kind}"][i], g[f"after_{kind}"][i])
        assert float(a) == g[f"a_{kind}"][i]
        assert isinstance(a, np.float32) == bool(g[f"a_is_f32_{kind}"][i])


def test_invalid_ring_is_value_error():
    with pytest.raises(ValueError):
        sphere.azim_to_interpolation_params(np.float64(0.1), np.float64(1.0))


def test_batch_params_equal_scalar_float64_branch():
    g = golden("interp2d.npz")
    rng = np.random.default_rng(5)
    e = np.concatenate([g["points"][:, 0], rng.uniform(-1.3, 1.9, 3000)])
    z = np.concatenate([g["points"][:, 1], rng.uniform(-9, 18, 3000)])
    idx, w = sphere.interpolation_params_batch(e, z)
    assert idx.dtype == np.int32 and w.dtype == np.float64
    for i in range(e.size):
        i4, w3 = sphere.interpolation_params(np.float64(e[i]), np.float64(z[i]))
        assert tuple(idx[i]) == tuple(i4) and tuple(w[i]) == tuple(w3), i
    # shape handling
    idx2, w2 = sphere.interpolation_params_batch(e[:6].reshape(2, 3), z[:6].reshape(2, 3))
    assert idx2.shape == (2, 3, 4) and w2.shape == (2, 3, 3)
    with pytest.raises(ValueError):
        sphere.interpolation_params_batch(np.array([np.nan]), np.array([0.0]))


def test_library_exports_every_declared_symbol():
    """include/bas.h and the built .so agree (no compute call: there is no GPU here)."""
    hdr = open(os.path.join(ROOT, "include", "bas.h")).read()
    declared = set(re.findall(r"\b(bas_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(bas._hip.SIGNATURES), declared ^ set(bas._hip.SIGNATURES)
    lib = bas._hip.lib()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.bas_version() == bas._hip.ABI_VERSION
    assert lib.bas_last_error() == b""


def test_argument_errors_are_reported_without_a_gpu():
    lib = bas._hip.lib()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.addressof(buf)
    # K % S != 0  -> BAS_E_SHAPE, message mirrors the reference's assert text (apply_hrtf.py:402)
    rc = lib.bas_render_mix_f32(p, 1024, p, 1, 1024, 512, 48, 128, p, 0, None, p, 64, None)
    assert rc == -2 and b"does not divide chunksize evenly" in lib.bas_last_error()
    rc = lib.bas_render_mix_f32(p, 1000, p, 1, 1000, 512, 32, 128, p, 0, None, p, 64, None)
    assert rc == -2 and b"multiple of K" in lib.bas_last_error()
    rc = lib.bas_render_mix_f32(None, 1024, None, 1, 1024, 512, 32, 128, None, 0, None, None, 0, None)
    assert rc == -1
    rc = lib.bas_interp2d_f32(None, None, None, None, 1, 187, 128, 8, None, None, 0, None)
    assert rc == -1
    rc = lib.bas_table_pack_f32(p, 187, 1001, 8, p, None)
    assert rc == -2
    assert lib.bas_render_workspace_bytes(256, 441344, 512, 32, 128) > 16


def test_synthetic_tables_roundtrip_through_mat(tmp_path, tables):
    import scipy.io
    t = tables["consistent"]
    p = str(tmp_path / "t.mat")
    bas.synth.save_table_mat(p, t)
    rec = scipy.io.loadmat(p)["irs_and_delaydiffs"][0][0]       # the indexing of apply_hrtf.py:38-44
    assert int(rec["upsampling"][0][0]) == 8
    assert np.array_equal(rec["irs_left"], t.irs_left) and np.array_equal(rec["diffs_right"], t.diffs_right)
    assert np.array_equal(t.irs_left.astype(np.float32).astype(np.float64), t.irs_left)   # lossless in float32
    assert np.allclose(t.diffs_left, -t.diffs_left.T) and np.all(np.diag(t.diffs_left) == 0)


def test_no_product_module_imports_the_oracle():
    pkg = os.path.join(ROOT, "binaural-audio-synthesis_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), f


def test_diagnostic_build_is_separate_and_only_it_reads_the_environment():
    """libbas_hip_diag.so (-DBAS_DIAG) exports the same ABI; the shipped library contains neither hook string."""
    hip = bas._hip
    with hip.use_library(hip.DIAG_LIB_PATH) as diag:
        assert hip.lib() is diag and diag.bas_version() == hip.ABI_VERSION
    assert hip.lib() is not diag
    shipped = open(hip.LIB_PATH, "rb").read()
    assert b"BAS_FORCE_KERNEL" not in shipped and b"BAS_DEBUG_FLAGS" not in shipped
    assert b"BAS_FORCE_KERNEL" in open(hip.DIAG_LIB_PATH, "rb").read()


def test_which_fused_kernel_a_shape_gets():
    """bas_render_fused_kernel_name is host logic (the plan of a shape: no launch, no GPU needed - without a device the
    library plans for MI355X's 256 CUs).  BASELINE config 4 and its per-rank shares run the split-role kernel, with the
    unit block for the segment lengths it exists for; scenes of at most two rounds of 2048-output tiles run four waves per
    tile; chunk sizes below ~448 and K = 448 (20 chunk slots do not fit LDS twice) keep the kernel with two workgroups per CU."""
    lib = bas._hip.lib()
    t = 441344
    name = lambda *a: lib.bas_render_fused_kernel_name(*a).decode()
    assert name(256, t, 512, 32, 128) == "bas_render_fs_kernel<128>"
    assert name(32, t, 512, 32, 128) == "bas_render_fs_kernel<128>"            # the N = 8 share of the scene
    assert name(256, t, 512, 32, 121) == "bas_render_fs_kernel<128>"
    assert name(256, t, 512, 32, 100) == "bas_render_fs_kernel<104>"           # the reference's default samples_to_keep
    assert name(256, t, 512, 32, 90) == "bas_render_fs_kernel<0>"
    assert name(256, t, 512, 32, 300) == "bas_render_fs_kernel<0>"             # three tap segments
    assert name(1024, 262656, 512, 32, 128) == "bas_render_fs_kernel<128>"     # a block of BASELINE config 5
    assert name(8, t, 512, 32, 128) == "bas_render_fs_kernel<128>"             # 432 units: some workgroups get two
    assert name(4, t, 512, 32, 128) == "bas_render_fq_kernel"                  # 864 tiles of 2048: two rounds of four-wave workgroups
    assert name(1, t, 512, 32, 128) == "bas_render_fq_kernel"                  # BASELINE configs 2 / 3
    assert name(256, 1024, 512, 32, 128) == "bas_render_fq_kernel"             # a real-time block: 256 sources x 512 samples + halo
    assert name(4, t, 1024, 64, 128) == "bas_render_fq_kernel"
    assert name(2048, 1024, 512, 32, 128) == "bas_render_fz_kernel<1,0>"       # 2048 short units: eight one-wave workgroups per CU
    assert name(256, t, 448, 32, 128) == "bas_render_fz_kernel<4,0>"
    assert name(256, t, 256, 32, 128) == "bas_render_fz_kernel<4,1>"
    # subchunks of 16 / 8 (round 4): two / four crossfaded tap sets per row in the unit blocks of the split-role kernel - big
    # scenes with L = 97 .. 104 or 121 .. 128; everything else with S < 32, and tiny chunks, are not fused
    assert name(256, t, 512, 16, 128) == "bas_render_fs_kernel<128,2>" and lib.bas_render_fused_supported(256, t, 512, 16, 128) == 1
    assert name(32, t, 512, 16, 100) == "bas_render_fs_kernel<104,2>"
    assert name(256, t, 512, 8, 128) == "bas_render_fs_kernel<128,4>" and name(64, t, 1024, 8, 100) == "bas_render_fs_kernel<104,4>"
    for shape in ((256, t, 64, 32, 128), (256, t, 512, 4, 128), (256, t, 512, 16, 90), (256, t, 512, 16, 300), (1, t, 512, 16, 128),
                  (256, t, 256, 16, 128), (256, t, 448, 16, 128)):
        assert name(*shape) == "" and lib.bas_render_fused_supported(*shape) == 0, shape


def test_small_upsampling_factor_is_refused_by_the_planned_entry_points():
    """the read plans step through phase planes assuming U >= 4; smaller factors must be an error
    there (bas_interp2d_f32 serves them with its plain kernel), not silent garbage.  Argument checks run before
    any launch, so this needs no GPU."""
    import ctypes
    hip = bas._hip
    buf = (ctypes.c_char * 4096)()
    a = ctypes.addressof(buf)
    a += (-a) % 16
    for u in (1, 2, 3):
        with pytest.raises(hip.BasError) as err:
            hip.call("bas_interp2d_plan_f32", a, a, a, 1, 187, 64, u, a, 4000, None)
        assert err.value.code == -2 and "upsampling" in str(err.value)
        with pytest.raises(hip.BasError) as err:
            hip.call("bas_render_mix_fused_f32", a, 512, a, a, 1, 512, 512, 32, 64, u, 187, a, 0, None, 0, a, 4000, None)
        assert err.value.code == -2 and "upsampling" in str(err.value)


def test_missing_library_fails_loudly(monkeypatch):
    """No CPU fallback: without the built .so every compute entry point raises."""
    hip = bas._hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", "/nonexistent/libbas_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback|is missing"):
        hip.lib()
    with pytest.raises(RuntimeError):
        hip.call("bas_version")


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    t = bas.synth.make_table("consistent", 0).truncated(16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        bas.irs_and_delaydiffs(t.upsampling, t.diffs_left, t.diffs_right, t.irs_left, t.irs_right)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        bas.delay_signal_float(np.zeros(8), 0.5)


def test_fast_scalar_params_equal_the_numpy_expressions():
    """sphere.interpolation_params takes a plain-Python path for np.float64 azimuths; it must return exactly
    what the reference's numpy expressions give (including the float32-rounded denominators of sphere.py:119),
    on random points and on every grid node +- eps."""
    from binaural_audio_synthesis_amd import sphere
    rng = np.random.default_rng(3)
    pts = [(np.float64(e), np.float64(a)) for e, a in zip(rng.uniform(-1.2, 1.9, 3000), rng.uniform(-20, 20, 3000))]
    for e in sphere.RING_ELEVS_DEG:
        for de in (0.0, 1e-9, -1e-9, 1e-6, -1e-4):
            for az in list(range(0, 361, 15)) + [359.999999, 1e-7]:
                for dz in (0.0, 1e-9, -1e-9):
                    pts.append((np.float64(np.deg2rad(float(e)) + de), np.float64(np.deg2rad(float(az)) + dz)))
                    pts.append((float(np.deg2rad(float(e)) + de), np.float64(np.float32(az) * np.float32(2 * np.pi / 360)) + dz))
    for e, a in pts:
        assert sphere.interpolation_params(e, a) == sphere._interpolation_params_numpy(e, a), (e, a)
    # a Python-float azimuth is NOT taken by the fast path (float32 branch of the reference under numpy 2)
    e, a = 0.3, 4.0
    assert sphere.interpolation_params(e, a) == sphere._interpolation_params_numpy(e, a)
    assert sphere.interpolation_params(np.float64(e), np.float64(a))[1][0] != sphere.interpolation_params(e, a)[1][0]


def test_legacy_index_split_matches_the_oracle():
    """apply_hrtf.ring_easy_params (host side of delay_compensated_interpolation_easy, apply_hrtf.py:116-122)."""
    from binaural_audio_synthesis_amd import apply_hrtf
    from oracle import bas_oracle as orc
    for ci in (73.0, 73.25, 95.999, 96.0, 96.5, 96.999999, 0.0, 0.75, 185.2, 120.000001):
        assert apply_hrtf.ring_easy_params(ci) == orc.ring_easy_params(ci)
    assert apply_hrtf.ring_easy_params(96.5) == (96, 73, 0.5)


def test_fast_fir_row_step_index_algebra():
    """The row step of the FIR kernels (csrc/bas_fir.h: ffa_octet_fma / ffa_combine) restated in numpy with the kernel's
    own index conventions - octet I holds the full-rate taps 8 I + j - 32 relative to the row distance, half-rate tap
    dk = 4 I + jj - 16 of g_e (j even) and g_o (j odd), accumulators A[16], B[17] (p = -1 .. 15), P[16] - against the
    direct 32 x 32 Toeplitz block it replaces, for every row distance of a 128-tap IR."""
    rng = np.random.default_rng(11)
    L = 128
    g = rng.standard_normal(L)
    for rp in range(5):                                          # input row rp rows above the output row
        x = rng.standard_normal(32)
        # direct: y[o] += g[32 rp + o - a] x[a]
        want = np.zeros(32)
        for o in range(32):
            for a in range(32):
                t = 32 * rp + o - a
                if 0 <= t < L:
                    want[o] += g[t] * x[a]
        fa, fb, fp = np.zeros(16), np.zeros(17), np.zeros(16)
        xs = x[0::2] + x[1::2]
        for I in range(8):
            t0 = 32 * rp - 32 + 8 * I
            if not (0 <= t0 < L):                                # the kernel's live mask
                continue
            taps = g[t0:t0 + 8]
            ge, go = taps[0::2], taps[1::2]
            gs = ge + go
            for jj in range(4):
                dk = 4 * I + jj - 16
                for p in range(-1, 16):
                    q = p - dk
                    if 0 <= q < 16:
                        if p >= 0:
                            fa[p] += ge[jj] * x[2 * q]
                            fp[p] += gs[jj] * xs[q]
                        fb[p + 1] += go[jj] * x[2 * q + 1]
        got = np.empty(32)
        got[0::2] = fa + fb[:16]                                 # y[2p] = A[p] + B[p-1]
        got[1::2] = (fp - fa) - fb[1:]                           # y[2p+1] = P[p] - A[p] - B[p]
        assert np.allclose(got, want, rtol=0, atol=1e-12), rp
