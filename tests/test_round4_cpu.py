"""CPU tests added in round 4 (no GPU): the root-weighted source sharding, the A/B builds of the Makefile still compile
(hipcc -fsyntax-only), argument errors of the new entry points."""
import ctypes
import os
import subprocess

import pytest

from conftest import ROOT
import binaural_audio_synthesis_amd as bas

CSRC = os.path.join(ROOT, "binaural-audio-synthesis_amd", "csrc")


@pytest.mark.parametrize("n_src,world", [(256, 8), (256, 4), (256, 2), (5, 2), (7, 3), (1024, 8), (8, 8)])
@pytest.mark.parametrize("weight", [1.0, 0.836, 0.5, 0.0])
def test_root_weighted_shards_partition_the_sources(n_src, world, weight):
    """distributed.shard_sources: contiguous, disjoint, complete for every root weight; weight 1 is the balanced split;
    a lighter root hands its sources to the others evenly."""
    from binaural_audio_synthesis_amd.distributed import shard_sources
    for root in (0, world - 1):
        shards = [shard_sources(n_src, world, r, root_weight=weight, root=root) for r in range(world)]
        assert shards[0].start == 0 and shards[-1].stop == n_src
        for a, b in zip(shards, shards[1:]):
            assert a.stop == b.start
        sizes = [len(s) for s in shards]
        others = [sz for r, sz in enumerate(sizes) if r != root]
        assert max(others) - min(others) <= 1
        if weight == 1.0:
            assert max(sizes) - min(sizes) <= 1
        else:
            assert sizes[root] <= min(others) or n_src < world
            assert abs(sizes[root] - weight * n_src / world) <= 0.5 + 1e-9 or sizes[root] == max(0, n_src - (world - 1))
    with pytest.raises(ValueError):
        shard_sources(n_src, world, 0, root_weight=1.5)


def test_bench_root_weight_rule():
    """bench.py --root-weight auto: six sources' worth of render time is what the root's receive + sum costs."""
    import argparse
    import bench
    a = argparse.Namespace(root_weight="auto", sources=256)
    assert bench.root_weight(a, 1) == 1.0
    assert abs(bench.root_weight(a, 8) - (1 - 42 / 256)) < 1e-12
    from binaural_audio_synthesis_amd.distributed import shard_sources
    assert [len(shard_sources(256, 8, r, root_weight=bench.root_weight(a, 8))) for r in range(8)] == [27, 33, 33, 33, 33, 33, 32, 32]
    assert bench.root_weight(argparse.Namespace(root_weight="auto", sources=5), 2) == 1.0      # toy scenes: equal shares
    assert bench.root_weight(argparse.Namespace(root_weight="0.75", sources=256), 4) == 0.75


@pytest.mark.parametrize("defs", [["-DFZ_ASM=0"], ["-DFZ_SPLIT=0", "-DFZ_QUAD=0"], ["-DBAS_DIAG", "-DBAS_STAMPS"]])
def test_ab_builds_of_the_makefile_still_compile(defs):
    """`make cppstep` (hipcc's own schedule of the row step: the A/B partner DESIGN.md cites for the generated assembly),
    `make nosplit` and `make stamps`: the three fused translation units pass hipcc -fsyntax-only with those definitions."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    for src in ("bas_fused.hip", "bas_fused_split.hip", "bas_fused_quad.hip"):
        r = subprocess.run([hipcc, "-std=c++17", "--offload-arch=gfx950", "-fsyntax-only", "-Wno-unused-function"] + defs + [src],
                           cwd=CSRC, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, f"{src} {defs}:\n{r.stderr[-3000:]}"


def test_argument_errors_of_the_round4_entry_points():
    lib = bas._hip.lib()
    buf = (ctypes.c_char * 4096)()
    a = ctypes.addressof(buf)
    a += (-a) % 16
    assert lib.bas_mix_workspace_bytes() >= 2048
    rc = lib.bas_mix_finish_f32(a, 2, 8, 16, a, None, 1, a, 64, None)               # part_stride < n
    assert rc == -2 and b"part_stride" in lib.bas_last_error()
    rc = lib.bas_mix_finish_f32(a, 2, 16, 16, a, None, 1, a, 32, None)              # workspace too small
    assert rc == -4 and b"bas_mix_workspace_bytes" in lib.bas_last_error()
    rc = lib.bas_mix_finish_f32(a, 2, 16, 16, a + 4, None, 1, a, 1 << 20, None)     # y not 16-byte aligned
    assert rc == -3
    rc = lib.bas_render_status(None, 0, None)
    assert rc == -4
    for name in ("bas_render_fused_fir_f32", "bas_render_fused_reduce_f32", "bas_render_mix_fused_f32"):
        with pytest.raises(bas._hip.BasError) as err:                               # K % S != 0, reference's assertion text
            bas._hip.call(name, a, 512, a, a, 1, 512, 512, 48, 64, 8, 187, a, 0, None, 0, a, 4000, None)
        assert err.value.code == -2 and "subchunksize does not divide chunksize evenly" in str(err.value)


def test_argument_errors_of_the_table_builder_entry_points():
    lib = bas._hip.lib()
    buf = (ctypes.c_char * 4096)()
    a = ctypes.addressof(buf)
    assert lib.bas_resample_up_f64(None, 1, 8, a, 3, 2, a, None) == -1                  # null x
    assert lib.bas_resample_up_f64(a, 0, 8, a, 3, 2, a, None) == -2                     # no rows
    assert lib.bas_resample_up_f64(a, 1, 8000, a, 300, 8, a, None) == -2 and b"LDS" in lib.bas_last_error()
    a += (-a) % 16
    assert lib.bas_delaydiffs_f64(a, 3, 16, a, 3, 2, a, None, None) == -1               # null status
    assert lib.bas_delaydiffs_f64(a, 3, 16, a, 3, 2, a, a + 4, None) == -3              # status not 8-byte aligned
    assert lib.bas_delaydiffs_f64(a, 70000, 16, a, 3, 2, a, a, None) == -2 and b"65535" in lib.bas_last_error()
    assert lib.bas_delaydiffs_f64(a, 3, 16, a, 0, 2, a, a, None) == -2                  # Lh = 0


def test_vectorized_trajectory_that_cannot_broadcast_falls_back(monkeypatch):
    """make_signal_move_2d(vectorized=True) with a function that raises on an array argument (math.sin) must take the
    scalar path instead of failing; checked on the host logic alone by stopping at the first device call."""
    import math
    import numpy as np
    calls = []

    def traj(t):
        calls.append(type(t))
        return 0.1, math.sin(t / 1000.0)                    # TypeError for an ndarray argument

    class Stop(Exception):
        pass

    def no_gpu(*a, **k):
        raise Stop()
    monkeypatch.setattr(bas.apply_hrtf, "as_device_table", lambda t: type("T", (), {"L": 128, "device": "cpu"})())
    monkeypatch.setattr(bas.apply_hrtf, "_params_to_device", no_gpu)
    with pytest.raises(Stop):
        bas.apply_hrtf.make_signal_move_2d(np.zeros(2000, dtype=np.float32), 512, 32, traj, object(), vectorized=True)
    assert np.ndarray in calls and int in calls             # tried the array call, then called chunk by chunk


@pytest.mark.parametrize("lseg,nsub,nbuf,psplit", [(128, 1, 3, False), (104, 1, 3, False), (128, 1, 2, True), (104, 1, 2, True),
                                                   (128, 2, 3, False), (104, 2, 3, False), (128, 4, 2, False), (104, 4, 2, False),
                                                   (128, 1, 2, False), (128, 1, 4, False)])
def test_generated_unit_blocks_compute_the_fir(lseg, nsub, nbuf, psplit):
    """tools/emulate_fir_asm.py executes the instruction list tools/gen_fir_asm.py writes for a unit block - LDS reads with
    the in-order return queue and the s_waitcnt counts, packed FMAs with op_sel - on random rows and taps (float64), combines
    the accumulators the way the flush does and compares with the defining sum (apply_hrtf.py:442-446): every variant the
    library ships (rolling x row; the P product split once more; two and four tap sets per row for subchunks of 16 / 8; both
    segment lengths) computes the FIR to 1e-12, and no register is read while a read into it is in flight."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import emulate_fir_asm as emu
    import gen_fir_asm as gen
    lines, last = gen.gen_unit_roll(261, lseg, nsub, nbuf, psplit=psplit)
    assert last + 11 + 12 <= 256                             # operands + what hipcc keeps around the block still fit a wave
    y, want = emu.run(lines, 261, lseg, nsub, psplit, seed=lseg + 7 * nsub)
    assert np.abs(y - want).max() <= 1e-12 * np.abs(want).max()
    if not psplit and nsub == 1:                             # round 3's block (two x buffers) through the same interpreter
        y0, want0 = emu.run(gen.gen_unit(261, lseg), 261, lseg, 1, False, seed=lseg + 7 * nsub)
        assert np.abs(y0 - want0).max() <= 1e-12 * np.abs(want0).max() and np.array_equal(want0, want)


def test_tile_filling_block():
    """stream.tile_filling_block: a multiple of the chunk size, not above the request, whose window [halo | block] + L - 1
    outputs does not spill into one more 8192-output tile; small requests come back rounded to chunks."""
    f = bas.stream.tile_filling_block
    assert f(1 << 18, 512, 128) == 261120                   # 2^18 would be 32.08 tiles
    for about, k, l in [(1 << 18, 512, 128), (1 << 20, 512, 128), (32768, 512, 100), (9000, 512, 128), (1 << 18, 1024, 128),
                        (100000, 512, 300)]:
        b = f(about, k, l)
        halo = -(-(l - 1) // k) * k
        assert b % k == 0 and 0 < b <= about
        t_out = halo + b + l - 1
        assert -(-t_out // 8192) == t_out // 8192 or t_out % 8192 > 8192 - k    # ends within a chunk of a tile boundary
        assert t_out // 8192 == (halo + about + l - 1) // 8192 or -(-t_out // 8192) == (halo + about + l - 1) // 8192
    assert f(512, 512, 128) == 512 and f(300, 128, 128) == 256 and f(100, 512, 128) == 512
