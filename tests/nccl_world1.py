"""Helper of tests/test_gpu_round3.py (run as a child process on the GPU box): the multi-GPU code paths of
binaural_audio_synthesis_amd.distributed with a REAL RCCL communicator - backend "nccl", world_size 1, the only
size a one-GPU box offers - against the single-process results, bit for bit.  Exercises what no gloo test can:
device tensors through dist.gather (async and blocking), work.wait() stream semantics, device-side all_reduce,
barrier, and communicator creation with device_id."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import binaural_audio_synthesis_amd as bas
    from binaural_audio_synthesis_amd import distributed as D

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1

    h = bas.synth.make_table("consistent", 0).truncated(128)
    d = bas.irs_and_delaydiffs(h.upsampling, h.diffs_left, h.diffs_right, h.irs_left, h.irs_right, device=dev)
    n_src, k, s, n = 5, 512, 32, 6 * 2048
    sigs = np.stack([bas.synth.integer_noise(300 + i, n, 0.3) for i in range(n_src)])     # loud: the peak rule fires
    t = np.arange(0, n + 1, k, dtype=np.float64)
    elev = np.empty((n_src, t.size))
    azim = np.empty((n_src, t.size))
    for i in range(n_src):
        elev[i], azim[i] = bas.synth.trajectory(("spiral", "circle_askew", "passing")[i % 3], period_s=0.09,
                                                length_s=n / 44100, turns=3.0)(t)
    whole = bas.render_sources(sigs, k, s, elev, azim, d)                                   # single process, peak rule
    assert float(bas.render_sources(sigs, k, s, elev, azim, d, normalize="none").abs().max()) > 1.0

    # 1. source sharding: one gather of the partial mix + fixed-order sum + peak rule
    got = D.render_sources_sharded(sigs, k, s, elev, azim, d)
    assert torch.equal(got, whole), "render_sources_sharded under nccl differs"
    # 2. the collective itself, asynchronous, as bench.py issues it
    part = bas.render_sources(sigs, k, s, elev, azim, d, normalize="none").t().contiguous()
    recv = torch.empty((1,) + tuple(part.shape), dtype=part.dtype, device=dev)
    work = dist.gather(part, gather_list=list(recv.unbind(0)), dst=0, async_op=True)
    work.wait()
    y, peak = D._hip_mix_partials(recv)
    torch.cuda.synchronize()
    assert torch.equal(y, part) and float(peak) == float(part.abs().max())
    # 3. time sharding
    got_t = D.render_time_sharded(torch.from_numpy(sigs).to(dev), k, s, elev, azim, d, 128)
    assert torch.equal(got_t, whole), "render_time_sharded under nccl differs"
    # 4. sharded stream: blocks + finish, one gather per block
    st = D.ShardedStreamRenderer(d, n_src, k, s)
    plain = bas.StreamRenderer(d, n_src, k, s)
    B = 2048
    for b in range(n // B):
        c0 = b * B // k
        args = (sigs[:, b * B:(b + 1) * B], elev[:, c0:c0 + B // k + 1], azim[:, c0:c0 + B // k + 1])
        assert torch.equal(st.process(*args), plain.process(*args)), f"sharded stream block {b} differs"
    assert torch.equal(st.finish(), plain.finish())
    assert abs(st.peak - plain.peak) == 0.0
    # 5. the small collectives of bench.py's settle loop and timing
    v = torch.tensor([3.5], dtype=torch.float64, device=dev)
    dist.all_reduce(v, op=dist.ReduceOp.MIN)
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    dist.barrier()
    assert float(v) == 3.5
    dist.destroy_process_group()
    print("nccl world_size=1: OK")


if __name__ == "__main__":
    main()
