"""The RCCL leg of the multi-GPU path, as far as ONE GPU allows: the `nccl` backend of torch.distributed IS RCCL on ROCm, and
avcer_amd/dist.py's collective is one `all_gather_into_tensor` of the per-clip records.  A process
group of world size 1 runs the same code path through the same library -- communicator creation, the device-side collective on
the current stream, the unpadding -- that the 8-GPU bench uses; the multi-rank logic (uneven shards, ordering) is covered on
CPU by tests/test_dist_cpu.py (gloo, world sizes 2, 4, 8).  No multi-GPU curve is measured here."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_rccl_all_gather_of_per_clip_records_world_size_1():
    import socket

    import torch.distributed as dist

    from avcer_amd import dist as adist

    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        g = torch.Generator().manual_seed(3)
        n, t, c = 128, 16, 8
        stat, dyn, aud = torch.rand(n, t, 7, generator=g), torch.randn(n, t, 7, generator=g), torch.randn(n, c, generator=g)
        rec = adist.pack_records(stat.to(dev), dyn.to(dev), aud.to(dev))
        assert tuple(rec.shape) == (n, 2 * t * 7 + c)                       # [128, 232] f32: 928 B per clip
        assert adist.all_gather_records(rec, n) is rec                      # the shortcut the 1-GPU bench takes
        out = adist.all_gather_records(rec, n, force=True)                  # ... and the collective itself, through RCCL
        torch.cuda.synchronize(dev)
        assert out.data_ptr() != rec.data_ptr() and torch.equal(out, rec)
        s2, d2, a2 = adist.unpack_records(out, t, c)
        assert torch.equal(s2.cpu(), stat) and torch.equal(d2.cpu(), dyn) and torch.equal(a2.cpu(), aud)
        # the timing exchange of bench.py's timed region (all_gather_into_tensor of one f64 + all_reduce MAX) on the same group
        mine = torch.tensor([1.25], device=dev, dtype=torch.float64)
        every = torch.zeros(1, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(every, mine)
        dist.all_reduce(mine, op=dist.ReduceOp.MAX)
        dist.barrier()
        assert float(every.item()) == 1.25 and float(mine.item()) == 1.25
    finally:
        dist.destroy_process_group()
