"""
-m gpu: the exchange of the path at the C boundary (prosstt_amd_comm_* / prosstt_amd_gather_counts, on RCCL directly).
A 1-GPU box can hold one rank only: the communicator comes up, moves bytes through RCCL's send / receive (to itself) and
the gather of the one rank is its rows; what several ranks do with it is the layout parallel.sample_and_gather(order="shard")
tests on gloo (tests/test_parallel_gloo.py).  No node with more than one GPU was available to the builder.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_moves_bytes_and_gathers():
    import torch
    from prosstt_amd import device
    ctx = device.get_context()
    uid = device.Comm.unique_id()
    assert len(uid) == 128 and uid != bytes(128)
    comm = device.Comm(ctx, uid, 0, 1)
    try:
        comm.selftest(1 << 20)
        comm.selftest(12345)
        rng = np.random.default_rng(4)
        rows = torch.as_tensor(rng.integers(0, 1000, (777, 512)).astype(np.int32)).to(ctx.torch_device)
        full = comm.gather_counts(rows, [777], root=0)
        torch.cuda.synchronize()
        assert torch.equal(full, rows)
        with pytest.raises(ValueError):
            comm.gather_counts(rows, [5], root=0)
    finally:
        comm.close()
