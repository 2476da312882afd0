"""The exchange step through the C ABI (include/dir_engine.h: dir_comm_init_all / dir_allgather_desc):
single process, every visible GPU, RCCL all-gather of padded descriptor blocks.  On the 1-GPU test box
the collective degenerates to a copy; with more devices the result must equal the concatenation of the
shards (the same property the gloo world-size-2 test pins for dirtorch_amd.distributed)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_single_process_allgather_over_rccl():
    from dirtorch_amd import _lib
    ndev = torch.cuda.device_count()
    comm = ctypes.c_void_p()
    _lib.call('dir_comm_init_all', ndev, None, ctypes.byref(comm))
    try:
        n = ctypes.c_int()
        _lib.call('dir_comm_size', comm, ctypes.byref(n))
        assert n.value == ndev
        rows, D = 37, 2048
        g = torch.Generator().manual_seed(5)
        shards = [torch.randn(rows, D, generator=g) for _ in range(ndev)]
        send = [s.to('cuda:%d' % i) for i, s in enumerate(shards)]
        recv = [torch.zeros(ndev * rows, D, device='cuda:%d' % i) for i in range(ndev)]
        for i in range(ndev):
            torch.cuda.synchronize(i)
        sp = (ctypes.c_void_p * ndev)(*[t.data_ptr() for t in send])
        rp = (ctypes.c_void_p * ndev)(*[t.data_ptr() for t in recv])
        _lib.call('dir_allgather_desc', comm, sp, rp, rows, D, None)
        want = torch.cat(shards)
        for i in range(ndev):
            torch.cuda.synchronize(i)
            assert torch.equal(recv[i].cpu(), want), i
        with pytest.raises(_lib.DirError):
            _lib.call('dir_allgather_desc', comm, sp, rp, rows, 0, None)
    finally:
        _lib.call('dir_comm_destroy', comm)
    with pytest.raises(_lib.DirError):
        _lib.call('dir_comm_init_all', ndev + 7, None, ctypes.byref(comm))     # more devices than the box has
