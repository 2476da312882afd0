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


def _nccl_worker(rank, world, port, n, out):
    import os
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY='0')
    from dirtorch_amd import distributed as dd
    dd.init_from_env('nccl')
    lo, hi = dd.shard_range(n)
    g = torch.Generator().manual_seed(11)
    full = torch.randn(n, 64, generator=g)
    got = dd.allgather_rows(full[lo:hi].cuda(), n)           # RCCL all_gather_into_tensor of the padded blocks
    out[rank] = bool(torch.equal(got.cpu(), full))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs at least two GPUs (the 1-GPU gpurun box skips it)')
@pytest.mark.timeout(600)
def test_torch_distributed_rccl_allgather_on_every_gpu():
    """One process per GPU, world size = torch.cuda.device_count(), torch.distributed over RCCL: the exchange step of
    dirtorch_amd.distributed (unequal shards: n % world != 0) equals the single-process concatenation bit for bit - what
    tests/test_host_cpu.py pins over gloo, on the real collective the first multi-GPU box runs."""
    import os
    import torch.multiprocessing as mp
    world = torch.cuda.device_count()
    n = 1000 * world + world - 1
    port = 29500 + (os.getpid() % 2000)
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_nccl_worker, args=(world, port, n, out), nprocs=world, join=True)
        res = dict(out)
    assert len(res) == world and all(res.values()), res


class _FakeNet(object):
    iscuda, out_dim, without_fc = True, 64, False


def _nccl_ws1_worker(rank, port, n, out):
    """ONE rank over RCCL: communicator creation with device_id=, the error-agreement all-reduce and the padded
    all_gather_into_tensor of extract_sharded / allgather_rows on CUDA buffers, the 'mesh' layout's peer loop (no peers at
    W = 1), barrier, teardown - everything the first multi-GPU run executes except the bytes crossing xGMI."""
    import os
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    from dirtorch_amd import distributed as dd
    r, w, local = dd.init_from_env('nccl', force=True)
    assert (r, w, local) == (0, 1, 0) and dd.is_initialized() and torch.distributed.get_backend() == 'nccl'
    g = torch.Generator().manual_seed(11)
    full = torch.randn(n, 64, generator=g)
    res = {}
    for algo in ('rccl', 'mesh'):
        got = dd.allgather_rows(full.cuda(), n, algo=algo)
        res['rows_' + algo] = bool(torch.equal(got.cpu(), full))
    calls = []

    def extract(ds, trfs, net):
        calls.append(len(ds))
        return full[ds.lo:ds.lo + len(ds)].cuda()

    class DS(object):
        def __len__(self):
            return n
    got = dd.extract_sharded(extract, DS(), None, _FakeNet())
    res['extract_sharded'] = bool(torch.equal(got.cpu(), full)) and calls == [n]

    def boom(ds, trfs, net):
        raise FloatingPointError('planted')
    try:
        dd.extract_sharded(boom, DS(), None, _FakeNet())
        res['error_agreement'] = False
    except FloatingPointError:
        res['error_agreement'] = True
    out.update(res)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_torch_distributed_rccl_at_world_size_one():
    """The multi-rank code on the ONE GPU a gpurun box has (round-5 review, item 4): a fresh process joins a one-rank
    'nccl' group and runs the exchange step of dirtorch_amd.distributed on CUDA tensors through RCCL itself."""
    import os
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_nccl_ws1_worker, args=(port, 1003, out), nprocs=1, join=True)
        res = dict(out)
    assert res and all(res.values()), res
