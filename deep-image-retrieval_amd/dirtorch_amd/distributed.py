"""Multi-GPU layout of the retrieval path: one process per GPU, image-parallel database shards,
ONE all-gather of the per-shard descriptor blocks before ranking (BASELINE.json north_star,
SURVEY.md §8e).

The reference's only parallelism is single-process nn.DataParallel (dirtorch/utils/common.py:155),
which is inert at its default batch size of 1.  Here rank r extracts the contiguous index range
[r*N/W, (r+1)*N/W) with replicated weights (no communication), then every rank receives all shards
with a single collective: RCCL all_gather_into_tensor over xGMI on GPUs ('nccl' backend), gloo on
CPU tensors for the tests.  Shards are padded to equal row counts (the collective needs equal
sizes) and trimmed after; the result is bit-identical to the single-process concatenation.
"""
import os

import torch


def is_initialized():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def rank():
    return torch.distributed.get_rank() if is_initialized() else 0


def world_size():
    return torch.distributed.get_world_size() if is_initialized() else 1


def init_from_env(backend=None, force=False):
    """Join the process group described by RANK/WORLD_SIZE/MASTER_* (torch.distributed.run);
    no-op for a single process unless `force` (a one-rank group: the same RCCL communicator set-up, buffer
    registration and collectives as on N GPUs - how the exchange step is exercised on a one-GPU box).
    Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or (force and 'RANK' in os.environ)) and not is_initialized():
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC only on this driver
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
            torch.distributed.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            torch.distributed.init_process_group(backend)
    return rank(), world_size(), local


def shard_range(n, r=None, w=None):
    """Contiguous index range of rank r among w ranks: [r*n//w, (r+1)*n//w)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    return (r * n) // w, ((r + 1) * n) // w


def shard_sizes(n, w=None):
    """[(lo, hi)] of every rank's contiguous range."""
    w = world_size() if w is None else w
    return [shard_range(n, r, w) for r in range(w)]


def padded_rows(n, w=None):
    """Rows of the common (padded) block the collective exchanges: the longest shard."""
    return max(h - l for l, h in shard_sizes(n, w))


def score_gathered(queries, gathered, n_total, w, similarity):
    """Scores [Q, n_total] of `queries` against a database that arrived through ONE all_gather_into_tensor of padded
    shard blocks: `gathered` is [w * rows, D] with shard r at rows [r * rows, r * rows + n_r).  Equal shards: one
    similarity call over the whole buffer.  Unequal shards (n_total % w != 0 - e.g. 1 006 322 rows on 8 GPUs): the
    padding rows sit BETWEEN the shards, so each block is scored on its own and the [Q, n_r] pieces are concatenated -
    no 8 GB compaction copy.  similarity(queries, block) -> [Q, rows of block] (dirtorch_amd.ops.similarity)."""
    rows = padded_rows(n_total, w)
    assert gathered.shape[0] == w * rows, (gathered.shape, w, rows)
    if rows * w == n_total:
        return similarity(queries, gathered)
    return torch.cat([similarity(queries, gathered[r * rows:r * rows + (h - l)])
                      for r, (l, h) in enumerate(shard_sizes(n_total, w))], dim=1)


def merge_score_blocks(score_blocks, n_total, w):
    """The cheaper exchange: every rank scored only its own (padded) rows, the [w, Q, rows] score blocks were
    all-gathered; trims every block's padding columns and concatenates in dataset order -> [Q, n_total]."""
    return torch.cat([score_blocks[r, :, :h - l] for r, (l, h) in enumerate(shard_sizes(n_total, w))], dim=1).contiguous()


def exchange_algo():
    """How the one exchange step moves its blocks: 'rccl' = all_gather_into_tensor (RCCL picks ring / tree), 'mesh' = W - 1
    direct point-to-point sends per rank in ONE grouped launch (every xGMI link of the GPU busy at once: SURVEY section 5 prices
    the 8.2 GB exchange of configs[3] at ~6.7 ms that way against ~47 ms for a ring bound by one link).  DIRTORCH_AMD_EXCHANGE."""
    algo = os.environ.get('DIRTORCH_AMD_EXCHANGE', 'rccl')
    if algo not in ('rccl', 'mesh'):
        raise ValueError("DIRTORCH_AMD_EXCHANGE must be 'rccl' or 'mesh', got %r" % algo)
    return algo


def allgather_blocks(out, block, algo=None):
    """out[r * rows:(r + 1) * rows] = rank r's `block` ([rows, ...], equal shapes on every rank), on every rank.  ONE exchange
    step either way: algo 'rccl' - the collective; 'mesh' - a full-mesh of direct sends / receives issued as one group
    (batch_isend_irecv: ncclGroupStart ... ncclSend / ncclRecv ... ncclGroupEnd under RCCL, so all W - 1 peers move at once)."""
    w, r = world_size(), rank()
    rows = block.shape[0]
    assert out.shape[0] == w * rows and out.shape[1:] == block.shape[1:], (out.shape, block.shape, w)
    algo = exchange_algo() if algo is None else algo
    if w == 1 and not (is_initialized() and block.is_cuda):
        out.copy_(block)
    elif algo == 'mesh':
        out[r * rows:(r + 1) * rows].copy_(block)
        ops = []
        for d in range(1, w):            # peer order staggered per rank: step d pairs r -> r + d with r - d -> r
            to, frm = (r + d) % w, (r - d) % w
            ops.append(torch.distributed.P2POp(torch.distributed.isend, block, to))
            ops.append(torch.distributed.P2POp(torch.distributed.irecv, out[frm * rows:(frm + 1) * rows], frm))
        for req in (torch.distributed.batch_isend_irecv(ops) if ops else ()):    # (a one-rank group has no peers)
            req.wait()
    elif block.is_cuda:
        torch.distributed.all_gather_into_tensor(out, block)
    else:
        torch.distributed.all_gather([out[i * rows:(i + 1) * rows] for i in range(w)], block)
    return out


def allgather_rows(local, n_total, algo=None):
    """local: this rank's [hi-lo, D] block (rows shard_range(n_total)) -> the full [n_total, D]
    on every rank, in dataset order.  One collective (or one group of direct sends: allgather_blocks)."""
    w = world_size()
    if w == 1 and not (is_initialized() and local.is_cuda):
        assert local.shape[0] == n_total
        return local
    # (a ONE-rank process group on the GPU takes the collective path too: RCCL runs the same all-gather on itself, so a
    # one-GPU box exercises communicator, buffers and kernels of the exchange step - tests/test_comm_gpu.py)
    sizes = [shard_range(n_total, r, w) for r in range(w)]
    lo, hi = sizes[rank()]
    assert local.shape[0] == hi - lo, 'shard has %d rows, expected %d' % (local.shape[0], hi - lo)
    rows = max(h - l for l, h in sizes)
    D = local.shape[1]
    padded = local.new_zeros((rows, D))
    padded[:hi - lo] = local
    out = allgather_blocks(local.new_empty((w * rows, D)), padded.contiguous(), algo)
    parts = [out[r * rows:r * rows + (h - l)] for r, (l, h) in enumerate(sizes)]
    return torch.cat(parts, dim=0)


class SubDataset(object):
    """View of dataset[lo:hi] exposing what the loader touches (get_image / get_key / len)."""

    def __init__(self, dataset, lo, hi):
        self.dataset, self.lo, self.nimg = dataset, lo, hi - lo

    def __len__(self):
        return self.nimg

    def get_image(self, i, *a, **k):
        return self.dataset.get_image(self.lo + i, *a, **k)

    def get_key(self, i):
        return self.dataset.get_key(self.lo + i)

    def get_label(self, i, *a, **k):
        return self.dataset.get_label(self.lo + i, *a, **k)


def extract_sharded(extract_fn, dataset, trfs, net, width=None, **kw):
    """extract_fn(dataset, trfs, net, **kw) -> [N, D]; under torch.distributed each rank runs it on
    its own shard and the blocks are all-gathered.  `width`: row length when it is not the network's
    descriptor size (the fused multi-scale extraction returns the scales side by side)."""
    if world_size() == 1 and not (is_initialized() and net.iscuda):
        return extract_fn(dataset, trfs, net, **kw)
    n = len(dataset)
    lo, hi = shard_range(n)
    D = width or (net._head_in_dim() if getattr(net, 'without_fc', False) else net.out_dim)
    dev = 'cuda' if net.iscuda else 'cpu'
    local, err = None, None
    if hi > lo:
        try:
            local = extract_fn(SubDataset(dataset, lo, hi), trfs, net, **kw)
        except Exception as e:     # fp16 overflow on this rank's shard (test_dir._check_finite), a DirError, an OOM ...
            err = e
    # every rank must reach the collective: a rank that raised on its own would leave the others blocked in
    # the all-gather until the RCCL timeout.  Agree on the failure first (ANY exception), then raise everywhere.
    bad = torch.tensor([1 if err else 0], dtype=torch.int32, device=dev)
    torch.distributed.all_reduce(bad, op=torch.distributed.ReduceOp.MAX)
    if int(bad.item()):
        if err:
            raise err
        raise FloatingPointError('another rank failed on its shard (see its log: non-finite descriptors / fp16 overflow, '
                                 'an engine error or out of memory); for an overflow run with DIRTORCH_AMD_DTYPE=bf16')
    if local is None:      # an empty shard (more ranks than images)
        local = torch.zeros((0, D), dtype=torch.float32, device=dev)
    return allgather_rows(local, n)
