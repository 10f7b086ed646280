"""The N > 1 path on CPU: world_size-2 gloo run of the chunk sharding + all-gather of per-chunk sizes
(the only exchange step of the multi-GPU design)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_bytes, chunk, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    entry.load_package()
    from dcz_amd import sharding
    orc = entry.load_oracle()  # test infrastructure: stands in for the per-rank GPU encoder on a CPU-only box
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        K = (total_bytes + chunk - 1) // chunk
        lo, hi = sharding.byte_range(total_bytes, chunk, world, rank)
        data = orc.gen_text(77, lo, hi - lo)  # this rank's contiguous span of the stream
        pay, sizes, offs, lens = orc.compress_blocks(data, chunk)
        all_sizes, offsets, base = sharding.gather_chunk_sizes(torch.from_numpy(sizes.astype(np.int64)), K)
        q.put((rank, lo, hi, pay.tobytes(), all_sizes.numpy().copy(), offsets.numpy().copy(), int(base)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(world, total_bytes, chunk):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total_bytes, chunk, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_two_ranks_reproduce_the_single_rank_file(orc):
    total, chunk = 11 * 4096 + 100, 4096  # 12 chunks, short last chunk; rank 1 gets the tail
    res = _run(2, total, chunk)
    whole = orc.gen_text(77, 0, total)
    pay, sizes, offs, lens = orc.compress_blocks(whole, chunk)
    file_bytes = bytearray(pay.size)
    for rank, lo, hi, part, all_sizes, offsets, base in res:
        assert (all_sizes == sizes.astype(np.int64)).all()          # every rank sees every chunk size
        assert (offsets == offs.astype(np.int64)).all()              # = the footer's compressedOffset column
        first = lo // chunk
        assert base == int(offs[first])
        file_bytes[base:base + len(part)] = part                     # each rank writes one contiguous span
    assert bytes(file_bytes) == pay.tobytes()


def test_uneven_split_leaves_last_rank_short_or_empty(pkg):
    from dcz_amd import sharding
    assert [sharding.chunk_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [sharding.chunk_range(2, 4, r) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert sharding.byte_range(10 * 100 - 7, 100, 4, 3) == (900, 993)
    assert sharding.chunk_range(0, 8, 3) == (0, 0)
