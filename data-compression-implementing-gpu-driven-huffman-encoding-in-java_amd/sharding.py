"""Multi-GPU sharding of one file's chunks.

Chunks are independent (own histogram, code table, byte-aligned bitstream: CpuCompressionService.java:210-261,
:731-735), so rank r owns the contiguous chunk range [r*ceil(K/G), min(K, (r+1)*ceil(K/G))) and its output is
one contiguous span of the file.  The only exchange step is an all-gather of the per-chunk compressedSize
(one u32 per chunk) so that every rank can compute the footer's compressedOffset column
(CompressionHeader.java:75) and its own write base.  Payload bytes never cross xGMI.
Works on any torch.distributed backend: "nccl" (= RCCL) with device tensors, "gloo" with CPU tensors.
"""
import torch
import torch.distributed as dist


def chunks_per_rank(num_chunks, world):
    return (num_chunks + world - 1) // world if world > 0 else num_chunks


def chunk_range(num_chunks, world, rank):
    """[first, last) chunk indices owned by `rank`."""
    per = chunks_per_rank(num_chunks, world)
    first = min(num_chunks, rank * per)
    return first, min(num_chunks, first + per)


def byte_range(total_bytes, chunk_bytes, world, rank):
    num_chunks = (total_bytes + chunk_bytes - 1) // chunk_bytes
    first, last = chunk_range(num_chunks, world, rank)
    return min(total_bytes, first * chunk_bytes), min(total_bytes, last * chunk_bytes)


def gather_chunk_sizes(local_sizes, num_chunks, group=None):
    """All-gather the per-chunk compressed sizes.

    local_sizes: int32/int64 tensor with this rank's chunk sizes (possibly empty).
    Returns (all_sizes int64[num_chunks], global_offsets int64[num_chunks], my_base 0-dim int64 tensor):
    global_offsets is the exclusive scan over ALL chunks = the footer's compressedOffset column; my_base is
    this rank's first byte in the file (left on the device so the exchange stays asynchronous)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    per = chunks_per_rank(num_chunks, world)
    dev = local_sizes.device
    if world > 1 and dev.type == "cuda" and dist.get_backend(group) == "gloo":
        dev = torch.device("cpu")  # rehearsal transport: gloo moves host tensors
    padded = torch.zeros(per, dtype=torch.int64, device=dev)
    padded[: local_sizes.numel()] = local_sizes.to(torch.int64)
    if world > 1:
        gathered = torch.empty(world * per, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(gathered, padded, group=group)
    else:
        gathered = padded
    all_sizes = gathered[:num_chunks]
    offsets = torch.cumsum(all_sizes, 0) - all_sizes
    first, _ = chunk_range(num_chunks, world, rank)
    my_base = offsets[first] if first < num_chunks else all_sizes.sum()
    return all_sizes, offsets, my_base
