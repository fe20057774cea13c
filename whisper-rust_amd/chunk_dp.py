"""Chunk-parallel data parallelism over the GPUs of one node.

The unit of work is one 30 s chunk with its own `WhisperState`; chunks are independent (the model is
`whisper_full_parallel`'s, whisper.cpp:7771-7806, minus its timestamp stitching), so ranks never exchange
anything on the data path: each rank transcribes its block of chunk ids, and the (tiny) segment lists are
gathered on the host at the end.  Weights are replicated: rank 0 reads the model file once and broadcasts the
bytes over RCCL/xGMI (bench.py), every rank then parses them from memory.
"""
from __future__ import annotations

from typing import Any, List, Sequence


def shard_chunks(n_chunks: int, rank: int, world: int) -> range:
    """Contiguous block partition of chunk ids [0, n_chunks): sizes differ by at most one, rank order = id order."""
    if world <= 0 or not (0 <= rank < world) or n_chunks < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(n_chunks, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def gather_results(local: Sequence[Any], dist=None) -> List[Any]:
    """All ranks contribute their per-chunk results (in chunk-id order); every rank gets the full, ordered list."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local)
    out: List[Any] = [None] * dist.get_world_size()
    dist.all_gather_object(out, list(local))
    return [x for part in out for x in part]
