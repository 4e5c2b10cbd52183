"""Sharding of demix work across the GPUs of one node.

The reference is single-process; its chunk loop (separator.py:153-229) carries
no state from one chunk to the next and ends in a hard ``torch.cat``
(separator.py:231), so tracks -- and, inside a track, (track, chunk) pairs -- are
independent work items.  Two levels, one process per GPU:

* ``demix_tracks`` (default, what ``bench.py --gpus N`` runs): whole tracks are dealt to
  ranks longest-first; a rank runs ``Separator.forward`` on its tracks (full chunks stacked
  along the batch axis) and keeps their stems.  No data-path collective: at ~9 ms per 240 s
  track an all-gather of everybody's stems (339 MB per track) would cost several times the
  compute, and nothing downstream needs every rank to hold every track.
* ``demix_sharded``: (track, chunk) items dealt longest-first for batches whose track lengths
  do not balance; the exchange step is an all-gather of the finished stems (RCCL over xGMI
  with the ``nccl`` backend), issued ``async_op=True`` per round so it overlaps the next
  round's kernels.  Chunks are never merged into one Wiener batch across tracks' statistics
  (the window maximum spans the batch dimension).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist
from torch import Tensor


@dataclass(frozen=True)
class WorkItem:
    track: int
    chunk: int
    start: int
    length: int


def chunk_items(track_lengths: Sequence[int], chunk_size: int) -> List[WorkItem]:
    """Work items in the order the reference would visit them (separator.py:147-158)."""
    items = []
    for t, N in enumerate(track_lengths):
        for c, start in enumerate(range(0, N, chunk_size)):
            items.append(WorkItem(t, c, start, min(chunk_size, N - start)))
    return items


def assign_lpt(items: Sequence[WorkItem], world_size: int) -> List[List[WorkItem]]:
    """Longest-processing-time-first: cost of an item ~ its length (every stage is linear
    in the number of slices).  Deterministic, identical on every rank."""
    queues: List[List[WorkItem]] = [[] for _ in range(world_size)]
    load = [0] * world_size
    for it in sorted(items, key=lambda i: (-i.length, i.track, i.chunk)):
        r = min(range(world_size), key=lambda k: (load[k], k))
        queues[r].append(it)
        load[r] += it.length
    return queues


def assign_tracks_lpt(track_lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Whole tracks to ranks, longest first onto the least loaded rank.  Deterministic."""
    queues: List[List[int]] = [[] for _ in range(world_size)]
    load = [0] * world_size
    for t in sorted(range(len(track_lengths)), key=lambda i: (-track_lengths[i], i)):
        r = min(range(world_size), key=lambda k: (load[k], k))
        queues[r].append(t)
        load[r] += track_lengths[t]
    return queues


def demix_tracks(separate: Callable[[Tensor], Tensor], tracks: Sequence[Tensor],
                 group: Optional[dist.ProcessGroup] = None, gather: bool = False) -> Dict[int, Tensor]:
    """Track-affine sharding: this rank demixes the tracks ``assign_tracks_lpt`` gives it and
    returns {track: (4, nb_samples, 2, N_t)} for those.  ``gather=True`` additionally all-gathers
    every track's stems to every rank (the north star's final waveform concat; off by default)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lengths = [int(t.shape[-1]) for t in tracks]
    queues = assign_tracks_lpt(lengths, world)
    out = {t: separate(tracks[t]) for t in queues[rank]}
    if not gather or world == 1:
        return out
    rounds = max(len(q) for q in queues)
    dev, dt, B = tracks[0].device, tracks[0].dtype, tracks[0].shape[0]
    for k in range(rounds):
        width = max(lengths[q[k]] for q in queues if k < len(q))
        send = torch.zeros(4, B, 2, width, dtype=dt, device=dev)
        if k < len(queues[rank]):
            send[..., :lengths[queues[rank][k]]] = out[queues[rank][k]]
        recv = torch.empty(world * 4, B, 2, width, dtype=dt, device=dev)
        dist.all_gather_into_tensor(recv, send, group=group)
        for r in range(world):
            if k < len(queues[r]):
                t = queues[r][k]
                out[t] = recv[4 * r:4 * r + 4, ..., :lengths[t]].clone()
    return out


def demix_sharded(separate_chunk: Callable[[Tensor], Tensor], tracks: Sequence[Tensor],
                  chunk_size: int, group: Optional[dist.ProcessGroup] = None,
                  gather: bool = True) -> Dict[int, Tensor]:
    """Demix ``tracks`` (each (nb_samples, 2, N_t), resident on this rank's device) with the
    chunk items sharded over the process group.

    separate_chunk: (nb_samples, 2, n) -> (4, nb_samples, 2, n) for ONE chunk (a Separator whose
    chunk_size is >= n).  Returns {track: (4, nb_samples, 2, N_t)}; with ``gather`` every rank
    holds every track (the final waveform concat of the north star), otherwise only the
    chunks this rank computed are filled in (zeros elsewhere).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lengths = [int(t.shape[-1]) for t in tracks]
    queues = assign_lpt(chunk_items(lengths, chunk_size), world)
    rounds = max(len(q) for q in queues)
    dev, dt = tracks[0].device, tracks[0].dtype
    B = tracks[0].shape[0]
    out = {t: torch.zeros(4, B, 2, lengths[t], dtype=dt, device=dev) for t in range(len(tracks))}
    pending = []   # (work handle, gathered buffer, round)
    for k in range(rounds):
        mine = queues[rank][k] if k < len(queues[rank]) else None
        est = None
        if mine is not None:
            est = separate_chunk(tracks[mine.track][..., mine.start:mine.start + mine.length])
        if world == 1 or not gather:
            if mine is not None:
                out[mine.track][..., mine.start:mine.start + mine.length] = est
            continue
        # all ranks contribute a buffer padded to the longest item of the round
        width = max(q[k].length for q in queues if k < len(q))
        send = torch.zeros(4, B, 2, width, dtype=dt, device=dev)
        if mine is not None:
            send[..., :mine.length] = est
        recv = torch.empty(world * 4, B, 2, width, dtype=dt, device=dev)   # rank-major concat
        work = dist.all_gather_into_tensor(recv, send, group=group, async_op=True)
        pending.append((work, recv, send, k))
    for work, recv, _send, k in pending:
        work.wait()
        for r in range(world):
            if k < len(queues[r]):
                it = queues[r][k]
                out[it.track][..., it.start:it.start + it.length] = recv[4 * r:4 * r + 4, ..., :it.length]
    return out
