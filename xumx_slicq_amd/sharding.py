"""Sharding of demix work across the GPUs of one node (BASELINE configs[3]).

The reference is single-process; its chunk loop (separator.py:147-158) carries no state from one
chunk to the next and ends in a hard ``torch.cat`` (separator.py:229-231), so (track, chunk) pairs
are independent work items.  One process per GPU; what this file holds, top down:

* ``ShardedDemixer`` -- what ``bench.py --gpus N`` runs for N > 1 and ``--workload testset50``.
  ``ShardPlan`` (identical on every rank, no communication) deals the items longest-first to the
  ranks and cuts every rank's queue into rounds of ``stack`` items; the equal-length (full-size)
  items of a round form ONE stacked pass through ``Separator.demix_into`` (the pass shape of the
  single-GPU headline), the short tails of the tracks go out first on a side stream.  Every rank
  holds the SAME flat per-track allocation (``flat``; track t = ``(4, nb, 2, N_t)`` at
  ``track_off[t]``, int64 element offsets) and the kernels write a rank's own rows straight into
  their final place -- that IS the waveform concat of separator.py:231.
* The exchange (``exchange="sendrecv"``, default): one ``RowExchange.exchange`` per pass kind and
  round on an exchange stream, beside the next pass's kernels.  The owner of a row ``ncclSend``s
  it to every peer, every other rank ``ncclRecv``s it at the same offset (``xsq_exchange_rows``,
  csrc/exchange.hip: grouped point-to-point over xGMI, no packing buffer, no placement pass; the
  library's own communicator on the RCCL build torch loaded).  ``exchange="allgather"``: the
  kernels write into this rank's slice of a per-exchange block, one in-place
  ``all_gather_into_tensor`` per pass kind and round, one ``xsq_place_rows`` launch per exchange
  moves the rows into ``flat``.  Which of the two runs is a COLLECTIVE decision (``RowExchange``
  construction, ``ShardedDemixer.settle``): all ranks exchange in place, or all fall back
  (``fallback=True``), or all raise ``ExchangeUnavailable``.
* With the ``gloo`` backend (CPU tests; several ranks sharing the one GPU of a test box, which RCCL
  refuses) the same rows travel host-staged: functional path only.
* ``demix_sharded`` / ``demix_tracks``: the simple forms (one item or one whole track per rank and
  round, ``all_gather_into_tensor`` of padded blocks).  Kept for callers without a ``Separator``
  that has ``demix_into``; ``bench.py`` does not run them.

Chunks of different tracks are never merged into one Wiener batch statistic: the window maximum is
taken per item (``group`` argument of ``demix_into``).
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


# --------------------------------------------------------------------------------------------------
# BASELINE configs[3]: a set of tracks as one chunk batch over the ranks, stems all-gathered
# --------------------------------------------------------------------------------------------------
class _Done:
    """Handle of the host-staged (gloo) exchange: ``wait()`` orders the CURRENT stream behind the copy of the
    gathered block into device memory (issued on the stream that was current when the exchange ran)."""

    def __init__(self, event=None):
        self.event = event

    def wait(self):
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)
        return True


def all_gather_stems(recv: Tensor, send: Tensor, group=None, async_op: bool = True):
    """``dist.all_gather_into_tensor`` (RCCL over xGMI with the ``nccl`` backend: device buffers, asynchronous on
    the backend's stream; ``send`` may be this rank's slice of ``recv`` -- the in-place form, no local copy).  With
    the ``gloo`` backend -- CPU tests, and two ranks sharing the one GPU of a test box, which RCCL refuses -- device
    tensors are staged through the host (functional path only)."""
    if dist.get_backend(group) == "gloo":
        if send.device.type == "cuda":
            h_send = send.cpu()                                    # synchronises the producing stream
            h_recv = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_gather_into_tensor(h_recv, h_send, group=group)
            recv.copy_(h_recv)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(recv.device))
            return _Done(ev)
        dist.all_gather_into_tensor(recv, send.clone(), group=group)     # host tensors: no aliasing of in / out
        return _Done()
    return dist.all_gather_into_tensor(recv, send, group=group, async_op=async_op)


class ExchangeUnavailable(RuntimeError):
    """The in-place send / recv exchange cannot be used -- raised on EVERY rank of the group together (the decision is an
    all-reduce of the ranks' local outcomes), so that callers may fall back to the all-gather form without the ranks
    ending up in different modes."""


def _flag_device(group, dev: torch.device):
    return dev if (dist.is_initialized() and dist.get_backend(group) == "nccl") else torch.device("cpu")


def all_ranks_ok(ok: bool, group=None, dev: Optional[torch.device] = None, world: Optional[int] = None) -> bool:
    """MIN over the ranks of a local success flag (True everywhere or False everywhere).  One small all-reduce over the
    torch process group; no-op without a group or on a group of one."""
    if not dist.is_initialized() or (world if world is not None else dist.get_world_size(group)) <= 1:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=_flag_device(group, dev or torch.device("cpu")))
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


def checksum_int32(flat: Tensor, piece: int = 1 << 26) -> int:
    """Exact integer checksum of a float32 tensor's BITS (sum of its words read as int32, in int64), taken in pieces: the
    int64 widening of a whole 4.8 G-float stem allocation would be a 38 GB temporary."""
    words = flat.reshape(-1).view(torch.int32)
    total = 0
    for o in range(0, words.numel(), piece):
        total += int(words[o:o + piece].sum(dtype=torch.int64).item())
    return total


_ROW_EXCHANGES: Dict[tuple, "RowExchange"] = {}


def shared_row_exchange(device: torch.device, group=None, world: Optional[int] = None, rank: Optional[int] = None) -> "RowExchange":
    """ONE RowExchange (= one RCCL communicator) per (device, process group): every ShardedDemixer of a process shares it
    (bench.py builds several over the same ranks).  Collective on first use; ``close_row_exchanges()`` destroys them --
    call it before ``dist.destroy_process_group()``."""
    dev = torch.device(device)
    key = (dev.type, dev.index, id(group) if group is not None else None, world, rank)
    rx = _ROW_EXCHANGES.get(key)
    if rx is None or rx.closed:
        rx = _ROW_EXCHANGES[key] = RowExchange(dev, group, world, rank)
    return rx


def close_row_exchanges():
    for rx in list(_ROW_EXCHANGES.values()):
        rx.close()
    _ROW_EXCHANGES.clear()


class RowExchange:
    """In-place exchange of stem rows between the ranks: every rank holds the same flat layout, the owner of a row sends it
    to every peer at the same offset (``xsq_exchange_rows``: grouped ncclSend / ncclRecv on the current
    stream -- RCCL over xGMI, one point-to-point link per peer, no packing buffer, no placement pass).  The communicator is
    the library's own (``xsq_comm_create``: ncclCommInitRank with an id handed round through ``torch.distributed``), on
    the RCCL build torch already loaded.  With the ``gloo`` backend -- CPU tests, two ranks sharing the one GPU of a test
    box -- the same rows travel as host-staged broadcasts (functional path only).

    Construction is COLLECTIVE and ends the same way on every rank: each rank catches its local error (librccl not found,
    a missing symbol, ncclCommInitRank failing), rank 0 broadcasts the unique id -- or a failure marker -- unconditionally,
    and an all-reduce (MIN) of the ranks' flags after the id and again after ncclCommInitRank decides; on any failure
    anywhere every rank raises ``ExchangeUnavailable`` (and frees what it had created)."""

    def __init__(self, device: torch.device, group=None, world: Optional[int] = None, rank: Optional[int] = None):
        self.dev, self.group = torch.device(device), group
        live = dist.is_initialized()
        self.world = world if world is not None else (dist.get_world_size(group) if live else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if live else 0)
        self.backend = dist.get_backend(group) if live else "none"
        self.comm, self.closed = None, False
        import ctypes as C
        import os
        use_rccl = self.dev.type == "cuda" and self.backend != "gloo"
        several = live and self.world > 1
        if not use_rccl and not several:
            return
        # tests only: XSQ_FAULT_INJECT="load:<rank>" / "init:<rank>" makes that rank fail the local step, so that the
        # collective decision can be exercised without breaking RCCL (tests/test_sharding_cpu.py)
        fault = os.environ.get("XSQ_FAULT_INJECT", "")
        uid, err = C.create_string_buffer(128), None
        try:                                                   # local step 1: resolve RCCL; rank 0 makes the id
            if fault == f"load:{self.rank}":
                raise RuntimeError("injected fault (XSQ_FAULT_INJECT=%s)" % fault)
            if use_rccl:
                from . import _lib
                path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
                _lib.check(_lib.lib.xsq_comm_load(path.encode() if os.path.exists(path) else None), "xsq_comm_load")
                if self.rank == 0:
                    _lib.check(_lib.lib.xsq_comm_unique_id(uid), "xsq_comm_unique_id")
        except Exception as e:                                 # noqa: BLE001 -- reported through the collective decision
            err = e
        if several:
            box = [uid.raw if err is None else None]           # the id or a failure marker: ALWAYS broadcast, nobody waits in vain
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            if box[0] is None and err is None:
                err = RuntimeError("rank 0 could not create the RCCL unique id")
            elif box[0] is not None:
                uid = C.create_string_buffer(box[0], 128)
        if not all_ranks_ok(err is None, group, self.dev, self.world):
            self.closed = True
            raise ExchangeUnavailable(f"RCCL not usable on every rank (this rank: {err!r})")
        out = C.c_void_p()
        try:                                                   # local step 2: ncclCommInitRank (itself collective: every rank calls it)
            if fault == f"init:{self.rank}":
                raise RuntimeError("injected fault (XSQ_FAULT_INJECT=%s)" % fault)
            if use_rccl:
                from . import _lib
                with torch.cuda.device(self.dev):
                    _lib.check(_lib.lib.xsq_comm_create(C.byref(out), uid, self.world, self.rank), "xsq_comm_create")
        except Exception as e:                                 # noqa: BLE001
            err = e
        if not all_ranks_ok(err is None, group, self.dev, self.world):
            if out.value:
                from . import _lib
                _lib.lib.xsq_comm_destroy(out)
            self.closed = True
            raise ExchangeUnavailable(f"ncclCommInitRank did not succeed on every rank (this rank: {err!r})")
        self.comm = out if use_rccl else None

    def version(self):
        from . import _lib
        return _lib.lib.xsq_comm_version() if self.comm is not None else None

    def exchange(self, flat: Tensor, table, dst: Optional[Tensor] = None, self_loop: bool = False):
        """table: numpy int64 (nrows, 4) = (owner, src offset, dst offset, length), identical on every rank.  Asynchronous
        on the current stream (RCCL); ``dst`` defaults to ``flat`` (in place)."""
        dst = flat if dst is None else dst
        import os
        if os.environ.get("XSQ_FAULT_INJECT", "") in ("exchange:all", f"exchange:{self.rank}"):     # tests only (see __init__)
            raise RuntimeError("injected fault at the exchange's enqueue")
        if self.comm is not None:
            from . import _lib
            _lib.check(_lib.lib.xsq_exchange_rows(self.comm, flat.data_ptr(), flat.numel(), dst.data_ptr(), dst.numel(),
                                                  table.ctypes.data, int(table.shape[0]), 1 if self_loop else 0,
                                                  _lib.stream_ptr()), "xsq_exchange_rows")
            return
        if self.world == 1 and not self_loop:
            return
        for owner, so, do, n in table.tolist():               # gloo / no process group: host-staged, functional only
            if n == 0:
                continue
            if self.world == 1:
                dst[do:do + n].copy_(flat[so:so + n])
                continue
            buf = flat[so:so + n].cpu() if owner == self.rank else torch.empty(n, dtype=flat.dtype)
            dist.broadcast(buf, src=dist.get_global_rank(self.group, owner) if self.group is not None else owner, group=self.group)
            if owner != self.rank:
                dst[do:do + n].copy_(buf)

    def close(self):
        self.closed = True
        if self.comm is not None:
            from . import _lib
            _lib.lib.xsq_comm_destroy(self.comm)
            self.comm = None

    def abort(self):
        """ncclCommAbort: give the communicator up without waiting for what is queued on it (``ShardedDemixer.settle``)."""
        self.closed = True
        if self.comm is not None:
            from . import _lib
            _lib.lib.xsq_comm_abort(self.comm)
            self.comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass(frozen=True)
class PlacedItem:
    item: WorkItem
    part: int            # exchange of the round the item travels in: 0 = the stacked full-size pass, 1 = the short tails
    offset: int          # float offset of the item's packed stems (4, nb, 2, length) in its rank's block of that exchange


class ShardPlan:
    """Deterministic schedule, identical on every rank: LPT assignment of the (track, chunk) items, each
    rank's queue cut into rounds of ``stack`` items (longest first, so the full-size chunks of a round form
    one stacked pass and the short tails come last), and per round the packed send layout.

    A round has up to two exchanges ("parts"), one per KIND of pass, so that a pass's stems leave as soon as the
    pass has finished: part 0 carries the round's full-size chunks (one stacked pass), part 1 its short tails.
    In part p of round k rank r packs its items back to back as (4, nb, 2, length_i) row blocks; the block width
    is the largest packed size over the ranks (ranks with less send padding that nobody reads), and a part that is
    empty on every rank does not exist (``width[k][p] == 0``)."""

    PARTS = 2

    def __init__(self, track_lengths: Sequence[int], chunk_size: int, world_size: int, nb_samples: int = 1,
                 stack: int = 4):
        self.lengths = [int(n) for n in track_lengths]
        self.chunk_size, self.world, self.nb, self.stack = int(chunk_size), int(world_size), int(nb_samples), int(stack)
        self.queues = assign_lpt(chunk_items(self.lengths, self.chunk_size), self.world)
        nrounds = max((len(q) + self.stack - 1) // self.stack for q in self.queues) if self.queues else 0
        self.rounds: List[List[List[PlacedItem]]] = []      # [round][rank] -> placed items
        self.width: List[List[int]] = []                    # [round][part] -> floats per rank
        for k in range(nrounds):
            per_rank, width = [], [0] * self.PARTS
            for q in self.queues:
                off, placed = [0] * self.PARTS, []
                for it in q[k * self.stack:(k + 1) * self.stack]:
                    part = 0 if it.length == self.chunk_size else 1
                    placed.append(PlacedItem(it, part, off[part]))
                    off[part] += 8 * self.nb * it.length
                per_rank.append(placed)
                width = [max(w, o) for w, o in zip(width, off)]
            self.rounds.append(per_rank)
            self.width.append([(w + 63) // 64 * 64 for w in width])      # 256-byte granules
        self.audio_seconds = sum(self.lengths)

    def passes(self, k: int, rank: int) -> List[List[PlacedItem]]:
        """Items of round k on ``rank`` grouped into passes: equal-length items share a stacked pass."""
        groups: Dict[int, List[PlacedItem]] = {}
        for p in self.rounds[k][rank]:
            groups.setdefault(p.item.length, []).append(p)
        return [groups[n] for n in sorted(groups, reverse=True)]

    def exchanges(self) -> List[tuple]:
        """(round, part) of every exchange of a step, in issue order (the same on every rank)."""
        return [(k, p) for k in range(len(self.rounds)) for p in range(self.PARTS) if self.width[k][p] > 0]

    def exchange_bytes(self) -> Dict[str, int]:
        """What one step moves per rank: bytes a rank RECEIVES from the others (incl. the padding of ragged blocks),
        the bytes of stems among them, and the number of collectives."""
        ex = self.exchanges()
        wire = sum(4 * (self.world - 1) * self.width[k][p] for k, p in ex)
        loads = [sum(32 * self.nb * i.length for i in q) for q in self.queues]
        return {"collectives_per_step": len(ex), "bytes_in_per_rank_per_step": wire,
                "stem_bytes_in_per_rank_per_step": sum(loads) - min(loads) if self.world > 1 else 0,
                "largest_collective_bytes_per_rank": max((4 * self.width[k][p] for k, p in ex), default=0)}

    def imbalance(self) -> float:
        """max rank load / mean rank load (1.0 = perfectly balanced)."""
        loads = [sum(i.length for i in q) for q in self.queues]
        return max(loads) * len(loads) / max(1, sum(loads))


class ShardedDemixer:
    """Demixes a fixed set of tracks as one chunk batch over the process group.

    ``separator``: a ``Separator`` (``demix_into`` is the compute step).  ``get_chunk(item)`` returns the
    (nb_samples, 2, item.length) audio of a work item, resident on this rank's device (a rank only ever asks
    for its own items).  All buffers (the exchange blocks, the per-track result tensors) are allocated once;
    ``run()`` may be called repeatedly (the bench's steps) and returns {track: (4, nb_samples, 2, N_t)}.

    gather=True  : every rank ends up holding every track.  The kernels of a pass write its stems into this rank's
                   slice of the exchange block, ONE in-place ``all_gather_into_tensor`` per pass kind and round is issued
                   right behind them (RCCL over xGMI, asynchronous: it runs beside the next pass's kernels), and ONE
                   placement launch per exchange (``xsq_place_rows``) moves every row of every rank into the per-track
                   tensors on a side stream -- the final waveform concat of separator.py:231.
    gather=False : the kernels write this rank's items straight into the result tensors; chunks computed by
                   other ranks stay zero (no data-path collective).
    solo=True    : ignore the process group and run the whole set on this rank (world = 1): the single-GPU
                   rate on the same workload.
    A group of ONE rank exchanges nothing and runs as gather=False -- unless ``gather="always"``: the exchange blocks,
    the (one-rank) collective and the placement launch then run all the same, which is how a 1-GPU box executes the
    RCCL branch (tests/test_sharding_gpu.py)."""

    def __init__(self, separator, track_lengths: Sequence[int], get_chunk: Callable[[WorkItem], Tensor],
                 device: torch.device, group: Optional[dist.ProcessGroup] = None, gather: bool = True,
                 nb_samples: int = 1, stack: int = 4, solo: bool = False, exchange: str = "sendrecv", fallback: bool = False):
        if exchange not in ("sendrecv", "allgather"):
            raise ValueError(f"exchange must be 'sendrecv' or 'allgather'; got {exchange!r}")
        self.sep, self.get_chunk, self.dev, self.group, self.gather = separator, get_chunk, torch.device(device), group, gather
        self.exchange, self.fallback, self.exchange_note = exchange, bool(fallback), None
        live = dist.is_initialized() and not solo
        self._live = live
        self.world = dist.get_world_size(group) if live else 1
        self.rank = dist.get_rank(group) if live else 0
        self.plan = ShardPlan(track_lengths, separator.chunk_size, self.world, nb_samples, stack)
        if self.world == 1 and gather != "always":
            self.gather = False
        self.gather = bool(self.gather)
        nb, dt = self.plan.nb, torch.float32
        lens = self.plan.lengths
        # one flat allocation for all tracks: the kernels address it through element offsets (int64: the 50-track set is
        # 4.8 G floats, past 2^32 elements)
        self.track_off = [0]
        for n in lens:
            self.track_off.append(self.track_off[-1] + 8 * nb * n)
        self.flat = torch.zeros(self.track_off[-1], dtype=dt, device=self.dev)
        self.out = {t: self.flat[self.track_off[t]:self.track_off[t + 1]].view(4, nb, 2, lens[t]) for t in range(len(lens))}
        self._place_stream = torch.cuda.Stream(device=self.dev) if self.dev.type == "cuda" else None
        self._tail_stream = None
        self._setup_exchange()

    def _setup_exchange(self):
        """Buffers and tables of the chosen exchange.  sendrecv needs the library's RCCL communicator: its construction is
        a collective decision (``RowExchange``) -- with ``fallback`` every rank switches to the all-gather form together
        and ``exchange_note`` says why; without it every rank raises ``ExchangeUnavailable``."""
        dt = torch.float32
        self.recv: Dict[tuple, Tensor] = {}       # (round, part) -> (world * width) floats, rank-major
        self.send: Dict[tuple, Tensor] = {}       # this rank's slice of it: what the kernels write (in-place all-gather)
        self._place: Dict[tuple, tuple] = {}      # (round, part) -> (row table on the device, rows, longest row)
        self._xtable: Dict[tuple, object] = {}    # (round, part) -> host table (owner, src offset, dst offset, length) of the in-place exchange
        self._offs = {}                           # (round, pass) -> row-offset tensor
        self._rowx = None
        if self.gather and self.exchange == "sendrecv":
            try:
                self._rowx = shared_row_exchange(self.dev, self.group if self._live else None, self.world, self.rank)
            except ExchangeUnavailable as e:
                if not self.fallback:
                    raise
                self.exchange_note = "sendrecv-inplace unavailable (%s); every rank fell back to allgather+place" % str(e)[:300]
                self.exchange = "allgather"
        if self.gather and self.exchange == "sendrecv":
            # every rank keeps the same flat layout; kernels write their rows in place, rows travel owner -> peers
            for key in self.plan.exchanges():
                self._xtable[key] = self._exchange_table(*key)
        elif self.gather:
            for key in self.plan.exchanges():
                w = self.plan.width[key[0]][key[1]]
                self.recv[key] = torch.zeros(self.world * w, dtype=dt, device=self.dev)
                self.send[key] = self.recv[key][self.rank * w:(self.rank + 1) * w]
                self._place[key] = self._place_table(*key)

    def settle(self) -> Optional[str]:
        """One warm-up step whose OUTCOME the ranks agree on.  The vote comes BEFORE any rank blocks: ``run()`` only queues
        work, so each rank knows on the host whether its sends / receives were accepted (a refused table, an RCCL enqueue
        error raise out of ``xsq_exchange_rows``), and an all-reduce (MIN) of that flag -- on a side stream with the
        ``nccl`` backend, so it does not queue behind the exchange it is voting about -- decides.  All ranks queued it:
        synchronize and vote once more on the outcome.  Somebody did not: peers may hold receives whose sender never
        arrives, so every rank ABORTS the communicator (``xsq_comm_abort``) instead of waiting, then all fall back to the
        all-gather form together (``fallback``) or all raise ``ExchangeUnavailable``.  Returns ``exchange_note`` (None =
        the chosen exchange ran).  (Not recoverable: a rank that dies outright -- that is the process group's timeout.)"""
        if not (self.gather and self.exchange == "sendrecv"):
            return self.exchange_note
        grp = self.group if self._live else None
        err, stage = None, "enqueue"
        try:
            self.run()
        except Exception as e:                                 # noqa: BLE001 -- reported through the collective decision
            err = e
        if self.dev.type == "cuda":
            if getattr(self, "_vote_stream", None) is None:
                self._vote_stream = torch.cuda.Stream(device=self.dev)
            with torch.cuda.stream(self._vote_stream):
                queued = all_ranks_ok(err is None, grp, self.dev, self.world)
        else:
            queued = all_ranks_ok(err is None, grp, self.dev, self.world)
        if queued:
            stage = "completion"
            try:
                if self.dev.type == "cuda":
                    torch.cuda.synchronize(self.dev)
            except Exception as e:                             # noqa: BLE001
                err = e
            if all_ranks_ok(err is None, grp, self.dev, self.world):
                return self.exchange_note
        if self._rowx is not None:
            self._rowx.abort()
        if not self.fallback:
            raise ExchangeUnavailable(f"the first in-place exchange failed at {stage} on at least one rank (this rank: {err!r})")
        self.exchange_note = ("first sendrecv-inplace exchange failed at %s on at least one rank (this rank: %r); every rank "
                              "aborted the communicator and fell back to allgather+place" % (stage, err))
        self.exchange = "allgather"
        self._setup_exchange()
        return self.exchange_note

    # element offsets of packed channel (target, item*nb + b, c) for one pass
    def _row_offsets(self, k: int, pi: int, placed: Sequence[PlacedItem]) -> Tensor:
        key = (k, pi)
        t = self._offs.get(key)
        if t is None:
            nb = self.plan.nb
            rows = torch.empty(4, len(placed) * nb, 2, dtype=torch.int64)
            for i, p in enumerate(placed):
                it = p.item
                for tg in range(4):
                    for b in range(nb):
                        for c in range(2):
                            r = (tg * nb + b) * 2 + c
                            if self.gather and self.exchange == "allgather":       # packed (4, nb, 2, length) block of the exchange buffer
                                rows[tg, i * nb + b, c] = p.offset + r * it.length
                            else:                 # final position inside the track's (4, nb, 2, N_t)
                                rows[tg, i * nb + b, c] = self.track_off[it.track] + r * self.plan.lengths[it.track] + it.start
            t = self._offs[key] = rows.to(self.dev)
        return t

    def _place_table(self, k: int, part: int):
        """Row table of one exchange for xsq_place_rows: every (rank, item, target, sample, channel) row of the
        gathered block -> its span of the flat per-track allocation."""
        nb, w = self.plan.nb, self.plan.width[k][part]
        rows = []
        for r in range(self.world):
            for p in self.plan.rounds[k][r]:
                if p.part != part:
                    continue
                it = p.item
                for row in range(8 * nb):
                    rows.append((r * w + p.offset + row * it.length,
                                 self.track_off[it.track] + row * self.plan.lengths[it.track] + it.start, it.length))
        table = torch.tensor(rows, dtype=torch.int64).reshape(-1, 3)
        return table.to(self.dev), len(rows), max((r[2] for r in rows), default=0)

    def _exchange_table(self, k: int, part: int):
        """Host table of one in-place exchange (xsq_exchange_rows): every (rank, item, target, sample, channel) row of the
        pass kind ``part`` of round k -> (owner, offset, offset, length) in the flat per-track allocation."""
        import numpy as np
        nb, rows = self.plan.nb, []
        for r in range(self.world):
            for p in self.plan.rounds[k][r]:
                if p.part != part:
                    continue
                it = p.item
                for row in range(8 * nb):
                    off = self.track_off[it.track] + row * self.plan.lengths[it.track] + it.start
                    rows.append((r, off, off, it.length))
        return np.ascontiguousarray(np.asarray(rows, dtype=np.int64).reshape(-1, 4))

    def _run_pass(self, k: int, pi: int, placed: Sequence[PlacedItem]):
        target = self.send[(k, placed[0].part)] if (self.gather and self.exchange == "allgather") else self.flat
        audio = [self.get_chunk(p.item) for p in placed]
        # (a list: the Separator reads the items where they lie -- no torch.cat of 84 MB per stacked pass; stand-in separators
        #  of the CPU tests get one packed tensor)
        if len(audio) == 1 or not getattr(self.sep, "accepts_item_lists", False):
            audio = audio[0] if len(audio) == 1 else torch.cat(audio, dim=0)
        self.sep.demix_into(audio, target, self._row_offsets(k, pi, placed), group=self.plan.nb)

    def _is_tail(self, placed: Sequence[PlacedItem]) -> bool:
        return placed[0].part == 1

    def _compute_round(self, k: int, part: Optional[int] = None, skip_tails: bool = False):
        for pi, placed in enumerate(self.plan.passes(k, self.rank)):
            if part is not None and placed[0].part != part:
                continue
            if not (skip_tails and self._is_tail(placed)):
                self._run_pass(k, pi, placed)

    def _tails_on_side_stream(self):
        """The short last chunks of the tracks are launch-bound passes (a dozen kernels of a few hundred workgroups):
        like ``Separator.forward`` does for one track, they go out FIRST, on a side stream with its own workspaces, and
        fill in beside the stacked passes; the caller's stream joins before the first collective that carries one.
        Returns True when tails were moved."""
        if self.dev.type != "cuda":
            return False
        tails = [(k, pi, placed) for k in range(len(self.plan.rounds))
                 for pi, placed in enumerate(self.plan.passes(k, self.rank)) if self._is_tail(placed)]
        if not tails:
            return False
        if self._tail_stream is None:
            self._tail_stream = torch.cuda.Stream(device=self.dev)
        main = torch.cuda.current_stream(self.dev)
        self._tail_stream.wait_stream(main)
        with torch.cuda.stream(self._tail_stream):
            for k, pi, placed in tails:
                self._run_pass(k, pi, placed)
        return True

    def _place_exchange(self, key):
        """recv[key] (rank-major packed blocks) -> the per-track tensors: the hard concat by placement, one launch."""
        table, nrows, longest = self._place[key]
        if nrows == 0:
            return
        if self.dev.type == "cuda":
            from . import _lib
            _lib.check(_lib.lib.xsq_place_rows(self.recv[key].data_ptr(), self.flat.data_ptr(), table.data_ptr(),
                                               nrows, longest, _lib.stream_ptr()), "xsq_place_rows")
            return
        src = self.recv[key]                      # host tensors: the gloo tests' stand-in separator
        for so, do, n in table.tolist():
            self.flat[do:do + n].copy_(src[so:so + n])

    @torch.no_grad()
    def run(self) -> Dict[int, Tensor]:
        nrounds = len(self.plan.rounds)
        cuda = self.dev.type == "cuda"
        main = torch.cuda.current_stream(self.dev) if cuda else None
        moved = self._tails_on_side_stream()
        if not self.gather:
            for k in range(nrounds):
                self._compute_round(k, skip_tails=moved)
            if moved:
                main.wait_stream(self._tail_stream)
            return self.out
        if self.exchange == "sendrecv":
            xs = self._place_stream
            joined = False
            for key in self.plan.exchanges():
                k, part = key
                if part == 1 and moved:
                    if not joined:
                        main.wait_stream(self._tail_stream)
                        joined = True
                else:
                    self._compute_round(k, part=part)
                # the rows of this pass kind leave behind its kernels on the exchange stream, beside the next pass's kernels
                # (which write other rows of the flat layout); incoming rows land where no kernel of this rank writes
                if xs is not None:
                    xs.wait_stream(main)
                    with torch.cuda.stream(xs):
                        self._rowx.exchange(self.flat, self._xtable[key])
                else:
                    self._rowx.exchange(self.flat, self._xtable[key])
            if xs is not None:
                main.wait_stream(xs)
            return self.out
        pending, joined = [], False
        for key in self.plan.exchanges():
            k, part = key
            if part == 1 and moved:
                if not joined:                            # every tail of the step was issued up front on the side stream
                    main.wait_stream(self._tail_stream)
                    joined = True
            else:
                self._compute_round(k, part=part)
            # async: the collective waits for this pass's kernels on the backend's own stream and runs beside
            # the next pass's kernels; the placement of the exchange before is queued behind its collective on a
            # side stream, so the host never blocks here with RCCL
            work = all_gather_stems(self.recv[key], self.send[key], group=self.group, async_op=True)
            pending.append((key, work))
            if len(pending) > 1:
                self._finish(*pending.pop(0))
        while pending:
            self._finish(*pending.pop(0))
        if cuda:
            main.wait_stream(self._place_stream)
        return self.out

    def _finish(self, key, work):
        if self._place_stream is None:
            work.wait()
            self._place_exchange(key)
            return
        with torch.cuda.stream(self._place_stream):
            work.wait()              # NCCL/RCCL: a stream-level wait of the place stream on the collective
            self._place_exchange(key)
