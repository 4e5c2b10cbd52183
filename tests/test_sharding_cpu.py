"""Multi-rank path on CPU: world_size-2 gloo processes shard the chunk items, exchange
stems with all-gather and must reproduce the single-process result exactly."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from xumx_slicq_amd.sharding import (ShardedDemixer, ShardPlan, WorkItem, assign_lpt, assign_tracks_lpt, chunk_items,
                                     demix_sharded, demix_tracks)


def fake_separate(x):
    """A stand-in for Separator on CPU (the HIP path needs a GPU): chunk-local and
    deterministic, so sharded and sequential runs must agree bit for bit."""
    cs = torch.cumsum(x, dim=-1)
    return torch.stack([x, 2.0 * x, cs - cs.mean(dim=-1, keepdim=True), x.flip(-1)])


def test_items_follow_the_reference_chunk_loop():
    items = chunk_items([10_584_000, 100], 2_621_440)
    assert [i.length for i in items if i.track == 0] == [2_621_440] * 4 + [98_240]
    assert items[-1] == WorkItem(1, 0, 0, 100)
    assert chunk_items([2_621_440], 2_621_440) == [WorkItem(0, 0, 0, 2_621_440)]


def test_lpt_balances_and_is_deterministic():
    lengths = [150 * 44100 + 9973 * i for i in range(50)]          # 50 tracks, 150..160 s
    items = chunk_items(lengths, 2_621_440)
    for world in (1, 2, 4, 8):
        q = assign_lpt(items, world)
        key = lambda i: (i.track, i.chunk)
        assert sorted((i for r in q for i in r), key=key) == sorted(items, key=key)     # a partition
        loads = [sum(i.length for i in r) for r in q]
        assert max(loads) - min(loads) <= 2_621_440
        assert q == assign_lpt(list(reversed(items)), world)
    q8 = assign_lpt(items, 8)
    assert max(sum(i.length for i in r) for r in q8) / (sum(lengths) / 8) < 1.05


def test_track_assignment_is_a_balanced_partition():
    lengths = [150 * 44100 + 9973 * i for i in range(50)]
    for world in (1, 2, 4, 8):
        q = assign_tracks_lpt(lengths, world)
        assert sorted(t for r in q for t in r) == list(range(50))
        loads = [sum(lengths[t] for t in r) for r in q]
        assert max(loads) - min(loads) <= max(lengths)
    assert assign_tracks_lpt([10, 10, 10, 10], 4) == [[0], [1], [2], [3]]   # bench shape: one track per rank


class FakeSeparator:
    """CPU stand-in with the Separator.demix_into contract (items stacked along the batch axis, stems placed
    through a row-offset table): what ShardedDemixer drives on the GPU."""
    chunk_size = 600

    def demix_into(self, audio, out, row_offsets, group=1):
        assert group == 1 and row_offsets.shape == (4, audio.shape[0], 2) and row_offsets.dtype == torch.int64
        flat, n = out.view(-1), audio.shape[-1]
        for i in range(audio.shape[0]):
            est = fake_separate(audio[i:i + 1])                 # (4, 1, 2, n): item-local, as chunks are
            for tg in range(4):
                for c in range(2):
                    o = int(row_offsets[tg, i, c])
                    flat[o:o + n] = est[tg, 0, c]


def _sequential(tracks, cs=600):
    return {t: torch.cat([fake_separate(x[..., s:s + cs]) for s in range(0, x.shape[-1], cs)], dim=-1)
            for t, x in enumerate(tracks)}


def test_shard_plan_rounds_cover_every_item_once():
    lengths = [2500, 700, 1301, 64, 600, 1800]
    for world in (1, 2, 3):
        for stack in (1, 2, 4):
            plan = ShardPlan(lengths, 600, world, stack=stack)
            seen = [p.item for rnd in plan.rounds for per_rank in rnd for p in per_rank]
            assert sorted(seen, key=lambda i: (i.track, i.chunk)) == chunk_items(lengths, 600)
            for k, rnd in enumerate(plan.rounds):
                for r, placed in enumerate(rnd):
                    assert len(placed) <= stack
                    end = [0, 0]
                    for p in placed:                          # packed back to back per exchange (full pass / tails)
                        assert p.part == (0 if p.item.length == 600 else 1)
                        assert p.offset == end[p.part]
                        end[p.part] += 8 * p.item.length
                    assert end[0] <= plan.width[k][0] and end[1] <= plan.width[k][1]
                    # passes: equal lengths only, every item exactly once
                    ps = plan.passes(k, r)
                    assert sorted(id(p) for g in ps for p in g) == sorted(id(p) for p in placed)
                    assert all(len({p.item.length for p in g}) == 1 for g in ps)
            assert plan.imbalance() >= 1.0
            # exchange accounting: one collective per non-empty (round, part); a rank receives every other rank's block
            ex = plan.exchanges()
            assert ex == [(k, p) for k in range(len(plan.rounds)) for p in (0, 1)
                          if any(q.part == p for per_rank in plan.rounds[k] for q in per_rank)]
            acct = plan.exchange_bytes()
            assert acct["collectives_per_step"] == len(ex)
            assert acct["bytes_in_per_rank_per_step"] == sum(4 * (world - 1) * plan.width[k][p] for k, p in ex)
            stems = 32 * sum(lengths)
            assert acct["stem_bytes_in_per_rank_per_step"] <= acct["bytes_in_per_rank_per_step"] or world == 1
            assert acct["stem_bytes_in_per_rank_per_step"] <= stems


def test_sharded_demixer_single_process_places_every_chunk():
    g = torch.Generator().manual_seed(1)
    tracks = [torch.randn(1, 2, n, generator=g) for n in (2500, 700, 1301, 64)]
    get = lambda it: tracks[it.track][..., it.start:it.start + it.length]
    out = ShardedDemixer(FakeSeparator(), [x.shape[-1] for x in tracks], get, torch.device("cpu"), stack=3).run()
    ref = _sequential(tracks)
    assert all(torch.equal(out[t], ref[t]) for t in ref)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = (rank, False, False)
    try:
        g = torch.Generator().manual_seed(0)
        tracks = [torch.randn(1, 2, n, generator=g) for n in (2500, 700, 1301, 64)]
        out = demix_sharded(fake_separate, tracks, chunk_size=600)
        ref = {t: torch.cat([fake_separate(x[..., s:s + 600]) for s in range(0, x.shape[-1], 600)], dim=-1)
               for t, x in enumerate(tracks)}
        ok = all(torch.equal(out[t], ref[t]) for t in ref)
        part = demix_sharded(fake_separate, tracks, chunk_size=600, gather=False)
        mine = assign_lpt(chunk_items([x.shape[-1] for x in tracks], 600), world)[rank]
        ok_part = all(torch.equal(part[i.track][..., i.start:i.start + i.length],
                                  ref[i.track][..., i.start:i.start + i.length]) for i in mine)
        # track-affine path: own tracks only, or everything with gather=True
        whole = {t: fake_separate(x) for t, x in enumerate(tracks)}
        own = demix_tracks(fake_separate, tracks)
        mine_t = assign_tracks_lpt([x.shape[-1] for x in tracks], world)[rank]
        ok_tracks = sorted(own) == sorted(mine_t) and all(torch.equal(own[t], whole[t]) for t in own)
        allg = demix_tracks(fake_separate, tracks, gather=True)
        ok_tracks = ok_tracks and sorted(allg) == [0, 1, 2, 3] and all(torch.equal(allg[t], whole[t]) for t in allg)
        # stacked rounds + one all-gather per round + placement (what bench.py --gpus N runs)
        get = lambda it: tracks[it.track][..., it.start:it.start + it.length]
        lens = [x.shape[-1] for x in tracks]
        ok_dmx = True
        for exchange in ("allgather", "sendrecv"):               # all-gather + placement | rows in place, owner -> peers
            dmx = ShardedDemixer(FakeSeparator(), lens, get, torch.device("cpu"), stack=2, exchange=exchange)
            for _ in range(2):                                   # buffers are reused across steps
                got = dmx.run()
                ok_dmx = ok_dmx and all(torch.equal(got[t], ref[t]) for t in ref)
            # the in-place exchange's table: every row of every item once, owned by the rank the plan gave the item to
            if exchange == "sendrecv":
                rows = sum((dmx._xtable[key].tolist() for key in dmx.plan.exchanges()), [])
                ok_dmx = ok_dmx and len(rows) == 8 * sum(len(q_) for q_ in dmx.plan.queues)
                ok_dmx = ok_dmx and sum(r[3] for r in rows) == 8 * sum(lens) and all(r[1] == r[2] for r in rows)
        own = ShardedDemixer(FakeSeparator(), lens, get, torch.device("cpu"), stack=2, gather=False).run()
        mine2 = [p.item for rnd in dmx.plan.rounds for p in rnd[rank]]
        ok_dmx = ok_dmx and sorted(mine2, key=lambda i: (i.track, i.chunk)) == sorted(mine, key=lambda i: (i.track, i.chunk))
        ok_dmx = ok_dmx and all(torch.equal(own[i.track][..., i.start:i.start + i.length],
                                            ref[i.track][..., i.start:i.start + i.length]) for i in mine2)
        theirs = [p.item for rnd in dmx.plan.rounds for r2 in range(world) if r2 != rank for p in rnd[r2]]
        ok_dmx = ok_dmx and all(float(own[i.track][..., i.start:i.start + i.length].abs().max()) == 0.0 for i in theirs)
        res = (rank, ok, ok_part and ok_tracks and ok_dmx)
    finally:
        q.put(res)
        dist.destroy_process_group()


def test_world_size_two_gloo_matches_sequential():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True, True), (1, True, True)]


# ---- world = 4 and 8: the north star quotes 1 / 2 / 4 / 8 GPUs -------------------------------
def _testset_lengths(n=50):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.testset_lengths(n)


@pytest.mark.parametrize("world", [4, 8])
def test_shard_plan_invariants_on_the_50_track_set(world):
    """configs[3] at its stated size, plan level: the 254 (track, chunk) items of bench.py's 50 track lengths over 4 and 8
    ranks.  Every item placed exactly once; a round holds <= stack items per rank; the exchange tables (what every rank
    would hand xsq_exchange_rows) are a function of the plan only, cover every row of every item once with the owner the
    plan chose, and the group cuts of csrc/exchange.hip (by ROW INDEX) pair every send with a receive of the same group."""
    import numpy as np
    lengths, cs = _testset_lengths(), 2_621_440
    plan = ShardPlan(lengths, cs, world, stack=4)
    items = chunk_items(lengths, cs)
    assert len(items) == 254
    seen = [p.item for rnd in plan.rounds for per_rank in rnd for p in per_rank]
    assert sorted(seen, key=lambda i: (i.track, i.chunk)) == items
    assert plan.imbalance() < 1.03
    owner_of = {(p.item.track, p.item.chunk): r for rnd in plan.rounds for r, per_rank in enumerate(rnd) for p in per_rank}
    assert all(len(per_rank) <= 4 for rnd in plan.rounds for per_rank in rnd)
    # at least two rounds per rank: the exchange of round k runs beside the kernels of round k + 1
    assert min(sum(1 for rnd in plan.rounds if rnd[r]) for r in range(world)) >= 2
    # the tables as ShardedDemixer builds them (no process group needed: they depend on the plan alone)
    track_off = [0]
    for n in lengths:
        track_off.append(track_off[-1] + 8 * n)

    class _D:                                              # just enough of a ShardedDemixer for _exchange_table
        pass
    d = _D()
    d.plan, d.world, d.track_off = plan, world, track_off
    total_rows, total_len, covered = 0, 0, set()
    per_group = max(1, 1024 // max(1, world - 1))          # csrc/exchange.hip: rows_per_group
    for key in plan.exchanges():
        tab = ShardedDemixer._exchange_table(d, *key)
        assert tab.dtype == np.int64 and tab.shape[1] == 4 and tab.flags["C_CONTIGUOUS"]
        assert (tab[:, 1] == tab[:, 2]).all() and (tab[:, 0] >= 0).all() and (tab[:, 0] < world).all()
        assert (tab[:, 1] + tab[:, 3] <= track_off[-1]).all()
        # group by group: the sends rank r queues to peer q in group g == the receives q queues from r in group g
        for g0 in range(0, len(tab), per_group):
            grp = tab[g0:g0 + per_group]
            # what each rank queues, walking the group as xsq_exchange_rows does (owner: a send to every peer; else one receive)
            sent = {r: [] for r in range(world)}           # rank -> [(peer, offset, length)] in issue order
            recvd = {r: [] for r in range(world)}          # rank -> [(source, offset, length)]
            for me in range(world):
                for ow, o, do, n in grp.tolist():
                    if ow == me:
                        sent[me] += [(peer, o, n) for peer in range(world) if peer != me]
                    else:
                        recvd[me].append((ow, do, n))
            for a in range(world):                         # point-to-point operations of a pair match IN ORDER
                for b in range(world):
                    if a != b:
                        assert [(o, n) for peer, o, n in sent[a] if peer == b] == [(o, n) for src, o, n in recvd[b] if src == a]
            assert max(len(sent[r]) + len(recvd[r]) for r in range(world)) <= 1024 + (world - 1)
        for ow, o, _, n in tab.tolist():
            span = (o, n)
            assert span not in covered
            covered.add(span)
        total_rows += len(tab)
        total_len += int(tab[:, 3].sum())
    assert total_rows == 8 * 254 and total_len == 8 * sum(lengths)
    # owners in the tables are the plan's owners
    for key in plan.exchanges():
        k, part = key
        for r in range(world):
            for p in plan.rounds[k][r]:
                assert owner_of[(p.item.track, p.item.chunk)] == r
    acct = plan.exchange_bytes()
    assert acct["collectives_per_step"] == len(plan.exchanges())
    loads = [sum(32 * i.length for i in q) for q in plan.queues]
    assert acct["stem_bytes_in_per_rank_per_step"] == sum(loads) - min(loads)
    assert acct["bytes_in_per_rank_per_step"] >= acct["stem_bytes_in_per_rank_per_step"] * 0.99


def _worker_n(rank, world, port, q):
    """world = 4 / 8: 14 tracks -> 63 items of <= 600 samples, >= 2 rounds of stack = 2 on every rank, both exchange forms,
    buffers reused across two steps, replicas compared."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = (rank, "not run")
    try:
        from xumx_slicq_amd.sharding import close_row_exchanges
        g = torch.Generator().manual_seed(7)
        lens = [2500, 700, 1301, 64, 3100, 1800, 4790, 601, 599, 1200, 2400, 3333, 4199, 5000]
        tracks = [torch.randn(1, 2, n, generator=g) for n in lens]
        ref = _sequential(tracks)
        get = lambda it: tracks[it.track][..., it.start:it.start + it.length]
        notes = []
        for exchange in ("sendrecv", "allgather"):
            dmx = ShardedDemixer(FakeSeparator(), lens, get, torch.device("cpu"), stack=2, exchange=exchange)
            assert dmx.world == world and dmx.settle() is None
            if min(sum(1 for rnd in dmx.plan.rounds if rnd[r]) for r in range(world)) < 2:
                notes.append("a rank has fewer than two rounds")
            for step in range(2):
                dmx.flat.fill_(float("nan"))
                got = dmx.run()
                if not all(torch.equal(got[t], ref[t]) for t in ref):
                    notes.append(f"{exchange} step {step}: differs from the sequential chunk loop")
            sums = [None] * world
            dist.all_gather_object(sums, int(dmx.flat.view(torch.int32).sum(dtype=torch.int64)))
            if len(set(sums)) != 1:
                notes.append(f"{exchange}: replicas differ {sums}")
        out = demix_sharded(fake_separate, tracks, chunk_size=600)
        if not all(torch.equal(out[t], ref[t]) for t in ref):
            notes.append("demix_sharded differs")
        close_row_exchanges()
        res = (rank, "ok" if not notes else "; ".join(notes))
    except Exception as e:                                  # noqa: BLE001
        import traceback
        res = (rank, f"{type(e).__name__}: {e} | {traceback.format_exc()[-400:]}")
    finally:
        q.put(res)
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_world_size_four_and_eight_gloo_matches_sequential(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_n, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, "ok") for r in range(world)], res


def test_single_process_without_process_group():
    tracks = [torch.arange(1500, dtype=torch.float32).view(1, 1, -1).repeat(1, 2, 1)]
    out = demix_sharded(fake_separate, tracks, chunk_size=400)
    ref = torch.cat([fake_separate(tracks[0][..., s:s + 400]) for s in range(0, 1500, 400)], dim=-1)
    assert torch.equal(out[0], ref)


def _fault_worker(rank, world, port, q, fault):
    """One rank fails a LOCAL step of the exchange's construction (XSQ_FAULT_INJECT); the decision has to come out the same
    on both ranks: both raise ExchangeUnavailable (no fallback) or both end up on the all-gather form (fallback) -- and the
    run after the decision is still bitwise the sequential result."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = (rank, "not run")
    try:
        from xumx_slicq_amd.sharding import ExchangeUnavailable, all_ranks_ok, close_row_exchanges
        g = torch.Generator().manual_seed(0)
        tracks = [torch.randn(1, 2, n, generator=g) for n in (2500, 700, 1301, 64)]
        ref = _sequential(tracks)
        get = lambda it: tracks[it.track][..., it.start:it.start + it.length]
        lens = [x.shape[-1] for x in tracks]
        notes = []
        assert all_ranks_ok(True) is True and all_ranks_ok(rank != 1) is False        # MIN over the ranks
        os.environ["XSQ_FAULT_INJECT"] = fault
        try:
            ShardedDemixer(FakeSeparator(), lens, get, torch.device("cpu"), stack=2, exchange="sendrecv")
            notes.append("no-fallback: did not raise")
        except ExchangeUnavailable:
            pass
        dmx = ShardedDemixer(FakeSeparator(), lens, get, torch.device("cpu"), stack=2, exchange="sendrecv", fallback=True)
        if dmx.exchange != "allgather" or not dmx.exchange_note:
            notes.append(f"fallback: exchange={dmx.exchange} note={dmx.exchange_note}")
        note = dmx.settle()
        got = dmx.run()
        if not all(torch.equal(got[t], ref[t]) for t in ref):
            notes.append("fallback run differs")
        # a failure at the first exchange's ENQUEUE (every rank: with gloo's blocking broadcasts a one-sided failure cannot be
        # staged): the vote before any rank blocks sends every rank to the all-gather form, or makes every rank raise
        os.environ["XSQ_FAULT_INJECT"] = ""
        late = ShardedDemixer(FakeSeparator(), lens, get, torch.device("cpu"), stack=2, exchange="sendrecv", fallback=True)
        strict = ShardedDemixer(FakeSeparator(), lens, get, torch.device("cpu"), stack=2, exchange="sendrecv")
        os.environ["XSQ_FAULT_INJECT"] = "exchange:all"
        n2 = late.settle()
        if late.exchange != "allgather" or not n2 or "enqueue" not in n2:
            notes.append(f"enqueue fault: exchange={late.exchange} note={n2}")
        try:
            strict.settle()
            notes.append("enqueue fault without fallback: did not raise")
        except ExchangeUnavailable:
            pass
        got = late.run()
        if not all(torch.equal(got[t], ref[t]) for t in ref):
            notes.append("run after the enqueue-fault fallback differs")
        os.environ["XSQ_FAULT_INJECT"] = ""
        ok = ShardedDemixer(FakeSeparator(), lens, get, torch.device("cpu"), stack=2, exchange="sendrecv", fallback=True)
        if ok.exchange != "sendrecv" or ok.exchange_note is not None or ok.settle() is not None:
            notes.append(f"healthy: exchange={ok.exchange} note={ok.exchange_note}")
        got = ok.run()
        if not all(torch.equal(got[t], ref[t]) for t in ref):
            notes.append("healthy run differs")
        close_row_exchanges()
        res = (rank, "ok" if not notes else "; ".join(notes) + f" ({note})")
    except Exception as e:                                  # noqa: BLE001
        res = (rank, f"{type(e).__name__}: {e}")
    finally:
        q.put(res)
        dist.destroy_process_group()


@pytest.mark.parametrize("fault", ["load:1", "load:0", "init:1"])
def test_exchange_fallback_is_a_collective_decision(fault):
    """a rank-local failure while the in-place exchange is set up must not leave the ranks in
    different modes.  Rank 0 or rank 1 fails before the id broadcast / at communicator creation."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fault_worker, args=(r, 2, port, q, fault)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, "ok"), (1, "ok")], res
