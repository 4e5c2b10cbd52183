"""The sharded chunk-batch path (BASELINE configs[3]) with the REAL Separator on the GPU: stacked passes over
work items of different tracks, stems placed through row offsets, and -- with two ranks sharing the one GPU of
the test box over gloo (RCCL refuses two ranks on one device) -- the all-gather + placement, all compared
bitwise with the single-process ``Separator.forward`` of every track."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

CS = 100_000                       # chunk size of these tests: S = 13 slices per full chunk
LENGTHS = (250_000, 130_000, 100_000, 9_500, 300_000, 5_000)


def _tracks(dev):
    from xumx_slicq_amd.synth import synth_audio
    return [synth_audio(n, seed=31 + t).to(dev) for t, n in enumerate(LENGTHS)]


def _reference(sep, tracks):
    """The reference's literal chunk loop (separator.py:147-231), one chunk per pass."""
    sep.batch_chunks = False
    try:
        return {t: sep(x).clone() for t, x in enumerate(tracks)}
    finally:
        sep.batch_chunks = True


@pytest.mark.parametrize("wiener", [False, True])
def test_stacked_items_of_different_tracks_equal_the_chunk_loop(wiener):
    from xumx_slicq_amd.separator import seeded_separator
    from xumx_slicq_amd.sharding import ShardedDemixer
    dev = torch.device("cuda", 0)
    sep = seeded_separator(realtime=False, wiener=wiener, device=dev, chunk_size=CS)
    tracks = _tracks(dev)
    ref = _reference(sep, tracks)
    get = lambda it: tracks[it.track][..., it.start:it.start + it.length]
    for stack in (1, 3):
        dmx = ShardedDemixer(sep, LENGTHS, get, dev, stack=stack)
        for _ in range(2):
            out = dmx.run()
            torch.cuda.synchronize()
            for t in ref:
                assert out[t].shape == ref[t].shape
                assert torch.equal(out[t], ref[t]), (wiener, stack, t, float((out[t] - ref[t]).abs().max()))


def test_demix_into_rejects_bad_arguments():
    from xumx_slicq_amd.separator import seeded_separator
    dev = torch.device("cuda", 0)
    sep = seeded_separator(realtime=False, wiener=False, device=dev, chunk_size=CS)
    out = torch.zeros(8 * 20_000, device=dev)
    offs = (torch.arange(8, device=dev) * 20_000).view(4, 1, 2)
    with pytest.raises(ValueError):
        sep.demix_into(torch.zeros(1, 2, CS + 1, device=dev), out, offs)
    with pytest.raises(ValueError):
        sep.demix_into(torch.zeros(1, 2, 20_000, device=dev), out, offs.view(8))
    sep.demix_into(torch.zeros(1, 2, 20_000, device=dev), out, offs)       # the valid call
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())


def test_place_rows_moves_ragged_rows_bitwise():
    """xsq_place_rows (the concat-by-placement launch of the sharded path): rows of any length and any 4-byte
    alignment on either side, incl. rows shorter than one vector, against plain slicing."""
    from xumx_slicq_amd import _lib
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    src = torch.randn(2_000_000, generator=g).to(dev)
    dst = torch.full((2_200_000,), -7.0, device=dev)
    want = dst.clone()
    rows, so, do = [], 3, 1
    for n in (1, 2, 3, 4, 5, 7, 8191, 8192, 8193, 100_001, 65_536, 33, 250_000):
        for shift in (0, 1, 2, 3):                       # destination alignment relative to the source's
            if so + n > src.numel() or do + shift + n > dst.numel():
                continue
            rows.append((so, do + shift, n))
            want[do + shift:do + shift + n] = src[so:so + n]
            so += n + 1
            do += shift + n + 2
    table = torch.tensor(rows, dtype=torch.int64, device=dev)
    _lib.check(_lib.lib.xsq_place_rows(src.data_ptr(), dst.data_ptr(), table.data_ptr(), len(rows),
                                       max(r[2] for r in rows), _lib.stream_ptr()), "xsq_place_rows")
    torch.cuda.synchronize()
    assert torch.equal(dst, want)
    with pytest.raises(_lib.XsqError):
        _lib.check(_lib.lib.xsq_place_rows(None, dst.data_ptr(), table.data_ptr(), 1, 4, _lib.stream_ptr()), "xsq_place_rows")


def _worker(rank, world, port, q, backend="gloo", gather=True, exchange="sendrecv"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    # gloo: both ranks share device 0 (the functional path of a 1-GPU box); nccl (= RCCL): one rank per device
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    res = (rank, "not run")
    try:
        from xumx_slicq_amd.separator import seeded_separator
        from xumx_slicq_amd.sharding import ShardedDemixer
        assert dist.get_backend() == backend
        msgs = []
        for wiener in (False, True):
            sep = seeded_separator(realtime=False, wiener=wiener, device=dev, chunk_size=CS)
            tracks = _tracks(dev)
            ref = _reference(sep, tracks)
            get = lambda it: tracks[it.track][..., it.start:it.start + it.length]
            dmx = ShardedDemixer(sep, LENGTHS, get, dev, stack=2, gather=gather, exchange=exchange)
            assert dmx.world == world and dmx.gather and dmx.exchange == exchange
            for step in range(2):
                out = dmx.run()
                torch.cuda.synchronize()
                for t in ref:
                    if not torch.equal(out[t], ref[t]):
                        msgs.append(f"wiener={wiener} step={step} track={t} differs by {float((out[t] - ref[t]).abs().max()):.3e}")
            own = ShardedDemixer(sep, LENGTHS, get, dev, stack=2, gather=False).run()
            torch.cuda.synchronize()
            for rnd in dmx.plan.rounds:
                for p in rnd[rank]:
                    i = p.item
                    if not torch.equal(own[i.track][..., i.start:i.start + i.length], ref[i.track][..., i.start:i.start + i.length]):
                        msgs.append(f"wiener={wiener} no-gather item {i} differs")
        res = (rank, "ok" if not msgs else "; ".join(msgs[:4]))
    except Exception as e:          # reported through the queue: a dead worker would only time the test out
        res = (rank, f"{type(e).__name__}: {e}")
    finally:
        q.put(res)
        from xumx_slicq_amd.sharding import close_row_exchanges
        close_row_exchanges()
        dist.destroy_process_group()


def _run_ranks(world, backend, gather=True, exchange="sendrecv"):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend, gather, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    assert res == [(r, "ok") for r in range(world)], res


EXCHANGES = ["sendrecv", "allgather"]        # in-place grouped send / recv (default) | all-gather + placement (fallback)


@pytest.mark.parametrize("exchange", EXCHANGES)
def test_two_ranks_exchange_bitwise(exchange):
    _run_ranks(2, "gloo", exchange=exchange)


@pytest.mark.parametrize("exchange", EXCHANGES)
def test_rccl_group_of_one_runs_the_collective_path(exchange):
    """What a 1-GPU box CAN execute of the RCCL branch: a process group of one rank on backend nccl with ``device_id``
    set, ``ShardedDemixer(gather="always")``.  allgather: communicator creation, the in-place ``all_gather_into_tensor``
    issued asynchronously behind the kernels, the place stream's wait on its work handle, ``xsq_place_rows``.  sendrecv:
    the library's own communicator (xsq_comm_create on torch's librccl), kernels writing their rows in place, one
    grouped exchange per pass on the exchange stream.  Bitwise equal to the single-process chunk loop, mix-phase and
    Wiener-EM.  (Several ranks: the test below, wherever >= 2 devices exist.)"""
    _run_ranks(1, "nccl", gather="always", exchange=exchange)


def test_rccl_self_loop_moves_rows_bitwise():
    """xsq_exchange_rows through RCCL on ONE device: a communicator of one rank, every row sent to the rank itself and
    received into a second buffer (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the caller's stream) -- ragged
    rows of odd offsets and lengths arrive bit for bit, untouched spans stay untouched, bad tables are refused."""
    import numpy as np
    from xumx_slicq_amd import _lib
    from xumx_slicq_amd.sharding import RowExchange
    dev = torch.device("cuda", 0)
    rx = RowExchange(dev, world=1, rank=0)
    assert rx.comm is not None and rx.version() > 20000
    g = torch.Generator().manual_seed(3)
    src = torch.randn(3_000_000, generator=g).to(dev)
    dst = torch.full((3_000_000,), -7.0, device=dev)
    rows = np.asarray([(0, 0, 5, 1), (0, 17, 1001, 333_333), (0, 1_000_001, 400_003, 2_000_000 - 13), (0, 9, 9, 0)], dtype=np.int64)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        rx.exchange(src, rows, dst=dst, self_loop=True)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    want = torch.full((3_000_000,), -7.0, device=dev)
    for _, so, do, n in rows.tolist():
        want[do:do + n] = src[so:so + n]
    assert torch.equal(dst, want)
    bad = np.asarray([(3, 0, 0, 4)], dtype=np.int64)
    with pytest.raises(_lib.XsqError, match="owner"):
        rx.exchange(src, bad, dst=dst, self_loop=True)
    # spans are checked against the buffers before anything is queued: a row that leaves either one is refused
    for row, what in (((0, 3_000_000 - 3, 0, 4), "source span"), ((0, 0, 3_000_000 - 3, 4), "destination span"), ((0, -1, 0, 4), "source span")):
        with pytest.raises(_lib.XsqError, match=what):
            rx.exchange(src, np.asarray([(0, 0, 16, 8), row], dtype=np.int64), dst=dst, self_loop=True)
    torch.cuda.synchronize()
    assert torch.equal(dst, want)                                   # ... and nothing of the refused tables was moved
    # groups closed every few rows (XSQ_EXCHANGE_GROUP_ROWS is read per call): same bytes
    os.environ["XSQ_EXCHANGE_GROUP_ROWS"] = "2"
    try:
        dst2 = torch.full((3_000_000,), -7.0, device=dev)
        rx.exchange(src, rows, dst=dst2, self_loop=True)
        torch.cuda.synchronize()
        assert torch.equal(dst2, want)
    finally:
        del os.environ["XSQ_EXCHANGE_GROUP_ROWS"]
    rx.close()


@pytest.mark.parametrize("exchange", EXCHANGES)
def test_rccl_ranks_exchange_bitwise(exchange):
    """The exchange as it ships: one rank per device, backend nccl (= RCCL over xGMI), ``device_id`` set, the real
    Separator through ``ShardedDemixer(gather=True)`` -- sendrecv: kernels write their rows of the shared flat layout in
    place, one grouped ncclSend / ncclRecv per pass kind and round moves them owner -> peers; allgather: in-place
    all-gather per pass kind and round + one placement launch per exchange -- bitwise equal to the single-process chunk
    loop, mix-phase and Wiener-EM.  Switches itself on wherever two or more devices are visible (the 1-GPU test box skips
    it: RCCL refuses two ranks on one device)."""
    n = torch.cuda.device_count()                # does not initialise the GPU in this (parent) process
    if n < 2:
        pytest.skip(f"needs >= 2 devices for RCCL (found {n}); the same path runs over gloo in the test above")
    _run_ranks(min(n, 8), "nccl", exchange=exchange)
