"""BASELINE configs[3] AT ITS STATED SIZE: the 50-track set of bench.py (13,514 s of stereo audio, 254 (track, chunk) work
items of <= 2,621,440 samples, one flat stem allocation of 4.8 G floats -- element offsets past 2^32) through
``ShardedDemixer`` with the exchange machinery on, against ``Separator.forward`` of every whole track (the call that holds
the hard concat of /root/reference/xumx_slicq_v2/separator.py:147-158,229-231).  Bitwise on every track, with an exact
int32-sum checksum per track on top; both post-filters; both forms of the exchange:

* one rank on backend nccl (= RCCL): the library's own communicator, grouped ncclSend / ncclRecv per pass (a group of one
  moves nothing but runs every call), or in-place all_gather_into_tensor + xsq_place_rows;
* two spawned gloo ranks sharing the one GPU of the test box: real ownership split, rows owner -> peer at identical
  offsets of each rank's own copy of the flat layout (host-staged: RCCL refuses two ranks on one device);
* ``bench.py --gpus 2 --workload testset50`` as the driver starts it (self_launch -> torch.distributed.run -> ranks -> ONE
  JSON line), over gloo, with its `verified` and `collective` blocks.

Several ranks over RCCL itself need a box with >= 2 devices (tests/test_sharding_gpu.py::test_rccl_ranks_exchange_bitwise)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHUNK = 2_621_440


def _worker(rank, world, port, q, backend, wiener, exchange, ntracks=50):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    res = (rank, "not run")
    try:
        sys.path.insert(0, ROOT)
        import bench
        from xumx_slicq_amd.separator import seeded_separator
        from xumx_slicq_amd.sharding import ShardedDemixer, chunk_items, close_row_exchanges
        from xumx_slicq_amd.synth import synth_audio_device
        lengths = bench.testset_lengths(ntracks)
        assert lengths == bench.testset_lengths()[:ntracks]
        assert ntracks != 50 or len(chunk_items(lengths, CHUNK)) == 254
        sep = seeded_separator(realtime=False, wiener=wiener, device=dev, chunk_size=CHUNK)
        # the bench's own audio: item (track, chunk) seeded on its own, a track = its items back to back
        def whole(t):
            return torch.cat([synth_audio_device(it.length, seed=20260101 + 64 * t + it.chunk, device=dev)
                              for it in chunk_items([lengths[t]], CHUNK)], dim=-1)
        if world <= 2:                                           # whole tracks resident (84 MB per 240 s), items are views
            tracks = [whole(t) for t in range(ntracks)]
            get = lambda it: tracks[it.track][..., it.start:it.start + it.length]
        else:                                                    # several ranks share the one device: a rank keeps its own items only
            tracks, items = None, {}
            def get(it):
                key = (it.track, it.chunk)
                if key not in items:
                    items[key] = synth_audio_device(it.length, seed=20260101 + 64 * it.track + it.chunk, device=dev)
                return items[key]
        dmx = ShardedDemixer(sep, lengths, get, dev, stack=4, gather=("always" if world == 1 else True), exchange=exchange)
        assert dmx.world == world and dmx.gather and dmx.exchange == exchange and dmx.settle() is None
        assert dmx.flat.numel() == 8 * sum(lengths) and (ntracks < 50 or dmx.flat.numel() > (1 << 32))
        cross = next((t for t in range(ntracks) if dmx.track_off[t] < (1 << 32) <= dmx.track_off[t + 1]), -1)
        if world > 2:                                            # every rank: >= 2 rounds, a stacked pass and tails, rows from every peer
            per_rank = [sum(1 for rnd in dmx.plan.rounds if rnd[r]) for r in range(world)]
            assert min(per_rank) >= 2, per_rank
            owners = {int(o) for key in dmx.plan.exchanges() for o in (dmx._xtable[key][:, 0] if exchange == "sendrecv" else [])}
            assert exchange != "sendrecv" or owners == set(range(world)), owners
        msgs = []
        for step in range(2 if world == 1 else 1):               # buffers and tables are reused across steps
            dmx.flat.fill_(float("nan"))                         # every element has to be written by the step
            out = dmx.run()
            torch.cuda.synchronize()
            for t in range(ntracks):
                ref = sep(tracks[t] if tracks is not None else whole(t))
                if out[t].shape != ref.shape or not torch.equal(out[t], ref):
                    msgs.append(f"step {step} track {t}{' (crosses 2^32)' if t == cross else ''} differs by "
                                f"{float((out[t] - ref).abs().nan_to_num(nan=9e9).max()):.3e}")
                a = int(out[t].view(torch.int32).sum(dtype=torch.int64).item())
                b = int(ref.view(torch.int32).sum(dtype=torch.int64).item())
                if a != b:
                    msgs.append(f"step {step} track {t}: checksum {a} != {b}")
                del ref
                if world > 2:
                    torch.cuda.empty_cache()                     # eight caching allocators share one device
        if world > 1:                                            # every rank must hold the same bits of everything
            from xumx_slicq_amd.sharding import checksum_int32
            mine = checksum_int32(dmx.flat)
            sums = [None] * world
            dist.all_gather_object(sums, mine)
            if len(set(sums)) != 1:
                msgs.append(f"replicas differ: {sums}")
        close_row_exchanges()
        res = (rank, "ok" if not msgs else "; ".join(msgs[:4]))
    except Exception as e:                                       # noqa: BLE001 -- reported through the queue
        import traceback
        res = (rank, f"{type(e).__name__}: {e} | {traceback.format_exc()[-600:]}")
    finally:
        q.put(res)
        dist.destroy_process_group()


def _run(world, backend, wiener, exchange, ntracks=50):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend, wiener, exchange, ntracks)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=1500) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    assert res == [(r, "ok") for r in range(world)], res


@pytest.mark.parametrize("exchange", ["sendrecv", "allgather"])
@pytest.mark.parametrize("wiener", [False, True])
def test_testset50_one_rank_with_the_exchange_machinery_on(wiener, exchange):
    _run(1, "nccl", wiener, exchange)


@pytest.mark.parametrize("wiener,exchange", [(False, "sendrecv"), (True, "allgather")])
def test_testset50_two_gloo_ranks_on_one_device(wiener, exchange):
    _run(2, "gloo", wiener, exchange)


@pytest.mark.parametrize("wiener,exchange", [(False, "sendrecv"), (True, "allgather")])
def test_testset_eight_gloo_ranks_on_one_device(wiener, exchange):
    """the world size the north star ends at.  Eight spawned ranks share device 0 over gloo (RCCL
    refuses several ranks per device), the real Separator, the first 16 tracks of the set (72 work items: three rounds per
    rank, stacked passes and tails, rows owned by every rank): bitwise against per-track Separator.forward on every rank,
    identical replica checksums."""
    _run(8, "gloo", wiener, exchange, ntracks=16)


def test_testset_four_gloo_ranks_on_one_device():
    _run(4, "gloo", False, "sendrecv", ntracks=12)


def test_bench_gpus8_child_process_prints_one_verified_line():
    """Item 1(c): `python bench.py --gpus 8 --workload testset50 --tracks 8` as the driver starts it, eight ranks over gloo
    on the one device of the test box; started from a process that has not touched the GPU.  (Eight tracks = 36 work items,
    two rounds per rank: eight ranks' workspaces -- 14 GB each for the stacked pass of the sharded and of the per-track path
    -- plus three copies of the stems per rank in the all-gather arm have to share ONE device's 288 GB here.)"""
    env = dict(os.environ, XSQ_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "testset50", "--tracks", "8",
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=1500, cwd=ROOT)
    assert r.returncode == 0, "\n".join([ln for ln in r.stderr.splitlines() if "Error" in ln][:6]) + "\n...\n" + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["n_gpus"] == 8 and d["steps"] == 1
    assert d["value"] > 0
    assert d["verified"]["bitwise"] is True and len(d["verified"]["tracks"]) >= 3
    c = d["collective"]
    assert c["world"] == 8 and c["replicas"]["identical_on_all_ranks"] is True and len(c["devices"]) == 8
    assert c["exchange"] == "sendrecv-inplace"
    assert "single_rank_same_workload" in d["variants"]


def test_bench_gpus2_testset50_child_process_prints_one_verified_line():
    """The driver's own path for N > 1, on the one box the tests have: `python bench.py --gpus 2 --workload testset50` starts
    a torch.distributed.run child (before anything of ITS process touches the GPU), the ranks share device 0 over gloo,
    rank 0 prints one JSON line.  The line must carry the contract's fields, `verified.bitwise`, identical replicas and a
    recorded exchange decision."""
    env = dict(os.environ, XSQ_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "testset50", "--steps", "1",
                        "--warmup", "1", "--no-variants"], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, "\n".join([ln for ln in r.stderr.splitlines() if "Error" in ln][:6]) + "\n...\n" + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "strong"
    assert "configs[3]" in d["config"]["workload"] and "254 chunk" in d["config"]["workload"]
    assert d["value"] > 0 and abs(d["value"] - 13514.0 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    v = d["verified"]
    assert v["bitwise"] is True and len(v["tracks"]) >= 3 and v["track_crossing_2^32_flat_elements"] in v["tracks"]
    c = d["collective"]
    assert c["world"] == 2 and c["replicas"]["identical_on_all_ranks"] is True
    assert c["exchange"] == "sendrecv-inplace" and c["exchange_requested"] == "sendrecv" and c["exchange_note"]
    assert len(c["devices"]) == 2
