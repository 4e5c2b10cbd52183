"""Counterpart of /root/reference/xumx_slicq_v2/inference.py for the ROCm backend:
``separate`` (inference.py:14-33, same timing convention: wall time of the
``separator(audio)`` call only) and a small CLI (``python -m xumx_slicq_amd``)."""
from __future__ import annotations

import argparse
import time
from pathlib import Path

import torch

from . import audio as xaudio
from .separator import Separator, seeded_separator


def separate(audio, separator, rate=None, device=None):
    """inference.py:14-33: returns ({target: (nb_samples, 2, T)}, seconds)."""
    if rate is None:
        raise Exception("rate` must be provided.")
    if device:
        audio = audio.to(device)
    audio = xaudio.preprocess_audio(audio, rate, separator.sample_rate)
    torch.cuda.synchronize(audio.device)
    start_time = time.time()
    estimates = separator(audio)
    torch.cuda.synchronize(audio.device)
    time_delta = time.time() - start_time
    return separator.to_dict(estimates), time_delta


class _PinnedPool:
    """Pinned host buffers handed round between the pipeline's threads (hipHostMalloc of a 339 MB stem block costs more
    than demixing the track: buffers are allocated once per size class and reused)."""

    def __init__(self):
        import queue
        self._free = queue.Queue()
        self._count = 0

    def take(self, numel: int, limit: int):
        """A pinned fp32 buffer of >= numel elements; blocks when `limit` buffers are out and none is free."""
        import queue
        while True:
            try:
                buf = self._free.get(block=self._count >= limit)
            except queue.Empty:
                buf = None
            if buf is None:
                self._count += 1
                return torch.empty(numel, dtype=torch.float32, pin_memory=True)      # (allocated pinned: no pageable twin, no copy)
            if buf.numel() >= numel:
                return buf
            self._count -= 1          # too small for this track: drop it, allocate a larger one
            del buf

    def give(self, buf):
        self._free.put(buf)


_POOLS: dict = {}        # process-wide pinned staging pools of demix_directory


def demix_directory(separator, wavs, out_dir, device="cuda", readers: int = 3, writers: int = 4, depth: int = 3, quiet=False):
    """The CLI's loop (inference.py:118-146) as a pipeline over the tracks: decode -> pinned host buffer (reader threads) |
    H2D on a copy stream | ``separator(audio)`` | channel interleave on the GPU (the wav payload layout, so the host never
    transposes 339 MB per track) | D2H into a pinned buffer on a second copy stream | header + payload written by writer
    threads.  At ~5 ms of GPU time per 240 s track the loop is bound by PCIe and file I/O; the stages of consecutive
    tracks overlap.  Returns [(name, audio seconds, separator milliseconds by HIP events)] in input order."""
    import queue
    import threading
    from concurrent.futures import ThreadPoolExecutor

    dev = torch.device(device)
    out_dir = Path(out_dir)
    # the staging buffers outlive the call (page-locking a 339 MB block costs more than demixing the track it carries)
    pool_in, pool_out = _POOLS.setdefault("in", _PinnedPool()), _POOLS.setdefault("out", _PinnedPool())
    copy_in, copy_out = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    results, errors = {}, []
    wq: "queue.Queue" = queue.Queue()

    def read(path):
        keep = []

        def take(numel):
            keep.append(pool_in.take(numel, depth + readers))
            return keep[0]
        view, rate = xaudio.load_audio_into(str(path), take)        # (2, N) float32 in the pinned buffer, decoded in one pass
        if float(rate) != float(separator.sample_rate):
            raise ValueError(f"{path}: {rate} Hz, the model runs at {float(separator.sample_rate)} Hz: resample first")
        return path, view, keep[0], rate

    lock = threading.Lock()

    def writer():
        # one job = ONE stem of one track: the four stems of a track go out on four threads at once (a track's last write is what
        # the end of a run waits for); the track's pinned buffer returns to the pool with its last stem
        while True:
            job = wq.get()
            if job is None:
                return
            path, host, buf, done, t0, t1, rate, n, k, left = job
            try:
                done.synchronize()
                target_dir = out_dir / path.stem
                target_dir.mkdir(parents=True, exist_ok=True)
                xaudio.save_wav_float_interleaved(str(target_dir / f"{separator.sources[k]}.wav"), host[k], rate)
            except Exception as e:                                   # noqa: BLE001 -- reported after the loop
                errors.append((str(path), e))
            finally:
                with lock:
                    left[0] -= 1
                    last = left[0] == 0
                if last:
                    results[str(path)] = (path.name, n / rate, t0.elapsed_time(t1))
                    pool_out.give(buf)

    threads = [threading.Thread(target=writer, daemon=True) for _ in range(writers)]
    for t in threads:
        t.start()
    wavs = [Path(w) for w in wavs]
    with ThreadPoolExecutor(max_workers=readers) as ex:
        pending = [ex.submit(read, w) for w in wavs]
        for fut in pending:
            path, view, buf_in, rate = fut.result()
            n = view.shape[-1]
            with torch.cuda.stream(copy_in):
                x = view.to(dev, non_blocking=True)[None]            # (1, 2, N)
                up = torch.cuda.Event()
                up.record(copy_in)
            main.wait_event(up)
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record(main)
            est = separator(x)                                       # (4, 1, 2, N)
            t1.record(main)
            inter = est[:, 0].transpose(1, 2).contiguous()           # (4, N, 2): the wav payload of each target
            x.record_stream(main)
            ready = torch.cuda.Event()
            ready.record(main)
            buf_out = pool_out.take(inter.numel(), depth + writers)
            host = buf_out[:inter.numel()].view(inter.shape)
            with torch.cuda.stream(copy_out):
                copy_out.wait_event(ready)
                host.copy_(inter, non_blocking=True)
                inter.record_stream(copy_out)
                done = torch.cuda.Event()
                done.record(copy_out)
            up.synchronize()                                         # the input buffer may be refilled once its upload is over
            pool_in.give(buf_in)
            left = [len(separator.sources)]
            for k in range(len(separator.sources)):
                wq.put((path, host, buf_out, done, t0, t1, rate, n, k, left))
    for _ in threads:
        wq.put(None)
    for t in threads:
        t.join()
    if errors:
        raise RuntimeError("demix_directory: %d track(s) failed, first: %s: %r" % (len(errors), errors[0][0], errors[0][1]))
    out = [results[str(w)] for w in wavs]
    if not quiet:
        for name, secs, ms in out:
            print(f"{name}: {secs:.1f} s demixed in {ms:.1f} ms")
    return out


def inference_main(argv=None):
    p = argparse.ArgumentParser(description="xumx-sliCQ-V2 inference on MI355X (hip-rocm backend)")
    p.add_argument("--input-dir", type=str, default="/input")
    p.add_argument("--output-dir", type=str, default="/output")
    p.add_argument("--ext", type=str, default=".wav")
    p.add_argument("--model-path", type=str, default=None,
                   help="directory with xumx_slicq_v2.json/.pth; omit for seeded synthetic weights")
    p.add_argument("--realtime", action="store_true")
    p.add_argument("--warmup", type=int, default=0)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--serial", action="store_true", help="one track at a time, as the reference's loop (inference.py:118-146)")
    args = p.parse_args(argv)
    if args.model_path:
        separator = Separator.load(model_path=args.model_path, runtime_backend="hip-rocm",
                                   warmup=args.warmup, realtime=args.realtime, device=args.device)
    else:
        separator = seeded_separator(realtime=args.realtime, device=args.device)
    out_dir = Path(args.output_dir)
    wavs = sorted(Path(args.input_dir).glob(f"*{args.ext}"))
    if not args.serial:
        t0 = time.time()
        done = demix_directory(separator, wavs, out_dir, device=args.device)
        wall = time.time() - t0
        if done:
            print(f"xumx-sliCQ-V2 inference time: {sum(d[2] for d in done) / len(done) / 1e3:.4f} s/track over {len(done)} track(s); "
                  f"{len(done) / wall:.2f} tracks/s end to end (decode, H2D, demix, D2H, encode)")
        return
    tot, n = 0.0, 0
    for wav in wavs:
        sig, rate = xaudio.load_audio(str(wav))
        estimates, dt = separate(sig, separator, rate=rate, device=args.device)
        tot, n = tot + dt, n + 1
        target_dir = out_dir / wav.stem
        target_dir.mkdir(parents=True, exist_ok=True)
        for target, est in estimates.items():
            xaudio.save_wav_float(str(target_dir / f"{target}.wav"), est[0], rate)
        print(f"{wav.name}: {sig.shape[-1] / rate:.1f} s demixed in {dt * 1e3:.1f} ms")
    if n:
        print(f"xumx-sliCQ-V2 inference time: {tot / n:.4f} s/track over {n} track(s)")


if __name__ == "__main__":
    inference_main()
