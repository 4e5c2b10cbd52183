#!/usr/bin/env python3
"""Cold start of the drop-in: a FRESH process from `import torch` to the first stems of a 10 s clip, through the reference's
own entry point ``Separator.load(model_path=<dir with xumx_slicq_v2.json + .pth>)`` (/root/reference/xumx_slicq_v2/
separator.py:50-93), split into the phases a user waits for.  bench.py runs this as a child process (`variants.cold_start`);
the model directory is written by the caller (``--make-dir``: seeded synthetic weights in the reference's layout) and is not
part of the measured time.

    python3 tools/cold_start.py --make-dir /tmp/xsq_model        # once: json + torch.save'd state_dict
    python3 tools/cold_start.py --model-path /tmp/xsq_model      # -> one JSON line
"""
import argparse
import json
import os
import sys
import time

T0 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_dir(path):
    import torch
    from xumx_slicq_amd.plan import build_plan
    from xumx_slicq_amd.weights import seeded_state_dict
    os.makedirs(path, exist_ok=True)
    plan = build_plan()
    sd = seeded_state_dict([(F, T) for (_, F, T) in plan.blocks], seed=1234)
    torch.save(sd, os.path.join(path, "xumx_slicq_v2.pth"))
    json.dump({"args": {"fscale": "bark", "fbins": 262, "fmin": 32.9, "sample_rate": 44100.0, "seq_dur": 2.0, "realtime": False}},
              open(os.path.join(path, "xumx_slicq_v2.json"), "w"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--make-dir", default=None)
    ap.add_argument("--model-path", default=None)
    ap.add_argument("--seconds", type=float, default=10.0)
    args = ap.parse_args()
    if args.make_dir:
        make_dir(args.make_dir)
        return
    marks = [("process_start", T0)]

    def mark(name):
        marks.append((name, time.perf_counter()))

    import torch
    mark("import_torch")
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    mark("hip_runtime_and_first_allocation")
    from xumx_slicq_amd.separator import Separator
    from xumx_slicq_amd.synth import synth_audio
    mark("import_package_and_library")
    x = synth_audio(int(args.seconds * 44100), seed=5).cuda()
    torch.cuda.synchronize()
    mark("input_clip_to_device")
    sep = Separator.load(model_path=args.model_path, device="cuda")
    torch.cuda.synchronize()
    mark("separator_load: plan (windows, dual windows), json + torch.load, load_state_dict")
    dev = x.device
    sep.xumx_model._model(dev)
    torch.cuda.synchronize()
    mark("xsq_model_create: pack parameters, BatchNorm fold, fp64 Winograd transform of layers 2 / 3, upload")
    y = sep(x)
    torch.cuda.synchronize()
    mark("first_forward: DFT matrices + tables upload, workspaces, code objects of every kernel, the pass itself")
    y2 = sep(x)
    torch.cuda.synchronize()
    mark("second_forward (warm)")
    assert y.shape == (4, 1, 2, x.shape[-1]) and torch.equal(y, y2) and bool(torch.isfinite(y).all())
    phases = [{"phase": b[0], "ms": round((b[1] - a[1]) * 1e3, 1)} for a, b in zip(marks, marks[1:])]
    first = next(i for i, p in enumerate(phases) if p["phase"].startswith("first_forward"))
    total = sum(p["ms"] for p in phases[:first + 1])
    ours = sum(p["ms"] for p in phases[2:first + 1])
    print(json.dumps({"cold_start_ms": round(total, 1), "cold_start_ms_without_torch_and_hip_init": round(ours, 1),
                      "clip_seconds": args.seconds, "phases": phases}))


if __name__ == "__main__":
    main()
