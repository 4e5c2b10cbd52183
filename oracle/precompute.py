"""The two CPU-heavy oracle results of the GPU suite, as a job that tests/conftest.py starts in the BACKGROUND when the
session is collected, so that their host time (2-3 minutes) runs beside the other tests instead of in front of them.

TEST INFRASTRUCTURE (see oracle/__init__.py): only tests/ start this.

    python -m oracle.precompute fullsize <out.pt>    the bench's 240 s track (10,584,000 samples, seed 20260101) through the
                                                     oracle's literal chunk loop, mix-phase AND Wiener-EM from ONE forward
                                                     transform and ONE set of CDAE masks per chunk (the masks do not depend on
                                                     the post-filter: model.py:264-268)
    python -m oracle.precompute train16 <out.pt>     the oracle's autograd on bench.py's configs[4] batch (B = 16 x 88,200)
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FULL_N, FULL_SEED = 10_584_000, 20260101
RISK16 = 3e-6      # |BatchNorm output| below this may land on the other side of the ReLU kink in another summation order


def _plan_sd():
    from oracle import slicqt as oslicqt
    from xumx_slicq_amd.weights import seeded_state_dict
    plan = oslicqt.make_plan()
    return plan, seeded_state_dict([(F, T) for (_, F, T) in plan.blocks], seed=1234)


def fullsize(plan=None, sd=None):
    """{False: stems with mix-phase, True: stems with Wiener-EM}, each (4, 1, 2, FULL_N)."""
    from oracle import separator as osep
    from xumx_slicq_amd.synth import synth_audio
    if plan is None:
        plan, sd = _plan_sd()
    x = synth_audio(FULL_N, seed=FULL_SEED)
    pm, wi = osep.separate_both(plan, sd, x, causal=False)
    return {False: pm, True: wi}


def inputs16(n=88200, B=16):
    from xumx_slicq_amd.synth import synth_audio
    y_t = torch.stack([0.5 * synth_audio(n, seed=700 + j, nb_samples=B) for j in range(4)])      # bench.py's batch
    return y_t.sum(0), y_t


def train16(plan=None, sd=None):
    from oracle import loss as oloss
    if plan is None:
        plan, sd = _plan_sd()
    x, y_t = inputs16()
    minima = {}
    _loss, mse, msk, grads = oloss.training_gradients(plan, sd, x, y_t, causal=False, wiener=True, minima=minima)
    risk = sorted({k.rsplit(".", 1)[0] for k, v in minima.items() if v < RISK16})          # "sliced_umx.<b>.cdaes.<t>"
    return {"mse": mse, "msk": msk, "grads": {k: v.detach() for k, v in grads.items()}, "risk": risk}


def main():
    what, out = sys.argv[1], sys.argv[2]
    res = {"fullsize": fullsize, "train16": train16}[what]()
    tmp = out + ".tmp"
    torch.save(res, tmp)
    os.replace(tmp, out)             # the reader never sees a half-written file


if __name__ == "__main__":
    main()
