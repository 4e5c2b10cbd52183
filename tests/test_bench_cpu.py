"""bench.py's bookkeeping on CPU: the algorithmic-work numerators, the BASELINE configs[3] workload
definition, and the lookups into the committed rocprofv3 summaries that fill `roofline.traffic` and `pmc`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from xumx_slicq_amd.plan import build_plan  # noqa: E402


def test_testset_is_fifty_seeded_lengths_in_range():
    a, b = bench.testset_lengths(), bench.testset_lengths()
    assert a == b and len(a) == 50
    assert all(150 * 44100 <= n <= 420 * 44100 for n in a)
    assert abs(sum(a) / 44100 - 13514) < 1.0          # the duration DESIGN.md / README.md quote


def test_algorithmic_work_of_the_bench_track():
    plan = build_plan()
    chunk, n = 2621440, 10584000
    w = bench.algorithmic_work(plan, 1, [chunk] * 4 + [n - 4 * chunk], wiener=True)
    # every instrumented kernel of the default path has a numerator of the right kind
    for k in ("slice_rfft", "slice_irfft_ola", "wiener_stats", "wiener_apply"):
        assert w[k][0] == "hbm" and w[k][1] > 0
    for k in ("band_analysis_dft4", "band_synthesis_dft4", "cdae_l1_gemm", "cdae_l2_slab", "cdae_l3_slab", "cdae_l4_gemm"):
        assert w[k][0] == "mfma" and w[k][1] > 0
    # the synthesis transforms four targets, the analysis one mix; layer 3 is counted like layer 2 (per input position)
    assert w["band_synthesis_dft4"][1] == 4 * w["band_analysis_dft4"][1]
    assert w["cdae_l3_slab"][1] == w["cdae_l2_slab"][1]
    # the figures DESIGN.md section 4 quotes (GFLOP per 240 s track)
    assert abs(w["cdae_l2_slab"][1] / 1e9 - 112.3) < 0.5
    assert abs(w["cdae_l1_gemm"][1] / 1e9 - 46.4) < 0.5
    # radix-4 split point: the bands at or above XSQ_D4_MIN_LG_DEFAULT = 24 carry 99 % of the band-DFT flops
    Lg = np.asarray(plan.Lg, dtype=np.int64)
    assert abs((Lg[Lg >= 24] ** 2).sum() / (Lg ** 2).sum() - 0.990) < 0.002 and int((Lg < 24).sum()) == 100
    src = open(os.path.join(os.path.dirname(bench.__file__), "xumx_slicq_amd", "csrc", "slicqt.hip")).read()
    assert "#define XSQ_D4_MIN_LG_DEFAULT 24" in src          # the number the header, bench.py and this test quote


def test_committed_profiles_fill_the_traffic_and_counter_fields():
    tr = bench.pmc_traffic("cdae_l3_slab")
    assert tr is not None
    per_step, launches = tr
    assert launches >= 1 and 0.8e9 < per_step / launches < 2.0e9      # ~1.2 GB per launch against 1.0 GB of activations
    pm = bench.pmc_issue("cdae_l3_slab")
    assert pm is not None and 0.4 < pm["mfma_busy"] <= 1.0 and 0.0 < pm["valu_issue"] < 1.0
    fft = bench.pmc_issue("slice_irfft_ola")
    assert fft is not None and fft["mfma_busy"] == 0.0 and fft["valu_issue"] > 0.4     # a vector-ALU kernel
    assert bench.pmc_traffic("no_such_kernel") is None and bench.pmc_issue("no_such_kernel") is None


def test_committed_bench_line_keeps_the_contract():
    """The newest committed bench line (profiles/*_bench.json, written by bench.py on an MI355X) against the contract: the
    metric and unit are BASELINE.json's, the workload is named, value = audio-seconds / wall-seconds of the timed steps,
    the roofline and cpu_baseline objects carry their fields, and the line says which bound the step ran at."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    files = sorted(f for f in glob.glob(os.path.join(root, "profiles", "r0*_bench.json")))
    d = json.load(open(files[-1]))
    assert d["metric"] == base["metric"] == bench.METRIC and d["unit"] == "x real-time"
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "f32" and "configs[1]" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] - bench.TRACK_SAMPLES / bench.FS / (d["ms_per_step"] * 1e-3)) < 0.002 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert d["host_enqueue_ms"] < 0.5 * d["ms_per_step"]            # GPU-bound: the host issues a step in a fraction of its time
    assert abs(d["gpu_span_ms"]["mean"] - d["ms_per_step"]) < 0.05 * d["ms_per_step"]
    for k in ("wiener", "train_step", "bf16x6", "bf16x3", "hip_graph"):
        assert k in d["variants"]


def test_issue_bound_of_the_bench_track():
    """`roofline_issue`: the issue bound of every fp32 MFMA kernel from the committed instruction budget (profiles/
    isa_budget.json, tools/isa_budget.py) and the tilings restated in bench.issue_bound.  Held here: every MFMA kernel of the
    step has a bound; the Winograd form of layers 2 / 3 lowers theirs by the ratio the instruction counts give; no bound
    exceeds the kernel's time in the newest committed bench line by more than the model's slack."""
    plan = build_plan()
    chunk, n = 2621440, 10584000
    items = [chunk] * 4 + [n - 4 * chunk]
    w = bench.issue_bound(plan, 1, items, winograd=True)
    d = bench.issue_bound(plan, 1, items, winograd=False)
    for k in ("cdae_l1_gemm", "cdae_l2_slab", "cdae_l3_slab", "cdae_l4_gemm", "band_analysis_dft4", "band_synthesis_dft4",
              "band_analysis_gemm", "band_synthesis_gemm"):
        assert w[k][0] > 0 and 0.2 < w[k][1] < 1.0, (k, w[k])
    # direct slab kernels: 9984 MFMA cycles + 4 x 367 (262) other vector instructions per tap and wave, 8 waves per 256 rows;
    # Winograd: 6240 + 4 x ~390 (~325) per tap and wave, 4 waves per 64 pairs = 128 rows
    for k in ("cdae_l2_slab", "cdae_l3_slab"):
        assert 0.70 < w[k][0] / d[k][0] < 0.82, (k, w[k], d[k])
        assert 0.80 < d[k][0] < 0.95          # ms: the direct kernels measured 1.00 ms = ~0.9 of this bound
    assert abs(bench.executed_mfma_flops(plan, 1, items, "cdae_l3_slab", True) / 1e9 - 72.9) < 0.5     # against 112.3 GFLOP algorithmic
    assert bench.executed_mfma_flops(plan, 1, items, "cdae_l1_gemm", True) is None


def test_roofline_issue_covers_the_whole_step():
    """every kernel of the step carries a vector-issue bound -- the MFMA kernels from the static
    instruction budget, the slice FFTs from the dynamic SQ_INSTS_VALU of the committed PMC pass -- and the table says what
    share of the step's kernel time it covers (>= 95 %)."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(f for f in glob.glob(os.path.join(root, "profiles", "r*_bench.json")) if "testset" not in f and "gpus" not in f and "wiener" not in f)
    d = json.load(open(files[-1]))
    prof = {k: (v["ms_per_step"], v["launches_per_step"]) for k, v in d["kernels"].items()}
    plan = build_plan()
    items = [2621440] * 4 + [10584000 - 4 * 2621440]
    table = bench.roofline_issue_table(bench.issue_bound(plan, 1, items, winograd=True), prof, 1)
    cov = table[-1]
    assert "coverage" in cov and cov["coverage"] >= 0.95, cov
    rows = {r["kernel"]: r for r in table[:-1]}
    for k in ("slice_rfft", "slice_irfft_ola"):
        assert rows[k]["source"].startswith("pmc") and 0.4 < rows[k]["frac_of_issue_bound"] < 1.0
        assert abs(rows[k]["t_issue_bound_ms"] - rows[k]["frac_of_issue_bound"] * rows[k]["ms_per_step"]) < 1e-3
    for k in ("cdae_l2_slab", "cdae_l1_gemm", "band_synthesis_dft4"):
        assert rows[k]["source"] == "isa" and 0.3 < rows[k]["frac_of_issue_bound"] < 1.0


def test_provenance_records_what_selected_the_kernels(monkeypatch):
    """`env` / `library` / `nondefault` of the JSON line: a stray XSQ_* switch or a diagnostic library must show up."""
    for k in [k for k in os.environ if k.startswith("XSQ_")]:
        monkeypatch.delenv(k)
    p = bench.provenance()
    assert p["env"] == {"xsq": {}, "selecting_kernels_or_paths": []} and p["nondefault"] is False
    assert p["library"]["abi"] == 2 and p["library"]["default_path"] and "arch=gfx950" in p["library"]["build"]
    assert "no-packed-fp32-ops" in p["library"]["build"]
    monkeypatch.setenv("XSQ_DIST_BACKEND", "gloo")                 # transport only: recorded, not flagged
    p = bench.provenance()
    assert p["env"]["xsq"] == {"XSQ_DIST_BACKEND": "gloo"} and p["nondefault"] is False
    monkeypatch.setenv("XSQ_D4_SYM", "0")                          # selects another band kernel: flagged
    p = bench.provenance()
    assert p["env"]["selecting_kernels_or_paths"] == ["XSQ_D4_SYM"] and p["nondefault"] is True
