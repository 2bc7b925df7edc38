"""Host logic of bench.py / train_bench.py that needs no device: the parity gate that makes the process exit non-zero, the 8-GPU
prediction arithmetic, the product's pipeline default."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gate_failures_cover_the_headline_and_every_sub_record():
    b = _bench()
    assert b.PARITY_MIN_DB == 80.0 and b.PIT_GATE_DB == 1e-3            # the bars the bench line is gated on (BASELINE north_star)
    ok = {"parity_ok": True, "alt_precision": {"parity_ok": True}, "large": {"parity_ok": True},
          "train": {"bf16": {"value": 1.0}, "bf16x3": {"error": "rc -6"}}}
    assert b.gate_failures(ok) == []
    assert b.gate_failures({}) == []                                     # a line without sub-records (N > 1, --mode train) has no gate to fail
    bad = dict(ok, parity_ok=False, parity_db_vs_golden=61.0, pit_si_snr_max_abs_delta_db=0.2)
    assert len(b.gate_failures(bad)) == 1 and "headline" in b.gate_failures(bad)[0]
    worse = dict(bad, large={"parity_ok": False, "parity_db_vs_golden": 40.0}, alt_precision={"parity_ok": False})
    assert [m.split(":")[0] for m in b.gate_failures(worse)] == ["headline", "alt_precision", "large"]
    assert b.gate_failures(dict(ok, train={"bf16": {"parity_ok": False, "parity_note": "x"}})) == ["train.bf16: x"]


def test_dp8_prediction_arithmetic():
    """Ring all-reduce over point-to-point xGMI: 2 (N-1)/N bytes / link bandwidth; the captured step does not overlap it; the
    1-rank floor this run already contains is not counted twice."""
    from sepreformer_amd import train_bench as tb
    assert tb.XGMI_LINK_GBS == 153
    p = tb.dp8_prediction(step_ms=80.0, batch=16, ar_bytes=58_766_336, ar_ms_here=0.02, world_here=1, overlapped=False)
    ring = 2.0 * 7 / 8 * 58_766_336 / 153e9 * 1e3
    assert abs(p["allreduce_ms_ring_model"] - round(ring, 3)) < 1e-9 and 0.6 < ring < 0.7
    assert abs(p["predicted_step_ms"] - round(80.0 + ring - 0.02, 3)) < 1e-9
    assert p["predicted_utt_per_s"] == round(8 * 16 / p["predicted_step_ms"] * 1e3, 1)
    assert 0.99 < p["predicted_scaling_efficiency"] < 1.0 and p["n_gpus"] == 8
    half = tb.dp8_prediction(80.0, 16, 58_766_336, None, 2, True)        # eager step: half of the all-reduce hides under the backward
    assert abs(half["predicted_step_ms"] - round(80.0 + 0.5 * ring, 3)) < 1e-9 and half["allreduce_ms_measured_here"] is None
    assert p["allreduce_ms_direct_lower_bound"] < p["allreduce_ms_ring_model"]


def test_pipeline_default_is_the_product_mode(monkeypatch):
    """Model.pipelines = 0 (auto): two half-batch pipelines from 16 utterances up, one below - the mode bench.py times; SEPR_PIPELINES
    overrides it, a fixed value wins over auto."""
    from sepreformer_amd.config import VARIANTS
    from sepreformer_amd.model import Model
    monkeypatch.delenv("SEPR_PIPELINES", raising=False)
    m = Model.from_config(VARIANTS["tiny"], init_seed=0)
    assert m.pipelines == 0
    assert [m.effective_pipelines(b) for b in (1, 15, 16, 32, 33)] == [1, 1, 2, 2, 2]
    m.pipelines = 3
    assert m.effective_pipelines(32) == 3 and m.effective_pipelines(2) <= 2
    monkeypatch.setenv("SEPR_PIPELINES", "1")
    assert Model.from_config(VARIANTS["tiny"], init_seed=0).pipelines == 1


def test_summary_is_last_key_and_small():
    """Round 5 (review item 5c): the line ends with a `summary` object of at most 1 500 characters that carries every headline number, so a
    consumer that keeps only the tail of the ~17 KB line still sees them.  Checked on the committed line of the round."""
    import json
    import bench
    path = os.path.join(ROOT, "profiles", "r06_v8_bench.json")
    line = open(path).read().strip().split("\n")[-1]
    rec = json.loads(line)
    assert list(rec)[-1] == "summary"
    s = rec["summary"]
    assert len(json.dumps(s)) <= 1500 and line.rstrip().endswith(json.dumps(s) + "}")
    assert s["utt_s"] == rec["value"] and s["parity_ok"] is True and s["large"]["utt_s"] == rec["large"]["value"]
    assert set(s["train"]) == {"bf16x3", "bf16", "bf16_b32"} and all(t["steps"] >= 8 for t in s["train"].values()) and s["large"]["steps"] >= 10
    assert all(t["tn_hbm"][1] is not None and t["mid"][0] is not None for t in s["train"].values())     # live PMC traffic + middle-kernel time for all three
    assert len(json.dumps(bench.make_summary(rec))) <= 1500
