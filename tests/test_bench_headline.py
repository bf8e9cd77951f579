"""bench.py's last stdout line is what the driver parses: ONE compact, strict JSON object under 4 KB with the contract's keys,
`roofline` and `cpu_baseline`; everything else goes to a side file.  (Round 5's line had grown to 20.5 KB and the driver's record
came back `parsed: null`.)  The emitter is run here on canned numbers — no GPU."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _canned(n_other=14, long_text=400):
    pad = "x" * long_text
    other = [{"config": f"cfg{i}", "ms_per_sweep": 0.1 * i, "workload": pad, "roofline": {"bound": "hbm", "frac": 0.5, "frac_note": pad, "achieved": 1.0},
              "parity": {"ok": True, "max_rel_err": 1e-15, "checker": pad}, "plan": {"stages": 100001, "note": pad}} for i in range(n_other)]
    other.append({"config": "broken", "error": "RuntimeError: " + pad})
    return {
        "metric": "edge-message updates/sec per sweep, 10M-edge Gaussian grid", "value": 2.8634e11, "unit": "edge-message updates/s", "n_gpus": 1,
        "steps": 20, "warmup": 5, "ms_per_step": 0.0559, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "C4: ONE 1415x1415 2-D Gaussian grid loopy BP; " + pad, "schedule": "fused " + pad, "partition": "1 row blocks", "seed": 1234},
        "timed_regions": {"count": 5, "ms_per_step_each": [0.05] * 5, "device_ms_per_step_each": [0.05] * 5},
        "roofline": {"bound": "hbm", "kernel": "void cx::k_sweep<0, false, true, 3, 5, false>(cx::SweepArgs)", "achieved": 6790.123456789, "peak": 8000.0,
                     "unit": "GB/s", "frac": 0.848765432, "traffic": 3.762e8, "basis": "counter bytes / avg launch duration " + pad, "avg_kernel_ms": 0.05544,
                     "payload_bytes_per_launch": 320174880.0, "frac_payload": 0.72, "survey_convention_bytes_per_launch": 512207360.0,
                     "frac_survey_convention": 1.15, "bound_detail": pad, "all_kernels_sampled_ms": {"a": 1.0}},
        "marginals_per_s": 3.6e10, "max_message_change_over_run": float("nan"),
        "cpu_baseline": {"value": 8.8e6, "unit": "edge-message updates/s", "cores": 1, "kind": "port", "sample": "12 sweeps " + pad,
                         "reference_julia": "not on this box", "flooding_all_cores": {"value": 1.3e8, "unit": "edge-message updates/s", "cores": 16, "kind": "port", "sample": pad}},
        "parity": {"max_rel_err_marginals": 1e-13, "max_rel_err_messages": float("inf"), "tolerance": 1e-6, "ok": False, "sweeps": 8, "sample": pad, "checker": pad},
        "other_configs": other,
    }


@pytest.mark.parametrize("n_other,long_text", [(0, 10), (14, 400), (60, 3000)])
def test_the_headline_is_one_compact_strict_json_line(n_other, long_text):
    full = _canned(n_other, long_text)
    line = bench.headline_line(full, "gpurun_out/bench_details_n1.json")
    assert "\n" not in line
    assert len(line.encode()) < 4096
    assert "NaN" not in line and "Infinity" not in line

    def no_constants(tok):
        raise AssertionError(f"non-finite token {tok} in the headline")
    d = json.loads(line, parse_constant=no_constants)
    assert set(d) <= set(bench.HEADLINE_KEYS) | {"other_configs_digest", "weak_scaling", "halo_check"}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "parity", "details"):
        assert k in d, k
    assert d["config"]["workload"].startswith("C4")
    assert "model" not in d["config"]
    ro = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "payload_bytes_per_launch", "frac_survey_convention"):
        assert k in ro, k
    assert not any(k.startswith("frac_algorithmic") for k in ro)
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb) and cb["kind"] in ("port", "reference")
    assert d["parity"]["ok"] is False and d["parity"]["max_rel_err_messages"] is None      # inf -> null, the verdict stays
    assert d["value"] == pytest.approx(2.8634e11) and d["steps"] == 20 and d["warmup"] == 5


def test_a_multi_gpu_line_without_cpu_baseline_is_still_valid():
    full = _canned(0, 10)
    for k in ("cpu_baseline", "parity", "other_configs"):
        del full[k]
    full["n_gpus"] = 8
    full["weak_scaling"] = {"value": 1e12, "unit": "edge-message updates/s", "ms_per_step": 0.06, "workload": "w" * 500, "schedule": "s" * 500}
    full["halo_check"] = "ok: last imported halo == neighbours' packed messages, bit for bit"
    d = json.loads(bench.headline_line(full, None))
    assert d["n_gpus"] == 8 and "cpu_baseline" not in d and d["details"] is None
    assert d["weak_scaling"]["value"] == 1e12 and "workload" not in d["weak_scaling"]


def test_the_side_file_holds_everything_and_is_strict_json(tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rel = bench.write_details(_canned(3, 50), 1)
    assert rel == os.path.join("gpurun_out", "bench_details_n1.json")
    text = open(tmp_path / rel).read()
    assert "NaN" not in text and "Infinity" not in text
    d = json.loads(text)
    assert len(d["other_configs"]) == 4 and "timed_regions" in d
    err = capsys.readouterr()
    assert err.out == "" and "[bench] details:" in err.err        # stdout stays clean for the one line


def test_bench_prints_the_headline_last_and_only_once():
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def run_rank"):src.index("def main")]
    assert body.count("print(headline_line(") == 1
    assert "print(json.dumps(out)" not in body
