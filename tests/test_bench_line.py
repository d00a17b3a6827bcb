"""bench.py's last stdout line: at most 4 096 bytes, carrying the keys the driver checks (CPU test, no GPU).

Round 5's single line had grown to 23 KB and the driver's record came back ``parsed: null``; the records of that round
(``profiles/r5_bench.json``: the --gpus 1 line; ``profiles/r5_bench_2ranks_shared_gpu.jsonl``: the N > 1 lines) are the inputs
here, plus a synthetic worst case with every string blown up."""
import copy
import json
import os

import pytest

import bench_line

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
               "dtype", "data", "config", "roofline", "cpu_baseline"}


def _records():
    out = [("r5_bench", json.load(open(os.path.join(ROOT, "profiles", "r5_bench.json"))))]
    with open(os.path.join(ROOT, "profiles", "r5_bench_2ranks_shared_gpu.jsonl")) as f:
        for i, l in enumerate(l for l in f if l.strip().startswith("{")):
            out.append(("r5_2ranks_%d" % i, json.loads(l)))
    return out


@pytest.mark.parametrize("name,full", _records(), ids=[n for n, _ in _records()])
def test_compact_line_fits_and_has_the_driver_keys(name, full):
    assert len(json.dumps(full)) > bench_line.LIMIT            # the inputs are the lines that were too long
    line = bench_line.compact(full)
    text = json.dumps(line)
    assert len(text) <= bench_line.LIMIT and "\n" not in text
    assert json.loads(text) == line
    missing = DRIVER_KEYS - set(line)
    assert not missing, missing
    assert line["vs_baseline"] is None and line["higher_is_better"] is True
    assert line["value"] == pytest.approx(full["value"], rel=1e-5) and line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert "workload" in line["config"] and "model" not in line["config"]
    roof = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-4)
    for k in ("value", "unit", "cores", "kind"):
        assert k in line["cpu_baseline"], k
    assert all(not isinstance(v, (dict, list)) for v in line["highlights"].values())
    assert line["full"] == bench_line.FULL_NAME


def test_n1_line_names_the_step_and_its_checks():
    full = json.load(open(os.path.join(ROOT, "profiles", "r5_bench.json")))
    line = bench_line.compact(full)
    assert line["parity"]["ok"] is True
    assert line["config"]["exec"] == "hipgraph" and line["config"]["edges_counted_per_step"] == 53937
    assert "frac" in line["roofline_bwd"] and "traffic" in line["roofline_bwd"]
    hl = line["highlights"]
    for k in ("pair.ms_per_step", "synth.fwd_frac_hbm", "synth.bwd_frac_hbm", "synth.fwd_bf16_frac_hbm", "sim.mfma_frac_of_f32_peak",
              "scoring.scored_triples_per_s", "sharded.ms_per_step"):
        assert k in hl, k


def test_worst_case_strings_still_fit():
    full = copy.deepcopy(json.load(open(os.path.join(ROOT, "profiles", "r5_bench.json"))))

    def blow(o):
        for k, v in o.items():
            if isinstance(v, str):
                o[k] = v + " " + "x" * 3000
            elif isinstance(v, dict):
                blow(v)
    blow(full)
    full["metric"], full["unit"] = "gnn_layer_edges_per_s", "edges/s"
    line = bench_line.compact(full)
    assert len(json.dumps(line)) <= bench_line.LIMIT
    assert DRIVER_KEYS <= set(line)


def test_floats_are_rounded_not_dropped():
    assert bench_line._sig(1.824197033) == 1.8242
    assert bench_line._sig(29567890.123) == 29567900.0
    assert bench_line._sig(True) is True and bench_line._sig(7) == 7 and bench_line._sig(float("nan")) is None


def test_write_full_round_trips(tmp_path, monkeypatch):
    monkeypatch.setattr(bench_line, "ROOT", str(tmp_path))
    os.mkdir(tmp_path / "gpurun_out")
    full = {"metric": "m", "value": 1.0, "sharded": {"scaling_model": {"band": list(range(100))}}}
    rel = bench_line.write_full(full)
    assert rel == bench_line.FULL_NAME
    for p in (tmp_path / bench_line.FULL_NAME, tmp_path / "gpurun_out" / bench_line.FULL_NAME):
        assert json.load(open(p)) == full
