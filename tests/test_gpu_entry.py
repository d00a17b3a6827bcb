"""The driver's entry points in ONE fresh process, build() before smoke(): build() maps libjmac_hip.so before anything has used
torch's HIP runtime, which is the load order that once gave "no ROCm-capable device" on the library's first launch
(jmac_amd/_lib.py:lib imports torch first so that both sides share one runtime)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("code", [
    "import __graft_entry__ as g; g.build(); g.smoke()",
    "import __graft_entry__ as g; g.smoke()",
    # the library mapped before torch is even imported by the caller
    "import jmac_amd; jmac_amd.lib(); import __graft_entry__ as g; g.smoke()",
])
def test_entry_points_in_one_process(code):
    res = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:]
    assert "smoke: layer fwd rel err" in res.stdout


@pytest.mark.gpu
def test_bench_ends_with_a_line_the_driver_can_parse(tmp_path):
    """The bench contract end to end, in the driver's form (`python bench.py --gpus 1 --steps K --warmup W`, the side sections that
    only add wall time switched off): the LAST stdout line is one JSON object of at most 4 096 bytes with the contract's keys, the
    K and W it was asked for, `roofline`, `parity.ok` -- and everything else is in bench_full.json, which the line names.
    (Round 5's 23 KB line came back from the driver as `parsed: null`; tests/test_bench_line.py covers the builder on CPU.)"""
    import json
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "7", "--warmup", "2", "--no-synth", "--no-side",
                          "--no-cpu-baseline", "--no-torch-adam-leg"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    last = lines[-1]
    assert len(last) <= 4096, len(last)
    line = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in line, k
    assert line["steps"] == 7 and line["warmup"] == 2 and line["n_gpus"] == 1 and line["vs_baseline"] is None
    assert line["unit"] == "edges/s" and line["higher_is_better"] is True and line["dtype"] == "f32"
    assert abs(line["value"] - line["config"]["edges_counted_per_step"] / (line["ms_per_step"] * 1e-3)) <= 1e-4 * line["value"]
    assert line["parity"]["ok"] is True and 0.3 < line["roofline"]["frac"] < 1.0
    assert line["roofline"]["frac"] <= line["roofline"]["frac_back_to_back"] + 1e-9
    full = json.load(open(os.path.join(ROOT, line["full"])))
    assert full["ms_per_step"] == pytest.approx(line["ms_per_step"], rel=1e-5)
    for k in ("layer", "scoring", "sim", "bench_wall_s"):
        assert k in full, k
    assert full["scoring"]["whole_split"]["ranks_equal_batched"] is True
