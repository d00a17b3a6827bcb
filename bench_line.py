"""The ONE line bench.py ends with, and the file everything else goes to.

The driver keeps a bounded tail of stdout and parses its last line: round 5's single line had grown to 23 KB and was cut
(BENCH_r05.json: parsed = null).  ``compact`` builds the line the driver checks -- the bench contract's keys, ``roofline``,
``roofline_bwd``, ``cpu_baseline``, ``parity.ok`` and a handful of scalar highlights -- from the full record, rounds every float
to six significant digits and guarantees ``len(json.dumps(line)) <= LIMIT``; ``write_full`` puts the whole record (every side
measurement, the scaling models, the wall-clock sections) into ``bench_full.json`` beside bench.py (and into ``gpurun_out/`` when
that directory exists, so it comes back from the GPU box).  No torch, no GPU: covered by tests/test_bench_line.py on CPU.
"""
import json
import os

LIMIT = 4096            # bytes of the last stdout line, hard bound (tests/test_bench_line.py)
ROOT = os.path.dirname(os.path.abspath(__file__))
FULL_NAME = "bench_full.json"

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
CONFIG_KEYS = ("workload", "exec", "bwd_mode", "optimizer", "edges_counted_per_step", "batch_reuse",
               "global_entities", "global_triples", "parallelism", "wire", "exchange", "scaling_base")
ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_in_step", "frac_back_to_back", "traffic",
             "algorithmic_bytes_per_launch", "avg_launch_ms", "in_step_launch_ms", "source")
ROOF_BWD_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "avg_launch_ms")
CPU_KEYS = ("value", "unit", "cores", "cpu_model", "kind", "sample", "port_vs_reference_cost_ratio")
# (name in the compact line, path in the full record)
HIGHLIGHTS = (("ms_per_step_torch_adam", ("ms_per_step_torch_adam",)),
              ("pair.ms_per_step", ("pair", "ms_per_step")),
              ("union.train_ms_per_step", ("union", "train_mode", "ms_per_step")),
              ("union.bwd_frac_hbm", ("union", "train_mode", "roofline_bwd", "frac")),
              ("union.bwd_traffic_over_algorithmic", ("union", "train_mode", "roofline_bwd", "traffic_over_algorithmic")),
              ("synth.fwd_frac_hbm", ("synth", "fwd_frac_hbm")),
              ("synth.bwd_frac_hbm", ("synth", "bwd_frac_hbm")),
              ("synth.fwd_bf16_frac_hbm", ("synth", "fwd_bf16_frac_hbm")),
              ("sim.mfma_frac_of_f32_peak", ("sim", "mfma_frac_of_f32_peak")),
              ("scoring.scored_triples_per_s", ("scoring", "scored_triples_per_s")),
              ("scoring.whole_split_triples_per_s", ("scoring", "whole_split", "scored_triples_per_s")),
              ("sharded.ms_per_step", ("sharded", "ms_per_step")),
              ("sharded.rehearsal_world8.agg_ms", ("sharded", "rehearsal_world8", "step_ms")),
              ("sharded.speedup8_rehearsed", ("sharded", "scaling_model", "predicted", "8", "speedup_vs_1")),
              ("gpu_over_cpu", ("gpu_over_cpu",)),
              ("replicas.value", ("replicas", "value")),
              ("comm.backend", ("comm", "backend")),
              ("comm.world_seen_by_all_reduce", ("comm", "world_seen_by_all_reduce")),
              ("comm.collective_ms_per_step", ("comm", "collective_ms_per_step")),
              ("bench_wall_s", ("bench_wall_s", "total")))


def _sig(x, n=6):
    """Floats to n significant digits (ints, bools, strings, None unchanged)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (n, x))


def _clip(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[: n - 1] + "~"


def _pick(src, keys, strlen):
    out = {}
    for k in keys:
        if k not in src or isinstance(src[k], (dict, list)):
            continue
        if src[k] is None and k != "traffic":              # "traffic": null is a statement (no PMC pass), the rest is noise
            continue
        out[k] = _clip(_sig(src[k]), strlen)
    return out


def _get(d, path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d if not isinstance(d, (dict, list)) else None


def compact(full, full_path=FULL_NAME, strlen=200):
    """The driver's line from the full record: a dict with ``len(json.dumps(.)) <= LIMIT``."""
    for attempt_strlen in (strlen, 120, 72, 40):
        line = {k: _clip(_sig(full[k]), attempt_strlen) for k in CONTRACT if k in full}
        line["config"] = _pick(full.get("config") or {}, CONFIG_KEYS, attempt_strlen)
        if full.get("roofline"):
            line["roofline"] = _pick(full["roofline"], ROOF_KEYS, attempt_strlen)
            line["roofline"].setdefault("traffic", None)
        if full.get("roofline_bwd"):
            line["roofline_bwd"] = _pick(full["roofline_bwd"], ROOF_BWD_KEYS, attempt_strlen)
        cb = full.get("cpu_baseline")
        if isinstance(cb, dict):
            line["cpu_baseline"] = _pick(cb, CPU_KEYS, attempt_strlen) if "error" not in cb else {"error": _clip(str(cb["error"]), attempt_strlen)}
        par = full.get("parity")
        if isinstance(par, dict):
            line["parity"] = {k: _sig(par[k]) for k in ("ok", "tol", "loss_rel_err", "align_out_rel_err", "comp_layer1_rel_err") if k in par}
        hl = {}
        for name, path in HIGHLIGHTS:
            v = _get(full, path)
            if v is not None:
                hl[name] = _clip(_sig(v), 48)
        if hl:
            line["highlights"] = hl
        line["full"] = full_path
        if len(json.dumps(line)) <= LIMIT:
            return line
    # last resort: the contract keys and the three objects without strings longer than 40 characters, no highlights
    line.pop("highlights", None)
    if len(json.dumps(line)) > LIMIT:          # pragma: no cover
        raise ValueError("bench line does not fit %d bytes" % LIMIT)
    return line


def write_full(full, name=FULL_NAME):
    """The whole record, beside bench.py and (when present) under gpurun_out/.  Returns the path named in the compact line."""
    text = json.dumps(full, indent=1)
    path = os.path.join(ROOT, name)
    wrote = None
    for p in (path, os.path.join(ROOT, "gpurun_out", name)):
        try:
            if os.path.isdir(os.path.dirname(p)):
                with open(p, "w") as f:
                    f.write(text + "\n")
                wrote = wrote or p
        except OSError:
            pass
    return os.path.relpath(wrote, ROOT) if wrote else None
