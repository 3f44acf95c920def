"""bench.py's stdout contract: ONE strict-JSON line, short enough for the driver to parse out of its stdout tail
(BENCH_r04.json had `parsed: null` for a 22 KB line)."""
import json
import os
import subprocess
import sys

import bench_emit

from conftest import ROOT


def _fat(n_extra=40):
    """a full-size result object shaped like round 4's (profiles/r04_bench.json) plus padding legs and non-finite floats"""
    with open(os.path.join(ROOT, "profiles", "r04_bench.json")) as f:
        out = json.load(f)
    assert len(json.dumps(out)) > 20000
    for i in range(n_extra):
        out["extra_leg_%d" % i] = {"workload": "x" * 300, "ms_per_step": 1.0 / 3.0, "nested": {"a": [1.0] * 50}}
    out["encode"]["images_per_s_f16"] = float("nan")
    out["roofline"]["traffic"] = float("inf")
    out["strong_scaling"] = {"configs2_nuswide_map": {"pairs_per_s": 1e12, "ms_per_step": 0.5, "mAP": 0.3, "workload": "w" * 200},
                             "configs4_topk_10M_256bit": {"legs": {"Q%d" % q: {"pairs_per_s": 1e10, "ms_per_call": 0.1, "gallery_GBps": 4000.0,
                                                                               "first_hit": [0, 1]} for q in (1, 8, 64)}}}
    return out


def _strict(line):
    def bad(c):
        raise ValueError("non-finite constant %s in the bench line" % c)
    return json.loads(line, parse_constant=bad)


def test_compact_line_is_short_strict_json_with_the_contract_keys():
    out = _fat()
    line = bench_emit.compact_line(out)
    assert "\n" not in line
    assert len(line.encode()) <= bench_emit.MAX_LINE_BYTES < 8192
    d = _strict(line)
    for k in bench_emit.CONTRACT_KEYS:
        assert k in d, k
    assert d["value"] == float("%.6g" % out["value"]) and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["config"]["workload"].startswith("configs[1]")
    r = d["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms"):
        assert k in r, k
    assert r["traffic"] is None                                  # inf -> null, never the bare token Infinity
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert d["summary"]["encode"]["images_per_s_f16"] is None    # nan -> null
    assert "strong" in d["summary"] and set(d["summary"]["strong"]["cfg4_topk"]) == {"Q1", "Q8", "Q64"}


def test_compact_line_sheds_summary_entries_before_it_exceeds_the_cap(monkeypatch):
    out = _fat()
    monkeypatch.setattr(bench_emit, "MAX_LINE_BYTES", 2048)
    line = bench_emit.compact_line(out)
    assert len(line.encode()) <= 2048
    d = _strict(line)
    assert "roofline" in d and "cpu_baseline" in d and all(k in d for k in bench_emit.CONTRACT_KEYS)


def test_bench_dry_run_prints_exactly_one_parseable_line():
    """bench.py itself, through the same emitter (dry run: no GPU work)"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [x for x in p.stdout.splitlines() if x.strip()]
    assert len(lines) == 1
    d = _strict(lines[0])
    assert d["dry_run"] is True and len(lines[0]) < 8192
