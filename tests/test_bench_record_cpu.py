"""The bench line is a record the driver has to be able to keep: short, strict JSON, the contract's keys.

bench.compact_record() is the only producer of bench.py's last stdout line; here it is fed the full detail dictionaries of
committed runs (the 24.6 KB line that round 5's driver could not parse among them) and the N > 1 shape."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

DETAILS = [f for f in ("profiles/r05zd_bench.json", "profiles/r04m_bench.json", "profiles/r05zd_bench_2ranks_host.json") if os.path.exists(os.path.join(ROOT, f))]


def _check(line):
    assert "\n" not in line and len(line) < bench.RECORD_LIMIT
    rec = json.loads(line, parse_constant=lambda c: pytest.fail(f"non-strict JSON constant {c}"))
    for k in bench.REQUIRED_KEYS:
        assert k in rec, k
    assert isinstance(rec["value"], float) and rec["value"] > 0 and rec["higher_is_better"] is True and rec["vs_baseline"] is None
    assert rec["dtype"] == "f64" and rec["data"] == "synthetic" and rec["unit"] == "obs/s"
    assert set(rec["config"]) >= {"workload", "camera_dof", "lm_iterations_per_step", "sharding", "comm"} and "model" not in rec["config"]
    rf = rec["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    for k in ("achieved", "peak", "frac", "avg_launch_us"):
        assert isinstance(rf[k], float), k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 * rf["frac"]
    assert "traffic" in rf and "kernel" in rf
    # nothing but numbers, booleans, nulls and short identifiers
    def walk(o, depth=0):
        assert depth <= 2
        for k, v in o.items():
            if isinstance(v, dict):
                walk(v, depth + 1)
            else:
                assert v is None or isinstance(v, (bool, int, float)) or (isinstance(v, str) and len(v) <= 200), (k, v)
    walk(rec)
    return rec


@pytest.mark.parametrize("path", DETAILS)
def test_record_of_a_committed_run_is_compact(path):
    d = json.loads([l for l in open(os.path.join(ROOT, path)).read().splitlines() if l.startswith("{")][-1])
    rec = _check(bench.record_line(d))
    assert rec["value"] == pytest.approx(d["value"], rel=1e-5) and rec["ms_per_step"] == pytest.approx(d["ms_per_step"], rel=1e-5)
    if d["n_gpus"] == 1:
        assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["cores"] >= 1 and rec["cpu_baseline"]["value"] > 0
        assert rec["roofline_hbm"]["bound"] == "hbm"


def test_record_survives_nan_and_long_strings():
    d = json.load(open(os.path.join(ROOT, DETAILS[0])))
    d = bench._jsonable(d)
    d["roofline"]["traffic"] = float("nan"); d["roofline"]["achieved"] = float("inf"); d["config"]["workload"] = "x" * 5000
    d["kernels"]["k_schur_gram"]["avg_us"] = float("nan")
    line = bench.record_line(d)
    assert len(line) < bench.RECORD_LIMIT and "NaN" not in line and "Infinity" not in line
    json.loads(line)


def test_record_refuses_a_line_without_the_contract_keys():
    with pytest.raises(ValueError):
        bench.record_line({"metric": "x"})


def test_kernel_source_digest_is_stable():
    a = bench.kernel_source_digest(); assert len(a) == 16 and a == bench.kernel_source_digest()
