"""The line bench.py prints is what the driver parses: it must be ONE short JSON object (round 4's 22 KB
line came back `parsed: null`, which left the round's headline unmeasured).  CPU tests of the line builder on canned
records: the round-4 record committed under profiles/, and the same record dressed up as an 8-rank run."""
import copy
import json
import os

import pytest

from conftest import ROOT

import bench


def _canned():
    return json.load(open(os.path.join(ROOT, "profiles", "r04fin_bench.json")))


def _as_world(rec, world):
    rec = copy.deepcopy(rec)
    rec["n_gpus"] = world
    rec["config"]["parallelism"] = "dp%d" % world
    rec["ranks_seen"] = [dict(rank=r, local_rank=r, device_index=r, device_uuid="GPU-%032x" % (0xabcdef0123456789 * (r + 1)),
                              pci_bus_id=5 + 16 * r, device_name="AMD Instinct MI355X", pid=100000 + r)
                         for r in range(world)]
    rec["distributed"] = {
        "backend": "nccl", "rccl_version": "2.26.6", "world_size": world, "distinct_devices": world,
        "bucket_bytes": 4790000, "allreduce_per_step": 1,
        "allreduce_us": {"p50": 61.234567, "p90": 88.7654321, "calls_timed_per_rank": 50, "how": "x" * 200},
        "step_ms_per_rank": [4.9 + 0.0001 * r for r in range(world)], "loss_per_rank": [0.0663682520389 + r for r in range(world)],
        "param_checksum_per_rank": [[-1234.567890123456, 0.123456789012345]] * world, "replicas_identical": True,
        "allreduce_in_graph": True, "allreduce_form": "one graph (captured all-reduce)",
        "n1_probe": {"n1_reference_ms": 4.81234567, "steps": 5, "how": "y" * 120},
        "launcher": "external launcher"}
    return rec


def test_one_gpu_line_is_short_and_complete():
    line = bench.compact_line(_canned())
    assert "\n" not in line and len(line) <= 8000, len(line)
    rec = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_check", "fp32_mfma_only",
                "step_ms_rank0"):
        assert key in rec, key
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rec["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(rec["cpu_baseline"])
    assert "workload" in rec["config"] and "model" not in rec["config"]
    # strings stay inside the window the driver keeps per string
    def strings(o):
        if isinstance(o, str):
            yield o
        elif isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
    assert max(len(s) for s in strings(rec)) <= 200
    step_rows = [k for k in rec["kernels"] if k.get("calls_per_step")]
    legs = [k for k in rec["kernels"] if not k.get("calls_per_step")]
    assert len(step_rows) <= 8 and len(legs) == 3
    assert rec["kernels_total"] == len(_canned()["kernels"])


@pytest.mark.parametrize("world", [2, 8])
def test_multi_gpu_line_is_short_and_complete(world):
    line = bench.compact_line(_as_world(_canned(), world))
    assert "\n" not in line and len(line) <= 12000, len(line)
    rec = json.loads(line)
    assert rec["n_gpus"] == world and len(rec["ranks"]) == world
    assert set(rec["ranks"][0]) == {"rank", "device_uuid", "device_index", "step_ms", "checksum"}
    d = rec["distributed"]
    assert d["replicas_identical"] is True and d["allreduce_in_graph"] is True and "how" not in d["allreduce_us"]
    assert abs(d["n1_probe"]["ratio"] - rec["ms_per_step"] / 4.81234567) < 1e-3
    for key in ("roofline", "cpu_baseline", "ms_per_step", "steps"):
        assert key in rec


def test_line_sheds_rows_rather_than_exceed_the_budget():
    rec = _as_world(_canned(), 8)
    line = bench.compact_line(rec, budget=5000)
    assert len(line) <= 5000
    out = json.loads(line)
    assert "roofline" in out and "cpu_baseline" in out and "ms_per_step" in out


def test_sustained_fraction_uses_the_mfma_only_roof():
    # 1448 TFLOP/s contained the GEMM stage's own LDS reads and vector work
    assert bench.SUSTAINED_F16_RANDOM_TFLOPS == 1760.0 and not hasattr(bench, "SUSTAINED_F16_MIX_TFLOPS")


def test_emit_prints_the_compact_line_last_and_writes_the_full_record(tmp_path, capsys, monkeypatch):
    monkeypatch.setenv("USIP_BENCH_OUT", str(tmp_path))
    rec = _canned()
    bench.emit(rec, 1)
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and len(out[0]) <= 8000
    assert json.loads(out[0])["kernels_total"] == len(rec["kernels"])
    full = json.load(open(tmp_path / "bench_full_n1.json"))
    assert len(full["kernels"]) == len(rec["kernels"])
