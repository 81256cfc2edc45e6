#!/usr/bin/env python
"""bench.py -- detector fwd+bwd throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model ball|som] [--pairs 8]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = ModelDetector.optimize (models/keypoint_detector.py:158-207): siamese forward on
B' = 2*pairs clouds, rigid transform, probabilistic chamfer + 2x keypoint-on-pc, backward,
gradient all-reduce (N > 1), Adam update.  Workload = BASELINE.json configs[2], the
configuration the metric is quoted on ("KITTI detector, N=16384, M=512, K=64, batch=8 on 1
MI355X"): 8 pairs -> 16 clouds per GPU, Cs=4, node_knn_k_1=16; the K=64 model is
RPN_Detector_Ball.  Weak scaling: every rank owns 8 pairs.  Synthetic "slab" clouds, seeds
1234+rank; random-init weights (seeded, identical on every rank).

Rank 0 prints ONE JSON line; `roofline` is for the kernel with the largest share of the timed
region (HIP events on the launch stream, usip_amd/prof.py), `kernels` lists the others, and
`cpu_baseline` is the oracle's PyTorch-CPU restatement of the same step timed on this box's
host cores on a bounded sample (1 pair).
"""
import argparse
import json
import os
import sys
import time

# the host driver of this pool only supports dmabuf IPC: without this RCCL's cross-process buffer sharing fails with
# "hipIpcGetMemHandle: invalid argument" (already exported on the GPU boxes; kept here for any other launcher)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch                     # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_F32_TFLOPS = 157.3         # f32-in MFMA / f32 vector peak
PEAK_BF16_TFLOPS = 2500.0       # dense 16-bit MFMA, data-sheet figure: what `roofline.peak` is priced against
# Measured in round 4 (profiles/r04_mfma_sustained_clock.txt): the data-sheet figure is reached with CONSTANT operands at
# 1.18 GHz; random operands hold 0.84 GHz = 1760 TFLOP/s with nothing but MFMAs in the loop.  (A GEMM stage's own LDS
# reads and vector work lower that to 1448 -- that figure contains the kernel's overhead and is NOT used as a roof.)
# Reported next to the contract's fraction, never instead of it.
SUSTAINED_F16_RANDOM_TFLOPS = 1760.0    # MFMA only, random operands (no LDS reads, no VALU work beside them)
# The LAST stdout line is what the driver parses; round 4's 22 KB line was not read.  Budget for that line (bytes):
LINE_BUDGET_N1 = 8000
LINE_BUDGET_MULTI = 12000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # warm-up 10, >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="ball", choices=["ball", "som", "descriptor"],
                    help="ball = RPN_Detector_Ball (the K=64 headline model), som = RPN_Detector, "
                         "descriptor = DescriptorLiteOld step (BASELINE configs[4], SURVEY 8 f-1)")
    ap.add_argument("--pairs", type=int, default=8, help="pairs per GPU (B); the detector sees 2B clouds")
    # --points / --nodes: the same options under names torch.distributed.run cannot mistake for abbreviations of its
    # own (--n.. / --m..) when it scans the command line
    ap.add_argument("--n", "--points", dest="n", type=int, default=16384)
    ap.add_argument("--m", "--nodes", dest="m", type=int, default=512)
    ap.add_argument("--cloud", default="slab")
    ap.add_argument("--precision", default="f32x2", choices=["f32", "f32x3", "f32x2", "bf16"],
                    help="f32x2 (default since round 3) = fp32-ACCURATE products on the 16-bit matrix cores for the "
                         "matrix-bound layers: two fp16 planes per operand, three plane products, operands scaled by "
                         "exact powers of two from rigorous bounds (tests/test_f32x2_mode_gpu.py; the whole-step parity "
                         "tests of tests/test_modules_gpu.py run in this mode), launches without an operand bound as "
                         "f32x3, fp32 MFMA for the HBM-bound layers; "
                         "f32x3 = three bf16 planes per operand, six plane products (round 2's default; "
                         "tests/test_f32x3_mode_gpu.py); "
                         "f32 = fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere; "
                         "bf16 = bf16 multiply / fp32 accumulate, tensors stay fp32 (perf mode of BASELINE configs[1]; "
                         "NOT a parity mode)")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel from the host instead of replaying the step from HIP graphs")
    ap.add_argument("--graph", action="store_true",
                    help="(default since round 2, kept for old command lines) replay from HIP graphs at every N: graph A "
                         "(forward+backward) / eager RCCL all-reduce / graph B (Adam); a refused capture falls back "
                         "to eager launches with a warning")
    ap.add_argument("--tune", action="append", default=[], metavar="KNOB=VALUE",
                    help="measurement aid: set a launch-geometry knob of the library (usip_set_tuning; speed only, never "
                         "results), e.g. --tune x3_gemm_tile=5; recorded in config.tuning")
    ap.add_argument("--no-optimizer", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-n1-probe", action="store_true",
                    help="(N > 1) skip rank 0's 5-step single-GPU probe of the same workload (`distributed.n1_probe`)")
    ap.add_argument("--no-fp32-leg", action="store_true",
                    help="skip the 10-step run of the same step with fp32 MFMA everywhere (`fp32_mfma_only` in the line)")
    ap.add_argument("--no-kernel-leg", action="store_true",
                    help="skip the stand-alone roofline leg of the two HBM-bound integer kernels the north star names "
                         "(dist-in ball_query, index_max), which runs after the timed region at N=1")
    ap.add_argument("--only-kernels", action="store_true",
                    help="run ONLY that leg and print its rows (what tools/profile_roofline.sh puts under rocprofv3)")
    return ap.parse_args()


def load_traffic_db(precision):
    """HBM traffic per launch from the committed rocprofv3 PMC passes (tools/profile_roofline.sh: separate
    --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command; FETCH_SIZE doubled per MI355X_MICROARCH.md).
    Keyed by kernel template + workgroup count; a key shared by several layer shapes carries their average."""
    import glob
    import re
    # one file per (round tag, mode): rNNx_traffic_<mode>.json (older rounds: _traffic.json = f32x3 default run,
    # _traffic_f32mode.json, _traffic_bf16.json).  Only files of THIS mode are read, and only the newest round tag
    # that has one -- rows never mix modes or rounds.
    def mode_of(name):
        if "bf16" in name:
            return "bf16"
        if "f32mode" in name or "_f32." in name:
            return "f32"
        if "som" in name or "desc" in name:
            return None
        if "x2" in name:
            return "f32x2"
        return "f32x3"
    cands = []
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*traffic*.json")):
        name = os.path.basename(f)
        m = re.match(r"(r\d+[a-z]*)_", name)
        if m and mode_of(name) == precision:
            cands.append((m.group(1), f))
    if not cands:
        return {}, None
    def order(t):                                    # r06ac is newer than r06m: session tags run a..z, then aa..az
        m = re.match(r"r(\d+)([a-z]*)", t)
        return (int(m.group(1)), len(m.group(2)), m.group(2))
    tag = max((t for t, _ in cands), key=order)
    db, src = {}, []
    for t, f in sorted(cands):
        if t != tag:
            continue
        try:
            db.update(json.load(open(f)))
            src.append(os.path.basename(f))
        except (OSError, ValueError):
            pass
    return db, (", ".join(src) if src else None)


def kernel_leg(dev, traffic_db, iters=12):
    """The two HBM-bound integer kernels of the path on their own, at BASELINE configs[2] sizes (B'=16 clouds,
    N=16384, M=512): dist-in ball_query on "cube" clouds (every row is a full scan: the defining case of SURVEY 8d)
    and index_max at C=64 / C=128.  Every launch reads a DIFFERENT buffer of a ring whose total size exceeds the
    256 MB Infinity Cache, so the bytes come from HBM, not from the cache the previous launch (or the producer)
    left them in.  HIP events on the launch stream around every launch; median."""
    import numpy as np
    from usip_amd import ops, synth
    B, N, M, K = 16, 16384, 512, 64
    rows = []

    def timed(fn_of_i, n_ring):
        for i in range(3):
            fn_of_i(i % n_ring)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for i, (s, e) in enumerate(evs):
            s.record()
            fn_of_i(i % n_ring)
            e.record()
        torch.cuda.synchronize()
        t = sorted(s.elapsed_time(e) for s, e in evs)
        return t[len(t) // 2] * 1e-3, t[0] * 1e-3, t[-1] * 1e-3

    def row(name, alg, t, key, note):
        med, lo, hi = t
        tr = traffic_db.get(key, {})
        rows.append({"kernel": name, "calls_per_step": 0, "avg_us": round(med * 1e6, 2), "min_us": round(lo * 1e6, 2),
                     "max_us": round(hi * 1e6, 2), "share_of_step": 0.0, "bound": "hbm",
                     "achieved": round(alg / med / 1e9, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                     "frac": round(alg / med / 1e9 / PEAK_HBM_GBPS, 4), "algorithmic_bytes": alg,
                     "traffic": tr.get("hbm_bytes_per_launch"), "rocprof_key": key, "rocprof_avg_us": tr.get("avg_us"),
                     "note": note})

    rng = np.random.default_rng(0)
    ring = []
    for _ in range(3):                                         # 3 x 537 MB distance matrices
        x = torch.from_numpy(np.stack([synth.make_cloud(rng, N, "cube") for _ in range(B)])).to(dev)
        ring.append(ops.pairwise_dist(x[:, :, :M].contiguous(), x))
    inside = ring[0] <= 2.0                                    # algorithmic bytes: scanned prefix of every row
    cs = torch.cumsum(inside.int(), -1)
    kth = (cs >= K).int().argmax(-1)
    prefix = int(torch.where(cs[..., -1] >= K, kth + 1, torch.full_like(kth, N)).sum().item())
    del inside, cs, kth
    alg = 4.0 * prefix + 4.0 * B * M * K
    t = timed(lambda i: ops.ball_query(ring[i], 2.0, K), len(ring))
    row("ball_query (dist-in, cube, B'=16)", alg, t, "ball_query_kernel<1, true> |wg=%d" % (B * M),
        "stand-alone leg, ring of 3 distinct 537 MB inputs (defeats the 256 MB Infinity Cache)")
    del ring
    for C in (64, 128):
        n_ring = 8 if C == 64 else 4                           # 8 x 67 MB / 4 x 134 MB of values
        data = [torch.randn(B, C, N, device=dev) for _ in range(n_ring)]
        idx = [torch.randint(0, M, (B, N), device=dev, dtype=torch.int32) for _ in range(n_ring)]
        alg = 4.0 * (B * C * N + B * N + B * C * M)
        t = timed(lambda i: ops.index_max(data[i], idx[i], M), n_ring)
        ch, u, th = ops.index_max_geometry(B, C, N, M)
        row("index_max (C=%d, B'=16)" % C, alg, t, "index_max_kernel<%d, %d, true, %d> |wg=%d" % (ch, u, th, B * C // ch),
            "stand-alone leg, ring of %d distinct inputs (%d MB)" % (n_ring, n_ring * B * C * N * 4 // 1000000))
        del data, idx
    return rows


def rel_by_channel(actual, expected, axis):
    """The loosest-to-tightest reading of "1e-5 relative" between the tensor-scale bar (tests/conftest.py::assert_close) and a
    per-element one: the error of every slice along `axis` against THAT slice's own maximum -- for
    keypoints / node [B,3,M] axis 1 = the coordinate (x, z ~ +-50 but y ~ N(0,1) for the slab clouds), for sigmas [B,M] axis 0 =
    the cloud.  -> (worst figure, list per slice)."""
    import numpy as np
    a = np.moveaxis(np.asarray(actual, dtype=np.float64), axis, 0)
    e = np.moveaxis(np.asarray(expected, dtype=np.float64), axis, 0)
    a, e = a.reshape(a.shape[0], -1), e.reshape(e.shape[0], -1)
    per = np.abs(a - e).max(axis=1) / np.maximum(np.abs(e).max(axis=1), 1e-30)
    return float(per.max()), [float("%.3e" % v) for v in per]


def cpu_baseline(args, model, dev=None):
    """the oracle (PyTorch-CPU restatement of the same step, proven equal to the reference by the golden
    fixtures) on a bounded sample -- 1 pair = 2 clouds of the same workload -- for BOTH detectors: (A)
    RPN_Detector_Ball, the K=64 headline model, and (B) RPN_Detector, the reference's default.  1 warm-up + 3 timed
    steps each, median.  `value` is the model this run benchmarks; the other is under `models`.

    The oracle's result for the benchmarked model is not thrown away: the HIP step (eager, the arithmetic mode this run
    times) runs on the SAME pair with the SAME parameters and `parity_check` reports what the parity tests assert --
    every index tensor equal, floats relative to the tensor's scale -- at the size the bench times (N=16384, M=512)."""
    import numpy as np
    from oracle import detector as od
    from usip_amd import synth
    from usip_amd.networks import DetectorOptions, detector_param_shapes
    # ATen's strided reductions oversubscribe badly on a many-core host (62 s/step with 256
    # threads vs ~5 s with 8-16): use at most 16 threads and report that count next to the host's.
    host_cores = os.cpu_count() or 1
    cores = min(host_cores, 16)
    torch.set_num_threads(cores)
    opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=16)
    batch_np = synth.make_pair_batch(99, 1, args.n, args.m, 4, args.cloud)
    batch = {k: torch.from_numpy(v) for k, v in batch_np.items()}
    me = model if model in ("ball", "som") else "ball"
    res, parity = {}, None
    WARM, TIMED = 1, 3           # ~6 s (Ball) / ~3.4 s (SOM) per step: ~38 s of CPU work for both models (contract: 10-30 s each)
    for mdl in ("ball", "som"):
        filled = synth.fill_parameters(detector_param_shapes(mdl, 4))
        P = {k: torch.from_numpy(v).requires_grad_(True) for k, v in filled.items()
             if not ("running_" in k or "num_batches" in k)}
        bufs = {k: torch.from_numpy(v.copy()) for k, v in filled.items() if "running_" in k}
        times, last = [], None
        for i in range(WARM + TIMED):
            for p in P.values():
                p.grad = None
            t0 = time.perf_counter()
            last = od.detector_step(P, bufs, batch, mdl, opt.node_knn_k_1, opt.loss_sigma_lower_bound,
                                    opt.keypoint_on_pc_alpha)
            times.append(time.perf_counter() - t0)
        t = sorted(times[WARM:])
        res[mdl] = dict(value=2.0 / t[len(t) // 2], s_per_step=round(t[len(t) // 2], 3), p10=round(t[0], 3),
                        p90=round(t[-1], 3))
        if mdl == me and dev is not None:
            from usip_amd.step import DetectorStep, batch_to_device
            st = DetectorStep(mdl, opt, dev)
            st.load_numpy_state(filled)
            st.step(batch_to_device(batch_np, dev))
            torch.cuda.synchronize()
            idx_equal, worst, names = {}, {}, ("node", "keypoints", "sigmas", "loss", "loss_chamfer", "chamfer_pure",
                                                 "chamfer_weighted")
            for k, v in st.detector.last_indices.items():
                idx_equal[k] = bool(np.array_equal(v.cpu().numpy(), last[k].numpy()))
            for k in names:
                a_, b_ = st.last[k].detach().double().cpu().numpy(), last[k].detach().double().numpy()
                worst[k] = float(np.abs(a_ - b_).max() / max(float(np.abs(b_).max()), 1e-30))
            # per coordinate (node, keypoints: axis 1) / per cloud (sigmas: axis 0), each slice against its own scale
            by_ch = {}
            for k, ax in (("node", 1), ("keypoints", 1), ("sigmas", 0)):
                by_ch[k] = rel_by_channel(st.last[k].detach().cpu().numpy(), last[k].detach().numpy(), ax)
            parity = dict(model={"ball": "RPN_Detector_Ball", "som": "RPN_Detector"}[mdl], pairs=1, n=args.n, m=args.m,
                          indices_equal=all(idx_equal.values()), index_tensors=idx_equal,
                          max_rel=max(worst.values()), rel_by_tensor={k: float("%.3e" % v) for k, v in worst.items()},
                          rel_by_channel={k: float("%.3e" % v[0]) for k, v in by_ch.items()},
                          rel_by_channel_slices={k: v[1] for k, v in by_ch.items()},
                          bar="indices bit-exact, floats <= 1e-5 of the tensor's scale (tests/conftest.py::assert_close)",
                          ok=bool(all(idx_equal.values()) and max(worst.values()) <= 1e-5))
            del st
    return dict(value=res[me]["value"], unit="point-clouds/s", cores=cores, host_cores=host_cores, kind="port",
                thread_cap="16 (ATen's strided reductions slow down beyond that: 62 s/step at 256 threads vs ~6 s)",
                models={"RPN_Detector_Ball": res["ball"], "RPN_Detector": res["som"]},
                sample="1 pair (2 clouds) N=%d M=%d, oracle/detector.py fwd+losses+bwd, median of 3 after 1 warm-up; "
                       "value = %s (%.2f s/step)" % (args.n, args.m, {"ball": "RPN_Detector_Ball", "som": "RPN_Detector"}[me],
                                                     res[me]["s_per_step"])), parity


def _short(v, n=160):
    """Strings of the compact line stay under the driver's per-string window."""
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 3] + "..."


def compact_line(out, budget=None):
    """The ONE line the driver parses (round 4's 22 KB line came back `parsed: null`).  Headline keys,
    `config`, `roofline`, `cpu_baseline`, `parity_check`, `fp32_mfma_only`, `step_ms_rank0`, the top-8 kernel rows of the
    step + the stand-alone roofline legs, and for N > 1 a census trimmed to rank / device / step time / checksum.  The
    full table goes to a file (`emit`), never into this line.  Rows are dropped from the bottom of the kernel table
    until the line fits `budget` bytes."""
    world = int(out.get("n_gpus", 1))
    budget = budget or (LINE_BUDGET_N1 if world == 1 else LINE_BUDGET_MULTI)
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data") if k in out}
    line["dtype"] = _short(line.get("dtype"), 120)
    cfg = dict(out.get("config", {}))
    cfg.pop("step_ms_rank0", None)
    line["config"] = {k: _short(v) for k, v in cfg.items()}
    for k in ("pairs_per_s", "loss", "step_ms_rank0"):
        if k in out:
            line[k] = out[k]
    if "roofline" in out:
        r = out["roofline"]
        line["roofline"] = {k: _short(r[k]) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel",
                                                       "launches_per_step", "avg_us", "share_of_step",
                                                       "algorithmic_per_launch", "timing", "traffic_source",
                                                       "frac_of_sustained") if r.get(k) is not None or k == "traffic"}
        if r.get("frac_of_sustained") is not None:
            line["roofline"]["sustained_roof"] = "MFMA-only loop on random operands: 1760 TFLOP/s (16-bit)"
    if "cpu_baseline" in out:
        c = out["cpu_baseline"]
        line["cpu_baseline"] = {k: _short(c[k], 200) for k in ("value", "unit", "cores", "host_cores", "kind", "sample",
                                                              "models") if k in c}
    if out.get("parity_check"):
        p = out["parity_check"]
        line["parity_check"] = {k: p[k] for k in ("ok", "indices_equal", "max_rel", "rel_by_channel", "model", "pairs", "n", "m")
                                if k in p}
    if "fp32_mfma_only" in out:
        f = out["fp32_mfma_only"]
        line["fp32_mfma_only"] = {k: f[k] for k in ("ms_per_step", "value", "unit", "steps") if k in f}
    if "distributed" in out:
        d = out["distributed"]
        line["distributed"] = {k: d[k] for k in ("backend", "rccl_version", "world_size", "distinct_devices",
                                                 "bucket_bytes", "allreduce_us", "replicas_identical",
                                                 "allreduce_in_graph", "allreduce_form", "launcher") if k in d}
        if isinstance(line["distributed"].get("allreduce_us"), dict):
            line["distributed"]["allreduce_us"] = {k: v for k, v in d["allreduce_us"].items() if k != "how"}
        np_ = d.get("n1_probe")
        if np_:
            line["distributed"]["n1_probe"] = {"n1_reference_ms": round(np_["n1_reference_ms"], 4),
                                               "ratio": round(out["ms_per_step"] / np_["n1_reference_ms"], 4)}
        steps_pr, sums_pr = d.get("step_ms_per_rank", []), d.get("param_checksum_per_rank", [])
        line["ranks"] = [{"rank": c["rank"], "device_uuid": c.get("device_uuid"), "device_index": c.get("device_index"),
                          "step_ms": steps_pr[i] if i < len(steps_pr) else None,
                          "checksum": (sums_pr[i] or [None])[0] if i < len(sums_pr) else None}
                         for i, c in enumerate(out.get("ranks_seen", []))]
    rows = out.get("kernels", [])
    step_rows = [r for r in rows if r.get("calls_per_step")][:8]
    legs = [r for r in rows if not r.get("calls_per_step")]
    keep = ("kernel", "calls_per_step", "avg_us", "share_of_step", "bound", "achieved", "unit", "frac", "traffic")

    def slim(r):
        return {k: (_short(r[k], 60) if k == "kernel" else r[k]) for k in keep if r.get(k) is not None}
    line["kernels"] = [slim(r) for r in legs] + [slim(r) for r in step_rows]
    if out.get("full_table"):
        line["full_table"] = out["full_table"]
    line["kernels_total"] = len(rows)
    text = json.dumps(line, separators=(",", ":"))
    while len(text) > budget and len(line["kernels"]) > len(legs):
        line["kernels"].pop()
        text = json.dumps(line, separators=(",", ":"))
    if len(text) > budget:                                  # cannot happen with the fields above; never print a long line
        for k in ("ranks", "kernels"):
            line.pop(k, None)
        text = json.dumps(line, separators=(",", ":"))
    return text


def emit(out, world):
    """Full record (every kernel row, every note) -> gpurun_out/bench_full_n<N>.json; compact line -> stdout, LAST."""
    path = os.path.join(os.environ.get("USIP_BENCH_OUT", os.path.join(ROOT, "gpurun_out")), "bench_full_n%d.json" % world)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f)
        out["full_table"] = os.path.relpath(path, ROOT)
    except OSError:
        out["full_table"] = None
    sys.stdout.flush()
    print(compact_line(out), flush=True)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the same
    command line under torch.distributed.run on 127.0.0.1 with a free port) and pass rank 0's JSON line through.
    Replaces nn.DataParallel's in-process replication (models/keypoint_detector.py:35-37)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, USIP_BENCH_SPAWNED="1")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # a launcher started another number of ranks than --gpus says: the line must not claim N it did not run
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    # One process per GPU over RCCL ("nccl" IS RCCL on ROCm).  USIP_DIST_BACKEND=gloo + USIP_SHARE_DEVICE=1 is a
    # debugging aid only: it lets the N>1 code path run with several ranks on ONE GPU (RCCL refuses that).
    backend = os.environ.get("USIP_DIST_BACKEND", "nccl")
    if os.environ.get("USIP_SHARE_DEVICE") == "1":
        local_rank = 0
    if world > 1 and hasattr(os, "sched_setaffinity"):
        # one disjoint core set per rank: eight launch threads (+ RCCL's proxy threads) on one host otherwise
        # migrate across each other's caches; harmless if the container restricts the set already
        try:
            cores = sorted(os.sched_getaffinity(0))
            per = max(1, len(cores) // world)
            lr = int(os.environ.get("LOCAL_RANK", "0"))
            mine = cores[lr * per:(lr + 1) * per]
            if len(mine) >= 2:
                os.sched_setaffinity(0, mine)
        except OSError:
            pass
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from usip_amd import functional as Fh
    from usip_amd import ops, prof, synth
    from usip_amd.networks import DetectorOptions
    assert not Fh.pins_active(), "bench.py never runs with the decision-pinning test hooks"
    from usip_amd import _lib
    for kv in args.tune:
        name, val = kv.split("=")
        _lib.check(_lib.lib().usip_set_tuning(name.encode(), int(val)), "usip_set_tuning(%s)" % kv)
    if args.only_kernels:
        print(json.dumps({"kernels": kernel_leg(dev, load_traffic_db(args.precision)[0])}), flush=True)
        return
    ops.set_matmul_mode(args.precision)
    mfma_peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
    # f32x3: rows of the split kernels are priced against the bf16 matrix peak / 6 (six plane products per fp32
    # product), rows that stayed on the fp32 kernel against the fp32 peak -- see the per-kernel `peak` fields
    from usip_amd.step import DetectorStep, batch_to_device

    opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=16)
    torch.manual_seed(0)                                   # identical replicas, no broadcast needed
    if args.model == "descriptor":
        # BASELINE configs[4]: N=16384, 256 keypoints, 4 (anchor, positive) pairs per GPU; keypoints are cloud
        # points, sigmas U(0.1, 3), in-batch negatives = the next pair's anchor
        import numpy as np
        from usip_amd.step import DescriptorStep
        args.no_cpu_baseline = True
        kp = 256 if args.m == 512 else args.m
        pairs = 4 if args.pairs == 8 else args.pairs
        st = DescriptorStep(opt, dev, with_optimizer=not args.no_optimizer, graph=not args.no_graph)
        b0 = synth.make_pair_batch(1234 + rank, pairs, args.n, kp, 4, args.cloud)
        rng = np.random.default_rng(99 + rank)
        batch = batch_to_device(dict(anc_pc=b0["src_pc"], pos_pc=b0["dst_pc"], anc_sn=b0["src_sn"], pos_sn=b0["dst_sn"],
                                     anc_kp=b0["src_node"], pos_kp=b0["dst_node"],
                                     anc_sigmas=rng.uniform(0.1, 3.0, (pairs, kp)).astype(np.float32),
                                     neg_idx=np.roll(np.arange(pairs), 1).astype(np.int64)), dev)
        args.pairs = pairs
    else:
        st = DetectorStep(args.model, opt, dev, with_optimizer=not args.no_optimizer, graph=not args.no_graph)
        batch = batch_to_device(synth.make_pair_batch(1234 + rank, args.pairs, args.n, args.m, 4, args.cloud), dev)
    if not args.no_graph:
        # set-up, not warm-up: two eager steps (allocator, rocBLAS handles) and the graph capture happen here,
        # so that the W warm-up steps and the K timed steps below are all steady-state steps
        for _ in range(3):
            st.step(batch)
        batch = st.static_batch(batch) or batch              # feed the captured input buffers directly

    graphed = (not args.no_graph) and st.use_graph          # False when the capture was refused (eager fallback)

    def barrier():
        if world > 1:
            dist.barrier()

    n1_probe = None
    if world > 1 and args.model != "descriptor" and not args.no_n1_probe:
        # Self-check of the scaling line: rank 0 alone times 5 steps of the SAME per-GPU workload without any exchange
        # (its own step object, so the replicas stay identical) while the other ranks wait; the N-GPU step should cost
        # that plus the all-reduce.  The driver computes scaling efficiency from separate runs; this is one run's view.
        if rank == 0:
            torch.manual_seed(0)
            st1 = DetectorStep(args.model, opt, dev, with_optimizer=not args.no_optimizer, graph=not args.no_graph)
            st1.solo = True                                # no all-reduce, no capture of one
            b1 = batch_to_device(synth.make_pair_batch(1234, args.pairs, args.n, args.m, 4, args.cloud), dev)
            for _ in range(6):
                st1.step(b1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                st1.step(b1)
            torch.cuda.synchronize()
            n1_probe = {"n1_reference_ms": (time.perf_counter() - t1) / 5 * 1e3, "steps": 5,
                        "how": "rank 0 alone, same per-GPU workload, no gradient exchange, before the timed region"}
            del st1, b1
            torch.cuda.empty_cache()
        barrier()
    for _ in range(args.warmup):
        st.step(batch)
    torch.cuda.synchronize()
    barrier()
    if world > 1 and not getattr(st, "allreduce_in_graph", False):
        st.allreduce_events = []                           # two HIP events around every gradient all-reduce
    # Per-kernel HIP events cost ~3 us of stream bubble each (~0.7 ms per step for ~220 of them): they are
    # recorded on every 4th step (1-GPU eager runs) or on one middle step only (graph replay, multi-GPU), which keeps the headline
    # number within ~1.5 % of an uninstrumented run while the kernel durations still come from inside the
    # timed region.
    # Sampled steps are launched eagerly (a graph replay cannot carry the events); the others replay the graphs.
    # light instrumentation (graph replay, and every multi-GPU run so that all N are measured alike): ONE step of
    # the timed region carries events, and only around the shared-MLP launches
    light = graphed or world > 1
    sample_every = 4
    timed_steps_sampled = 0
    if not args.no_kernel_timing:
        prof.reset()
    # one HIP event per step boundary (median and p10/p90 of the step time); 3 us each
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        # graph replay: instrument the MIDDLE step -- the host is then a few replays ahead of the GPU, so the ~8 ms it
        # needs to launch an eager step never leave the GPU waiting (at step 0 they do: the queue starts empty)
        sampled = (not args.no_kernel_timing) and ((i == args.steps // 2) if light else (i % sample_every == 0))
        # graph replay: the one instrumented step brackets only the shared-MLP launches (the roofline kernel's
        # family); the other operators are timed in an extra step after the timed region (see below)
        prof.enable(sampled, only="shared_mlp" if light else None)
        timed_steps_sampled += int(sampled)
        if graphed:
            st.step(batch, eager=sampled)
        else:
            st.step(batch)
        marks[i + 1].record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    prof.enable(False)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(st.last["loss"].item())
    census = None
    if world > 1:
        # who took part: every rank reports its device; all-reduce durations (max over ranks per percentile) and the
        # per-rank step time, so that the line shows that RCCL saw N ranks on N devices and what the exchange cost
        ar_events, st.allreduce_events = (st.allreduce_events or [])[:args.steps], None
        ar = sorted(s.elapsed_time(e) * 1e3 for s, e in ar_events)
        props = torch.cuda.get_device_properties(dev)
        mine = dict(rank=rank, local_rank=local_rank, device_index=dev.index, pid=os.getpid(),
                    cores=sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
                    device_name=props.name, device_uuid=str(getattr(props, "uuid", "")),
                    pci_bus_id=int(getattr(props, "pci_bus_id", -1)), step_ms_wall=elapsed / args.steps * 1e3,
                    allreduce_us_p50=ar[len(ar) // 2] if ar else None,
                    allreduce_us_p90=ar[min(len(ar) - 1, int(0.9 * len(ar)))] if ar else None,
                    allreduce_calls=len(ar), loss=loss_val,
                    # replicas start identical and receive identical reduced gradients: the parameters must agree bit
                    # for bit on every rank after the run (fp64 sum of the flat parameter buffer, and its first words)
                    param_checksum=[float(st.bucket.flat_param.detach().double().sum().item()),
                                    float(st.bucket.flat_param.detach()[:8].double().abs().sum().item())]
                    if st.bucket.flat_param is not None else None)
        census = [None] * world
        dist.all_gather_object(census, mine)
    summ = None
    if not args.no_kernel_timing:
        summ = prof.summary()
        if light:
            # the remaining operators: one fully instrumented eager step AFTER the timed region (not part of
            # `value`; on every rank, it contains the gradient all-reduce); shared-MLP entries keep their in-region
            # timings
            prof.reset()
            prof.enable(True)
            if graphed:
                st.step(batch, eager=True)
            else:
                st.step(batch)
            prof.enable(False)
            for name, r in prof.summary().items():
                if name not in summ:
                    r = dict(r)
                    r["calls"] = r["calls"] * timed_steps_sampled         # normalised per sampled step below
                    r["total_ms"] = r["total_ms"] * timed_steps_sampled
                    summ[name] = r

    if rank == 0:
        clouds = world * 2 * args.pairs * args.steps
        out = {
            "metric": "point-clouds/sec %s fwd+bwd" % ("descriptor" if args.model == "descriptor" else "detector"),
            "value": clouds / elapsed, "unit": "point-clouds/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "f32x3": "f32 (matrix-bound products as six bf16-plane MFMAs of an exact three-way "
                                             "split, fp32 accumulate: fp32-accurate)",
                      "f32x2": "f32 results; matrix-bound products emulated from 2 fp16 planes per operand (22-bit "
                               "operands, 3 plane products, exact power-of-two operand scaling, fp32 accumulate; launches "
                               "without an operand bound: 3 bf16 planes / 6 products); fp32_mfma_only = the same step with "
                               "fp32 MFMA everywhere",
                      "bf16": "bf16 multiply, f32 accumulate and storage (perf mode)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": ("KITTI descriptor head N=%d, 256 keypoints, K=64, batch=%d pairs/GPU (BASELINE "
                                    "configs[4])" % (args.n, args.pairs)) if args.model == "descriptor" else
                                   ("ModelNet40-shaped detector N=%d M=%d batch=%d pairs/GPU (BASELINE configs[1])"
                                    % (args.n, args.m, args.pairs)) if (args.n, args.m) == (5000, 64) else
                                   ("KITTI detector N=%d M=%d K=64 batch=%d pairs/GPU (BASELINE configs[2])"
                                    % (args.n, args.m, args.pairs)),
                       "detector": {"ball": "RPN_Detector_Ball", "som": "RPN_Detector",
                                    "descriptor": "DescriptorLiteOld (descriptor head, 256 keypoints)"}[args.model],
                       "clouds_per_gpu": 2 * args.pairs, "surface_normal_len": 4, "node_knn_k_1": 16,
                       "ball_radius": 2, "ball_k": 64, "cloud": args.cloud, "matmul": args.precision,
                       "step": "fwd+losses+bwd" + ("+allreduce" if world > 1 else "") +
                               ("" if args.no_optimizer else "+adam"),
                       "launch": ("HIP graph replay (ONE graph per step: forward, backward, RCCL all-reduce, Adam)"
                                  if getattr(st, "allreduce_in_graph", False) else
                                  "HIP graph replay (ONE graph per step: forward, backward, Adam)"
                                  if any(e["fused"] for e in getattr(st, "_graphs", {}).values()) else
                                  "HIP graph replay (2 graphs per step, all-reduce between them)") if graphed else "eager",
                       "parallelism": "dp%d" % world, "tuning": args.tune or None},
            "pairs_per_s": clouds / elapsed / 2, "loss": loss_val,
        }
        if census is not None:
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:                              # noqa: BLE001  (gloo debugging runs have no RCCL)
                rccl = None
            out["ranks_seen"] = [{k: c[k] for k in ("rank", "local_rank", "device_index", "device_uuid", "pci_bus_id",
                                                    "device_name", "pid", "cores")} for c in census]
            out["distributed"] = {
                "backend": backend, "rccl_version": rccl if backend == "nccl" else None, "world_size": world,
                "distinct_devices": len({(c["device_index"], c["device_uuid"], c["pci_bus_id"]) for c in census}),
                "bucket_bytes": int(st.bucket.flat.numel() * 4), "allreduce_per_step": 1,
                "allreduce_us": {"p50": max(c["allreduce_us_p50"] or 0.0 for c in census),
                                 "p90": max(c["allreduce_us_p90"] or 0.0 for c in census),
                                 "calls_timed_per_rank": min(c["allreduce_calls"] for c in census),
                                 "how": "HIP events on the launch stream around all_reduce(SUM) + 1/world scale, every "
                                        "step of the timed region, max over ranks of each rank's percentile"},
                "step_ms_per_rank": [round(c["step_ms_wall"], 4) for c in census],
                "loss_per_rank": [c["loss"] for c in census],
                "param_checksum_per_rank": [c["param_checksum"] for c in census],
                "replicas_identical": len({str(c["param_checksum"]) for c in census}) == 1,
                "allreduce_in_graph": bool(getattr(st, "allreduce_in_graph", False)),
                "allreduce_form": st.allreduce_form(), "fused_fallbacks": getattr(st, "fused_fallbacks", 0),
                "n1_probe": n1_probe,
                "launcher": "self-spawned torch.distributed.run" if os.environ.get("USIP_BENCH_SPAWNED") else
                            "external launcher"}
        if census is not None and backend == "nccl":
            # a line that claims N GPUs must have run on N GPUs with identical replicas: anything else is an error, not
            # a number (USIP_DIST_BACKEND=gloo + USIP_SHARE_DEVICE=1, the one-GPU debugging mode, is exempt by design)
            d = out["distributed"]
            if d["distinct_devices"] != world or not d["replicas_identical"]:
                print(json.dumps({"error": "bench.py --gpus %d: %d distinct devices, replicas_identical=%s"
                                           % (world, d["distinct_devices"], d["replicas_identical"]),
                                  "distributed": d, "ranks_seen": out["ranks_seen"]}), flush=True)
                sys.exit(3)
        raw_steps = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
        per_step = sorted(raw_steps)
        if per_step:
            pick = lambda q: per_step[min(len(per_step) - 1, int(q * len(per_step)))]
            out["step_ms_rank0"] = {"p10": round(pick(0.1), 4), "median": round(pick(0.5), 4), "p90": round(pick(0.9), 4),
                                    "max": round(per_step[-1], 4), "first": round(raw_steps[0], 4)}
            # the same inside `config` (the contract's own keys): with the driver's --steps 20 the timed region is 0.1 s
            out["config"]["step_ms_rank0"] = out["step_ms_rank0"]
        if not args.no_kernel_timing:
            traffic_db, traffic_src = load_traffic_db(args.precision)
            def products(r):
                """Matrix products per fp32 product of a split-product launch: 6 (three bf16 planes), 3 (two fp16
                planes: template argument NPL = 2, kernel names x2h / x2d / x2r / layer_bwd_x2), 0 for every other kernel."""
                key = (r.get("rocprof_key") or "").split(" |wg=")[0]
                if "x2h" in key or "x2d_kernel" in key or "x2f_kernel" in key or "x2r_kernel" in key or "x2l_kernel" in key or \
                        "layer_bwd_x2_kernel" in key or \
                        (("x3p_kernel" in key or "wgrad_x3_kernel" in key) and key.rstrip(">").endswith(", 2")):
                    return 3
                if "x3" in key or ("bf16_kernel" in key and key.rstrip(">").endswith(", 3")):
                    return 6
                return 0

            def is_x3(r):
                return products(r) > 0

            kernels = []
            for name, r in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"]):
                gemm = r["flops_per_call"] > 0 and name.startswith("shared_mlp")
                # a split-product launch issues `products` 16-bit matrix products per fp32 product: its ceiling in
                # fp32-equivalent flops is the dense 16-bit peak / products
                mm_peak = PEAK_BF16_TFLOPS / products(r) if is_x3(r) else mfma_peak
                row = {"kernel": name, "calls_per_step": r["calls"] / timed_steps_sampled, "avg_us": round(r["avg_us"], 2),
                       "share_of_step": round(r["total_ms"] / timed_steps_sampled / (elapsed / args.steps * 1e3), 4)}
                if gemm:
                    # both rooflines of a GEMM launch; `bound` = the one whose floor is higher (the narrow layers, 16
                    # flop/B, are HBM-bound; 512 x 256 sits at the ridge: 50 us of HBM against 41 us of matrix pipe)
                    f_mm, f_hbm = r["TFLOPs"] / mm_peak, r["GBps"] / PEAK_HBM_GBPS
                    mfma = r["bytes_per_call"] / (PEAK_HBM_GBPS * 1e9) <= r["flops_per_call"] / (mm_peak * 1e12)
                    row.update({"bound": "mfma" if mfma else "hbm", "achieved": round(r["TFLOPs"] if mfma else r["GBps"], 3),
                                "peak": mm_peak if mfma else PEAK_HBM_GBPS, "unit": "TFLOP/s" if mfma else "GB/s",
                                "frac": round(f_mm if mfma else f_hbm, 4), "frac_mfma": round(f_mm, 4),
                                "frac_hbm": round(f_hbm, 4)})
                elif r["flops_per_call"] > 0:
                    # brute-force distance kernels (ball_query_coords, nearest, knn, som_assign): SURVEY 8d prices them in
                    # pair evaluations against the fp32 vector peak (8 flop per pair: 3 sub, 3 mul/fma, compare, select);
                    # their HBM traffic is tiny by design
                    pairs = r["flops_per_call"] / 8.0
                    row.update({"bound": "valu", "achieved": round(r["TFLOPs"], 3), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(r["TFLOPs"] / PEAK_F32_TFLOPS, 4),
                                "pair_evals_per_s": round(pairs / (r["avg_us"] * 1e-6), 1) if r["avg_us"] > 0 else None,
                                "hbm_GBps": round(r["GBps"], 1)})
                elif r["bytes_per_call"] < 4.0e6:
                    # a few hundred KB: neither roofline applies, the launch's own latency does (and, for the rows of the
                    # instrumented eager step, whatever the host did between the two event records: `fill_scaled`, the
                    # first launch of the backward pass, reads 5 us in the rocprof trace and 20-60 us here)
                    row.update({"bound": "latency", "achieved": round(r["avg_us"], 2), "peak": None, "unit": "us",
                                "frac": None, "hbm_GBps": round(r["GBps"], 1)})
                else:
                    row.update({"bound": "hbm", "achieved": round(r["GBps"], 3), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                "frac": round(r["GBps"] / PEAK_HBM_GBPS, 4)})
                row.update({"traffic": (traffic_db.get(r.get("rocprof_key") or "", {}).get("hbm_bytes_per_launch")),
                            "rocprof_key": r.get("rocprof_key")})
                kernels.append(row)
            if kernels:
                # The dominant KERNEL (device function = template instantiation, as rocprofv3 lists it), all its
                # launches in the timed region together.
                def family(label, r):
                    # the device function as rocprof names it (template instantiation); launches of one
                    # instantiation at different layer shapes are the same kernel
                    key = r.get("rocprof_key")
                    return key.split(" |wg=")[0] if key else label.split()[0]
                fam = {}
                for name, r in summ.items():
                    f = fam.setdefault(family(name, r), dict(ms=0.0, calls=0, flops=0.0, nbytes=0.0, traffic=0.0,
                                                          traffic_calls=0, mfma=False, moved=0.0))
                    f["moved"] += r.get("moved_per_call", 0.0) * r["calls"]
                    f["ms"] += r["total_ms"]
                    f["calls"] += r["calls"]
                    f["flops"] += r["flops_per_call"] * r["calls"]
                    f["nbytes"] += r["bytes_per_call"] * r["calls"]
                    f["mfma"] = f["mfma"] or name.startswith("shared_mlp")
                    t = traffic_db.get(r.get("rocprof_key") or "", {}).get("hbm_bytes_per_launch")
                    if t is not None:
                        f["traffic"] += t * r["calls"]
                        f["traffic_calls"] += r["calls"]
                top_name, top = max(fam.items(), key=lambda kv: kv[1]["ms"])
                avg_s = top["ms"] * 1e-3 / top["calls"]
                if top["mfma"]:
                    top_peak = (PEAK_BF16_TFLOPS / products({"rocprof_key": top_name})
                                if is_x3({"rocprof_key": top_name}) else mfma_peak)
                    ach, peak, unit, bound = top["flops"] / top["calls"] / avg_s / 1e12, top_peak, "TFLOP/s", "mfma"
                else:
                    ach, peak, unit, bound = top["nbytes"] / top["calls"] / avg_s / 1e9, PEAK_HBM_GBPS, "GB/s", "hbm"
                out["roofline"] = {
                    "bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
                    "traffic": (top["traffic"] / top["traffic_calls"]) if top["traffic_calls"] else None,
                    "kernel": top_name + (" (csrc/gemm_x2f.hip)" if "x2f" in top_name else
                                          " (csrc/gemm_x2d.hip)" if "x2d" in top_name else
                                          " (csrc/shared_mlp_x3.hip)" if "x3" in top_name else
                                          " (csrc/shared_mlp_bf16.hip)" if "bf16" in top_name else " (csrc/shared_mlp.hip)")
                    if top["mfma"] else top_name, "launches_per_step": top["calls"] / timed_steps_sampled,
                    "avg_us": round(avg_s * 1e6, 2),
                    "share_of_step": round(top["ms"] / timed_steps_sampled / (elapsed / args.steps * 1e3), 4),
                    "timing": ("HIP events on the launch stream, one eager step in the middle of the timed region"
                               if light else "HIP events on the launch stream, every %d-th step of the timed region "
                               "(%d steps)" % (sample_every, timed_steps_sampled)),
                    "algorithmic_per_launch": (top["flops"] if top["mfma"] else top["nbytes"]) / top["calls"],
                    "fp32_mfma_peak_ratio": (round(ach / PEAK_F32_TFLOPS, 3) if (top["mfma"] and peak != mfma_peak) else None),
                    "traffic_source": traffic_src,
                    "peak_note": ("fp32-equivalent: a split-product launch issues %d 16-bit matrix products per fp32 "
                                  "product, so its ceiling is the dense bf16/fp16 peak (2500 TFLOP/s) / %d; the same "
                                  "launch priced in 16-bit flops actually issued: %.0f of 2500 TFLOP/s"
                                  % (products({"rocprof_key": top_name}), products({"rocprof_key": top_name}),
                                     products({"rocprof_key": top_name}) * ach))
                    if (top["mfma"] and peak != mfma_peak) else None,
                    "attainable_peak_note": (
                        "a pure fp32-MFMA loop (tools/mfma_peak.hip) sustains 121-141 TFLOP/s with random operands on "
                        "this chip (clock 1.85-2.15 GHz under load), see profiles/r01_mfma_attainable_peak.txt"
                        if (top["mfma"] and args.precision == "f32") else
                        # round 4: what the 16-bit matrix pipe sustains depends on the operand DATA -- measured, not assumed
                        "2500 TFLOP/s needs CONSTANT operands (2467 measured at 1.18 GHz); back-to-back MFMAs on random "
                        "operands hold 0.84 GHz = 1760 TFLOP/s = %.0f fp32-equivalent at %d products "
                        "(profiles/r04_mfma_sustained_clock.txt): frac_of_sustained = achieved / that"
                        % (SUSTAINED_F16_RANDOM_TFLOPS / products({"rocprof_key": top_name}), products({"rocprof_key": top_name}))
                        if (top["mfma"] and peak != mfma_peak) else None),
                    "frac_of_sustained": (round(ach / (SUSTAINED_F16_RANDOM_TFLOPS / products({"rocprof_key": top_name})), 4)
                                          if (top["mfma"] and peak != mfma_peak) else None)}
                out["kernels"] = kernels
                if light:
                    out["kernels_note"] = ("shared_mlp_* rows: HIP events inside the timed region (one eager step); the "
                                           "other rows: one instrumented eager step run after the timed region")
        if world == 1 and args.precision != "f32" and not args.no_fp32_leg and args.model != "descriptor":
            # the same step with plain fp32 MFMA everywhere (v_mfma_f32_32x32x2_f32), timed the same way on a short
            # region: what the headline would be without the 16-bit-plane emulation of the matrix-bound products
            prev_mode = ops.set_matmul_mode("f32")
            torch.manual_seed(0)
            st32 = DetectorStep(args.model, opt, dev, with_optimizer=not args.no_optimizer, graph=not args.no_graph)
            b32 = batch_to_device(synth.make_pair_batch(1234 + rank, args.pairs, args.n, args.m, 4, args.cloud), dev)
            for _ in range(3):
                st32.step(b32)
            if not args.no_graph:
                b32 = st32.static_batch(b32) or b32
            for _ in range(3):
                st32.step(b32)
            torch.cuda.synchronize()
            t32 = time.perf_counter()
            n32 = 10
            for _ in range(n32):
                st32.step(b32)
            torch.cuda.synchronize()
            t32 = (time.perf_counter() - t32) / n32
            out["fp32_mfma_only"] = {"ms_per_step": t32 * 1e3, "value": 2 * args.pairs / t32, "unit": "point-clouds/s",
                                     "steps": n32, "dtype": "f32 (fp32 MFMA for every product)"}
            del st32, b32
            ops.set_matmul_mode(prev_mode)
        if world == 1 and not args.no_kernel_leg and not args.no_kernel_timing and args.model != "descriptor":
            del st, batch
            torch.cuda.empty_cache()
            out.setdefault("kernels", []).extend(kernel_leg(dev, load_traffic_db(args.precision)[0]))
            out["kernels_note"] = (out.get("kernels_note", "") + "; rows with calls_per_step 0: stand-alone roofline leg "
                                   "after the timed region (see their note)").lstrip("; ")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["parity_check"] = cpu_baseline(args, args.model, dev)
        emit(out, world)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
