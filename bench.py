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

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_F32_TFLOPS = 157.3         # f32-in MFMA / f32 vector peak
PEAK_BF16_TFLOPS = 2500.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="ball", choices=["ball", "som", "descriptor"],
                    help="ball = RPN_Detector_Ball (the K=64 headline model), som = RPN_Detector, "
                         "descriptor = DescriptorLiteOld step (BASELINE configs[4], SURVEY 8 f-1)")
    ap.add_argument("--pairs", type=int, default=8, help="pairs per GPU (B); the detector sees 2B clouds")
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--m", type=int, default=512)
    ap.add_argument("--cloud", default="slab")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16"],
                    help="f32 = fp32 MFMA (parity mode, the headline); bf16 = bf16 multiply / fp32 accumulate in the "
                         "shared-MLP kernels, tensors stay fp32 (perf mode of BASELINE configs[1]; NOT the headline)")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel from the host instead of replaying the step from HIP graphs")
    ap.add_argument("--graph", action="store_true",
                    help="replay from HIP graphs at N > 1 as well (default there: eager launches -- capture next to "
                         "an RCCL communicator cannot be exercised on the 1-GPU development boxes)")
    ap.add_argument("--no-optimizer", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    return ap.parse_args()


def cpu_baseline(args, model):
    """The oracle (PyTorch-CPU restatement, proven equal to the reference by the golden fixtures)
    on a bounded sample: 1 pair = 2 clouds of the same workload, 1 warm-up + 2 timed steps."""
    import numpy as np
    from oracle import detector as od
    from usip_amd import synth
    from usip_amd.networks import detector_param_shapes
    # ATen's strided reductions oversubscribe badly on a many-core host (62 s/step with 256
    # threads vs ~5 s with 8-16): use at most 16 threads and report that count.
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    shapes = detector_param_shapes(model, 4)
    filled = synth.fill_parameters(shapes)
    P = {k: torch.from_numpy(v).requires_grad_(True) for k, v in filled.items()
         if not ("running_" in k or "num_batches" in k)}
    bufs = {k: torch.from_numpy(v.copy()) for k, v in filled.items() if "running_" in k}
    batch = {k: torch.from_numpy(v) for k, v in
             synth.make_pair_batch(99, 1, args.n, args.m, 4, args.cloud).items()}
    times = []
    for i in range(3):
        for p in P.values():
            p.grad = None
        t0 = time.perf_counter()
        od.detector_step(P, bufs, batch, model, 16, 1e-3, 0.01)
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[1:]))
    return dict(value=2.0 / t, unit="point-clouds/s", cores=cores, kind="port",
                sample="1 pair (2 clouds) N=%d M=%d model=%s, oracle/detector.py fwd+losses+bwd, "
                       "median of 2 after 1 warm-up (%.2f s/step)" % (args.n, args.m, model, t))


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not args.graph:
        args.no_graph = True
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    # One process per GPU over RCCL ("nccl" IS RCCL on ROCm).  USIP_DIST_BACKEND=gloo + USIP_SHARE_DEVICE=1 is a
    # debugging aid only: it lets the N>1 code path run with several ranks on ONE GPU (RCCL refuses that).
    backend = os.environ.get("USIP_DIST_BACKEND", "nccl")
    if os.environ.get("USIP_SHARE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from usip_amd import ops, prof, synth
    from usip_amd.networks import DetectorOptions
    ops.set_matmul_mode(args.precision)
    mfma_peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
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

    graphed = not args.no_graph

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        st.step(batch)
    torch.cuda.synchronize()
    barrier()
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
            "dtype": "f32" if args.precision == "f32" else "bf16 multiply, f32 accumulate and storage (perf mode)",
            "data": "synthetic",
            "config": {"workload": ("KITTI descriptor head N=%d, 256 keypoints, K=64, batch=%d pairs/GPU (BASELINE "
                                    "configs[4])" % (args.n, args.pairs)) if args.model == "descriptor" else
                                   ("KITTI detector N=%d M=%d K=64 batch=%d pairs/GPU (BASELINE configs[2])"
                                    % (args.n, args.m, args.pairs)),
                       "detector": {"ball": "RPN_Detector_Ball", "som": "RPN_Detector",
                                    "descriptor": "DescriptorLiteOld (descriptor head, 256 keypoints)"}[args.model],
                       "clouds_per_gpu": 2 * args.pairs, "surface_normal_len": 4, "node_knn_k_1": 16,
                       "ball_radius": 2, "ball_k": 64, "cloud": args.cloud,
                       "step": "fwd+losses+bwd" + ("+allreduce" if world > 1 else "") +
                               ("" if args.no_optimizer else "+adam"),
                       "launch": "HIP graph replay (2 graphs per step, all-reduce between them)" if graphed
                                 else "eager",
                       "parallelism": "dp%d" % world},
            "pairs_per_s": clouds / elapsed / 2, "loss": loss_val,
        }
        raw_steps = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
        per_step = sorted(raw_steps)
        if per_step:
            pick = lambda q: per_step[min(len(per_step) - 1, int(q * len(per_step)))]
            out["step_ms_rank0"] = {"p10": round(pick(0.1), 4), "median": round(pick(0.5), 4), "p90": round(pick(0.9), 4),
                                    "max": round(per_step[-1], 4), "first": round(raw_steps[0], 4)}
        if not args.no_kernel_timing:
            # HBM traffic per launch from the committed rocprofv3 PMC passes (tools/profile_roofline.sh:
            # separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command; FETCH_SIZE doubled per
            # MI355X_MICROARCH.md).  Keyed by kernel template + workgroup count; a key shared by several
            # layer shapes carries their average.
            traffic_db, traffic_src = {}, None
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
                if ("bf16" in os.path.basename(f)) != (args.precision == "bf16"):
                    continue                               # each precision mode has its own kernels and PMC passes
                try:
                    traffic_db, traffic_src = json.load(open(f)), os.path.basename(f)
                except (OSError, ValueError):
                    pass
            kernels = []
            for name, r in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"]):
                mfma = r["flops_per_call"] > 0 and name.startswith("shared_mlp")
                ach = r["TFLOPs"] if mfma else r["GBps"]
                peak = mfma_peak if mfma else PEAK_HBM_GBPS
                kernels.append({"kernel": name, "calls_per_step": r["calls"] / timed_steps_sampled,
                                "avg_us": round(r["avg_us"], 2),
                                "share_of_step": round(r["total_ms"] / timed_steps_sampled / (elapsed / args.steps * 1e3), 4),
                                "bound": "mfma" if mfma else "hbm", "achieved": round(ach, 3), "peak": peak,
                                "unit": "TFLOP/s" if mfma else "GB/s", "frac": round(ach / peak, 4),
                                "traffic": (traffic_db.get(r.get("rocprof_key") or "", {}).get("hbm_bytes_per_launch")),
                                "rocprof_key": r.get("rocprof_key")})
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
                                                          traffic_calls=0, mfma=False))
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
                    ach, peak, unit, bound = top["flops"] / top["calls"] / avg_s / 1e12, mfma_peak, "TFLOP/s", "mfma"
                else:
                    ach, peak, unit, bound = top["nbytes"] / top["calls"] / avg_s / 1e9, PEAK_HBM_GBPS, "GB/s", "hbm"
                out["roofline"] = {
                    "bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
                    "traffic": (top["traffic"] / top["traffic_calls"]) if top["traffic_calls"] else None,
                    "kernel": top_name + (" (csrc/shared_mlp_bf16.hip)" if "bf16" in top_name else " (csrc/shared_mlp.hip)")
                    if top["mfma"] else top_name, "launches_per_step": top["calls"] / timed_steps_sampled,
                    "avg_us": round(avg_s * 1e6, 2),
                    "share_of_step": round(top["ms"] / timed_steps_sampled / (elapsed / args.steps * 1e3), 4),
                    "timing": ("HIP events on the launch stream, one eager step in the middle of the timed region"
                               if light else "HIP events on the launch stream, every %d-th step of the timed region "
                               "(%d steps)" % (sample_every, timed_steps_sampled)),
                    "algorithmic_per_launch": (top["flops"] if top["mfma"] else top["nbytes"]) / top["calls"],
                    "traffic_source": traffic_src,
                    "attainable_peak_note": "a pure fp32-MFMA loop (tools/mfma_peak.hip) sustains 121-141 TFLOP/s with "
                                            "random operands on this chip (clock 1.85-2.15 GHz under load), see "
                                            "profiles/r01_mfma_attainable_peak.txt" if (top["mfma"] and args.precision == "f32") else None}
                out["kernels"] = kernels
                if light:
                    out["kernels_note"] = ("shared_mlp_* rows: HIP events inside the timed region (one eager step); the "
                                           "other rows: one instrumented eager step run after the timed region")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.model)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
