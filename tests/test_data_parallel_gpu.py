"""The N > 1 step on a ONE-GPU box: what can be executed of the data-parallel path without
an 8-GPU node.

* the fall-back chain of the one-graph step -- a captured all-reduce that turns out not to reduce is detected on the
  first replay, the replicas are re-synchronised from rank 0 and the two-graph form takes over (two gloo ranks sharing
  cuda:0; the "captured collective that is a no-op" is injected, gloo itself cannot be captured);
* RCCL itself: a ONE-rank "nccl" process group (RCCL refuses two ranks per device) puts the real all-reduce enqueue
  through the eager path, through a HIP-graph capture and through replays of the whole step;
* `bench.py --gpus 8` end to end with eight gloo ranks on the one device: census, affinity slicing, identical
  replicas, and the length of the line the driver parses.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_step(rank, graph=True):
    from usip_amd import ops, synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DetectorStep, batch_to_device
    ops.set_matmul_mode("f32x2")
    opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=16)
    torch.manual_seed(0)                                       # identical replicas
    st = DetectorStep("ball", opt, DEV, with_optimizer=True, graph=graph)
    batch = batch_to_device(synth.make_pair_batch(1234 + rank, 2, 2048, 64, 4, "slab"), DEV)
    return st, batch


def _params_bits(st):
    return st.bucket.flat_param.detach().clone()


def _fallback_worker(rank, world, port, q):
    import warnings
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    os.environ["USIP_GRAPH_ALLREDUCE"] = "1"                   # the one-graph form is opt-in
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st, batch = _make_step(rank)
    st._test_fused_without_reduce = True                       # the captured "collective" does nothing
    seen = []
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for i in range(7):
            st.step(batch)
            torch.cuda.synchronize()
            seen.append((st.allreduce_in_graph, st.allreduce_form(), getattr(st, "fused_fallbacks", 0)))
    msgs = [str(x.message) for x in w]
    p = _params_bits(st).cpu()
    both = [torch.empty_like(p) for _ in range(world)]
    dist.all_gather(both, p)
    # a reference: the same seven steps with the eager all-reduce from the start (no graph): the step on un-averaged
    # gradients was UNDONE by the re-synchronisation only in the sense that the replicas agree again -- rank 0's
    # trajectory is what everybody follows, so the comparison is across ranks, not against this run
    q.put(dict(rank=rank, seen=seen, warned=any("re-synchronised from rank 0" in m for m in msgs),
               identical=bool(torch.equal(both[0], both[1])), finite=bool(torch.isfinite(p).all()),
               fused_off=bool(getattr(st, "solo_fuse_off", False)),
               graphs=[(e["fused"], e["reduces"]) for e in st._graphs.values()]))
    dist.barrier()
    dist.destroy_process_group()


def test_a_captured_allreduce_that_does_not_reduce_is_caught_and_the_two_graph_form_takes_over():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fallback_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in res:
        # steps 0, 1 eager; step 2 captures the (fake) one-graph form, replays it, the check fails -> fallback
        assert r["seen"][1][0] is False and r["seen"][1][1].startswith("two graphs")
        assert r["seen"][2] == (False, "two graphs (eager all-reduce between them); captured form refused or failed its check", 1)
        assert r["seen"][-1][2] == 1                             # one fallback, not one per step
        assert r["warned"] and r["fused_off"] and r["finite"]
        assert r["graphs"] == [(False, False)]                   # the fused entry is gone, the two-graph one serves
        assert r["identical"]                                    # replicas bit-identical after the recovery + 4 more steps


def _rccl_one_rank_worker(port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    os.environ["USIP_GRAPH_ALLREDUCE"] = "1"                   # the one-graph form is opt-in
    torch.cuda.set_device(0)
    dev = torch.device(DEV)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {"rccl": ".".join(str(v) for v in torch.cuda.nccl.version())}
    # (1) eager RCCL all-reduce of the flat bucket, one rank: sum over one rank / 1 = the same bits
    st, batch = _make_step(0, graph=False)
    st.exchange_even_alone = True
    ref, _ = _make_step(0, graph=False)                        # the same steps with no process-group traffic at all
    for _ in range(3):
        st.step(batch)
        ref.step(batch)
    torch.cuda.synchronize()
    out["eager_equal"] = bool(torch.equal(_params_bits(st), _params_bits(ref)))
    # (2) the whole step as ONE graph with RCCL's all-reduce captured inside, replayed
    g, _ = _make_step(0, graph=True)
    g.exchange_even_alone = True
    e, _ = _make_step(0, graph=False)
    forms = []
    for _ in range(6):
        g.step(batch)
        e.step(batch)
        forms.append(g.allreduce_form())
    torch.cuda.synchronize()
    out["forms"] = forms
    out["in_graph"] = bool(g.allreduce_in_graph)
    out["entries"] = [(x["fused"], x["reduces"], x["checked"]) for x in g._graphs.values()]
    out["graph_equal"] = bool(torch.equal(_params_bits(g), _params_bits(e)))
    out["fallbacks"] = getattr(g, "fused_fallbacks", 0)
    # (3) the DEFAULT (no opt-in): graph A / eager RCCL all-reduce / graph B
    os.environ.pop("USIP_GRAPH_ALLREDUCE")
    d, _ = _make_step(0, graph=True)
    d.exchange_even_alone = True
    e2, _ = _make_step(0, graph=False)
    for _ in range(5):
        d.step(batch)
        e2.step(batch)
    torch.cuda.synchronize()
    out["default_form"] = d.allreduce_form()
    out["default_entries"] = [(x["fused"], x["reduces"]) for x in d._graphs.values()]
    out["default_equal"] = bool(torch.equal(_params_bits(d), _params_bits(e2)))
    q.put(out)
    dist.destroy_process_group()


def test_rccl_allreduce_runs_eagerly_and_inside_the_captured_step_on_one_rank():
    """RCCL executes: `dist.all_reduce` on backend "nccl" with the flat gradient bucket, eagerly and captured into the
    one-graph step (forward, backward, all-reduce, Adam) that is replayed four times.  One rank, because RCCL refuses two
    per device: the collective is the degenerate one, the ENQUEUE path (stream capture of RCCL's launch, the
    all-ranks-agree flag, the first-replay check) is the real one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert out["eager_equal"], out
    assert all(f.startswith("two graphs") for f in out["forms"][:2])      # the two eager set-up steps: nothing captured yet
    assert out["in_graph"] and out["forms"][-1] == "one graph (captured all-reduce)", out
    assert out["entries"] == [(True, True, True)] and out["fallbacks"] == 0, out
    assert out["graph_equal"], out                              # replaying RCCL from the graph changes no bit
    assert out["default_form"].startswith("two graphs") and out["default_entries"] == [(False, False)], out
    assert out["default_equal"], out                            # graph A / eager RCCL all-reduce / graph B: the same bits


def test_bench_eight_ranks_gloo_one_gpu(tmp_path):
    """`python bench.py --gpus 8`, self-spawned, eight gloo ranks on the one device, tiny clouds: the census lists
    eight ranks, every rank got its own core slice, the replicas end bit-identical, and the line is one the driver
    can read (<= 12 KB, `roofline` aside -- kernel timing is off here -- every contract key present)."""
    env = dict(os.environ, USIP_DIST_BACKEND="gloo", USIP_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               USIP_BENCH_OUT=str(tmp_path))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--pairs", "1",
           "--points", "2048", "--nodes", "64", "--no-cpu-baseline", "--no-fp32-leg"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    text = res.stdout.decode()
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and text.strip().splitlines()[-1] == lines[0]
    assert len(lines[0]) <= 12000, len(lines[0])
    line = json.loads(lines[0])
    full = json.load(open(tmp_path / "bench_full_n8.json"))
    assert line["n_gpus"] == 8 and line["config"]["parallelism"] == "dp8" and line["scaling"] == "weak"
    assert [r["rank"] for r in line["ranks"]] == list(range(8))
    assert len({r["checksum"] for r in line["ranks"]}) == 1 and line["distributed"]["replicas_identical"] is True
    assert line["distributed"]["world_size"] == 8 and line["distributed"]["allreduce_form"].startswith("two graphs")
    assert line["distributed"]["n1_probe"]["ratio"] > 0
    assert "roofline" in line and "ms_per_step" in line and line["steps"] == 3
    assert len({r["pid"] for r in full["ranks_seen"]}) == 8
    cores = [tuple(r.get("cores") or ()) for r in full["ranks_seen"]]
    if all(cores):                                             # (a host with fewer than 16 cores keeps the inherited set)
        flat = [c for cs in cores for c in cs]
        assert len(flat) == len(set(flat)), "core slices overlap"
    assert value_is_sum_of_ranks(line)


def value_is_sum_of_ranks(line):
    clouds = line["n_gpus"] * line["config"]["clouds_per_gpu"] * line["steps"]
    return abs(line["value"] - clouds / (line["ms_per_step"] * 1e-3 * line["steps"])) <= 1e-6 * line["value"]
