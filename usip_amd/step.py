"""One detector training step = ModelDetector.optimize of the reference
(models/keypoint_detector.py:158-207), as a data-parallel unit:

    siamese forward on cat(src, dst)  ->  R*kp*s + t  ->  probabilistic chamfer
    + alpha * mean(keypoint-on-pc src) + alpha * mean(keypoint-on-pc dst)  ->  backward
    [-> one RCCL all-reduce of the flat gradient bucket]  [-> Adam]

Sharding: every (src, dst) pair is independent in forward, losses and backward;
BatchNorm statistics are per replica in the reference (nn.DataParallel, no SyncBN) and stay
per rank here.  Rank r owns pairs [r*B/W, (r+1)*B/W); the only exchange is the gradient sum.
"""
import os
from collections import OrderedDict
from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import functional as Fh
from .losses import ChamferLoss_Brute, KeypointOnPCLoss
from .networks import build_detector


class FlatGradBucket:
    """All parameter gradients of a module as views into ONE contiguous fp32 buffer, so the
    data-parallel exchange is a single all-reduce (4.79 MB for the detector) instead of 46.

    flat_params=True lays the PARAMETERS out the same way (every p.data becomes a view into `flat_param`, whose
    .grad is the gradient buffer): the optimizer then updates one 1.2 M-element tensor in one launch instead of
    walking 50 small ones (88 -> ~15 us per step for Adam).  Values, names and shapes of the module's parameters
    do not change; load_state_dict copies into the views."""

    def __init__(self, module: torch.nn.Module, flat_params: bool = False):
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_param = None
        if flat_params and all(p.dtype == torch.float32 for p in self.params):
            store = torch.empty(n, dtype=torch.float32, device=dev)
            self.flat_param = torch.nn.Parameter(store)
            self.flat_param.grad = self.flat
        off = 0
        self.offsets = {}                                     # id(parameter) -> offset of its values in the flat buffers
        for p in self.params:
            self.offsets[id(p)] = off
            if self.flat_param is not None:
                view = self.flat_param.data[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()
        for p in self.params:                                 # a new step: every parameter may be written once again
            p._usip_sink_used = False

    def all_reduce_mean(self, group=None, even_alone=False):
        """sum over ranks then / world: gradients of the mean-of-means loss (equal shards).  even_alone: issue the
        collective in a one-rank process group too (the one-GPU test that puts RCCL's all-reduce through a HIP graph)."""
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or even_alone):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))


class FlatAdam:
    """torch.optim.Adam(lr, betas=(0.9, 0.999)) (models/keypoint_detector.py:42-45) for ONE flat fp32 parameter whose
    .grad is the flat gradient bucket: a single HIP launch over five arrays (usip_adam_step_f32) instead of torch's
    fused multi-tensor kernel, which spreads a single 1.2 M-element tensor over 19 workgroups (48 -> ~6 us).  The step
    count lives on the device, so the update can be captured into a HIP graph.  `state` / `param_groups` follow
    torch.optim.Optimizer's layout (step, exp_avg, exp_avg_sq) so that a checkpoint of it reads like Adam's."""

    def __init__(self, flat_param: torch.nn.Parameter, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
        self.param = flat_param
        self.param_groups = [dict(params=[flat_param], lr=float(lr), betas=tuple(betas), eps=float(eps), weight_decay=0.0)]
        dev = flat_param.device
        self.state = {flat_param: dict(step=torch.zeros(1, dtype=torch.float32, device=dev),
                                       exp_avg=torch.zeros_like(flat_param.data),
                                       exp_avg_sq=torch.zeros_like(flat_param.data))}
        # (lr, beta1, beta2, eps) on the device: the kernel reads them there, so that a launch captured into a HIP graph
        # follows param_groups[0]['lr'] (ModelDetector.update_learning_rate, keypoint_detector.py:356-366)
        self._hyper = torch.zeros(4, dtype=torch.float32, device=dev)
        self._hyper_host = None
        self.sync_hyper()

    def sync_hyper(self):
        """Refresh the device copy of (lr, betas, eps) from param_groups when they changed (a 16-byte copy; never
        inside a stream capture: the step calls it before it replays the update)."""
        g = self.param_groups[0]
        now = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))
        if now != self._hyper_host:
            self._hyper.copy_(torch.tensor(now, dtype=torch.float32))
            self._hyper_host = now

    def step(self):
        from . import ops
        if self.param.grad is None:
            return
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        st = self.state[self.param]
        g = self.param_groups[0]
        ops.adam_step(self.param.data, self.param.grad, st["exp_avg"], st["exp_avg_sq"], st["step"], g["lr"],
                      g["betas"][0], g["betas"][1], g["eps"], hyper=self._hyper)

    def zero_grad(self, set_to_none: bool = False):
        if self.param.grad is not None:
            self.param.grad.zero_()

    def state_dict(self):
        """torch.optim.Adam's layout for ONE parameter (index 0): state[0] = {step (0-dim), exp_avg, exp_avg_sq} over
        the FLAT parameter buffer -- the per-parameter tensors are views into it in nn.Module.parameters() order, which
        is how a per-parameter Adam checkpoint (the CPU / gloo path builds torch.optim.Adam) maps onto it."""
        st = self.state[self.param]
        return dict(state={0: dict(step=st["step"].reshape(()).clone(), exp_avg=st["exp_avg"].clone(),
                                   exp_avg_sq=st["exp_avg_sq"].clone())},
                    param_groups=[{k: (v if k != "params" else [0]) for k, v in self.param_groups[0].items()}])

    def load_state_dict(self, sd):
        """Takes its own checkpoints AND a per-parameter torch.optim.Adam one of the same module (what the CPU / gloo
        path writes): the per-parameter moments are laid end to end in parameter order -- the order the flat buffer
        holds the parameters in."""
        st = self.state[self.param]
        state = sd["state"]
        if len(state) > 1 or (len(state) == 1 and torch.as_tensor(state[next(iter(state))]["exp_avg"]).numel() != st["exp_avg"].numel()):
            keys = sorted(state)
            n = sum(torch.as_tensor(state[k]["exp_avg"]).numel() for k in keys)
            if n != st["exp_avg"].numel():
                raise ValueError("FlatAdam.load_state_dict: the checkpoint holds %d moments, the flat parameter has %d"
                                 % (n, st["exp_avg"].numel()))
            for name in ("exp_avg", "exp_avg_sq"):
                st[name].copy_(torch.cat([torch.as_tensor(state[k][name]).reshape(-1).to(st[name].dtype) for k in keys]))
            st["step"].fill_(float(torch.as_tensor(state[keys[0]]["step"])))
            state = None
        for k, v in (state[0].items() if state is not None else ()):
            st[k].copy_(torch.as_tensor(v).reshape(st[k].shape))
        group = (sd.get("param_groups") or [{}])[0]
        # the kernel implements plain Adam (what the reference constructs): a checkpoint that asks for more is refused,
        # not silently trained without it
        if float(group.get("weight_decay", 0.0) or 0.0) != 0.0 or group.get("amsgrad", False) or group.get("maximize", False):
            raise ValueError("FlatAdam.load_state_dict: weight_decay / amsgrad / maximize are not implemented "
                             "(models/keypoint_detector.py:42-45 uses none of them)")
        for k, v in group.items():                            # a decayed learning rate survives a resume
            if k in ("lr", "betas", "eps"):
                self.param_groups[0][k] = tuple(v) if k == "betas" else v
        self.sync_hyper()


class _GraphedStep:
    """What DetectorStep and DescriptorStep share: the gradient bucket, the optimizer, and the optional replay
    of the step from HIP graphs.

    graph=True: the kernel launches of one step are captured once (third call; the first two run eagerly and
    are ordinary training steps) into
        graph A = zero the gradient bucket, forward, losses, backward, BatchNorm counters
        graph B = the Adam update
    with the gradient all-reduce issued eagerly between them when world > 1.  Every call copies the batch into
    the captured input buffers (skipped when the caller already passes those buffers) and replays; the kernels,
    their order and their results are those of the eager step -- only the per-launch host work and the gaps
    between launches go away (detector: 9.65 -> 9.44 ms per step).  A batch of another shape, or another
    BatchNorm momentum (epoch-dependent decay), captures a new graph; the cache keeps the `max_graphs` most recently
    used ones."""

    def _setup(self, module, opt, device, with_optimizer: bool, graph: bool):
        self.opt = opt
        self.device = torch.device(device)
        self.module = module
        on_gpu = self.device.type == "cuda"
        self.bucket = FlatGradBucket(module, flat_params=with_optimizer and on_gpu)
        self.use_graph = bool(graph) and on_gpu
        self.optimizer = None
        if with_optimizer:                                    # keypoint_detector.py:42-45, keypoint_descriptor.py:38-41
            # Adam over ONE flat parameter (the module's parameters are views into it): a single fused launch;
            # capturable keeps its step counter on the device so that the update can live in a HIP graph
            if self.bucket.flat_param is not None:
                self.optimizer = FlatAdam(self.bucket.flat_param, lr=opt.lr, betas=(0.9, 0.999))
            else:                                             # CPU (gloo tests): torch's own implementation
                self.optimizer = torch.optim.Adam(module.parameters(), lr=opt.lr, betas=(0.9, 0.999))
        # K-major copies of all convolution weights ([Cout, Cin, 1(, 1)] -> [Cin, Cout]) by ONE launch per step
        self._wt = None
        if self.bucket.flat_param is not None:
            rows, views, dof, tiles = [], {}, 0, 0
            for p in self.bucket.params:
                if p.dim() >= 3 and p.numel() == p.shape[0] * p.shape[1]:
                    co, ci = int(p.shape[0]), int(p.shape[1])
                    rows.append((self.bucket.offsets[id(p)], co, ci, dof, tiles))
                    views[p.data_ptr()] = (dof, ci, co)
                    dof += co * ci
                    tiles += ((co + 31) // 32) * ((ci + 31) // 32)
            if rows:
                flat_wt = torch.empty(dof, dtype=torch.float32, device=self.device)
                table = torch.tensor(rows, dtype=torch.int32, device=self.device)
                self._wt = (flat_wt, table, tiles,
                            {ptr: flat_wt[o:o + ci * co].view(ci, co) for ptr, (o, ci, co) in views.items()})
        self.last: Dict[str, torch.Tensor] = {}
        # key -> dict(a = graph A, b = graph B or None, static = captured input buffers, last, loss, fused = one graph
        # for the whole step, reduces = that graph holds the all-reduce, checked), least recently used first.  Every entry owns
        # a private memory pool with all activations of a step (GBs at N=16384), so the cache is bounded: a
        # training run with epoch-decayed BatchNorm momentum or varying cloud sizes evicts instead of growing.
        self._graphs: "OrderedDict" = OrderedDict()
        self.max_graphs = 3
        # bench.py: set to a list to have every gradient all-reduce bracketed by two HIP events on the launch stream
        # (the collective itself runs on RCCL's stream; the launch stream waits for it, so the pair spans it)
        self.allreduce_events = None
        self.allreduce_in_graph = False                      # True once a step graph contains the all-reduce (RCCL)
        self.exchange_even_alone = False                     # tests: treat a ONE-rank process group as data-parallel
        self._eager_calls = 0
        self._bns = [m for m in module.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
        self._bn_counters = [m.num_batches_tracked for m in self._bns if m.num_batches_tracked is not None]

    def load_numpy_state(self, state: Dict):
        sd = self.module.state_dict()
        self.module.load_state_dict({k: torch.as_tensor(v).reshape(sd[k].shape) for k, v in state.items()})
        # load_state_dict copies in place, gradients keep pointing into the flat bucket

    def _prepare(self, batch):
        """Hook: host-side inputs the step adds to the batch (the descriptor's point permutation)."""
        return batch

    def _static_from(self, batch):
        """Hook: the persistent input buffers a captured graph reads (the caller's tensors are copied into them before
        every replay)."""
        return {k: v.clone() for k, v in batch.items()}

    def forward_losses(self, batch, epoch=None):
        raise NotImplementedError

    def _forward_backward(self, batch, epoch):
        from . import functional as Fh
        from . import ops
        if Fh.pins_active() and not getattr(self, "allow_pinned_decisions", False):
            # the decision-pinning hooks are test instruments (tests/test_modules_gpu.py): a training step never runs
            # with them unless the step object was built for such a test, and never inside a captured graph
            raise RuntimeError("usip_amd: functional.pinned_decisions is active during a training step")
        if Fh.pins_active() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("usip_amd: pinned decisions cannot be captured into a HIP graph")
        self.bucket.zero()                                    # zero_grad()
        Fh.GRAD_SINK = True       # every parameter is used once per step and the bucket was just zeroed:
        Fh.DEFER_BN_COUNTERS = True   # the backward kernels write dW/dgamma/dbeta straight into the bucket
        try:
            if self._wt is not None:                          # fresh K-major weights for every layer, one launch
                flat_wt, table, tiles, views = self._wt
                ops.multi_transpose(self.bucket.flat_param.data, flat_wt, table, tiles)
                Fh.WT_CACHE = views
            # f32x3 mode: every weight operand is split into its bf16 planes once per step -- by one launch for all of
            # them once a completed step has shown which operands are asked for
            plan = getattr(self, "_planes_plan", None)
            ops.PLANES_CACHE = plan.refresh() if plan is not None else {}
            Fh.PRE_BN_SUMS.clear()
            loss = self.forward_losses(batch, epoch)
            # the weight gradients are read by the all-reduce / the optimizer only: their fifteen fixed-order sums of
            # partial tiles are recorded during backward and issued together behind it (USIP_DEFER_WGRAD=0: one by one)
            defer = self.device.type == "cuda" and os.environ.get("USIP_DEFER_WGRAD", "1") not in ("0", "off")
            if defer:
                ops.wgrad_defer(True, self.device)
            try:
                loss.backward()
                if defer:
                    self.deferred_reductions = ops.wgrad_flush(self.device)
            finally:
                if defer:
                    ops.wgrad_defer(False)
            if plan is None and ops.PLANES_CACHE and not torch.cuda.is_current_stream_capturing():
                self._planes_plan = ops.PlanesPlan(ops.PLANES_CACHE, [self.bucket.flat_param,
                                                                      self._wt[0] if self._wt is not None else None])
        finally:
            Fh.GRAD_SINK = False
            Fh.DEFER_BN_COUNTERS = False
            Fh.WT_CACHE = None
            ops.PLANES_CACHE = None
            Fh.PRE_BN_SUMS.clear()
        if self._bn_counters:
            torch._foreach_add_(self._bn_counters, 1)         # every BatchNorm ran exactly once
        return loss

    def step(self, batch: Dict[str, torch.Tensor], epoch: Optional[int] = None, group=None, eager: bool = False):
        """forward + losses + backward (+ gradient all-reduce) (+ Adam when constructed with it).
        eager=True forces plain launches for this call (bench.py does so on the steps it instruments with
        HIP events, which a graph replay cannot carry)."""
        self._replay_this_call = True
        self._prepare_group = group                           # the ranks this step's batch is sharded over (random point dropout)
        batch = self._prepare(batch)
        if self.use_graph and not eager and self._replay_this_call:
            return self._step_graph(batch, epoch, group)
        return self._step_eager(batch, epoch, group)

    def _exchanges(self, group) -> bool:
        """True when this step object exchanges gradients: a process group of more than one rank (or, for the one-GPU
        RCCL test, of exactly one with exchange_even_alone), and not bench.py's solo probe."""
        if getattr(self, "solo", False) or not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size(group) > 1 or self.exchange_even_alone

    def _all_reduce(self, group):
        if getattr(self, "solo", False):                      # bench.py's single-rank probe inside a multi-rank job
            return
        if self.allreduce_events is not None and self.device.type == "cuda":
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            self.bucket.all_reduce_mean(group, self.exchange_even_alone)
            e.record()
            self.allreduce_events.append((s, e))
        else:
            self.bucket.all_reduce_mean(group, self.exchange_even_alone)

    def _step_eager(self, batch, epoch, group):
        loss = self._forward_backward(batch, epoch)
        self._all_reduce(group)
        if self.optimizer is not None:
            self.optimizer.step()
        return loss

    def _bn_momenta(self, epoch):
        """The momentum every BatchNorm will run this step with (a kernel argument, so part of what a captured
        graph froze): epoch-driven modules switch to their decayed value (models/layers.py:61-71), the others --
        and all of them while the rule does not apply -- keep what they have.  The value changes once every
        `bn_momentum_decay_step` epochs and stops at the 0.01 clamp, so keying on it (not on the epoch) re-captures
        only when a replay would be wrong."""
        out = []
        for bn in self._bns:
            m = bn.momentum
            if getattr(bn, "epoch_driven", False):
                nxt = bn.momentum_for(epoch)
                if nxt is not None:
                    m = nxt
            out.append(None if m is None else round(float(m), 12))
        return tuple(out)

    def _step_graph(self, batch, epoch, group):
        decays = getattr(self.opt, "bn_momentum_decay_step", None)
        key = (tuple((k, tuple(v.shape)) for k, v in sorted(batch.items())),
               self._bn_momenta(epoch) if (decays is not None and decays > 0) else None)
        entry = self._graphs.get(key)
        if entry is not None:
            self._graphs.move_to_end(key)
        if entry is None:
            if self._eager_calls < 2:                         # allocator, rocBLAS handles, lazily built state
                self._eager_calls += 1
                return self._step_eager(batch, epoch, group)
            static = self._static_from(batch)
            torch.cuda.synchronize(self.device)
            world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
            # One graph for the whole step when the gradient exchange can be captured: RCCL (backend "nccl") enqueues
            # its collectives on a stream and supports stream capture, so forward + backward + all-reduce + Adam replay
            # as ONE graph launch per step and nothing is left exposed between two replays.  OPT-IN since round 5
            # (USIP_GRAPH_ALLREDUCE=1): the form has run on ONE rank only (tests/test_data_parallel_gpu.py) and the first
            # multi-device run is the driver's -- until a run on N devices has validated it the default is the plain
            # graph A / eager all-reduce / graph B.  gloo (the CPU tests, the several-ranks-on-one-GPU
            # debugging mode) cannot be captured at all, and a refused capture falls back.
            # (only ever attempted with RCCL: a gloo all-reduce inside a capture aborts the process -- tried, r04aa)
            distributed = self._exchanges(group)
            fuse = (distributed and not getattr(self, "solo_fuse_off", False)
                    and os.environ.get("USIP_GRAPH_ALLREDUCE", "0") in ("1", "on")
                    and (dist.get_backend(group) == "nccl" or getattr(self, "_test_fused_without_reduce", False))
                    and self.allreduce_events is None)
            # a single process has nothing between backward and update either: one graph (USIP_GRAPH_ONE=0: two, for A/B)
            fuse = fuse or (not distributed and self.optimizer is not None
                            and os.environ.get("USIP_GRAPH_ONE", "1") not in ("0", "off"))
            entry = None
            if fuse:
                g1 = loss = None
                try:
                    g1 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                        loss = self._forward_backward(static, epoch)
                        # (_test_fused_without_reduce: tests/ only -- stands in for a captured collective that turns out
                        # to be a no-op, the failure the check below exists for)
                        if distributed and not getattr(self, "_test_fused_without_reduce", False):
                            self.bucket.all_reduce_mean(group, self.exchange_even_alone)
                        if self.optimizer is not None:
                            self.optimizer.step()
                    ok = True
                except Exception as err:                      # noqa: BLE001  (a backend that cannot be captured says so its own way)
                    import warnings
                    warnings.warn("usip_amd: the gradient all-reduce could not be captured (%s); replaying two graphs "
                                  "with the all-reduce between them" % err)
                    torch.cuda.synchronize(self.device)
                    ok = False
                if distributed:
                    # every rank adopts the one-graph form or none does: a rank whose capture failed would otherwise
                    # issue an eager all-reduce its peers never join (they replay theirs from the graph) -- and hang
                    ok = self._all_ranks_agree(ok, group)
                if ok:
                    entry = dict(a=g1, b=None, static=static, last=dict(self.last), loss=loss, fused=True,
                                 reduces=distributed, checked=not distributed)
                    self.allreduce_in_graph = distributed
                else:
                    del g1
                    if distributed:
                        self.solo_fuse_off = True             # one refusal is enough: later captures go two-graph directly
            if entry is None:
                try:
                    # thread_local: calls other threads make meanwhile (a collective watchdog polling its events)
                    # must not invalidate the capture
                    ga = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                        loss = self._forward_backward(static, epoch)
                    last = dict(self.last)
                    gb = None
                    if self.optimizer is not None:
                        gb = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gb, capture_error_mode="thread_local"):
                            self.optimizer.step()
                except RuntimeError as err:                   # capture refused: keep training with plain launches
                    import warnings
                    warnings.warn("usip_amd: HIP graph capture failed (%s); continuing with eager launches" % err)
                    torch.cuda.synchronize(self.device)
                    self.use_graph = False
                    return self._step_eager(batch, epoch, group)
                entry = dict(a=ga, b=gb, static=static, last=last, loss=loss, fused=False, reduces=False, checked=True)
            self._graphs[key] = entry                         # capture launches nothing: replay below
            while len(self._graphs) > self.max_graphs:
                _, old = self._graphs.popitem(last=False)     # drops the graphs and, with them, their memory pool
                del old
        static = entry["static"]
        for k, v in batch.items():
            if v.data_ptr() != static[k].data_ptr():
                static[k].copy_(v, non_blocking=True)
        if key[1] is not None:                                # what the Python forward would have left behind
            for bn, m in zip(self._bns, key[1]):
                bn.momentum = m
        if entry["fused"]:
            if hasattr(self.optimizer, "sync_hyper"):
                self.optimizer.sync_hyper()                   # (before the replay: the update is inside it)
            entry["a"].replay()
            self.last = entry["last"]
            if entry["reduces"] and not entry["checked"]:
                # once per CAPTURED GRAPH (a re-capture for another shape or BatchNorm momentum is checked again): after
                # an all-reduce every rank holds the same gradient -- if the captured collective did not do what the
                # eager one does, say so now instead of training on garbage
                entry["checked"] = True
                if not self._gradients_agree(group):
                    # the captured collective did not reduce: one step was taken on un-averaged gradients.  Put the
                    # replicas back together (rank 0's parameters and optimizer state), give up the fused form for
                    # this step object and go on with graph A / eager all-reduce / graph B.
                    import warnings
                    warnings.warn("usip_amd: gradients differ across ranks after the captured all-reduce; replicas "
                                  "re-synchronised from rank 0, continuing with the all-reduce between two graphs")
                    self._resync_from_rank0(group)
                    self.solo_fuse_off = True
                    self.allreduce_in_graph = False
                    self.fused_fallbacks = getattr(self, "fused_fallbacks", 0) + 1
                    for k in [k for k, e in self._graphs.items() if e["reduces"]]:
                        del self._graphs[k]
            return entry["loss"]
        entry["a"].replay()
        self._all_reduce(group)
        if entry["b"] is not None:
            if hasattr(self.optimizer, "sync_hyper"):
                self.optimizer.sync_hyper()                   # a changed learning rate reaches the captured update
            entry["b"].replay()
        self.last = entry["last"]
        return entry["loss"]

    def _all_ranks_agree(self, ok: bool, group) -> bool:
        """Logical AND of `ok` over the ranks (an eager MIN all-reduce of one flag)."""
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return bool(int(flag.item()))

    def _gradients_agree(self, group) -> bool:
        """True when every rank holds the same reduced gradient: the fp64 sum of the flat bucket, compared as BIT
        PATTERNS (a NaN gradient is the same NaN everywhere after a sum all-reduce; comparing floats would call that
        a disagreement and re-synchronise for nothing)."""
        mine = self.bucket.flat.detach().double().sum().reshape(1).view(torch.int64)
        every = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
        dist.all_gather(every, mine, group=group)
        return all(torch.equal(e, every[0]) for e in every)

    def allreduce_form(self) -> str:
        """How the gradient exchange is issued right now (bench.py logs it): "one graph" = captured RCCL all-reduce,
        "two graphs" = eager all-reduce between graph A and graph B, "eager", or "none" (single process)."""
        if not self._exchanges(None):
            return "none"
        if not self.use_graph:
            return "eager"
        if self.allreduce_in_graph:
            return "one graph (captured all-reduce)"
        return "two graphs (eager all-reduce between them)" + (
            "; captured form refused or failed its check" if getattr(self, "solo_fuse_off", False) else "")

    def _resync_from_rank0(self, group):
        """Every replica takes rank 0's parameters, BatchNorm buffers and optimizer state (used once, if ever: when a
        captured all-reduce turned out not to reduce)."""
        src = dist.get_global_rank(group, 0) if group is not None else 0
        tensors = [p.data for p in self.module.parameters()] + [b for b in self.module.buffers()]
        opt = self.optimizer
        if opt is not None and hasattr(opt, "state"):
            for st in opt.state.values():
                tensors += [v for v in st.values() if torch.is_tensor(v)]
        seen = set()
        nccl = dist.get_backend(group) == "nccl"
        for t in tensors:
            if nccl and not t.is_cuda:
                continue                                      # (a host-side step counter: RCCL moves device memory only)
            base = t._base if t._base is not None else t
            if base.data_ptr() in seen:
                continue                                      # views of one flat buffer: once
            seen.add(base.data_ptr())
            dist.broadcast(base, src=src, group=group)

    def static_batch(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """The captured input buffers for batches shaped like `batch` (None before capture): a loader that
        writes into them saves the per-step device copies."""
        for entry in self._graphs.values():
            static = entry["static"]
            if all(k in static and static[k].shape == v.shape for k, v in batch.items()):
                return {k: static[k] for k in batch}
        return None


class DetectorStep(_GraphedStep):
    """ModelDetector.optimize (models/keypoint_detector.py:158-207) on one device: owns detector + criteria
    (+ optimizer); see _GraphedStep for graph=True."""

    def __init__(self, model: str, opt, device, with_optimizer: bool = False, graph: bool = False):
        self.detector = build_detector(model, opt).to(torch.device(device))
        self.chamfer_criteria = ChamferLoss_Brute(opt)
        self.keypoint_on_pc_criteria = KeypointOnPCLoss(opt)
        self._setup(self.detector, opt, device, with_optimizer, graph)

    _SIAMESE = (("src_pc", "dst_pc", "_cat_pc"), ("src_sn", "dst_sn", "_cat_sn"), ("src_node", "dst_node", "_cat_node"))

    def _static_from(self, batch):
        # src and dst halves as views of ONE buffer each: the siamese concatenation (keypoint_detector.py:141-146) then
        # exists already when the captured step starts -- three torch.cat launches fewer per replay
        static = {k: v.clone() for k, v in batch.items() if not any(k in t[:2] for t in self._SIAMESE)}
        for a, b, both in self._SIAMESE:
            cat = torch.cat((batch[a], batch[b]), 0)
            n = batch[a].shape[0]
            static[a], static[b], static[both] = cat[:n], cat[n:], cat
        return static

    def _prepare(self, batch):
        """Random point dropout of ModelDetector.optimize (keypoint_detector.py:160-168): when
        opt.random_pc_dropout_lower_limit < 0.99, a keep ratio U(limit, 1) and that many point indices WITHOUT
        replacement are drawn on the host every step (random.uniform + np.random.choice, as there) and ONE index set
        selects the same points of src and dst clouds and normals.  The indices travel as batch["keep_idx"] (int64 [n]
        device tensor; pass it yourself to fix the choice -- the parity fixture does).  With several ranks rank 0 draws
        and broadcasts, so the whole (sharded) batch sees one set per step as under nn.DataParallel.  The number of kept
        points changes from step to step, so such steps run from plain launches even with graph=True."""
        limit = getattr(self.opt, "random_pc_dropout_lower_limit", 1.0)
        self._replay_this_call = True
        if "keep_idx" not in batch and limit < 0.99:
            import random
            import numpy as np
            n_in = int(getattr(self.opt, "input_pc_num", batch["src_pc"].shape[2]))
            # ONE draw per step for the whole batch, as in the reference (a single process behind nn.DataParallel):
            # rank 0 draws, every rank trains on the same keep ratio and index set -- equal cloud sizes per step
            # (the group handed to step(), not the world: a sub-group's ranks must not wait for ranks that never call; a solo
            # step object -- bench.py's one-rank probe inside a multi-rank job -- draws for itself: ADVICE r5)
            group = getattr(self, "_prepare_group", None)
            multi = (not getattr(self, "solo", False) and dist.is_available() and dist.is_initialized()
                     and dist.get_world_size(group) > 1)
            first = multi and dist.get_rank(group) == 0
            if not multi or first:
                keep = round(random.uniform(limit, 1.0) * n_in)
                idx = torch.from_numpy(np.random.choice(n_in, keep, replace=False)).to(self.device)
            if multi:
                src = dist.get_global_rank(group, 0) if group is not None else 0
                count = torch.tensor([keep if first else 0], dtype=torch.int64, device=self.device)
                dist.broadcast(count, src=src, group=group)
                if not first:
                    idx = torch.empty(int(count.item()), dtype=torch.int64, device=self.device)
                dist.broadcast(idx, src=src, group=group)
            batch = dict(batch, keep_idx=idx)
        if "keep_idx" in batch:
            if self.use_graph and not getattr(self, "_warned_dropout", False):
                import warnings                                # a new point count almost every step: replay cannot pay off
                warnings.warn("usip_amd: random point dropout changes the cloud size every step; steps that carry "
                              "keep_idx run from plain launches, not from a HIP graph")
                self._warned_dropout = True
            self._replay_this_call = False                    # this call only: a later batch without keep_idx replays again
            batch = dict(batch)
            idx = batch.pop("keep_idx").long()
            self.last_keep = int(idx.numel())                 # points per cloud this step trains on
            for k in ("src_pc", "src_sn", "dst_pc", "dst_sn"):
                batch[k] = torch.index_select(batch[k], 2, idx).contiguous()
            for k in [both for _, _, both in self._SIAMESE if both in batch]:
                del batch[k]                                 # concatenated views of the undropped clouds
        return batch

    def forward_losses(self, batch: Dict[str, torch.Tensor], epoch: Optional[int] = None):
        B = batch["src_pc"].shape[0]
        self.detector.train()                                 # keypoint_detector.py:171
        pc, sn, node = (batch[both] if both in batch else torch.cat((batch[a], batch[b]), 0)
                        for a, b, both in self._SIAMESE)
        nodes, kp, sg, _ = self.detector(pc, sn, node, True, epoch)   # forward_siamese :141-156
        kp_src, kp_dst = torch.split(kp, B, dim=0)            # :147-149 (split: one backward node, no zero fills)
        sg_src, sg_dst = torch.split(sg, B, dim=0)
        # :182-184  R.kp*s + t  as one batched GEMM with the scale folded into R (inputs, no gradient)
        kp_t = Fh.rigid_transform(kp_src, batch["R"], batch["scale"], batch["shift"])
        loss_chamfer, pure, weighted = self.chamfer_criteria(kp_t, kp_dst, sg_src, sg_dst)
        alpha = self.opt.keypoint_on_pc_alpha
        # :196-203  keypoint-on-pc for src and dst: both clouds of every pair in ONE nearest-neighbour launch
        # (rows [0,B) = src, [B,2B) = dst; every cloud is independent, so the values are the reference's)
        if getattr(self.opt, "keypoint_on_pc_type", "point_to_point") == "point_to_plane":     # :197-201
            on = self.keypoint_on_pc_criteria(kp, pc, sn)                    # 2B x M x 1 x 1
            on_src, on_dst = torch.mean(on[:B]) * alpha, torch.mean(on[B:]) * alpha
            loss = loss_chamfer + on_src + on_dst
        else:
            loss, on_src, on_dst = Fh.detector_loss_combine(self.keypoint_on_pc_criteria(kp, pc, None), loss_chamfer,
                                                            alpha)   # :204
        self.last = dict(node=nodes, keypoints=kp, sigmas=sg, loss=loss, loss_chamfer=loss_chamfer,
                         chamfer_pure=pure, chamfer_weighted=weighted, loss_on_pc_src=on_src,
                         loss_on_pc_dst=on_dst)
        return loss


class DescriptorStep(_GraphedStep):
    """ModelDescriptor.optimize (models/keypoint_descriptor.py:126-159): siamese descriptor forward on
    cat(anchor, positive), triplet loss with in-batch negatives, backward (+ all-reduce) (+ Adam).
    The random point permutation of the reference (networks.py:345-347) is drawn on the host every step and
    handed to the model as a device tensor (batch["perm"]; given by the caller to fix it), which also keeps the
    host-to-device copy out of a captured graph."""

    def __init__(self, opt, device, with_optimizer: bool = False, graph: bool = False):
        from .losses import DescPairScanLoss
        from .networks import DescriptorLiteOld
        self.descriptor = DescriptorLiteOld(opt).to(torch.device(device))
        self.triplet_criteria = DescPairScanLoss(opt)
        self._setup(self.descriptor, opt, device, with_optimizer, graph)

    def _prepare(self, batch):
        if "perm" in batch:
            return batch
        import numpy as np
        perm = self.descriptor.fixed_permutation
        if perm is None:
            perm = np.random.permutation(batch["anc_pc"].shape[2])
        return dict(batch, perm=torch.as_tensor(perm, dtype=torch.int64).to(self.device))

    def forward_losses(self, batch: Dict[str, torch.Tensor], epoch: Optional[int] = None):
        B = batch["anc_pc"].shape[0]
        self.descriptor.train()
        desc, x_feat = self.descriptor(torch.cat((batch["anc_pc"], batch["pos_pc"]), 0),
                                       torch.cat((batch["anc_sn"], batch["pos_sn"]), 0),
                                       torch.cat((batch["anc_kp"], batch["pos_kp"]), 0), True, epoch,
                                       perm=batch.get("perm"))
        anc, pos = torch.split(desc, B, dim=0)
        triplet, active = self.triplet_criteria(anc, pos, anc[batch["neg_idx"], :, :], batch["anc_sigmas"])
        loss = torch.mean(triplet)
        self.last = dict(descriptors=desc, x_features=x_feat, triplet=triplet, active=active, loss=loss)
        return loss


def batch_to_device(batch_np: Dict, device) -> Dict[str, torch.Tensor]:
    return {k: torch.as_tensor(v).to(device) for k, v in batch_np.items()}


def shard_pairs(batch_np: Dict, rank: int, world: int) -> Dict:
    """Rank r's contiguous slice of pairs."""
    B = batch_np["src_pc"].shape[0]
    assert B % world == 0, "pairs must divide evenly over ranks"
    per = B // world
    return {k: v[rank * per:(rank + 1) * per] for k, v in batch_np.items()}
