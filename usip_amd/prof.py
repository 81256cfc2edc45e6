"""Per-kernel HIP-event timing on the launch stream.

Disabled by default (one boolean test per operator call).  bench.py enables it over the timed
region: every operator wrapper brackets its kernel launch with two events recorded on torch's
current stream -- the stream the kernel is launched on -- and tags the launch with its
ALGORITHMIC work (bytes and/or flops), so that achieved GB/s or TFLOP/s per kernel come from
live measurements of the same run that produces the throughput number.
"""
from collections import defaultdict
from contextlib import contextmanager

import torch

enabled = False
_only = None                        # when set: only operators whose name starts with this prefix are bracketed
_records = defaultdict(list)        # name -> [(start_event, end_event, bytes, flops)]
_keys = {}                          # name -> rocprof key


def enable(flag: bool = True, only: str = None):
    """only="shared_mlp": bracket just the shared-MLP launches (~80 events instead of ~450 per step)."""
    global enabled, _only
    enabled = flag
    _only = only if flag else None


def reset():
    _records.clear()


@contextmanager
def kernel(name: str, nbytes: float = 0.0, flops: float = 0.0, rocprof_key=None, moved: float = 0.0):
    """rocprof_key: "<kernel template> |wg=<workgroups>" as tools/pmc_summary.py names launches, so that
    PMC traffic collected in a separate rocprofv3 pass can be attached to this operator.
    moved: bytes the launch moves INTO its CUs by construction, L2-served re-reads included (the split GEMMs re-read
    their weight planes for every position tile) -- what the per-CU memory path (~25 GB/s) has to carry."""
    if not enabled or (_only is not None and not name.startswith(_only)):
        yield
        return
    if rocprof_key is not None:
        _keys[name] = rocprof_key() if callable(rocprof_key) else rocprof_key
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()                      # current stream == launch stream of the wrapped kernel
    yield
    e.record()
    _records[name].append((s, e, nbytes, flops, moved))


def summary():
    """{name: dict(calls, total_ms, avg_us, bytes_per_call, flops_per_call, GBps, TFLOPs)} (synchronises)."""
    torch.cuda.synchronize()
    out = {}
    for name, recs in _records.items():
        ms = [r[0].elapsed_time(r[1]) for r in recs]
        tot = sum(ms)
        nb = sum(r[2] for r in recs) / len(recs)
        fl = sum(r[3] for r in recs) / len(recs)
        avg_s = tot / len(recs) * 1e-3
        out[name] = dict(rocprof_key=_keys.get(name), calls=len(recs), total_ms=tot, avg_us=avg_s * 1e6, bytes_per_call=nb,
                         moved_per_call=sum(r[4] for r in recs) / len(recs),
                         flops_per_call=fl, GBps=(nb / avg_s / 1e9) if avg_s > 0 else 0.0,
                         TFLOPs=(fl / avg_s / 1e12) if avg_s > 0 else 0.0)
    return out
