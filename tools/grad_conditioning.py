"""Why whole-step gradients are compared at EQUAL discrete decisions.  CPU only.

For each detector fixture this runs the oracle (oracle/detector.py, the PyTorch-CPU restatement of the reference's
step) four ways and reports the largest per-parameter gradient difference relative to the tensor's scale:

  free fp32 vs fp64            both forwards make their own discrete decisions (what a naive parity test compares)
  free fp32, conv vs matmul    the same fp32 math through another ATen kernel: a different summation order only
  pinned fp32 vs fp64          fp64 replays every decision of the fp32 run (pool arg-max, ReLU on/off, arg-mins, ...)
  pinned fp32 conv vs matmul   both take the fixture's (the reference's) pool arg-max and near-zero ReLU decisions

    python tools/grad_conditioning.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden                             # noqa: E402
from oracle import detector as od                            # noqa: E402
from usip_amd import synth                                   # noqa: E402
from usip_amd.networks import detector_param_shapes          # noqa: E402

torch.set_num_threads(8)


def main():
    for fix in ("detector_ball_micro.npz", "detector_som_cfg1.npz", "detector_som_micro.npz"):
        g = load_golden(fix)
        model, cs = str(g["cfg_model"]), g["in/src_sn"].shape[1]
        filled = synth.fill_parameters(detector_param_shapes(model, cs))
        batch_np = {k[3:]: v for k, v in g.items() if k.startswith("in/")}
        n_pools = sum(k.startswith("idx/pool_arg_") for k in g)
        n_relu = sum(k.startswith("idx/relu_near_idx_") for k in g)

        def fixture_tape():
            return od.DecisionTape(
                pools=[torch.from_numpy(g["idx/pool_arg_%d" % i].astype(np.int64)) for i in range(n_pools)],
                relu_fix=[(torch.from_numpy(g["idx/relu_near_idx_%d" % i].astype(np.int64)),
                           torch.from_numpy(g["idx/relu_near_on_%d" % i])) for i in range(n_relu)])

        def run(dtype, tape=None, matmul=False):
            P = {k: torch.from_numpy(v).to(dtype).requires_grad_(True) for k, v in filled.items()
                 if not ("running_" in k or "num_batches" in k)}
            bufs = {k: torch.from_numpy(v.copy()).to(dtype) for k, v in filled.items() if "running_" in k}
            od.TAPE, od.MATMUL = tape, matmul
            try:
                od.detector_step(P, bufs, {k: torch.from_numpy(v).to(dtype) for k, v in batch_np.items()}, model,
                                 int(g["cfg_knn"]), float(g["cfg_sigma_lb"]), float(g["cfg_alpha"]))
            finally:
                od.TAPE, od.MATMUL = None, False
            return P

        def worst(A, B):
            big = max(float(B[k].grad.abs().max()) for k in B)
            w, wk = 0.0, None
            for k in A:
                b = B[k].grad.double().numpy().ravel()
                s = np.abs(b).max()
                if s < 1e-5 * big:
                    continue                                  # analytically zero gradients (bias before BatchNorm)
                e = np.abs(A[k].grad.double().numpy().ravel() - b).max() / s
                if e > w:
                    w, wk = e, k
            return "%.1e (%s)" % (w, wk)

        free32, free64, free32mm = run(torch.float32), run(torch.float64), run(torch.float32, matmul=True)
        tape = fixture_tape()
        pin32 = run(torch.float32, tape)
        pin64 = run(torch.float64, od.DecisionTape(replay=tape.rec))
        pin32mm = run(torch.float32, fixture_tape(), matmul=True)
        print(fix)
        print("  free-running  fp32 vs fp64          :", worst(free32, free64))
        print("  free-running  fp32 conv vs matmul   :", worst(free32mm, free32))
        print("  equal decisions  fp32 vs fp64       :", worst(pin32, pin64))
        print("  equal decisions  fp32 conv vs matmul:", worst(pin32mm, pin32))


if __name__ == "__main__":
    main()
