"""Per-parameter gradient figures of the non-default-options fixtures (debug aid for
tests/test_modules_gpu.py::test_detector_step_with_non_default_layer_options_matches_reference)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import test_modules_gpu as T
from usip_amd import ops
ops.set_matmul_mode(os.environ.get("USIP_MATMUL_MODE", "f32"))

for fix in sys.argv[1:] or ["detector_ball_elu_instance.npz", "detector_som_swish.npz", "detector_som_k2.npz"]:
    g, st = T._run_step(fix)
    print("==", fix)
    for k, p in st.detector.named_parameters():
        if p.grad is None:
            print("  %-50s NO GRAD" % k); continue
        gr = p.grad.detach().cpu().numpy().ravel().astype(np.float64)
        ref = g["grad_head/" + k].astype(np.float64)
        n = float(g["grad_norm/" + k])
        print("  %-50s head %.3e  norm %.4e ref %.4e" % (k, np.abs(gr[:48] - ref).max() / max(np.abs(gr).max(), 1e-30),
                                                          np.sqrt((gr ** 2).sum()), n))
