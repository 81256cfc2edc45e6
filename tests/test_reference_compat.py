"""Drop-in check that needs the reference checkout (build container only; skipped on the GPU box):
the reference's OWN models/networks.py, imported unchanged on top of usip_amd's modules as INTEGRATION.md
describes, builds the same detectors with the same state_dict keys and shapes as with its own layers."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

REF = "/root/reference"

SCRIPT = textwrap.dedent("""
    import sys, types, json
    import matplotlib; matplotlib.use("Agg")
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    sys.modules["torchvision"] = types.ModuleType("torchvision")
    mode = sys.argv[1]
    if mode == "ours":
        import usip_amd
        usip_amd.install()                                   # index_max, ball_query drop-ins
        from usip_amd import layers, losses, operations
        import models                                        # the reference's package
        sys.modules["models.layers"] = layers
        sys.modules["models.losses"] = losses
        sys.modules["models.operations"] = operations
        models.layers, models.losses, models.operations = layers, losses, operations
    else:
        for name in ("index_max", "ball_query"):
            sys.modules[name] = types.ModuleType(name)       # never called: construction only
    from models import networks                              # the reference's file, unchanged
    class Opt: pass
    opt = Opt()
    opt.surface_normal_len, opt.activation, opt.normalization = 4, "relu", "batch"
    opt.bn_momentum, opt.bn_momentum_decay_step, opt.bn_momentum_decay = 0.1, None, 0.6
    opt.k, opt.node_knn_k_1, opt.loss_sigma_lower_bound = 1, 16, 1e-3
    out = {}
    for cls in ("RPN_Detector", "RPN_DetectorLite", "RPN_Detector_Ball", "RPN_Detector_KNN"):
        net = getattr(networks, cls)(opt)
        out[cls] = {k: list(v.shape) for k, v in net.state_dict().items()}
        out[cls + "/layer_module"] = type(net.mlp1).__module__
        if mode == "ours":                                   # the fused classes of usip_amd.networks: same keys, shapes
            from usip_amd import networks as fused
            out[cls + "/fused"] = {k: list(v.shape) for k, v in getattr(fused, cls)(opt).state_dict().items()}
    print(json.dumps(out))
""") % (ROOT, REF)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_reference_networks_builds_unchanged_on_our_modules():
    import json
    res = {}
    for mode in ("ours", "theirs"):
        p = subprocess.run([sys.executable, "-c", SCRIPT, mode], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        res[mode] = json.loads(p.stdout.strip().splitlines()[-1])
    for cls in ("RPN_Detector", "RPN_DetectorLite", "RPN_Detector_Ball", "RPN_Detector_KNN"):
        assert res["ours"][cls] == res["theirs"][cls], cls           # same keys, same shapes
        assert res["ours"][cls + "/fused"] == res["theirs"][cls], cls    # usip_amd.networks' own classes too
        assert res["ours"][cls + "/layer_module"] == "usip_amd.layers"
        assert res["theirs"][cls + "/layer_module"] == "models.layers"
