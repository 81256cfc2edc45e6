"""Which weight operands does a steady-state step still split on its own (outside the recorded PlanesPlan)?  Prints a MISS
line with the call stack for each; none since round 3 (the 3 absmax_one launches per step a six-step rocprof trace shows
are the FIRST step's 18, averaged)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from usip_amd import ops, synth
from usip_amd.networks import DetectorOptions
from usip_amd.step import DetectorStep, batch_to_device
ops.set_matmul_mode("f32x2")
dev = torch.device("cuda:0")
st = DetectorStep("ball", DetectorOptions(surface_normal_len=4, node_knn_k_1=16), dev, with_optimizer=True)
batch = batch_to_device(synth.make_pair_batch(1234, 8, 16384, 512, 4, "slab"), dev)
st.step(batch); st.step(batch)
plan = st._planes_plan
print("plan entries", len(plan.entries))
orig = ops.weight_planes
def wp(At, a_offset, M, K, npl=3, P=0, nb=1):
    rows = int(ops._lib.lib().usip_mlp_x3p_tile_rows(int(M), int(P), int(nb)))
    key = (At.data_ptr(), At.shape[1], int(a_offset), int(M), int(K), int(npl), rows)
    if ops.PLANES_CACHE is not None and key not in ops.PLANES_CACHE:
        import traceback
        print("MISS", key[1:], "shape", tuple(At.shape), [l.strip() for l in traceback.format_stack(limit=6)[:-1]][-4:])
    return orig(At, a_offset, M, K, npl, P, nb)
ops.weight_planes = wp
st.step(batch)
