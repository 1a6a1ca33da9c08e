"""The actor forward only (65 536 rows x the native 1 750-float obs, fused chains), 30 times: the workload tools/collect_policy_profile.sh
profiles.  Per forward the launches are: encoder 0 chain (634 -> 80 -> 60), encoder 1 chain (1112 -> 80 -> 60), MLP chain (124 -> 256 -> 160 -> 128 -> 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaac_rover_amd import _lib
from isaac_rover_amd.learning.model import HeightmapNet

e, ns, nd = 65536, 634, 1112
eng = _lib.Engine(e, device=0)
obs = torch.rand(e, 4 + ns + nd, device="cuda")
net = HeightmapNet(eng, 4 + ns + nd, ns, nd, 2, "tanh")
for _ in range(30):
    net.compute(obs, fused=True)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    net.compute(obs, fused=True)
b.record()
torch.cuda.synchronize()
flops = 2 * e * sum(l.weight.numel() for l in net.encoder0 + net.encoder1 + net.network)
ms = a.elapsed_time(b) / 20
print(f'{{"actor_forward_ms": {ms:.4f}, "tflops_f32": {flops / ms / 1e9:.1f}, "frac_of_157_tflops": {flops / ms / 1e9 / 157.3:.3f}, "rows": {e}, "obs": {4 + ns + nd}}}')
eng.close()
