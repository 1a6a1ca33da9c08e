"""Per-kernel times of the small-batch actor forward (run under rocprofv3 --kernel-trace --stats): python tools/policy_small_profile.py ROWS"""
import sys, torch
sys.path.insert(0, '.')
from isaac_rover_amd import _lib
from isaac_rover_amd.learning.model import HeightmapNet
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = _lib.Engine(8, device=0)
net = HeightmapNet(eng, 1750, 634, 1112, 2, "tanh")
x = torch.rand(rows, 1750, device="cuda")
for _ in range(300):
    net.compute(x, fused=True)
torch.cuda.synchronize()
