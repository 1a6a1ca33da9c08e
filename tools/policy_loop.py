"""The actor forward in a loop for ~8 s (for tools/clock_probe.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaac_rover_amd import _lib
from isaac_rover_amd.learning.model import HeightmapNet
e, ns, nd = 65536, 634, 1112
eng = _lib.Engine(e, device=0)
obs = torch.rand(e, 4 + ns + nd, device="cuda")
net = HeightmapNet(eng, 4 + ns + nd, ns, nd, 2, "tanh")
t0 = time.time(); n = 0
while time.time() - t0 < 8.0:
    for _ in range(200): net.compute(obs, fused=True)
    torch.cuda.synchronize(); n += 200
dt = time.time() - t0
print(f"{n} forwards in {dt:.2f} s: {dt / n * 1e3:.4f} ms each")
