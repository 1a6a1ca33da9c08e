"""Time the actor forward (f-4) on the GPU box: native 1750-float obs and the 41-float bench obs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaac_rover_amd import _lib
from isaac_rover_amd.learning.model import HeightmapNet
for e, ns, nd in ((65536, 37, 0), (4096, 634, 1112), (65536, 634, 1112)):
    eng = _lib.Engine(e, device=0)
    w = 4 + ns + nd
    obs = torch.rand(e, w, device="cuda")
    net = HeightmapNet(eng, w, ns, max(nd, 0), 2, "tanh") if nd else None
    if net is None:      # no dense part: feed an empty dense slice through a 1-wide dummy so the class stays generic
        obs = torch.rand(e, w + 1, device="cuda"); net = HeightmapNet(eng, w + 1, ns, 1, 2, "tanh")
    for _ in range(3): net.compute(obs)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): net.compute(obs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    flops = 2 * e * sum(l.weight.numel() for l in net.encoder0 + net.encoder1 + net.network)
    print(f"E={e} obs={obs.shape[1]}: actor forward {dt*1e3:.3f} ms  ({flops/dt/1e12:.1f} TFLOP/s f32, {obs.numel()*4/dt/1e9:.0f} GB/s of obs)")
    eng.close()
