"""Time the actor forward (f-4) on the GPU box, one launch per layer vs the fused chain kernels; HIP-event time per forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaac_rover_amd import _lib
from isaac_rover_amd.learning.model import HeightmapNet
for e, ns, nd in ((4096, 634, 1112), (65536, 634, 1112), (65536, 37, 1)):
    eng = _lib.Engine(e, device=0)
    w = 4 + ns + nd
    obs = torch.rand(e, w, device="cuda")
    net = HeightmapNet(eng, w, ns, nd, 2, "tanh")
    for fused in (False, True):
        for _ in range(3): net.compute(obs, fused=fused)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t = time.perf_counter(); a.record()
        for _ in range(20): net.compute(obs, fused=fused)
        b.record(); th = (time.perf_counter() - t) / 20
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
        gpu = a.elapsed_time(b) / 20
        flops = 2 * e * sum(l.weight.numel() for l in net.encoder0 + net.encoder1 + net.network)
        print(f"E={e} obs={w} fused={fused}: gpu {gpu:.3f} ms ({flops / gpu / 1e9:.1f} TFLOP/s f32), wall {dt*1e3:.3f} ms, host enqueue {th*1e3:.3f} ms")
    eng.close()
