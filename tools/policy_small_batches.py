import sys, torch, time
sys.path.insert(0, '.')
from isaac_rover_amd import _lib
from isaac_rover_amd.learning.model import HeightmapNet
for rows in (512, 2048, 4096, 8192, 16384):
    eng = _lib.Engine(8, device=0)
    net = HeightmapNet(eng, 1750, 634, 1112, 2, "tanh")
    x = torch.rand(rows, 1750, device="cuda")
    for fused in (False, True):
        for _ in range(20): net.compute(x, fused=fused)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200): net.compute(x, fused=fused)
        b.record(); torch.cuda.synchronize()
        print(rows, "fused" if fused else "per-layer", round(a.elapsed_time(b) / 200, 4), "ms")
    eng.close()
