import sys, time; sys.path.insert(0, '.')
import torch, numpy as np
from isaac_rover_amd import _lib, synth
verts, tris, _ = synth.grid_mesh(601, seed=0)
eng = _lib.Engine(8, device=0)
for _ in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    idx = eng.build_knn_map(verts, tris, 600, 600, 0.1, 200)
    torch.cuda.synchronize(); print("build 600x600 K=200 over 720k triangles: %.3f s" % (time.perf_counter() - t))
ref = synth.knn_map_grid(600, 601, 200, torch.device("cuda"))
same = (idx.cpu() == ref.cpu())
print("identical to the integer-exact grid ranking:", float(same.float().mean()))
a = np.sort(idx.cpu().numpy()[100:110, 100:110], axis=-1); b = np.sort(ref.cpu().numpy()[100:110, 100:110], axis=-1)
print("same SETS on a 10x10 patch:", float((a == b).mean()))
