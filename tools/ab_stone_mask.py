import os, sys, statistics
sys.path.insert(0, "/root/repo")
import torch
from isaac_rover_amd import _lib, synth
E = 65536
scene = synth.make_scene(n_cells=600, k=200, n_stones=1024, device="cuda")
distn = synth.ray_distribution("37")
eng = _lib.Engine(E, device=0)
eng.set_scene(scene, distn)
st = {k: v.cuda() for k, v in synth.make_states(E, 60.0, seed=0).items()}
sin = eng.make_in(st["pos"], st["quat"], st["joints"], st["target"], st["lin_hist"], st["ang_hist"], st["euler_pre"], st["progress"])
obs = torch.zeros(E, eng.num_observations, device="cuda")
z64 = lambda: torch.zeros(E, dtype=torch.int64, device="cuda")
keep = [torch.zeros(E, device="cuda"), z64(), z64(), z64(), torch.zeros(1, dtype=torch.int32, device="cuda"), z64()]
outs = [eng.make_out(obs, rew=keep[0], reset=keep[1], rock_collision=keep[2], reset_ids=keep[3], n_reset=keep[4]),
        eng.make_out(obs, rew=keep[0], reset=keep[1], rock_collision=keep[2], reset_ids=keep[3], n_reset=keep[4], stone_collision=keep[5], stone_margin=0.0)]
times = {0: [], 1: []}
for r in range(14):
    for i, so in enumerate(outs):
        eng.step(sin, so, compact=True)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            eng.step(sin, so, compact=True)
        b.record(); torch.cuda.synchronize()
        if r >= 2: times[i].append(a.elapsed_time(b) / 5)
for i in times:
    print("stone mask" if i else "no mask   ", f"median {statistics.median(times[i]):.4f} min {min(times[i]):.4f}")
print("mask ones:", int(keep[5].sum()))
