"""Diagnostic: host-side enqueue cost of the step path (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from isaac_rover_amd import _lib, synth
from hip_helpers import make_engine

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scene = synth.make_scene(n_cells=128, k=64, n_stones=16)
distn = synth.ray_distribution("37")
st = {k: v.cuda() for k, v in synth.make_states(E, 12.8, seed=1).items()}
for variant in (1, 2):
    eng = make_engine(scene, distn, E, variant=variant)
    sin = eng.make_in(st["pos"], st["quat"], st["joints"], st["target"], st["lin_hist"], st["ang_hist"], st["euler_pre"], st["progress"])
    obs = torch.zeros(E, eng.num_observations, device="cuda")
    sout = eng.make_out(obs, rew=torch.zeros(E, device="cuda"), reset=torch.zeros(E, dtype=torch.int64, device="cuda"),
                        rock_collision=torch.zeros(E, dtype=torch.int64, device="cuda"),
                        reset_ids=torch.zeros(E, dtype=torch.int64, device="cuda"), n_reset=torch.zeros(1, dtype=torch.int32, device="cuda"))
    for prof in (False, True):
        eng.set_profiling(prof)
        for _ in range(5):
            eng.step(sin, sout, compact=True)
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            eng.step(sin, sout, compact=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"E={E} variant={variant} profiling={prof}: enqueue {1e6*(t1-t0)/n:.1f} us/step, total {1e6*(t2-t0)/n:.1f} us/step")
        eng.set_profiling(False)
    q = st["quat"]; out = torch.empty(E, 3, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(1000):
        eng.quat_to_euler(q, out=out)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"  single-launch API call: enqueue {1e6*(t1-t0)/1000:.1f} us, total {1e6*(t2-t0)/1000:.1f} us")
