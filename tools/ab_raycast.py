"""In-process A/B of ray-cast kernel options (run on the GPU box): interleaved rounds, median / min per arm.

    python tools/ab_raycast.py [envs] [rounds] -- arms are (variant, run, static_guard) tuples below
"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaac_rover_amd import _lib, synth

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
# keys starting with ENV_ set / clear an environment variable the launcher reads (experiments only)
ARMS = [dict(raycast_variant=3, raycast_run=0, bin_low_bits=10),
        dict(raycast_variant=3, raycast_run=0, bin_low_bits=9),
        dict(raycast_variant=3, raycast_run=0, bin_low_bits=8),
        dict(raycast_variant=3, raycast_run=0, bin_low_bits=11),
        ]
FULL_STEP = "--step" in sys.argv
scene = synth.make_scene(n_cells=600, k=200, n_stones=1024, device="cuda")
distn = synth.ray_distribution("37")
eng = _lib.Engine(E, device=0)
eng.set_scene(scene, distn)
st = {k: v.cuda() for k, v in synth.make_states(E, 60.0, seed=0).items()}
sin = eng.make_in(st["pos"], st["quat"], st["joints"], st["target"], st["lin_hist"], st["ang_hist"], st["euler_pre"], st["progress"])
obs = torch.zeros(E, eng.num_observations, device="cuda")
z64 = lambda: torch.zeros(E, dtype=torch.int64, device="cuda")
sout = eng.make_out(obs, rew=torch.zeros(E, device="cuda"), reset=z64(), rock_collision=z64(), reset_ids=z64(),
                    n_reset=torch.zeros(1, dtype=torch.int32, device="cuda"))
torch.cuda.synchronize()
times = {i: [] for i in range(len(ARMS))}
for r in range(ROUNDS + 2):
    for i, arm in enumerate(ARMS):
        for k, v in arm.items():
            if k.startswith("ENV_"):
                if v is None:
                    os.environ.pop(k[4:], None)
                else:
                    os.environ[k[4:]] = v
            else:
                eng.set_option(k, v)
        eng.step(sin, sout, compact=True)      # (re)build this arm's ray records / bins
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            if FULL_STEP:
                eng.step(sin, sout, compact=True)
            else:
                eng.replay_raycast()
        b.record()
        torch.cuda.synchronize()
        if r >= 2:
            times[i].append(a.elapsed_time(b) / 5)
for i, arm in enumerate(ARMS):
    t = times[i]
    print(f"{arm}: median {statistics.median(t):.4f} ms  min {min(t):.4f} ms  max {max(t):.4f}")
