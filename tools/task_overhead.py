"""Where a VecEnv.step() of the RoverTask mirror spends host time (run on the GPU box)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaac_rover_amd import config, synth, vec_env
from isaac_rover_amd.tasks.rover import RoverTask

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scene = synth.make_scene(n_cells=600, k=200, n_stones=128, device="cuda")
cfg = config.SimConfig(num_envs=E, device="cuda:0")
env = vec_env.VecEnv(headless=True)
g = torch.Generator().manual_seed(0)
spawn = torch.zeros(E, 3)
spawn[:, 0:2] = 9 + 42 * torch.rand(E, 2, generator=g)
task = RoverTask("Rover", cfg, env, scene=scene, distribution=synth.ray_distribution("37"))
env.set_task(task, sim_params={"dt": 0.05}, spawn_positions=spawn)
obs = env.reset()
acts = [2 * torch.rand(E, 2, device="cuda") - 1 for _ in range(8)]
for i in range(50):
    env.step(acts[i % 8])
torch.cuda.synchronize()
t = time.perf_counter()
N = 500
for i in range(N):
    env.step(acts[i % 8])
t_enq = time.perf_counter() - t
torch.cuda.synchronize()
t_all = time.perf_counter() - t
print(f"E={E}: host enqueue {1e3 * t_enq / N:.3f} ms/step, wall {1e3 * t_all / N:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    env.step(acts[i % 8])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
