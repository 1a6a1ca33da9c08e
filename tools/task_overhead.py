"""What the drop-in task layer costs on top of Engine.step (run on the GPU box): RoverTask.pre_physics_step + post_physics_step on static
poses, eager and replayed from captured hipGraphs (graph=True), next to Engine.step alone on the same buffers.

    python tools/task_overhead.py [--profile]     -> one line per (envs, ray set, mode); --profile adds a cProfile of the eager 512-env case
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isaac_rover_amd import config, synth, vec_env
from isaac_rover_amd.tasks.rover import RoverTask


def make(E, rays, scene, graph):
    cfg = config.SimConfig(num_envs=E, device="cuda:0")
    distn = None if rays == "native" else synth.ray_distribution(rays)
    task = RoverTask("Rover", cfg, vec_env.VecEnv(headless=True), scene=scene, distribution=distn, graph=graph, stone_mask_margin=0.0)
    st = synth.make_states(E, 60.0, seed=7)
    task.set_up_scene(spawn_positions=st["pos"].cuda())
    task.post_reset()
    task.initial_pos.copy_(st["pos"].cuda())      # (not the shifted spawns: the synthetic scene's stone density piles those up at the map's edge)
    task._rover.feed(positions=task.initial_pos, orientations=st["quat"].cuda(), joint_positions=st["joints"].cuda())
    task.reset()
    return task


def measure(task, steps):
    E = task.num_envs
    g = torch.Generator().manual_seed(3)
    acts = [(2 * torch.rand(E, 2, generator=g) - 1).cuda() for _ in range(8)]

    def step(i):
        task.pre_physics_step(acts[i % 8])
        task.post_physics_step()

    for i in range(40):
        step(i)
    torch.cuda.synchronize()
    t_pre = time.perf_counter()
    i = 0
    while time.perf_counter() - t_pre < 0.4:          # clock ramp
        step(i)
        i += 1
        if i % 16 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    eng = task._engine
    t0 = time.perf_counter()
    for i in range(steps):
        eng.step(task._sin, task._sout, increment_progress=True, compact=True)
    torch.cuda.synchronize()
    t_eng = time.perf_counter() - t0
    return 1e3 * t_all / steps, 1e3 * t_enq / steps, 1e3 * t_eng / steps, step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    scene = synth.make_scene(n_cells=600, k=200, n_stones=1024, device="cuda")
    for E, rays, steps in ((512, "37", 1000), (512, "native", 400), (4096, "37", 600), (65536, "37", 100)):
        for graph in (False, True):
            task = make(E, rays, scene, graph)
            wall, enq, eng, step = measure(task, steps)
            print(f"E={E:6d} rays={rays:>6s} {'graph' if graph else 'eager'}: task wall {wall:.4f} ms/step, host enqueue {enq:.4f}, "
                  f"Engine.step alone {eng:.4f}, task / engine {wall / eng:.2f}", flush=True)
            if args.profile and not graph and E == 512 and rays == "37":
                import cProfile
                import pstats
                pr = cProfile.Profile()
                pr.enable()
                for i in range(300):
                    step(i)
                pr.disable()
                torch.cuda.synchronize()
                pstats.Stats(pr).sort_stats("cumulative").print_stats(16)
            task.close()
            del task
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
