#!/usr/bin/env python3
"""train.py-shaped rollout on the MI355X rover step path (no Isaac Sim, no learner).

Mirrors the loop of the reference's `omniisaacgymenvs/train.py:57-125` (env = load...("Rover"); trainer loop:
actions = agent.act(obs); obs, rew, done, info = env.step(actions)) with a random policy standing in for the skrl PPO
agent and `vec_env.KinematicSim` standing in for PhysX.

    python examples/rollout.py --envs 4096 --steps 200 [--assets /path/to/omniisaacgymenvs]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from isaac_rover_amd import assets, config, synth, vec_env  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--assets", default="", help="directory holding the reference's tasks/utils/terrain/... files")
    ap.add_argument("--native-rays", action="store_true", help="the reference's 1634-point distribution (1750-float obs)")
    args = ap.parse_args()

    scene = assets.load_reference_assets(args.assets) if args.assets else synth.make_scene(n_cells=600, k=200, n_stones=128, device="cuda")
    cfg = config.SimConfig(num_envs=args.envs, device="cuda:0")
    env = vec_env.VecEnv(headless=True)
    extent = scene.terrain.map_indices.shape[0] * scene.terrain.cell_size
    g = torch.Generator().manual_seed(0)
    spawn = torch.zeros(args.envs, 3)
    spawn[:, 0:2] = 0.15 * extent + 0.7 * extent * torch.rand(args.envs, 2, generator=g)
    from isaac_rover_amd.tasks.rover import RoverTask
    task = RoverTask("Rover", cfg, env, scene=scene, distribution=None if args.native_rays else synth.ray_distribution("37"))
    env.set_task(task, sim_params={"dt": 0.05}, spawn_positions=spawn)          # utils/task_util.py:45
    obs = env.reset()
    print(f"obs {tuple(obs.shape)}  actions {task.num_actions}  device {task.device}")
    ret = torch.zeros(args.envs, device=task.device)
    episodes = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        actions = 2 * torch.rand(args.envs, 2, device=task.device) - 1          # agent.act(obs) goes here
        obs, rew, done, info = env.step(actions)
        ret += rew
        episodes += int(done.sum())                                             # host sync, like a logger would do
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{args.steps} steps x {args.envs} envs in {dt:.2f} s = {args.steps * args.envs / dt:,.0f} env-steps/s "
          f"(incl. random policy + toy pose feeder); episodes finished: {episodes}; mean return {float(ret.mean()):.4f}")
    env.close()


if __name__ == "__main__":
    main()
