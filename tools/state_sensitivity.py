"""How the step's time depends on the state distribution (run on the GPU box): Engine.step at configs[2] on SURVEY 8(d)'s random states and on
variants of them (level rovers, zero joints, spawn-like poses)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isaac_rover_amd import _lib, synth

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
scene = synth.make_scene(n_cells=600, k=200, n_stones=1024, device="cuda")
distn = synth.ray_distribution("37")
eng = _lib.Engine(E, device=0)
eng.set_scene(scene, distn)
dev = eng.device
W = eng.num_observations


def run(label, mutate):
    batches = []
    for b in range(4):
        st = synth.make_states(E, 60.0, seed=b)
        mutate(st)
        batches.append({k: v.to(dev) for k, v in st.items()})
    obs = torch.zeros(E, W, device=dev)
    i64 = torch.int64
    sout = eng.make_out(obs, rew=torch.zeros(E, device=dev), reset=torch.ones(E, dtype=i64, device=dev),
                        rock_collision=torch.zeros(E, dtype=i64, device=dev), reset_ids=torch.zeros(E, dtype=i64, device=dev),
                        n_reset=torch.zeros(1, dtype=torch.int32, device=dev))
    sins = [eng.make_in(b["pos"], b["quat"], b["joints"], b["target"], b["lin_hist"], b["ang_hist"], b["euler_pre"], b["progress"]) for b in batches]
    eng.set_profiling(True)
    t0 = time.perf_counter()
    i = 0
    while time.perf_counter() - t0 < 0.4:
        eng.step(sins[i % 4], sout, True, True)
        i += 1
        if i % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    eng.set_profiling(True, every=4)
    n = 200
    t0 = time.perf_counter()
    for i in range(n):
        eng.step(sins[i % 4], sout, True, True)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    p = eng.get_profile()
    ci = eng.cull_info()
    print(f"{label:34s} {1e3 * t / n:.4f} ms/step, ray cast {p.raycast_ms / max(p.launches, 1):.4f} ms, pairs/ray {ci['pairs_per_ray']:.2f}, "
          f"both tests {ci['rays_both_tests'] / max(ci['rays'], 1):.3f}, not scanned {ci['rays_not_scanned'] / max(ci['rays'], 1):.3f}, "
          f"items/ray {ci['lane_items'] / max(ci['rays'], 1):.2f}", flush=True)


def level(st):
    e = st["quat"].shape[0]
    g = torch.Generator().manual_seed(1)
    yaw = (2 * torch.rand(e, generator=g) - 1) * 3.14159
    st["quat"] = synth.quat_from_euler(torch.zeros(e), torch.zeros(e), yaw)


def small_tilt(st):
    e = st["quat"].shape[0]
    g = torch.Generator().manual_seed(1)
    yaw = (2 * torch.rand(e, generator=g) - 1) * 3.14159
    st["quat"] = synth.quat_from_euler(0.01 * torch.randn(e, generator=g), 0.01 * torch.randn(e, generator=g), yaw)


def zero_joints(st):
    st["joints"].zero_()


def both(st):
    level(st)
    zero_joints(st)


run("SURVEY 8(d) states", lambda st: None)
run("level rovers (roll = pitch = 0)", level)
run("tilt sigma 0.01", small_tilt)
run("zero joints", zero_joints)
run("level + zero joints", both)
run("SURVEY 8(d) states again", lambda st: None)
