"""Helpers for the -m gpu parity tests: drive the HIP path through the C ABI (isaac_rover_amd._lib)."""
import torch

from isaac_rover_amd import _lib

EXTRA_DT = {k: (torch.int64 if k == "collision_penalty" else torch.float32) for k in _lib.EXTRAS}


def make_engine(scene, distribution, num_envs, variant=3, run=None, **kw):
    """``variant`` defaults to the culled ray cast (the kernel full batches run; the library itself falls back to the binned
    kernel 2 for the as-shipped fp16 maths): at test sizes the library's auto choice would be the env-order kernel.
    None = leave the auto choice."""
    eng = _lib.Engine(num_envs, device=0, **kw)
    eng.set_scene(scene, distribution)
    if variant is not None:
        eng.set_option("raycast_variant", variant)
    if run is not None:
        eng.set_option("raycast_run", run)
    return eng


def hip_step(eng, st, compact=True, fused=True, stone_margin=None):
    """One post_physics_step on the GPU. Returns a dict of CPU numpy arrays named like the oracle's.
    ``stone_margin``: also request the stone_info occupancy mask (BASELINE configs[2]) -> out["stone_collision"]."""
    dev = eng.device
    e = eng.num_envs
    d = {k: v.to(dev).contiguous() for k, v in st.items()}
    progress = d["progress"].clone()
    sin = eng.make_in(d["pos"], d["quat"], d["joints"], d["target"], d["lin_hist"], d["ang_hist"], d["euler_pre"], progress)
    f, i64 = torch.float32, torch.int64
    obs = torch.zeros(e, eng.num_observations, device=dev)
    bufs = dict(rew=torch.zeros(e, device=dev), reset=torch.ones(e, dtype=i64, device=dev),
                rock_collision=torch.zeros(e, dtype=i64, device=dev),
                reset_ids=torch.full((e,), -1, dtype=i64, device=dev), n_reset=torch.zeros(1, dtype=torch.int32, device=dev),
                euler=torch.zeros(e, 3, device=dev), heading_diff=torch.zeros(e, device=dev),
                ray_dist=torch.zeros(e, eng.P, device=dev), wheel_dist=torch.zeros(e, 24, device=dev),
                body_dist=torch.zeros(e, 2, device=dev))
    extras = {k: torch.zeros(e, dtype=EXTRA_DT[k], device=dev) for k in _lib.EXTRAS}
    done_u8 = torch.full((e,), 7, dtype=torch.uint8, device=dev)
    stone = None if stone_margin is None else torch.full((e,), -1, dtype=i64, device=dev)
    sout = eng.make_out(obs, extras=extras, done_u8=done_u8, stone_collision=stone, stone_margin=stone_margin or 0.0, **bufs)
    if fused:
        eng.step(sin, sout, increment_progress=True, compact=compact)
    else:   # the reference's method split, rl_task.py:250-257
        progress += 1
        eng.get_observations(sin, sout)
        eng.calculate_metrics(sin, sout)
        eng.is_done(sin, sout)
        if compact:
            eng.compact_resets(bufs["reset"], bufs["reset_ids"], bufs["n_reset"])
    torch.cuda.synchronize()
    out = dict(euler=bufs["euler"], heading_diff=bufs["heading_diff"], ray_dist=bufs["ray_dist"],
               wheel_dist=bufs["wheel_dist"], body_dist=bufs["body_dist"], rock_collision=bufs["rock_collision"],
               obs_buf=obs, rew_buf=bufs["rew"], reset_buf=bufs["reset"], progress_buf=progress)
    for k in _lib.EXTRAS:
        out["extras_" + k] = extras[k]
    if fused or True:
        out["done_u8"] = done_u8
    if stone is not None:
        out["stone_collision"] = stone
    out = {k: v.cpu().numpy() for k, v in out.items()}
    n = int(bufs["n_reset"].item())
    out["reset_ids"] = bufs["reset_ids"][:n].cpu().numpy()
    return out
