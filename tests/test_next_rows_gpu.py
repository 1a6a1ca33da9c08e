"""GPU (-m gpu): the "next" rows f-1 / f-2 / f-3 / f-4 against golden vectors captured from the reference's OWN code
(oracle/gen_golden.py: RoverTask.reset_idx, rover_utils._get_knn_triangles, learning/model.py's networks) — not against
restatements written for the test."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

TOL_QUAT = 1e-6          # reset orientation: f32 sin/cos of d/2 vs scipy's float64 quaternion cast to f32
TOL_NET_ABS, TOL_NET_REL = 2e-5, 2e-4      # f32 MFMA accumulation order vs the reference's nn.Linear on CPU (MKL sgemm)


# --------------------------------------------------------------------------------------------------------------- f-1
def test_pre_physics_step_matches_reference():
    """`rover_pre_physics_step` (one kernel) against what the reference's own `RoverTask.pre_physics_step` (rover.py:338-343,379-414,
    run unbound on its own RoverView by oracle/ref_harness.py) left behind: `rover_rot`, both Memory trackers, and the
    (positions, joint_indices) / (velocities, joint_indices) it handed to `set_joint_position_targets` / `set_joint_velocity_targets`
    (robots/articulations/views/rover_view.py:45-46) — including straight-line / turn-on-the-spot / 0-0 / NaN / inf actions.  The task's
    host-side control flow (device_reset=False: the reference's method split) must produce the same."""
    from isaac_rover_amd import _lib
    fx = load_golden("pre_physics_step")
    e = fx["in_actions"].shape[0]
    eng = _lib.Engine(e, device=0)
    dev = eng.device
    actions = torch.from_numpy(fx["in_actions"]).to(dev)
    quat = torch.from_numpy(fx["in_quat"]).to(dev)
    lin = torch.from_numpy(fx["in_lin_hist"]).to(dev).clone()
    ang = torch.from_numpy(fx["in_ang_hist"]).to(dev).clone()
    euler = torch.full((e, 3), 7.0, device=dev)
    jpt = torch.full((e, 13), -5.0, device=dev)
    jvt = torch.full((e, 13), -6.0, device=dev)
    ann = torch.from_numpy(fx["in_actions_nn"]).to(dev).contiguous().clone()
    eng.pre_physics_step(actions, quat, lin, ang, euler_pre=euler, pos_targets13=jpt, vel_targets13=jvt, actions_nn=ann)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(ann.cpu().numpy(), fx["out_actions_nn"])                           # :366 (bitwise, NaN included)
    np.testing.assert_allclose(euler.cpu().numpy(), fx["out_rover_rot"], rtol=1e-5, atol=1e-5)       # :343
    np.testing.assert_array_equal(lin.cpu().numpy(), fx["out_lin_tracker"])                         # :379-380 (bitwise, NaN included)
    np.testing.assert_array_equal(ang.cpu().numpy(), fx["out_ang_tracker"])
    pi, vi = fx["out_pos_joint_indices"], fx["out_vel_joint_indices"]
    np.testing.assert_array_equal(pi, [6, 8, 4, 7])
    np.testing.assert_array_equal(vi, [10, 5, 12, 9, 3, 11])
    jp, jv = jpt.cpu().numpy(), jvt.cpu().numpy()
    np.testing.assert_allclose(jp[:, pi], fx["out_positions"], rtol=1e-5, atol=1e-5, equal_nan=True)   # :400-412
    np.testing.assert_allclose(jv[:, vi], fx["out_velocities"], rtol=1e-5, atol=1e-4, equal_nan=True)  # :404-414
    assert np.isnan(fx["out_velocities"]).any() and np.isinf(fx["out_velocities"]).any()                # the edge rows are in
    rest_p = np.setdiff1d(np.arange(13), pi)
    rest_v = np.setdiff1d(np.arange(13), vi)
    assert (jp[:, rest_p] == -5.0).all() and (jv[:, rest_v] == -6.0).all()                            # other joints' targets untouched
    eng.close()


def test_task_pre_physics_step_matches_reference():
    """The drop-in `RoverTask.pre_physics_step` in both control flows against the same reference capture (`actions_nn` too)."""
    from isaac_rover_amd import synth
    from isaac_rover_amd.config import SimConfig
    from isaac_rover_amd.tasks.rover import RoverTask
    from isaac_rover_amd.vec_env import VecEnv
    fx = load_golden("pre_physics_step")
    e = fx["in_actions"].shape[0]
    scene = synth.make_scene(n_cells=128, k=16, n_stones=10)
    for device_reset in (True, False):
        env = VecEnv(headless=True)
        task = RoverTask("Rover", SimConfig(num_envs=e, device="cuda:0"), env, scene=scene, distribution=synth.ray_distribution("9"),
                         device_reset=device_reset)
        env.set_task(task, sim_params={"dt": 0.05})
        dev = "cuda:0"
        task._rover._quat.copy_(torch.from_numpy(fx["in_quat"]).to(dev))
        task.linear_velocity.tracker.copy_(torch.from_numpy(fx["in_lin_hist"]).to(dev).reshape(task.linear_velocity.tracker.shape))
        task.angular_velocity.tracker.copy_(torch.from_numpy(fx["in_ang_hist"]).to(dev).reshape(task.angular_velocity.tracker.shape))
        task.actions_nn = torch.from_numpy(fx["in_actions_nn"]).to(dev)
        task.reset_buf.zero_()
        task._compaction_fresh = False
        task.global_step = 20
        task.pre_physics_step(torch.from_numpy(fx["in_actions"]).to(dev))
        torch.cuda.synchronize()
        np.testing.assert_allclose(task.rover_rot.cpu().numpy(), fx["out_rover_rot"], rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(task.linear_velocity.tracker.reshape(e, 3).cpu().numpy(), fx["out_lin_tracker"])
        np.testing.assert_array_equal(task.angular_velocity.tracker.reshape(e, 3).cpu().numpy(), fx["out_ang_tracker"])
        np.testing.assert_array_equal(task.actions_nn.cpu().numpy(), fx["out_actions_nn"])
        np.testing.assert_array_equal(task._rover.actuated_pos_indices, fx["out_pos_joint_indices"])
        np.testing.assert_array_equal(task._rover.actuated_vel_indices, fx["out_vel_joint_indices"])
        jp = task._rover._joint_pos_targets.cpu().numpy()
        jv = task._rover._joint_vel_targets.cpu().numpy()
        np.testing.assert_allclose(jp[:, fx["out_pos_joint_indices"]], fx["out_positions"], rtol=1e-5, atol=1e-5, equal_nan=True)
        np.testing.assert_allclose(jv[:, fx["out_vel_joint_indices"]], fx["out_velocities"], rtol=1e-5, atol=1e-4, equal_nan=True)
        env.close()


# --------------------------------------------------------------------------------------------------------------- f-2
def _reset_state(fx, dev):
    e = fx["initial_pos"].shape[0]
    g = torch.Generator().manual_seed(3)
    return dict(initial=torch.from_numpy(fx["initial_pos"]).to(dev), pos=torch.randn(e, 3, generator=g).to(dev),
                quat=torch.randn(e, 4, generator=g).to(dev), jp=torch.randn(e, 13, generator=g).to(dev),
                jv=torch.randn(e, 13, generator=g).to(dev), base=torch.from_numpy(fx["base_pos_in"]).to(dev),
                reset=torch.from_numpy(fx["reset_buf_in"]).to(dev), progress=torch.from_numpy(fx["progress_buf_in"]).to(dev))


def _check_reset(fx, st, before):
    ids = fx["env_ids"]
    others = np.setdiff1d(np.arange(fx["initial_pos"].shape[0]), ids)
    got = {k: v.cpu().numpy() for k, v in st.items()}
    # what reset_idx handed to set_world_poses (rover.py:449): base_pos[ids] and the scipy (x,y,z,w) quaternion read as (w,x,y,z)
    np.testing.assert_array_equal(got["pos"][ids], fx["out_pose_pos"])
    np.testing.assert_allclose(got["quat"][ids], fx["out_pose_quat"], atol=TOL_QUAT, rtol=0)
    np.testing.assert_array_equal(fx["out_pose_indices"], ids)
    # set_joint_positions / set_joint_velocities (:439-440): zeros for the listed envs
    np.testing.assert_array_equal(got["jp"][ids], fx["out_joint_pos"])
    np.testing.assert_array_equal(got["jv"][ids], fx["out_joint_vel"])
    # self.base_pos (:442), reset_buf / progress_buf (:452-453): whole arrays, untouched envs included
    np.testing.assert_array_equal(got["base"], fx["out_base_pos"])
    np.testing.assert_array_equal(got["reset"], fx["out_reset_buf"])
    np.testing.assert_array_equal(got["progress"], fx["out_progress_buf"])
    assert got["reset"][3] == 1                                    # flagged but not listed: stays flagged
    for k in ("pos", "quat", "jp", "jv"):
        np.testing.assert_array_equal(got[k][others], before[k][others], err_msg=k)


@pytest.mark.parametrize("count_on_device", [False, True])
def test_reset_envs_matches_reference_reset_idx(count_on_device):
    """rover_reset_envs (reset_envs_kernel) vs RoverTask.reset_idx of the reference (rover.py:416-453) run unbound with
    random.randint fed the same degrees: poses, the scipy-order yaw quirk (:429-431,:449), joint zeroing, bookkeeping."""
    from isaac_rover_amd import _lib
    fx = load_golden("reset_idx")
    e = fx["initial_pos"].shape[0]
    eng = _lib.Engine(e, device=0)
    dev = eng.device
    st = _reset_state(fx, dev)
    before = {k: v.cpu().numpy().copy() for k, v in st.items()}
    n = len(fx["env_ids"])
    ids = torch.zeros(e, dtype=torch.int64, device=dev)
    ids[:n] = torch.from_numpy(fx["env_ids"]).to(dev)
    yaw = torch.zeros(e, dtype=torch.int32, device=dev)
    yaw[:n] = torch.from_numpy(fx["degrees"]).to(dev)
    kw = dict(n_reset_dev=torch.tensor([n], dtype=torch.int32, device=dev)) if count_on_device else dict(n_reset_host=n)
    eng.reset_envs(ids, st["initial"], st["pos"], st["quat"], st["reset"], st["progress"], joint_pos13=st["jp"],
                   joint_vel13=st["jv"], base_pos3=st["base"], yaw_deg=yaw, **kw)
    torch.cuda.synchronize()
    _check_reset(fx, st, before)
    eng.close()


def test_task_reset_idx_matches_reference():
    """The host-side RoverTask.reset_idx of this package (the reference-shaped control flow, device_reset=False) against the
    same fixture."""
    from isaac_rover_amd import synth
    from isaac_rover_amd.config import SimConfig
    from isaac_rover_amd.tasks.rover import RoverTask
    from isaac_rover_amd.vec_env import VecEnv
    fx = load_golden("reset_idx")
    e = fx["initial_pos"].shape[0]
    scene = synth.make_scene(n_cells=128, k=16, n_stones=10)
    env = VecEnv(headless=True)
    task = RoverTask("Rover", SimConfig(num_envs=e, device="cuda:0"), env, scene=scene, distribution=synth.ray_distribution("9"),
                     device_reset=False)
    env.set_task(task, sim_params={"dt": 0.05})
    dev = "cuda:0"
    st = _reset_state(fx, dev)
    rv = task._rover
    rv._pos.copy_(st["pos"]); rv._quat.copy_(st["quat"]); rv._joint_pos.copy_(st["jp"]); rv._joint_vel.copy_(st["jv"])
    task.initial_pos = st["initial"]
    task.base_pos = st["base"]
    task.reset_buf.copy_(st["reset"]); task.progress_buf.copy_(st["progress"])
    before = {k: v.cpu().numpy().copy() for k, v in st.items()}
    task.reset_idx(torch.from_numpy(fx["env_ids"]).to(dev), yaw_deg=torch.from_numpy(fx["degrees"]).to(dev))
    torch.cuda.synchronize()
    got = dict(pos=rv._pos, quat=rv._quat, jp=rv._joint_pos, jv=rv._joint_vel, base=task.base_pos, reset=task.reset_buf,
               progress=task.progress_buf, initial=st["initial"])
    _check_reset(fx, got, before)
    env.close()


# --------------------------------------------------------------------------------------------------------------- f-3
def _fp16_distances(verts, tris, cell_x, cell_y, i):
    """The reference's distances for map row i (rover_utils.py:68-72,99-100): fp16 centroids (float64 mean -> float -> half),
    fp16(c - p) per axis, norm in f32 rounded to fp16.  Returns [Y, T] fp16."""
    v = verts.astype(np.float64)
    cx = ((v[tris[:, 0], 0] + v[tris[:, 1], 0] + v[tris[:, 2], 0]) / 3).astype(np.float32).astype(np.float16).astype(np.float32)
    cy = ((v[tris[:, 0], 1] + v[tris[:, 1], 1] + v[tris[:, 2], 1]) / 3).astype(np.float32).astype(np.float16).astype(np.float32)
    dx = (cx - np.float32(cell_x[i])).astype(np.float16).astype(np.float32)
    dy = (cy[None, :] - cell_y.astype(np.float32)[:, None]).astype(np.float16).astype(np.float32)
    return np.sqrt(dx[None, :] * dx[None, :] + dy * dy).astype(np.float16)


@pytest.mark.parametrize("kind", ["grid10m", "soup50m"])
def test_knn_builder_matches_the_reference_builder(kind):
    """rover_build_knn_map_ref vs maps built by the reference's own _get_knn_triangles (tasks/utils/rover_utils.py:52-118;
    only o3d.io.read_triangle_mesh replaced).  Per cell: the multiset of fp16 distances of the K chosen triangles is identical,
    every triangle strictly nearer than the K-th distance is in both lists, nothing farther is in either — i.e. the lists may
    differ only by swaps among triangles tied with the K-th fp16 distance (torch.topk leaves that order unspecified).  The
    on-disk vertices / triangles tensors (:113-118) are byte-identical."""
    from isaac_rover_amd import _lib, assets, synth
    fx = load_golden("knn_" + kind)
    verts, tris, kw = synth.knn_test_mesh(kind)
    h = hashlib.sha256(); h.update(np.ascontiguousarray(verts).tobytes()); h.update(np.ascontiguousarray(tris).tobytes())
    assert h.hexdigest() == str(fx["mesh_digest"]), "test mesh drifted from the one the reference built its map on"
    x, y, res, k = int(fx["res_x"]), int(fx["res_y"]), float(fx["res"]), int(fx["n_triangles"])
    eng = _lib.Engine(8, device=0)
    m = assets.build_knn_map(eng, verts, tris, n_cells=x, res=res, k=k, ranking="reference_fp16", cell_x_f16=fx["cell_x_f16"],
                             cell_y_f16=fx["cell_y_f16"])
    got = m.map_indices.numpy().astype(np.int64)                          # [X, Y, K]
    ref = np.transpose(fx["map_indices_kxy"].astype(np.int64), (1, 2, 0))  # saved as [K, X, Y] (rover_utils.py:108-113)
    assert hashlib.sha256(m.vertices.numpy().tobytes()).hexdigest() == str(fx["vertices_f16_digest"])
    assert hashlib.sha256(m.triangles.numpy().tobytes()).hexdigest() == str(fx["triangles_digest"])
    n_swapped = 0
    ids = np.arange(tris.shape[0])
    for i in range(x):
        d = _fp16_distances(verts, tris, fx["cell_x_f16"], fx["cell_y_f16"], i)      # [Y, T]
        dg = np.take_along_axis(d, got[i], axis=1).astype(np.float32)                 # [Y, K]
        dr = np.take_along_axis(d, ref[i], axis=1).astype(np.float32)
        np.testing.assert_array_equal(np.sort(dg, axis=1), np.sort(dr, axis=1), err_msg=f"row {i}: fp16 distance multisets")
        assert (np.diff(dg, axis=1) >= 0).all(), f"row {i}: list not ascending"
        kth = dg[:, -1:]
        nearer = d.astype(np.float32) < kth                                           # must be in both lists
        for lst in (got[i], ref[i]):
            member = np.zeros_like(nearer)
            np.put_along_axis(member, lst, True, axis=1)
            assert not (nearer & ~member).any(), f"row {i}: a strictly nearer triangle is missing"
            assert (member.sum(axis=1) == k).all(), f"row {i}: duplicate triangle ids"
        # own tie rule: among equal fp16 distances ascending triangle id
        tie = np.diff(dg, axis=1) == 0
        assert (np.diff(got[i], axis=1)[tie] > 0).all()
        n_swapped += int((np.sort(got[i], axis=1) != np.sort(ref[i], axis=1)).any(axis=1).sum())
    print(f"{kind}: {n_swapped} of {x * y} cells differ from the reference by swaps inside the K-th tie class")
    # default coordinate tables = ATen's CUDA arange, fp16(float(i) * res): same call without tables must equal explicit ones
    tab = (np.arange(x, dtype=np.float32) * np.float32(res)).astype(np.float16)
    a = eng.build_knn_map(verts, tris, x, y, res, k, ranking="reference_fp16").cpu().numpy()
    b = eng.build_knn_map(verts, tris, x, y, res, k, ranking="reference_fp16", cell_x_f16=tab, cell_y_f16=tab).cpu().numpy()
    np.testing.assert_array_equal(a, b)
    eng.close()


# --------------------------------------------------------------------------------------------------------------- f-4
def _load(net, fx, tag):
    sd = {k[len(tag) + 1:]: torch.from_numpy(v.astype(np.float32)) for k, v in fx.items() if k.startswith(tag + ".")}
    assert set(sd) == set(net.state_dict()), "parameter names must be the reference's (state_dict interop)"
    net.load_state_dict(sd)


@pytest.mark.parametrize("name", ["policy_native", "policy_p37"])
def test_policy_forward_matches_reference_modules(name):
    """HeightmapNet (rover_linear_forward, f32 MFMA) with a state_dict saved from the reference's StochasticActorHeightmap /
    DeterministicHeightmap (learning/model.py:152-241, built by the reference's constructors with skrl's base class stubbed)
    against the outputs those modules computed: both encoders (:122-150), the actor mean (Tanh head, :185-195) and the
    critic value (:231-241).  policy_p37 has an EMPTY dense slice: Encoder(0, ...) = activation(bias)."""
    from isaac_rover_amd import _lib
    from isaac_rover_amd.learning.model import HeightmapNet
    fx = load_golden(name)
    nobs, ns, nd = int(fx["num_observations"]), int(fx["num_sparse"]), int(fx["num_dense"])
    eng = _lib.Engine(8, device=0)
    x = torch.from_numpy(fx["states"].astype(np.float32)).cuda()
    actor = HeightmapNet(eng, nobs, ns, nd, 2, "tanh")
    _load(actor, fx, "actor")
    # the fused chain kernels (one launch per encoder, one for MLP + head) against the same reference outputs
    fused = actor.compute(x, fused=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(fused.cpu().numpy(), fx["out_actor"], atol=TOL_NET_ABS, rtol=TOL_NET_REL)
    p0 = nobs - ns - nd
    cat_f = actor._bufs["cat"].cpu().numpy()
    np.testing.assert_array_equal(cat_f[:, :p0], fx["states"].astype(np.float32)[:, :p0])      # (copied by the encoder pair's second launch)
    np.testing.assert_allclose(cat_f[:, p0:p0 + 60], fx["out_actor_encoder0"], atol=TOL_NET_ABS, rtol=TOL_NET_REL)
    np.testing.assert_allclose(cat_f[:, p0 + 60:p0 + 120], fx["out_actor_encoder1"], atol=TOL_NET_ABS, rtol=TOL_NET_REL)
    out = actor.compute(x, fused=False)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), fx["out_actor"], atol=TOL_NET_ABS, rtol=TOL_NET_REL)
    cat = actor._bufs["cat"].cpu().numpy()
    p = nobs - ns - nd
    np.testing.assert_array_equal(cat[:, :p], fx["states"].astype(np.float32)[:, :p])
    np.testing.assert_allclose(cat[:, p:p + 60], fx["out_actor_encoder0"], atol=TOL_NET_ABS, rtol=TOL_NET_REL)
    np.testing.assert_allclose(cat[:, p + 60:p + 120], fx["out_actor_encoder1"], atol=TOL_NET_ABS, rtol=TOL_NET_REL)
    np.testing.assert_array_equal(actor.log_std_parameter.cpu().numpy(), fx["out_log_std"])
    if "out_critic" in fx:
        critic = HeightmapNet(eng, nobs, ns, nd, 1, None)
        _load(critic, fx, "critic")
        for fused_mode in (True, False):
            v = critic.compute(x, fused=fused_mode)
            torch.cuda.synchronize()
            np.testing.assert_allclose(v.cpu().numpy(), fx["out_critic"], atol=TOL_NET_ABS, rtol=TOL_NET_REL)
    eng.close()


@pytest.mark.parametrize("rows", [1, 17, 512, 4096, 20000, 24000])
def test_encoder_pair_equals_two_chain_calls(rows):
    """rover_mlp_chain_pair_forward (both encoders + the proprioception copy in one launch per stage below 20 480 rows, one chain
    after the other above) against two rover_mlp_chain_forward calls and a tensor copy: bit-identical."""
    from isaac_rover_amd import _lib
    from isaac_rover_amd.learning.model import HeightmapNet
    eng = _lib.Engine(8, device=0)
    torch.manual_seed(rows)
    net = HeightmapNet(eng, 1750, 634, 1112, 2, "tanh")
    x = torch.rand(rows, 1750, device="cuda") * 2 - 1
    p, ef = 4, 60
    want = torch.full((rows, p + 2 * ef), float("nan"), device="cuda")
    want[:, :p] = x[:, :p]
    eng.chain_forward(x[:, p:p + 634], net.encoder0, want[:, p:p + ef])
    eng.chain_forward(x[:, p + 634:], net.encoder1, want[:, p + ef:])
    got = torch.full((rows, p + 2 * ef), float("nan"), device="cuda")
    eng.chain_pair_forward(x[:, p:p + 634], net.encoder0, got[:, p:p + ef], x[:, p + 634:], net.encoder1, got[:, p + ef:],
                           copy_src=x, copy_dst=got, copy_cols=p)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # without the copy the first columns stay untouched
    got2 = torch.full((rows, p + 2 * ef), 7.0, device="cuda")
    eng.chain_pair_forward(x[:, p:p + 634], net.encoder0, got2[:, p:p + ef], x[:, p + 634:], net.encoder1, got2[:, p + ef:])
    torch.cuda.synchronize()
    assert torch.equal(got2[:, p:], want[:, p:]) and bool((got2[:, :p] == 7.0).all())
    # the whole forward through the pair equals the layer-by-layer path within the net tolerance
    a, b = net.compute(x, fused=True).clone(), net.compute(x, fused=False).clone()
    torch.cuda.synchronize()
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=TOL_NET_ABS, rtol=TOL_NET_REL)
    eng.close()
