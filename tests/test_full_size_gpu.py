"""GPU (-m gpu): BASELINE.json's full size (65 536 envs, 37 + 26 rays, 600 x 600 cells, K = 200): EVERY env of configs[2] and
configs[4] against the CPU oracle, in the fp32 parity mode and in the reference's as-shipped fp16 arithmetic, on the regular grid
scene and on the irregular (decimated-style) mesh — plus size-independent properties of the path itself."""
import numpy as np
import pytest
import torch

from conftest import assert_step_close

pytestmark = pytest.mark.gpu

E, CELLS, K = 65536, 600, 200


@pytest.fixture(scope="module")
def full():
    from isaac_rover_amd import synth
    scene = synth.make_scene(n_cells=CELLS, k=K, n_stones=1024, device="cuda")
    distn = synth.ray_distribution("37")
    st = synth.make_states(E, CELLS * 0.1, seed=7)
    return scene, distn, st


def _run(scene, distn, st, num_envs=None, **kw):
    from hip_helpers import hip_step, make_engine
    n = num_envs or st["pos"].shape[0]
    eng = make_engine(scene, distn, n, **kw)
    out = hip_step(eng, st)
    eng.close()
    return out


def test_full_size_invariants(full):
    scene, distn, st = full
    a = _run(scene, distn, st, variant=3)
    # (1) three independent ray-cast algorithms (culled: conservative sphere / normal test + exact candidates; binned:
    #     register-resident cells, every triangle evaluated; env-order streaming) agree bit for bit
    for other in (4, 2, 1):
        b = _run(scene, distn, st, variant=other)
        for k in a:
            np.testing.assert_array_equal(a[k], b[k], err_msg=f"{k} vs variant {other}")
    # (2) compaction = nonzero(reset_buf), ascending; the one-byte done flags that travel in the multi-GPU gather agree
    np.testing.assert_array_equal(a["reset_ids"], np.nonzero(a["reset_buf"])[0])
    np.testing.assert_array_equal(a["done_u8"], a["reset_buf"].astype(np.uint8))
    # (3) obs layout: [dist/9, heading/pi, lin, ang | ray distances / 2]; misses are exactly 11/2, hits bounded
    assert a["obs_buf"].shape == (E, 41)
    np.testing.assert_array_equal(a["obs_buf"][:, 4:], a["ray_dist"] / 2.0)
    assert (a["ray_dist"] <= 11.0).all() and np.isfinite(a["ray_dist"]).all()
    assert 0.5 < (a["ray_dist"] < 11.0).mean() <= 1.0
    assert (np.abs(a["obs_buf"][:, 1]) <= 1.0 + 1e-6).all()
    # (4) reward / done consistency with the mask: collided envs are done and carry the -300/3000 penalty
    coll = a["rock_collision"] == 1
    assert coll.any() and (~coll).any()
    assert (a["reset_buf"][coll] == 1).all()
    assert (a["extras_collision_penalty"][coll] == E).all() and (a["extras_collision_penalty"][~coll] == 0).all()
    assert (a["rew_buf"][coll] < -0.09).all()
    np.testing.assert_array_equal(a["progress_buf"], st["progress"].numpy() + 1)
    # (5) a sample of envs against the CPU oracle
    from oracle import oracle as orc
    idx = np.random.default_rng(0).choice(E, 384, replace=False)
    sub = {k: v[idx] for k, v in st.items()}
    t = orc.KnnMap(scene.terrain.map_indices, scene.terrain.triangles, scene.terrain.vertices)
    r = orc.KnnMap(scene.rocks.map_indices, scene.rocks.triangles, scene.rocks.vertices)
    want = orc.step(t, r, sub, *distn, num_envs_global=E)
    got = {k: (v[idx] if k != "reset_ids" else v) for k, v in a.items()}
    got.pop("reset_ids")
    assert_step_close(got, {"out_" + k: v for k, v in want.items()}, "full-size sample")


def test_full_size_permutation_and_sharding(full):
    """Envs are independent: permuting them permutes the outputs; two half-size shards reproduce the whole."""
    scene, distn, st = full
    a = _run(scene, distn, st)
    perm = torch.from_numpy(np.random.default_rng(1).permutation(E))
    p = _run(scene, distn, {k: v[perm] for k, v in st.items()})
    for k in ("obs_buf", "rew_buf", "reset_buf", "rock_collision", "ray_dist", "wheel_dist", "extras_pos_reward"):
        np.testing.assert_array_equal(p[k], a[k][perm.numpy()], err_msg=k)
    halves = [_run(scene, distn, {k: v[o:o + E // 2] for k, v in st.items()}, num_envs=E // 2, num_envs_global=E, env_offset=o)
              for o in (0, E // 2)]
    for k in a:
        np.testing.assert_array_equal(a[k], np.concatenate([h[k] for h in halves]), err_msg=k)


def test_full_size_step_is_repeatable(full):
    """Same state in, same bits out (the bin order produced by atomics must not leak into the results)."""
    scene, distn, st = full
    a = _run(scene, distn, st)
    b = _run(scene, distn, st)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_full_size_stone_mask_equals_clearance(full):
    """BASELINE configs[2] "stone_info collision mask" on ALL 65 536 envs over 1 024 stones: the mask the step emits through
    the stone-occupancy grid equals (rover_clearance(pos_xy) <= margin) — the exact clearance min over every stone
    (rover.py:536-539) — for the margins 0, 1.0 (:539) and 1.4 (:660); envs whose clearance is within 1e-4 of the margin are
    exempt.  A 512-env sample of the clearance itself is checked against the oracle."""
    from hip_helpers import hip_step, make_engine
    from isaac_rover_amd import synth
    from oracle import oracle as orc
    scene, distn, st = full
    eng = make_engine(scene, distn, E)
    xy = st["pos"][:, 0:2].contiguous().cuda()
    clear = eng.clearance(xy).cpu().numpy()
    info = synth.read_stone_info_array(scene.stone_info_raw)
    idx = np.random.default_rng(5).choice(E, 512, replace=False)
    np.testing.assert_allclose(clear[idx], orc.clearance(info, st["pos"][idx, 0:2]), rtol=0, atol=2e-5)
    base = None
    for margin in (0.0, 1.0, 1.4):
        out = hip_step(eng, st, stone_margin=margin)
        want = clear <= margin
        sure = np.abs(clear - margin) > 1e-4
        np.testing.assert_array_equal(out["stone_collision"][sure] == 1, want[sure], err_msg=f"margin {margin}")
        assert set(np.unique(out["stone_collision"])) <= {0, 1}
        assert 0.02 < want.mean() < 0.98, "the case must have both outcomes"
        if base is None:
            base = out
        for k in ("obs_buf", "rew_buf", "reset_buf", "rock_collision"):      # the mask never feeds reward or done
            np.testing.assert_array_equal(out[k], base[k], err_msg=k)
    eng.close()


def test_full_size_config4_dense_rays_and_goal_validation():
    """BASELINE configs[4]: 65 536 envs, 120-point dense heightmap, then the device-side reset + spawn-goal validation of the
    envs the step flagged (rover_reset_envs over 1 024 stones, count read on the device).  Properties: the three ray-cast
    algorithms agree bit for bit on all 9.6 M rays; every re-drawn goal has clearance > 1.0 (rover.py:539) and sits at radius
    8 (:578) from its spawn; reset / progress are zeroed exactly for the compacted ids; a 256-entry list with caller-supplied
    draws equals the sequential oracle."""
    from hip_helpers import hip_step, make_engine
    from isaac_rover_amd import synth
    from oracle import oracle as orc
    scene = synth.make_scene(n_cells=CELLS, k=K, n_stones=1024, device="cuda")
    distn = synth.ray_distribution("120")
    st = synth.make_states(E, CELLS * 0.1, seed=9)
    outs = {}
    for variant in (4, 3, 2, 1):
        eng = make_engine(scene, distn, E, variant=variant)
        assert eng.info().raycast_variant == variant
        outs[variant] = hip_step(eng, st)
        if variant != 3:
            eng.close()
        else:
            eng2 = eng
    a = outs[3]
    for k in a:
        np.testing.assert_array_equal(a[k], outs[1][k], err_msg=k)
        np.testing.assert_array_equal(a[k], outs[2][k], err_msg=k)
        np.testing.assert_array_equal(a[k], outs[4][k], err_msg=k)
    assert a["obs_buf"].shape == (E, 124)
    np.testing.assert_array_equal(a["obs_buf"][:, 4:], a["ray_dist"] / 2.0)
    idx = np.random.default_rng(2).choice(E, 128, replace=False)
    t = orc.KnnMap(scene.terrain.map_indices, scene.terrain.triangles, scene.terrain.vertices)
    r = orc.KnnMap(scene.rocks.map_indices, scene.rocks.triangles, scene.rocks.vertices)
    want = orc.step(t, r, {k: v[idx] for k, v in st.items()}, *distn, num_envs_global=E)
    got = {k: v[idx] for k, v in a.items() if k != "reset_ids"}
    assert_step_close(got, {"out_" + k: v for k, v in want.items()}, "cfg4 sample")

    # ---- reset + goal validation for the flagged envs, all on the device ----
    eng = eng2
    dev = eng.device
    n = len(a["reset_ids"])
    assert 1000 < n < E
    ids = torch.zeros(E, dtype=torch.int64, device=dev)
    ids[:n] = torch.from_numpy(a["reset_ids"]).to(dev)
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    initial = st["pos"].clone().cuda()
    pos = torch.zeros(E, 3, device=dev); quat = torch.zeros(E, 4, device=dev)
    target_before = st["target"].clone()
    target = st["target"].clone().cuda()
    reset = torch.from_numpy(a["reset_buf"]).to(dev); progress = torch.from_numpy(a["progress_buf"]).to(dev)
    used = torch.zeros(1, dtype=torch.int32, device=dev)
    eng.reset_envs(ids, initial, pos, quat, reset, progress, n_reset_dev=n_dev, target3=target, radius=8.0, seed=4242,
                   max_draws=256, n_draws_used=used)
    torch.cuda.synchronize()
    assert int(used.item()) >= 1, "every listed env found a clear goal within 256 draws"
    rid = a["reset_ids"]
    touched = np.union1d(rid, [0])                                    # env 0: the env_ids = mask*env_ids quirk (rover.py:540)
    tgt = target.cpu().numpy()
    c = eng.clearance(target[torch.from_numpy(touched).to(dev)][:, 0:2].contiguous()).cpu().numpy()
    assert (c > 1.0).all(), f"{(c <= 1.0).sum()} accepted goals violate the clearance"
    rad = np.linalg.norm(tgt[touched, 0:2] - st["pos"].numpy()[touched, 0:2], axis=1)
    np.testing.assert_allclose(rad, 8.0, atol=1e-4)
    np.testing.assert_array_equal(tgt[touched, 2], eng.sample_height(target[torch.from_numpy(touched).to(dev)][:, 0:2].contiguous()).cpu().numpy())
    others = np.setdiff1d(np.arange(E), touched)
    np.testing.assert_array_equal(tgt[others], target_before.numpy()[others])
    np.testing.assert_array_equal(pos.cpu().numpy()[rid], st["pos"].numpy()[rid])
    np.testing.assert_allclose(np.linalg.norm(quat.cpu().numpy()[rid], axis=1), 1.0, atol=1e-6)
    assert (reset.cpu().numpy()[rid] == 0).all() and (progress.cpu().numpy()[rid] == 0).all()
    np.testing.assert_array_equal(reset.cpu().numpy()[others], a["reset_buf"][others])
    np.testing.assert_array_equal(progress.cpu().numpy()[others], a["progress_buf"][others])
    # ---- 256 entries with supplied draws vs the sequential oracle (1 024 stones) ----
    info = synth.read_stone_info_array(scene.stone_info_raw)
    sub = rid[:: max(1, n // 256)][:256].astype(np.int64)
    draws = np.random.default_rng(8).random((128, len(sub))).astype(np.float32)
    want_t, want_used = orc.generate_goals(info, sub, st["pos"].numpy(), draws, radius=8.0)
    t2 = torch.zeros(E, 3, device=dev)
    used2 = torch.zeros(1, dtype=torch.int32, device=dev)
    eng.generate_goals(torch.from_numpy(sub).to(dev), initial, t2, radius=8.0, draws=torch.from_numpy(draws).to(dev), n_draws_used=used2)
    assert int(used2.item()) == want_used
    np.testing.assert_allclose(t2[:, 0:2].cpu().numpy(), want_t[:, 0:2], rtol=1e-6, atol=1e-5)
    eng.close()


# ---------------------------------------------------------------------------------------------------------------------
# ALL envs against the oracle (round 4).  The C oracle (OpenMP) does ~0.3 M env-steps/s on the GPU box's host cores in fp32
# and ~60 k in its fp16 mode, so a whole BASELINE-size batch costs it 0.2-3 s — no reason to sample.
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_maps(scene):
    from oracle import oracle as orc
    return (orc.KnnMap(scene.terrain.map_indices, scene.terrain.triangles, scene.terrain.vertices),
            orc.KnnMap(scene.rocks.map_indices, scene.rocks.triangles, scene.rocks.vertices))


def _rays_vs_oracle(eng, maps, half, label):
    """The ray phase on its own, free of the pose trigonometry: the device's OWN rays of the last step (origins and ray-record
    directions as prep_rays_kernel made them, `rover_export_rays`) through the oracle's per-ray arithmetic (ray_casting.py:34-59 + the
    cell lookup + min over K, `oracle.raycast_unit`) must give the device's distances bit for bit — every ray, both maps."""
    from oracle import oracle as orc
    src, dirs, cell, dist = (x.cpu().numpy() for x in eng.export_rays())
    t, r = maps
    want_rock = orc.raycast_unit(r, src[:, :26], dirs[:, :26], half=half).reshape(dist[:, :26].shape)
    want_terr = orc.raycast_unit(t, src[:, 26:], dirs[:, 26:], half=half).reshape(dist[:, 26:].shape)
    # equal as IEEE values (a distance of exactly zero may come out as +0 from one triangle and -0 from another: which of the two equal
    # values a min returns is the reduction order's business — torch.min's too; 1 ray of 9.6 M at configs[4]), NaN where the oracle has NaN
    want_all = np.concatenate((want_rock, want_terr), axis=1)
    differ = ~((dist == want_all) | (np.isnan(dist) & np.isnan(want_all)))
    n_bad = int(differ.sum())
    n_bits = int((dist.view(np.uint32) != want_all.view(np.uint32)).sum())
    print(f"[{label}] ray phase on the device's own rays vs the oracle: {n_bad} of {dist.size} distances differ ({n_bits} in their bits: +0 / -0)")
    for e, sl in list(zip(*np.nonzero(differ)))[:8]:
        print(f"   env {e} slot {sl}: src {src[e, sl].tolist()} ({src[e, sl].view(np.uint32).tolist()}) dir {dirs[e, sl].tolist()} "
              f"({dirs[e, sl].view(np.uint32).tolist()}) cell {cell[e, sl]}: device {dist[e, sl]!r} oracle {want_all[e, sl]!r}")
    assert n_bad == 0, f"{label}: {n_bad} of {dist.size} ray distances differ from the oracle on IDENTICAL rays"
    return src, cell


def _all_envs_vs_oracle(scene, distn, st, label, budget=None):
    """Two comparisons per arithmetic (fp32 parity mode, the reference's as-shipped fp16 arithmetic), each on every env:
      (a) the ray phase on identical rays (`_rays_vs_oracle`): ZERO differing distances;
      (b) the whole step against the oracle's own step.  What may differ there is what the pose trigonometry moves: the device's
          sin / cos / atan2 differ from the host libm's by an ulp on some poses, which moves a ray origin by an ulp (a fp16 rounding
          in the as-shipped mode), and with it — rarely — a cell index or an eps-edge decision.  The numbers of rays whose origin /
          cell differ are measured and printed; the distance / flag budgets are twice what was measured on MI355X (`budget`)."""
    from hip_helpers import hip_step, make_engine
    from oracle import oracle as orc
    n = st["pos"].shape[0]
    maps = _oracle_maps(scene)
    t, r = maps
    # Budgets = twice what was MEASURED on MI355X (round 5, gpurun_out/fullsize_2.log; the numbers are printed on every run): fp32 hit <-> miss
    # flips 4e-7 of the terrain rays (configs[2]; 0 elsewhere), flags 0 (allowed: two envs of 65 536); as shipped 3.3e-5 ... 6.4e-5 of the
    # terrain rays, <= 2.5e-6 of the wheel rays, <= 7.6e-6 of the body rays, flags 0 — all of it on rays whose ORIGIN the pose trigonometry
    # moved (26 % of the fp32 origins differ from the host libm's by an ulp, 0.0003 % land in another cell; 0.015 % of the as-shipped
    # origins round to another fp16 value); on identical rays nothing differs (`_rays_vs_oracle`).
    b = dict(flips32=1e-6, flags32=3.1e-5, rays16=1.3e-4, flags16=3.1e-5)
    b.update(budget or {})
    eng = make_engine(scene, distn, n, variant=None)                 # the library's own choice of ray-cast kernel
    got = hip_step(eng, st)
    src, cell = _rays_vs_oracle(eng, maps, False, f"{label} fp32")
    eng.close()
    want = orc.step(t, r, st, *distn, num_envs_global=n, precision="fp32")
    g = dict(got)
    np.testing.assert_array_equal(g.pop("reset_ids"), np.nonzero(got["reset_buf"])[0])      # compaction of the flags the step itself set
    assert_step_close(g, {"out_" + k: v for k, v in want.items()}, f"{label} fp32, all {n} envs")
    # what the trigonometry moved: terrain ray origins that differ at all (an ulp), and how many of them land in another cell
    ws = want["ray_sources"].reshape(src[:, 26:].shape)
    moved = (src[:, 26:].view(np.uint32) != ws.view(np.uint32)).any(axis=2)
    cx = np.rint(np.clip((ws[..., 0] - scene.shift[0]) / np.float32(0.1), 0, scene.terrain.map_indices.shape[0] - 1))
    cy = np.rint(np.clip((ws[..., 1] - scene.shift[1]) / np.float32(0.1), 0, scene.terrain.map_indices.shape[0] - 1))
    cy = np.minimum(cy, scene.terrain.map_indices.shape[1] - 1)
    other_cell = (cell[:, 26:] != (cx * scene.terrain.map_indices.shape[1] + cy).astype(np.int64))
    flips = float(((got["ray_dist"] < 11.0) != (want["ray_dist"] < 11.0)).mean())
    f_reset = float((got["reset_buf"] != want["reset_buf"]).mean())
    f_coll = float((got["rock_collision"] != want["rock_collision"]).mean())
    print(f"[{label} fp32] whole step vs the oracle: {moved.mean():.4%} of the terrain ray origins differ by an ulp, "
          f"{other_cell.mean():.5%} land in another cell; hit<->miss flips {flips:.5%}, reset flags {f_reset:.5%}, collision flags {f_coll:.5%}")
    assert flips <= b["flips32"], f"{label}: {flips:.5%} of the terrain rays flip hit <-> miss against the oracle"
    assert f_reset <= b["flags32"] and f_coll <= b["flags32"]

    outs = {}
    for variant in (4, 3, 2):
        eng = make_engine(scene, distn, n, variant=variant)
        eng.set_option("ray_precision", 2)
        assert eng.info().raycast_variant == variant
        outs[variant] = hip_step(eng, st)
        if variant == 3:
            src16, _ = _rays_vs_oracle(eng, maps, True, f"{label} as shipped")
        eng.close()
    for k in outs[3]:
        np.testing.assert_array_equal(outs[3][k], outs[2][k], err_msg=f"{label} as shipped: {k}, culled vs every-triangle kernel")
        np.testing.assert_array_equal(outs[4][k], outs[2][k], err_msg=f"{label} as shipped: {k}, staged vs every-triangle kernel")
    want16 = orc.step(t, r, st, *distn, num_envs_global=n, precision="fp16_as_shipped")
    moved16 = (src16[:, 26:].view(np.uint32) != want16["ray_sources"].reshape(src16[:, 26:].shape).view(np.uint32)).any(axis=2)
    bad = {k: float((outs[3][k] != want16[k]).mean()) for k in ("ray_dist", "wheel_dist", "body_dist")}
    f_reset = float((outs[3]["reset_buf"] != want16["reset_buf"]).mean())
    f_coll = float((outs[3]["rock_collision"] != want16["rock_collision"]).mean())
    print(f"[{label} as shipped] whole step vs the oracle: {moved16.mean():.4%} of the terrain ray origins round to another fp16 value; "
          f"distances that differ: {bad}; reset flags {f_reset:.5%}, collision flags {f_coll:.5%}")
    for k, v in bad.items():
        assert v <= b["rays16"], f"{label} as shipped: {k}: {v:.4%} of the rays differ from the oracle's fp16 mode"
    assert f_reset <= b["flags16"] and f_coll <= b["flags16"]
    np.testing.assert_array_equal(outs[3]["progress_buf"], want16["progress_buf"])
    return got


def test_full_size_config2_every_env_against_the_oracle(full):
    """BASELINE configs[2]: all 65 536 envs (4.1 M rays x K = 200) against the oracle, fp32 and as shipped."""
    scene, distn, st = full
    _all_envs_vs_oracle(scene, distn, st, "configs[2]")


def test_full_size_config4_every_env_against_the_oracle():
    """BASELINE configs[4]'s step (65 536 envs, 120 + 26 rays): all envs against the oracle, fp32 and as shipped."""
    from isaac_rover_amd import synth
    scene = synth.make_scene(n_cells=CELLS, k=K, n_stones=1024, device="cuda")
    _all_envs_vs_oracle(scene, synth.ray_distribution("120"), synth.make_states(E, CELLS * 0.1, seed=9), "configs[4]")


def test_full_size_irregular_mesh_every_env_against_the_oracle():
    """configs[2]'s batch on the scene of `bench.py --mesh irregular` (decimated-style mesh: 349 k triangles from millimetres to
    metres, ~80 degree rock flanks, needle / zero-area triangles, shuffled ids; K = 200 maps by rover_build_knn_map): all 65 536
    envs against the oracle, fp32 and as shipped — the geometry the reference's real terrain has
    (utils/terrain_utils/terrain_generation.py:217-243)."""
    from isaac_rover_amd import _lib, assets, synth
    spec = synth.IrregularSpec(extent_x=CELLS * 0.1, extent_y=CELLS * 0.1, n_rocks=1024, seed=5, fine=0.05)
    tool = _lib.Engine(8, device=0)
    scene, zf = assets.build_irregular_scene(tool, spec, K)
    tool.close()
    st = synth.make_states(E, CELLS * 0.1, seed=11, heightfn=zf)
    got = _all_envs_vs_oracle(scene, synth.ray_distribution("37"), st, "irregular mesh")
    assert 0.3 < (got["ray_dist"] < 11.0).mean() <= 1.0


def test_reference_operating_point_512_envs_native_rays():
    """The reference's own operating point: numEnvs 512 (cfg/task/Rover.yaml:11) x its native 1 634-point heightmap + 26 rock rays
    (heightmap_distribution.py:36-115), K = 200 (rover_utils.py:49), on a decimated-style mesh: every env against the oracle in
    both arithmetics, with the kernel the library itself picks at this size."""
    from isaac_rover_amd import _lib, assets, synth
    from isaac_rover_amd.tasks.utils.heightmap_distribution import generate_native
    spec = synth.IrregularSpec(extent_x=30.0, extent_y=30.0, n_rocks=256, seed=6, fine=0.05)
    tool = _lib.Engine(8, device=0)
    scene, zf = assets.build_irregular_scene(tool, spec, K)
    tool.close()
    distn = tuple(np.asarray(x) for x in generate_native())
    assert distn[0].shape[0] == 1634
    st = synth.make_states(512, 30.0, seed=12, heightfn=zf)
    got = _all_envs_vs_oracle(scene, distn, st, "512 envs x native rays")
    assert got["obs_buf"].shape == (512, 1750)
