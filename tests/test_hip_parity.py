"""GPU (-m gpu): the HIP path, called through the C ABI, against (a) golden vectors captured from the
reference and (b) the CPU oracle on the same seeded inputs.  Tolerances: conftest.py."""
import os

import numpy as np
import pytest
import torch

from conftest import (RAY_FLIP_BUDGET, STEP_FIXTURES_AS_SHIPPED, STEP_FIXTURES_FP32, TOL_RAY, assert_step_close, load_golden, scene_for,
                      states_of)

pytestmark = pytest.mark.gpu


def _oracle_maps(scene):
    from oracle import oracle as orc
    return (orc.KnnMap(scene.terrain.map_indices, scene.terrain.triangles, scene.terrain.vertices),
            orc.KnnMap(scene.rocks.map_indices, scene.rocks.triangles, scene.rocks.vertices))


@pytest.mark.parametrize("name", STEP_FIXTURES_FP32)
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("variant", [None, 4, 3])
def test_step_matches_reference_golden(name, fused, variant):
    """Every fp32 fixture captured from the reference, through the library's own choice of ray-cast kernel (None), the staged kernel
    (4: the one full batches and bench.py run) and the culled kernel (3)."""
    from hip_helpers import hip_step, make_engine
    fx = load_golden(name)
    scene = scene_for(fx)
    st = states_of(fx)
    eng = make_engine(scene, (fx["distribution"], fx["sparse_idx"], fx["dense_idx"]), st["pos"].shape[0], variant=variant,
                      curriculum_level=int(fx["curriculum_level"]), num_envs_global=int(fx["num_envs_global"]))
    if variant is not None:
        assert eng.info().raycast_variant == variant
    out = hip_step(eng, st, fused=fused)
    assert_step_close(out, fx, name)
    np.testing.assert_array_equal(out["reset_ids"], np.nonzero(out["reset_buf"])[0])
    eng.close()


@pytest.mark.parametrize("name,precision", [("step_lattice_fp32", "fp32"), ("step_lattice_fp16_as_shipped", "fp16_as_shipped")])
@pytest.mark.parametrize("variant", [4, 3, 2, 1])
def test_exact_ties_follow_the_reference(name, precision, variant):
    """The lattice fixture (exact threshold / det-guard / degenerate-triangle hits, see tests/test_oracle_golden.py):
    on its rounding-free envs the HIP ray casts equal the reference bit for bit, in both kernels."""
    from hip_helpers import hip_step, make_engine
    if precision == "fp16_as_shipped" and variant == 1:
        pytest.skip("the env-order kernel has no as-shipped fp16 arithmetic")
    fx = load_golden(name)
    scene = scene_for(fx)
    st = states_of(fx)
    eng = make_engine(scene, (fx["distribution"], fx["sparse_idx"], fx["dense_idx"]), st["pos"].shape[0], variant=variant)
    eng.set_option("ray_precision", {"fp32": 0, "fp16_as_shipped": 2}[precision])
    assert eng.info().raycast_variant == variant
    out = hip_step(eng, st)
    n = int(fx["exact_envs"])
    for key in ("ray_dist", "wheel_dist", "body_dist", "rock_collision", "reset_buf"):
        np.testing.assert_array_equal(out[key][:n], fx["out_" + key][:n], err_msg=key)
    np.testing.assert_array_equal(out["obs_buf"][:n, 4:], fx["out_obs_buf"][:n, 4:])
    eng.close()


@pytest.mark.parametrize("dist_name,num_envs,seed", [("37", 4096, 11), ("120", 1024, 12), ("9", 2048, 13)])
def test_step_matches_oracle(dist_name, num_envs, seed):
    """BASELINE.json configs[1] size (4096 envs, 37 rays) against the CPU oracle on the same seeded inputs."""
    from hip_helpers import hip_step, make_engine
    from isaac_rover_amd import synth
    from oracle import oracle as orc
    scene = synth.make_scene(n_cells=128, k=24, n_stones=48)
    distn = synth.ray_distribution(dist_name)
    st = synth.make_states(num_envs, 12.8, seed=seed)
    t, r = _oracle_maps(scene)
    want = orc.step(t, r, st, *distn)
    eng = make_engine(scene, distn, num_envs)
    got = hip_step(eng, st)
    assert_step_close(got, {"out_" + k: v for k, v in want.items()}, f"oracle:{dist_name}")
    # the ray kernel and the oracle are both IEEE op-by-op: only the trig ulps of the pose stage separate them
    d = np.abs(got["ray_dist"] - want["ray_dist"])
    assert np.quantile(d, 0.999) < 5e-5
    assert (got["reset_buf"] != want["reset_buf"]).mean() < 1e-3
    eng.close()


@pytest.mark.parametrize("dist_name,num_envs,k", [("37", 4096, 24), ("120", 512, 200), ("9", 300, 7), ("37", 700, 100),
                                                  ("9", 200, 16), ("37", 600, 256), ("37", 300, 250)])   # 256 / 250: a full id row (64 lanes x 4)
def test_raycast_variants_bit_identical(dist_name, num_envs, k):
    """Variant 1 (half-wave per ray, env order), variant 2 (rays binned by cell, shared-reciprocal IEEE division,
    any run length, early out on or off) and variant 3 (culled: bounding-sphere / normal test first, exact arithmetic on the
    candidates only) must agree bit for bit, and with the oracle's ray maths given the same rays."""
    from hip_helpers import hip_step, make_engine
    from isaac_rover_amd import synth
    scene = synth.make_scene(n_cells=64, k=k, n_stones=24)
    distn = synth.ray_distribution(dist_name)
    st = synth.make_states(num_envs, 6.4, seed=21)
    ref = hip_step(make_engine(scene, distn, num_envs, variant=1), st)
    for variant, run, early_out in ((2, 1, 1), (2, 5, 0), (2, 16, 1), (2, 16, 0), (2, 64, 1),
                                    (3, 1, 1), (3, 5, 1), (3, 32, 1), (3, 64, 1), (3, 200, 1)):
        eng = make_engine(scene, distn, num_envs, variant=variant, run=run)
        eng.set_option("raycast_early_out", early_out)       # conservative whole-pair rejection: same bits on or off
        assert eng.info().raycast_variant == variant
        got = hip_step(eng, st)
        for key in ref:
            np.testing.assert_array_equal(got[key], ref[key], err_msg=f"{key} variant={variant} run={run} early_out={early_out}")
        eng.close()


def test_crowded_cell_variants_bit_identical():
    """All 4 096 envs stand on the same spot (two poses): a handful of (map, cell) bins hold every ray — one coarse bucket of
    150 k rays (the bucket sort's general path, past the 16 384 rays its one-pass path takes), runs that never leave a bin, bins
    that span hundreds of runs.  Variants 3 and 2 must still agree with variant 1 bit for bit."""
    from hip_helpers import hip_step, make_engine
    from isaac_rover_amd import synth
    num_envs = 4096
    scene = synth.make_scene(n_cells=64, k=40, n_stones=24)
    distn = synth.ray_distribution("37")
    one = synth.make_states(2, 6.4, seed=5)
    st = {k: v[torch.arange(num_envs) % 2].clone() for k, v in one.items()}
    ref = hip_step(make_engine(scene, distn, num_envs, variant=1), st)
    for variant in (3, 2):
        eng = make_engine(scene, distn, num_envs, variant=variant)
        got = hip_step(eng, st)
        for key in ref:
            np.testing.assert_array_equal(got[key], ref[key], err_msg=f"{key} variant={variant}")
        eng.close()


def test_fp16_source_option_matches_oracle_and_as_shipped_reference():
    """ray_precision 1: origins / directions rounded to fp16 like the reference as shipped, f32 maths after."""
    from hip_helpers import hip_step, make_engine
    from oracle import oracle as orc
    fx16 = load_golden("step_e64_p37_fp16_as_shipped")
    scene = scene_for(fx16)
    st = states_of(fx16)
    distn = (fx16["distribution"], fx16["sparse_idx"], fx16["dense_idx"])
    eng = make_engine(scene, distn, 64)
    eng.set_option("ray_precision", 1)
    got = hip_step(eng, st)
    t, r = _oracle_maps(scene)
    want = orc.step(t, r, st, *distn, precision="fp16_sources")
    assert_step_close(got, {"out_" + k: v for k, v in want.items()}, "fp16-sources vs oracle")
    d = np.abs(got["ray_dist"].astype(np.float64) - fx16["out_ray_dist"])
    assert d.mean() < 1e-3 and (d > 0.05).mean() == 0.0                     # vs the reference AS SHIPPED
    np.testing.assert_array_equal(got["reset_buf"], fx16["out_reset_buf"])
    np.testing.assert_array_equal(got["rock_collision"], fx16["out_rock_collision"])
    eng.close()


@pytest.mark.parametrize("variant", [4, 3, 2])
@pytest.mark.parametrize("name", STEP_FIXTURES_AS_SHIPPED)
def test_as_shipped_fp16_mode_is_bit_identical_to_the_reference(name, variant):
    """ray_precision 2: the reference AS SHIPPED (Camera.dtype = float16).  Ray distances, wheel / body distances, the
    collision mask, done flags and the heightmap part of obs are bit-identical to the golden vectors captured from the
    unmodified reference (9 / 37 / 120 / native 1634 rays, K = 200, the irregular mesh); reward differs by sin/cos/atan2 ulps
    only.  Variant 3: the culled ray cast with the fp16 proof tables and the fp16 exact phase; variant 2: every triangle."""
    from hip_helpers import hip_step, make_engine
    from oracle import oracle as orc
    fx16 = load_golden(name)
    scene = scene_for(fx16)
    st = states_of(fx16)
    distn = (fx16["distribution"], fx16["sparse_idx"], fx16["dense_idx"])
    eng = make_engine(scene, distn, st["pos"].shape[0], variant=variant)
    eng.set_option("ray_precision", 2)
    assert eng.info().raycast_variant == variant
    got = hip_step(eng, st)
    for k in ("ray_dist", "wheel_dist", "body_dist", "rock_collision", "reset_buf", "progress_buf", "extras_collision_penalty"):
        np.testing.assert_array_equal(got[k], fx16["out_" + k], err_msg=k)
    np.testing.assert_array_equal(got["obs_buf"][:, 4:], fx16["out_obs_buf"][:, 4:])
    np.testing.assert_allclose(got["obs_buf"][:, :4], fx16["out_obs_buf"][:, :4], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got["rew_buf"], fx16["out_rew_buf"], rtol=1e-5, atol=1e-5)
    if name != "step_e64_p37_fp16_as_shipped":
        eng.close()
        return
    # and against the oracle's fp16 mode on a bigger random batch
    from isaac_rover_amd import synth
    st2 = synth.make_states(2048, 12.8, seed=77)
    t, r = _oracle_maps(scene)
    want = orc.step(t, r, st2, *distn, precision="fp16_as_shipped")
    eng2 = make_engine(scene, distn, 2048, variant=variant)
    eng2.set_option("ray_precision", 2)
    got2 = hip_step(eng2, st2)
    for k in ("ray_dist", "wheel_dist", "body_dist"):
        bad = (got2[k] != want[k])
        assert bad.mean() < 2e-3, f"{k}: {bad.mean():.4%} of rays differ (trig ulps moving an fp16 rounding)"
    assert (got2["reset_buf"] != want["reset_buf"]).mean() < 2e-3
    eng.close(); eng2.close()


def _custom_scene(n_x, n_y, k_t, k_r, shift=(0.0, 0.0, 0.0), seed=0):
    """Non-square maps with different K per map, cut out of a square synthetic scene."""
    from isaac_rover_amd import synth
    n = max(n_x, n_y)
    base_t = synth.make_scene(n_cells=n, k=k_t, n_stones=12, seed=seed)
    base_r = base_t if k_r == k_t else synth.make_scene(n_cells=n, k=k_r, n_stones=12, seed=seed)
    t = synth.KnnMap(base_t.terrain.map_indices[:n_x, :n_y].contiguous(), base_t.terrain.triangles, base_t.terrain.vertices)
    r = synth.KnnMap(base_r.rocks.map_indices[:n_x, :n_y].contiguous(), base_r.rocks.triangles, base_r.rocks.vertices)
    return synth.Scene(terrain=t, rocks=r, stone_info_raw=base_t.stone_info_raw, heightmap=base_t.heightmap, shift=shift)


@pytest.mark.parametrize("num_envs,n_x,n_y,k_t,k_r,dist_name,shift", [
    (1, 40, 40, 16, 16, "9", (0.0, 0.0, 0.0)),          # a single env
    (257, 48, 32, 16, 16, "37", (0.0, 0.0, 0.0)),       # E not a multiple of the block size, X > Y
    (100, 32, 48, 1, 1, "9", (0.0, 0.0, 0.0)),          # K = 1, X < Y
    (64, 40, 40, 255, 255, "9", (0.0, 0.0, 0.0)),       # K8 = 256: the widest the register-resident variants take
    (64, 40, 40, 300, 300, "9", (0.0, 0.0, 0.0)),       # K8 > 256: falls back to the streaming variant
    (200, 40, 40, 24, 40, "37", (0.0, 0.0, 0.0)),       # different K for terrain and rocks
    (128, 40, 40, 16, 16, "120", (-1.3, 0.7, 0.0)),     # shifted map origin (camera.py:239-241)
])
def test_odd_shapes_match_oracle(num_envs, n_x, n_y, k_t, k_r, dist_name, shift):
    from hip_helpers import hip_step
    from isaac_rover_amd import _lib, synth
    from oracle import oracle as orc
    scene = _custom_scene(n_x, n_y, k_t, k_r, shift)
    distn = synth.ray_distribution(dist_name)
    st = synth.make_states(num_envs, min(n_x, n_y) * 0.1, seed=31)
    t = orc.KnnMap(scene.terrain.map_indices, scene.terrain.triangles, scene.terrain.vertices, shift=shift[0:2])
    r = orc.KnnMap(scene.rocks.map_indices, scene.rocks.triangles, scene.rocks.vertices, shift=shift[0:2])
    want = orc.step(t, r, st, *distn)
    results = []
    for variant in (4, 3, 2, 1):
        eng = _lib.Engine(num_envs, device=0)
        eng.set_scene(scene, distn)
        eng.set_option("raycast_variant", variant)
        got = hip_step(eng, st)
        assert_step_close(got, {"out_" + k: v for k, v in want.items()}, f"odd:{variant}")
        np.testing.assert_array_equal(got["reset_ids"], np.nonzero(got["reset_buf"])[0])
        results.append(got)
        eng.close()
    for other in results[1:]:
        for k in results[0]:
            np.testing.assert_array_equal(results[0][k], other[k], err_msg=k)


@pytest.mark.parametrize("precision", [0, 2])
@pytest.mark.parametrize("k", [200, 40])
def test_early_out_changes_no_bit(precision, k):
    """The conservative whole-pair rejection of the binned kernels (f32 and as-shipped fp16) on a batch large enough to
    hit its margins from both sides: every distance identical with the option off."""
    from hip_helpers import hip_step, make_engine
    from isaac_rover_amd import synth
    scene = synth.make_scene(n_cells=96, k=k, n_stones=40)
    distn = synth.ray_distribution("120")
    st = synth.make_states(3000, 9.6, seed=77)
    st["quat"] = synth.quat_from_euler(0.5 * torch.randn(3000), 0.5 * torch.randn(3000), 3.0 * torch.randn(3000))   # steep tilts
    outs = []
    for early_out in (1, 0):
        eng = make_engine(scene, distn, 3000, variant=2)
        eng.set_option("ray_precision", precision)
        eng.set_option("raycast_early_out", early_out)
        outs.append(hip_step(eng, st))
        eng.close()
    for key in outs[0]:
        np.testing.assert_array_equal(outs[0][key], outs[1][key], err_msg=key)
    assert (outs[0]["ray_dist"] < 11.0).mean() > 0.5


@pytest.mark.parametrize("k,cells", [(200, 96), (40, 96), (255, 48)])
def test_culled_raycast_changes_no_bit(k, cells):
    """The culled kernel (variant 3) against the binned kernel with its early out off, on a batch with steep tilts, rays
    parallel to facets (exact axis-aligned poses on the vertex lattice) and huge / NaN poses: every output identical, for
    several run lengths."""
    from hip_helpers import hip_step, make_engine
    from isaac_rover_amd import synth
    n = 3000
    scene = synth.make_scene(n_cells=cells, k=k, n_stones=40)
    distn = synth.ray_distribution("120")
    st = synth.make_states(n, cells * 0.1, seed=78)
    g = torch.Generator().manual_seed(5)
    st["quat"] = synth.quat_from_euler(0.5 * torch.randn(n, generator=g), 0.5 * torch.randn(n, generator=g), 3.0 * torch.randn(n, generator=g))
    q = torch.randn(1000, 4, generator=g)
    st["quat"][1000:2000] = q / q.norm(dim=1, keepdim=True)                      # arbitrary orientations
    axis = torch.tensor([[1.0, 0, 0, 0], [0.70710678, 0.70710678, 0, 0], [0.70710678, 0, 0.70710678, 0], [0, 1.0, 0, 0]])
    st["quat"][2000:2900] = axis[torch.randint(0, 4, (900,), generator=g)]       # rays in facet planes, through vertices
    st["pos"][2000:2900, 0:2] = torch.round(st["pos"][2000:2900, 0:2] * 20) / 20
    st["pos"][2900:2950] *= 1.0e4                                                 # far outside the map
    st["pos"][2950:2960] = float("nan")
    st["pos"][2960:2970, 2] += 500.0                                              # high above the terrain
    for precision in (0, 2):                         # f32 arithmetic / the reference's as-shipped fp16 arithmetic
        eng = make_engine(scene, distn, n, variant=2)
        eng.set_option("ray_precision", precision)
        eng.set_option("raycast_early_out", 0)
        ref = hip_step(eng, st)
        eng.close()
        for run in (0, 7, 64):                       # auto / odd / longest runs of sorted rays per wave
            eng = make_engine(scene, distn, n, variant=3, run=run or None)
            eng.set_option("ray_precision", precision)
            assert eng.info().raycast_variant == 3
            got = hip_step(eng, st)
            got2 = hip_step(eng, st)                 # and again on the same engine (queue regions are reused)
            eng.close()
            for key in ref:
                np.testing.assert_array_equal(got[key], ref[key], err_msg=f"{key} run={run} precision={precision}")
                np.testing.assert_array_equal(got2[key], ref[key], err_msg=f"{key} run={run} precision={precision} (second step)")
        assert (ref["ray_dist"] < 11.0).mean() > 0.3


def _adversarial_states(n, extent, seed=78, heightfn=None, margin_m=None):
    from isaac_rover_amd import synth
    kw = {} if heightfn is None else dict(heightfn=heightfn, margin_m=margin_m)
    st = synth.make_states(n, extent, seed=seed, **kw)
    g = torch.Generator().manual_seed(5)
    st["quat"] = synth.quat_from_euler(0.5 * torch.randn(n, generator=g), 0.5 * torch.randn(n, generator=g), 3.0 * torch.randn(n, generator=g))
    st["quat"][0:600] = synth.quat_from_euler(0.08 * torch.randn(600, generator=g), 0.08 * torch.randn(600, generator=g), 3.0 * torch.randn(600, generator=g))
    q = torch.randn(1000, 4, generator=g)
    st["quat"][1000:2000] = q / q.norm(dim=1, keepdim=True)                      # arbitrary orientations
    axis = torch.tensor([[1.0, 0, 0, 0], [0.70710678, 0.70710678, 0, 0], [0.70710678, 0, 0.70710678, 0], [0, 1.0, 0, 0]])
    st["quat"][2000:2900] = axis[torch.randint(0, 4, (900,), generator=g)]       # rays in facet planes, through vertices
    st["pos"][2000:2900, 0:2] = torch.round(st["pos"][2000:2900, 0:2] * 20) / 20
    st["pos"][2900:2950] *= 1.0e4                                                 # far outside the map
    st["pos"][2950:2960] = float("nan")
    st["pos"][2960:2970, 2] += 500.0                                              # high above the terrain
    return st


def _staged_against_every_triangle(scene, distn, st):
    """variant 4 (both ways of casting the rocks part, short and long runs) against the binned kernel with its early out off, in the f32
    arithmetic and in the reference's as-shipped fp16 arithmetic: every output identical."""
    from hip_helpers import hip_step, make_engine
    n = st["pos"].shape[0]
    for precision in (0, 2):
        eng = make_engine(scene, distn, n, variant=2)
        eng.set_option("ray_precision", precision)
        eng.set_option("raycast_early_out", 0)
        ref = hip_step(eng, st)
        eng.close()
        for lane_rocks, run, env_order in ((0, None, 0), (1, None, 0), (1, 7, 0), (0, 64, 0), (1, None, 1)):
            eng = make_engine(scene, distn, n, variant=4, run=run)
            eng.set_option("ray_precision", precision)
            eng.set_option("lane_rocks", lane_rocks)
            eng.set_option("lane_env_order", env_order)         # 1: no sort, the ray slots in env order (padding slots skipped)
            assert eng.info().raycast_variant == 4
            got = hip_step(eng, st)
            got2 = hip_step(eng, st)
            ci = eng.cull_info()
            eng.close()
            for key in ref:
                np.testing.assert_array_equal(got[key], ref[key], err_msg=f"{key} precision={precision} lane_rocks={lane_rocks} run={run}")
                np.testing.assert_array_equal(got2[key], ref[key], err_msg=f"{key} precision={precision} lane_rocks={lane_rocks} run={run} (second step)")
            assert ci["rays"] == n * (distn[0].shape[0] + 26), (lane_rocks, run, env_order)
    return ref


@pytest.mark.parametrize("k,cells,rays", [(200, 96, "120"), (200, 96, "37"), (40, 96, "120"), (255, 48, "37"), (8, 64, "9")])
def test_staged_raycast_changes_no_bit(k, cells, rays):
    """The staged kernel (variant 4: lane = (ray, chunk of 8 pairs) over per-cell record rows in group-bound order — fp16 records
    relative to the cell, suffix bounds and suffix cones every 8 pairs (K = 255: all 16 levels in use), tests (A) and (B) for rays off
    their cell's cone) against the
    binned kernel with its early out off — every triangle evaluated — on the batch of test_culled_raycast_changes_no_bit: steep tilts,
    arbitrary orientations, rays in facet planes, poses far outside the map, NaN."""
    from isaac_rover_amd import synth
    scene = synth.make_scene(n_cells=cells, k=k, n_stones=40)
    ref = _staged_against_every_triangle(scene, synth.ray_distribution(rays), _adversarial_states(3000, cells * 0.1))
    assert (ref["ray_dist"] < 11.0).mean() > 0.3


@pytest.mark.parametrize("seed,k,coarse,fine", [(1, 200, 1.2, 0.0375), (2, 64, 3.0, 0.05)])
def test_staged_raycast_changes_no_bit_on_irregular_meshes(seed, k, coarse, fine):
    """The same on irregular meshes (needles, flanks, always-candidate triangles, cells without a cone: most rays run both tests, the
    steep triangles sit in front of their cells' rows)."""
    from isaac_rover_amd import _lib, assets, synth
    spec = synth.IrregularSpec(extent_x=12.0, extent_y=12.0, n_rocks=24, seed=seed, coarse=coarse, fine=fine)
    tool = _lib.Engine(8, device=0)
    scene, zf = assets.build_irregular_scene(tool, spec, k)
    tool.close()
    distn = synth.ray_distribution("120" if seed != 2 else "37")
    _staged_against_every_triangle(scene, distn, _adversarial_states(3000, 12.0, seed=80 + seed, heightfn=zf, margin_m=1.0))


@pytest.mark.parametrize("name,precision", [("step_irregular_p37_fp32", 0), ("step_irregular_native_fp32", 0),
                                            ("step_e32_p37_k200_fp32", 0), ("step_irregular_p37_fp16_as_shipped", 2),
                                            ("step_e32_p37_k200_fp16_as_shipped", 2)])
def test_irregular_mesh_and_k200_all_variants(name, precision):
    """Fixtures captured from the reference on (a) an IRREGULAR mesh — non-uniform Delaunay triangulation, edges from millimetres
    to metres, ~80 degree rock flanks, needle / zero-area triangles, duplicated vertices, mixed windings, shuffled ids — with
    K = 200 maps built by the reference's own _get_knn_triangles, and (b) the regular scene at the reference's K = 200:
    the culled (3), binned (2) and env-order (1) kernels agree with the reference within the stated tolerance and with EACH
    OTHER bit for bit; in the as-shipped fp16 mode the distances equal the reference's bit for bit."""
    from hip_helpers import hip_step, make_engine
    fx = load_golden(name)
    scene = scene_for(fx)
    st = states_of(fx)
    distn = (fx["distribution"], fx["sparse_idx"], fx["dense_idx"])
    outs = {}
    for variant in (4, 3, 2, 1):
        if precision == 2 and variant == 1:
            continue                                  # the env-order kernel has no as-shipped fp16 arithmetic
        eng = make_engine(scene, distn, st["pos"].shape[0], variant=variant)
        eng.set_option("ray_precision", precision)
        if eng.info().raycast_variant != variant:
            eng.close()
            continue                                  # (a precision this variant does not implement)
        outs[variant] = hip_step(eng, st)
        if variant == 3:
            ci = eng.cull_info()
            assert ci["rays"] == st["pos"].shape[0] * (distn[0].shape[0] + 26)
            if "irregular" in name:                   # the paths a regular grid mesh never takes are taken here
                assert ci["always_candidate_triangles"][0] > 0 and (ci["cells_without_cone"][0] > 0 or precision == 2)
                assert ci["rays_both_tests"] > 0 and ci["candidate_pairs"] > 0
        eng.close()
    assert 2 in outs and 4 in outs and (3 in outs or precision == 2)
    for variant, out in outs.items():
        if precision == 2:
            for key in ("ray_dist", "wheel_dist", "body_dist", "rock_collision", "reset_buf"):
                np.testing.assert_array_equal(out[key], fx["out_" + key], err_msg=f"{key} variant={variant}")
            np.testing.assert_array_equal(out["obs_buf"][:, 4:], fx["out_obs_buf"][:, 4:])
        else:
            assert_step_close(out, fx, f"{name} variant={variant}")
        for key in out:
            np.testing.assert_array_equal(out[key], outs[2][key], err_msg=f"{key} variant={variant} vs 2")


@pytest.mark.parametrize("seed,k,coarse,fine", [(1, 200, 1.2, 0.0375), (2, 64, 3.0, 0.05), (3, 200, 0.6, 0.03)])
def test_culled_raycast_changes_no_bit_on_irregular_meshes(seed, k, coarse, fine):
    """The culled kernel against the binned kernel without early out and the env-order kernel on irregular meshes (maps by
    the GPU builder): steep tilts, arbitrary orientations, rovers parked on rock flanks, poses outside the map, NaN."""
    from hip_helpers import hip_step, make_engine
    from isaac_rover_amd import _lib, assets, synth
    n = 3000
    spec = synth.IrregularSpec(extent_x=12.0, extent_y=12.0, n_rocks=24, seed=seed, coarse=coarse, fine=fine)
    tool = _lib.Engine(8, device=0)
    scene, zf = assets.build_irregular_scene(tool, spec, k)
    tool.close()
    distn = synth.ray_distribution("120" if seed != 2 else "37")
    st = synth.make_states(n, 12.0, seed=80 + seed, heightfn=zf, margin_m=1.0)
    g = torch.Generator().manual_seed(seed)
    st["quat"] = synth.quat_from_euler(0.5 * torch.randn(n, generator=g), 0.5 * torch.randn(n, generator=g), 3.0 * torch.randn(n, generator=g))
    q = torch.randn(1000, 4, generator=g)
    st["quat"][1000:2000] = q / q.norm(dim=1, keepdim=True)
    _zf, (rxy, rr, _rh, _rp) = synth.irregular_height(spec)
    for e in range(2000, 2600):                                                   # on the rocks' flanks, close to the surface
        i = e % len(rr)
        a = 0.37 * e
        x, y = rxy[i, 0] + 0.75 * rr[i] * np.cos(a), rxy[i, 1] + 0.75 * rr[i] * np.sin(a)
        st["pos"][e, 0], st["pos"][e, 1], st["pos"][e, 2] = float(x), float(y), float(zf(x, y)) + 0.3
    st["pos"][2900:2950] *= 1.0e4
    st["pos"][2950:2960] = float("nan")
    st["pos"][2960:2970, 2] += 500.0
    for precision in (0, 2):                          # f32 arithmetic / the reference's as-shipped fp16 arithmetic
        eng = make_engine(scene, distn, n, variant=2)
        eng.set_option("ray_precision", precision)
        eng.set_option("raycast_early_out", 0)
        ref = hip_step(eng, st)
        eng.close()
        if precision == 0:
            eng = make_engine(scene, distn, n, variant=1)
            ref1 = hip_step(eng, st)
            eng.close()
            for key in ref:
                np.testing.assert_array_equal(ref1[key], ref[key], err_msg=f"{key} variant 1 vs 2")
        most = 0
        # (lazy: the scan kernel that fetches a bin's far records on demand — chosen by the library for small ray sets on meshes whose
        #  cells mostly have a far bound, which this one is not; forced here through the experiment variable, read at rover_create)
        for run, queue_mb, lazy in ((0, None, None), (7, None, "1"), (64, None, "1"), (64, None, "0"), (64, 1, None)):
            if lazy is not None:
                os.environ["ROVER_CULL_LAZY"] = lazy
            try:
                eng = make_engine(scene, distn, n, variant=3, run=run or None)
            finally:
                os.environ.pop("ROVER_CULL_LAZY", None)
            if lazy is not None and precision == 0:
                assert eng.cull_info()["far_records_on_demand"] == int(lazy)
            eng.set_option("ray_precision", precision)
            if queue_mb:                              # a 1 MB queue budget: the step's ray cast is cut into slices that re-use the regions
                eng.set_option("cull_queue_mb", queue_mb)
            assert eng.info().raycast_variant == 3
            got = hip_step(eng, st)
            ci = eng.cull_info()
            eng.close()
            for key in ref:
                np.testing.assert_array_equal(got[key], ref[key], err_msg=f"{key} run={run} queue_mb={queue_mb} precision={precision}")
            assert ci["always_candidate_triangles"][0] > 0 and ci["rays_both_tests"] > 0
            assert ci["launches_per_step"] == (1 if not queue_mb else ci["launches_per_step"]) and (not queue_mb or ci["launches_per_step"] > 4)
            assert ci["queue_bytes"] <= (queue_mb or 1536) << 20
            most = max(most, ci["max_pairs_per_run"])
        # runs that found more candidates than a queue region holds (1 024 entries) were cast in several segments
        assert most > 1024, most
        assert (ref["ray_dist"] < 11.0).mean() > 0.3


def test_budget_capped_queue_and_a_larger_ray_set_later():
    """A queue capped by its budget keeps its size when the ray set grows (rover_set_distribution with more points), but the
    per-wave counters are sized by the ray count: they must be re-allocated, or the scan kernel's counter stores run past the
    buffer (round-3 advisor finding).  37 -> 120 points at a 1 MB budget, then the step and its counters."""
    from hip_helpers import hip_step, make_engine
    from isaac_rover_amd import synth
    n = 8192
    scene = synth.make_scene(n_cells=96, k=24, n_stones=12)
    st = synth.make_states(n, 9.6, seed=3)
    eng = make_engine(scene, synth.ray_distribution("37"), n, variant=3)
    eng.set_option("cull_queue_mb", 1)
    a = hip_step(eng, st)
    ci = eng.cull_info()
    assert ci["rays"] == n * 63 and ci["launches_per_step"] > 1 and ci["queue_bytes"] <= 1 << 20
    dense = synth.ray_distribution("120")
    eng.set_distribution(*dense)
    b = hip_step(eng, st)
    ci = eng.cull_info()
    assert ci["rays"] == n * 146 and ci["queue_bytes"] <= 1 << 20
    eng.close()
    ref = make_engine(scene, dense, n, variant=1)
    want = hip_step(ref, st)
    ref.close()
    for k in want:
        np.testing.assert_array_equal(b[k], want[k], err_msg=k)
    assert a["obs_buf"].shape[1] == 41 and b["obs_buf"].shape[1] == 124


def test_ray_sort_entry_layouts_agree():
    """The bucket sort keeps an entry in one dword (low bin bits | slot) while the slot ids leave room, else in two: 36 864
    envs x 64 slots with 4 096 bins per bucket (bin_low_bits 12) take the two-dword layout, the default the packed one — both
    with the large tiles of the first two passes (>= 2 M slots); 8 192 envs the small tiles.  The sorted kernels' results must
    equal the env-order kernel's either way."""
    from hip_helpers import hip_step, make_engine
    from isaac_rover_amd import synth
    _ray_sort_layouts(36864)
    _ray_sort_layouts(8192)


def _ray_sort_layouts(n):
    from hip_helpers import hip_step, make_engine
    from isaac_rover_amd import synth
    scene = synth.make_scene(n_cells=96, k=24, n_stones=12)
    distn = synth.ray_distribution("37")
    st = synth.make_states(n, 9.6, seed=77)
    eng = make_engine(scene, distn, n, variant=1)
    ref = hip_step(eng, st)
    eng.close()
    for variant in (3, 2):
        for low_bits in (None, 12, 8):
            eng = make_engine(scene, distn, n, variant=variant)
            if low_bits:
                eng.set_option("bin_low_bits", low_bits)
            got = hip_step(eng, st)
            eng.close()
            for key in ref:
                np.testing.assert_array_equal(got[key], ref[key], err_msg=f"{key} variant={variant} bin_low_bits={low_bits}")


def test_fused_sort_histogram_survives_option_and_shape_changes():
    """With 37 + 26 rays prep_rays_kernel counts the sort's coarse buckets itself into a table that has to be all zero when a step starts
    (bucket_sort_kernel clears it again; bin_hist_fused, rover_kernels.hip).  One engine through everything that changes the table's
    layout or who fills it — other poses every step, the sorted variants and the env-order one in turn, bin_low_bits 12 / 8 / default
    (other bucket counts in the same allocation), a ray set that keeps bucket_hist_kernel (120 + 26: R8 = 152) and back — against a fresh
    env-order engine each time."""
    from hip_helpers import hip_step, make_engine
    from isaac_rover_amd import synth
    n = 4096
    scene = synth.make_scene(n_cells=96, k=24, n_stones=12)
    sparse, dense = synth.ray_distribution("37"), synth.ray_distribution("120")
    eng = make_engine(scene, sparse, n, variant=3)
    script = [("variant", 3), ("variant", 3), ("variant", 2), ("bin_low_bits", 12), ("variant", 1), ("variant", 3), ("bin_low_bits", 8),
              ("dist", dense), ("variant", 2), ("dist", sparse), ("bin_low_bits", 0), ("variant", 3)]
    cur = sparse
    for i, (what, val) in enumerate(script):
        if what == "dist":
            eng.set_distribution(*val)
            cur = val
        else:
            eng.set_option("raycast_variant" if what == "variant" else what, val)
        st = synth.make_states(n, 9.6, seed=100 + i)
        got = hip_step(eng, st)
        ref = make_engine(scene, cur, n, variant=1)
        want = hip_step(ref, st)
        ref.close()
        for key in want:
            np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} after step {i}: {what}")
    eng.close()


def test_step_captured_in_a_graph_first_then_eager_and_replayed():
    """The fused sort histogram needs its count table zero at the start of every step and leaves it zero (bucket_sort_kernel).  A step
    that is only CAPTURED (torch.cuda.graph = hipGraph) on a fresh engine runs nothing: the eager step after it, the replays after that
    and another eager step must all see a zero table — each compared with a fresh env-order engine on the same poses."""
    from hip_helpers import hip_step, make_engine
    from isaac_rover_amd import synth
    n = 4096
    scene = synth.make_scene(n_cells=96, k=24, n_stones=12)
    distn = synth.ray_distribution("37")
    dev = torch.device("cuda:0")
    eng = make_engine(scene, distn, n, variant=3)
    states = [synth.make_states(n, 9.6, seed=200 + i) for i in range(4)]
    keys = ("pos", "quat", "joints", "target", "lin_hist", "ang_hist", "euler_pre")
    stat = {k: states[0][k].to(dev).contiguous().clone() for k in keys}
    progress = states[0]["progress"].to(dev).clone()
    sin = eng.make_in(*(stat[k] for k in keys), progress)
    obs = torch.zeros(n, eng.num_observations, device=dev)
    bufs = dict(rew=torch.zeros(n, device=dev), reset=torch.ones(n, dtype=torch.int64, device=dev),
                rock_collision=torch.zeros(n, dtype=torch.int64, device=dev),
                reset_ids=torch.full((n,), -1, dtype=torch.int64, device=dev), n_reset=torch.zeros(1, dtype=torch.int32, device=dev),
                ray_dist=torch.zeros(n, eng.P, device=dev), wheel_dist=torch.zeros(n, 24, device=dev), body_dist=torch.zeros(n, 2, device=dev))
    sout = eng.make_out(obs, **bufs)
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):                                       # captured, not run
        eng.step(sin, sout, increment_progress=False, compact=True)

    def check(i, how):
        torch.cuda.synchronize()
        ref = make_engine(scene, distn, n, variant=1)
        want = hip_step(ref, states[i])
        ref.close()
        np.testing.assert_array_equal(obs.cpu().numpy(), want["obs_buf"], err_msg=f"obs, {how}")
        np.testing.assert_array_equal(bufs["ray_dist"].cpu().numpy(), want["ray_dist"], err_msg=f"ray_dist, {how}")
        np.testing.assert_array_equal(bufs["wheel_dist"].cpu().numpy(), want["wheel_dist"], err_msg=f"wheel_dist, {how}")
        np.testing.assert_array_equal(bufs["rock_collision"].cpu().numpy(), want["rock_collision"], err_msg=f"rock_collision, {how}")

    def load(i):
        for k in keys:
            stat[k].copy_(states[i][k].to(dev))
        progress.copy_(states[i]["progress"].to(dev))

    load(0); eng.step(sin, sout, increment_progress=False, compact=True); check(0, "eager step after a capture")
    load(1); g.replay(); check(1, "first replay")
    load(2); g.replay(); check(2, "second replay")
    load(3); eng.step(sin, sout, increment_progress=False, compact=True); check(3, "eager step after the replays")
    eng.close()


def test_rays_that_clear_their_cell_are_not_scanned():
    """On a regular mesh the scan kernels do not scan rays that provably clear BOTH halves of their cell's triangles — most rock
    rays — and drop bins without a live ray (on an irregular mesh the eager kernel scans everything: test_culled_raycast_changes_no_bit_on_
    irregular_meshes forces both kernels there).  Same bits as the env-order kernel."""
    from hip_helpers import hip_step, make_engine
    from isaac_rover_amd import synth
    n = 8192
    scene = synth.make_scene(n_cells=160, k=40, n_stones=24)
    distn = synth.ray_distribution("37")
    st = synth.make_states(n, 16.0, seed=5)
    outs, infos = {}, {}
    for lazy in ("1", "0"):
        os.environ["ROVER_CULL_LAZY"] = lazy
        try:
            eng = make_engine(scene, distn, n, variant=3)
        finally:
            os.environ.pop("ROVER_CULL_LAZY", None)
        outs[lazy] = hip_step(eng, st)
        infos[lazy] = eng.cull_info()
        eng.close()
    eng = make_engine(scene, distn, n, variant=1)
    ref = hip_step(eng, st)
    eng.close()
    for lazy in outs:
        for key in ref:
            np.testing.assert_array_equal(outs[lazy][key], ref[key], err_msg=f"{key} lazy={lazy}")
    assert infos["1"]["far_records_on_demand"] == 1 and infos["0"]["far_records_on_demand"] == 0
    assert infos["1"]["rays_not_scanned"] > 0.1 * infos["1"]["rays"], infos["1"]
    # (the eager kernel leaves them out too where the library expects enough of them — most terrain cells with a far bound, rock rays a
    #  tenth of the ray set — and then counts the same rays; otherwise it scans everything)
    assert infos["0"]["rays_not_scanned"] in (0, infos["1"]["rays_not_scanned"])
    assert infos["1"]["bins"] <= infos["0"]["bins"]


def test_auto_variant_and_run_selection():
    """raycast_variant 0 (auto), f32 arithmetic: the env-order kernel below 24 576 rays per step, the staged kernel (4) from there on —
    in env order (no sort) while a terrain cell holds fewer than 1.5 heightmap rays and more than 64 cells hold one rover —; as shipped: the binned kernel up to 24 576 rays,
    above that the staged kernel — in env order below 98 304 rays, behind the sort beyond (on an irregular terrain mesh only from two heightmap
    rays per terrain cell: the culled kernel below that); K8 > 256 always falls back to the env-order kernel."""
    from hip_helpers import hip_step, make_engine
    from isaac_rover_amd import synth
    scene = synth.make_scene(n_cells=64, k=16, n_stones=8)
    distn = synth.ray_distribution("37")
    small = make_engine(scene, distn, 256, variant=None)
    assert small.info().raycast_variant == 1
    small.set_option("ray_precision", 2)
    assert small.info().raycast_variant == 2
    for n, want32, want16 in ((384, 1, 2), (512, 4, 4), (1024, 4, 4)):   # 24 192 / 32 256 / 64 512 rays: either side of the f32 and of the fp16 switch
        mid = make_engine(scene, distn, n, variant=None)
        assert mid.info().raycast_variant == want32, n
        mid.set_option("ray_precision", 2)
        assert mid.info().raycast_variant == want16, n
        assert mid.info().raycast_sorted == 0 or want16 != 4, n           # (as shipped, small batches: the staged kernel in env order)
        mid.close()
    # the staged kernel without the sort: fewer than 1.5 heightmap rays per terrain cell AND fewer than one rover per 64 cells
    wide = synth.make_scene(n_cells=160, k=16, n_stones=8)
    for n, want_sorted in ((392, 0), (400, 1), (1024, 1)):  # 25 600 cells: 65.3 / 64 / 25 cells per rover
        e = make_engine(wide, distn, n, variant=None)
        assert e.info().raycast_variant == 4 and e.info().raycast_sorted == want_sorted, n
        e.close()
    wider = synth.make_scene(n_cells=256, k=16, n_stones=8)
    sparse = make_engine(wider, distn, 2048, variant=None)   # as shipped, 129 024 rays, 1.2 heightmap rays per terrain cell, and — K = 16 — fewer than
    sparse.set_option("ray_precision", 2)                    # half of the cells with a usable far bound (the mark of an irregular mesh): the culled kernel
    assert sparse.info().raycast_variant == 3
    sparse.close()
    k200 = synth.make_scene(n_cells=96, k=200, n_stones=8)   # a regular mesh with the reference's K: most cells have a far bound
    reg = make_engine(k200, distn, 2048, variant=None)       # as shipped, 129 024 rays, 8.2 heightmap rays per terrain cell: staged behind the sort
    reg.set_option("ray_precision", 2)                       # since round 6 (rounds 4-5: the culled kernel below ten rays per cell)
    assert reg.info().raycast_variant == 4 and reg.info().raycast_sorted == 1 and reg.info().raycast_rocks_staged == 1
    reg.close()
    big = make_engine(scene, distn, 4096, variant=None)
    assert big.info().raycast_variant == 4 and big.info().raycast_sorted == 1
    big.set_option("ray_precision", 2)
    assert big.info().raycast_variant == 4 and big.info().raycast_sorted == 1      # as shipped, 37 heightmap rays per terrain cell of this small map: staged, sorted
    big.set_option("ray_precision", 0)
    st = synth.make_states(4096, 6.4, seed=3)
    a = hip_step(big, st)                                   # auto run length
    big.set_option("raycast_run", 16)
    b = hip_step(big, st)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    small.close(); big.close()


def test_nan_pose_does_not_fault():
    """A NaN quaternion / position (the reference would raise at the cell lookup): cell 0, every ray misses."""
    from hip_helpers import hip_step, make_engine
    from isaac_rover_amd import synth
    scene = synth.make_scene(n_cells=32, k=8, n_stones=4)
    distn = synth.ray_distribution("9")
    st = synth.make_states(16, 3.2, seed=1)
    st["quat"][3] = float("nan")
    st["pos"][5, 0] = float("nan")
    got = hip_step(make_engine(scene, distn, 16), st)
    assert (got["ray_dist"][[3, 5]] == 11.0).all()
    assert np.isfinite(got["obs_buf"][[0, 1, 2, 4]]).all()


def test_sharded_equals_whole():
    """Two ctxs over env shards (env_offset, num_envs_global) reproduce one ctx over all envs bit for bit."""
    from hip_helpers import hip_step, make_engine
    fx = load_golden("step_e64_p37_fp32")
    scene = scene_for(fx)
    st = states_of(fx)
    distn = (fx["distribution"], fx["sparse_idx"], fx["dense_idx"])
    whole = hip_step(make_engine(scene, distn, 64), st)
    parts = []
    for off in (0, 32):
        eng = make_engine(scene, distn, 32, num_envs_global=64, env_offset=off)
        parts.append(hip_step(eng, {k: v[off:off + 32] for k, v in st.items()}))
    for k in whole:
        np.testing.assert_array_equal(whole[k], np.concatenate([p[k] for p in parts]), err_msg=k)


def test_reset_path_matches_reference_golden():
    from isaac_rover_amd import _lib, synth
    fx = load_golden("reset_path")
    scene = scene_for(fx)
    eng = _lib.Engine(32, device=0)
    eng.set_scene(scene, synth.ray_distribution("9"))
    eng.set_stones(fx["stone_info"])
    dev = eng.device
    xy = torch.from_numpy(fx["xy"]).to(dev)
    np.testing.assert_allclose(eng.clearance(xy).cpu().numpy(), fx["clearance"], rtol=0, atol=2e-4)
    np.testing.assert_array_equal(eng.sample_height(xy).cpu().numpy(), fx["heights"])
    sp = torch.from_numpy(fx["spawn_in"]).to(dev)
    np.testing.assert_allclose(eng.shift_spawns(sp).cpu().numpy(), fx["spawn_out"], rtol=0, atol=1e-5)
    for sfx in ("", "_b"):
        ids = torch.from_numpy(fx["goal_env_ids" + sfx]).to(dev)
        initial = torch.from_numpy(fx["goal_initial"]).to(dev)
        target = torch.zeros(32, 3, device=dev)
        used = torch.zeros(1, dtype=torch.int32, device=dev)
        eng.generate_goals(ids, initial, target, draws=torch.from_numpy(fx["goal_draws" + sfx]).to(dev), n_draws_used=used)
        assert int(used.item()) == int(fx["goal_used" + sfx])
        np.testing.assert_allclose(target[:, 0:2].cpu().numpy(), fx["goal_targets" + sfx][:, 0:2], rtol=1e-6, atol=1e-5)
    # goal z of set_targets (rover.py:582-583) for the reset envs of the first case
    ids = torch.from_numpy(fx["goal_env_ids"]).to(dev)
    target = torch.zeros(32, 3, device=dev)
    eng.generate_goals(ids, initial, target, draws=torch.from_numpy(fx["goal_draws"]).to(dev))
    np.testing.assert_array_equal(target[ids, 2].cpu().numpy(), fx["goal_height"])
    # compaction order = nonzero order
    e = fx["reset_buf"].shape[0]
    eng2 = _lib.Engine(e, device=0, env_offset=5000)
    r = torch.from_numpy(fx["reset_buf"]).to(dev)
    ids_out = torch.zeros(e, dtype=torch.int64, device=dev)
    n = torch.zeros(1, dtype=torch.int32, device=dev)
    eng2.compact_resets(r, ids_out, n)
    np.testing.assert_array_equal(ids_out[: int(n.item())].cpu().numpy(), fx["reset_ids"] + 5000)


def test_generate_goals_matches_sequential_oracle_random_cases():
    """The parallel decomposition (per-entry first clear draw + env-0 replay) against the sequential loop of
    rover.py:544-549 as restated by the oracle, over random stone sets / id lists (with and without env 0)."""
    from isaac_rover_amd import _lib, synth
    from oracle import oracle as orc
    fx = load_golden("reset_path")
    scene = scene_for(fx)
    e = 48
    eng = _lib.Engine(e, device=0)
    eng.set_scene(scene, synth.ray_distribution("9"))
    dev = eng.device
    rng = np.random.default_rng(123)
    for case in range(40):
        s_cnt = int(rng.integers(1, 40))
        info = np.zeros((s_cnt, 7), np.float32)
        info[:, 0:2] = rng.uniform(0, 12.8, (s_cnt, 2))
        info[:, 6] = rng.uniform(0.05, 0.45, s_cnt)
        eng.set_stones(info)
        n = int(rng.integers(1, e + 1))
        ids = np.sort(rng.choice(np.arange(0 if case % 2 else 1, e), size=min(n, e - 1), replace=False)).astype(np.int64)
        initial = np.zeros((e, 3), np.float32)
        initial[:, 0:2] = rng.uniform(1.0, 11.8, (e, 2))
        draws = rng.random((64, len(ids))).astype(np.float32)
        want, used = orc.generate_goals(info, ids, initial, draws, radius=3.0)
        target = torch.zeros(e, 3, device=dev)
        used_d = torch.zeros(1, dtype=torch.int32, device=dev)
        eng.generate_goals(torch.from_numpy(ids).to(dev), torch.from_numpy(initial).to(dev), target, radius=3.0,
                           draws=torch.from_numpy(draws).to(dev), n_draws_used=used_d)
        assert int(used_d.item()) == used, f"case {case}: draws used {int(used_d.item())} != {used}"
        np.testing.assert_allclose(target[:, 0:2].cpu().numpy(), want[:, 0:2], rtol=1e-6, atol=1e-5, err_msg=f"case {case}")


def test_stone_mask_matches_reference_clearance():
    """The additional stone_info occupancy mask of the step (BASELINE.json configs[2]): positions = the query points
    of the reset-path golden fixture, expected = (the REFERENCE's clearance values <= margin), for the three margins the
    reference's own tests of that clearance use (0 here, 1.0 rover.py:539, 1.4 rover.py:660).  Exact except where the
    reference's clearance is within its own cdist tolerance (2e-4) of the margin.  Reward / done are untouched."""
    from hip_helpers import make_engine
    from isaac_rover_amd import synth
    fx = load_golden("reset_path")
    scene = scene_for(fx)
    distn = synth.ray_distribution("9")
    xy = torch.from_numpy(fx["xy"])
    n = xy.shape[0]
    st = synth.make_states(n, 12.8, seed=5)
    st["pos"][:, 0:2] = xy
    eng = make_engine(scene, distn, n)
    dev = eng.device
    d = {k: v.to(dev).contiguous() for k, v in st.items()}
    sin = eng.make_in(d["pos"], d["quat"], d["joints"], d["target"], d["lin_hist"], d["ang_hist"], d["euler_pre"], d["progress"])
    base = None
    for margin in (0.0, 1.0, 1.4):
        obs = torch.zeros(n, eng.num_observations, device=dev)
        rew = torch.zeros(n, device=dev)
        reset = torch.zeros(n, dtype=torch.int64, device=dev)
        rock = torch.zeros(n, dtype=torch.int64, device=dev)
        mask = torch.full((n,), -7, dtype=torch.int64, device=dev)
        sout = eng.make_out(obs, rew=rew, reset=reset, rock_collision=rock, stone_collision=mask, stone_margin=margin)
        eng.step(sin, sout, increment_progress=False)
        torch.cuda.synchronize()
        want = fx["clearance"] <= margin
        decided = np.abs(fx["clearance"] - margin) > 2e-4
        got = mask.cpu().numpy()
        assert set(np.unique(got)) <= {0, 1}
        np.testing.assert_array_equal(got[decided] != 0, want[decided])
        assert want.any() or margin == 0.0
        cur = (obs.cpu().numpy(), rew.cpu().numpy(), reset.cpu().numpy(), rock.cpu().numpy())
        if base is not None:
            for a, b in zip(base, cur):
                np.testing.assert_array_equal(a, b)
        base = cur
    # the mask needs the stone list; a margin beyond the grid's reach is refused
    bad = eng.make_out(obs, rew=rew, reset=reset, rock_collision=rock, stone_collision=mask, stone_margin=2.0)
    with pytest.raises(Exception, match="stone_margin"):
        eng.step(sin, bad, increment_progress=False)
    eng.close()


def test_reset_envs_with_zero_resets_is_a_no_op():
    """Empty input: no env flagged done -> nothing moves, no goal is redrawn (count read from device memory)."""
    from isaac_rover_amd import _lib, synth
    fx = load_golden("reset_path")
    scene = scene_for(fx)
    e = 32
    eng = _lib.Engine(e, device=0)
    eng.set_scene(scene, synth.ray_distribution("9"))
    dev = eng.device
    pos = torch.rand(e, 3, device=dev); quat = torch.rand(e, 4, device=dev); tgt = torch.rand(e, 3, device=dev)
    before = (pos.clone(), quat.clone(), tgt.clone())
    reset = torch.zeros(e, dtype=torch.int64, device=dev); progress = torch.full((e,), 7, dtype=torch.int64, device=dev)
    ids = torch.zeros(e, dtype=torch.int64, device=dev); n = torch.zeros(1, dtype=torch.int32, device=dev)
    used = torch.full((1,), -5, dtype=torch.int32, device=dev)
    eng.compact_resets(reset, ids, n)
    eng.reset_envs(ids, torch.zeros(e, 3, device=dev), pos, quat, reset, progress, n_reset_dev=n, target3=tgt, n_draws_used=used)
    torch.cuda.synchronize()
    assert int(n.item()) == 0 and int(used.item()) == 0
    for a, b in zip(before, (pos, quat, tgt)):
        assert torch.equal(a, b)
    assert bool((progress == 7).all())


def test_cell_index_mode_cuda_rcp_matches_oracle_and_differs_on_ties():
    """Option cell_index_mode: 0 = `x / 0.1` as ATen's CPU kernel divides (what the golden vectors pin), 1 = `x * (1 / 0.1)` as
    ATen's CUDA kernel evaluates the same line (camera.py:241) on the device the reference really runs on.  Both modes equal
    the oracle in the same mode bit for bit on the ray results; they differ from each other on tie envs only."""
    from conftest import tie_points, tie_scene_and_states
    from hip_helpers import hip_step, make_engine
    from oracle import oracle as orc
    scene, distn, st, tie_envs = tie_scene_and_states()
    t = orc.KnnMap(scene.terrain.map_indices, scene.terrain.triangles, scene.terrain.vertices)
    r = orc.KnnMap(scene.rocks.map_indices, scene.rocks.triangles, scene.rocks.vertices)
    n = st["pos"].shape[0]
    got, want = {}, {}
    for mode, name in ((0, "cpu_div"), (1, "cuda_rcp")):
        for variant in (3, 2, 1):
            eng = make_engine(scene, distn, n, variant=variant)
            eng.set_option("cell_index_mode", mode)
            got[(mode, variant)] = hip_step(eng, st)
            eng.close()
        try:
            orc.set_cell_index_mode(name)
            want[mode] = orc.step(t, r, st, *distn)
        finally:
            orc.set_cell_index_mode("cpu_div")
        for variant in (3, 2, 1):
            g = got[(mode, variant)]
            # tie envs: identity orientation -> no trig ulps between OCML and glibc -> the terrain ray must agree bit for bit
            np.testing.assert_array_equal(g["ray_dist"][tie_envs], want[mode]["ray_dist"][tie_envs], err_msg=f"{name} v{variant}")
            for key in ("ray_dist", "wheel_dist", "body_dist"):
                np.testing.assert_allclose(g[key], want[mode][key], rtol=1e-5, atol=1e-5, err_msg=f"{name} v{variant} {key}")
            for key in ("rock_collision", "reset_buf"):
                np.testing.assert_array_equal(g[key], want[mode][key], err_msg=f"{name} v{variant} {key}")
        np.testing.assert_array_equal(got[(mode, 2)]["ray_dist"], got[(mode, 1)]["ray_dist"])
    differs = np.nonzero((got[(0, 2)]["ray_dist"] != got[(1, 2)]["ray_dist"]).any(axis=1))[0]
    assert len(differs) >= 4 and set(differs) <= set(tie_envs)
    # heightfield lookup (rover.py:588-608): unique value per cell -> the chosen cell is visible
    n0 = 600
    hm = torch.arange(n0, dtype=torch.float32)[:, None] * 1000.0 + torch.arange(n0, dtype=torch.float32)[None, :]
    v, ia, ib = tie_points(0.025, n0)
    xy = torch.from_numpy(np.stack((v, np.full_like(v, 0.26)), axis=1)).cuda()
    eng = make_engine(scene, distn, n)
    eng.set_heightfield(hm, 0.025, 1.0)
    np.testing.assert_array_equal(eng.sample_height(xy).cpu().numpy(), ia * 1000.0 + 10.0)
    eng.set_option("cell_index_mode", 1)
    np.testing.assert_array_equal(eng.sample_height(xy).cpu().numpy(), ib * 1000.0 + 10.0)
    with pytest.raises(Exception, match="cell_index_mode"):
        eng.set_option("cell_index_mode", 2)
    eng.close()


def test_library_rng_is_keyed_by_global_env_id_and_seed():
    """Shards of a multi-GPU run must not replay each other's reset randomisation: the Philox counter is the GLOBAL env id
    (id + env_offset), the key is the caller's seed.  Two engines that differ only in env_offset draw different yaws and goal
    angles for the same local envs; the same (seed, global id) reproduces the same draw on any shard."""
    from isaac_rover_amd import _lib, synth
    fx = load_golden("reset_path")
    scene = scene_for(fx)
    e = 32

    def run(env_offset, seed):
        eng = _lib.Engine(e, device=0, num_envs_global=4 * e, env_offset=env_offset)
        eng.set_scene(scene, synth.ray_distribution("9"))
        eng.set_stones(fx["stone_info"])
        dev = eng.device
        initial = torch.from_numpy(fx["goal_initial"]).to(dev)
        pos = torch.zeros(e, 3, device=dev); quat = torch.zeros(e, 4, device=dev); tgt = torch.zeros(e, 3, device=dev)
        reset = torch.ones(e, dtype=torch.int64, device=dev); progress = torch.ones(e, dtype=torch.int64, device=dev)
        ids = torch.arange(env_offset, env_offset + e, device=dev)
        n = torch.full((1,), e, dtype=torch.int32, device=dev)
        eng.reset_envs(ids, initial, pos, quat, reset, progress, n_reset_dev=n, target3=tgt, seed=seed)
        torch.cuda.synchronize()
        out = quat.cpu().numpy().copy(), tgt.cpu().numpy().copy()
        eng.close()
        return out

    q0, t0 = run(0, 99)
    q1, t1 = run(e, 99)
    q0b, t0b = run(0, 99)
    q0c, t0c = run(0, 100)
    np.testing.assert_array_equal(q0, q0b); np.testing.assert_array_equal(t0, t0b)          # deterministic
    assert (q0 != q1).any(axis=1).mean() > 0.9 and (t0[1:, 0:2] != t1[1:, 0:2]).any(axis=1).mean() > 0.9      # other shard
    assert (q0 != q0c).any(axis=1).mean() > 0.9 and (t0[1:, 0:2] != t0c[1:, 0:2]).any(axis=1).mean() > 0.9    # other seed


def test_yaw_deg_length_is_validated():
    """yaw_deg[i] is read for every i below the reset count; with the count on the device that can be any i < num_envs."""
    from isaac_rover_amd import _lib, synth
    fx = load_golden("reset_path")
    scene = scene_for(fx)
    e = 32
    eng = _lib.Engine(e, device=0)
    eng.set_scene(scene, synth.ray_distribution("9"))
    dev = eng.device
    z3 = torch.zeros(e, 3, device=dev); q = torch.zeros(e, 4, device=dev)
    reset = torch.ones(e, dtype=torch.int64, device=dev); progress = torch.ones(e, dtype=torch.int64, device=dev)
    ids = torch.arange(e, device=dev); n = torch.full((1,), 4, dtype=torch.int32, device=dev)
    short = torch.zeros(4, dtype=torch.int32, device=dev)
    with pytest.raises(_lib.RoverError, match="yaw_deg"):
        eng.reset_envs(ids, z3, z3.clone(), q, reset, progress, n_reset_dev=n, yaw_deg=short)
    with pytest.raises(_lib.RoverError, match="yaw_deg"):
        eng.reset_envs(ids, z3, z3.clone(), q, reset, progress, n_reset_host=8, yaw_deg=short)
    # the C ABI checks too (a caller that bypasses the Python binding)
    import ctypes as C
    io = _lib.ResetIO(ids.data_ptr(), n.data_ptr(), 0, z3.data_ptr(), z3.data_ptr(), q.data_ptr(), None, None, None,
                      reset.data_ptr(), progress.data_ptr(), short.data_ptr(), None, 8.0, None, 0, 0, None, 4)
    assert eng.lib.rover_reset_envs(eng._h, C.byref(io), None) == -1
    assert b"yaw_deg" in eng.lib.rover_last_error(eng._h)
    eng.reset_envs(ids, z3, z3.clone(), q, reset, progress, n_reset_host=4, yaw_deg=short)      # long enough: fine
    torch.cuda.synchronize()
    eng.close()


def test_philox_goals_have_clearance():
    """Library RNG path: every accepted goal has clearance > 1.0 and sits `radius` from its spawn."""
    from isaac_rover_amd import _lib, synth
    fx = load_golden("reset_path")
    scene = scene_for(fx)
    eng = _lib.Engine(32, device=0)
    eng.set_scene(scene, synth.ray_distribution("9"))
    eng.set_stones(fx["stone_info"])
    dev = eng.device
    ids = torch.arange(1, 32, device=dev)
    initial = torch.from_numpy(fx["goal_initial"]).to(dev)
    target = torch.zeros(32, 3, device=dev)
    used = torch.zeros(1, dtype=torch.int32, device=dev)
    eng.generate_goals(ids, initial, target, seed=1234, max_draws=256, n_draws_used=used)
    assert int(used.item()) >= 1
    c = eng.clearance(target[ids][:, 0:2].contiguous())
    assert bool((c > 1.0).all())
    r = (target[ids][:, 0:2] - initial[ids][:, 0:2]).norm(dim=1)
    np.testing.assert_allclose(r.cpu().numpy(), 8.0, atol=1e-4)


@pytest.mark.parametrize("precision,tag", [(0, "fp32"), (2, "fp16")])
def test_get_depths_returns_the_reference_triple(precision, tag):
    """All three return values of Camera.get_depths (camera.py:145: distances, "intersection points" sources - d k of
    ray_casting.py:63, ray sources) against the reference's own (tests/golden/get_depths_e64_p37.npz): `rover_get_depths` on the
    reference's (positions, euler rotations), and the optional `ray_src` / `hit_pt` outputs of `rover_step`.  As shipped (fp16
    tensors): bit for bit, every ray."""
    from hip_helpers import make_engine
    fx = load_golden("get_depths_e64_p37")
    scene = scene_for(fx)
    distn = (fx["distribution"], fx["sparse_idx"], fx["dense_idx"])
    e, p_n = fx["in_pos"].shape[0], fx["distribution"].shape[0]
    want_d, want_pt, want_src = fx[f"out_{tag}_dist"], fx[f"out_{tag}_pt"], fx[f"out_{tag}_src"]
    for variant in (4, 3, 2, 1) if precision == 0 else (4, 3, 2):
        eng = make_engine(scene, distn, e, variant=variant)
        eng.set_option("ray_precision", precision)
        dev = eng.device
        d, pt, src = eng.get_depths(torch.from_numpy(fx["in_pos"]).to(dev), torch.from_numpy(fx["in_euler"]).to(dev))
        # the step's optional outputs (euler from the quaternion on the device)
        st = states_of(load_golden("step_e64_p37_fp32" if precision == 0 else "step_e64_p37_fp16_as_shipped"))
        np.testing.assert_array_equal(st["pos"].numpy(), fx["in_pos"])
        dd = {k: v.to(dev).contiguous() for k, v in st.items()}
        sin = eng.make_in(dd["pos"], dd["quat"], dd["joints"], dd["target"], dd["lin_hist"], dd["ang_hist"], dd["euler_pre"], dd["progress"].clone())
        obs = torch.zeros(e, eng.num_observations, device=dev)
        rsrc = torch.zeros(e, p_n, 3, device=dev); hpt = torch.zeros(e, p_n, 3, device=dev); rd = torch.zeros(e, p_n, device=dev)
        sout = eng.make_out(obs, rew=torch.zeros(e, device=dev), reset=torch.ones(e, dtype=torch.int64, device=dev),
                            rock_collision=torch.zeros(e, dtype=torch.int64, device=dev), ray_dist=rd, ray_src=rsrc, hit_pt=hpt)
        eng.step(sin, sout)
        torch.cuda.synchronize()
        for label, (gd, gp, gs) in (("get_depths", (d, pt, src)), ("step", (rd, hpt, rsrc))):
            gd, gp, gs = gd.cpu().numpy(), gp.cpu().numpy(), gs.cpu().numpy()
            if precision == 2:
                np.testing.assert_array_equal(gs, want_src, err_msg=f"{label} sources v{variant}")
                np.testing.assert_array_equal(gd, want_d, err_msg=f"{label} distances v{variant}")
                np.testing.assert_array_equal(gp, want_pt, err_msg=f"{label} points v{variant}")
            else:
                np.testing.assert_allclose(gs, want_src, rtol=1e-6, atol=2e-6, err_msg=f"{label} sources v{variant}")
                close = np.abs(gd - want_d) <= 2e-3
                assert close.mean() >= 0.999, f"{label} distances v{variant}"
                np.testing.assert_allclose(gp[close], want_pt[close], rtol=0, atol=2.5e-3, err_msg=f"{label} points v{variant}")
        eng.close()
    assert (want_d < 11.0).mean() > 0.5 and (want_d == 11.0).any()          # hits and misses (k = 11: the point 11 m along the ray)


@pytest.mark.parametrize("precision,name", [(0, "step_e64_p37_fp32"), (2, "step_e64_p37_fp16_as_shipped"),
                                            (0, "step_irregular_p37_fp32"), (2, "step_irregular_p37_fp16_as_shipped")])
def test_get_collisions_matches_the_reference(precision, name):
    """Rock_Detection.get_collisions (rock_detect.py:52-149) as its own C-ABI call: `rover_get_collisions` on the reference's
    (positions, euler rotations, joint positions) against the reference's own wheel / body distances of the same fixture (the
    values its get_observations handed to check_collision, rover.py:291).  As shipped (fp16 tensors): bit for bit."""
    from hip_helpers import make_engine
    fx = load_golden(name)
    scene = scene_for(fx)
    distn = (fx["distribution"], fx["sparse_idx"], fx["dense_idx"])
    e = fx["in_pos"].shape[0]
    for variant in (4, 3, 2, 1) if precision == 0 else (4, 3, 2):
        eng = make_engine(scene, distn, e, variant=variant)
        eng.set_option("ray_precision", precision)
        dev = eng.device
        wheel, body = eng.get_collisions(torch.from_numpy(fx["in_pos"]).to(dev), torch.from_numpy(fx["out_euler"]).to(dev),
                                         torch.from_numpy(fx["in_joints"]).to(dev))
        torch.cuda.synchronize()
        wheel, body = wheel.cpu().numpy(), body.cpu().numpy()
        assert wheel.shape == (e, 24) and body.shape == (e, 2)
        if precision == 2:
            np.testing.assert_array_equal(wheel, fx["out_wheel_dist"], err_msg=f"wheel v{variant}")
            np.testing.assert_array_equal(body, fx["out_body_dist"], err_msg=f"body v{variant}")
        else:
            for got, want in ((wheel, fx["out_wheel_dist"]), (body, fx["out_body_dist"])):
                d = np.abs(got.astype(np.float64) - want.astype(np.float64))
                assert float((d > TOL_RAY).mean()) <= RAY_FLIP_BUDGET, f"v{variant}: {(d > TOL_RAY).mean():.4%} of rays differ"
        eng.close()
    assert (fx["out_wheel_dist"] < 11.0).any()


@pytest.mark.parametrize("precision,name", [(0, "step_e64_p37_fp32"), (2, "step_e64_p37_fp16_as_shipped"),
                                            (0, "step_irregular_p37_fp32"), (2, "step_irregular_p37_fp16_as_shipped"),
                                            (0, "step_e32_p37_k200_fp32"), (0, "step_lattice_fp32")])
def test_ray_phase_on_exported_and_supplied_rays(precision, name):
    """`rover_export_rays` / `rover_cast_rays`: the device's own rays of a step through the oracle's per-ray arithmetic
    (ray_casting.py:34-59, cell lookup camera.py:233-264, min over K) give the device's distances BIT FOR BIT in every ray-cast kernel
    and either arithmetic — the comparison that does not depend on the pose trigonometry; and the same rays handed back through
    `rover_cast_rays` reproduce them."""
    from hip_helpers import hip_step, make_engine
    from oracle import oracle as orc
    fx = load_golden(name)
    scene = scene_for(fx)
    distn = (fx["distribution"], fx["sparse_idx"], fx["dense_idx"])
    st = states_of(fx)
    e = st["pos"].shape[0]
    t = orc.KnnMap(scene.terrain.map_indices, scene.terrain.triangles, scene.terrain.vertices)
    r = orc.KnnMap(scene.rocks.map_indices, scene.rocks.triangles, scene.rocks.vertices)
    for variant in (4, 3, 2, 1) if precision == 0 else (4, 3, 2):
        eng = make_engine(scene, distn, e, variant=variant)
        eng.set_option("ray_precision", precision)
        got = hip_step(eng, st)
        src, dirs, cell, dist = eng.export_rays()
        torch.cuda.synchronize()
        s_, d_, dist_ = src.cpu().numpy(), dirs.cpu().numpy(), dist.cpu().numpy()
        np.testing.assert_array_equal(dist_[:, 26:], got["ray_dist"])
        np.testing.assert_array_equal(dist_[:, :24], got["wheel_dist"])
        want = np.concatenate((orc.raycast_unit(r, s_[:, :26], d_[:, :26], half=precision == 2).reshape(e, 26),
                               orc.raycast_unit(t, s_[:, 26:], d_[:, 26:], half=precision == 2).reshape(e, -1)), axis=1)
        np.testing.assert_array_equal(dist_, want, err_msg=f"{name} v{variant}")          # (IEEE equality: NaN == NaN here, +0 == -0)
        again = eng.cast_rays(src, dirs)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(again.cpu().numpy().view(np.uint32), dist_.view(np.uint32), err_msg=f"{name} v{variant} cast_rays")
        # supplied rays that are not a step's: every ray of env 0 given to every env, shifted by the env's index in x
        s2 = src[0:1].repeat(e, 1, 1).clone(); d2 = dirs[0:1].repeat(e, 1, 1).clone()
        s2[:, :, 0] += (torch.arange(e, device=src.device, dtype=torch.float32) * 0.0371)[:, None]
        if precision == 2:
            s2 = s2.half().float()
        got2 = eng.cast_rays(s2.contiguous(), d2.contiguous()).cpu().numpy()
        s2n, d2n = s2.cpu().numpy(), d2.cpu().numpy()
        want2 = np.concatenate((orc.raycast_unit(r, s2n[:, :26], d2n[:, :26], half=precision == 2).reshape(e, 26),
                                orc.raycast_unit(t, s2n[:, 26:], d2n[:, 26:], half=precision == 2).reshape(e, -1)), axis=1)
        np.testing.assert_array_equal(got2, want2, err_msg=f"{name} v{variant} supplied rays")
        eng.close()


def test_quat_to_euler_and_ackermann():
    from isaac_rover_amd import _lib
    fx = load_golden("step_e64_p37_fp32")
    eng = _lib.Engine(64, device=0)
    q = torch.from_numpy(fx["in_quat"]).to(eng.device)
    np.testing.assert_allclose(eng.quat_to_euler(q).cpu().numpy(), fx["out_euler"], rtol=1e-5, atol=1e-5)
    ak = load_golden("ackermann")
    steer, vel = eng.ackermann(torch.from_numpy(ak["lin"]).to(eng.device), torch.from_numpy(ak["ang"]).to(eng.device))
    np.testing.assert_allclose(steer.cpu().numpy(), ak["steer"], rtol=1e-5, atol=1e-5, equal_nan=True)
    np.testing.assert_allclose(vel.cpu().numpy(), ak["vel"], rtol=1e-5, atol=1e-4, equal_nan=True)


def test_errors_are_loud():
    from isaac_rover_amd import _lib
    eng = _lib.Engine(8, device=0)
    with pytest.raises(_lib.RoverError, match="maps must be set"):
        z = torch.zeros(8, 3, device=eng.device)
        sin = eng.make_in(z, torch.zeros(8, 4, device=eng.device), torch.zeros(8, 13, device=eng.device), z, z, z, z,
                          torch.zeros(8, dtype=torch.int64, device=eng.device))
        eng.Ns = 1
        sout = eng.make_out(torch.zeros(8, 5, device=eng.device))
        eng.step(sin, sout)
    with pytest.raises(_lib.RoverError):
        _lib.Engine(0, device=0)


def _knn_brute_force(verts, tris, n_x, n_y, res, k):
    """numpy restatement of rover_utils.py:68-108 with the builder's stated ranking (f32 squared distance, ties by id)."""
    v = verts.astype(np.float32)
    cx = (v[tris[:, 0], 0] + v[tris[:, 1], 0] + v[tris[:, 2], 0]) / np.float32(3)
    cy = (v[tris[:, 0], 1] + v[tris[:, 1], 1] + v[tris[:, 2], 1]) / np.float32(3)
    out = np.zeros((n_x, n_y, k), np.int32)
    ids = np.arange(len(tris), dtype=np.uint64)
    for x in range(n_x):
        px = np.float32(x) * np.float32(res)
        dx = cx - px
        for y in range(n_y):
            dy = cy - np.float32(y) * np.float32(res)
            d2 = (dx * dx + dy * dy).astype(np.float32)
            key = (d2.view(np.uint32).astype(np.uint64) << np.uint64(32)) | ids
            out[x, y] = np.sort(key)[:k].astype(np.uint64) & np.uint64(0xffffffff)
    return out


@pytest.mark.parametrize("n_tri,n_x,n_y,k,res,extent", [(3000, 24, 20, 16, 0.1, 2.4), (900, 16, 16, 200, 0.1, 1.5),
                                                       (5000, 30, 30, 7, 0.05, 3.0), (64, 8, 8, 64, 0.1, 0.8)])
def test_knn_builder_matches_brute_force(n_tri, n_x, n_y, k, res, extent):
    """rover_build_knn_map (bucketed ring search + LDS bitonic sort) against brute force over every centroid, on an
    irregular random triangle soup whose bounding box is larger AND smaller than the map in different cases."""
    from isaac_rover_amd import _lib
    rng = np.random.default_rng(n_tri)
    centers = rng.uniform(-0.3, extent, (n_tri, 1, 2))
    verts = np.concatenate([centers + rng.normal(0, 0.05, (n_tri, 3, 2)), rng.normal(0, 0.1, (n_tri, 3, 1))], axis=2)
    verts = verts.reshape(-1, 3).astype(np.float32)
    tris = np.arange(3 * n_tri, dtype=np.int32).reshape(n_tri, 3)
    eng = _lib.Engine(8, device=0)
    got = eng.build_knn_map(verts, tris, n_x, n_y, res, k).cpu().numpy()
    want = _knn_brute_force(verts, tris, n_x, n_y, res, k)
    np.testing.assert_array_equal(got, want)
    with pytest.raises(_lib.RoverError, match="fewer than K"):
        eng.build_knn_map(verts[:30], tris[:10], 4, 4, res, 11)
    eng.close()


def test_built_map_drives_the_step():
    """Mesh -> rover_build_knn_map -> set_knn_map -> step: same result as the oracle fed the same built map."""
    from hip_helpers import hip_step
    from isaac_rover_amd import _lib, assets, synth
    from oracle import oracle as orc
    verts, tris, _ = synth.grid_mesh(41, seed=3)
    eng = _lib.Engine(64, device=0)
    terrain = assets.build_knn_map(eng, verts, tris, n_cells=40, res=0.1, k=20)
    scene = synth.make_scene(n_cells=40, k=20, n_stones=6, seed=3)
    scene.terrain = terrain
    distn = synth.ray_distribution("37")
    eng.set_scene(scene, distn)
    st = synth.make_states(64, 4.0, seed=2)
    got = hip_step(eng, st)
    t = orc.KnnMap(terrain.map_indices, terrain.triangles, terrain.vertices)
    r = orc.KnnMap(scene.rocks.map_indices, scene.rocks.triangles, scene.rocks.vertices)
    want = orc.step(t, r, st, *distn)
    assert_step_close(got, {"out_" + k: v for k, v in want.items()}, "built-map")
    assert (got["ray_dist"] < 11.0).mean() > 0.5


def _write_binary_ply(path, verts, tris):
    with open(path, "wb") as fh:
        fh.write(f"ply\nformat binary_little_endian 1.0\nelement vertex {len(verts)}\nproperty float x\nproperty float y\n"
                 f"property float z\nelement face {len(tris)}\nproperty list uchar int vertex_indices\nend_header\n".encode())
        fh.write(np.asarray(verts, "<f4").tobytes())
        rec = np.zeros(len(tris), dtype=[("n", "u1"), ("i", "<i4", 3)])
        rec["n"] = 3
        rec["i"] = tris
        fh.write(rec.tobytes())


def test_generate_knn_triangles_from_ply(tmp_path):
    """f-3 end to end: map.ply + big_stones.ply -> knn_terrain/ + knn_rocks/ in the reference's on-disk format
    (rover_utils.py:48-50,113-118) -> load_reference_assets -> a step equal to the one on the maps built in memory."""
    from hip_helpers import hip_step
    from isaac_rover_amd import _lib, assets, synth
    verts, tris, _ = synth.grid_mesh(41, seed=4)
    rock_tris = tris[::5]
    terrain_dir = tmp_path / "tasks" / "utils" / "terrain"
    terrain_dir.mkdir(parents=True)
    _write_binary_ply(terrain_dir / "map.ply", verts, tris)
    _write_binary_ply(terrain_dir / "big_stones.ply", verts, rock_tris)
    eng = _lib.Engine(32, device=0)
    maps = assets.generate_knn_triangles(eng, str(terrain_dir), res_x=40, res_y=40, res=0.1, n_triangles=24)
    raw = torch.load(terrain_dir / "knn_terrain" / "map_indices.pt")
    assert tuple(raw.shape) == (24, 40, 40) and raw.dtype == torch.int32                       # [K, X, Y] like :108,116
    assert torch.load(terrain_dir / "knn_rocks" / "vertices.pt").dtype == torch.float16
    base = synth.make_scene(n_cells=40, k=24, n_stones=6, seed=4)
    np.save(terrain_dir / "stone_info.npy", base.stone_info_raw)
    torch.save(base.heightmap, terrain_dir / "heightmap_tensor.pt")
    scene = assets.load_reference_assets(str(tmp_path))
    assert torch.equal(scene.terrain.map_indices, maps["knn_terrain"].map_indices)
    assert torch.equal(scene.rocks.triangles, torch.as_tensor(rock_tris, dtype=torch.int32))
    distn = synth.ray_distribution("9")
    st = synth.make_states(32, 4.0, seed=6)
    eng.set_scene(scene, distn)
    a = hip_step(eng, st)
    mem = synth.Scene(terrain=maps["knn_terrain"], rocks=maps["knn_rocks"], stone_info_raw=base.stone_info_raw, heightmap=base.heightmap)
    eng2 = _lib.Engine(32, device=0)
    eng2.set_scene(mem, distn)
    b = hip_step(eng2, st)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert (a["ray_dist"] < 11.0).mean() > 0.5
    eng.close(); eng2.close()


def _torch_net_reference(net, states):
    """fp32 PyTorch restatement of model.py:185-195 on the same weights."""
    import torch.nn.functional as F
    act = {"leakyrelu": lambda v: F.leaky_relu(v, 0.01), "tanh": torch.tanh, None: lambda v: v, "relu": F.relu, "elu": F.elu}
    p, ns, nd = net.num_proprioception, net.num_sparse, net.num_dense
    def run(layers, x):
        for l in layers:
            x = act[l.activation](F.linear(x, l.weight, l.bias))
        return x
    x0 = run(net.encoder0, states[:, p:p + ns])
    x1 = run(net.encoder1, states[:, p + ns:p + ns + nd])
    return run(net.network, torch.cat((states[:, 0:p], x0, x1), dim=1))


@pytest.mark.parametrize("num_envs,ns,nd", [(300, 634, 1112), (4096, 37, 0 + 5), (1, 9, 3), (65536 + 100, 37, 5)])
def test_policy_forward_matches_torch_fp32(num_envs, ns, nd):
    """f-4: actor and critic forward (f32 MFMA linear layers reading obs slices in place) vs a PyTorch fp32 reference."""
    from isaac_rover_amd import _lib
    from isaac_rover_amd.learning.model import HeightmapNet
    eng = _lib.Engine(max(num_envs, 1), device=0)
    w = 4 + ns + nd
    g = torch.Generator().manual_seed(1)
    states = (torch.rand(num_envs, w, generator=g) * 4 - 1).cuda()
    prev = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    try:
        for outputs, head in ((2, "tanh"), (1, None)):
            net = HeightmapNet(eng, w, ns, nd, outputs, head, device="cuda:0", seed=outputs)
            want = _torch_net_reference(net, states)
            for fused in (False, True):        # one launch per layer / one fused chain kernel per encoder and for MLP + head
                got = net.compute(states, fused=fused)
                torch.cuda.synchronize()
                assert got.shape == (num_envs, outputs)
                np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-4, atol=2e-5, err_msg=f"fused={fused}")
    finally:
        torch.backends.cuda.matmul.allow_tf32 = prev
    sd = net.state_dict()
    assert "encoder0.encoder.0.layer.0.weight" in sd and "network.3.bias" in sd        # the reference's parameter names
    eng.close()


@pytest.mark.parametrize("mlp", [(64, 48, 32), (100, 37, 20), (128, 160, 128), (129, 16, 5)])
def test_fused_chain_narrow_and_odd_widths(mlp):
    """The 4-layer chain computes its first layer in two halves of 128 features: a first hidden width below 128 leaves the second
    half empty (no bias read past the tensor, every value 0).  The bias tensors sit at the END of NaN-filled buffers, so a read past
    them returns NaN and poisons the row.  Fused vs one launch per layer vs PyTorch fp32."""
    from isaac_rover_amd import _lib
    from isaac_rover_amd.learning.model import HeightmapNet
    eng = _lib.Engine(8, device=0)
    ns, nd, rows = 37, 0, 300
    w = 4 + ns + nd
    g = torch.Generator().manual_seed(9)
    states = (torch.rand(rows, w, generator=g) * 4 - 1).cuda()
    net = HeightmapNet(eng, w, ns, nd, 2, "tanh", mlp_features=mlp, encoder_features=(50, 33), device="cuda:0", seed=4)
    for layer in net.encoder0 + net.encoder1 + net.network:      # bias as the tail of a NaN buffer: bias[n + i] is NaN
        n = layer.bias.numel()
        if n == 0:
            continue
        buf = torch.full((n + 512,), float("nan"), device="cuda:0")
        buf[:n] = layer.bias
        layer.bias = buf[:n]
        layer._keep = buf
    assert eng.chain_fits(net.network) and eng.chain_fits(net.encoder0)
    want = _torch_net_reference(net, states)
    for fused in (False, True):
        got = net.compute(states, fused=fused)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(got).all()), f"fused={fused}: non-finite output (a read past a bias tensor)"
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-4, atol=2e-5, err_msg=f"fused={fused}")
    eng.close()


@pytest.mark.parametrize("act", ["none", "leakyrelu", "tanh", "relu", "elu"])
def test_linear_forward_shapes_and_activations(act):
    from isaac_rover_amd import _lib
    import torch.nn.functional as F
    eng = _lib.Engine(8, device=0)
    g = torch.Generator().manual_seed(3)
    # m >= 65 536 takes the 128-row workgroups (all columns per workgroup), smaller m the 32 x 32 tiles
    for m, k, n in ((257, 33, 1), (64, 634, 80), (130, 5, 256), (31, 1112, 80), (65536 + 77, 40, 100), (65536, 70, 256)):
        wide = torch.randn(m, k + 9, generator=g).cuda()
        x = wide[:, 4:4 + k]
        w = (torch.randn(n, k, generator=g) / k ** 0.5).cuda()
        b = torch.randn(n, generator=g).cuda()
        out_wide = torch.full((m, n + 3), 7.0).cuda()
        eng.linear_forward(x, w, b, act, out_wide[:, 1:1 + n])
        want = F.linear(x.double(), w.double(), b.double())
        want = {"none": want, "leakyrelu": F.leaky_relu(want, 0.01), "tanh": torch.tanh(want), "relu": F.relu(want), "elu": F.elu(want)}[act]
        torch.cuda.synchronize()
        np.testing.assert_allclose(out_wide[:, 1:1 + n].cpu().numpy(), want.float().cpu().numpy(), rtol=1e-4, atol=1e-5)
        assert bool((out_wide[:, 0] == 7.0).all()) and bool((out_wide[:, 1 + n:] == 7.0).all())
    eng.close()
