"""GPU (-m gpu): error behaviour of the C ABI — every misuse is refused with a negative code and a message,
nothing faults, and contexts can be created / destroyed repeatedly."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(n=16):
    from isaac_rover_amd import _lib, synth
    scene = synth.make_scene(n_cells=32, k=8, n_stones=4)
    eng = _lib.Engine(n, device=0)
    eng.set_scene(scene, synth.ray_distribution("9"))
    return eng, scene


def test_missing_tables_are_reported():
    from isaac_rover_amd import _lib, synth
    eng = _lib.Engine(8, device=0)
    dev = eng.device
    with pytest.raises(_lib.RoverError, match="rover_set_stones"):
        eng.clearance(torch.zeros(4, 2, device=dev))
    with pytest.raises(_lib.RoverError, match="rover_set_heightfield"):
        eng.sample_height(torch.zeros(4, 2, device=dev))
    with pytest.raises(_lib.RoverError, match="set_distribution|index"):
        eng.set_distribution(np.zeros((3, 3)), [0, 5], [])
    with pytest.raises(_lib.RoverError, match="bad shape"):
        eng.set_knn_map(0, np.zeros((4, 4, 2), np.int32), np.zeros((1, 3), np.int32), np.zeros((3, 3), np.float16), cell_size=0.0)
    eng.close()


def test_wrong_tensor_shapes_are_refused_on_the_host():
    from isaac_rover_amd import _lib
    eng, _ = _engine(16)
    dev = eng.device
    z3 = torch.zeros(16, 3, device=dev)
    with pytest.raises(_lib.RoverError, match="quat"):
        eng.make_in(z3, torch.zeros(16, 3, device=dev), torch.zeros(16, 13, device=dev), z3, z3, z3, z3,
                    torch.zeros(16, dtype=torch.int64, device=dev))
    with pytest.raises(_lib.RoverError, match="progress"):
        eng.make_in(z3, torch.zeros(16, 4, device=dev), torch.zeros(16, 13, device=dev), z3, z3, z3, z3,
                    torch.zeros(16, dtype=torch.int32, device=dev))
    with pytest.raises(_lib.RoverError, match="obs"):
        eng.make_out(torch.zeros(16, 5, device=dev))
    with pytest.raises(_lib.RoverError, match="expected a tensor on"):
        eng.make_in(torch.zeros(16, 3), torch.zeros(16, 4, device=dev), torch.zeros(16, 13, device=dev), z3, z3, z3, z3,
                    torch.zeros(16, dtype=torch.int64, device=dev))
    eng.close()


def test_get_depths_argument_checks():
    """rover_get_depths: null poses and a ctx without its tables are refused; wrong shapes are refused on the host; the call leaves
    the observation state of the ctx (euler / heading of the last step) alone."""
    from isaac_rover_amd import _lib, synth
    eng, scene = _engine(16)
    lib, dev = eng.lib, eng.device
    assert lib.rover_get_depths(eng._h, None, None, None, None, None, None) == -1
    assert b"positions and rotations" in lib.rover_last_error(eng._h)
    with pytest.raises(_lib.RoverError, match="rotations"):
        eng.get_depths(torch.zeros(16, 3, device=dev), torch.zeros(16, 4, device=dev))
    bare = _lib.Engine(16, device=0)
    with pytest.raises(_lib.RoverError, match="maps must be set"):
        bare.get_depths(torch.zeros(16, 3, device=dev), torch.zeros(16, 3, device=dev))
    bare.close()
    # a step, then get_depths on OTHER poses, then the metrics of the step: the heading the metrics read is still the step's
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from hip_helpers import hip_step
    st = synth.make_states(16, 3.2, seed=4)
    a = hip_step(eng, st, fused=False)
    d = {k: v.to(dev).contiguous() for k, v in st.items()}
    sin = eng.make_in(d["pos"], d["quat"], d["joints"], d["target"], d["lin_hist"], d["ang_hist"], d["euler_pre"], d["progress"].clone())
    obs = torch.zeros(16, eng.num_observations, device=dev)
    rew = torch.zeros(16, device=dev); rock = torch.zeros(16, dtype=torch.int64, device=dev)
    sout = eng.make_out(obs, rew=rew, reset=torch.ones(16, dtype=torch.int64, device=dev), rock_collision=rock)
    eng.get_observations(sin, sout)
    dist, pts, src = eng.get_depths(torch.zeros(16, 3, device=dev) + 1.0, torch.zeros(16, 3, device=dev))
    eng.calculate_metrics(sin, sout)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(rew.cpu().numpy(), a["rew_buf"])
    assert dist.shape == (16, 9) and pts.shape == (16, 9, 3) and torch.isfinite(dist).all()
    np.testing.assert_allclose(src[:, :, 2].cpu().numpy(), 1.0 - 0.26878, atol=1e-6)      # identity pose: the distribution's z below the rover
    eng.close()


def test_get_collisions_argument_checks_and_state():
    """rover_get_collisions: null poses / a bare ctx / wrong shapes are refused; NULL joints mean zero joints; and neither it nor
    rover_get_depths stands in for rover_get_observations: rover_calculate_metrics on a ctx that only ran them is refused."""
    from isaac_rover_amd import _lib
    eng, scene = _engine(16)
    lib, dev = eng.lib, eng.device
    assert lib.rover_get_collisions(eng._h, None, None, None, None, None, None) == -1
    assert b"positions and rotations" in lib.rover_last_error(eng._h)
    with pytest.raises(_lib.RoverError, match="joints"):
        eng.get_collisions(torch.zeros(16, 3, device=dev), torch.zeros(16, 3, device=dev), torch.zeros(16, 12, device=dev))
    bare = _lib.Engine(16, device=0)
    with pytest.raises(_lib.RoverError, match="maps must be set"):
        bare.get_collisions(torch.zeros(16, 3, device=dev), torch.zeros(16, 3, device=dev))
    bare.close()
    pos = torch.zeros(16, 3, device=dev) + 1.5
    w0, b0 = eng.get_collisions(pos, torch.zeros(16, 3, device=dev))
    w1, b1 = eng.get_collisions(pos, torch.zeros(16, 3, device=dev), torch.zeros(16, 13, device=dev))
    torch.cuda.synchronize()
    assert torch.equal(w0, w1) and torch.equal(b0, b1) and w0.shape == (16, 24) and b0.shape == (16, 2)
    # no observation yet on this ctx: the metrics must refuse (their heading state would be the memset's zeros)
    eng.get_depths(pos, torch.zeros(16, 3, device=dev))
    sin, sout = _lib.StepIn(), _lib.StepOut()
    assert lib.rover_calculate_metrics(eng._h, C.byref(sin), C.byref(sout), None) == -2
    assert b"rover_get_observations first" in lib.rover_last_error(eng._h)
    eng.close()


def test_cast_rays_refuses_directions_that_are_not_unit_length_where_the_proofs_need_them():
    """rover_cast_rays takes the directions as they are.  The culled / staged ray cast (variants 3, 4) prove their rejections for
    |d|^2 <= 1.00001 — what -normalize() gives —: a finite direction of another length is ROVER_E_INVALID there, not a silent difference;
    the every-triangle kernels (1, 2) take it; a NaN direction (keeps all its candidates by itself) is no error anywhere."""
    from isaac_rover_amd import _lib
    eng, scene = _engine(16)
    dev = eng.device
    pos = torch.zeros(16, 3, device=dev) + 1.5
    eng.get_depths(pos, torch.zeros(16, 3, device=dev))
    src, dirs, cell, dist = eng.export_rays()
    torch.cuda.synchronize()
    long = dirs.clone(); long[3, 30] *= 1.01
    nan = dirs.clone(); nan[5, 7, 1] = float("nan")
    results = {}
    for variant in (1, 2, 3, 4):
        eng.set_option("raycast_variant", variant)
        base = eng.cast_rays(src, dirs).clone()
        with_nan = eng.cast_rays(src, nan).clone()
        results[variant] = (base, with_nan)
        if variant >= 3:
            with pytest.raises(_lib.RoverError, match="not of unit length"):
                eng.cast_rays(src, long)
            again = eng.cast_rays(src, dirs)                  # the ctx is usable after the refusal
            torch.cuda.synchronize()
            assert torch.equal(again, base)
        else:
            results[variant] += (eng.cast_rays(src, long).clone(),)
    torch.cuda.synchronize()
    for variant in (2, 3, 4):
        assert torch.equal(results[variant][0], results[1][0])
        assert torch.equal(results[variant][1].view(torch.int32), results[1][1].view(torch.int32))
    assert torch.equal(results[2][2], results[1][2])
    eng.close()


def test_c_level_argument_checks():
    from isaac_rover_amd import _lib
    eng, _ = _engine(16)
    lib = eng.lib
    assert lib.rover_step(eng._h, None, None, 0, None) == -1
    assert b"null struct" in lib.rover_last_error(eng._h)
    sin, sout = _lib.StepIn(), _lib.StepOut()
    assert lib.rover_step(eng._h, C.byref(sin), C.byref(sout), 0, None) == -1          # null pointers inside
    assert lib.rover_set_option(eng._h, b"no_such_option", 1) == -1
    assert lib.rover_set_option(eng._h, b"raycast_variant", 9) == -1
    assert lib.rover_replay_raycast(eng._h, None) == -2                                   # no step yet
    assert lib.rover_calculate_metrics(eng._h, C.byref(sin), C.byref(sout), None) == -2   # get_observations first
    assert lib.rover_create(None, None) == -1
    cfg = _lib.Cfg(num_envs=4, device=99)
    h = C.c_void_p()
    assert lib.rover_create(C.byref(cfg), C.byref(h)) == -1 and b"out of range" in lib.rover_last_error(None)
    eng.close()


def test_obs_row_stride_and_create_destroy_loop():
    """obs may be a strided view (e.g. the front columns of a wider learner buffer); contexts do not leak."""
    from hip_helpers import hip_step
    from isaac_rover_amd import _lib, synth
    eng, scene = _engine(16)
    st = synth.make_states(16, 3.2, seed=3)
    want = hip_step(eng, st)["obs_buf"]
    dev = eng.device
    wide = torch.full((16, eng.num_observations + 7), -7.0, device=dev)
    d = {k: v.to(dev) for k, v in st.items()}
    sin = eng.make_in(d["pos"], d["quat"], d["joints"], d["target"], d["lin_hist"], d["ang_hist"], d["euler_pre"], d["progress"].clone())
    sout = eng.make_out(wide[:, : eng.num_observations], rew=torch.zeros(16, device=dev),
                        reset=torch.zeros(16, dtype=torch.int64, device=dev),
                        rock_collision=torch.zeros(16, dtype=torch.int64, device=dev))
    eng.step(sin, sout)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(wide[:, : eng.num_observations].cpu().numpy(), want)
    assert bool((wide[:, eng.num_observations:] == -7.0).all())
    eng.close()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(20):
        e2, _ = _engine(64)
        e2.close()
    assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20)


def test_mlp_chain_forward_shapes_and_errors():
    """rover_mlp_chain_forward: layer widths outside the built tile shapes, wrong depths and mismatched outputs are refused;
    odd sizes inside them (K0 not a multiple of 32, M not a multiple of 128, narrow layers) match PyTorch."""
    import torch.nn.functional as F
    from isaac_rover_amd import _lib
    from isaac_rover_amd.learning.model import Layer
    eng = _lib.Engine(8, device=0)
    g = torch.Generator().manual_seed(7)
    x = (torch.rand(301, 45, generator=g) * 2 - 1).cuda()
    ok2 = [Layer(45, 70, "leakyrelu", generator=g), Layer(70, 33, "tanh", generator=g)]
    out = torch.empty(301, 33, device="cuda")
    eng.chain_forward(x, ok2, out)
    want = torch.tanh(F.linear(F.leaky_relu(F.linear(x, ok2[0].weight, ok2[0].bias)), ok2[1].weight, ok2[1].bias))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), want.cpu().numpy(), rtol=2e-4, atol=2e-5)
    ok4 = [Layer(45, 200, None, generator=g), Layer(200, 130, "relu", generator=g), Layer(130, 100, "leakyrelu", generator=g),
           Layer(100, 3, "elu", generator=g)]
    out4 = torch.empty(301, 3, device="cuda")
    eng.chain_forward(x, ok4, out4)
    h = F.linear(x, ok4[0].weight, ok4[0].bias)
    h = F.relu(F.linear(h, ok4[1].weight, ok4[1].bias))
    h = F.leaky_relu(F.linear(h, ok4[2].weight, ok4[2].bias))
    want4 = F.elu(F.linear(h, ok4[3].weight, ok4[3].bias))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out4.cpu().numpy(), want4.cpu().numpy(), rtol=2e-4, atol=2e-5)
    assert eng.chain_fits(ok2) and eng.chain_fits(ok4)
    # a column slice at an odd float offset with an odd row stride: the 16-byte staging loads are only 4-byte aligned
    big = (torch.rand(301, 51, generator=g) * 2 - 1).cuda()
    xs = big[:, 3:48]
    out_s = torch.empty(301, 33, device="cuda")
    eng.chain_forward(xs, ok2, out_s)
    want_s = torch.tanh(F.linear(F.leaky_relu(F.linear(xs, ok2[0].weight, ok2[0].bias)), ok2[1].weight, ok2[1].bias))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out_s.cpu().numpy(), want_s.cpu().numpy(), rtol=2e-4, atol=2e-5)
    ok2b = [Layer(45, 90, "elu", generator=g), Layer(90, 64, "relu", generator=g)]             # the <= 96 -> <= 64 shape, full last tile
    out2b = torch.empty(301, 64, device="cuda")
    eng.chain_forward(x, ok2b, out2b)
    want2b = F.relu(F.linear(F.elu(F.linear(x, ok2b[0].weight, ok2b[0].bias)), ok2b[1].weight, ok2b[1].bias))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out2b.cpu().numpy(), want2b.cpu().numpy(), rtol=2e-4, atol=2e-5)
    # the 4-layer kernel is built for none / LeakyReLU / ReLU on its hidden layers and <= 16 outputs: other nets run layer by layer
    tanh_hidden = [Layer(45, 200, "tanh", generator=g)] + ok4[1:]
    wide_head = ok4[:3] + [Layer(100, 20, None, generator=g)]
    assert not eng.chain_fits(tanh_hidden) and not eng.chain_fits(wide_head)
    with pytest.raises(_lib.RoverError, match="tile shapes"):
        eng.chain_forward(x, tanh_hidden, out4)
    with pytest.raises(_lib.RoverError, match="tile shapes"):
        eng.chain_forward(x, wide_head, torch.empty(301, 20, device="cuda"))
    wide = [Layer(45, 120, "relu", generator=g), Layer(120, 33, "relu", generator=g)]          # 120 > 96: no built shape
    assert not eng.chain_fits(wide)
    with pytest.raises(_lib.RoverError, match="tile shapes"):
        eng.chain_forward(x, wide, out)
    with pytest.raises(_lib.RoverError, match="layers"):
        eng.chain_forward(x, ok4[:3], torch.empty(301, 100, device="cuda"))                   # 3 layers: not a built depth
    with pytest.raises(_lib.RoverError, match="out must be"):
        eng.chain_forward(x, ok2, torch.empty(301, 34, device="cuda"))
    eng.close()


def test_staged_tables_option_and_explicit_variant_4():
    """"staged_tables": which proofs' tables of the staged ray cast rover_set_knn_map builds.  Without them the auto choice never picks
    variant 4, asking for it by name is an error (never a silent change of kernel), and results do not change."""
    from hip_helpers import hip_step
    from isaac_rover_amd import _lib, synth
    scene = synth.make_scene(n_cells=96, k=40, n_stones=16)
    distn = synth.ray_distribution("37")
    e = 1024                                          # 64 512 rays per step: the staged kernel's range in f32 arithmetic
    st = synth.make_states(e, 9.6, seed=3)
    outs = {}
    for tables in (3, 0, 1, 2):
        eng = _lib.Engine(e, device=0)
        eng.set_option("staged_tables", tables)
        eng.set_scene(scene, distn)
        assert eng.info().raycast_variant == (4 if tables & 1 else 3), tables
        outs[tables] = hip_step(eng, st)
        if tables & 1:
            eng.set_option("raycast_variant", 4)
        else:
            with pytest.raises(_lib.RoverError, match="staged"):
                eng.set_option("raycast_variant", 4)
        eng.set_option("raycast_variant", 0)
        eng.set_option("ray_precision", 2)            # as shipped: needs the fp16 proof's tables
        if tables & 2:
            eng.set_option("raycast_variant", 4)
            hip_step(eng, st)
        else:
            with pytest.raises(_lib.RoverError, match="staged"):
                eng.set_option("raycast_variant", 4)
        eng.close()
    # a variant requested BEFORE the maps are set is checked when the step runs
    eng = _lib.Engine(e, device=0)
    eng.set_option("staged_tables", 0)
    eng.set_option("raycast_variant", 4)
    eng.set_scene(scene, distn)
    with pytest.raises(_lib.RoverError, match="staged"):
        hip_step(eng, st)
    eng.close()
    with pytest.raises(_lib.RoverError, match="staged_tables"):
        _lib.Engine(8, device=0).set_option("staged_tables", 4)
    for tables in (0, 1, 2):
        for k in outs[3]:
            np.testing.assert_array_equal(outs[3][k], outs[tables][k], err_msg=f"staged_tables {tables}: {k}")
