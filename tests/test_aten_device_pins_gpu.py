"""GPU (-m gpu): pins of the library's DEFAULT arithmetic choices to ATen's own device kernels.

The golden vectors were captured from the reference on a CPU (no GPU where the reference can be imported), so they pin
``cell_index_mode = cpu_div``.  The reference as deployed runs on ``cuda:0`` (rover.py:90), where ATen evaluates the
same source lines differently in two places; the library's defaults follow the device, and these tests hold them to
what torch-ROCm's kernels compute ON THE GPU BOX, on plain tensors (no reference code involved — the op sequences
below restate the cited lines):

  * ``(depth_points - shift) / horizontal_scale`` with a Python-float divisor (camera.py:241, rock_detect.py:381,
    rover.py:590): ATen's device kernel multiplies by the reciprocal of the scalar -> ``cell_index_mode = cuda_rcp``;
  * ``torch.arange(0, X * res, res, dtype=float16, device='cuda')`` (rover_utils.py:78-79): the device kernel computes
    fp16(start + float(i) * step) -> the KNN builder's default coordinate tables (``rover_build_knn_map_ref``).
"""
import dataclasses

import numpy as np
import pytest
import torch

from conftest import tie_points

pytestmark = pytest.mark.gpu

N_CELLS = 600


def _coords(cell, n_cells, n_random, seed):
    """float32 coordinates: every .5-tie neighbourhood where division and reciprocal-multiply disagree, exact ties, cell centres,
    out-of-range values (the clamp), and uniform random ones."""
    v, _, _ = tie_points(cell, n_cells)
    rng = np.random.default_rng(seed)
    k = np.arange(n_cells, dtype=np.float64)
    extra = np.concatenate(((k + 0.5) * cell, k * cell, [-1.0, -1e-7, 0.0, n_cells * cell - 1e-3, n_cells * cell, n_cells * cell + 3.0]))
    rnd = rng.uniform(-2.0 * cell, (n_cells + 2) * cell, n_random)
    return np.concatenate((v, extra.astype(np.float32), rnd.astype(np.float32))).astype(np.float32)


def _aten_cells(xy, shift2, scale, n0, n1_for_id):
    """camera.py:241-253 / rover.py:590-601 on whatever device ``xy`` lives on: -> (ix, iy) int64."""
    scaledmap = (xy - shift2) / scale                       # Python-float divisor, like the reference's horizontal_scale
    scaledmap = torch.clamp(scaledmap, min=0, max=n0 - 1)   # the reference clamps BOTH axes with size()[0] - 1
    scaledmap = torch.round(scaledmap)
    return scaledmap[:, 0].long(), scaledmap[:, 1].long()


@pytest.mark.parametrize("shift", [(0.0, 0.0), (-3.7, 2.3)])
@pytest.mark.parametrize("precision", ["fp32", "fp16_as_shipped"])
def test_cuda_rcp_cell_lookup_is_what_aten_computes_on_the_device(shift, precision):
    """The heightmap ray of a rover with identity orientation and distribution point (0, 0) starts exactly at (pos.x, pos.y): the cell
    id in its ray record (`rover_export_rays`) is the library's cell lookup of that coordinate.  With ``cuda_rcp`` it must equal ATen's
    device result on every coordinate — tie neighbourhoods included —, with ``cpu_div`` ATen's CPU result; the two modes differ on the
    tie points only."""
    from hip_helpers import hip_step, make_engine
    from isaac_rover_amd import synth
    scene = synth.make_scene(n_cells=N_CELLS, k=2, n_stones=16, device="cuda")
    scene = dataclasses.replace(scene, shift=(shift[0], shift[1], 0.0))
    x = _coords(0.1, N_CELLS, 150_000, seed=1) + np.float32(shift[0])
    y = np.random.default_rng(2).permutation(_coords(0.1, N_CELLS, 150_000, seed=3)) + np.float32(shift[1])
    n = len(x)
    st = synth.make_states(n, N_CELLS * 0.1, seed=5)
    st["pos"][:, 0] = torch.from_numpy(x)
    st["pos"][:, 1] = torch.from_numpy(y)
    st["quat"][:] = torch.tensor([1.0, 0.0, 0.0, 0.0])
    distn = (np.array([[0.0, 0.0, -0.26878]]), np.array([0], dtype=np.int64), np.array([], dtype=np.int64))
    xy = st["pos"][:, 0:2].clone()
    if precision == "fp16_as_shipped":
        xy = xy.half()                                     # camera.py:212: the ray sources are fp16; `- shift` (f32) promotes them back
    shift2 = torch.tensor(shift, dtype=torch.float32)
    cells = {}
    for mode, name in ((1, "cuda_rcp"), (0, "cpu_div")):
        eng = make_engine(scene, distn, n, variant=None)
        eng.set_option("ray_precision", {"fp32": 0, "fp16_as_shipped": 2}[precision])
        eng.set_option("cell_index_mode", mode)
        hip_step(eng, st)
        src, _, cell, _ = eng.export_rays()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(src[:, 26, 0:2].cpu().numpy(), xy.float().numpy())      # the origin IS the position
        cells[name] = cell[:, 26].cpu().numpy().astype(np.int64)
        eng.close()
    ix, iy = _aten_cells(xy.cuda(), shift2.cuda(), 0.1, N_CELLS, N_CELLS)
    want_dev = (ix * N_CELLS + iy).cpu().numpy()
    ix, iy = _aten_cells(xy, shift2, 0.1, N_CELLS, N_CELLS)
    want_cpu = (ix * N_CELLS + iy).numpy()
    np.testing.assert_array_equal(cells["cuda_rcp"], want_dev, err_msg="cuda_rcp vs ATen on the device")
    np.testing.assert_array_equal(cells["cpu_div"], want_cpu, err_msg="cpu_div vs ATen on the CPU")
    n_diff = int((want_dev != want_cpu).sum())
    print(f"shift {shift}, {precision}: ATen device and CPU pick different cells for {n_diff} of {n} coordinates")
    if precision == "fp32" and shift == (0.0, 0.0):
        assert n_diff >= 100, "the tie neighbourhoods must tell the two modes apart (else this test pins nothing)"


def test_cuda_rcp_heightfield_lookup_is_what_aten_computes_on_the_device():
    """rover.py:588-608 at horizontal_scale 0.025 on a heightfield with one unique value per cell: `rover_sample_height` in the default
    mode == ATen's device result, in cpu_div mode == ATen's CPU result."""
    from isaac_rover_amd import _lib, synth
    n0 = 2400
    hm = (torch.arange(n0, dtype=torch.float32)[:, None] * 4096.0 + torch.arange(n0, dtype=torch.float32)[None, :]).contiguous()
    x = _coords(0.025, n0, 200_000, seed=7)
    y = np.random.default_rng(8).permutation(_coords(0.025, n0, 200_000, seed=9))
    for shift in ((0.0, 0.0), (1.25, -0.4)):
        xy = torch.from_numpy(np.stack((x + np.float32(shift[0]), y + np.float32(shift[1])), axis=1)).contiguous()
        shift2 = torch.tensor(shift, dtype=torch.float32)
        eng = _lib.Engine(8, device=0)
        eng.set_heightfield(hm, 0.025, 1.0, shift)
        for mode, dev in ((1, "cuda"), (0, "cpu")):
            eng.set_option("cell_index_mode", mode)
            got = eng.sample_height(xy.cuda()).cpu().numpy()
            ix, iy = _aten_cells(xy.to(dev), shift2.to(dev), 0.025, n0, n0)
            want = (hm.to(dev)[ix, iy] * 1.0).cpu().numpy()
            np.testing.assert_array_equal(got, want, err_msg=f"mode {mode} vs ATen on {dev}, shift {shift}")
        eng.close()


@pytest.mark.parametrize("n,res", [(600, 0.1), (1200, 0.05), (100, 0.1), (500, 0.1)])
def test_knn_builder_default_tables_are_atens_device_arange(n, res):
    """rover_utils.py:78-79: xx = torch.arange(0, res_x * res, res, device='cuda:0', dtype=float16).  The builder's default coordinate
    tables (no caller tables) are documented as that kernel's values, fp16(float(i) * res): compared here with the kernel itself on the
    GPU box, and a map built with the default tables equals one built with the kernel's output handed in."""
    dev_tab = torch.arange(0, n * res, res, device="cuda", dtype=torch.float16)
    assert dev_tab.numel() >= n                            # (600 * 0.1 / 0.1 rounds up to 601 entries: the reference uses the first res_x)
    dev_tab = dev_tab[:n].cpu().numpy()
    ours = (np.arange(n, dtype=np.float32) * np.float32(res)).astype(np.float16)
    np.testing.assert_array_equal(dev_tab.view(np.uint16), ours.view(np.uint16))
    if n != 100:
        return
    from isaac_rover_amd import _lib, synth
    verts, tris, kw = synth.knn_test_mesh("grid10m")
    eng = _lib.Engine(8, device=0)
    a = eng.build_knn_map(verts, tris, n, n, res, 16, ranking="reference_fp16").cpu().numpy()
    b = eng.build_knn_map(verts, tris, n, n, res, 16, ranking="reference_fp16", cell_x_f16=dev_tab, cell_y_f16=dev_tab).cpu().numpy()
    np.testing.assert_array_equal(a, b)
    eng.close()
