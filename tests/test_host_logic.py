"""CPU: host-side logic around the hot path — distribution table, asset formats, history buffer, sharding,
the world_size-2 gather, and that the C-ABI library loads and exports every declared symbol."""
import os
import types
import re
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


def test_native_distribution_bit_exact():
    """heightmap_distribution.py:36-115 re-derived: 1634 points, 634 sparse (0..633), 1112 dense, 112 shared."""
    from isaac_rover_amd.tasks.utils.heightmap_distribution import Heightmap, generate_native
    g = load_golden("heightmap_native")
    d, c, f = generate_native()
    np.testing.assert_array_equal(d, g["distribution"])
    np.testing.assert_array_equal(c, g["coarse_idx"])
    np.testing.assert_array_equal(f, g["fine_idx"])
    assert d.shape == (1634, 3) and len(c) == 634 and len(f) == 1112
    assert len(np.intersect1d(c, f)) == 112 and (c == np.arange(634)).all()
    hm = Heightmap("cpu")
    rays = torch.arange(1634.0).repeat(2, 1)
    assert hm.get_sparse_vector(rays).shape == (2, 634) and hm.get_dense_vector(rays).shape == (2, 1112)
    assert hm.get_num_sparse_vector() + hm.get_num_dense_vector() + 4 == 1750


def test_asset_round_trip(tmp_path):
    """On-disk formats of camera.py:154-161 / rover_utils.py:113-118: map_indices.pt is [K,X,Y] int32."""
    from isaac_rover_amd import assets, synth
    scene = synth.make_scene(n_cells=32, k=8, n_stones=4)
    assets.save_reference_assets(scene, str(tmp_path))
    raw = torch.load(tmp_path / "tasks/utils/terrain/knn_terrain/map_indices.pt")
    assert tuple(raw.shape) == (8, 32, 32) and raw.dtype == torch.int32
    assert torch.load(tmp_path / "tasks/utils/terrain/knn_terrain/vertices.pt").dtype == torch.float16
    back = assets.load_reference_assets(str(tmp_path))
    assert torch.equal(back.terrain.map_indices, scene.terrain.map_indices)
    assert torch.equal(back.rocks.map_indices, scene.rocks.map_indices)
    assert torch.equal(back.terrain.vertices, scene.terrain.vertices)
    assert torch.equal(back.heightmap, scene.heightmap)
    info = assets.read_stone_info(str(tmp_path / "tasks/utils/terrain/stone_info.npy"))
    assert info.shape == (4, 7) and info.dtype == torch.float32
    np.testing.assert_allclose(info[:, 6].numpy(), np.maximum(scene.stone_info_raw[:, 3], scene.stone_info_raw[:, 4]) / 4, rtol=1e-6)


def test_ply_reader_ascii_and_binary(tmp_path):
    from isaac_rover_amd import assets
    v = np.array([[0, 0, 0], [1, 0, 0.5], [0, 1, 0.25], [1, 1, -0.5]], np.float32)
    f = np.array([[0, 1, 2], [1, 3, 2]], np.int32)
    a = tmp_path / "a.ply"
    a.write_text("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 4\nproperty float x\nproperty float y\n"
                 "property float z\nelement face 2\nproperty list uchar int vertex_indices\nend_header\n"
                 + "".join(f"{x} {y} {z}\n" for x, y, z in v) + "".join(f"3 {i} {j} {k}\n" for i, j, k in f))
    b = tmp_path / "b.ply"
    with open(b, "wb") as fh:
        fh.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty double x\nproperty double y\nproperty double z\n"
                 b"element face 2\nproperty list uchar uint vertex_indices\nend_header\n")
        fh.write(v.astype("<f8").tobytes())
        for tri in f:
            fh.write(np.uint8(3).tobytes() + tri.astype("<u4").tobytes())
    for path in (a, b):
        vv, ff = assets.load_ply(str(path))
        np.testing.assert_array_equal(vv, v)
        np.testing.assert_array_equal(ff, f)
        assert vv.dtype == np.float32 and ff.dtype == np.int32


def test_knn_map_is_exact_nearest():
    """synth's windowed integer KNN equals brute force over all triangle centroids (rover_utils.py:68-108)."""
    from isaac_rover_amd import synth
    n_cells, k = 24, 12
    scene = synth.make_scene(n_cells=n_cells, k=k, n_stones=2)
    tris = scene.terrain.triangles.numpy()
    ii, jj = np.meshgrid(np.arange(n_cells + 1), np.arange(n_cells + 1), indexing="ij")
    vx, vy = ii.reshape(-1).astype(np.int64), jj.reshape(-1).astype(np.int64)
    cx, cy = vx[tris].sum(1), vy[tris].sum(1)                      # centroid * 3 in cell units
    for (x, y) in [(0, 0), (5, 17), (23, 23), (12, 0)]:
        key = ((cx - 3 * x) ** 2 + (cy - 3 * y) ** 2) * len(tris) + np.arange(len(tris))
        want = np.argsort(key)[:k]
        np.testing.assert_array_equal(scene.terrain.map_indices[x, y].numpy(), want)


def test_memory_history_semantics():
    """rover.py:60-77: newest first, 3 deep."""
    from isaac_rover_amd.tasks.rover import Memory
    m = Memory(4, 1, 3, "cpu")
    ptr = m.tracker.data_ptr()
    for v in (1.0, 2.0, 3.0, 4.0):
        m.input_state(torch.full((4,), v))
    assert m.tracker.data_ptr() == ptr                              # in place: the kernels borrow this pointer
    np.testing.assert_array_equal(m.tracker[0].numpy(), [4.0, 3.0, 2.0])
    assert m.get_state(1).shape == (4,) and float(m.get_state(1)[0]) == 3.0


def test_shard_range():
    from isaac_rover_amd.distributed import shard_range
    assert shard_range(262144, 8, 3) == (98304, 131072)
    with pytest.raises(ValueError):
        shard_range(10, 4, 0)


def test_c_abi_exports_every_declared_symbol():
    """librover_step.so loads (no GPU needed) and exports exactly what include/rover_step.h declares."""
    from isaac_rover_amd import _lib
    _lib.build()
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "rover_step.h")).read()
    declared = set(re.findall(r"ROVER_API\s+[\w\s\*]+?\b(rover_\w+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.rover_version()
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert declared <= exported
    assert not any(s.startswith("oracle_") for s in exported)      # the product never links the oracle


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: no import, include, link or dlopen of it in the shipped package."""
    pkg = os.path.join(ROOT, "isaac_rover_2.0_amd")
    bad = re.compile(r"^\s*(from\s+oracle|import\s+oracle)|#include\s+[\"<][^\">]*oracle|librover_oracle|rover_oracle\.(c|so)\b(?!, which)", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".sh")):
                src = open(os.path.join(dirpath, f)).read()
                assert not bad.search(src), f"{f} reaches into oracle/"


def test_engine_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from isaac_rover_amd import _lib
    with pytest.raises(_lib.RoverError, match="no HIP device"):
        _lib.Engine(8, device=0)


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["ROVER_ROOT"])
from isaac_rover_amd.distributed import StepGather, shard_range
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
E, W = 6, 5
lo, hi = shard_range(E * world, world, rank)
g = StepGather(E, W, "cpu", world, rank)
obs, rew, reset = g.local_views()
for step in range(3):
    ids = torch.arange(lo, hi, dtype=torch.float32)
    obs.copy_(ids[:, None] * 10 + torch.arange(W) + step)
    rew.copy_(ids + 0.5 * step)
    reset.copy_((torch.arange(lo, hi) + step) % 2)
    out = g.gather()
    if rank == 0:
        O, R, D = (t.reshape(E * world, *t.shape[2:]) for t in out)      # packed shards [N, E, ...] -> one flat batch
        assert out[0].shape == (world, E, W) and out[2].dtype == torch.uint8
        allids = torch.arange(E * world, dtype=torch.float32)
        assert torch.equal(O, allids[:, None] * 10 + torch.arange(W) + step)
        assert torch.equal(R, allids + 0.5 * step)
        assert torch.equal(D.long(), (torch.arange(E * world) + step) % 2)
        assert D.dtype == torch.uint8                 # done travels as one byte per env
    else:
        assert out is None
# overlapped mode: two buffer sets, the transfer of step i is only waited for when set i % 2 is needed again
g2 = StepGather(E, W, "cpu", world, rank, depth=2)
for step in range(5):
    d = step % 2
    g2.wait(d)
    if rank == 0 and step >= 2:                      # the learner reads step - 2 before the set is overwritten
        O, R, D = (t.reshape(E * world, *t.shape[2:]) for t in g2.global_views(d))
        FO, FR, FD = g2.flat_views(d)                # the documented flat accessor: one batch in global env order
        assert FO.shape == (E * world, W) and FO.is_contiguous() and torch.equal(FO, O) and torch.equal(FR, R) and torch.equal(FD, D)
        allids = torch.arange(E * world, dtype=torch.float32)
        assert torch.equal(O, allids[:, None] * 10 + torch.arange(W) + (step - 2))
        assert torch.equal(D.long(), (torch.arange(E * world) + step - 2) % 2)
    obs, rew, reset = g2.local_views(d)
    ids = torch.arange(lo, hi, dtype=torch.float32)
    obs.copy_(ids[:, None] * 10 + torch.arange(W) + step)
    rew.copy_(ids + 0.5 * step)
    reset.copy_((torch.arange(lo, hi) + step) % 2)
    g2.gather(d, wait=False)
for d in range(2):
    g2.wait(d)
if rank == 0:
    O, R, D = g2.global_views(0)
    assert torch.equal(R.reshape(-1), torch.arange(E * world, dtype=torch.float32) + 0.5 * 4)
    # one message per peer: the root's buffer is world x shard_bytes, a sender's exactly one shard
    from isaac_rover_amd.distributed import shard_bytes
    assert g2._bufs[0].numel() == world * shard_bytes(E, W) and shard_bytes(E, W) % 256 == 0
else:
    from isaac_rover_amd.distributed import shard_bytes
    assert g2._bufs[0].numel() == shard_bytes(E, W)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_gather_world_size_2_gloo(tmp_path):
    """env shards + one grouped P2P gather of (obs, reward, done); gloo, 2 processes."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ROVER_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


_DEPTH2_WORKER = r"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.environ["ROVER_ROOT"])
from isaac_rover_amd.distributed import StepGather
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
E, W, STEPS = 512, 41, 9
g = StepGather(E, W, torch.device("cpu"), world, rank, depth=2)
def value(step, r):          # what rank r's shard holds after step `step`
    return 1000.0 * step + 10.0 * r
seen = []
for i in range(STEPS):
    d = i % 2
    g.wait(d)                                    # the transfer that last read / filled set d (step i - 2) must be through ...
    if rank == 0 and i >= 2:                     # ... and then the root holds step i - 2 of EVERY shard, whatever happened to the other set since
        O, R, D = g.global_views(d)                  # [N, E, W], [N, E], [N, E]: shard r = rank r
        for r in range(world):
            assert torch.all(O[r] == value(i - 2, r)), (i, r, O[r][0, 0].item())
            assert torch.all(R[r] == value(i - 2, r) + 1.0) and torch.all(D[r] == (i - 2 + r) % 2)
        seen.append(i - 2)
    o, rw, dn = g.local_views(d)                 # "the step kernels" of step i write set d
    o.fill_(value(i, rank)); rw.fill_(value(i, rank) + 1.0); dn.fill_((i + rank) % 2)
    # injected skew: the sender runs ahead of a slow root on some steps, the root ahead of a slow sender on others, so that a
    # transfer of set d is still in flight while the other set is being overwritten by the next step
    if (rank == 0 and i % 3 == 1) or (rank == 1 and i % 3 == 2):
        time.sleep(0.15)
    g.gather(d, wait=False)
for d in range(2):
    g.wait(d)
if rank == 0:
    for i in (STEPS - 2, STEPS - 1):
        O, R, D = g.global_views(i % 2)
        for r in range(world):
            assert torch.all(O[r] == value(i, r)), (i, r)
    assert seen == list(range(STEPS - 2))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_gather_depth_2_delivers_every_step_under_skew_gloo(tmp_path):
    """The overlapped gather of bench.py's N > 1 loop (StepGather(depth=2): the transfer of step i runs while step i + 1 writes the
    other buffer set; wait(d) before set d is written again): with sleeps injected on alternating sides the root must still find
    step i - 2 of BOTH shards in set d when it comes back to it — an ordering bug here would be silent on hardware."""
    script = tmp_path / "worker_depth2.py"
    script.write_text(_DEPTH2_WORKER)
    port = 31500 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ROVER_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


_BENCH_WORKER = r"""
import os, sys, json, types, torch
root = os.environ["ROVER_ROOT"]
sys.path.insert(0, root)
import bench

class FakeEngine:
    # stands in for _lib.Engine: same call surface as bench.py uses, writes recognisable values with torch on CPU
    def __init__(self, n, rank_offset):
        self.n, self.off, self.num_observations, self.Ns, self.Nd, self.P = n, rank_offset, 41, 37, 0, 37
        self.steps = 0
    def set_scene(self, scene, distn): pass
    def set_option(self, k, v): pass
    def info(self): return types.SimpleNamespace(table_bytes=[1, 1], raycast_variant=2)
    def make_in(self, *t): return t
    def make_out(self, obs, **kw): return dict(obs=obs, **kw)
    def step(self, sin, sout, increment_progress=True, compact=False):
        self.steps += 1
        ids = torch.arange(self.off, self.off + self.n, dtype=torch.float32)
        sout["obs"].copy_(ids[:, None] + 0.001 * self.steps)
        sout["rew"].copy_(ids)
        flags = (torch.arange(self.off, self.off + self.n) + self.steps) % 2
        sout["reset"].copy_(flags)
        sout["done_u8"].copy_(flags)
    def reset_envs(self, *a, **k): pass
    def set_profiling(self, on, every=1): self.steps0 = self.steps
    def get_profile(self):
        n = self.steps - self.steps0
        return types.SimpleNamespace(raycast_ms=1.0 * n, launches=n, pairs_per_launch=self.n * 63 * 200)

bench._device = lambda local_rank: torch.device("cpu")
bench._init_process_group = lambda dist, device: dist.init_process_group("gloo")
bench._make_engine = lambda n, local_rank, n_global, off: FakeEngine(n, off)
bench._sync = lambda: None
bench.load_scene = lambda args, device, local_rank=0: (None, None)
rc = bench.main(sys.argv[1:])
print("rank", os.environ["RANK"], "done", file=sys.stderr)
sys.exit(rc)
"""

_BENCH_ARGS = ["--steps", "7", "--warmup", "3", "--envs-per-gpu", "64", "--preroll-ms", "20", "--passes", "3"]


def _check_bench_line(out, extra):
    import json
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    assert out.rstrip("\n").splitlines()[-1] == lines[0], "the record must be the LAST stdout line"
    assert len(lines[0]) < 4000, "the driver keeps a bounded tail of stdout: the record line must stay compact"

    def no_constants(x):
        raise ValueError(f"not strict JSON: {x}")
    d = json.loads(lines[0], parse_constant=no_constants)
    assert d["n_gpus"] == 2 and d["steps"] == 7 and d["warmup"] == 3 and d["scaling"] == "weak" and d["config"]["envs_total"] == 128
    assert d["value"] > 0 and abs(d["value"] - 128 * 7 / (d["ms_per_step"] * 7e-3)) < 1e-6 * d["value"]
    assert ("overlapped" in d["config"]["workload"]) == (extra == "")
    assert "cpu_baseline" not in d and set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert d["rccl_ranks"] == 2 and d["backend"] == "gloo"          # the backend really saw both ranks
    assert d["gather_check"] is True                                # root-side buffers == per-rank checksums
    assert len(d["per_rank"]["ms_per_step"]) == 2 and all(x > 0 for x in d["per_rank"]["ms_per_step"])
    alt = d["alt_sync_gather" if extra == "" else "alt_overlapped"]
    assert alt["gather_check"] is True and alt["value"] > 0
    assert d["gather_bytes_per_rank_per_step"] == 64 * (4 * 41 + 4 + 1) and d["gather_messages_per_peer_per_step"] == 1
    assert d["passes"]["n"] == 3 and d["passes"]["min_ms_per_step"] <= d["ms_per_step"] <= d["passes"]["max_ms_per_step"]
    assert "BASELINE configs" not in d["config"]["workload"]      # 64 envs per rank is none of BASELINE.json's configs
    return d


def test_bench_record_line_stays_compact_and_strict():
    """bench.compact_line: whatever the full record holds (seven `also` workloads with their own rooflines and 900-byte notes, NaN
    from an empty profile, a failed workload), the line printed on stdout is strict JSON well under the ~8 KB of stdout the driver
    keeps (round 5's 21 KB line was lost) and carries the contract's keys, roofline and cpu_baseline."""
    import json
    import bench
    note = "x" * 900
    roof = {"bound": "hbm", "achieved": 3161.8123456789, "peak": 8000.0, "unit": "GB/s", "frac": 0.395123456, "traffic": 1.1631e9, "kernel": "lane_scan_kernel",
            "avg_launch_ms": 0.3678123, "stall_frac": float("nan"), "limited_by": "latency", "profile_key": "E65536_P37_K200_C600", "profile_stale": False,
            "valu": {"frac": 0.62, "issue_rates": {"a": 4.1} , "unit": note}, "hbm": {"traffic": 1.0}, "note": note, "algorithmic_equiv_GBps": float("inf")}
    also = {f"workload_{i}": {"value": 1.23456789e8, "unit": "env-steps/s", "ms_per_step": 0.4567891, "steps": 50, "passes": {"n": 5}, "dtype": "f32",
                              "workload": "BASELINE configs[1]: 4096 envs x (37 + 26) rays, K=200, 600x600 cells, mesh=grid, ray_precision=fp32, cell_index_mode=cuda_rcp",
                              "roofline": dict(roof), "cull": {"triangles": [1, 2]}} for i in range(9)}
    also["task_api"] = {"value": 1e7, "ms_per_step": 0.2, "dtype": "f32", "host_enqueue_ms": 0.15, "engine_step_ms": 0.17,
                        "workload": "RoverTask.pre_physics_step + post_physics_step (custom workload (no BASELINE config)): 512 envs x (native + 26) rays, K=200, mesh=irregular"}
    also["broken"] = {"error": "RoverError('" + note + "')"}
    full = {"metric": "env-steps/sec (obs+reward+done)", "value": 1.42e8, "unit": "env-steps/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.46,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "passes": {"n": 30, "statistic": note, "min_ms_per_step": 0.45, "max_ms_per_step": 0.47, "spread": 0.02},
            "config": {"workload": "BASELINE configs[2]: 65536 envs/GPU x 1 GPU, 37-point heightmap + 26 rock rays, K=200", "envs_total": 65536, "rays_per_env": 63,
                       "obs_dim": 41, "algorithmic_bytes_per_env_step": 227148, "table_bytes": 1 << 32},
            "rccl_ranks": 1, "backend": None, "roofline": roof, "lib": "rover_step 0.3 (gfx950) src-000000000000", "lib_built_from_tree": True,
            "cull": {"candidate_pairs_per_ray": 3.6, "rays_not_scanned": 0.25, "triangles": [720000, 720000]},
            "cpu_baseline": {"value": 374760.123, "unit": "env-steps/s", "cores": 256, "kind": "port", "cpu_model": "AMD EPYC", "sample": note,
                             "sample_short": "2048 envs x 20 reps, best rep", "torch_ref": {"torch_threads": 8, "fp32": {"value": 374.0}, "fp16_as_shipped": {"value": 79.7}},
                             "host": {"sockets": 2}, "reference_pytorch": note},
            "also": also}
    line = json.dumps(bench.compact_line(full, "gpurun_out/bench_full.json"), allow_nan=False, separators=(",", ":"))
    assert "\n" not in line and len(line) < bench.COMPACT_LIMIT, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert d[k] == full[k]
    assert d["config"]["workload"].startswith("BASELINE configs[2]") and "model" not in d["config"]
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "stall_frac", "profile_key", "profile_stale"}
    assert d["roofline"]["stall_frac"] is None and d["roofline"]["frac"] == 0.39512
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and len(d["cpu_baseline"]["sample"]) < 100
    assert set(d["also"]) == set(also) and set(d["also"]["workload_0"]) == {"value", "ms_per_step", "dtype", "workload", "frac"}
    assert d["also"]["task_api"]["host_enqueue_ms"] == 0.15 and "error" in d["also"]["broken"] and len(d["also"]["broken"]["error"]) <= 120
    assert d["full_record"] == "gpurun_out/bench_full.json"


def test_bench_labels_name_a_baseline_config_only_when_everything_matches():
    """bench.config_label: "BASELINE configs[k]" needs the batch size, the ray set, K, cells, stones, mesh, arithmetic and GPU count of
    that config (BASELINE.json); `--gpus 8 --envs-per-gpu 32768` IS configs[3], 8 x 65 536 is weak scaling of configs[2]."""
    import bench
    base = dict(k=200, cells=600, stones=1024, mesh="grid", ray_precision="fp32", graph=False, rays="37", validate_goals=False)
    A = lambda **kw: types.SimpleNamespace(**{**base, **kw})
    assert bench.config_label(A(), 65536, 1, 65536) == "BASELINE configs[2]"
    assert bench.config_label(A(), 4096, 1, 4096) == "BASELINE configs[1]"
    assert bench.config_label(A(rays="120", validate_goals=True), 65536, 1, 65536) == "BASELINE configs[4]"
    assert bench.config_label(A(), 32768, 8, 262144) == "BASELINE configs[3]"
    assert "shard of BASELINE configs[3]" in bench.config_label(A(), 32768, 1, 32768)
    assert bench.config_label(A(), 65536, 8, 524288).startswith("weak scaling of BASELINE configs[2]")
    for kw, e in ((dict(rays="native"), 4096), (dict(mesh="irregular"), 65536), (dict(ray_precision="fp16_as_shipped"), 65536), (dict(k=64), 65536),
                  (dict(cells=300), 4096), (dict(rays="120"), 65536), (dict(), 512), (dict(), 32000)):
        assert "BASELINE" not in bench.config_label(A(**kw), e, 1, e).replace("no BASELINE config", ""), (kw, e)


@pytest.mark.parametrize("extra", ["", "--sync-gather"])
def test_bench_self_launch_gloo(tmp_path, extra):
    """`python bench.py --gpus 2` with NO rank environment (how the driver calls it): the parent spawns two rank processes
    itself, relays exactly one JSON line and returns 0.  CPU: gloo + a stand-in engine substituted through ROVER_BENCH_CHILD;
    the real bench.main() runs in every rank (rank bookkeeping, pre-roll without collectives, overlapped / blocking gather of
    (obs f32, rew f32, done u8), integer checksums of what the root received, barrier + max over ranks, both gather modes)."""
    script = tmp_path / "bench_worker.py"
    script.write_text(_BENCH_WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ROVER_ROOT=ROOT, ROVER_BENCH_CHILD=str(script))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + _BENCH_ARGS + extra.split(),
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "rank 0 done" in p.stderr and "rank 1 done" in p.stderr
    _check_bench_line(p.stdout, extra)


def test_bench_self_launch_reports_a_failed_rank(tmp_path):
    """A rank that dies makes the parent exit non-zero without a JSON line — at once, although rank 0 would block FOREVER (a rank
    stuck in RCCL's rendezvous or in a collective whose peer is gone): every child is watched, the survivors are killed."""
    script = tmp_path / "bad_worker.py"
    script.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    time.sleep(1)\n    sys.exit(7)\ntime.sleep(3600)\nprint('{}')\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(ROVER_BENCH_CHILD=str(script))
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 7 and p.stdout.strip() == "" and "rank 1 exited with code 7" in p.stderr
    assert time.time() - t0 < 60, "the parent waited for a rank that can never finish"


def test_bench_self_launch_rank_timeout(tmp_path):
    """--rank-timeout-s: ranks that neither finish nor fail are killed and the parent exits 124 without a JSON line."""
    script = tmp_path / "stuck_worker.py"
    script.write_text("import time\ntime.sleep(3600)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(ROVER_BENCH_CHILD=str(script))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rank-timeout-s", "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 124 and p.stdout.strip() == "" and "still running after --rank-timeout-s" in p.stderr


def test_bench_under_torchrun_env_does_not_spawn(tmp_path):
    """With WORLD_SIZE / RANK already set (python -m torch.distributed.run ...) every process is one rank: nothing is spawned."""
    script = tmp_path / "bench_worker.py"
    script.write_text(_BENCH_WORKER)
    port = 31000 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ROVER_ROOT=ROOT, ROVER_BENCH_CHILD="/nonexistent/never-started")
        procs.append(subprocess.Popen([sys.executable, str(script), "--gpus", "2"] + _BENCH_ARGS, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, e[-3000:]
    _check_bench_line(outs[0][0], "")
    assert not any(l.startswith("{") for l in outs[1][0].splitlines())


def test_bench_parent_never_touches_the_gpu_runtime():
    """The launcher half of bench.py imports nothing that could initialise HIP: no torch import at module level or in
    launch_ranks()."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    names = {a.name.split(".")[0] for n in top if isinstance(n, ast.Import) for a in n.names} | \
            {n.module.split(".")[0] for n in top if isinstance(n, ast.ImportFrom) and n.module}
    assert "torch" not in names and "isaac_rover_amd" not in names and "numpy" not in names
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "launch_ranks")
    assert not any(isinstance(n, (ast.Import, ast.ImportFrom)) for n in ast.walk(fn))


def test_fastdiv_matches_integer_division(tmp_path):
    """`make_fastdiv` / `FastDiv::div` (csrc/rover_internal.h, used by assemble_obs_kernel for i / W): the host half builds the
    magic numbers, the device half is three integer instructions — restated here on the host and compared with `/` for divisors
    around powers of two, the obs widths of the configs and the 32-bit extremes."""
    import shutil
    if not (shutil.which("g++") and os.path.isdir("/opt/rocm/include")):
        pytest.skip("needs g++ and the ROCm headers")
    src = tmp_path / "fd.cpp"
    src.write_text(r'''
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "rover_internal.h"
int main() {
    unsigned long long bad = 0;
    const uint32_t ds[] = {1, 2, 3, 4, 5, 7, 41, 63, 64, 65, 124, 146, 150, 1000, 1750, 4099, 65535, 65536, 65537, 0x7fffffffu,
                           0x80000000u, 0x80000001u, 0xfffffffeu, 0xffffffffu};
    for (uint32_t d : ds) {
        const rover::FastDiv f = rover::make_fastdiv(d);
        for (uint64_t k = 0; k < 3000000; ++k) {
            const uint32_t n = k < 1000000 ? (uint32_t)k : (k < 2000000 ? 0xffffffffu - (uint32_t)(k - 1000000) : (uint32_t)(k * 2654435761ull + 12345u));
            const uint32_t t = (uint32_t)(((uint64_t)f.m * n) >> 32);              // __umulhi
            const uint32_t q = (t + ((n - t) >> f.s1)) >> f.s2;
            if (q != n / d) ++bad;
        }
    }
    std::printf("%llu\n", bad);
    return bad != 0;
}
''')
    exe = tmp_path / "fd"
    inc = os.path.join(ROOT, "isaac_rover_2.0_amd", "csrc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", inc, "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", str(exe), str(src)],
                   check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True)
    assert out.stdout.strip() == "0"
