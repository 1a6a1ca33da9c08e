#!/usr/bin/env python3
"""bench.py — env-steps/s of the rover step hot path (obs + reward + done + done compaction) on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one fused RLTask.post_physics_step (rl_task.py:239-259) over one batch of synthetic sim states
that are already resident in HBM: get_observations (P terrain rays + 26 rock rays per rover against the
K triangles of their 0.1 m cell) + calculate_metrics + is_done + done compaction, through the C ABI
(rover_step).  Default workload = BASELINE.json configs[2] — the config the north star's >= 4 M env-steps/s
target is quoted on: 65 536 envs per GPU, 37-point radial heightmap + 26 rock-collision rays + the stone_info
occupancy mask, 600 x 600 cell maps with K = 200.

N > 1 (weak scaling, 65 536 envs per GPU, one process per GPU over RCCL):
  * `python bench.py --gpus N` with no rank environment starts the N ranks ITSELF: the parent process never touches
    the GPU runtime, spawns N fresh children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays
    rank 0's single JSON line and exits non-zero if any child fails.
  * under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (WORLD_SIZE already set) nothing is
    spawned: the process is one rank.
  Every step hands (obs f32, reward f32, done u8) of every shard to rank 0 with one grouped RCCL send/recv.  Two timed
  passes of K steps each: the headline pass lets that transfer run on RCCL's stream under the kernels of the next step
  (double-buffered outputs), the second pass serialises it (`alt_sync_gather` in the JSON line); --sync-gather makes the
  serialised pass the headline one.  After each pass the root compares integer checksums of what it received with
  checksums every rank computed of what it sent (`gather_check`).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     — the ray-cast kernel: f32 VALU issue utilisation (instructions per launch from profiles/valu.json x 4
                 cycles / (1 024 SIMDs x 2.4 GHz x HIP-event time)), with the HBM side (PMC traffic, GB/s, fraction of
                 8 TB/s) and SURVEY §8(d)'s no-reuse byte model (`algorithmic_equiv_GBps`, `reuse_factor`) next to it
  cpu_baseline — the CPU oracle (oracle/rover_oracle.c, OpenMP) timed on a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMD = 1024                # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4              # max shader clock (rocm-smi shows 2.38 GHz while the step loop runs, tools/clock_probe.sh)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preroll-ms", type=float, default=400.0,
                    help="untimed load before the warm-up steps so that the GPU clocks have ramped (DVFS: the first ~60 ms of "
                         "load run up to 30 %% slower, measured with --debug-timing); 0 disables")
    ap.add_argument("--no-torch-ref", action="store_true", help="skip the PyTorch tensor-program CPU baseline (oracle/torch_ref.py)")
    ap.add_argument("--envs-per-gpu", type=int, default=65536)
    ap.add_argument("--rays", default="37", choices=["9", "37", "120", "native"],
                    help="BASELINE ray sets; native = the reference's own 1634-point distribution (1750-float obs)")
    ap.add_argument("--cells", type=int, default=600)
    ap.add_argument("--k", type=int, default=200)
    ap.add_argument("--stones", type=int, default=1024)
    ap.add_argument("--mesh", default="grid", choices=["grid", "shuffled", "irregular"],
                    help="grid = SURVEY 8(d)'s regular 0.1 m heightfield mesh in row-major triangle order (the headline workload); "
                         "shuffled = the same mesh with its triangle table in random order; irregular = a decimated-style mesh "
                         "(synth.irregular_mesh: non-uniform Delaunay triangulation, millimetre to metre edges, ~80 degree rock "
                         "flanks, needle / zero-area triangles, shuffled ids) with K-nearest maps built by rover_build_knn_map")
    ap.add_argument("--validate-goals", action="store_true",
                    help="configs[4]: also run the reset/spawn-goal validation kernel on the envs flagged done each step")
    ap.add_argument("--ray-precision", default="fp32", choices=["fp32", "fp16_sources", "fp16_as_shipped"],
                    help="fp32 = the reference's fp32 mode (default, the north star's parity mode); fp16_as_shipped = bit-identical to the "
                         "reference as shipped (Camera.dtype = float16)")
    ap.add_argument("--cell-index-mode", default="cuda_rcp", choices=["cpu_div", "cuda_rcp"],
                    help="camera.py:241 `x / 0.1`: the multiply by the reciprocal ATen's CUDA kernels turn it into (default: the "
                         "reference as deployed, rover.py:90) or a true division as ATen evaluates it on CPU (what the golden vectors pin)")
    ap.add_argument("--sync-gather", action="store_true",
                    help="N > 1: headline pass waits for the RCCL gather of a step before the next step starts (default: the gather "
                         "of step i runs on RCCL's stream under the kernels of step i + 1; the other mode is timed as the alt pass)")
    ap.add_argument("--no-alt-pass", action="store_true", help="N > 1: skip the second timed pass in the other gather mode")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (A/B of the launch overhead)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--debug-timing", action="store_true")
    ap.add_argument("--cpu-sample-envs", type=int, default=2048)
    ap.add_argument("--scene-cache", default=os.environ.get("ROVER_SCENE_CACHE", ""))
    ap.add_argument("--event-every", type=int, default=8,
                    help="bracket the ray-cast launch of every n-th timed step with HIP events (roofline.avg_launch_ms is their mean): "
                         "an event pair costs the stream ~12 us around the kernel, 1.8 %% of a step if every step carried one")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the `also` object of the default N = 1 run: the same 65 536-env batch on the irregular (decimated-style) mesh "
                         "and in the reference's as-shipped fp16 arithmetic, timed in the same process after the headline pass")
    ap.add_argument("--also-only", default="", help="comma-separated names of the `also` workloads to run (default: all)")
    ap.add_argument("--also-steps", type=int, default=50, help="timed steps per pass of each `also` workload (>= 20)")
    ap.add_argument("--passes", type=int, default=30,
                    help="back-to-back timed passes of --steps steps (each bracketed by barrier + device sync); value / ms_per_step are the MEDIAN "
                         "pass, min / max in `passes`")
    ap.add_argument("--also-passes", type=int, default=5)
    ap.add_argument("--rank-timeout-s", type=float, default=900.0,
                    help="N > 1 self-launch: kill every rank and exit 124 when the run has not finished after this many seconds")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# parent: start one process per GPU.  Nothing here may initialise the GPU runtime (no torch.cuda call, no HIP call): on
# this pool a process that has touched the GPU must not exec / be replaced, and the children must start from a clean slate.
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, rank_timeout_s=900.0, poll_s=0.2):
    """Spawn ``n`` rank processes of this script, relay rank 0's JSON line to stdout; returns the exit code.

    All children are watched together: the first one that exits non-zero (or the overall ``rank_timeout_s``) ends the run — the
    others are killed (exactly the PIDs started here) instead of being left blocked in a collective or in RCCL's rendezvous."""
    child = os.environ.get("ROVER_BENCH_CHILD", os.path.abspath(__file__))      # tests substitute a stand-in engine here
    port = _free_port()
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")         # rank 0's stdout: a file, so a long line can never block the child on a full pipe
    rc = 0
    t_start = time.time()
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: required for RCCL between processes on this host
            procs.append(subprocess.Popen([sys.executable, child] + list(argv), env=env, text=True,
                                          stdout=out0 if r == 0 else sys.stderr))
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                for r, c in bad:
                    print(f"bench.py: rank {r} exited with code {c}; stopping the other ranks", file=sys.stderr)
                rc = bad[0][1] if bad[0][1] > 0 else 1
                break
            if all(c == 0 for c in codes):
                break
            if time.time() - t_start > rank_timeout_s:
                alive = [r for r, c in enumerate(codes) if c is None]
                print(f"bench.py: ranks {alive} still running after --rank-timeout-s {rank_timeout_s:.0f}; killing them", file=sys.stderr)
                rc = 124
                break
            time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                     # exactly the PIDs started above
                p.wait()
                rc = rc or 1
    out0.seek(0)
    text = out0.read()
    out0.close()
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    for ln in text.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if rc == 0 and len(lines) != 1:
        print(f"bench.py: expected ONE JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        rc = 1
    if rc == 0:
        print(lines[0], flush=True)
    return rc


# ---------------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------------
def algorithmic_bytes_per_env_step(p, k, ns, nd, r=26):
    """B = (P + R) K 18 + 128 + (4 + Ns + Nd) 4 + 56."""
    return (p + r) * k * 18 + 128 + (4 + ns + nd) * 4 + 56


def load_scene(args, device, local_rank=0):
    """-> (scene, height function or None)"""
    import torch
    from isaac_rover_amd import synth
    key = f"scene_{args.mesh}_c{args.cells}_k{args.k}_s{args.stones}.pt"
    path = os.path.join(args.scene_cache, key) if args.scene_cache else ""
    zf = None
    spec = None
    if args.mesh == "irregular":
        spec = synth.IrregularSpec(extent_x=args.cells * 0.1, extent_y=args.cells * 0.1, n_rocks=args.stones, seed=5, fine=0.05)
        zf, _ = synth.irregular_height(spec)
    if path and os.path.exists(path):
        return torch.load(path, weights_only=False), zf
    if args.mesh == "irregular":
        from isaac_rover_amd import _lib, assets
        tool = _lib.Engine(8, device=local_rank)
        scene, _ = assets.build_irregular_scene(tool, spec, args.k)
        tool.close()
    else:
        scene = synth.make_scene(n_cells=args.cells, k=args.k, n_stones=args.stones, device=device)
        if args.mesh == "shuffled":
            import dataclasses
            host = lambda m: dataclasses.replace(m, map_indices=m.map_indices.cpu(), triangles=m.triangles.cpu(), vertices=m.vertices.cpu())
            scene = synth.shuffle_triangle_ids(dataclasses.replace(scene, terrain=host(scene.terrain), rocks=host(scene.rocks)), seed=9)
    if path:
        os.makedirs(args.scene_cache, exist_ok=True)
        torch.save(scene, path)
    return scene, zf


def cpu_baseline(args, scene, distn, states):
    """Oracle (plain C + OpenMP, kind='port') on a bounded sample of the same workload."""
    from oracle import oracle as orc
    n = min(args.cpu_sample_envs, states["pos"].shape[0])
    st = {k: v[:n].cpu() for k, v in states.items()}
    t = orc.KnnMap(scene.terrain.map_indices, scene.terrain.triangles, scene.terrain.vertices)
    r = orc.KnnMap(scene.rocks.map_indices, scene.rocks.triangles, scene.rocks.vertices)
    prec = {"precision": args.ray_precision}
    orc.step(t, r, st, *distn, **prec)              # warm-up (page-in, thread pool)
    best, reps, t_all = float("inf"), 0, time.perf_counter()
    while reps < 3 or (time.perf_counter() - t_all < 10.0 and reps < 20):
        t0 = time.perf_counter()
        orc.step(t, r, st, *distn, **prec)
        best = min(best, time.perf_counter() - t0)
        reps += 1
    logical = os.cpu_count() or 1
    threads = int(os.environ.get("OMP_NUM_THREADS", logical))
    model, sockets, phys = "", set(), set()
    try:
        with open("/proc/cpuinfo") as f:
            pid = cid = None
            for ln in f:
                if ln.startswith("model name") and not model:
                    model = ln.split(":", 1)[1].strip()
                elif ln.startswith("physical id"):
                    pid = ln.split(":", 1)[1].strip()
                    sockets.add(pid)
                elif ln.startswith("core id"):
                    cid = ln.split(":", 1)[1].strip()
                    phys.add((pid, cid))
    except OSError:
        pass
    # second CPU baseline: the reference's APPROACH — dense [E, p, K, 3, 3] gathers, batched
    # ray_distance, min over K, as a PyTorch tensor program (oracle/torch_ref.py) — in fp32 mode and as shipped (fp16 tensors)
    torch_ref = None
    if not args.no_torch_ref:
        import torch
        from oracle import torch_ref as tr
        import dataclasses
        host = lambda m: dataclasses.replace(m, map_indices=m.map_indices.cpu(), triangles=m.triangles.cpu(), vertices=m.vertices.cpu())
        scene_cpu = dataclasses.replace(scene, terrain=host(scene.terrain), rocks=host(scene.rocks))
        torch_ref = {"torch_threads": torch.get_num_threads()}
        for name, dt, m in (("fp32", torch.float32, min(n, 256)), ("fp16_as_shipped", torch.float16, min(n, 64))):
            sub = {k: v[:m] for k, v in st.items()}
            tr.step(scene_cpu, sub, *distn, dtype=dt)                # warm-up
            tb, tr_reps, t_all = float("inf"), 0, time.perf_counter()
            while tr_reps < 2 or (time.perf_counter() - t_all < 4.0 and tr_reps < 5):
                t0 = time.perf_counter()
                tr.step(scene_cpu, sub, *distn, dtype=dt)
                tb = min(tb, time.perf_counter() - t0)
                tr_reps += 1
            torch_ref[name] = {"value": m / tb, "unit": "env-steps/s", "sample": f"{m} envs x {tr_reps} reps, best rep"}
    return {"value": n / best, "unit": "env-steps/s", "cores": threads, "kind": "port", "cpu_model": model, "torch_ref": torch_ref,
            "host": {"sockets": len(sockets) or None, "physical_cores": len(phys) or None, "logical_cpus": logical,
                     "omp_threads": threads},
            "sample_short": f"{n} envs x {reps} reps, best rep (oracle/rover_oracle.c, OpenMP)",
            "sample": f"{n} envs x {reps} reps of the same workload (P={distn[0].shape[0]}, K={args.k}, "
                      f"{args.cells}x{args.cells} cells), best rep; oracle/rover_oracle.c, gcc -O2 -fopenmp on every logical CPU",
            "reference_pytorch": "the reference's own PyTorch path is not runnable on the GPU box (it cannot travel); measured in the "
                                 "build container (8 threads): 303-1525 env-steps/s, BASELINE.md §2; torch_ref = the same op "
                                 "sequence restated in oracle/torch_ref.py, timed here"}


# The touch points with the GPU runtime, as functions so that tests/test_host_logic.py can drive the N > 1 control flow (rank
# bookkeeping, overlapped gather, checksums, max-over-ranks timing, the JSON line) on CPU with gloo and a stand-in engine.
def _device(local_rank):
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rover step path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    return torch.device("cuda", local_rank)


def _init_process_group(dist, device):
    dist.init_process_group("nccl", device_id=device)


def _make_engine(num_envs, local_rank, num_envs_global, env_offset):
    from isaac_rover_amd import _lib
    return _lib.Engine(num_envs, device=local_rank, num_envs_global=num_envs_global, env_offset=env_offset)


def _sync():
    import torch
    torch.cuda.synchronize()


def _lib_version():
    try:
        from isaac_rover_amd import _lib
        return _lib.version()
    except Exception:
        return ""


def _lib_matches_tree():
    try:
        from isaac_rover_amd import _lib
        return _lib.version().endswith("src-" + _lib.source_hash())
    except Exception:
        return None


def _gpu_event():
    """A timing event on the current stream, or None without a GPU (the CPU stand-in of tests/test_host_logic.py)."""
    import torch
    if not torch.cuda.is_available():
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


def _profile_entry(fname, key):
    path = os.path.join(ROOT, "profiles", fname)
    try:
        with open(path) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


def roofline(args, E, n_rays, prof, info, lib_version=""):
    """The ray-cast kernel's roofline object.  Live: HIP-event time per launch.  From profiles/: per launch of the same workload the
    VALU work in SIMD cycles (PMC instruction counts by kind x MEASURED cycles per instruction, profiles/issue_rates.json) and the HBM
    bytes (rocprofv3 --pmc, tools/collect_profiles.sh); formulas in profiles/README.md.  `profile_stale`: the counts were taken on a
    library built from other sources than the one running (rover_version() carries a hash of its sources)."""
    ray_ms = prof.raycast_ms / max(prof.launches, 1)
    t = ray_ms * 1e-3
    key = f"E{E}_P{args.rays}_K{args.k}_C{args.cells}" + ("" if args.ray_precision == "fp32" else "_" + args.ray_precision) \
        + ("" if args.mesh == "grid" else "_" + args.mesh)
    valu = _profile_entry("valu.json", key)
    traf = _profile_entry("traffic.json", key)
    rays = E * (n_rays + 26)
    peak = N_SIMD * CLOCK_GHZ                      # G SIMD-cycles per second
    achieved = frac = insts_per_ray = None
    if valu and t > 0:
        insts = float(valu["valu_insts_per_launch"])
        insts_per_ray = insts / rays
        cyc = valu.get("valu_simd_cycles_per_launch")
        if cyc:
            achieved = float(cyc) / t / 1e9
            frac = achieved / peak
    traffic = traf.get("hbm_bytes_per_launch") if traf else None
    algo = 18.0 * prof.pairs_per_launch            # 18 B per (ray, triangle), no reuse credited
    algo_gbs = algo / t / 1e9 if t > 0 else None
    hbm = None
    if traffic and t > 0:
        hbm = {"traffic": traffic, "GBps": traffic / t / 1e9, "frac_of_8TBps": traffic / t / 1e9 / HBM_PEAK_GBS}
    valu_obj = {"achieved": achieved, "peak": peak, "unit": "G VALU-busy SIMD cycles/s (peak: 1024 SIMDs x 2.4 GHz)", "frac": frac,
                "valu_insts_per_ray": insts_per_ray,
                "cycles_per_inst": (float(valu["valu_simd_cycles_per_launch"]) / float(valu["valu_insts_per_launch"])
                                    if valu and valu.get("valu_simd_cycles_per_launch") else None),
                "frac_lower_bound": (float(valu["valu_simd_cycles_lower_bound"]) / t / 1e9 / peak
                                     if valu and valu.get("valu_simd_cycles_lower_bound") and t > 0 else None),
                "issue_rates": valu.get("issue_rates_cycles") if valu else None}
    profiled_lib = (valu or traf or {}).get("lib")
    stale = bool(profiled_lib) and bool(lib_version) and profiled_lib != lib_version
    if (valu or traf) and not profiled_lib:
        stale = True                               # an entry from before the library carried a source hash
    # The line leads with the counter-measured HBM fraction (north star: "rocprof HBM GB/s against the 8 TB/s roofline"); the VALU issue
    # model is the secondary entry `valu`, and `stall_frac` = SQ_WAIT_ANY / SQ_WAVE_CYCLES (the share of the resident waves' cycles spent
    # in s_waitcnt) says how far from either roof the kernel's own latency chains keep it.
    stall = valu.get("stall_frac") if valu else None
    if stale or hbm is None:
        # no counters for this workload key, or counters of another build of the library: no bound and no fraction is claimed
        # (the live launch time and the no-reuse byte model below are still this run's)
        head = {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None,
                "profile_missing": not (valu or traf)}
        if stale:
            hbm = None
            stall = None
            valu_obj.update({"achieved": None, "frac": None})
    else:
        head = {"bound": "hbm", "achieved": hbm["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm["frac_of_8TBps"]}
    head["stall_frac"] = stall
    # `bound` names the roofline the contract asks for on this path (HBM: the north star's "rocprof HBM GB/s against the 8 TB/s roofline";
    # there is no dense contraction, so never "mfma").  Which of the two measured resources lies nearer its roof is `nearest_roof`, from the
    # HBM fraction and the VALU issue BRACKET [frac_lower_bound, frac] (the staged kernel's plain and mixed / packed f32 instructions share the
    # kind counters: the lower end prices every f32 instruction at the plain rate) — reported as a bracket, no midpoint is taken.
    v_hi, v_lo = valu_obj["frac"], valu_obj.get("frac_lower_bound")
    head["valu_frac_bracket"] = None if v_hi is None else [v_lo if v_lo is not None else v_hi, v_hi]
    if head["frac"] is None:
        head["nearest_roof"] = head["limited_by"] = None
    else:
        lo = v_lo if v_lo is not None else (v_hi or 0.0)
        head["nearest_roof"] = "hbm" if head["frac"] >= (v_hi or 0.0) else ("valu issue" if lo > head["frac"] else "hbm or valu issue (inside the bracket)")
        top = max(head["frac"], v_hi or 0.0)
        head["limited_by"] = head["nearest_roof"] if top >= 0.7 else "latency (neither roof reached)"
    head.update({"kernel": {4: "lane_scan_kernel", 3: "cull_scan_kernel", 2: "raycast_binned_kernel"}.get(info.raycast_variant, "raycast_kernel"),
                 "traffic": traffic, "avg_launch_ms": ray_ms, "launches": int(prof.launches), "launches_timed_every": int(getattr(args, "event_every", 1)), "rays_per_launch": rays,
                 "hbm": hbm, "valu": valu_obj, "algorithmic_bytes_per_launch": algo, "algorithmic_equiv_GBps": algo_gbs,
                 "reuse_factor": (algo / traffic) if traffic else None,
                 "profile_key": key, "profile_lib": profiled_lib, "lib": lib_version, "profile_stale": stale,
                 "note": "hbm: PMC-measured HBM bytes per launch (profiles/traffic.json: 2*FETCH_SIZE + WRITE_SIZE, gfx950 correction) / live "
                         "HIP-event time per launch; valu.frac = VALU-busy SIMD cycles per launch (profiles/valu.json: PMC instruction counts by "
                         "kind x the measured cycles per instruction of profiles/issue_rates.json — 4.11 for packed f32 / compares / conversions, "
                         "2.20 for 32-bit integer ops, 8 waves per SIMD) / (1024 SIMDs x 2.4 GHz x the same time); algorithmic_equiv_GBps = SURVEY "
                         "8(d)'s no-reuse byte model (18 B per (ray, triangle) pair), which the kernel beats by not touching provably "
                         "rejected triangles: reuse_factor = model bytes / measured bytes; formulas in profiles/README.md"})
    return head


def config_label(args, E, world, E_global):
    """The BASELINE.json config a run IS — only when batch size, ray set, K, cells, stones, mesh, arithmetic and GPU count all match —
    else a plain description."""
    std = (args.k == 200 and args.cells == 600 and args.stones == 1024 and args.mesh == "grid" and args.ray_precision == "fp32"
           and not args.graph)
    plain37 = std and args.rays == "37" and not args.validate_goals
    if plain37 and world == 1 and E == 65536:
        return "BASELINE configs[2]"
    if plain37 and world == 1 and E == 4096:
        return "BASELINE configs[1]"
    if std and world == 1 and E == 65536 and args.rays == "120" and args.validate_goals:
        return "BASELINE configs[4]"
    if plain37 and world == 8 and E_global == 262144:
        return "BASELINE configs[3]"
    if plain37 and world == 1 and E == 32768:
        return "one rank's shard of BASELINE configs[3] (32 768 of its 262 144 envs, no gather)"
    if plain37 and world > 1 and E == 65536:
        return f"weak scaling of BASELINE configs[2] ({E_global} envs in total)"
    return "custom workload (no BASELINE config)"


def _median_pass(times):
    """(median, min, max) of the passes' elapsed times; the median of an even count is the upper middle one (a pass that ran)."""
    t = sorted(times)
    return t[len(t) // 2], t[0], t[-1]



# ---------------------------------------------------------------------------------------------------------------------
# the record: ONE compact JSON line on stdout (the driver keeps the last ~8 KB of stdout: a longer line is lost — round 5's
# 21 KB line was), everything else in a side file
# ---------------------------------------------------------------------------------------------------------------------
COMPACT_LIMIT = 4000         # bytes: tests/test_host_logic.py holds the line to this


def _r(x, digits=5):
    """A float at `digits` significant digits (the line is a record, not a checkpoint); anything else unchanged; NaN / inf -> None
    (strict JSON has neither)."""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{digits}g}")
    return x


def _short_workload(w):
    """"BASELINE configs[1]: 4096 envs x (37 + 26) rays, K=200, ..." -> label + what differs from the headline."""
    head, _, rest = w.partition(": ")
    task = head.startswith("RoverTask")                      # the task-level entries: "RoverTask.pre_physics_step + ... (<label>)"
    if task:
        head = head[head.find("(") + 1:head.rfind(")")]
    head = head.replace(" (32 768 of its 262 144 envs, no gather)", "")
    keep = [p for p in rest.split(", ") if p.startswith(("mesh=", "ray_precision=", "+ goal")) and p not in ("mesh=grid", "ray_precision=fp32")]
    envs = rest.split(", ")[0] if rest else ""
    label = head if "BASELINE configs" in head else "custom"
    return ", ".join((["RoverTask pre+post_physics_step"] if task else []) + [label, envs] + keep)


def compact_line(full, full_path=None):
    """The line the driver parses: the contract's keys, the roofline and cpu_baseline objects at their required fields, `also` at
    five fields per workload.  `full` (every number this run measured) goes to `full_path`."""
    rf = full.get("roofline") or {}
    line = {k: full[k] for k in
            ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = full["config"]
    line["config"] = {k: cfg[k] for k in ("workload", "envs_total", "rays_per_env", "obs_dim", "algorithmic_bytes_per_env_step") if k in cfg}
    ps = full.get("passes") or {}
    line["passes"] = {k: ps.get(k) for k in ("n", "min_ms_per_step", "max_ms_per_step")}
    line["passes"]["spread"] = _r(ps.get("spread"))
    line["roofline"] = {k: _r(rf.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms",
                                                   "stall_frac", "limited_by", "nearest_roof", "profile_key", "profile_stale")}
    line["roofline"]["valu_frac"] = [_r(x) for x in rf["valu_frac_bracket"]] if rf.get("valu_frac_bracket") else None
    line["roofline"]["algorithmic_equiv_GBps"] = _r(rf.get("algorithmic_equiv_GBps"))
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "cpu_model": cb.get("cpu_model"),
                                "sample": cb.get("sample_short") or cb.get("sample")}
        tr = cb.get("torch_ref") or {}
        for name in ("fp32", "fp16_as_shipped"):
            if isinstance(tr.get(name), dict):
                line["cpu_baseline"]["torch_ref_" + name] = _r(tr[name]["value"])
    for k in ("rccl_ranks", "backend", "lib", "lib_built_from_tree", "gather_check", "gather_bytes_per_rank_per_step",
              "gather_messages_per_peer_per_step"):
        if k in full:
            line[k] = full[k]
    if full.get("per_rank"):
        line["per_rank"] = {k: [_r(x) for x in v] for k, v in full["per_rank"].items()}
    for k in ("alt_sync_gather", "alt_overlapped"):
        if k in full:
            a = full[k]
            line[k] = {"value": _r(a["value"], 7), "ms_per_step": _r(a["ms_per_step"], 7), "gather_check": a.get("gather_check")}
    if full.get("cull"):
        line["cull"] = {k: _r(full["cull"].get(k)) for k in ("candidate_pairs_per_ray", "rays_not_scanned")}
    if full.get("also"):
        line["also"] = {}
        for name, a in full["also"].items():
            if "error" in a:
                line["also"][name] = {"error": str(a["error"])[:120]}
                continue
            e = {"value": _r(a["value"]), "ms_per_step": _r(a["ms_per_step"]), "dtype": a.get("dtype"),
                 "workload": _short_workload(a.get("workload", "")), "frac": _r((a.get("roofline") or {}).get("frac"))}
            for k in ("host_enqueue_ms", "engine_step_ms"):          # the task-level entries
                if k in a:
                    e[k] = _r(a[k])
            line["also"][name] = e
    if full_path:
        line["full_record"] = full_path
    return line


def write_full(full):
    """The whole record as indented JSON: gpurun_out/bench_full.json when that directory exists (it is merged back from the GPU box),
    else next to bench.py, else the temp dir; -> the path written (None if nowhere is writable)."""
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT, tempfile.gettempdir()):
        if not os.path.isdir(d):
            continue
        path = os.path.join(d, "bench_full.json")
        try:
            with open(path, "w") as f:
                json.dump(full, f, indent=1, default=str)
            return os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
        except OSError:
            continue
    return None


_SCENES = {}


def _checksums(torch, obs, rew, done):
    """Order-independent exact checksums of one (obs, rew, done) set: int64 sums of the raw bits."""
    return torch.stack((obs.contiguous().view(torch.int32).sum(dtype=torch.int64),
                        rew.contiguous().view(torch.int32).sum(dtype=torch.int64),
                        done.sum(dtype=torch.int64)))


def measure_also(args, device, local_rank, **override):
    """One more workload in THIS process (N = 1): the same batch size, ray set and maps as the headline with `override` applied
    (mesh="irregular" / ray_precision="fp16_as_shipped"): scene, engine, four resident state batches, the headline's warm-up rule
    (clocks are already up: no pre-roll), >= 20 timed steps between device synchronisations, the ray-cast launch bracketed by HIP
    events on every `--event-every`-th step.  Returns the line's sub-object (value, ms_per_step, ray-cast ms, its own roofline)."""
    import argparse as _ap
    import numpy as np
    import torch
    from isaac_rover_amd import _lib, synth
    a = _ap.Namespace(**vars(args))
    for k, v in override.items():
        setattr(a, k, v)
    E = a.envs_per_gpu
    skey = (a.mesh, a.cells, a.k, a.stones)
    if skey not in _SCENES:
        _SCENES.clear()                      # one scene resident at a time
        _SCENES[skey] = load_scene(a, device, local_rank)
    scene, zf = _SCENES[skey]
    if a.rays == "native":
        from isaac_rover_amd.tasks.utils.heightmap_distribution import generate_native
        distn = tuple(np.asarray(x) for x in generate_native())
    else:
        distn = synth.ray_distribution(a.rays)
    n_rays = int(distn[0].shape[0])
    eng = _make_engine(E, local_rank, E, 0)
    eng.set_scene(scene, distn)
    eng.set_option("ray_precision", {"fp32": 0, "fp16_sources": 1, "fp16_as_shipped": 2}[a.ray_precision])
    eng.set_option("cell_index_mode", {"cpu_div": 0, "cuda_rcp": 1}[a.cell_index_mode])
    info = eng.info()
    W = eng.num_observations
    f, i64 = torch.float32, torch.int64
    batches = []
    for b in range(4):
        st = synth.make_states(E, a.cells * 0.1, seed=b, heightfn=zf)
        batches.append({k: v.to(device) for k, v in st.items()})
    obs = torch.zeros(E, W, device=device)
    rew = torch.zeros(E, device=device)
    done = torch.zeros(E, dtype=torch.uint8, device=device)
    reset = torch.ones(E, dtype=i64, device=device)
    rock = torch.zeros(E, dtype=i64, device=device)
    extras = {k: torch.zeros(E, dtype=(i64 if k == "collision_penalty" else f), device=device) for k in _lib.EXTRAS}
    reset_ids = torch.zeros(E, dtype=i64, device=device)
    n_reset = torch.zeros(1, dtype=torch.int32, device=device)
    stone = torch.zeros(E, dtype=i64, device=device)
    sout = eng.make_out(obs, rew=rew, reset=reset, rock_collision=rock, extras=extras, reset_ids=reset_ids, n_reset=n_reset,
                        stone_collision=stone, stone_margin=0.0, done_u8=done)
    sins = [eng.make_in(b["pos"], b["quat"], b["joints"], b["target"], b["lin_hist"], b["ang_hist"], b["euler_pre"], b["progress"])
            for b in batches]
    steps = max(20, int(a.also_steps))
    if a.validate_goals:
        n_used = torch.zeros(1, dtype=torch.int32, device=device)
        initial = [b["pos"].clone() for b in batches]
        joint_vel = torch.zeros(E, 13, device=device)

    def step(i):
        eng.step(sins[i % 4], sout, increment_progress=True, compact=True)
        if a.validate_goals:                 # configs[4]: the device-side reset + goal re-draw / validation of the envs the step flagged done
            st = batches[i % 4]
            eng.reset_envs(reset_ids, initial[i % 4], st["pos"], st["quat"], reset, st["progress"], n_reset_dev=n_reset,
                           joint_pos13=st["joints"], joint_vel13=joint_vel, target3=st["target"], radius=8.0, seed=i,
                           max_draws=256, n_draws_used=n_used)

    eng.set_profiling(True)
    if a.preroll_ms > 0:                     # the scene build left the GPU idle: the same untimed clock ramp as the headline gets
        t_pre, i_pre = time.perf_counter(), 0
        while time.perf_counter() - t_pre < a.preroll_ms * 1e-3:
            for _ in range(8):
                step(i_pre)
                i_pre += 1
            _sync()
    for i in range(max(10, a.warmup)):
        step(i)
    _sync()
    eng.set_profiling(True, every=a.event_every)
    times, i_step = [], 0
    for _ in range(max(1, int(a.also_passes))):
        t0 = time.perf_counter()
        for _i in range(steps):
            step(i_step)
            i_step += 1
        _sync()
        times.append(time.perf_counter() - t0)
    elapsed, t_min, t_max = _median_pass(times)
    prof = eng.get_profile()
    eng.set_profiling(False)
    out = {"value": E * steps / elapsed, "unit": "env-steps/s", "ms_per_step": 1e3 * elapsed / steps, "steps": steps,
           "passes": {"n": len(times), "min_ms_per_step": 1e3 * t_min / steps, "max_ms_per_step": 1e3 * t_max / steps,
                      "spread": (t_max - t_min) / elapsed},
           "raycast_ms": prof.raycast_ms / max(prof.launches, 1), "dtype": "f16" if a.ray_precision == "fp16_as_shipped" else "f32",
           "workload": f"{config_label(a, E, 1, E)}: {E} envs x ({a.rays} + 26) rays, K={a.k}, {a.cells}x{a.cells} cells, mesh={a.mesh}, "
                       f"ray_precision={a.ray_precision}, cell_index_mode={a.cell_index_mode}"
                       + (", + goal validation" if a.validate_goals else ""),
           "raycast_variant": int(info.raycast_variant),
           "roofline": roofline(a, E, n_rays, prof, info, _lib_version())}
    if info.raycast_variant >= 3:
        ci = eng.cull_info()
        out["cull"] = {"candidate_pairs_per_ray": ci["pairs_per_ray"], "rays_with_both_tests": ci["rays_both_tests"] / max(ci["rays"], 1),
                       "rays_far_skipped": ci["rays_far_skipped"] / max(ci["rays"], 1), "triangles": ci["triangles"]}
    eng.close()
    del batches, sins, sout, obs
    torch.cuda.empty_cache()
    return out



def measure_task_api(args, device, local_rank, steps=200, graph=None, **override):
    """The drop-in surface itself (N = 1): what the reference's caller invokes per env.step() — `RoverTask.pre_physics_step(actions)`
    (rover.py:338-414: done compaction consumed, device-side reset_idx + set_targets, history, Ackermann) and
    `RLTask.post_physics_step()` (rl_task.py:239-259: the fused rover_step) — on static poses (the pose feeder / physics stand-in is
    skipped; envs the step flags done are reset to their spawn by the task itself, as in training).  Reports wall ms per step, the host's
    enqueue time per step (the Python + ctypes cost of the task layer) and, next to them, `Engine.step` alone on the same buffers."""
    import argparse as _ap
    import numpy as np
    import torch
    from isaac_rover_amd import config as rcfg, synth, vec_env
    from isaac_rover_amd.tasks.rover import RoverTask
    a = _ap.Namespace(**vars(args))
    for k, v in override.items():
        setattr(a, k, v)
    E = a.envs_per_gpu
    skey = (a.mesh, a.cells, a.k, a.stones)
    if skey not in _SCENES:
        _SCENES.clear()
        _SCENES[skey] = load_scene(a, device, local_rank)
    scene, zf = _SCENES[skey]
    distn = None if a.rays == "native" else synth.ray_distribution(a.rays)
    cfg = rcfg.SimConfig(num_envs=E, device=f"cuda:{local_rank}")
    kw = {} if graph is None else {"graph": graph}
    task = RoverTask("Rover", cfg, vec_env.VecEnv(headless=True), scene=scene, distribution=distn, fused=True, device_reset=True,
                     ray_precision=a.ray_precision, cell_index_mode=a.cell_index_mode, stone_mask_margin=0.0, **kw)
    st = synth.make_states(E, a.cells * 0.1, seed=7, heightfn=zf)
    task.set_up_scene(spawn_positions=st["pos"].to(device))
    task.post_reset()
    # Spawn poses = SURVEY 8(d)'s positions themselves: the synthetic scene's 1 024 stones leave `avoid_pos_rock_collision` (clearance
    # > 1.4 m, rover.py:649-661) almost no free ground — it walks most rovers to the map's edge, where thousands of rays share the border
    # cells (measured: Engine.step 0.575 ms on those poses against 0.47 ms) — a property of the stone density, not of the task layer.
    task.initial_pos.copy_(st["pos"].to(device))
    task._rover.feed(positions=task.initial_pos, orientations=st["quat"].to(device), joint_positions=st["joints"].to(device))
    task.reset()
    g = torch.Generator().manual_seed(3)
    acts = [(2 * torch.rand(E, 2, generator=g) - 1).to(device) for _ in range(8)]

    def step(i):
        task.pre_physics_step(acts[i % 8])
        return task.post_physics_step()

    for i in range(40):                      # past global step 10 (the curriculum switch, rover.py:344-353) and any graph capture
        step(i)
    _sync()
    t_pre, i_pre = time.perf_counter(), 0    # the task's construction left the GPU idle: the same untimed clock ramp as the headline gets
    while time.perf_counter() - t_pre < a.preroll_ms * 1e-3:
        for _ in range(8):
            step(i_pre)
            i_pre += 1
        _sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    t_enq = time.perf_counter() - t0
    _sync()
    t_all = time.perf_counter() - t0
    n_done = int(task.reset_buf.sum().item())
    eng = task._engine
    for i in range(10):
        eng.step(task._sin, task._sout, increment_progress=True, compact=True)
    _sync()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.step(task._sin, task._sout, increment_progress=True, compact=True)
    _sync()
    t_eng = time.perf_counter() - t0
    info = eng.info()
    out = {"value": E * steps / t_all, "unit": "env-steps/s", "ms_per_step": 1e3 * t_all / steps, "steps": steps,
           "host_enqueue_ms": 1e3 * t_enq / steps, "engine_step_ms": 1e3 * t_eng / steps,
           "task_over_engine": t_all / t_eng, "graph": bool(getattr(task, "_use_graph", False)),
           "envs_done_last_step": n_done, "dtype": "f16" if a.ray_precision == "fp16_as_shipped" else "f32",
           "raycast_variant": int(info.raycast_variant),
           "workload": f"RoverTask.pre_physics_step + post_physics_step ({config_label(a, E, 1, E)}): {E} envs x ({a.rays} + 26) rays, "
                       f"K={a.k}, {a.cells}x{a.cells} cells, mesh={a.mesh}, ray_precision={a.ray_precision}, device reset + goal validation"}
    task.close()
    del task, acts
    torch.cuda.empty_cache()
    return out


def also_workloads(args, device, local_rank):
    """The other single-GPU workloads of BASELINE.json and the two representative variants of the headline, each under this run's
    clock (the driver times one command): configs[1], configs[4], one rank's shard of configs[3]; the geometry the reference's real
    terrain has (a decimated mesh, utils/terrain_utils/terrain_generation.py:217-243) and its real arithmetic (fp16, camera.py:55); and
    the reference's own operating point — numEnvs 512 (cfg/task/Rover.yaml:11) x its native 1 634 + 26 rays on that mesh — in both
    arithmetics (the irregular scene of the entry before them is still resident)."""
    todo = [("configs1", measure_also, dict(envs_per_gpu=4096, also_steps=max(200, args.also_steps))),
            ("task_api_configs1", measure_task_api, dict(envs_per_gpu=4096, steps=300)),
            ("task_api_configs2", measure_task_api, dict(steps=100)),
            ("configs3_shard", measure_also, dict(envs_per_gpu=32768)),
            ("configs4", measure_also, dict(rays="120", validate_goals=True)),
            ("fp16_as_shipped", measure_also, dict(ray_precision="fp16_as_shipped")),
            ("mesh_irregular", measure_also, dict(mesh="irregular")),
            ("ref_operating_point", measure_also, dict(mesh="irregular", rays="native", envs_per_gpu=512, also_steps=max(500, args.also_steps))),
            ("task_api_ref_operating_point", measure_task_api, dict(mesh="irregular", rays="native", envs_per_gpu=512, steps=300)),
            ("ref_operating_point_as_shipped", measure_also, dict(mesh="irregular", rays="native", envs_per_gpu=512,
                                                                  ray_precision="fp16_as_shipped", also_steps=max(500, args.also_steps)))]
    out = {}
    for name, fn, kw in todo:
        if args.also_only and name not in args.also_only.split(","):
            continue
        try:
            out[name] = fn(args, device, local_rank, **kw)
        except Exception as e:           # one extra workload failing must not lose the headline (it is reported in its place)
            print(f"bench.py: also[{name}] failed: {e!r}", file=sys.stderr)
            out[name] = {"error": repr(e)}
    return out


def run_rank(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    device = _device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        _init_process_group(dist, device)

    from isaac_rover_amd import _lib, synth
    from isaac_rover_amd.distributed import StepGather

    E = args.envs_per_gpu
    E_global = E * world
    scene, zf = load_scene(args, device, local_rank)
    if args.rays == "native":
        from isaac_rover_amd.tasks.utils.heightmap_distribution import generate_native
        distn = tuple(np.asarray(x) for x in generate_native())
    else:
        distn = synth.ray_distribution(args.rays)
    n_rays = int(distn[0].shape[0])
    eng = _make_engine(E, local_rank, E_global, rank * E)
    eng.set_scene(scene, distn)
    eng.set_option("ray_precision", {"fp32": 0, "fp16_sources": 1, "fp16_as_shipped": 2}[args.ray_precision])
    eng.set_option("cell_index_mode", {"cpu_div": 0, "cuda_rcp": 1}[args.cell_index_mode])
    info = eng.info()
    W = eng.num_observations

    # 4 resident state batches, rotated, so consecutive steps do not replay identical rays
    batches = []
    for b in range(4):
        st = synth.make_states(E, args.cells * 0.1, seed=100 * rank + b, heightfn=zf)
        batches.append({k: v.to(device) for k, v in st.items()})
    can_overlap = world > 1 and not args.validate_goals
    depth = 2 if can_overlap else 1
    gather = StepGather(E, W, device, world, rank, depth=depth)
    reset = torch.ones(E, dtype=torch.int64, device=device)          # reset_buf (rl_task.py:105); the u8 copy travels
    rock = torch.zeros(E, dtype=torch.int64, device=device)
    extras = {k: torch.zeros(E, dtype=(torch.int64 if k == "collision_penalty" else torch.float32), device=device)
              for k in _lib.EXTRAS}
    reset_ids = torch.zeros(E, dtype=torch.int64, device=device)
    n_reset = torch.zeros(1, dtype=torch.int32, device=device)
    # configs[2]/[4]: "+ stone_info collision mask" — the additional occupancy-mask output of the step (margin 0 m)
    stone_mask = torch.zeros(E, dtype=torch.int64, device=device)
    souts = []
    for d in range(depth):
        obs_d, rew_d, done_d = gather.local_views(d)
        souts.append(eng.make_out(obs_d, rew=rew_d, reset=reset, rock_collision=rock, extras=extras, reset_ids=reset_ids,
                                  n_reset=n_reset, stone_collision=stone_mask, stone_margin=0.0, done_u8=done_d))
    sins = [eng.make_in(b["pos"], b["quat"], b["joints"], b["target"], b["lin_hist"], b["ang_hist"], b["euler_pre"],
                        b["progress"]) for b in batches]
    if args.validate_goals:
        n_used = torch.zeros(1, dtype=torch.int32, device=device)
        initial = [b["pos"].clone() for b in batches]
        joint_vel = torch.zeros(E, 13, device=device)

    graphs = {}

    def launch_step(b, d):
        if not args.graph:
            eng.step(sins[b], souts[d], increment_progress=True, compact=True)
            return
        g = graphs.get((b, d))
        if g is None:                        # one captured graph per (state batch, output set)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                eng.step(sins[b], souts[d], increment_progress=True, compact=True)
            graphs[(b, d)] = g
        g.replay()

    wait_events = []                         # N > 1: (before, after) pairs around every point where the compute stream waits for a transfer
    sampled_steps = [0]

    def one_step(i, overlap, timed=False):
        b = i % len(batches)
        d = i % depth
        # (timing events on the compute stream cost it ~6 us each: like the in-library ray-cast events they bracket every
        #  --event-every-th step only, and the waits they measure are scaled up — the N > 1 headline then carries what the N = 1 one does)
        ev = max(1, args.event_every)
        sampled = timed and world > 1 and ((i + i // ev) % ev) == 0      # one step in `ev`, its phase moving on by one each time: both buffer sets are sampled
        e0 = _gpu_event() if sampled else None
        gather.wait(d)                       # overlapped mode: the transfer that last read buffer set d must be through
        if e0 is not None:
            wait_events.append((e0, _gpu_event()))
            sampled_steps[0] += 1
        launch_step(b, d)
        e0 = _gpu_event() if (sampled and not overlap) else None
        gather.gather(d, wait=not overlap)   # (obs, rew, done) of this step to the learner rank
        if e0 is not None:
            wait_events.append((e0, _gpu_event()))
        if args.validate_goals:
            # configs[4]: reset_idx + set_targets (goal re-draw + stone-clearance validation + goal z) for the envs the
            # step flagged done, consuming the compacted ids on the device — no host sync (rover.py:356-361 has one)
            st = batches[b]
            eng.reset_envs(reset_ids, initial[b], st["pos"], st["quat"], reset, st["progress"], n_reset_dev=n_reset,
                           joint_pos13=st["joints"], joint_vel13=joint_vel, target3=st["target"], radius=8.0, seed=i,
                           max_draws=256, n_draws_used=n_used)

    def fence():
        for d in range(depth):
            gather.wait(d)
        _sync()
        if world > 1:
            dist.barrier()
            _sync()

    def gather_check():
        """Root-side (obs, rew, done) of every buffer set == what every rank sent (exact integer checksums)."""
        if world == 1:
            return None
        mine = torch.stack([_checksums(torch, *gather.local_views(d)) for d in range(depth)])       # [depth, 3]
        allsums = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allsums, mine)
        ok = True
        if rank == 0:
            for d in range(depth):
                og, rg, dg = gather.global_views(d)            # [N, E, W], [N, E], [N, E]: shard r = rank r's packed message
                for r in range(world):
                    ok = ok and bool(torch.equal(_checksums(torch, og[r], rg[r], dg[r]).cpu(), allsums[r][d].cpu()))
        return ok

    def timed_pass(first_step, overlap, check=True):
        fence()
        t0 = time.perf_counter()
        stamps, evs = [], []
        wait_events.clear()
        sampled_steps[0] = 0
        for i in range(args.steps):
            one_step(first_step + i, overlap, timed=True)
            if args.debug_timing:
                stamps.append(time.perf_counter())
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                evs.append(ev)
        t_enq = time.perf_counter()
        fence()
        elapsed = time.perf_counter() - t0
        if args.debug_timing and rank == 0:
            dd = [1e6 * (b - a) for a, b in zip([t0] + stamps[:-1], stamps)]
            print("gpu ms/step:", " ".join(f"{a.elapsed_time(b):.2f}" for a, b in zip(evs[:-1], evs[1:])), file=sys.stderr)
            print("enqueue us/step:", " ".join(f"{x:.0f}" for x in dd), "| enqueue total ms", 1e3 * (t_enq - t0),
                  "| fence ms", 1e3 * (elapsed - (t_enq - t0)), file=sys.stderr)
        ok = gather_check() if check else None
        per_rank = None
        if world > 1:
            # per rank: its own elapsed time and the time its compute stream spent waiting for (obs, rew, done) transfers — EXTRAPOLATED
            # from the sampled steps (every --event-every-th) to all of them
            waited = sum(a.elapsed_time(b) for a, b in wait_events if a is not None and b is not None) * 1e-3 \
                * (args.steps / max(1, sampled_steps[0]))
            mine = torch.tensor([elapsed, waited], dtype=torch.float64, device=device)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            per_rank = {"ms_per_step": [1e3 * float(x[0]) / args.steps for x in every],
                        "transfer_wait_ms_per_step_extrapolated": [1e3 * float(x[1]) / args.steps for x in every]}
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, ok, per_rank

    def timed_passes(first_step, overlap, n):
        """n back-to-back passes of exactly --steps steps, each bracketed by barrier + device sync on both sides with the max over the
        ranks as its time; -> (median pass's time, min, max, gather check of the last pass, per-rank record of the median pass, steps run)"""
        eng.set_profiling(True, every=args.event_every)      # resets the in-library ray-cast event counters
        recs = []
        for p in range(n):
            e, ok_p, pr = timed_pass(first_step + p * args.steps, overlap, check=(p == n - 1))
            recs.append((e, ok_p, pr))
        med, t_min, t_max = _median_pass([r[0] for r in recs])
        pr_med = next(r[2] for r in recs if r[0] == med)
        return med, t_min, t_max, recs[-1][1], pr_med, n * args.steps

    # Event timing is switched on BEFORE the warm-up: the first timed hipEventRecord on a stream makes the runtime
    # enable queue profiling once (tens of ms); the warm-up absorbs that, then the counters are reset.
    eng.set_profiling(True)
    if args.preroll_ms > 0:                  # clock ramp: not part of the W warm-up steps or the K timed steps
        t_pre, i_pre = time.perf_counter(), 0
        while time.perf_counter() - t_pre < args.preroll_ms * 1e-3:      # local compute only: no collective, so the
            for _ in range(8):                                           # ranks need not agree on the iteration count
                launch_step(i_pre % len(sins), 0)
                i_pre += 1
            _sync()
        if world > 1:
            dist.barrier()
    overlap = can_overlap and not args.sync_gather
    for i in range(args.warmup):
        one_step(i, overlap)
    n_pass = max(1, args.passes)
    elapsed, t_min, t_max, ok, per_rank, ran = timed_passes(args.warmup, overlap, n_pass)
    prof = eng.get_profile()
    alt = None
    if world > 1 and can_overlap and not args.no_alt_pass:
        e2, e2_min, e2_max, ok2, pr2, _ = timed_passes(args.warmup + ran, not overlap, min(5, n_pass))
        alt = {"mode": "overlapped" if not overlap else "sync_gather", "value": E_global * args.steps / e2,
               "ms_per_step": 1e3 * e2 / args.steps, "gather_check": ok2, "per_rank": pr2}
    eng.set_profiling(False)
    if world > 1:            # every rank says who it was (a run that dies later still leaves this on stderr)
        print(f"bench.py: rank {rank}/{world} on cuda:{local_rank} ({torch.cuda.get_device_name(local_rank) if torch.cuda.is_available() else 'cpu'}), "
              f"backend {dist.get_backend()}, {1e3 * elapsed / args.steps:.4f} ms per step (median of {n_pass} passes)", file=sys.stderr, flush=True)
    # the kernel the line reports is the kernel that ran: the library has no silent change of ray-cast variant, and this guards it
    if eng.info().raycast_variant != info.raycast_variant:
        raise SystemExit(f"bench.py: ray-cast variant changed during the run ({info.raycast_variant} -> {eng.info().raycast_variant})")

    rc = 0
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = E_global * args.steps / elapsed
        cfg_name = config_label(args, E, world, E_global)
        line = {
            "metric": "env-steps/sec (obs+reward+done)", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "passes": {"n": n_pass, "statistic": "median pass of --steps steps (barrier + device sync on both sides, max over ranks)",
                       "min_ms_per_step": 1e3 * t_min / args.steps, "max_ms_per_step": 1e3 * t_max / args.steps,
                       "spread": (t_max - t_min) / elapsed},
            "scaling": "weak", "vs_baseline": None, "dtype": "f16" if args.ray_precision == "fp16_as_shipped" else "f32",
            "data": "synthetic",
            "config": {"workload": f"{cfg_name}: {E} envs/GPU x {world} GPU, "
                                   f"{args.rays}-point heightmap + 26 rock rays, K={args.k}, {args.cells}x{args.cells} "
                                   f"cells @0.1 m, stone_info mask over {args.stones} stones"
                                   + (f", mesh={args.mesh}" if args.mesh != "grid" else "")
                                   + (", + goal validation" if args.validate_goals else "")
                                   + (f", ray_precision={args.ray_precision}" if args.ray_precision != "fp32" else "")
                                   + f", cell_index_mode={args.cell_index_mode}"
                                   + (", step replayed from a hipGraph" if args.graph else "")
                                   + ((", RCCL gather(one packed message per rank: obs f32 | rew f32 | done u8)->rank0"
                                       + (" overlapped with the next step" if overlap else " serialised with the steps"))
                                      if world > 1 else ""),
                       "envs_total": E_global, "rays_per_env": n_rays + 26, "obs_dim": W,
                       "algorithmic_bytes_per_env_step": algorithmic_bytes_per_env_step(n_rays, args.k, eng.Ns, eng.Nd),
                       "table_bytes": int(info.table_bytes[0] + info.table_bytes[1])},
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "backend": dist.get_backend() if world > 1 else None,
            "roofline": roofline(args, E, n_rays, prof, info, _lib_version()),
            "lib": _lib_version(),
            "lib_built_from_tree": _lib_matches_tree(),      # false: the .so was built from other sources than the ones next to it
        }
        if info.raycast_variant >= 3:
            ci = eng.cull_info()
            line["cull"] = {"candidate_pairs_per_ray": ci["pairs_per_ray"], "rays_with_both_tests": ci["rays_both_tests"] / max(ci["rays"], 1),
                            "rays_per_bin": ci["rays"] / max(ci["bins"], 1), "max_pairs_per_run": ci["max_pairs_per_run"],
                            "rays_far_skipped": ci["rays_far_skipped"] / max(ci["rays"], 1),
                            "rays_not_scanned": ci["rays_not_scanned"] / max(ci["rays"], 1),
                            "always_candidate_triangles": ci["always_candidate_triangles"], "cells_without_cone": ci["cells_without_cone"],
                            "triangles": ci["triangles"], "queue_bytes": ci["queue_bytes"],
                            "cells_with_far_bound": ci["cells_with_far_bound"], "far_records_on_demand": bool(ci["far_records_on_demand"])}
            if info.raycast_variant == 4:
                line["cull"].update({"lane_items_per_ray": ci["lane_items"] / max(ci["rays"], 1),
                                     "lane_flushes": ci["lane_flushes"]})
        if world > 1:
            line["gather_check"] = ok
            line["per_rank"] = per_rank          # [rank]: ms per step of that rank; ms per step its compute stream waited for a transfer (rank 0 = the root's receive time that was not hidden)
            line["gather_bytes_per_rank_per_step"] = E * (4 * W + 4 + 1)
            line["gather_messages_per_peer_per_step"] = 1                   # the packed shard (isaac_rover_amd.distributed.shard_bytes, padded to 256 B)
            if alt is not None:
                line["alt_" + alt.pop("mode")] = alt
            if ok is False or (alt is not None and alt.get("gather_check") is False):
                rc = 3
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args, scene, distn, batches[0])
        default_workload = (world == 1 and args.mesh == "grid" and args.ray_precision == "fp32" and args.rays == "37"
                            and not args.validate_goals and not args.graph and E == 65536)
        if default_workload and not args.no_also:
            # the representative workloads under the same clock: the geometry the reference's real terrain has (a decimated mesh,
            # utils/terrain_utils/terrain_generation.py:217-243) and its real arithmetic (fp16, camera.py:55) — headline fields unchanged
            eng.close()
            del sins, souts, batches
            torch.cuda.empty_cache()
            _SCENES[(args.mesh, args.cells, args.k, args.stones)] = (scene, zf)
            line["also"] = also_workloads(args, device, local_rank)
        out_line = json.dumps(compact_line(line, write_full(line)), allow_nan=False, separators=(",", ":"))
        if len(out_line) > COMPACT_LIMIT:
            print(f"bench.py: the record line is {len(out_line)} bytes (> {COMPACT_LIMIT}): a driver that keeps a bounded tail may lose it",
                  file=sys.stderr)
        print(out_line, flush=True)
        if line["lib_built_from_tree"] is False:
            # a number measured on a library built from OTHER sources than the ones next to it is not a number of this tree
            print("bench.py: librover_step.so was not built from the sources in the tree (run isaac_rover_2.0_amd/csrc/build.sh)",
                  file=sys.stderr)
            rc = rc or 4
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def main(argv=None):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # (before anything touches HIP: a launcher's ranks need dmabuf IPC for RCCL on this host)
    args = parse(argv)
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and (world_env is None or (world_env == "1" and "RANK" not in os.environ)):
        return launch_ranks(args.gpus, sys.argv[1:] if argv is None else argv, rank_timeout_s=args.rank_timeout_s)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
