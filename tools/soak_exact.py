"""Bit-exactness soak (run on the GPU box): the culled ray cast (variant 3) and the binned ray cast with the conservative
early out against the env-order kernel (no early out, different traversal) and against the binned kernel with the option off, over many seeds, arbitrary rover
orientations (rays parallel to facets included), several K and both precisions.  Prints the number of compared rays.

    python tools/soak_exact.py [rounds] [irregular_rounds]

Second leg: IRREGULAR meshes (synth.irregular_mesh: non-uniform Delaunay triangulation, edges from millimetres to metres, ~80
degree rock flanks, needle / zero-area triangles, duplicated vertices, mixed windings, shuffled ids), K-nearest maps by the GPU
builder, K = 200 / 64.  Prints how often the paths a regular grid never takes were taken (rover_get_cull_info).
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isaac_rover_amd import _lib, synth

print("library:", _lib.version(), flush=True)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
irr_rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
E = 32768
total = 0


def compare(outs, pairs, label):
    for a, b in pairs:
        for x, y, what in zip(outs[a], outs[b], ("ray", "wheel", "body", "coll", "reset")):
            same = torch.equal(x, y) or bool(((x == y) | (x.isnan() & y.isnan())).all())
            if not same:
                bad = (x != y).nonzero()[:5]
                raise SystemExit(f"MISMATCH {a} vs {b} {what} {label}: {bad.tolist()} {x[tuple(bad[0])]} {y[tuple(bad[0])]}")


def run_engines(engs, st, P):
    d = {kk: v.cuda().contiguous() for kk, v in st.items()}
    outs = {}
    for name, e in engs.items():
        sin = e.make_in(d["pos"], d["quat"], d["joints"], d["target"], d["lin_hist"], d["ang_hist"], d["euler_pre"], d["progress"].clone())
        obs = torch.zeros(E, e.num_observations, device="cuda")
        bufs = dict(rew=torch.zeros(E, device="cuda"), reset=torch.zeros(E, dtype=torch.int64, device="cuda"),
                    rock_collision=torch.zeros(E, dtype=torch.int64, device="cuda"), ray_dist=torch.zeros(E, P, device="cuda"),
                    wheel_dist=torch.zeros(E, 24, device="cuda"), body_dist=torch.zeros(E, 2, device="cuda"))
        e.step(sin, e.make_out(obs, **bufs), increment_progress=False)
        torch.cuda.synchronize()
        outs[name] = (bufs["ray_dist"], bufs["wheel_dist"], bufs["body_dist"], bufs["rock_collision"], bufs["reset"])
    return outs


def make_engine(name):
    """"culled" is the scan kernel the library picks for the scene and ray set; "culled_other" the other one (far records on demand +
    rays that clear their whole cell not scanned, or neither), forced through the experiment variable read at rover_create."""
    if name != "culled_other":
        return _lib.Engine(E, device=0)
    probe = _lib.Engine(E, device=0)
    return probe            # (the choice needs the scene: set below, see other_kernel())


def other_kernel(engs, scene, distn):
    """Re-create engs["culled_other"] with the scan kernel "culled" did NOT get."""
    lazy = engs["culled"].cull_info()["far_records_on_demand"]
    engs["culled_other"].close()
    os.environ["ROVER_CULL_LAZY"] = "0" if lazy else "1"
    try:
        e = _lib.Engine(E, device=0)
    finally:
        os.environ.pop("ROVER_CULL_LAZY", None)
    e.set_scene(scene, distn)
    e.set_option("raycast_variant", 3); e.set_option("raycast_early_out", 1); e.set_option("ray_precision", 0)
    assert e.cull_info()["far_records_on_demand"] != lazy
    engs["culled_other"] = e


irr_total = 0
from isaac_rover_amd import assets
for seed, k, extent, n_rocks, coarse, fine, dist_name in ((11, 200, 24.0, 110, 1.2, 0.0375, "120"), (12, 64, 30.0, 60, 3.0, 0.05, "37"),
                                                         (13, 200, 16.0, 90, 0.6, 0.03, "120")):
    if irr_rounds <= 0:
        break
    spec = synth.IrregularSpec(extent_x=extent, extent_y=extent, n_rocks=n_rocks, seed=seed, coarse=coarse, fine=fine)
    tool = _lib.Engine(8, device=0)
    scene, zf = assets.build_irregular_scene(tool, spec, k)
    tool.close()
    _zf, (rxy, rr, _rh, _rp) = synth.irregular_height(spec)
    distn = synth.ray_distribution(dist_name)
    P = distn[0].shape[0]
    engs = {}
    for name, (variant, early, prec) in {"culled": (3, 1, 0), "culled_other": (3, 1, 0), "binned": (2, 1, 0), "binned_noearly": (2, 0, 0), "envorder": (1, 0, 0),
                                         "h": (2, 1, 2), "h_noearly": (2, 0, 2), "h_culled": (3, 1, 2),
                                         "staged": (4, 1, 0), "staged_rocks": (4, 1, 0), "staged_envorder": (4, 1, 0), "h_staged": (4, 1, 2)}.items():
        e = make_engine(name)
        e.set_scene(scene, distn)
        e.set_option("raycast_variant", variant); e.set_option("raycast_early_out", early); e.set_option("ray_precision", prec)
        if variant == 4:       # the staged kernel: rocks part on the culled kernel / on the staged kernel too / no sort at all
            e.set_option("lane_rocks", 1 if name in ("staged_rocks", "h_staged") else 0)
            e.set_option("lane_env_order", 1 if name == "staged_envorder" else 0)
        engs[name] = e
    other_kernel(engs, scene, distn)
    print(f"irregular seed {seed}: {scene.terrain.triangles.shape[0]} triangles ({scene.rocks.triangles.shape[0]} on rocks), K={k}, {extent} m", flush=True)
    for r in range(irr_rounds):
        st = synth.make_states(E, extent, seed=7000 + 100 * seed + r, heightfn=zf, margin_m=0.5)
        g = torch.Generator().manual_seed(100 * seed + r)
        if r % 4 == 1:      # arbitrary orientations
            q = torch.randn(E, 4, generator=g); st["quat"] = q / q.norm(dim=1, keepdim=True)
        elif r % 4 == 2:    # steep tilts
            st["quat"] = synth.quat_from_euler(0.5 * torch.randn(E, generator=g), 0.5 * torch.randn(E, generator=g), 3.0 * torch.randn(E, generator=g))
        elif r % 4 == 3:    # every rover on a rock flank, 0.3 m above the surface
            i = torch.randint(0, len(rr), (E,), generator=g).numpy()
            a = (6.2831853 * torch.rand(E, generator=g)).numpy()
            f = (0.4 + 0.8 * torch.rand(E, generator=g)).numpy()
            x = np.clip(rxy[i, 0] + f * rr[i] * np.cos(a), 0.2, extent - 0.2); y = np.clip(rxy[i, 1] + f * rr[i] * np.sin(a), 0.2, extent - 0.2)
            st["pos"][:, 0] = torch.from_numpy(x).float(); st["pos"][:, 1] = torch.from_numpy(y).float()
            st["pos"][:, 2] = torch.from_numpy(zf(x, y)).float() + 0.3
        outs = run_engines(engs, st, P)
        compare(outs, (("culled", "envorder"), ("culled_other", "envorder"), ("culled", "binned_noearly"), ("binned", "envorder"), ("binned", "binned_noearly"), ("h", "h_noearly"), ("h_culled", "h_noearly"),
                      ("staged", "binned_noearly"), ("staged_rocks", "binned_noearly"), ("staged_envorder", "binned_noearly"), ("h_staged", "h_noearly")),
                f"irregular seed={seed} K={k} round={r}")
        ci = engs["culled"].cull_info()
        irr_total += E * (P + 26)
        hit = float((outs["binned"][0] < 11.0).float().mean())
        print(f"irregular seed {seed} round {r}: ok, terrain hit rate {hit:.3f}, candidate pairs / ray {ci['pairs_per_ray']:.2f} (max per run {ci['max_pairs_per_run']}), "
              f"rays with both tests {ci['rays_both_tests'] / ci['rays']:.3f}, not scanned {engs['culled_other'].cull_info()['rays_not_scanned'] / ci['rays']:.3f} (forced kernel), always-candidate triangles {ci['always_candidate_triangles']}, "
              f"cells without cone {ci['cells_without_cone']}", flush=True)
    for e in engs.values():
        e.close()
if irr_rounds > 0:
    print(f"irregular soak ok: {irr_total / 1e6:.1f} M rays x 11 comparisons on irregular meshes, all bit-identical", flush=True)

for k, cells, dist_name in ((200, 300, "120"), (100, 200, "37"), (40, 160, "120"), (16, 128, "9")):
    scene = synth.make_scene(n_cells=cells, k=k, n_stones=max(8, cells * cells // 400), device="cuda")
    distn = synth.ray_distribution(dist_name)
    engs = {}
    for name, (variant, early, prec) in {"culled": (3, 1, 0), "culled_other": (3, 1, 0), "binned": (2, 1, 0), "binned_noearly": (2, 0, 0), "envorder": (1, 0, 0),
                                         "h": (2, 1, 2), "h_noearly": (2, 0, 2), "h_culled": (3, 1, 2),
                                         "staged": (4, 1, 0), "staged_rocks": (4, 1, 0), "staged_envorder": (4, 1, 0), "h_staged": (4, 1, 2)}.items():
        e = make_engine(name)
        e.set_scene(scene, distn)
        e.set_option("raycast_variant", variant); e.set_option("raycast_early_out", early); e.set_option("ray_precision", prec)
        if variant == 4:       # the staged kernel: rocks part on the culled kernel / on the staged kernel too / no sort at all
            e.set_option("lane_rocks", 1 if name in ("staged_rocks", "h_staged") else 0)
            e.set_option("lane_env_order", 1 if name == "staged_envorder" else 0)
        engs[name] = e
    other_kernel(engs, scene, distn)
    P = distn[0].shape[0]
    for r in range(rounds):
        st = synth.make_states(E, cells * 0.1, seed=1000 * k + r)
        g = torch.Generator().manual_seed(r)
        if r % 3 == 1:      # arbitrary orientations: random unit quaternions
            q = torch.randn(E, 4, generator=g); st["quat"] = q / q.norm(dim=1, keepdim=True)
        elif r % 3 == 2:    # exactly axis-aligned poses on a 0.05 m lattice (rays in facet planes, through vertices)
            st["quat"] = torch.tensor([[1.0, 0, 0, 0], [0.70710678, 0.70710678, 0, 0], [0.70710678, 0, 0.70710678, 0], [0, 1.0, 0, 0]])[torch.randint(0, 4, (E,), generator=g)]
            st["pos"][:, 0:2] = torch.round(st["pos"][:, 0:2] * 20) / 20
        d = {kk: v.cuda().contiguous() for kk, v in st.items()}
        outs = {}
        for name, e in engs.items():
            sin = e.make_in(d["pos"], d["quat"], d["joints"], d["target"], d["lin_hist"], d["ang_hist"], d["euler_pre"], d["progress"].clone())
            obs = torch.zeros(E, e.num_observations, device="cuda")
            bufs = dict(rew=torch.zeros(E, device="cuda"), reset=torch.zeros(E, dtype=torch.int64, device="cuda"),
                        rock_collision=torch.zeros(E, dtype=torch.int64, device="cuda"), ray_dist=torch.zeros(E, P, device="cuda"),
                        wheel_dist=torch.zeros(E, 24, device="cuda"), body_dist=torch.zeros(E, 2, device="cuda"))
            e.step(sin, e.make_out(obs, **bufs), increment_progress=False)
            torch.cuda.synchronize()
            outs[name] = (bufs["ray_dist"], bufs["wheel_dist"], bufs["body_dist"], bufs["rock_collision"], bufs["reset"])
        for a, b in (("culled", "envorder"), ("culled_other", "envorder"), ("culled", "binned_noearly"), ("binned", "envorder"), ("binned", "binned_noearly"),
                     ("h", "h_noearly"), ("h_culled", "h_noearly"),
                     ("staged", "binned_noearly"), ("staged_rocks", "binned_noearly"), ("staged_envorder", "binned_noearly"), ("h_staged", "h_noearly")):
            for x, y, what in zip(outs[a], outs[b], ("ray", "wheel", "body", "coll", "reset")):
                same = torch.equal(x, y) or bool(((x == y) | (x.isnan() & y.isnan())).all())
                if not same:
                    bad = (x != y).nonzero()[:5]
                    raise SystemExit(f"MISMATCH {a} vs {b} {what} K={k} round={r}: {bad.tolist()} {x[tuple(bad[0])]} {y[tuple(bad[0])]}")
        total += E * (P + 26)
        hit = float((outs["binned"][0] < 11.0).float().mean())
        ns = max(engs[n].cull_info()["rays_not_scanned"] for n in ("culled", "culled_other")) / (E * (P + 26))
        print(f"K={k} round {r}: ok, terrain hit rate {hit:.3f}, rays not scanned (on-demand kernel) {ns:.3f}", flush=True)
    for e in engs.values():
        e.close()
print(f"soak ok: {total / 1e6:.1f} M rays x 11 comparisons, all bit-identical")
