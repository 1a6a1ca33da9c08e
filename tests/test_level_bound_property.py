"""Property test (CPU, numpy) of the staged ray cast's LEVEL bound — `lane_build_kernel` / `lane_scan_kernel` in
csrc/rover_cull.hip, DESIGN.md §4.5 and §5.3.

A suffix of a cell's record row is skipped as a group when the ray clears its bound {G, z0, z1, rho_out}; the claim is that every
triangle of the suffix then passes test (A), `c_a |h|^2 - (h.d)^2 > r2`, on its own — with room for the f32 rounding of that test
(< 3e-6 |h|^2 + 3e-7 r2).  The GPU suites check the consequence (bit-identical distances against the every-triangle
kernel); this test checks the inequality itself, at its tightest: for random sets of sphere records and random steep rays the set is
moved towards the ray until the bound (evaluated as the kernel does, in float32 on fp16-rounded level records, with the kernel's
margins) JUST clears it, and there every record is put through test (A) in float64.

The restatement below follows the kernel's operation order; `v_fma_mix_f32` / `v_fma_f32` are single roundings, numpy rounds each
operation: the difference is an ulp per operation, three orders of magnitude below the 1.0004 / 0.9999 margins.
"""
import math

import numpy as np

f32 = np.float32
C_A = f32(0.99925)                      # CullK<0>::c_a
C_RHO = 1.06                           # CullK<0>::c_rho
ALPHA = 0.005                          # CullK<0>::alpha: test (A) must give W >= rho + ALPHA (|h| + 2 rho)
CONE_TAU = 1.25e-2                     # ROVER_CONE_TAU


def _far_consts(c_a=0.99925, dd=1.00001):
    """cull_far_consts (rover_cull.hip)."""
    ca = c_a - 1.0e-5
    k1 = f32(1.00001 / (0.9 * math.sqrt(ca)))
    k2 = f32(1.00001 * math.sqrt(dd + 1.0e-5 - ca) / (0.9 * math.sqrt(ca)))
    return k1, k2


def _h_down(v):
    """fp16 <= v (lane_build_kernel's `down`)."""
    h = np.float16(v)
    if np.isfinite(h) and float(h) > float(v):
        h = np.nextafter(h, np.float16(-np.inf))
    return h


def _h_up(v):
    """fp16 >= v (lane_build_kernel's `up` / half_bits_up)."""
    h = np.float16(v)
    if np.isfinite(h) and float(h) < float(v):
        h = np.nextafter(h, np.float16(np.inf))
    return h


def _level_record(hx, hy, hz, r2h, k1):
    """The suffix bound lane_build_kernel stores for a set of sphere records (hx, hy, hz: fp16 values; r2h: the records' fp16 r2)."""
    hx = hx.astype(f32); hy = hy.astype(f32); hz = hz.astype(f32); r2h = r2h.astype(f32)
    dxy = np.sqrt(hx * hx + hy * hy, dtype=f32)
    g = (dxy - k1 * np.sqrt(r2h, dtype=f32) * f32(1.00001)) * f32(0.99999) - f32(1.0e-6)
    pro = dxy * f32(1.00001)
    G = f32(g.min()) * f32(0.9999) - f32(1.0e-5)
    return _h_down(G), _h_down(hz.min()), _h_up(hz.max()), _h_up(pro.max())


def _clears(level, s, d, k2):
    """lane_scan_kernel's level test for a ray with cell-relative origin s and direction d (float32 arithmetic)."""
    Gq, z0, z1, ro = (f32(x) for x in level)
    sx, sy, sz = (f32(x) for x in s)
    dx, dy, dz = (f32(x) for x in d)
    o = np.sqrt(sx * sx + sy * sy, dtype=f32)
    dxy2 = dx * dx + dy * dy
    adz = abs(dz)
    sq = np.sqrt(dxy2, dtype=f32)
    steep = adz * adz >= f32(0.81) * (dxy2 + adz * adz) * f32(1.0001)
    c3 = k2 * adz * sq * f32(1.0004)
    c4 = (sq + k2 * adz * adz) * f32(1.0004)
    base = -((adz * f32(1.0004) + c3) * o)
    dzm = max(abs(sz - z0), abs(sz - z1))
    lhs = Gq * adz + (ro * (-c3) + base)
    return bool(steep) and bool(lhs > dzm * c4)


def _test_a_margin(hx, hy, hz, r2h, s, d):
    """c_a |h|^2 - (h.d)^2 - r2 minus the rounding allowance of the f32 test, per record, in float64 (h = s - m)."""
    h = np.stack([s[0] - hx.astype(np.float64), s[1] - hy.astype(np.float64), s[2] - hz.astype(np.float64)], axis=1)
    q = (h * h).sum(axis=1)
    t = h @ np.asarray(d, dtype=np.float64)
    r2 = r2h.astype(np.float64)
    return float(C_A) * q - t * t - r2 - (3.0e-6 * q + 3.0e-7 * r2)


def _random_case(rng):
    m = int(rng.integers(1, 40))
    ang = rng.uniform(0.0, 2.0 * math.pi, m)
    spread = rng.uniform(0.0, 1.5, m) * rng.uniform(0.0, 1.0)          # how far behind the nearest record the others lie
    hz = rng.uniform(-0.6, 0.6, m) * rng.uniform(0.05, 1.0)
    r = rng.uniform(0.01, 0.25, m)
    # a steep ray (cos beta >= ~0.9) from inside or near the cell, at any height above or below the records
    tilt = rng.uniform(0.0, 0.44)
    az = rng.uniform(0.0, 2.0 * math.pi)
    d64 = np.array([math.sin(tilt) * math.cos(az), math.sin(tilt) * math.sin(az), -math.cos(tilt)]) * rng.choice([1.0, -1.0])
    d = d64.astype(f32)
    s = np.array([rng.uniform(-0.08, 0.08), rng.uniform(-0.08, 0.08), rng.uniform(-0.6, 3.0) * rng.choice([1.0, 0.2])]).astype(f32)
    return ang, spread, hz, r, s, d


def _records(D, ang, spread, hz, r):
    """fp16 sphere records of the set moved to xy distance D (+ spread) from the cell centre; None when one leaves the +-4 m the
    builder admits (such a triangle is stored as always-a-candidate: no bound depends on it)."""
    dist = D + spread
    if dist.max() > 3.9:
        return None
    hx = np.float16(dist * np.cos(ang)); hy = np.float16(dist * np.sin(ang)); hzh = np.float16(hz)
    r2h = np.array([_h_up(f32(C_RHO) * f32(x) * f32(x)) for x in r], dtype=np.float16)
    return hx, hy, hzh, r2h


def test_a_cleared_suffix_passes_test_a_record_by_record():
    rng = np.random.default_rng(20251004)
    k1, k2 = _far_consts()
    tight = 0
    worst = math.inf
    for _ in range(1500):
        ang, spread, hz, r, s, d = _random_case(rng)
        lo, hi = 0.0, 3.9 - float(spread.max())
        rec = _records(hi, ang, spread, hz, r)
        if rec is None or not _clears(_level_record(*rec, k1), s, d, k2):
            continue                                   # never cleared inside the builder's range: nothing to check
        for _ in range(40):                            # the smallest distance at which the set is still cleared
            mid = 0.5 * (lo + hi)
            rec = _records(mid, ang, spread, hz, r)
            if _clears(_level_record(*rec, k1), s, d, k2):
                hi = mid
            else:
                lo = mid
        rec = _records(hi, ang, spread, hz, r)
        assert _clears(_level_record(*rec, k1), s, d, k2)
        margin = _test_a_margin(*rec, s.astype(np.float64), d.astype(np.float64))
        assert (margin > 0.0).all(), f"cleared as a group at D = {hi}, but a record fails test (A): {margin.min()}"
        worst = min(worst, float(margin.min()))
        tight += 1
    assert tight > 500                                 # the search did reach the bound's edge in most cases
    assert worst < 0.5                                 # ... and the edge is not far from test (A)'s own (the bound is not vacuous)


def test_a_ray_that_is_not_steep_clears_nothing():
    """cos beta < 0.9: the bound does not apply (the horizontal body rays), whatever the distances."""
    k1, k2 = _far_consts()
    rec = _records(3.0, np.array([0.3]), np.array([0.0]), np.array([0.0]), np.array([0.01]))
    level = _level_record(*rec, k1)
    for tilt in (0.47, 0.8, 1.2, math.pi / 2):
        d = np.array([math.sin(tilt), 0.0, -math.cos(tilt)], dtype=f32)
        assert not _clears(level, np.zeros(3, dtype=f32), d, k2)
    assert _clears(level, np.zeros(3, dtype=f32), np.array([0.0, 0.0, -1.0], dtype=f32), k2)


# ---------------------------------------------------------------------------------------------------
# Test (A) on the row's cell-relative fp16 records: u >= +0 as lane_scan_kernel computes it must give what
# the rejection proof uses of test (A): W >= rho + ALPHA (|h| + 2 rho) for the TRUE sphere (centre m, padded radius rho) and the true ray.
# ---------------------------------------------------------------------------------------------------
def _fma32(a, b, c):
    return f32(np.float64(a) * np.float64(b) + np.float64(c))          # one rounding (the products of two f32 are exact in f64)


def _record_of(m, C, rho_true):
    """lane_build_kernel's (A) record of a triangle whose ctab entry has centre m and r2 = C_RHO (rho + 1e-4)^2: m' (fp16), r2' (fp16)."""
    r2c = f32(C_RHO * (rho_true + 1.0e-4) ** 2 * (1.0 + 1.0e-6))
    rel = (np.asarray(m, dtype=np.float64) - np.asarray(C, dtype=np.float64)).astype(f32)
    mh = rel.astype(np.float16)
    enc = float(np.linalg.norm(np.asarray(C, dtype=np.float64) + mh.astype(np.float64) - np.asarray(m, dtype=np.float64)))
    rho = math.sqrt(float(r2c) / C_RHO)
    need = C_RHO * (rho + enc + 5.0e-5) ** 2 * 1.000001
    r2h = _h_up(f32(need * 1.0000001))
    if float(r2h) < 2.0 ** -14:
        r2h = np.float16(2.0 ** -14)
    return mh, r2h


def _u_of(mh, r2h, C, s, d):
    """u of test (A) in lane_scan_kernel's operation order (float32; v_fma_mix_f32 and v_fma_f32 round once)."""
    sp = (np.asarray(s, dtype=f32) - np.asarray(C, dtype=f32)).astype(f32)
    hx, hy, hz = (f32(sp[i] - f32(mh[i])) for i in range(3))
    t = _fma32(hz, d[2], _fma32(hy, d[1], f32(hx * d[0])))
    qq = _fma32(hz, hz, _fma32(hy, hy, f32(hx * hx)))
    u = _fma32(qq, C_A, -f32(r2h))
    return _fma32(-t, t, u)


def test_test_a_on_cell_relative_fp16_records_implies_the_proofs_premise():
    rng = np.random.default_rng(77)
    checked = 0
    tightest = math.inf
    for _ in range(3000):
        C = np.array([rng.uniform(-30.0, 30.0), rng.uniform(-30.0, 30.0), rng.uniform(-2.0, 2.0)]).astype(f32).astype(np.float64)
        m = C + np.array([rng.uniform(-3.5, 3.5), rng.uniform(-3.5, 3.5), rng.uniform(-1.0, 1.0)]) * rng.choice([1.0, 0.1])
        m = m.astype(f32).astype(np.float64)                   # ctab holds f32 (x, y) and fp16-decoded z: values the tables can hold
        rho = float(rng.uniform(0.005, 0.3))
        mh, r2h = _record_of(m, C, rho)
        dv = rng.normal(size=3); dv /= np.linalg.norm(dv)
        d = dv.astype(f32)
        dhat = d.astype(np.float64) / np.linalg.norm(d.astype(np.float64))
        # rays through points at lateral offset w from m: find the smallest w at which the kernel's u is >= +0
        perp = np.cross(dhat, rng.normal(size=3)); perp /= np.linalg.norm(perp)
        along = float(rng.uniform(-3.0, 3.0))

        def origin(w):
            return (m + perp * w + dhat * along).astype(f32)
        lo, hi = 0.0, 6.0
        if not _u_of(mh, r2h, C.astype(f32), origin(hi), d) >= 0.0:
            continue
        for _ in range(50):
            mid = 0.5 * (lo + hi)
            if _u_of(mh, r2h, C.astype(f32), origin(mid), d) >= 0.0:
                hi = mid
            else:
                lo = mid
        s = origin(hi).astype(np.float64)
        h = s - m
        W = float(np.linalg.norm(h - (h @ dhat) * dhat))
        need = rho + ALPHA * (float(np.linalg.norm(h)) + 2.0 * rho)
        assert W >= need, f"u >= 0 at W = {W}, the proof needs {need} (rho {rho}, |h| {np.linalg.norm(h)})"
        tightest = min(tightest, W / need)
        checked += 1
    assert checked > 2500
    assert tightest < 1.2                                   # the test sits close to what it has to imply


# ---------------------------------------------------------------------------------------------------
# Test (B) for a whole set (last sentence): the ray record carries qm = CONE_TAU |d_z| + |d_xy| (+ 2e-5, rounded UP to 16
# bits: ray_cone_bound, rover_kernels.hip), a cell or a suffix q = min |N_z| / |N| over its triangles (rounded DOWN to 16 bits:
# lane_build_kernel / idx4_build_kernel).  q16 >= rq must give |N . d| > (CONE_TAU - 1e-4) |N| for every triangle of the set (CONE_TAU for an exactly unit
# d) — more than the 3e-3 that (B) with the stored normal's 1e-3 error needs.
# ---------------------------------------------------------------------------------------------------
def _ray_q16(d):
    dx, dy, dz = (f32(x) for x in d)
    qm = f32(CONE_TAU) * abs(dz) + np.sqrt(dx * dx + dy * dy, dtype=f32) + f32(2.0e-5)
    return 0xffff if not qm < f32(0.9999) else int(math.ceil(float(qm * f32(65535.0))))


def _set_q16(qn):
    q16 = int(math.floor(float(f32(qn) * f32(65535.0)))) if qn > 0.0 else 0
    return min(q16, 0xfffe)


def test_a_set_cone_that_covers_the_ray_gives_test_b_for_every_triangle():
    rng = np.random.default_rng(5)
    covered = 0
    tightest = math.inf
    for _ in range(4000):
        tilt = rng.uniform(0.0, 1.3) * rng.choice([1.0, 0.3])
        az = rng.uniform(0.0, 2.0 * math.pi)
        # (a record's direction is -normalize() in f32, or checked by rover_cast_rays: |d|^2 within 1e-5 of 1.  The bound leans on that: at a
        #  tilt of 1.3 rad a direction 0.1 % short would leave 1e-5 of the threshold)
        d = (np.array([math.sin(tilt) * math.cos(az), math.sin(tilt) * math.sin(az), -math.cos(tilt)]) * rng.uniform(1.0 - 4.0e-6, 1.0 + 4.0e-6)).astype(f32)
        rq = _ray_q16(d)
        # the set's cone JUST covers the ray; its triangles' normals lie anywhere on or inside that cone, worst azimuth included
        q16 = rq
        if q16 > 0xfffe:
            continue
        qn_min = q16 / 65535.0                               # every q that rounds down to q16 is >= this
        n_t = 50
        nz = np.concatenate([[qn_min], rng.uniform(qn_min, 1.0, n_t - 1)])
        na = np.concatenate([[az, az + math.pi], rng.uniform(0.0, 2.0 * math.pi, n_t - 2)])      # towards / against the ray's tilt: the worst cases
        nxy = np.sqrt(np.maximum(0.0, 1.0 - nz * nz))
        N = np.stack([nxy * np.cos(na), nxy * np.sin(na), nz * rng.choice([1.0, -1.0], n_t)], axis=1)
        assert _set_q16(qn_min) <= q16
        dd = d.astype(np.float64)
        c = np.abs(N @ dd)                                   # |N| = 1
        assert (c > CONE_TAU - 1.0e-4).all(), f"cone {q16} >= ray bound {rq}, but |N.d| / |N| = {c.min()}"
        tightest = min(tightest, float(c.min()))
        covered += 1
    assert covered > 2000
    assert tightest < 0.02                                   # the worst-azimuth normal on the cone's rim sits close to the threshold


# ---------------------------------------------------------------------------------------------------
# The constants of test (A) themselves (rover_cull.hip: CullK<0>, cull_proof_h): c_a |h|^2 - (h.d)^2 > c_rho rho^2 has to
# give W - rho >= eta (|h| + 2 rho) — f32 proof: c_a = 0.9984, c_rho = 1.12, eta = ALPHA = 0.005(rounds 2-5: 0.995, 1.19, 0.02), |d|^2 <= 1.00001; as-shipped fp16 proof: c_a, c_rho
# derived from eta by cull_proof_h, |d|^2 within 4e-3 of 1 (the direction is normalised in fp16).
# ---------------------------------------------------------------------------------------------------
def _proof_h(eta, split=8.0):
    """cull_proof_h (rover_cull.hip): `split` = how the cross term 2 alpha beta rho |h| is shared between the rho^2 and the |h|^2 part."""
    a = eta + 1.0e-3
    b = 1.0 + 2.0 * a
    c_rho = (b * b + split * a * b) * 1.004 + 0.005
    c_a = float(f32(0.996 * (1.0 - (a * a + a * b / split)) - 0.0005))
    return c_a, c_rho


def _edge_of_test_a(rng, c_a, c_rho, eta, dd_lo, dd_hi, n):
    """Rays at the edge of `c_a |h|^2 - (h.d)^2 > c_rho rho^2`: the smallest (W - rho) / (|h| + 2 rho) over them must stay >= eta."""
    worst = math.inf
    for _ in range(n):
        rho = float(rng.uniform(1.0e-3, 0.5))
        dlen = math.sqrt(float(rng.uniform(dd_lo, dd_hi)))
        dv = rng.normal(size=3); dv *= dlen / np.linalg.norm(dv)
        dhat = dv / np.linalg.norm(dv)
        perp = np.cross(dhat, rng.normal(size=3)); perp /= np.linalg.norm(perp)
        along = float(rng.uniform(-1.0, 1.0)) * float(rng.choice([0.1, 1.0, 10.0]))
        lo, hi = 0.0, 50.0 + 50.0 * abs(along)

        def passes(w):
            h = perp * w + dhat * along
            return c_a * float(h @ h) - float(h @ dv) ** 2 > c_rho * rho * rho
        if not passes(hi):
            continue
        for _ in range(60):
            mid = 0.5 * (lo + hi)
            if passes(mid):
                hi = mid
            else:
                lo = mid
        h = perp * hi + dhat * along
        W = hi                                                   # distance from the centre to the ray's line
        worst = min(worst, (W - rho) / (float(np.linalg.norm(h)) + 2.0 * rho))
    return worst


def test_test_a_constants_give_the_margin_the_proofs_use():
    rng = np.random.default_rng(11)
    w32 = _edge_of_test_a(rng, float(C_A), C_RHO, ALPHA, 0.99999, 1.00001, 3000)
    assert w32 >= ALPHA, w32
    assert w32 < 0.03                                            # ... and not much more: the constants are not wasteful
    for eta, split in ((0.04, 2.0), (0.06, 2.0), (0.10, 2.0), (0.06, 6.0), (0.08, 8.0), (0.08, 16.0)):   # (0.08, 8): the library's choice (cull_eta_h, cull_split_h)
        c_a, c_rho = _proof_h(eta, split)
        wh = _edge_of_test_a(rng, c_a, c_rho, eta, 0.996, 1.004, 3000)
        assert wh >= eta, (eta, split, wh)
        assert wh < eta + 0.03


# ---------------------------------------------------------------------------------------------------------------------
# Round 6: test (B) on 4-byte records (B4, rover_cull.hip: LN_B4_*)
# ---------------------------------------------------------------------------------------------------------------------
LN_B4_TAU, LN_B4_ERR = 1.55e-2, 3.0e-3
LN_B4_C = f32((512.0 * LN_B4_TAU) * (512.0 * LN_B4_TAU) * 1.0001)


def _b4_encode(n_fp16):
    """lane_build_kernel: the ctab record's fp16 normal (any length) -> the three signed 10-bit integers, or None ("always a candidate")."""
    n = n_fp16.astype(np.float64)
    nn = math.sqrt(float(n @ n))
    if not (0.0 < nn < 1.0e30):
        return None
    q = np.rint(n / nn * 511.0)
    ql = math.sqrt(float(q @ q))
    err = float(np.linalg.norm(q / ql - n / nn))
    if ql <= 0.0 or ql > 512.0 or err > LN_B4_ERR - 5.0e-4 - 1.0e-6 or np.abs(q).max() > 511:
        return None
    return q.astype(np.int32)


def _b4_culls(q, d):
    """lane_scan_kernel's test on the decoded integers and the ray's direction, in float32: culled iff fl(t * t - C) >= +0."""
    t = f32(q[0]) * f32(d[0])
    t = f32(f32(q[1]) * f32(d[1]) + t)      # (the kernel's fma rounds once where this rounds twice: an ulp of |t| <= 512, covered below)
    t = f32(f32(q[2]) * f32(d[2]) + t)
    u = f32(t * t - LN_B4_C)
    return not np.signbit(u)


def test_b4_records_decode_within_their_allowance_and_cull_only_off_plane_rays():
    """Test (B) on 4-byte records.  For random triangle normals — stored the way ctab_build_kernel stores them, as fp16 components of a vector of
    length r / tau — the code lane_build_kernel keeps decodes to a direction within LN_B4_ERR of the TRUE unit normal, |n4| <= 512; and whenever
    the kernel's float32 test culls a ray, the true normal has |cos(N, d)| > LN_B4_TAU - LN_B4_ERR, with the slack the proof's static_assert uses
    (0.999 (LN_B4_TAU - LN_B4_ERR) sigma > 1e-4).  Rays are drawn near each triangle's plane, where the test decides, and at random."""
    rng = np.random.default_rng(5)
    n_codes = n_culled = n_kept_near = 0
    worst_err, worst_cos = 0.0, 1.0
    for _ in range(4000):
        N = rng.normal(size=3)
        if rng.random() < 0.3:
            N[rng.integers(3)] *= 1.0e-3                       # normals near a coordinate plane / axis: where the quantisation is coarsest
        N /= np.linalg.norm(N)
        length = 10.0 ** rng.uniform(0.0, 3.0)                # |stored normal| = r / tau: 1 ... 1000
        n16 = (N * length).astype(np.float16)
        q = _b4_encode(n16)
        if q is None:
            continue
        n_codes += 1
        assert float(np.sqrt((q.astype(np.float64) ** 2).sum())) <= 512.0
        err = float(np.linalg.norm(q / np.linalg.norm(q) - N))
        worst_err = max(worst_err, err)
        assert err <= LN_B4_ERR, err
        # directions: in the triangle's plane tilted out of it by a small angle around the threshold, and anywhere
        a = np.cross(N, rng.normal(size=3)); a /= np.linalg.norm(a)
        for k in range(24):
            if k < 16:
                ang = rng.uniform(0.0, 2.5) * LN_B4_TAU * (1 if rng.random() < 0.5 else -1)
                d = math.cos(ang) * a + math.sin(ang) * N
            else:
                d = rng.normal(size=3); d /= np.linalg.norm(d)
            d = (d * rng.uniform(0.999995, 1.000005)).astype(np.float32)      # |d|^2 within 1e-5 of 1, as the proofs assume
            cos_true = abs(float(N @ d.astype(np.float64))) / float(np.linalg.norm(d.astype(np.float64)))
            if _b4_culls(q, d):
                n_culled += 1
                worst_cos = min(worst_cos, cos_true)
                assert cos_true > LN_B4_TAU - LN_B4_ERR, (cos_true, q, d)
            elif k < 16:
                n_kept_near += 1
    assert n_codes > 3500 and n_culled > 20000 and n_kept_near > 5000       # both outcomes were exercised near the threshold
    assert 0.999 * (LN_B4_TAU - LN_B4_ERR) * 0.05 > 1.45 * 2.0 * 1.0e-6 / ALPHA      # the static_assert of rover_cull.hip
    print(f"B4: {n_codes} codes, worst decoded-normal error {worst_err:.2e} (allowance {LN_B4_ERR}), smallest |cos| of a culled ray {worst_cos:.2e} "
          f"(needs > {LN_B4_TAU - LN_B4_ERR})")
    # the "always a candidate" code: n4 = 0 gives u = -C < 0 for every direction
    assert not _b4_culls(np.zeros(3, np.int32), np.array([0.3, -0.5, 0.8], np.float32))
