"""CPU emulation of the matrix-core filter's arithmetic (split-bf16 operands, f32 accumulation, the margins the host
folds into the bounds) against the reference-order f32 discriminant: the filter must never reject a (ray, group) pair
for which a member sphere has a positive reference discriminant.  No GPU: the bounds come from rt_unit_layout (host
code of librt_hip.so), the arithmetic is re-stated here in numpy (rt_scan.h scan_list_mfma)."""
import ctypes as C

import numpy as np
import pytest

F = np.float32
K_MFMA = 4096.0
EPS = 2.0 ** -24


def bf16(x):
    """round-to-nearest-even bfloat16 of float32 values, returned as float32"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def split(x):
    h = bf16(x)
    l = bf16((np.asarray(x, dtype=F) - h).astype(F))
    return h, l


def dot_split(xs, ys, drop_lolo=False):
    """sum_i x_i y_i with every operand carried as hi + lo bf16 and f32 accumulation (sequential; the matrix core's
    internal order is unspecified, the margin has 64 eps per accumulation for that)"""
    acc = np.zeros(np.broadcast(xs[0], ys[0]).shape, dtype=F)
    for x, y in zip(xs, ys):
        xh, xl = split(x)
        yh, yl = split(y)
        for a, b in ((xh, yh), (xh, yl), (xl, yh)) + (() if drop_lolo else ((xl, yl),)):
            acc = (acc + (a * b).astype(F)).astype(F)
    return acc


def layout(built, sph):
    from cpuraytracer_amd import _capi
    L = _capi.load()
    n = C.c_uint32(0)
    _capi.check(L.rt_unit_layout(sph.ctypes.data, sph.shape[0], 0, C.byref(n), None, None))
    orig = np.zeros(n.value * 4, dtype=np.uint32)
    bounds = np.zeros((n.value, 4), dtype=np.float32)
    _capi.check(L.rt_unit_layout(sph.ctypes.data, sph.shape[0], n.value, C.byref(n), orig.ctypes.data, bounds.ctypes.data))
    return orig.reshape(-1, 4), bounds


@pytest.mark.parametrize("seed,scale,offset", [(1, 1.0, 0.0), (2, 1.0, 3000.0), (3, 0.01, 5.0), (4, 100.0, 2.0e4), (5, 1.0, 0.0)])
def test_split_bf16_filter_never_rejects_a_real_candidate(built, oracle, seed, scale, offset):
    rng = np.random.default_rng(seed)
    n = 400  # <= 128 groups: the groups are the matrix-core level (K = 4096)
    sph = np.zeros(n, dtype=oracle.SPHERE_DTYPE)
    c = rng.uniform(-8, 8, (n, 3))
    c[:, 1] = np.abs(c[:, 1]) * 0.2
    sph["cx"], sph["cy"], sph["cz"] = ((c * scale + offset).astype(F)).T
    sph["r"] = (np.exp(rng.uniform(np.log(0.05), np.log(0.8), n)) * scale).astype(F)
    orig, bounds = layout(built, sph)
    assert orig.shape[0] <= 128
    m = 1500
    # rays: origins around and far outside the cluster, directions roughly towards it (normalised in f32) or random
    o = (rng.uniform(-12, 12, (m, 3)) * rng.choice([1.0, 1.0, 6.0], (m, 1)) * scale + offset).astype(F)
    tgt = (rng.uniform(-8, 8, (m, 3)) * scale + offset)
    d = np.where(rng.random((m, 1)) < 0.7, tgt - o, rng.normal(size=(m, 3)))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F)
    a = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(F) + d[:, 2] * d[:, 2]).astype(F)
    dO = ((d[:, 0] * o[:, 0] + d[:, 1] * o[:, 1]).astype(F) + d[:, 2] * o[:, 2]).astype(F)
    oo = ((o[:, 0] * o[:, 0] + o[:, 1] * o[:, 1]).astype(F) + o[:, 2] * o[:, 2]).astype(F)
    m2a = (F(-2.0) * a).astype(F)
    g = (m2a[:, None] * o).astype(F)
    cr = ((a * oo).astype(F) * F(1.0 - 2.0 * K_MFMA * EPS)).astype(F)
    missed = 0
    for gi in range(orig.shape[0]):
        ids = orig[gi][orig[gi] != 0xFFFFFFFF]
        if len(ids) == 0:
            continue
        Cx, Cy, Cz, W = (np.full(m, v, dtype=F) for v in bounds[gi])
        # filter arithmetic: b~ = [d, d.o].[-C, 1]; t~ = [-2a o, a].[C, W] + a|o|^2 (1 - 2 K eps) (constant as hi + lo)
        bt = dot_split((d[:, 0], d[:, 1], d[:, 2], dO), (-Cx, -Cy, -Cz, np.ones(m, dtype=F)))
        crh, crl = split(cr)
        tt = dot_split((g[:, 0], g[:, 1], g[:, 2], a), (Cx, Cy, Cz, W), drop_lolo=True)
        tt = ((tt + crh).astype(F) + crl).astype(F)
        Fv = (bt * bt - tt.astype(np.float64)).astype(F)  # one fma: exact product, one rounding
        passes = Fv >= 0
        # reference-order discriminant of every member (ray-tracing.cpp:44-50), separate f32 mul/add
        for i in ids:
            ocx = (o[:, 0] - F(sph["cx"][i])).astype(F)
            ocy = (o[:, 1] - F(sph["cy"][i])).astype(F)
            ocz = (o[:, 2] - F(sph["cz"][i])).astype(F)
            b = (((ocx * d[:, 0]).astype(F) + (ocy * d[:, 1]).astype(F)).astype(F) + (ocz * d[:, 2]).astype(F)).astype(F)
            r2 = F(sph["r"][i]) * F(sph["r"][i])
            cc = ((((ocx * ocx).astype(F) + (ocy * ocy).astype(F)).astype(F) + (ocz * ocz).astype(F)).astype(F) - r2).astype(F)
            disc = ((b * b).astype(F) - (a * cc).astype(F)).astype(F)
            missed += int(np.count_nonzero((disc > 0) & ~passes))
    assert missed == 0


def _fma(a, b, c):
    """f32 fma: exact product and sum in f64 (53 bits hold a 24 x 24 bit product), one rounding"""
    return (np.asarray(a, dtype=np.float64) * np.asarray(b, dtype=np.float64) + np.asarray(c, dtype=np.float64)).astype(F)


def _reference_root(o, d, a, cx, cy, cz, r):
    """Sphere::Intersect in the reference's operation order (ray-tracing.cpp:44-71): accepted?, t"""
    ocx, ocy, ocz = (o[:, 0] - F(cx)).astype(F), (o[:, 1] - F(cy)).astype(F), (o[:, 2] - F(cz)).astype(F)
    b = (((ocx * d[:, 0]).astype(F) + (ocy * d[:, 1]).astype(F)).astype(F) + (ocz * d[:, 2]).astype(F)).astype(F)
    cc = ((((ocx * ocx).astype(F) + (ocy * ocy).astype(F)).astype(F) + (ocz * ocz).astype(F)).astype(F) - F(r) * F(r)).astype(F)
    disc = ((b * b).astype(F) - (a * cc).astype(F)).astype(F)
    with np.errstate(invalid="ignore"):
        sq = np.sqrt(disc).astype(F)
        t = ((-b - sq).astype(F) / a).astype(F)
        t2 = ((-b + sq).astype(F) / a).astype(F)
    t = np.where(t > F(0.001), t, t2)
    return (disc > 0) & (t > F(0.001)), t


@pytest.mark.parametrize("seed,scale,offset", [(11, 1.0, 0.0), (12, 1.0, 3000.0), (13, 0.01, 5.0), (14, 100.0, 2.0e4), (15, 1.0, 0.0)])
def test_far_limit_of_the_descent_never_rejects_a_nearer_or_equal_root(built, oracle, seed, scale, offset):
    """rt_scan.h bound_rejected_far (hierarchy descent): a bound is dropped when it lies wholly beyond the ray's closest hit
    so far.  Emulated in f32 against the reference-order roots of the member spheres: whenever the far test fires for tmax,
    every root the reference would accept inside that bound is > tmax -- for group bounds (K = 2048, from the host's layout)
    and for one-sphere bounds (K = 64, rebuilt here by the host's formula), with tmax taken from real hits of other spheres
    along the same ray, scaled by 1/2, 1 and 2."""
    rng = np.random.default_rng(seed)
    n = 1600  # > 128 groups: the hierarchy scan's VALU levels (K = 2048)
    sph = np.zeros(n, dtype=oracle.SPHERE_DTYPE)
    c = rng.uniform(-20, 20, (n, 3))
    c[:, 1] = np.abs(c[:, 1]) * 0.1
    sph["cx"], sph["cy"], sph["cz"] = ((c * scale + offset).astype(F)).T
    sph["r"] = (np.exp(rng.uniform(np.log(0.05), np.log(0.6), n)) * scale).astype(F)
    orig, bounds = layout(built, sph)
    assert orig.shape[0] > 128
    m = 1200
    o = (rng.uniform(-25, 25, (m, 3)) * rng.choice([1.0, 1.0, 4.0], (m, 1)) * scale + offset).astype(F)
    o[:, 1] = (np.abs(o[:, 1] - F(offset)) * F(0.3) + F(offset)).astype(F)
    tgt = (rng.uniform(-20, 20, (m, 3)) * [1, 0.05, 1] * scale + offset)
    d = np.where(rng.random((m, 1)) < 0.8, tgt - o, rng.normal(size=(m, 3)))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True) * rng.choice([1.0, 1.0, 0.3, 7.0], (m, 1))).astype(F)  # not all unit length
    a = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(F) + d[:, 2] * d[:, 2]).astype(F)
    dO = ((d[:, 0] * o[:, 0] + d[:, 1] * o[:, 1]).astype(F) + d[:, 2] * o[:, 2]).astype(F)
    oo = ((o[:, 0] * o[:, 0] + o[:, 1] * o[:, 1]).astype(F) + o[:, 2] * o[:, 2]).astype(F)
    g = ((F(-2.0) * a).astype(F)[:, None] * o).astype(F)
    Cn = np.linalg.norm(bounds[:, :3].astype(np.float64), axis=1)
    Rf = np.sqrt(np.maximum(0.0, Cn * Cn - bounds[:, 3].astype(np.float64)))
    real = bounds[:, 3] < 1e29
    bound_norm = F((Cn[real] + Rf[real]).max() * 1.001)
    bt = (F(1e-4) * np.sqrt(a).astype(F) * (np.sqrt(oo).astype(F) + bound_norm)).astype(F)
    # closest hits along every ray: the far limits a descent would really see (every sphere, reference order)
    hits = np.full((m, 3), np.inf, dtype=F)
    roots = {}
    for i in range(n):
        ok, t = _reference_root(o, d, a, sph["cx"][i], sph["cy"][i], sph["cz"][i], sph["r"][i])
        roots[i] = (ok, t)
        tt = np.where(ok, t, np.inf).astype(F)
        hits[:, 0] = np.minimum(hits[:, 0], tt)                       # the closest hit
        hits[:, 1] = np.where(rng.random(m) < 0.03, np.minimum(hits[:, 1], tt), hits[:, 1])  # some hit or other
    hits[:, 2] = hits[:, 0]
    tmaxes = [hits[:, 0], hits[:, 1], (hits[:, 1] * F(0.5)).astype(F), (hits[:, 2] * F(2.0)).astype(F)]

    def far_rejects(B, K, tmax):
        cr = ((a * oo).astype(F) * F(1.0 - 2.0 * K * EPS)).astype(F)
        b = _fma(-d[:, 2], B[2], _fma(-d[:, 1], B[1], _fma(-d[:, 0], B[0], dO)))
        t = _fma(g[:, 2], B[2], _fma(g[:, 1], B[1], _fma(g[:, 0], B[0], _fma(a, B[3], cr))))
        u = ((a * tmax).astype(F) * F(1.0 + 2.0 ** -10)).astype(F)
        bu = (b + u).astype(F)
        with np.errstate(invalid="ignore", over="ignore"):
            fo = _fma(u, (b + bu).astype(F), t)
        return np.isfinite(tmax) & (bu < -bt) & (fo > 0)

    wrong = fired = 0
    for gi in range(orig.shape[0]):
        ids = orig[gi][orig[gi] != 0xFFFFFFFF]
        if len(ids) == 0 or not real[gi]:
            continue
        for tmax in tmaxes:
            rej = far_rejects(bounds[gi], 2048.0, tmax)
            fired += int(rej.sum())
            for i in ids:
                ok, t = roots[i]
                wrong += int(np.count_nonzero(rej & ok & ~(t > tmax)))
        for i in ids:  # the member's one-sphere bound (host: BoundOf of a single sphere, K = 64)
            cx, cy, cz, r = (float(sph[k][i]) for k in ("cx", "cy", "cz", "r"))
            C2 = cx * cx + cy * cy + cz * cz
            Rf2 = r * r * (1.0 + 1e-5) + 64.0 * EPS * (2.0 * (np.sqrt(C2) + r) ** 2 + r * r)
            w = np.nextafter(np.nextafter(F(C2 - Rf2), F(-np.inf)), F(-np.inf))
            ok, t = roots[i]
            for tmax in tmaxes:
                rej = far_rejects((F(cx), F(cy), F(cz), w), 64.0, tmax)
                fired += int(rej.sum())
                wrong += int(np.count_nonzero(rej & ok & ~(t > tmax)))
    assert fired > 10000  # the test does exercise the far limit
    assert wrong == 0


@pytest.mark.parametrize("seed,scale,offset", [(21, 1.0, 0.0), (22, 1.0, 3000.0), (23, 0.01, 5.0), (24, 100.0, 2.0e4), (25, 1.0, 0.0)])
def test_box_clip_of_the_descent_keeps_every_accepted_root(built, oracle, seed, scale, offset):
    """rt_scan.h, hierarchy descent: rays are clipped to the box around the spheres in the hierarchy, padded per ray by the reach
    of an accepted root outside its sphere (the reference's discriminant is off by up to 16 eps a G).  Emulated in f32, with
    rays that graze spheres exactly where they touch the box -- from 10, 300 and 3,000 scene units away, where the rounding is
    largest: (I) no accepted reference root lies outside the widened interval, and a ray declared to miss the box has none;
    (II) whenever the near/far test of a bound fires, no member sphere has an accepted root."""
    rng = np.random.default_rng(seed)
    n = 900  # > 128 groups; a flat layer like the cover scene's small spheres
    sph = np.zeros(n, dtype=oracle.SPHERE_DTYPE)
    c = rng.uniform(-15, 15, (n, 3))
    c[:, 1] = 0.2 + 0.05 * rng.random(n)
    sph["cx"], sph["cy"], sph["cz"] = ((c * scale + offset).astype(F)).T
    sph["r"] = (np.exp(rng.uniform(np.log(0.1), np.log(0.3), n)) * scale).astype(F)
    orig, bounds = layout(built, sph)
    assert orig.shape[0] > 128
    cs = np.stack([sph["cx"], sph["cy"], sph["cz"]], 1).astype(np.float64)
    rs = sph["r"].astype(np.float64)
    # the host's box (rt_capi.hip BuildLayout) and the padding constants
    lo = np.array([np.nextafter(F((cs[:, k] - rs * (1 + 1e-5)).min()), F(-np.inf)) for k in range(3)], dtype=F)
    hi = np.array([np.nextafter(F((cs[:, k] + rs * (1 + 1e-5)).max()), F(np.inf)) for k in range(3)], dtype=F)
    absmax = F(max(np.abs(lo.astype(np.float64)).max(), np.abs(hi.astype(np.float64)).max()) * 1.001)
    A = (np.linalg.norm(cs, axis=1) + rs).max()
    A2 = F(3.0 * A * A * 1.001)
    inv2r = F(1.001 / (2.0 * rs.min()))
    # rays: random ones, and grazers of the sphere tops / sides from far away
    m0 = 600
    o = (rng.uniform(-20, 20, (m0, 3)) * [1, 0.2, 1] * scale + offset).astype(np.float64)
    tgt = (rng.uniform(-15, 15, (m0, 3)) * [1, 0.02, 1] * scale + offset)
    d = tgt - o
    graz_o, graz_d = [], []
    for i in rng.integers(0, n, 300):
        for L in (10.0, 300.0, 3000.0):
            ax = int(rng.integers(0, 3)); sgn = rng.choice([-1.0, 1.0])
            pnt = cs[i].copy(); pnt[ax] += sgn * rs[i] * (1.0 + rng.choice([-1e-4, -1e-6, 0.0, 1e-6, 1e-4]))
            dirv = rng.normal(size=3); dirv[ax] = rng.normal() * 1e-4; dirv /= np.linalg.norm(dirv)
            graz_o.append(pnt - L * scale * dirv); graz_d.append(dirv * rng.choice([1.0, 0.3, 5.0]))
    o = np.concatenate([o, np.array(graz_o)]).astype(F)
    d = np.concatenate([d / np.linalg.norm(d, axis=1, keepdims=True), np.array(graz_d)]).astype(F)
    m = o.shape[0]
    a = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(F) + d[:, 2] * d[:, 2]).astype(F)
    dO = ((d[:, 0] * o[:, 0] + d[:, 1] * o[:, 1]).astype(F) + d[:, 2] * o[:, 2]).astype(F)
    oo = ((o[:, 0] * o[:, 0] + o[:, 1] * o[:, 1]).astype(F) + o[:, 2] * o[:, 2]).astype(F)
    g = ((F(-2.0) * a).astype(F)[:, None] * o).astype(F)
    # device: the clip
    X = (F(32.0) * F(EPS) * _fma(F(2.0), oo, A2)).astype(F)
    reach = np.minimum(np.sqrt(X).astype(F), (X * inv2r).astype(F))
    tn = np.zeros(m, dtype=F); tf = np.full(m, np.inf, dtype=F)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for ax in range(3):
            pad = (reach + (F(1e-6) * (np.abs(o[:, ax]) + absmax).astype(F)).astype(F)).astype(F)
            inv = (F(1.0) / d[:, ax]).astype(F)
            t0 = ((((lo[ax] - pad).astype(F) - o[:, ax]).astype(F)) * inv).astype(F)
            t1 = ((((hi[ax] + pad).astype(F) - o[:, ax]).astype(F)) * inv).astype(F)
            tn = np.fmax(tn, np.fmin(t0, t1)); tf = np.fmin(tf, np.fmax(t0, t1))
    tn = (tn * F(1.0 - 2.0 ** -10)).astype(F); tf = (tf * F(1.0 + 2.0 ** -10)).astype(F)
    empty = (tn > tf) | ~(tf > 0)
    un = np.where(tn > 0, (a * tn).astype(F), F(-np.inf)).astype(F)
    uf = (a * tf).astype(F)
    roots = {}
    outside = 0
    hits = 0
    for i in range(n):
        ok, t = _reference_root(o, d, a, sph["cx"][i], sph["cy"][i], sph["cz"][i], sph["r"][i])
        roots[i] = (ok, t)
        hits += int(ok.sum())
        outside += int(np.count_nonzero(ok & (empty | (t < tn) | (t > tf))))
    assert hits > 3000 and outside == 0  # (I)
    Cn = np.linalg.norm(bounds[:, :3].astype(np.float64), axis=1)
    Rf = np.sqrt(np.maximum(0.0, Cn * Cn - bounds[:, 3].astype(np.float64)))
    real = bounds[:, 3] < 1e29
    bt = (F(1e-4) * np.sqrt(a).astype(F) * (np.sqrt(oo).astype(F) + F((Cn[real] + Rf[real]).max() * 1.001))).astype(F)
    cr = ((a * oo).astype(F) * F(1.0 - 2.0 * 2048.0 * EPS)).astype(F)
    wrong = fired = 0
    with np.errstate(invalid="ignore", over="ignore"):
        for gi in range(orig.shape[0]):
            ids = orig[gi][orig[gi] != 0xFFFFFFFF]
            if len(ids) == 0 or not real[gi]:
                continue
            B = bounds[gi]
            b = _fma(-d[:, 2], B[2], _fma(-d[:, 1], B[1], _fma(-d[:, 0], B[0], dO)))
            t = _fma(g[:, 2], B[2], _fma(g[:, 1], B[1], _fma(g[:, 0], B[0], _fma(a, B[3], cr))))
            bf = (b + uf).astype(F); ff = _fma(uf, (b + bf).astype(F), t)
            bn = (b + un).astype(F); fn = _fma(un, (b + bn).astype(F), t)
            rej = ~empty & (((bf < -bt) & (ff > 0)) | ((bn > bt) & (fn > 0)))
            fired += int(rej.sum())
            for i in ids:
                wrong += int(np.count_nonzero(rej & roots[i][0]))
    assert wrong == 0 and (fired > 10000 or offset != 0.0)  # (II); far from the origin the padding swallows the layer: nothing to fire
