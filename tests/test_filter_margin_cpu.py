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
