"""CPU emulation of the cell-grid scan's walk (rt_scan.h: grid_slab_rows + the feed of scan_list_grid), binary32 operation by
operation: the set of cells it visits for a segment must contain EVERY cell within D cells (Chebyshev) of the segment — that
is what makes the grid scan's candidate set conservative: the hit point of an accepted root lies on the clipped segment and
within r + reach of its sphere's centre, so the sphere's home cell is within D = (r_max + reach) / h of the segment.  No GPU."""
import ctypes as C

import numpy as np
import pytest

F = np.float32


def shipped_slab_rows(queries, ius, nv):
    """grid_segment_slope + grid_slab_rows AS SHIPPED: librt_hip.so's rt_unit_grid_rows runs the very functions of csrc/rt_scan.h
    (compiled for the host; no GPU needed), so an edit to them -- kGridSlack, the padding, the clamps -- meets these tests."""
    from cpuraytracer_amd import _capi
    L = _capi.load()
    L.rt_unit_grid_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p]
    q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, 5)
    iu = np.ascontiguousarray(ius, dtype=np.int32)
    rows = np.zeros((q.shape[0], 2), dtype=np.int32)
    se = np.zeros(q.shape[0], dtype=np.float32)
    assert L.rt_unit_grid_rows(q.ctypes.data, iu.ctypes.data, q.shape[0], nv, rows.ctypes.data, se.ctypes.data) == 0
    return rows, se
SLACK = F(1e-3)


def fma(a, b, c):
    return F(np.float64(a) * np.float64(b) + np.float64(c))


def slab_rows(su, sv, eu, ev, D, iu, nv):
    """grid_slab_rows, same operations in the same order; returns (r0, r1) with r0 > r1 for 'no rows'"""
    du, dv = F(eu - su), F(ev - sv)
    ulo, uhi = min(su, eu), max(su, eu)
    a = max(ulo, F(F(F(iu) - D) - SLACK))
    b = min(uhi, F(F(F(iu + 1) + D) + SLACK))
    steep = abs(du) < F(F(1e-4) * max(abs(dv), F(1.0)))
    if steep:
        vlo, vhi = min(sv, ev), max(sv, ev)
    else:
        slope = F(dv / du)
        va, vb = fma(F(a - su), slope, sv), fma(F(b - su), slope, sv)
        pad = F(F(1e-4) * abs(slope))
        vlo, vhi = F(min(va, vb) - pad), F(max(va, vb) + pad)
    if a > b:
        return 1, 0
    f0 = np.floor(F(F(vlo - D) - SLACK))
    f1 = np.floor(F(F(vhi + D) + SLACK))
    r0 = 0 if f0 < 0 else int(f0)
    r1 = nv - 1 if f1 > nv - 1 else int(f1)
    if f1 < 0:
        r1 = -1
    return r0, r1


def walk(su, sv, eu, ev, D, nu, nv):
    """The slabs the feed of scan_list_grid lists, with the rows of each from the SHIPPED grid_slab_rows -- which must also equal
    the operation-by-operation model above (the model is the documentation; the shipped function is what is tested)."""
    fa = np.floor(F(F(min(su, eu) - D) - SLACK))
    fb = np.floor(F(F(max(su, eu) + D) + SLACK))
    iuA = 0 if fa < 0 else int(fa)
    iuB = nu - 1 if fb > nu - 1 else int(fb)
    cells = set()
    if iuA <= iuB and fb >= 0:
        ius = list(range(iuA, iuB + 1))
        rows, _ = shipped_slab_rows([[su, sv, eu, ev, D]] * len(ius), ius, nv)
        for iu, (r0, r1) in zip(ius, rows.tolist()):
            m0, m1 = slab_rows(su, sv, eu, ev, D, iu, nv)
            assert (r0 > r1 and m0 > m1) or (r0, r1) == (m0, m1), ("shipped grid_slab_rows differs from its model", su, sv, eu, ev, D, iu)
            for iv in range(r0, r1 + 1):
                cells.add((iu, iv))
    return cells


@pytest.mark.parametrize("seed", range(6))
def test_walk_visits_every_cell_within_d_of_the_segment(seed):
    rng = np.random.default_rng(500 + seed)
    nu, nv = 40, 57
    missed = 0
    for _ in range(400):
        kind = rng.integers(0, 5)
        s = rng.uniform(-3, [nu + 3, nv + 3])
        e = rng.uniform(-3, [nu + 3, nv + 3])
        if kind == 1:    # nearly parallel to v
            e[0] = s[0] + rng.normal() * 10.0 ** rng.uniform(-9, -2)
        elif kind == 2:  # nearly parallel to u
            e[1] = s[1] + rng.normal() * 10.0 ** rng.uniform(-9, -2)
        elif kind == 3:  # short
            e = s + rng.normal(size=2) * 10.0 ** rng.uniform(-6, 0)
        elif kind == 4:  # ends on cell borders
            s, e = np.round(s), np.round(e * 2) / 2
        su, sv, eu, ev = F(s[0]), F(s[1]), F(e[0]), F(e[1])
        D = F(rng.choice([0.05, 0.4, 0.9, 1.6, 3.2]) * rng.uniform(0.5, 1.0))
        got = walk(su, sv, eu, ev, D, nu, nv)
        # every cell that holds a point within D (Chebyshev) of a point of the segment -- sampled densely, in binary64
        t = np.linspace(0.0, 1.0, 801)[:, None]
        pts = np.array([su, sv], dtype=np.float64) + t * (np.array([eu, ev], dtype=np.float64) - np.array([su, sv], dtype=np.float64))
        offs = np.float64(D) * (1.0 - 1e-6) * np.array([[-1, -1], [-1, 1], [1, -1], [1, 1], [0, 0], [-1, 0], [1, 0], [0, -1], [0, 1]], dtype=np.float64)
        q = (pts[:, None, :] + offs[None, :, :]).reshape(-1, 2)
        c = np.floor(q).astype(np.int64)
        ok = (c[:, 0] >= 0) & (c[:, 0] < nu) & (c[:, 1] >= 0) & (c[:, 1] < nv)
        need = set(map(tuple, np.unique(c[ok], axis=0).tolist()))
        missed += len(need - got)
        # and the walk is not wildly generous: within D + 1.5 cells of the segment's bounding box
        for (iu, iv) in got:
            assert min(su, eu) - D - 1.5 <= iu + 1 and iu <= max(su, eu) + D + 1.5
    assert missed == 0
