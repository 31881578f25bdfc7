"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/rt_api.h
declares, fails loudly without a device, and the host-side row-set / sharding logic is right."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def _declared_functions(header):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built):
    from cpuraytracer_amd import _capi
    L = _capi.load()
    declared = _declared_functions(os.path.join(ROOT, "include", "rt_api.h"))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), "librt_hip.so does not export %s" % name
    assert L.rt_api_version() == 2  # 2: rt_scene_upload takes the light LIST


def test_library_contains_gfx950_code_object(built):
    from cpuraytracer_amd import LIB_PATH
    blob = open(LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"rt_trace_kernel" in blob


def test_struct_layouts_match_header(built):
    from cpuraytracer_amd import _capi
    assert C.sizeof(_capi.RtSphere) == 16 and C.sizeof(_capi.RtMaterial) == 48
    assert C.sizeof(_capi.RtCamera) == 72 and C.sizeof(_capi.RtLight) == 28 and C.sizeof(_capi.RtRowset) == 20
    assert _capi.SPHERE_DTYPE.itemsize == 16 and _capi.MATERIAL_DTYPE.itemsize == 48


def _has_gpu():
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device failure mode")
def test_no_cpu_fallback_without_device(built):
    from cpuraytracer_amd import HipRenderer, RtError
    with pytest.raises(RtError) as e:
        HipRenderer(0)
    assert e.value.code == 1 and "no CPU fallback" in str(e.value)
    cli = os.path.join(ROOT, "cpuraytracer_amd", "lib", "spheres")
    p = subprocess.run([cli, "--width", "8", "--height", "8", "--spp", "1"], capture_output=True, text=True)
    assert p.returncode == 1 and "no HIP device" in p.stderr


def test_null_arguments_are_rejected_without_touching_the_gpu(built):
    from cpuraytracer_amd import _capi
    L = _capi.load()
    assert L.rt_render(None, 8, 8, _capi.whole_image(8), 1, 2, 8, 1, None) == 2
    assert b"null ctx" in L.rt_last_error()
    assert L.rt_scene_upload(None, None, None, 0, None, None, 0, None, 1.0) == 2
    assert L.rt_resolve(None, 1) == 2 and L.rt_download(None, None, None) == 2


def test_rowset_arithmetic_matches_oracle_and_python(built, oracle):
    from cpuraytracer_amd import _capi, distributed as D
    L = _capi.load()
    O = oracle.lib()
    assert D.BLOCK_ROWS == _capi.BLOCK_ROWS
    for br in (1, 4):  # the partition's default (single rows) and round 1's 4-row blocks
        for H in (1, 4, 50, 100, 800, 1080):
            for world in (1, 2, 3, 4, 8):
                seen = np.zeros(H, dtype=int)
                for rank in range(world):
                    rs = _capi.cyclic_rows(H, rank, world, br)
                    n = L.rt_rowset_local_rows(rs)
                    assert n == O.orc_rowset_local_rows(oracle.RtRowset(0, H, br, rank, world)) == D.local_rows(H, rank, world, br)
                    for lr in range(n):
                        j = L.rt_rowset_global_row(rs, lr)
                        assert j == D.global_row(lr, rank, world, br) == O.orc_rowset_global_row(oracle.RtRowset(0, H, br, rank, world), lr)
                        seen[j] += 1
                assert (seen == 1).all()  # a partition of the rows
    assert L.rt_rowset_local_rows(_capi.RtRowset(0, 8, 0, 0, 1)) == 0  # degenerate: block_rows 0


def test_assemble_deinterleaves(built):
    from cpuraytracer_amd import distributed as D
    H, W = 50, 3
    full = np.arange(H * W * 2, dtype=np.float32).reshape(H, W, 2)
    for world in (1, 2, 4, 8):
        parts = []
        for r in range(world):
            rows = [D.global_row(lr, r, world) for lr in range(D.local_rows(H, r, world))]
            p = full[rows]
            pad = np.full((D.max_local_rows(H, world) - len(rows), W, 2), -1, dtype=np.float32)
            parts.append(np.concatenate([p, pad], 0))
        assert np.array_equal(D.assemble(parts, H, world), full)


def _layout(spheres):
    import ctypes as C
    from cpuraytracer_amd import _capi
    L = _capi.load()
    n = C.c_uint32(0)
    sph = np.ascontiguousarray(spheres)
    _capi.check(L.rt_unit_layout(sph.ctypes.data, sph.shape[0], 0, C.byref(n), None, None))
    orig = np.zeros(n.value * 4, dtype=np.uint32)
    bounds = np.zeros((n.value, 4), dtype=np.float32)
    _capi.check(L.rt_unit_layout(sph.ctypes.data, sph.shape[0], n.value, C.byref(n), orig.ctypes.data, bounds.ctypes.data))
    return orig.reshape(-1, 4), bounds


@pytest.mark.parametrize("name", ["cover", "three", "grid10k"])
def test_clustered_layout_is_a_partition_with_enclosing_bounds(built, oracle, name):
    """Every sphere appears in exactly one group, and each group's bound encloses its members with the margins
    the filter's conservativeness argument needs."""
    sc = oracle.build_scene(name, 1, 1.5)
    orig, bounds = _layout(sc.spheres)
    assert orig.shape[0] % 2 == 0
    members = orig[orig != 0xFFFFFFFF]
    assert sorted(members.tolist()) == list(range(sc.n))
    c = np.stack([sc.spheres["cx"], sc.spheres["cy"], sc.spheres["cz"]], 1).astype(np.float64)
    r = sc.spheres["r"].astype(np.float64)
    # rt_scan.h: kMarginK = 4096 when the groups ARE the matrix-core level (<= 128 groups), kMarginKValu = 2048
    # when a hierarchy sits above them and they are tested on the VALU
    keps = (4096 if orig.shape[0] <= 128 else 2048) * 2.0 ** -24
    for g in range(orig.shape[0]):
        ids = orig[g][orig[g] != 0xFFFFFFFF]
        if len(ids) == 0:
            assert bounds[g, 3] >= 1e29  # padding group can never pass the filter
            continue
        if orig.shape[0] > 128 and len(ids) == 1 and g < 8 and r[ids[0]] > 4 * np.median(r) and bounds[g, 3] >= 1e29:
            # hierarchy scan: a leading big-sphere group stays OUT of the bounds (TraceParams::n_always: every ray tests it
            # exactly) -- its bound must then be the never-a-candidate one, and the sphere is still listed (checked above)
            continue
        C_ = bounds[g, :3].astype(np.float64)
        s_i = np.linalg.norm(c[ids] - C_, axis=1)
        R = (s_i + r[ids]).max()
        Rf2 = float(C_ @ C_) - float(bounds[g, 3])
        Cn = np.linalg.norm(C_)
        need = R * R + 0.01 * s_i.max() ** 2 + keps * (2 * (Cn + R) ** 2 + R * R)
        assert Rf2 >= need * (1 - 1e-9), (g, Rf2, need)
    if name == "cover":
        # compact groups: the k-d split keeps the small-sphere bounds under two grid cells
        small = [g for g in range(orig.shape[0]) if (orig[g] != 0xFFFFFFFF).sum() > 1]
        Rs = [np.sqrt(float(bounds[g, :3].astype(np.float64) @ bounds[g, :3].astype(np.float64)) - float(bounds[g, 3])) for g in small]
        assert max(Rs) < 2.2


def test_invariant_division_by_multiplication_is_exact():
    """rt_params.h FastDiv (the trace kernel's path index -> tile / row / row block): the Granlund-Montgomery formula,
    restated here, equals n // d for every 32-bit n; the device code itself is covered by every image parity test (a wrong
    quotient moves a path to another pixel)."""
    def make(d):
        l = 0
        while l < 32 and (1 << l) < d:
            l += 1
        return ((((1 << l) - d) << 32) // d + 1) & 0xffffffff, min(l, 1), max(l - 1, 0)
    rng = np.random.default_rng(5)
    divisors = [1, 2, 3, 5, 7, 64, 1200, 1920, 4096, 8192, 64 * 128, 64 * 1024, 65535, 65536, 2**31 - 1, 2**31, 2**31 + 1, 2**32 - 1]
    divisors += [int(x) for x in rng.integers(1, 2**32, 200)] + [int(x) for x in rng.integers(1, 2**16, 200)]
    for d in divisors:
        m, s1, s2 = make(d)
        n = np.concatenate([rng.integers(0, 2**32, 20000, dtype=np.uint64),
                            np.array([0, 1, d - 1, d, min(d + 1, 2**32 - 1), min(2 * d, 2**32 - 1), 2**32 - 1, 2**31], dtype=np.uint64)])
        t = (n * np.uint64(m)) >> np.uint64(32)
        q = ((t + ((n - t) >> np.uint64(s1))) & np.uint64(0xffffffff)) >> np.uint64(s2)
        # (t + ((n - t) >> s1)) never exceeds 32 bits for a valid magic: the mask only documents the register width
        assert np.array_equal(q, n // np.uint64(d)), d


def test_which_scan_a_scene_is_laid_out_for(built, oracle):
    """rt_scene_upload's choice of closest-hit scan (host logic, no GPU): the cover scene fits the flat matrix-core filter; the
    10,004-sphere layer gets the cell grid (about one sphere per cell, the four big spheres tested for every ray); a layer whose
    spheres are clumped, or a scene with more than eight big spheres, goes to the bounds hierarchy."""
    from cpuraytracer_amd import _capi
    L = _capi.load()

    def info(spheres):
        sph = np.ascontiguousarray(spheres)
        out = (C.c_uint32 * 5)()
        _capi.check(L.rt_unit_layout_info(sph.ctypes.data, sph.shape[0], out))
        return list(out)
    assert info(oracle.build_scene("cover", 1, 1.5).spheres)[0] == 0
    assert info(oracle.build_scene("three", 1, 2.0).spheres)[0] == 0
    g = info(oracle.build_scene("grid10k", 1, 1.0).spheres)
    assert g[0] == 1 and g[3] == 4 and 0.6 * 10000 <= g[1] * g[2] <= 1.6 * 10000 and g[4] == 1
    rng = np.random.default_rng(4)

    def layer(n, side, clump=0, n_big=0):
        sph = np.zeros(n + 1 + n_big, dtype=oracle.SPHERE_DTYPE)
        a, b = rng.uniform(-side, side, n), rng.uniform(-side, side, n)
        a[:clump], b[:clump] = rng.uniform(-1.5, 1.5, clump), rng.uniform(-1.5, 1.5, clump)
        sph["cx"][:n], sph["cy"][:n], sph["cz"][:n], sph["r"][:n] = a, 0.2, b, 0.2
        sph["cy"][n], sph["r"][n] = -1000.0, 1000.0
        for k in range(n_big):
            sph[n + 1 + k] = (rng.uniform(-side, side), 3.0, rng.uniform(-side, side), 3.0)
        return sph
    assert info(layer(2000, 30.0))[0] == 1
    assert info(layer(2000, 100.0, clump=1500))[0] == 2   # clumped: the average sphere shares its cell with many others
    assert info(layer(2000, 30.0, n_big=9))[0] == 2       # ten big spheres with the floor: more than the every-ray list takes
    assert info(layer(300, 10.0))[0] == 0                 # 76 groups: the flat filter
