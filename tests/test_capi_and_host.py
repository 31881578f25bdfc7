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
    assert L.rt_api_version() == 1


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
    assert L.rt_scene_upload(None, None, None, 0, None, None, None, 1.0) == 2
    assert L.rt_resolve(None, 1) == 2 and L.rt_download(None, None, None) == 2


def test_rowset_arithmetic_matches_oracle_and_python(built, oracle):
    from cpuraytracer_amd import _capi, distributed as D
    L = _capi.load()
    O = oracle.lib()
    for H in (1, 4, 50, 100, 800, 1080):
        for world in (1, 2, 3, 4, 8):
            seen = np.zeros(H, dtype=int)
            for rank in range(world):
                rs = _capi.cyclic_rows(H, rank, world)
                n = L.rt_rowset_local_rows(rs)
                assert n == O.orc_rowset_local_rows(oracle.RtRowset(0, H, 4, rank, world)) == D.local_rows(H, rank, world)
                for lr in range(n):
                    j = L.rt_rowset_global_row(rs, lr)
                    assert j == D.global_row(lr, rank, world) == O.orc_rowset_global_row(oracle.RtRowset(0, H, 4, rank, world), lr)
                    seen[j] += 1
            assert (seen == 1).all()
    assert L.rt_rowset_local_rows(_capi.RtRowset(0, 8, 0, 0, 1)) == 0  # degenerate: block_rows 0
    assert L.rt_rowset_local_rows(_capi.RtRowset(0, 8, 4, 2, 2)) == 0  # shard >= nshards


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
