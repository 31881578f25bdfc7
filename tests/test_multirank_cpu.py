"""world_size-2 (and 3) gloo test of the N>1 path on CPU: row sharding, the padded equal-count
gather to rank 0 and the de-interleave, with the oracle standing in for the per-rank renderer
(tests may use the oracle; the product's multi-GPU code under test is cpuraytracer_amd/distributed.py)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from cpuraytracer_amd import distributed as D
    from oracle import oracle_py as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = O.build_scene("three", 1, W / float(H))
        orc = O.Oracle()
        orc.upload(sc)
        rs = O.RtRowset(0, H, D.BLOCK_ROWS, rank, world)
        orc.render(W, H, 1, 3, 8, 1, rowset=rs)
        hdr, _ = orc.download()
        assert hdr.shape[0] == D.local_rows(H, rank, world)
        parts = D.gather_strip(torch.from_numpy(hdr), H, rank, world)
        # bench.py's exchange: HDR and LDR strip in one buffer, one gather for both (ragged heights: padded to the largest shard)
        orc.resolve()
        hdr2, ldr2 = orc.download()
        xch = D.StripExchange(H, W, rank, world, "cpu")
        xch.hdr[:hdr2.shape[0]] = torch.from_numpy(hdr2)
        xch.ldr[:ldr2.shape[0]] = torch.from_numpy(ldr2)
        ph, pl = xch.gather(through_host=True)
        dist.barrier()
        if rank == 0:
            full = D.assemble([p.numpy() for p in parts], H, world)
            np.save(out_path, full)
            np.save(out_path + ".xh.npy", D.assemble([p.numpy() for p in ph], H, world))
            np.save(out_path + ".xl.npy", D.assemble([p.numpy() for p in pl], H, world))
        else:
            assert ph is None and pl is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H", [(2, 48), (2, 50), (3, 50), (4, 50), (8, 96)])
def test_sharded_render_gathers_to_the_single_rank_image(built, oracle, tmp_path, world, H):
    import torch.multiprocessing as mp
    W = 64
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(world, _free_port(), H, W, out), nprocs=world, join=True)
    got = np.load(out)
    sc = oracle.build_scene("three", 1, W / float(H))
    orc = oracle.Oracle()
    orc.upload(sc)
    orc.render(W, H, 1, 3, 8, 1)
    orc.resolve()
    want, want_ldr = orc.download()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(np.load(out + ".xh.npy").view(np.uint32), want.view(np.uint32))  # the one-buffer exchange: HDR ...
    assert np.array_equal(np.load(out + ".xl.npy"), want_ldr)                              # ... and LDR
