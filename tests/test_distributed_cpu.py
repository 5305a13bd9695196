"""CPU tier, world_size = 2, gloo: the multi-rank path of bench.py -- 8-row strips dealt
round-robin over ranks, every peer's strips sent to rank 0, scattered into the image.  On the
CPU the strips are rendered by the oracle (there is no GPU here); the partition, the gather
and the assembly are the code the GPU path uses (sdf_playground_amd strip helpers)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

W, H = 40, 27  # ragged last strip


def _free_port():
    # from below the ephemeral range: a port handed out by bind(0) can become the source port of somebody's outgoing connection
    # before the rendezvous listens on it (bench.py, _free_port)
    import random

    for _ in range(64):
        p = random.randrange(20000, 32000)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", p))
            return p
        except OSError:
            continue
        finally:
            s.close()
    raise RuntimeError("no free port")


def _worker(rank, world, port, out_path, wire):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import sdf_playground_amd as sp
    from oracle import pyoracle as po

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f = po.default_frame("labyrinth", W, H, stime=0.5)
    n = sp.strip_buffer_pixels_host(W, H, world)
    local = np.zeros((n // W, W, 4), np.float32)
    rows = sp.strip_rows_of_rank(H, rank, world)
    full = np.zeros((H, W, 4), np.float32)
    for row in rows:  # this rank renders only its own rows
        po.render("labyrinth", f, region=(0, row, W, row + 1), out=full, nthreads=1)
    for row in rows:
        strip = row // sp.STRIP_ROWS
        local[(strip // world) * sp.STRIP_ROWS + row % sp.STRIP_ROWS] = full[row]
    # strips travel packed (uint8 buffers) exactly as on the GPUs: 13 bytes per pixel (fp32, lossless) or
    # 7 (the reference's RGBA16F target); the transfer is the pattern of sdfr_render_gather: every peer
    # sends its buffer to rank 0, which receives them into consecutive slots (slot 0 = its own strips)
    pack, unpack = (sp.pack_strip_host, sp.unpack_strip_host) if wire == "f32" else (sp.pack_strip16_host, sp.unpack_strip16_host)
    t = torch.from_numpy(pack(local.reshape(n, 4)))
    if rank == 0:
        gathered = [t] + [torch.empty_like(t) for _ in range(1, world)]
        reqs = [dist.irecv(gathered[p], src=p) for p in range(1, world)]
        for q in reqs:
            q.wait()
        unpacked = np.stack([unpack(g.numpy(), n) for g in gathered])
        img = sp.assemble_strips_host(W, H, world, unpacked)
        np.save(out_path, img)
    else:
        dist.send(t, dst=0)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,wire", [(2, "f32"), (2, "f16")])
def test_two_rank_strip_gather_matches_single_rank(oracle, tmp_path, world, wire):
    out_path = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), out_path, wire), nprocs=world, join=True)
    img = np.load(out_path)
    f = oracle.default_frame("labyrinth", W, H, stime=0.5)
    ref, _, _ = oracle.render("labyrinth", f)
    if wire == "f32":
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    else:  # the RGBA16F target: the fp32 result rounded to nearest even (Postprocessing.cpp:23)
        assert img.dtype == np.float16
        with np.errstate(over="ignore"):
            assert np.array_equal(img.view(np.uint16), ref.astype(np.float16).view(np.uint16))
        assert np.array_equal(img.view(np.uint16), oracle.float_to_half(ref).view(np.uint16))


def test_strip_partition_covers_every_row_once():
    import sdf_playground_amd as sp

    for h in (1, 7, 8, 9, 27, 64, 2160):
        for world in (1, 2, 3, 4, 8):
            for split in ((0, 1), (1, 4), (5, 16), (3, 4)):
                rows = sorted([r for k in range(world) for r in sp.strip_rows_of_rank(h, k, world, split)] + sp.private_rows_host(h, split))
                assert rows == list(range(h)), (h, world, split)
                assert all(len(sp.strip_rows_of_rank(h, k, world, split)) <= sp.strip_buffer_pixels_host(1, h, world, split) for k in range(world))
                # rows sit in a rank's buffer in increasing order, whole strips at 8-row boundaries
                for k in range(world):
                    rk = sp.strip_rows_of_rank(h, k, world, split)
                    assert rk == sorted(rk)
