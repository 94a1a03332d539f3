"""CPU-only, world_size 2 over gloo: the rank decomposition and the counter all-reduce that bench.py
uses on GPUs.  The per-rank verdicts come from the CPU oracle here (no GPU in this test)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import unet_amd
from unet_amd import shard
from oracle import oracle_ffi as O

SEED = 0x756E6574
N_PER_RANK = 3000


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ifc = unet_amd.Iface(ext_tcp=True)
        if mode == "index":
            first, n = shard.rank_index_range(rank, world, N_PER_RANK)
            arena, off16, lens = unet_amd.synth_host(unet_amd.SYNTH_IMIX, SEED, first, n, ifc)
            res = O.recv_batch(arena, off16, lens, ifc)
            mine = np.arange(first, first + n)
        else:
            n_all = N_PER_RANK * world
            arena, off16, lens = unet_amd.synth_host(unet_amd.SYNTH_IMIX, SEED, 0, n_all, ifc)
            hashes = np.array([O.flow_hash_record(arena[int(o) * 16:int(o) * 16 + int(l)].tobytes(), ifc)
                               for o, l in zip(off16, lens)], dtype=np.uint32)
            mine = np.nonzero(shard.shard_of_hash(hashes, world) == rank)[0]
            res = O.recv_batch(arena, off16[mine], lens[mine], ifc)
        counts = torch.from_numpy(np.bincount(res["verdict"], minlength=unet_amd.V_COUNT).astype(np.int64))
        local = counts.clone()
        shard.allreduce_counts(counts, dist)
        t = shard.max_over_ranks(1.0 + rank, "cpu", dist)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), total=counts.numpy(), local=local.numpy(), mine=mine, tmax=t)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["index", "flowhash"])
def test_two_ranks_cover_the_trace_and_counters_add_up(tmp_path, mode):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"r{i}.npz") for i in range(world)]
    ifc = unet_amd.Iface(ext_tcp=True)
    arena, off16, lens = unet_amd.synth_host(unet_amd.SYNTH_IMIX, SEED, 0, N_PER_RANK * world, ifc)
    exp = np.bincount(O.recv_batch(arena, off16, lens, ifc)["verdict"], minlength=unet_amd.V_COUNT)
    # disjoint cover
    allidx = np.concatenate([x["mine"] for x in r])
    assert np.array_equal(np.sort(allidx), np.arange(N_PER_RANK * world))
    # every rank ends with the same global counters, equal to the single-process answer
    for x in r:
        assert np.array_equal(x["total"], exp)
        assert float(x["tmax"]) == 2.0
    assert np.array_equal(r[0]["local"] + r[1]["local"], exp)


def test_rank_ranges_and_hash_placement():
    assert shard.rank_index_range(0, 8, 1 << 22) == (0, 1 << 22)
    assert shard.rank_index_range(7, 8, 1 << 22) == (7 << 22, 1 << 22)
    with pytest.raises(ValueError):
        shard.rank_index_range(8, 8, 4)
    h = np.array([0, 1, 7, 8, 0xFFFFFFFF], dtype=np.uint32)
    assert shard.shard_of_hash(h, 8).tolist() == [0, 1, 7, 0, 7]
    # a flow keeps its GPU: identical 5-tuples hash identically; the hash spreads evenly
    rng = np.random.default_rng(0)
    hs = np.array([unet_amd.flow_hash5(int(a), int(b), 17, int(c), int(d))
                   for a, b, c, d in rng.integers(0, 2**16, size=(20000, 4))], dtype=np.uint32)
    for world in (2, 4, 8):
        assert shard.shard_skew(shard.shard_of_hash(hs, world), world) < 1.08
    assert unet_amd.flow_hash5(1, 2, 17, 3, 4) == unet_amd.flow_hash5(1, 2, 17, 3, 4)
    with pytest.raises(ValueError):
        shard.allreduce_counts(torch.zeros(3, dtype=torch.int64))
