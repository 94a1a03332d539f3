#!/usr/bin/env python3
"""Headline benchmark: receive-path verdicts (IPv4 validate + UDP pseudo-header checksum) over
4,194,304 synthetic 1500-byte packets per GPU, resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one launch of the verify kernel over the rank's whole batch.  Ranks hold disjoint index
ranges of the same trace (weak scaling); nothing crosses GPUs on the data path, only the per-verdict
counters are all-reduced (RCCL) after the timed region.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

import unet_amd
from unet_amd import Verdict, shard

SEED = 0x756E6574
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak

WORKLOADS = {
    # name: (trace kind, frames per GPU, fixed stride in 16-byte slots or 0)
    "udp1500": (unet_amd.SYNTH_UDP1500, 1 << 22, 94),
    "imix": (unet_amd.SYNTH_IMIX, 1 << 23, 0),
    "hdr20": (unet_amd.SYNTH_IPV4_HDR20, 1 << 24, 0),
}


def cpu_baseline(kind, ifc, budget_s):
    """The CPU oracle (scalar restatement of the reference, one thread -- the reference processes every
    packet on its main thread) timed on a bounded sample of the same trace."""
    from oracle import oracle_ffi as O

    n = 1 << 15 if kind != unet_amd.SYNTH_IPV4_HDR20 else 1 << 20
    arena, off16, lens = unet_amd.synth_host(kind, SEED, 0, n, ifc)
    if kind == unet_amd.SYNTH_IPV4_HDR20:
        run = lambda: O.ipv4_validate_batch(arena[:n * 20], ifc.ext_tcp)
    else:
        run = lambda: O.recv_batch(arena, off16, lens, ifc)
    run()
    passes, t0 = 0, time.perf_counter()
    while True:
        run()
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    return {"value": n * passes / dt / 1e9, "unit": "Gpkt/s", "cores": 1, "kind": "port",
            "sample": f"{n} records of the same trace x {passes} passes ({dt:.1f} s, oracle/unet_oracle.c, gcc -O2, 1 thread)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="udp1500", choices=sorted(WORKLOADS))
    ap.add_argument("--ext-tcp", type=int, default=None, help="TCP extension on/off (default: on for imix only)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--frames", type=int, default=0, help="override frames per GPU")
    ap.add_argument("--rotate", type=int, default=0, help="number of distinct batches cycled through (default: 6 for hdr20, else 1)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the offload has no CPU path")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    kind, n, stride16 = WORKLOADS[args.workload]
    if args.frames:
        n = args.frames
    ext_tcp = (args.workload == "imix") if args.ext_tcp is None else bool(args.ext_tcp)
    ifc = unet_amd.Iface(ext_tcp=ext_tcp)
    ctx = unet_amd.Offload(local)
    # A dedicated torch stream: its raw handle goes to the C ABI, so the kernels, torch's own ops and the
    # HIP events used for timing (torch.cuda.Event records on the current stream) share one queue.
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream, "need a non-null stream handle (NULL selects the library's private stream)"
    first_idx, n = shard.rank_index_range(rank, world, n)

    # ---- build the batch(es) in HBM.  A workload whose arena is not far larger than the 256 MiB Infinity
    # Cache is rotated over several distinct batches (consecutive index ranges of the same trace) so that
    # no launch finds its input in cache.
    rotate = args.rotate if args.rotate else (6 if kind == unet_amd.SYNTH_IPV4_HDR20 else 1)
    steps_fns, keep = [], []
    for r in range(rotate):
        f_idx = first_idx + r * world * n
        off16, lens, nbytes = unet_amd.synth_layout(kind, SEED, f_idx, n)
        d_arena = torch.empty(nbytes + 16, dtype=torch.uint8, device="cuda")
        d_out = torch.zeros(n, dtype=torch.int32, device="cuda")
        if kind == unet_amd.SYNTH_IPV4_HDR20:
            ctx.synth_dev(kind, SEED, f_idx, n, ifc, None, None, d_arena.data_ptr(), nbytes, stream)
            alg_bytes = n * 24  # 20 read + 4 written per header
            fn = (lambda a=d_arena, o=d_out: ctx.ipv4_validate_dev(a.data_ptr(), n, ifc, o.data_ptr(), stream))
            kernel = "k_ipv4_validate"
        else:
            d_off = torch.from_numpy(off16).cuda()
            d_len = torch.from_numpy(lens).cuda()
            ctx.synth_dev(kind, SEED, f_idx, n, ifc, d_off.data_ptr(), d_len.data_ptr(), d_arena.data_ptr(), nbytes, stream)
            alg_bytes = int(lens.astype(np.int64).sum()) + 8 * n  # record bytes + 4 (length) + 4 (result)
            off_ptr = None if stride16 else d_off.data_ptr()
            fn = (lambda a=d_arena, o=d_out, l=d_len, op=off_ptr, nb=nbytes:
                  ctx.verify_dev(a.data_ptr(), nb, op, stride16, l.data_ptr(), n, ifc, o.data_ptr(), stream))
            kernel = "k_verify<MODE_L3>"
            keep.append((d_off, d_len))
        steps_fns.append(fn)
        keep.append((d_arena, d_out))
        if r == 0:
            d_arena0, d_out0, nbytes0, alg_bytes0 = d_arena, d_out, nbytes, alg_bytes
    d_arena, d_out, nbytes, alg_bytes = d_arena0, d_out0, nbytes0, alg_bytes0
    counter = [0]

    def step():
        steps_fns[counter[0] % rotate]()
        counter[0] += 1
    torch.cuda.synchronize()

    # ---- timed region
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream
    wall = shard.max_over_ranks(wall, "cuda", dist)

    # ---- epilogue: per-verdict counters, summed across GPUs (the only cross-GPU traffic)
    d_cnt = torch.zeros(32, dtype=torch.int64, device="cuda")
    ctx.count_verdicts_dev(d_out.data_ptr(), n, d_cnt.data_ptr(), stream)
    torch.cuda.synchronize()
    local_cnt = d_cnt.cpu().numpy().copy()
    cnt = shard.allreduce_counts(d_cnt, dist).cpu().numpy()
    parity = None
    if args.workload == "udp1500":
        # per-index fault rules of the trace: idx % 256 == 33 -> length mismatch; idx % 64 == 7 or
        # idx % 256 == 97 -> checksum failure (a zeroed field can verify only if the true value was 0)
        idx = np.arange(first_idx, first_idx + n, dtype=np.int64)
        n_len = int((idx % 256 == 33).sum())
        n_ck = int(((idx % 64 == 7) | (idx % 256 == 97)).sum())
        parity = bool(local_cnt[Verdict.DROP_UDP_LEN] == n_len and n_ck - 16 <= local_cnt[Verdict.DROP_UDP_CSUM] <= n_ck
                      and local_cnt[Verdict.ACCEPT_UDP] == n - n_len - local_cnt[Verdict.DROP_UDP_CSUM])

    # ---- attainable-bandwidth yardstick: read-only streaming kernel over the same buffer
    d_sink = torch.zeros(4096, dtype=torch.int32, device="cuda")
    for _ in range(2):
        ctx.stream_read_dev(d_arena.data_ptr(), nbytes, d_sink.data_ptr(), stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ctx.stream_read_dev(d_arena.data_ptr(), nbytes, d_sink.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    stream_gbps = nbytes * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9

    if rank == 0:
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes of this same command (tools/pmc_traffic.py), if the
        # committed measurement matches this workload and batch size
        traffic = None
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(args.workload)
            if rec and rec["frames"] == n:
                traffic = rec["traffic_bytes"]
        except (OSError, ValueError, KeyError):
            pass
        line = {
            "metric": "Gpkt/s and achieved HBM GB/s, batched IPv4+TCP/UDP checksum+parse, 1500B",
            "value": n * world * args.steps / wall / 1e9,
            "unit": "Gpkt/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": {"udp1500": "4,194,304 x 1500-byte IPv4/UDP packets per GPU (BASELINE configs[2])",
                                    "imix": "8,388,608 mixed 64-9000-byte TCP/UDP/ICMP packets per GPU (BASELINE configs[3])",
                                    "hdr20": "16,777,216 bare 20-byte IPv4 headers per GPU (BASELINE configs[1])"}[args.workload],
                       "frames_per_gpu": n, "arena_bytes_per_gpu": nbytes, "ext_tcp": ext_tcp, "batches_rotated": rotate,
                       "sharding": "disjoint index ranges per rank, no data-path collective; counters all-reduced (RCCL) after the timed region",
                       "inputs": "generated in HBM by unet_synth_dev before the timed region"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "kernel": kernel,
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "stream_read_GBps": stream_gbps, "frac_of_stream_read": achieved / stream_gbps},
            "hbm_GBps_all_gpus": alg_bytes * world * args.steps / wall / 1e9,
            "verdict_counts": {Verdict(i).name: int(c) for i, c in enumerate(cnt) if c},
            "parity_check": parity,
        }
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(kind, ifc, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    ctx.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
