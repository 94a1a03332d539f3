#!/usr/bin/env python3
"""Benchmark of the offloaded receive path on MI355X.

    python bench.py --gpus N --steps K --warmup W                    # headline: udp1500
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json configs):
  udp1500  4,194,304 x 1500-byte IPv4/UDP packets per GPU          (configs[2], the headline metric)
  imix     8,388,608 mixed 64..9000-byte TCP/UDP/ICMP packets      (configs[3])
  hdr20    16,777,216 bare 20-byte IPv4 headers                    (configs[1])
  trace    a long flow-table trace generated chunk by chunk in HBM (configs[4]; --total-frames)
  eth1514  4,194,304 Ethernet frames of 1514 bytes judged in L2 mode
  csum / generate / txbuild / tximix   the other entry points (calculate_checksum, in-place checksum fill,
           packet building from payloads of 1472 bytes or of IMIX sizes)

One "step" = one launch of the kernel over the rank's whole resident batch.  With N > 1 the global batch
is N times larger and every rank takes the records whose 5-tuple flow hash % N is its rank (computed
from the trace definition, before any record exists), so a flow stays on one GPU and nothing crosses
GPUs on the data path; only the per-verdict counters (UNET_V_COUNT u64) are all-reduced (RCCL) after the timed region.
Rank 0 prints one JSON line.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
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
TX_LAYOUT = "auto"   # --tx-layout: where the transmit workloads put their packets ("packed", or "lines" = on 128-byte lines)
TX_FILL_LINES = 1    # UNET_TX_FILL_LINES
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak

WORKLOADS = {
    # name: (trace kind, records per GPU per batch, fixed stride in 16-byte slots or 0)
    "udp1500": (unet_amd.SYNTH_UDP1500, 1 << 22, 94),
    "imix": (unet_amd.SYNTH_IMIX, 1 << 23, 0),
    "odd": (unet_amd.SYNTH_ODD, 1 << 23, 0),
    "hdr20": (unet_amd.SYNTH_IPV4_HDR20, 1 << 24, 0),
    "trace": (unet_amd.SYNTH_FLOWS, 1 << 24, 0),
    # the other entry points over the udp1500 batch
    "csum": (unet_amd.SYNTH_UDP1500, 1 << 22, 94),
    "generate": (unet_amd.SYNTH_UDP1500, 1 << 22, 94),
    "fields": (unet_amd.SYNTH_UDP1500, 1 << 22, 94),
    "txbuild": (unet_amd.SYNTH_UDP1500, 1 << 22, 94),
    "eth1514": (unet_amd.SYNTH_UDP1500, 1 << 22, 94),
    "tximix": (unet_amd.SYNTH_IMIX, 1 << 23, 0),
    # extension (no TCP in the reference): verdicts, then per-flow TCP tracking of the accepted segments
    "tcptrack": (unet_amd.SYNTH_FLOWS, 1 << 24, 0),
}
DESCRIPTION = {
    "udp1500": "4,194,304 x 1500-byte IPv4/UDP packets per GPU (BASELINE configs[2])",
    "imix": "8,388,608 mixed 64-9000-byte TCP/UDP/ICMP packets per GPU (BASELINE configs[3])",
    "odd": "8,388,608 mixed TCP/UDP/ICMP packets per GPU of 65 / 577 / 1499 / 4097 / 8999 bytes (odd lengths, one byte past or short of a 16-byte boundary)",
    "hdr20": "16,777,216 bare 20-byte IPv4 headers per GPU (BASELINE configs[1])",
    "trace": "flow-table trace (2^20 flows, IMIX sizes) generated in HBM in chunks of 16,777,216 records per GPU (BASELINE configs[4])",
    "csum": "calculate_checksum over 4,194,304 x 1500-byte records per GPU (unet_csum_batch_dev)",
    "generate": "transmit-side checksum fill of 4,194,304 x 1500-byte IPv4/UDP packets per GPU, in place (unet_generate_dev)",
    "fields": "transmit-side checksum values of 4,194,304 x 1500-byte IPv4/UDP packets per GPU, packets left untouched (unet_checksum_fields_dev)",
    "txbuild": "4,194,304 x 1500-byte IPv4/UDP packets per GPU built from 1472-byte payloads + descriptors (unet_tx_build_dev)",
    "tximix": "8,388,608 IPv4/UDP packets per GPU built from payloads of IMIX sizes (36-8972 bytes) + descriptors (unet_tx_build_dev)",
    "tcptrack": "EXTENSION, parity unpinned (the reference has no TCP): 16,777,216 records of the flow trace per GPU judged (unet_verify_dev, ext_tcp) and their "
                "accepted TCP segments applied to a per-flow table of 4,194,304 slots (unet_tcp_track_ws_dev)",
    "eth1514": "4,194,304 x 1514-byte Ethernet frames per GPU (IPv4/UDP inside, built by unet_tx_build_dev in L2 mode, faults injected), judged in L2 mode",
}


KERNEL_SOURCES = ("unet_amd/csrc/kernels.h", "unet_amd/csrc/csum_core.h", "unet_amd/csrc/parse.h")


def kernels_sha256():
    """Digest of the kernel sources: ties a committed PMC traffic figure to the code it was measured on."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def committed_traffic(workload, frames, single_gpu):
    """(bytes per launch or None, where the figure comes from).  The figure is a committed PMC measurement
    (profiles/traffic.json), not a measurement of this run: it is only reported when it was taken on this
    workload, this batch size and exactly these kernel sources."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(workload)
    except (OSError, ValueError):
        return None, "profiles/traffic.json unreadable"
    if not rec:
        return None, "no committed PMC pass for this workload"
    if rec.get("frames") != frames or not single_gpu:
        return None, "committed PMC pass was taken on another batch size or sharding"
    if rec.get("kernels_sha256") != kernels_sha256():
        return None, f"stale: the committed PMC pass ({rec.get('source', '?')}) was taken on other kernel sources"
    return rec["traffic_bytes"], f"committed PMC pass, not this run: {rec.get('source', '?')}"


def self_launch(argv, gpus):
    """`python bench.py --gpus N` with no launcher around it: start N fresh ranks under torch.distributed.run as a
    child process (this process has not touched the GPU and never does) and hand back its exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "--max-restarts", "0", os.path.abspath(__file__)] + list(argv)
    # the launcher ends every rank when one of them fails and exits non-zero itself: that code is this job's
    rc = subprocess.call(cmd)
    if rc != 0:
        print(f"bench.py: the {gpus}-rank job failed (launcher exit code {rc})", file=sys.stderr, flush=True)
    return rc if rc else 0


def cpu_baseline(kind, ifc, budget_s):
    """The CPU oracle (scalar restatement of the reference) timed on a bounded sample of the same trace.
    `value` is one thread -- the reference processes every packet on its main thread.  `all_cores` adds
    the same code over disjoint record ranges on every host core this job may use, and the same verdict
    logic with a 64-bit-accumulator checksum as a best-effort CPU line."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle_ffi as O

    n = 1 << 15 if kind != unet_amd.SYNTH_IPV4_HDR20 else 1 << 20
    arena, off16, lens = unet_amd.synth_host(kind, SEED, 0, n, ifc)
    hdr = kind == unet_amd.SYNTH_IPV4_HDR20

    def run_range(lo, hi, fast=False):
        if hdr:
            return O.ipv4_validate_batch(arena[lo * 20:hi * 20], ifc.ext_tcp)
        return O.recv_batch(arena, off16[lo:hi], lens[lo:hi], ifc, fast=fast)

    def timed(fn, budget):
        fn()
        passes, t0 = 0, time.perf_counter()
        while True:
            fn()
            passes += 1
            dt = time.perf_counter() - t0
            if dt >= budget:
                return n * passes / dt / 1e9, passes, dt

    v1, passes, dt = timed(lambda: run_range(0, n), budget_s)
    out = {"value": v1, "unit": "Gpkt/s", "cores": 1, "kind": "port",
           "sample": f"{n} records of the same trace x {passes} passes ({dt:.1f} s, oracle/unet_oracle.c, gcc -O2, 1 thread)"}
    cores = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1), os.cpu_count() or 1, 64))
    if cores > 1:
        cuts = [n * i // cores for i in range(cores + 1)]
        with ThreadPoolExecutor(cores) as pool:  # ctypes drops the GIL inside the C batch drivers
            par = lambda fast: list(pool.map(lambda i: run_range(cuts[i], cuts[i + 1], fast), range(cores)))
            vn, _, _ = timed(lambda: par(False), min(3.0, budget_s))
            out["all_cores"] = {"cores": cores, "value": vn, "unit": "Gpkt/s"}
            if not hdr:
                vf, _, _ = timed(lambda: par(True), min(3.0, budget_s))
                out["all_cores"]["value_fast_checksum"] = vf
    return out


class Batch:
    """One resident batch of a rank: arena + tables + result buffer in HBM."""

    def __init__(self, ctx, stream, kind, ifc, first_idx, n_global, world, rank, stride16, op="verify"):
        self.kind, self.ifc, self.ctx, self.stream, self.op = kind, ifc, ctx, stream, op
        self.rel = None
        if kind == unet_amd.SYNTH_IPV4_HDR20:
            # bare headers carry no usable flow state: ranks take index ranges
            per = n_global // world
            self.first, self.n = first_idx + rank * per, per
            self.nbytes = (self.n * 20 + 15) // 16 * 16
            self.arena = torch.empty(self.nbytes + 16, dtype=torch.uint8, device="cuda")
            ctx.synth_dev(kind, SEED, self.first, self.n, ifc, None, None, self.arena.data_ptr(), self.nbytes, stream)
            self.off = self.len = None
            self.alg_bytes = self.n * 24  # 20 read + 4 written per header
            self.stride16 = 0
        else:
            self.first = first_idx
            if world == 1:
                self.n, rel_ptr = n_global, None
            else:
                d_shard = torch.empty(n_global, dtype=torch.uint8, device="cuda")
                ctx.synth_shard_dev(kind, SEED, first_idx, n_global, ifc, world, d_shard.data_ptr(), stream)
                self.rel = torch.nonzero(d_shard == rank).flatten().to(torch.int32)
                del d_shard
                self.n, rel_ptr = self.rel.numel(), self.rel.data_ptr()
            # lengths from the trace definition, layout = exclusive scan of the 16-byte slot counts
            self.len = torch.empty(self.n, dtype=torch.int32, device="cuda")
            ctx.synth_len_dev(kind, SEED, first_idx, rel_ptr, self.n, self.len.data_ptr(), stream)
            slots = (self.len.to(torch.int64) + 15) // 16
            if stride16:  # fixed-stride arena: every record gets stride16 slots (94 = packed for 1500 bytes)
                assert int(slots.max().item()) <= stride16
                slots = torch.full_like(slots, stride16)
            self.off = (torch.cumsum(slots, 0) - slots).to(torch.int32)
            self.nbytes = int(slots.sum().item()) * 16
            self.arena = torch.zeros(self.nbytes + 16, dtype=torch.uint8, device="cuda")
            ctx.synth_indexed_dev(kind, SEED, first_idx, rel_ptr, self.n, ifc, self.off.data_ptr(), self.len.data_ptr(),
                                  self.arena.data_ptr(), self.nbytes, stream)
            self.alg_bytes = int(self.len.to(torch.int64).sum().item()) + 8 * self.n  # bytes + 4 (length) + 4 (result)
            self.stride16 = stride16
        self.out = torch.zeros(self.n, dtype=torch.int32, device="cuda")
        if op == "csum":
            self.alg_bytes -= 2 * self.n      # 2-byte results
        elif op in ("generate", "fields"):
            self.alg_bytes += self.n          # 4 bytes of checksum fields + 1 status byte instead of a 4-byte result
            self.status = torch.zeros(self.n, dtype=torch.uint8, device="cuda")
        elif op == "tcptrack":
            from unet_amd._ffi import tcp_track_workspace
            self.table_slots = 1 << 22
            self.table = torch.zeros(self.table_slots * 32, dtype=torch.uint8, device="cuda")
            self.seg_state = torch.zeros(self.n, dtype=torch.uint8, device="cuda")
            self.ws_bytes = tcp_track_workspace(self.n, self.table_slots)
            self.ws = torch.zeros(self.ws_bytes, dtype=torch.uint8, device="cuda")
            self.tcp_stats = torch.zeros(2, dtype=torch.int32, device="cuda")
            self.alg_bytes += 5 * self.n  # the verdict read back and one state byte written per record (sort and table traffic not counted)
        elif op == "tximix":
            # the resident IMIX records serve as payloads: a record of L bytes gives an L-28-byte payload, so the
            # built packets have the IMIX sizes again
            from unet_amd._ffi import TX_DESC_DTYPE
            self.rx = unet_amd.Iface("198.51.100.7", "255.255.255.0")
            self.tx_ifc = ifc.replace(mode=0)
            desc = np.zeros(self.n, dtype=TX_DESC_DTYPE)
            rng = np.random.default_rng(SEED + first_idx)
            desc["src"], desc["dst"], desc["proto"] = ifc.unicast, self.rx.unicast, 17
            desc["l4a"] = rng.integers(0, 65536, size=self.n)
            desc["l4b"] = rng.integers(0, 65536, size=self.n)
            desc["ip_id"] = np.arange(self.n) & 0xFFFF
            self.desc = torch.from_numpy(desc.view(np.uint8).reshape(-1)).cuda()
            self.plen = torch.clamp(self.len - 28, min=0)
            oslots = (self.plen.to(torch.int64) + 28 + 15) // 16
            self.tx_flags = 0
            if TX_LAYOUT == "lines":
                oslots, self.tx_flags = (oslots + 7) // 8 * 8, TX_FILL_LINES
            self.ooff = (torch.cumsum(oslots, 0) - oslots).to(torch.int32)
            self.out_bytes = int(oslots.sum().item()) * 16
            self.out_arena = torch.zeros(self.out_bytes + 16, dtype=torch.uint8, device="cuda")
            self.olen = torch.zeros(self.n, dtype=torch.int32, device="cuda")
            self.status = torch.zeros(self.n, dtype=torch.uint8, device="cuda")
            psum = int(self.plen.to(torch.int64).sum().item())
            # payload + packet + descriptor + payload offset and length, output offset, output length, status
            self.alg_bytes = psum + (psum + 28 * self.n) + self.n * (28 + 4 + 4 + 4 + 4 + 1)
        elif op in ("txbuild", "eth1514"):
            # the resident records serve as 1472-byte payloads; every packet goes to one receiver address
            from unet_amd._ffi import TX_DESC_DTYPE
            l2 = op == "eth1514"
            self.rx = unet_amd.Iface("198.51.100.7", "255.255.255.0", mac=bytes.fromhex("02c0ffee0007"), mode=1 if l2 else 0)
            self.tx_ifc = ifc.replace(mode=1 if l2 else 0)
            self.ostride = 95 if l2 else 94  # 1514 / 1500 bytes in 16-byte slots
            self.tx_flags = 0
            if op == "txbuild" and TX_LAYOUT == "lines":
                # every packet starts on a 128-byte line (unet_tx_layout_aligned, align16 = 8) and the kernel may write
                # its last line whole: the layout the library recommends for payloads above about 1 KiB
                self.ostride, self.tx_flags = 96, TX_FILL_LINES
            desc = np.zeros(self.n, dtype=TX_DESC_DTYPE)
            desc["dst_mac"] = np.frombuffer(bytes(self.rx.mac), dtype=np.uint8)
            rng = np.random.default_rng(SEED + first_idx)
            desc["src"], desc["dst"], desc["proto"] = ifc.unicast, self.rx.unicast, 17
            desc["l4a"] = rng.integers(0, 65536, size=self.n)
            desc["l4b"] = rng.integers(0, 65536, size=self.n)
            desc["ip_id"] = np.arange(self.n) & 0xFFFF
            self.desc = torch.from_numpy(desc.view(np.uint8).reshape(-1)).cuda()
            self.plen = torch.full((self.n,), 1472, dtype=torch.int32, device="cuda")
            self.ooff = (torch.arange(self.n, dtype=torch.int64, device="cuda") * self.ostride).to(torch.int32)
            self.out_bytes = self.n * self.ostride * 16
            self.out_arena = torch.zeros(self.out_bytes + 16, dtype=torch.uint8, device="cuda")
            self.olen = torch.zeros(self.n, dtype=torch.int32, device="cuda")
            self.status = torch.zeros(self.n, dtype=torch.uint8, device="cuda")
            # payload + packet + descriptor + payload length, output offset, output length, status
            self.alg_bytes = self.n * (1472 + 1500 + 28 + 4 + 4 + 4 + 1)
            if l2:
                # build the frames once, then damage some: a payload byte (idx % 64 == 5), the destination MAC
                # (idx % 128 == 9), the EtherType (idx % 256 == 77)
                self.build()
                fr = self.out_arena[:self.out_bytes].view(self.n, self.ostride * 16)
                idx = torch.arange(self.n, device="cuda")
                fr[idx % 64 == 5, 700] ^= 0x40
                fr[idx % 128 == 9, 3] ^= 0x01
                fr[idx % 256 == 77, 13] = 0x01
                self.alg_bytes = self.n * (1514 + 4 + 4)  # frame + length + result

    def run(self):
        if self.op == "csum":
            off_ptr = None if self.stride16 else self.off.data_ptr()
            self.ctx.csum_batch_dev(self.arena.data_ptr(), self.nbytes, off_ptr, self.stride16, self.len.data_ptr(), None,
                                    self.n, self.out.data_ptr(), self.stream)
        elif self.op == "generate":
            off_ptr = None if self.stride16 else self.off.data_ptr()
            self.ctx.generate_dev(self.arena.data_ptr(), self.nbytes, off_ptr, self.stride16, self.len.data_ptr(), self.n,
                                  self.out.data_ptr(), self.stream)
        elif self.op == "fields":
            off_ptr = None if self.stride16 else self.off.data_ptr()
            self.ctx.checksum_fields_dev(self.arena.data_ptr(), self.nbytes, off_ptr, self.stride16, self.len.data_ptr(), self.n,
                                         self.out.data_ptr(), self.status.data_ptr(), self.stream)
        elif self.op == "tcptrack":
            self.ctx.verify_dev(self.arena.data_ptr(), self.nbytes, self.off.data_ptr(), 0, self.len.data_ptr(), self.n, self.ifc,
                                self.out.data_ptr(), self.stream)
            self.ctx.tcp_track_ws_dev(self.arena.data_ptr(), self.nbytes, self.off.data_ptr(), 0, self.len.data_ptr(), self.out.data_ptr(), self.n,
                                      self.ifc, self.table.data_ptr(), self.table_slots, self.seg_state.data_ptr(), self.ws.data_ptr(),
                                      self.ws_bytes, self.tcp_stats.data_ptr(), self.stream)
        elif self.op in ("txbuild", "tximix"):
            self.build()
        elif self.op == "eth1514":
            self.ctx.verify_dev(self.out_arena.data_ptr(), self.out_bytes, None, self.ostride, self.olen.data_ptr(), self.n, self.rx,
                                self.out.data_ptr(), self.stream)
        elif self.kind == unet_amd.SYNTH_IPV4_HDR20:
            self.ctx.ipv4_validate_dev(self.arena.data_ptr(), self.n, self.ifc, self.out.data_ptr(), self.stream)
        else:
            off_ptr = None if self.stride16 else self.off.data_ptr()
            self.ctx.verify_dev(self.arena.data_ptr(), self.nbytes, off_ptr, self.stride16, self.len.data_ptr(), self.n,
                                self.ifc, self.out.data_ptr(), self.stream)

    def build(self):
        if self.op == "tximix":
            self.ctx.tx_build_dev(self.arena.data_ptr(), self.nbytes, self.off.data_ptr(), 0, self.plen.data_ptr(), self.desc.data_ptr(),
                                  self.n, self.tx_ifc, self.out_arena.data_ptr(), self.out_bytes, self.ooff.data_ptr(),
                                  self.olen.data_ptr(), self.status.data_ptr(), self.stream, flags=self.tx_flags)
            return
        self.ctx.tx_build_dev(self.arena.data_ptr(), self.nbytes, None, 94, self.plen.data_ptr(), self.desc.data_ptr(), self.n,
                              self.tx_ifc, self.out_arena.data_ptr(), self.out_bytes, self.ooff.data_ptr(), self.olen.data_ptr(),
                              self.status.data_ptr(), self.stream, flags=self.tx_flags)

    def counts(self):
        d_cnt = torch.zeros(unet_amd.V_COUNT, dtype=torch.int64, device="cuda")
        self.ctx.count_verdicts_dev(self.out.data_ptr(), self.n, d_cnt.data_ptr(), self.stream)
        return d_cnt

    def trace_indices(self):
        if self.rel is None:
            return np.arange(self.first, self.first + self.n, dtype=np.int64)
        return self.first + self.rel.cpu().numpy().astype(np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="udp1500", choices=sorted(WORKLOADS))
    ap.add_argument("--ext-tcp", type=int, default=None, help="TCP extension on/off (default: on for imix/trace)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--frames", type=int, default=0, help="override records per GPU per batch")
    ap.add_argument("--stride16", type=int, default=0, help="fixed-stride workloads: record stride in 16-byte slots (default 94 = packed)")
    ap.add_argument("--rotate", type=int, default=0, help="distinct resident batches cycled through (default 6 for hdr20, else 1)")
    ap.add_argument("--total-frames", type=int, default=1 << 30, help="trace workload: records in the whole trace (all GPUs)")
    ap.add_argument("--emulate-world", type=int, default=0, help="single-GPU rehearsal of the sharded path: pretend to be one of W ranks")
    ap.add_argument("--emulate-rank", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--all-on-gpu0", action="store_true", help="rehearsal on a one-GPU box: every rank uses cuda:0")
    ap.add_argument("--tx-layout", default="auto", choices=["auto", "packed", "lines"],
                    help="txbuild / tximix: output packets packed on 16-byte boundaries, or on 128-byte lines with UNET_TX_FILL_LINES "
                         "(auto: lines for txbuild, whose payloads are above 1 KiB; packed for tximix)")
    ap.add_argument("--dry-run", action="store_true", help="start the ranks, rendezvous, print who is there, stop before any GPU call")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group (RCCL with --backend nccl) even with one rank, so that the barrier, the counter "
                         "all-reduce and the size all-gather of the multi-GPU path run on a one-GPU box")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help="with --dry-run: this rank exits with an error before the rendezvous")
    ap.add_argument("--rendezvous-seconds", type=float, default=180.0, help="give up when the ranks have not all met after this long")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    # Native libraries write to file descriptor 1 as they please (RCCL prints a version banner there when it is
    # initialised): from here on descriptor 1 is the error stream, and the one JSON line goes to the real one.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.dry_run:
        import datetime

        import torch.distributed as dist
        if args.dry_run_fail_rank == rank:
            raise SystemExit(f"rank {rank}: asked to fail before the rendezvous (--dry-run-fail-rank)")
        seen = [rank]
        if world > 1 or args.force_dist:
            if "MASTER_ADDR" not in os.environ:
                with socket.socket() as s:
                    s.bind(("127.0.0.1", 0))
                    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s.getsockname()[1]))
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=args.rendezvous_seconds), rank=rank, world_size=world)
            t = torch.zeros(world, dtype=torch.int64)
            t[rank] = rank + 1
            dist.all_reduce(t)
            seen = [int(x) - 1 for x in t.tolist()]
            dist.destroy_process_group()
        if rank == 0:
            emit({"dry_run": True, "n_gpus": world, "ranks": seen, "workload": args.workload})
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the offload has no CPU path")
    if args.all_on_gpu0:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or args.force_dist:
        import datetime

        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:  # --force-dist without a launcher: a rendezvous of one
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s.getsockname()[1]))
        # a rank that died before the rendezvous must not leave the others waiting for ever
        limit = datetime.timedelta(seconds=args.rendezvous_seconds)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=limit, rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, timeout=limit, rank=rank, world_size=world)
    shard_world, shard_rank = (args.emulate_world, args.emulate_rank) if args.emulate_world else (world, rank)

    global TX_LAYOUT
    TX_LAYOUT = args.tx_layout if args.tx_layout != "auto" else ("lines" if args.workload == "txbuild" else "packed")
    kind, n_per_gpu, stride16 = WORKLOADS[args.workload]
    if args.stride16 and stride16:
        stride16 = args.stride16
    if args.frames:
        n_per_gpu = args.frames
    ext_tcp = (args.workload in ("imix", "trace", "odd", "tcptrack")) if args.ext_tcp is None else bool(args.ext_tcp)
    ifc = unet_amd.Iface(ext_tcp=ext_tcp)
    ctx = unet_amd.Offload(local)
    # A dedicated torch stream: its raw handle goes to the C ABI, so the kernels, torch's own ops and the
    # HIP events used for timing (torch.cuda.Event records on the current stream) share one queue.
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream, "need a non-null stream handle (NULL selects the library's private stream)"
    n_global = n_per_gpu * shard_world  # records per global batch

    def sync_all():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload == "trace":
        # ---- config 5: the trace does not fit in HBM, so it is generated, judged and dropped chunk by
        # chunk; only the verify launches are timed (inputs resident when each timed region starts).
        n_chunks = min(args.steps, max(1, args.total_frames // n_global))
        kernel_s, frames, alg_bytes_total = 0.0, 0, 0
        total_cnt = torch.zeros(unet_amd.V_COUNT, dtype=torch.int64, device="cuda")
        sizes = []
        sync_all()
        t_all = time.perf_counter()
        for c in range(n_chunks):
            b = Batch(ctx, stream, kind, ifc, c * n_global, n_global, shard_world, shard_rank, 0)
            if c == 0:
                b.run()  # warm the code path once
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            b.run()
            e1.record()
            torch.cuda.synchronize()
            kernel_s += e0.elapsed_time(e1) * 1e-3
            total_cnt += b.counts()
            frames += b.n
            alg_bytes_total += b.alg_bytes
            sizes.append(b.n)
            del b
        sync_all()
        wall_all = time.perf_counter() - t_all
        kernel_s_max = shard.max_over_ranks(kernel_s, "cuda", dist)
        frames_t = torch.tensor([frames, alg_bytes_total], dtype=torch.int64, device="cuda")
        if dist:
            dist.all_reduce(frames_t)
        frames_all, bytes_all = int(frames_t[0].item()), int(frames_t[1].item())
        cnt = shard.allreduce_counts(total_cnt, dist).cpu().numpy()
        if rank == 0:
            achieved = alg_bytes_total / kernel_s / 1e9
            line = {
                "metric": "Gpkt/s and achieved HBM GB/s, batched IPv4+TCP/UDP checksum+parse, 1500B",
                "value": frames_all / kernel_s_max / 1e9, "unit": "Gpkt/s", "n_gpus": world, "steps": n_chunks, "warmup": 0,
                "ms_per_step": kernel_s_max * 1e3 / n_chunks, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "u32", "data": "synthetic",
                "config": {"workload": DESCRIPTION["trace"], "records_judged": frames_all, "chunks": n_chunks,
                           "ext_tcp": ext_tcp, "shard_world": shard_world, "shard_rank": shard_rank,
                           "rank0_shard_sizes_min_max": [min(sizes), max(sizes)],
                           "rank0_shard_skew_vs_even": max(sizes) / n_per_gpu,
                           "timing": "sum of HIP-event times around the verify launches (max over ranks); generation excluded",
                           "end_to_end_Gpkt_s_incl_generation": frames_all / wall_all / 1e9},
                "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "traffic_source": "no PMC pass for the chunked trace",
                             "kernels_sha256": kernels_sha256(), "kernel": "k_verify<MODE_L3>",
                             "kernel_ms": kernel_s * 1e3 / n_chunks, "algorithmic_bytes_per_launch": alg_bytes_total // n_chunks},
                "hbm_GBps_all_gpus": bytes_all / kernel_s_max / 1e9,
                "verdict_counts": {Verdict(i).name: int(c) for i, c in enumerate(cnt) if c},
            }
        ctx.close()
        if dist:
            dist.destroy_process_group()
        if rank == 0:
            # the CPU path beside every scaling point: rank 0 times it once the GPUs are idle
            if args.cpu_seconds > 0:
                line["cpu_baseline"] = cpu_baseline(kind, ifc, args.cpu_seconds)
            emit(line)
        return

    # ---- resident batches.  A workload whose arena is not far larger than the 256 MiB Infinity Cache is
    # rotated over several distinct batches (consecutive index ranges of the same trace) so that no
    # launch finds its input in cache.
    rotate = args.rotate if args.rotate else (6 if kind == unet_amd.SYNTH_IPV4_HDR20 else 1)
    op = args.workload if args.workload in ("csum", "generate", "fields", "txbuild", "eth1514", "tximix", "tcptrack") else "verify"
    batches = [Batch(ctx, stream, kind, ifc, r * n_global, n_global, shard_world, shard_rank, stride16, op) for r in range(rotate)]
    b0 = batches[0]
    kernel = {"tcptrack": "k_verify<MODE_L3> + k_tcp_keys + DeviceRadixSort + k_tcp_apply + k_tcp_apply_long", "csum": "k_csum", "generate": "k_generate", "fields": "k_generate<read only>", "txbuild": "k_tx_build_wave<MODE_L3>", "tximix": "k_tx_build<MODE_L3>", "eth1514": "k_verify<MODE_L2>"}.get(op, "k_ipv4_validate_rows" if kind == unet_amd.SYNTH_IPV4_HDR20 else "k_verify<MODE_L3>")
    torch.cuda.synchronize()

    # ---- timed region
    k = 0
    for _ in range(args.warmup):
        batches[k % rotate].run()
        k += 1
    sync_all()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        batches[k % rotate].run()
        k += 1
    ev1.record()
    sync_all()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream
    wall = shard.max_over_ranks(wall, "cuda", dist)

    # ---- epilogue: per-verdict counters, summed across GPUs (the only cross-GPU traffic)
    if op == "txbuild":
        # judge the built packets as the receiver would: every one must be accepted
        ctx.verify_dev(b0.out_arena.data_ptr(), b0.out_bytes, None, b0.ostride, b0.olen.data_ptr(), b0.n, b0.rx, b0.out.data_ptr(), stream)
    if op == "tximix":
        ctx.verify_dev(b0.out_arena.data_ptr(), b0.out_bytes, b0.ooff.data_ptr(), 0, b0.olen.data_ptr(), b0.n, b0.rx, b0.out.data_ptr(), stream)
    if op in ("csum", "fields"):
        b0.out.zero_()  # checksum values are not verdicts: nothing to count
    d_cnt = b0.counts()
    torch.cuda.synchronize()
    local_cnt = d_cnt.cpu().numpy().copy()
    cnt = shard.allreduce_counts(d_cnt, dist).cpu().numpy()
    sizes_t = torch.tensor([sum(b.n for b in batches), sum(b.alg_bytes for b in batches)], dtype=torch.int64, device="cuda")
    all_sizes, all_bytes = [int(sizes_t[0].item())], [int(sizes_t[1].item())]
    if dist:
        gathered = [torch.zeros_like(sizes_t) for _ in range(world)]
        dist.all_gather(gathered, sizes_t)
        all_sizes = [int(g[0].item()) for g in gathered]
        all_bytes = [int(g[1].item()) for g in gathered]
    frames_per_step_all = sum(all_sizes) / rotate  # records judged per step over all ranks
    parity = None
    if op == "generate":
        # the fill is idempotent: the packets must still verify exactly as the trace rules say
        b0.op = "verify"
        b0.run()
        d_cnt = b0.counts()
        torch.cuda.synchronize()
        local_cnt = d_cnt.cpu().numpy().copy()
        cnt = shard.allreduce_counts(d_cnt, dist).cpu().numpy()
        parity = bool(local_cnt[Verdict.ACCEPT_UDP] + local_cnt[Verdict.DROP_UDP_LEN] == b0.n)
    if op == "tximix":
        parity = bool(local_cnt[Verdict.ACCEPT_UDP] == b0.n and int(b0.status.max().item()) == 0
                      and bool((b0.olen == b0.plen + 28).all().item()))
    if op == "txbuild":
        parity = bool(local_cnt[Verdict.ACCEPT_UDP] == b0.n and int(b0.status.max().item()) == 0
                      and int(b0.olen.min().item()) == 1500 == int(b0.olen.max().item()))
    if op == "eth1514":
        i = np.arange(b0.n)
        bad_type = i % 256 == 77
        bad_mac = (i % 128 == 9) & ~bad_type
        bad_sum = (i % 64 == 5) & ~bad_type & ~bad_mac
        parity = bool(local_cnt[Verdict.DROP_L2_ETHERTYPE] == bad_type.sum() and local_cnt[Verdict.DROP_L2_NOT_FOR_ME] == bad_mac.sum()
                      and local_cnt[Verdict.DROP_UDP_CSUM] == bad_sum.sum()
                      and local_cnt[Verdict.ACCEPT_UDP] == b0.n - bad_type.sum() - bad_mac.sum() - bad_sum.sum())
    tcp_extra = None
    if op == "tcptrack":
        st = b0.seg_state.cpu().numpy()
        tcp_extra = {"segment_states": {k: int((st == v).sum()) for k, v in (("not_tcp", 0), ("new", 1), ("in_order", 2), ("out_of_order", 3), ("after_close", 4))},
                     "longest_run_on_one_slot": int(b0.tcp_stats[0].item()), "runs_over_64": int(b0.tcp_stats[1].item()) // (args.steps + args.warmup),
                     "table_slots": b0.table_slots, "workspace_bytes": b0.ws_bytes,
                     "note": "the table carries over from step to step, as it would from batch to batch; tracking is an extension with no reference counterpart"}
        parity = bool(tcp_extra["segment_states"]["not_tcp"] == b0.n - int(local_cnt[Verdict.ACCEPT_TCP]))
    if args.workload == "udp1500":
        # per-index fault rules of the trace: idx % 256 == 33 -> length mismatch; idx % 64 == 7 or
        # idx % 256 == 97 -> checksum failure (a zeroed field can verify only if the true value was 0)
        idx = b0.trace_indices()
        n_len = int((idx % 256 == 33).sum())
        n_ck = int(((idx % 64 == 7) | (idx % 256 == 97)).sum())
        parity = bool(local_cnt[Verdict.DROP_UDP_LEN] == n_len and n_ck - 16 <= local_cnt[Verdict.DROP_UDP_CSUM] <= n_ck
                      and local_cnt[Verdict.ACCEPT_UDP] == b0.n - n_len - local_cnt[Verdict.DROP_UDP_CSUM])

    # ---- attainable-bandwidth yardstick: read-only streaming kernel over the same buffer
    d_sink = torch.zeros(4096, dtype=torch.int32, device="cuda")
    if op == "eth1514":
        b0.arena, b0.nbytes = b0.out_arena, b0.out_bytes  # the frames are what this workload reads
    # (shapes of the same read -- slabs swept by resident workgroups, short-lived waves that own 24 KiB each (with three or
    # with eight workgroups to a CU), and
    # the record kernels' own loads over records of the batch's mean size with only a sum taken; which one is fastest
    # differs from box to box: the yardstick is the best one)
    # The epilogue above leaves the GPU idle for milliseconds and its clocks drop: each shape is warmed up for 30 ms
    # first, and -- this being a ceiling -- the fastest of three groups of eight launches counts.
    shapes = {}
    rec_bytes = max(16, min(65520, b0.nbytes // max(1, b0.n)))  # the batch's mean record size, for the third shape

    def read_records(buf, nbytes, sink, s):
        ctx.stream_read_records_dev(buf, nbytes, rec_bytes, sink, s)

    yardsticks = [("slabs", ctx.stream_read_dev), ("pieces", ctx.stream_read_pieces_dev), ("pieces_full_occupancy", ctx.stream_read_pieces_full_dev),
                  ("records", read_records)]
    if op not in ("txbuild", "tximix", "eth1514") and b0.kind != unet_amd.SYNTH_IPV4_HDR20 and getattr(b0, "len", None) is not None:
        # and the batch's own records through the checksum-only kernel (k_csum: the record kernels' requests, head and tail
        # rows and all, a sum and two bytes written per record, no parse): what the access pattern itself allows
        d_sums = torch.empty(b0.n, dtype=torch.int16, device="cuda")

        def read_record_sums(buf, nbytes, sink, s):
            ctx.csum_batch_dev(buf, nbytes, None if b0.stride16 else b0.off.data_ptr(), b0.stride16, b0.len.data_ptr(), None, b0.n, d_sums.data_ptr(), s)

        yardsticks.append(("record_sums", read_record_sums))
    for name, fn in yardsticks:
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.03:
            fn(b0.arena.data_ptr(), b0.nbytes, d_sink.data_ptr(), stream)
            torch.cuda.synchronize()
        best = 0.0
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                fn(b0.arena.data_ptr(), b0.nbytes, d_sink.data_ptr(), stream)
            e1.record()
            torch.cuda.synchronize()
            best = max(best, b0.nbytes * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        shapes[name] = best
    stream_gbps = max(shapes.values())
    copy_gbps = None
    if op in ("txbuild", "tximix"):
        # read+write yardstick: a plain device-to-device copy of the payload arena into the packet arena
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cb = min(b0.nbytes, b0.out_bytes)
        b0.out_arena[:cb].copy_(b0.arena[:cb])
        e0.record()
        for _ in range(5):
            b0.out_arena[:cb].copy_(b0.arena[:cb])
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 2 * cb * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ctx.stream_copy_dev(b0.arena.data_ptr(), b0.out_arena.data_ptr(), cb, stream)
        e0.record()
        for _ in range(5):
            ctx.stream_copy_dev(b0.arena.data_ptr(), b0.out_arena.data_ptr(), cb, stream)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = {"torch_copy": copy_gbps, "k_stream_copy": 2 * cb * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9}

    if rank == 0:
        alg_bytes = sum(b.alg_bytes for b in batches) / rotate
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes of this same command (tools/pmc_traffic.py)
        traffic, traffic_source = committed_traffic(args.workload, b0.n, shard_world == 1)
        if shard_world == 1:
            sharding = "single GPU, contiguous trace indices"
        elif kind == unet_amd.SYNTH_IPV4_HDR20:
            sharding = "index ranges (bare headers)"
        else:
            sharding = f"5-tuple flow hash % {shard_world} over a global batch of {n_global} records"
        line = {
            "metric": "Gpkt/s and achieved HBM GB/s, batched IPv4+TCP/UDP checksum+parse, 1500B",
            "value": frames_per_step_all * args.steps / wall / 1e9,
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
            "config": {"workload": DESCRIPTION[args.workload], "frames_per_gpu": n_per_gpu, "rank0_frames": b0.n,
                       "arena_bytes_rank0": b0.nbytes, "ext_tcp": ext_tcp, "batches_rotated": rotate,
                       "sharding": sharding, "shard_sizes": [s // rotate for s in all_sizes],
                       "shard_skew_max_over_mean": max(all_sizes) / (sum(all_sizes) / len(all_sizes)),
                       "collectives": f"none on the data path; {unet_amd.V_COUNT} u64 verdict counters all-reduced (RCCL) after the timed region",
                       "inputs": "generated in HBM by unet_synth_indexed_dev before the timed region",
                       **({"tx_output_layout": TX_LAYOUT, "tx_output_arena_bytes": b0.out_bytes} if op in ("txbuild", "tximix") else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernels_sha256": kernels_sha256(), "kernel": kernel,
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "stream_read_GBps": stream_gbps, "stream_read_shapes_GBps": shapes,
                         # like with like: the rate at which the kernel gets through the bytes of this very buffer (the
                         # tables it reads as well and the results it writes are left out) over the pure read's rate
                         "arena_GBps": b0.nbytes / (kernel_ms * 1e-3) / 1e9 if op not in ("txbuild", "tximix") else None,
                         "frac_of_stream_read": (b0.nbytes / (kernel_ms * 1e-3) / 1e9) / stream_gbps if op not in ("txbuild", "tximix") else None,
                         **({"d2d_copy_GBps": copy_gbps} if copy_gbps else {})},
            "hbm_GBps_all_gpus": sum(all_bytes) / rotate * args.steps / wall / 1e9,
            "verdict_counts": {Verdict(i).name: int(c) for i, c in enumerate(cnt) if c},
            "parity_check": parity,
            **({"tcp_tracking": tcp_extra} if tcp_extra else {}),
        }
    ctx.close()
    if dist:
        dist.destroy_process_group()
    if rank == 0:
        # the CPU path beside every scaling point: rank 0 times it once the GPUs are idle
        if args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(kind, ifc, args.cpu_seconds)
        emit(line)


if __name__ == "__main__":
    main()
