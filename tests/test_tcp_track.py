"""Per-flow TCP tracking (extension, parity unpinned: the reference has no TCP).  The GPU form -- extract, stable
sort by table slot, one lane per slot walking its run -- must equal the sequential CPU definition bit for bit:
every per-segment state and every byte of the flow table, on hand-built conversations with known outcomes, on
interleaved random flows with losses, retransmissions, resets and slot collisions, and across consecutive
batches that share one table."""
import struct

import numpy as np
import pytest

import unet_amd
from unet_amd import _ffi
from helpers import py_ck
from oracle import oracle_ffi as O

pytestmark = pytest.mark.gpu

ME = 0xC0000202
FIN, SYN, RST, PSH, ACK = 1, 2, 4, 8, 16


def tcp_packet(src, sport, dport, seq, flags, payload=b"", opts=b"", l2=None):
    doff = 5 + len(opts) // 4
    seg = bytearray(struct.pack(">HHIIBBHHH", sport, dport, seq & 0xFFFFFFFF, 0, doff << 4, flags, 4096, 0, 0) + opts + payload)
    pseudo = struct.pack(">IIBBH", src, ME, 0, 6, len(seg))
    struct.pack_into(">H", seg, 16, py_ck(bytes(seg), ~py_ck(pseudo, 0) & 0xFFFF))
    pkt = O.create_ip_header(1, 6, src, ME, len(seg)) + bytes(seg)
    return (l2 + pkt) if l2 else pkt


def run_gpu(offload, frames, ifc, table):
    import torch
    arena, off16, lens = unet_amd.pack_arena(frames)
    n = len(frames)
    d_arena = torch.from_numpy(arena).cuda()
    d_off = torch.from_numpy(off16.view(np.int32)).cuda()
    d_len = torch.from_numpy(lens.view(np.int32)).cuda()
    d_res = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_tab = torch.from_numpy(table.view(np.uint8).reshape(-1).copy()).cuda()
    d_st = torch.full((n,), 0xEE, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    offload.verify_dev(d_arena.data_ptr(), arena.nbytes, d_off.data_ptr(), 0, d_len.data_ptr(), n, ifc, d_res.data_ptr(), s or None)
    offload.tcp_track_dev(d_arena.data_ptr(), arena.nbytes, d_off.data_ptr(), 0, d_len.data_ptr(), d_res.data_ptr(), n, ifc, d_tab.data_ptr(),
                          len(table), d_st.data_ptr(), s or None)
    torch.cuda.synchronize()
    res = d_res.cpu().numpy().view(unet_amd.RESULT_DTYPE)
    return res, d_st.cpu().numpy(), d_tab.cpu().numpy().view(_ffi.TCP_FLOW_DTYPE), (arena, off16, lens)


def check_against_oracle(offload, frames, ifc, table):
    res, st, tab, (arena, off16, lens) = run_gpu(offload, frames, ifc, table)
    exp_res = O.recv_batch(arena, off16, lens, ifc)
    assert np.array_equal(res, exp_res)
    exp_tab = table.copy()
    exp_st = O.tcp_track(arena, off16, lens, exp_res, ifc, exp_tab)
    assert np.array_equal(st, exp_st), np.nonzero(st != exp_st)[0][:10]
    assert np.array_equal(tab.view(np.uint8), exp_tab.view(np.uint8))
    return res, st, tab


def test_one_conversation_with_known_outcomes(offload):
    ifc = unet_amd.Iface(ext_tcp=True)
    a = 0x0A000001
    frames = [
        tcp_packet(a, 1000, 80, 100, SYN),                       # 0 NEW, consumes 1
        tcp_packet(a, 1000, 80, 101, ACK, b"x" * 50),            # 1 IN_ORDER -> 151
        tcp_packet(a, 1000, 80, 251, ACK, b"y" * 10),            # 2 OUT_OF_ORDER (100 bytes missing), expectation -> 261
        tcp_packet(a, 1000, 80, 151, ACK, b"z" * 100),           # 3 OUT_OF_ORDER (old data), expectation stays 261
        tcp_packet(a, 1000, 80, 261, ACK | FIN, b"end"),         # 4 IN_ORDER, FIN -> 265
        struct.pack(">BBHHHBBHII", 0x45, 0, 20, 0, 0, 64, 47, 0, a, ME),  # 5 not TCP (protocol 47)
        tcp_packet(a, 1000, 80, 265, RST),                       # 6 IN_ORDER, reset
        tcp_packet(a, 1000, 80, 265, ACK, b"late"),              # 7 AFTER_CLOSE
        tcp_packet(a, 1000, 80, 9000, SYN),                      # 8 NEW (SYN reopens a reset flow)
        tcp_packet(a, 1000, 80, 9001, ACK, b"q", opts=bytes([8, 0, 0, 0])),  # 9 DROP_TCP_OPTION: not tracked
    ]
    table = np.zeros(1 << 10, dtype=_ffi.TCP_FLOW_DTYPE)
    res, st, tab = check_against_oracle(offload, frames, ifc, table)
    assert st.tolist() == [1, 2, 3, 3, 2, 0, 2, 4, 1, 0]
    assert int(res["verdict"][9]) == unet_amd.Verdict.DROP_TCP_OPTION
    slot = O.tcp_slot(a, ME, 1000, 80, 1 << 10)
    assert slot == _ffi.load_library().unet_tcp_slot(a, ME, 1000, 80, 1 << 10)
    e = tab[slot]
    assert (int(e["state"]), int(e["next_seq"]), int(e["segments"])) == (_ffi.TCP_SYN, 9001, 1)
    assert np.count_nonzero(tab["state"]) == 1


@pytest.mark.parametrize("mode,log2_slots", [(0, 16), (1, 16), (0, 4), (0, 0)])
def test_interleaved_flows_equal_the_sequential_definition(offload, mode, log2_slots):
    """300 flows interleaved at random, with losses, retransmissions, resets and (with 16 or 1 slots) constant
    take-overs of table slots; three consecutive batches share the table."""
    rng = np.random.default_rng(100 + mode + log2_slots)
    mac = bytes.fromhex("02aabbccddee")
    ifc = unet_amd.Iface(mac=mac, mode=mode, ext_tcp=True)
    l2 = (mac + b"\x02\x00\x00\x00\x00\x09\x08\x00") if mode else None
    flows = [dict(src=int(rng.integers(1, 2**32 - 1)), sport=int(rng.integers(1, 65536)), dport=int(rng.integers(1, 65536)),
                  seq=int(rng.integers(0, 2**32)), started=False) for _ in range(300)]
    flows[0]["seq"] = 2**32 - 300  # sequence numbers wrap
    table = np.zeros(1 << log2_slots, dtype=_ffi.TCP_FLOW_DTYPE)
    seen = set()
    for batch in range(3):
        frames = []
        for _ in range(6000):
            f = flows[int(rng.integers(0, len(flows)) ** 1) if rng.random() < 0.7 else int(rng.integers(0, 8))]  # a few hot flows
            r = rng.random()
            payload = rng.integers(0, 256, size=int(rng.integers(0, 600)), dtype=np.uint8).tobytes()
            if not f["started"]:
                frames.append(tcp_packet(f["src"], f["sport"], f["dport"], f["seq"], SYN, l2=l2))
                f["seq"] += 1
                f["started"] = True
            elif r < 0.05:   # lost segment: the sequence space moves on without a packet
                f["seq"] += len(payload) + 1
            elif r < 0.10:   # retransmission of old data
                frames.append(tcp_packet(f["src"], f["sport"], f["dport"], f["seq"] - 700, ACK, payload, l2=l2))
            elif r < 0.12:
                frames.append(tcp_packet(f["src"], f["sport"], f["dport"], f["seq"], RST | ACK, l2=l2))
            elif r < 0.14:
                frames.append(tcp_packet(f["src"], f["sport"], f["dport"], f["seq"], FIN | ACK, payload, l2=l2))
                f["seq"] += len(payload) + 1
            elif r < 0.16:   # a damaged segment: dropped by the checksum, never reaches the table
                p = bytearray(tcp_packet(f["src"], f["sport"], f["dport"], f["seq"], ACK, payload + b"!", l2=l2))
                p[-1] ^= 0x10
                frames.append(bytes(p))
            elif r < 0.18:   # restart
                f["seq"] = int(rng.integers(0, 2**32))
                frames.append(tcp_packet(f["src"], f["sport"], f["dport"], f["seq"], SYN, l2=l2))
                f["seq"] += 1
            else:
                frames.append(tcp_packet(f["src"], f["sport"], f["dport"], f["seq"], ACK | PSH, payload, l2=l2))
                f["seq"] += len(payload)
        res, st, table = check_against_oracle(offload, frames, ifc, table)
        table = table.copy()
        seen |= set(st.tolist())
    assert seen == {0, 1, 2, 3, 4}


def test_trace_chunk_tracks_like_the_oracle(offload):
    """A chunk of the flow-table trace (2^20 flows of skewed popularity, random sequence numbers): 200,000 records,
    L3 mode, a table far smaller than the flow count."""
    import torch
    ifc = unet_amd.Iface(ext_tcp=True)
    n = 200_000
    arena, off16, lens = unet_amd.synth_host(unet_amd.SYNTH_FLOWS, 0x756E6574, 0, n, ifc)
    table = np.zeros(1 << 14, dtype=_ffi.TCP_FLOW_DTYPE)
    d_arena = torch.from_numpy(arena).cuda()
    d_off = torch.from_numpy(off16.view(np.int32)).cuda()
    d_len = torch.from_numpy(lens.view(np.int32)).cuda()
    d_res = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_tab = torch.from_numpy(table.view(np.uint8).reshape(-1).copy()).cuda()
    d_st = torch.zeros(n, dtype=torch.uint8, device="cuda")
    offload.verify_dev(d_arena.data_ptr(), arena.nbytes, d_off.data_ptr(), 0, d_len.data_ptr(), n, ifc, d_res.data_ptr())
    offload.tcp_track_dev(d_arena.data_ptr(), arena.nbytes, d_off.data_ptr(), 0, d_len.data_ptr(), d_res.data_ptr(), n, ifc, d_tab.data_ptr(),
                          len(table), d_st.data_ptr())
    offload.sync()
    res = O.recv_batch(arena, off16, lens, ifc)
    exp_st = O.tcp_track(arena, off16, lens, res, ifc, table)
    assert np.array_equal(d_st.cpu().numpy(), exp_st)
    assert np.array_equal(d_tab.cpu().numpy(), table.view(np.uint8).reshape(-1))
    assert (exp_st != 0).sum() > n // 2


def test_callers_workspace_long_runs_and_stats(offload):
    """unet_tcp_track_ws_dev: the workspace is the caller's (nothing is allocated by the call), one hot flow puts
    3,000 segments onto one table slot -- walked by a whole wave out of LDS -- between 4,000 segments of 500 other
    flows, and the stats say how the batch fell onto the table.  Two calls in a row reuse workspace and table."""
    import torch
    rng = np.random.default_rng(77)
    ifc = unet_amd.Iface(ext_tcp=True)
    hot = dict(src=0x0A010203, sport=4444, dport=443, seq=1000)
    others = [dict(src=int(rng.integers(1, 2**32 - 1)), sport=int(rng.integers(1, 65536)), dport=80, seq=int(rng.integers(0, 2**32))) for _ in range(500)]
    slots = 1 << 12
    table = np.zeros(slots, dtype=_ffi.TCP_FLOW_DTYPE)
    d_tab = torch.from_numpy(table.view(np.uint8).reshape(-1).copy()).cuda()
    n_max = 8000
    need = _ffi.tcp_track_workspace(n_max, slots)
    assert 30 * n_max < need < 80 * n_max + (1 << 20)
    d_ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    assert d_ws.data_ptr() % 256 == 0
    for call in range(2):
        frames, who = [], []
        for k in range(7000 if call == 0 else 3500):
            f = hot if k % 7 < 3 else others[int(rng.integers(0, len(others)))]
            payload = rng.integers(0, 256, size=int(rng.integers(0, 200)), dtype=np.uint8).tobytes()
            seq = f["seq"] + (400 if rng.random() < 0.03 else 0)   # now and then a gap
            frames.append(tcp_packet(f["src"], f["sport"], f["dport"], seq, ACK | PSH, payload))
            f["seq"] = seq + len(payload)
            who.append(f is hot)
        n = len(frames)
        arena, off16, lens = unet_amd.pack_arena(frames)
        d_arena = torch.from_numpy(arena).cuda()
        d_off = torch.from_numpy(off16.view(np.int32)).cuda()
        d_len = torch.from_numpy(lens.view(np.int32)).cuda()
        d_res = torch.zeros(n, dtype=torch.int32, device="cuda")
        d_st = torch.full((n,), 0xEE, dtype=torch.uint8, device="cuda")
        d_stats = torch.zeros(2, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        offload.verify_dev(d_arena.data_ptr(), arena.nbytes, d_off.data_ptr(), 0, d_len.data_ptr(), n, ifc, d_res.data_ptr())
        offload.tcp_track_ws_dev(d_arena.data_ptr(), arena.nbytes, d_off.data_ptr(), 0, d_len.data_ptr(), d_res.data_ptr(), n, ifc, d_tab.data_ptr(),
                                 slots, d_st.data_ptr(), d_ws.data_ptr(), need, d_stats.data_ptr())
        offload.sync()
        exp_res = O.recv_batch(arena, off16, lens, ifc)
        assert np.array_equal(d_res.cpu().numpy().view(unet_amd.RESULT_DTYPE), exp_res)
        exp_st = O.tcp_track(arena, off16, lens, exp_res, ifc, table)     # the oracle advances `table` in place
        assert np.array_equal(d_st.cpu().numpy(), exp_st)
        assert np.array_equal(d_tab.cpu().numpy(), table.view(np.uint8).reshape(-1))
        # every record here is an accepted TCP segment: the hot flow's slot sees all of its segments (and any other
        # flow that shares the slot), no other slot comes near
        longest, long_runs = (int(x) for x in d_stats.cpu().numpy())
        assert sum(who) <= longest <= sum(who) + 200 and 1 <= long_runs <= 3
    # a workspace that is too small or misaligned is refused before anything is queued
    with pytest.raises(_ffi.UnetError) as e:
        offload.tcp_track_ws_dev(d_arena.data_ptr(), arena.nbytes, d_off.data_ptr(), 0, d_len.data_ptr(), d_res.data_ptr(), n, ifc, d_tab.data_ptr(),
                                 slots, d_st.data_ptr(), d_ws.data_ptr(), 1024, None)
    assert e.value.code == -5 and "unet_tcp_track_workspace" in str(e.value)
    with pytest.raises(_ffi.UnetError) as e:
        offload.tcp_track_ws_dev(d_arena.data_ptr(), arena.nbytes, d_off.data_ptr(), 0, d_len.data_ptr(), d_res.data_ptr(), n, ifc, d_tab.data_ptr(),
                                 slots, d_st.data_ptr(), d_ws.data_ptr() + 16, need - 16, None)
    assert e.value.code == -1


def test_bad_arguments(offload):
    ifc = unet_amd.Iface(ext_tcp=True)
    with pytest.raises(_ffi.UnetError) as e:
        offload.tcp_track_dev(16, 16, None, 1, 16, 16, 1, ifc, 16, 3, 16)
    assert e.value.code == -1 and "power of two" in str(e.value)
