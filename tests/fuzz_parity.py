"""Randomised parity soak: batches with random length mixes, layouts, modes and contents through unet_verify (and its
static and asynchronous forms), unet_ipv4_validate, unet_csum_batch, unet_generate and unet_tx_build, every result against the CPU oracle.  Not collected by pytest (it runs for as
long as it is told to); tests/test_fuzz_short.py runs a few rounds of it.  dev_round sends a batch through the device-pointer entry
points, where the library chooses records per wave from the arena size, the stride or a caller's hint instead of from the lengths;
tx_dev_round does the same for unet_tx_build_dev, into packed and into line-aligned output layouts, with and without
UNET_TX_FILL_LINES and a (true or wrong) unet_hint_max_len.  The transmit rounds go through a rotation of contexts opened
with the library's kernel knobs set (tile kernel only, one packet per wave or per group of waves, fixed packets per wave),
so that every route builds every kind of batch, not only the ones the library would choose it for.

usage: python tests/fuzz_parity.py [seconds] [first seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd  # noqa: E402
from oracle import oracle_ffi as O  # noqa: E402

LENGTH_CLASSES = [
    lambda r, n: r.integers(0, 65, size=n),
    lambda r, n: r.integers(20, 129, size=n),
    lambda r, n: r.integers(65, 113, size=n),
    lambda r, n: r.integers(1, 1025, size=n),
    lambda r, n: r.integers(900, 1100, size=n),
    lambda r, n: r.integers(1, 3000, size=n),
    lambda r, n: r.choice([64, 576, 1500, 4096, 9000], p=[.56, .32, .08, .02, .02], size=n),
    lambda r, n: r.choice([65, 577, 1499, 4097, 8999], p=[.56, .32, .08, .02, .02], size=n),
    lambda r, n: np.where(r.random(n) < 0.01, 65535, r.integers(40, 400, size=n)),
    lambda r, n: np.full(n, int(r.integers(1, 2100))),
    lambda r, n: np.full(n, int(r.choice([64, 80, 96, 112, 128, 256, 512, 1024, 1040, 1504]))),
    # batches whose mean makes the library take few records per wave, with short records in them
    lambda r, n: np.where(r.random(n) < 0.1, r.choice([3000, 9000, 30000, 65535], size=n), r.integers(20, 120, size=n)),
    lambda r, n: np.where(np.arange(n) % int(r.integers(17, 70)) == 0, int(r.choice([20000, 40000, 65535])), r.integers(1, 113, size=n)),
    lambda r, n: r.choice([1100, 1300, 1536, 1537, 2048, 2049, 4096, 8192, 8193, 16384, 49151, 49152], size=n),
    # around what a lane, a lane group and a wave's LDS staging area take on the transmit side
    lambda r, n: r.choice([18, 20, 36, 52, 53, 84, 85, 100, 104, 105, 244, 320, 321, 548], size=n),
    lambda r, n: r.integers(int(r.choice([900, 960, 976])), 1010, size=n),
    lambda r, n: np.where(r.random(n) < 0.05, r.integers(981, 1100, size=n), r.integers(0, 200, size=n)),
    lambda r, n: r.choice([36, 548, 1472, 4068, 8972], p=[.56, .32, .08, .02, .02], size=n),
]


def make_frames(rng, n, ifc):
    """Half valid traffic from the synthetic generator (cut or padded to the drawn length now and then), half noise."""
    lens = np.asarray(LENGTH_CLASSES[int(rng.integers(len(LENGTH_CLASSES)))](rng, n), dtype=np.int64)
    kind = unet_amd.SYNTH_ODD if rng.random() < 0.5 else unet_amd.SYNTH_IMIX
    a, o, l = unet_amd.synth_host(kind, int(rng.integers(1, 2**31)), int(rng.integers(0, 2**20)), n, ifc)
    frames = []
    for i in range(n):
        if rng.random() < 0.5:
            f = a[int(o[i]) * 16:int(o[i]) * 16 + int(l[i])].tobytes()
            if rng.random() < 0.2:
                f = f[:int(lens[i])]
        else:
            f = rng.integers(0, 256, size=int(lens[i]), dtype=np.uint8).tobytes()
        frames.append(f)
    return frames


def layout(rng, frames):
    lens = np.array([len(f) for f in frames], dtype=np.uint32)
    n = len(frames)
    slots = (lens.astype(np.int64) + 15) // 16
    how = int(rng.integers(6))
    stride16 = 0
    packed = np.concatenate([[0], np.cumsum(slots)[:-1]])
    if how == 0:
        off = packed
    elif how == 1:
        off = packed + int(rng.integers(1, 9))  # the first tile does not start on a 128-byte line
    elif how == 2:
        stride16 = int(slots.max()) + int(rng.integers(0, 2))
        off = np.arange(n, dtype=np.int64) * stride16
    elif how == 3:
        off = packed + np.cumsum(rng.random(n) < 0.02)  # a gap now and then
    elif how == 4:
        perm = rng.permutation(n)
        off = np.zeros(n, dtype=np.int64)
        at = 0
        for k in perm:
            off[k] = at
            at += int(slots[k])
    else:
        off = (int(packed[-1] + slots[-1]) - packed - slots)  # descending
    size = int((off + slots).max()) * 16
    arena = rng.integers(0, 256, size=max(size, 16), dtype=np.uint8)
    for f, o in zip(frames, off):
        arena[int(o) * 16:int(o) * 16 + len(f)] = np.frombuffer(f, dtype=np.uint8)
    return arena, (None if stride16 else off.astype(np.uint32)), stride16, lens, how


def one_round(ctx, seed):
    rng = np.random.default_rng(seed)
    mode = unet_amd.MODE_L2 if rng.random() < 0.3 else unet_amd.MODE_L3
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode, ext_tcp=bool(rng.random() < 0.7))
    n = int(rng.choice([1, 31, 32, 33, 63, 64, 65, 500, 2000, 5000]))
    frames = make_frames(rng, n, ifc)
    arena, off16, stride16, lens, how = layout(rng, frames)
    tag = (seed, n, mode, how, int(lens.max()), round(float(lens.mean()), 1))
    got = ctx.verify(arena, off16, lens, ifc, stride16=stride16)
    exp = O.recv_batch(arena, off16, lens, ifc, stride16=stride16)
    bad = np.nonzero(got != exp)[0]
    assert bad.size == 0, ("verify", tag, bad[:5], got[bad[:5]], exp[bad[:5]], lens[bad[:5]])
    # the other two host-pointer forms of the same call
    if n <= 65536 and arena.nbytes <= (8 << 20):
        assert np.array_equal(ctx.verify_static(arena, off16, lens, ifc, stride16=stride16), exp), ("verify_static", tag)
    t, out = ctx.verify_async(arena, off16, lens, ifc, stride16=stride16)
    t2, out2 = ctx.verify_async(arena, off16, lens, ifc, stride16=stride16)
    ctx.wait(t)
    ctx.wait(t2)
    assert np.array_equal(out, exp) and np.array_equal(out2, exp), ("verify_async", tag)
    # bare 20-byte headers: the generator's (valid and faulty) with a few random ones among them
    nh = int(rng.choice([1, 3, 255, 256, 257, 1000, 5000]))
    hdrs, _, _ = unet_amd.synth_host(unet_amd.SYNTH_IPV4_HDR20, int(rng.integers(1, 2**31)), int(rng.integers(0, 2**20)), nh, ifc)
    hdrs = hdrs[:nh * 20].copy()
    noisy = rng.random(nh) < 0.1
    hdrs.reshape(nh, 20)[noisy] = rng.integers(0, 256, size=(int(noisy.sum()), 20), dtype=np.uint8)
    assert np.array_equal(ctx.ipv4_validate(hdrs, ifc), O.ipv4_validate_batch(hdrs, ifc.ext_tcp)), ("ipv4_validate", tag, nh)
    seeds = rng.integers(0, 65536, size=n).astype(np.uint16)
    g = ctx.csum_batch(arena, off16, lens, seeds, stride16=stride16)
    e = O.csum_batch(arena, off16, lens, seeds, stride16=stride16)
    assert (g == e).all(), ("csum", tag, np.nonzero(g != e)[0][:5])
    if mode == unet_amd.MODE_L3:
        ga, gs = ctx.generate(arena, off16, lens, stride16=stride16)
        ea, es = O.generate_batch(arena, off16, lens, stride16=stride16)
        assert (gs == es).all(), ("generate status", tag)
        assert (ga == ea).all(), ("generate bytes", tag)
    return tag


def tx_round(ctx, seed):
    """unet_tx_build on payloads of a random length mix and layout, every packet byte against the oracle."""
    from unet_amd._ffi import TX_DESC_DTYPE, tx_out_layout
    rng = np.random.default_rng(seed)
    mode = unet_amd.MODE_L2 if rng.random() < 0.4 else unet_amd.MODE_L3
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode)
    n = int(rng.choice([1, 15, 16, 17, 33, 64, 65, 300, 1500]))
    lens = np.asarray(LENGTH_CLASSES[int(rng.integers(len(LENGTH_CLASSES)))](rng, n), dtype=np.int64)
    frames = [rng.integers(0, 256, size=int(l), dtype=np.uint8).tobytes() for l in lens]
    arena, off16, stride16, plen, how = layout(rng, frames)
    desc = np.zeros(n, dtype=TX_DESC_DTYPE)
    desc["src"] = np.where(rng.random(n) < 0.2, 0, ifc.unicast)
    desc["dst"] = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    desc["l4a"] = rng.integers(0, 65536, size=n)
    desc["l4b"] = rng.integers(0, 65536, size=n)
    desc["values"] = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    desc["ip_id"] = rng.integers(0, 65536, size=n)
    desc["proto"] = np.where(rng.random(n) < 0.3, 1, np.where(rng.random(n) < 0.02, 99, 17))
    desc["dst_mac"] = rng.integers(0, 256, size=(n, 6), dtype=np.uint8)
    tag = (seed, n, mode, how, int(plen.max()), round(float(plen.mean()), 1))
    got, got_off, got_len, got_status = ctx.tx_build(arena, off16, plen, desc, ifc, payload_stride16=stride16)
    ooff, obytes = tx_out_layout(plen, mode)
    exp, exp_len, exp_status = O.tx_build_batch(arena, off16, plen, desc, ifc, ooff, obytes, pay_stride16=stride16)
    assert (got_off == ooff).all() and got.nbytes == obytes, ("tx layout", tag)
    assert (got_status == exp_status).all() and (got_len == exp_len).all(), ("tx status", tag)
    for o, l in zip(ooff, exp_len):
        a, b = int(o) * 16, int(o) * 16 + int(l)
        assert (got[a:b] == exp[a:b]).all(), ("tx bytes", tag, int(o), int(l))
    return tag


def dev_round(ctx, seed):
    """The device-pointer forms on a batch already in device memory, with no hint, a true one or a wrong one."""
    import torch
    rng = np.random.default_rng(seed)
    mode = unet_amd.MODE_L2 if rng.random() < 0.3 else unet_amd.MODE_L3
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode, ext_tcp=bool(rng.random() < 0.7))
    n = int(rng.choice([1, 7, 8, 9, 16, 33, 64, 65, 129, 500, 3000]))
    frames = make_frames(rng, n, ifc)
    arena, off16, stride16, lens, how = layout(rng, frames)
    hint = int(rng.choice([0, 0, max(1, int(lens.mean())), 40, 300, 1500, 9000, 60000]))
    max_hint = int(rng.choice([0, 0, 0, max(1, int(lens.max())), 64, 80, 96, 1000]))  # right, absent or wrong: the verdicts may not depend on it
    tag = (seed, n, mode, how, int(lens.max()), round(float(lens.mean()), 1), hint, max_hint)
    dev = torch.device("cuda:0")
    d_arena = torch.from_numpy(arena).to(dev)
    d_off = None if off16 is None else torch.from_numpy(off16.view(np.int32)).to(dev)
    p_off = 0 if d_off is None else d_off.data_ptr()
    d_len = torch.from_numpy(lens.view(np.int32)).to(dev)
    d_res = torch.empty(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.hint_mean_len(hint)
    ctx.hint_max_len(max_hint)
    try:
        ctx.verify_dev(d_arena.data_ptr(), arena.nbytes, p_off, stride16, d_len.data_ptr(), n, ifc, d_res.data_ptr())
        seeds = rng.integers(0, 65536, size=n).astype(np.uint16)
        d_seeds = torch.from_numpy(seeds.view(np.int16)).to(dev)
        d_sum = torch.empty(n, dtype=torch.int16, device=dev)
        torch.cuda.synchronize()
        ctx.csum_batch_dev(d_arena.data_ptr(), arena.nbytes, p_off, stride16, d_len.data_ptr(), d_seeds.data_ptr(), n, d_sum.data_ptr())
        torch.cuda.synchronize()
        got = d_res.cpu().numpy().view(unet_amd.RESULT_DTYPE)
        exp = O.recv_batch(arena, off16, lens, ifc, stride16=stride16)
        bad = np.nonzero(got != exp)[0]
        assert bad.size == 0, ("verify_dev", tag, bad[:5], got[bad[:5]], exp[bad[:5]], lens[bad[:5]])
        g = d_sum.cpu().numpy().view(np.uint16)
        e = O.csum_batch(arena, off16, lens, seeds, stride16=stride16)
        assert (g == e).all(), ("csum_batch_dev", tag, np.nonzero(g != e)[0][:5])
        if mode == unet_amd.MODE_L3:
            d_status = torch.empty(n, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            ctx.generate_dev(d_arena.data_ptr(), arena.nbytes, p_off, stride16, d_len.data_ptr(), n, d_status.data_ptr())
            torch.cuda.synchronize()
            ea, es = O.generate_batch(arena, off16, lens, stride16=stride16)
            assert (d_status.cpu().numpy() == es).all(), ("generate_dev status", tag)
            assert (d_arena.cpu().numpy() == ea).all(), ("generate_dev bytes", tag)
    finally:
        ctx.hint_mean_len(0)
        ctx.hint_max_len(0)
    return tag


def tx_dev_round(ctx, seed):
    """unet_tx_build_dev on payloads already in device memory, packets packed or on 128-byte lines, with no length hint, a
    true one or a wrong one; every packet byte against the oracle and every byte between packets untouched."""
    import torch
    from unet_amd._ffi import TX_DESC_DTYPE, TX_FILL_LINES, tx_out_layout
    rng = np.random.default_rng(seed)
    mode = unet_amd.MODE_L2 if rng.random() < 0.4 else unet_amd.MODE_L3
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode)
    n = int(rng.choice([1, 3, 4, 5, 16, 33, 64, 65, 300, 1500]))
    lens = np.asarray(LENGTH_CLASSES[int(rng.integers(len(LENGTH_CLASSES)))](rng, n), dtype=np.int64)
    frames = [rng.integers(0, 256, size=int(l), dtype=np.uint8).tobytes() for l in lens]
    arena, off16, stride16, plen, how = layout(rng, frames)
    desc = np.zeros(n, dtype=TX_DESC_DTYPE)
    desc["src"] = np.where(rng.random(n) < 0.2, 0, ifc.unicast)
    desc["dst"] = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    desc["l4a"] = rng.integers(0, 65536, size=n)
    desc["l4b"] = rng.integers(0, 65536, size=n)
    desc["values"] = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    desc["ip_id"] = rng.integers(0, 65536, size=n)
    desc["proto"] = np.where(rng.random(n) < 0.3, 1, 17)
    desc["dst_mac"] = rng.integers(0, 256, size=(n, 6), dtype=np.uint8)
    align16 = int(rng.choice([1, 1, 8, 8, 2]))
    hint = int(rng.choice([0, 0, max(1, int(plen.mean())), 40, 700, 3000, 60000]))
    max_hint = int(rng.choice([0, 0, max(1, int(plen.max())), 100, 1000]))  # right, absent or wrong: the bytes may not depend on it
    fill_lines = align16 == 8 and bool(rng.random() < 0.6)  # (only where nothing else lies between a packet's end and its line's)
    ooff, obytes = tx_out_layout(plen, mode, align16)
    tag = (seed, n, mode, how, int(plen.max()), round(float(plen.mean()), 1), align16, hint, max_hint, fill_lines)
    exp, exp_len, exp_status = O.tx_build_batch(arena, off16, plen, desc, ifc, ooff, obytes, pay_stride16=stride16)
    dev = torch.device("cuda:0")
    d_arena = torch.from_numpy(arena).to(dev)
    d_off = None if off16 is None else torch.from_numpy(off16.view(np.int32)).to(dev)
    d_len = torch.from_numpy(plen.view(np.int32)).to(dev)
    d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to(dev)
    d_ooff = torch.from_numpy(ooff.view(np.int32)).to(dev)
    d_out = torch.full((obytes + 256,), 0xAB, dtype=torch.uint8, device=dev)
    d_olen = torch.zeros(n, dtype=torch.int32, device=dev)
    d_st = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.hint_mean_len(hint)
    ctx.hint_max_len(max_hint)
    try:
        ctx.tx_build_dev(d_arena.data_ptr(), arena.nbytes, 0 if d_off is None else d_off.data_ptr(), stride16, d_len.data_ptr(),
                         d_desc.data_ptr(), n, ifc, d_out.data_ptr(), obytes, d_ooff.data_ptr(), d_olen.data_ptr(), d_st.data_ptr(),
                         flags=TX_FILL_LINES if fill_lines else 0)
        torch.cuda.synchronize()
    finally:
        ctx.hint_mean_len(0)
        ctx.hint_max_len(0)
    got, got_len, got_st = d_out.cpu().numpy(), d_olen.cpu().numpy().view(np.uint32), d_st.cpu().numpy()
    assert (got_st == exp_status).all() and (got_len == exp_len).all(), ("tx_dev status", tag)
    keep = np.ones(obytes + 256, dtype=bool)  # bytes no packet's chunks cover
    may_zero = np.zeros(obytes + 256, dtype=bool)  # with UNET_TX_FILL_LINES: from a packet's last chunk to the end of its 128-byte line
    for o, l in zip(ooff, exp_len):
        a, b = int(o) * 16, int(o) * 16 + int(l)
        assert (got[a:b] == exp[a:b]).all(), ("tx_dev bytes", tag, int(o), int(l))
        chunk_end = a + (int(l) + 15) // 16 * 16
        assert (got[b:chunk_end] == 0).all(), ("tx_dev pad of the last chunk", tag, int(o), int(l))
        keep[a:chunk_end] = False
        if fill_lines and l:
            may_zero[chunk_end:min((b + 127) // 128 * 128, obytes)] = True
    may_zero &= keep  # (a later packet's own bytes are its own)
    strict = keep & ~may_zero
    assert (got[strict] == 0xAB).all(), ("tx_dev wrote outside its packets", tag)
    rest = got[may_zero]
    assert ((rest == 0) | (rest == 0xAB)).all(), ("tx_dev: something other than zeros behind a packet", tag)
    return tag


# The library reads these when a context is opened: which transmit kernel builds a batch, and in what shape.
TX_CONTEXT_KNOBS = [
    {},
    {"UNET_TX_WAVE": "0"},
    {"UNET_TX_WAVE": "2", "UNET_TXW_GROUP": "1"},
    {"UNET_TX_WAVE": "2", "UNET_TXW_GROUP": "2"},
    {"UNET_TX_WAVE": "2", "UNET_TXW_GROUP": "8"},
    {"UNET_TX_WAVE": "2", "UNET_TXW_GROUP": "16"},
    {"UNET_TX_WAVE": "2", "UNET_TXW_GROUP": "4", "UNET_TXW_BLOCK": "64", "UNET_TXW_POLICY": "0"},
    {"UNET_TX_WAVE": "2", "UNET_TXW_GROUP": "1", "UNET_TXW_BLOCK": "128", "UNET_TXW_POLICY": "2"},
    {"UNET_TX_WAVE": "0", "UNET_TX_TILE": "64"},
    {"UNET_TX_WAVE": "0", "UNET_TX_TILE": "32"},
    {"UNET_TX_WAVE": "0", "UNET_TX_TILE": "16"},
    {"UNET_TX_WAVE": "0", "UNET_TX_TILE": "8"},
    {"UNET_TX_STAGE": "0"},
]


def open_with(knobs):
    names = {k for kn in TX_CONTEXT_KNOBS for k in kn}
    old = {k: os.environ.pop(k, None) for k in names}
    os.environ.update(knobs)
    try:
        return unet_amd.Offload(0)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = last = time.time()
    rounds = 0
    ctxs = [open_with(k) for k in TX_CONTEXT_KNOBS]
    try:
        ctx = ctxs[0]
        while time.time() - t0 < seconds:
            one_round(ctx, seed + rounds)
            dev_round(ctx, seed + rounds)
            tx_ctx = ctxs[rounds % len(ctxs)]
            tx_round(tx_ctx, seed + rounds)
            tx_dev_round(tx_ctx, seed + rounds)
            rounds += 1
            if time.time() - last > 30:
                last = time.time()
                print(f"{rounds} rounds, {time.time() - t0:.0f} s", flush=True)
    finally:
        bad = [c.debug_lds_violations() for c in ctxs]
        for c in ctxs:
            c.close()
    assert not any(bad), f"-DUNET_DEBUG library: out-of-area LDS stores counted per context {bad}"
    print(f"ok: {rounds} rounds from seed {seed}, {len(ctxs)} transmit configurations, every result equal to the oracle's")


if __name__ == "__main__":
    main()
