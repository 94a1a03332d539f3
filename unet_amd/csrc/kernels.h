// gfx950 kernels of the offload.  All of them are HBM-bound byte/integer work: 16-byte
// coalesced loads, one v_sad_u16 per dword, DPP wave reductions, no LDS except where a
// record is not a multiple of 16 bytes (the bare 20-byte header kernel).
#pragma once
#include "csum_core.h"
#include "parse.h"
#include "synth_spec.h"

namespace unetdev {

constexpr u32 kTile = 64;        // records per wave: lane j keeps record j's metadata and result
constexpr u32 kBlock = 256;      // 4 waves
constexpr u32 kMaxLen = 65535;   // IPv4 maximum (src/protocols/ipv4.rs:18), loopback MTU (src/devices/loopback.rs:16)

// Where the records of a batch live.
struct Batch {
    const uint8_t* arena;
    u64 arena_bytes;
    const u32* off16;  // may be null: fixed stride
    u32 stride16;
    const u32* len;
    u32 n;
};

// Header dwords of the record a wave is working on, held in row 0 (chunk `lane` of the record).
struct WaveFetch {
    u32x4 r;
    __device__ __forceinline__ u32 operator()(u32 k) const {
        const u32 l = k >> 2, s = k & 3u;
        const u32 x = rdlane(r.x, l), y = rdlane(r.y, l), z = rdlane(r.z, l), w = rdlane(r.w, l);
        return s == 0u ? x : (s == 1u ? y : (s == 2u ? z : w));
    }
};

// Per-lane tail masking of the last 16-byte chunk of a record (bytes at and beyond `flen` are
// pad and must not be summed).  m[] are wave-uniform.
struct TailMask {
    u32 last;  // index of the last chunk
    u32 m0, m1, m2, m3;
    __device__ __forceinline__ TailMask(u32 flen, u32 nch) {
        last = nch - 1u;
        const int tb = (int)(flen - last * 16u);  // 1..16 valid bytes in the last chunk
        m0 = keep_bytes(tb);
        m1 = keep_bytes(tb - 4);
        m2 = keep_bytes(tb - 8);
        m3 = keep_bytes(tb - 12);
    }
    __device__ __forceinline__ u32x4 apply(u32x4 v, u32 chunk) const {
        if (chunk == last) {
            v.x &= m0;
            v.y &= m1;
            v.z &= m2;
            v.w &= m3;
        }
        return v;
    }
};

// Record bytes are read exactly once and never reused: non-temporal 16-byte loads keep them from
// displacing anything in L2 / Infinity Cache (measured on MI355X: 7.0 TB/s streaming read with
// them against 6.2 TB/s with default-policy loads).
__device__ __forceinline__ u32x4 ld_stream(const u32x4* p) { return __builtin_nontemporal_load(p); }

// `p` points at the record, `nch` = ceil(flen / 16).
// Rows from chunk `from` on, four 1 KiB loads in flight per iteration; per-lane partial sums.
__device__ __forceinline__ u32 rows_sum(const u32x4* __restrict__ p, u32 nch, u32 from, const TailMask& tm, u32 lane, u32 acc) {
    for (u32 c0 = from; c0 < nch; c0 += 4u * kWave) {
        const u32 i0 = c0 + lane, i1 = i0 + kWave, i2 = i1 + kWave, i3 = i2 + kWave;
        u32x4 a = {0, 0, 0, 0}, b = a, c = a, d = a;
        if (i0 < nch) a = ld_stream(p + i0);
        if (i1 < nch) b = ld_stream(p + i1);
        if (i2 < nch) c = ld_stream(p + i2);
        if (i3 < nch) d = ld_stream(p + i3);
        acc = sum_chunk(tm.apply(a, i0), acc);
        acc = sum_chunk(tm.apply(b, i1), acc);
        acc = sum_chunk(tm.apply(c, i2), acc);
        acc = sum_chunk(tm.apply(d, i3), acc);
    }
    return acc;
}

__device__ __forceinline__ u32 record_sum(const u32x4* __restrict__ p, u32 flen, u32 nch, u32x4 row0, u32 lane) {
    const TailMask tm(flen, nch);
    const u32 acc = sum_chunk(tm.apply(row0, lane), 0u);
    return wave_sum(rows_sum(p, nch, kWave, tm, lane, acc));
}

// Tile bookkeeping shared by the wave-per-record kernels.
struct Tile {
    u32 first, cnt, lane;
    u32 my_len, my_slot;
    bool my_ok;
    template <int TILE = (int)kTile>
    __device__ __forceinline__ bool init(const Batch& b) {
        lane = threadIdx.x & 63u;
        const u32 wave = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
        first = wave * (u32)TILE;
        if (first >= b.n) return false;
        cnt = min((u32)TILE, b.n - first);
        my_len = 0;
        my_slot = 0;
        my_ok = false;
        if (lane < cnt) {
            const u32 i = first + lane;
            my_len = b.len[i];
            my_slot = b.off16 ? b.off16[i] : i * b.stride16;
            // the record and its pad to 16 bytes must lie inside the arena
            const u64 end = (u64)my_slot * 16u + (((u64)my_len + 15u) & ~15ull);
            my_ok = my_len <= kMaxLen && end <= b.arena_bytes;
        }
        return true;
    }
};

// ---------------------------------------------------------------- a1: calculate_checksum
// out[i] = calculate_checksum(record i, seeds[i])  (src/utils.rs:1-17)
__global__ __launch_bounds__(kBlock) void k_csum(Batch b, const uint16_t* __restrict__ seeds,
                                                 uint16_t* __restrict__ out) {
    Tile t;
    if (!t.init(b)) return;
    u32 my_seed = 0;
    if (seeds && t.lane < t.cnt) my_seed = seeds[t.first + t.lane];
    const u64 okmask = __ballot(t.my_ok);
    u32 my_res = 0;
    for (u32 j = 0; j < t.cnt; ++j) {
        const u32 flen = rdlane(t.my_len, j);
        const u32 slot = rdlane(t.my_slot, j);
        const u32 seed = rdlane(my_seed, j);
        if (!((okmask >> j) & 1ull)) continue;
        u32 total = 0;
        if (flen != 0u) {
            const u32x4* p = reinterpret_cast<const u32x4*>(b.arena + (u64)slot * 16u);
            const u32 nch = (flen + 15u) >> 4;
            u32x4 r0 = {0, 0, 0, 0};
            if (t.lane < nch) r0 = ld_stream(p + t.lane);
            total = record_sum(p, flen, nch, r0, t.lane);
        }
        const u32 res = finish(total + bswap16(seed));
        if (t.lane == j) my_res = res;
    }
    if (t.lane < t.cnt) out[t.first + t.lane] = (uint16_t)my_res;
}

// ------------------------------------------- a2-a6, a8, a11: receive-path verdicts
//
// Work split inside one wave (TILE consecutive records, lane j owns record j's metadata and result):
//   * streaming, wave-cooperative, one record at a time (records longer than kSmallChunks chunks):
//     all 64 lanes load the record in 16-byte chunks, sum the FULL chunks, and a DPP reduction leaves
//     the exact sum in lane j.  Nothing here depends on the record's contents.  The first six chunks
//     (every header byte any check can look at) and the partial last chunk are parked in LDS.
//   * lane-parallel: records of at most kSmallChunks chunks are loaded and summed by their own lane;
//     then every lane parses ITS record's headers out of LDS, adds the masked tail chunk, and turns
//     sum + headers into the verdict.  The whole O(1)-per-record decision logic therefore runs once
//     per tile on the vector unit instead of once per record on the (per-CU, shared) scalar unit.
constexpr u32 kSmallChunks = 8;   // records <= 128 bytes are handled by a single lane
constexpr u32 kHdrChunks = 6;     // 96 bytes: 14 (Ethernet) + 60 (max IPv4 header) + 16 (deepest L4 field) = 90
constexpr u32 kLdsChunksPerRec = kHdrChunks + 1;  // + the record's partial last chunk

// Header dwords of a lane's own record, parked in LDS.
struct LdsFetch {
    const u32* w;
    __device__ __forceinline__ u32 operator()(u32 k) const { return w[k]; }
};

// Rows 0 and 1 (the first 2 KiB) of one record, as loaded by a whole wave.
struct Rows2 {
    u32x4 r0, r1;
    const u32x4* p;
    u32 flen, nch, j;
    bool live;
};

// Issue (not wait for) the loads of record j of the tile.  Exactly two load instructions are issued
// whatever the record's length -- lanes past the end re-read the record's last chunk (same address,
// no extra HBM traffic) -- so the compiler can retire them with counted waits (vmcnt(2)) while the
// next record's pair is still in flight.  `live` = false issues the same two loads on the arena's
// first chunk and marks the buffer unused.
__device__ __forceinline__ Rows2 issue_rows(const Batch& b, u32 my_len, u32 my_slot, u32 lane, u32 j, bool live) {
    Rows2 x;
    x.j = j;
    x.live = live;
    x.flen = rdlane(my_len, j);
    const u32 slot = rdlane(my_slot, j);
    x.nch = live ? (x.flen + 15u) >> 4 : 0u;
    x.p = reinterpret_cast<const u32x4*>(b.arena + (live ? (u64)slot * 16u : 0ull));
    const u32 last = x.nch != 0u ? x.nch - 1u : 0u;
    x.r0 = ld_stream(x.p + min(lane, last));
    x.r1 = ld_stream(x.p + min(lane + kWave, last));
    return x;
}

__device__ __forceinline__ u32x4 zero_unless(u32x4 v, bool keep) {
    v.x = keep ? v.x : 0u;
    v.y = keep ? v.y : 0u;
    v.z = keep ? v.z : 0u;
    v.w = keep ? v.w : 0u;
    return v;
}

// Sum of the full chunks of one streamed record (wave-uniform result); parks header and tail chunk.
__device__ __forceinline__ u32 stream_one(const Rows2& cur, u32 lane, u32x4* __restrict__ park) {
    const u32 nfull = cur.flen >> 4;         // chunks that are summed whole
    const bool partial = (cur.flen & 15u) != 0u;  // chunk `nfull` exists and is cut short
    u32x4* mine = park + cur.j * kLdsChunksPerRec;
    if (lane < kHdrChunks) mine[lane] = cur.r0;
    const u32 i0 = lane, i1 = lane + kWave;
    if (partial && nfull < kWave) {
        if (i0 == nfull) mine[kHdrChunks] = cur.r0;
    } else if (partial && nfull < 2u * kWave) {
        if (i1 == nfull) mine[kHdrChunks] = cur.r1;
    }
    u32x4 a = cur.r0, c = cur.r1;
    if (nfull < kWave) a = zero_unless(a, i0 < nfull);
    if (nfull < 2u * kWave) c = zero_unless(c, i1 < nfull);
    u32 acc = sum_chunk(a, 0u);
    acc = sum_chunk(c, acc);
    for (u32 c0 = 2u * kWave; c0 < cur.nch; c0 += 4u * kWave) {
        const u32 k0 = c0 + lane, k1 = k0 + kWave, k2 = k1 + kWave, k3 = k2 + kWave;
        u32x4 v0 = {0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0;
        if (k0 < cur.nch) v0 = ld_stream(cur.p + k0);
        if (k1 < cur.nch) v1 = ld_stream(cur.p + k1);
        if (k2 < cur.nch) v2 = ld_stream(cur.p + k2);
        if (k3 < cur.nch) v3 = ld_stream(cur.p + k3);
        if (partial && nfull >= c0 && nfull < c0 + 4u * kWave) {  // the cut chunk is in this group
            if (k0 == nfull) { mine[kHdrChunks] = v0; v0 = u32x4{0, 0, 0, 0}; }
            if (k1 == nfull) { mine[kHdrChunks] = v1; v1 = u32x4{0, 0, 0, 0}; }
            if (k2 == nfull) { mine[kHdrChunks] = v2; v2 = u32x4{0, 0, 0, 0}; }
            if (k3 == nfull) { mine[kHdrChunks] = v3; v3 = u32x4{0, 0, 0, 0}; }
        }
        acc = sum_chunk(v0, acc);
        acc = sum_chunk(v1, acc);
        acc = sum_chunk(v2, acc);
        acc = sum_chunk(v3, acc);
    }
    return wave_sum(acc);
}

// Per-lane byte masks of a record's cut last chunk: keep the first (flen & 15) bytes.
struct LaneTail {
    u32 m0, m1, m2, m3;
    __device__ __forceinline__ explicit LaneTail(u32 flen) {
        const int tb = (int)(flen & 15u);
        m0 = keep_bytes(tb);
        m1 = keep_bytes(tb - 4);
        m2 = keep_bytes(tb - 8);
        m3 = keep_bytes(tb - 12);
    }
    __device__ __forceinline__ u32x4 apply(u32x4 v) const {
        v.x &= m0;
        v.y &= m1;
        v.z &= m2;
        v.w &= m3;
        return v;
    }
};

template <int MODE, int TILE>
__global__ __launch_bounds__(kBlock, 6) void k_verify(Batch b, IfaceDev ifc, u32* __restrict__ out) {
    static_assert(TILE >= 1 && TILE <= 64, "one lane per record of the tile");
    __shared__ u32x4 lds[(kBlock / kWave) * TILE * kLdsChunksPerRec];
    Tile t;
    if (!t.init<TILE>(b)) return;
    u32x4* park = lds + (threadIdx.x >> 6) * (TILE * kLdsChunksPerRec);
    u32x4* mine = park + t.lane * kLdsChunksPerRec;  // only meaningful for lane < TILE
    const u32 my_nch = t.my_ok ? (t.my_len + 15u) >> 4 : 0u;
    const bool my_small = t.my_ok && my_nch <= kSmallChunks;
    const bool my_big = t.my_ok && my_nch > kSmallChunks;
    const LaneTail tail(t.my_len);
    const bool my_partial = (t.my_len & 15u) != 0u;
    u32 my_total = 0;

    // ---- streamed records: first pair of loads goes out before anything else
    u64 todo = __ballot(my_big);
    Rows2 A, B;
    A.live = false;
    if (todo != 0ull) {
        const u32 jA = (u32)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        A = issue_rows(b, t.my_len, t.my_slot, t.lane, jA, true);
    }

    // ---- lane-parallel records: each lane pulls its own <= 8 chunks
    if (my_small) {
        const u32x4* p = reinterpret_cast<const u32x4*>(b.arena + (u64)t.my_slot * 16u);
        u32x4 c[kSmallChunks];
#pragma unroll
        for (u32 i = 0; i < kSmallChunks; ++i) {
            c[i] = u32x4{0, 0, 0, 0};
            if (i < my_nch) c[i] = ld_stream(p + i);
        }
        const u32 last = my_nch - 1u;  // my_nch == 0 never matches below
#pragma unroll
        for (u32 i = 0; i < kSmallChunks; ++i) {
            if (i < kHdrChunks) mine[i] = c[i];
            u32x4 v = c[i];
            if (my_partial && i == last) v = tail.apply(v);
            my_total = sum_chunk(v, my_total);
        }
    }

    // ---- stream the long records, two register buffers used alternately
    if (A.live) {
        while (true) {
            const bool hasB = todo != 0ull;
            const u32 jB = hasB ? (u32)__builtin_ctzll(todo) : A.j;
            todo &= todo - 1ull;
            B = issue_rows(b, t.my_len, t.my_slot, t.lane, jB, hasB);
            const u32 sA = stream_one(A, t.lane, park);
            if (t.lane == A.j) my_total = sA;
            if (!hasB) break;
            const bool hasA = todo != 0ull;
            const u32 jA = hasA ? (u32)__builtin_ctzll(todo) : B.j;
            todo &= todo - 1ull;
            A = issue_rows(b, t.my_len, t.my_slot, t.lane, jA, hasA);
            const u32 sB = stream_one(B, t.lane, park);
            if (t.lane == B.j) my_total = sB;
            if (!hasA) break;
        }
    }

    // ---- every lane finishes its own record: headers and tail chunk come back out of LDS
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u32 res = 0;  // V_NONE for records whose length/offset is out of range
    if (t.my_ok) {
        if (my_big && my_partial) my_total = sum_chunk(tail.apply(mine[kHdrChunks]), my_total);
        LdsFetch f;
        f.w = reinterpret_cast<const u32*>(mine);
        const Parsed P = parse_record<MODE>(f, t.my_len, ifc);
        res = P.kind == K_DONE ? P.res : finalize_record(P, my_total);
    }
    if (t.lane < t.cnt) out[t.first + t.lane] = res;
}

// ------------------------------------ a2-a4: Ipv4Header::try_from + validate, bare headers
// 20-byte records back to back.  Four headers are 80 bytes = five aligned 16-byte chunks, so a lane
// owns G groups of four headers: 5*G dwordx4 loads, all issued before the first use (the wave's
// instructions together cover one contiguous span, and with the default cache policy every line is
// fetched from HBM once and then hit in L1 by the neighbouring lanes' chunks -- non-temporal loads
// lose that reuse and run at 0.6x), twenty dwords per group in registers at compile-time offsets
// 0/5/10/15, one dwordx4 store of four results per group.  No LDS, no barrier.
constexpr u32 kHdrPerGroup = 4;

__global__ __launch_bounds__(kBlock) void k_ipv4_validate(const uint8_t* __restrict__ hdrs, u32 n, u32 ext_tcp,
                                                          u32* __restrict__ out) {
    const u32 ngroups = (n + kHdrPerGroup - 1) / kHdrPerGroup;
    for (u32 g = blockIdx.x * kBlock + threadIdx.x; g < ngroups; g += gridDim.x * kBlock) {
        const u32 h0 = g * kHdrPerGroup;
        if (h0 + kHdrPerGroup <= n) {
            const u32x4* src = reinterpret_cast<const u32x4*>(hdrs) + (u64)g * 5u;
            u32x4 d[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) d[k] = src[k];
            const u32 w[20] = {d[0].x, d[0].y, d[0].z, d[0].w, d[1].x, d[1].y, d[1].z, d[1].w, d[2].x, d[2].y,
                               d[2].z, d[2].w, d[3].x, d[3].y, d[3].z, d[3].w, d[4].x, d[4].y, d[4].z, d[4].w};
            u32x4 r;
            r.x = ipv4_header_verdict(w[0], w[1], w[2], w[3], w[4], ext_tcp);
            r.y = ipv4_header_verdict(w[5], w[6], w[7], w[8], w[9], ext_tcp);
            r.z = ipv4_header_verdict(w[10], w[11], w[12], w[13], w[14], ext_tcp);
            r.w = ipv4_header_verdict(w[15], w[16], w[17], w[18], w[19], ext_tcp);
            __builtin_nontemporal_store(r, reinterpret_cast<u32x4*>(out + h0));
        } else {
            // last 1..3 headers of the batch: dword loads, nothing beyond byte 20*n is touched
            const u32* w = reinterpret_cast<const u32*>(hdrs) + (u64)h0 * 5u;
            for (u32 h = h0; h < n; ++h, w += 5) out[h] = ipv4_header_verdict(w[0], w[1], w[2], w[3], w[4], ext_tcp);
        }
    }
}

// ------------------------------- a7, a9, a10: transmit-side checksum fill, in place
// status[i] = 0 when filled, else the reason the record was left alone.
__global__ __launch_bounds__(kBlock) void k_generate(Batch b, uint8_t* __restrict__ arena_rw,
                                                     uint8_t* __restrict__ status) {
    Tile t;
    if (!t.init(b)) return;
    const u64 okmask = __ballot(t.my_ok);
    u32 my_st = V_NONE;
    for (u32 j = 0; j < t.cnt; ++j) {
        const u32 flen = rdlane(t.my_len, j);
        const u32 slot = rdlane(t.my_slot, j);
        if (!((okmask >> j) & 1ull)) { if (t.lane == j) my_st = V_DROP_L3_TRUNCATED; continue; }
        const u32x4* p = reinterpret_cast<const u32x4*>(b.arena + (u64)slot * 16u);
        uint8_t* rec = arena_rw + (u64)slot * 16u;
        const u32 nch = (flen + 15u) >> 4;
        WaveFetch f;
        f.r = u32x4{0, 0, 0, 0};
        if (t.lane < nch) f.r = p[t.lane];
        u32 st = 0;
        if (flen < 20u) {
            st = V_DROP_L3_TRUNCATED;
        } else {
            const u32 w0 = f(0), w1 = f(1), w2 = f(2), w3 = f(3), w4 = f(4);
            const u32 proto = (w2 >> 8) & 0xffu;
            const u32 vhl = w0 & 0xffu;
            const u32 seglen = flen - 20u;
            if (!(proto == 1u || proto == 17u)) st = V_DROP_L3_PROTO;
            else if ((vhl >> 4) != 4u) st = V_DROP_L3_VERSION;
            else if ((vhl & 0xfu) < 5u) st = V_DROP_L3_IHL_SHORT;
            else if ((vhl & 0xfu) > 5u) st = V_DROP_L3_IHL_LONG;
            else if (seglen < 8u) st = proto == 17u ? V_DROP_UDP_SHORT : V_DROP_ICMP_TRUNCATED;
            else {
                const u32 total = record_sum(p, flen, nch, f.r, t.lane);
                // IPv4 header field zeroed, then summed (src/protocols/ipv4.rs:374-389)
                const u32 hdr = halves(w0) + halves(w1) + (w2 & 0xffffu) + halves(w3) + halves(w4);
                const u32 ipck = finish(hdr);
                const u32 prefix = hdr + (w2 >> 16);
                u32 l4ck, l4off;
                if (proto == 17u) {
                    // s = ck(pseudo, 0); c = ck(hdr(cksum 0) || payload, !s)  (src/transport/udp.rs:200-203)
                    const u32 u1 = f(6);
                    const u32 pseudo = halves(w3) + halves(w4) + 0x1100u + bswap16(seglen);
                    l4ck = finish(total - prefix - (u1 >> 16) + pseudo);
                    l4off = 26u;
                } else {
                    // ck(hdr(cksum 0) || data, 0)  (src/transport/icmp.rs:81-85)
                    const u32 m0 = f(5);
                    l4ck = finish(total - prefix - (m0 >> 16));
                    l4off = 22u;
                }
                if (t.lane == 0u) {
                    *reinterpret_cast<uint16_t*>(rec + 10) = (uint16_t)bswap16(ipck);
                    *reinterpret_cast<uint16_t*>(rec + l4off) = (uint16_t)bswap16(l4ck);
                }
            }
        }
        if (t.lane == j) my_st = st;
    }
    if (status && t.lane < t.cnt) status[t.first + t.lane] = (uint8_t)my_st;
}

// ------------------------------------------------------------------ verdict histogram
__global__ __launch_bounds__(kBlock) void k_count_verdicts(const u32* __restrict__ results, u32 n,
                                                           unsigned long long* __restrict__ counts) {
    __shared__ u32 hist[V_COUNT];
    if (threadIdx.x < V_COUNT) hist[threadIdx.x] = 0;
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * kBlock + threadIdx.x; i < n; i += (u64)gridDim.x * kBlock) {
        const u32 v = (results[i] >> 16) & 0xffu;
        atomicAdd(&hist[v < V_COUNT ? v : 0u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < V_COUNT && hist[threadIdx.x] != 0u) atomicAdd(&counts[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

// ------------------------------------------------------------------ flow hash
__host__ __device__ inline u32 rotl32(u32 x, int r) { return (x << r) | (x >> (32 - r)); }
// MurmurHash3 x86_32 over {src, dst, sport << 16 | dport, proto}, seed "unet".
__host__ __device__ inline u32 flow_hash5(u32 src, u32 dst, u32 proto, u32 sport, u32 dport) {
    const u32 w[4] = {src, dst, (sport << 16) | dport, proto};
    u32 h = 0x756e6574u;
    for (int i = 0; i < 4; i++) {
        u32 k = w[i] * 0xcc9e2d51u;
        k = rotl32(k, 15);
        k *= 0x1b873593u;
        h ^= k;
        h = rotl32(h, 13);
        h = h * 5u + 0xe6546b64u;
    }
    h ^= 16u;
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

// One lane per record; only the first few header bytes are touched.
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_flow_hash(Batch b, u32* __restrict__ out) {
    const u32 i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= b.n) return;
    const u32 flen = b.len[i];
    const u64 slot = b.off16 ? b.off16[i] : (u64)i * b.stride16;
    const u64 end = slot * 16u + (((u64)flen + 15u) & ~15ull);
    u32 h = 0;
    if (flen <= kMaxLen && end <= b.arena_bytes) {
        const u32* w = reinterpret_cast<const u32*>(b.arena + slot * 16u);
        auto f = [w](u32 k) { return w[k]; };
        u32 l3len = flen;
        bool ok = true;
        if (MODE == MODE_L2) {
            ok = flen >= 14u && bswap16(w[3] & 0xffffu) == 0x0800u;
            l3len = flen - 14u;
        }
        if (ok && l3len >= 20u) {
            const u32 w0 = l3raw<MODE>(f, 0), w2 = l3raw<MODE>(f, 8);
            const u32 src = bswap32(l3raw<MODE>(f, 12)), dst = bswap32(l3raw<MODE>(f, 16));
            const u32 proto = (w2 >> 8) & 0xffu;
            const u32 hl = (w0 & 0xfu) * 4u;
            u32 sport = 0, dport = 0;
            if ((proto == 17u || proto == 6u) && hl >= 20u && hl + 4u <= l3len) {
                const u32 p0 = l3raw<MODE>(f, hl);
                sport = bswap16(p0 & 0xffffu);
                dport = bswap16(p0 >> 16);
            }
            h = flow_hash5(src, dst, proto, sport, dport);
        }
    }
    out[i] = h;
}

// ------------------------------------------------------------------ bandwidth yardstick
// Pure streaming read: every workgroup owns one contiguous slab, four non-temporal 16-byte loads in
// flight per lane.  The best read-only shape found on MI355X; reported beside the 8 TB/s peak.
__global__ __launch_bounds__(kBlock) void k_stream_read(const u32x4* __restrict__ buf, u64 nchunks,
                                                        u32* __restrict__ sink) {
    u32 acc = 0;
    const u64 per = (nchunks + gridDim.x - 1) / gridDim.x;
    const u64 lo = (u64)blockIdx.x * per;
    const u64 hi = lo + per < nchunks ? lo + per : nchunks;
    u64 i = lo + threadIdx.x;
    for (; i + 3 * kBlock < hi; i += 4 * kBlock) {
        const u32x4 a = ld_stream(buf + i), b = ld_stream(buf + i + kBlock), c = ld_stream(buf + i + 2 * kBlock),
                    d = ld_stream(buf + i + 3 * kBlock);
        acc = sum_chunk(a, acc);
        acc = sum_chunk(b, acc);
        acc = sum_chunk(c, acc);
        acc = sum_chunk(d, acc);
    }
    for (; i < hi; i += kBlock) acc = sum_chunk(ld_stream(buf + i), acc);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63u) == 0u) atomicAdd(&sink[blockIdx.x & 4095u], acc);
}

// ------------------------------------------------------------------ synthetic traces
// Layer 1 (+ the post-checksum payload flip): one wave per record, 16 bytes per lane.
__global__ __launch_bounds__(kBlock) void k_synth_raw(int kind, u64 seed, u64 first_idx, u32 n,
                                                      const u32* __restrict__ off16, u32 stride16,
                                                      const u32* __restrict__ len, uint8_t* __restrict__ arena) {
    const u32 lane = threadIdx.x & 63u;
    const u32 rec = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (rec >= n) return;
    const u64 idx = first_idx + rec;
    const u64 key = unetsynth::key_of(seed, idx);
    const unetsynth::Shape s = unetsynth::shape_of(kind, seed, idx);
    const u32 flen = len[rec];
    const u64 slot = off16 ? off16[rec] : (u64)rec * stride16;
    u32 fpos = 0, fmask = 0;
    const bool has_flip = unetsynth::flip_of(kind, key, s, &fpos, &fmask);
    u64* dst = reinterpret_cast<u64*>(arena + slot * 16u);
    const u32 nch = (flen + 15u) >> 4;
    for (u32 c = lane; c < nch; c += kWave) {
        u64 a = unetsynth::raw_word(key, 2u * c), b2 = unetsynth::raw_word(key, 2u * c + 1u);
        if (has_flip && (fpos >> 4) == c) {
            const u64 m = (u64)fmask << (8u * (fpos & 7u));
            if (fpos & 8u) b2 ^= m; else a ^= m;
        }
        dst[2u * c] = a;
        dst[2u * c + 1u] = b2;
    }
}

// Layer 2: one lane per record builds the headers (with their checksums) and stores them.
__global__ __launch_bounds__(kBlock) void k_synth_hdr(int kind, u64 seed, u64 first_idx, u32 n, u32 unicast,
                                                      u32 broadcast, const u32* __restrict__ off16, u32 stride16,
                                                      uint8_t* __restrict__ arena) {
    const u32 rec = blockIdx.x * kBlock + threadIdx.x;
    if (rec >= n) return;
    const u64 idx = first_idx + rec;
    const u64 key = unetsynth::key_of(seed, idx);
    const unetsynth::Shape s = unetsynth::shape_of(kind, seed, idx);
    uint8_t h[96];
    const u32 hb = unetsynth::build_header(kind, key, s, unicast, broadcast, h);
    uint8_t* base = kind == unetsynth::kKindHdr20 ? arena + (u64)rec * 20u
                                                  : arena + (off16 ? (u64)off16[rec] : (u64)rec * stride16) * 16u;
    u32* dst = reinterpret_cast<u32*>(base);
    for (u32 i = 0; i < hb; i += 4) {
        dst[i >> 2] = (u32)h[i] | ((u32)h[i + 1] << 8) | ((u32)h[i + 2] << 16) | ((u32)h[i + 3] << 24);
    }
}

}  // namespace unetdev
