// gfx950 kernels of the offload.  All of them are HBM-bound byte/integer work: 16-byte coalesced
// loads, one v_sad_u16 per dword, DPP wave reductions; LDS only parks the few header bytes that the
// per-record decision logic reads back lane-parallel.
#pragma once
#include "csum_core.h"
#include "parse.h"
#include "synth_spec.h"

namespace unetdev {

#ifndef UNET_BLOCK
#define UNET_BLOCK 256
#endif
constexpr u32 kBlock = UNET_BLOCK;  // 4 waves
constexpr u32 kTxFillLines = 1;  // = UNET_TX_FILL_LINES of the C header
constexpr u32 kTxNoStage = 1u << 16;  // internal flag of k_tx_build (UNET_TX_STAGE=0 in the environment): no output staging, for A/B runs
constexpr u32 UNET_TX_OK_ = 0, UNET_TX_E_PROTO_ = 1, UNET_TX_E_TOO_LONG_ = 2, UNET_TX_E_RANGE_ = 3;  // = UNET_TX_* of the C header
constexpr u32 kMaxLen = 65535;   // IPv4 maximum (src/protocols/ipv4.rs:18), loopback MTU (src/devices/loopback.rs:16)

// Where the records of a batch live.
struct Batch {
    const uint8_t* arena;
    u64 arena_bytes;
    const u32* off16;  // may be null: fixed stride
    u32 stride16;
    const u32* len;
    u32 n;
    u32 tile_n = 0;  // records per wave if fewer than the kernel's TILE (0: TILE)
};

// Every store into a wave's LDS area goes through this: in a -DUNET_DEBUG build a store whose chunk index lies outside the
// area it is meant for is counted (unet_debug_lds_violations) and dropped instead of landing in a neighbouring
// wave's area, where it would show up as one wrong header byte in a few thousand batches.  Release builds store
// unconditionally: the index arithmetic is the same, the check costs registers the short-record kernels do not have.
#ifdef UNET_DEBUG
__device__ unsigned int g_lds_out_of_area = 0;
#define UNET_LDS_STORE(BASE, IDX, LIMIT, V)                        \
    do {                                                           \
        const u32 i_ = (IDX);                                      \
        if (i_ < (u32)(LIMIT)) (BASE)[i_] = (V);                   \
        else atomicAdd(&g_lds_out_of_area, 1u);                    \
    } while (0)
#else
#define UNET_LDS_STORE(BASE, IDX, LIMIT, V) ((void)(LIMIT), (BASE)[(IDX)] = (V))
#endif

// Build-time switches for A/B measurements (each path can be compiled out; the wave-cooperative path then
// takes those records).
#ifndef UNET_MID_PATH
#define UNET_MID_PATH 1       // tile_sums: lane groups for records of 81..1024 bytes
#endif
#ifndef UNET_OCC64
#define UNET_OCC64 8          // workgroups per CU the 64-record-tile kernels leave register room for
#endif
#ifndef UNET_SPAN_PATH
#define UNET_SPAN_PATH 1      // tile_sums: tiles of short records staged in LDS as one coalesced byte range
#endif
#ifndef UNET_STREAM_PATH
#define UNET_STREAM_PATH 1    // tile_run: 32-record tiles whose records lie back to back are read as one byte stream
#endif
#ifndef UNET_STREAM_TILE16
#define UNET_STREAM_TILE16 0  // stream route for the 16-record tiles of long records too
#endif
#ifndef UNET_STREAM_MAX_CHUNKS
#define UNET_STREAM_MAX_CHUNKS 0  // stream route only for tiles of at most this many chunks (0: any)
#endif
#ifndef UNET_STREAM_DEPTH
#define UNET_STREAM_DEPTH 4   // stream route: register rows in rotation (one less is in flight)
#endif
#ifndef UNET_SPAN_ONLY64
#define UNET_SPAN_ONLY64 0    // experiment: 32-record tiles of short records take the stream route instead of LDS staging
#endif
#ifndef UNET_TX_SMALL_PATH
#define UNET_TX_SMALL_PATH 1  // k_tx_build: one lane per packet up to 48-byte payloads
#endif
#ifndef UNET_TX_MID_PATH
#define UNET_TX_MID_PATH 1    // k_tx_build: lane groups for packets of up to 64 chunks
#endif
#ifndef UNET_TX_TILE_BURST
#define UNET_TX_TILE_BURST 4   // k_tx_build: rows of a long payload past its first two requested together (4; 2 saves three registers and
                               // costs the IMIX payload mix 1 %: 4.74 against 4.79 TB/s)
#endif
#ifndef UNET_TX_MID_DEPTH
#define UNET_TX_MID_DEPTH 1   // k_tx_build, lane groups: steps in flight.  2 (the next step's loads go out before the current one is built) costs
                              // 20 registers and with them the fifth wave per SIMD: the IMIX payload mix 4.70 -> 4.79 TB/s with 1, uniform sizes
                              // within +-1.5 % (profiles/r3/tx_mid_depth_ab.log)
#endif
#ifndef UNET_TX_STAGE
#define UNET_TX_STAGE 1       // k_tx_build: tiles of short packets that lie back to back in the output arena are assembled in LDS and stored as whole rows
#endif
#ifndef UNET_TX_G4
#define UNET_TX_G4 1          // k_tx_build: groups of 4 lanes for packets of up to 16 chunks
#endif

// Cache policy of the record loads.  Record bytes are read once, so bulk loads are non-temporal (pure
// streaming read on MI355X: 7.0 TB/s non-temporal vs 6.2 TB/s default).  But records are only 16-byte
// aligned, so a record's first and last 128-byte lines are shared with its neighbours, and a
// non-temporal load does not leave a line behind for the next touch: every load that holds a record's
// end, and every load of the lane-parallel and lane-group paths (neighbouring lanes or steps pull pieces
// of the same lines with separate instructions), uses the default policy.  UNET_NT_MASK (bit set =
// non-temporal) is the experiment knob: 1 the `head` row of a streamed record, 2 lane-parallel records,
// 4 rows of long transmit payloads.  ld_stream() is always non-temporal.
#ifndef UNET_NT_MASK
#define UNET_NT_MASK 1
#endif
__device__ __forceinline__ u32x4 ld_stream(const u32x4* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ u32x4 ld_row(const u32x4* p) { return (UNET_NT_MASK & 1) ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ u32x4 ld_small(const u32x4* p) { return (UNET_NT_MASK & 2) ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ u32x4 ld_far(const u32x4* p) { return (UNET_NT_MASK & 4) ? __builtin_nontemporal_load(p) : *p; }
#ifndef UNET_NT_W16
#define UNET_NT_W16 0
#endif
#ifndef UNET_NT_W8
#define UNET_NT_W8 0
#endif
#ifndef UNET_NT_W4
#define UNET_NT_W4 0
#endif

// ------------------------------------------------------------------------------------------------
// Tile machinery shared by k_verify, k_csum and k_generate.
//
// One wave owns TILE consecutive records; lane j keeps record j's length, slot and, at the end, its
// exact byte-swapped-word sum over [0, len).  Work split inside the wave, by record length:
//   * longer than 1 KiB -- streaming, wave-cooperative, one record at a time: all 64 lanes load the
//     record in 16-byte chunks, sum the FULL chunks, and a DPP reduction leaves the sum in lane j;
//   * up to 1 KiB -- the same with groups of 4, 8 or 16 lanes per record, 16, 8 or 4 records at a time;
//   * at most kSmallChunks chunks -- lane-parallel: loaded and summed by the record's own lane.
//   Nothing in the first two depends on the record's contents; the first six chunks (every header byte
//   any check can look at) and the partial last chunk are parked in LDS, and afterwards every lane adds
//   the byte-masked tail chunk of its own record.
// Callers then run their O(1)-per-record logic lane-parallel on headers read back from LDS, i.e. once
// per tile on the vector unit instead of once per record on the (per-CU, shared) scalar unit.
#ifndef UNET_SMALL_CHUNKS
#define UNET_SMALL_CHUNKS 5   // above 80 bytes a record is read by a group of 4, 8 or 16 lanes (measured break-even)
#endif
#ifndef UNET_ROW4_CHUNKS
#define UNET_ROW4_CHUNKS 16
#endif
constexpr u32 kSmallChunks = UNET_SMALL_CHUNKS;  // records of up to this many chunks are handled by a single lane
constexpr u32 kHdrChunks = 6;     // 96 bytes: 14 (Ethernet) + 60 (max IPv4 header) + 16 (deepest L4 field) = 90
// + the record's partial last chunk.  7 chunks = 28 dwords: lanes writing the same chunk of their own
// records spread over the LDS banks (a power-of-two stride would put them all on the same few).
#ifndef UNET_LDS_PER_REC
#define UNET_LDS_PER_REC (kHdrChunks + 1)  // (smaller values only for occupancy experiments on uniform short records)
#endif
constexpr u32 kLdsChunksPerRec = UNET_LDS_PER_REC;
constexpr u32 kRow4Chunks = UNET_ROW4_CHUNKS;  // longer records up to 16 chunks (256 bytes) are handled by 4 lanes,
constexpr u32 kRow8Chunks = 32;   // up to 32 chunks (512 bytes) by 8 lanes,
constexpr u32 kMidChunks = 64;    // up to 64 chunks (1024 bytes) by a 16-lane row

// LDS geometry of a tile.  A wave works through its tile in passes of at most 32 records, so that the park area
// -- and with it the number of workgroups a CU can hold -- does not grow with 64-record tiles: 4.5 KiB per wave
// lets eight workgroups share a CU (with 7.5 KiB it was five, and short records, which keep few bytes in flight
// per wave, ran 10 % slower).
template <int TILE>
struct TileGeom {
    static constexpr u32 kPass = TILE < 32 ? (u32)TILE : 32u;   // records per pass; record j is parked at slot j % kPass
    static constexpr u32 kPasses = (u32)TILE / kPass;           // 1 or 2
    // chunks of the park area: kPass record areas, rounded up to whole 1 KiB rows for the span route
    static constexpr bool kStream = UNET_STREAM_PATH && (TILE == 32 || (UNET_STREAM_TILE16 && TILE == 16));  // the stream route
    // (its bitmaps, 32 chunks, sit behind the record areas: in the unused end of the 256 chunks, or appended)
    static constexpr u32 kPark = TILE >= 32 ? 256u : kPass * kLdsChunksPerRec + (kStream ? 32u : 0u);
    static constexpr u32 kLists = (2u * (u32)TILE + 3u) / 4u;            // TILE u32 sums + TILE u32 list entries
    static constexpr u32 kWaveLds = kPark + kLists;
    static_assert(kPark >= kPass * kLdsChunksPerRec, "park area holds a pass of records");
};

// Header dwords of a lane's own record, parked in LDS.
struct LdsFetch {
    const u32* w;
    __device__ __forceinline__ u32 operator()(u32 k) const { return w[k]; }
};
// Any dword of the lane's own record: out of LDS while it is parked there, from memory past that.
struct DeepFetch {
    const u32* w;     // LDS
    const u32* g;     // the record in the arena
    u32 lds_dwords;   // dwords of the record that LDS holds
    __device__ __forceinline__ u32 operator()(u32 k) const { return k < lds_dwords ? w[k] : g[k]; }
};

struct Tile {
    u32 first, cnt, lane;
    u32 my_len, my_slot;
    bool my_ok;
    const u32x4* my_hdr;  // set by the tile routes: where the lane's own record starts in LDS (at least its first six chunks)
    u32 hdr_dwords;       // how many dwords of the record LDS holds there
    template <int TILE>
    __device__ __forceinline__ bool init(const Batch& b) {
        lane = threadIdx.x & 63u;
        const u32 wave = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
        const u32 per = b.tile_n != 0u && b.tile_n < (u32)TILE ? b.tile_n : (u32)TILE;
        first = wave * per;
        if (first >= b.n) return false;
        cnt = min(per, b.n - first);
        my_len = 0;
        my_slot = 0;
        my_ok = false;
        if (lane < cnt) {
            const u32 i = first + lane;
            my_len = b.len[i];
            // slots are 32-bit: a fixed-stride batch whose last record starts past slot 2^32 - 1 (an arena of
            // 64 GiB or more) must not wrap onto another record
            const u64 slot = b.off16 ? (u64)b.off16[i] : (u64)i * b.stride16;
            my_slot = (u32)slot;
            // the record and its pad to 16 bytes must lie inside the arena
            const u64 end = slot * 16u + (((u64)my_len + 15u) & ~15ull);
            my_ok = my_len <= kMaxLen && slot <= 0xffffffffull && end <= b.arena_bytes;
        }
        return true;
    }
};

// The two ends of one record, as loaded by a whole wave: `tail` always holds the record's end (its last
// 64 chunks when the record is longer than 2 KiB, chunks 64.. when it is longer than 1 KiB, all of it
// otherwise), `head` holds chunks 0..63 of a record longer than 1 KiB and is a dummy otherwise.  What
// lies between the two in a record longer than 2 KiB is read by stream_one itself.
struct Rows2 {
    u32x4 head, tail;

    const u32x4* p;
    u32 flen, nch, j;
    bool live;
    __device__ __forceinline__ bool two() const { return nch > kWave; }           // head is in use
    // chunk index of tail's lane 0
    __device__ __forceinline__ u32 tail_base() const { return nch > 2u * kWave ? nch - kWave : (two() ? kWave : 0u); }
};

// Issue (not wait for) the loads of record j of the tile.  Exactly two load instructions are issued
// whatever the record's length -- lanes past the end re-read the record's last chunk (same address,
// no extra HBM traffic), an unused `head` reads the arena's first chunk -- so the compiler can retire
// them with counted waits (vmcnt(2)) while the next record's pair is still in flight.  `live` = false
// issues the same two loads on the arena's first chunk and marks the buffer unused.
//
// Cache policy: the 128-byte line a record ends in is shared with the next record, which the same wave
// reads next.  The load that holds the record's end (`tail`) therefore uses the default policy, so that
// line is still in L2 then; `head` ends inside the record and streams past the cache (non-temporal).
__device__ __forceinline__ Rows2 issue_rows(const Batch& b, u32 my_len, u32 my_slot, u32 lane, u32 j, bool live) {
    Rows2 x;
    x.j = j;
    x.live = live;
    x.flen = rdlane(my_len, j);
    const u32 slot = rdlane(my_slot, j);
    x.nch = live ? (x.flen + 15u) >> 4 : 0u;
    x.p = reinterpret_cast<const u32x4*>(b.arena + (live ? (u64)slot * 16u : 0ull));
    const u32 last = x.nch != 0u ? x.nch - 1u : 0u;
    const bool two = x.two();
    const u32x4* hp = two ? x.p : reinterpret_cast<const u32x4*>(b.arena);
    x.head = ld_row(hp + (two ? lane : 0u));
    x.tail = x.p[min(lane + x.tail_base(), last)];
    return x;
}

__device__ __forceinline__ u32x4 zero_unless(u32x4 v, bool keep) {
    v.x = keep ? v.x : 0u;
    v.y = keep ? v.y : 0u;
    v.z = keep ? v.z : 0u;
    v.w = keep ? v.w : 0u;
    return v;
}

__device__ __forceinline__ u32x4 pick_chunk(bool first, u32x4 a, u32x4 b) {
    return u32x4{first ? a.x : b.x, first ? a.y : b.y, first ? a.z : b.z, first ? a.w : b.w};
}

// Sum of the full chunks of one streamed record (wave-uniform result); parks header and tail chunk.
template <bool PARK_HDR, u32 PASS>
__device__ __forceinline__ u32 stream_one(const Rows2& cur, u32 lane, u32x4* __restrict__ park) {
    const u32 nfull = cur.flen >> 4;              // chunks that are summed whole
    const bool partial = (cur.flen & 15u) != 0u;  // chunk `nfull` exists and is cut short
    const bool two = cur.two();
    const u32 i1 = lane + cur.tail_base();        // chunk held by this lane's `tail`
    const u32 at = (cur.j % PASS) * kLdsChunksPerRec;  // the record's slot of the park area (PASS slots)
    if (PARK_HDR && lane < kHdrChunks) UNET_LDS_STORE(park, at + lane, PASS * kLdsChunksPerRec, pick_chunk(two, cur.head, cur.tail));
    // the cut chunk is the record's last one: always in `tail`
    if (partial && i1 == nfull) UNET_LDS_STORE(park, at + kHdrChunks, PASS * kLdsChunksPerRec, cur.tail);
    const u32x4 a = zero_unless(cur.head, two);
    const u32x4 c = zero_unless(cur.tail, i1 < nfull);
    u32 acc = sum_chunk(a, 0u);
    acc = sum_chunk(c, acc);
    // between head and tail (records longer than 2 KiB): four 1 KiB streaming loads in flight per step
    const u32 mid_end = cur.tail_base();
    u32 c0 = kWave;
#pragma nounroll
    for (; c0 + 4u * kWave <= mid_end; c0 += 4u * kWave) {
        const u32x4* q = cur.p + c0 + lane;
        const u32x4 v0 = ld_stream(q), v1 = ld_stream(q + kWave), v2 = ld_stream(q + 2u * kWave), v3 = ld_stream(q + 3u * kWave);
        acc = sum_chunk(v0, acc);
        acc = sum_chunk(v1, acc);
        acc = sum_chunk(v2, acc);
        acc = sum_chunk(v3, acc);
    }
    if (c0 < mid_end) {
        const u32 k0 = c0 + lane, k1 = k0 + kWave, k2 = k1 + kWave, k3 = k2 + kWave;
        u32x4 v0 = {0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0;
        if (k0 < mid_end) v0 = ld_stream(cur.p + k0);
        if (k1 < mid_end) v1 = ld_stream(cur.p + k1);
        if (k2 < mid_end) v2 = ld_stream(cur.p + k2);
        if (k3 < mid_end) v3 = ld_stream(cur.p + k3);
        acc = sum_chunk(v0, acc);
        acc = sum_chunk(v1, acc);
        acc = sum_chunk(v2, acc);
        acc = sum_chunk(v3, acc);
    }
    return wave_sum(acc);
}

// Wave-uniform byte masks for the cut last chunk (index nch - 1) of a buffer of `flen` bytes.
struct TailMask {
    u32 last;
    u32 m0, m1, m2, m3;
    __device__ __forceinline__ TailMask(u32 flen, u32 nch) {
        last = nch - 1u;
        const int tb = (int)(flen - last * 16u);  // 1..16 valid bytes in the last chunk (16 when nch == 0)
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

__device__ __forceinline__ u32 from_lane(u32 v, u32 src_lane) {
    return (u32)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}

// Lane i receives lane i+1's value; lane 63 receives `after` (DPP wave_shl:1, unwritten lanes keep `old`).
__device__ __forceinline__ u32 from_next_lane(u32 v, u32 after) {
    return (u32)__builtin_amdgcn_update_dpp((int)after, (int)v, 0x130, 0xf, 0xf, false);
}

// Mid-size records, 64/W at a time, W (8 or 16) neighbouring lanes per record -- a whole wave on one such
// record would leave most lanes idle and keep little in flight.  Lane `sub` of a group loads chunks sub,
// sub+W, sub+2W, sub+3W of its group's record (exactly four loads, clamped to the last chunk, default
// cache policy: each holds or neighbours a record end), so W = 8 serves records of up to 32 chunks with one
// full 128-byte line per record and load, W = 16 records of up to 64 chunks.  Two steps are in flight: the
// next step's loads go out before the current one is summed.  The sum of a record's full chunks is reduced
// inside its group with DPP butterflies and left in `sums[record]`; the first six chunks and the cut chunk
// go to the record's LDS area.  `list` holds the tile's records of this class, in lane order.
struct RowStep {
    u32x4 v0, v1, v2, v3;
    u32 rec, len;    // per group
    bool row_live;   // per group
};

template <u32 W>
__device__ __forceinline__ RowStep issue_rows_step(const Batch& b, const Tile& t, const u32* __restrict__ list, u32 first, u32 count) {
    const u32 sub = t.lane & (W - 1u), k = first + t.lane / W;
    RowStep g;
    g.row_live = k < count;
    g.rec = list[g.row_live ? k : 0u];
    g.len = from_lane(t.my_len, g.rec);
    const u32 slot = from_lane(t.my_slot, g.rec);
    const u32 nch = g.row_live ? (g.len + 15u) >> 4 : 0u;
    const u32 last = nch != 0u ? nch - 1u : 0u;
    const u32x4* p = reinterpret_cast<const u32x4*>(b.arena + (g.row_live ? (u64)slot * 16u : 0ull));
    // bit k of the mask: load k streams past the cache (see the cache-policy note at the top)
    constexpr u32 nt = W == 16u ? UNET_NT_W16 : (W == 8u ? UNET_NT_W8 : UNET_NT_W4);
    g.v0 = (nt & 1u) ? ld_stream(p + min(sub, last)) : p[min(sub, last)];
    g.v1 = (nt & 2u) ? ld_stream(p + min(sub + W, last)) : p[min(sub + W, last)];
    g.v2 = (nt & 4u) ? ld_stream(p + min(sub + 2u * W, last)) : p[min(sub + 2u * W, last)];
    g.v3 = (nt & 8u) ? ld_stream(p + min(sub + 3u * W, last)) : p[min(sub + 3u * W, last)];
    return g;
}

// Sum over each group of W neighbouring lanes, left in every lane of the group.
template <u32 W>
__device__ __forceinline__ u32 group_sum(u32 v) {
    v = dpp_add<0xb1, 0xf>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4e, 0xf>(v);   // quad_perm [2,3,0,1]
    if (W >= 8u) v = dpp_add<0x141, 0xf>(v);   // row_half_mirror: the other quad of the 8
    if (W == 16u) v = dpp_add<0x140, 0xf>(v);  // row_mirror: the other 8 of the 16
    return v;
}

template <u32 W, bool PARK_HDR, u32 PASS>
__device__ __forceinline__ void consume_rows_step(const RowStep& g, u32 lane, u32x4* __restrict__ park, u32* __restrict__ sums) {
    const u32 sub = lane & (W - 1u);
    const u32 nfull = g.row_live ? g.len >> 4 : 0u;
    const u32 at = (g.rec % PASS) * kLdsChunksPerRec;
    constexpr u32 kLimit = PASS * kLdsChunksPerRec;
    if (PARK_HDR && g.row_live) {  // chunks 0..5: in v0 for W >= 8, in v0 and v1 for W = 4
        if (sub < kHdrChunks) UNET_LDS_STORE(park, at + sub, kLimit, g.v0);
        if (W < kHdrChunks && sub + W < kHdrChunks) UNET_LDS_STORE(park, at + sub + W, kLimit, g.v1);
    }
    if (g.row_live && (g.len & 15u) != 0u && sub == (nfull & (W - 1u))) {
        const u32 w = nfull / W;  // which of the four loads holds the cut chunk
        UNET_LDS_STORE(park, at + kHdrChunks, kLimit, w == 0u ? g.v0 : (w == 1u ? g.v1 : (w == 2u ? g.v2 : g.v3)));
    }
    u32 acc = sum_chunk(zero_unless(g.v0, sub < nfull), 0u);
    acc = sum_chunk(zero_unless(g.v1, sub + W < nfull), acc);
    acc = sum_chunk(zero_unless(g.v2, sub + 2u * W < nfull), acc);
    acc = sum_chunk(zero_unless(g.v3, sub + 3u * W < nfull), acc);
    acc = group_sum<W>(acc);
    if (g.row_live && sub == 0u) sums[g.rec] = acc;
}

template <u32 W, bool PARK_HDR, u32 PASS>
__device__ __forceinline__ void rows_path(const Batch& b, const Tile& t, bool mine_here, u32* __restrict__ list,
                                          u32x4* __restrict__ park, u32* __restrict__ sums) {
    constexpr u32 kPerStep = kWave / W;
    const u64 mask = __ballot(mine_here);
    if (mask == 0ull) return;
    const u32 count = (u32)__builtin_popcountll(mask);
    // the records of this class, compacted in lane order
    if (mine_here) list[__builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u))] = t.lane;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u32 next = 0;
    RowStep g0 = issue_rows_step<W>(b, t, list, next, count), g1;
    next += kPerStep;
#pragma nounroll
    while (true) {
        g1 = issue_rows_step<W>(b, t, list, next, count);
        consume_rows_step<W, PARK_HDR, PASS>(g0, t.lane, park, sums);
        if (next >= count) break;
        next += kPerStep;
        g0 = issue_rows_step<W>(b, t, list, next, count);
        consume_rows_step<W, PARK_HDR, PASS>(g1, t.lane, park, sums);
        if (next >= count) break;
        next += kPerStep;
    }
}

// ---- span route.  A tile made only of short records (at most 8 chunks each) that lie close together -- any
// packed or fixed-stride batch of short frames: the byte range that holds the tile's records (at most 256 chunks)
// is staged in the wave's LDS area with whole 1 KiB streaming rows, all issued up front, and every lane then
// sums its own record out of LDS.  One lane loading its own 64-byte record straight from memory makes every load
// instruction touch 32 different 128-byte lines; here each line is requested once.
constexpr u32 kSpanMaxChunks = 8;

struct SpanPlan {
    bool on;               // the tile takes this route
    u32 base, span, rows;  // first chunk, chunks, 1 KiB rows
    u32 max_nch;           // longest record of the tile, in chunks
    u32x4 r0, r1, r2, r3;  // the rows, loads in flight
};

template <int TILE>
__device__ __forceinline__ SpanPlan plan_span(const Batch& b, const Tile& t, u32 my_nch) {
    typedef TileGeom<TILE> G;
    static_assert(G::kPark <= 4u * kWave, "four rows hold the park area");
    SpanPlan sp;
    sp.on = false;
    sp.base = sp.span = sp.rows = sp.max_nch = 0;
    sp.r0 = sp.r1 = sp.r2 = sp.r3 = u32x4{0, 0, 0, 0};
    const bool in = t.lane < t.cnt;
    const bool fits = !in || (t.my_ok && my_nch <= kSpanMaxChunks && t.my_slot <= 0xfffffff0u);
    if (__ballot(fits) != ~0ull) return sp;
    const bool some = in && my_nch != 0u;
    const u32 lo = wave_minmax<false>(some ? t.my_slot : 0xffffffffu);
    const u32 end = wave_minmax<true>(some ? t.my_slot + my_nch : 0u);
    if (end > lo) {
        // start on a 128-byte line when the range still fits then
        sp.base = end - (lo & ~7u) <= G::kPark ? (lo & ~7u) : lo;
        sp.span = end - sp.base;
    }
    if (sp.span > G::kPark) return sp;
    sp.on = true;
    sp.rows = (sp.span + kWave - 1u) / kWave;
    sp.max_nch = wave_minmax<true>(my_nch);
    const u32x4* p = reinterpret_cast<const u32x4*>(b.arena) + (u64)sp.base;
    // wave-uniform branches: exactly the rows that exist are requested
    if (sp.rows > 0u) sp.r0 = ld_stream(p + min(t.lane, sp.span - 1u));
    if (sp.rows > 1u) sp.r1 = ld_stream(p + min(kWave + t.lane, sp.span - 1u));
    if (sp.rows > 2u) sp.r2 = ld_stream(p + min(2u * kWave + t.lane, sp.span - 1u));
    if (sp.rows > 3u) sp.r3 = ld_stream(p + min(3u * kWave + t.lane, sp.span - 1u));
    return sp;
}

// The rows go to LDS, every lane sums its own record; t.my_hdr then points at it in LDS.
template <int TILE>
__device__ __forceinline__ u32 span_sums(const SpanPlan& sp, Tile& t, u32x4* __restrict__ park) {
    // With 16-record tiles the wave's LDS area is 112 chunks, less than two rows: there only the chunks of the range
    // are stored.  (The 256 chunks of the larger tiles hold four whole rows; testing there as well costs them 16
    // registers and two waves per SIMD.)
    // (UNET_DEBUG_SPAN_WHOLE_ROWS brings back the fault this condition repairs -- whole rows stored into the small area of
    // 16-record tiles -- as a self-test of the -DUNET_DEBUG build: its counter must then move)
#ifdef UNET_DEBUG_SPAN_WHOLE_ROWS
    constexpr bool kWholeRows = true;
#else
    constexpr bool kWholeRows = TileGeom<TILE>::kPark >= 4u * kWave;
#endif
    constexpr u32 kLimit = TileGeom<TILE>::kPark;
    if (sp.rows > 0u && (kWholeRows || t.lane < sp.span)) UNET_LDS_STORE(park, t.lane, kLimit, sp.r0);
    if (sp.rows > 1u && (kWholeRows || kWave + t.lane < sp.span)) UNET_LDS_STORE(park, kWave + t.lane, kLimit, sp.r1);
    if (sp.rows > 2u && (kWholeRows || 2u * kWave + t.lane < sp.span)) UNET_LDS_STORE(park, 2u * kWave + t.lane, kLimit, sp.r2);
    if (sp.rows > 3u && (kWholeRows || 3u * kWave + t.lane < sp.span)) UNET_LDS_STORE(park, 3u * kWave + t.lane, kLimit, sp.r3);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u32 my_nch = (t.my_len + 15u) >> 4;
    const bool some = t.lane < t.cnt && my_nch != 0u;
    const u32x4* mine = park + (some ? t.my_slot - sp.base : 0u);
    t.my_hdr = mine;
    t.hdr_dwords = 0xffffffffu;  // the whole record is staged
    const LaneTail tail(t.my_len);
    const bool partial = (t.my_len & 15u) != 0u;
    const u32 nfull = t.my_len >> 4;
    u32 acc = 0;
#pragma unroll
    for (u32 k = 0; k < kSpanMaxChunks; ++k) {
        if (k < sp.max_nch) {  // wave-uniform
            u32x4 c = mine[k];
            const u32x4 cut = tail.apply(c);
            const bool whole = k < nfull, last = partial && k == nfull;
            c.x = whole ? c.x : (last ? cut.x : 0u);
            c.y = whole ? c.y : (last ? cut.y : 0u);
            c.z = whole ? c.z : (last ? cut.z : 0u);
            c.w = whole ? c.w : (last ? cut.w : 0u);
            acc = sum_chunk(c, acc);
        }
    }
    return some ? acc : 0u;
}

// ---- stream route.  A tile whose records lie back to back in the arena, in index order and without gaps -- any
// packed batch -- is one byte range.  The wave reads that range in whole 1 KiB rows (lane L of row r holds chunk
// 64 r + L of the range), streaming, three rows in flight, whatever the records' lengths: every 128-byte line is
// requested by exactly one instruction, where the lane-group route asks for the two ends of a record's lines from
// different instructions and so has to keep them cached.
//   Which record a chunk belongs to comes from two bitmaps with one bit per chunk, built once per tile in LDS:
// S marks the chunks a record starts on, C the cut last chunks (records whose length is not a multiple of 16).
// A row's 64 bits of each are wave-uniform, so the masks "summed whole", "last chunk of its record in this row"
// and the count of records begun in earlier rows are scalar arithmetic; the record number of a lane is that
// count plus the S bits up to the lane.  Per row the lanes take an (unconditional) inclusive prefix sum P of
// their chunk sums; the last lane e of each record's run adds P[e] to the record's total in LDS and takes it off
// the next record's, which leaves every record with exactly the sum of its own chunks (mod 2^32, and the true
// sums are below 2^31).  Header chunks and cut chunks are parked as in the general route.
constexpr u32 kStreamRows = 32;  // rows per bitmap build (16 bytes per row): a longer range is worked through in segments of 32 KiB

// Inclusive prefix sum over the 64 lanes.
__device__ __forceinline__ u32 wave_prefix(u32 v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ u64 uniform64(u32x4 w, bool second) {
    const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(second ? w.z : w.x));
    const u32 hi = (u32)__builtin_amdgcn_readfirstlane((int)(second ? w.w : w.y));
    return (u64)hi << 32 | lo;
}

// Wave-uniform: does the tile take the stream route, and over which chunks [base, base + span) of the arena;
// `lead` chunks at the front of the range (it starts on a 128-byte line) belong to no record of the tile.
template <int TILE>
__device__ __forceinline__ bool plan_stream(const Tile& t, u32 my_nch, u32* base, u32* span, u32* lead) {
    static_assert(TileGeom<TILE>::kPark >= (u32)TILE * kLdsChunksPerRec + kStreamRows, "bitmaps behind the record areas");
    const bool in = t.lane < t.cnt;
    const u32 next_slot = from_next_lane(t.my_slot, 0xffffffffu);
    const bool is_last = t.lane + 1u >= t.cnt;
    // every record valid, non-empty (two records may not start on the same chunk) and followed at once by the next
    const bool fits = !in || (t.my_ok && my_nch >= 1u && t.my_slot <= 0xffff0000u && (is_last || t.my_slot + my_nch == next_slot));
    if (__ballot(fits) != ~0ull) return false;
    const u32 lo = rdlane(t.my_slot, 0u);
    const u32 end = rdlane(t.my_slot + my_nch, t.cnt - 1u);
    *base = lo & ~7u;
    *lead = lo & 7u;
    *span = end - *base;
    return UNET_STREAM_MAX_CHUNKS == 0 || *span <= (u32)UNET_STREAM_MAX_CHUNKS;
}

// One row.  `w` = {S, C} of this row; `carry` = records begun in earlier rows.
template <int TILE, bool PARK_HDR>
__device__ __forceinline__ void stream_row(const u32x4 v, const u32x4 w, u32 row, u32 span, u32 lead, const Tile& t, u32 my_rel,
                                           u32x4* __restrict__ park, u32* __restrict__ sums, u32& carry) {
    const u64 S = uniform64(w, false), C = uniform64(w, true);
    const u32 left = span - row * kWave;  // chunks of the range from this row on
    u64 valid = left >= kWave ? ~0ull : (1ull << left) - 1ull;
    if (row == 0u) valid &= ~0ull << lead;
    // record number: records begun before this row + S bits at or below the lane, less one
    const u64 S1 = S >> 1;
    const u32 c0 = carry + (u32)(S & 1ull) - 1u;
    const u32 rid = __builtin_amdgcn_mbcnt_hi((u32)(S1 >> 32), __builtin_amdgcn_mbcnt_lo((u32)S1, c0));
    carry += (u32)__builtin_popcountll(S);
    const u32 at = rid * kLdsChunksPerRec;  // record rid's slot: one per record of the tile, the bitmaps lie behind them
    constexpr u32 kLimit = (u32)TILE * kLdsChunksPerRec;
    // wave-uniform masks become per-lane conditions with no vector arithmetic
    if (PARK_HDR) {
        const u32 ci = row * kWave + t.lane - from_lane(my_rel, rid);
        if (__builtin_amdgcn_inverse_ballot_w64(valid) && ci < kHdrChunks) UNET_LDS_STORE(park, at + ci, kLimit, v);
    }
    if (__builtin_amdgcn_inverse_ballot_w64(C & valid)) UNET_LDS_STORE(park, at + kHdrChunks, kLimit, v);
    const u32 P = wave_prefix(sum_chunk(zero_unless(v, __builtin_amdgcn_inverse_ballot_w64(valid & ~C)), 0u));
    // the last lane of every run: the lane before a record start, and the row's last chunk
    const u64 before_start = S1 & valid;
    const u64 row_end = 1ull << (min(left, kWave) - 1u);
    if (__builtin_amdgcn_inverse_ballot_w64(before_start | row_end)) {
        __hip_atomic_fetch_add(sums + rid, P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (__builtin_amdgcn_inverse_ballot_w64(before_start)) __hip_atomic_fetch_sub(sums + rid + 1u, P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}

// Exact sum of the lane's own record; t.my_hdr points at its parked chunks 0..5 afterwards.
template <int TILE, bool PARK_HDR>
__device__ __forceinline__ u32 stream_sums(const Batch& b, Tile& t, u32 base, u32 span, u32 lead, u32x4* __restrict__ park) {
    typedef TileGeom<TILE> G;
    static_assert(G::kPasses == 1u, "one park slot per record of the tile");
    u32* sums = reinterpret_cast<u32*>(park + G::kPark);
    u32x4* bm = park + (u32)TILE * kLdsChunksPerRec;  // row r: {S lo, S hi, C lo, C hi}
    const bool in = t.lane < t.cnt;
    const u32 my_rel = t.my_slot - base;
    // D register rows in rotation, D - 1 loads in flight; the first ones go out before the bitmaps are built
    constexpr u32 D = UNET_STREAM_DEPTH;
    const u32 nrows = (span + kWave - 1u) / kWave, lastc = span - 1u;
    const u32x4* p = reinterpret_cast<const u32x4*>(b.arena) + (u64)base;
#define UNET_STREAM_LD(R) ld_stream(p + min((R) * kWave + t.lane, lastc))
    u32x4 r[D];
#pragma unroll
    for (u32 k = 0; k + 1u < D; ++k) r[k] = UNET_STREAM_LD(k);
    static_assert(kStreamRows % D == 0u, "a bitmap segment is a whole number of rotations");
    if (t.lane < (u32)TILE) sums[t.lane] = 0u;
    const u32 my_cut = (t.my_len & 15u) != 0u ? my_rel + (t.my_len >> 4) : 0xffffffffu;
    static_assert(D % 2u == 0u, "the two bitmap-row registers alternate with the row's parity");
    u32 carry = 0, row = 0;
    u32 next_start = 0;     // where the next record begins (the range's end after the last one): wave-uniform
    u32 hdr_end = 0;        // where the header chunks of the running record end: wave-uniform
    u32 acc_run = 0;        // per lane: chunk sums of rows that lie wholly inside one record, not yet credited to it
    bool pending = false;   // wave-uniform: acc_run holds something
    u32x4 wv[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
#pragma nounroll
    while (true) {
        if (row % kStreamRows == 0u) {
            // S and C bits of the chunks [64 row, 64 row + 2048)
            if (t.lane < kStreamRows) bm[t.lane] = u32x4{0, 0, 0, 0};
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            u32* bits = reinterpret_cast<u32*>(bm);
            const u32 s_at = my_rel - row * kWave, c_at = my_cut - row * kWave;
            if (in && s_at < kStreamRows * kWave)
                __hip_atomic_fetch_or(bits + (s_at >> 6) * 4u + ((s_at >> 5) & 1u), 1u << (s_at & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (in && my_cut != 0xffffffffu && c_at < kStreamRows * kWave)
                __hip_atomic_fetch_or(bits + (c_at >> 6) * 4u + 2u + ((c_at >> 5) & 1u), 1u << (c_at & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            wv[0] = bm[0];
        }
        bool done = false;
#pragma unroll
        for (u32 k = 0; k < D; ++k) {
            if (!done) {
                const u32 rr = row + k;
                r[(k + D - 1u) % D] = UNET_STREAM_LD(rr + D - 1u);
                // the next row's bits, ahead of time (stale at a segment's end: re-read above)
                wv[(k + 1u) & 1u] = bm[(rr + 1u) % kStreamRows];
                if (rr != 0u && next_start > (rr + 1u) * kWave && rr * kWave >= hdr_end) {
                    // the whole row, and the chunk after it, lie inside the record that is already running, past its
                    // header chunks: nothing to park, nothing cut, no boundary
                    acc_run = sum_chunk(r[k], acc_run);
                    pending = true;
                } else {
                    if (pending) {
                        const u32 whole_rows = wave_sum(acc_run);
                        if (t.lane == 0u) __hip_atomic_fetch_add(sums + carry - 1u, whole_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        acc_run = 0u;
                        pending = false;
                    }
                    stream_row<TILE, PARK_HDR>(r[k], wv[k & 1u], rr, span, lead, t, my_rel, park, sums, carry);
                    next_start = carry < t.cnt ? rdlane(my_rel, carry) : span;
                    if (PARK_HDR && carry != 0u) hdr_end = rdlane(my_rel, carry - 1u) + kHdrChunks;
                }
                done = rr + 1u >= nrows;  // (the last row holds the range's end: never skipped above, so nothing is pending here)
            }
        }
        if (done) break;
        row += D;
    }
#undef UNET_STREAM_LD
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u32x4* mine = park + (t.lane % G::kPass) * kLdsChunksPerRec;
    t.my_hdr = mine;
    t.hdr_dwords = kHdrChunks * 4u;
    u32 total = in ? sums[t.lane] : 0u;
    if (in && (t.my_len & 15u) != 0u) total = sum_chunk(LaneTail(t.my_len).apply(mine[kHdrChunks]), total);
    return total;
}

// General route, one pass: exact sum of the lane's own record if it belongs to this pass (lanes pass*kPass ..
// pass*kPass + kPass - 1), 0 otherwise.  On return t.my_hdr points at the lane's own parked chunks 0..5 (if
// PARK_HDR; stale bytes past the record's end).
template <int TILE, bool PARK_HDR>
__device__ __forceinline__ u32 tile_sums(const Batch& b, Tile& t, u32 pass, u32x4* __restrict__ park, bool* mine_now) {
    typedef TileGeom<TILE> G;
    constexpr u32 PASS = G::kPass;
    u32x4* mine = park + (t.lane % PASS) * kLdsChunksPerRec;
    const bool now = t.lane < t.cnt && (G::kPasses == 1u || t.lane / PASS == pass);
    *mine_now = now;
    const bool ok = t.my_ok && now;
    const u32 my_nch = ok ? (t.my_len + 15u) >> 4 : 0u;
    t.my_hdr = mine;
    t.hdr_dwords = kHdrChunks * 4u;
    const bool my_small = ok && my_nch >= 1u && my_nch <= kSmallChunks;
    const bool my_mid = UNET_MID_PATH && ok && my_nch > kSmallChunks && my_nch <= kMidChunks;
    const bool my_row4 = my_mid && my_nch <= kRow4Chunks;
    const bool my_row8 = my_mid && !my_row4 && my_nch <= kRow8Chunks;
    const bool my_big = ok && my_nch > kSmallChunks && !my_mid;
    const LaneTail tail(t.my_len);
    const bool my_partial = (t.my_len & 15u) != 0u;
    u32 my_total = 0;

    // ---- mid-size records (up to 1 KiB), four at a time (its register buffers are free again afterwards)
    u32* sums = reinterpret_cast<u32*>(park + G::kPark);  // one u32 per record of the tile
    u32* list = sums + TILE;                                              // and one list entry
    if (UNET_MID_PATH) {
        if (kRow4Chunks > kSmallChunks) rows_path<4, PARK_HDR, PASS>(b, t, my_row4, list, park, sums);
        rows_path<8, PARK_HDR, PASS>(b, t, my_row8, list, park, sums);
        rows_path<16, PARK_HDR, PASS>(b, t, my_mid && !my_row4 && !my_row8, list, park, sums);
    }

    // ---- streamed records: their first loads go out before the lane-parallel part
    u64 todo = __ballot(my_big);
#ifndef UNET_DEPTH
#define UNET_DEPTH 3  // register buffers in rotation: two records' loads in flight while a third is summed
#endif
    Rows2 A, B;
#if UNET_DEPTH == 3
    Rows2 C;
#endif
    // next record to stream, or a dummy (same number of load instructions) once the list is empty
#define UNET_NEXT(R)                                                                 \
    {                                                                                \
        const bool has = todo != 0ull;                                               \
        const u32 j = has ? (u32)__builtin_ctzll(todo) : 0u;                         \
        todo &= todo - 1ull;                                                         \
        R = issue_rows(b, t.my_len, t.my_slot, t.lane, j, has);                      \
    }
    const bool any_big = todo != 0ull;
    if (any_big) {
        UNET_NEXT(A)
#if UNET_DEPTH == 3
        UNET_NEXT(B)
#endif
    }

    // ---- lane-parallel records: each lane pulls its own 1..kSmallChunks chunks.  Straight-line code per group
    // of four: unconditional loads (indices past the end re-read the last chunk and are zeroed by
    // selects), so all of them are in flight together and no wide value is merged across branches.
    if (my_small) {
        const u32x4* p = reinterpret_cast<const u32x4*>(b.arena + (u64)t.my_slot * 16u);
        const u32 last = my_nch - 1u;
        const u32 k0 = my_partial ? tail.m0 : 0xffffffffu, k1 = my_partial ? tail.m1 : 0xffffffffu;
        const u32 k2 = my_partial ? tail.m2 : 0xffffffffu, k3 = my_partial ? tail.m3 : 0xffffffffu;
        u32 acc = 0;
#define UNET_SMALL_LOAD(i) const u32x4 s##i = ld_small(p + min((u32)(i), last))
#define UNET_SMALL_USE(i)                                                           \
        {                                                                           \
            if (PARK_HDR && (i) < kHdrChunks) UNET_LDS_STORE(park, (t.lane % PASS) * kLdsChunksPerRec + (i), PASS * kLdsChunksPerRec, s##i); \
            const bool in = (u32)(i) <= last, cut = (u32)(i) == last;               \
            u32x4 v = zero_unless(s##i, in);                                        \
            v.x &= cut ? k0 : 0xffffffffu;                                          \
            v.y &= cut ? k1 : 0xffffffffu;                                          \
            v.z &= cut ? k2 : 0xffffffffu;                                          \
            v.w &= cut ? k3 : 0xffffffffu;                                          \
            acc = sum_chunk(v, acc);                                                \
        }
        UNET_SMALL_LOAD(0); UNET_SMALL_LOAD(1); UNET_SMALL_LOAD(2); UNET_SMALL_LOAD(3);
        if (last >= 4u) {  // 65..128 bytes
            UNET_SMALL_LOAD(4); UNET_SMALL_LOAD(5); UNET_SMALL_LOAD(6); UNET_SMALL_LOAD(7);
            UNET_SMALL_USE(4) UNET_SMALL_USE(5) UNET_SMALL_USE(6) UNET_SMALL_USE(7)
        }
#if UNET_SMALL_CHUNKS > 8
        if (last >= 8u) {
            UNET_SMALL_LOAD(8); UNET_SMALL_LOAD(9); UNET_SMALL_LOAD(10); UNET_SMALL_LOAD(11);
            UNET_SMALL_USE(8) UNET_SMALL_USE(9) UNET_SMALL_USE(10) UNET_SMALL_USE(11)
        }
#endif
#if UNET_SMALL_CHUNKS > 12
        if (last >= 12u) {
            UNET_SMALL_LOAD(12); UNET_SMALL_LOAD(13); UNET_SMALL_LOAD(14); UNET_SMALL_LOAD(15);
            UNET_SMALL_USE(12) UNET_SMALL_USE(13) UNET_SMALL_USE(14) UNET_SMALL_USE(15)
        }
#endif
        UNET_SMALL_USE(0) UNET_SMALL_USE(1) UNET_SMALL_USE(2) UNET_SMALL_USE(3)
#undef UNET_SMALL_LOAD
#undef UNET_SMALL_USE
        my_total = acc;
    }

    // ---- stream the long records; the register buffers are used in rotation (no register copies, so
    // nothing forces an early wait) and the loads of the next record(s) are in flight while one is summed
#define UNET_CONSUME(R)                                                              \
    {                                                                                \
        const u32 s_ = stream_one<PARK_HDR, PASS>(R, t.lane, park);                        \
        if (t.lane == R.j) my_total = s_;                                            \
    }
    if (any_big) {
#pragma nounroll
        while (true) {
#if UNET_DEPTH == 3
            UNET_NEXT(C)
            UNET_CONSUME(A)
            if (!B.live) break;
            UNET_NEXT(A)
            UNET_CONSUME(B)
            if (!C.live) break;
            UNET_NEXT(B)
            UNET_CONSUME(C)
            if (!A.live) break;
#else
            UNET_NEXT(B)
            UNET_CONSUME(A)
            if (!B.live) break;
            UNET_NEXT(A)
            UNET_CONSUME(B)
            if (!A.live) break;
#endif
        }
    }
#undef UNET_NEXT
#undef UNET_CONSUME

    // ---- LDS written by other lanes of this wave is read back below and by the caller
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (my_mid) my_total = sums[t.lane];
    if ((my_big || my_mid) && my_partial) {
        // parked by the wave while it held the record's last row
        my_total = sum_chunk(tail.apply(mine[kHdrChunks]), my_total);
    }
    return my_total;
}

// Per wave: the park area, then TILE u32 sums and TILE u32 list entries.
#define UNET_TILE_LDS(TILE)                                                                         \
    constexpr u32 kWaveLds = TileGeom<TILE>::kWaveLds;                                              \
    __shared__ u32x4 lds[(kBlock / kWave) * kWaveLds];                                              \
    u32x4* park = lds + (threadIdx.x >> 6) * kWaveLds

// General route of the 64-record tiles, which exist for batches of short records (the host picks them only when
// it has reason to expect such a batch): a tile that cannot take the span route -- a long record in it, records
// scattered over the arena -- is read by plain per-lane loops, 32 records per pass, four loads in flight per
// lane.  Any length is handled; it is simply not fast for long records, and it is cheap in registers, which is
// what lets these kernels keep eight workgroups on a CU for the route they are meant for.
template <int TILE, bool PARK_HDR>
__device__ __forceinline__ u32 lane_loop_pass(const Batch& b, Tile& t, u32 pass, u32x4* __restrict__ park, bool* mine_now) {
    constexpr u32 PASS = TileGeom<TILE>::kPass;
    u32x4* mine = park + (t.lane % PASS) * kLdsChunksPerRec;
    const bool now = t.lane < t.cnt && (TileGeom<TILE>::kPasses == 1u || t.lane / PASS == pass);
    *mine_now = now;
    t.my_hdr = mine;
    t.hdr_dwords = kHdrChunks * 4u;
    const bool ok = t.my_ok && now;
    const u32 nch = ok ? (t.my_len + 15u) >> 4 : 0u;
    const u32 nfull = t.my_len >> 4;
    const bool partial = (t.my_len & 15u) != 0u;
    const LaneTail tail(t.my_len);
    const u32x4* p = reinterpret_cast<const u32x4*>(b.arena + (ok ? (u64)t.my_slot * 16u : 0ull));
    const u32 last = nch != 0u ? nch - 1u : 0u;
    u32 acc = 0;
#pragma nounroll
    for (u32 c = 0; c < nch; c += 4u) {
        const u32x4 v0 = ld_small(p + min(c, last)), v1 = ld_small(p + min(c + 1u, last));
        const u32x4 v2 = ld_small(p + min(c + 2u, last)), v3 = ld_small(p + min(c + 3u, last));
        const u32 at = (t.lane % PASS) * kLdsChunksPerRec;
        constexpr u32 kLimit = PASS * kLdsChunksPerRec;
        if (PARK_HDR && c == 0u) {
            UNET_LDS_STORE(park, at, kLimit, v0);
            UNET_LDS_STORE(park, at + 1u, kLimit, v1);
            UNET_LDS_STORE(park, at + 2u, kLimit, v2);
            UNET_LDS_STORE(park, at + 3u, kLimit, v3);
        }
        if (PARK_HDR && c == 4u) {
            UNET_LDS_STORE(park, at + 4u, kLimit, v0);
            UNET_LDS_STORE(park, at + 5u, kLimit, v1);
        }
#define UNET_LOOP_USE(V, I)                                                         \
        {                                                                           \
            const u32 k_ = c + (I);                                                 \
            const bool whole_ = k_ < nfull, cut_ = partial && k_ == nfull;          \
            const u32x4 m_ = tail.apply(V);                                         \
            u32x4 u_;                                                               \
            u_.x = whole_ ? V.x : (cut_ ? m_.x : 0u);                               \
            u_.y = whole_ ? V.y : (cut_ ? m_.y : 0u);                               \
            u_.z = whole_ ? V.z : (cut_ ? m_.z : 0u);                               \
            u_.w = whole_ ? V.w : (cut_ ? m_.w : 0u);                               \
            acc = sum_chunk(u_, acc);                                               \
        }
        UNET_LOOP_USE(v0, 0u) UNET_LOOP_USE(v1, 1u) UNET_LOOP_USE(v2, 2u) UNET_LOOP_USE(v3, 3u)
#undef UNET_LOOP_USE
    }
    // LDS written here is read back by the caller
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return acc;
}

// Works through a tile: fin(total, mine_now) is called once per pass by the whole wave, with `mine_now` set for
// the lanes whose record was handled in that pass, `total` = that record's exact byte sum and t.my_hdr = its first
// chunks in LDS.  The LDS area is reused by the next pass.  The two routes are kept apart (rather than chosen
// inside one loop) so that the staged rows of the span route never share registers' lifetimes with the rotation
// buffers of the general route.
template <int TILE, bool PARK_HDR, class Fin>
__device__ __forceinline__ void tile_run(const Batch& b, Tile& t, u32x4* __restrict__ park, Fin&& fin) {
    if (UNET_SPAN_PATH && (TILE == 64 || !UNET_SPAN_ONLY64)) {
        const SpanPlan sp = plan_span<TILE>(b, t, t.my_ok ? (t.my_len + 15u) >> 4 : 0u);
        if (sp.on) {
            const u32 total = span_sums<TILE>(sp, t, park);
            fin(total, t.lane < t.cnt);
            return;
        }
    }
    if constexpr (TileGeom<TILE>::kStream) {
        u32 base, span, lead;
        if (plan_stream<TILE>(t, t.my_ok ? (t.my_len + 15u) >> 4 : 0u, &base, &span, &lead)) {
            const u32 total = stream_sums<TILE, PARK_HDR>(b, t, base, span, lead, park);
            fin(total, t.lane < t.cnt);
            return;
        }
    }
#pragma nounroll
    for (u32 pass = 0; pass < TileGeom<TILE>::kPasses; ++pass) {
        bool mine_now;
        u32 total;
        if constexpr (TileGeom<TILE>::kPasses > 1u) total = lane_loop_pass<TILE, PARK_HDR>(b, t, pass, park, &mine_now);
        else total = tile_sums<TILE, PARK_HDR>(b, t, pass, park, &mine_now);
        fin(total, mine_now);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------- a1: calculate_checksum
// out[i] = calculate_checksum(record i, seeds[i])  (src/utils.rs:1-17)
template <int TILE>
__global__ __launch_bounds__(kBlock, (TILE == 64 ? UNET_OCC64 : 6) * (256 / UNET_BLOCK)) void k_csum(Batch b, const uint16_t* __restrict__ seeds,
                                                    uint16_t* __restrict__ out) {
    static_assert(TILE >= 1 && TILE <= 64, "one lane per record of the tile");
    UNET_TILE_LDS(TILE);
    Tile t;
    if (!t.init<TILE>(b)) return;
    u32 my_seed = 0;
    if (seeds && t.lane < t.cnt) my_seed = seeds[t.first + t.lane];
    u32 res = 0;
    tile_run<TILE, false>(b, t, park, [&](u32 total, bool mine_now) __attribute__((always_inline)) {
        // the seed is an already-folded partial sum in the big-endian domain (src/utils.rs:2)
        if (mine_now && t.my_ok) res = finish(total + bswap16(my_seed));
    });
    if (t.lane < t.cnt) out[t.first + t.lane] = (uint16_t)res;
}

// ------------------------------------------- a2-a6, a8, a11: receive-path verdicts
template <int MODE, int TILE>
__global__ __launch_bounds__(kBlock, (TILE == 64 ? UNET_OCC64 : 6) * (256 / UNET_BLOCK)) void k_verify(Batch b, IfaceDev ifc, u32* __restrict__ out) {
    static_assert(TILE >= 1 && TILE <= 64, "one lane per record of the tile");
    UNET_TILE_LDS(TILE);
    Tile t;
    if (!t.init<TILE>(b)) return;
    u32 res = 0;  // V_NONE for records whose length/offset is out of range
    tile_run<TILE, true>(b, t, park, [&](u32 total, bool mine_now) __attribute__((always_inline)) {
        if (mine_now && t.my_ok) {
            LdsFetch f;
            f.w = reinterpret_cast<const u32*>(t.my_hdr);
            const DeepFetch deep{f.w, reinterpret_cast<const u32*>(b.arena + (u64)t.my_slot * 16u), t.hdr_dwords};
            const Parsed P = parse_record<MODE>(f, deep, t.my_len, ifc);
            res = P.kind == K_DONE ? P.res : finalize_record(P, total);
        }
    });
    if (t.lane < t.cnt) out[t.first + t.lane] = res;
}

// ------------------------------------------- a few long records: one wave per 4 KiB segment
// One wave per record leaves a batch of a few jumbo frames (the loopback device carries up to 65535 bytes,
// src/devices/loopback.rs:16) on a few waves, each pulling its record through row by row.  Here every segment of
// kSegChunks chunks gets a wave of its own: wave (i, s) sums chunks [s * kSegChunks, (s + 1) * kSegChunks) of record i
// -- exact u32 sums, segments start on 16-byte boundaries, so the segments' sums simply add up -- and leaves the
// result in partial[i * max_segs + s]; k_verify_presummed then judges record i from its header bytes and that total.
constexpr u32 kSegChunks = 256;  // 4 KiB

__global__ __launch_bounds__(kBlock) void k_segment_sums(Batch b, u32 max_segs, u32* __restrict__ partial) {
    const u32 lane = threadIdx.x & 63u;
    const u32 w = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const u32 i = w / max_segs, sg = w - i * max_segs;
    if (i >= b.n) return;
    const u32 len = b.len[i];
    const u64 slot = b.off16 ? (u64)b.off16[i] : (u64)i * b.stride16;
    const bool ok = len <= kMaxLen && slot <= 0xffffffffull && slot * 16u + (((u64)len + 15u) & ~15ull) <= b.arena_bytes;
    const u32 nch = ok ? (len + 15u) >> 4 : 0u;
    const u32 lo = sg * kSegChunks;
    u32 acc = 0;
    if (lo < nch) {
        const u32 hi = min(nch, lo + kSegChunks);  // chunks [lo, hi) are this wave's
        const u32x4* p = reinterpret_cast<const u32x4*>(b.arena) + slot;
        const TailMask tm(len, nch);
        // the four rows of a segment, requested together (clamped to the segment's last chunk, then left out)
        const u32 last = hi - 1u;
        const u32 c0 = lo + lane, c1 = c0 + kWave, c2 = c1 + kWave, c3 = c2 + kWave;
        const u32x4 v0 = ld_stream(p + min(c0, last)), v1 = ld_stream(p + min(c1, last));
        const u32x4 v2 = ld_stream(p + min(c2, last)), v3 = ld_stream(p + min(c3, last));
        acc = sum_chunk(tm.apply(zero_unless(v0, c0 < hi), c0), acc);
        acc = sum_chunk(tm.apply(zero_unless(v1, c1 < hi), c1), acc);
        acc = sum_chunk(tm.apply(zero_unless(v2, c2 < hi), c2), acc);
        acc = sum_chunk(tm.apply(zero_unless(v3, c3 < hi), c3), acc);
    }
    acc = wave_sum(acc);
    if (lane == 0u) partial[w] = acc;
}

// Dword k of a record in the arena (per-lane loads: only the few header dwords a verdict needs are touched).
struct GlobalFetch {
    const u32* w;
    __device__ __forceinline__ u32 operator()(u32 k) const { return w[k]; }
};

template <int MODE>
__global__ __launch_bounds__(kBlock) void k_verify_presummed(Batch b, IfaceDev ifc, const u32* __restrict__ partial, u32 max_segs,
                                                             u32* __restrict__ out) {
    const u32 i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= b.n) return;
    const u32 len = b.len[i];
    const u64 slot = b.off16 ? (u64)b.off16[i] : (u64)i * b.stride16;
    const bool ok = len <= kMaxLen && slot <= 0xffffffffull && slot * 16u + (((u64)len + 15u) & ~15ull) <= b.arena_bytes;
    u32 res = 0;  // V_NONE, as the tile kernels answer for a record outside the arena -- and for one longer than the segments cover
    if (ok && ((len + 15u) >> 4) <= max_segs * kSegChunks) {
        u32 total = 0;
        for (u32 s = 0; s < max_segs; ++s) total += partial[i * max_segs + s];
        const GlobalFetch f{reinterpret_cast<const u32*>(b.arena + slot * 16u)};
        const Parsed P = parse_record<MODE>(f, f, len, ifc);
        res = P.kind == K_DONE ? P.res : finalize_record(P, total);
    }
    out[i] = res;
}

// ------------------------------------------- the same verdicts for records of ONE length at a BYTE stride
// Records of `len` bytes, record i at byte i * stride of the buffer (stride >= len, both at most kBytesMaxLen): no
// 16-byte slots, no tables.  A record of 65 bytes costs 80 bytes of arena and 8 bytes of tables in the slot layout
// (1.35 x its own bytes); here it costs 65.  A wave takes the byte range of its records (at most 4 KiB, starting
// on a 128-byte line) into LDS with whole streaming 1 KiB rows, all requested up front; every lane then reads its
// own record out of LDS at its byte offset -- the LDS serves 4- and 16-byte reads at any byte address -- so sums and
// header fields are taken in the record's own alignment and the rest is the code of the slot kernels.
constexpr u32 kBytesMaxLen = 256;   // longest record (and stride)
constexpr u32 kBytesStage = 4096;   // record bytes one wave stages
constexpr u32 kBytesArea = (127u + kBytesStage + 16u + 15u) / 16u + 1u;  // chunks: lead, records, one chunk of read-ahead
constexpr u32 kBytesRows = (kBytesArea + kWave - 1u) / kWave;            // 5

// Dword k / chunk k of a record that starts at any byte address of LDS.
struct LdsBytesFetch {
    const uint8_t* p;
    __device__ __forceinline__ u32 operator()(u32 k) const {
        u32 v;
        __builtin_memcpy(&v, p + 4u * k, 4);
        return v;
    }
    __device__ __forceinline__ u32x4 chunk(u32 k) const {
        u32x4 v;
        __builtin_memcpy(&v, p + 16u * k, 16);
        return v;
    }
};

// Records one wave takes: as many as fit its 4 KiB, whatever their number -- a lane works through records lane,
// lane + 64, ... of the wave's group (twenty-byte records: 204 to a wave, four to a lane at most).  Short records
// are latency-bound: what a wave has in flight is what counts, so it always asks for its whole 4 KiB at once.
__host__ __device__ inline u32 bytes_per_wave(u32 stride) { return stride == 0u ? 0u : kBytesStage / stride; }

template <int MODE>
__global__ __launch_bounds__(kBlock, 8) void k_verify_bytes(const uint8_t* __restrict__ buf, u32 stride, u32 len, u32 n, IfaceDev ifc,
                                                            u32* __restrict__ out) {
    __shared__ u32x4 lds[(kBlock / kWave) * kBytesArea];
    const u32 lane = threadIdx.x & 63u;
    u32x4* area = lds + (threadIdx.x >> 6) * kBytesArea;
    const u32 per = bytes_per_wave(stride);
    const u64 first = ((u64)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * per;
    if (first >= n) return;
    const u32 cnt = n - first < per ? (u32)(n - first) : per;
    const u64 b0 = first * stride;
    const u64 base = b0 & ~127ull;
    const u32 lead = (u32)(b0 - base);
    // chunks that hold bytes of these records (the host has checked that the buffer is readable that far)
    const u32 span = (lead + (cnt - 1u) * stride + len + 15u) >> 4;
    const u32x4* p = reinterpret_cast<const u32x4*>(buf) + (base >> 4);
    u32x4 r[kBytesRows];
#pragma unroll
    for (u32 k = 0; k < kBytesRows; ++k) {
        r[k] = u32x4{0, 0, 0, 0};
        if (k * kWave < span) r[k] = ld_stream(p + min(k * kWave + lane, span - 1u));  // wave-uniform: only the rows that exist
    }
#pragma unroll
    for (u32 k = 0; k < kBytesRows; ++k)
        if (k * kWave + lane < span) UNET_LDS_STORE(area, k * kWave + lane, kBytesArea, r[k]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u32 nfull = len >> 4;
    const LaneTail tail(len);
#pragma nounroll
    for (u32 i = lane; i < cnt; i += kWave) {
        const LdsBytesFetch f{reinterpret_cast<const uint8_t*>(area) + lead + i * stride};
        u32 total = 0;
#pragma unroll
        for (u32 k = 0; k < kBytesMaxLen / 16u; ++k) {
            if (k < nfull) total = sum_chunk(f.chunk(k), total);  // (wave-uniform: every record has the same length)
        }
        if ((len & 15u) != 0u) total = sum_chunk(tail.apply(f.chunk(nfull)), total);
        const Parsed P = parse_record<MODE>(f, f, len, ifc);
        out[first + i] = P.kind == K_DONE ? P.res : finalize_record(P, total);
    }
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

// Same verdicts for whole groups of 256 headers per wave (one wave per workgroup): the 5 KiB of a group are
// pulled with five fully coalesced, streaming 1 KiB loads, turned through the wave's LDS area so that
// each lane gets the five chunks of its own four headers, and answered with one coalesced 1 KiB store.
constexpr u32 kHdrPerWave = kWave * kHdrPerGroup;          // 256 headers
constexpr u32 kHdrChunksPerWave = kHdrPerWave * 20u / 16u;  // 320 chunks

#ifndef UNET_HDR_BLOCK
#define UNET_HDR_BLOCK 64   // one wave per workgroup: 6.04 vs 5.88 TB/s on 16.8 M headers (128: 5.93)
#endif
constexpr u32 kHdrBlock = UNET_HDR_BLOCK;  // threads per workgroup of the row kernel

__global__ __launch_bounds__(kHdrBlock) void k_ipv4_validate_rows(const u32x4* __restrict__ hdr_chunks, u32 ext_tcp,
                                                                  u32x4* __restrict__ out4) {
    __shared__ u32x4 lds[(kHdrBlock / kWave) * kHdrChunksPerWave];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    u32x4* mine = lds + wave * kHdrChunksPerWave;
    const u64 g = (u64)blockIdx.x * (kHdrBlock / kWave) + wave;
    const u32x4* src = hdr_chunks + g * kHdrChunksPerWave + lane;
    u32x4 v[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = ld_stream(src + k * kWave);
#pragma unroll
    for (int k = 0; k < 5; ++k) mine[k * kWave + lane] = v[k];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u32x4 d[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) d[k] = mine[lane * 5u + k];
    u32x4 r;
    r.x = ipv4_header_verdict(d[0].x, d[0].y, d[0].z, d[0].w, d[1].x, ext_tcp);
    r.y = ipv4_header_verdict(d[1].y, d[1].z, d[1].w, d[2].x, d[2].y, ext_tcp);
    r.z = ipv4_header_verdict(d[2].z, d[2].w, d[3].x, d[3].y, d[3].z, ext_tcp);
    r.w = ipv4_header_verdict(d[3].w, d[4].x, d[4].y, d[4].z, d[4].w, ext_tcp);
    __builtin_nontemporal_store(r, out4 + g * kWave + lane);
}

// ------------------------------- a7, a9, a10: transmit-side checksum fill, in place
// status[i] = 0 when filled, else the reason the record was left alone.  The sums are taken over the
// packet as it stands; the stale contents of the checksum fields are subtracted exactly.
// IN_PLACE = false leaves the packets alone and writes the two values to fields[i] = ip_csum | l4_csum << 16
// instead (0 for a record that is not filled): a pure read pass.
template <int TILE, bool IN_PLACE>
__global__ __launch_bounds__(kBlock, (TILE == 64 ? UNET_OCC64 : 6) * (256 / UNET_BLOCK)) void k_generate(Batch b, uint8_t* __restrict__ arena_rw,
                                                        u32* __restrict__ fields, uint8_t* __restrict__ status) {
    static_assert(TILE >= 1 && TILE <= 64, "one lane per record of the tile");
    UNET_TILE_LDS(TILE);
    Tile t;
    if (!t.init<TILE>(b)) return;
    u32 st = V_DROP_L3_TRUNCATED;  // also reported for out-of-range lengths/offsets
    u32 my_fields = 0;
    tile_run<TILE, true>(b, t, park, [&](u32 total, bool mine_now) __attribute__((always_inline)) {
    if (mine_now && t.my_ok && t.my_len >= 20u) {
        // (locals, assigned to the captured results once at the end: writes through the captures from several
        // branches end up as stores to stack memory)
        u32 st_ = 0, fields_ = 0;
        const u32* w = reinterpret_cast<const u32*>(t.my_hdr);
        const u32 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
        const u32 proto = (w2 >> 8) & 0xffu;
        const u32 vhl = w0 & 0xffu;
        const u32 seglen = t.my_len - 20u;
        if (!(proto == 1u || proto == 17u)) st_ = V_DROP_L3_PROTO;
        else if ((vhl >> 4) != 4u) st_ = V_DROP_L3_VERSION;
        else if ((vhl & 0xfu) < 5u) st_ = V_DROP_L3_IHL_SHORT;
        else if ((vhl & 0xfu) > 5u) st_ = V_DROP_L3_IHL_LONG;
        else if (seglen < 8u) st_ = proto == 17u ? V_DROP_UDP_SHORT : V_DROP_ICMP_TRUNCATED;
        else {
            st_ = 0;
            // IPv4 header field zeroed, then summed (src/protocols/ipv4.rs:374-389)
            const u32 hdr = halves(w0) + halves(w1) + (w2 & 0xffffu) + halves(w3) + halves(w4);
            const u32 prefix = hdr + (w2 >> 16);
            u32 w5 = w[5], w6 = w[6];
            u32 l4ck;
            if (proto == 17u) {
                // s = ck(pseudo, 0); c = ck(hdr(cksum 0) || payload, !s)  (src/transport/udp.rs:200-203)
                const u32 pseudo = halves(w3) + halves(w4) + 0x1100u + bswap16(seglen);
                l4ck = finish(total - prefix - (w6 >> 16) + pseudo);
                w6 = (w6 & 0xffffu) | (bswap16(l4ck) << 16);  // segment bytes 6..7 = packet bytes 26..27
            } else {
                // ck(hdr(cksum 0) || data, 0)  (src/transport/icmp.rs:81-85)
                l4ck = finish(total - prefix - (w5 >> 16));
                w5 = (w5 & 0xffffu) | (bswap16(l4ck) << 16);  // message bytes 2..3 = packet bytes 22..23
            }
            const u32 ipck = finish(hdr);
            fields_ = ipck | (l4ck << 16);
            // Both fields live in the packet's first 32 bytes: store those two chunks whole instead of
            // two 2-byte pokes.  (Measured: the kernel without these stores runs at 6.9 TB/s, with them
            // at 4.7 TB/s -- one partial-line write per record into the stream being read costs a
            // read-modify-write in the memory system whatever its width.)
            if (IN_PLACE) {
                u32x4* rec = reinterpret_cast<u32x4*>(arena_rw + (u64)t.my_slot * 16u);
                rec[0] = u32x4{w0, w1, (w2 & 0xffffu) | (bswap16(ipck) << 16), w3};
                rec[1] = u32x4{w4, w5, w6, w[7]};
            }
        }
        st = st_;
        my_fields = fields_;
    }
    });
    if (!IN_PLACE && t.lane < t.cnt) fields[t.first + t.lane] = my_fields;
    if (status && t.lane < t.cnt) status[t.first + t.lane] = (uint8_t)st;
}

// ------------------- transmit path, whole packets: udp::send / icmp::send -> ipv4::send -> tap::send
struct TxDesc {  // mirrors unet_tx_desc (28 bytes, 4-byte aligned)
    u32 src, dst;
    u32 l4ab;      // l4a | l4b << 16
    u32 values;
    u32 id_proto;  // ip_id | proto << 16
    u32 mac_lo;    // dst_mac[0..3]
    u32 mac_hi;    // dst_mac[4..5]
};
static_assert(sizeof(TxDesc) == 28, "descriptor layout");

// 16 output bytes = bytes R..15 of `lo` followed by bytes 0..R-1 of `hi` (R = 4 or 6).
template <int R>
__device__ __forceinline__ u32x4 splice(u32x4 lo, u32x4 hi) {
    if (R == 4) return u32x4{lo.y, lo.z, lo.w, hi.x};
    return u32x4{(lo.y >> 16) | (lo.z << 16), (lo.z >> 16) | (lo.w << 16), (lo.w >> 16) | (hi.x << 16),
                 (hi.x >> 16) | (hi.y << 16)};
}

__device__ __forceinline__ void tx_store(u32x4 v, u32x4* p) {
#ifdef UNET_TX_DEFAULT_STORE
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}

struct TxRows {
    u32x4 r0, r1;
    const u32x4* p;
    u32 flen, nch, j;
    bool live;
};

// Short packets are stored chunk by chunk by their own lane; neighbouring lanes' packets share 128-byte
// lines, so these stores must stay in L2 until the line is complete (default policy: 4.2 TB/s on 18-byte
// payloads; streaming stores of partial lines: 3.5 TB/s there and 0.7 TB/s on 128-byte payloads).
__device__ __forceinline__ void tx_store_small(u32x4 v, u32x4* p) { *p = v; }

// Where the single-lane and lane-group paths put a packet's output chunk m: into the output arena, or -- when the tile's
// packets lie back to back there -- into LDS at the chunk's place in the tile's output range, which then leaves as whole
// 1 KiB rows (tx_flush_staged).  A tile that is longer than the wave's staging area stages as many of its first
// packets as fit and stores the rest directly.  `staged` is wave-uniform.
struct TxSink {
    uint8_t* out_arena;
    u32x4* area;   // LDS
    u32 base;      // first chunk of the staged range (on a 128-byte line)
    u32 limit;     // chunks the area holds
    u32 span;      // chunks of the staged range: the tile's output, or as many of its first packets as the area holds
    bool staged;
    // is a packet of `nout` output chunks at `oslot` inside the staged range?
    __device__ __forceinline__ bool holds(u32 oslot, u32 nout) const { return staged && oslot - base + nout <= span; }
    __device__ __forceinline__ void store(u32 oslot, u32 m, u32x4 v, bool in_stage) const {
        if (in_stage) UNET_LDS_STORE(area, oslot - base + m, limit, v);
        else tx_store_small(v, reinterpret_cast<u32x4*>(out_arena + (u64)oslot * 16u) + m);
    }
};

// Chunks [lead, span) of the staged range, row by row: whole lines with streaming stores, the range's first and last
// line -- shared with the neighbouring tiles when they start or end inside them -- with the default policy.
__device__ __forceinline__ void tx_flush_staged(const TxSink& k, u32 lane, u32 lead) {
    const u32 span = k.span;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u32x4* q = reinterpret_cast<u32x4*>(k.out_arena + (u64)k.base * 16u);
    const u32 last_line = (span - 1u) >> 3;
    const bool cut_first = lead != 0u, cut_last = (span & 7u) != 0u;
#pragma nounroll
    for (u32 c = lane; c < span; c += kWave) {
        if (c < lead) continue;
        const u32x4 v = k.area[c];
        const u32 line = c >> 3;
        if ((cut_first && line == 0u) || (cut_last && line == last_line)) q[c] = v;
        else tx_store(v, q + c);
    }
}

// Rows 0/1 of a payload, loaded SH lanes to the right: lane L holds payload chunk L - SH (row 0) and
// L + 64 - SH (row 1), so that lane L ends up owning output chunk L (L + 64) of the packet.  Exactly two
// unconditional, clamped load instructions, as in issue_rows.
template <u32 SH>
__device__ __forceinline__ TxRows issue_rows_shifted(const Batch& b, u32 my_len, u32 my_slot, u32 lane, u32 j, bool live) {
    TxRows x;
    x.j = j;
    x.live = live;
    x.flen = rdlane(my_len, j);
    const u32 slot = rdlane(my_slot, j);
    x.nch = live ? (x.flen + 15u) >> 4 : 0u;
    x.p = reinterpret_cast<const u32x4*>(b.arena + (live ? (u64)slot * 16u : 0ull));
    const u32 last = x.nch != 0u ? x.nch - 1u : 0u;
    x.r0 = ld_row(x.p + min(lane < SH ? 0u : lane - SH, last));
    x.r1 = ld_row(x.p + min(lane + kWave - SH, last));
    return x;
}

// Where tx_build_one's lanes 0..SH-1 take their header chunk from: parked in LDS by the packet's lane (tile kernel),
// or in every lane's registers (one packet per wave).
struct TxHdrLds {
    const u32x4* p;
    __device__ __forceinline__ u32x4 operator()(u32 lane) const { return p[lane]; }
};
template <u32 SH>
struct TxHdrRegs {
    u32x4 c[SH];
    __device__ __forceinline__ u32x4 operator()(u32 lane) const {
        u32x4 v = c[0];
#pragma unroll
        for (u32 k = 1; k < SH; ++k) v = pick_chunk(lane == k, c[k], v);
        return v;
    }
};

// One packet as the row code sees it: payload chunks in, output chunks out.
struct TxPkt {
    const u32x4* p;   // payload
    u32x4* pout;      // packet
    u32 nin, nout;    // payload chunks, output chunks the stores may cover
    u32 last;         // index of the last payload chunk (0 for an empty payload)
    TailMask tm;
    __device__ __forceinline__ TxPkt(const u32x4* p_, u32 payload_len, u32 nin_, u32 store_len, u32x4* pout_)
        : p(p_), pout(pout_), nin(nin_), nout((store_len + 15u) >> 4), last(nin_ != 0u ? nin_ - 1u : 0u), tm(payload_len, nin_) {}
    // payload chunk k as it enters the packet: zero past the payload's end, the cut last chunk masked
    __device__ __forceinline__ u32x4 chunk(u32 k) const { return tm.apply(zero_unless(ld_stream(p + min(k, last)), k < nin), k); }
};

// Payload chunks [k, k_end) of a packet, k on a row boundary of the output (k + SH a multiple of 64): rows of 64 chunks,
// four rows requested together with the first chunk of the row after them (one address for the whole wave), then each
// row spliced with the first dwords of the next and stored; the last step asks only for the rows that exist.  The loads
// are clamped to the payload's last chunk and zeroed past its end.  more_behind: the packet goes on past k_end (in
// another wave's hands).  Returns acc + the lanes' sums over these chunks.
template <int R, u32 SH, int BURST = 4>
__device__ __forceinline__ u32 tx_rows(const TxPkt& t, u32 lane, u32 k, u32 k_end, u32 acc, bool more_behind) {
#define UNET_TX_ROW(X, K) X = t.chunk((K) + lane)
#define UNET_TX_EMIT(X, NEXT, K)                                                                  \
    {                                                                                             \
        acc = sum_chunk(X, acc);                                                                  \
        u32x4 h_ = {0, 0, 0, 0};                                                                  \
        h_.x = from_next_lane(X.x, rdlane(NEXT.x, 0));                                            \
        if (R > 4) h_.y = from_next_lane(X.y, rdlane(NEXT.y, 0));                                 \
        if ((K) + lane + SH < t.nout) tx_store(splice<R>(X, h_), t.pout + (K) + lane + SH);       \
    }
    if (BURST == 2) {
        // the tile kernel's form: two rows at a time (eight registers fewer, which is its sixth wave per SIMD)
        u32x4 ra, rb;
#pragma nounroll
        for (; k + 2u * kWave <= k_end; k += 2u * kWave) {
            UNET_TX_ROW(ra, k);
            UNET_TX_ROW(rb, k + kWave);
            u32x4 pk = {0, 0, 0, 0};
            if (more_behind || k + 2u * kWave < k_end) pk = t.chunk(k + 2u * kWave);
            UNET_TX_EMIT(ra, rb, k)
            UNET_TX_EMIT(rb, pk, k + kWave)
        }
        if (k < k_end) {
            u32x4 pk = {0, 0, 0, 0};
            if (more_behind) pk = t.chunk(k + kWave);
            UNET_TX_ROW(ra, k);
            UNET_TX_EMIT(ra, pk, k)
        }
        return acc;
    }
    u32x4 ra, rb, rc, rd;
#pragma nounroll
    for (; k + 4u * kWave <= k_end; k += 4u * kWave) {
        UNET_TX_ROW(ra, k);
        UNET_TX_ROW(rb, k + kWave);
        UNET_TX_ROW(rc, k + 2u * kWave);
        UNET_TX_ROW(rd, k + 3u * kWave);
        const u32x4 pk = t.chunk(k + 4u * kWave);
        UNET_TX_EMIT(ra, rb, k)
        UNET_TX_EMIT(rb, rc, k + kWave)
        UNET_TX_EMIT(rc, rd, k + 2u * kWave)
        UNET_TX_EMIT(rd, pk, k + 3u * kWave)
    }
    if (k < k_end) {  // fewer than 256 chunks are left: only the rows that hold some of them are requested
        const bool has_b = k + kWave < k_end, has_c = k + 2u * kWave < k_end, has_d = k + 3u * kWave < k_end;
        // the chunk behind the last of these rows: zero when the packet ends here, another wave's when only the range does
        u32x4 pk = {0, 0, 0, 0};
        if (more_behind) pk = t.chunk(k + (has_d ? 4u : (has_c ? 3u : (has_b ? 2u : 1u))) * kWave);
        rb = rc = rd = pk;
        UNET_TX_ROW(ra, k);
        if (has_b) { UNET_TX_ROW(rb, k + kWave); }
        if (has_c) { UNET_TX_ROW(rc, k + 2u * kWave); }
        if (has_d) { UNET_TX_ROW(rd, k + 3u * kWave); }
        UNET_TX_EMIT(ra, rb, k)
        if (has_b) UNET_TX_EMIT(rb, rc, k + kWave)
        if (has_c) UNET_TX_EMIT(rc, rd, k + 2u * kWave)
        if (has_d) UNET_TX_EMIT(rd, pk, k + 3u * kWave)
    }
#undef UNET_TX_ROW
#undef UNET_TX_EMIT
    return acc;
}

// Rows 0/1 of a packet, stored whole once the payload's exact sum `total` has completed the L4 checksum (the last
// 16-bit word of virtual chunk SH-1 for UDP, the third-last for ICMP).  a, c: the masked payload chunks of rows 0/1 as
// loaded by issue_rows_shifted; fx, fy: first dwords of the payload chunk behind row 1.
template <int R, u32 SH, class Hdr>
__device__ __forceinline__ void tx_finish(const TxPkt& t, u32 lane, u32x4 a, u32x4 c, u32 fx, u32 fy, u32 total, const Hdr& hdr,
                                          u32 l4_partial_be, bool udp) {
    const u32 pay_be = bswap16(fold16(total));  // payload's folded sum, big-endian domain
    const u32 ck = bswap16((~fold16(l4_partial_be + pay_be)) & 0xffffu) << 16;
    if (lane < SH) a = hdr(lane);
    if (lane == SH - 1u) {
        a.w |= udp ? ck : 0u;
        a.z |= udp ? 0u : ck;
    }
    u32x4 ha = {0, 0, 0, 0}, hc = ha;
    ha.x = from_next_lane(a.x, rdlane(c.x, 0));
    hc.x = from_next_lane(c.x, fx);
    if (R > 4) {
        ha.y = from_next_lane(a.y, rdlane(c.y, 0));
        hc.y = from_next_lane(c.y, fy);
    }
    if (lane < t.nout) tx_store(splice<R>(a, ha), t.pout + lane);
    if (lane + kWave < t.nout) tx_store(splice<R>(c, hc), t.pout + lane + kWave);
}

constexpr u32 tx_rows01_chunks(u32 SH) { return 2u * kWave - SH; }  // payload chunks that rows 0/1 of a packet hold

// Builds one packet from the payload in `cur` and its prepared headers.  The packet is treated as one
// byte stream, headers then payload, cut into the payload's 16-byte chunks: "virtual" chunk m holds stream
// bytes [16m - R, 16m - R + 16), so chunks 0..SH-1 are the headers (L4 checksum still blank) and chunk SH + k is
// payload chunk k.  Output chunk m = bytes R..15 of virtual chunk m followed by bytes 0..R-1 of virtual
// chunk m + 1, which sits in the next lane's registers.  The part of a payload past rows 0/1 is streamed
// through first (tx_rows); rows 0/1 are stored last (tx_finish).
template <int R, u32 SH, int BURST, class Hdr>
__device__ __forceinline__ void tx_build_one(const TxRows& cur, u32 lane, u32 frame_len, u32x4* __restrict__ pout,
                                             const Hdr& hdr, u32 l4_partial_be, bool udp) {
    const TxPkt t(cur.p, cur.flen, cur.nch, frame_len, pout);
    const u32 k0 = lane - SH, k1 = lane + kWave - SH;  // k0 wraps for the header lanes: they hold no payload
    const u32x4 a = t.tm.apply(zero_unless(cur.r0, lane >= SH && k0 < t.nin), k0);
    const u32x4 c = t.tm.apply(zero_unless(cur.r1, k1 < t.nin), k1);
    u32 acc = sum_chunk(c, sum_chunk(a, 0u));
    constexpr u32 kRows = tx_rows01_chunks(SH);
    const u32 kend = max(t.nin, t.nout - SH);
    u32 fx = 0, fy = 0;  // first dwords of the chunk after row 1, wanted by lane 63 of row 1
    if (kRows < kend) {
        const u32x4 pk = t.chunk(kRows);
        fx = rdlane(pk.x, 0);
        fy = rdlane(pk.y, 0);
        acc = tx_rows<R, SH, BURST>(t, lane, kRows, kend, acc, false);
    }
    tx_finish<R, SH>(t, lane, a, c, fx, fy, wave_sum(acc), hdr, l4_partial_be, udp);
}

// A packet with a short payload, built by its own lane: the same virtual-chunk splice as tx_build_one, with every
// chunk of the packet in this lane's registers.  Up to 3 chunks (48 bytes) when the lane stores its chunks into the
// output arena itself -- the measured break-even with the 4-lane groups, whose stores fill lines better --, up to 5
// (80 bytes) when the tile's output is staged in LDS (64-byte payloads: 5.1 -> 5.8 TB/s against the groups).
// N: chunks the lane's registers hold (3 or 5); max_chunks (wave-uniform): the longest such payload of the tile --
// chunks past the third are only requested if some lane has them (18-byte payloads run 9 % slower with five
// unconditional loads), and a tile without them runs the three-chunk form (the IMIX payload mix 3 % faster).
constexpr u32 kTxSmallChunks = 3;
constexpr u32 kTxSmallStaged = 5;

template <int R, u32 SH, u32 N>
__device__ __forceinline__ void tx_build_small(const Batch& pay, u32 my_len, u32 my_slot, u32 frame_len, const TxSink& sink,
                                               u32 my_oslot, const u32x4* __restrict__ hdr, u32 l4_partial_be, bool udp,
                                               u32 max_chunks) {
    const u32 nin = (my_len + 15u) >> 4, last = nin != 0u ? nin - 1u : 0u;
    const u32 nout = (frame_len + 15u) >> 4;
    // an empty payload may sit at the very end of the arena: read the arena's first chunk instead
    const u32x4* p = reinterpret_cast<const u32x4*>(pay.arena + (nin != 0u ? (u64)my_slot * 16u : 0ull));
    u32x4 v[SH + N + 1u];
#pragma unroll
    for (u32 k = 0; k < N; ++k) {
        v[SH + k] = u32x4{0, 0, 0, 0};
        if (k < kTxSmallChunks || k < max_chunks) v[SH + k] = p[min(k, last)];
    }
    const LaneTail lt(my_len);
    const bool partial = (my_len & 15u) != 0u;
    u32 acc = 0;
#pragma unroll
    for (u32 k = 0; k < N; ++k) {
        u32x4 x = zero_unless(v[SH + k], k < nin);
        if (partial && k == last) x = lt.apply(x);
        v[SH + k] = x;
        acc = sum_chunk(x, acc);
    }
    v[SH + N] = u32x4{0, 0, 0, 0};
#pragma unroll
    for (u32 c = 0; c < SH; ++c) v[c] = hdr[c];
    const u32 pay_be = bswap16(fold16(acc));
    const u32 ck = bswap16((~fold16(l4_partial_be + pay_be)) & 0xffffu) << 16;
    v[SH - 1u].w |= udp ? ck : 0u;
    v[SH - 1u].z |= udp ? 0u : ck;
    const bool in_stage = sink.holds(my_oslot, nout);
#pragma unroll
    for (u32 c = 0; c < SH + N; ++c)
        if (c < nout) sink.store(my_oslot, c, splice<R>(v[c], v[c + 1u]), in_stage);
}

// Packets of at most 4*W chunks (payloads of up to 4*W - SH chunks), 64/W at a time, one per group of W
// (4, 8 or 16) neighbouring lanes: lane `sub` of a group holds virtual chunks sub, sub+W, sub+2W, sub+3W of
// its group's packet (header chunks from LDS, payload chunks loaded SH chunks further left), the payload's
// sum is reduced inside the group, and a chunk's neighbour comes from the next lane of the group -- or,
// for the group's last lane, from its first lane's chunk of the next step.  One step is in flight (UNET_TX_MID_DEPTH).
// `list` holds the tile's packets of this class, in lane order.
struct TxRowStep {
    u32x4 v0, v1, v2, v3;
    u32 rec, len, flen, oslot, partial;  // per group
    bool udp, row_live;                  // per group
};

template <u32 W, u32 SH>
__device__ __forceinline__ TxRowStep issue_tx_step(const Batch& pay, const Tile& t, u32 my_flen, u32 my_oslot, u32 my_partial,
                                                   bool my_udp, const u32* __restrict__ list, u32 first, u32 count) {
    const u32 sub = t.lane & (W - 1u), k = first + t.lane / W;
    TxRowStep g;
    g.row_live = k < count;
    g.rec = list[g.row_live ? k : 0u];
    g.len = from_lane(t.my_len, g.rec);
    g.flen = from_lane(my_flen, g.rec);
    g.oslot = from_lane(my_oslot, g.rec);
    g.partial = from_lane(my_partial, g.rec);
    g.udp = from_lane((u32)my_udp, g.rec) != 0u;
    const u32 slot = from_lane(t.my_slot, g.rec);
    const u32 nin = g.row_live ? (g.len + 15u) >> 4 : 0u;
    const u32 last = nin != 0u ? nin - 1u : 0u;
    // an empty payload may sit at the very end of the arena: read the arena's first chunk instead
    const u32x4* p = reinterpret_cast<const u32x4*>(pay.arena + (nin != 0u ? (u64)slot * 16u : 0ull));
    g.v0 = p[min(sub < SH ? 0u : sub - SH, last)];
    g.v1 = p[min(sub + W - SH, last)];
    g.v2 = p[min(sub + 2u * W - SH, last)];
    g.v3 = p[min(sub + 3u * W - SH, last)];
    return g;
}

// Lane i of a group of W receives lane i+1's value; the group's last lane receives its first lane's
// `nxt` (that lane's chunk of the next step).
template <u32 W>
__device__ __forceinline__ u32 from_next_in_group(u32 v, u32 nxt, bool last_of_group) {
    const u32 wrap = (u32)__builtin_amdgcn_update_dpp(0, (int)nxt, 0x110 + (W - 1u), 0xf, 0xf, false);  // row_shr:W-1
    const u32 next = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, false);               // row_shl:1
    return last_of_group ? wrap : next;
}

template <u32 W, int R, u32 SH>
__device__ __forceinline__ void consume_tx_step(const TxRowStep& g, u32 lane, const TxSink& sink, const u32x4* __restrict__ park) {
    const u32 sub = lane & (W - 1u);
    const u32 nin = g.row_live ? (g.len + 15u) >> 4 : 0u, last = nin != 0u ? nin - 1u : 0u;
    const u32 nout = g.row_live ? (g.flen + 15u) >> 4 : 0u;
    const LaneTail lt(g.len);
    const bool partial = (g.len & 15u) != 0u;
    // (named values, not an array: conditional updates of array elements end up in scratch memory)
    u32 acc = 0;
#define UNET_TXM_MASK(X, I)                                                              \
    {                                                                                    \
        const u32 m_ = W * (I) + sub, k_ = m_ - SH; /* k_ wraps for the header lanes */   \
        X = zero_unless(X, m_ >= SH && k_ < nin);                                        \
        const bool cut_ = partial && k_ == last;                                         \
        X.x &= cut_ ? lt.m0 : 0xffffffffu;                                               \
        X.y &= cut_ ? lt.m1 : 0xffffffffu;                                               \
        X.z &= cut_ ? lt.m2 : 0xffffffffu;                                               \
        X.w &= cut_ ? lt.m3 : 0xffffffffu;                                               \
        acc = sum_chunk(X, acc);                                                         \
    }
    u32x4 x0 = g.v0, x1 = g.v1, x2 = g.v2, x3 = g.v3;
    UNET_TXM_MASK(x0, 0u)
    UNET_TXM_MASK(x1, 1u)
    UNET_TXM_MASK(x2, 2u)
    UNET_TXM_MASK(x3, 3u)
#undef UNET_TXM_MASK
    const u32 total = group_sum<W>(acc);  // the payload's exact sum, in every lane of the group
    const u32 ck = bswap16((~fold16(g.partial + bswap16(fold16(total)))) & 0xffffu) << 16;
    const bool hdr_lane = g.row_live && sub < SH;
    const u32x4 hv = park[hdr_lane ? g.rec * SH + sub : 0u];
    x0.x = hdr_lane ? hv.x : x0.x;
    x0.y = hdr_lane ? hv.y : x0.y;
    x0.z = hdr_lane ? hv.z : x0.z;
    x0.w = hdr_lane ? hv.w : x0.w;
    const bool ck_lane = sub == SH - 1u;
    x0.w |= ck_lane && g.udp ? ck : 0u;
    x0.z |= ck_lane && !g.udp ? ck : 0u;
    const bool last_of_group = sub == W - 1u;
    const bool in_stage = sink.holds(g.oslot, nout);
#define UNET_TXM_EMIT(X, NX, NY, I)                                                      \
    {                                                                                    \
        u32x4 h_ = {0, 0, 0, 0};                                                         \
        h_.x = from_next_in_group<W>(X.x, NX, last_of_group);                            \
        if (R > 4) h_.y = from_next_in_group<W>(X.y, NY, last_of_group);                 \
        const u32 m_ = W * (I) + sub;                                                    \
        if (m_ < nout) sink.store(g.oslot, m_, splice<R>(X, h_), in_stage);              \
    }
    UNET_TXM_EMIT(x0, x1.x, x1.y, 0u)
    UNET_TXM_EMIT(x1, x2.x, x2.y, 1u)
    UNET_TXM_EMIT(x2, x3.x, x3.y, 2u)
    UNET_TXM_EMIT(x3, 0u, 0u, 3u)
#undef UNET_TXM_EMIT
}

template <u32 W, int R, u32 SH>
__device__ __forceinline__ void tx_rows_path(const Batch& pay, const Tile& t, bool mine_here, u32 my_flen, u32 my_oslot,
                                             u32 my_partial, bool my_udp, u32* __restrict__ list, const TxSink& sink,
                                             const u32x4* __restrict__ park) {
    constexpr u32 kPerStep = kWave / W;
    const u64 mask = __ballot(mine_here);
    if (mask == 0ull) return;
    const u32 count = (u32)__builtin_popcountll(mask);
    if (mine_here) list[__builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u))] = t.lane;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u32 next = 0;
#if UNET_TX_MID_DEPTH == 1
#pragma nounroll
    for (; next < count; next += kPerStep) {
        const TxRowStep g = issue_tx_step<W, SH>(pay, t, my_flen, my_oslot, my_partial, my_udp, list, next, count);
        consume_tx_step<W, R, SH>(g, t.lane, sink, park);
    }
#else
    TxRowStep g0 = issue_tx_step<W, SH>(pay, t, my_flen, my_oslot, my_partial, my_udp, list, next, count), g1;
    next += kPerStep;
#pragma nounroll
    while (true) {
        g1 = issue_tx_step<W, SH>(pay, t, my_flen, my_oslot, my_partial, my_udp, list, next, count);
        consume_tx_step<W, R, SH>(g0, t.lane, sink, park);
        if (next >= count) break;
        next += kPerStep;
        g0 = issue_tx_step<W, SH>(pay, t, my_flen, my_oslot, my_partial, my_udp, list, next, count);
        consume_tx_step<W, R, SH>(g1, t.lane, sink, park);
        if (next >= count) break;
        next += kPerStep;
    }
#endif
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the list is reused by the next class
    __builtin_amdgcn_wave_barrier();
}

// Chunks of LDS one wave of k_tx_build<MODE, TILE> has for staging its tile's output (0: that instantiation never
// stages), and the dynamic LDS a launch that may stage needs.
#ifndef UNET_TX_WAVE_BUDGET
#define UNET_TX_WAVE_BUDGET 640
#endif
constexpr u32 tx_stage_chunks(int mode, int tile) {
    return UNET_TX_STAGE && tile >= 8 ? (u32)UNET_TX_WAVE_BUDGET - (u32)tile * (mode == MODE_L2 ? 3u : 2u) - ((u32)tile + 3u) / 4u : 0u;
}
constexpr u32 tx_stage_bytes(int mode, int tile) { return (kBlock / kWave) * tx_stage_chunks(mode, tile) * 16u; }

// The headers of one packet as its first SH virtual chunks (see tx_build_one): little-endian halves of the wire
// bytes, R bytes to the right, the L4 checksum field still zero.  Returns the L4 checksum's partial sum (pseudo-header
// and L4 header for UDP, the ICMP header otherwise) in the big-endian domain.
template <int MODE>
__device__ __forceinline__ u32 tx_headers(const TxDesc& d, u32 payload_len, u32 proto, const IfaceDev& ifc,
                                          u32x4 (&chunk)[MODE == MODE_L2 ? 3 : 2]) {
    constexpr u32 SH = MODE == MODE_L2 ? 3u : 2u;
    constexpr int R = MODE == MODE_L2 ? 6 : 4;
    const u32 src = ifc.unicast, dst = d.dst;
    // create_ip_header, src/protocols/ipv4.rs:366-391
    const u32 ip[10] = {0x4500u, (20u + (8u + payload_len)) & 0xffffu, d.id_proto & 0xffffu, 0u, (64u << 8) | proto, 0u,
                        src >> 16, src & 0xffffu, dst >> 16, dst & 0xffffu};
    u32 ipsum = 0;
#pragma unroll
    for (int q = 0; q < 10; ++q) ipsum += ip[q];
    const u32 ipck = (~fold16(ipsum)) & 0xffffu;
    u32 l4[4], partial;
    if (proto == 17u) {
        // udp::send, src/transport/udp.rs:181-203: the pseudo-header uses the caller's src address
        const u32 length = (8u + payload_len) & 0xffffu;
        l4[0] = d.l4ab & 0xffffu;
        l4[1] = d.l4ab >> 16;
        l4[2] = length;
        l4[3] = 0u;
        partial = (d.src >> 16) + (d.src & 0xffffu) + (dst >> 16) + (dst & 0xffffu) + 17u + length + l4[0] + l4[1] + l4[2];
    } else {
        // icmp::send, src/transport/icmp.rs:75-85
        l4[0] = d.l4ab & 0xffffu;  // type << 8 | code
        l4[1] = 0u;
        l4[2] = d.values >> 16;
        l4[3] = d.values & 0xffffu;
        partial = l4[0] + l4[2] + l4[3];
    }
    u32 h[8 * SH];
    int n = 0;
#pragma unroll
    for (int q = 0; q < R / 2; ++q) h[n++] = 0u;
    if (MODE == MODE_L2) {
        h[n++] = d.mac_lo & 0xffffu;
        h[n++] = d.mac_lo >> 16;
        h[n++] = d.mac_hi & 0xffffu;
        h[n++] = ifc.mac_lo & 0xffffu;
        h[n++] = ifc.mac_lo >> 16;
        h[n++] = ifc.mac_hi & 0xffffu;
        h[n++] = 0x0008u;  // 08 00
    }
#pragma unroll
    for (int q = 0; q < 10; ++q) h[n++] = bswap16(q == 5 ? ipck : ip[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) h[n++] = bswap16(l4[q]);
#pragma unroll
    for (u32 c = 0; c < SH; ++c)
        chunk[c] = u32x4{h[8 * c] | (h[8 * c + 1] << 16), h[8 * c + 2] | (h[8 * c + 3] << 16), h[8 * c + 4] | (h[8 * c + 5] << 16),
                         h[8 * c + 6] | (h[8 * c + 7] << 16)};
    return partial;
}

// MODE_L3: packet = 28 header bytes + payload.  MODE_L2: 42 header bytes + payload (+ tap::send's pad).
// Lane j first prepares packet j's headers (every field and the IPv4 checksum; the L4 checksum as a
// partial sum) and parks them in LDS; the wave then builds the packets one after another, reading each
// payload once and writing each packet once, in whole 1 KiB rows from its first byte.
#ifndef UNET_TX_DEPTH
#define UNET_TX_DEPTH 3
#endif
#ifndef UNET_TX_OCC
#define UNET_TX_OCC 1   // workgroups per CU the register allocator must leave room for
#endif
template <int MODE, int TILE>
__global__ __launch_bounds__(kBlock, UNET_TX_OCC * (256 / UNET_BLOCK)) void k_tx_build(Batch pay, const TxDesc* __restrict__ desc, IfaceDev ifc,
                                                     uint8_t* __restrict__ out_arena, u64 out_bytes,
                                                     const u32* __restrict__ out_off16, u32* __restrict__ out_len,
                                                     uint8_t* __restrict__ status, u32 flags) {
    constexpr u32 H = MODE == MODE_L2 ? 42u : 28u;   // header bytes in front of the payload
    constexpr u32 SH = MODE == MODE_L2 ? 3u : 2u;    // output chunks that contain header bytes
    constexpr int R = MODE == MODE_L2 ? 6 : 4;       // payload bytes inside the last header chunk
    static_assert(16u * SH - H == (u32)R, "header geometry");
    // Staging area for the output of tiles of short packets (UNET_TX_STAGE): whatever is left of 10 KiB per wave, so
    // that four workgroups still share a CU's 160 KiB of LDS (three measured 10-20 % slower on every short size, staged
    // or not): 432-622 chunks, e.g. 64 packets of up to six or seven chunks, 32 of up to sixteen, 16 of up to 36
    // (548-byte payloads), less the seven chunks the range may start inside its line.  It lies in the launch's
    // dynamic LDS (tx_stage_bytes; kTxNoStage and none of it for batches that are not expected to stage: a mix of
    // IMIX payload sizes, whose tiles hardly ever qualify, runs 3 % slower with 40 KiB per workgroup reserved).
    constexpr u32 kStage = tx_stage_chunks(MODE, TILE);
    constexpr u32 kWaveLds = TILE * SH + (TILE + 3u) / 4u;  // header chunks, one list entry per packet
    __shared__ u32x4 lds[(kBlock / kWave) * kWaveLds];
    extern __shared__ u32x4 lds_stage[];
    u32x4* park = lds + (threadIdx.x >> 6) * kWaveLds;
    u32* list = reinterpret_cast<u32*>(park + TILE * SH);
    Tile t;
    if (!t.init<TILE>(pay)) return;
    const u32 i = t.first + t.lane;
    TxDesc d = {};
    u32 my_oslot = 0;
    if (t.lane < t.cnt) {
        d = desc[i];
        my_oslot = out_off16[i];
    }
    const u32 proto = (d.id_proto >> 16) & 0xffu;
    const u32 l3len = 28u + t.my_len;
    u32 frame_len = MODE == MODE_L2 ? 14u + max(l3len, 60u) : l3len;  // src/driver/tap.rs:147-152
    u32 st = UNET_TX_OK_;
    if (!t.my_ok) st = UNET_TX_E_RANGE_;
    else if (!(proto == 1u || proto == 17u)) st = UNET_TX_E_PROTO_;
    else if (t.my_len > 65535u - 20u - 8u) st = UNET_TX_E_TOO_LONG_;  // src/transport/udp.rs:174-180
    else if ((u64)my_oslot * 16u + (((u64)frame_len + 15u) & ~15ull) > out_bytes) st = UNET_TX_E_RANGE_;
    const bool my_go = t.lane < t.cnt && st == UNET_TX_OK_;
    // Chunks this packet's stores may cover, as a length: the packet itself, or -- when the caller has said that
    // nothing else lives between a packet's end and the next 128-byte line of the output arena (kTxFillLines) -- up
    // to that line, so that the packet's last line is written whole (zeros behind the packet) instead of in part.
    u32 store_len = frame_len;
    if ((flags & kTxFillLines) != 0u) {
        const u32 end = my_oslot + ((frame_len + 15u) >> 4);
        const u32 upto = (end + 7u) & ~7u;
        if (upto >= end && (u64)upto * 16u <= out_bytes) store_len = (upto - my_oslot) * 16u;
    }
    // payloads of up to 4*W - SH chunks fit the four steps of a group of W lanes; up to three chunks, one lane
    const bool fits_groups = UNET_TX_MID_PATH && my_go && t.my_len <= 16u * (kWave - SH);
    const bool small3 = UNET_TX_SMALL_PATH && my_go && t.my_len <= 16u * kTxSmallChunks;

    u64 todo = __ballot(my_go && !small3 && !fits_groups);
    constexpr int D = UNET_TX_DEPTH;  // register buffers in rotation: D - 1 payloads' loads in flight while one packet is built
    TxRows buf[D];
#define UNET_NEXT(X)                                                                 \
    {                                                                                \
        const bool has = todo != 0ull;                                               \
        const u32 j = has ? (u32)__builtin_ctzll(todo) : 0u;                         \
        todo &= todo - 1ull;                                                         \
        X = issue_rows_shifted<SH>(pay, t.my_len, t.my_slot, t.lane, j, has);        \
    }
    const bool any = todo != 0ull;
    // Output staging: every packet of the tile is built by one lane or a lane group, and their output chunks lie back to
    // back in the arena (any packed layout) within the staging area's size.
    TxSink sink{out_arena, lds_stage + (threadIdx.x >> 6) * kStage, 0u, kStage, 0u, false};
    u32 stage_lead = 0;
    const bool short_if_staged = UNET_TX_SMALL_PATH && my_go && t.my_len <= 16u * kTxSmallStaged;
    const u32 my_cover = ((short_if_staged ? frame_len : store_len) + 15u) >> 4;  // chunks this packet's stores cover
    if (kStage != 0u && !any && (flags & kTxNoStage) == 0u) {
        const bool in = t.lane < t.cnt, is_last = t.lane + 1u >= t.cnt;
        const u32 next_oslot = from_next_lane(my_oslot, 0xffffffffu);
        if (__ballot(!in || (my_go && (is_last || my_oslot + my_cover == next_oslot))) == ~0ull) {
            const u32 lo = rdlane(my_oslot, 0u);
            sink.base = lo & ~7u;
            stage_lead = lo & 7u;
            // the packets whose end lies inside the area: all of them, or -- tiles of 32 and 64 short packets, where
            // staging gains most; sixteen 600-byte payloads lose 2 % with fifteen staged -- the tile's first ones, at least
            // half of it
            const u64 inside = __ballot(in && my_oslot + my_cover - sink.base <= kStage);
            if (inside != 0ull) {
                const u32 last_in = 63u - (u32)__builtin_clzll(inside);
                sink.span = rdlane(my_oslot + my_cover, last_in) - sink.base;
                sink.staged = TILE >= 32 ? 2u * (last_in + 1u) >= t.cnt : last_in + 1u == t.cnt;
            }
        }
    }
    const bool my_small = sink.holds(my_oslot, my_cover) ? short_if_staged : small3;
    const bool my_mid = fits_groups && !my_small;
    const bool my_g4 = UNET_TX_G4 && my_mid && t.my_len <= 16u * (16u - SH);
    const bool my_g8 = my_mid && !my_g4 && t.my_len <= 16u * (32u - SH);
    const u64 mid_mask = __ballot(my_mid);
    const u32 small_chunks = __ballot(my_small && t.my_len > 64u) != 0ull ? 5u : (__ballot(my_small && t.my_len > 48u) != 0ull ? 4u : 3u);
#ifndef UNET_TX_EARLY_BIG
#define UNET_TX_EARLY_BIG 0   // k_tx_build: the first whole-wave payloads are requested ahead of the lane-group and single-lane packets too
#endif
    if (any && (UNET_TX_EARLY_BIG || mid_mask == 0ull)) {  // the first payloads are on their way while the headers are prepared
#pragma unroll
        for (int k = 0; k < D - 1; ++k) UNET_NEXT(buf[k])
    }

    // ---- headers (tx_headers), parked as virtual chunks 0..SH-1
    u32 my_partial = 0;
    const bool my_udp = proto == 17u;
    if (my_go) {
        u32x4 hc[SH];
        my_partial = tx_headers<MODE>(d, t.my_len, proto, ifc, hc);
#pragma unroll
        for (u32 c = 0; c < SH; ++c) UNET_LDS_STORE(park, t.lane * SH + c, TILE * SH, hc[c]);
    }
    if (t.lane < t.cnt) {
        out_len[i] = st == UNET_TX_OK_ ? frame_len : 0u;
        if (status) status[i] = (uint8_t)st;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the parked headers are read by lanes 0..SH-1
    __builtin_amdgcn_wave_barrier();

    // ---- mid-size payloads, 16, 8 or 4 packets at a time (the register buffers are free again afterwards)
    if (mid_mask != 0ull) {
        // (a group of W lanes covers 4 W chunks, a multiple of eight: the filled-up length still fits)
        if (UNET_TX_G4) tx_rows_path<4, R, SH>(pay, t, my_g4, store_len, my_oslot, my_partial, my_udp, list, sink, park);
        tx_rows_path<8, R, SH>(pay, t, my_g8, store_len, my_oslot, my_partial, my_udp, list, sink, park);
        tx_rows_path<16, R, SH>(pay, t, my_mid && !my_g4 && !my_g8, store_len, my_oslot, my_partial, my_udp, list, sink, park);
        if (any && !UNET_TX_EARLY_BIG) {
#pragma unroll
            for (int k = 0; k < D - 1; ++k) UNET_NEXT(buf[k])
        }
    }

    // ---- short payloads: one packet per lane
    if (small_chunks > kTxSmallChunks) {  // (wave-uniform)
        if (my_small)
            tx_build_small<R, SH, kTxSmallStaged>(pay, t.my_len, t.my_slot, frame_len, sink, my_oslot, park + t.lane * SH, my_partial, my_udp,
                                                  small_chunks);
    } else if (my_small) {
        tx_build_small<R, SH, kTxSmallChunks>(pay, t.my_len, t.my_slot, frame_len, sink, my_oslot, park + t.lane * SH, my_partial, my_udp,
                                              kTxSmallChunks);
    }
    if (sink.staged) tx_flush_staged(sink, t.lane, stage_lead);

    // ---- three register buffers in rotation, as in tile_sums: the loads of the next two payloads are
    // in flight while one packet is assembled and stored
#define UNET_CONSUME(X)                                                              \
    tx_build_one<R, SH, UNET_TX_TILE_BURST>(X, t.lane, rdlane(store_len, X.j),                           \
                        reinterpret_cast<u32x4*>(out_arena + (u64)rdlane(my_oslot, X.j) * 16u), TxHdrLds{park + X.j * SH}, \
                        rdlane(my_partial, X.j), rdlane((u32)my_udp, X.j) != 0u);
    if (any) {
        bool more = true;
#pragma nounroll
        while (more) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                if (more) {
                    UNET_NEXT(buf[(k + D - 1) % D])
                    UNET_CONSUME(buf[k])
                    more = buf[(k + 1) % D].live;
                }
            }
        }
    }
#undef UNET_NEXT
#undef UNET_CONSUME
}

// One packet per wave -- or per group of 2, 4, 8 or 16 waves of one workgroup -- for batches of long payloads (mean
// above 1 KiB).  A wave reads its packet's table entries and descriptor through the scalar unit; the packet's rows of
// 64 output chunks are dealt out to the group's waves in consecutive runs (at least two rows each, about four at the
// batch's mean length); every wave copies its run (tx_rows), the group's first wave also holds rows 0/1 back until
// the payload's sum is known: each wave leaves the exact sum of its rows in LDS, one workgroup barrier, and the first
// wave completes the L4 checksum and stores rows 0/1 (tx_finish).  Why not tiles: a device-to-device copy runs 10 %
// faster in waves that move one 1.5 KiB record and end than in waves that loop over four or eight of them (6.0 against
// 5.3 TB/s: short waves keep the chip's reads and writes inside one narrow moving window), and the tile kernel's
// per-wave set-up (64 lanes of table loads, LDS parking, 100 registers) is what made one packet per wave slow there.
// Correct for every payload length; short payloads waste their wave.
constexpr u32 kTxGroupMax = 16;  // waves per packet at most (one workgroup of 1024 threads)

// Packet i, built by wave w of a group of 1 << group_log2 waves (`wv`: this wave's number in its workgroup, `part`:
// the workgroup's sums; neither is used by a group of one).  POLICY bit 0: row 0 of the payload non-temporal, bit 1:
// row 1 (which holds the end of most payloads).  WRITE_META: out_len and status are stored too.
template <int MODE, int POLICY, bool WRITE_META>
__device__ __forceinline__ void tx_wave_packet(const Batch& pay, const TxDesc* __restrict__ desc, const IfaceDev& ifc,
                                               uint8_t* __restrict__ out_arena, u64 out_bytes, const u32* __restrict__ out_off16,
                                               u32* __restrict__ out_len, uint8_t* __restrict__ status, u32 flags, u32 i, bool have,
                                               u32 w, u32 group_log2, u32 wv, u32* __restrict__ part, u32 lane) {
    constexpr u32 SH = MODE == MODE_L2 ? 3u : 2u;
    constexpr int R = MODE == MODE_L2 ? 6 : 4;
    constexpr u32 kRows = tx_rows01_chunks(SH);
    const u32 group = 1u << group_log2;
    u32 len = 0, oslot = 0, st = UNET_TX_E_RANGE_, frame_len = 0, proto = 0;
    u64 slot = 0;
    TxDesc d = {};
    if (have) {
        len = pay.len[i];
        slot = pay.off16 ? (u64)pay.off16[i] : (u64)i * pay.stride16;
        oslot = out_off16[i];
        d = desc[i];
        proto = (d.id_proto >> 16) & 0xffu;
        const u32 l3len = 28u + len;
        frame_len = MODE == MODE_L2 ? 14u + max(l3len, 60u) : l3len;  // src/driver/tap.rs:147-152
        // the same checks, in the same order, as the tile kernel
        const bool in_ok = len <= kMaxLen && slot <= 0xffffffffull && slot * 16u + (((u64)len + 15u) & ~15ull) <= pay.arena_bytes;
        st = UNET_TX_OK_;
        if (!in_ok) st = UNET_TX_E_RANGE_;
        else if (!(proto == 1u || proto == 17u)) st = UNET_TX_E_PROTO_;
        else if (len > 65535u - 20u - 8u) st = UNET_TX_E_TOO_LONG_;  // src/transport/udp.rs:174-180
        else if ((u64)oslot * 16u + (((u64)frame_len + 15u) & ~15ull) > out_bytes) st = UNET_TX_E_RANGE_;
        if (WRITE_META && w == 0u && lane == 0u) {
            out_len[i] = st == UNET_TX_OK_ ? frame_len : 0u;
            if (status) status[i] = (uint8_t)st;
        }
    }
    const bool go = have && st == UNET_TX_OK_;
    u32 store_len = frame_len;  // (kTxFillLines: see k_tx_build)
    if (go && (flags & kTxFillLines) != 0u) {
        const u32 end = oslot + ((frame_len + 15u) >> 4);
        const u32 upto = (end + 7u) & ~7u;
        if (upto >= end && (u64)upto * 16u <= out_bytes) store_len = (upto - oslot) * 16u;
    }
    const u32 nin = go ? (len + 15u) >> 4 : 0u;
    // an empty payload may sit at the very end of the arena: read the arena's first chunk instead
    const TxPkt t(reinterpret_cast<const u32x4*>(pay.arena + (nin != 0u ? slot * 16u : 0ull)), len, nin, go ? store_len : 0u,
                  reinterpret_cast<u32x4*>(out_arena + (u64)oslot * 16u));
    const u32 kend = max(t.nin, t.nout - SH);                 // payload chunks (or zeros) the output covers
    const u32 rows = go ? (kend + SH + kWave - 1u) / kWave : 0u;  // rows of 64 output chunks
    const u32 per = max(2u, (rows + group - 1u) >> group_log2);  // rows per wave of the group
    u32 acc = 0, fx = 0, fy = 0;
    u32x4 a = {0, 0, 0, 0}, c = a;
    if (w == 0u) {
        if (go) {
            const u32x4* p0 = t.p + min(lane < SH ? 0u : lane - SH, t.last);
            const u32x4* p1 = t.p + min(lane + kWave - SH, t.last);
            const u32x4 r0 = (POLICY & 1) ? __builtin_nontemporal_load(p0) : *p0;
            const u32x4 r1 = (POLICY & 2) ? __builtin_nontemporal_load(p1) : *p1;
            const u32 k0 = lane - SH, k1 = lane + kWave - SH;  // k0 wraps for the header lanes: they hold no payload
            a = t.tm.apply(zero_unless(r0, lane >= SH && k0 < t.nin), k0);
            c = t.tm.apply(zero_unless(r1, k1 < t.nin), k1);
            acc = sum_chunk(c, sum_chunk(a, 0u));
            if (kRows < kend) {
                const u32x4 pk = t.chunk(kRows);
                fx = rdlane(pk.x, 0);
                fy = rdlane(pk.y, 0);
                if (per > 2u) acc = tx_rows<R, SH>(t, lane, kRows, min(kend, per * kWave - SH), acc, per * kWave - SH < kend);
            }
        }
    } else {
        const u32 r_begin = w * per;
        if (r_begin < rows) acc = tx_rows<R, SH>(t, lane, r_begin * kWave - SH, min(kend, (r_begin + per) * kWave - SH), acc, (r_begin + per) * kWave - SH < kend);
    }
    u32 total = wave_sum(acc);
    if (group > 1u) {  // (wave-uniform, and the same for every wave of the launch)
        if (lane == 0u) part[wv] = total;
        __syncthreads();
        if (w == 0u) {
            total = 0u;
            for (u32 k = 0; k < group; ++k) total += part[wv + k];  // exact: a payload's bytes sum to less than 2^31
        }
    }
    if (w == 0u && go) {
        TxHdrRegs<SH> hdr;
        const u32 partial = tx_headers<MODE>(d, len, proto, ifc, hdr.c);
        tx_finish<R, SH>(t, lane, a, c, fx, fy, total, hdr, partial, proto == 17u);
    }
}

template <int MODE, int POLICY>
__global__ __launch_bounds__(kTxGroupMax * kWave) void k_tx_build_wave(Batch pay, const TxDesc* __restrict__ desc, IfaceDev ifc,
                                                                       uint8_t* __restrict__ out_arena, u64 out_bytes,
                                                                       const u32* __restrict__ out_off16, u32* __restrict__ out_len,
                                                                       uint8_t* __restrict__ status, u32 flags, u32 group_log2) {
    __shared__ u32 part[kTxGroupMax];  // per wave of the workgroup: the sum of its rows
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const u32 gw = (u32)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + wv));
    const u32 i = gw >> group_log2, w = gw & ((1u << group_log2) - 1u);  // packet, wave of its group
    tx_wave_packet<MODE, POLICY, true>(pay, desc, ifc, out_arena, out_bytes, out_off16, out_len, status, flags, i, i < pay.n, w, group_log2,
                                       wv, part, lane);
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
    const u64 per = ((nchunks + gridDim.x - 1) / gridDim.x + 63u) & ~63ull;  // slabs start on 1 KiB boundaries
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

// The same bytes read by short-lived waves, each owning one contiguous piece of kPieceChunks chunks (24 KiB, what a
// wave of k_verify covers on 1500-byte records), four non-temporal 1 KiB rows requested ahead of the four being
// summed, and -- the launch adds unused LDS to every workgroup -- at most three workgroups resident on a CU: the
// fastest pure read found on MI355X (7.1 TB/s on pseudo-random bytes; eight resident workgroups 6.9-7.0, the slabs
// above 6.7-6.8; a buffer of one repeated byte reads 5 % faster than pseudo-random bytes).  The benchmark reports
// the better of the two shapes as what a pure read of its buffer attains.
constexpr u32 kPieceChunks = 1536;
__global__ __launch_bounds__(kBlock) void k_stream_read_pieces(const u32x4* __restrict__ buf, u64 nchunks, u32* __restrict__ sink) {
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = (u64)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const u64 lo = wave * kPieceChunks;
    if (lo >= nchunks) return;
    const u64 left = nchunks - lo;
    const u32 n = left < kPieceChunks ? (u32)left : kPieceChunks;
    const u32x4* p = buf + lo;
    const u32 last = n - 1u;
    u32 acc = 0;
    u32x4 v0 = ld_stream(p + min(lane, last)), v1 = ld_stream(p + min(lane + kWave, last));
    u32x4 v2 = ld_stream(p + min(lane + 2u * kWave, last)), v3 = ld_stream(p + min(lane + 3u * kWave, last));
    for (u32 c = 4u * kWave; c < n + 4u * kWave; c += 4u * kWave) {
        // rows past the piece's end re-read its last chunk (same line, no traffic) and are not summed
        const u32x4 n0 = ld_stream(p + min(c + lane, last)), n1 = ld_stream(p + min(c + lane + kWave, last));
        const u32x4 n2 = ld_stream(p + min(c + lane + 2u * kWave, last)), n3 = ld_stream(p + min(c + lane + 3u * kWave, last));
        const u32 at = c - 4u * kWave + lane;
        acc = sum_chunk(zero_unless(v0, at < n), acc);
        acc = sum_chunk(zero_unless(v1, at + kWave < n), acc);
        acc = sum_chunk(zero_unless(v2, at + 2u * kWave < n), acc);
        acc = sum_chunk(zero_unless(v3, at + 3u * kWave < n), acc);
        v0 = n0;
        v1 = n1;
        v2 = n2;
        v3 = n3;
    }
    acc = wave_sum(acc);
    if (lane == 0u) atomicAdd(&sink[wave & 4095u], acc);
}

// Third shape: the record kernels' own read loop and nothing else -- fixed-size records of `rec_chunks` chunks, sixteen
// to a wave, requested by issue_rows and summed by stream_one exactly as k_verify and k_csum do (three register
// buffers in rotation, streaming rows plus one default-policy row that holds the record's end), with no headers parked,
// no parse and no result.  (On about half of the boxes every one of these shapes comes out 1-2.5 % below k_verify
// itself, 6.75-6.88 against 6.95 TB/s on the same buffer -- this one, with less work per byte on the same loads,
// included; it is fastest with four workgroups to a CU.  The verdict kernel's waves alternate between requesting and
// arithmetic, which thins the request stream out much as an occupancy cap does.)
__global__ __launch_bounds__(kBlock, 6) void k_stream_read_records(Batch b, u32* __restrict__ sink) {
    Tile t;
    if (!t.init<16>(b)) return;
    u64 todo = __ballot(t.lane < t.cnt && t.my_ok);
    u32 acc = 0;
    Rows2 A, B, C;
#define UNET_NEXT(R)                                                                 \
    {                                                                                \
        const bool has = todo != 0ull;                                               \
        const u32 j = has ? (u32)__builtin_ctzll(todo) : 0u;                         \
        todo &= todo - 1ull;                                                         \
        R = issue_rows(b, t.my_len, t.my_slot, t.lane, j, has);                      \
    }
#define UNET_CONSUME(R) acc += stream_one<false, 16u>(R, t.lane, nullptr);
    if (todo != 0ull) {
        UNET_NEXT(A)
        UNET_NEXT(B)
#pragma nounroll
        while (true) {
            UNET_NEXT(C)
            UNET_CONSUME(A)
            if (!B.live) break;
            UNET_NEXT(A)
            UNET_CONSUME(B)
            if (!C.live) break;
            UNET_NEXT(B)
            UNET_CONSUME(C)
            if (!A.live) break;
        }
    }
#undef UNET_NEXT
#undef UNET_CONSUME
    if (t.lane == 0u) atomicAdd(&sink[(blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) & 4095u], acc);
}

// Read+write yardstick: plain copy with the same access pattern (slab per workgroup, 16 bytes per lane).
__global__ __launch_bounds__(kBlock) void k_stream_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, u64 nchunks) {
    const u64 per = ((nchunks + gridDim.x - 1) / gridDim.x + 63u) & ~63ull;  // slabs start on 1 KiB boundaries
    const u64 lo = (u64)blockIdx.x * per;
    const u64 hi = lo + per < nchunks ? lo + per : nchunks;
    u64 i = lo + threadIdx.x;
    for (; i + 3 * kBlock < hi; i += 4 * kBlock) {
        const u32x4 a = ld_stream(src + i), b = ld_stream(src + i + kBlock), c = ld_stream(src + i + 2 * kBlock),
                    d = ld_stream(src + i + 3 * kBlock);
        __builtin_nontemporal_store(a, dst + i);
        __builtin_nontemporal_store(b, dst + i + kBlock);
        __builtin_nontemporal_store(c, dst + i + 2 * kBlock);
        __builtin_nontemporal_store(d, dst + i + 3 * kBlock);
    }
    for (; i < hi; i += kBlock) __builtin_nontemporal_store(ld_stream(src + i), dst + i);
}

// ------------------------------------------------------------------ synthetic traces
// Record r of a generated batch is trace index first_idx + (rel_idx ? rel_idx[r] : r).
__device__ __forceinline__ u64 trace_index(u64 first_idx, const u32* __restrict__ rel_idx, u32 r) {
    return first_idx + (rel_idx ? (u64)rel_idx[r] : (u64)r);
}

// Layer 1 (+ the post-checksum payload flip): one wave per record, 16 bytes per lane.
__global__ __launch_bounds__(kBlock) void k_synth_raw(int kind, u64 seed, u64 first_idx, const u32* __restrict__ rel_idx,
                                                      u32 n, const u32* __restrict__ off16,
                                                      const u32* __restrict__ len, uint8_t* __restrict__ arena, u64 arena_bytes) {
    const u32 lane = threadIdx.x & 63u;
    const u32 rec = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (rec >= n) return;
    const u64 idx = trace_index(first_idx, rel_idx, rec);
    const u64 key = unetsynth::key_of(seed, idx);
    const unetsynth::Shape s = unetsynth::shape_of(kind, seed, idx);
    const u32 flen = len[rec];
    const u64 slot = off16[rec];
    // a table that does not describe this trace, or points outside the arena, writes nothing
    if (flen != s.len || slot * 16u + (((u64)flen + 15u) & ~15ull) > arena_bytes) return;
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
__global__ __launch_bounds__(kBlock) void k_synth_hdr(int kind, u64 seed, u64 first_idx, const u32* __restrict__ rel_idx,
                                                      u32 n, u32 unicast, u32 broadcast,
                                                      const u32* __restrict__ off16, const u32* __restrict__ len,
                                                      uint8_t* __restrict__ arena, u64 arena_bytes) {
    const u32 rec = blockIdx.x * kBlock + threadIdx.x;
    if (rec >= n) return;
    const u64 idx = trace_index(first_idx, rel_idx, rec);
    const u64 key = unetsynth::key_of(seed, idx);
    const unetsynth::Shape s = unetsynth::shape_of(kind, seed, idx);
    const unetsynth::Tuple tp = unetsynth::tuple_of(kind, seed, idx, s, unicast, broadcast);
    uint8_t h[96];
    const u32 hb = unetsynth::build_header(kind, key, s, tp, unicast, h);
    if (kind == unetsynth::kKindHdr20) {
        if ((u64)rec * 20u + 20u > arena_bytes) return;
    } else {
        // same rule as k_synth_raw: a mismatched table writes nothing
        if (len[rec] != s.len || (u64)off16[rec] * 16u + (((u64)s.len + 15u) & ~15ull) > arena_bytes) return;
    }
    uint8_t* base = kind == unetsynth::kKindHdr20 ? arena + (u64)rec * 20u : arena + (u64)off16[rec] * 16u;
    u32* dst = reinterpret_cast<u32*>(base);
    for (u32 i = 0; i < hb; i += 4) {
        dst[i >> 2] = (u32)h[i] | ((u32)h[i + 1] << 8) | ((u32)h[i + 2] << 16) | ((u32)h[i + 3] << 24);
    }
}

// Record lengths of a set of trace indices (the layout is an exclusive scan of their 16-byte slots).
__global__ __launch_bounds__(kBlock) void k_synth_len(int kind, u64 seed, u64 first_idx, const u32* __restrict__ rel_idx,
                                                      u32 n, u32* __restrict__ len) {
    const u32 rec = blockIdx.x * kBlock + threadIdx.x;
    if (rec >= n) return;
    len[rec] = unetsynth::shape_of(kind, seed, trace_index(first_idx, rel_idx, rec)).len;
}

// GPU that owns each trace index of [first_idx, first_idx + n): flow hash of the record's 5-tuple
// modulo the number of GPUs.  Needs no record bytes, so a rank can pick its records before it
// generates (or receives) any of them.
__global__ __launch_bounds__(kBlock) void k_synth_shard(int kind, u64 seed, u64 first_idx, u32 n, u32 unicast,
                                                        u32 broadcast, u32 world, uint8_t* __restrict__ shard) {
    const u32 rec = blockIdx.x * kBlock + threadIdx.x;
    if (rec >= n) return;
    const u64 idx = first_idx + rec;
    const unetsynth::Shape s = unetsynth::shape_of(kind, seed, idx);
    const unetsynth::Tuple tp = unetsynth::tuple_of(kind, seed, idx, s, unicast, broadcast);
    shard[rec] = (uint8_t)(flow_hash5(tp.src, tp.dst, tp.proto, tp.sport, tp.dport) % world);
}

}  // namespace unetdev
