// Per-flow TCP tracking on the GPU that owns the flow.  An extension with no counterpart in the reference (it has
// no TCP at all, src/transport.rs:10-28); the only conventions taken from it are "the segment is whatever follows
// the IPv4 header in the buffer" (src/protocols/ipv4.rs:415) and the 5-tuple of the pseudo-header.
//
// Sequential definition (what the CPU oracle does): the records of a batch are visited in index order; every
// record judged ACCEPT_TCP looks up slot = mix(flow_hash5(5-tuple)) of a direct-mapped table and runs the small
// state machine of tcp_apply() on it.  Parallel form: (1) one lane per record extracts {slot, 5-tuple, seq,
// flags, sequence space consumed}; (2) a stable radix sort of (slot, index) brings the records of a slot together
// in index order; (3) one lane per slot walks its run sequentially with the slot's entry in registers.  Records of
// different slots never touch the same entry, so the result equals the sequential one bit for bit.
#pragma once
#include "kernels.h"

namespace unetdev {

// One slot of the flow table: mirrors unet_tcp_flow (32 bytes).
struct TcpFlow {
    u32 src, dst;
    u32 ports;      // sport << 16 | dport
    u32 next_seq;   // sequence number expected next
    u32 state;      // UNET_TCP_* in the low byte
    u32 segments, out_of_order, after_close;
};
static_assert(sizeof(TcpFlow) == 32, "flow table slot");

// What one accepted TCP segment contributes.
struct TcpSeg {
    u32 src, dst, ports, seq;
    u32 consumed;   // payload bytes + SYN + FIN: the sequence space the segment occupies
    u32 flags;      // TCP flags byte
};

enum : u32 { TCP_FREE = 0, TCP_SYN = 1, TCP_ESTABLISHED = 2, TCP_FIN = 3, TCP_RESET = 4 };
enum : u32 { TCPS_NOT_TCP = 0, TCPS_NEW = 1, TCPS_IN_ORDER = 2, TCPS_OUT_OF_ORDER = 3, TCPS_AFTER_CLOSE = 4 };
constexpr u32 kTcpFin = 0x01, kTcpSyn = 0x02, kTcpRst = 0x04;

// Table slot of a flow: the high bits of a multiplicative mix of the flow hash (its low bits already chose the
// GPU, so they are the same for every flow one table sees).
__host__ __device__ inline u32 tcp_slot(u32 hash, u32 log2_slots) { return log2_slots ? (hash * 0x9E3779B1u) >> (32u - log2_slots) : 0u; }

// The state machine, shared by the kernel and (restated) by the oracle.
__host__ __device__ inline u32 tcp_apply(TcpFlow& f, const TcpSeg& s) {
    const bool same = f.state != TCP_FREE && f.src == s.src && f.dst == s.dst && f.ports == s.ports;
    const bool reopen = same && f.state == TCP_RESET && (s.flags & kTcpSyn);
    if (!same || reopen) {
        // first segment of a flow in this slot (free, taken over from another flow, or a new SYN after a reset)
        f.src = s.src;
        f.dst = s.dst;
        f.ports = s.ports;
        f.next_seq = s.seq + s.consumed;
        f.state = (s.flags & kTcpRst) ? TCP_RESET : ((s.flags & kTcpFin) ? TCP_FIN : ((s.flags & kTcpSyn) ? TCP_SYN : TCP_ESTABLISHED));
        f.segments = 1;
        f.out_of_order = 0;
        f.after_close = 0;
        return TCPS_NEW;
    }
    f.segments++;
    if (f.state == TCP_RESET) {
        f.after_close++;
        return TCPS_AFTER_CLOSE;
    }
    u32 out;
    if (s.seq == f.next_seq) {
        out = TCPS_IN_ORDER;
        f.next_seq = s.seq + s.consumed;
        if (s.flags & kTcpFin) f.state = TCP_FIN;
        else if (f.state == TCP_SYN && !(s.flags & kTcpSyn)) f.state = TCP_ESTABLISHED;
    } else {
        out = TCPS_OUT_OF_ORDER;
        f.out_of_order++;
        // a gap ahead moves the expectation forward; old data (a retransmission) leaves it alone
        if ((int)(s.seq - f.next_seq) > 0) f.next_seq = s.seq + s.consumed;
    }
    if (s.flags & kTcpRst) f.state = TCP_RESET;
    return out;
}

// (1) one lane per record
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_tcp_keys(Batch b, const u32* __restrict__ results, u32 log2_slots, u32* __restrict__ keys,
                                                     u32* __restrict__ idx, TcpSeg* __restrict__ segs, uint8_t* __restrict__ seg_state) {
    const u32 i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= b.n) return;
    idx[i] = i;
    u32 key = 1u << log2_slots;  // sorts behind every slot: not an accepted TCP segment
    const u32 flen = b.len[i];
    const u64 slot = b.off16 ? (u64)b.off16[i] : (u64)i * b.stride16;
    // d_results is only trusted as far as the tables allow: a verdict that claims a record the arena does not hold
    // (results of another batch, a table changed in between) is treated as "not TCP"
    const bool inside = flen <= kMaxLen && flen >= (MODE == MODE_L2 ? 54u : 40u) && slot * 16u + (((u64)flen + 15u) & ~15ull) <= b.arena_bytes;
    const u32* w = reinterpret_cast<const u32*>(b.arena + (inside ? slot * 16u : 0ull));
    auto f = [w](u32 k) { return w[k]; };
    const u32 l3len = MODE == MODE_L2 ? flen - 14u : flen;
    const u32 hl = inside ? (l3raw<MODE>(f, 0) & 0xfu) * 4u : 0u;
    if (inside && ((results[i] >> 16) & 0xffu) == V_ACCEPT_TCP && hl >= 20u && hl + 20u <= l3len) {
        // the receive path has accepted it: the headers are complete and consistent
        TcpSeg s;
        s.src = bswap32(l3raw<MODE>(f, 12));
        s.dst = bswap32(l3raw<MODE>(f, 16));
        const u32 t0 = l3raw<MODE>(f, hl), t3 = l3raw<MODE>(f, hl + 12u);
        const u32 sport = bswap16(t0 & 0xffffu), dport = bswap16(t0 >> 16);
        s.ports = (sport << 16) | dport;
        s.seq = bswap32(l3raw<MODE>(f, hl + 4u));
        s.flags = (t3 >> 8) & 0xffu;
        const u32 doff = (t3 >> 4) & 0xfu;
        // The segment is everything behind the IPv4 header in the buffer (src/protocols/ipv4.rs:415), the same rule
        // the pseudo-header length of the verdict used: an Ethernet frame padded past its IPv4 total length fails that
        // checksum and never gets here, so buffer length and total length agree for every accepted segment a sender
        // built by that rule.
        const u32 payload = hl + doff * 4u <= l3len ? l3len - hl - doff * 4u : 0u;
        s.consumed = payload + ((s.flags & kTcpSyn) ? 1u : 0u) + ((s.flags & kTcpFin) ? 1u : 0u);
        segs[i] = s;
        key = tcp_slot(flow_hash5(s.src, s.dst, 6u, sport, dport), log2_slots);
    } else {
        seg_state[i] = (uint8_t)TCPS_NOT_TCP;
    }
    keys[i] = key;
}

// Counters one call leaves behind (mirrors unet_tcp_stats).
struct TcpStats {
    u32 longest_run;  // most segments that met one table slot in this batch
    u32 long_runs;    // runs longer than kTcpShortRun (each walked by a whole wave out of LDS)
};
constexpr u32 kTcpShortRun = 64;  // runs up to this length are walked by the lane that finds their start

// (3) one lane per sorted position; the lane at the start of a slot's run finds the run's end (the keys are sorted:
// a binary search) and walks the run if it is short.  A long run -- a hot flow, or a table far too small for the
// batch -- would hold one lane on a chain of dependent loads for as long as it lasts: it goes onto a list instead.
__global__ __launch_bounds__(kBlock) void k_tcp_apply(const u32* __restrict__ keys, const u32* __restrict__ idx, const TcpSeg* __restrict__ segs,
                                                      u32 n, u32 log2_slots, TcpFlow* __restrict__ table, uint8_t* __restrict__ seg_state,
                                                      u32* __restrict__ long_list, u32* __restrict__ long_count, TcpStats* __restrict__ stats) {
    const u32 p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const u32 key = keys[p];
    if (key >= (1u << log2_slots)) return;
    if (p > 0u && keys[p - 1u] == key) return;
    // first position behind the run: the first q in (p, n] with keys[q] != key (keys[n] counts as different)
    u32 lo = p + 1u, hi = n;
    if (lo < n && keys[lo] == key) {
        while (lo < hi) {
            const u32 mid = lo + ((hi - lo) >> 1);
            if (keys[mid] == key) lo = mid + 1u; else hi = mid;
        }
    }
    const u32 end = lo < n && keys[lo] == key ? n : lo;
    const u32 run = end - p;
    if (stats) atomicMax(&stats->longest_run, run);
    if (run > kTcpShortRun) {
        const u32 at = atomicAdd(long_count, 1u);
        long_list[2u * at] = p;
        long_list[2u * at + 1u] = run;
        if (stats) atomicAdd(&stats->long_runs, 1u);
        return;
    }
    TcpFlow f = table[key];
    for (u32 q = p; q < end; ++q) {
        const u32 i = idx[q];
        seg_state[i] = (uint8_t)tcp_apply(f, segs[i]);
    }
    table[key] = f;
}

// (3b) the long runs, one wave each (the waves share the list out between them): 64 segments at a time are gathered
// into LDS by the whole wave, lane 0 runs the state machine over them out of LDS, and the wave stores the 64
// outcomes.  The walk stays sequential -- it is the definition -- but no step of it waits for memory.
__global__ __launch_bounds__(kBlock) void k_tcp_apply_long(const u32* __restrict__ keys, const u32* __restrict__ idx, const TcpSeg* __restrict__ segs,
                                                           TcpFlow* __restrict__ table, uint8_t* __restrict__ seg_state,
                                                           const u32* __restrict__ long_list, const u32* __restrict__ long_count) {
    __shared__ TcpSeg seg_lds[kBlock / kWave][kWave];
    __shared__ u32 out_lds[kBlock / kWave][kWave];
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const u32 waves = gridDim.x * (kBlock / kWave);
    const u32 count = *long_count;
    for (u32 r = blockIdx.x * (kBlock / kWave) + w; r < count; r += waves) {
        const u32 p = long_list[2u * r], run = long_list[2u * r + 1u];
        const u32 key = keys[p];
        TcpFlow f = table[key];  // (every lane holds a copy; lane 0's is the one that advances)
        for (u32 at = 0; at < run; at += kWave) {
            const u32 m = run - at < kWave ? run - at : kWave;
            u32 i = 0;
            if (lane < m) {
                i = idx[p + at + lane];
                seg_lds[w][lane] = segs[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0u) {
                for (u32 k = 0; k < m; ++k) out_lds[w][k] = tcp_apply(f, seg_lds[w][k]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < m) seg_state[i] = (uint8_t)out_lds[w][lane];
        }
        if (lane == 0u) table[key] = f;
    }
}

}  // namespace unetdev
