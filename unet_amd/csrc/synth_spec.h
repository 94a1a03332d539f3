// Definition of the synthetic frame traces, shared verbatim by the host twin and the
// device generator kernels (every function is __host__ __device__), so "record idx of
// trace (kind, seed)" has exactly one definition.
//
// A record is built in three layers:
//   1. raw stream  : byte p of the record is byte (p % 8) of the little-endian 64-bit word
//                    raw_word(key, p / 8); this covers the record and its pad up to the next
//                    16-byte boundary (pad bytes are deliberately garbage, not zero).
//   2. headers     : IPv4 header (+ random option bytes when IHL > 5) and the UDP / TCP / ICMP
//                    header overwrite bytes [0, payload_start) with valid checksums.
//   3. fault       : chosen by idx modulo 64 (and 256); either a header field is made invalid
//                    before the checksums are computed, or a bit is flipped afterwards.
// The generator carries its own plain word-loop Internet checksum; it shares no code with the
// kernels under test.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define UNET_HD __host__ __device__ inline
#else
#define UNET_HD inline
#endif

namespace unetsynth {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr u64 kGolden = 0x9E3779B97F4A7C15ull;
constexpr int kKindHdr20 = 2;
constexpr int kKindUdp1500 = 3;
constexpr int kKindImix = 4;
constexpr int kKindFlows = 5;  // IMIX shapes, but the 5-tuple comes from a table of 2^20 flows (skewed popularity)
constexpr int kKindOdd = 6;    // IMIX rules with lengths that end one byte past or short of a 16-byte boundary, or odd
constexpr u32 kFlowCount = 1u << 20;

UNET_HD u64 fin(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
UNET_HD u64 key_of(u64 seed, u64 idx) { return fin(seed + (idx + 1) * kGolden); }
UNET_HD u64 field(u64 key, u32 f) { return fin(key + (u64)(f + 1) * kGolden); }
UNET_HD u64 raw_word(u64 key, u32 k) { return fin((key ^ 0xA5A5A5A5A5A5A5A5ull) + (u64)k * kGolden); }
UNET_HD uint8_t raw_byte(u64 key, u32 pos) { return (uint8_t)(raw_word(key, pos >> 3) >> (8 * (pos & 7))); }

struct Shape {
    u32 len;            // record length in bytes
    u32 ihl;            // real IHL in dwords (layout); the header nibble may be faulted
    u32 proto;          // 1 / 6 / 17 (layout); the header byte may be faulted
    u32 l4hl;           // L4 header bytes (8 UDP/ICMP, 20 or 32 TCP); 0 for bare headers
    u32 payload_start;  // ihl*4 + l4hl
    u32 fault;          // idx % 64
    u32 fault256;       // idx % 256
};

// Fault selectors (idx % 64 unless noted).
constexpr u32 kFaultIpCsum = 3;      // bit flip in the source address after the header checksum
constexpr u32 kFaultPayload = 7;     // bit flip in the L4 payload after the L4 checksum
constexpr u32 kFaultVersion = 11;    // version 6
constexpr u32 kFaultUdpLen = 13;     // IMIX, UDP only: length field 4 short
constexpr u32 kFaultIhlShort = 19;   // IHL nibble 4
constexpr u32 kFaultNotForMe = 21;   // IMIX: destination on another host
constexpr u32 kFaultIhlLong = 27;    // IHL nibble 15
constexpr u32 kFaultIcmpType = 29;   // IMIX, ICMP only: type 3
constexpr u32 kFaultMoreFrag = 35;   // MF flag
constexpr u32 kFaultTcpOpt = 45;     // IMIX, TCP with options only: malformed option area (checksum stays valid)
constexpr u32 kFaultFragOff = 43;    // fragment offset != 0
constexpr u32 kFaultProto = 51;      // protocol 47
constexpr u32 kFaultIhlOpt = 59;     // HDR20: IHL 6..14 (valid: only 20 bytes are summed)
constexpr u32 kFault256UdpLen = 33;  // UDP1500: length field short by 8..32
constexpr u32 kFault256Csum0 = 97;   // UDP1500: transmitted checksum 0x0000

// The addressing 5-tuple a record carries before any fault is applied (ports are 0 for ICMP).
struct Tuple {
    u32 src, dst, proto, sport, dport;
};

// kKindFlows: flow popularity is the product of two uniform 20-bit draws (a long-tailed, Zipf-like
// law: half of all records fall on ~19 % of the flows, 10 % of them on ~0.7 %).
UNET_HD u32 flow_id_of(u64 key) {
    const u64 f = field(key, 15);
    const u64 a = (f >> 44) & 0xfffff, b = (f >> 24) & 0xfffff;
    return (u32)((a * b) >> 20);
}
UNET_HD u64 flow_key(u64 seed, u32 flow) { return fin((seed ^ 0xF10F10F10F10F10Full) + (u64)(flow + 1) * kGolden); }
UNET_HD u32 flow_proto(u64 fk) {
    const u32 b = (u32)((fin(fk) >> 8) % 100);
    return b < 88 ? 6u : (b < 98 ? 17u : 1u);
}

UNET_HD Shape shape_of(int kind, u64 seed, u64 idx) {
    Shape s;
    u64 key = key_of(seed, idx);
    s.fault = (u32)(idx % 64);
    s.fault256 = (u32)(idx % 256);
    s.ihl = 5;
    if (kind == kKindHdr20) {
        u64 f2 = field(key, 2);
        u32 sel = (u32)((f2 >> 8) % 16);
        s.len = 20;
        s.proto = sel < 9 ? 17u : (sel < 13 ? 1u : 6u);
        s.l4hl = 0;
    } else if (kind == kKindUdp1500) {
        s.len = 1500;
        s.proto = 17;
        s.l4hl = 8;
    } else {
        u64 f6 = field(key, 6);
        u32 a = (u32)(f6 % 100);
        s.len = a < 56 ? 64u : (a < 88 ? 576u : (a < 96 ? 1500u : (a < 98 ? 4096u : 9000u)));
        if (kind == kKindOdd) s.len = a < 56 ? 65u : (a < 88 ? 577u : (a < 96 ? 1499u : (a < 98 ? 4097u : 8999u)));
        u32 b = (u32)((f6 >> 8) % 100);
        s.proto = b < 88 ? 6u : (b < 98 ? 17u : 1u);
        if (kind == kKindFlows) s.proto = flow_proto(flow_key(seed, flow_id_of(key)));
        if (((f6 >> 16) % 20) == 0) s.ihl = ((f6 >> 24) & 1) ? 8u : 6u;
        s.l4hl = s.proto == 6 ? ((((f6 >> 32) % 8) == 0) ? 32u : 20u) : 8u;
    }
    s.payload_start = s.ihl * 4 + s.l4hl;
    return s;
}

UNET_HD Tuple tuple_of(int kind, u64 seed, u64 idx, const Shape& s, u32 unicast, u32 broadcast) {
    const u64 key = key_of(seed, idx);
    Tuple t;
    u64 ports, dsel_src;
    if (kind == kKindFlows) {
        const u64 fk = flow_key(seed, flow_id_of(key));
        t.src = (u32)fk;
        ports = fk >> 32;
        dsel_src = fin(fk) >> 32;
    } else {
        t.src = (u32)field(key, 1);
        ports = field(key, 3);
        dsel_src = field(key, 2);
    }
    const u32 dsel = (u32)(dsel_src % 16);
    t.dst = dsel == 0 ? broadcast : (dsel == 1 ? 0xffffffffu : unicast);
    t.proto = s.proto;
    const bool has_ports = s.proto == 17 || s.proto == 6;
    t.sport = has_ports ? (u32)ports & 0xffff : 0u;
    t.dport = has_ports ? (u32)(ports >> 16) & 0xffff : 0u;
    return t;
}

// Position and mask of the post-checksum payload bit flip, if this record has one.
UNET_HD bool flip_of(int kind, u64 key, const Shape& s, u32* pos, u32* mask) {
    if (kind == kKindHdr20 || s.fault != kFaultPayload) return false;
    if (s.len <= s.payload_start) return false;
    u64 f = field(key, 11);
    *pos = s.payload_start + (u32)(f % (s.len - s.payload_start));
    *mask = 1u << ((f >> 40) & 7);
    return true;
}

UNET_HD void put16(uint8_t* p, u32 v) {
    p[0] = (uint8_t)(v >> 8);
    p[1] = (uint8_t)v;
}
UNET_HD void put32(uint8_t* p, u32 v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}
// Exact (unfolded) sum of big-endian 16-bit words of an even-length byte range.
UNET_HD u32 sum_be_even(const uint8_t* p, u32 n) {
    u32 s = 0;
    for (u32 i = 0; i + 1 < n; i += 2) s += ((u32)p[i] << 8) | p[i + 1];
    return s;
}
// Exact sum of BE words of the raw stream over [start, end), start even; an odd tail byte
// counts as the high byte of a zero-padded word.
UNET_HD u32 sum_be_raw(u64 key, u32 start, u32 end) {
    u32 s = 0;
    u64 w = 0;
    u32 have = 0xffffffffu;
    for (u32 p = start; p < end; p += 2) {
        if ((p >> 3) != have) {
            have = p >> 3;
            w = raw_word(key, have);
        }
        u32 hi = (u32)(w >> (8 * (p & 7))) & 0xff;
        u32 lo = (p + 1 < end) ? ((u32)(w >> (8 * ((p + 1) & 7))) & 0xff) : 0u;
        s += (hi << 8) | lo;
    }
    return s;
}
UNET_HD u32 fold_not(u32 s) {
    while (s >> 16) s = (s & 0xffff) + (s >> 16);
    return (~s) & 0xffff;
}

// Writes header bytes [0, s.payload_start) of record idx into h (at least 96 bytes) and
// returns payload_start.  For kKindHdr20 writes the 20 header bytes.
UNET_HD u32 build_header(int kind, u64 key, const Shape& s, const Tuple& tp, u32 unicast, uint8_t* h) {
    const u64 f0 = field(key, 0), f3 = field(key, 3);
    const u64 f4 = field(key, 4), f5 = field(key, 5), f7 = field(key, 7), f8 = field(key, 8);
    const u32 hl = s.ihl * 4;
    const bool imix = kind == kKindImix || kind == kKindFlows || kind == kKindOdd;
    // ---- IPv4 header
    u32 ihl_nibble = s.ihl, version = 4, proto = s.proto;
    u32 flags_frag = ((f0 >> 24) & 1) ? 0x4000u : 0u;
    u32 dst = tp.dst;
    if (kind != kKindUdp1500) {
        if (s.fault == kFaultVersion) version = 6;
        if (s.fault == kFaultIhlShort) ihl_nibble = 4;
        if (s.fault == kFaultIhlLong) ihl_nibble = 15;
        if (s.fault == kFaultMoreFrag) flags_frag |= 0x2000u;
        if (s.fault == kFaultFragOff) flags_frag |= 1u + (u32)(f5 % 0x1fff);
        if (s.fault == kFaultProto) proto = 47;
        if (kind == kKindHdr20 && s.fault == kFaultIhlOpt) ihl_nibble = 6 + (u32)(f5 % 9);
        if (imix && s.fault == kFaultNotForMe) dst = unicast ^ 0x00005500u;
    }
    h[0] = (uint8_t)((version << 4) | ihl_nibble);
    h[1] = (uint8_t)f0;
    put16(h + 2, s.len);
    put16(h + 4, (u32)(f0 >> 8) & 0xffff);
    put16(h + 6, flags_frag);
    h[8] = (uint8_t)(1 + (u32)((f0 >> 32) % 255));
    h[9] = (uint8_t)proto;
    h[10] = 0;
    h[11] = 0;
    put32(h + 12, tp.src);
    put32(h + 16, dst);
    for (u32 i = 20; i < hl; i++) h[i] = (uint8_t)(field(key, 12 + (i >> 3)) >> (8 * (i & 7)));
    put16(h + 10, fold_not(sum_be_even(h, 20)));  // options are not part of the sum (ipv4.rs:159-182)
    if (kind == kKindHdr20) {
        if (s.fault == kFaultIpCsum) h[12 + (f5 % 4)] ^= (uint8_t)(1u << ((f5 >> 8) % 8));
        return 20;
    }
    // ---- L4 header
    uint8_t* l4 = h + hl;
    const u32 seglen = s.len - hl;
    u32 csum_off;
    u32 pseudo = 0;
    if (s.proto == 17) {
        u32 length = seglen;
        if (kind == kKindUdp1500 && s.fault256 == kFault256UdpLen) length -= 8 * (1 + (u32)(f4 % 4));
        if (imix && s.fault == kFaultUdpLen) length -= 4;
        put16(l4, tp.sport);
        put16(l4 + 2, tp.dport);
        put16(l4 + 4, length);
        csum_off = 6;
        pseudo = (tp.src >> 16) + (tp.src & 0xffff) + (dst >> 16) + (dst & 0xffff) + 17 + length;
    } else if (s.proto == 6) {
        put16(l4, tp.sport);
        put16(l4 + 2, tp.dport);
        put32(l4 + 4, (u32)f7);
        put32(l4 + 8, (u32)(f7 >> 32));
        l4[12] = (uint8_t)((s.l4hl / 4) << 4);
        l4[13] = (uint8_t)(0x10 | (f8 & 0x0f));
        put16(l4 + 14, (u32)(f8 >> 8) & 0xffff);
        put16(l4 + 18, 0);
        if (s.l4hl == 32) {
            // 12 option bytes: NOP, NOP, timestamps (kind 8, length 10, two random 32-bit values)
            const u64 ts = field(key, 9);
            l4[20] = 1;
            l4[21] = 1;
            l4[22] = 8;
            l4[23] = 10;
            put32(l4 + 24, (u32)ts);
            put32(l4 + 28, (u32)(ts >> 32));
            if (imix && s.fault == kFaultTcpOpt) {
                const u32 how = (u32)(field(key, 10) % 3);
                if (how == 0) l4[23] = 1;        // length below 2
                else if (how == 1) l4[23] = 11;  // runs past the data offset
                else l4[22] = 2;                 // MSS with length 10
            }
        }
        csum_off = 16;
        pseudo = (tp.src >> 16) + (tp.src & 0xffff) + (dst >> 16) + (dst & 0xffff) + 6 + seglen;
    } else {
        u32 type = (f3 & 1) ? 8u : 0u;
        if (imix && s.fault == kFaultIcmpType) type = 3;
        l4[0] = (uint8_t)type;
        l4[1] = 0;
        put32(l4 + 4, (u32)f7);
        csum_off = 2;
    }
    l4[csum_off] = 0;
    l4[csum_off + 1] = 0;
    u32 total = pseudo + sum_be_even(l4, s.l4hl) + sum_be_raw(key, s.payload_start, s.len);
    u32 ck = fold_not(total);  // stored as computed: no 0x0000 -> 0xffff substitution (udp.rs:202-203)
    if (kind == kKindUdp1500 && s.fault256 == kFault256Csum0) ck = 0;
    put16(l4 + csum_off, ck);
    if (imix && s.fault == kFaultIpCsum) h[12 + (f5 % 4)] ^= (uint8_t)(1u << ((f5 >> 8) % 8));
    return s.payload_start;
}

}  // namespace unetsynth
