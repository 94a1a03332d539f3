// Header parse + verdict logic of the receive path, written once over an abstract "dword k of the
// record" fetch: the kernels run it lane-parallel on headers parked in LDS (one lane per record),
// the flow-hash kernel on plain per-lane global loads.
// The order of checks is the reference's order; each block cites the lines it restates.
#pragma once
#include "csum_core.h"

namespace unetdev {

// Numeric verdict codes: same values as enum unet_verdict in include/unet_offload.h.
enum : u32 {
    V_NONE = 0,
    V_DROP_L2_TRUNCATED = 1,
    V_DROP_L2_ETHERTYPE = 2,
    V_DROP_L2_NOT_FOR_ME = 3,
    V_PASS_ARP = 4,
    V_DROP_L3_TRUNCATED = 5,
    V_DROP_L3_PROTO = 6,
    V_DROP_L3_VERSION = 7,
    V_DROP_L3_IHL_SHORT = 8,
    V_DROP_L3_IHL_LONG = 9,
    V_DROP_L3_FRAGMENT = 10,
    V_DROP_L3_CSUM = 11,
    V_IGNORE_L3_NOT_FOR_ME = 12,
    V_DROP_ICMP_TRUNCATED = 13,
    V_DROP_ICMP_TYPE = 14,
    V_ACCEPT_ICMP_ECHO = 15,
    V_ACCEPT_ICMP_ECHOREPLY = 16,
    V_DROP_UDP_SHORT = 17,
    V_DROP_UDP_LEN = 18,
    V_DROP_UDP_CSUM = 19,
    V_ACCEPT_UDP = 20,
    V_DROP_TCP_SHORT = 21,
    V_DROP_TCP_DOFF = 22,
    V_DROP_TCP_CSUM = 23,
    V_ACCEPT_TCP = 24,
    V_L3_HEADER_OK = 25,
    V_DROP_ARP_TRUNCATED = 26,
    V_DROP_ARP_HTYPE = 27,
    V_DROP_ARP_PTYPE = 28,
    V_DROP_ARP_HLEN = 29,
    V_DROP_ARP_PLEN = 30,
    V_IGNORE_ARP_NOT_FOR_ME = 31,
    V_DROP_TCP_OPTION = 32,
    V_COUNT = 40
};

constexpr int MODE_L3 = 0;
constexpr int MODE_L2 = 1;

enum : u32 { K_DONE = 0, K_UDP = 1, K_TCP = 2, K_ECHO = 3 };

// Outcome of the O(1) part.  When kind != K_DONE the verdict still needs the record's byte
// sum: with T = exact sum of the byte-swapped 16-bit words of the whole record [0, len),
//   L4 sum  = T - sub + add        (sub removes everything in front of the L4 segment and,
//                                   for an echo, the type byte and checksum word the reply
//                                   zeroes; add is the pseudo-header)
struct Parsed {
    u32 res;   // packed result if kind == K_DONE
    u32 kind;
    u32 sub;
    u32 add;
    u32 opt_bad;  // K_TCP: the option area is malformed (reported only if the checksum is good)
};

// TCP option area (extension; RFC 9293 3.1 / 3.2, RFC 7323, RFC 2018 lengths): `byte(i)` gives byte i of the
// area, 0 <= i < optlen <= 40.  Returns 1 when it is malformed.
template <class B>
__device__ __forceinline__ u32 tcp_options_malformed(const B& byte, u32 optlen) {
    u32 pos = 0;
    while (pos < optlen) {
        const u32 kind = byte(pos);
        if (kind == 0u) break;  // end of option list: the rest is padding
        if (kind == 1u) { pos += 1u; continue; }  // no-operation
        if (pos + 1u >= optlen) return 1u;  // no room for the length byte
        const u32 l = byte(pos + 1u);
        if (l < 2u || pos + l > optlen) return 1u;
        if ((kind == 2u && l != 4u) || (kind == 3u && l != 3u) || (kind == 4u && l != 2u) || (kind == 8u && l != 10u) ||
            (kind == 5u && (l < 10u || ((l - 2u) & 7u) != 0u)))
            return 1u;
        pos += l;
    }
    return 0u;
}

// Ipv4Header::try_from (src/protocols/ipv4.rs:185-207) + validate (:128-157) on the five
// little-endian dwords of the first 20 header bytes.  Returns the packed result.
__device__ __forceinline__ u32 ipv4_header_verdict(u32 w0, u32 w1, u32 w2, u32 w3, u32 w4, u32 ext_tcp) {
    const u32 proto = (w2 >> 8) & 0xffu;  // byte 9: converted first (ipv4.rs:189, transport.rs:18-26)
    if (!(proto == 1u || proto == 17u || (ext_tcp != 0u && proto == 6u))) return pack_result(V_DROP_L3_PROTO, proto, 0);
    const u32 vhl = w0 & 0xffu;
    if ((vhl >> 4) != 4u) return pack_result(V_DROP_L3_VERSION, proto, 0);  // :129-133
    const u32 hl = (vhl & 0x0fu) * 4u;                                     // :116-118
    if (hl < 20u) return pack_result(V_DROP_L3_IHL_SHORT, proto, 0);       // :134-138
    if (hl >= 60u) return pack_result(V_DROP_L3_IHL_LONG, proto, 0);       // :139-143
    const u32 ff = bswap16(w1 >> 16);                                      // bytes 6..7, big-endian
    if (((ff >> 13) & 1u) != 0u || (ff & 0x1fffu) != 0u) return pack_result(V_DROP_L3_FRAGMENT, proto, 0);  // :145-147
    // validate_checksum over to_bytes(): exactly the first 20 bytes (:152-182)
    const u32 ck = finish(halves(w0) + halves(w1) + halves(w2) + halves(w3) + halves(w4));
    if (ck != 0u) return pack_result(V_DROP_L3_CSUM, proto, ck);
    return pack_result(V_L3_HEADER_OK, proto, 0);
}

// 4 bytes at L3-relative offset `rel` (a multiple of 4) as a little-endian dword.  In MODE_L2
// the IPv4 header starts 14 bytes into the record, so every such dword straddles two record
// dwords.
template <int MODE, class F>
__device__ __forceinline__ u32 l3raw(const F& f, u32 rel) {
    if (MODE == MODE_L3) {
        return f(rel >> 2);
    } else {
        const u32 k = (rel + 12u) >> 2;
        return (f(k) >> 16) | (f(k + 1) << 16);
    }
}

// `f(k)` = dword k of the record for the header bytes every check can reach within the first 96 bytes;
// `deep(k)` the same for any dword inside the record (only the TCP option walk needs it).
template <int MODE, class F, class D>
__device__ __forceinline__ Parsed parse_record(const F& f, const D& deep, u32 flen, const IfaceDev& ifc) {
    Parsed P;
    P.kind = K_DONE;
    P.sub = 0;
    P.add = 0;
    P.opt_bad = 0;
    u32 l3len = flen;
    u32 l2sum = 0;
    if (MODE == MODE_L2) {
        // EthernetHeader::try_from + ethernet::recv, src/devices/ethernet.rs:46-75
        if (flen < 14u) { P.res = pack_result(V_DROP_L2_TRUNCATED, 0, 0); return P; }
        const u32 d0 = f(0), d1 = f(1), d2 = f(2), d3 = f(3);
        const u32 ethertype = bswap16(d3 & 0xffffu);
        if (!(ethertype == 0x0800u || ethertype == 0x0806u)) {  // src/protocols.rs:30-43, before the MAC filter
            P.res = pack_result(V_DROP_L2_ETHERTYPE, 0, 0); return P;
        }
        const bool mine = d0 == ifc.mac_lo && (d1 & 0xffffu) == ifc.mac_hi;
        const bool bcast = d0 == 0xffffffffu && (d1 & 0xffffu) == 0xffffu;
        if (!mine && !bcast) { P.res = pack_result(V_DROP_L2_NOT_FOR_ME, 0, 0); return P; }  // :63-71
        if (ethertype == 0x0806u) {
            // arp::recv, src/protocols/arp.rs:206-246: header checks in order, then "is the target us?"
            const u32 alen = flen - 14u;
            if (alen < 8u) { P.res = pack_result(V_DROP_ARP_TRUNCATED, 0, 0); return P; }       // [0..8] panics, :211
            const u32 a0 = l3raw<MODE>(f, 0), a1 = l3raw<MODE>(f, 4);  // ARP bytes 0..7 (same 14-byte skew as IPv4)
            if (bswap16(a0 & 0xffffu) != 1u) { P.res = pack_result(V_DROP_ARP_HTYPE, 0, 0); return P; }       // :212-214
            if (bswap16(a0 >> 16) != 0x0800u) { P.res = pack_result(V_DROP_ARP_PTYPE, 0, 0); return P; }      // :215-217
            if ((a1 & 0xffu) != 6u) { P.res = pack_result(V_DROP_ARP_HLEN, 0, 0); return P; }                 // :218-220
            if (((a1 >> 8) & 0xffu) != 4u) { P.res = pack_result(V_DROP_ARP_PLEN, 0, 0); return P; }          // :221-223
            if (alen < 28u) { P.res = pack_result(V_DROP_ARP_TRUNCATED, 0, 0); return P; }      // ArpMessage::from, :225
            const u32 tpa = bswap32(l3raw<MODE>(f, 24));
            P.res = pack_result(tpa == ifc.unicast ? V_PASS_ARP : V_IGNORE_ARP_NOT_FOR_ME, 0, 0);              // :238-246
            return P;
        }
        l3len = flen - 14u;  // everything after the header, padding included (:73)
        l2sum = halves(d0) + halves(d1) + halves(d2) + (d3 & 0xffffu);
    }
    // ipv4::recv, src/protocols/ipv4.rs:394-422
    if (l3len < 20u) { P.res = pack_result(V_DROP_L3_TRUNCATED, 0, 0); return P; }  // try_from indexes [9]..[19]
    const u32 w0 = l3raw<MODE>(f, 0), w1 = l3raw<MODE>(f, 4), w2 = l3raw<MODE>(f, 8);
    const u32 w3 = l3raw<MODE>(f, 12), w4 = l3raw<MODE>(f, 16);
    const u32 hres = ipv4_header_verdict(w0, w1, w2, w3, w4, ifc.ext_tcp);
    if (((hres >> 16) & 0xffu) != V_L3_HEADER_OK) { P.res = hres; return P; }
    const u32 proto = hres >> 24;
    const u32 dst = bswap32(w4);
    if (dst != ifc.unicast && dst != ifc.broadcast && dst != 0xffffffffu) {  // :402-407
        P.res = pack_result(V_IGNORE_L3_NOT_FOR_ME, proto, 0); return P;
    }
    const u32 hl = (w0 & 0x0fu) * 4u;
    if (hl > l3len) { P.res = pack_result(V_DROP_L3_TRUNCATED, proto, 0); return P; }  // &data[hl..len] panics, :415
    const u32 plen = l3len - hl;  // total_length is never consulted (TODO at :144)
    // exact sum of everything in front of the L4 segment
    u32 prefix = l2sum + halves(w0) + halves(w1) + halves(w2) + halves(w3) + halves(w4);
    for (u32 rel = 20u; rel < hl; rel += 4u) prefix += halves(l3raw<MODE>(f, rel));

    if (proto == 1u) {
        // icmp::recv, src/transport/icmp.rs:97-122; try_from (:48-63) converts the type byte
        // before it indexes bytes 1..7, and never looks at the checksum.
        if (plen < 1u) { P.res = pack_result(V_DROP_ICMP_TRUNCATED, proto, 0); return P; }
        const u32 m0 = l3raw<MODE>(f, hl);
        const u32 type = m0 & 0xffu;
        if (!(type == 0u || type == 8u)) { P.res = pack_result(V_DROP_ICMP_TYPE, proto, 0); return P; }
        if (plen < 8u) { P.res = pack_result(V_DROP_ICMP_TRUNCATED, proto, 0); return P; }
        if (type == 0u) { P.res = pack_result(V_ACCEPT_ICMP_ECHOREPLY, proto, 0); return P; }
        // Echo: the reply is the same message with type 0 and a fresh checksum (:110-120, :75-85).
        P.kind = K_ECHO;
        P.sub = prefix + (m0 & 0xffu) + (m0 >> 16);
        return P;
    }
    if (proto == 17u) {
        // udp::recv, src/transport/udp.rs:219-254
        if (plen < 8u) { P.res = pack_result(V_DROP_UDP_SHORT, proto, 0); return P; }  // :225-228
        const u32 u1 = l3raw<MODE>(f, hl + 4u);
        const u32 length = bswap16(u1 & 0xffffu);
        if (plen != length) { P.res = pack_result(V_DROP_UDP_LEN, proto, 0); return P; }  // :232-238
        P.kind = K_UDP;
        P.sub = prefix;
        P.add = halves(w3) + halves(w4) + 0x1100u + (u1 & 0xffffu);  // src, dst, 0x00 0x11, length (:26-36)
        return P;
    }
    // TCP extension (ext_tcp only; the reference drops protocol 6 at the top).
    if (plen < 20u) { P.res = pack_result(V_DROP_TCP_SHORT, proto, 0); return P; }
    const u32 t3 = l3raw<MODE>(f, hl + 12u);
    const u32 doff = (t3 >> 4) & 0x0fu;
    if (doff < 5u || doff * 4u > plen) { P.res = pack_result(V_DROP_TCP_DOFF, proto, 0); return P; }
    P.kind = K_TCP;
    P.sub = prefix;
    P.add = halves(w3) + halves(w4) + 0x0600u + bswap16(plen);
    if (doff > 5u) {
        const u32 at = (MODE == MODE_L2 ? 14u : 0u) + hl + 20u;  // record byte where the options start
        P.opt_bad = tcp_options_malformed([&](u32 i) { const u32 b = at + i; return (deep(b >> 2) >> (8u * (b & 3u))) & 0xffu; }, doff * 4u - 20u);
    }
    return P;
}

// Second half: the record's byte sum is known.
__device__ __forceinline__ u32 finalize_record(const Parsed& P, u32 total) {
    const u32 ck = finish(total - P.sub + P.add);
    if (P.kind == K_UDP) return ck != 0u ? pack_result(V_DROP_UDP_CSUM, 17, ck) : pack_result(V_ACCEPT_UDP, 17, 0);  // udp.rs:247-254
    if (P.kind == K_TCP) return ck != 0u ? pack_result(V_DROP_TCP_CSUM, 6, ck) : pack_result(P.opt_bad ? V_DROP_TCP_OPTION : V_ACCEPT_TCP, 6, 0);
    return pack_result(V_ACCEPT_ICMP_ECHO, 1, ck);
}

}  // namespace unetdev
