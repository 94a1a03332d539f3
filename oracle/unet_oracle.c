/*
 * unet_oracle.c -- TEST INFRASTRUCTURE (see unet_oracle.h): scalar CPU restatement of the
 * checksum / parse / verdict path of ikanago/unet @ 2024_10_08.  Every function cites the
 * reference lines it follows.  Written from the behaviour of those lines, in C; the
 * control flow deliberately keeps the reference's order of checks because "first failing
 * check wins" is part of the contract.
 */
#include "unet_oracle.h"

#include <string.h>

/* ------------------------------------------------------------------ a1 */

/* src/utils.rs:1-17.  The seed is added to a u32 accumulator (:2); the data is consumed in
 * 2-byte chunks read big-endian, a trailing single byte becoming the HIGH byte of a word
 * whose low byte is zero (:6-7); the u32 sum (:12) is folded while any bit above 16 is set
 * (:13-15) and complemented (:16). */
uint16_t oracle_calculate_checksum(const uint8_t* data, size_t len, uint16_t sum) {
    uint32_t acc = sum;
    size_t i = 0;
    while (i < len) {
        size_t chunk = len - i >= 2 ? 2 : 1;
        uint32_t word;
        if (chunk == 1) {
            word = (uint32_t)data[i] << 8;
        } else {
            word = ((uint32_t)data[i] << 8) | data[i + 1];
        }
        acc += word; /* u32 wrap-around like release-mode Rust; unreachable below 131 KB */
        i += chunk;
    }
    while ((acc >> 16) != 0) {
        acc = (acc & 0xffffu) + (acc >> 16);
    }
    return (uint16_t)~acc;
}

/* Same value, different arithmetic: 64-bit accumulator over native-endian 32-bit loads, one
 * byte swap at the end (the ones'-complement sum commutes with a byte swap of every word).
 * Not in the reference; it is the "best effort CPU" comparison line and a second,
 * independently written implementation the tests check the first one against. */
uint16_t oracle_calculate_checksum_fast(const uint8_t* data, size_t len, uint16_t sum) {
    uint64_t acc = 0;
    size_t i = 0;
    for (; i + 16 <= len; i += 16) {
        uint32_t w[4];
        memcpy(w, data + i, 16);
        acc += (uint64_t)w[0] + w[1] + w[2] + w[3];
    }
    for (; i + 4 <= len; i += 4) {
        uint32_t w;
        memcpy(&w, data + i, 4);
        acc += w;
    }
    if (i < len) {
        uint32_t w = 0;
        memcpy(&w, data + i, len - i); /* missing bytes stay zero: odd tail is padded low */
        acc += w;
    }
    /* fold the little-endian-word sum to 16 bits */
    acc = (acc & 0xffffffffu) + (acc >> 32);
    acc = (acc & 0xffffffffu) + (acc >> 32);
    acc = (acc & 0xffffu) + (acc >> 16);
    acc = (acc & 0xffffu) + (acc >> 16);
    acc = (acc & 0xffffu) + (acc >> 16);
    uint32_t le = (uint32_t)acc;
    uint32_t be = ((le & 0xffu) << 8) | (le >> 8); /* back to the big-endian word domain */
    uint32_t tot = be + sum;
    while ((tot >> 16) != 0) {
        tot = (tot & 0xffffu) + (tot >> 16);
    }
    return (uint16_t)~tot;
}

/* ------------------------------------------------------------ helpers */

/* Which checksum routine the verdict code calls: the faithful 2-byte loop (default) or the 64-bit
 * accumulator.  Thread-local so that concurrent batch drivers can differ. */
static __thread int g_use_fast = 0;
static uint16_t csum_of(const uint8_t* data, size_t len, uint16_t sum) {
    return g_use_fast ? oracle_calculate_checksum_fast(data, len, sum) : oracle_calculate_checksum(data, len, sum);
}

static uint16_t be16(const uint8_t* p) { return (uint16_t)(((uint16_t)p[0] << 8) | p[1]); }
static uint32_t be32(const uint8_t* p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}
static void put_be16(uint8_t* p, uint16_t v) {
    p[0] = (uint8_t)(v >> 8);
    p[1] = (uint8_t)v;
}
static void put_be32(uint8_t* p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}
static oracle_result mk(uint8_t verdict, uint8_t proto, uint16_t csum) {
    oracle_result r;
    r.csum = csum;
    r.verdict = verdict;
    r.proto = proto;
    return r;
}

/* src/protocols/ipv4.rs:220 */
uint32_t oracle_iface_broadcast(uint32_t unicast, uint32_t netmask) {
    return (unicast & 0xffffff00u) | (~netmask & 0xffu);
}

/* ------------------------------------------------------------ a2-a4 */

/* Ipv4Header::try_from (src/protocols/ipv4.rs:185-207) then validate (:128-150).
 * try_from's first statement converts byte 9 to TransportProtocolNumber (:189, enum at
 * src/transport.rs:10-28), so an unknown protocol fails before the version is looked at. */
oracle_result oracle_ipv4_header_validate(const uint8_t* b, int ext_tcp) {
    uint8_t proto = b[9];
    if (!(proto == 1 || proto == 17 || (ext_tcp && proto == 6))) {
        return mk(OV_DROP_L3_PROTO, proto, 0);
    }
    uint8_t vhl = b[0];
    uint16_t flags_frag = be16(b + 6);
    /* validate(), :128-150 */
    if ((vhl >> 4) != 4) return mk(OV_DROP_L3_VERSION, proto, 0);                 /* :129-133 */
    uint8_t hl = (uint8_t)((vhl & 0x0f) * 4);                                     /* :116-118 */
    if (hl < 20) return mk(OV_DROP_L3_IHL_SHORT, proto, 0);                       /* :134-138 */
    if (!(hl < 60)) return mk(OV_DROP_L3_IHL_LONG, proto, 0);                     /* :139-143, const :17 */
    uint8_t flags = (uint8_t)(flags_frag >> 13);                                  /* :120-122 */
    if ((flags & 0x1) > 0 || (flags_frag & 0x1fff) > 0) {                         /* :145-147 */
        return mk(OV_DROP_L3_FRAGMENT, proto, 0);
    }
    /* validate_checksum (:152-157) sums to_bytes() (:159-182): a fresh 20-byte vector built
     * from the parsed fields.  Each field round-trips to its original bytes, and the options
     * of a longer header are NOT part of it. */
    uint8_t copy[20];
    memcpy(copy, b, 20);
    uint16_t ck = csum_of(copy, 20, 0);
    if (ck != 0) return mk(OV_DROP_L3_CSUM, proto, ck);
    return mk(OV_L3_HEADER_OK, proto, 0);
}

/* ------------------------------------------------------------ a6 */

/* udp::recv up to the checksum decision, src/transport/udp.rs:219-254. */
static oracle_result udp_recv(const uint8_t* seg, size_t len, uint32_t src, uint32_t dst) {
    if (len < 8) return mk(OV_DROP_UDP_SHORT, 17, 0);                             /* :225-228 */
    uint16_t sum = csum_of(seg, len, 0);                        /* :229 */
    uint16_t length = be16(seg + 4);                                              /* :57-66 */
    if (len != (size_t)length) return mk(OV_DROP_UDP_LEN, 17, 0);                 /* :232-238 */
    uint8_t pseudo[12];                                                           /* :26-36, :240-246 */
    put_be32(pseudo, src);
    put_be32(pseudo + 4, dst);
    pseudo[8] = 0;
    pseudo[9] = 17;
    put_be16(pseudo + 10, length);
    uint16_t fin = csum_of(pseudo, 12, (uint16_t)~sum);         /* :247 */
    if (fin != 0) return mk(OV_DROP_UDP_CSUM, 17, fin);                           /* :248-254 */
    return mk(OV_ACCEPT_UDP, 17, 0);
}

/* TCP extension (nothing like it in the reference): same pseudo-header arithmetic with
 * protocol 6 and the buffer's segment length, plus the data-offset sanity check. */
/* Option area of a TCP header (bytes 20 .. data offset): 1 when malformed.  No counterpart in the reference;
 * lengths per RFC 9293 (MSS 4), RFC 7323 (window scale 3, timestamps 10), RFC 2018 (SACK permitted 2, SACK 2 + 8k). */
static int tcp_options_malformed(const uint8_t* opt, uint32_t optlen) {
    uint32_t pos = 0;
    while (pos < optlen) {
        uint32_t kind = opt[pos];
        if (kind == 0) break;
        if (kind == 1) { pos++; continue; }
        if (pos + 1 >= optlen) return 1;
        uint32_t l = opt[pos + 1];
        if (l < 2 || pos + l > optlen) return 1;
        if ((kind == 2 && l != 4) || (kind == 3 && l != 3) || (kind == 4 && l != 2) || (kind == 8 && l != 10) ||
            (kind == 5 && (l < 10 || (l - 2) % 8 != 0)))
            return 1;
        pos += l;
    }
    return 0;
}

static oracle_result tcp_recv_ext(const uint8_t* seg, size_t len, uint32_t src, uint32_t dst) {
    if (len < 20) return mk(OV_DROP_TCP_SHORT, 6, 0);
    uint32_t doff = seg[12] >> 4;
    if (doff < 5 || doff * 4 > len) return mk(OV_DROP_TCP_DOFF, 6, 0);
    uint16_t sum = csum_of(seg, len, 0);
    uint8_t pseudo[12];
    put_be32(pseudo, src);
    put_be32(pseudo + 4, dst);
    pseudo[8] = 0;
    pseudo[9] = 6;
    put_be16(pseudo + 10, (uint16_t)len);
    uint16_t fin = csum_of(pseudo, 12, (uint16_t)~sum);
    if (fin != 0) return mk(OV_DROP_TCP_CSUM, 6, fin);
    if (tcp_options_malformed(seg + 20, doff * 4 - 20)) return mk(OV_DROP_TCP_OPTION, 6, 0);
    return mk(OV_ACCEPT_TCP, 6, 0);
}

/* ------------------------------------------------------------ a8, a9 */

/* icmp::send's checksum, src/transport/icmp.rs:75-85: header with checksum 0, then data. */
uint16_t oracle_icmp_send_checksum(uint8_t type, uint8_t code, uint32_t values, const uint8_t* data,
                                   size_t data_len) {
    uint8_t hdr[8];
    hdr[0] = type;
    hdr[1] = code;
    hdr[2] = 0;
    hdr[3] = 0;
    put_be32(hdr + 4, values);
    /* ck(hdr || data, 0): chain through the seed.  An 8-byte prefix keeps word alignment. */
    uint16_t part = csum_of(hdr, 8, 0);
    return csum_of(data, data_len, (uint16_t)~part);
}

/* icmp::recv, src/transport/icmp.rs:97-122.  IcmpHeader::try_from (:48-63) reads data[0]
 * and converts it (:52, failing on an unknown type) before it touches data[1..8]; a short
 * buffer therefore panics only if the type byte exists and is valid.  The checksum field is
 * parsed (:54) and never verified.  An Echo makes the stack send an EchoReply with the same
 * code, values and data[8..] (:110-120); its checksum is what icmp::send computes. */
static oracle_result icmp_recv(const uint8_t* msg, size_t len) {
    if (len < 1) return mk(OV_DROP_ICMP_TRUNCATED, 1, 0);
    uint8_t type = msg[0];
    if (!(type == 0 || type == 8)) return mk(OV_DROP_ICMP_TYPE, 1, 0);            /* :13-23 */
    if (len < 8) return mk(OV_DROP_ICMP_TRUNCATED, 1, 0);
    if (type == 8) {
        uint16_t reply = oracle_icmp_send_checksum(0, msg[1], be32(msg + 4), msg + 8, len - 8);
        return mk(OV_ACCEPT_ICMP_ECHO, 1, reply);
    }
    return mk(OV_ACCEPT_ICMP_ECHOREPLY, 1, 0);
}

/* ------------------------------------------------------------ a5 */

/* ipv4::recv, src/protocols/ipv4.rs:394-422. */
oracle_result oracle_ipv4_recv(const uint8_t* data, size_t len, const oracle_iface* iface) {
    if (len < 20) return mk(OV_DROP_L3_TRUNCATED, 0, 0); /* try_from indexes value[9]..value[19] */
    oracle_result h = oracle_ipv4_header_validate(data, iface->ext_tcp);           /* :400-401 */
    if (h.verdict != OV_L3_HEADER_OK) return h;
    uint8_t proto = data[9];
    uint32_t src = be32(data + 12), dst = be32(data + 16);
    if (dst != iface->unicast && dst != iface->broadcast && dst != 0xffffffffu) { /* :402-407 */
        return mk(OV_IGNORE_L3_NOT_FOR_ME, proto, 0);
    }
    size_t hl = (size_t)(data[0] & 0x0f) * 4;
    if (hl > len) return mk(OV_DROP_L3_TRUNCATED, proto, 0); /* &data[hl..len] panics, :415 */
    const uint8_t* payload = data + hl; /* total_length is never consulted (TODO at :144) */
    size_t plen = len - hl;
    if (proto == 1) return icmp_recv(payload, plen);                              /* :417 */
    if (proto == 17) return udp_recv(payload, plen, src, dst);                    /* :418 */
    return tcp_recv_ext(payload, plen, src, dst); /* only reachable with ext_tcp */
}

/* arp::recv up to its decision, src/protocols/arp.rs:206-246.  ArpHeader::from takes data[0..8]
 * (:211, panics below 8 bytes); the four field checks follow in order (:212-223); ArpMessage::from
 * (:225) slices up to byte 28 (panics below 28); a message whose target protocol address is ours makes
 * the stack cache the sender and reply (:238-245), any other is ignored. */
static oracle_result arp_recv(const uint8_t* data, size_t len, const oracle_iface* iface) {
    if (len < 8) return mk(OV_DROP_ARP_TRUNCATED, 0, 0);
    if (be16(data) != 1) return mk(OV_DROP_ARP_HTYPE, 0, 0);
    if (be16(data + 2) != 0x0800) return mk(OV_DROP_ARP_PTYPE, 0, 0);
    if (data[4] != 6) return mk(OV_DROP_ARP_HLEN, 0, 0);
    if (data[5] != 4) return mk(OV_DROP_ARP_PLEN, 0, 0);
    if (len < 28) return mk(OV_DROP_ARP_TRUNCATED, 0, 0);
    if (be32(data + 24) == iface->unicast) return mk(OV_PASS_ARP, 0, 0);
    return mk(OV_IGNORE_ARP_NOT_FOR_ME, 0, 0);
}

/* ------------------------------------------------------------ a11 */

/* ethernet::recv, src/devices/ethernet.rs:58-75, then the queue hand-off of
 * src/devices.rs:188-201 and ipv4::recv.  EthernetHeader::try_from (:46-55) fails on an
 * unknown ethertype (src/protocols.rs:30-43) before the MAC filter (:63-71) runs; the L3
 * packet is everything after byte 14, padding included (:73). */
oracle_result oracle_ethernet_recv(const uint8_t* frame, size_t len, const oracle_iface* iface) {
    static const uint8_t bcast[6] = {0xff, 0xff, 0xff, 0xff, 0xff, 0xff};
    if (len < 14) return mk(OV_DROP_L2_TRUNCATED, 0, 0);
    uint16_t ethertype = be16(frame + 12);
    if (!(ethertype == 0x0800 || ethertype == 0x0806)) return mk(OV_DROP_L2_ETHERTYPE, 0, 0);
    if (memcmp(frame, iface->mac, 6) != 0 && memcmp(frame, bcast, 6) != 0) {
        return mk(OV_DROP_L2_NOT_FOR_ME, 0, 0);
    }
    if (ethertype == 0x0806) return arp_recv(frame + 14, len - 14, iface);
    return oracle_ipv4_recv(frame + 14, len - 14, iface);
}

oracle_result oracle_recv(const uint8_t* rec, size_t len, const oracle_iface* iface) {
    return iface->mode ? oracle_ethernet_recv(rec, len, iface) : oracle_ipv4_recv(rec, len, iface);
}

/* ------------------------------------------------------------ a7, a10 */

/* udp::send, src/transport/udp.rs:181-203: s = ck(pseudo, 0); c = ck(hdr(cksum 0) || payload, !s).
 * The result is stored as is: a computed 0x0000 is NOT replaced by 0xffff. */
uint16_t oracle_udp_send_checksum(uint32_t src, uint32_t dst, uint16_t sport, uint16_t dport,
                                  const uint8_t* payload, size_t payload_len) {
    uint16_t length = (uint16_t)(8 + payload_len);                                /* :181 */
    uint8_t pseudo[12];
    put_be32(pseudo, src);
    put_be32(pseudo + 4, dst);
    pseudo[8] = 0;
    pseudo[9] = 17;
    put_be16(pseudo + 10, length);
    uint8_t hdr[8];
    put_be16(hdr, sport);
    put_be16(hdr + 2, dport);
    put_be16(hdr + 4, length);
    put_be16(hdr + 6, 0);
    uint16_t s = csum_of(pseudo, 12, 0);                        /* :200 */
    /* ck(hdr || payload, !s), chained through the seed across the 8-byte header */
    uint16_t h = csum_of(hdr, 8, (uint16_t)~s);
    return csum_of(payload, payload_len, (uint16_t)~h);         /* :202 */
}

/* create_ip_header, src/protocols/ipv4.rs:366-391. */
void oracle_create_ip_header(uint16_t id, uint8_t protocol, uint32_t src, uint32_t dst, size_t data_len,
                             uint8_t out[20]) {
    uint16_t total = (uint16_t)(20 + (uint16_t)data_len);                         /* :373 (wraps like `as u16`) */
    out[0] = 0x45;
    out[1] = 0;
    put_be16(out + 2, total);
    put_be16(out + 4, id);
    put_be16(out + 6, 0);
    out[8] = 64;
    out[9] = protocol;
    out[10] = 0;
    out[11] = 0;
    put_be32(out + 12, src);
    put_be32(out + 16, dst);
    uint16_t ck = csum_of(out, 20, 0);                          /* :387 */
    put_be16(out + 10, ck);                                                       /* :388-389 */
}

/* Fill the checksum fields of a ready-made L3 packet the way the send functions would have:
 * IPv4 header field from the 20 header bytes with the field zeroed (ipv4.rs:374-389), UDP
 * field with pseudo-header seed chaining (udp.rs:200-203), ICMP field over the whole
 * message (icmp.rs:81-85).  Only the shapes the reference can emit are filled: version 4,
 * IHL 5, protocol 1 or 17. */
uint8_t oracle_generate(uint8_t* pkt, size_t len) {
    if (len < 20) return OV_DROP_L3_TRUNCATED;
    uint8_t proto = pkt[9];
    if (!(proto == 1 || proto == 17)) return OV_DROP_L3_PROTO;
    if ((pkt[0] >> 4) != 4) return OV_DROP_L3_VERSION;
    if ((pkt[0] & 0x0f) < 5) return OV_DROP_L3_IHL_SHORT;
    if ((pkt[0] & 0x0f) > 5) return OV_DROP_L3_IHL_LONG;
    size_t seglen = len - 20;
    if (seglen < 8) return proto == 17 ? OV_DROP_UDP_SHORT : OV_DROP_ICMP_TRUNCATED;
    uint8_t* seg = pkt + 20;
    if (proto == 17) {
        uint8_t pseudo[12];
        memcpy(pseudo, pkt + 12, 8);
        pseudo[8] = 0;
        pseudo[9] = 17;
        put_be16(pseudo + 10, (uint16_t)seglen);
        seg[6] = 0;
        seg[7] = 0;
        uint16_t s = csum_of(pseudo, 12, 0);
        uint16_t c = csum_of(seg, seglen, (uint16_t)~s);
        put_be16(seg + 6, c);
    } else {
        seg[2] = 0;
        seg[3] = 0;
        uint16_t c = csum_of(seg, seglen, 0);
        put_be16(seg + 2, c);
    }
    pkt[10] = 0;
    pkt[11] = 0;
    uint16_t hc = csum_of(pkt, 20, 0);
    put_be16(pkt + 10, hc);
    return 0;
}

/* One whole transmitted packet: udp::send (udp.rs:168-216) or icmp::send (icmp.rs:66-95) ->
 * ipv4::send (ipv4.rs:300-364: the header's source is interface.unicast) -> in L2 mode the Ethernet
 * header and tap::send's padding (tap.rs:138-152: zero bytes until the L3 packet -- not the frame --
 * is 60 bytes long).  Returns the number of bytes written to out (0 with *status != 0 otherwise).
 * Payloads over 65535-20-8 bytes are refused for both protocols: UDP by udp.rs:174-180; for ICMP the
 * reference would let `total_length as u16` wrap (ipv4.rs:373), which this build reports instead. */
uint32_t oracle_tx_build(const oracle_tx_desc* d, const uint8_t* payload, size_t plen, const oracle_iface* iface,
                         uint8_t* out, uint8_t* status) {
    uint8_t st = 0;
    if (!(d->proto == 1 || d->proto == 17)) st = 1;
    else if (plen > 65535u - 20u - 8u) st = 2;
    if (status) *status = st;
    if (st) return 0;
    uint8_t* p = out;
    if (iface->mode == 1) {
        memcpy(p, d->dst_mac, 6);                                                 /* tap.rs:138-142 */
        memcpy(p + 6, iface->mac, 6);
        p[12] = 0x08;
        p[13] = 0x00;
        p += 14;
    }
    size_t l4len = 8 + plen;
    oracle_create_ip_header(d->ip_id, d->proto, iface->unicast, d->dst, l4len, p); /* ipv4.rs:337-339 */
    uint8_t* l4 = p + 20;
    if (d->proto == 17) {
        put_be16(l4, d->l4a);
        put_be16(l4 + 2, d->l4b);
        put_be16(l4 + 4, (uint16_t)l4len);
        put_be16(l4 + 6, oracle_udp_send_checksum(d->src, d->dst, d->l4a, d->l4b, payload, plen));
    } else {
        l4[0] = (uint8_t)(d->l4a >> 8);
        l4[1] = (uint8_t)(d->l4a & 0xff);
        put_be16(l4 + 2, oracle_icmp_send_checksum(l4[0], l4[1], d->values, payload, plen));
        put_be32(l4 + 4, d->values);
    }
    if (plen) memcpy(l4 + 8, payload, plen);
    size_t l3len = 20 + l4len;
    if (iface->mode == 1 && l3len < 60) {                                         /* tap.rs:147-152 */
        memset(p + l3len, 0, 60 - l3len);
        l3len = 60;
    }
    return (uint32_t)((size_t)(p - out) + l3len);
}

/* ------------------------------------------------------------ batches */

static const uint8_t* rec_ptr(const uint8_t* arena, const uint32_t* off16, uint32_t stride16, uint32_t i) {
    uint64_t slot = off16 ? (uint64_t)off16[i] : (uint64_t)i * stride16;
    return arena + slot * 16u;
}

void oracle_recv_batch(const uint8_t* arena, const uint32_t* off16, uint32_t stride16,
                       const uint32_t* len, uint32_t n, const oracle_iface* iface, oracle_result* out) {
    for (uint32_t i = 0; i < n; i++) {
        out[i] = oracle_recv(rec_ptr(arena, off16, stride16, i), len[i], iface);
    }
}

void oracle_recv_batch_fast(const uint8_t* arena, const uint32_t* off16, uint32_t stride16,
                            const uint32_t* len, uint32_t n, const oracle_iface* iface, oracle_result* out) {
    g_use_fast = 1;
    oracle_recv_batch(arena, off16, stride16, len, n, iface, out);
    g_use_fast = 0;
}

void oracle_csum_batch(const uint8_t* arena, const uint32_t* off16, uint32_t stride16,
                       const uint32_t* len, const uint16_t* seeds, uint32_t n, uint16_t* out) {
    for (uint32_t i = 0; i < n; i++) {
        out[i] = csum_of(rec_ptr(arena, off16, stride16, i), len[i],
                                           seeds ? seeds[i] : (uint16_t)0);
    }
}

void oracle_ipv4_validate_batch(const uint8_t* hdrs, uint32_t n, int ext_tcp, oracle_result* out) {
    for (uint32_t i = 0; i < n; i++) {
        out[i] = oracle_ipv4_header_validate(hdrs + (size_t)i * 20, ext_tcp);
    }
}

void oracle_generate_batch(uint8_t* arena, const uint32_t* off16, uint32_t stride16, const uint32_t* len,
                           uint32_t n, uint8_t* status) {
    for (uint32_t i = 0; i < n; i++) {
        uint8_t s = oracle_generate((uint8_t*)rec_ptr(arena, off16, stride16, i), len[i]);
        if (status) status[i] = s;
    }
}

void oracle_tx_build_batch(const uint8_t* pay_arena, const uint32_t* pay_off16, uint32_t pay_stride16,
                           const uint32_t* pay_len, const oracle_tx_desc* desc, uint32_t n, const oracle_iface* iface,
                           uint8_t* out_arena, const uint32_t* out_off16, uint32_t* out_len, uint8_t* status) {
    for (uint32_t i = 0; i < n; i++) {
        uint8_t st = 0;
        out_len[i] = oracle_tx_build(&desc[i], rec_ptr(pay_arena, pay_off16, pay_stride16, i), pay_len[i], iface,
                                     out_arena + (size_t)out_off16[i] * 16u, &st);
        if (status) status[i] = st;
    }
}

/* ------------------------------------------------------------ flow hash */

static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

/* MurmurHash3 x86_32 of the four words {src, dst, sport<<16|dport, proto} with seed "unet". */
uint32_t oracle_flow_hash5(uint32_t src, uint32_t dst, uint8_t proto, uint16_t sport, uint16_t dport) {
    uint32_t w[4] = {src, dst, ((uint32_t)sport << 16) | dport, proto};
    uint32_t h = 0x756e6574u;
    for (int i = 0; i < 4; i++) {
        uint32_t k = w[i] * 0xcc9e2d51u;
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

uint32_t oracle_flow_hash_record(const uint8_t* rec, size_t len, const oracle_iface* iface) {
    if (iface->mode) {
        if (len < 14 || be16(rec + 12) != 0x0800) return 0;
        rec += 14;
        len -= 14;
    }
    if (len < 20) return 0;
    uint8_t proto = rec[9];
    size_t hl = (size_t)(rec[0] & 0x0f) * 4;
    uint16_t sport = 0, dport = 0;
    if ((proto == 17 || proto == 6) && hl >= 20 && hl + 4 <= len) {
        sport = be16(rec + hl);
        dport = be16(rec + hl + 2);
    }
    return oracle_flow_hash5(be32(rec + 12), be32(rec + 16), proto, sport, dport);
}

/* ------------------------------------------------------------ per-flow TCP tracking (extension) */

uint32_t oracle_tcp_slot(uint32_t src, uint32_t dst, uint16_t sport, uint16_t dport, uint32_t table_slots) {
    uint32_t l = 0;
    while ((1u << l) < table_slots) l++;
    if (l == 0) return 0;
    return (uint32_t)(oracle_flow_hash5(src, dst, 6, sport, dport) * 0x9E3779B1u) >> (32 - l);
}

void oracle_tcp_track(const uint8_t* arena, const uint32_t* off16, uint32_t stride16, const uint32_t* len,
                      const oracle_result* results, uint32_t n, const oracle_iface* iface, oracle_tcp_flow* table,
                      uint32_t table_slots, uint8_t* seg_state) {
    enum { FREE = 0, SYN = 1, EST = 2, FIN = 3, RESET = 4 };
    for (uint32_t i = 0; i < n; i++) {
        seg_state[i] = 0;
        if (results[i].verdict != OV_ACCEPT_TCP) continue;
        const uint8_t* d = arena + (size_t)(off16 ? off16[i] : (uint64_t)i * stride16) * 16;
        size_t l3len = len[i];
        if (iface->mode == 1) { d += 14; l3len -= 14; }
        size_t hl = (size_t)(d[0] & 0x0f) * 4;
        const uint8_t* t = d + hl;
        uint32_t src = be32(d + 12), dst = be32(d + 16);
        uint16_t sport = be16(t), dport = be16(t + 2);
        uint32_t seq = be32(t + 4), doff = t[12] >> 4, flags = t[13];
        uint32_t consumed = (uint32_t)(l3len - hl - doff * 4) + ((flags & 2) ? 1 : 0) + ((flags & 1) ? 1 : 0);
        uint32_t ports = ((uint32_t)sport << 16) | dport;
        oracle_tcp_flow* f = &table[oracle_tcp_slot(src, dst, sport, dport, table_slots)];
        int same = f->state != FREE && f->src == src && f->dst == dst && f->ports == ports;
        if (!same || (f->state == RESET && (flags & 2))) {
            f->src = src; f->dst = dst; f->ports = ports;
            f->next_seq = seq + consumed;
            f->state = (flags & 4) ? RESET : ((flags & 1) ? FIN : ((flags & 2) ? SYN : EST));
            f->segments = 1; f->out_of_order = 0; f->after_close = 0;
            seg_state[i] = 1;
            continue;
        }
        f->segments++;
        if (f->state == RESET) { f->after_close++; seg_state[i] = 4; continue; }
        if (seq == f->next_seq) {
            seg_state[i] = 2;
            f->next_seq = seq + consumed;
            if (flags & 1) f->state = FIN;
            else if (f->state == SYN && !(flags & 2)) f->state = EST;
        } else {
            seg_state[i] = 3;
            f->out_of_order++;
            if ((int32_t)(seq - f->next_seq) > 0) f->next_seq = seq + consumed;
        }
        if (flags & 4) f->state = RESET;
    }
}
