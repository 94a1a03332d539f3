/*
 * oracle_san_check.c -- TEST INFRASTRUCTURE.  The oracle is what every GPU result is compared with, so the oracle itself
 * must not read a byte past a record: this program feeds it records in heap buffers of exactly their length (random
 * bytes, and well-formed IPv4/UDP/ICMP/TCP/Ethernet/ARP headers with random fields, cut at every length), built with
 * the address and undefined-behaviour sanitizers (`make oracle_san_check`).  It also checks the two properties the
 * reference's own tests assert of the checksum (src/utils.rs:43-60): ck(d, ck(d, 0)) == 0, and that the 64-bit
 * variant agrees with the 2-byte loop.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "unet_oracle.h"

static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd64(void) {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
}
static uint32_t rnd(uint32_t n) { return (uint32_t)(rnd64() % n); }

#define REQUIRE(x)                                                      \
    do {                                                                \
        if (!(x)) {                                                     \
            printf("oracle_san_check: %s failed, line %d\n", #x, __LINE__); \
            exit(1);                                                    \
        }                                                               \
    } while (0)

/* a frame for iface: Ethernet header (mode 1), IPv4 header with random IHL, then an L4 header; fields mostly valid.
 * The headers are made in a scratch image and the record takes as much of it as it is long. */
static void make_frame(uint8_t* rec, size_t len, const oracle_iface* ifc) {
    uint8_t f[160];
    size_t at = 0;
    for (size_t i = 0; i < sizeof f; i++) f[i] = (uint8_t)rnd64();
    for (size_t i = 0; i < len; i++) rec[i] = (uint8_t)rnd64();
    if (ifc->mode == 1) {
        if (rnd(8)) memcpy(f, ifc->mac, 6);
        f[12] = 0x08;
        f[13] = rnd(6) == 0 ? 0x06 : 0x00; /* ARP now and then */
        at = 14;
    }
    const uint32_t ihl = rnd(10) == 0 ? rnd(16) : 5 + rnd(3);
    f[at] = (uint8_t)((rnd(20) == 0 ? rnd(16) : 4) << 4 | ihl);
    const size_t total = len > at ? len - at : 0; /* total length: what the buffer holds, most of the time */
    if (rnd(8)) f[at + 2] = (uint8_t)(total >> 8), f[at + 3] = (uint8_t)total;
    f[at + 6] = rnd(10) == 0 ? (uint8_t)rnd64() : 0;
    f[at + 7] = rnd(10) == 0 ? (uint8_t)rnd64() : 0;
    static const uint8_t protos[] = {1, 6, 17, 17, 6, 47};
    f[at + 9] = protos[rnd(6)];
    f[at + 10] = f[at + 11] = 0;
    if (rnd(4)) {
        const uint32_t dst = rnd(6) ? ifc->unicast : ifc->broadcast;
        f[at + 16] = (uint8_t)(dst >> 24), f[at + 17] = (uint8_t)(dst >> 16), f[at + 18] = (uint8_t)(dst >> 8), f[at + 19] = (uint8_t)dst;
    }
    if (rnd(6)) {
        const uint16_t ck = oracle_calculate_checksum(f + at, 20, 0);
        f[at + 10] = (uint8_t)(ck >> 8), f[at + 11] = (uint8_t)ck;
    }
    const size_t l4 = at + (size_t)ihl * 4;
    if (f[at + 9] == 6) f[l4 + 12] = (uint8_t)((5 + rnd(11)) << 4); /* TCP data offset 5..15: option walks */
    if (f[at + 9] == 17 && rnd(4)) { /* UDP length = what lies behind the IPv4 header */
        const size_t ul = len > l4 ? len - l4 : 0;
        f[l4 + 4] = (uint8_t)(ul >> 8), f[l4 + 5] = (uint8_t)ul;
    }
    memcpy(rec, f, len < sizeof f ? len : sizeof f);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
    oracle_iface ifc = {0xC0000202u, 0xC00002FFu, {2, 0xaa, 0xbb, 0xcc, 0xdd, 0xee}, 0, 1};
    for (int r = 0; r < rounds; r++) {
        ifc.mode = (uint8_t)rnd(2);
        ifc.ext_tcp = (uint8_t)rnd(2);
        /* every short length often, long ones now and then */
        const size_t len = rnd(4) == 0 ? rnd(65536) : (rnd(2) ? rnd(200) : rnd(2100));
        uint8_t* rec = malloc(len ? len : 1);
        REQUIRE(rec);
        if (rnd(3) == 0)
            for (size_t i = 0; i < len; i++) rec[i] = (uint8_t)rnd64();
        else
            make_frame(rec, len, &ifc);
        const oracle_result a = oracle_recv(rec, len, &ifc);
        REQUIRE(a.verdict < OV_COUNT);
        (void)oracle_flow_hash_record(rec, len, &ifc);
        const uint16_t seed = (uint16_t)rnd64();
        const uint16_t ck = oracle_calculate_checksum(rec, len, seed);
        REQUIRE(ck == oracle_calculate_checksum_fast(rec, len, seed));
        REQUIRE(oracle_calculate_checksum(rec, len, oracle_calculate_checksum(rec, len, 0)) == 0 || len == 0);
        if (len >= 20) (void)oracle_ipv4_header_validate(rec, ifc.ext_tcp);
        /* transmit side: fill in place (L3), and a whole packet built into a buffer of exactly its size */
        if (ifc.mode == 0) (void)oracle_generate(rec, len);
        if (len <= 65507 + 100) {
            oracle_tx_desc d;
            memset(&d, 0, sizeof d);
            d.src = rnd(5) ? ifc.unicast : 0, d.dst = (uint32_t)rnd64(), d.l4a = (uint16_t)rnd64(), d.l4b = (uint16_t)rnd64();
            d.values = (uint32_t)rnd64(), d.ip_id = (uint16_t)rnd64(), d.proto = rnd(20) ? (rnd(2) ? 17 : 1) : 99;
            const size_t l3 = 28 + len, frame = ifc.mode ? 14 + (l3 < 60 ? 60 : l3) : l3;
            uint8_t* out = malloc(frame);
            REQUIRE(out);
            uint8_t st = 0xEE;
            const uint32_t n = oracle_tx_build(&d, rec, len, &ifc, out, &st);
            REQUIRE(n == 0 || n == frame);
            if (n) {
                /* What was built verifies at its destination -- when the datagram's source was the interface's own
                 * address: udp::send sums its pseudo-header over the caller's source (src/transport/udp.rs:186-192)
                 * while ipv4::send writes the interface's (src/protocols/ipv4.rs:338), so a datagram sent from 0.0.0.0
                 * carries a checksum its receiver rejects.  The oracle keeps that, as the reference does. */
                oracle_iface rx = ifc;
                rx.unicast = d.dst;
                memcpy(rx.mac, d.dst_mac, 6);
                const oracle_result v = oracle_recv(out, n, &rx);
                /* (tap::send pads the packet to 60 bytes, src/driver/tap.rs:147-152, and the receive path takes everything
                 * behind the IPv4 header as the datagram, src/protocols/ipv4.rs:415: a padded short datagram fails
                 * udp::recv's length check, src/transport/udp.rs:232-238 -- in the reference too) */
                const int padded = ifc.mode == 1 && l3 < 60;
                REQUIRE(v.verdict != OV_DROP_L3_CSUM && (padded || v.verdict != OV_DROP_UDP_LEN));
                if (d.src == ifc.unicast || d.proto == 1) REQUIRE(v.verdict != OV_DROP_UDP_CSUM);
                if (d.src == ifc.unicast && !padded) REQUIRE(v.verdict == (d.proto == 17 ? OV_ACCEPT_UDP : (d.l4a >> 8) == 8 ? OV_ACCEPT_ICMP_ECHO : v.verdict));
            }
            free(out);
        }
        free(rec);
    }
    printf("oracle_san_check ok: %d records\n", rounds);
    return 0;
}
