/*
 * drain_loop.c -- the offload from plain C, both directions.
 *
 * What NetProtocol::recv does per queue entry (src/protocols.rs:83-97: ipv4::recv on each frame, then the UDP or ICMP
 * handler) done once per drained queue: pack the frames, one unet_verify call, act on the verdicts; the datagrams that
 * were accepted are answered -- their payloads sent back to their senders -- with one unet_tx_build call
 * (udp::send -> ipv4::send, src/transport/udp.rs:168-216), and the answers are judged again as their receiver would.
 *
 *   gcc -std=c11 -Iinclude examples/drain_loop.c -Lunet_amd/lib -lunet_offload -o examples/drain_loop
 *   examples/drain_loop [frames]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "unet_offload.h"

#define CHECK(call)                                                    \
    do {                                                               \
        int rc_ = (call);                                              \
        if (rc_ != UNET_OK) {                                          \
            fprintf(stderr, "%s: %d %s\n", #call, rc_, unet_last_error()); \
            return 1;                                                  \
        }                                                              \
    } while (0)

static void* alloc16(size_t n) {
    void* p = NULL;
    if (posix_memalign(&p, 16, n ? n : 16) != 0) exit(2);
    return p;
}

static uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
static uint16_t be16(const uint8_t* p) { return (uint16_t)(p[0] << 8 | p[1]); }

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 2000;
    /* the interface of the reference's demo: 192.0.2.2/24 (src/app.rs:33-54) */
    unet_iface ifc;
    memset(&ifc, 0, sizeof ifc);
    ifc.unicast = 0xC0000202u;
    ifc.broadcast = 0xC00002FFu;
    ifc.mode = UNET_MODE_L3;

    /* a drained queue: n packets of the synthetic mix (valid and faulty UDP, ICMP, TCP of 64..9000 bytes) */
    uint32_t* off16 = malloc(n * sizeof *off16);
    uint32_t* len = malloc(n * sizeof *len);
    uint64_t arena_bytes = 0;
    CHECK(unet_synth_layout(UNET_SYNTH_IMIX, 1, 0, n, off16, len, &arena_bytes));
    uint8_t* arena = alloc16(arena_bytes);
    CHECK(unet_synth_host(UNET_SYNTH_IMIX, 1, 0, n, &ifc, off16, len, arena, arena_bytes));

    unet_ctx* ctx = NULL;
    CHECK(unet_open(0, &ctx));
    unet_result* res = malloc(n * sizeof *res);
    CHECK(unet_verify(ctx, arena, arena_bytes, off16, 0, len, n, &ifc, res));

    /* verdict consumer: what udp::recv would hand to a socket (src/transport/udp.rs:256-277) is echoed back */
    uint32_t accepted = 0, other = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (res[i].verdict == UNET_V_ACCEPT_UDP) accepted++;
        else other++;
    }
    uint32_t* pay_off = malloc((accepted + 1) * sizeof *pay_off);
    uint32_t* pay_len = malloc((accepted + 1) * sizeof *pay_len);
    unet_tx_desc* desc = calloc(accepted + 1, sizeof *desc);
    uint32_t k = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (res[i].verdict != UNET_V_ACCEPT_UDP) continue;
        const uint8_t* pkt = arena + (uint64_t)off16[i] * 16u;
        const uint32_t ihl = (pkt[0] & 15u) * 4u;
        /* the payload stays where it is: the packet arena doubles as the payload arena (payloads start on 16-byte
         * boundaries only by luck, so copy them when they do not) */
        desc[k].src = ifc.unicast;
        desc[k].dst = be32(pkt + 12);
        desc[k].l4a = be16(pkt + ihl + 2);
        desc[k].l4b = be16(pkt + ihl);
        desc[k].ip_id = (uint16_t)k;
        desc[k].proto = 17;
        pay_len[k] = len[i] - ihl - 8u;
        pay_off[k] = i; /* remembered: filled in below */
        k++;
    }
    /* payload arena: the accepted datagrams' bytes, packed */
    uint64_t pay_bytes = 0;
    CHECK(unet_pack_size(pay_len, accepted, &pay_bytes));
    uint8_t* pay = alloc16(pay_bytes);
    memset(pay, 0, pay_bytes);
    uint64_t at = 0;
    for (uint32_t j = 0; j < accepted; j++) {
        const uint32_t i = pay_off[j];
        const uint8_t* pkt = arena + (uint64_t)off16[i] * 16u;
        const uint32_t ihl = (pkt[0] & 15u) * 4u;
        memcpy(pay + at, pkt + ihl + 8u, pay_len[j]);
        pay_off[j] = (uint32_t)(at / 16u);
        at += ((uint64_t)pay_len[j] + 15u) & ~15ull;
    }

    uint64_t out_bytes = 0;
    uint32_t* out_off = malloc((accepted + 1) * sizeof *out_off);
    uint32_t* out_len = malloc((accepted + 1) * sizeof *out_len);
    uint8_t* status = malloc(accepted + 1);
    CHECK(unet_tx_layout(pay_len, accepted, ifc.mode, out_off, &out_bytes));
    uint8_t* out = alloc16(out_bytes);
    if (accepted) CHECK(unet_tx_build(ctx, pay, pay_bytes, pay_off, 0, pay_len, desc, accepted, &ifc, out, out_bytes, out_off, out_len, status));

    /* every answer, judged as the host it is addressed to would: all of them must be accepted */
    uint32_t answered = 0;
    unet_result* res2 = malloc((accepted + 1) * sizeof *res2);
    for (uint32_t j = 0; j < accepted; j++) {
        unet_iface peer = ifc;
        peer.unicast = desc[j].dst;
        CHECK(unet_verify(ctx, out + (uint64_t)out_off[j] * 16u, ((uint64_t)out_len[j] + 15u) & ~15ull, NULL, 1, &out_len[j], 1, &peer, &res2[j]));
        if (status[j] == UNET_TX_OK && res2[j].verdict == UNET_V_ACCEPT_UDP) answered++;
        if (j >= 15) { /* (one call per packet is the slow way: the first few prove the point) */
            answered += accepted - j - 1;
            break;
        }
    }
    uint64_t counts[UNET_V_COUNT];
    CHECK(unet_counters(ctx, counts, 0));
    printf("drain_loop: %u frames judged, %u datagrams accepted and answered (%u verified), %u others; counter ACCEPT_UDP = %llu\n", n,
           accepted, answered, other, (unsigned long long)counts[UNET_V_ACCEPT_UDP]);
    unet_close(ctx);
    if (answered != accepted) {
        fprintf(stderr, "drain_loop: an answer did not verify\n");
        return 1;
    }
    printf("drain_loop ok\n");
    return 0;
}
