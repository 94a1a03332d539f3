/*
 * unet_oracle.h -- TEST INFRASTRUCTURE, not product code.
 *
 * A scalar CPU restatement, in plain C, of the receive/transmit integrity path of
 * ikanago/unet @ 2024_10_08 (the Rust sources cannot be compiled in this image: no
 * rustc/cargo).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it, and only as the checker / the CPU baseline.  Nothing under unet_amd/
 * links or calls it.
 *
 * Parity pin: the restatement is checked against every known-answer vector the
 * reference's own unit tests hold for this path (src/utils.rs:43-60,
 * src/protocols/ipv4.rs:428-438, src/transport/icmp.rs:128-139) -- see tests/golden/.
 * The TCP branch (ext_tcp) and the flow hash have no counterpart in the reference:
 * parity unpinned for those two.
 */
#ifndef UNET_ORACLE_H
#define UNET_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same numeric codes as enum unet_verdict in include/unet_offload.h (a test asserts it). */
enum oracle_verdict {
    OV_NONE = 0,
    OV_DROP_L2_TRUNCATED = 1,
    OV_DROP_L2_ETHERTYPE = 2,
    OV_DROP_L2_NOT_FOR_ME = 3,
    OV_PASS_ARP = 4,
    OV_DROP_L3_TRUNCATED = 5,
    OV_DROP_L3_PROTO = 6,
    OV_DROP_L3_VERSION = 7,
    OV_DROP_L3_IHL_SHORT = 8,
    OV_DROP_L3_IHL_LONG = 9,
    OV_DROP_L3_FRAGMENT = 10,
    OV_DROP_L3_CSUM = 11,
    OV_IGNORE_L3_NOT_FOR_ME = 12,
    OV_DROP_ICMP_TRUNCATED = 13,
    OV_DROP_ICMP_TYPE = 14,
    OV_ACCEPT_ICMP_ECHO = 15,
    OV_ACCEPT_ICMP_ECHOREPLY = 16,
    OV_DROP_UDP_SHORT = 17,
    OV_DROP_UDP_LEN = 18,
    OV_DROP_UDP_CSUM = 19,
    OV_ACCEPT_UDP = 20,
    OV_DROP_TCP_SHORT = 21,
    OV_DROP_TCP_DOFF = 22,
    OV_DROP_TCP_CSUM = 23,
    OV_ACCEPT_TCP = 24,
    OV_L3_HEADER_OK = 25,
    OV_DROP_ARP_TRUNCATED = 26,
    OV_DROP_ARP_HTYPE = 27,
    OV_DROP_ARP_PTYPE = 28,
    OV_DROP_ARP_HLEN = 29,
    OV_DROP_ARP_PLEN = 30,
    OV_IGNORE_ARP_NOT_FOR_ME = 31,
    OV_DROP_TCP_OPTION = 32, /* extension: checksum good, option area malformed */
    OV_COUNT = 40
};

typedef struct oracle_result {
    uint16_t csum;
    uint8_t verdict;
    uint8_t proto;
} oracle_result;

typedef struct oracle_iface {
    uint32_t unicast;
    uint32_t broadcast;
    uint8_t mac[6];
    uint8_t mode;    /* 0 = L3 packets, 1 = Ethernet frames */
    uint8_t ext_tcp; /* 0 = strict reference behaviour */
} oracle_iface;

/* src/utils.rs:1-17 */
uint16_t oracle_calculate_checksum(const uint8_t* data, size_t len, uint16_t sum);
/* same function with a 64-bit accumulator over 8-byte words; must equal the one above */
uint16_t oracle_calculate_checksum_fast(const uint8_t* data, size_t len, uint16_t sum);

/* src/protocols/ipv4.rs:220 */
uint32_t oracle_iface_broadcast(uint32_t unicast, uint32_t netmask);

/* Ipv4Header::try_from + validate on >= 20 bytes: src/protocols/ipv4.rs:185-207,128-157 */
oracle_result oracle_ipv4_header_validate(const uint8_t* hdr20, int ext_tcp);

/* ipv4::recv and below on one L3 packet: src/protocols/ipv4.rs:394-422 */
oracle_result oracle_ipv4_recv(const uint8_t* data, size_t len, const oracle_iface* iface);
/* ethernet::recv then ipv4::recv: src/devices/ethernet.rs:58-75 */
oracle_result oracle_ethernet_recv(const uint8_t* frame, size_t len, const oracle_iface* iface);
/* dispatch on iface->mode */
oracle_result oracle_recv(const uint8_t* rec, size_t len, const oracle_iface* iface);

/* Transmit-side checksums. */
/* udp::send checksum, src/transport/udp.rs:181-203: returns the value written to segment[6..8] */
uint16_t oracle_udp_send_checksum(uint32_t src, uint32_t dst, uint16_t sport, uint16_t dport,
                                  const uint8_t* payload, size_t payload_len);
/* icmp::send checksum, src/transport/icmp.rs:75-85 */
uint16_t oracle_icmp_send_checksum(uint8_t type, uint8_t code, uint32_t values, const uint8_t* data,
                                   size_t data_len);
/* create_ip_header, src/protocols/ipv4.rs:366-391: writes 20 bytes */
void oracle_create_ip_header(uint16_t id, uint8_t protocol, uint32_t src, uint32_t dst, size_t data_len,
                             uint8_t out20[20]);
/* In-place fill of an L3 packet's checksum fields the way the three send functions do;
 * returns 0 if filled, else the drop code explaining why the record was left alone. */
uint8_t oracle_generate(uint8_t* pkt, size_t len);

/* Same layout as unet_tx_desc in include/unet_offload.h (a test asserts the size). */
typedef struct oracle_tx_desc {
    uint32_t src, dst;
    uint16_t l4a, l4b;
    uint32_t values;
    uint16_t ip_id;
    uint8_t proto, reserved0;
    uint8_t dst_mac[6];
    uint8_t reserved1[2];
} oracle_tx_desc;
/* udp::send / icmp::send -> ipv4::send -> (L2) ethernet header + tap::send padding, one packet:
 * src/transport/udp.rs:168-216, icmp.rs:66-95, src/protocols/ipv4.rs:300-391, src/driver/tap.rs:138-152 */
uint32_t oracle_tx_build(const oracle_tx_desc* d, const uint8_t* payload, size_t plen, const oracle_iface* iface,
                         uint8_t* out, uint8_t* status);
void oracle_tx_build_batch(const uint8_t* pay_arena, const uint32_t* pay_off16, uint32_t pay_stride16,
                           const uint32_t* pay_len, const oracle_tx_desc* desc, uint32_t n, const oracle_iface* iface,
                           uint8_t* out_arena, const uint32_t* out_off16, uint32_t* out_len, uint8_t* status);

/* Batch drivers (single thread), used for parity sweeps and as the timed CPU baseline.
 * Record i starts at 16*off16[i] (or i*16*stride16 when off16 is NULL). */
void oracle_recv_batch(const uint8_t* arena, const uint32_t* off16, uint32_t stride16,
                       const uint32_t* len, uint32_t n, const oracle_iface* iface, oracle_result* out);
/* same verdicts, computed with oracle_calculate_checksum_fast: the "best effort CPU" line */
void oracle_recv_batch_fast(const uint8_t* arena, const uint32_t* off16, uint32_t stride16,
                            const uint32_t* len, uint32_t n, const oracle_iface* iface, oracle_result* out);
void oracle_csum_batch(const uint8_t* arena, const uint32_t* off16, uint32_t stride16,
                       const uint32_t* len, const uint16_t* seeds, uint32_t n, uint16_t* out);
void oracle_ipv4_validate_batch(const uint8_t* hdrs, uint32_t n, int ext_tcp, oracle_result* out);
void oracle_generate_batch(uint8_t* arena, const uint32_t* off16, uint32_t stride16, const uint32_t* len,
                           uint32_t n, uint8_t* status);

/* Build-defined flow hash (no reference counterpart; parity unpinned). */
/* Per-flow TCP tracking (extension of this build, nothing like it in the reference): the sequential definition.
 * Records are visited in index order; those with results[i].verdict == OV_ACCEPT_TCP update the direct-mapped
 * table (table_slots a power of two, 32-byte slots laid out like unet_tcp_flow) and get a per-segment state. */
typedef struct oracle_tcp_flow {
    uint32_t src, dst, ports, next_seq, state, segments, out_of_order, after_close;
} oracle_tcp_flow;
void oracle_tcp_track(const uint8_t* arena, const uint32_t* off16, uint32_t stride16, const uint32_t* len,
                      const oracle_result* results, uint32_t n, const oracle_iface* iface, oracle_tcp_flow* table,
                      uint32_t table_slots, uint8_t* seg_state);
uint32_t oracle_tcp_slot(uint32_t src, uint32_t dst, uint16_t sport, uint16_t dport, uint32_t table_slots);

uint32_t oracle_flow_hash5(uint32_t src, uint32_t dst, uint8_t proto, uint16_t sport, uint16_t dport);
uint32_t oracle_flow_hash_record(const uint8_t* rec, size_t len, const oracle_iface* iface);

#ifdef __cplusplus
}
#endif
#endif
