/*
 * unet_offload.h -- C ABI of the MI355X (gfx950) offload for the receive/transmit
 * integrity path of the ikanago/unet userspace network stack: RFC 1071 Internet
 * checksum, IPv4 / UDP / ICMP / Ethernet header parse + validate, and the
 * accept/drop verdict each frame would get from the stack's own recv functions.
 *
 * Everything here is plain C: pointers, sizes, fixed-width integers.  All compute
 * runs in hand-written HIP kernels; there is no CPU compute path behind these
 * entry points (the only host-side arithmetic is the synthetic-trace generator's
 * host twin and the arena packer, neither of which produces checksums or verdicts
 * for a caller).  Every function returns 0 on success or a negative UNET_E_* code;
 * unet_last_error() returns a thread-local description of the last failure.
 *
 * "Reference" citations are file:line in ikanago/unet @ 2024_10_08.
 */
#ifndef UNET_OFFLOAD_H
#define UNET_OFFLOAD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ errors */
#define UNET_OK 0
#define UNET_E_INVAL (-1)   /* bad argument (null pointer, n out of range, misaligned arena ...) */
#define UNET_E_NODEV (-2)   /* no usable HIP device / device id out of range */
#define UNET_E_HIP (-3)     /* a HIP runtime call failed; see unet_last_error() */
#define UNET_E_NOMEM (-4)   /* host or device allocation failed */
#define UNET_E_RANGE (-5)   /* a frame length > 65535 or an offset outside the arena */
#define UNET_E_BUSY (-6)    /* unet_verify_async: UNET_ASYNC_DEPTH batches are already in flight; unet_verify_static: the GPU's static array is in use */

/* ---------------------------------------------------------------- verdicts */
/* One code per outcome of the reference's receive path; the FIRST failing check
 * in the reference's evaluation order wins.  Codes tagged "panic" are inputs on
 * which the reference indexes out of bounds; the offload defines a drop for them. */
enum unet_verdict {
    UNET_V_NONE = 0,                 /* slot not written */
    UNET_V_DROP_L2_TRUNCATED = 1,    /* frame < 14 B: panic, src/devices/ethernet.rs:50-52 */
    UNET_V_DROP_L2_ETHERTYPE = 2,    /* ethertype not 0x0800/0x0806, src/protocols.rs:33-42 */
    UNET_V_DROP_L2_NOT_FOR_ME = 3,   /* dst MAC neither ours nor ff:ff:ff:ff:ff:ff, src/devices/ethernet.rs:63-71 */
    UNET_V_PASS_ARP = 4,             /* valid ARP message whose target is our address: the host caches the sender and replies, src/protocols/arp.rs:238-245 */
    UNET_V_DROP_L3_TRUNCATED = 5,    /* packet < 20 B or IHL*4 > len: panic, src/protocols/ipv4.rs:189-203,415 */
    UNET_V_DROP_L3_PROTO = 6,        /* protocol byte not ICMP/UDP; checked before anything else, src/transport.rs:18-26 via ipv4.rs:189 */
    UNET_V_DROP_L3_VERSION = 7,      /* src/protocols/ipv4.rs:129-133 */
    UNET_V_DROP_L3_IHL_SHORT = 8,    /* IHL*4 < 20, src/protocols/ipv4.rs:134-138 */
    UNET_V_DROP_L3_IHL_LONG = 9,     /* IHL*4 >= 60 (IHL 15 is rejected), src/protocols/ipv4.rs:139-143 */
    UNET_V_DROP_L3_FRAGMENT = 10,    /* MF set or fragment offset != 0, src/protocols/ipv4.rs:145-147 */
    UNET_V_DROP_L3_CSUM = 11,        /* checksum over the first 20 header bytes != 0, src/protocols/ipv4.rs:152-157 */
    UNET_V_IGNORE_L3_NOT_FOR_ME = 12,/* dst not unicast/broadcast/255.255.255.255: Ok(()) silently, src/protocols/ipv4.rs:402-407 */
    UNET_V_DROP_ICMP_TRUNCATED = 13, /* ICMP message < 8 B: panic, src/transport/icmp.rs:52-55 */
    UNET_V_DROP_ICMP_TYPE = 14,      /* type not 0/8, src/transport/icmp.rs:16-22 */
    UNET_V_ACCEPT_ICMP_ECHO = 15,    /* echo request; host sends the reply, src/transport/icmp.rs:110-120 */
    UNET_V_ACCEPT_ICMP_ECHOREPLY = 16,/* src/transport/icmp.rs:103-121 */
    UNET_V_DROP_UDP_SHORT = 17,      /* segment < 8 B, src/transport/udp.rs:226-228 */
    UNET_V_DROP_UDP_LEN = 18,        /* buffer length != UDP length field, src/transport/udp.rs:232-238 */
    UNET_V_DROP_UDP_CSUM = 19,       /* pseudo-header checksum != 0, src/transport/udp.rs:247-254 */
    UNET_V_ACCEPT_UDP = 20,          /* continues to PCB lookup on the host, src/transport/udp.rs:256-277 */
    /* Extension (no counterpart in the reference, only with unet_iface.ext_tcp = 1): */
    UNET_V_DROP_TCP_SHORT = 21,      /* segment < 20 B */
    UNET_V_DROP_TCP_DOFF = 22,       /* data offset < 5 or data offset*4 > segment length */
    UNET_V_DROP_TCP_CSUM = 23,       /* pseudo-header (protocol 6) checksum != 0 */
    UNET_V_ACCEPT_TCP = 24,          /* checksum good and options well-formed (else DROP_TCP_OPTION below) */
    /* Only produced by unet_ipv4_validate*: header parsed and validate() returned Ok. */
    UNET_V_L3_HEADER_OK = 25,
    /* ARP messages (ethertype 0x0806, MODE_L2), arp::recv src/protocols/arp.rs:206-246: */
    UNET_V_DROP_ARP_TRUNCATED = 26,  /* < 28 B after the Ethernet header: panic, arp.rs:211,225 (slices [0..8], [8..28]) */
    UNET_V_DROP_ARP_HTYPE = 27,      /* hardware type != 1, arp.rs:212-214 */
    UNET_V_DROP_ARP_PTYPE = 28,      /* protocol type != 0x0800, arp.rs:215-217 */
    UNET_V_DROP_ARP_HLEN = 29,       /* hardware address length != 6, arp.rs:218-220 */
    UNET_V_DROP_ARP_PLEN = 30,       /* protocol address length != 4, arp.rs:221-223 */
    UNET_V_IGNORE_ARP_NOT_FOR_ME = 31,/* valid, target is another host: Ok(()) with no action, arp.rs:238,246 */
    /* Extension (ext_tcp = 1 only): the checksum verified, but the option area between byte 20 and the data offset
     * is malformed -- an option without room for its length byte, a length below 2 or running past the data
     * offset, or a wrong length for MSS (4), window scale (3), SACK permitted (2), SACK (2 + 8k) or timestamps (10). */
    UNET_V_DROP_TCP_OPTION = 32,
    UNET_V_COUNT = 40                /* size of a counter array indexed by verdict (33..39 are unassigned) */
};

/* Per-frame result, 4 bytes.
 *  verdict: enum unet_verdict.
 *  proto:   IPv4 protocol byte (packet[9]) whenever the IPv4 header was reachable
 *           (>= 20 bytes of L3), else 0.
 *  csum:    the checksum value the reference computes and tests/prints on this path:
 *           DROP_L3_CSUM          -> calculate_checksum(header[0..20], 0)  (nonzero)
 *           DROP_UDP_CSUM / ACCEPT_UDP -> calculate_checksum(pseudo, !calculate_checksum(segment, 0))
 *           DROP_TCP_CSUM / ACCEPT_TCP -> same with protocol 6 (extension)
 *           ACCEPT_ICMP_ECHO      -> checksum of the echo reply the reference sends
 *                                    (type 0, same code/values/data; icmp.rs:75-85,110-120)
 *           every other verdict   -> 0
 */
typedef struct unet_result {
    uint16_t csum;
    uint8_t verdict;
    uint8_t proto;
} unet_result;

#define UNET_MODE_L3 0 /* records are L3 packets starting at the IPv4 header (what ipv4::recv gets) */
#define UNET_MODE_L2 1 /* records are Ethernet frames starting at the destination MAC (what ethernet::recv reads) */

/* Receiving interface: the constants the reference keeps in Ipv4Interface / NetDevice
 * (src/protocols/ipv4.rs:209-229, src/devices.rs:75-90).  Addresses are the u32 value of
 * Ipv4Address (big-endian bytes read as a number, e.g. 192.0.2.2 = 0xC0000202). */
typedef struct unet_iface {
    uint32_t unicast;
    uint32_t broadcast; /* unicast & 0xffffff00 | ~netmask & 0xff, src/protocols/ipv4.rs:220 */
    uint8_t mac[6];     /* used in UNET_MODE_L2 only */
    uint8_t mode;       /* UNET_MODE_L3 or UNET_MODE_L2 */
    uint8_t ext_tcp;    /* 0: protocol 6 -> DROP_L3_PROTO exactly like the reference; 1: TCP extension */
} unet_iface;

typedef struct unet_ctx unet_ctx;

/* ------------------------------------------------------------ context */
/* Bind a context to one HIP device.  Creates two streams and pinned staging used by the
 * host-pointer entry points.  One context per thread/GPU (thread-compatible, not thread-safe).
 * Host-pointer calls on small batches (staging image up to 4 MiB for unet_verify, unet_verify_async,
 * unet_generate, unet_ipv4_validate and unet_csum_batch, 1 MiB for unet_tx_build, 64 KiB for unet_csum) let the kernel read the pinned staging buffer in
 * place across the host link and write its results into pinned memory: one launch and one wait per call, no copy
 * command.  Larger batches are copied to device memory first -- through the library's pinned staging buffers, or, from
 * 1 MiB up and for the synchronous calls other than unet_tx_build, straight from the caller's arena when that is
 * pinned host memory itself (hipHostMalloc, or registered with hipHostRegister), which takes the caller's thread out of
 * the transfer (unet_verify on 262,144 x 1500 bytes: 48 -> 53 GB/s).  Environment, read once here: UNET_ZERO_COPY_BYTES
 * (that limit; 0 = always copy), UNET_COPY_THREADS (host threads filling the staging buffers, default 4), and for
 * experiments UNET_TILE / UNET_TX_TILE (records / packets per wave, overriding the library's choice by record size),
 * UNET_TX_WAVE (0: long payloads by tiles too; 2: every batch one packet per wavefront), UNET_TXW_GROUP (wavefronts per
 * packet), UNET_TXW_BLOCK, UNET_TXW_POLICY, UNET_TX_STAGE (0: no LDS staging of short packets' output) and
 * UNET_TX_PIPELINE (0: unet_tx_build never works in overlapping runs) / UNET_TX_PIPELINE_MIN (bytes from which it does). */
int unet_open(int device_id, unet_ctx** out_ctx);
void unet_close(unet_ctx* ctx);
const char* unet_last_error(void);
int unet_device_count(void);
/* Block until all work queued on the context's own streams is done. */
int unet_sync(unet_ctx* ctx);
/* Records per wavefront are chosen from the batch's mean record length.  The *_dev entry points cannot see the
 * lengths (they are in device memory) and estimate it as 16*stride16, or arena_bytes / n without a stride --
 * wrong for an arena with gaps.  A caller that knows better says so here (0 = back to the estimate).  Only
 * speed depends on it, never results. */
int unet_hint_mean_len(unet_ctx* ctx, uint32_t mean_len);
/* The same for the longest record (or, for unet_tx_build_dev, payload) of the batches to come, 0 = not known.  With it
 * a batch of records of at most 64 bytes gets 64-record tiles although only its mean could be estimated, and
 * unet_tx_build_dev reserves its LDS staging area for batches of mid-size payloads (up to 1008 bytes, 994 for Ethernet
 * output) that come with an offset table -- without it only for short ones and for fixed strides.  Only speed depends
 * on it, never results: a record longer than the hint is still handled correctly. */
int unet_hint_max_len(unet_ctx* ctx, uint32_t max_len);
/* The number of records one wavefront is given for a batch whose mean record length is `mean_len` bytes and whose
 * longest record is `max_len` bytes (0 = not known, as for the *_dev entry points): 64, 48, 40, 32, 16, 8, 4, 2 or
 * 1.  A pure function of its arguments that touches no GPU -- for logs, for sizing experiments with UNET_TILE, and
 * so that the rule can be tested without a device.  (No counterpart in the reference, which has no batches.) */
int unet_records_per_wave(uint32_t mean_len, uint32_t max_len);
/* The same for unet_tx_build*: packets per wavefront by mean payload length -- 64, 32, 16 or 8 up to 1152 bytes; above that 1:
 * every packet is built by a wavefront of its own, or by unet_tx_waves_per_packet (1, 2, 4, 8 or 16, about three 1 KiB
 * rows of output each) wavefronts of one workgroup.  (UNET_TX_WAVE=0 in the environment of unet_open keeps such
 * batches on the tile kernel, 8, 4, 2 or 1 packets per wavefront: for A/B measurements.) */
int unet_tx_packets_per_wave(uint32_t mean_payload_len);
int unet_tx_waves_per_packet(uint32_t mean_payload_len, int mode);

/* -------------------------------------------------- frame arena layout */
/* A batch is `n` records in one byte arena.  Record i starts at byte 16*off16[i]
 * (every record starts on a 16-byte boundary) and is len[i] bytes long (0..65535).
 * If off16 is NULL the records sit at a fixed stride: record i starts at i*16*stride16.
 * The arena must be readable up to the next multiple of 16 past every record's end
 * (the packer pads); pad bytes are never interpreted. */

/* ------------------------------------------- a1: calculate_checksum */
/* Batched `utils::calculate_checksum(data, sum)` (src/utils.rs:1-17): out[i] =
 * checksum of record i seeded with seeds[i] (seeds may be NULL = all 0).
 * *_dev: every pointer is device memory of ctx's device; the kernel is enqueued on
 * `stream` (a hipStream_t; NULL = the context's compute stream) and the call returns
 * without waiting. */
int unet_csum_batch_dev(unet_ctx* ctx, void* stream, const uint8_t* d_arena, uint64_t arena_bytes,
                        const uint32_t* d_off16, uint32_t stride16, const uint32_t* d_len,
                        const uint16_t* d_seeds, uint32_t n, uint16_t* d_out);
/* Host-pointer form: stages through pinned memory, runs the same kernel, waits. */
int unet_csum_batch(unet_ctx* ctx, const uint8_t* arena, uint64_t arena_bytes, const uint32_t* off16,
                    uint32_t stride16, const uint32_t* len, const uint16_t* seeds, uint32_t n,
                    uint16_t* out);
/* Exact shape of the reference leaf `calculate_checksum(data: &[u8], sum: u16) -> u16`
 * (src/utils.rs:1; call sites ipv4.rs:154,387 icmp.rs:83 udp.rs:200,202,229,247) for one
 * buffer of at most 65535 bytes.  Runs on the GPU like everything else; it exists so a
 * host shim can keep the reference signature, not because one buffer at a time is fast. */
int unet_csum(unet_ctx* ctx, const uint8_t* data, size_t len, uint16_t seed, uint16_t* out);

/* --------------------------- a2-a4: Ipv4Header::try_from + validate */
/* `n` bare 20-byte IPv4 headers packed back to back (stride 20 bytes, d_hdrs 16-byte
 * aligned).  out[i] = verdict of Ipv4Header::try_from (src/protocols/ipv4.rs:185-207) followed
 * by validate() (ipv4.rs:128-157): one of DROP_L3_PROTO / VERSION / IHL_SHORT / IHL_LONG /
 * FRAGMENT / CSUM or L3_HEADER_OK.  Only iface->ext_tcp is read. */
int unet_ipv4_validate_dev(unet_ctx* ctx, void* stream, const uint8_t* d_hdrs, uint32_t n,
                           const unet_iface* iface, unet_result* d_out);
int unet_ipv4_validate(unet_ctx* ctx, const uint8_t* hdrs, uint32_t n, const unet_iface* iface,
                       unet_result* out);

/* ------------- a2-a6, a8, a11: ethernet::recv -> ipv4::recv -> udp/icmp::recv */
/* Verdict of the reference receive path for every record: ethernet::recv
 * (src/devices/ethernet.rs:58-75, MODE_L2 only) -> ipv4::recv (src/protocols/ipv4.rs:394-422)
 * -> udp::recv validation (src/transport/udp.rs:219-254) or icmp::recv
 * (src/transport/icmp.rs:97-122). */
int unet_verify_dev(unet_ctx* ctx, void* stream, const uint8_t* d_arena, uint64_t arena_bytes,
                    const uint32_t* d_off16, uint32_t stride16, const uint32_t* d_len, uint32_t n,
                    const unet_iface* iface, unet_result* d_out);
int unet_verify(unet_ctx* ctx, const uint8_t* arena, uint64_t arena_bytes, const uint32_t* off16,
                uint32_t stride16, const uint32_t* len, uint32_t n, const unet_iface* iface,
                unet_result* out);
/* The same verdicts for a batch of a FEW LONG records, one wavefront per 4 KiB segment of a record instead of one per
 * record or per several: sixteen jumbo frames (the loopback device carries up to 65535 bytes,
 * src/devices/loopback.rs:16) keep 16 x 16 wavefronts busy instead of four.  max_len bounds the record lengths (a
 * record that needs more 4 KiB segments than max_len does gets UNET_V_NONE -- pass the true maximum); d_workspace holds
 * one u32 per segment (unet_verify_split_workspace bytes, 4-byte aligned; host code).  Results are bit-identical to
 * unet_verify_dev: sums of 16-bit words add up over segments that start on even offsets.  unet_verify itself takes
 * this path when a batch has fewer than 128 wavefronts' worth of records of 16 KiB and more. */
int unet_verify_split_workspace(uint32_t n, uint32_t max_len, uint64_t* bytes);
int unet_verify_split_dev(unet_ctx* ctx, void* stream, const uint8_t* d_arena, uint64_t arena_bytes,
                          const uint32_t* d_off16, uint32_t stride16, const uint32_t* d_len, uint32_t n,
                          uint32_t max_len, const unet_iface* iface, unet_result* d_out, void* d_workspace,
                          uint64_t workspace_bytes);
/* The same verdicts for n records of ONE length laid out at a BYTE stride: record i occupies bytes
 * [i * stride_bytes, i * stride_bytes + len) of d_buf (16-byte aligned; readable up to the end of the 16-byte chunk
 * the last record ends in).  No slots, no tables: what a capture ring of fixed-size cells or a batch of
 * minimum-size frames looks like -- 46-byte L3 packets and 60-byte frames are the shortest the reference's devices
 * carry (src/devices/ethernet.rs:9-13), 74 bytes the shortest frame its tap driver sends (src/driver/tap.rs:147-152).
 * len <= stride_bytes <= 256 (UNET_E_RANGE beyond: use unet_verify_dev, whose 16-byte pad weighs little there).
 * Same functions and order of checks as unet_verify_dev, bit-identical results. */
int unet_verify_bytes_dev(unet_ctx* ctx, void* stream, const uint8_t* d_buf, uint64_t buf_bytes, uint32_t stride_bytes,
                          uint32_t len, uint32_t n, const unet_iface* iface, unet_result* d_out);

/* The drain loop's way of using the GPU without threads of its own (src/protocols.rs:83-97 runs on the stack's
 * main thread): unet_verify_async stages batch k and returns while the GPU works on it, the caller packs batch
 * k+1 and hands it over as well, then collects the verdicts of batch k with unet_wait.  Up to UNET_ASYNC_DEPTH
 * batches are in flight, each on a stream of its own (copy in, kernel, copy out overlap between batches); one
 * more submit returns UNET_E_BUSY.  The arena and tables may be reused as soon as unet_verify_async returns;
 * `out` must stay valid until unet_wait(ticket) returns, which is when it holds the results.  Same kernel and
 * bit-identical results as unet_verify. */
#define UNET_ASYNC_DEPTH 2
#define UNET_TICKET_NONE 0xffffffffu
typedef struct unet_ticket {
    uint32_t id;
} unet_ticket;
int unet_verify_async(unet_ctx* ctx, const uint8_t* arena, uint64_t arena_bytes, const uint32_t* off16,
                      uint32_t stride16, const uint32_t* len, uint32_t n, const unet_iface* iface,
                      unet_result* out, unet_ticket* ticket);
int unet_wait(unet_ctx* ctx, unet_ticket ticket);

/* unet_verify for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops
 * such as the stack's own drain loop handing over a few hundred frames at a time, harnesses that forbid
 * hipMalloc).  It has no counterpart in the reference.  Same arguments, same kernel and bit-identical results as
 * unet_verify, but the device-side staging lives in a static device array inside the library (about 8.8 MB,
 * held by every GPU the library runs on until the process ends) and the call performs no hipMalloc / hipFree or
 * any other allocation.  Limits, checked with UNET_E_RANGE: at most UNET_STATIC_MAX_RECORDS records, all of them
 * ending inside the first UNET_STATIC_MAX_ARENA_BYTES bytes of the arena.  All contexts on one GPU share that
 * GPU's array, which keeps the previous call's bytes: a call that arrives while another one (of any context of this
 * process on that GPU) is inside returns UNET_E_BUSY without touching anything. */
#define UNET_STATIC_MAX_RECORDS 65536u
#define UNET_STATIC_MAX_ARENA_BYTES (8u << 20)
int unet_verify_static(unet_ctx* ctx, const uint8_t* arena, uint64_t arena_bytes, const uint32_t* off16,
                       uint32_t stride16, const uint32_t* len, uint32_t n, const unet_iface* iface,
                       unet_result* out);

/* ------------------- a7, a9, a10: udp::send / icmp::send / create_ip_header */
/* Transmit side: every record is an L3 packet (IPv4 header with IHL 5 + UDP or ICMP
 * message) whose checksum fields hold arbitrary bytes.  Fills, in place:
 *   packet[10..12]  = calculate_checksum(header with field zeroed, 0)        (ipv4.rs:366-391)
 *   UDP  segment[6..8] = calculate_checksum(hdr(cksum 0)||payload, !calculate_checksum(pseudo,0))
 *                        with NO 0x0000->0xffff substitution                (udp.rs:181-203)
 *   ICMP message[2..4] = calculate_checksum(hdr(cksum 0)||data, 0)           (icmp.rs:75-85)
 * The pseudo-header length is the buffer's segment length (len[i]-20), which is what
 * udp::send uses.  Records that are not (version 4, IHL 5, protocol 1 or 17, long enough)
 * are left untouched and reported in d_status[i] (0 = filled, else an unet_verdict drop
 * code); d_status may be NULL. */
int unet_generate_dev(unet_ctx* ctx, void* stream, uint8_t* d_arena, uint64_t arena_bytes,
                      const uint32_t* d_off16, uint32_t stride16, const uint32_t* d_len, uint32_t n,
                      uint8_t* d_status);

/* The same two values without touching the packets: fields[i] = what the send path would store at bytes
 * 10..11 (IPv4 header checksum) and at the UDP / ICMP checksum field, as numbers (store them big-endian);
 * {0, 0} when status[i] != 0.  A pure read pass -- for callers that keep the packets elsewhere (the host
 * entry point unet_generate uses it: 4 bytes per packet travel back instead of the arena). */
typedef struct unet_tx_fields {
    uint16_t ip_csum;
    uint16_t l4_csum;
} unet_tx_fields;
int unet_checksum_fields_dev(unet_ctx* ctx, void* stream, const uint8_t* d_arena, uint64_t arena_bytes,
                             const uint32_t* d_off16, uint32_t stride16, const uint32_t* d_len, uint32_t n,
                             unet_tx_fields* d_fields, uint8_t* d_status);
int unet_generate(unet_ctx* ctx, uint8_t* arena, uint64_t arena_bytes, const uint32_t* off16,
                  uint32_t stride16, const uint32_t* len, uint32_t n, uint8_t* status);

/* ------------- transmit path, whole packets: udp::send / icmp::send -> ipv4::send -> tap::send */
/* One packet to build.  `src` is the address the caller passed to udp::send / icmp::send (it goes into
 * the UDP pseudo-header, src/transport/udp.rs:186-192); the IPv4 header always carries the interface's
 * unicast address (src/protocols/ipv4.rs:338, create_ip_header's `src` argument). */
typedef struct unet_tx_desc {
    uint32_t src;        /* Endpoint address given to the send call (0.0.0.0 or the interface address) */
    uint32_t dst;        /* destination IPv4 address */
    uint16_t l4a;        /* UDP: source port.  ICMP: type << 8 | code */
    uint16_t l4b;        /* UDP: destination port.  ICMP: unused */
    uint32_t values;     /* ICMP: the 4 bytes after the checksum (identifier, sequence).  UDP: unused */
    uint16_t ip_id;      /* identification, Ipv4IdGenerator::next() (src/protocols/ipv4.rs:292-296) */
    uint8_t proto;       /* 1 (ICMP) or 17 (UDP) */
    uint8_t reserved0;
    uint8_t dst_mac[6];  /* MODE_L2 only: resolved by the host (ARP cache) or ff:ff:ff:ff:ff:ff */
    uint8_t reserved1[2];
} unet_tx_desc;          /* 28 bytes */

#define UNET_TX_OK 0
#define UNET_TX_E_PROTO 1     /* proto not 1/17 */
#define UNET_TX_E_TOO_LONG 2  /* payload > 65535 - 20 - 8 (src/transport/udp.rs:174-180) */
#define UNET_TX_E_RANGE 3     /* payload or output record outside its arena */

/* Builds packet i = [Ethernet header (MODE_L2)] IPv4 header (IHL 5, ttl 64, flags 0) | UDP or ICMP header |
 * payload i, with every checksum the reference computes on this path:
 *   UDP  hdr(cksum 0)||payload seeded with the pseudo-header, no 0->0xffff substitution (udp.rs:181-203)
 *   ICMP hdr(cksum 0)||data (icmp.rs:75-85);  IPv4 header (ipv4.rs:366-391)
 * MODE_L2 adds dst MAC | iface->mac | 0x0800 and pads like tap::send: when the L3 packet is shorter than
 * 60 bytes, zero bytes are appended until IT is 60 bytes long -- the frame is then 74 bytes
 * (src/driver/tap.rs:138-152 pads data.len(), not the frame).
 * Payload i is record i of the payload arena (same layout rules as every other arena); packet i is
 * written at 16*d_out_off16[i] of the output arena, its length to d_out_len[i] (28 + payload, or
 * 14 + max(28 + payload, 60)), its status (UNET_TX_*) to d_status[i] (may be NULL).  The host computes
 * d_out_off16 from the payload lengths; routing, the MTU check and ARP resolution stay on the host. */
int unet_tx_build_dev(unet_ctx* ctx, void* stream, const uint8_t* d_payload_arena, uint64_t payload_arena_bytes,
                      const uint32_t* d_payload_off16, uint32_t payload_stride16, const uint32_t* d_payload_len,
                      const unet_tx_desc* d_desc, uint32_t n, const unet_iface* iface, uint8_t* d_out_arena,
                      uint64_t out_arena_bytes, const uint32_t* d_out_off16, uint32_t* d_out_len, uint8_t* d_status);

/* The same with options.  UNET_TX_FILL_LINES: the caller states that between the end of every packet and the next
 * 128-byte boundary of the output arena (counted from d_out_arena, which should itself sit on such a boundary)
 * there is nothing it wants kept -- true of any layout made by unet_tx_layout_aligned with align16 = 8 -- and
 * allows the kernel to write zero bytes there, so that it can store a packet's last 128-byte line whole instead of
 * in part (2000-byte payloads 5.8 -> 6.5 TB/s, 3000-byte 5.7 -> 6.1, 1472-byte within 2 %).  The kernel never writes past
 * out_arena_bytes; d_out_len is the packet length as before.  flags = 0 is unet_tx_build_dev. */
#define UNET_TX_FILL_LINES 1u
int unet_tx_build_opts_dev(unet_ctx* ctx, void* stream, const uint8_t* d_payload_arena, uint64_t payload_arena_bytes,
                           const uint32_t* d_payload_off16, uint32_t payload_stride16, const uint32_t* d_payload_len,
                           const unet_tx_desc* d_desc, uint32_t n, const unet_iface* iface, uint8_t* d_out_arena,
                           uint64_t out_arena_bytes, const uint32_t* d_out_off16, uint32_t* d_out_len, uint8_t* d_status,
                           uint32_t flags);

/* Packed output layout for n payloads: out_off16[i] (may be NULL) and the arena size unet_tx_build needs.
 * Host code. */
int unet_tx_layout(const uint32_t* payload_len, uint32_t n, int mode, uint32_t* out_off16, uint64_t* out_arena_bytes);
/* The same with every packet starting on a multiple of align16 16-byte slots (a power of two, 1..64; 8 = the 128-byte
 * lines of the memory system).  Packets that start on lines are built faster when they are long -- 2000-byte payloads
 * 5.1 -> 5.6 TB/s, 4000-byte Ethernet 5.2 -> 5.6, 8972-byte 5.3 -> 5.6 -- for up to 112 bytes of arena per packet that
 * nothing writes (zero them once if the consumer reads whole slots); packets of less than about 1 KiB are slower that
 * way, the padding outweighs it.  unet_tx_build_dev takes any such table. */
int unet_tx_layout_aligned(const uint32_t* payload_len, uint32_t n, int mode, uint32_t align16, uint32_t* out_off16,
                           uint64_t* out_arena_bytes);
/* Host-pointer form: stages payloads and descriptors, builds on the GPU, copies the packets back packed
 * back to back (layout as unet_tx_layout; out_off16 and out_len are written).  Slots of refused packets
 * are zero-filled.  From 16 MiB of payloads or packets up, a batch whose payloads lie in the arena in index order is
 * cut into runs of packets: while one run is built the next one's payloads go up and the last one's packets come
 * back, both directions of the host link at once (262,144 datagrams of 1472 bytes: 31 -> 14 ms per call). */
int unet_tx_build(unet_ctx* ctx, const uint8_t* payload_arena, uint64_t payload_arena_bytes, const uint32_t* payload_off16,
                  uint32_t payload_stride16, const uint32_t* payload_len, const unet_tx_desc* desc, uint32_t n,
                  const unet_iface* iface, uint8_t* out_arena, uint64_t out_arena_bytes, uint32_t* out_off16,
                  uint32_t* out_len, uint8_t* status);

/* ------------------------------------------------------------ counters */
/* Histogram of verdicts: d_counts[v] += number of results with verdict v (UNET_V_COUNT
 * u64 slots in device memory, caller zeroes them; NULL = the context's own counters, see unet_counters).  The reference keeps no counters; these are what a
 * multi-GPU job sums at the end (the only value that ever crosses GPUs). */
int unet_count_verdicts_dev(unet_ctx* ctx, void* stream, const unet_result* d_results, uint32_t n,
                            uint64_t* d_counts);
/* Every context keeps UNET_V_COUNT counters of its own in device memory: the host-pointer entry points
 * (unet_verify, unet_verify_async, unet_verify_static, unet_ipv4_validate) add the verdicts of every record they
 * judge, and unet_count_verdicts_dev adds to them when d_counts is NULL.  unet_counters waits for the context's
 * streams and copies them out (reset != 0: and zeroes them); unet_counters_sum adds up the counters of several
 * contexts, e.g. one per GPU of this process.  Across processes (one rank per GPU) the UNET_V_COUNT * 8 = 320 bytes
 * are summed by the job's own collective: an RCCL all-reduce, or the launcher's.  unet_counters waits for the
 * context's OWN streams only: a unet_count_verdicts_dev(d_counts = NULL) enqueued on a caller's stream must have been
 * synchronized by the caller before unet_counters is asked.  A counter array holds UNET_V_COUNT entries (40; it was
 * 32 before DROP_TCP_OPTION = 32 existed): size d_counts with the constant, never with a literal. */
int unet_counters(unet_ctx* ctx, uint64_t* out /* UNET_V_COUNT */, int reset);
int unet_counters_sum(unet_ctx* const* ctxs, uint32_t n_ctx, uint64_t* out /* UNET_V_COUNT */, int reset);

/* ------------------------------------------- per-flow TCP tracking (extension) */
/* No counterpart in the reference (it has no TCP, src/transport.rs:10-28); works on records unet_verify* has
 * judged UNET_V_ACCEPT_TCP with iface.ext_tcp = 1.  A flow is one direction of a connection (its 5-tuple).  The
 * table lives in device memory on the GPU that owns the flow -- flows are placed on GPUs by unet_flow_hash5 %
 * n_gpus, so all segments of a flow meet the same table and nothing about it ever crosses GPUs.  It is
 * direct-mapped: a flow whose slot holds another flow takes the slot over. */
typedef struct unet_tcp_flow {   /* one table slot, 32 bytes; all zero = free */
    uint32_t src;
    uint32_t dst;
    uint32_t ports;              /* source port << 16 | destination port */
    uint32_t next_seq;           /* sequence number expected next */
    uint32_t state;              /* UNET_TCP_* */
    uint32_t segments;           /* segments seen since the flow took the slot */
    uint32_t out_of_order;       /* of those: sequence number not the expected one */
    uint32_t after_close;        /* of those: arrived after a reset */
} unet_tcp_flow;
#define UNET_TCP_FREE 0u
#define UNET_TCP_SYN 1u          /* SYN seen, nothing after it yet */
#define UNET_TCP_ESTABLISHED 2u
#define UNET_TCP_FIN 3u          /* FIN seen in order */
#define UNET_TCP_RESET 4u        /* RST seen; only a new SYN reopens the flow */
/* per-segment outcome, one byte per record */
#define UNET_TCPS_NOT_TCP 0u     /* not an accepted TCP segment: the table is not touched */
#define UNET_TCPS_NEW 1u         /* first segment of its flow in the slot (free slot, take-over, or SYN after a reset) */
#define UNET_TCPS_IN_ORDER 2u    /* sequence number = the expected one */
#define UNET_TCPS_OUT_OF_ORDER 3u/* a gap (expectation moves past it) or old data (expectation unchanged) */
#define UNET_TCPS_AFTER_CLOSE 4u /* segment of a reset flow */
/* Applies the records of a batch to the table IN RECORD ORDER per flow (record i before record j > i) and writes
 * each record's outcome to d_seg_state[i].  d_results are the verdicts unet_verify_dev wrote for the same batch
 * (same iface mode).  table_slots must be a power of two (1 .. 2^30); the caller zeroes a new table.  The
 * library's own workspace (about 45 bytes per record) grows on demand, so this call may allocate, and a
 * reallocation waits for the device (unet_tcp_track_ws_dev below takes the caller's workspace instead); it is
 * enqueued on `stream` like every *_dev call. */
int unet_tcp_track_dev(unet_ctx* ctx, void* stream, const uint8_t* d_arena, uint64_t arena_bytes,
                       const uint32_t* d_off16, uint32_t stride16, const uint32_t* d_len,
                       const unet_result* d_results, uint32_t n, const unet_iface* iface,
                       unet_tcp_flow* d_table, uint32_t table_slots, uint8_t* d_seg_state);
/* The same with the workspace in the caller's hands, so that the call itself never allocates or frees (a reallocation
 * of the library's own buffer waits for the whole device): unet_tcp_track_workspace says how many bytes n records
 * need (about 45 per record; host code), d_workspace is device memory of at least that size on a 256-byte boundary.
 * d_stats (may be NULL; the caller zeroes it) receives how the batch fell onto the table: the walk over the
 * segments of one table slot is sequential by definition, so one hot flow -- or a table far too small -- bounds the
 * call.  Runs of more than 64 segments are walked by a whole wavefront that gathers them into LDS 64 at a time (no
 * step of the walk waits for memory: about 20 ns per segment instead of a microsecond); longest_run tells the caller
 * when to split its batches or grow its table. */
typedef struct unet_tcp_stats {
    uint32_t longest_run;  /* most segments that met one table slot in one call (maximum over the calls since it was zeroed) */
    uint32_t long_runs;    /* runs of more than 64 segments (sum over those calls) */
} unet_tcp_stats;
int unet_tcp_track_workspace(uint32_t n, uint32_t table_slots, uint64_t* bytes);
int unet_tcp_track_ws_dev(unet_ctx* ctx, void* stream, const uint8_t* d_arena, uint64_t arena_bytes,
                          const uint32_t* d_off16, uint32_t stride16, const uint32_t* d_len,
                          const unet_result* d_results, uint32_t n, const unet_iface* iface,
                          unet_tcp_flow* d_table, uint32_t table_slots, uint8_t* d_seg_state,
                          void* d_workspace, uint64_t workspace_bytes, unet_tcp_stats* d_stats);
/* Table slot a flow uses: a mix of unet_flow_hash5(src, dst, 6, sport, dport) independent of the bits that chose
 * the GPU.  Host code. */
uint32_t unet_tcp_slot(uint32_t src, uint32_t dst, uint16_t sport, uint16_t dport, uint32_t table_slots);

/* --------------------------------------------------- flow-hash sharding */
/* Build-defined 32-bit mix of the 5-tuple (ports are 0 unless protocol is UDP, or TCP
 * with ext_tcp): the shard of a frame is hash % n_gpus, so a flow stays on one GPU.
 * unet_flow_hash5 is the scalar definition; the _dev form parses each record's headers
 * (MODE_L2 or MODE_L3 as in iface) and writes one hash per record (0 if the record has
 * no complete IPv4 header). */
uint32_t unet_flow_hash5(uint32_t src, uint32_t dst, uint8_t proto, uint16_t sport, uint16_t dport);
/* Host code: the same hash from the bytes of one record (mode as in unet_iface: UNET_MODE_L3 packet or
 * UNET_MODE_L2 frame).  Ports count for UDP and TCP when the buffer reaches them, else 0; a buffer that
 * holds no IPv4 header hashes to 0.  Lets a dispatcher place frames on GPUs before any GPU has seen them;
 * equals unet_flow_hash_dev's result for the same bytes. */
uint32_t unet_flow_hash_record(const uint8_t* rec, size_t len, int mode);
int unet_flow_hash_dev(unet_ctx* ctx, void* stream, const uint8_t* d_arena, uint64_t arena_bytes,
                       const uint32_t* d_off16, uint32_t stride16, const uint32_t* d_len, uint32_t n,
                       const unet_iface* iface, uint32_t* d_hash);

/* ------------------------------------------------ synthetic frame traces */
/* Deterministic counter-based traces used by the benchmarks and parity tests: record
 * `idx` of trace (kind, seed) is a pure function of (kind, seed, idx), identical between
 * the host twin and the device kernel.  All traces are UNET_MODE_L3 records addressed to
 * `iface` with per-index fault injection. */
#define UNET_SYNTH_IPV4_HDR20 2 /* bare 20-byte headers, stride 20 (use with unet_ipv4_validate) */
#define UNET_SYNTH_UDP1500 3    /* 1500-byte IPv4+UDP packets, stride16 = 94 */
#define UNET_SYNTH_IMIX 4       /* mixed 64..9000-byte TCP/UDP/ICMP packets, packed at 16-byte boundaries */
#define UNET_SYNTH_FLOWS 5      /* IMIX shapes whose 5-tuples come from a table of 2^20 flows of skewed popularity */
#define UNET_SYNTH_ODD 6        /* the IMIX mix with lengths 65 / 577 / 1499 / 4097 / 8999: odd, and one past or short of a 16-byte boundary */

/* Length of record idx (bytes) for a trace kind; pure function, host side. */
uint32_t unet_synth_len(int kind, uint64_t seed, uint64_t idx);
/* Fill off16[0..n) / len[0..n) for records first_idx..first_idx+n and return the arena size. */
int unet_synth_layout(int kind, uint64_t seed, uint64_t first_idx, uint32_t n, uint32_t* off16,
                      uint32_t* len, uint64_t* arena_bytes);
/* Host twin: write the records into `arena` (layout as produced by unet_synth_layout). */
int unet_synth_host(int kind, uint64_t seed, uint64_t first_idx, uint32_t n, const unet_iface* iface,
                    const uint32_t* off16, const uint32_t* len, uint8_t* arena, uint64_t arena_bytes);
/* Device kernel: same bytes, generated in HBM. */
int unet_synth_dev(unet_ctx* ctx, void* stream, int kind, uint64_t seed, uint64_t first_idx, uint32_t n,
                   const unet_iface* iface, const uint32_t* d_off16, const uint32_t* d_len,
                   uint8_t* d_arena, uint64_t arena_bytes);

/* Same as unet_synth_dev for an arbitrary set of trace indices: record r is trace index
 * first_idx + d_rel_idx[r] (d_rel_idx NULL = r).  d_off16 / d_len describe where each goes. */
int unet_synth_indexed_dev(unet_ctx* ctx, void* stream, int kind, uint64_t seed, uint64_t first_idx,
                           const uint32_t* d_rel_idx, uint32_t n, const unet_iface* iface, const uint32_t* d_off16,
                           const uint32_t* d_len, uint8_t* d_arena, uint64_t arena_bytes);
/* Lengths of those records (their layout is an exclusive scan of ceil(len/16)). */
int unet_synth_len_dev(unet_ctx* ctx, void* stream, int kind, uint64_t seed, uint64_t first_idx,
                       const uint32_t* d_rel_idx, uint32_t n, uint32_t* d_len);
/* Flow-hash placement of a trace without touching record bytes: d_shard[i] = GPU (0..world-1) that owns
 * trace index first_idx + i, i.e. unet_flow_hash5 of the record's 5-tuple modulo `world`.
 * unet_synth_flow_hash is the host-side scalar form of the hash. */
int unet_synth_shard_dev(unet_ctx* ctx, void* stream, int kind, uint64_t seed, uint64_t first_idx, uint32_t n,
                         const unet_iface* iface, uint32_t world, uint8_t* d_shard);
uint32_t unet_synth_flow_hash(int kind, uint64_t seed, uint64_t idx, const unet_iface* iface);

/* ------------------------------------------------------- measurement aid */
/* Read-only streaming kernel over `bytes` of device memory (16 B per lane, grid-stride):
 * the attainable-bandwidth yardstick reported beside the 8 TB/s peak.  d_sink receives
 * one u32 per workgroup (at least 4096 slots). */
int unet_stream_read_dev(unet_ctx* ctx, void* stream, const uint8_t* d_buf, uint64_t bytes,
                         uint32_t* d_sink);
/* The same read as short-lived wavefronts that own 24 KiB each (the shape of the record kernels' own reads).
 * A benchmark takes the better of the two as what a pure read of its buffer attains on that GPU. */
int unet_stream_read_pieces_dev(unet_ctx* ctx, void* stream, const uint8_t* d_buf, uint64_t bytes,
                                uint32_t* d_sink);
/* The same pieces with as many resident wavefronts as the CU holds (the call above caps them at three workgroups per
 * CU, which reads faster on most GPUs of this type and slower on some). */
int unet_stream_read_pieces_full_dev(unet_ctx* ctx, void* stream, const uint8_t* d_buf, uint64_t bytes,
                                     uint32_t* d_sink);
/* A third shape: the buffer read as packed records of `record_bytes` each (1..65520, rounded up to 16), sixteen to a wavefront,
 * with the loads the record kernels issue -- whole streaming rows and one cached row holding the record's end -- and
 * only a sum taken.  (A measurement aid: it keeps a table of n equal lengths in the context and may allocate it.) */
int unet_stream_read_records_dev(unet_ctx* ctx, void* stream, const uint8_t* d_buf, uint64_t bytes, uint32_t record_bytes,
                                 uint32_t* d_sink);
/* Debug builds only (hipcc -DUNET_DEBUG, e.g. tools/variants.sh debug): every store the kernels make into a
 * wavefront's LDS area is checked against that area's bounds; a store outside is dropped and counted, and this call
 * waits for the device and returns the count (reset != 0: and zeroes it).  A release build returns UNET_E_INVAL.
 * No counterpart in the reference; it exists so that one run of the parity soak can rule out a whole class of
 * wave-against-wave LDS races. */
int unet_debug_lds_violations(unet_ctx* ctx, uint32_t* out, int reset);
/* ------------------------------------------------ arena packing (host code, no GPU involved) */
/* Bytes an arena needs for n records of these lengths, each padded to a 16-byte boundary (at least 16). */
int unet_pack_size(const uint32_t* len, uint32_t n, uint64_t* arena_bytes);
/* Copies n separately owned buffers (e.g. the Vec<u8> of each Ipv4QueueEntry, src/protocols.rs:52-57) into one
 * arena, back to back on 16-byte boundaries, pad bytes zeroed; writes off16[i].  The arena must be 16-byte
 * aligned and at least unet_pack_size() bytes long. */
int unet_pack_frames(const uint8_t* const* frames, const uint32_t* len, uint32_t n, uint8_t* arena, uint64_t arena_bytes,
                     uint32_t* off16);
/* The same from one length-prefixed byte stream: n records, each a little-endian u32 length followed by that
 * many bytes, no padding between records.  n_out receives the record count; with arena == NULL only the
 * count and *arena_bytes are computed (off16/len may be NULL then). */
int unet_pack_length_prefixed(const uint8_t* stream, uint64_t stream_bytes, uint8_t* arena, uint64_t* arena_bytes,
                              uint32_t* off16, uint32_t* len, uint32_t cap, uint32_t* n_out);

/* ------------------------------------------------ capture files <-> arenas (host code, no GPU involved) */
/* What the reference's two devices hand to the stack: tap::read fills a 1514-byte buffer
 * (src/driver/tap.rs:168-173, ETHERNET_FRAME_MAX_SIZE src/devices/ethernet.rs:10), so a longer frame arrives
 * cut at 1514 bytes; the loopback device carries up to 65535 bytes (src/devices/loopback.rs:16). */
#define UNET_TAP_FRAME_MAX 1514u
#define UNET_LOOPBACK_MTU 65535u

typedef struct unet_pcap_info {
    uint32_t linktype;      /* 0 NULL, 1 EN10MB, 101 RAW, 228 IPV4 */
    uint32_t snaplen;
    uint32_t mode;          /* UNET_MODE_L2 for Ethernet captures, UNET_MODE_L3 for the others */
    uint32_t skip;          /* link-layer bytes dropped in front of every record (4 for NULL) */
    uint32_t n_records;     /* complete records in the file */
    uint32_t n_snapped;     /* records the capture itself had cut (caplen < origlen) */
    uint32_t n_clipped;     /* records cut to max_len by this import */
    uint32_t nanosecond;    /* 1 if timestamps are in nanoseconds */
    uint64_t arena_bytes;   /* size of the arena unet_pcap_pack needs (>= 16) */
    uint64_t trailing_bytes;/* bytes after the last complete record */
} unet_pcap_info;

/* Classic pcap file (either byte order, micro- or nanosecond variant) held in memory -> counts and sizes.
 * Records longer than max_len (UNET_TAP_FRAME_MAX, UNET_LOOPBACK_MTU, at most 65535) are cut to it. */
int unet_pcap_scan(const uint8_t* file, uint64_t file_bytes, uint32_t max_len, unet_pcap_info* info);
/* Packs the records into an arena (16-byte aligned, info.arena_bytes long; pad bytes are zeroed):
 * record i at 16*off16[i], len[i] bytes; n must be info.n_records.  ts_ns (may be NULL) receives the
 * timestamps in nanoseconds. */
int unet_pcap_pack(const uint8_t* file, uint64_t file_bytes, uint32_t max_len, uint8_t* arena, uint64_t arena_bytes,
                   uint32_t* off16, uint32_t* len, uint64_t* ts_ns, uint32_t n);
/* Arena -> pcap (little-endian, microsecond timestamps i microseconds apart, snaplen 65535).  Returns the number
 * of bytes the file needs in *file_bytes; writes it when `file` is non-NULL and file_cap is large enough.
 * Records with len 0 (e.g. packets unet_tx_build_dev refused) are skipped. */
int unet_pcap_write(const uint8_t* arena, uint64_t arena_bytes, const uint32_t* off16, uint32_t stride16,
                    const uint32_t* len, uint32_t n, uint32_t linktype, uint8_t* file, uint64_t file_cap,
                    uint64_t* file_bytes);

/* Read+write yardstick for the packet builder: copies bytes/16 16-byte chunks with streaming loads and
 * stores.  No counterpart in the reference. */
int unet_stream_copy_dev(unet_ctx* ctx, void* stream, const uint8_t* d_src, uint8_t* d_dst, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* UNET_OFFLOAD_H */
