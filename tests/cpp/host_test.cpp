// Exercises unet_amd/host/unet_host.hpp.  `--cpu`: host logic only (arena packer, PCB table).
// `--gpu`: the batched drain loop against the CPU oracle, packet by packet.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../oracle/unet_oracle.h"
#include "../../unet_amd/host/unet_host.hpp"

using namespace unet;

static int g_fail = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);    \
            g_fail++;                                                      \
        }                                                                  \
    } while (0)

static std::vector<uint8_t> ip_packet(uint8_t proto, uint32_t src, uint32_t dst, const std::vector<uint8_t>& l4, uint16_t id = 7) {
    std::vector<uint8_t> p(20 + l4.size());
    oracle_create_ip_header(id, proto, src, dst, l4.size(), p.data());
    std::memcpy(p.data() + 20, l4.data(), l4.size());
    return p;
}
static std::vector<uint8_t> udp_segment(uint32_t src, uint32_t dst, uint16_t sport, uint16_t dport, const std::vector<uint8_t>& payload) {
    std::vector<uint8_t> s(8 + payload.size());
    put16(&s[0], sport);
    put16(&s[2], dport);
    put16(&s[4], uint16_t(s.size()));
    std::memcpy(s.data() + 8, payload.data(), payload.size());
    put16(&s[6], oracle_udp_send_checksum(src, dst, sport, dport, payload.data(), payload.size()));
    return s;
}
static std::vector<uint8_t> icmp_message(uint8_t type, uint8_t code, uint32_t values, const std::vector<uint8_t>& data) {
    std::vector<uint8_t> m(8 + data.size());
    m[0] = type;
    m[1] = code;
    put32(&m[4], values);
    std::memcpy(m.data() + 8, data.data(), data.size());
    put16(&m[2], oracle_icmp_send_checksum(type, code, values, data.data(), data.size()));
    return m;
}

// Longest-prefix match as in the reference's own router tests (src/protocols/ipv4.rs:440-491).
static void router_tests() {
    auto a = std::make_shared<Ipv4Interface>(Ipv4Address{0xC0000202}, Ipv4Address{0xFFFFFF00});
    auto b = std::make_shared<Ipv4Interface>(Ipv4Address{0xC0000001}, Ipv4Address{0xFFFF0000});
    Ipv4Router r;
    CHECK(r.lookup(Ipv4Address{0xC0000203}) == nullptr);
    r.register_route(Ipv4Address{0xC0000000}, b);
    r.register_route(Ipv4Address{0xC0000200}, a);
    const IpRoute* x = r.lookup(Ipv4Address{0xC0000203});
    CHECK(x && x->interface == a && !x->next_hop);   // /24 beats /16
    x = r.lookup(Ipv4Address{0xC0000303});
    CHECK(x && x->interface == b);
    CHECK(r.lookup(Ipv4Address{0x08080808}) == nullptr);
    r.register_default(a, Ipv4Address{0xC0000201});
    x = r.lookup(Ipv4Address{0x08080808});
    CHECK(x && x->interface == a && x->next_hop && x->next_hop->v == 0xC0000201);
    x = r.lookup(Ipv4Address{0xC0000203});
    CHECK(x && !x->next_hop);                        // a more specific route still wins over the default
}

static void cpu_tests() {
    router_tests();
    FrameArena a;
    const uint8_t x[40] = {1, 2, 3};
    CHECK(a.push(x, 0) == 0);
    CHECK(a.push(x, 1) == 1);
    CHECK(a.push(x, 16) == 2);
    CHECK(a.push(x, 17) == 3);
    CHECK(a.off16()[0] == 0 && a.off16()[1] == 0 && a.off16()[2] == 1 && a.off16()[3] == 2);
    CHECK(a.bytes() == 64 && (reinterpret_cast<uintptr_t>(a.data()) & 15) == 0);
    CHECK(a.record(3)[0] == 1 && a.record(3)[16] == 0 && a.record_len(3) == 17);
    // udp::bind / select_pcb semantics, src/transport/udp.rs:84-165
    UdpContext u;
    const Ipv4Address any{0}, me{0xC0000202}, other{0xC0000203};
    CHECK(u.bind(Endpoint{any, 8000}).value_or(99) == 0);
    CHECK(!u.bind(Endpoint{me, 8000}).has_value());      // wildcard already holds the port
    CHECK(u.bind(Endpoint{me, 8001}).value_or(99) == 1);
    CHECK(!u.bind(Endpoint{any, 8001}).has_value());     // binding ANY collides with any address on that port
    CHECK(u.bind(Endpoint{other, 8002}).value_or(99) == 2);
    CHECK(u.select_pcb(me, 8000) == u.slot(0));
    CHECK(u.select_pcb(other, 8001) == nullptr);
    CHECK(u.select_pcb(other, 8002) == u.slot(2));
    for (int i = 3; i < 16; i++) CHECK(u.bind(Endpoint{any, uint16_t(9000 + i)}).has_value());
    CHECK(!u.bind(Endpoint{any, 9999}).has_value());     // 16 slots
    Ipv4Interface tap(Ipv4Address{0xC0000202}, Ipv4Address{0xFFFFFF00});
    CHECK(tap.broadcast.v == 0xC00002FF);
    Ipv4Interface lo(Ipv4Address{0x7F000001}, Ipv4Address{0xFF000000});
    CHECK(lo.broadcast.v == 0x7F0000FF);
    Ipv4IdGenerator g;
    CHECK(g.next() == 0 && g.next() == 1);
}

static void gpu_tests() {
    Offload gpu(0);
    // the reference's own checksum test, through the C++ wrapper (src/utils.rs:43-50)
    const uint8_t kat[18] = {0x1f, 0x40, 0x1f, 0x41, 0x0, 0x1c, 0x0, 0x0, 0x30, 0x31, 0x32, 0x33, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39};
    const uint16_t s = gpu.calculate_checksum(kat, 18, 0);
    CHECK(s == 0xbc58 && gpu.calculate_checksum(kat, 18, s) == 0);

    auto tap = std::make_shared<Ipv4Interface>(Ipv4Address{0xC0000202}, Ipv4Address{0xFFFFFF00});
    auto lo = std::make_shared<Ipv4Interface>(Ipv4Address{0x7F000001}, Ipv4Address{0xFF000000});
    lo->mtu = 65535;
    const uint32_t peer = 0xC0000201, me = tap->unicast.v;
    std::vector<uint8_t> pay(300), big(3000);
    for (size_t i = 0; i < pay.size(); i++) pay[i] = uint8_t(i * 7 + 3);
    for (size_t i = 0; i < big.size(); i++) big[i] = uint8_t(i * 13 + 1);

    NetProtocol proto;
    auto push = [&](std::vector<uint8_t> p, std::shared_ptr<Ipv4Interface> i) { proto.queue.push_back(Ipv4QueueEntry{std::move(p), std::move(i)}); };
    push(ip_packet(17, peer, me, udp_segment(peer, me, 5000, 8000, pay)), tap);                       // 0 accepted, bound port
    push(ip_packet(17, peer, me, udp_segment(peer, me, 5000, 8123, pay)), tap);                       // 1 accepted, no socket
    push(ip_packet(1, peer, me, icmp_message(8, 0, 0x00800001, std::vector<uint8_t>(56, 0x5a))), tap); // 2 echo
    push(ip_packet(1, peer, me, icmp_message(0, 0, 0x00800002, pay)), tap);                            // 3 echo reply
    {
        auto p = ip_packet(17, peer, me, udp_segment(peer, me, 1, 8000, pay));
        p[100] ^= 0x10;
        push(p, tap);                                                                                   // 4 bad udp checksum
    }
    push(ip_packet(6, peer, me, std::vector<uint8_t>(40, 0)), tap);                                    // 5 tcp, strict mode
    push(ip_packet(1, 0x7F000001, 0x7F000001, icmp_message(8, 3, 42, big)), lo);                       // 6 echo on loopback, 3 KB
    push(ip_packet(1, peer, tap->broadcast.v, icmp_message(8, 0, 1, pay)), tap);                       // 7 echo to broadcast
    push(ip_packet(17, peer, 0xC0000299, udp_segment(peer, 0xC0000299, 1, 8000, pay)), tap);           // 8 not for me
    push(ip_packet(17, peer, me, udp_segment(peer, me, 77, 8000, {})), tap);                           // 9 empty datagram
    const std::vector<Ipv4QueueEntry> copy(proto.queue.begin(), proto.queue.end());

    UdpContext pcbs;
    CHECK(pcbs.bind(Endpoint{Ipv4Address{0}, 8000}).has_value());
    Ipv4IdGenerator ids;
    const RecvReport rep = proto.recv_batch(gpu, pcbs, ids);
    CHECK(proto.queue.empty() && rep.results.size() == copy.size());

    // every verdict equals the oracle's for the same bytes and interface
    for (size_t i = 0; i < copy.size(); i++) {
        const unet_iface ci = copy[i].interface->c_iface();
        oracle_iface oi{};
        std::memcpy(&oi, &ci, sizeof oi);
        const oracle_result o = oracle_ipv4_recv(copy[i].data.data(), copy[i].data.size(), &oi);
        CHECK(o.verdict == rep.results[i].verdict && o.csum == rep.results[i].csum && o.proto == rep.results[i].proto);
    }
    const int want[10] = {20, 20, 15, 16, 19, 6, 15, 15, 12, 20};
    for (int i = 0; i < 10; i++) CHECK(rep.results[i].verdict == want[i]);

    // UDP: two datagrams reached the bound socket, in queue order, with the reference's endpoint quirk
    CHECK(rep.udp_enqueued == 2);
    UdpContext::Pcb* sock = pcbs.slot(0);
    CHECK(sock && sock->queue.size() == 2);
    if (sock && sock->queue.size() == 2) {
        CHECK(sock->queue[0].data == pay && sock->queue[0].foreign.port == 5000 && sock->queue[0].foreign.address.v == me);
        CHECK(sock->queue[1].data.empty() && sock->queue[1].foreign.port == 77);
    }
    // errors: no socket, bad checksum, protocol 6, broadcast echo cannot be answered (the reply's source would
    // be the broadcast address, which ipv4::send refuses: ipv4.rs:324-329)
    CHECK(rep.errors.size() == 4);
    bool saw_socket = false, saw_ck = false, saw_proto = false, saw_src = false;
    for (const std::string& e : rep.errors) {
        saw_socket |= e.find("udp socket not found, dst: 192.0.2.2, port: 8123") == 0;
        saw_ck |= e.find("invalid udp checksum: 0x") == 0;
        saw_proto |= e == "unknown transport protocol number: 6";
        saw_src |= e.find("unable to send packet with the source address, src: 192.0.2.255") == 0;
    }
    CHECK(saw_socket && saw_ck && saw_proto && saw_src);

    // replies: byte-identical to what icmp::send + create_ip_header produce on the CPU
    CHECK(rep.replies.size() == 2);
    if (rep.replies.size() == 2) {
        // the loopback interface's packets form the second launch, so its reply takes the second id
        const std::vector<uint8_t> r0 = ip_packet(1, me, peer, icmp_message(0, 0, 0x00800001, std::vector<uint8_t>(56, 0x5a)), 0);
        const std::vector<uint8_t> r1 = ip_packet(1, 0x7F000001, 0x7F000001, icmp_message(0, 3, 42, big), 1);
        CHECK(rep.replies[0].bytes == r0 && rep.replies[0].dst.v == peer);
        CHECK(rep.replies[1].bytes == r1 && rep.replies[1].dst.v == 0x7F000001);
    }
}

// udp::send / icmp::send batches against oracle_tx_build, packet by packet; the reference's refusals.
static void gpu_send_tests() {
    Offload gpu(0);
    Ipv4Interface tap(Ipv4Address{0xC0000202}, Ipv4Address{0xFFFFFF00});
    tap.mac = {0x02, 0xaa, 0xbb, 0xcc, 0xdd, 0xee};
    const MacAddress peer_mac = {0x02, 0x11, 0x22, 0x33, 0x44, 0x55};
    const ArpResolver arp = [&](Ipv4Address a) -> std::optional<MacAddress> {
        if (a.v == 0xC0000201) return peer_mac;
        return std::nullopt;
    };
    std::vector<TxRequest> reqs;
    auto udp = [&](uint32_t src, uint32_t dst, size_t n) {
        TxRequest r;
        r.proto = 17;
        r.src = Endpoint{Ipv4Address{src}, 8000};
        r.dst = Endpoint{Ipv4Address{dst}, 8001};
        r.data.resize(n);
        for (size_t i = 0; i < n; i++) r.data[i] = uint8_t(i * 31 + n);
        return r;
    };
    reqs.push_back(udp(0xC0000202, 0xC0000201, 20));     // 0 ok
    reqs.push_back(udp(0, 0xC0000201, 1400));            // 1 ok, ANY source (pseudo-header uses 0.0.0.0)
    reqs.push_back(udp(0, 0xFFFFFFFF, 10));              // 2 "source address is required for broadcast packet"
    reqs.push_back(udp(0xC0000203, 0xC0000201, 10));     // 3 "unable to send packet with the source address"
    reqs.push_back(udp(0xC0000202, 0xC0000201, 1492));   // 4 "packet too long" (8 + 1492 is not < 1500)
    reqs.push_back(udp(0xC0000202, 0xC0000201, 70000));  // 5 "udp packet too long"
    reqs.push_back(udp(0xC0000202, 0xC00002FF, 0));      // 6 ok, subnet broadcast, empty datagram
    reqs.push_back(udp(0xC0000202, 0xC0000207, 33));     // 7 L2: no ARP entry -> unresolved; L3: ok
    {
        TxRequest r;
        r.proto = 1;
        r.src.address.v = 0xC0000202;
        r.dst.address.v = 0xC0000201;
        r.icmp_type = 8;
        r.icmp_code = 0;
        r.icmp_values = 0x00800001;
        r.data.assign(56, 0x5a);
        reqs.push_back(r);                               // 8 ok, echo request
    }
    for (int l2 = 0; l2 < 2; l2++) {
        Ipv4IdGenerator ids;
        ids.id = 128;
        const TxReport rep = send_batch(gpu, tap, ids, reqs, l2 != 0, arp);
        const std::vector<int> ok = l2 ? std::vector<int>{0, 1, 6, 8} : std::vector<int>{0, 1, 6, 7, 8};
        CHECK(rep.packets.size() == ok.size());
        CHECK(rep.errors.size() == 4 && rep.unresolved == (l2 ? 1u : 0u));
        CHECK(ids.id == 128 + 5);  // the unresolved packet consumed an id as well
        const unet_iface ci = tap.c_iface(l2 != 0);
        oracle_iface oi{};
        std::memcpy(&oi, &ci, sizeof oi);
        const int id_of[9] = {128, 129, -1, -1, -1, -1, 130, 131, 132};
        for (size_t k = 0; k < ok.size() && k < rep.packets.size(); k++) {
            const TxRequest& r = reqs[ok[k]];
            oracle_tx_desc d{};
            d.src = r.src.address.v;
            d.dst = r.dst.address.v;
            d.proto = r.proto;
            d.ip_id = uint16_t(id_of[ok[k]]);
            if (r.proto == 17) { d.l4a = r.src.port; d.l4b = r.dst.port; }
            else { d.l4a = uint16_t((r.icmp_type << 8) | r.icmp_code); d.values = r.icmp_values; }
            if (l2) std::memcpy(d.dst_mac, r.dst.address.v == 0xC00002FF ? MacAddress{0xff, 0xff, 0xff, 0xff, 0xff, 0xff}.data() : peer_mac.data(), 6);
            std::vector<uint8_t> want(14 + 20 + 8 + r.data.size() + 64);
            uint8_t st = 0;
            want.resize(oracle_tx_build(&d, r.data.data(), r.data.size(), &oi, want.data(), &st));
            CHECK(st == 0 && rep.packets[k].bytes == want && rep.packets[k].dst.v == r.dst.address.v);
        }
        bool e0 = false, e1 = false, e2 = false, e3 = false;
        for (const std::string& e : rep.errors) {
            e0 |= e == "source address is required for broadcast packet";
            e1 |= e == "unable to send packet with the source address, src: 192.0.2.3, interface: 192.0.2.2";
            e2 |= e == "packet too long, len: 1500, mtu: 1500";
            e3 |= e == "udp packet too long, len: 70000, max: 65507";
        }
        CHECK(e0 && e1 && e2 && e3);
    }
}

// Three contexts on one GPU stand in for three GPUs: flow-hash placement, per-shard threads, merged verdicts.
static void gpu_sharded_tests() {
    const uint32_t n = 6000;
    unet_iface ifc{};
    ifc.unicast = 0xC0000202;
    ifc.broadcast = 0xC00002FF;
    ifc.ext_tcp = 1;
    std::vector<uint32_t> off(n), len(n);
    uint64_t bytes = 0;
    CHECK(unet_synth_layout(UNET_SYNTH_FLOWS, 0x756e6574ull, 4242, n, off.data(), len.data(), &bytes) == UNET_OK);
    std::vector<uint8_t> arena(bytes + 64);
    uint8_t* base = arena.data() + ((16 - (reinterpret_cast<uintptr_t>(arena.data()) & 15)) & 15);
    CHECK(unet_synth_host(UNET_SYNTH_FLOWS, 0x756e6574ull, 4242, n, &ifc, off.data(), len.data(), base, bytes) == UNET_OK);
    std::vector<std::vector<uint8_t>> frames(n);
    for (uint32_t i = 0; i < n; i++) frames[i].assign(base + size_t(off[i]) * 16, base + size_t(off[i]) * 16 + len[i]);

    ShardedOffload many({0, 0, 0});
    const ShardedOffload::Report rep = many.verify(frames, ifc);
    Offload one(0);
    FrameArena a;
    for (auto& f : frames) a.push(f.data(), f.size());
    const std::vector<unet_result> ref = one.verify(a, ifc);
    oracle_iface oi{};
    std::memcpy(&oi, &ifc, sizeof oi);
    size_t sizes[3] = {0, 0, 0};
    bool same = true, placed = true;
    for (uint32_t i = 0; i < n; i++) {
        same &= std::memcmp(&rep.results[i], &ref[i], sizeof(unet_result)) == 0;
        const oracle_result o = oracle_ipv4_recv(frames[i].data(), frames[i].size(), &oi);
        same &= o.verdict == rep.results[i].verdict && o.csum == rep.results[i].csum;
        placed &= rep.shard[i] == oracle_flow_hash_record(frames[i].data(), frames[i].size(), &oi) % 3;
        sizes[rep.shard[i]]++;
    }
    CHECK(same);
    CHECK(placed);
    CHECK(rep.shard_sizes[0] == sizes[0] && rep.shard_sizes[1] == sizes[1] && rep.shard_sizes[2] == sizes[2]);
    CHECK(sizes[0] > n / 6 && sizes[1] > n / 6 && sizes[2] > n / 6);
    // frames of one flow always meet on the same shard, and the L2 form of a frame hashes like its L3 form
    std::vector<uint8_t> l2(14 + frames[7].size());
    l2[12] = 0x08;
    std::memcpy(l2.data() + 14, frames[7].data(), frames[7].size());
    CHECK(unet_flow_hash_record(l2.data(), l2.size(), UNET_MODE_L2) == unet_flow_hash_record(frames[7].data(), frames[7].size(), UNET_MODE_L3));
    CHECK(unet_flow_hash_record(l2.data(), 13, UNET_MODE_L2) == 0 && unet_flow_hash_record(frames[7].data(), 19, UNET_MODE_L3) == 0);

    // the counters of the three contexts add up to the histogram of the merged verdicts
    std::array<uint64_t, UNET_V_COUNT> hist{};
    for (const unet_result& r : rep.results) hist[r.verdict]++;
    CHECK(many.counters(true) == hist);
    const std::array<uint64_t, UNET_V_COUNT> zero{};
    CHECK(many.counters() == zero);

    // transmit-side fill over the shards: blank the checksum fields of the UDP/ICMP packets, fill on three
    // contexts, compare byte for byte with the oracle's fill of the same packets
    std::vector<std::vector<uint8_t>> blank, want;
    for (uint32_t i = 0; i < n && blank.size() < 2000; i++) {
        std::vector<uint8_t> p = frames[i];
        if (p.size() < 28 || (p[9] != 17 && p[9] != 1)) continue;
        p[10] = p[11] = 0;
        blank.push_back(p);
        want.push_back(p);
    }
    std::vector<uint8_t> want_status(blank.size());
    for (size_t i = 0; i < want.size(); i++) want_status[i] = oracle_generate(want[i].data(), want[i].size());
    const std::vector<uint8_t> got_status = many.generate(blank);
    CHECK(got_status == want_status);
    CHECK(blank == want);
    CHECK(blank.size() > 100);

    // whole packets over the shards: same bytes as one context builds, and as the oracle builds
    std::vector<std::vector<uint8_t>> payloads;
    std::vector<unet_tx_desc> desc;
    for (uint32_t i = 0; i < 1500; i++) {
        const size_t plen = (i * 37u) % 1400u;
        std::vector<uint8_t> pl(plen);
        for (size_t k = 0; k < plen; k++) pl[k] = uint8_t(k * 7 + i);
        unet_tx_desc d{};
        d.src = ifc.unicast;
        d.dst = 0xC0000200u | (i % 200u + 1u);
        d.proto = i % 5 == 0 ? 1 : 17;
        d.l4a = d.proto == 17 ? uint16_t(1000 + i % 50) : uint16_t(8 << 8);
        d.l4b = uint16_t(2000 + i % 7);
        d.values = 0x00010000u + i;
        d.ip_id = uint16_t(i);
        payloads.push_back(std::move(pl));
        desc.push_back(d);
    }
    unet_iface tx_if = ifc;
    tx_if.ext_tcp = 0;
    std::vector<uint8_t> st_many;
    const std::vector<std::vector<uint8_t>> built = many.tx_build(payloads, desc, tx_if, &st_many);
    bool tx_same = built.size() == payloads.size();
    oracle_iface toi{};
    std::memcpy(&toi, &tx_if, sizeof toi);
    for (size_t i = 0; i < built.size(); i++) {
        oracle_tx_desc od{};
        std::memcpy(&od, &desc[i], sizeof od);
        std::vector<uint8_t> w(28 + payloads[i].size() + 64);
        uint8_t st = 0;
        w.resize(oracle_tx_build(&od, payloads[i].data(), payloads[i].size(), &toi, w.data(), &st));
        tx_same &= st == 0 && st_many[i] == 0 && built[i] == w;
    }
    CHECK(tx_same);
}

// ipv4::send's routing step: a destination behind the default gateway is framed with the GATEWAY's MAC.
static void gpu_routed_send_tests() {
    Offload gpu(0);
    auto tap = std::make_shared<Ipv4Interface>(Ipv4Address{0xC0000202}, Ipv4Address{0xFFFFFF00});
    tap->mac = {0x02, 0xaa, 0xbb, 0xcc, 0xdd, 0xee};
    auto lo = std::make_shared<Ipv4Interface>(Ipv4Address{0x7F000001}, Ipv4Address{0xFF000000});
    lo->mtu = 65535;
    Ipv4Router router;
    router.register_route(Ipv4Address{0x7F000000}, lo);
    router.register_route(Ipv4Address{0xC0000200}, tap);
    router.register_default(tap, Ipv4Address{0xC0000201});
    const MacAddress gw_mac = {0x02, 0x11, 0x22, 0x33, 0x44, 0x55}, peer_mac = {0x02, 0x66, 0x66, 0x66, 0x66, 0x66};
    const ArpResolver arp = [&](Ipv4Address a) -> std::optional<MacAddress> {
        if (a.v == 0xC0000201) return gw_mac;
        if (a.v == 0xC0000209) return peer_mac;
        return std::nullopt;
    };
    auto udp = [&](uint32_t dst, size_t n) {
        TxRequest r;
        r.proto = 17;
        r.src = Endpoint{Ipv4Address{0}, 8000};
        r.dst = Endpoint{Ipv4Address{dst}, 53};
        r.data.assign(n, 0x42);
        return r;
    };
    std::vector<TxRequest> reqs = {udp(0x08080808, 30), udp(0xC0000209, 31), udp(0xC0000277, 32)};
    Ipv4IdGenerator ids;
    const TxReport rep = send_batch_routed(gpu, router, ids, reqs, true, arp);
    CHECK(rep.packets.size() == 2 && rep.unresolved == 1 && rep.errors.empty());
    if (rep.packets.size() == 2) {
        CHECK(std::memcmp(rep.packets[0].bytes.data(), gw_mac.data(), 6) == 0);    // off-subnet: the gateway's MAC
        CHECK(be32(rep.packets[0].bytes.data() + 14 + 16) == 0x08080808);          // ... the real destination address
        CHECK(std::memcmp(rep.packets[1].bytes.data(), peer_mac.data(), 6) == 0);  // on-subnet: the peer's own MAC
    }
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "--cpu";
    cpu_tests();
    if (mode == "--gpu") {
        gpu_tests();
        gpu_send_tests();
        gpu_sharded_tests();
        gpu_routed_send_tests();
    }
    if (g_fail) {
        std::printf("%d check(s) failed\n", g_fail);
        return 1;
    }
    std::printf("host_test %s ok\n", mode.c_str());
    return 0;
}
