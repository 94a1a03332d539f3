// Host side of the offload for a C++ caller: the batching caller that replaces the reference's
// per-packet drain loop, and the host-side follow-ups that act on the verdicts.
//
// The reference (Rust) handles one packet per call:
//     NetProtocol::recv      src/protocols.rs:83-97     pops queue entries, calls ipv4::recv on each
//     ipv4::recv             src/protocols/ipv4.rs:394-422
//     udp::recv              src/transport/udp.rs:219-279   (validation, then PCB lookup + enqueue)
//     icmp::recv / send      src/transport/icmp.rs:66-122   (echo -> echo reply)
// Here the queue is drained into one arena, ONE unet_verify call decides every packet on the GPU, and
// this file performs what the reference does after its checks have passed: UDP PCB lookup / enqueue
// (udp.rs:256-277) and the echo replies (icmp.rs:110-120).  Transmission is batched the same way:
// udp::send / icmp::send requests (udp.rs:168-216, icmp.rs:66-95) pass ipv4::send's host-side checks
// (ipv4.rs:320-335) here and are then built -- headers, checksums, Ethernet framing -- by ONE
// unet_tx_build call: no checksum is computed and no header is serialised on the host.
//
// Header-only, C++17, needs only include/unet_offload.h and the shared library.
#pragma once
#include <array>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/unet_offload.h"

namespace unet {

// ---------------------------------------------------------------------------------------------
struct Ipv4Address {  // src/protocols/ipv4.rs:22-28
    uint32_t v = 0;
    static constexpr uint32_t ANY = 0x00000000u, BROADCAST = 0xffffffffu;
    bool operator==(const Ipv4Address& o) const { return v == o.v; }
    bool operator!=(const Ipv4Address& o) const { return v != o.v; }
    std::string to_string() const {
        return std::to_string(v >> 24) + "." + std::to_string((v >> 16) & 255) + "." + std::to_string((v >> 8) & 255) + "." +
               std::to_string(v & 255);
    }
};

struct Ipv4Interface {  // src/protocols/ipv4.rs:209-229
    Ipv4Address unicast, netmask, broadcast;
    std::array<uint8_t, 6> mac{};  // the owning device's hw_addr (src/devices.rs:75-90); L2 mode only
    size_t mtu = 1500;             // src/devices/ethernet.rs:13 payload size; loopback uses 65535
    Ipv4Interface(Ipv4Address u, Ipv4Address m) : unicast(u), netmask(m) {
        broadcast.v = (u.v & 0xffffff00u) | (~m.v & 0xffu);  // ipv4.rs:220
    }
    unet_iface c_iface(bool l2 = false, bool ext_tcp = false) const {
        unet_iface i{};
        i.unicast = unicast.v;
        i.broadcast = broadcast.v;
        std::memcpy(i.mac, mac.data(), 6);
        i.mode = l2 ? UNET_MODE_L2 : UNET_MODE_L3;
        i.ext_tcp = ext_tcp ? 1 : 0;
        return i;
    }
};

struct Endpoint {  // src/transport.rs:30-43
    Ipv4Address address;
    uint16_t port = 0;
};

// ---------------------------------------------------------------------------------------------
// Arena packer: every record starts on a 16-byte boundary of a 16-byte-aligned buffer.
class FrameArena {
  public:
    FrameArena() = default;
    FrameArena(const FrameArena&) = delete;
    FrameArena& operator=(const FrameArena&) = delete;
    ~FrameArena() { std::free(buf_); }

    void clear() {
        used_ = 0;
        off16_.clear();
        len_.clear();
    }
    // Appends one record; returns its index.
    size_t push(const uint8_t* data, size_t n) {
        if (n > 65535) throw std::length_error("record longer than 65535 bytes");
        const size_t padded = (n + 15) & ~size_t(15);
        reserve(used_ + padded);
        off16_.push_back(static_cast<uint32_t>(used_ / 16));
        len_.push_back(static_cast<uint32_t>(n));
        if (n) std::memcpy(buf_ + used_, data, n);
        if (padded > n) std::memset(buf_ + used_ + n, 0, padded - n);
        used_ += padded;
        return len_.size() - 1;
    }
    size_t size() const { return len_.size(); }
    uint8_t* data() { reserve(16); return buf_; }
    uint64_t bytes() const { return used_ < 16 ? 16 : used_; }
    const uint32_t* off16() const { return off16_.data(); }
    const uint32_t* len() const { return len_.data(); }
    uint8_t* record(size_t i) { return buf_ + size_t(off16_[i]) * 16; }
    uint32_t record_len(size_t i) const { return len_[i]; }

  private:
    void reserve(size_t want) {
        if (want <= cap_) return;
        size_t cap = cap_ ? cap_ : 4096;
        while (cap < want) cap *= 2;
        void* p = nullptr;
        if (posix_memalign(&p, 64, cap) != 0) throw std::bad_alloc();
        if (used_) std::memcpy(p, buf_, used_);
        std::free(buf_);
        buf_ = static_cast<uint8_t*>(p);
        cap_ = cap;
    }
    uint8_t* buf_ = nullptr;
    size_t cap_ = 0, used_ = 0;
    std::vector<uint32_t> off16_, len_;
};

// ---------------------------------------------------------------------------------------------
// RAII handle on one GPU.
class Offload {
  public:
    explicit Offload(int device = 0) {
        if (unet_open(device, &ctx_) != UNET_OK) throw std::runtime_error(std::string("unet_open: ") + unet_last_error());
    }
    Offload(const Offload&) = delete;
    Offload& operator=(const Offload&) = delete;
    ~Offload() { unet_close(ctx_); }
    unet_ctx* get() const { return ctx_; }

    // `calculate_checksum(data, sum)` of src/utils.rs:1-17, on the GPU.
    uint16_t calculate_checksum(const uint8_t* data, size_t len, uint16_t sum) const {
        uint16_t out = 0;
        if (unet_csum(ctx_, data, len, sum, &out) != UNET_OK) throw std::runtime_error(std::string("unet_csum: ") + unet_last_error());
        return out;
    }
    std::vector<unet_result> verify(FrameArena& a, const unet_iface& iface) const {
        std::vector<unet_result> out(a.size());
        if (a.size() == 0) return out;
        if (unet_verify(ctx_, a.data(), a.bytes(), a.off16(), 0, a.len(), static_cast<uint32_t>(a.size()), &iface, out.data()) != UNET_OK)
            throw std::runtime_error(std::string("unet_verify: ") + unet_last_error());
        return out;
    }
    std::vector<uint8_t> generate(FrameArena& a) const {
        std::vector<uint8_t> status(a.size());
        if (a.size() == 0) return status;
        if (unet_generate(ctx_, a.data(), a.bytes(), a.off16(), 0, a.len(), static_cast<uint32_t>(a.size()), status.data()) != UNET_OK)
            throw std::runtime_error(std::string("unet_generate: ") + unet_last_error());
        return status;
    }

    // Packets built from payloads + descriptors (unet_tx_build); returned packed, one vector per packet
    // (empty for a refused one).
    std::vector<std::vector<uint8_t>> tx_build(FrameArena& payloads, const std::vector<unet_tx_desc>& desc, const unet_iface& iface,
                                               std::vector<uint8_t>* status_out = nullptr) const {
        const uint32_t n = static_cast<uint32_t>(payloads.size());
        std::vector<std::vector<uint8_t>> out(n);
        if (n == 0) return out;
        if (desc.size() != n) throw std::invalid_argument("one descriptor per payload");
        uint64_t need = 0;
        if (unet_tx_layout(payloads.len(), n, iface.mode, nullptr, &need) != UNET_OK)
            throw std::runtime_error(std::string("unet_tx_layout: ") + unet_last_error());
        void* raw = nullptr;
        if (posix_memalign(&raw, 64, need) != 0) throw std::bad_alloc();
        std::unique_ptr<uint8_t, decltype(&std::free)> arena(static_cast<uint8_t*>(raw), &std::free);
        std::vector<uint32_t> off(n), len(n);
        std::vector<uint8_t> status(n);
        if (unet_tx_build(ctx_, payloads.data(), payloads.bytes(), payloads.off16(), 0, payloads.len(), desc.data(), n, &iface, arena.get(),
                          need, off.data(), len.data(), status.data()) != UNET_OK)
            throw std::runtime_error(std::string("unet_tx_build: ") + unet_last_error());
        for (uint32_t i = 0; i < n; i++)
            if (status[i] == UNET_TX_OK) out[i].assign(arena.get() + size_t(off[i]) * 16, arena.get() + size_t(off[i]) * 16 + len[i]);
        if (status_out) *status_out = std::move(status);
        return out;
    }

  private:
    unet_ctx* ctx_ = nullptr;
};

// ---------------------------------------------------------------------------------------------
// Several GPUs behind one call.  Frames are independent, so each goes to the GPU chosen by the hash of its
// flow 5-tuple (all frames of a flow meet on one GPU; nothing is exchanged between GPUs).  The calling thread
// only hashes; every GPU has a host thread of its own that packs that GPU's arena, runs the offload call and
// scatters the answers back into the caller's order.  The per-verdict counters of all contexts are summed on
// request (the only values that are ever combined across GPUs).
class ShardedOffload {
  public:
    // One context per entry; a device id may appear more than once (several contexts on one GPU).
    explicit ShardedOffload(const std::vector<int>& devices) {
        if (devices.empty()) throw std::invalid_argument("need at least one device");
        for (int d : devices) gpus_.push_back(std::make_unique<Offload>(d));
        arenas_.resize(gpus_.size());
        for (auto& a : arenas_) a = std::make_unique<FrameArena>();
    }
    size_t shards() const { return gpus_.size(); }
    static uint32_t shard_of(const uint8_t* frame, size_t len, int mode, size_t nshards) {
        return static_cast<uint32_t>(unet_flow_hash_record(frame, len, mode) % nshards);
    }

    struct Report {
        std::vector<unet_result> results;   // input order
        std::vector<uint32_t> shard;        // input order: which GPU judged the frame
        std::vector<size_t> shard_sizes;
    };
    Report verify(const std::vector<std::vector<uint8_t>>& frames, const unet_iface& iface) {
        const size_t n = frames.size();
        Report rep;
        rep.results.resize(n);
        place(n, rep.shard, rep.shard_sizes, [&](size_t i) { return shard_of(frames[i].data(), frames[i].size(), iface.mode, gpus_.size()); });
        on_every_shard([&](size_t s) {
            FrameArena& a = *arenas_[s];
            a.clear();
            for (size_t i : index_[s]) a.push(frames[i].data(), frames[i].size());
            const std::vector<unet_result> part = gpus_[s]->verify(a, iface);
            for (size_t k = 0; k < index_[s].size(); k++) rep.results[index_[s][k]] = part[k];
        });
        return rep;
    }

    // Transmit-side checksum fill (unet_generate) of L3 packets, in place in the caller's buffers; returns the
    // per-packet status in input order.
    std::vector<uint8_t> generate(std::vector<std::vector<uint8_t>>& packets) {
        const size_t n = packets.size();
        std::vector<uint8_t> status(n);
        std::vector<uint32_t> shard;
        std::vector<size_t> sizes;
        place(n, shard, sizes, [&](size_t i) { return shard_of(packets[i].data(), packets[i].size(), UNET_MODE_L3, gpus_.size()); });
        on_every_shard([&](size_t s) {
            FrameArena& a = *arenas_[s];
            a.clear();
            for (size_t i : index_[s]) a.push(packets[i].data(), packets[i].size());
            const std::vector<uint8_t> st = gpus_[s]->generate(a);
            for (size_t k = 0; k < index_[s].size(); k++) {
                const size_t i = index_[s][k];
                status[i] = st[k];
                if (!packets[i].empty()) std::memcpy(packets[i].data(), a.record(k), packets[i].size());
            }
        });
        return status;
    }

    // Whole packets from payloads + descriptors (unet_tx_build), placed by the flow hash of the descriptor's
    // 5-tuple.  Returns one vector per packet in input order (empty for a refused one).
    std::vector<std::vector<uint8_t>> tx_build(const std::vector<std::vector<uint8_t>>& payloads, const std::vector<unet_tx_desc>& desc,
                                               const unet_iface& iface, std::vector<uint8_t>* status_out = nullptr) {
        const size_t n = payloads.size();
        if (desc.size() != n) throw std::invalid_argument("one descriptor per payload");
        std::vector<std::vector<uint8_t>> out(n);
        std::vector<uint8_t> status(n);
        std::vector<uint32_t> shard;
        std::vector<size_t> sizes;
        place(n, shard, sizes, [&](size_t i) {
            const unet_tx_desc& d = desc[i];
            const bool ports = d.proto == 17;
            return static_cast<uint32_t>(unet_flow_hash5(iface.unicast, d.dst, d.proto, ports ? d.l4a : 0, ports ? d.l4b : 0) % gpus_.size());
        });
        on_every_shard([&](size_t s) {
            FrameArena& a = *arenas_[s];
            a.clear();
            std::vector<unet_tx_desc> dd;
            dd.reserve(index_[s].size());
            for (size_t i : index_[s]) {
                a.push(payloads[i].data(), payloads[i].size());
                dd.push_back(desc[i]);
            }
            std::vector<uint8_t> st;
            std::vector<std::vector<uint8_t>> built = gpus_[s]->tx_build(a, dd, iface, &st);
            for (size_t k = 0; k < index_[s].size(); k++) {
                out[index_[s][k]] = std::move(built[k]);
                status[index_[s][k]] = st[k];
            }
        });
        if (status_out) *status_out = std::move(status);
        return out;
    }

    // Per-verdict counts of everything judged so far, summed over the GPUs (unet_counters_sum).
    std::array<uint64_t, UNET_V_COUNT> counters(bool reset = false) {
        std::vector<unet_ctx*> h;
        for (auto& g : gpus_) h.push_back(g->get());
        std::array<uint64_t, UNET_V_COUNT> out{};
        if (unet_counters_sum(h.data(), static_cast<uint32_t>(h.size()), out.data(), reset ? 1 : 0) != UNET_OK)
            throw std::runtime_error(std::string("unet_counters_sum: ") + unet_last_error());
        return out;
    }

  private:
    template <class ShardOf>
    void place(size_t n, std::vector<uint32_t>& shard, std::vector<size_t>& sizes, ShardOf&& shard_of_item) {
        const size_t g = gpus_.size();
        shard.resize(n);
        index_.assign(g, {});
        for (size_t i = 0; i < n; i++) {
            shard[i] = shard_of_item(i);
            index_[shard[i]].push_back(i);
        }
        sizes.assign(g, 0);
        for (size_t s = 0; s < g; s++) sizes[s] = index_[s].size();
    }
    // Runs work(s) for every shard on a thread of its own; the first failure is rethrown with its shard.
    template <class Work>
    void on_every_shard(Work&& work) {
        const size_t g = gpus_.size();
        std::vector<std::string> err(g);
        std::vector<std::thread> th;
        for (size_t s = 0; s < g; s++)
            th.emplace_back([&, s] {
                try {
                    work(s);
                } catch (const std::exception& e) {
                    err[s] = e.what()[0] ? e.what() : "error";
                }
            });
        for (auto& t : th) t.join();
        for (size_t s = 0; s < g; s++)
            if (!err[s].empty()) throw std::runtime_error("shard " + std::to_string(s) + ": " + err[s]);
    }

    std::vector<std::unique_ptr<Offload>> gpus_;
    std::vector<std::unique_ptr<FrameArena>> arenas_;
    std::vector<std::vector<size_t>> index_;  // per shard: the input indices it owns, in input order
};

// ---------------------------------------------------------------------------------------------
// UDP protocol control blocks: src/transport/udp.rs:68-165.
struct UdpPcbQueueEntry {
    Endpoint foreign;
    std::vector<uint8_t> data;
};

class UdpContext {
  public:
    static constexpr size_t UDP_PCB_LENGTH = 16;  // udp.rs:15
    struct Pcb {
        bool open = true;  // PcbState::Open
        Endpoint local;
        std::deque<UdpPcbQueueEntry> queue;
        // udp.rs:84-90
        bool can_be_bound(Ipv4Address address, uint16_t port) const {
            return open && (local.address.v == Ipv4Address::ANY || address.v == Ipv4Address::ANY || local.address == address) &&
                   local.port == port;
        }
    };
    // udp.rs:110-136: first PCB, in slot order, that can be bound
    Pcb* select_pcb(Ipv4Address address, uint16_t port) {
        for (auto& p : pcbs_)
            if (p && p->can_be_bound(address, port)) return &*p;
        return nullptr;
    }
    // udp::bind, udp.rs:139-165: nullopt when the endpoint is taken or the table is full
    std::optional<size_t> bind(const Endpoint& e) {
        if (select_pcb(e.address, e.port)) return std::nullopt;
        for (size_t i = 0; i < pcbs_.size(); i++)
            if (!pcbs_[i]) {
                pcbs_[i] = Pcb{true, e, {}};
                return i;
            }
        return std::nullopt;
    }
    Pcb* slot(size_t i) { return pcbs_[i] ? &*pcbs_[i] : nullptr; }

  private:
    std::array<std::optional<Pcb>, UDP_PCB_LENGTH> pcbs_{};
};

// Routing table: src/protocols/ipv4.rs:231-280.  Longest netmask wins among the routes whose network holds the
// destination; a default route carries the gateway as next hop.
struct IpRoute {
    Ipv4Address network, netmask;
    std::shared_ptr<Ipv4Interface> interface;
    std::optional<Ipv4Address> next_hop;
};
class Ipv4Router {
  public:
    void register_route(Ipv4Address network, std::shared_ptr<Ipv4Interface> interface) {  // ipv4.rs:243-250
        const Ipv4Address mask = interface->netmask;
        routes_.push_back(IpRoute{network, mask, std::move(interface), std::nullopt});
    }
    void register_default(std::shared_ptr<Ipv4Interface> interface, Ipv4Address gateway) {  // ipv4.rs:252-259
        routes_.push_front(IpRoute{Ipv4Address{Ipv4Address::ANY}, Ipv4Address{Ipv4Address::ANY}, std::move(interface), gateway});
    }
    const IpRoute* lookup(Ipv4Address dst) const {  // ipv4.rs:261-271
        const IpRoute* candidate = nullptr;
        for (const IpRoute& r : routes_)
            if ((dst.v & r.netmask.v) == r.network.v && (!candidate || r.netmask.v > candidate->netmask.v)) candidate = &r;
        return candidate;
    }

  private:
    std::deque<IpRoute> routes_;
};

struct Ipv4IdGenerator {  // src/protocols/ipv4.rs:282-297
    uint16_t id = 0;
    uint16_t next() { return id++; }
};

// ---------------------------------------------------------------------------------------------
struct Ipv4QueueEntry {  // src/protocols.rs:52-57
    std::vector<uint8_t> data;
    std::shared_ptr<Ipv4Interface> interface;
};

// A finished packet: what ipv4::send hands to the device (L3 bytes), or in L2 mode the frame tap::send
// writes to the TAP file.
struct TxPacket {
    Ipv4Address dst;
    std::vector<uint8_t> bytes;
};

// The arguments of one udp::send (udp.rs:168-173) or icmp::send (icmp.rs:66-74) call.
struct TxRequest {
    uint8_t proto = 17;   // 17: udp::send, 1: icmp::send
    Endpoint src, dst;    // ports are unused for ICMP
    uint8_t icmp_type = 0, icmp_code = 0;
    uint32_t icmp_values = 0;
    std::vector<uint8_t> data;
};

struct TxReport {
    std::vector<TxPacket> packets;    // in request order, refused and unresolved requests left out
    std::vector<std::string> errors;  // the reference's anyhow messages
    size_t unresolved = 0;            // L2 mode: next hop not in the ARP cache (ipv4.rs:352-357 returns Ok)
};

using MacAddress = std::array<uint8_t, 6>;
using ArpResolver = std::function<std::optional<MacAddress>(Ipv4Address)>;  // the Resolved state of the ARP cache

// Collects send requests that passed the host-side checks, then builds them all in one offload call.
class TxBatch {
  public:
    // udp::send's length check (udp.rs:174-180) and ipv4::send's checks (ipv4.rs:320-335); takes the next
    // identification only for a packet that will be built (ipv4.rs:337).  Returns false and reports why
    // when the reference would have bailed out.
    // `next_hop`: the route's gateway when it has one; the ARP cache is asked for that address instead of the
    // destination (ipv4.rs:346-353).
    bool add(const TxRequest& r, const Ipv4Interface& ifc, Ipv4IdGenerator& ids, bool l2, const ArpResolver& arp, TxReport& rep,
             std::optional<Ipv4Address> next_hop = std::nullopt) {
        constexpr size_t kIpv4PayloadMax = 65535 - 20;  // ipv4.rs:18-20
        if (r.proto == 17 && r.data.size() > kIpv4PayloadMax - 8) {
            rep.errors.push_back("udp packet too long, len: " + std::to_string(r.data.size()) + ", max: " + std::to_string(kIpv4PayloadMax - 8));
            return false;
        }
        const Ipv4Address src = r.src.address, dst = r.dst.address;
        if (!(src.v != Ipv4Address::ANY || dst.v != Ipv4Address::BROADCAST)) {
            rep.errors.push_back("source address is required for broadcast packet");
            return false;
        }
        if (!(src.v == Ipv4Address::ANY || src == ifc.unicast)) {
            rep.errors.push_back("unable to send packet with the source address, src: " + src.to_string() + ", interface: " + ifc.unicast.to_string());
            return false;
        }
        const size_t l4len = 8 + r.data.size();
        if (!(l4len < ifc.mtu)) {
            rep.errors.push_back("packet too long, len: " + std::to_string(l4len) + ", mtu: " + std::to_string(ifc.mtu));
            return false;
        }
        unet_tx_desc d{};
        d.src = src.v;
        d.dst = dst.v;
        d.proto = r.proto;
        if (r.proto == 17) {
            d.l4a = r.src.port;
            d.l4b = r.dst.port;
        } else {
            d.l4a = uint16_t((r.icmp_type << 8) | r.icmp_code);
            d.values = r.icmp_values;
        }
        if (l2) {  // ipv4.rs:341-362
            if (dst == ifc.broadcast || dst.v == Ipv4Address::BROADCAST) {
                std::memset(d.dst_mac, 0xff, 6);
            } else {
                const std::optional<MacAddress> mac = arp ? arp(next_hop.value_or(dst)) : std::nullopt;
                d.ip_id = ids.next();  // the reference has built the packet before it consults the ARP cache
                if (!mac) {
                    rep.unresolved++;
                    return false;
                }
                std::memcpy(d.dst_mac, mac->data(), 6);
                payloads_.push(r.data.data(), r.data.size());
                desc_.push_back(d);
                dst_.push_back(dst);
                return true;
            }
        }
        d.ip_id = ids.next();
        payloads_.push(r.data.data(), r.data.size());
        desc_.push_back(d);
        dst_.push_back(dst);
        return true;
    }
    size_t size() const { return desc_.size(); }
    void flush(const Offload& gpu, const Ipv4Interface& ifc, bool l2, TxReport& rep) {
        if (desc_.empty()) return;
        const unet_iface ci = ifc.c_iface(l2);
        std::vector<uint8_t> status;
        std::vector<std::vector<uint8_t>> built = gpu.tx_build(payloads_, desc_, ci, &status);
        for (size_t i = 0; i < built.size(); i++) {
            if (status[i] != UNET_TX_OK) { rep.errors.push_back("packet not built, status " + std::to_string(status[i])); continue; }
            rep.packets.push_back(TxPacket{dst_[i], std::move(built[i])});
        }
        payloads_.clear();
        desc_.clear();
        dst_.clear();
    }

  private:
    FrameArena payloads_;
    std::vector<unet_tx_desc> desc_;
    std::vector<Ipv4Address> dst_;
};

// Batched udp::send / icmp::send over one interface: host checks per request, one build call for all.
inline TxReport send_batch(const Offload& gpu, const Ipv4Interface& ifc, Ipv4IdGenerator& ids, const std::vector<TxRequest>& requests,
                           bool l2 = false, const ArpResolver& arp = nullptr) {
    TxReport rep;
    TxBatch batch;
    for (const TxRequest& r : requests) batch.add(r, ifc, ids, l2, arp, rep);
    batch.flush(gpu, ifc, l2, rep);
    return rep;
}

// ipv4::send's first step as well (ipv4.rs:307-309): the route picks the interface and the next hop.  One build
// call per interface that got packets.
inline TxReport send_batch_routed(const Offload& gpu, const Ipv4Router& router, Ipv4IdGenerator& ids, const std::vector<TxRequest>& requests,
                                  bool l2 = false, const ArpResolver& arp = nullptr) {
    TxReport rep;
    std::vector<std::pair<const Ipv4Interface*, std::unique_ptr<TxBatch>>> batches;
    for (const TxRequest& r : requests) {
        const IpRoute* route = router.lookup(r.dst.address);
        if (!route) {
            rep.errors.push_back("no route found, dst: " + r.dst.address.to_string());
            continue;
        }
        const Ipv4Interface* ifp = route->interface.get();
        TxBatch* b = nullptr;
        for (auto& kv : batches)
            if (kv.first == ifp) b = kv.second.get();
        if (!b) {
            batches.emplace_back(ifp, std::make_unique<TxBatch>());
            b = batches.back().second.get();
        }
        b->add(r, *ifp, ids, l2, arp, rep, route->next_hop);
    }
    for (auto& kv : batches) kv.second->flush(gpu, *kv.first, l2, rep);
    return rep;
}

struct RecvReport {
    std::vector<unet_result> results;  // one per drained entry, in queue order
    std::vector<std::string> errors;   // the reference's anyhow messages for dropped packets
    std::vector<TxPacket> replies;     // echo replies, checksums filled
    size_t udp_enqueued = 0;
};

inline uint16_t be16(const uint8_t* p) { return static_cast<uint16_t>((p[0] << 8) | p[1]); }
inline uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
inline void put16(uint8_t* p, uint16_t v) { p[0] = uint8_t(v >> 8); p[1] = uint8_t(v); }
inline void put32(uint8_t* p, uint32_t v) { p[0] = uint8_t(v >> 24); p[1] = uint8_t(v >> 16); p[2] = uint8_t(v >> 8); p[3] = uint8_t(v); }

// The IPv4 protocol queue with a batched drain.
class NetProtocol {
  public:
    std::deque<Ipv4QueueEntry> queue;  // src/protocols.rs:59-63

    // Batched `NetProtocol::recv` (protocols.rs:83-97): every queued packet is decided by one offload call
    // per interface.  Unlike the reference, a bad packet does not abort the drain (protocols.rs:92-93 leaves
    // the rest queued until the next IRQ); each packet gets its own verdict and, if dropped, its own error.
    RecvReport recv_batch(const Offload& gpu, UdpContext& pcbs, Ipv4IdGenerator& ids, bool ext_tcp = false) {
        RecvReport rep;
        std::vector<Ipv4QueueEntry> entries(std::make_move_iterator(queue.begin()), std::make_move_iterator(queue.end()));
        queue.clear();
        rep.results.resize(entries.size());
        std::vector<char> done(entries.size(), 0);
        for (size_t first = 0; first < entries.size(); first++) {
            if (done[first]) continue;
            // one arena + one launch per distinct interface, queue order kept inside it
            const Ipv4Interface* ifp = entries[first].interface.get();
            arena_.clear();
            std::vector<size_t> idx;
            for (size_t i = first; i < entries.size(); i++)
                if (!done[i] && entries[i].interface.get() == ifp) {
                    arena_.push(entries[i].data.data(), entries[i].data.size());
                    idx.push_back(i);
                    done[i] = 1;
                }
            const unet_iface ci = ifp->c_iface(false, ext_tcp);
            const std::vector<unet_result> res = gpu.verify(arena_, ci);
            TxBatch tx;
            TxReport txr;
            for (size_t k = 0; k < idx.size(); k++) {
                rep.results[idx[k]] = res[k];
                follow_up(entries[idx[k]], res[k], *ifp, pcbs, ids, rep, tx, txr);
            }
            tx.flush(gpu, *ifp, false, txr);  // every echo reply of this interface: one build call
            for (TxPacket& p : txr.packets) rep.replies.push_back(std::move(p));
            for (std::string& e : txr.errors) rep.errors.push_back(std::move(e));
        }
        return rep;
    }

  private:
    FrameArena arena_;

    static void follow_up(const Ipv4QueueEntry& e, const unet_result& r, const Ipv4Interface& ifc, UdpContext& pcbs,
                          Ipv4IdGenerator& ids, RecvReport& rep, TxBatch& tx, TxReport& txr) {
        const uint8_t* d = e.data.data();
        switch (r.verdict) {
            case UNET_V_ACCEPT_UDP: {  // udp.rs:256-277
                const size_t hl = size_t(d[0] & 0x0f) * 4;
                const Ipv4Address dst{be32(d + 16)};
                const uint16_t sport = be16(d + hl), dport = be16(d + hl + 2);
                UdpContext::Pcb* pcb = pcbs.select_pcb(dst, dport);
                if (!pcb) {
                    rep.errors.push_back("udp socket not found, dst: " + dst.to_string() + ", port: " + std::to_string(dport));
                    break;
                }
                // the reference stores the DESTINATION address as the foreign endpoint (udp.rs:271-274)
                pcb->queue.push_back(UdpPcbQueueEntry{Endpoint{dst, sport}, std::vector<uint8_t>(d + hl + 8, d + e.data.size())});
                rep.udp_enqueued++;
                break;
            }
            case UNET_V_ACCEPT_ICMP_ECHO: {  // icmp.rs:110-120: send(EchoReply, code, values, &data[8..], dst, src)
                const size_t hl = size_t(d[0] & 0x0f) * 4;
                const uint8_t* msg = d + hl;
                TxRequest q;
                q.proto = 1;
                q.src.address.v = be32(d + 16);
                q.dst.address.v = be32(d + 12);
                q.icmp_type = 0;
                q.icmp_code = msg[1];
                q.icmp_values = be32(msg + 4);
                q.data.assign(msg + 8, d + e.data.size());
                tx.add(q, ifc, ids, false, nullptr, txr);
                break;
            }
            case UNET_V_ACCEPT_ICMP_ECHOREPLY:
            case UNET_V_IGNORE_L3_NOT_FOR_ME:
            case UNET_V_ACCEPT_TCP:
                break;  // Ok(()) in the reference / nothing more on the host
            default:
                rep.errors.push_back(describe(r));
        }
    }

    // The reference's error text for a dropped packet (anyhow messages at the cited lines).
    static std::string describe(const unet_result& r) {
        char hex[8];
        std::snprintf(hex, sizeof hex, "%04x", r.csum);
        switch (r.verdict) {
            case UNET_V_DROP_L3_PROTO: return "unknown transport protocol number: " + std::to_string(r.proto);  // transport.rs:22-25
            case UNET_V_DROP_L3_VERSION: return "invalid version";                                             // ipv4.rs:129-133
            case UNET_V_DROP_L3_IHL_SHORT: return "ipv4 header too short";                                     // ipv4.rs:134-138
            case UNET_V_DROP_L3_IHL_LONG: return "ipv4 header too long";                                       // ipv4.rs:139-143
            case UNET_V_DROP_L3_FRAGMENT: return "fragmentation is not supported";                             // ipv4.rs:145-147
            case UNET_V_DROP_L3_CSUM: return std::string("invalid checksum: ") + hex;                          // ipv4.rs:155
            case UNET_V_DROP_ICMP_TYPE: return "unknown icmp type";                                            // icmp.rs:20
            case UNET_V_DROP_UDP_SHORT: return "udp packet too short";                                         // udp.rs:227
            case UNET_V_DROP_UDP_LEN: return "invalid udp packet length";                                      // udp.rs:233-237
            case UNET_V_DROP_UDP_CSUM: return std::string("invalid udp checksum: 0x") + hex;                   // udp.rs:249-253
            case UNET_V_DROP_L3_TRUNCATED:
            case UNET_V_DROP_ICMP_TRUNCATED: return "truncated packet (the reference indexes out of bounds here)";
            default: return "dropped, verdict " + std::to_string(r.verdict);
        }
    }
};

}  // namespace unet
