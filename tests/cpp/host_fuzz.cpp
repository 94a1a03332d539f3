// Host-only entry points that read bytes they did not make (capture files, length-prefixed streams) and the arena
// packers, driven with random and damaged input into buffers of exactly the size the library asked for.  Built with the
// host-side address and undefined-behaviour sanitizers (see tests/test_host_fuzz.py): a read or write one byte past any
// of these heap buffers stops the program.  No GPU call is made.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/unet_offload.h"

static std::mt19937_64 rng(12345);
static uint32_t rnd(uint32_t n) { return (uint32_t)(rng() % n); }

#define REQUIRE(x)                                                                \
    do {                                                                          \
        if (!(x)) {                                                               \
            printf("host_fuzz: %s failed at line %d (%s)\n", #x, __LINE__, unet_last_error()); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

// 16-byte aligned heap block of exactly n bytes (the sanitizer's red zone starts right behind it)
struct Block {
    uint8_t* raw;
    uint8_t* p;
    explicit Block(size_t n) {
        raw = nullptr;
        if (posix_memalign((void**)&raw, 16, n ? n : 1) != 0) exit(2);
        p = raw;
    }
    ~Block() { free(raw); }
};

static std::vector<uint8_t> make_pcap(uint32_t n, bool swapped, bool nanos, uint32_t linktype, uint32_t max_caplen) {
    std::vector<uint8_t> f(24);
    auto put32 = [&](size_t at, uint32_t v) {
        if (swapped) v = __builtin_bswap32(v);
        memcpy(&f[at], &v, 4);
    };
    put32(0, nanos ? 0xa1b23c4du : 0xa1b2c3d4u);
    put32(4, 0x00040002u);
    put32(16, 65535);
    put32(20, linktype);
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t cap = rnd(max_caplen + 1);
        const size_t at = f.size();
        f.resize(at + 16 + cap);
        put32(at, i);
        put32(at + 4, rnd(1000000));
        put32(at + 8, cap);
        put32(at + 12, cap + (rnd(4) == 0 ? rnd(100) : 0));
        for (uint32_t k = 0; k < cap; k++) f[at + 16 + k] = (uint8_t)rng();
    }
    return f;
}

static void pcap_round(int round) {
    static const uint32_t links[] = {0, 1, 101, 228, 7, 0x10000001u};
    const uint32_t n = rnd(40);
    std::vector<uint8_t> f = make_pcap(n, rnd(2), rnd(2), links[rnd(6)], round % 3 == 0 ? 70000 : 2000);
    // damage: cut the file anywhere, flip bytes in it, or blow up a caplen
    const int how = rnd(5);
    if (how == 1 && !f.empty()) f.resize(rnd((uint32_t)f.size() + 1));
    if (how == 2) for (int k = 0; k < 8 && !f.empty(); k++) f[rnd((uint32_t)f.size())] ^= (uint8_t)(1u << rnd(8));
    if (how == 3 && f.size() > 40) { const uint32_t big = 0xfffffff0u; memcpy(&f[24 + 8], &big, 4); }
    Block file(f.size());
    if (!f.empty()) memcpy(file.p, f.data(), f.size());
    const uint32_t max_len = rnd(3) == 0 ? UNET_TAP_FRAME_MAX : (rnd(2) ? UNET_LOOPBACK_MTU : 1 + rnd(3000));
    unet_pcap_info info;
    const int rc = unet_pcap_scan(file.p, f.size(), max_len, &info);
    if (rc != UNET_OK) return;  // refused with a message: fine
    REQUIRE(info.arena_bytes >= 16);
    Block arena(info.arena_bytes);
    std::vector<uint32_t> off(info.n_records + 1), len(info.n_records + 1);
    std::vector<uint64_t> ts(info.n_records + 1);
    // exactly n_records entries are written: guard words behind them
    off[info.n_records] = len[info.n_records] = 0xdeadbeefu;
    ts[info.n_records] = 0xdeadbeefdeadbeefull;
    REQUIRE(unet_pcap_pack(file.p, f.size(), max_len, arena.p, info.arena_bytes, off.data(), len.data(), rnd(2) ? ts.data() : nullptr, info.n_records) == UNET_OK);
    REQUIRE(off[info.n_records] == 0xdeadbeefu && len[info.n_records] == 0xdeadbeefu && ts[info.n_records] == 0xdeadbeefdeadbeefull);
    for (uint32_t i = 0; i < info.n_records; i++) {
        REQUIRE(len[i] <= max_len);
        REQUIRE((uint64_t)off[i] * 16 + len[i] <= info.arena_bytes);
    }
    // a wrong record count or a short arena is refused, not overrun
    if (info.n_records) REQUIRE(unet_pcap_pack(file.p, f.size(), max_len, arena.p, info.arena_bytes, off.data(), len.data(), nullptr, info.n_records - 1) != UNET_OK);
    if (info.arena_bytes > 16) {
        Block small(info.arena_bytes - 16);
        REQUIRE(unet_pcap_pack(file.p, f.size(), max_len, small.p, info.arena_bytes - 16, off.data(), len.data(), nullptr, info.n_records) != UNET_OK);
    }
    // and back out to a capture file of exactly the size asked for
    uint64_t need = 0;
    REQUIRE(unet_pcap_write(arena.p, info.arena_bytes, off.data(), 0, len.data(), info.n_records, info.mode, nullptr, 0, &need) == UNET_OK);
    Block out(need);
    uint64_t wrote = 0;
    REQUIRE(unet_pcap_write(arena.p, info.arena_bytes, off.data(), 0, len.data(), info.n_records, info.mode, out.p, need, &wrote) == UNET_OK);
    REQUIRE(wrote == need);
    if (need > 24) {
        Block shorter(need - 1);
        REQUIRE(unet_pcap_write(arena.p, info.arena_bytes, off.data(), 0, len.data(), info.n_records, info.mode, shorter.p, need - 1, &wrote) != UNET_OK);
    }
    // what was written scans again to the same records
    unet_pcap_info again;
    REQUIRE(unet_pcap_scan(out.p, need, UNET_LOOPBACK_MTU, &again) == UNET_OK);
    uint32_t nonempty = 0;
    for (uint32_t i = 0; i < info.n_records; i++) nonempty += len[i] != 0;
    REQUIRE(again.n_records == nonempty);
}

static void prefixed_round() {
    // a stream of [u32 length][bytes] records, then damaged
    const uint32_t n = rnd(50);
    std::vector<uint8_t> s;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t l = rnd(5) == 0 ? rnd(70000) : rnd(300);
        const size_t at = s.size();
        s.resize(at + 4 + l);
        memcpy(&s[at], &l, 4);
        for (uint32_t k = 0; k < l; k++) s[at + 4 + k] = (uint8_t)rng();
    }
    const int how = rnd(4);
    if (how == 1 && !s.empty()) s.resize(rnd((uint32_t)s.size() + 1));
    if (how == 2) for (int k = 0; k < 4 && !s.empty(); k++) s[rnd((uint32_t)s.size())] ^= (uint8_t)(1u << rnd(8));
    Block stream(s.size());
    if (!s.empty()) memcpy(stream.p, s.data(), s.size());
    uint64_t need = 0;
    uint32_t count = 0;
    if (unet_pack_length_prefixed(stream.p, s.size(), nullptr, &need, nullptr, nullptr, 0, &count) != UNET_OK) return;
    REQUIRE(need >= 16);
    Block arena(need);
    std::vector<uint32_t> off(count + 1), len(count + 1);
    off[count] = len[count] = 0xdeadbeefu;
    uint64_t cap_bytes = need;
    uint32_t got = 0;
    REQUIRE(unet_pack_length_prefixed(stream.p, s.size(), arena.p, &cap_bytes, off.data(), len.data(), count, &got) == UNET_OK);
    REQUIRE(got == count && off[count] == 0xdeadbeefu && len[count] == 0xdeadbeefu);
    for (uint32_t i = 0; i < count; i++) REQUIRE((uint64_t)off[i] * 16 + len[i] <= need);
    if (count) {
        uint64_t cb = need;
        REQUIRE(unet_pack_length_prefixed(stream.p, s.size(), arena.p, &cb, off.data(), len.data(), count - 1, &got) != UNET_OK);
    }
    if (need > 16) {
        Block small(need - 16);
        uint64_t cb = need - 16;
        REQUIRE(unet_pack_length_prefixed(stream.p, s.size(), small.p, &cb, off.data(), len.data(), count, &got) != UNET_OK);
    }
}

static void frames_round() {
    const uint32_t n = 1 + rnd(60);
    std::vector<std::vector<uint8_t>> frames(n);
    std::vector<const uint8_t*> ptr(n);
    std::vector<uint32_t> len(n), off(n + 1), out_off(n + 1);
    for (uint32_t i = 0; i < n; i++) {
        frames[i].resize(rnd(4) == 0 ? rnd(65536) : rnd(200));
        for (auto& b : frames[i]) b = (uint8_t)rng();
        ptr[i] = frames[i].data();
        len[i] = (uint32_t)frames[i].size();
    }
    uint64_t need = 0;
    REQUIRE(unet_pack_size(len.data(), n, &need) == UNET_OK);
    Block arena(need);
    off[n] = 0xdeadbeefu;
    REQUIRE(unet_pack_frames(ptr.data(), len.data(), n, arena.p, need, off.data()) == UNET_OK);
    REQUIRE(off[n] == 0xdeadbeefu);
    for (uint32_t i = 0; i < n; i++) REQUIRE(len[i] == 0 || memcmp(arena.p + (uint64_t)off[i] * 16, frames[i].data(), len[i]) == 0);
    if (need > 16) {
        Block small(need - 16);
        REQUIRE(unet_pack_frames(ptr.data(), len.data(), n, small.p, need - 16, off.data()) != UNET_OK);
    }
    // transmit layouts: offsets for exactly n packets
    for (int mode = 0; mode < 2; mode++) {
        for (uint32_t i = 0; i < n; i++) len[i] = rnd(65508);
        out_off[n] = 0xdeadbeefu;
        uint64_t obytes = 0;
        REQUIRE(unet_tx_layout_aligned(len.data(), n, mode, 1u << rnd(7), out_off.data(), &obytes) == UNET_OK);
        REQUIRE(out_off[n] == 0xdeadbeefu && obytes >= 16);
    }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 3000;
    for (int r = 0; r < rounds; r++) {
        pcap_round(r);
        prefixed_round();
        if (r % 4 == 0) frames_round();
    }
    printf("host_fuzz ok: %d rounds\n", rounds);
    return 0;
}
