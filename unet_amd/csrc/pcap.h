// Classic pcap (libpcap savefile) import/export for frame arenas.  Host code only.
//
// File: 24-byte global header {magic, version 2.4, thiszone, sigfigs, snaplen, linktype}, then per record a
// 16-byte header {ts_sec, ts_frac, caplen, origlen} and caplen bytes.  The magic tells the byte order of all
// header fields (0xa1b2c3d4 microsecond, 0xa1b23c4d nanosecond timestamps; byte-swapped when written on a
// machine of the other endianness).
#pragma once
#include <cstdint>
#include <cstring>

namespace unetpcap {

constexpr uint32_t kMagicUs = 0xa1b2c3d4u, kMagicNs = 0xa1b23c4du;
constexpr uint32_t kLinkNull = 0, kLinkEthernet = 1, kLinkRaw = 101, kLinkIpv4 = 228;

inline uint32_t rd32(const uint8_t* p, bool swapped) {
    uint32_t v;
    std::memcpy(&v, p, 4);  // host is little-endian (x86-64)
    return swapped ? __builtin_bswap32(v) : v;
}

struct Header {
    bool swapped, nanos;
    uint32_t snaplen, linktype;
    uint32_t skip;  // link-layer bytes in front of what the stack receives (4 for DLT_NULL)
    int mode;       // 1 = records are Ethernet frames, 0 = records are IPv4 packets
};

// 0 on success, else a static message.
inline const char* parse_header(const uint8_t* f, uint64_t bytes, Header* h) {
    if (bytes < 24) return "pcap: file shorter than the 24-byte global header";
    uint32_t magic;
    std::memcpy(&magic, f, 4);
    if (magic == kMagicUs || magic == kMagicNs) h->swapped = false;
    else if (__builtin_bswap32(magic) == kMagicUs || __builtin_bswap32(magic) == kMagicNs) h->swapped = true;
    else return "pcap: bad magic (pcapng and other formats are not supported)";
    const uint32_t m = h->swapped ? __builtin_bswap32(magic) : magic;
    h->nanos = m == kMagicNs;
    h->snaplen = rd32(f + 16, h->swapped);
    h->linktype = rd32(f + 20, h->swapped) & 0x0fffffffu;  // upper bits: FCS length flags
    switch (h->linktype) {
        case kLinkEthernet: h->mode = 1; h->skip = 0; break;
        case kLinkRaw:
        case kLinkIpv4: h->mode = 0; h->skip = 0; break;
        case kLinkNull: h->mode = 0; h->skip = 4; break;
        default: return "pcap: unsupported link type (supported: 0 NULL, 1 EN10MB, 101 RAW, 228 IPV4)";
    }
    return nullptr;
}

// Calls fn(record_index, record_header, data, kept_len, caplen, origlen) for every complete record; kept_len is what
// the device would hand to the stack: the captured bytes after `skip`, cut at max_len.  Returns the number
// of records; *trailing = bytes at the end that do not form a complete record.
template <class F>
inline uint32_t for_each_record(const uint8_t* f, uint64_t bytes, const Header& h, uint32_t max_len, uint64_t* trailing, F&& fn) {
    uint64_t pos = 24;
    uint32_t n = 0;
    while (bytes - pos >= 16) {
        const uint32_t caplen = rd32(f + pos + 8, h.swapped), origlen = rd32(f + pos + 12, h.swapped);
        if ((uint64_t)caplen > bytes - pos - 16) break;
        const uint8_t* data = f + pos + 16;
        uint32_t kept = caplen > h.skip ? caplen - h.skip : 0u;
        if (kept > max_len) kept = max_len;
        fn(n, f + pos, caplen > h.skip ? data + h.skip : data, kept, caplen, origlen);
        pos += 16 + (uint64_t)caplen;
        if (++n == 0xffffffffu) break;
    }
    if (trailing) *trailing = bytes - pos;
    return n;
}

}  // namespace unetpcap
