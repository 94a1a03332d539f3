// C ABI of the offload (include/unet_offload.h): argument checking, launches, and the
// host-pointer wrappers that stage batches through pinned memory.  No checksum or verdict is
// ever computed on the host here.
#include "../../include/unet_offload.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "kernels.h"
#include "pcap.h"
#include "tcp_track.h"

#include <hipcub/hipcub.hpp>

using namespace unetdev;

#include <cstddef>
// Layouts other languages bind to (rust/unet-offload-sys mirrors them as #[repr(C)] structs).
static_assert(sizeof(unet_result) == 4, "unet_result must stay 4 bytes");
static_assert(offsetof(unet_result, csum) == 0 && offsetof(unet_result, verdict) == 2 && offsetof(unet_result, proto) == 3, "unet_result layout");
static_assert(sizeof(unet_iface) == 16, "unet_iface layout");
static_assert(offsetof(unet_iface, broadcast) == 4 && offsetof(unet_iface, mac) == 8 && offsetof(unet_iface, mode) == 14 && offsetof(unet_iface, ext_tcp) == 15, "unet_iface layout");
static_assert(sizeof(unet_tx_desc) == 28, "unet_tx_desc layout");
static_assert(offsetof(unet_tx_desc, dst) == 4 && offsetof(unet_tx_desc, l4a) == 8 && offsetof(unet_tx_desc, l4b) == 10 && offsetof(unet_tx_desc, values) == 12 &&
              offsetof(unet_tx_desc, ip_id) == 16 && offsetof(unet_tx_desc, proto) == 18 && offsetof(unet_tx_desc, dst_mac) == 20, "unet_tx_desc layout");
static_assert(sizeof(unet_tx_fields) == 4 && offsetof(unet_tx_fields, l4_csum) == 2, "unet_tx_fields layout");
static_assert(sizeof(unet_pcap_info) == 48 && offsetof(unet_pcap_info, nanosecond) == 28 && offsetof(unet_pcap_info, arena_bytes) == 32 &&
              offsetof(unet_pcap_info, trailing_bytes) == 40, "unet_pcap_info layout");
static_assert(sizeof(unet_ticket) == 4, "unet_ticket layout");
static_assert(sizeof(unet_tcp_stats) == 8, "unet_tcp_stats layout");
static_assert(sizeof(unet_tcp_flow) == 32 && offsetof(unet_tcp_flow, next_seq) == 12 && offsetof(unet_tcp_flow, after_close) == 28, "unet_tcp_flow layout");
static_assert((int)UNET_V_COUNT == (int)V_COUNT, "verdict tables out of sync");
static_assert((int)UNET_V_L3_HEADER_OK == (int)V_L3_HEADER_OK, "verdict tables out of sync");
static_assert((int)UNET_V_ACCEPT_UDP == (int)V_ACCEPT_UDP, "verdict tables out of sync");
static_assert((int)UNET_V_ACCEPT_TCP == (int)V_ACCEPT_TCP, "verdict tables out of sync");

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(UNET_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr size_t kStageBytes = 32u << 20;  // two pinned staging buffers of this size

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

// Helper threads for the copy from the caller's (pageable) memory into the pinned staging buffers: one
// core moves about 10-15 GB/s, the PCIe link several times that.
class CopyPool {
  public:
    explicit CopyPool(int helpers) {
        for (int i = 0; i < helpers; i++) th_.emplace_back([this, i] { work(i + 1); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            gen_++;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void copy(void* dst, const void* src, size_t bytes) {
        const size_t parts = th_.size() + 1;
        if (th_.empty() || bytes < (1u << 20)) {
            memcpy(dst, src, bytes);
            return;
        }
        {
            std::lock_guard<std::mutex> g(m_);
            dst_ = (char*)dst;
            src_ = (const char*)src;
            per_ = ((bytes + parts - 1) / parts + 4095) & ~(size_t)4095;
            bytes_ = bytes;
            pending_ = (int)th_.size();
            gen_++;
        }
        cv_.notify_all();
        slice(0);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
    }

  private:
    void slice(size_t i) {
        const size_t lo = i * per_;
        if (lo < bytes_) memcpy(dst_ + lo, src_ + lo, bytes_ - lo < per_ ? bytes_ - lo : per_);
    }
    void work(size_t id) {
        int seen = 0;
        while (true) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
            }
            slice(id);
            {
                std::lock_guard<std::mutex> g(m_);
                pending_--;
            }
            done_.notify_one();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    char* dst_ = nullptr;
    const char* src_ = nullptr;
    size_t per_ = 0, bytes_ = 0;
    int pending_ = 0, gen_ = 0;
    bool stop_ = false;
};

}  // namespace

// ---- device memory of unet_verify_static: held by every GPU the library runs on until the process ends.
// Layout per call: [len table | offset table | arena bytes] from byte 0 (one H2D copy), results at kStaticOutAt.
constexpr u32 kStaticMaxRecords = UNET_STATIC_MAX_RECORDS;
constexpr u64 kStaticMaxArena = UNET_STATIC_MAX_ARENA_BYTES;
constexpr u64 kStaticInBytes = 2ull * kStaticMaxRecords * 4u + kStaticMaxArena;  // both tables + arena
constexpr u64 kStaticOutAt = kStaticInBytes;                                     // multiple of 256
constexpr u64 kStaticBytes = kStaticOutAt + (u64)kStaticMaxRecords * 4u;
static_assert(kStaticInBytes % 256 == 0, "results start on a 256-byte boundary");
static_assert(kStaticInBytes <= kStageBytes && (u64)kStaticMaxRecords * 4u <= kStageBytes, "one pinned staging buffer holds a whole call");
static __device__ __align__(256) uint8_t scratch[kStaticBytes];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
uint8_t* g_scratch_ptr[kMaxDevices];  // per device: each GPU holds its own copy of the array at its own address

// Device address of `scratch` on device `dev` (which must be the current device).  In host code the array's
// name is only a handle for hipGetSymbolAddress, never an address a kernel or a copy may be given.
int static_scratch_of(int dev, uint8_t** out) {
    if (dev < 0 || dev >= kMaxDevices) return fail(UNET_E_NODEV, "device %d: at most %d devices are supported", dev, kMaxDevices);
    std::lock_guard<std::mutex> g(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return fail(UNET_E_HIP, "hipGetSymbolAddress(scratch) on device %d failed: %s", dev, hipGetErrorString(e));
        g_scratch_ptr[dev] = (uint8_t*)p;
    }
    *out = g_scratch_ptr[dev];
    return UNET_OK;
}

// One call at a time may use a device's copy of the array (all contexts of the process on that GPU share it).
std::atomic<bool> g_scratch_busy[kMaxDevices];
struct StaticInUse {
    int dev;
    bool mine;
    explicit StaticInUse(int d) : dev(d), mine(!g_scratch_busy[d].exchange(true, std::memory_order_acquire)) {}
    ~StaticInUse() {
        if (mine) g_scratch_busy[dev].store(false, std::memory_order_release);
    }
};
}  // namespace

struct unet_ctx {
    int device = 0;
    hipStream_t s_comp = nullptr;  // kernels (and D2H of results)
    hipStream_t s_copy = nullptr;  // H2D staging
    hipStream_t s_down = nullptr;  // unet_tx_build on large batches: packets back to the host while the next payloads go up
    hipEvent_t tx_up[2] = {nullptr, nullptr}, tx_built[2] = {nullptr, nullptr}, tx_down[2] = {nullptr, nullptr};
    void* pinned[2] = {nullptr, nullptr};
    void* pinned_dev[2] = {nullptr, nullptr};  // the same buffers as the device addresses kernels may use
    uint64_t host_counts[UNET_V_COUNT] = {};   // verdicts of the batches judged in place (added to d_counts by unet_counters)
    u64 zero_copy_bytes = 4u << 20;            // staging images up to this size may be read in place (UNET_ZERO_COPY_BYTES)
    hipEvent_t pinned_free[2] = {nullptr, nullptr};
    hipEvent_t copied = nullptr;
    DevBuf d_arena, d_off, d_len, d_out, d_aux, d_tx, d_tx_tab, d_tcp, d_len_fill;
    u64 len_fill_n = 0;           // unet_stream_read_records_dev: d_len_fill holds this many copies of len_fill_v
    u32 len_fill_v = 0;
    CopyPool* pool = nullptr;
    int tile_override = 0;        // UNET_TILE, read once at unet_open
    int tx_tile_override = 0;     // UNET_TX_TILE (1, 2, 4, 8, 16, 32, 64), read once at unet_open
    int tile_once = 0;            // records per wave for the launch a host-pointer entry point is about to make
    int tx_wave = 1;              // UNET_TX_WAVE: 0 = every batch goes through the tile kernel, 2 = every batch one packet per wave (A/B, tests)
    u32 tx_wave_block = 256;      // UNET_TXW_BLOCK: threads per workgroup of k_tx_build_wave (64, 128 or 256; more when a packet's group of waves needs it)
    int tx_wave_policy = -1;      // UNET_TXW_POLICY (experiments): cache policy of the two row loads of k_tx_build_wave instead of the choice by mean payload
    u64 tx_pipeline_min = 16u << 20;  // UNET_TX_PIPELINE_MIN: payload or packet bytes from which unet_tx_build works in overlapping runs
    bool tx_pipeline = true;      // UNET_TX_PIPELINE=0: unet_tx_build on large batches uploads everything, builds, downloads everything (A/B, tests)
    bool tx_stage = true;         // UNET_TX_STAGE=0: k_tx_build stores every packet straight into the output arena (A/B)
    int tx_group_log2 = -1;       // UNET_TXW_GROUP (1, 2, 4, 8, 16): waves per packet of k_tx_build_wave instead of the choice by mean payload
    bool hdr_plain = false;       // UNET_HDR_PLAIN, read once at unet_open
    u32 lds_pad = 0;              // UNET_LDS_PAD: bytes of unused LDS added to every workgroup of the record kernels (experiments: fewer resident workgroups)
    u32 mean_len_hint = 0;        // unet_hint_mean_len
    u32 max_len_hint = 0;         // unet_hint_max_len
    bool len_known = false;       // set by the host-pointer entry points around their *_dev call
    u32 mean_len_known = 0, max_len_known = 0;
    uint8_t* scratch_dev = nullptr;  // this device's copy of the static scratch array
    unsigned long long* d_counts = nullptr;  // UNET_V_COUNT per-verdict counters of the host-pointer entry points
    hipEvent_t results_ready = nullptr;
    // unet_verify_async: batches in flight, each with its own stream, pinned image and device image
    struct AsyncSlot {
        hipStream_t s = nullptr;
        hipEvent_t done = nullptr;
        void* pin = nullptr;
        void* pin_dev = nullptr;   // the same buffer as the device sees it
        bool in_place = false;     // this batch is judged on the pinned image itself
        size_t pin_cap = 0;
        DevBuf dev;
        bool busy = false;
        u32 seq = 0, n = 0;
        size_t res_at = 0;
        unet_result* out = nullptr;
    } slot[UNET_ASYNC_DEPTH];
};

namespace {

int ensure(unet_ctx* c, DevBuf& b, size_t bytes) {
    (void)c;
    if (bytes <= b.cap) return UNET_OK;
    if (b.p) HIP_TRY(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = (bytes + (1u << 20) - 1) & ~((size_t)(1u << 20) - 1);
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) return fail(UNET_E_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    b.cap = want;
    return UNET_OK;
}

// Is [p, p + bytes) host memory the copy engines reach directly -- allocated as pinned memory by the caller
// (hipHostMalloc) or registered by it (hipHostRegister)?  Asked for large transfers only: the answer costs a
// microsecond or two.
constexpr size_t kDirectMin = 1u << 20;
inline bool caller_pinned(const void* p, size_t bytes) {
    if (bytes < kDirectMin) return false;
    for (const char* q : {(const char*)p, (const char*)p + bytes - 1}) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, q) != hipSuccess) {
            (void)hipGetLastError();  // (older runtimes answer "invalid value" for ordinary host memory)
            return false;
        }
        if (a.type != hipMemoryTypeHost) return false;
    }
    return true;
}

// Host -> device on the copy stream: through the two pinned buffers, or straight from the caller's memory when that is
// pinned itself (every caller of this waits for the copy stream before it returns).
int h2d(unet_ctx* c, void* dst, const void* src, size_t bytes) {
    if (caller_pinned(src, bytes)) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->s_copy));
        return UNET_OK;
    }
    size_t off = 0;
    int which = 0;
    while (off < bytes) {
        const size_t nb = bytes - off < kStageBytes ? bytes - off : kStageBytes;
        HIP_TRY(hipEventSynchronize(c->pinned_free[which]));
        c->pool->copy(c->pinned[which], (const char*)src + off, nb);
        HIP_TRY(hipMemcpyAsync((char*)dst + off, c->pinned[which], nb, hipMemcpyHostToDevice, c->s_copy));
        HIP_TRY(hipEventRecord(c->pinned_free[which], c->s_copy));
        off += nb;
        which ^= 1;
    }
    return UNET_OK;
}

// Make the compute stream wait for everything queued on the copy stream so far.
int copies_before_compute(unet_ctx* c) {
    HIP_TRY(hipEventRecord(c->copied, c->s_copy));
    HIP_TRY(hipStreamWaitEvent(c->s_comp, c->copied, 0));
    return UNET_OK;
}

int d2h_sync(unet_ctx* c, void* dst, const void* src, size_t bytes) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->s_comp));
    HIP_TRY(hipStreamSynchronize(c->s_comp));
    return UNET_OK;
}

IfaceDev to_dev(const unet_iface* i) {
    IfaceDev d;
    d.unicast = i->unicast;
    d.broadcast = i->broadcast;
    d.mac_lo = (u32)i->mac[0] | ((u32)i->mac[1] << 8) | ((u32)i->mac[2] << 16) | ((u32)i->mac[3] << 24);
    d.mac_hi = (u32)i->mac[4] | ((u32)i->mac[5] << 8);
    d.ext_tcp = i->ext_tcp ? 1u : 0u;
    return d;
}

int check_batch(unet_ctx* ctx, const void* arena, const void* off16, u32 stride16, const void* len, u32 n) {
    if (!ctx) return fail(UNET_E_INVAL, "null context");
    if (n == 0) return UNET_OK;
    if (!arena || !len) return fail(UNET_E_INVAL, "null arena or length table");
    if (((uintptr_t)arena & 15u) != 0) return fail(UNET_E_INVAL, "arena must be 16-byte aligned");
    if (!off16 && stride16 == 0) return fail(UNET_E_INVAL, "need an offset table or a non-zero stride");
    return UNET_OK;
}
// the kernels may touch the arena's first 16 bytes even when every record is empty
inline bool arena_too_small(u64 arena_bytes) { return arena_bytes < 16; }
// Records per wave.  16 down to 1 for long records (about 16 KiB per wave), 32 up to 1 KiB, and 64-record tiles
// for batches of the shortest records only (at most 64 bytes): their kernel stages a tile's byte range in LDS when
// it fits 4 KiB and reads any other tile with plain per-lane loops (correct for every length, slow for long
// ones).  So it is chosen when the longest record is known to be that short, or -- lengths unseen -- when the
// mean is so low that hardly any record can be longer.
constexpr u64 kInPlaceOneWave = 64u << 10;  // what one wave may be left to read across the host link (see unet_verify)
constexpr u64 kTxInPlaceBytes = 1u << 20;    // unet_tx_build: input and output images up to this size stay in pinned host memory
constexpr u64 kShortMax = 64;
inline int tile_for(u64 mean_len, bool max_known, u64 max_len) {
    if (mean_len > 1024) {  // (records of exactly 1 KiB are still read faster as a stream, by 32-record tiles)
        // Long records: about 16 KiB per wave (8, 4, 2 records or 1).  A wave that works through sixteen 9000-byte
        // records holds its slot for 144 KB; shorter waves keep more of them starting and ending all the time
        // (2048 B: 6.57 -> 7.0 TB/s, 4096 B: 6.68 -> 7.08, 9000 B: 6.67 -> 6.9).  Up to 1536 bytes sixteen is best.
        if (mean_len <= 1536) return 16;
        if (mean_len >= 49152) return 8;  // (the very longest run best eight to a wave: 6.98 vs 6.81 singly)
        const u64 r = 16384u / mean_len;
        return r >= 8 ? 8 : (r >= 4 ? 4 : (r >= 2 ? 2 : 1));
    }
    const bool all_short = max_known ? max_len <= kShortMax : mean_len <= kShortMax;
    if (all_short) return 64;
    // Records of 65..96 bytes, the longest known: the same kernel, with as many records per wave as are sure to
    // fit its 256-chunk staging area -- a multiple of eight, so that tiles of full-length records begin on
    // 128-byte lines (48 of five chunks, 40 of six).  Those tiles are latency-bound and everything a wave reads
    // is in flight at once: with 32 such records a wave would keep only 2.5-3 KiB of its 4 KiB on the way.
    if (max_known && max_len > kShortMax && max_len <= 96) return (int)((256u / (u32)((max_len + 15u) / 16u)) & ~7u);
    return 32;
}
// tile_for's answer -> the kernel instantiation that runs it (16, 32 or 64 records per wave at most)
inline int kernel_tile(int tile) { return tile <= 16 ? 16 : (tile <= 32 ? 32 : 64); }
// The device-pointer entry points cannot see the lengths: the mean record size is the caller's hint if one was
// given (unet_hint_mean_len), else the fixed stride (which also bounds the length), else arena_bytes / n (right
// for packed arenas).  The host-pointer entry points pass the true mean and maximum.
inline int pick_tile(const unet_ctx* c, u64 arena_bytes, u32 stride16, u32 n) {
    if (c->tile_once) return c->tile_once;  // set around one launch by a host-pointer entry point
    if (c->tile_override) return c->tile_override;
    if (c->len_known) return tile_for(c->mean_len_known, true, c->max_len_known);
    if (c->mean_len_hint) return tile_for(c->mean_len_hint, c->max_len_hint != 0, c->max_len_hint);
    if (stride16) return tile_for((u64)stride16 * 16u, true, (u64)stride16 * 16u);
    return tile_for(arena_bytes / n, c->max_len_hint != 0, c->max_len_hint);
}

// The mean record length as far as it is known (same sources as pick_tile).
inline u64 pick_mean(const unet_ctx* c, u64 arena_bytes, u32 stride16, u32 n) {
    if (c->len_known) return c->mean_len_known;
    if (c->mean_len_hint) return c->mean_len_hint;
    if (stride16) return (u64)stride16 * 16u;
    return arena_bytes / n;
}
// Packets per wave of k_tx_build for payloads that take a whole wave each (above 1 KiB): fewer than sixteen, for the
// reason given in tile_for, down to one from 8 KB up.  Measured, TB/s at 8 / 4 / 2 / 1 per wave, raw IP output:
// 1472-byte payloads 5.36 / 5.29 / 4.60 / 3.13, 3000 B 4.99 / 5.14 / 5.03 / 4.16, 4000 B 5.23 / 5.60 / 5.57 / 4.90,
// 6000 B 4.84 / 5.04 / 5.29 / 5.17, 8972 B 4.65 / 4.78 / 5.18 / 5.29, 20000 B 4.85 / 5.01 / 5.24 / 5.46, 65507 B
// 4.73 / 4.84 / 4.85 / 5.01; Ethernet output likewise (8972 B: 5.01 / 5.16 / 5.50 / 5.59).  Counts below the smallest
// instantiation of the kernel (4, Ethernet 8) run that instantiation with part of its lanes.
inline int tx_tile_long(u64 mean_payload) {
    if (mean_payload < 2500) return 8;
    if (mean_payload < 5000) return 4;
    return mean_payload < 8000 ? 2 : 1;
}

// Mean payload above which a batch goes to k_tx_build_wave.  Not at 1 KiB, where a packet begins to need a second row:
// up to about 1150 bytes the tile kernel's eight packets per wave are still ahead (1040 B 5.55 against 4.94 TB/s on
// lines, 5.64 against 4.99 as packed Ethernet frames; 1100 B 5.76 / 5.23 and 5.77 / 5.28; packed raw IP even), at 1200
// bytes the wave kernel is (5.66 against 5.25 packed).
constexpr u64 kTxWaveFrom = 1152;

// Packets per wave of k_tx_build by mean payload length.  Up to 128 bytes (one lane or four to a packet) 64, up to 320
// bytes 32, up to 720 bytes sixteen, up to kTxWaveFrom eight (above 1 KiB by way of tx_tile_long).  The steps follow what a wave's LDS
// staging area holds (496-622 chunks of output: 64 packets of up to seven chunks, 32 of up to seventeen, 16 of up to 37,
// 8 of up to 77) and what measured best where a tile is staged in part or not at all: 100 B 64 against 32 per wave
// 5.46 / 4.96 TB/s, 128 B 5.66 / 5.50 (Ethernet: the same either way), 200 B 5.16 / 5.35; 280 B 5.30 with 32 (not staged) against 5.22 with 16 (staged), 400 B
// 5.23 against 5.73, 700 B 5.62 with 16 against 5.63 with 8, 800 B 5.61 against 5.74, 900 B 5.56 against 5.75.
inline int tx_tile_for(u64 mean_payload) {
    if (mean_payload > 1024) return tx_tile_long(mean_payload);
    if (mean_payload > 720) return 8;
    if (mean_payload > 320) return 16;
    return mean_payload <= 128 ? 64 : 32;
}

// Waves per packet of k_tx_build_wave (as a power of two) by mean payload: about three rows of 64 output chunks per wave.
inline u32 tx_group_log2_for(u64 mean_payload, bool l2) {
    const u64 rows = (mean_payload + (l2 ? 42u : 28u) + 1023u) / 1024u;
    return rows <= 3 ? 0u : (rows <= 6 ? 1u : (rows <= 12 ? 2u : (rows <= 24 ? 3u : 4u)));
}

inline hipStream_t pick(unet_ctx* c, void* stream) { return stream ? (hipStream_t)stream : c->s_comp; }
inline u32 tiles_grid(u32 n, u32 tile) {
    const u32 tiles = (u32)(((u64)n + tile - 1) / tile);
    return (tiles + (kBlock / kWave) - 1) / (kBlock / kWave);
}
// A launch may hold fewer than 2^32 threads: with `per_wave` records to a wave that bounds the batch.  (Only
// reachable with one or two records per wave -- batches of jumbo records, or the UNET_TILE override.)
inline int check_tiles(u32 n, u32 per_wave) {
    if ((u64)tiles_grid(n, per_wave) * kBlock >= (1ull << 32))
        return fail(UNET_E_RANGE, "%u records at %u per wavefront need 2^32 threads or more in one launch: split the batch (at most %llu records per call at this size)",
                    n, per_wave, (unsigned long long)per_wave * ((1ull << 26) - 4u));
    return UNET_OK;
}

// Shared body of the host-pointer batch calls: upload arena + tables, leave device views in `b`.
// Every record (plus its pad to 16 bytes) must lie inside the caller's arena.  Also: mean and longest length, and
// whether the records lie in the arena in index order (each one starting at or behind its predecessor).
int check_records(uint64_t arena_bytes, const uint32_t* off16, uint32_t stride16, const uint32_t* len, uint32_t n, u32* mean_len, u32* max_len,
                  bool* ascending) {
    u64 sum_len = 0, prev = 0;
    u32 longest = 0;
    bool asc = true;
    for (u32 i = 0; i < n; i++) {
        if (len[i] > kMaxLen) return fail(UNET_E_RANGE, "record %u: length %u > 65535", i, len[i]);
        const u64 slot = off16 ? off16[i] : (u64)i * stride16;
        if (slot > 0xffffffffull) return fail(UNET_E_RANGE, "record %u starts past 2^32 16-byte slots", i);
        const u64 end = slot * 16u + (((u64)len[i] + 15u) & ~15ull);
        if (end > arena_bytes) return fail(UNET_E_RANGE, "record %u ends at byte %llu, arena is %llu bytes (records are read in 16-byte chunks)", i, (unsigned long long)end, (unsigned long long)arena_bytes);
        sum_len += len[i];
        if (len[i] > longest) longest = len[i];
        asc = asc && slot >= prev;
        prev = slot;
    }
    if (mean_len) *mean_len = (u32)(sum_len / n);
    if (max_len) *max_len = longest;
    if (ascending) *ascending = asc;
    return UNET_OK;
}

int upload_batch(unet_ctx* c, const uint8_t* arena, uint64_t arena_bytes, const uint32_t* off16, uint32_t stride16,
                 const uint32_t* len, uint32_t n, Batch* b, u32* mean_len, u32* max_len) {
    int rc;
    if ((rc = check_records(arena_bytes, off16, stride16, len, n, mean_len, max_len, nullptr))) return rc;
    if ((rc = ensure(c, c->d_arena, arena_bytes))) return rc;
    if ((rc = ensure(c, c->d_len, (size_t)n * 4))) return rc;
    if (off16 && (rc = ensure(c, c->d_off, (size_t)n * 4))) return rc;
    if ((rc = h2d(c, c->d_len.p, len, (size_t)n * 4))) return rc;
    if (off16 && (rc = h2d(c, c->d_off.p, off16, (size_t)n * 4))) return rc;
    if ((rc = h2d(c, c->d_arena.p, arena, arena_bytes))) return rc;
    if ((rc = copies_before_compute(c))) return rc;
    b->arena = (const uint8_t*)c->d_arena.p;
    b->arena_bytes = arena_bytes;
    b->off16 = off16 ? (const u32*)c->d_off.p : nullptr;
    b->stride16 = stride16;
    b->len = (const u32*)c->d_len.p;
    b->n = n;
    return UNET_OK;
}

// Is a staging image better read by the kernel where it lies (pinned host memory, across the host link) than copied
// to HBM first?  Yes for a drained queue of a few to a few thousand frames, where the two copy commands' fixed cost
// is most of the call; no for a few jumbo frames: one wave pulling 100 KiB across the link row by row is slower than
// the copy engine.
inline bool read_in_place(const unet_ctx* c, u64 image_bytes, u32 n, int records_per_wave) {
    const u64 waves = (n + (u32)records_per_wave - 1u) / (u32)records_per_wave;
    return image_bytes <= c->zero_copy_bytes && (image_bytes <= kInPlaceOneWave || image_bytes <= waves * kInPlaceOneWave || waves >= 8u);
}
// Long records read in place: four to a wave instead of sixteen, so that more waves pull on the link at once.
inline int in_place_tile(int tile) { return tile <= 16 ? (tile < 4 ? tile : 4) : tile; }
// The verdicts of a batch judged in place are counted here rather than by a launch of their own.
inline void count_on_host(unet_ctx* c, const u32* res, u32 n) {
    for (u32 i = 0; i < n; i++) {
        const u32 v = (res[i] >> 16) & 0xffu;
        c->host_counts[v < UNET_V_COUNT ? v : 0u]++;  // as k_count_verdicts does
    }
}

int launch_verify(hipStream_t s, const Batch& b_in, const unet_iface* iface, int records_per_wave, u32* d_out, u32 lds_pad = 0) {
    const IfaceDev ifc = to_dev(iface);
    Batch b = b_in;
    b.tile_n = (u32)records_per_wave;
    const u32 n = b.n;
    if (int rc = check_tiles(n, (u32)records_per_wave)) return rc;
    const int tile = kernel_tile(records_per_wave);
#define UNET_LAUNCH_VERIFY(M, T) \
    hipLaunchKernelGGL((k_verify<M, T>), dim3(tiles_grid(n, (u32)records_per_wave)), dim3(kBlock), lds_pad, s, b, ifc, d_out)
    if (iface->mode == UNET_MODE_L2) {
        if (tile == 16) UNET_LAUNCH_VERIFY(MODE_L2, 16);
        else if (tile == 32) UNET_LAUNCH_VERIFY(MODE_L2, 32);
        else UNET_LAUNCH_VERIFY(MODE_L2, 64);
    } else {
        if (tile == 16) UNET_LAUNCH_VERIFY(MODE_L3, 16);
        else if (tile == 32) UNET_LAUNCH_VERIFY(MODE_L3, 32);
        else UNET_LAUNCH_VERIFY(MODE_L3, 64);
    }
#undef UNET_LAUNCH_VERIFY
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

// The host-pointer entry points have seen the lengths; that steers the tile choice of the *_dev call they make.
struct LenScope {
    unet_ctx* c;
    LenScope(unet_ctx* ctx, u32 mean, u32 max) : c(ctx) {
        c->len_known = true;
        c->mean_len_known = mean;
        c->max_len_known = max;
    }
    ~LenScope() { c->len_known = false; }
};

// One batch as a single staging image [len | off16 | arena bytes that hold records]: one host-to-device copy.
struct Image {
    u64 at_off, at_arena, used, total, sum_len;
    u64 src_bytes;  // the caller's arena_bytes: `used` may exceed it by the 16-byte floor when every record is empty
    u32 max_len;
};

int plan_image(uint64_t arena_bytes, const uint32_t* off16, uint32_t stride16, const uint32_t* len, uint32_t n, Image* im) {
    u64 used = 16, sum_len = 0;
    u32 longest = 0;
    for (u32 i = 0; i < n; i++) {
        if (len[i] > longest) longest = len[i];
        if (len[i] > kMaxLen) return fail(UNET_E_RANGE, "record %u: length %u > 65535", i, len[i]);
        const u64 slot = off16 ? off16[i] : (u64)i * stride16;
        if (slot > 0xffffffffull) return fail(UNET_E_RANGE, "record %u starts past 2^32 16-byte slots", i);
        const u64 end = slot * 16u + (((u64)len[i] + 15u) & ~15ull);
        if (end > arena_bytes) return fail(UNET_E_RANGE, "record %u ends at byte %llu, arena is %llu bytes (records are read in 16-byte chunks)", i, (unsigned long long)end, (unsigned long long)arena_bytes);
        if (end > used) used = end;
        sum_len += len[i];
    }
    const u64 tab = ((u64)n * 4u + 15u) & ~15ull;
    im->at_off = tab;
    im->at_arena = off16 ? 2u * tab : tab;
    im->used = used;
    im->total = im->at_arena + used;
    im->sum_len = sum_len;
    im->src_bytes = arena_bytes;
    im->max_len = longest;
    return UNET_OK;
}

void fill_image(unet_ctx* c, uint8_t* pin, const Image& im, const uint8_t* arena, const uint32_t* off16, const uint32_t* len, uint32_t n) {
    memcpy(pin, len, (size_t)n * 4u);
    if (off16) memcpy(pin + im.at_off, off16, (size_t)n * 4u);
    // `used` is at least 16 (the kernels may touch the arena's first chunk even when every record is empty), the
    // caller's arena may be shorter than that: copy what exists, zero the rest
    const u64 have = im.used < im.src_bytes ? im.used : im.src_bytes;
    if (have) c->pool->copy(pin + im.at_arena, arena, have);
    if (have < im.used) memset(pin + im.at_arena + have, 0, im.used - have);
}

Batch image_batch(const uint8_t* dev, const Image& im, bool has_off, uint32_t stride16, uint32_t n) {
    return Batch{dev + im.at_arena, im.used, has_off ? (const u32*)(dev + im.at_off) : nullptr, stride16, (const u32*)dev, n};
}

// Adds the verdicts of n results to the context's own counters (device-resident, read by unet_counters).
int count_into_ctx(unet_ctx* c, hipStream_t s, const u32* d_results, u32 n) {
    if (n == 0) return UNET_OK;
    u32 grid = (n + kBlock - 1) / kBlock;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_count_verdicts, dim3(grid), dim3(kBlock), 0, s, d_results, n, c->d_counts);
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

}  // namespace

extern "C" {

const char* unet_last_error(void) { return g_err; }

int unet_hint_mean_len(unet_ctx* c, uint32_t mean_len) {
    if (!c) return fail(UNET_E_INVAL, "null context");
    c->mean_len_hint = mean_len;
    return UNET_OK;
}

int unet_hint_max_len(unet_ctx* c, uint32_t max_len) {
    if (!c) return fail(UNET_E_INVAL, "null context");
    c->max_len_hint = max_len;
    return UNET_OK;
}

int unet_records_per_wave(uint32_t mean_len, uint32_t max_len) { return tile_for(mean_len, max_len != 0, max_len); }
int unet_tx_packets_per_wave(uint32_t mean_payload_len) { return mean_payload_len > kTxWaveFrom ? 1 : tx_tile_for(mean_payload_len); }
int unet_tx_waves_per_packet(uint32_t mean_payload_len, int mode) {
    return mean_payload_len > kTxWaveFrom ? 1 << tx_group_log2_for(mean_payload_len, mode == UNET_MODE_L2) : 1;
}

int unet_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int unet_open(int device_id, unet_ctx** out_ctx) {
    if (!out_ctx) return fail(UNET_E_INVAL, "null out_ctx");
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(UNET_E_NODEV, "no HIP device visible");
    if (device_id < 0 || device_id >= n) return fail(UNET_E_NODEV, "device %d out of range (0..%d)", device_id, n - 1);
    HIP_TRY(hipSetDevice(device_id));
    unet_ctx* c = new (std::nothrow) unet_ctx();
    if (!c) return fail(UNET_E_NOMEM, "out of host memory");
    c->device = device_id;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->s_comp, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->s_copy, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->s_down, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipEventCreateWithFlags(&c->tx_up[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->tx_built[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->tx_down[i], hipEventDisableTiming);
    }
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipHostMalloc(&c->pinned[i], kStageBytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostGetDevicePointer(&c->pinned_dev[i], c->pinned[i], 0);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->pinned_free[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->copied, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->results_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_counts, UNET_V_COUNT * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(c->d_counts, 0, UNET_V_COUNT * sizeof(unsigned long long));
    for (int i = 0; i < UNET_ASYNC_DEPTH && e == hipSuccess; i++) {
        e = hipStreamCreateWithFlags(&c->slot[i].s, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->slot[i].done, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        fail(UNET_E_HIP, "context setup failed: %s", hipGetErrorString(e));
        unet_close(c);
        return UNET_E_HIP;
    }
    // staging-copy helpers: UNET_COPY_THREADS counts the calling thread too (default 4, at most 16)
    int threads = 4;
    if (const char* t = getenv("UNET_COPY_THREADS")) threads = atoi(t);
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && threads > hw) threads = hw;
    threads = threads < 1 ? 1 : (threads > 16 ? 16 : threads);
    // a thread that cannot be started (process or thread limits) is not fatal: copy with fewer helpers
    for (; threads >= 1 && !c->pool; threads = threads > 1 ? 1 : 0) {
        try {
            c->pool = new CopyPool(threads - 1);
        } catch (...) {
            c->pool = nullptr;
        }
    }
    if (!c->pool) {
        unet_close(c);
        return fail(UNET_E_NOMEM, "cannot create the staging-copy pool");
    }
    if (const char* t = getenv("UNET_TILE")) {
        const int v = atoi(t);
        c->tile_override = v < 1 ? 1 : (v > 64 ? 64 : v);  // records per wave, any value from 1 to 64
    }
    c->hdr_plain = getenv("UNET_HDR_PLAIN") != nullptr;
    if (const char* t = getenv("UNET_LDS_PAD")) c->lds_pad = (u32)strtoul(t, nullptr, 10);
    if (const char* z = getenv("UNET_ZERO_COPY_BYTES")) c->zero_copy_bytes = strtoull(z, nullptr, 10);
    if (const char* t = getenv("UNET_TX_WAVE")) c->tx_wave = atoi(t);
    if (const char* t = getenv("UNET_TXW_BLOCK")) {
        const int v = atoi(t);
        c->tx_wave_block = v <= 64 ? 64u : (v <= 128 ? 128u : 256u);
    }
    if (const char* t = getenv("UNET_TX_STAGE")) c->tx_stage = atoi(t) != 0;
    if (const char* t = getenv("UNET_TX_PIPELINE")) c->tx_pipeline = atoi(t) != 0;
    if (const char* t = getenv("UNET_TX_PIPELINE_MIN")) c->tx_pipeline_min = strtoull(t, nullptr, 10);
    if (const char* t = getenv("UNET_TXW_POLICY")) c->tx_wave_policy = atoi(t);
    if (const char* t = getenv("UNET_TXW_GROUP")) {
        const int v = atoi(t);
        c->tx_group_log2 = v <= 1 ? 0 : (v <= 2 ? 1 : (v <= 4 ? 2 : (v <= 8 ? 3 : 4)));
    }
    if (const char* t = getenv("UNET_TX_TILE")) {
        const int v = atoi(t);
        c->tx_tile_override = v <= 1 ? 1 : (v <= 2 ? 2 : (v <= 4 ? 4 : (v <= 8 ? 8 : (v <= 16 ? 16 : (v <= 32 ? 32 : 64)))));
    }
    int rc = static_scratch_of(device_id, &c->scratch_dev);
    if (rc) {
        unet_close(c);
        return rc;
    }
    *out_ctx = c;
    return UNET_OK;
}

void unet_close(unet_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->s_comp) (void)hipStreamSynchronize(c->s_comp);
    if (c->s_copy) (void)hipStreamSynchronize(c->s_copy);
    if (c->s_down) (void)hipStreamSynchronize(c->s_down);
    DevBuf* bufs[] = {&c->d_arena, &c->d_off, &c->d_len, &c->d_out, &c->d_aux, &c->d_tx, &c->d_tx_tab, &c->d_tcp, &c->d_len_fill};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (int i = 0; i < 2; i++) {
        if (c->pinned[i]) (void)hipHostFree(c->pinned[i]);
        if (c->pinned_free[i]) (void)hipEventDestroy(c->pinned_free[i]);
        if (c->tx_up[i]) (void)hipEventDestroy(c->tx_up[i]);
        if (c->tx_built[i]) (void)hipEventDestroy(c->tx_built[i]);
        if (c->tx_down[i]) (void)hipEventDestroy(c->tx_down[i]);
    }
    delete c->pool;
    for (auto& sl : c->slot) {
        if (sl.s) (void)hipStreamSynchronize(sl.s);
        if (sl.pin) (void)hipHostFree(sl.pin);
        if (sl.dev.p) (void)hipFree(sl.dev.p);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.s) (void)hipStreamDestroy(sl.s);
    }
    if (c->d_counts) (void)hipFree(c->d_counts);
    if (c->results_ready) (void)hipEventDestroy(c->results_ready);
    if (c->copied) (void)hipEventDestroy(c->copied);
    if (c->s_comp) (void)hipStreamDestroy(c->s_comp);
    if (c->s_copy) (void)hipStreamDestroy(c->s_copy);
    if (c->s_down) (void)hipStreamDestroy(c->s_down);
    delete c;
}

int unet_sync(unet_ctx* c) {
    if (!c) return fail(UNET_E_INVAL, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->s_copy));
    HIP_TRY(hipStreamSynchronize(c->s_comp));
    HIP_TRY(hipStreamSynchronize(c->s_down));
    for (auto& sl : c->slot) HIP_TRY(hipStreamSynchronize(sl.s));
    return UNET_OK;
}

/* ---------------------------------------------------------------- a1 */

int unet_csum_batch_dev(unet_ctx* ctx, void* stream, const uint8_t* d_arena, uint64_t arena_bytes,
                        const uint32_t* d_off16, uint32_t stride16, const uint32_t* d_len,
                        const uint16_t* d_seeds, uint32_t n, uint16_t* d_out) {
    int rc = check_batch(ctx, d_arena, d_off16, stride16, d_len, n);
    if (rc) return rc;
    if (n == 0) return UNET_OK;
    if (!d_out) return fail(UNET_E_INVAL, "null output");
    HIP_TRY(hipSetDevice(ctx->device));
    Batch b{d_arena, arena_bytes, d_off16, stride16, d_len, n};
    if (arena_too_small(arena_bytes)) return fail(UNET_E_INVAL, "arena must hold at least 16 bytes");
    b.tile_n = (u32)pick_tile(ctx, arena_bytes, stride16, n);
    if ((rc = check_tiles(n, b.tile_n))) return rc;
    const dim3 grid(tiles_grid(n, b.tile_n));
    switch (kernel_tile((int)b.tile_n)) {
        case 16: hipLaunchKernelGGL(k_csum<16>, grid, dim3(kBlock), 0, pick(ctx, stream), b, d_seeds, d_out); break;
        case 32: hipLaunchKernelGGL(k_csum<32>, grid, dim3(kBlock), 0, pick(ctx, stream), b, d_seeds, d_out); break;
        default: hipLaunchKernelGGL(k_csum<64>, grid, dim3(kBlock), 0, pick(ctx, stream), b, d_seeds, d_out); break;
    }
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

int unet_csum_batch(unet_ctx* c, const uint8_t* arena, uint64_t arena_bytes, const uint32_t* off16,
                    uint32_t stride16, const uint32_t* len, const uint16_t* seeds, uint32_t n, uint16_t* out) {
    if (!c) return fail(UNET_E_INVAL, "null context");
    if (n == 0) return UNET_OK;
    if (!arena || !len || !out) return fail(UNET_E_INVAL, "null argument");
    if (!off16 && stride16 == 0) return fail(UNET_E_INVAL, "need an offset table or a non-zero stride");
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if (arena_bytes + 3ull * n * 4u + 64u <= kStageBytes) {
        // a few buffers: summed where the staging image [lengths | offsets | arena | seeds] lies (see unet_verify)
        Image im;
        if ((rc = plan_image(arena_bytes, off16, stride16, len, n, &im))) return rc;
        const int tile = in_place_tile(c->tile_override ? c->tile_override : tile_for(im.sum_len / n, true, im.max_len));
        const u64 seeds_at = (im.total + 15u) & ~15ull, total = seeds_at + (seeds ? (u64)n * 2u : 0u);
        if (read_in_place(c, total, n, tile)) {
            HIP_TRY(hipEventSynchronize(c->pinned_free[0]));
            HIP_TRY(hipEventSynchronize(c->pinned_free[1]));
            fill_image(c, (uint8_t*)c->pinned[0], im, arena, off16, len, n);
            if (seeds) memcpy((uint8_t*)c->pinned[0] + seeds_at, seeds, (size_t)n * 2u);
            const uint8_t* di = (const uint8_t*)c->pinned_dev[0];
            c->tile_once = tile;
            rc = unet_csum_batch_dev(c, nullptr, di + im.at_arena, im.used, off16 ? (const uint32_t*)(di + im.at_off) : nullptr, stride16,
                                     (const uint32_t*)di, seeds ? (const uint16_t*)(di + seeds_at) : nullptr, n, (uint16_t*)c->pinned_dev[1]);
            c->tile_once = 0;
            if (rc) return rc;
            HIP_TRY(hipStreamSynchronize(c->s_comp));
            memcpy(out, c->pinned[1], (size_t)n * 2u);
            return UNET_OK;
        }
    }
    Batch b;
    u32 mean = 0, maxlen = 0;
    rc = upload_batch(c, arena, arena_bytes, off16, stride16, len, n, &b, &mean, &maxlen);
    LenScope hint(c, mean, maxlen);
    if (rc) return rc;
    if ((rc = ensure(c, c->d_out, (size_t)n * 2))) return rc;
    const uint16_t* d_seeds = nullptr;
    if (seeds) {
        if ((rc = ensure(c, c->d_aux, (size_t)n * 2))) return rc;
        if ((rc = h2d(c, c->d_aux.p, seeds, (size_t)n * 2))) return rc;
        if ((rc = copies_before_compute(c))) return rc;
        d_seeds = (const uint16_t*)c->d_aux.p;
    }
    rc = unet_csum_batch_dev(c, nullptr, b.arena, b.arena_bytes, b.off16, b.stride16, b.len, d_seeds, n,
                             (uint16_t*)c->d_out.p);
    if (rc) return rc;
    return d2h_sync(c, out, c->d_out.p, (size_t)n * 2);
}

int unet_csum(unet_ctx* c, const uint8_t* data, size_t len, uint16_t seed, uint16_t* out) {
    if (!c || !out || (!data && len)) return fail(UNET_E_INVAL, "null argument");
    if (len > kMaxLen) return fail(UNET_E_RANGE, "length %zu > 65535", len);
    HIP_TRY(hipSetDevice(c->device));
    // one staging image [len u32 | seed u16 | pad to 16][record padded to whole chunks]: one copy in, one launch,
    // one copy out, all on the compute stream
    const size_t padded = (len + 15) & ~(size_t)15;
    const size_t total = 16 + (padded ? padded : 16);
    int rc;
    if ((rc = ensure(c, c->d_arena, total + 256))) return rc;
    HIP_TRY(hipEventSynchronize(c->pinned_free[0]));
    HIP_TRY(hipEventSynchronize(c->pinned_free[1]));
    uint8_t* pin = (uint8_t*)c->pinned[0];
    memset(pin, 0, total);
    const u32 l32 = (u32)len;
    memcpy(pin, &l32, 4);
    memcpy(pin + 4, &seed, 2);
    if (len) memcpy(pin + 16, data, len);
    if (total <= kInPlaceOneWave && total <= c->zero_copy_bytes) {
        // a buffer one wave reads in a few rows: the kernel takes it out of the pinned buffer itself (see unet_verify)
        const uint8_t* hd = (const uint8_t*)c->pinned_dev[0];
        LenScope hint(c, (u32)len, (u32)len);
        if ((rc = unet_csum_batch_dev(c, nullptr, hd + 16, total - 16, nullptr, 1, (const u32*)hd, (const uint16_t*)(hd + 4), 1,
                                      (uint16_t*)c->pinned_dev[1])))
            return rc;
        HIP_TRY(hipStreamSynchronize(c->s_comp));
        memcpy(out, c->pinned[1], 2);
        return UNET_OK;
    }
    uint8_t* dev = (uint8_t*)c->d_arena.p;
    HIP_TRY(hipMemcpyAsync(dev, pin, total, hipMemcpyHostToDevice, c->s_comp));
    uint16_t* d_out = (uint16_t*)(dev + ((total + 255) & ~(size_t)255));
    {
        LenScope hint(c, (u32)len, (u32)len);
        rc = unet_csum_batch_dev(c, nullptr, dev + 16, total - 16, nullptr, 1, (const u32*)dev, (const uint16_t*)(dev + 4), 1, d_out);
    }
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->pinned[1], d_out, 2, hipMemcpyDeviceToHost, c->s_comp));
    HIP_TRY(hipStreamSynchronize(c->s_comp));
    memcpy(out, c->pinned[1], 2);
    return UNET_OK;
}

/* ---------------------------------------------------------------- a2-a4 */

int unet_ipv4_validate_dev(unet_ctx* ctx, void* stream, const uint8_t* d_hdrs, uint32_t n, const unet_iface* iface,
                           unet_result* d_out) {
    if (!ctx) return fail(UNET_E_INVAL, "null context");
    if (n == 0) return UNET_OK;
    if (!d_hdrs || !iface || !d_out) return fail(UNET_E_INVAL, "null argument");
    if (((uintptr_t)d_hdrs & 15u) != 0) return fail(UNET_E_INVAL, "header buffer must be 16-byte aligned");
    HIP_TRY(hipSetDevice(ctx->device));
    if (((uintptr_t)d_out & 15u) != 0) return fail(UNET_E_INVAL, "result buffer must be 16-byte aligned");
    const u32 ext = iface->ext_tcp ? 1u : 0u;
    // whole workgroups (256 headers per wave) take the row kernel; the remainder the per-lane kernel
    const u32 per_block = (kHdrBlock / kWave) * kHdrPerWave;
    const u32 full_blocks = ctx->hdr_plain ? 0u : n / per_block;
    const u32 done = full_blocks * per_block;
    if (full_blocks)
        hipLaunchKernelGGL(k_ipv4_validate_rows, dim3(full_blocks), dim3(kHdrBlock), 0, pick(ctx, stream), (const u32x4*)d_hdrs, ext,
                           (u32x4*)d_out);
    if (n > done) {
        const u32 rest = n - done, per = kBlock * kHdrPerGroup;
        hipLaunchKernelGGL(k_ipv4_validate, dim3((rest + per - 1) / per), dim3(kBlock), 0, pick(ctx, stream),
                           d_hdrs + (size_t)done * 20u, rest, ext, (u32*)d_out + done);
    }
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

int unet_ipv4_validate(unet_ctx* c, const uint8_t* hdrs, uint32_t n, const unet_iface* iface, unet_result* out) {
    if (!c) return fail(UNET_E_INVAL, "null context");
    if (n == 0) return UNET_OK;
    if (!hdrs || !iface || !out) return fail(UNET_E_INVAL, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t)n * 20;
    int rc;
    if (bytes + 16 <= c->zero_copy_bytes && bytes + 16 <= kStageBytes) {
        // a few headers: judged where the pinned staging buffer lies (see unet_verify)
        HIP_TRY(hipEventSynchronize(c->pinned_free[0]));
        HIP_TRY(hipEventSynchronize(c->pinned_free[1]));
        memcpy(c->pinned[0], hdrs, bytes);
        memset((uint8_t*)c->pinned[0] + bytes, 0, 16);
        if ((rc = unet_ipv4_validate_dev(c, nullptr, (const uint8_t*)c->pinned_dev[0], n, iface, (unet_result*)c->pinned_dev[1]))) return rc;
        HIP_TRY(hipStreamSynchronize(c->s_comp));
        memcpy(out, c->pinned[1], (size_t)n * 4u);
        count_on_host(c, (const u32*)c->pinned[1], n);
        return UNET_OK;
    }
    if ((rc = ensure(c, c->d_arena, bytes + 16))) return rc;  // device buffer covers the last partial chunk
    if ((rc = ensure(c, c->d_out, (size_t)n * 4))) return rc;
    if ((rc = h2d(c, c->d_arena.p, hdrs, bytes))) return rc;
    if ((rc = copies_before_compute(c))) return rc;
    if ((rc = unet_ipv4_validate_dev(c, nullptr, (const uint8_t*)c->d_arena.p, n, iface, (unet_result*)c->d_out.p))) return rc;
    if ((rc = count_into_ctx(c, c->s_comp, (const u32*)c->d_out.p, n))) return rc;
    return d2h_sync(c, out, c->d_out.p, (size_t)n * 4);
}

/* ---------------------------------------------------------------- receive path */

int unet_verify_dev(unet_ctx* ctx, void* stream, const uint8_t* d_arena, uint64_t arena_bytes,
                    const uint32_t* d_off16, uint32_t stride16, const uint32_t* d_len, uint32_t n,
                    const unet_iface* iface, unet_result* d_out) {
    int rc = check_batch(ctx, d_arena, d_off16, stride16, d_len, n);
    if (rc) return rc;
    if (n == 0) return UNET_OK;
    if (!iface || !d_out) return fail(UNET_E_INVAL, "null iface or output");
    if (arena_too_small(arena_bytes)) return fail(UNET_E_INVAL, "arena must hold at least 16 bytes");
    if (iface->mode != UNET_MODE_L3 && iface->mode != UNET_MODE_L2) return fail(UNET_E_INVAL, "bad iface mode %d", iface->mode);
    HIP_TRY(hipSetDevice(ctx->device));
    Batch b{d_arena, arena_bytes, d_off16, stride16, d_len, n};
    if (arena_too_small(arena_bytes)) return fail(UNET_E_INVAL, "arena must hold at least 16 bytes");
    return launch_verify(pick(ctx, stream), b, iface, pick_tile(ctx, arena_bytes, stride16, n), (u32*)d_out, ctx->lds_pad);
}

namespace {
// Segments a record of max_len bytes is cut into by the split form (at least one).
inline u32 split_segments(u32 max_len) { return max_len == 0 ? 1u : (((max_len + 15u) >> 4) + kSegChunks - 1u) / kSegChunks; }

// A few long records: is the batch better judged one wave per 4 KiB segment than one wave per record?  Yes when the
// usual tiling would leave most of the 256 CUs without a wave and the records are long enough to be worth cutting.
inline bool want_split(u32 n, u64 mean_len, u32 max_len, int records_per_wave) {
    const u64 waves = ((u64)n + (u32)records_per_wave - 1u) / (u32)records_per_wave;
    return max_len >= 4u * kSegChunks * 16u && mean_len >= 2u * kSegChunks * 16u && waves < 128u && (u64)n * split_segments(max_len) <= 65536u;
}

int launch_split(hipStream_t s, const Batch& b, const unet_iface* iface, u32 max_len, u32* d_partial, u32* d_out) {
    const u32 segs = split_segments(max_len), n = b.n;
    const u64 waves = (u64)n * segs;
    if (waves >= (1ull << 26)) return fail(UNET_E_RANGE, "%u records of up to %u bytes make %llu segments: too many for one launch", n, max_len, (unsigned long long)waves);
    hipLaunchKernelGGL(k_segment_sums, dim3((u32)((waves + (kBlock / kWave) - 1) / (kBlock / kWave))), dim3(kBlock), 0, s, b, segs, d_partial);
    HIP_TRY(hipGetLastError());
    const IfaceDev ifc = to_dev(iface);
    const dim3 grid((n + kBlock - 1) / kBlock);
    if (iface->mode == UNET_MODE_L2)
        hipLaunchKernelGGL(k_verify_presummed<MODE_L2>, grid, dim3(kBlock), 0, s, b, ifc, (const u32*)d_partial, segs, d_out);
    else
        hipLaunchKernelGGL(k_verify_presummed<MODE_L3>, grid, dim3(kBlock), 0, s, b, ifc, (const u32*)d_partial, segs, d_out);
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}
}  // namespace

int unet_verify_split_workspace(uint32_t n, uint32_t max_len, uint64_t* bytes) {
    if (!bytes) return fail(UNET_E_INVAL, "null argument");
    if (max_len > kMaxLen) return fail(UNET_E_RANGE, "max_len %u > 65535", max_len);
    *bytes = (uint64_t)(n ? n : 1u) * split_segments(max_len) * 4u;
    return UNET_OK;
}

int unet_verify_split_dev(unet_ctx* ctx, void* stream, const uint8_t* d_arena, uint64_t arena_bytes, const uint32_t* d_off16,
                          uint32_t stride16, const uint32_t* d_len, uint32_t n, uint32_t max_len, const unet_iface* iface,
                          unet_result* d_out, void* d_workspace, uint64_t workspace_bytes) {
    int rc = check_batch(ctx, d_arena, d_off16, stride16, d_len, n);
    if (rc) return rc;
    if (n == 0) return UNET_OK;
    if (!iface || !d_out || !d_workspace) return fail(UNET_E_INVAL, "null argument");
    if (arena_too_small(arena_bytes)) return fail(UNET_E_INVAL, "arena must hold at least 16 bytes");
    if (iface->mode != UNET_MODE_L3 && iface->mode != UNET_MODE_L2) return fail(UNET_E_INVAL, "bad iface mode %d", iface->mode);
    if (max_len > kMaxLen) return fail(UNET_E_RANGE, "max_len %u > 65535", max_len);
    if (((uintptr_t)d_workspace & 3u) != 0) return fail(UNET_E_INVAL, "workspace must be 4-byte aligned");
    uint64_t need = 0;
    if ((rc = unet_verify_split_workspace(n, max_len, &need))) return rc;
    if (workspace_bytes < need)
        return fail(UNET_E_RANGE, "workspace of %llu bytes, %u records of up to %u bytes need %llu (unet_verify_split_workspace)",
                    (unsigned long long)workspace_bytes, n, max_len, (unsigned long long)need);
    HIP_TRY(hipSetDevice(ctx->device));
    Batch b{d_arena, arena_bytes, d_off16, stride16, d_len, n};
    return launch_split(pick(ctx, stream), b, iface, max_len, (u32*)d_workspace, (u32*)d_out);
}

int unet_verify_bytes_dev(unet_ctx* ctx, void* stream, const uint8_t* d_buf, uint64_t buf_bytes, uint32_t stride_bytes,
                          uint32_t len, uint32_t n, const unet_iface* iface, unet_result* d_out) {
    if (!ctx) return fail(UNET_E_INVAL, "null context");
    if (n == 0) return UNET_OK;
    if (!d_buf || !iface || !d_out) return fail(UNET_E_INVAL, "null argument");
    if (((uintptr_t)d_buf & 15u) != 0) return fail(UNET_E_INVAL, "buffer must be 16-byte aligned");
    if (iface->mode != UNET_MODE_L3 && iface->mode != UNET_MODE_L2) return fail(UNET_E_INVAL, "bad iface mode %d", iface->mode);
    if (stride_bytes == 0 || stride_bytes > kBytesMaxLen || len > stride_bytes)
        return fail(UNET_E_RANGE, "byte-stride records: need len <= stride <= %u, got len %u, stride %u (longer records: unet_verify_dev)", kBytesMaxLen, len,
                    stride_bytes);
    // the records are read in 16-byte chunks: the buffer must reach the end of the chunk the last record ends in
    const u64 need = (((u64)(n - 1u) * stride_bytes + len) + 15u) & ~15ull;
    if (need > buf_bytes || buf_bytes < 16)
        return fail(UNET_E_RANGE, "%u records of %u bytes at stride %u end at byte %llu (in whole 16-byte chunks), the buffer has %llu", n, len, stride_bytes,
                    (unsigned long long)need, (unsigned long long)buf_bytes);
    HIP_TRY(hipSetDevice(ctx->device));
    const u32 per = bytes_per_wave(stride_bytes);
    int rc;
    if ((rc = check_tiles(n, per))) return rc;
    const IfaceDev ifc = to_dev(iface);
    if (iface->mode == UNET_MODE_L2)
        hipLaunchKernelGGL(k_verify_bytes<MODE_L2>, dim3(tiles_grid(n, per)), dim3(kBlock), 0, pick(ctx, stream), d_buf, stride_bytes, len, n, ifc, (u32*)d_out);
    else
        hipLaunchKernelGGL(k_verify_bytes<MODE_L3>, dim3(tiles_grid(n, per)), dim3(kBlock), 0, pick(ctx, stream), d_buf, stride_bytes, len, n, ifc, (u32*)d_out);
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

int unet_verify(unet_ctx* c, const uint8_t* arena, uint64_t arena_bytes, const uint32_t* off16, uint32_t stride16,
                const uint32_t* len, uint32_t n, const unet_iface* iface, unet_result* out) {
    if (!c) return fail(UNET_E_INVAL, "null context");
    if (n == 0) return UNET_OK;
    if (!arena || !len || !iface || !out) return fail(UNET_E_INVAL, "null argument");
    if (!off16 && stride16 == 0) return fail(UNET_E_INVAL, "need an offset table or a non-zero stride");
    if (iface->mode != UNET_MODE_L3 && iface->mode != UNET_MODE_L2) return fail(UNET_E_INVAL, "bad iface mode %d", iface->mode);
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if (arena_bytes + 2ull * n * 4u + 64u <= kStageBytes && (u64)n * 4u <= kStageBytes) {
        // small batch (a drained protocol queue): one staging image, one copy each way, everything on one stream
        Image im;
        if ((rc = plan_image(arena_bytes, off16, stride16, len, n, &im))) return rc;
        const u64 res_at = (im.total + 255u) & ~255ull;
        HIP_TRY(hipEventSynchronize(c->pinned_free[0]));
        HIP_TRY(hipEventSynchronize(c->pinned_free[1]));
        fill_image(c, (uint8_t*)c->pinned[0], im, arena, off16, len, n);
        const int tile0 = c->tile_override ? c->tile_override : tile_for(im.sum_len / n, true, im.max_len);
        const int tile_here = in_place_tile(tile0);
        if (!c->tile_override && want_split(n, im.sum_len / n, im.max_len, tile0)) {
            // a few jumbo records: one wave per 4 KiB segment instead of a few waves working through whole records
            const u64 part_bytes = (u64)n * split_segments(im.max_len) * 4u;  // at most 256 KiB (want_split)
            if (im.total <= c->zero_copy_bytes) {
                // read where the pinned image lies, like any small batch: now by hundreds of waves at once; results
                // and the segments' sums go to the other pinned buffer
                const u64 part_at = ((u64)n * 4u + 255u) & ~255ull;
                if ((rc = launch_split(c->s_comp, image_batch((const uint8_t*)c->pinned_dev[0], im, off16 != nullptr, stride16, n), iface, im.max_len,
                                       (u32*)((uint8_t*)c->pinned_dev[1] + part_at), (u32*)c->pinned_dev[1])))
                    return rc;
            } else {
                const u64 part_at = (res_at + (u64)n * 4u + 255u) & ~255ull;
                if ((rc = ensure(c, c->d_arena, part_at + part_bytes))) return rc;
                uint8_t* dev = (uint8_t*)c->d_arena.p;
                HIP_TRY(hipMemcpyAsync(dev, c->pinned[0], im.total, hipMemcpyHostToDevice, c->s_comp));
                u32* d_out = (u32*)(dev + res_at);
                if ((rc = launch_split(c->s_comp, image_batch(dev, im, off16 != nullptr, stride16, n), iface, im.max_len, (u32*)(dev + part_at), d_out))) return rc;
                HIP_TRY(hipMemcpyAsync(c->pinned[1], d_out, (size_t)n * 4u, hipMemcpyDeviceToHost, c->s_comp));
            }
            HIP_TRY(hipStreamSynchronize(c->s_comp));
            memcpy(out, c->pinned[1], (size_t)n * 4u);
            count_on_host(c, (const u32*)c->pinned[1], n);
            return UNET_OK;
        }
        if (read_in_place(c, im.total, n, tile_here)) {
            // the kernel reads the image straight out of the pinned host buffer and writes the results into the
            // other one (both are mapped into the device's address space): one launch, one wait, no copy command
            if ((rc = launch_verify(c->s_comp, image_batch((const uint8_t*)c->pinned_dev[0], im, off16 != nullptr, stride16, n), iface, tile_here,
                                    (u32*)c->pinned_dev[1])))
                return rc;
            HIP_TRY(hipStreamSynchronize(c->s_comp));
            memcpy(out, c->pinned[1], (size_t)n * 4u);
            count_on_host(c, (const u32*)c->pinned[1], n);
            return UNET_OK;
        }
        // (device memory is only needed on this side of the decision)
        if ((rc = ensure(c, c->d_arena, res_at + (u64)n * 4u))) return rc;
        uint8_t* dev = (uint8_t*)c->d_arena.p;
        HIP_TRY(hipMemcpyAsync(dev, c->pinned[0], im.total, hipMemcpyHostToDevice, c->s_comp));
        u32* d_out = (u32*)(dev + res_at);
        const int tile = c->tile_override ? c->tile_override : tile_for(im.sum_len / n, true, im.max_len);
        if ((rc = launch_verify(c->s_comp, image_batch(dev, im, off16 != nullptr, stride16, n), iface, tile, d_out))) return rc;
        // the counting kernel goes ahead of the copy: when the wait below ends nothing reads d_out any more, so the
        // next call (of any context that shares the memory) may overwrite it
        if ((rc = count_into_ctx(c, c->s_comp, d_out, n))) return rc;
        HIP_TRY(hipMemcpyAsync(c->pinned[1], d_out, (size_t)n * 4u, hipMemcpyDeviceToHost, c->s_comp));
        HIP_TRY(hipEventRecord(c->results_ready, c->s_comp));
        HIP_TRY(hipEventSynchronize(c->results_ready));
        memcpy(out, c->pinned[1], (size_t)n * 4u);
        return UNET_OK;
    }
    Batch b;
    u32 mean = 0, maxlen = 0;
    rc = upload_batch(c, arena, arena_bytes, off16, stride16, len, n, &b, &mean, &maxlen);
    LenScope hint(c, mean, maxlen);
    if (rc) return rc;
    if ((rc = ensure(c, c->d_out, (size_t)n * 4))) return rc;
    rc = unet_verify_dev(c, nullptr, b.arena, b.arena_bytes, b.off16, b.stride16, b.len, n, iface, (unet_result*)c->d_out.p);
    if (rc) return rc;
    if ((rc = count_into_ctx(c, c->s_comp, (const u32*)c->d_out.p, n))) return rc;
    return d2h_sync(c, out, c->d_out.p, (size_t)n * 4);
}

int unet_verify_async(unet_ctx* c, const uint8_t* arena, uint64_t arena_bytes, const uint32_t* off16, uint32_t stride16,
                      const uint32_t* len, uint32_t n, const unet_iface* iface, unet_result* out, unet_ticket* ticket) {
    if (!c || !ticket) return fail(UNET_E_INVAL, "null context or ticket");
    ticket->id = UNET_TICKET_NONE;
    if (n == 0) return UNET_OK;
    if (!arena || !len || !iface || !out) return fail(UNET_E_INVAL, "null argument");
    if (!off16 && stride16 == 0) return fail(UNET_E_INVAL, "need an offset table or a non-zero stride");
    if (iface->mode != UNET_MODE_L3 && iface->mode != UNET_MODE_L2) return fail(UNET_E_INVAL, "bad iface mode %d", iface->mode);
    int k = -1;
    for (int i = 0; i < UNET_ASYNC_DEPTH; i++)
        if (!c->slot[i].busy) { k = i; break; }
    if (k < 0) return fail(UNET_E_BUSY, "%d batches are already in flight: unet_wait for one first", UNET_ASYNC_DEPTH);
    unet_ctx::AsyncSlot& sl = c->slot[k];
    HIP_TRY(hipSetDevice(c->device));
    Image im;
    int rc = plan_image(arena_bytes, off16, stride16, len, n, &im);
    if (rc) return rc;
    const u64 res_at = (im.total + 255u) & ~255ull, need = res_at + (u64)n * 4u;
    if (need > sl.pin_cap) {
        if (sl.pin) HIP_TRY(hipHostFree(sl.pin));
        sl.pin = nullptr;
        sl.pin_cap = 0;
        const size_t want = (need + (1u << 20) - 1) & ~((size_t)(1u << 20) - 1);
        hipError_t e = hipHostMalloc(&sl.pin, want, hipHostMallocDefault);
        if (e != hipSuccess) return fail(UNET_E_NOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        sl.pin_cap = want;
        HIP_TRY(hipHostGetDevicePointer(&sl.pin_dev, sl.pin, 0));
    }
    const int tile = c->tile_override ? c->tile_override : tile_for(im.sum_len / n, true, im.max_len);
    const int tile_here = in_place_tile(tile);
    sl.in_place = read_in_place(c, im.total, n, tile_here);
    // the slot's device image grows in steps of two, so that a steady stream of batches soon stops reallocating
    // (a hipFree waits for the whole device, the other slot's batch included)
    if (!sl.in_place && need > sl.dev.cap && (rc = ensure(c, sl.dev, need > 2u * sl.dev.cap ? need : 2u * sl.dev.cap))) return rc;
    // the caller's buffers are free again when this returns: everything the GPU reads sits in the pinned image
    fill_image(c, (uint8_t*)sl.pin, im, arena, off16, len, n);
    // From the first command queued on the slot's stream on, a failure must not hand the slot back while the GPU
    // may still be reading its pinned image: wait for the stream before reporting it.
    auto enqueue = [&]() -> int {
        int rc2;
        if (sl.in_place) {
            // (see unet_verify: the kernel works on the pinned image itself)
            const uint8_t* hd = (const uint8_t*)sl.pin_dev;
            if ((rc2 = launch_verify(sl.s, image_batch(hd, im, off16 != nullptr, stride16, n), iface, tile_here, (u32*)((uint8_t*)sl.pin_dev + res_at)))) return rc2;
            HIP_TRY(hipEventRecord(sl.done, sl.s));
        } else {
            uint8_t* dev = (uint8_t*)sl.dev.p;
            HIP_TRY(hipMemcpyAsync(dev, sl.pin, im.total, hipMemcpyHostToDevice, sl.s));
            u32* d_out = (u32*)(dev + res_at);
            if ((rc2 = launch_verify(sl.s, image_batch(dev, im, off16 != nullptr, stride16, n), iface, tile, d_out))) return rc2;
            if ((rc2 = count_into_ctx(c, sl.s, d_out, n))) return rc2;
            HIP_TRY(hipMemcpyAsync((uint8_t*)sl.pin + res_at, d_out, (size_t)n * 4u, hipMemcpyDeviceToHost, sl.s));
            HIP_TRY(hipEventRecord(sl.done, sl.s));
        }
        return UNET_OK;
    };
    if ((rc = enqueue())) {
        (void)hipStreamSynchronize(sl.s);  // g_err keeps the first failure's text
        return rc;
    }
    sl.busy = true;
    sl.seq = (sl.seq + 1u) & 0xffffffu;
    sl.n = n;
    sl.res_at = res_at;
    sl.out = out;
    ticket->id = (u32)k | (sl.seq << 8);
    return UNET_OK;
}

int unet_wait(unet_ctx* c, unet_ticket ticket) {
    if (!c) return fail(UNET_E_INVAL, "null context");
    if (ticket.id == UNET_TICKET_NONE) return UNET_OK;  // an empty batch
    const u32 k = ticket.id & 0xffu;
    if (k >= UNET_ASYNC_DEPTH || !c->slot[k].busy || c->slot[k].seq != (ticket.id >> 8)) return fail(UNET_E_INVAL, "ticket %u is not in flight", ticket.id);
    unet_ctx::AsyncSlot& sl = c->slot[k];
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(sl.done));
    memcpy(sl.out, (const uint8_t*)sl.pin + sl.res_at, (size_t)sl.n * 4u);
    if (sl.in_place) count_on_host(c, (const u32*)((const uint8_t*)sl.pin + sl.res_at), sl.n);
    sl.busy = false;
    return UNET_OK;
}

// No counterpart in the reference: unet_verify for callers that cannot allocate.  Same kernel; the device-side
// staging that unet_verify keeps in hipMalloc'ed buffers lives in the static `scratch` array instead, reached
// only through the per-device address taken at unet_open, and the host side uses the pinned buffers the
// context already owns.  Nothing here allocates or frees.
int unet_verify_static(unet_ctx* c, const uint8_t* arena, uint64_t arena_bytes, const uint32_t* off16, uint32_t stride16,
                       const uint32_t* len, uint32_t n, const unet_iface* iface, unet_result* out) {
    if (!c) return fail(UNET_E_INVAL, "null context");
    if (n == 0) return UNET_OK;
    if (!arena || !len || !iface || !out) return fail(UNET_E_INVAL, "null argument");
    if (!off16 && stride16 == 0) return fail(UNET_E_INVAL, "need an offset table or a non-zero stride");
    if (iface->mode != UNET_MODE_L3 && iface->mode != UNET_MODE_L2) return fail(UNET_E_INVAL, "bad iface mode %d", iface->mode);
    if (n > kStaticMaxRecords)
        return fail(UNET_E_RANGE, "unet_verify_static takes at most %u records per call, got %u (use unet_verify)", kStaticMaxRecords, n);
    // the records, wherever they sit in the caller's arena, must end inside its first kStaticMaxArena bytes
    Image im;
    int rc = plan_image(arena_bytes, off16, stride16, len, n, &im);
    if (rc) return rc;
    if (im.used > kStaticMaxArena)
        return fail(UNET_E_RANGE, "unet_verify_static takes records inside the arena's first %llu bytes, these reach byte %llu (use unet_verify)",
                    (unsigned long long)kStaticMaxArena, (unsigned long long)im.used);
    HIP_TRY(hipSetDevice(c->device));
    // every context on this GPU shares the one array: a call that finds another one inside is refused
    StaticInUse in_use(c->device);
    if (!in_use.mine) return fail(UNET_E_BUSY, "unet_verify_static: another call is using device %d's static array", c->device);
    // one staging image [len | off16 | arena], one H2D copy, one launch, one D2H copy, all on the compute stream
    HIP_TRY(hipEventSynchronize(c->pinned_free[0]));
    HIP_TRY(hipEventSynchronize(c->pinned_free[1]));
    fill_image(c, (uint8_t*)c->pinned[0], im, arena, off16, len, n);
    uint8_t* dev = c->scratch_dev;  // never the array's name: that is not a device address in host code
    HIP_TRY(hipMemcpyAsync(dev, c->pinned[0], im.total, hipMemcpyHostToDevice, c->s_comp));
    u32* d_out = (u32*)(dev + kStaticOutAt);
    const int tile = c->tile_override ? c->tile_override : tile_for(im.sum_len / n, true, im.max_len);
    if ((rc = launch_verify(c->s_comp, image_batch(dev, im, off16 != nullptr, stride16, n), iface, tile, d_out))) return rc;
    // counted ahead of the copy, so that nothing reads the shared array once the wait below has ended
    if ((rc = count_into_ctx(c, c->s_comp, d_out, n))) return rc;
    HIP_TRY(hipMemcpyAsync(c->pinned[1], d_out, (size_t)n * 4u, hipMemcpyDeviceToHost, c->s_comp));
    HIP_TRY(hipEventRecord(c->results_ready, c->s_comp));
    HIP_TRY(hipEventSynchronize(c->results_ready));
    memcpy(out, c->pinned[1], (size_t)n * 4u);
    return UNET_OK;
}

/* ---------------------------------------------------------------- transmit path */

int unet_generate_dev(unet_ctx* ctx, void* stream, uint8_t* d_arena, uint64_t arena_bytes, const uint32_t* d_off16,
                      uint32_t stride16, const uint32_t* d_len, uint32_t n, uint8_t* d_status) {
    int rc = check_batch(ctx, d_arena, d_off16, stride16, d_len, n);
    if (rc) return rc;
    if (n == 0) return UNET_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    Batch b{d_arena, arena_bytes, d_off16, stride16, d_len, n};
    if (arena_too_small(arena_bytes)) return fail(UNET_E_INVAL, "arena must hold at least 16 bytes");
    b.tile_n = (u32)pick_tile(ctx, arena_bytes, stride16, n);
    if ((rc = check_tiles(n, b.tile_n))) return rc;
    const dim3 grid(tiles_grid(n, b.tile_n));
    switch (kernel_tile((int)b.tile_n)) {
        case 16: hipLaunchKernelGGL((k_generate<16, true>), grid, dim3(kBlock), 0, pick(ctx, stream), b, d_arena, (u32*)nullptr, d_status); break;
        case 32: hipLaunchKernelGGL((k_generate<32, true>), grid, dim3(kBlock), 0, pick(ctx, stream), b, d_arena, (u32*)nullptr, d_status); break;
        default: hipLaunchKernelGGL((k_generate<64, true>), grid, dim3(kBlock), 0, pick(ctx, stream), b, d_arena, (u32*)nullptr, d_status); break;
    }
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

int unet_checksum_fields_dev(unet_ctx* ctx, void* stream, const uint8_t* d_arena, uint64_t arena_bytes, const uint32_t* d_off16,
                             uint32_t stride16, const uint32_t* d_len, uint32_t n, unet_tx_fields* d_fields, uint8_t* d_status) {
    static_assert(sizeof(unet_tx_fields) == 4, "field pair layout");
    int rc = check_batch(ctx, d_arena, d_off16, stride16, d_len, n);
    if (rc) return rc;
    if (n == 0) return UNET_OK;
    if (!d_fields) return fail(UNET_E_INVAL, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    Batch b{d_arena, arena_bytes, d_off16, stride16, d_len, n};
    if (arena_too_small(arena_bytes)) return fail(UNET_E_INVAL, "arena must hold at least 16 bytes");
    b.tile_n = (u32)pick_tile(ctx, arena_bytes, stride16, n);
    if ((rc = check_tiles(n, b.tile_n))) return rc;
    const dim3 grid(tiles_grid(n, b.tile_n));
    switch (kernel_tile((int)b.tile_n)) {
        case 16: hipLaunchKernelGGL((k_generate<16, false>), grid, dim3(kBlock), 0, pick(ctx, stream), b, (uint8_t*)nullptr, (u32*)d_fields, d_status); break;
        case 32: hipLaunchKernelGGL((k_generate<32, false>), grid, dim3(kBlock), 0, pick(ctx, stream), b, (uint8_t*)nullptr, (u32*)d_fields, d_status); break;
        default: hipLaunchKernelGGL((k_generate<64, false>), grid, dim3(kBlock), 0, pick(ctx, stream), b, (uint8_t*)nullptr, (u32*)d_fields, d_status); break;
    }
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

namespace {
// Stores the checksum values the GPU computed into the caller's packets (those whose status is 0).
void patch_fields(uint8_t* arena, const uint32_t* off16, uint32_t stride16, uint32_t n, const unet_tx_fields* fields, const uint8_t* st,
                  uint8_t* status) {
    for (u32 i = 0; i < n; i++) {
        if (status) status[i] = st[i];
        if (st[i] != 0) continue;
        uint8_t* p = arena + (off16 ? (u64)off16[i] : (u64)i * stride16) * 16u;
        p[10] = (uint8_t)(fields[i].ip_csum >> 8);
        p[11] = (uint8_t)fields[i].ip_csum;
        uint8_t* f = p + (p[9] == 17 ? 26 : 22);  // UDP checksum field / ICMP checksum field behind a 20-byte header
        f[0] = (uint8_t)(fields[i].l4_csum >> 8);
        f[1] = (uint8_t)fields[i].l4_csum;
    }
}
}  // namespace

int unet_generate(unet_ctx* c, uint8_t* arena, uint64_t arena_bytes, const uint32_t* off16, uint32_t stride16,
                  const uint32_t* len, uint32_t n, uint8_t* status) {
    if (!c) return fail(UNET_E_INVAL, "null context");
    if (n == 0) return UNET_OK;
    if (!arena || !len) return fail(UNET_E_INVAL, "null argument");
    if (!off16 && stride16 == 0) return fail(UNET_E_INVAL, "need an offset table or a non-zero stride");
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if (arena_bytes + 2ull * n * 4u + 64u <= kStageBytes && (u64)n * 5u + 16u <= kStageBytes) {
        // a few packets: the kernel reads them where the staging image lies and leaves values and status in the
        // other pinned buffer (see unet_verify)
        Image im;
        if ((rc = plan_image(arena_bytes, off16, stride16, len, n, &im))) return rc;
        const int tile = in_place_tile(c->tile_override ? c->tile_override : tile_for(im.sum_len / n, true, im.max_len));
        if (read_in_place(c, im.total, n, tile)) {
            HIP_TRY(hipEventSynchronize(c->pinned_free[0]));
            HIP_TRY(hipEventSynchronize(c->pinned_free[1]));
            fill_image(c, (uint8_t*)c->pinned[0], im, arena, off16, len, n);
            const uint8_t* di = (const uint8_t*)c->pinned_dev[0];
            const u64 st_at = ((u64)n * 4u + 15u) & ~15ull;
            c->tile_once = tile;
            rc = unet_checksum_fields_dev(c, nullptr, di + im.at_arena, im.used, off16 ? (const uint32_t*)(di + im.at_off) : nullptr, stride16,
                                          (const uint32_t*)di, n, (unet_tx_fields*)c->pinned_dev[1], (uint8_t*)c->pinned_dev[1] + st_at);
            c->tile_once = 0;
            if (rc) return rc;
            HIP_TRY(hipStreamSynchronize(c->s_comp));
            patch_fields(arena, off16, stride16, n, (const unet_tx_fields*)c->pinned[1], (const uint8_t*)c->pinned[1] + st_at, status);
            return UNET_OK;
        }
    }
    Batch b;
    u32 mean = 0, maxlen = 0;
    rc = upload_batch(c, arena, arena_bytes, off16, stride16, len, n, &b, &mean, &maxlen);
    LenScope hint(c, mean, maxlen);
    if (rc) return rc;
    // the GPU computes the two checksum values per packet; only those 4 bytes (and the status) travel
    // back, and are stored into the caller's packets here
    if ((rc = ensure(c, c->d_out, (size_t)n * 5))) return rc;
    unet_tx_fields* d_fields = (unet_tx_fields*)c->d_out.p;
    uint8_t* d_status = (uint8_t*)c->d_out.p + (size_t)n * 4;
    rc = unet_checksum_fields_dev(c, nullptr, (const uint8_t*)c->d_arena.p, arena_bytes, b.off16, b.stride16, b.len, n, d_fields, d_status);
    if (rc) return rc;
    std::vector<unet_tx_fields> fields;
    std::vector<uint8_t> st;
    try {
        fields.resize(n);
        st.resize(n);
    } catch (const std::bad_alloc&) {
        return fail(UNET_E_NOMEM, "out of host memory");
    }
    if ((rc = d2h_sync(c, fields.data(), d_fields, (size_t)n * 4))) return rc;
    if ((rc = d2h_sync(c, st.data(), d_status, (size_t)n))) return rc;
    patch_fields(arena, off16, stride16, n, fields.data(), st.data(), status);
    return UNET_OK;
}

int unet_tx_build_dev(unet_ctx* ctx, void* stream, const uint8_t* d_payload_arena, uint64_t payload_arena_bytes,
                      const uint32_t* d_payload_off16, uint32_t payload_stride16, const uint32_t* d_payload_len,
                      const unet_tx_desc* d_desc, uint32_t n, const unet_iface* iface, uint8_t* d_out_arena,
                      uint64_t out_arena_bytes, const uint32_t* d_out_off16, uint32_t* d_out_len, uint8_t* d_status) {
    return unet_tx_build_opts_dev(ctx, stream, d_payload_arena, payload_arena_bytes, d_payload_off16, payload_stride16, d_payload_len, d_desc, n,
                                  iface, d_out_arena, out_arena_bytes, d_out_off16, d_out_len, d_status, 0u);
}

int unet_tx_build_opts_dev(unet_ctx* ctx, void* stream, const uint8_t* d_payload_arena, uint64_t payload_arena_bytes,
                           const uint32_t* d_payload_off16, uint32_t payload_stride16, const uint32_t* d_payload_len,
                           const unet_tx_desc* d_desc, uint32_t n, const unet_iface* iface, uint8_t* d_out_arena,
                           uint64_t out_arena_bytes, const uint32_t* d_out_off16, uint32_t* d_out_len, uint8_t* d_status,
                           uint32_t flags) {
    static_assert(sizeof(unet_tx_desc) == sizeof(TxDesc), "descriptor layout");
    static_assert(UNET_TX_FILL_LINES == kTxFillLines, "flag values");
    if ((flags & ~(uint32_t)UNET_TX_FILL_LINES) != 0u) return fail(UNET_E_INVAL, "unknown flag bits 0x%x", flags & ~(uint32_t)UNET_TX_FILL_LINES);
    static_assert(UNET_TX_E_PROTO == UNET_TX_E_PROTO_ && UNET_TX_E_TOO_LONG == UNET_TX_E_TOO_LONG_ && UNET_TX_E_RANGE == UNET_TX_E_RANGE_, "status codes");
    int rc = check_batch(ctx, d_payload_arena, d_payload_off16, payload_stride16, d_payload_len, n);
    if (rc) return rc;
    if (n == 0) return UNET_OK;
    if (!d_desc || !iface || !d_out_arena || !d_out_off16 || !d_out_len) return fail(UNET_E_INVAL, "null argument");
    if (((uintptr_t)d_out_arena & 15u) != 0) return fail(UNET_E_INVAL, "output arena must be 16-byte aligned");
    if (iface->mode != UNET_MODE_L3 && iface->mode != UNET_MODE_L2) return fail(UNET_E_INVAL, "bad iface mode %d", iface->mode);
    HIP_TRY(hipSetDevice(ctx->device));
    Batch b{d_payload_arena, payload_arena_bytes, d_payload_off16, payload_stride16, d_payload_len, n};
    const IfaceDev ifc = to_dev(iface);
#define UNET_TX_LAUNCH(M, T)                                                                                             \
    hipLaunchKernelGGL((k_tx_build<M, T>), dim3(tiles_grid(n, b.tile_n)), dim3(kBlock), ctx->lds_pad + (stage ? tx_stage_bytes(M, T) : 0u), pick(ctx, stream), b, (const TxDesc*)d_desc, ifc, \
                       d_out_arena, out_arena_bytes, d_out_off16, d_out_len, d_status, flags | (stage ? 0u : kTxNoStage))
    const u64 mean_payload = pick_mean(ctx, payload_arena_bytes, payload_stride16, n);
    int tile = tx_tile_for(mean_payload);
    if (ctx->tx_tile_override) tile = ctx->tx_tile_override;
    b.tile_n = (u32)tile;  // fewer packets per wave than the smallest instantiation: that kernel, part of its lanes
    // Long payloads (mean above 1 KiB): one packet per wave or group of waves, see k_tx_build_wave.  (A launch holds
    // fewer than 2^32 threads; a batch too large for that stays with the tile kernel.)
    if (ctx->tx_wave == 2 || (ctx->tx_wave == 1 && !ctx->tx_tile_override && mean_payload > kTxWaveFrom)) {
        const u32 g2 = ctx->tx_group_log2 >= 0 ? (u32)ctx->tx_group_log2 : tx_group_log2_for(mean_payload, iface->mode == UNET_MODE_L2);
        const u32 group = 1u << g2;
        const u32 wpb = group > ctx->tx_wave_block / kWave ? group : ctx->tx_wave_block / kWave;  // waves per workgroup
        const u64 blocks = ((u64)n * group + wpb - 1) / wpb;
        if (blocks * wpb * kWave < (1ull << 32)) {
            const dim3 grid((u32)blocks), block(wpb * kWave);
#define UNET_TXW_LAUNCH(M, P)                                                                                                      \
    hipLaunchKernelGGL((k_tx_build_wave<M, P>), grid, block, 0, pick(ctx, stream), b, (const TxDesc*)d_desc, ifc, d_out_arena, \
                       out_arena_bytes, d_out_off16, d_out_len, d_status, flags, g2)
            const bool l2 = iface->mode == UNET_MODE_L2;
            // Both rows streaming while row 1 is at most half full (1472-byte payloads: 6.17 against 5.95 TB/s), row 1 --
            // the row a payload of up to 2 KiB ends in -- cached beyond that (2000 B: 6.53 against 6.14); row 0 cached too
            // is never better by more than 2 % and costs 7 % at 1100 B.
            const int policy = ctx->tx_wave_policy >= 0 ? ctx->tx_wave_policy : (mean_payload <= 1600 ? 3 : 1);
            switch (policy & 3) {
            case 0: if (l2) UNET_TXW_LAUNCH(MODE_L2, 0); else UNET_TXW_LAUNCH(MODE_L3, 0); break;
            case 1: if (l2) UNET_TXW_LAUNCH(MODE_L2, 1); else UNET_TXW_LAUNCH(MODE_L3, 1); break;
            case 2: if (l2) UNET_TXW_LAUNCH(MODE_L2, 2); else UNET_TXW_LAUNCH(MODE_L3, 2); break;
            default: if (l2) UNET_TXW_LAUNCH(MODE_L2, 3); else UNET_TXW_LAUNCH(MODE_L3, 3); break;
            }
#undef UNET_TXW_LAUNCH
            HIP_TRY(hipGetLastError());
            return UNET_OK;
        }
    }
    if ((rc = check_tiles(n, b.tile_n))) return rc;
    // Output staging in LDS (k_tx_build) costs the launch 40 KiB of LDS per workgroup, which only pays if tiles qualify:
    // batches of short packets (up to 320 bytes of mean payload), and longer ones when no payload can be long -- a fixed
    // stride of at most 1008 bytes, lengths the host has seen, or the caller's word (unet_hint_max_len).  A packed batch of
    // mixed sizes goes without.
    const u64 longest_group_payload = 16u * (kWave - (iface->mode == UNET_MODE_L2 ? 3u : 2u));
    const bool stage = ctx->tx_stage && (tile >= 32 || (payload_stride16 != 0 && (u64)payload_stride16 * 16u <= longest_group_payload + 15u) ||
                                         (ctx->len_known && ctx->max_len_known <= longest_group_payload) ||
                                         (ctx->max_len_hint != 0 && ctx->max_len_hint <= longest_group_payload));
    if (iface->mode == UNET_MODE_L2) {
        if (tile <= 8) UNET_TX_LAUNCH(MODE_L2, 8);
        else if (tile == 16) UNET_TX_LAUNCH(MODE_L2, 16);
        else if (tile == 32) UNET_TX_LAUNCH(MODE_L2, 32);
        else UNET_TX_LAUNCH(MODE_L2, 64);
    } else {
        if (tile <= 4) UNET_TX_LAUNCH(MODE_L3, 4);
        else if (tile == 8) UNET_TX_LAUNCH(MODE_L3, 8);
        else if (tile == 16) UNET_TX_LAUNCH(MODE_L3, 16);
        else if (tile == 32) UNET_TX_LAUNCH(MODE_L3, 32);
        else UNET_TX_LAUNCH(MODE_L3, 64);
    }
#undef UNET_TX_LAUNCH
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

int unet_tx_layout_aligned(const uint32_t* payload_len, uint32_t n, int mode, uint32_t align16, uint32_t* out_off16,
                           uint64_t* out_arena_bytes) {
    if ((n && !payload_len) || !out_arena_bytes) return fail(UNET_E_INVAL, "null argument");
    if (mode != UNET_MODE_L3 && mode != UNET_MODE_L2) return fail(UNET_E_INVAL, "bad mode %d", mode);
    if (align16 == 0 || align16 > 64 || (align16 & (align16 - 1u)) != 0) return fail(UNET_E_INVAL, "align16 must be 1, 2, 4 ... 64, got %u", align16);
    u64 slot = 0;
    for (u32 i = 0; i < n; i++) {
        if (payload_len[i] > kMaxLen) return fail(UNET_E_RANGE, "payload %u: length %u > 65535", i, payload_len[i]);
        const u64 l3 = 28ull + payload_len[i];
        const u64 frame = mode == UNET_MODE_L2 ? 14ull + (l3 < 60 ? 60 : l3) : l3;
        if (slot > 0xffffffffull) return fail(UNET_E_RANGE, "output needs more than 2^32 16-byte slots");
        if (out_off16) out_off16[i] = (u32)slot;
        slot += (((frame + 15u) >> 4) + align16 - 1u) & ~(u64)(align16 - 1u);
    }
    *out_arena_bytes = slot ? slot * 16u : 16u;
    return UNET_OK;
}

int unet_tx_layout(const uint32_t* payload_len, uint32_t n, int mode, uint32_t* out_off16, uint64_t* out_arena_bytes) {
    return unet_tx_layout_aligned(payload_len, n, mode, 1, out_off16, out_arena_bytes);
}

// unet_tx_build on a batch too large for one staging image, its payloads in index order in the caller's arena: the
// batch is cut into runs of packets whose payloads and whose packets each fit half a pinned buffer; while run k is built,
// run k + 1's payloads travel up (pinned[0], two halves in turn, copy stream) and run k - 1's packets travel down
// (pinned[1], two halves, a stream of their own) -- the host link carries both directions at once, and the caller's
// thread copies into and out of the pinned halves in between.  (Packets copied by the runtime straight into the
// caller's pageable memory come back at a third of the link's rate: 262,144 datagrams of 1472 bytes took 31 ms that
// way, 8 of them going up.)  The tables go up first and come back last, as in the unpipelined form.
int tx_build_runs(unet_ctx* c, const uint8_t* payload_arena, uint64_t payload_arena_bytes, const uint32_t* payload_off16,
                  uint32_t payload_stride16, const uint32_t* payload_len, const unet_tx_desc* desc, uint32_t n, const unet_iface* iface,
                  uint8_t* out_arena, uint64_t need, const uint32_t* out_off16, uint32_t* out_len, uint8_t* status);
int tx_build_pipelined(unet_ctx* c, const uint8_t* payload_arena, uint64_t payload_arena_bytes, const uint32_t* payload_off16,
                       uint32_t payload_stride16, const uint32_t* payload_len, const unet_tx_desc* desc, uint32_t n, const unet_iface* iface,
                       uint8_t* out_arena, uint64_t need, const uint32_t* out_off16, uint32_t* out_len, uint8_t* status) {
    const int rc = tx_build_runs(c, payload_arena, payload_arena_bytes, payload_off16, payload_stride16, payload_len, desc, n, iface, out_arena, need,
                                 out_off16, out_len, status);
    if (rc) {
        // nothing may still be reading or writing the pinned halves when the next call starts (g_err keeps this failure's text)
        (void)hipStreamSynchronize(c->s_copy);
        (void)hipStreamSynchronize(c->s_comp);
        (void)hipStreamSynchronize(c->s_down);
    }
    return rc;
}
int tx_build_runs(unet_ctx* c, const uint8_t* payload_arena, uint64_t payload_arena_bytes, const uint32_t* payload_off16,
                  uint32_t payload_stride16, const uint32_t* payload_len, const unet_tx_desc* desc, uint32_t n, const unet_iface* iface,
                  uint8_t* out_arena, uint64_t need, const uint32_t* out_off16, uint32_t* out_len, uint8_t* status) {
    // bytes per run and direction: half a pinned buffer, less for batches of a few MiB so that they still make six runs or more
    const u64 larger = payload_arena_bytes > need ? payload_arena_bytes : need;
    u64 kHalf = (larger / 6u + (1u << 20) - 1u) & ~(u64)((1u << 20) - 1u);
    kHalf = kHalf < (1u << 20) ? (1u << 20) : (kHalf > kStageBytes / 2 ? kStageBytes / 2 : kHalf);
    int rc;
    if ((rc = ensure(c, c->d_arena, payload_arena_bytes))) return rc;
    if ((rc = ensure(c, c->d_len, (size_t)n * 4))) return rc;
    if (payload_off16 && (rc = ensure(c, c->d_off, (size_t)n * 4))) return rc;
    const size_t o_desc = 0, o_off = (size_t)n * sizeof(unet_tx_desc), o_len = o_off + (size_t)n * 4, o_st = o_len + (size_t)n * 4;
    if ((rc = ensure(c, c->d_tx_tab, o_st + n))) return rc;
    if ((rc = ensure(c, c->d_tx, need))) return rc;
    char* tab = (char*)c->d_tx_tab.p;
    if ((rc = h2d(c, c->d_len.p, payload_len, (size_t)n * 4))) return rc;
    if (payload_off16 && (rc = h2d(c, c->d_off.p, payload_off16, (size_t)n * 4))) return rc;
    if ((rc = h2d(c, tab + o_desc, desc, (size_t)n * sizeof(unet_tx_desc)))) return rc;
    if ((rc = h2d(c, tab + o_off, out_off16, (size_t)n * 4))) return rc;
    if ((rc = copies_before_compute(c))) return rc;
    HIP_TRY(hipMemsetAsync(c->d_tx.p, 0, need, c->s_comp));  // refused packets leave their slots untouched
    HIP_TRY(hipEventSynchronize(c->pinned_free[0]));        // the table uploads are through with the pinned buffers
    HIP_TRY(hipEventSynchronize(c->pinned_free[1]));
    // (Arenas that are pinned memory of the caller's own go through the halves all the same: copied straight from and
    // into them the same batch took 16.7 ms against 16.2 -- the two directions of the link do not overlap on this
    // platform, so the staging copies, which run beside the transfers, cost nothing.)
    auto slot_of = [&](u32 i) { return payload_off16 ? (u64)payload_off16[i] : (u64)i * payload_stride16; };
    auto out_at = [&](u32 i) { return i < n ? (u64)out_off16[i] * 16u : need; };
    struct Run { u32 a, b; u64 in_lo, in_hi; };
    Run prev{0, 0, 0, 0};
    u32 k = 0;
    for (u32 a = 0; a < n; ++k) {
        // packets a .. b-1: as many as keep both byte ranges within half a pinned buffer (one packet always fits)
        Run r{a, a, slot_of(a) * 16u, slot_of(a) * 16u};
        while (r.b < n) {
            const u64 end = slot_of(r.b) * 16u + (((u64)payload_len[r.b] + 15u) & ~15ull);
            const u64 hi = end > r.in_hi ? end : r.in_hi;
            if (r.b > a && (hi - r.in_lo > kHalf || out_at(r.b + 1u) - out_at(a) > kHalf)) break;
            r.in_hi = hi;
            ++r.b;
        }
        const int h = (int)(k & 1u);
        uint8_t* up = (uint8_t*)c->pinned[0] + (size_t)h * kHalf;
        if (k >= 2u) HIP_TRY(hipEventSynchronize(c->tx_up[h]));  // run k - 2 has left this half
        if (r.in_hi > r.in_lo) {
            c->pool->copy(up, payload_arena + r.in_lo, (size_t)(r.in_hi - r.in_lo));
            HIP_TRY(hipMemcpyAsync((char*)c->d_arena.p + r.in_lo, up, (size_t)(r.in_hi - r.in_lo), hipMemcpyHostToDevice, c->s_copy));
        }
        HIP_TRY(hipEventRecord(c->tx_up[h], c->s_copy));
        HIP_TRY(hipStreamWaitEvent(c->s_comp, c->tx_up[h], 0));
        {
            // (with a fixed stride the kernel finds packet i at i strides from the arena it is given)
            const u64 skip = payload_off16 ? 0u : (u64)r.a * payload_stride16 * 16u;
            rc = unet_tx_build_dev(c, nullptr, (const uint8_t*)c->d_arena.p + skip, payload_arena_bytes - skip,
                                   payload_off16 ? (const uint32_t*)c->d_off.p + r.a : nullptr, payload_stride16, (const uint32_t*)c->d_len.p + r.a,
                                   (const unet_tx_desc*)(tab + o_desc) + r.a, r.b - r.a, iface, (uint8_t*)c->d_tx.p, need,
                                   (const uint32_t*)(tab + o_off) + r.a, (uint32_t*)(tab + o_len) + r.a, (uint8_t*)(tab + o_st) + r.a);
            if (rc) return rc;
        }
        HIP_TRY(hipEventRecord(c->tx_built[h], c->s_comp));
        HIP_TRY(hipStreamWaitEvent(c->s_down, c->tx_built[h], 0));
        // (this half's previous contents, run k - 2's packets, were copied out below while run k - 1 was in flight)
        HIP_TRY(hipMemcpyAsync((uint8_t*)c->pinned[1] + (size_t)h * kHalf, (const uint8_t*)c->d_tx.p + out_at(r.a), (size_t)(out_at(r.b) - out_at(r.a)),
                               hipMemcpyDeviceToHost, c->s_down));
        HIP_TRY(hipEventRecord(c->tx_down[h], c->s_down));
        if (k >= 1u) {
            HIP_TRY(hipEventSynchronize(c->tx_down[h ^ 1]));
            c->pool->copy(out_arena + out_at(prev.a), (const uint8_t*)c->pinned[1] + (size_t)(h ^ 1) * kHalf, (size_t)(out_at(prev.b) - out_at(prev.a)));
        }
        prev = r;
        a = r.b;
    }
    const int hl = (int)((k - 1u) & 1u);
    HIP_TRY(hipEventSynchronize(c->tx_down[hl]));
    c->pool->copy(out_arena + out_at(prev.a), (const uint8_t*)c->pinned[1] + (size_t)hl * kHalf, (size_t)(out_at(prev.b) - out_at(prev.a)));
    HIP_TRY(hipStreamSynchronize(c->s_down));
    HIP_TRY(hipStreamSynchronize(c->s_copy));
    if ((rc = d2h_sync(c, out_len, tab + o_len, (size_t)n * 4))) return rc;
    if (status && (rc = d2h_sync(c, status, tab + o_st, (size_t)n))) return rc;
    return UNET_OK;
}

int unet_tx_build(unet_ctx* c, const uint8_t* payload_arena, uint64_t payload_arena_bytes, const uint32_t* payload_off16,
                  uint32_t payload_stride16, const uint32_t* payload_len, const unet_tx_desc* desc, uint32_t n,
                  const unet_iface* iface, uint8_t* out_arena, uint64_t out_arena_bytes, uint32_t* out_off16,
                  uint32_t* out_len, uint8_t* status) {
    if (!c) return fail(UNET_E_INVAL, "null context");
    if (n == 0) return UNET_OK;
    if (!payload_arena || !payload_len || !desc || !iface || !out_arena || !out_off16 || !out_len) return fail(UNET_E_INVAL, "null argument");
    if (!payload_off16 && payload_stride16 == 0) return fail(UNET_E_INVAL, "need an offset table or a non-zero stride");
    u64 need = 0;
    int rc = unet_tx_layout(payload_len, n, iface->mode, out_off16, &need);
    if (rc) return rc;
    if (out_arena_bytes < need) return fail(UNET_E_RANGE, "output arena of %llu bytes, need %llu", (unsigned long long)out_arena_bytes, (unsigned long long)need);
    HIP_TRY(hipSetDevice(c->device));
    {
        // A few packets (the echo replies and datagrams of one drained queue): everything the kernel reads goes into
        // one pinned image [lengths | offsets | output offsets | descriptors | payloads] that it reads in place, and
        // it writes [output lengths | status | packets] into the other pinned buffer: one launch, one wait.
        Image im;
        if ((rc = plan_image(payload_arena_bytes, payload_off16, payload_stride16, payload_len, n, &im))) return rc;
        const u64 tab = ((u64)n * 4u + 15u) & ~15ull, dtab = ((u64)n * sizeof(unet_tx_desc) + 15u) & ~15ull;
        const u64 in_off = tab, in_ooff = 2u * tab, in_desc = 3u * tab, in_arena = in_desc + dtab, in_total = in_arena + im.used;
        const u64 o_len = 0, o_st = tab, o_arena = (tab + (((u64)n + 15u) & ~15ull) + 255u) & ~255ull, o_total = o_arena + need;
        const bool in_place = in_total <= kTxInPlaceBytes && o_total <= kTxInPlaceBytes && in_total <= c->zero_copy_bytes;
        // From 16 MiB up a batch in index order is better cut into runs that overlap the two directions of the host
        // link (tx_build_pipelined) than sent as one image each way: 16,384 datagrams of 1472 bytes 1.55 -> 1.30 ms,
        // 32,768 4.4 -> 2.4 ms; 8192 the same either way, 4096 0.44 ms as one image against 0.70 in runs.
        bool as_runs = false;
        if (!in_place && c->tx_pipeline && (im.used >= c->tx_pipeline_min || need >= c->tx_pipeline_min))
            if ((rc = check_records(payload_arena_bytes, payload_off16, payload_stride16, payload_len, n, nullptr, nullptr, &as_runs))) return rc;
        if (!as_runs && (in_place || (in_total <= kStageBytes && o_total <= kStageBytes))) {
            // (larger batches that still fit the pinned buffers: the same two images, one copy command each way)
            HIP_TRY(hipEventSynchronize(c->pinned_free[0]));
            HIP_TRY(hipEventSynchronize(c->pinned_free[1]));
            uint8_t* pi = (uint8_t*)c->pinned[0];
            uint8_t* po = (uint8_t*)c->pinned[1];
            memcpy(pi, payload_len, (size_t)n * 4u);
            if (payload_off16) memcpy(pi + in_off, payload_off16, (size_t)n * 4u);
            memcpy(pi + in_ooff, out_off16, (size_t)n * 4u);
            memcpy(pi + in_desc, desc, (size_t)n * sizeof(unet_tx_desc));
            c->pool->copy(pi + in_arena, payload_arena, (size_t)(im.used < payload_arena_bytes ? im.used : payload_arena_bytes));
            const uint8_t* di;
            uint8_t* dout;
            if (in_place) {
                memset(po, 0, o_total);  // refused packets and pad bytes read as zeros
                di = (const uint8_t*)c->pinned_dev[0];
                dout = (uint8_t*)c->pinned_dev[1];
            } else {
                if ((rc = ensure(c, c->d_arena, in_total))) return rc;
                if ((rc = ensure(c, c->d_tx, o_total))) return rc;
                HIP_TRY(hipMemcpyAsync(c->d_arena.p, pi, in_total, hipMemcpyHostToDevice, c->s_comp));
                HIP_TRY(hipMemsetAsync(c->d_tx.p, 0, o_total, c->s_comp));
                di = (const uint8_t*)c->d_arena.p;
                dout = (uint8_t*)c->d_tx.p;
            }
            LenScope hint(c, (u32)(im.sum_len / n), im.max_len);
            rc = unet_tx_build_dev(c, nullptr, di + in_arena, im.used, payload_off16 ? (const uint32_t*)(di + in_off) : nullptr, payload_stride16,
                                   (const uint32_t*)di, (const unet_tx_desc*)(di + in_desc), n, iface, dout + o_arena, need,
                                   (const uint32_t*)(di + in_ooff), (uint32_t*)(dout + o_len), dout + o_st);
            if (rc) return rc;
            if (!in_place) HIP_TRY(hipMemcpyAsync(po, dout, o_total, hipMemcpyDeviceToHost, c->s_comp));
            HIP_TRY(hipStreamSynchronize(c->s_comp));
            memcpy(out_len, po + o_len, (size_t)n * 4u);
            if (status) memcpy(status, po + o_st, n);
            c->pool->copy(out_arena, po + o_arena, need);
            return UNET_OK;
        }
    }
    Batch b;
    u32 mean = 0, maxlen = 0;
    bool ascending = false;
    if ((rc = check_records(payload_arena_bytes, payload_off16, payload_stride16, payload_len, n, &mean, &maxlen, &ascending))) return rc;
    if (ascending && c->tx_pipeline) {
        LenScope hint(c, mean, maxlen);
        return tx_build_pipelined(c, payload_arena, payload_arena_bytes, payload_off16, payload_stride16, payload_len, desc, n, iface, out_arena, need,
                                  out_off16, out_len, status);
    }
    if ((rc = upload_batch(c, payload_arena, payload_arena_bytes, payload_off16, payload_stride16, payload_len, n, &b, &mean, &maxlen))) return rc;
    LenScope hint(c, mean, maxlen);
    // tables: descriptors | output offsets | output lengths | status
    const size_t o_desc = 0, o_off = (size_t)n * sizeof(unet_tx_desc), o_len = o_off + (size_t)n * 4, o_st = o_len + (size_t)n * 4;
    if ((rc = ensure(c, c->d_tx_tab, o_st + n))) return rc;
    if ((rc = ensure(c, c->d_tx, need))) return rc;
    char* tab = (char*)c->d_tx_tab.p;
    if ((rc = h2d(c, tab + o_desc, desc, (size_t)n * sizeof(unet_tx_desc)))) return rc;
    if ((rc = h2d(c, tab + o_off, out_off16, (size_t)n * 4))) return rc;
    if ((rc = copies_before_compute(c))) return rc;
    // refused packets leave their slots untouched: start from zeros so the caller's copy is defined
    HIP_TRY(hipMemsetAsync(c->d_tx.p, 0, need, c->s_comp));
    rc = unet_tx_build_dev(c, nullptr, b.arena, b.arena_bytes, b.off16, b.stride16, b.len, (const unet_tx_desc*)(tab + o_desc), n, iface,
                           (uint8_t*)c->d_tx.p, need, (const uint32_t*)(tab + o_off), (uint32_t*)(tab + o_len), (uint8_t*)(tab + o_st));
    if (rc) return rc;
    if ((rc = d2h_sync(c, out_len, tab + o_len, (size_t)n * 4))) return rc;
    if (status && (rc = d2h_sync(c, status, tab + o_st, (size_t)n))) return rc;
    return d2h_sync(c, out_arena, c->d_tx.p, need);
}

/* ---------------------------------------------------------------- counters */

int unet_count_verdicts_dev(unet_ctx* ctx, void* stream, const unet_result* d_results, uint32_t n, uint64_t* d_counts) {
    if (!ctx) return fail(UNET_E_INVAL, "null context");
    if (n == 0) return UNET_OK;
    if (!d_results) return fail(UNET_E_INVAL, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    u32 grid = (n + kBlock - 1) / kBlock;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_count_verdicts, dim3(grid), dim3(kBlock), 0, pick(ctx, stream), (const u32*)d_results, n,
                       d_counts ? (unsigned long long*)d_counts : ctx->d_counts);
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

int unet_counters(unet_ctx* c, uint64_t* out, int reset) {
    if (!c || !out) return fail(UNET_E_INVAL, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    // the counting kernels run behind the results on the streams that produced them
    HIP_TRY(hipStreamSynchronize(c->s_comp));
    for (auto& sl : c->slot) HIP_TRY(hipStreamSynchronize(sl.s));
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "counter width");
    HIP_TRY(hipMemcpy(out, c->d_counts, UNET_V_COUNT * sizeof(uint64_t), hipMemcpyDeviceToHost));
    for (u32 v = 0; v < UNET_V_COUNT; v++) out[v] += c->host_counts[v];  // batches small enough to be counted on the host
    if (reset) {
        HIP_TRY(hipMemset(c->d_counts, 0, UNET_V_COUNT * sizeof(uint64_t)));
        for (u32 v = 0; v < UNET_V_COUNT; v++) c->host_counts[v] = 0;
    }
    return UNET_OK;
}

int unet_counters_sum(unet_ctx* const* ctxs, uint32_t n_ctx, uint64_t* out, int reset) {
    if ((n_ctx && !ctxs) || !out) return fail(UNET_E_INVAL, "null argument");
    for (u32 v = 0; v < UNET_V_COUNT; v++) out[v] = 0;
    for (u32 i = 0; i < n_ctx; i++) {
        uint64_t one[UNET_V_COUNT];
        int rc = unet_counters(ctxs[i], one, reset);
        if (rc) return rc;
        for (u32 v = 0; v < UNET_V_COUNT; v++) out[v] += one[v];
    }
    return UNET_OK;
}

/* ---------------------------------------------------------------- per-flow TCP tracking (extension) */

static_assert(sizeof(unet_tcp_flow) == sizeof(TcpFlow), "flow table slot layout");
static_assert(UNET_TCP_RESET == TCP_RESET && UNET_TCPS_AFTER_CLOSE == TCPS_AFTER_CLOSE && UNET_TCP_FIN == TCP_FIN, "tcp codes");

static int log2_exact(uint32_t v) {
    if (v == 0 || (v & (v - 1)) != 0) return -1;
    int l = 0;
    while ((1u << l) != v) l++;
    return l;
}

uint32_t unet_tcp_slot(uint32_t src, uint32_t dst, uint16_t sport, uint16_t dport, uint32_t table_slots) {
    const int l = log2_exact(table_slots);
    if (l < 0) return 0;
    return tcp_slot(flow_hash5(src, dst, 6u, sport, dport), (u32)l);
}

namespace {
// Workspace of one tracking call over n records: keys and indices in and out of the sort (4 x u32 per record), the
// extracted segments, the list of long runs with its counter, the sort's own scratch.
struct TcpWs {
    size_t a4, aseg, alist, sort_bytes, total;
};
int tcp_workspace(u32 n, int l2s, TcpWs* w) {
    w->sort_bytes = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, w->sort_bytes, (const u32*)nullptr, (u32*)nullptr, (const u32*)nullptr, (u32*)nullptr, (int)n, 0,
                                               l2s + 1, (hipStream_t) nullptr));
    w->a4 = ((size_t)n * 4u + 255u) & ~(size_t)255u;
    w->aseg = ((size_t)n * sizeof(TcpSeg) + 255u) & ~(size_t)255u;
    w->alist = (((size_t)n / kTcpShortRun + 1u) * 8u + 256u + 255u) & ~(size_t)255u;  // a long run holds more than kTcpShortRun records
    w->total = 4u * w->a4 + w->aseg + w->alist + ((w->sort_bytes + 255u) & ~(size_t)255u) + 256u;
    return UNET_OK;
}
}  // namespace

int unet_tcp_track_workspace(uint32_t n, uint32_t table_slots, uint64_t* bytes) {
    if (!bytes) return fail(UNET_E_INVAL, "null argument");
    const int l2s = log2_exact(table_slots);
    if (l2s < 0 || l2s > 30) return fail(UNET_E_INVAL, "table_slots must be a power of two up to 2^30, got %u", table_slots);
    if (n > 0x7fffffffu) return fail(UNET_E_RANGE, "at most 2^31 - 1 records per call");
    TcpWs w;
    int rc = tcp_workspace(n ? n : 1u, l2s, &w);
    if (rc) return rc;
    *bytes = w.total;
    return UNET_OK;
}

int unet_tcp_track_ws_dev(unet_ctx* ctx, void* stream, const uint8_t* d_arena, uint64_t arena_bytes, const uint32_t* d_off16,
                          uint32_t stride16, const uint32_t* d_len, const unet_result* d_results, uint32_t n, const unet_iface* iface,
                          unet_tcp_flow* d_table, uint32_t table_slots, uint8_t* d_seg_state, void* d_workspace, uint64_t workspace_bytes,
                          unet_tcp_stats* d_stats) {
    static_assert(sizeof(unet_tcp_stats) == sizeof(TcpStats), "stats layout");
    int rc = check_batch(ctx, d_arena, d_off16, stride16, d_len, n);
    if (rc) return rc;
    if (n == 0) return UNET_OK;
    if (!d_results || !iface || !d_table || !d_seg_state || !d_workspace) return fail(UNET_E_INVAL, "null argument");
    if (iface->mode != UNET_MODE_L3 && iface->mode != UNET_MODE_L2) return fail(UNET_E_INVAL, "bad iface mode %d", iface->mode);
    const int l2s = log2_exact(table_slots);
    if (l2s < 0 || l2s > 30) return fail(UNET_E_INVAL, "table_slots must be a power of two up to 2^30, got %u", table_slots);
    if (n > 0x7fffffffu) return fail(UNET_E_RANGE, "at most 2^31 - 1 records per call");
    if (((uintptr_t)d_workspace & 255u) != 0) return fail(UNET_E_INVAL, "workspace must be 256-byte aligned");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    TcpWs ws;
    if ((rc = tcp_workspace(n, l2s, &ws))) return rc;
    if (workspace_bytes < ws.total)
        return fail(UNET_E_RANGE, "workspace of %llu bytes, %u records need %llu (unet_tcp_track_workspace)", (unsigned long long)workspace_bytes, n,
                    (unsigned long long)ws.total);
    uint8_t* w = (uint8_t*)d_workspace;
    u32 *keys_in = (u32*)w, *keys_out = (u32*)(w + ws.a4), *idx_in = (u32*)(w + 2u * ws.a4), *idx_out = (u32*)(w + 3u * ws.a4);
    TcpSeg* segs = (TcpSeg*)(w + 4u * ws.a4);
    u32* long_count = (u32*)(w + 4u * ws.a4 + ws.aseg);  // counter in the first 256 bytes, the list behind it
    u32* long_list = long_count + 64;
    void* sort_tmp = w + 4u * ws.a4 + ws.aseg + ws.alist;
    HIP_TRY(hipMemsetAsync(long_count, 0, 256, s));
    Batch b{d_arena, arena_bytes, d_off16, stride16, d_len, n};
    const u32 grid = (n + kBlock - 1) / kBlock;
    if (iface->mode == UNET_MODE_L2)
        hipLaunchKernelGGL(k_tcp_keys<MODE_L2>, dim3(grid), dim3(kBlock), 0, s, b, (const u32*)d_results, (u32)l2s, keys_in, idx_in, segs, d_seg_state);
    else
        hipLaunchKernelGGL(k_tcp_keys<MODE_L3>, dim3(grid), dim3(kBlock), 0, s, b, (const u32*)d_results, (u32)l2s, keys_in, idx_in, segs, d_seg_state);
    HIP_TRY(hipGetLastError());
    // radix sort is stable: records of one slot stay in index order
    size_t sort_bytes = ws.sort_bytes;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(sort_tmp, sort_bytes, (const u32*)keys_in, keys_out, (const u32*)idx_in, idx_out, (int)n, 0, l2s + 1, s));
    hipLaunchKernelGGL(k_tcp_apply, dim3(grid), dim3(kBlock), 0, s, (const u32*)keys_out, (const u32*)idx_out, (const TcpSeg*)segs, n, (u32)l2s,
                       (TcpFlow*)d_table, d_seg_state, long_list, long_count, (TcpStats*)d_stats);
    HIP_TRY(hipGetLastError());
    // the long runs (their number is known on the device only: a fixed grid shares them out; with none, its waves end at once)
    const u32 long_grid = grid < 1024u ? grid : 1024u;
    hipLaunchKernelGGL(k_tcp_apply_long, dim3(long_grid), dim3(kBlock), 0, s, (const u32*)keys_out, (const u32*)idx_out, (const TcpSeg*)segs,
                       (TcpFlow*)d_table, d_seg_state, (const u32*)long_list, (const u32*)long_count);
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

int unet_tcp_track_dev(unet_ctx* ctx, void* stream, const uint8_t* d_arena, uint64_t arena_bytes, const uint32_t* d_off16,
                       uint32_t stride16, const uint32_t* d_len, const unet_result* d_results, uint32_t n, const unet_iface* iface,
                       unet_tcp_flow* d_table, uint32_t table_slots, uint8_t* d_seg_state) {
    // the form that keeps its workspace in the context: it grows on demand, so this call may allocate (and a
    // reallocation waits for the device); callers that cannot have that pass their own to unet_tcp_track_ws_dev
    if (!ctx) return fail(UNET_E_INVAL, "null context");
    if (n == 0) return UNET_OK;
    uint64_t need = 0;
    int rc = unet_tcp_track_workspace(n, table_slots, &need);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    if ((rc = ensure(ctx, ctx->d_tcp, (size_t)need))) return rc;
    return unet_tcp_track_ws_dev(ctx, stream, d_arena, arena_bytes, d_off16, stride16, d_len, d_results, n, iface, d_table, table_slots, d_seg_state,
                                 ctx->d_tcp.p, ctx->d_tcp.cap, nullptr);
}

/* ---------------------------------------------------------------- flow hash */

uint32_t unet_flow_hash5(uint32_t src, uint32_t dst, uint8_t proto, uint16_t sport, uint16_t dport) {
    return flow_hash5(src, dst, proto, sport, dport);
}

uint32_t unet_flow_hash_record(const uint8_t* rec, size_t len, int mode) {
    if (!rec) return 0;
    if (mode == UNET_MODE_L2) {
        if (len < 14 || rec[12] != 0x08 || rec[13] != 0x00) return 0;
        rec += 14;
        len -= 14;
    }
    if (len < 20) return 0;
    auto be16 = [](const uint8_t* p) { return (u32)((p[0] << 8) | p[1]); };
    auto be32 = [](const uint8_t* p) { return ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | p[3]; };
    const u32 proto = rec[9];
    const size_t hl = (size_t)(rec[0] & 0x0f) * 4;
    u32 sport = 0, dport = 0;
    if ((proto == 17 || proto == 6) && hl >= 20 && hl + 4 <= len) {
        sport = be16(rec + hl);
        dport = be16(rec + hl + 2);
    }
    return flow_hash5(be32(rec + 12), be32(rec + 16), proto, sport, dport);
}

int unet_flow_hash_dev(unet_ctx* ctx, void* stream, const uint8_t* d_arena, uint64_t arena_bytes,
                       const uint32_t* d_off16, uint32_t stride16, const uint32_t* d_len, uint32_t n,
                       const unet_iface* iface, uint32_t* d_hash) {
    int rc = check_batch(ctx, d_arena, d_off16, stride16, d_len, n);
    if (rc) return rc;
    if (n == 0) return UNET_OK;
    if (!iface || !d_hash) return fail(UNET_E_INVAL, "null iface or output");
    if (iface->mode != UNET_MODE_L3 && iface->mode != UNET_MODE_L2) return fail(UNET_E_INVAL, "bad iface mode %d", iface->mode);
    HIP_TRY(hipSetDevice(ctx->device));
    Batch b{d_arena, arena_bytes, d_off16, stride16, d_len, n};
    const u32 grid = (n + kBlock - 1) / kBlock;
    if (iface->mode == UNET_MODE_L2)
        hipLaunchKernelGGL(k_flow_hash<MODE_L2>, dim3(grid), dim3(kBlock), 0, pick(ctx, stream), b, d_hash);
    else
        hipLaunchKernelGGL(k_flow_hash<MODE_L3>, dim3(grid), dim3(kBlock), 0, pick(ctx, stream), b, d_hash);
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

/* ---------------------------------------------------------------- synthetic traces */

static bool kind_ok(int kind) {
    return kind == UNET_SYNTH_IPV4_HDR20 || kind == UNET_SYNTH_UDP1500 || kind == UNET_SYNTH_IMIX || kind == UNET_SYNTH_FLOWS || kind == UNET_SYNTH_ODD;
}

uint32_t unet_synth_len(int kind, uint64_t seed, uint64_t idx) {
    if (!kind_ok(kind)) return 0;
    return unetsynth::shape_of(kind, seed, idx).len;
}

uint32_t unet_synth_flow_hash(int kind, uint64_t seed, uint64_t idx, const unet_iface* iface) {
    if (!kind_ok(kind) || !iface) return 0;
    const unetsynth::Shape s = unetsynth::shape_of(kind, seed, idx);
    const unetsynth::Tuple tp = unetsynth::tuple_of(kind, seed, idx, s, iface->unicast, iface->broadcast);
    return flow_hash5(tp.src, tp.dst, tp.proto, tp.sport, tp.dport);
}

int unet_synth_layout(int kind, uint64_t seed, uint64_t first_idx, uint32_t n, uint32_t* off16, uint32_t* len,
                      uint64_t* arena_bytes) {
    if (!kind_ok(kind)) return fail(UNET_E_INVAL, "unknown trace kind %d", kind);
    u64 slot = 0;
    for (u32 i = 0; i < n; i++) {
        const u32 l = unetsynth::shape_of(kind, seed, first_idx + i).len;
        if (len) len[i] = l;
        if (kind == UNET_SYNTH_IPV4_HDR20) {
            if (off16) off16[i] = 0;  // stride 20: not addressable in 16-byte slots
            continue;
        }
        if (slot > 0xffffffffull) return fail(UNET_E_RANGE, "trace does not fit 2^32 slots");
        if (off16) off16[i] = (u32)slot;
        slot += (l + 15u) >> 4;
    }
    if (arena_bytes) *arena_bytes = kind == UNET_SYNTH_IPV4_HDR20 ? (((u64)n * 20u + 15u) & ~15ull) : slot * 16u;
    return UNET_OK;
}

int unet_synth_host(int kind, uint64_t seed, uint64_t first_idx, uint32_t n, const unet_iface* iface,
                    const uint32_t* off16, const uint32_t* len, uint8_t* arena, uint64_t arena_bytes) {
    if (!kind_ok(kind)) return fail(UNET_E_INVAL, "unknown trace kind %d", kind);
    if (!iface || !arena) return fail(UNET_E_INVAL, "null argument");
    if (kind != UNET_SYNTH_IPV4_HDR20 && !off16) return fail(UNET_E_INVAL, "offset table required");
    (void)len;
    for (u32 r = 0; r < n; r++) {
        const u64 idx = first_idx + r;
        const u64 key = unetsynth::key_of(seed, idx);
        const unetsynth::Shape s = unetsynth::shape_of(kind, seed, idx);
        const unetsynth::Tuple tp = unetsynth::tuple_of(kind, seed, idx, s, iface->unicast, iface->broadcast);
        uint8_t h[96];
        if (kind == UNET_SYNTH_IPV4_HDR20) {
            if ((u64)r * 20u + 20u > arena_bytes) return fail(UNET_E_RANGE, "arena too small");
            unetsynth::build_header(kind, key, s, tp, iface->unicast, h);
            memcpy(arena + (u64)r * 20u, h, 20);
            continue;
        }
        const u64 base = (u64)off16[r] * 16u;
        const u32 padded = (s.len + 15u) & ~15u;
        if (base + padded > arena_bytes) return fail(UNET_E_RANGE, "arena too small");
        uint8_t* rec = arena + base;
        for (u32 k = 0; k < padded / 8; k++) {
            const u64 w = unetsynth::raw_word(key, k);
            memcpy(rec + 8u * k, &w, 8);  // little-endian host
        }
        const u32 hb = unetsynth::build_header(kind, key, s, tp, iface->unicast, h);
        memcpy(rec, h, hb);
        u32 fpos, fmask;
        if (unetsynth::flip_of(kind, key, s, &fpos, &fmask)) rec[fpos] ^= (uint8_t)fmask;
    }
    return UNET_OK;
}

int unet_synth_indexed_dev(unet_ctx* ctx, void* stream, int kind, uint64_t seed, uint64_t first_idx,
                           const uint32_t* d_rel_idx, uint32_t n, const unet_iface* iface, const uint32_t* d_off16,
                           const uint32_t* d_len, uint8_t* d_arena, uint64_t arena_bytes) {
    if (!ctx) return fail(UNET_E_INVAL, "null context");
    if (!kind_ok(kind)) return fail(UNET_E_INVAL, "unknown trace kind %d", kind);
    if (n == 0) return UNET_OK;
    if (!iface || !d_arena) return fail(UNET_E_INVAL, "null argument");
    if (kind != UNET_SYNTH_IPV4_HDR20 && (!d_off16 || !d_len)) return fail(UNET_E_INVAL, "offset and length tables required");
    if (kind == UNET_SYNTH_IPV4_HDR20 && (u64)n * 20u > arena_bytes) return fail(UNET_E_RANGE, "arena too small");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    if (kind != UNET_SYNTH_IPV4_HDR20) {
        // one wave per record, and a launch holds fewer than 2^32 threads: slices of 2^25 records
        constexpr u32 kSlice = 1u << 25;
        for (u32 at = 0; at < n; at += kSlice) {
            const u32 m = n - at < kSlice ? n - at : kSlice;
            const u32 grid = (m + (kBlock / kWave) - 1) / (kBlock / kWave);
            hipLaunchKernelGGL(k_synth_raw, dim3(grid), dim3(kBlock), 0, s, kind, seed, d_rel_idx ? first_idx : first_idx + at,
                               d_rel_idx ? d_rel_idx + at : nullptr, m, d_off16 + at, d_len + at, d_arena, arena_bytes);
            HIP_TRY(hipGetLastError());
        }
    }
    hipLaunchKernelGGL(k_synth_hdr, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, kind, seed, first_idx, d_rel_idx, n,
                       iface->unicast, iface->broadcast, d_off16, d_len, d_arena, arena_bytes);
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

int unet_synth_dev(unet_ctx* ctx, void* stream, int kind, uint64_t seed, uint64_t first_idx, uint32_t n,
                   const unet_iface* iface, const uint32_t* d_off16, const uint32_t* d_len, uint8_t* d_arena,
                   uint64_t arena_bytes) {
    return unet_synth_indexed_dev(ctx, stream, kind, seed, first_idx, nullptr, n, iface, d_off16, d_len, d_arena, arena_bytes);
}

int unet_synth_len_dev(unet_ctx* ctx, void* stream, int kind, uint64_t seed, uint64_t first_idx,
                       const uint32_t* d_rel_idx, uint32_t n, uint32_t* d_len) {
    if (!ctx) return fail(UNET_E_INVAL, "null context");
    if (!kind_ok(kind)) return fail(UNET_E_INVAL, "unknown trace kind %d", kind);
    if (n == 0) return UNET_OK;
    if (!d_len) return fail(UNET_E_INVAL, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_synth_len, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, pick(ctx, stream), kind, seed, first_idx,
                       d_rel_idx, n, d_len);
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

int unet_synth_shard_dev(unet_ctx* ctx, void* stream, int kind, uint64_t seed, uint64_t first_idx, uint32_t n,
                         const unet_iface* iface, uint32_t world, uint8_t* d_shard) {
    if (!ctx) return fail(UNET_E_INVAL, "null context");
    if (!kind_ok(kind)) return fail(UNET_E_INVAL, "unknown trace kind %d", kind);
    if (n == 0) return UNET_OK;
    if (!iface || !d_shard) return fail(UNET_E_INVAL, "null argument");
    if (world == 0 || world > 255) return fail(UNET_E_INVAL, "world size %u outside 1..255", world);
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_synth_shard, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, pick(ctx, stream), kind, seed, first_idx, n,
                       iface->unicast, iface->broadcast, world, d_shard);
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

/* ---------------------------------------------------------------- yardstick */

int unet_stream_read_dev(unet_ctx* ctx, void* stream, const uint8_t* d_buf, uint64_t bytes, uint32_t* d_sink) {
    if (!ctx || !d_buf || !d_sink) return fail(UNET_E_INVAL, "null argument");
    if (((uintptr_t)d_buf & 15u) != 0) return fail(UNET_E_INVAL, "buffer must be 16-byte aligned");
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_stream_read, dim3(8192), dim3(kBlock), 0, pick(ctx, stream), (const u32x4*)d_buf, bytes / 16, d_sink);
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

static int stream_read_pieces(unet_ctx* ctx, void* stream, const uint8_t* d_buf, uint64_t bytes, uint32_t* d_sink, u32 pad_lds) {
    if (!ctx || !d_buf || !d_sink) return fail(UNET_E_INVAL, "null argument");
    if (((uintptr_t)d_buf & 15u) != 0) return fail(UNET_E_INVAL, "buffer must be 16-byte aligned");
    const u64 nchunks = bytes / 16, per_block = (u64)kPieceChunks * (kBlock / kWave);
    const u64 grid = (nchunks + per_block - 1) / per_block;
    if (grid == 0) return UNET_OK;
    if (grid * kBlock >= (1ull << 32)) return fail(UNET_E_RANGE, "buffer of %llu bytes is too large for one launch", (unsigned long long)bytes);
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_stream_read_pieces, dim3((u32)grid), dim3(kBlock), pad_lds, pick(ctx, stream), (const u32x4*)d_buf, nchunks, d_sink);
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

int unet_stream_read_pieces_dev(unet_ctx* ctx, void* stream, const uint8_t* d_buf, uint64_t bytes, uint32_t* d_sink) {
    // unused LDS so that at most three workgroups share a CU's 160 KiB: fewer resident waves read faster on most boxes (see the kernel)
    return stream_read_pieces(ctx, stream, d_buf, bytes, d_sink, (160u << 10) / 3u - 1024u);
}

int unet_stream_read_pieces_full_dev(unet_ctx* ctx, void* stream, const uint8_t* d_buf, uint64_t bytes, uint32_t* d_sink) {
    return stream_read_pieces(ctx, stream, d_buf, bytes, d_sink, 0u);
}

int unet_stream_read_records_dev(unet_ctx* ctx, void* stream, const uint8_t* d_buf, uint64_t bytes, uint32_t record_bytes, uint32_t* d_sink) {
    if (!ctx || !d_buf || !d_sink) return fail(UNET_E_INVAL, "null argument");
    if (((uintptr_t)d_buf & 15u) != 0) return fail(UNET_E_INVAL, "buffer must be 16-byte aligned");
    // (rounded up to 16 bytes a record may not be longer than the longest record the kernels take, 65535 bytes)
    if (record_bytes == 0 || record_bytes > 65520u) return fail(UNET_E_RANGE, "record size %u: 1..65520 bytes", record_bytes);
    const u32 stride16 = (record_bytes + 15u) / 16u;
    const u64 nrec = bytes / 16 / stride16;
    if (nrec == 0) return UNET_OK;
    if (nrec > 0xffffffffull) return fail(UNET_E_RANGE, "buffer of %llu bytes holds more than 2^32 records of %u bytes", (unsigned long long)bytes, record_bytes);
    int rc = check_tiles((u32)nrec, 16);
    if (rc) return rc;
    // the records' lengths: one table of equal entries, kept in the context (this call is a measurement aid: it may allocate)
    if (ctx->d_len_fill.cap < (size_t)nrec * 4) ctx->len_fill_n = 0;
    if ((rc = ensure(ctx, ctx->d_len_fill, (size_t)nrec * 4))) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = pick(ctx, stream);
    if (ctx->len_fill_n < nrec || ctx->len_fill_v != stride16 * 16u) {  // (filled once per record size, not per call)
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)ctx->d_len_fill.p, (int)(stride16 * 16u), (size_t)nrec, s));
        ctx->len_fill_n = nrec;
        ctx->len_fill_v = stride16 * 16u;
    }
    Batch b{d_buf, bytes, nullptr, stride16, (const u32*)ctx->d_len_fill.p, (u32)nrec};
    // unused LDS so that four workgroups share a CU (measured best of 2..8: 6.87 TB/s against 6.75 with eight, 6.55 with three)
    const u32 pad = (160u << 10) / 4u - 1024u;
    hipLaunchKernelGGL(k_stream_read_records, dim3(tiles_grid((u32)nrec, 16)), dim3(kBlock), pad, s, b, d_sink);
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

int unet_debug_lds_violations(unet_ctx* ctx, uint32_t* out, int reset) {
    if (!ctx || !out) return fail(UNET_E_INVAL, "null argument");
#ifdef UNET_DEBUG
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned int v = 0;
    HIP_TRY(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_lds_out_of_area), sizeof v));
    *out = v;
    if (reset) {
        v = 0;
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_lds_out_of_area), &v, sizeof v));
    }
    return UNET_OK;
#else
    (void)reset;
    *out = 0;
    return fail(UNET_E_INVAL, "this library was not built with -DUNET_DEBUG: its kernels do not check their LDS stores");
#endif
}

/* ---------------------------------------------------------------- arena packing */
int unet_pack_size(const uint32_t* len, uint32_t n, uint64_t* arena_bytes) {
    if ((n && !len) || !arena_bytes) return fail(UNET_E_INVAL, "null argument");
    u64 slots = 0;
    for (u32 i = 0; i < n; i++) {
        if (len[i] > kMaxLen) return fail(UNET_E_RANGE, "record %u: length %u > 65535", i, len[i]);
        slots += ((u64)len[i] + 15u) >> 4;
    }
    if (slots > 0xffffffffull) return fail(UNET_E_RANGE, "more than 2^32 16-byte slots");
    *arena_bytes = slots ? slots * 16u : 16u;
    return UNET_OK;
}

int unet_pack_frames(const uint8_t* const* frames, const uint32_t* len, uint32_t n, uint8_t* arena, uint64_t arena_bytes,
                     uint32_t* off16) {
    if (!arena || !off16 || (n && (!frames || !len))) return fail(UNET_E_INVAL, "null argument");
    if (((uintptr_t)arena & 15u) != 0) return fail(UNET_E_INVAL, "arena must be 16-byte aligned");
    u64 need = 0;
    int rc = unet_pack_size(len, n, &need);
    if (rc) return rc;
    if (arena_bytes < need) return fail(UNET_E_RANGE, "arena of %llu bytes, need %llu", (unsigned long long)arena_bytes, (unsigned long long)need);
    u64 slot = 0;
    for (u32 i = 0; i < n; i++) {
        if (len[i] && !frames[i]) return fail(UNET_E_INVAL, "record %u: null buffer", i);
        const u64 padded = ((u64)len[i] + 15u) & ~15ull;
        uint8_t* dst = arena + slot * 16u;
        if (len[i]) memcpy(dst, frames[i], len[i]);
        memset(dst + len[i], 0, padded - len[i]);
        off16[i] = (u32)slot;
        slot += padded >> 4;
    }
    if (slot == 0) memset(arena, 0, 16);
    return UNET_OK;
}

int unet_pack_length_prefixed(const uint8_t* stream, uint64_t stream_bytes, uint8_t* arena, uint64_t* arena_bytes,
                              uint32_t* off16, uint32_t* len, uint32_t cap, uint32_t* n_out) {
    if ((!stream && stream_bytes) || !arena_bytes || !n_out) return fail(UNET_E_INVAL, "null argument");
    if (arena && (!off16 || !len)) return fail(UNET_E_INVAL, "null table");
    if (arena && ((uintptr_t)arena & 15u) != 0) return fail(UNET_E_INVAL, "arena must be 16-byte aligned");
    u64 pos = 0, slot = 0;
    u32 n = 0;
    while (pos < stream_bytes) {
        if (stream_bytes - pos < 4) return fail(UNET_E_INVAL, "record %u: truncated length prefix at byte %llu", n, (unsigned long long)pos);
        u32 l;
        memcpy(&l, stream + pos, 4);
        if (l > kMaxLen) return fail(UNET_E_RANGE, "record %u: length %u > 65535", n, l);
        if (stream_bytes - pos - 4 < l) return fail(UNET_E_INVAL, "record %u: %u bytes announced, %llu left", n, l, (unsigned long long)(stream_bytes - pos - 4));
        const u64 padded = ((u64)l + 15u) & ~15ull;
        if (arena) {
            if (n >= cap) return fail(UNET_E_RANGE, "more than %u records", cap);
            if ((slot * 16u) + padded > *arena_bytes) return fail(UNET_E_RANGE, "arena too small");
            uint8_t* dst = arena + slot * 16u;
            memcpy(dst, stream + pos + 4, l);
            memset(dst + l, 0, padded - l);
            off16[n] = (u32)slot;
            len[n] = l;
        }
        slot += padded >> 4;
        pos += 4ull + l;
        if (slot > 0xffffffffull || n == 0xffffffffu) return fail(UNET_E_RANGE, "stream too large");
        n++;
    }
    *n_out = n;
    if (!arena) *arena_bytes = slot ? slot * 16u : 16u;
    else if (slot == 0 && *arena_bytes >= 16) memset(arena, 0, 16);
    return UNET_OK;
}

/* ---------------------------------------------------------------- capture files */
int unet_pcap_scan(const uint8_t* file, uint64_t file_bytes, uint32_t max_len, unet_pcap_info* info) {
    if (!file || !info) return fail(UNET_E_INVAL, "null argument");
    if (max_len == 0 || max_len > 65535u) return fail(UNET_E_RANGE, "max_len %u outside 1..65535", max_len);
    unetpcap::Header h;
    if (const char* e = unetpcap::parse_header(file, file_bytes, &h)) return fail(UNET_E_INVAL, "%s", e);
    std::memset(info, 0, sizeof(*info));
    info->linktype = h.linktype;
    info->snaplen = h.snaplen;
    info->mode = (uint32_t)h.mode;
    info->skip = h.skip;
    info->nanosecond = h.nanos ? 1u : 0u;
    uint64_t slots = 0;
    uint32_t snapped = 0, clipped = 0;
    info->n_records = unetpcap::for_each_record(file, file_bytes, h, max_len, &info->trailing_bytes,
        [&](uint32_t, const uint8_t*, const uint8_t*, uint32_t kept, uint32_t caplen, uint32_t origlen) {
            slots += (kept + 15u) >> 4;
            snapped += caplen < origlen;
            clipped += (caplen > h.skip ? caplen - h.skip : 0u) > kept;
        });
    info->n_snapped = snapped;
    info->n_clipped = clipped;
    info->arena_bytes = slots ? slots * 16u : 16u;
    if (slots > 0xffffffffull) return fail(UNET_E_RANGE, "capture needs more than 2^32 16-byte slots");
    return UNET_OK;
}

int unet_pcap_pack(const uint8_t* file, uint64_t file_bytes, uint32_t max_len, uint8_t* arena, uint64_t arena_bytes,
                   uint32_t* off16, uint32_t* len, uint64_t* ts_ns, uint32_t n) {
    if (!file || !arena || !off16 || !len) return fail(UNET_E_INVAL, "null argument");
    if (((uintptr_t)arena & 15u) != 0) return fail(UNET_E_INVAL, "arena must be 16-byte aligned");
    unet_pcap_info info;
    int rc = unet_pcap_scan(file, file_bytes, max_len, &info);
    if (rc) return rc;
    if (n != info.n_records) return fail(UNET_E_INVAL, "n = %u but the capture holds %u records", n, info.n_records);
    if (arena_bytes < info.arena_bytes) return fail(UNET_E_RANGE, "arena of %llu bytes, need %llu", (unsigned long long)arena_bytes,
                                                    (unsigned long long)info.arena_bytes);
    unetpcap::Header h;
    unetpcap::parse_header(file, file_bytes, &h);
    uint64_t slot = 0;
    unetpcap::for_each_record(file, file_bytes, h, max_len, nullptr,
        [&](uint32_t i, const uint8_t* rh, const uint8_t* data, uint32_t kept, uint32_t, uint32_t) {
            const uint64_t padded = ((uint64_t)kept + 15u) & ~15ull;
            uint8_t* dst = arena + slot * 16u;
            std::memcpy(dst, data, kept);
            std::memset(dst + kept, 0, padded - kept);
            off16[i] = (uint32_t)slot;
            len[i] = kept;
            if (ts_ns) {
                const uint64_t sec = unetpcap::rd32(rh, h.swapped), frac = unetpcap::rd32(rh + 4, h.swapped);
                ts_ns[i] = sec * 1000000000ull + (h.nanos ? frac : frac * 1000ull);
            }
            slot += padded >> 4;
        });
    if (slot == 0) std::memset(arena, 0, 16);
    return UNET_OK;
}

int unet_pcap_write(const uint8_t* arena, uint64_t arena_bytes, const uint32_t* off16, uint32_t stride16,
                    const uint32_t* len, uint32_t n, uint32_t linktype, uint8_t* file, uint64_t file_cap,
                    uint64_t* file_bytes) {
    if (!file_bytes || (n && (!arena || !len))) return fail(UNET_E_INVAL, "null argument");
    if (n && !off16 && stride16 == 0) return fail(UNET_E_INVAL, "need off16 or a non-zero stride16");
    uint64_t need = 24;
    for (uint32_t i = 0; i < n; ++i) {
        if (len[i] > 65535u) return fail(UNET_E_RANGE, "record %u: length %u exceeds 65535", i, len[i]);
        const uint64_t at = (off16 ? (uint64_t)off16[i] : (uint64_t)i * stride16) * 16u;
        if (at + len[i] > arena_bytes) return fail(UNET_E_RANGE, "record %u lies outside the arena", i);
        if (len[i]) need += 16u + len[i];
    }
    *file_bytes = need;
    if (!file) return UNET_OK;
    if (file_cap < need) return fail(UNET_E_RANGE, "file buffer of %llu bytes, need %llu", (unsigned long long)file_cap, (unsigned long long)need);
    const uint32_t gh[6] = {unetpcap::kMagicUs, 2u | (4u << 16), 0u, 0u, 65535u, linktype};
    std::memcpy(file, gh, 24);
    uint8_t* p = file + 24;
    for (uint32_t i = 0; i < n; ++i) {
        if (!len[i]) continue;
        const uint64_t at = (off16 ? (uint64_t)off16[i] : (uint64_t)i * stride16) * 16u;
        const uint32_t rh[4] = {i / 1000000u, i % 1000000u, len[i], len[i]};
        std::memcpy(p, rh, 16);
        std::memcpy(p + 16, arena + at, len[i]);
        p += 16u + len[i];
    }
    return UNET_OK;
}

int unet_stream_copy_dev(unet_ctx* ctx, void* stream, const uint8_t* d_src, uint8_t* d_dst, uint64_t bytes) {
    if (!ctx || !d_src || !d_dst) return fail(UNET_E_INVAL, "null argument");
    if ((((uintptr_t)d_src | (uintptr_t)d_dst) & 15u) != 0) return fail(UNET_E_INVAL, "buffers must be 16-byte aligned");
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_stream_copy, dim3(4096), dim3(kBlock), 0, pick(ctx, stream), (const u32x4*)d_src, (u32x4*)d_dst, bytes / 16);
    HIP_TRY(hipGetLastError());
    return UNET_OK;
}

}  // extern "C"
