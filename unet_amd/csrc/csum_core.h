// Device-side building blocks for the gfx950 checksum / parse kernels.
//
// Arithmetic convention used by every kernel in this directory
// ------------------------------------------------------------
// The reference sums big-endian 16-bit words (src/utils.rs:3-12).  The kernels load the
// same bytes as little-endian 32-bit words and add their two 16-bit halves with one
// v_sad_u16 per dword (|lo-0| + |hi-0| + acc).  That is the sum of the BYTE-SWAPPED words.
// End-around-carry addition commutes with a byte swap of every operand (swapping is a
// rotation by 8 bits, i.e. multiplication by 256 modulo 65535, and fold() maps 0 to 0 and
// any positive sum to the unique representative in [1, 65535]), so
//     fold(sum of BE words) == bswap16(fold(sum of swapped words))
// holds bit for bit, including the 0x0000 / 0xffff distinction.  Sums are carried as
// EXACT u32 integers until the final fold: a 65535-byte record gives at most
// 32768 * 65535 < 2^31, the same headroom the reference's u32 accumulator has.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace unetdev {

typedef uint32_t u32;
typedef uint64_t u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr u32 kWave = 64;

// lo16(x) + hi16(x) + acc in one VALU instruction.
__device__ __forceinline__ u32 sad16(u32 x, u32 acc) { return __builtin_amdgcn_sad_u16(x, 0u, acc); }

__device__ __forceinline__ u32 sum_chunk(u32x4 v, u32 acc) {
    acc = sad16(v.x, acc);
    acc = sad16(v.y, acc);
    acc = sad16(v.z, acc);
    acc = sad16(v.w, acc);
    return acc;
}

__device__ __forceinline__ u32 halves(u32 x) { return (x & 0xffffu) + (x >> 16); }

// src/utils.rs:13-15 for a u32: two steps are enough (first leaves at most 0x1fffe).
__device__ __forceinline__ u32 fold16(u32 s) {
    s = (s & 0xffffu) + (s >> 16);
    s = (s & 0xffffu) + (s >> 16);
    return s;
}
__device__ __forceinline__ u32 bswap16(u32 x) { return ((x & 0xffu) << 8) | ((x >> 8) & 0xffu); }
__device__ __forceinline__ u32 bswap32(u32 x) { return __builtin_bswap32(x); }

// Exact sum of swapped words -> the u16 the reference returns (src/utils.rs:16).
__device__ __forceinline__ u32 finish(u32 sum_le) { return (~bswap16(fold16(sum_le))) & 0xffffu; }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 dpp_add(u32 v) {
    return v + (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}

// Sum over the 64 lanes of a wave, returned wave-uniform.  Needs all 64 lanes active.
// Hillis-Steele inside each 16-lane row, then the two gfx9 row broadcasts.
__device__ __forceinline__ u32 wave_sum(u32 v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}

// Minimum / maximum over the 64 lanes of a wave, returned wave-uniform.  Same DPP ladder as wave_sum; a lane whose
// DPP source does not exist keeps its own value (`old` operand), which is harmless for min and max.
template <bool MAX>
__device__ __forceinline__ u32 wave_minmax(u32 v) {
#define UNET_MM_STEP(CTRL, ROWS)                                                                      \
    {                                                                                                 \
        const u32 o_ = (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWS, 0xf, false);      \
        v = MAX ? (o_ > v ? o_ : v) : (o_ < v ? o_ : v);                                              \
    }
    UNET_MM_STEP(0x111, 0xf) UNET_MM_STEP(0x112, 0xf) UNET_MM_STEP(0x114, 0xf) UNET_MM_STEP(0x118, 0xf)
    UNET_MM_STEP(0x142, 0xa) UNET_MM_STEP(0x143, 0xc)
#undef UNET_MM_STEP
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ u32 rdlane(u32 v, u32 lane) {
    return (u32)__builtin_amdgcn_readlane((int)v, (int)lane);
}

// Mask keeping the first `vb` bytes (0..4) of a little-endian dword.
__device__ __forceinline__ u32 keep_bytes(int vb) {
    return vb >= 4 ? 0xffffffffu : (vb <= 0 ? 0u : ((1u << (8 * vb)) - 1u));
}

// Receiving-interface constants in the form the kernels compare against.
struct IfaceDev {
    u32 unicast;    // Ipv4Address value
    u32 broadcast;  // Ipv4Address value
    u32 mac_lo;     // MAC bytes 0..3 as a little-endian dword (frame dword 0)
    u32 mac_hi;     // MAC bytes 4..5 (low half of frame dword 1)
    u32 ext_tcp;
};

// Packed per-frame result: csum | verdict << 16 | proto << 24 (the 4-byte unet_result).
__device__ __forceinline__ u32 pack_result(u32 verdict, u32 proto, u32 csum) {
    return (csum & 0xffffu) | (verdict << 16) | (proto << 24);
}

// splitmix64 finalizer, the synthetic traces' only source of randomness.
__host__ __device__ __forceinline__ u64 fin64(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace unetdev
