// Cache-policy probe for gfx950: streaming reads and the "record" pattern of k_verify (1504-byte records,
// a 1 KiB row and a 480-byte row per record) with every combination of the sc0 / nt / sc1 bits, through
// raw buffer loads (the compiler tracks those, unlike inline-asm loads).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, u32 bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);  // raw buffer, DATA_FORMAT 32
}

template <int AUX>
__device__ __forceinline__ u32x4 bload(rsrc_t rsrc, u32 voff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, AUX));
}

__device__ __forceinline__ u32 sum4(u32x4 v, u32 a) { return a + v.x + v.y + v.z + v.w; }

// slab streaming read: each block owns a contiguous slab, 4 loads in flight per lane
template <int AUX>
__global__ __launch_bounds__(256) void k_slab(const uint8_t* buf, u64 nchunks, u32* sink) {
    const u64 per = (nchunks + gridDim.x - 1) / gridDim.x;
    const u64 lo = (u64)blockIdx.x * per;
    const u64 hi = lo + per < nchunks ? lo + per : nchunks;
    const rsrc_t rsrc = make_rsrc(buf + lo * 16, (u32)((hi - lo) * 16));
    u32 acc = 0;
    u32 i = threadIdx.x;
    const u32 n = (u32)(hi - lo);
    for (; i + 768 < n; i += 1024) {
        u32x4 a = bload<AUX>(rsrc, i * 16), b = bload<AUX>(rsrc, (i + 256) * 16), c = bload<AUX>(rsrc, (i + 512) * 16), d = bload<AUX>(rsrc, (i + 768) * 16);
        acc = sum4(a, acc); acc = sum4(b, acc); acc = sum4(c, acc); acc = sum4(d, acc);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// record pattern: one wave per 16 records of 94 chunks; row 0 with policy A, row 1 with policy B; the loads
// of the next record are issued before the current one is summed
template <int A, int B>
__global__ __launch_bounds__(256) void k_records(const uint8_t* buf, u32 nrec, u32* sink) {
    const u32 lane = threadIdx.x & 63u, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const u32 first = wave * 16;
    if (first >= nrec) return;
    u32 acc = 0;
    rsrc_t rs = make_rsrc(buf + (u64)first * 1504, 1504);
    u32x4 r0 = bload<A>(rs, lane * 16), r1 = bload<B>(rs, (64 + (lane < 30 ? lane : 29)) * 16);
    for (u32 j = 1; j <= 16; ++j) {
        const u32 nx = first + (j < 16 ? j : 15);
        const rsrc_t rn = make_rsrc(buf + (u64)nx * 1504, 1504);
        const u32x4 n0 = bload<A>(rn, lane * 16), n1 = bload<B>(rn, (64 + (lane < 30 ? lane : 29)) * 16);
        acc = sum4(r0, acc);
        acc = sum4(r1, acc);
        r0 = n0;
        r1 = n1;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <class F>
double timeit(F f, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; i++) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main() {
    const u64 bytes = 6308233216ull;  // 4,194,304 records of 1504 bytes
    uint8_t* d; u32* sink;
    CK(hipMalloc(&d, bytes + 4096)); CK(hipMalloc(&sink, 4096));
    CK(hipMemset(d, 1, bytes + 4096));
    const u64 nch = bytes / 16;
    const u32 nrec = 4194304;
#define SLAB(AUX) { double ms = timeit([&] { hipLaunchKernelGGL(k_slab<AUX>, dim3(8192), dim3(256), 0, 0, d, nch, sink); }, 10); \
        printf("slab    aux %2d            : %7.3f ms %7.1f GB/s\n", AUX, ms, bytes / ms / 1e6); fflush(stdout); }
    SLAB(0) SLAB(1) SLAB(2) SLAB(3) SLAB(16) SLAB(17) SLAB(18) SLAB(19)
#define REC(A, B) { double ms = timeit([&] { hipLaunchKernelGGL((k_records<A, B>), dim3(nrec / 64), dim3(256), 0, 0, d, nrec, sink); }, 10); \
        printf("records row0 aux %2d row1 aux %2d : %7.3f ms %7.1f GB/s\n", A, B, ms, (double)nrec * 1504 / ms / 1e6); fflush(stdout); }
    REC(0, 0) REC(2, 2) REC(2, 0) REC(2, 1) REC(2, 16) REC(2, 17) REC(18, 0) REC(18, 16) REC(3, 0) REC(16, 16) REC(1, 1) REC(18, 18) REC(2, 3) REC(3, 3) REC(19, 19) REC(2, 18)
    return 0;
}
