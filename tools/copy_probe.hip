// Read+write bandwidth probe for MI355X: what does a plain 16-byte-per-lane device-to-device copy reach, with
// which launch shape, how many loads in flight and which cache-policy bits on the loads and on the stores?
// It is the ceiling k_tx_build (payload arena -> packet arena) is measured against.
// Build: hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip -o tools/copy_probe.bin
// usage: copy_probe.bin [bytes per direction]   (default 3 GiB read + 3 GiB written per launch)
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
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// AUX bits: 1 = sc0, 2 = nt, 16 = sc1
template <int AUX>
__device__ __forceinline__ u32x4 bload(rsrc_t r, u32 voff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, AUX));
}
template <int AUX>
__device__ __forceinline__ void bstore(rsrc_t r, u32 voff, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, (int)voff, 0, AUX);
}

// Every workgroup owns one contiguous slab (start on a 4 KiB boundary); U loads in flight per lane, then U stores.
template <int U, int LD, int ST>
__global__ __launch_bounds__(1024) void k_copy_slab(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, u64 nchunks) {
    const u64 per = ((nchunks + gridDim.x - 1) / gridDim.x + 255u) & ~255ull;
    const u64 lo = (u64)blockIdx.x * per;
    if (lo >= nchunks) return;
    const u64 hi = lo + per < nchunks ? lo + per : nchunks;
    const u32 n = (u32)(hi - lo), bs = blockDim.x;
    const rsrc_t rs = make_rsrc(src + lo * 16, n * 16), rd = make_rsrc(dst + lo * 16, n * 16);
    u32 i = threadIdx.x;
    for (; i + (U - 1) * bs < n; i += U * bs) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = bload<LD>(rs, (i + k * bs) * 16);
#pragma unroll
        for (int k = 0; k < U; k++) bstore<ST>(rd, (i + k * bs) * 16, v[k]);
    }
    for (; i < n; i += bs) bstore<ST>(rd, i * 16, bload<LD>(rs, i * 16));
}

// Grid-stride: the whole grid sweeps one window.
template <int U, int LD, int ST>
__global__ __launch_bounds__(1024) void k_copy_gs(const u32x4* __restrict__ src, u32x4* __restrict__ dst, u64 n) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = (LD & 2) ? __builtin_nontemporal_load(&src[i + k * stride]) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; k++) {
            if (ST & 2) __builtin_nontemporal_store(v[k], &dst[i + k * stride]);
            else dst[i + k * stride] = v[k];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// One wave owns one 4 KiB piece at a time (what a packet builder does: a wave per packet), pieces dealt out
// consecutively to the waves of the grid.
template <int LD, int ST>
__global__ __launch_bounds__(256) void k_copy_wave4k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, u64 n) {
    const u32 lane = threadIdx.x & 63u;
    const u64 waves = (u64)gridDim.x * 4u;
    const u64 pieces = n / 256u;
    for (u64 p = (u64)blockIdx.x * 4u + (threadIdx.x >> 6); p < pieces; p += waves) {
        const u32x4* s = src + p * 256u + lane;
        u32x4* d = dst + p * 256u + lane;
        u32x4 a, b, c, e;
        if (LD & 2) { a = __builtin_nontemporal_load(s); b = __builtin_nontemporal_load(s + 64); c = __builtin_nontemporal_load(s + 128); e = __builtin_nontemporal_load(s + 192); }
        else { a = s[0]; b = s[64]; c = s[128]; e = s[192]; }
        if (ST & 2) { __builtin_nontemporal_store(a, d); __builtin_nontemporal_store(b, d + 64); __builtin_nontemporal_store(c, d + 128); __builtin_nontemporal_store(e, d + 192); }
        else { d[0] = a; d[64] = b; d[128] = c; d[192] = e; }
    }
}

// One piece of P KiB per wave and nothing else: grid = pieces / (block / 64).  All loads, then all stores.
template <int P, int LD, int ST>
__global__ __launch_bounds__(256) void k_copy_piece(const u32x4* __restrict__ src, u32x4* __restrict__ dst, u64 n) {
    const u32 lane = threadIdx.x & 63u;
    const u64 piece = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const u64 at = piece * (u64)(P * 64) + lane;
    if (at - lane + P * 64 > n) return;
    u32x4 v[P];
#pragma unroll
    for (int k = 0; k < P; k++) v[k] = (LD & 2) ? __builtin_nontemporal_load(src + at + k * 64) : src[at + k * 64];
#pragma unroll
    for (int k = 0; k < P; k++) {
        if (ST & 2) __builtin_nontemporal_store(v[k], dst + at + k * 64);
        else dst[at + k * 64] = v[k];
    }
}

// Piece of P KiB per wave, loop of `reps` pieces per wave with the next piece's loads issued before the current
// piece's stores (two register sets): what a pipelined packet builder would do.
template <int P, int LD, int ST>
__global__ __launch_bounds__(256) void k_copy_piece_pipe(const u32x4* __restrict__ src, u32x4* __restrict__ dst, u64 n, u32 reps) {
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    u64 at = wave * reps * (u64)(P * 64) + lane;
    u32x4 a[P], b[P];
    if (at - lane + P * 64 > n) return;
#pragma unroll
    for (int k = 0; k < P; k++) a[k] = (LD & 2) ? __builtin_nontemporal_load(src + at + k * 64) : src[at + k * 64];
    for (u32 r = 0; r < reps; r++) {
        const u64 nx = at + P * 64;
        const bool more = r + 1 < reps && nx - lane + P * 64 <= n;
        if (more) {
#pragma unroll
            for (int k = 0; k < P; k++) b[k] = (LD & 2) ? __builtin_nontemporal_load(src + nx + k * 64) : src[nx + k * 64];
        }
#pragma unroll
        for (int k = 0; k < P; k++) {
            if (ST & 2) __builtin_nontemporal_store(a[k], dst + at + k * 64);
            else dst[at + k * 64] = a[k];
        }
        if (!more) break;
#pragma unroll
        for (int k = 0; k < P; k++) a[k] = b[k];
        at = nx;
    }
}

// One table-described record per wave, the way a packet builder meets its input: wave w reads off[w], len[w]
// and out_off[w] (wave-uniform, so scalar loads), copies the record's chunks with two clamped 1 KiB row loads
// and stores them 28 bytes further (header room), i.e. through a one-lane shift like the real kernel.
template <bool TABLES>
__global__ __launch_bounds__(256) void k_copy_record_wave(const u32x4* __restrict__ src, u32x4* __restrict__ dst, const u32* __restrict__ off,
                                                          const u32* __restrict__ len, const u32* __restrict__ out_off, u32 n,
                                                          u32 in_stride, u32 out_stride) {
    const u32 lane = threadIdx.x & 63u;
    const u32 w = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (w >= n) return;
    const u32 l = TABLES ? len[w] : 1500u, nch = (l + 15u) >> 4, last = nch - 1u;
    const u32x4* p = src + (TABLES ? off[w] : w * in_stride);
    u32x4* q = dst + (TABLES ? out_off[w] : w * out_stride);
    const u32x4 a = __builtin_nontemporal_load(p + min(lane, last));
    const u32x4 b = __builtin_nontemporal_load(p + min(lane + 64u, last));
    if (lane < nch) __builtin_nontemporal_store(a, q + lane);
    if (lane + 64u < nch) __builtin_nontemporal_store(b, q + lane + 64u);
}

template <int ST>
__global__ __launch_bounds__(256) void k_fill(uint8_t* __restrict__ dst, u64 nchunks) {
    const u64 per = ((nchunks + gridDim.x - 1) / gridDim.x + 255u) & ~255ull;
    const u64 lo = (u64)blockIdx.x * per;
    if (lo >= nchunks) return;
    const u64 hi = lo + per < nchunks ? lo + per : nchunks;
    const u32 n = (u32)(hi - lo);
    const rsrc_t rd = make_rsrc(dst + lo * 16, n * 16);
    const u32x4 v = {blockIdx.x, threadIdx.x, 3u, 4u};
    for (u32 i = threadIdx.x; i < n; i += 256) bstore<ST>(rd, i * 16, v);
}

template <typename F>
double timeit(F f, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; i++) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms / iters;
}

__global__ void k_fill_random(u64* buf, u64 nwords) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (u64)gridDim.x * blockDim.x) {
        u64 z = i * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        buf[i] = z ^ (z >> 31);
    }
}

int main(int argc, char** argv) {
    const u64 bytes = argc > 1 ? strtoull(argv[1], 0, 10) : (3ull << 30);
    uint8_t *s, *d;
    CK(hipMalloc(&s, bytes));
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(s, 0x5a, bytes));
    CK(hipMemset(d, 0, bytes));
    const u64 n = bytes / 16;
    // argv[2] = "random": the source holds pseudo-random bytes instead of one repeated byte (what is moved changes
    // how fast it moves on this part: fewer bits toggle on a constant buffer); argv[3] present: the shapes that
    // matter only (pieces per wave, 1500-byte records, the library copy)
    const bool random_fill = argc > 2 && argv[2][0] == 'r', brief = argc > 3;
    if (random_fill) hipLaunchKernelGGL(k_fill_random, dim3(65536), dim3(256), 0, 0, (u64*)s, bytes / 8);
    CK(hipDeviceSynchronize());
    printf("copy %.2f GB -> %.2f GB per launch (%s); GB/s counts read + written bytes\n", bytes / 1e9, bytes / 1e9,
           random_fill ? "pseudo-random contents" : "every byte 0x5a");
#define REPORT(name, grid, block, ms) { printf("%-34s grid %6d block %4d : %7.3f ms  %7.1f GB/s\n", name, grid, block, ms, 2.0 * bytes / (ms) / 1e6); fflush(stdout); }
#define SLAB(U, LD, ST, grid, block) { double ms = timeit([&] { hipLaunchKernelGGL((k_copy_slab<U, LD, ST>), dim3(grid), dim3(block), 0, 0, s, d, n); }, 6); \
        REPORT("slab U" #U " ld" #LD " st" #ST, grid, block, ms); }
#define GS(U, LD, ST, grid, block) { double ms = timeit([&] { hipLaunchKernelGGL((k_copy_gs<U, LD, ST>), dim3(grid), dim3(block), 0, 0, (const u32x4*)s, (u32x4*)d, n); }, 6); \
        REPORT("gridstride U" #U " ld" #LD " st" #ST, grid, block, ms); }
    {
        double ms = timeit([&] { CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0)); }, 6);
        REPORT("hipMemcpy D2D", 0, 0, ms);
    }
    if (!brief) {
    // cache-policy bits on the loads and stores, fixed shape
    SLAB(4, 0, 0, 4096, 256) SLAB(4, 2, 0, 4096, 256) SLAB(4, 0, 2, 4096, 256) SLAB(4, 2, 2, 4096, 256)
    SLAB(4, 2, 1, 4096, 256) SLAB(4, 2, 3, 4096, 256) SLAB(4, 2, 16, 4096, 256) SLAB(4, 2, 17, 4096, 256)
    SLAB(4, 2, 18, 4096, 256) SLAB(4, 2, 19, 4096, 256) SLAB(4, 18, 2, 4096, 256) SLAB(4, 3, 2, 4096, 256)
    // loads in flight
    SLAB(1, 2, 2, 4096, 256) SLAB(2, 2, 2, 4096, 256) SLAB(8, 2, 2, 4096, 256) SLAB(16, 2, 2, 4096, 256)
    SLAB(1, 2, 0, 4096, 256) SLAB(2, 2, 0, 4096, 256) SLAB(8, 2, 0, 4096, 256)
    // launch shape
    for (int grid : {256, 512, 1024, 2048, 8192, 16384, 65536}) { SLAB(4, 2, 2, grid, 256) }
    for (int grid : {256, 512, 1024, 2048, 4096}) { SLAB(4, 2, 2, grid, 512) SLAB(4, 2, 2, grid, 1024) SLAB(2, 2, 2, grid, 1024) }
    for (int grid : {1024, 2048, 4096, 8192, 16384}) { GS(4, 2, 2, grid, 256) GS(4, 0, 0, grid, 256) GS(2, 2, 2, grid, 256) GS(1, 2, 2, grid, 256) GS(4, 2, 0, grid, 256) }
    for (int grid : {1024, 4096, 16384, 65536}) {
        double ms = timeit([&] { hipLaunchKernelGGL((k_copy_wave4k<2, 2>), dim3(grid), dim3(256), 0, 0, (const u32x4*)s, (u32x4*)d, n); }, 6);
        REPORT("wave per 4 KiB piece ld2 st2", grid, 256, ms);
        ms = timeit([&] { hipLaunchKernelGGL((k_copy_wave4k<2, 0>), dim3(grid), dim3(256), 0, 0, (const u32x4*)s, (u32x4*)d, n); }, 6);
        REPORT("wave per 4 KiB piece ld2 st0", grid, 256, ms);
        ms = timeit([&] { hipLaunchKernelGGL((k_copy_wave4k<0, 0>), dim3(grid), dim3(256), 0, 0, (const u32x4*)s, (u32x4*)d, n); }, 6);
        REPORT("wave per 4 KiB piece ld0 st0", grid, 256, ms);
    }
    } else { SLAB(4, 2, 2, 4096, 256) SLAB(4, 2, 2, 65536, 256) SLAB(4, 2, 2, 1024, 1024) }
#define PIECE(P, LD, ST, block) { const u64 pieces = n / (P * 64); const int grid = (int)((pieces + (block / 64) - 1) / (block / 64)); \
        double ms = timeit([&] { hipLaunchKernelGGL((k_copy_piece<P, LD, ST>), dim3(grid), dim3(block), 0, 0, (const u32x4*)s, (u32x4*)d, n); }, 6); \
        REPORT("one " #P " KiB piece per wave ld" #LD " st" #ST, grid, block, ms); }
    PIECE(1, 2, 2, 256) PIECE(2, 2, 2, 256) PIECE(4, 2, 2, 256) PIECE(8, 2, 2, 256) PIECE(16, 2, 2, 256)
    PIECE(1, 2, 2, 64) PIECE(4, 2, 2, 64) PIECE(8, 2, 2, 64)
    PIECE(4, 0, 0, 256) PIECE(4, 2, 0, 256) PIECE(4, 0, 2, 256) PIECE(8, 0, 0, 256) PIECE(8, 2, 0, 256)
#define PIPE(P, LD, ST, reps) { const u64 pieces = n / (P * 64); const int grid = (int)((pieces / reps + 3) / 4); \
        double ms = timeit([&] { hipLaunchKernelGGL((k_copy_piece_pipe<P, LD, ST>), dim3(grid), dim3(256), 0, 0, (const u32x4*)s, (u32x4*)d, n, (u32)reps); }, 6); \
        REPORT("pipelined " #P " KiB pieces x" #reps " ld" #LD " st" #ST, grid, 256, ms); }
    PIPE(2, 2, 2, 4) PIPE(2, 2, 2, 16) PIPE(4, 2, 2, 2) PIPE(4, 2, 2, 4) PIPE(4, 2, 2, 8) PIPE(4, 2, 2, 16) PIPE(8, 2, 2, 4) PIPE(8, 2, 2, 16)
    PIPE(4, 2, 0, 4) PIPE(4, 0, 0, 4)
    {
        // 1500-byte records described by tables or by a fixed stride; stride 94 chunks = packed (rows start on any
        // 16-byte boundary), 96 = every record starts on a 128-byte line
        for (u32 is : {94u, 96u}) for (u32 os : {94u, 96u}) {
            const u32 nrec = (u32)(n / 96);
            u32 *h = (u32*)malloc(3ull * nrec * 4), *dt;
            for (u32 i = 0; i < nrec; i++) { h[i] = i * is; h[nrec + i] = 1500u; h[2 * nrec + i] = i * os; }
            CK(hipMalloc(&dt, 3ull * nrec * 4));
            CK(hipMemcpy(dt, h, 3ull * nrec * 4, hipMemcpyHostToDevice));
            const int block = 256, grid = (int)((nrec + 3) / 4);
            double ms = timeit([&] { hipLaunchKernelGGL(k_copy_record_wave<true>, dim3(grid), dim3(block), 0, 0, (const u32x4*)s, (u32x4*)d, dt, dt + nrec, dt + 2 * nrec, nrec, is, os); }, 6);
            printf("1500-B record per wave, tables, in stride %u out stride %u : %7.3f ms  %7.1f GB/s\n", is, os, ms, 2.0 * nrec * 1504.0 / ms / 1e6);
            ms = timeit([&] { hipLaunchKernelGGL(k_copy_record_wave<false>, dim3(grid), dim3(block), 0, 0, (const u32x4*)s, (u32x4*)d, dt, dt + nrec, dt + 2 * nrec, nrec, is, os); }, 6);
            printf("1500-B record per wave, no tables, in stride %u out stride %u : %7.3f ms  %7.1f GB/s\n", is, os, ms, 2.0 * nrec * 1504.0 / ms / 1e6);
            CK(hipFree(dt));
            free(h);
        }
    }
    // write only (GB/s here = written bytes * 2 / time, halve it)
    {
        double ms = timeit([&] { hipLaunchKernelGGL((k_fill<0>), dim3(4096), dim3(256), 0, 0, d, n); }, 6);
        printf("fill st0 : %7.3f ms  %7.1f GB/s written\n", ms, bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((k_fill<2>), dim3(4096), dim3(256), 0, 0, d, n); }, 6);
        printf("fill st2 : %7.3f ms  %7.1f GB/s written\n", ms, bytes / ms / 1e6);
        ms = timeit([&] { CK(hipMemsetAsync(d, 0, bytes, 0)); }, 6);
        printf("hipMemset : %7.3f ms  %7.1f GB/s written\n", ms, bytes / ms / 1e6);
    }
    return 0;
}
