// How many workgroups per CU should a streaming kernel keep resident on MI355X?  A pure read (and a copy) of a
// pseudo-random buffer by short-lived waves that own one contiguous piece each, with the number of resident
// workgroups capped by a dynamic LDS allocation (160 KiB per CU / the cap).
// Build: hipcc --offload-arch=gfx950 -O3 tools/occupancy_probe.hip -o tools/occupancy_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ u32 sum4(u32x4 v, u32 a) {
    a = __builtin_amdgcn_sad_u16(v.x, 0u, a); a = __builtin_amdgcn_sad_u16(v.y, 0u, a);
    a = __builtin_amdgcn_sad_u16(v.z, 0u, a); a = __builtin_amdgcn_sad_u16(v.w, 0u, a); return a;
}

__global__ void k_fill_random(u64* buf, u64 nwords) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (u64)gridDim.x * blockDim.x) {
        u64 z = i * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        buf[i] = z ^ (z >> 31);
    }
}

extern __shared__ u32 dyn_lds[];

// one wave per piece of `rows` 1 KiB rows; PAIR rows requested ahead of the PAIR being summed
template <int PAIR>
__global__ __launch_bounds__(256) void k_read_pieces(const u32x4* __restrict__ buf, u64 nchunks, u32 rows, u32* sink) {
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const u64 lo = wave * rows * 64;
    if (lo + (u64)rows * 64 > nchunks) return;
    const u32x4* p = buf + lo + lane;
    u32 acc = 0;
    u32x4 v[PAIR];
#pragma unroll
    for (int k = 0; k < PAIR; k++) v[k] = __builtin_nontemporal_load(p + k * 64);
    for (u32 r = PAIR; r < rows + PAIR; r += PAIR) {
        u32x4 nx[PAIR];
#pragma unroll
        for (int k = 0; k < PAIR; k++) nx[k] = __builtin_nontemporal_load(p + (r + k < rows ? r + k : rows - 1) * 64);
#pragma unroll
        for (int k = 0; k < PAIR; k++) acc = sum4(v[k], acc);
#pragma unroll
        for (int k = 0; k < PAIR; k++) v[k] = nx[k];
    }
    if (acc == 0x12345678u) sink[0] = acc + dyn_lds[0];
}

// copy: one wave per piece of `rows` rows, PAIR rows loaded ahead
template <int PAIR>
__global__ __launch_bounds__(256) void k_copy_pieces(const u32x4* __restrict__ src, u32x4* __restrict__ dst, u64 nchunks, u32 rows, u32* sink) {
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const u64 lo = wave * rows * 64;
    if (lo + (u64)rows * 64 > nchunks) return;
    const u32x4* p = src + lo + lane;
    u32x4* q = dst + lo + lane;
    u32x4 v[PAIR];
#pragma unroll
    for (int k = 0; k < PAIR; k++) v[k] = __builtin_nontemporal_load(p + ((u32)k < rows ? (u32)k : rows - 1) * 64);  // never past the piece
    for (u32 r = 0; r < rows; r += PAIR) {
        u32x4 nx[PAIR];
#pragma unroll
        for (int k = 0; k < PAIR; k++) nx[k] = __builtin_nontemporal_load(p + (r + PAIR + k < rows ? r + PAIR + k : rows - 1) * 64);
#pragma unroll
        for (int k = 0; k < PAIR; k++) if (r + k < rows) __builtin_nontemporal_store(v[k], q + (r + k) * 64);
#pragma unroll
        for (int k = 0; k < PAIR; k++) v[k] = nx[k];
    }
    if (sink == (u32*)1) sink[0] = dyn_lds[0];
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
    const u64 bytes = 3ull << 30;
    u32x4 *s, *d; u32* sink;
    CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4096));
    hipLaunchKernelGGL(k_fill_random, dim3(65536), dim3(256), 0, 0, (u64*)s, bytes / 8);
    CK(hipDeviceSynchronize());
    const u64 n = bytes / 16;
    CK(hipFuncSetAttribute((const void*)k_read_pieces<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10));
    CK(hipFuncSetAttribute((const void*)k_read_pieces<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10));
    CK(hipFuncSetAttribute((const void*)k_copy_pieces<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10));
    CK(hipFuncSetAttribute((const void*)k_copy_pieces<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 << 10));
    printf("%.2f GB of pseudo-random bytes; workgroups of 256 threads; GB/s = bytes read (read) or read + written (copy)\n", bytes / 1e9);
    for (u32 rows : {8u, 24u}) {
        for (int cap : {8, 6, 5, 4, 3, 2}) {
            const u32 lds = cap == 8 ? 0u : (u32)((160u << 10) / cap) - 512u;
            const u32 grid = (u32)(n / (4ull * rows * 64));
            double a = timeit([&] { hipLaunchKernelGGL(k_read_pieces<2>, dim3(grid), dim3(256), lds, 0, s, n, rows, sink); }, 8);
            double b = timeit([&] { hipLaunchKernelGGL(k_read_pieces<4>, dim3(grid), dim3(256), lds, 0, s, n, rows, sink); }, 8);
            printf("read  %2u KiB per wave, at most %d workgroups per CU: 2 rows ahead %7.1f GB/s   4 rows ahead %7.1f GB/s\n", rows, cap,
                   bytes / a / 1e6, bytes / b / 1e6);
            fflush(stdout);
        }
    }
    for (u32 rows : {1u, 2u, 8u, 24u}) {
        for (int cap : {8, 6, 4, 3, 2}) {
            const u32 lds = cap == 8 ? 0u : (u32)((160u << 10) / cap) - 512u;
            const u32 grid = (u32)(n / (4ull * rows * 64));
            double a = timeit([&] { hipLaunchKernelGGL(k_copy_pieces<1>, dim3(grid), dim3(256), lds, 0, s, d, n, rows, sink); }, 8);
            double b = timeit([&] { hipLaunchKernelGGL(k_copy_pieces<2>, dim3(grid), dim3(256), lds, 0, s, d, n, rows, sink); }, 8);
            printf("copy  %2u KiB per wave, at most %d workgroups per CU: 1 row ahead %7.1f GB/s   2 rows ahead %7.1f GB/s\n", rows, cap,
                   2.0 * bytes / a / 1e6, 2.0 * bytes / b / 1e6);
            fflush(stdout);
        }
    }
    return 0;
}
