// Read-bandwidth probe for MI355X: what does a pure 16-byte-per-lane streaming read reach, and which
// launch shape gets there?  Build: hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o tools/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ u32 sum4(u32x4 v, u32 a) {
    a = __builtin_amdgcn_sad_u16(v.x, 0u, a); a = __builtin_amdgcn_sad_u16(v.y, 0u, a);
    a = __builtin_amdgcn_sad_u16(v.z, 0u, a); a = __builtin_amdgcn_sad_u16(v.w, 0u, a); return a;
}

// grid-stride, U loads in flight per lane
template <int U, bool NT>
__global__ void k_gs(const u32x4* __restrict__ buf, u64 n, u32* sink) {
    u32 acc = 0;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(&buf[i + k * stride]) : buf[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; k++) acc = sum4(v[k], acc);
    }
    for (; i < n; i += stride) acc = sum4(buf[i], acc);
    if (acc == 0x12345678u) sink[0] = acc;
}

// each block owns a contiguous slab; U loads in flight per lane, consecutive 4 KiB*... pieces
template <int U, bool NT>
__global__ void k_slab(const u32x4* __restrict__ buf, u64 n, u32* sink) {
    u32 acc = 0;
    const u64 per = (n + gridDim.x - 1) / gridDim.x;
    const u64 lo = (u64)blockIdx.x * per;
    const u64 hi = lo + per < n ? lo + per : n;
    u64 i = lo + threadIdx.x;
    const u64 bs = blockDim.x;
    for (; i + (U - 1) * bs < hi; i += U * bs) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(&buf[i + k * bs]) : buf[i + k * bs];
#pragma unroll
        for (int k = 0; k < U; k++) acc = sum4(v[k], acc);
    }
    for (; i < hi; i += bs) acc = sum4(buf[i], acc);
    if (acc == 0x12345678u) sink[0] = acc;
}

// pseudo-random contents (splitmix64 of the word index): does what is read change how fast it is read?
__global__ void k_fill_random(u64* buf, u64 nwords) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (u64)gridDim.x * blockDim.x) {
        u64 z = i * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        buf[i] = z ^ (z >> 31);
    }
}

template <typename F>
double timeit(F f, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main(int argc, char** argv) {
    const u64 bytes = argc > 1 ? strtoull(argv[1], 0, 10) : 6308233216ull;
    void* d; u32* sink;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4096));
    CK(hipMemset(d, 0x5a, bytes));
    const u64 n = bytes / 16;
    const bool random_fill = argc > 2 && argv[2][0] == 'r';
    if (random_fill) hipLaunchKernelGGL(k_fill_random, dim3(65536), dim3(256), 0, 0, (u64*)d, bytes / 8);
    CK(hipDeviceSynchronize());
    printf("buffer %.2f GB, %s\n", bytes / 1e9, random_fill ? "pseudo-random contents" : "every byte 0x5a");
#define RUN(name, K, grid, block) { double ms = timeit([&] { hipLaunchKernelGGL(K, dim3(grid), dim3(block), 0, 0, (const u32x4*)d, n, sink); }, 10); \
        printf("%-28s grid %6d block %4d : %7.3f ms  %7.1f GB/s\n", name, grid, block, ms, bytes / ms / 1e6); fflush(stdout); }
    const bool brief = argc > 3;
    if (brief) {
        for (int grid : {4096, 8192}) {
            RUN("gs U4 nt", (k_gs<4, true>), grid, 256);
            RUN("slab U4", (k_slab<4, false>), grid, 256);
            RUN("slab U4 nt", (k_slab<4, true>), grid, 256);
        }
        return 0;
    }
    for (int block : {256, 512, 1024}) {
        for (int grid : {1024, 2048, 4096, 8192, 16384, 65536}) {
            if ((long)grid * block > 16384L * 1024) continue;
            RUN("gs U4", (k_gs<4, false>), grid, block);
        }
    }
    for (int grid : {2048, 4096, 8192}) {
        RUN("gs U8", (k_gs<8, false>), grid, 256);
        RUN("gs U2", (k_gs<2, false>), grid, 256);
        RUN("gs U1", (k_gs<1, false>), grid, 256);
        RUN("gs U4 nt", (k_gs<4, true>), grid, 256);
        RUN("gs U8 nt", (k_gs<8, true>), grid, 256);
        RUN("slab U4", (k_slab<4, false>), grid, 256);
        RUN("slab U8", (k_slab<8, false>), grid, 256);
        RUN("slab U4 nt", (k_slab<4, true>), grid, 256);
    }
    for (int grid : {256, 512, 768, 1024, 1280, 1536, 1792, 2048}) RUN("slab U8 b1024", (k_slab<8, false>), grid, 1024);
    return 0;
}
