// Does the LDS of gfx950 serve reads at addresses that are not multiples of the access size?  Every lane reads 4 and
// 16 bytes at byte offset `lane` of an LDS array holding the byte pattern 0, 1, 2 ... and the host checks each byte.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_unaligned_probe.hip -o tools/lds_unaligned_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_probe(u32* out32, u32x4* out128, u32x4* outc) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[1024];
    for (u32 i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = (unsigned char)(i * 7u + 3u);
    __syncthreads();
    const u32 addr = (u32)(size_t)(lds) + threadIdx.x;  // LDS byte address
    u32 a;
    u32x4 b;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr) : "memory");
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(addr) : "memory");
    out32[threadIdx.x] = a;
    out128[threadIdx.x] = b;
    // what the compiler makes of an under-aligned 16-byte read
    u32x4 c;
    __builtin_memcpy(&c, lds + threadIdx.x, 16);
    outc[threadIdx.x] = c;
}

int main() {
    u32 *d32; u32x4 *d128, *dc;
    CK(hipMalloc(&d32, 64 * 4)); CK(hipMalloc(&d128, 64 * 16)); CK(hipMalloc(&dc, 64 * 16));
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d32, d128, dc);
    CK(hipDeviceSynchronize());
    unsigned char h32[64 * 4], h128[64 * 16], hc[64 * 16];
    CK(hipMemcpy(h32, d32, sizeof h32, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h128, d128, sizeof h128, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost));
    int bad32 = 0, bad128 = 0, badc = 0;
    for (int l = 0; l < 64; l++) {
        for (int k = 0; k < 4; k++) bad32 += h32[l * 4 + k] != (unsigned char)((l + k) * 7 + 3);
        for (int k = 0; k < 16; k++) bad128 += h128[l * 16 + k] != (unsigned char)((l + k) * 7 + 3);
        for (int k = 0; k < 16; k++) badc += hc[l * 16 + k] != (unsigned char)((l + k) * 7 + 3);
    }
    printf("ds_read_b32 at byte offsets 0..63: %d wrong bytes\nds_read_b128 at byte offsets 0..63: %d wrong bytes\ncompiler's 16-byte copy: %d wrong bytes\n",
           bad32, bad128, badc);
    printf("lane 1 b128 bytes:");
    for (int k = 0; k < 16; k++) printf(" %02x", h128[16 + k]);
    printf("  (expected");
    for (int k = 0; k < 16; k++) printf(" %02x", (unsigned char)((1 + k) * 7 + 3));
    printf(")\n");
    return 0;
}
