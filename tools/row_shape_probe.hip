// Does a packet builder's copy depend on how full its 1 KiB row instructions are, and on where its rows start?
// One record of P chunks per wave (or two records per wave with their short second rows sharing one instruction),
// input at a stride of IS chunks, output at OS chunks; every wave copies and ends.
// Build: hipcc --offload-arch=gfx950 -O3 tools/row_shape_probe.hip -o tools/row_shape_probe.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// one record per wave: rows of 64 chunks, the last one as full as it is
__global__ __launch_bounds__(256) void k_single(const u32x4* __restrict__ src, u32x4* __restrict__ dst, u32 n, u32 P, u32 IS, u32 OS,
                                                u32 in_lead) {
    const u32 lane = threadIdx.x & 63u;
    const u32 w = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (w >= n) return;
    const u32x4* p = src + in_lead + (u64)w * IS;
    u32x4* q = dst + (u64)w * OS;
    const u32 last = P - 1u;
    const u32x4 a = __builtin_nontemporal_load(p + min(lane, last));
    const u32x4 b = __builtin_nontemporal_load(p + min(lane + 64u, last));
    if (lane < P) __builtin_nontemporal_store(a, q + lane);
    if (lane + 64u < P) __builtin_nontemporal_store(b, q + lane + 64u);
}

// two records of P = 64 + T chunks (T <= 32) per wave: three instructions each way, the middle one carrying both
// records' second rows (lanes 0..31 record A, lanes 32..63 record B)
__global__ __launch_bounds__(256) void k_pair(const u32x4* __restrict__ src, u32x4* __restrict__ dst, u32 n, u32 P, u32 IS, u32 OS,
                                              u32 in_lead) {
    const u32 lane = threadIdx.x & 63u;
    const u32 w = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (2u * w + 1u >= n) return;
    const u32x4* pa = src + in_lead + (u64)(2u * w) * IS;
    const u32x4* pb = pa + IS;
    u32x4* qa = dst + (u64)(2u * w) * OS;
    u32x4* qb = qa + OS;
    const u32 T = P - 64u, half = lane & 31u;
    const bool second = lane >= 32u;
    const u32x4 a = __builtin_nontemporal_load(pa + lane);
    const u32x4 m = __builtin_nontemporal_load((second ? pb : pa) + 64u + min(half, T - 1u));
    const u32x4 b = __builtin_nontemporal_load(pb + lane);
    __builtin_nontemporal_store(a, qa + lane);
    if (half < T) __builtin_nontemporal_store(m, (second ? qb : qa) + 64u + half);
    __builtin_nontemporal_store(b, qb + lane);
}

// R records per wave, one after another, the next record's two rows requested before the current one is stored
__global__ __launch_bounds__(256) void k_single_loop(const u32x4* __restrict__ src, u32x4* __restrict__ dst, u32 n, u32 P, u32 IS, u32 OS,
                                                     u32 in_lead, u32 R) {
    const u32 lane = threadIdx.x & 63u;
    const u32 w = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const u32 first = w * R;
    if (first >= n) return;
    const u32 cnt = min(R, n - first), last = P - 1u;
    const u32x4* p = src + in_lead + (u64)first * IS;
    u32x4* q = dst + (u64)first * OS;
    u32x4 a = __builtin_nontemporal_load(p + min(lane, last));
    u32x4 b = __builtin_nontemporal_load(p + min(lane + 64u, last));
    for (u32 r = 0; r < cnt; ++r) {
        const u32x4* pn = p + (r + 1u < cnt ? IS : 0u);
        const u32x4 na = __builtin_nontemporal_load(pn + min(lane, last));
        const u32x4 nb = __builtin_nontemporal_load(pn + min(lane + 64u, last));
        if (lane < P) __builtin_nontemporal_store(a, q + lane);
        if (lane + 64u < P) __builtin_nontemporal_store(b, q + lane + 64u);
        a = na;
        b = nb;
        p = pn;
        q += OS;
    }
}

// Records of up to 4 W chunks, 64 / W of them per step, one per group of W lanes (lane `sub` of a group moves chunks
// sub, sub + W, sub + 2 W, sub + 3 W); `steps` steps per wave, the next step's loads requested before the current
// step's stores.
template <u32 W>
__global__ __launch_bounds__(256) void k_groups(const u32x4* __restrict__ src, u32x4* __restrict__ dst, u32 n, u32 P, u32 IS, u32 OS,
                                                u32 steps) {
    const u32 lane = threadIdx.x & 63u, sub = lane & (W - 1u), g = lane / W;
    constexpr u32 per = 64u / W;
    const u32 w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    u32 rec = w * per * steps + g;
    const u32 last = P - 1u;
    u32x4 v[4], nv[4];
    {
        const u32x4* p = src + (u64)min(rec, n - 1u) * IS;
#pragma unroll
        for (u32 j = 0; j < 4; ++j) v[j] = __builtin_nontemporal_load(p + min(sub + j * W, last));
    }
    for (u32 s = 0; s < steps; ++s) {
        const u32 nrec = rec + per;
        if (s + 1u < steps) {
            const u32x4* p = src + (u64)min(nrec, n - 1u) * IS;
#pragma unroll
            for (u32 j = 0; j < 4; ++j) nv[j] = __builtin_nontemporal_load(p + min(sub + j * W, last));
        }
        if (rec < n) {
            u32x4* q = dst + (u64)rec * OS;
#pragma unroll
            for (u32 j = 0; j < 4; ++j)
                if (sub + j * W < P) q[sub + j * W] = v[j];
        }
#pragma unroll
        for (u32 j = 0; j < 4; ++j) v[j] = nv[j];
        rec = nrec;
    }
}

// The same records by the same lane groups, `steps` steps per wave, but the copies go to LDS at their place in the
// wave's output range first and leave as whole 1 KiB rows (streaming stores) once all of them are there.
// LD: 0 = default-policy loads, 2 = non-temporal.
template <u32 W, int LD>
__global__ __launch_bounds__(256) void k_groups_staged(const u32x4* __restrict__ src, u32x4* __restrict__ dst, u32 n, u32 P, u32 steps) {
    extern __shared__ u32x4 lds[];
    const u32 lane = threadIdx.x & 63u, sub = lane & (W - 1u), g = lane / W, wv = threadIdx.x >> 6;
    constexpr u32 per = 64u / W;
    const u32 w = blockIdx.x * (blockDim.x >> 6) + wv;
    const u32 per_wave = per * steps;
    const u32 first = w * per_wave;
    if (first >= n) return;
    const u32 cnt = min(per_wave, n - first);
    u32x4* area = lds + wv * (per_wave * P);
    const u32 last = P - 1u;
    for (u32 s = 0; s < steps; ++s) {
        const u32 r = s * per + g;
        const u32x4* p = src + (u64)min(first + r, n - 1u) * P;
        u32x4 v[4];
#pragma unroll
        for (u32 j = 0; j < 4; ++j) v[j] = LD ? __builtin_nontemporal_load(p + min(sub + j * W, last)) : p[min(sub + j * W, last)];
#pragma unroll
        for (u32 j = 0; j < 4; ++j)
            if (r < cnt && sub + j * W < P) area[r * P + sub + j * W] = v[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u32 span = cnt * P;
    u32x4* q = dst + (u64)first * P;
    for (u32 c = lane; c < span; c += 64u) __builtin_nontemporal_store(area[c], q + c);
}

// A wave's whole input range -- K rows of 64 chunks, `lead` chunks off a 128-byte line -- read as whole rows, all
// requested at once (streaming loads), placed in LDS `shift` dwords further on (a packet header's worth: the output of
// a packet builder is its input moved by 28 bytes per packet), and stored from there as whole rows.  The ceiling for
// a tile kernel that reads its payloads as one stream instead of packet by packet.
template <u32 K>
__global__ __launch_bounds__(256) void k_span_staged(const u32x4* __restrict__ src, u32x4* __restrict__ dst, u32 nspans, u32 lead, u32 shift) {
    extern __shared__ u32x4 lds[];
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const u32 w = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + wv);
    if (w >= nspans) return;
    u32* area = reinterpret_cast<u32*>(lds + wv * (K * 64u + 8u));
    const u32x4* p = src + lead + (u64)w * (K * 64u);
    u32x4* q = dst + (u64)w * (K * 64u);
    u32x4 v[K];
#pragma unroll
    for (u32 k = 0; k < K; ++k) v[k] = __builtin_nontemporal_load(p + k * 64u + lane);
#pragma unroll
    for (u32 k = 0; k < K; ++k) {
        u32* at = area + shift + (k * 64u + lane) * 4u;
        at[0] = v[k].x; at[1] = v[k].y; at[2] = v[k].z; at[3] = v[k].w;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (u32 k = 0; k < K; ++k) __builtin_nontemporal_store(reinterpret_cast<u32x4*>(area)[k * 64u + lane], q + k * 64u + lane);
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
    const bool only_span = argc > 1;  // any argument: only the last section (a tile's input range as rows)
    const u64 bytes = 3ull << 30;
    uint8_t *s, *d;
    CK(hipMalloc(&s, bytes + 4096));
    CK(hipMalloc(&d, bytes + 4096));
    hipLaunchKernelGGL(k_fill_random, dim3(65536), dim3(256), 0, 0, (u64*)s, bytes / 8);
    CK(hipMemset(d, 0, bytes));
    CK(hipDeviceSynchronize());
    const u64 nchunks = bytes / 16;
    struct Case { u32 P, IS, OS, lead; };
    // P chunks copied per record; lead: the input starts that many chunks off a 128-byte line
    const Case cases[] = {{64, 64, 64, 0},   {96, 96, 96, 0},   {128, 128, 128, 0}, {96, 96, 128, 0},  {96, 128, 128, 0},
                          {94, 94, 94, 0},   {94, 92, 96, 0},   {96, 92, 96, 0},    {96, 92, 96, 2},   {72, 72, 72, 0},
                          {80, 80, 80, 0},   {112, 112, 112, 0}, {65, 65, 72, 0},   {72, 65, 72, 0}};
    for (const Case& c : cases) {
        if (only_span) break;
        const u32 stride = c.IS > c.OS ? c.IS : c.OS;
        const u32 n = (u32)((nchunks - 64) / stride) & ~1u;
        const double moved = 2.0 * n * c.P * 16.0;
        for (int block : {256, 64}) {
            const int wpb = block / 64;
            double ms = timeit([&] { hipLaunchKernelGGL(k_single, dim3((n + wpb - 1) / wpb), dim3(block), 0, 0, (const u32x4*)s, (u32x4*)d, n, c.P, c.IS, c.OS, c.lead); }, 6);
            printf("P %3u in-stride %3u out-stride %3u lead %u block %3d  single %7.1f GB/s", c.P, c.IS, c.OS, c.lead, block, moved / ms / 1e6);
            if (c.P > 64 && c.P <= 96) {
                ms = timeit([&] { hipLaunchKernelGGL(k_pair, dim3((n / 2 + wpb - 1) / wpb), dim3(block), 0, 0, (const u32x4*)s, (u32x4*)d, n, c.P, c.IS, c.OS, c.lead); }, 6);
                printf("  pair %7.1f", moved / ms / 1e6);
            } else printf("  pair       -");
            for (u32 R : {4u, 8u}) {
                const u32 waves = (n + R - 1) / R;
                ms = timeit([&] { hipLaunchKernelGGL(k_single_loop, dim3((waves + wpb - 1) / wpb), dim3(block), 0, 0, (const u32x4*)s, (u32x4*)d, n, c.P, c.IS, c.OS, c.lead, R); }, 6);
                printf("  loop%u %7.1f", R, moved / ms / 1e6);
            }
            printf("\n");
            fflush(stdout);
        }
    }
    // mid-size and short records by lane groups: one step per wave against loops
    struct GCase { u32 W, P; };
    const GCase gcases[] = {{16, 36}, {16, 64}, {16, 52}, {8, 18}, {8, 32}, {4, 10}, {4, 16}, {4, 6}, {4, 3}};
    for (const GCase& c : gcases) {
        if (only_span) break;
        const u32 n = (u32)((nchunks - 64) / c.P);
        const double moved = 2.0 * n * c.P * 16.0;
        printf("groups of %2u lanes, records of %2u chunks packed:", c.W, c.P);
        for (u32 steps : {1u, 2u, 4u, 8u}) {
            const u32 per_wave = 64u / c.W * steps;
            const u32 waves = (n + per_wave - 1) / per_wave;
            double ms = 0;
            if (c.W == 16) ms = timeit([&] { hipLaunchKernelGGL(k_groups<16>, dim3((waves + 3) / 4), dim3(256), 0, 0, (const u32x4*)s, (u32x4*)d, n, c.P, c.P, c.P, steps); }, 6);
            if (c.W == 8) ms = timeit([&] { hipLaunchKernelGGL(k_groups<8>, dim3((waves + 3) / 4), dim3(256), 0, 0, (const u32x4*)s, (u32x4*)d, n, c.P, c.P, c.P, steps); }, 6);
            if (c.W == 4) ms = timeit([&] { hipLaunchKernelGGL(k_groups<4>, dim3((waves + 3) / 4), dim3(256), 0, 0, (const u32x4*)s, (u32x4*)d, n, c.P, c.P, c.P, steps); }, 6);
            printf("  %u step%s/wave %7.1f", steps, steps > 1 ? "s" : "", moved / ms / 1e6);
        }
        printf(" GB/s\n");
        printf("   ... staged through LDS, rows out:              ");
        for (u32 steps : {1u, 2u, 4u, 8u}) {
            const u32 per_wave = 64u / c.W * steps;
            if (per_wave * c.P * 16u * 4u > 64u * 1024u) { printf("  %u steps       -", steps); continue; }
            const u32 waves = (n + per_wave - 1) / per_wave;
            const u32 shm = per_wave * c.P * 16u * 4u;
            double ms = 0, ms0 = 0;
            if (c.W == 16) { ms = timeit([&] { hipLaunchKernelGGL((k_groups_staged<16, 2>), dim3((waves + 3) / 4), dim3(256), shm, 0, (const u32x4*)s, (u32x4*)d, n, c.P, steps); }, 6);
                             ms0 = timeit([&] { hipLaunchKernelGGL((k_groups_staged<16, 0>), dim3((waves + 3) / 4), dim3(256), shm, 0, (const u32x4*)s, (u32x4*)d, n, c.P, steps); }, 6); }
            if (c.W == 8) { ms = timeit([&] { hipLaunchKernelGGL((k_groups_staged<8, 2>), dim3((waves + 3) / 4), dim3(256), shm, 0, (const u32x4*)s, (u32x4*)d, n, c.P, steps); }, 6);
                            ms0 = timeit([&] { hipLaunchKernelGGL((k_groups_staged<8, 0>), dim3((waves + 3) / 4), dim3(256), shm, 0, (const u32x4*)s, (u32x4*)d, n, c.P, steps); }, 6); }
            if (c.W == 4) { ms = timeit([&] { hipLaunchKernelGGL((k_groups_staged<4, 2>), dim3((waves + 3) / 4), dim3(256), shm, 0, (const u32x4*)s, (u32x4*)d, n, c.P, steps); }, 6);
                            ms0 = timeit([&] { hipLaunchKernelGGL((k_groups_staged<4, 0>), dim3((waves + 3) / 4), dim3(256), shm, 0, (const u32x4*)s, (u32x4*)d, n, c.P, steps); }, 6); }
            printf("  %u steps nt %6.1f / default %6.1f", steps, moved / ms / 1e6, moved / ms0 / 1e6);
        }
        printf(" GB/s\n");
        fflush(stdout);
    }
    // a tile's whole input range as rows, through LDS, out as rows
    for (u32 lead : {0u, 3u}) {
        for (u32 shift : {0u, 7u}) {
            printf("span of K rows per wave through LDS (input %u chunks off a line, placed %u dwords on):", lead, shift);
#define SPAN(K_)                                                                                                                   \
    {                                                                                                                              \
        const u32 nsp = (u32)((nchunks - 64) / (K_ * 64u));                                                                        \
        const u32 shm = (K_ * 64u + 8u) * 16u * 4u;                                                                                \
        const double ms = timeit([&] { hipLaunchKernelGGL((k_span_staged<K_>), dim3((nsp + 3) / 4), dim3(256), shm, 0, (const u32x4*)s, (u32x4*)d, nsp, lead, shift); }, 6); \
        printf("  K=%u %7.1f", (u32)K_, 2.0 * nsp * K_ * 1024.0 / ms / 1e6);                                                        \
    }
            SPAN(1) SPAN(2) SPAN(4) SPAN(6) SPAN(8) SPAN(10)
#undef SPAN
            printf(" GB/s\n");
            fflush(stdout);
        }
    }
    return 0;
}
