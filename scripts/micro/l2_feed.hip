// Micro-benchmark: how fast can one CU pull L2-resident bytes, (a) into registers with buffer_load_dwordx4,
// (b) into LDS with `buffer_load_dwordx4 ... lds` (LDS-DMA), (c) registers + ds_write_b128?
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_feed scripts/micro/l2_feed.hip && /tmp/l2_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

template <int MODE, int UNROLL>
__global__ __launch_bounds__(1024) void feed(const char* src, unsigned bytes, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, bytes, 0x00020000);
    // each wave walks its own 1-KiB pieces: rows of 128 B, 8 rows per piece (the igemm staging shape)
    unsigned voff = (unsigned)(((blockIdx.x * 37 + wave * 11) * 1024) % bytes) + (lane >> 3) * 128 + (lane & 7) * 16;
    u32x4 acc = {0, 0, 0, 0};
    char* lbase = smem + wave * (UNROLL * 1024);
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            u32x4 r[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) r[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, u * 8192, 0);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= r[u];
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(lbase + u * 1024), 16, voff, u * 8192, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u32x4 r[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) r[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, u * 8192, 0);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) *reinterpret_cast<u32x4*>(lbase + u * 1024 + lane * 16) = r[u];
        }
        voff += UNROLL * 8192;
        if (voff >= bytes - UNROLL * 8192 - 2048) voff -= (bytes - UNROLL * 8192 - 2048) & ~1023u;
    }
    if (MODE != 0) { __syncthreads(); acc = *reinterpret_cast<u32x4*>(smem + tid * 16); }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int MODE, int UNROLL>
void run(const char* name, const char* src, unsigned bytes, unsigned* sink, int waves, int iters) {
    const size_t lds = (size_t)waves * UNROLL * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(feed<MODE, UNROLL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    feed<MODE, UNROLL><<<256, waves * 64, lds>>>(src, bytes, iters, sink);
    hipEventRecord(a);
    feed<MODE, UNROLL><<<256, waves * 64, lds>>>(src, bytes, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double total = 256.0 * waves * (double)iters * UNROLL * 1024.0;
    printf("%-10s waves/CU %2d unroll %d footprint %4u KiB: %7.1f GB/s per CU, %6.2f TB/s chip\n", name, waves, UNROLL, bytes >> 10,
           total / 256.0 / (ms * 1e-3) / 1e9, total / (ms * 1e-3) / 1e12);
}

int main() {
    char* src; unsigned* sink;
    const unsigned maxb = 64u << 20;
    hipMalloc(&src, maxb); hipMemset(src, 1, maxb); hipMalloc(&sink, 4);
    for (unsigned bytes : {1u << 20, 16u << 20}) {
        for (int waves : {4, 8, 16}) {
            run<0, 8>("regs", src, bytes, sink, waves, 2000);
            run<1, 8>("lds-dma", src, bytes, sink, waves, 2000);
            run<2, 8>("regs+dsw", src, bytes, sink, waves, 2000);
        }
    }
    hipDeviceSynchronize();
    return 0;
}
