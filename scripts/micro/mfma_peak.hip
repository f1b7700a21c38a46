// Calibration micro-benchmark (not part of the product): sustained bf16 MFMA rate on this chip for
// register-resident operands (random data), as a function of waves per CU, MFMA shape and an
// optional workgroup barrier every 32 MFMAs.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int SHAPE, bool BARRIER>
__global__ void k(const u32x4* in, float* out, int iters) {
    u32x4 r0 = in[threadIdx.x], r1 = in[threadIdx.x + 512], r2 = in[threadIdx.x + 1024], r3 = in[threadIdx.x + 1536];
    bf16x8 a0 = __builtin_bit_cast(bf16x8, r0), a1 = __builtin_bit_cast(bf16x8, r1);
    bf16x8 b0 = __builtin_bit_cast(bf16x8, r2), b1 = __builtin_bit_cast(bf16x8, r3);
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((i & 1) ? a0 : a1, (i & 2) ? b0 : b1, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((i & 2) ? a0 : a1, (i & 1) ? b0 : b1, acc[i], 0, 0, 0);
            if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
        }
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((i & 1) ? a0 : a1, (r & 1) ? b0 : b1, acc[i], 0, 0, 0);
            if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
        }
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, bool BARRIER>
void run(const u32x4* in, float* out, int wg_per_cu, int threads, int iters) {
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<SHAPE, BARRIER>), dim3(grid), dim3(threads), 0, 0, in, out, iters);
    hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<SHAPE, BARRIER>), dim3(grid), dim3(threads), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    // flops per iteration per wave: 32 x (2*16*16*32) for SHAPE 16; 16 x (2*32*32*16) for SHAPE 32 -> both 524288
    const double flops = (double)grid * (threads / 64) * iters * 524288.0;
    printf("shape %dx%d  wg/cu %d  threads %d  barrier %d : %8.3f ms  %7.1f TFLOP/s\n", SHAPE, SHAPE, wg_per_cu, threads, (int)BARRIER, ms,
           flops / ms / 1e9);
}

int main() {
    std::vector<unsigned> h(2048 * 4);
    srand(1);
    for (auto& v : h) { unsigned lo = 0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15), hi = 0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15); v = lo | (hi << 16); }
    u32x4* in; float* out;
    hipMalloc(&in, h.size() * 4); hipMalloc(&out, 256 * 8 * 1024 * 4);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int iters = 4000;
    run<16, false>(in, out, 1, 256, iters);
    run<16, false>(in, out, 2, 256, iters);
    run<16, false>(in, out, 1, 512, iters);
    run<16, false>(in, out, 4, 256, iters);
    run<16, true>(in, out, 1, 256, iters);
    run<16, true>(in, out, 2, 256, iters);
    run<16, true>(in, out, 1, 512, iters);
    run<32, false>(in, out, 1, 256, iters);
    run<32, false>(in, out, 2, 256, iters);
    run<32, true>(in, out, 2, 256, iters);
    run<16, false>(in, out, 1, 256, 40);     // short kernels: launch/ramp overhead (26 us of MFMA at peak)
    run<16, false>(in, out, 3, 256, 40);
    return 0;
}
