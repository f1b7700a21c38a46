// Calibration micro-benchmark (not part of the product): sustained rate of the two fp8 MFMA forms on this chip for register-resident
// operands (random bytes that are valid e4m3 values): v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (what the fp8 conv
// path issues) and v_mfma_f32_16x16x32_fp8_fp8, beside v_mfma_f32_16x16x32_bf16, by waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_fp8_peak mfma_fp8_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// KIND 0: bf16 16x16x32; 1: fp8 16x16x32; 2: scaled f8f6f4 16x16x128
template <int KIND>
__global__ void k(const u32x4* in, float* out, int iters) {
    const u32x4 r0 = in[threadIdx.x], r1 = in[threadIdx.x + 512], r2 = in[threadIdx.x + 1024], r3 = in[threadIdx.x + 1536];
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (KIND == 0) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (i & 1) ? r0 : r1), __builtin_bit_cast(bf16x8, (i & 2) ? r2 : r3),
                                                                 acc[i], 0, 0, 0);
            } else if constexpr (KIND == 1) {
                const long a = ((long)((i & 1) ? r0[0] : r1[0]) << 32) | ((i & 1) ? r0[1] : r1[1]);
                const long b = ((long)((i & 2) ? r2[0] : r3[0]) << 32) | ((i & 2) ? r2[1] : r3[1]);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, acc[i], 0, 0, 0);
            } else {
                const u32x4 al = (i & 1) ? r0 : r1, ah = (i & 1) ? r1 : r0, bl = (i & 2) ? r2 : r3, bh = (i & 2) ? r3 : r2;
                const i32x8 a = {(int)al[0], (int)al[1], (int)al[2], (int)al[3], (int)ah[0], (int)ah[1], (int)ah[2], (int)ah[3]};
                const i32x8 b = {(int)bl[0], (int)bl[1], (int)bl[2], (int)bl[3], (int)bh[0], (int)bh[1], (int)bh[2], (int)bh[3]};
                acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const u32x4* in, float* out, int threads, int iters) {
    const int grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<KIND>), dim3(grid), dim3(threads), 0, 0, in, out, iters);
    hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<KIND>), dim3(grid), dim3(threads), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double kdim = KIND == 2 ? 128.0 : 32.0;
    const double flops = (double)grid * (threads / 64) * iters * 16.0 * (2.0 * 16 * 16 * kdim);
    const char* names[3] = {"bf16 16x16x32 ", "fp8  16x16x32 ", "f8f6f4 16x16x128 (unit scales)"};
    printf("%s waves/SIMD %d : %8.3f ms  %7.1f TFLOP/s\n", names[KIND], threads / 256, ms, flops / ms / 1e9);
}

int main() {
    std::vector<unsigned> h(4 * 2048);
    srand(1);
    for (auto& v : h) {                                   // random e4m3 bytes without the NaN encodings (0x7f / 0xff); also finite bf16 pairs
        unsigned w = 0;
        for (int b = 0; b < 4; ++b) { unsigned byte = rand() & 0xff; if ((byte & 0x7f) == 0x7f) byte ^= 1; if ((byte & 0x78) == 0x78) byte ^= 0x40; w |= byte << (8 * b); }
        v = w;
    }
    u32x4* in; float* out;
    hipMalloc((void**)&in, h.size() * 4); hipMalloc((void**)&out, 256 * 1024 * 4);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int threads : {256, 512, 1024}) {
        run<0>(in, out, threads, 4000);
        run<1>(in, out, threads, 4000);
        run<2>(in, out, threads, 2000);
    }
    return 0;
}
