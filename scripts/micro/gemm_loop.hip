// Calibration micro-benchmark (not part of the product): the inner K-step of the implicit GEMM in isolation.
// 4 waves (2x2), wave tile 64x64 (16 accumulators), 32 MFMA 16x16x32 per step, per variant:
//   V=0  MFMA + barrier                       V=1  + 16 ds_read_b128 fragment reads per step
//   V=2  + 8 buffer_load..lds (32 KiB/WG/step) from an L2-resident panel, vmcnt(0) before the barrier
//   V=3  as V=2 but 3 LDS stages and counted vmcnt (loads of 2 steps in flight)
// Build: hipcc --offload-arch=gfx950 -O3 -o gemm_loop gemm_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS_AS __attribute__((address_space(3)))

template <int V>
__global__ __launch_bounds__(256) void k(const char* panel, unsigned panel_bytes, float* out, int nk) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = 32768;
    constexpr int NST = (V == 3) ? 3 : 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wp = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int fphys0 = (fq ^ (fr & 7)) << 4;
    const int w_frag = (wc * 64 + fr) * 128, x_frag = 16384 + (wp * 64 + fr) * 128;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(panel), 0, panel_bytes, 0x00020000);
    for (int i = tid; i < NST * STAGE / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0x3f803f80u + (i & 0xff);
    __syncthreads();
    f32x4 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int j = 0; j < 4; ++j) acc[m][j] = (f32x4){0, 0, 0, 0};
    unsigned voff[8];
    for (int i = 0; i < 8; ++i) voff[i] = (unsigned)(((blockIdx.x * 8 + i) * 4096 + tid * 16) % (panel_bytes - 65536));
    int buf = 0, ibuf = 0, sofs = 0;
    auto issue = [&]() {
        if constexpr (V >= 2) {
            char* sb = smem + ibuf * STAGE;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)(sb + i * 4096 + wave * 1024), 16, voff[i], sofs, 0, 0);
            ibuf = (ibuf == NST - 1) ? 0 : ibuf + 1;
            sofs = (sofs + 128) & 32767;
        }
    };
    issue();
    if (V == 3) issue();
    for (int kstep = 0; kstep < nk; ++kstep) {
        if (V == 3 && kstep + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kstep + (V == 3 ? 2 : 1) < nk) issue();
        const char* sb = smem + buf * STAGE;
        bf16x8 wf[2][4], xf[2][4];
        if constexpr (V >= 1) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ph = fphys0 ^ (kk << 6);
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[kk][j] = *reinterpret_cast<const bf16x8*>(sb + x_frag + j * 2048 + ph);
#pragma unroll
                for (int m = 0; m < 4; ++m) wf[kk][m] = *reinterpret_cast<const bf16x8*>(sb + w_frag + m * 2048 + ph);
            }
        } else {
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
            u32x4 g = (u32x4){0x3f803f81u + lane, 0x3f803f80u, 0x3f813f80u, 0x3f803f82u};
            asm volatile("" : "+v"(g));
            for (int kk = 0; kk < 2; ++kk) for (int j = 0; j < 4; ++j) { xf[kk][j] = __builtin_bit_cast(bf16x8, g); wf[kk][j] = __builtin_bit_cast(bf16x8, g); }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][m], xf[kk][j], acc[m][j], 0, 0, 0);
        buf = (buf == NST - 1) ? 0 : buf + 1;
    }
    float s = 0.f;
    for (int m = 0; m < 4; ++m) for (int j = 0; j < 4; ++j) s += acc[m][j][0] + acc[m][j][1] + acc[m][j][2] + acc[m][j][3];
    out[blockIdx.x * 256 + tid] = s;
#endif
}

template <int V>
void run(const char* panel, unsigned pb, float* out, int wg_per_cu, int nk) {
    const int grid = 256 * wg_per_cu;
    const size_t lds = (V == 3 ? 3 : 2) * 32768;
    auto kern = k<V>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, panel, pb, out, nk);
    hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, panel, pb, out, nk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double flops = (double)grid * 4 * nk * 32 * 16384.0;
    printf("V%d wg/cu %d nk %d: %8.3f ms %7.1f TFLOP/s  (%.0f cycles/step/WG at 2.4 GHz)\n", V, wg_per_cu, nk, ms, flops / ms / 1e9,
           ms * 1e-3 * 2.4e9 / nk / 1.0);
}

int main() {
    const unsigned pb = 8u << 20;     // 8 MiB panel: L2 / MALL resident
    char* panel; float* out;
    hipMalloc(&panel, pb); hipMalloc(&out, 256 * 4 * 256 * 4);
    std::vector<unsigned> h(pb / 4); for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3f803f80u + (unsigned)(i * 2654435761u >> 24);
    hipMemcpy(panel, h.data(), pb, hipMemcpyHostToDevice);
    const int nk = 512;
    run<0>(panel, pb, out, 1, nk); run<0>(panel, pb, out, 2, nk);
    run<1>(panel, pb, out, 1, nk); run<1>(panel, pb, out, 2, nk);
    run<2>(panel, pb, out, 1, nk); run<2>(panel, pb, out, 2, nk);
    run<3>(panel, pb, out, 1, nk);
    return 0;
}
