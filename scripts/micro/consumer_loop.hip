// Micro-benchmark: the consumer loop of the role-specialised igemm in isolation (wave tile 32 couts x 112 pixels: per
// K-step of 64: 18 ds_read_b128 + 28 v_mfma_f32_16x16x32_bf16), 8 consumer waves per CU, LDS pre-filled once.
//   variant bit 0: s_barrier per step            bit 1: 4 extra waves issue 11 LDS-DMA (1 KiB each, L2-resident) per step
//   variant bit 2: no LDS reads (operands stay in registers)
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/consumer_loop scripts/micro/consumer_loop.hip && /tmp/consumer_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define LDS_AS __attribute__((address_space(3)))

template <int V>
__global__ __launch_bounds__(768) void k(const char* src, unsigned src_bytes, float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = (128 + 224) * 128;           // 44 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 3 * STAGE / 16; i += 768) reinterpret_cast<u32x4*>(smem)[i] = (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    if (wave >= 8) {
        if (!(V & 2)) { if (V & 1) for (int g = 0; g < steps; ++g) __builtin_amdgcn_s_barrier(); return; }
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, src_bytes, 0x00020000);
        const int lw = wave - 8;
        unsigned voff = (unsigned)((blockIdx.x * 4 + lw) * 11 * 1024 + lane * 16) % (src_bytes - 64 * 1024);
        int buf = 0;
        for (int g = 0; g < steps; ++g) {
            char* sb = smem + buf * STAGE + lw * 1024;
#pragma unroll
            for (int i = 0; i < 11; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(sb + i * 4096), 16, voff, i * 4096, 0, 0);
            asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
            if (V & 1) __builtin_amdgcn_s_barrier();
            buf = buf == 2 ? 0 : buf + 1;
            voff += 44 * 1024; if (voff > src_bytes - 128 * 1024) voff -= src_bytes - 128 * 1024;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    const int wave_c = wave >> 1, wave_p = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int fphys0 = (fq ^ (fr & 7)) << 4;
    const int w_frag = (wave_c * 32 + fr) * 128, x_frag = 128 * 128 + (wave_p * 112 + fr) * 128;
    f32x4 acc[2][7];
    for (int m = 0; m < 2; ++m) for (int j = 0; j < 7; ++j) acc[m][j] = (f32x4){0, 0, 0, 0};
    bf16x8 wf[2], xf[7];
    for (int m = 0; m < 2; ++m) wf[m] = *reinterpret_cast<const bf16x8*>(smem + w_frag + m * 2048 + fphys0);
    for (int j = 0; j < 7; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(smem + x_frag + j * 2048 + fphys0);
    int buf = 0;
    for (int g = 0; g < steps; ++g) {
        const char* sb = smem + buf * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ph = fphys0 ^ (kk << 6);
            if (!(V & 4)) {
#pragma unroll
                for (int j = 0; j < 7; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(sb + x_frag + j * 2048 + ph);
#pragma unroll
                for (int m = 0; m < 2; ++m) wf[m] = *reinterpret_cast<const bf16x8*>(sb + w_frag + m * 2048 + ph);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < 7; ++j) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[m], xf[j], acc[m][j], 0, 0, 0);
        }
        if (V & 1) __builtin_amdgcn_s_barrier();
        buf = buf == 2 ? 0 : buf + 1;
    }
    float s = 0.f;
    for (int m = 0; m < 2; ++m) for (int j = 0; j < 7; ++j) s += acc[m][j][0] + acc[m][j][1] + acc[m][j][2] + acc[m][j][3];
    if (s == 12345.f) out[0] = s;
}

template <int V>
void run(const char* name, const char* src, unsigned bytes, float* out) {
    const int steps = 2000;
    const size_t lds = 3 * (128 + 224) * 128;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<V><<<256, 768, lds>>>(src, bytes, out, steps);
    hipEventRecord(a);
    k<V><<<256, 768, lds>>>(src, bytes, out, steps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double flops = 256.0 * 8 * steps * 28 * 16384.0;
    printf("%-46s %7.3f ms  %7.1f TFLOP/s  (%.0f cycles@2.09GHz per step)\n", name, ms, flops / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.09e9 / steps);
}

int main() {
    char* src; float* out;
    const unsigned bytes = 8u << 20;
    hipMalloc(&src, bytes); hipMemset(src, 0, bytes); hipMalloc(&out, 4);
    run<4>("MFMA only (operands in registers)", src, bytes, out);
    run<5>("MFMA only + barrier", src, bytes, out);
    run<0>("LDS reads + MFMA", src, bytes, out);
    run<1>("LDS reads + MFMA + barrier", src, bytes, out);
    run<2>("LDS reads + MFMA + loader DMAs (no barrier)", src, bytes, out);
    run<3>("LDS reads + MFMA + loader DMAs + barrier", src, bytes, out);
    run<6>("MFMA only + loader DMAs (no barrier)", src, bytes, out);
    hipDeviceSynchronize();
    return 0;
}
