// Device kernels for the ResNet-50 feature path on gfx950 (MI355X / CDNA4).  wave = 64.
//
// Activations: bf16 NHWC.  Every convolution is an implicit GEMM on v_mfma_f32_16x16x32_bf16 with
// the *weights* as the MFMA A operand (rows = output channels) and the *pixels* as the B operand
// (columns), so an accumulator lane holds 4 consecutive output channels of one pixel and NHWC
// stores are 16 B per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ float bf16_bits_to_f32(unsigned int hi16) { return __uint_as_float(hi16 << 16); }

// Two fp32 -> packed bf16x2 (round-to-nearest-even; the plain cast lowers to v_cvt_pk_bf16_f32).
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 v = {lo, hi};
    bf16x2 b = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned int, b);
}

// ------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution (k = 1 or 3, any stride/pad), fused folded-BN bias + residual + ReLU.
//
// GEMM view:  Y[cout][pixel] = sum_k Wt[cout][k] * X[k][pixel],  k = (tap, cin), BK = 64 per step.
// Workgroup tile BC couts x BP pixels, 256 threads = WC x WP waves.  LDS image per operand is
// [rows][128 B] (one row = 64 bf16 of K), 16-B chunk c of row r stored at chunk c ^ (r & 7)
// (conflict-free ds_read_b128 for MFMA fragments).  Staging is either LDS-DMA
// (global_load_lds_dwordx4: destination is lane-linear, so the XOR is applied to the per-lane
// SOURCE address) or register staging (global_load_dwordx4 + ds_write_b128).
// Padding pixels and the M tail read from a 128-B zero page.
// Output-channel order inside each 32-row group of the W tile is permuted at staging time so that
// lane (q = lane>>4) of MFMA block pair (2t, 2t+1) ends up with channels 32t + 8q .. +7: one 16-B
// NHWC store per pixel.
// ------------------------------------------------------------------------------------------------
struct ConvArgs {
    const __bf16* x;      // (N,H,W,Cin)
    const __bf16* w;      // (Cout, taps, Cin)   K-major
    const float* bias;    // (Cout)
    const __bf16* res;    // (N,Ho,Wo,Cout) or nullptr
    __bf16* y;            // (N,Ho,Wo,Cout)
    const void* zero;     // >= 128 B of zeros
    int N, H, W, Cin, Ho, Wo, Cout;
    int ks, stride, pad, relu;
    int M;                // N*Ho*Wo
    int HoWo;
    int cin_chunks;       // Cin / 64
    int nk;               // ks*ks*cin_chunks
    int Ktot;             // ks*ks*Cin
    int n_ctiles;         // Cout / BC
    int n_blocks;
    int x_back;           // bytes the X descriptor base sits before x: (pad*W + pad)*Cin*2
    unsigned x_records;   // X descriptor size: activation bytes + x_back (< 2^31)
    unsigned w_bytes;     // W descriptor size
};

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    // Blocks b and b+8 share an XCD (observed round-robin; speed only, never correctness).  Give each
    // XCD a contiguous run of logical tile ids so tiles sharing an X panel hit the same L2.
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// MODE: 0 = LDS-DMA staging, 2 LDS stages, one barrier per K-step (vmcnt(0) before it);
//       1 = register staging (buffer_load_dwordx4 -> ds_write_b128), 2 LDS stages;
//       2 = LDS-DMA staging, 3 LDS stages, raw s_barrier + COUNTED vmcnt: two K-steps of loads stay in
//           flight across the barrier (one workgroup of 8 waves per CU).
//
// Addressing (the K loop carries almost no vector ALU work): both operands are fetched with
// buffer_load_dwordx4 [... lds]: address = SRD base + per-lane voffset (fixed for the whole kernel)
// + scalar soffset (the K-step's uniform displacement).
//   W: voffset = (cout_row * Ktot + chunk*8) * 2,  soffset = step * 128.
//   X: voffset = byte offset of the lane's pixel at the REFERENCE tap (pad,pad) -- always inside
//      the image -- and soffset = ((dh*W + dw)*Cin + cc*64) * 2 against an SRD base moved back by
//      (pad*W + pad)*Cin*2 bytes, so every component is non-negative.  A lane whose tap falls in the
//      padding (or whose row is past M) uses voffset = 2^31 >= num_records: the buffer range check
//      returns zeros for it (requires activation bytes < 2^31, checked on the host).
enum { IGEMM_GLDS2 = 0, IGEMM_REG2 = 1, IGEMM_GLDS3 = 2 };

// Diagnostic builds only (scripts/ablate.sh): R50_ABLATE = 1 no global loads inside the K loop,
// 2 = no MFMAs (fragment reads kept alive), 3 = no fragment reads and no MFMAs (fill rate only).
#ifndef R50_ABLATE
#define R50_ABLATE 0
#endif

constexpr unsigned kOobOffset = 0x80000000u;

template <int BC, int BP, int WC, int WP, int MODE>
__global__ __launch_bounds__(WC * WP * 64) void igemm_bf16_kernel(const ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // body only in the device pass: the host pass needs just the launch stub
                                      // (the LDS-DMA buffer builtin has no host-side lowering)
    constexpr int NT = WC * WP * 64;      // threads per workgroup
    constexpr bool GLDS = (MODE != IGEMM_REG2);
    constexpr int NSTAGE = (MODE == IGEMM_GLDS3) ? 3 : 2;
    constexpr int MR = BC / WC / 16;      // cout blocks per wave
    constexpr int NR = BP / WP / 16;      // pixel blocks per wave
    static_assert(MR >= 2 && (MR % 2) == 0, "wave needs >= 32 couts");
    constexpr int RPP = NT / 8;           // tile rows staged per pass (8 lanes x 16 B per 128-B row)
    static_assert(BC % RPP == 0 && BP % RPP == 0, "tile rows must be a multiple of the staging pass");
    constexpr int WROWS = BC / RPP;       // staging rows per thread
    constexpr int XROWS = BP / RPP;
    constexpr int PASS_BYTES = NT * 16;
    constexpr int STAGE_BYTES = (BC + BP) * 128;
    constexpr int LOADS_PER_STAGE = WROWS + XROWS;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_c = wave / WP, wave_p = wave % WP;

    const int tile = xcd_remap(blockIdx.x, a.n_blocks);
    const int ct = tile % a.n_ctiles;
    const int pt = tile / a.n_ctiles;
    const int c0 = ct * BC;
    const int p0 = pt * BP;

    const int fr = lane & 15, fq = lane >> 4;

    // ---- epilogue operands fetched FIRST: the residual / bias latency hides under the K loop ----
    u32x4 res_reg[MR / 2][NR];
    f32x4 bias_lo[MR / 2], bias_hi[MR / 2];
    const bool has_res = (a.res != nullptr);
#pragma unroll
    for (int t = 0; t < MR / 2; ++t) {
        const int cout = c0 + wave_c * MR * 16 + 32 * t + 8 * fq;
        bias_lo[t] = *reinterpret_cast<const f32x4*>(a.bias + cout);
        bias_hi[t] = *reinterpret_cast<const f32x4*>(a.bias + cout + 4);
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int m = p0 + (wave_p * NR + j) * 16 + fr;
            u32x4 r = (u32x4){0u, 0u, 0u, 0u};
            if (has_res && m < a.M) r = *reinterpret_cast<const u32x4*>(a.res + (size_t)m * a.Cout + cout);
            res_reg[t][j] = r;
        }
    }

    // ---- staging geometry: thread -> (row = i*RPP + srow, physical 16-B slot) ----
    const int srow = tid >> 3;
    const int slot = tid & 7;
    const int lchunk = slot ^ (srow & 7);          // logical K chunk this thread fetches

    const __amdgpu_buffer_rsrc_t rsrc_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w), 0, (unsigned)a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.x) - a.x_back), 0, (unsigned)a.x_records, 0x00020000);

    unsigned x_voff[XROWS];                        // byte offset of the lane's chunk at the reference tap
    unsigned x_mask[XROWS];                        // bit t set <=> tap t of this row is inside the image
#pragma unroll
    for (int i = 0; i < XROWS; ++i) {
        const int m = p0 + i * RPP + srow;
        unsigned mask = 0u, voff = kOobOffset;
        if (m < a.M) {
            const int n = m / a.HoWo;
            const int r = m - n * a.HoWo;
            const int ho = r / a.Wo;
            const int wo = r - ho * a.Wo;
            const int hc = ho * a.stride, wc = wo * a.stride;          // reference tap (pad,pad): always inside
            voff = (unsigned)(((n * a.H + hc) * a.W + wc) * a.Cin + lchunk * 8) * 2u;
            for (int dh = 0; dh < a.ks; ++dh)
                for (int dw = 0; dw < a.ks; ++dw) {
                    const int hi = hc - a.pad + dh, wi = wc - a.pad + dw;
                    if ((unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W) mask |= 1u << (dh * a.ks + dw);
                }
        }
        x_voff[i] = voff;
        x_mask[i] = mask;
    }
    unsigned w_voff[WROWS];
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
        const int rho = i * RPP + srow;            // LDS row
        const int cl = (rho & ~31) | (rho & 3) | (((rho >> 4) & 1) << 2) | (((rho >> 2) & 3) << 3);
        w_voff[i] = (unsigned)((c0 + cl) * a.Ktot + lchunk * 8) * 2u;
    }

    f32x4 acc[MR][NR];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read addresses (bytes, relative to the stage's W / X base)
    const int fphys0 = (fq ^ (fr & 7)) << 4;       // kk = 0; kk = 1 is ^ 64
    const int w_frag = (wave_c * MR * 16 + fr) * 128;
    const int x_frag = BC * 128 + (wave_p * NR * 16 + fr) * 128;

    u32x4 wreg[WROWS], xreg[XROWS];                // register staging only
    (void)wreg; (void)xreg;

    // scalar K-step state of the stage being ISSUED
    int s_tap = 0, s_cc = 0, s_dw = 0;
    int s_wofs = 0;                                // W soffset: step * 128
    int s_tapofs = 0;                              // X soffset of the tap: (dh*W + dw)*Cin*2
    const int row_adv = (a.W - a.ks) * a.Cin * 2;  // extra displacement when dw wraps to the next kernel row

    bool ablate_first = true;
    (void)ablate_first;
    auto stage_issue = [&](int buf) {
#if R50_ABLATE == 1 || R50_ABLATE == 4 || R50_ABLATE == 5
        if (!ablate_first) return;
        ablate_first = false;
#endif
        char* sbase = smem + buf * STAGE_BYTES;
        const int xofs = s_tapofs + s_cc * 128;
#pragma unroll
        for (int i = 0; i < WROWS; ++i) {
            if constexpr (GLDS) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (LDS_AS void*)(sbase + i * PASS_BYTES + wave * 1024), 16,
                                                         w_voff[i], s_wofs, 0, 0);
            } else {
                wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff[i], s_wofs, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < XROWS; ++i) {
            const unsigned voff = ((x_mask[i] >> s_tap) & 1u) ? x_voff[i] : kOobOffset;
            if constexpr (GLDS) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (LDS_AS void*)(sbase + BC * 128 + i * PASS_BYTES + wave * 1024),
                                                         16, voff, xofs, 0, 0);
            } else {
                xreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, voff, xofs, 0);
            }
        }
        s_wofs += 128;
        if (++s_cc == a.cin_chunks) {
            s_cc = 0;
            ++s_tap;
            s_tapofs += a.Cin * 2;
            if (++s_dw == a.ks) { s_dw = 0; s_tapofs += row_adv; }
        }
    };
    auto stage_write = [&](int buf) {              // register staging: regs -> LDS
        if constexpr (!GLDS) {
            char* sbase = smem + buf * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < WROWS; ++i)
                *reinterpret_cast<u32x4*>(sbase + i * PASS_BYTES + tid * 16) = wreg[i];
#pragma unroll
            for (int i = 0; i < XROWS; ++i)
                *reinterpret_cast<u32x4*>(sbase + BC * 128 + i * PASS_BYTES + tid * 16) = xreg[i];
        }
    };
    auto compute = [&](int buf) {
        // All 2*(MR+NR) fragment reads of the K-step are in flight early: the first half's, then the
        // second half's slotted between the first half's MFMAs (pinned below).
        const char* sbase = smem + buf * STAGE_BYTES;
#if R50_ABLATE == 3
        asm volatile("" ::"v"(sbase));
        return;
#endif
        bf16x8 wf[2][MR], xf[2][NR];
#if R50_ABLATE == 4
        {   // no LDS reads: opaque register garbage as fragments (timing only)
            u32x4 g = (u32x4){(unsigned)tid, 0x3f803f80u, (unsigned)lane, 0x3f803f80u};
            asm volatile("" : "+v"(g));
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int j = 0; j < NR; ++j) xf[kk][j] = __builtin_bit_cast(bf16x8, g);
#pragma unroll
                for (int m = 0; m < MR; ++m) wf[kk][m] = __builtin_bit_cast(bf16x8, g);
            }
            asm volatile("" ::"v"(sbase));
        }
#else
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ph = fphys0 ^ (kk << 6);
#pragma unroll
            for (int j = 0; j < NR; ++j)
                xf[kk][j] = *reinterpret_cast<const bf16x8*>(sbase + x_frag + j * 2048 + ph);
#pragma unroll
            for (int m = 0; m < MR; ++m)
                wf[kk][m] = *reinterpret_cast<const bf16x8*>(sbase + w_frag + m * 2048 + ph);
        }
#endif
#if R50_ABLATE == 2
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int m = 0; m < MR; ++m) asm volatile("" ::"v"(wf[kk][m]));
#pragma unroll
            for (int j = 0; j < NR; ++j) asm volatile("" ::"v"(xf[kk][j]));
        }
#else
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int j = 0; j < NR; ++j)
                    acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][m], xf[kk][j], acc[m][j], 0, 0, 0);
#endif
    };

    if constexpr (MODE == IGEMM_GLDS3) {
        // ---- 3 LDS stages; stage k+2 is issued right after the barrier that opens step k.  The wait
        //      before that barrier leaves the newest stage's LOADS_PER_STAGE DMAs in flight.
        //      RAW: own-wave vmcnt, then barrier, then ds_read.  WAR: buffer (k+2)%3 == (k-1)%3 was last
        //      read in step k-1, which every wave finished before arriving at this barrier. ----
        stage_issue(0);
        if (a.nk > 1) stage_issue(1);
        int buf = 0;
        for (int k = 0; k < a.nk; ++k) {
            if (k + 1 < a.nk) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS_PER_STAGE) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#if R50_ABLATE != 5
            __builtin_amdgcn_s_barrier();
#endif
            if (k + 2 < a.nk) stage_issue(buf >= 1 ? buf - 1 : 2);      // (buf + 2) % 3
            compute(buf);
            buf = (buf == 2) ? 0 : buf + 1;
        }
    } else {
        // ---- 2 LDS stages, one barrier per K-step ----
        stage_issue(0);
        stage_write(0);
        if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int buf = 0;
        for (int k = 0; k < a.nk; ++k) {
            const bool more = (k + 1 < a.nk);
            if (more) stage_issue(buf ^ 1);
            compute(buf);
            if (more) stage_write(buf ^ 1);
            if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if R50_ABLATE != 5
            __syncthreads();
#endif
            buf ^= 1;
        }
    }

    // ---- epilogue: + bias (+ residual) -> ReLU -> bf16 -> 16-B NHWC stores ----
#pragma unroll
    for (int t = 0; t < MR / 2; ++t) {
        const int cout = c0 + wave_c * MR * 16 + 32 * t + 8 * fq;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int m = p0 + (wave_p * NR + j) * 16 + fr;
            if (m < a.M) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[2 * t][j][e] + bias_lo[t][e];
                    v[4 + e] = acc[2 * t + 1][j][e] + bias_hi[t][e];
                }
                if (has_res) {
                    const u32x4 r = res_reg[t][j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] += bf16_bits_to_f32(r[e] & 0xffffu);
                        v[2 * e + 1] += bf16_bits_to_f32(r[e] >> 16);
                    }
                }
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                u32x4 out;
#pragma unroll
                for (int e = 0; e < 4; ++e) out[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                *reinterpret_cast<u32x4*>(a.y + (size_t)m * a.Cout + cout) = out;
            }
        }
    }
#else
    (void)a;
#endif
}

// ------------------------------------------------------------------------------------------------
// Stem, step 1: fp32 NCHW (N,3,224,224) -> bf16 "NHWC4" with a zero border:
//   xp[n][hp][wp][4], hp = hi + 3 in [0,230), wp = wi + 4 in [0,232); channel 3 = 0.
// One thread per output pixel (8 B).  The border is rewritten every call.
// ------------------------------------------------------------------------------------------------
constexpr int STEM_HP = 230, STEM_WP = 232;

__global__ __launch_bounds__(256) void stem_pack_kernel(const float* __restrict__ x, u32x2* __restrict__ xp, int n_img) {
    const long long total = (long long)n_img * STEM_HP * STEM_WP;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int wp = (int)(idx % STEM_WP);
        const long long t = idx / STEM_WP;
        const int hp = (int)(t % STEM_HP);
        const int n = (int)(t / STEM_HP);
        const int hi = hp - 3, wi = wp - 4;
        u32x2 o = (u32x2){0u, 0u};
        if ((unsigned)hi < 224u && (unsigned)wi < 224u) {
            const float* p = x + ((size_t)n * 3 * 224 + hi) * 224 + wi;
            const float c0 = p[0], c1 = p[224 * 224], c2 = p[2 * 224 * 224];
            o[0] = pack_bf16x2(c0, c1);
            o[1] = pack_bf16x2(c2, 0.f);
        }
        xp[idx] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// Stem, step 2: conv 7x7 s2 p3 (3->64) + bias + ReLU on the packed image.
// K is laid out per kernel row: k-step = kh (7 steps), 32 K-elements = 8 window pixels x 4 ch,
// window pixel j <-> input column 2*wo - 4 + j (j = 0 carries zero weights, kw = j - 1).
// Workgroup = one image x 4 output rows (4 waves, one row of 112 pixels each) x 64 couts.
// LDS: weights [kh][64 cout rows][64 B] (28,672 B) then 13 padded input rows (13*1856 B).
// Packed weight (global) is already in the LDS image order.
// ------------------------------------------------------------------------------------------------
constexpr int STEM_ROWS_PER_WG = 4;
constexpr int STEM_W_BYTES = 7 * 64 * 64;
constexpr int STEM_IN_ROWS = 2 * STEM_ROWS_PER_WG + 5;
constexpr int STEM_ROW_BYTES = STEM_WP * 8;
constexpr int STEM_LDS_BYTES = STEM_W_BYTES + STEM_IN_ROWS * STEM_ROW_BYTES;

__global__ __launch_bounds__(256) void stem_conv_kernel(const char* __restrict__ xp, const char* __restrict__ wpk,
                                                        const float* __restrict__ bias, __bf16* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x / (112 / STEM_ROWS_PER_WG);
    const int ho0 = (blockIdx.x % (112 / STEM_ROWS_PER_WG)) * STEM_ROWS_PER_WG;

    // weights: contiguous copy
    for (int c = tid; c < STEM_W_BYTES / 16; c += 256)
        *reinterpret_cast<u32x4*>(smem + c * 16) = *reinterpret_cast<const u32x4*>(wpk + c * 16);
    // input rows hp = 2*ho0 .. 2*ho0+12 of image n: one contiguous block
    const char* src = xp + ((size_t)n * STEM_HP + 2 * ho0) * STEM_ROW_BYTES;
    for (int c = tid; c < STEM_IN_ROWS * STEM_ROW_BYTES / 16; c += 256)
        *reinterpret_cast<u32x4*>(smem + STEM_W_BYTES + c * 16) = *reinterpret_cast<const u32x4*>(src + c * 16);
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[4][7];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // W fragment rows: LDS row rho = 16*m + fr holds the permuted cout (see igemm comment)
    const int w_frag = fr * 64 + fq * 16;
    // X fragment: pixel wo = 16*j + fr, bytes (2*wo + 2*fq) * 8 within input row (2*wave + kh)
    const int x_frag = STEM_W_BYTES + (2 * wave) * STEM_ROW_BYTES + fr * 16 + fq * 16;
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
        bf16x8 wf[4], xf[7];
#pragma unroll
        for (int m = 0; m < 4; ++m)
            wf[m] = *reinterpret_cast<const bf16x8*>(smem + kh * 4096 + m * 1024 + w_frag);
#pragma unroll
        for (int j = 0; j < 7; ++j)
            xf[j] = *reinterpret_cast<const bf16x8*>(smem + x_frag + kh * STEM_ROW_BYTES + j * 256);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 7; ++j)
                acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[m], xf[j], acc[m][j], 0, 0, 0);
    }

    const int ho = ho0 + wave;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int cout = 32 * t + 8 * fq;
        const f32x4 b_lo = *reinterpret_cast<const f32x4*>(bias + cout);
        const f32x4 b_hi = *reinterpret_cast<const f32x4*>(bias + cout + 4);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int wo = 16 * j + fr;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = fmaxf(acc[2 * t][j][e] + b_lo[e], 0.f);
                v[4 + e] = fmaxf(acc[2 * t + 1][j][e] + b_hi[e], 0.f);
            }
            u32x4 out;
#pragma unroll
            for (int e = 0; e < 4; ++e) out[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
            *reinterpret_cast<u32x4*>(y + (((size_t)n * 112 + ho) * 112 + wo) * 64 + cout) = out;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(3, stride 2, pad 1), bf16 NHWC.  One thread per (output pixel, 8 channels): nine 16-B
// loads, fp32 max (padding = -inf), one 16-B store.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y,
                                                           int N, int H, int W, int C, int Ho, int Wo) {
    const int cg = C >> 3;
    const long long total = (long long)N * Ho * Wo * cg;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(idx % cg);
        long long t = idx / cg;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -__builtin_huge_valf();
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int hi = 2 * ho - 1 + dh;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int wi = 2 * wo - 1 + dw;
                if ((unsigned)wi >= (unsigned)W) continue;
                const u32x4 v = *reinterpret_cast<const u32x4*>(x + (((size_t)n * H + hi) * W + wi) * C + g * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    m[2 * e] = fmaxf(m[2 * e], bf16_bits_to_f32(v[e] & 0xffffu));
                    m[2 * e + 1] = fmaxf(m[2 * e + 1], bf16_bits_to_f32(v[e] >> 16));
                }
            }
        }
        u32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = pack_bf16x2(m[2 * e], m[2 * e + 1]);
        *reinterpret_cast<u32x4*>(y + (size_t)idx * 8) = out;
    }
}

// ------------------------------------------------------------------------------------------------
// Global average pool: (N, HW, C) bf16 -> (N, C) fp32.  One thread per (n, 8 channels); the HW rows
// are summed in order in fp32 and multiplied by 1/HW.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool_kernel(const __bf16* __restrict__ x, float* __restrict__ y,
                                                      int N, int HW, int C, float inv_hw) {
    const int cg = C >> 3;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * cg) return;
    const int g = idx % cg, n = idx / cg;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    const __bf16* p = x + (size_t)n * HW * C + g * 8;
    for (int r = 0; r < HW; ++r) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(p + (size_t)r * C);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s[2 * e] += bf16_bits_to_f32(v[e] & 0xffffu);
            s[2 * e + 1] += bf16_bits_to_f32(v[e] >> 16);
        }
    }
    float* o = y + (size_t)n * C + g * 8;
    *reinterpret_cast<f32x4*>(o) = (f32x4){s[0] * inv_hw, s[1] * inv_hw, s[2] * inv_hw, s[3] * inv_hw};
    *reinterpret_cast<f32x4*>(o + 4) = (f32x4){s[4] * inv_hw, s[5] * inv_hw, s[6] * inv_hw, s[7] * inv_hw};
}
