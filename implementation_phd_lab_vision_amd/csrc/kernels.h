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

// Element type of operands and activations: ET 0 = bf16, ET 1 = fp16 (IEEE half).  Both are 2 bytes, so every layout,
// DMA and LDS image is shared; only the MFMA opcode and the fp32 <-> element conversions differ.  (`bf16x8` is used as the
// name of "a 16-byte MFMA operand register" for both.)  fp16 carries 11 significand bits instead of 8: with fp16 operands
// the whole network lands 3e-4 from the fp32 reference (bf16: 2.5e-3, almost all of it weight rounding) at the same speed;
// its range (65504) is far above anything a BN-folded ResNet-50 produces, and the conversion saturates instead of
// overflowing to infinity.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

template <int ET>
__device__ __forceinline__ f32x4 mfma_e(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (ET == 0) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int ET>
__device__ __forceinline__ f32x16 mfma32_e(bf16x8 a, bf16x8 b, f32x16 c) {       // v_mfma_f32_32x32x16_{bf16,f16}: 32 cycles, 8 of them on the issue port
    if constexpr (ET == 0) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// ET 2 = fp8 (OCP e4m3), igemm_ws_kernel only (BASELINE configs[4]).  One byte per element: a 128-byte LDS row holds 128 values of K
// instead of 64, everything the loaders do is unchanged (the host describes the tensors in 2-byte units), and the two 16-byte
// fragments a lane reads per row-step are together the 32-byte operand of ONE v_mfma_scale_f32_16x16x128_f8f6f4 (K = 128; both
// block scales 2^0) -- twice the multiply-adds per LDS byte and per issue slot of the 16-bit path.  A and B fragments are read
// with the same (lane, byte) -> chunk mapping, so whatever order the instruction assigns to K inside a lane, both operands agree.
typedef __attribute__((ext_vector_type(8))) int i32x8;
__device__ __forceinline__ f32x4 mfma_fp8_k128(bf16x8 a0, bf16x8 a1, bf16x8 b0, bf16x8 b1, f32x4 c) {
    const u32x4 al = __builtin_bit_cast(u32x4, a0), ah = __builtin_bit_cast(u32x4, a1);
    const u32x4 bl = __builtin_bit_cast(u32x4, b0), bh = __builtin_bit_cast(u32x4, b1);
    const i32x8 a = {(int)al[0], (int)al[1], (int)al[2], (int)al[3], (int)ah[0], (int)ah[1], (int)ah[2], (int)ah[3]};
    const i32x8 b = {(int)bl[0], (int)bl[1], (int)bl[2], (int)bl[3], (int)bh[0], (int)bh[1], (int)bh[2], (int)bh[3]};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}
__device__ __forceinline__ unsigned pack4_fp8(float v0, float v1, float v2, float v3) {   // saturating at +-448 (e4m3 max)
    v0 = __builtin_amdgcn_fmed3f(v0, -448.0f, 448.0f); v1 = __builtin_amdgcn_fmed3f(v1, -448.0f, 448.0f);
    v2 = __builtin_amdgcn_fmed3f(v2, -448.0f, 448.0f); v3 = __builtin_amdgcn_fmed3f(v3, -448.0f, 448.0f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(v0, v1, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v2, v3, w, true);
    return (unsigned)w;
}
template <int ET>
__device__ __forceinline__ unsigned int pack2_e(float lo, float hi) {     // two fp32 -> packed pair, round to nearest even
    if constexpr (ET == 0) {
        return pack_bf16x2(lo, hi);
    } else {
        lo = __builtin_amdgcn_fmed3f(lo, -65504.0f, 65504.0f);             // saturate: one v_med3_f32 per value
        hi = __builtin_amdgcn_fmed3f(hi, -65504.0f, 65504.0f);
        const f16x2 b = {(_Float16)lo, (_Float16)hi};                   // v_cvt_pk_f16_f32
        return __builtin_bit_cast(unsigned int, b);
    }
}
template <int ET>
__device__ __forceinline__ float unpack_lo_e(unsigned int u) {
    if constexpr (ET == 0) return bf16_bits_to_f32(u & 0xffffu);
    else return (float)__builtin_bit_cast(f16x2, u)[0];
}
template <int ET>
__device__ __forceinline__ float unpack_hi_e(unsigned int u) {
    if constexpr (ET == 0) return __uint_as_float(u & 0xffff0000u);
    else return (float)__builtin_bit_cast(f16x2, u)[1];
}

// ------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution (k = 1 or 3, stride 1 or 2), fused folded-BN bias + residual + ReLU.
//
// GEMM view:  Y[cout][pixel] = sum_k Wt[cout][k] * X[k][pixel],  k = (tap, cin), BK = 64 per step.
// Workgroup tile BC couts x BP pixels, WC x WP waves (4 or 8).  LDS image per operand is
// [rows][128 B] (one row = 64 bf16 of K), 16-B chunk c of row r stored at chunk c ^ (r & 7)
// (conflict-free ds_read_b128 for MFMA fragments).  Staging is LDS-DMA (buffer_load_dwordx4 ... lds:
// the destination is lane-linear, so the XOR is applied to the per-lane SOURCE offset).
// Output-channel order inside each 32-row group of the W tile is permuted at staging time so that
// lane (q = lane>>4) of MFMA block pair (2t, 2t+1) ends up with channels 32t + 8q .. +7: one 16-B
// NHWC store per pixel.
//
// Addressing keeps vector-ALU work out of the K loop and the epilogue: every global access is a
// buffer instruction = SRD base + per-lane voffset (fixed per tile) + scalar soffset.
//   W: voffset = (cout_row * Ktot + chunk*8) * 2,  soffset = step * 128.
//   X: voffset = byte offset of the lane's pixel at the REFERENCE tap (pad,pad) -- always inside
//      the image -- and soffset = ((dh*W + dw)*Cin + cc*64) * 2 against an SRD base moved back by
//      (pad*W + pad)*Cin*2 bytes, so every component is non-negative.  A lane whose tap falls in the
//      padding (or whose row is past M) uses voffset = 2^31 >= num_records: the buffer range check
//      returns zeros for it (activation bytes < 2^31, checked on the host).
//   Y / residual: voffset = ((tile_pixel0 + lane_pixel + 16*j) * Cout + lane_cout + 32*t) * 2; rows
//      past M are dropped by the range check (num_records = M*Cout*2).  The range check looks at
//      voffset (+ immediate) only, never at soffset: anything that must be checked lives in voffset.
//
// The kernel is PERSISTENT-capable: workgroup b walks tiles b', b' + grid, ... and treats its (tile,
// K-step) pairs as one stream through the LDS stage ring, so tile t+1's first K-steps are in flight
// while tile t finishes and tile t's stores drain under tile t+1's MFMAs.  grid = #tiles gives the
// one-tile-per-workgroup form.  NSTAGE = 2 (4 waves, several workgroups per CU) or 3 (8 waves,
// counted vmcnt: two K-steps of DMA in flight).  Per stream step g (D = NSTAGE-1):
//     s_waitcnt vmcnt((D-1)*LOADS)  own DMAs of step g landed (other vector-memory ops in between are
//                                   ordered by issue, so they can only make the wait stricter)
//     s_barrier                     RAW: everybody's DMAs of step g landed;  WAR: everybody finished
//                                   reading the buffer that step g+D overwrites
//     issue DMAs of step g+D ; ds_read + MFMA of step g ; [epilogue of a finished tile]
// ------------------------------------------------------------------------------------------------
struct FastDiv {          // n / d for n < 2^31: d == 1 -> mul == 0;  else (umulhi(n, mul) >> shr)
    unsigned mul, shr;
};

struct ConvArgs {
    const __bf16* x;      // (N,H,W,Cin)
    const __bf16* w;      // (Cout, taps, Cin)   K-major
    const float* bias;    // (Cout)
    const __bf16* res;    // (N,Ho,Wo,Cout) or nullptr
    __bf16* y;            // (N,Ho,Wo,Cout)
    int N, H, W, Cin, Ho, Wo, Cout;
    int ks, stride, pad, relu;
    int M;                // N*Ho*Wo
    int HoWo;
    int cin_chunks;       // Cin / 64
    int nk;               // ks*ks*cin_chunks
    int Ktot;             // ks*ks*Cin
    int n_ctiles;         // Cout / BC
    int n_blocks;         // tiles
    int x_back;           // bytes the X descriptor base sits before x: (pad*W + pad)*Cin*2
    unsigned x_records;   // X descriptor size: activation bytes + x_back (< 2^31)
    unsigned w_bytes;     // W descriptor size
    unsigned y_bytes;     // Y / residual descriptor size: M*y_cstride*2
    FastDiv div_howo, div_wo, div_ctiles;
    int x_cstride;        // channels per pixel of the X tensor (Cin; 2*Cin in split mode: [hi | lo])
    int x_wrap;           // K chunk index at which the X chunk index wraps to 0 again (split: 2*Cin/64; else huge)
    int y_cstride;        // channels per pixel of Y / residual (Cout; 2*Cout in split mode)
    int et;               // element type of operands / activations: 0 = bf16, 1 = fp16 (host-side dispatch only)
    float oscale, rscale; // fp8 only: y = fp8(act(acc * oscale + residual * rscale)); bias arrives divided by the dequantisation scale
    float q_inv;          // 16-bit igemm_ws_kernel only, > 0: the output is written as e4m3, fp8(value16 * q_inv) with value16 the 16-bit result
                          // (the hand-over of layer1's output to the fp8 stack folded into layer1.2.conv3's epilogue; y_bytes = M * Cout then)
    // Second K source of a 1x1 conv (igemm_ws_kernel only): K = [x (Cin = 64*cc1 channels) | x2 (the rest of cin_chunks)], the
    // second one read at stride2 from its own tensor -- conv3 and the downsample conv of a stage's first bottleneck as ONE GEMM
    // against [W3 | Wd].  x2 == nullptr: ordinary conv (cc1 is then huge).
    const __bf16* x2;     // (N,H2,W2,x2_cstride) or nullptr
    int H2, W2, stride2, x2_cstride, cc1;
    unsigned x2_records;
#if defined(R50_STAMP)    // diagnostic build (scripts/stamp_conv.py): per-wave cycle sums, 8 slots per wave
    unsigned long long* dbg;
#endif
};

// In-kernel cycle stamps of the role-specialised kernel (diagnostic build -DR50_STAMP=1 only).
#if defined(R50_STAMP)
#define R50_STAMP_DECL unsigned long long st_prev = __builtin_readcyclecounter(), st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long st_rt0 = __builtin_amdgcn_s_memrealtime();
#define R50_MARK(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long st_now = __builtin_readcyclecounter(); st_sum[i] += st_now - st_prev; st_prev = st_now; __builtin_amdgcn_sched_barrier(0); }
#define R50_STAMP_FLUSH(nw) st_sum[7] = __builtin_amdgcn_s_memrealtime() - st_rt0;   /* slot 7: 100-MHz ticks over the stamped region (held clock = cycles / ticks x 100 MHz) */ \
    if (a.dbg && (threadIdx.x & 63) == 0) { _Pragma("unroll") for (int q = 0; q < 8; ++q) \
        a.dbg[((size_t)blockIdx.x * (nw) + (threadIdx.x >> 6)) * 8 + q] = st_sum[q]; }
#else
#define R50_STAMP_DECL
#define R50_MARK(i)
#define R50_STAMP_FLUSH(nw)
#endif

__device__ __forceinline__ unsigned fast_div(unsigned n, FastDiv d) {
    return d.mul == 0u ? n : (__umulhi(n, d.mul) >> d.shr);
}

#ifndef XCD_REMAP
#define XCD_REMAP 1                                    // input-resident 3x3 kernels (A/B knob)
#endif
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    // Blocks b and b+8 share an XCD (observed round-robin; speed only, never correctness).  Give each
    // XCD a contiguous run of logical ids so tiles sharing an X panel hit the same L2.
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// s_waitcnt vmcnt(n) for a wave-uniform RUNTIME n (the instruction takes an immediate): six scalar compares for 0..63 instead of a
// linear chain of `if (n == e)` candidates on the loader waves that every barrier of a role-specialised kernel waits for.
template <int LO, int HI>
__device__ __forceinline__ void wait_vmcnt_range(int n) {
    if constexpr (LO == HI) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LO) : "memory");
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (n <= MID) wait_vmcnt_range<LO, MID>(n);
        else wait_vmcnt_range<MID + 1, HI>(n);
    }
}
__device__ __forceinline__ void wait_vmcnt(int n) { wait_vmcnt_range<0, 63>(n < 0 ? 0 : (n > 63 ? 63 : n)); }

__device__ __forceinline__ unsigned relu_bf16x2(unsigned v) {   // max(x, 0) on two packed bf16 = v_pk_max_i16
    typedef __attribute__((ext_vector_type(2))) short s16x2;
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z));
}

// (nt cache policies on the epilogues' residual loads and output stores were measured slower: profiles/r03_tail_cache_policy_ab.txt)
constexpr unsigned kOobOffset = 0x80000000u;

template <int ET, int BC, int BP, int WC, int WP, int NSTAGE, bool SPLIT = false>
__global__ __launch_bounds__(WC * WP * 64) void igemm_bf16_kernel(const ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // body only in the device pass: the host pass needs just the launch stub
                                      // (the LDS-DMA buffer builtin has no host-side lowering)
    constexpr int NT = WC * WP * 64;
    constexpr int MR = BC / WC / 16;      // cout blocks per wave
    constexpr int NR = BP / WP / 16;      // pixel blocks per wave
    static_assert(MR >= 2 && (MR % 2) == 0, "wave needs >= 32 couts");
    static_assert(NSTAGE == 2 || NSTAGE == 3, "2 or 3 LDS stages");
    constexpr int RPP = NT / 8;           // tile rows staged per pass (8 lanes x 16 B per 128-B row)
    static_assert(BC % RPP == 0 && BP % (16 * WP) == 0, "cout rows fill whole staging passes; pixels come in 16-blocks per wave");
    constexpr int WROWS = BC / RPP;       // staging rows per thread
    constexpr int XROWS = (BP + RPP - 1) / RPP;   // the last pass may be partial (e.g. BP = 208): its rows read as zeros
    constexpr int BP_PAD = XROWS * RPP;           // X rows held in LDS
    constexpr int PASS_BYTES = NT * 16;
    constexpr int STAGE_BYTES = (BC + BP_PAD) * 128;
    constexpr int LOADS_PER_STAGE = WROWS + XROWS;
    constexpr int D = NSTAGE - 1;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_c = wave / WP, wave_p = wave % WP;
    const int fr = lane & 15, fq = lane >> 4;
    const int srow = tid >> 3;
    const int slot = tid & 7;
    const int lchunk = slot ^ (srow & 7);          // logical K chunk this thread fetches

    const int grid = gridDim.x;
    const int first = xcd_remap(blockIdx.x, grid);
    const int my_tiles = (a.n_blocks - first + grid - 1) / grid;      // >= 1 (grid <= n_blocks)
    const int total = my_tiles * a.nk;

    const __amdgpu_buffer_rsrc_t rsrc_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w), 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.x) - a.x_back), 0, a.x_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_r =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.res), 0, a.res ? a.y_bytes : 0u, 0x00020000);

    // ---------------- issue side: geometry of the tile whose K-steps are being fetched --------------
    unsigned x_voff[XROWS], x_mask[XROWS], w_voff[WROWS];
    auto decode_tile = [&](int tile) {
        const int pt = (int)fast_div((unsigned)tile, a.div_ctiles);
        const int ct = tile - pt * a.n_ctiles;
        const int c0 = ct * BC, p0 = pt * BP;
#pragma unroll
        for (int i = 0; i < XROWS; ++i) {
            const int m = p0 + i * RPP + srow;
            unsigned mask = 0u, voff = kOobOffset;
            if ((BP_PAD == BP || i * RPP + srow < BP) && m < a.M) {
                const int n = (int)fast_div((unsigned)m, a.div_howo);
                const int r = m - n * a.HoWo;
                const int ho = (int)fast_div((unsigned)r, a.div_wo);
                const int wo = r - ho * a.Wo;
                const int hc = ho * a.stride, wc = wo * a.stride;          // reference tap (pad,pad): always inside
                voff = (unsigned)(((n * a.H + hc) * a.W + wc) * a.x_cstride + lchunk * 8) * 2u;
                if (a.ks == 1) {
                    mask = 1u;
                } else {                                                    // ks == 3: bit (dh*3 + dw)
                    const unsigned hm = (hc >= a.pad ? 1u : 0u) | 2u | (hc - a.pad + 2 < a.H ? 4u : 0u);
                    const unsigned wm = (wc >= a.pad ? 1u : 0u) | 2u | (wc - a.pad + 2 < a.W ? 4u : 0u);
                    mask = ((hm & 1u) ? wm : 0u) | ((hm & 2u) ? (wm << 3) : 0u) | ((hm & 4u) ? (wm << 6) : 0u);
                }
            }
            x_voff[i] = voff;
            x_mask[i] = mask;
        }
#pragma unroll
        for (int i = 0; i < WROWS; ++i) {
            const int rho = i * RPP + srow;            // LDS row -> channel (permuted inside 32-row groups)
            const int cl = (rho & ~31) | (rho & 3) | (((rho >> 4) & 1) << 2) | (((rho >> 2) & 3) << 3);
            w_voff[i] = (unsigned)((c0 + cl) * a.Ktot + lchunk * 8) * 2u;
        }
    };
    int i_tile = first, i_k = 0, i_tap = 0, i_cc = 0, i_dw = 0, i_wofs = 0, i_tapofs = 0, i_buf = 0;
    const int row_adv = (a.W - a.ks) * a.x_cstride * 2;  // extra displacement when dw wraps to the next kernel row
    decode_tile(i_tile);
    auto stage_issue = [&]() {
        char* sbase = smem + i_buf * STAGE_BYTES;
        // split mode: K per tap is [x_hi | x_lo | x_hi] against [w_hi | w_hi | w_lo]; the third block re-reads x_hi
        const int xcc = (i_cc >= a.x_wrap) ? i_cc - a.x_wrap : i_cc;
        // soffset has to be an SGPR: without the readfirstlane the compiler proves nothing about these
        // loop-carried counters and wraps every X DMA in a waterfall loop (readfirstlane/cmp/saveexec/branch)
        const int xofs = __builtin_amdgcn_readfirstlane(i_tapofs + xcc * 128);
        const int wofs = __builtin_amdgcn_readfirstlane(i_wofs);
        const int tap = __builtin_amdgcn_readfirstlane(i_tap);
#pragma unroll
        for (int i = 0; i < WROWS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (LDS_AS void*)(sbase + i * PASS_BYTES + wave * 1024), 16,
                                                     w_voff[i], wofs, 0, 0);
#pragma unroll
        for (int i = 0; i < XROWS; ++i) {
            const unsigned voff = ((x_mask[i] >> tap) & 1u) ? x_voff[i] : kOobOffset;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (LDS_AS void*)(sbase + BC * 128 + i * PASS_BYTES + wave * 1024),
                                                     16, voff, xofs, 0, 0);
        }
        i_buf = (i_buf == NSTAGE - 1) ? 0 : i_buf + 1;
        i_wofs += 128;
        if (++i_cc == a.cin_chunks) {
            i_cc = 0;
            ++i_tap;
            i_tapofs += a.x_cstride * 2;
            if (++i_dw == a.ks) { i_dw = 0; i_tapofs += row_adv; }
        }
        if (++i_k == a.nk) {                       // next tile of this workgroup's stream
            i_k = 0; i_tap = 0; i_cc = 0; i_dw = 0; i_wofs = 0; i_tapofs = 0;
            i_tile += grid;
            if (i_tile < a.n_blocks) decode_tile(i_tile);
        }
    };

    // ---------------- compute side ----------------------------------------------------------------
    f32x4 acc[MR][NR];
    constexpr bool BIG = (MR * NR > 16);          // 128+ accumulator registers: keep the other register users small
    constexpr bool LEAN = (NT >= 1024);           // 16 waves = 4 per SIMD: 128 registers per lane, one K half's fragments at a time
    constexpr bool PREFETCH_RES = !SPLIT && !BIG && !LEAN;
    u32x4 res_reg[PREFETCH_RES ? MR / 2 : 1][PREFETCH_RES ? NR : 1];
    const bool has_res = (a.res != nullptr);
    const int fphys0 = (fq ^ (fr & 7)) << 4;       // kk = 0; kk = 1 is ^ 64
    const int w_frag = (wave_c * MR * 16 + fr) * 128;
    const int x_frag = BC * 128 + (wave_p * NR * 16 + fr) * 128;
    const int cout_lane = wave_c * MR * 16 + 8 * fq;
    const int pix_lane = wave_p * NR * 16 + fr;
    int c_tile = first, c_k = 0, c_buf = 0;
    unsigned y_voff = 0u;                          // byte offset of (tile pixel0 + lane pixel, tile cout0 + lane cout)
    const unsigned y_rowstep = (unsigned)(16 * a.y_cstride * 2);   // 16 pixels further (next pixel block of the wave)

    auto tile_begin = [&]() {                      // accumulators start at the bias; residual fetched early
        const int pt = (int)fast_div((unsigned)c_tile, a.div_ctiles);
        const int c0 = (c_tile - pt * a.n_ctiles) * BC, p0 = pt * BP;
        y_voff = (unsigned)((p0 + pix_lane) * a.y_cstride + c0 + cout_lane) * 2u;
#pragma unroll
        for (int t = 0; t < MR / 2; ++t) {
            const f32x4 b_lo = *reinterpret_cast<const f32x4*>(a.bias + c0 + cout_lane + 32 * t);
            const f32x4 b_hi = *reinterpret_cast<const f32x4*>(a.bias + c0 + cout_lane + 32 * t + 4);
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                acc[2 * t][j] = b_lo;
                acc[2 * t + 1][j] = b_hi;
                if constexpr (PREFETCH_RES) if (has_res)     // row displacement in voffset: soffset is not part of the range check
                    res_reg[t][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, y_voff + j * y_rowstep + 64 * t, 0, 0);
            }
        }
    };
    auto compute = [&]() {
        const char* sbase = smem + c_buf * STAGE_BYTES;
        if constexpr (LEAN || (BIG && (MR * NR + 2 * (MR + NR)) * 4 > 208)) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ph = fphys0 ^ (kk << 6);
                bf16x8 wf[MR], xf[NR];
#pragma unroll
                for (int j = 0; j < NR; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(sbase + x_frag + j * 2048 + ph);
#pragma unroll
                for (int m = 0; m < MR; ++m) wf[m] = *reinterpret_cast<const bf16x8*>(sbase + w_frag + m * 2048 + ph);
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int j = 0; j < NR; ++j)
                        acc[m][j] = mfma_e<ET>(wf[m], xf[j], acc[m][j]);
            }
        } else {
        bf16x8 wf[2][MR], xf[2][NR];
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
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int j = 0; j < NR; ++j)
                    acc[m][j] = mfma_e<ET>(wf[kk][m], xf[kk][j], acc[m][j]);
        if constexpr (BIG) {     // 8-wave tile with room for both halves' fragments: every read of the step first
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MR + NR), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * MR * NR, 0);
        }
        }
    };
    auto epilogue = [&]() {                        // (+ residual) -> bf16 -> ReLU on the packed pair -> 16-B store
#pragma unroll
        for (int t = 0; t < MR / 2; ++t) {
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                f32x4 lo = acc[2 * t][j], hi = acc[2 * t + 1][j];
                const unsigned voff = y_voff + j * y_rowstep + 64 * t;
                if constexpr (!SPLIT) {
                    if (has_res) {
                        u32x4 r;
                        if constexpr (PREFETCH_RES) r = res_reg[t][j];
                        else r = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, voff, 0, 0);
                        lo[0] += unpack_lo_e<ET>(r[0]); lo[1] += unpack_hi_e<ET>(r[0]);
                        lo[2] += unpack_lo_e<ET>(r[1]); lo[3] += unpack_hi_e<ET>(r[1]);
                        hi[0] += unpack_lo_e<ET>(r[2]); hi[1] += unpack_hi_e<ET>(r[2]);
                        hi[2] += unpack_lo_e<ET>(r[3]); hi[3] += unpack_hi_e<ET>(r[3]);
                    }
                    u32x4 out = (u32x4){pack2_e<ET>(lo[0], lo[1]), pack2_e<ET>(lo[2], lo[3]), pack2_e<ET>(hi[0], hi[1]),
                                        pack2_e<ET>(hi[2], hi[3])};
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) out[e] = relu_bf16x2(out[e]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(out, rsrc_y, voff, 0, 0);
                } else {
                    // split mode: values travel as bf16 pairs (head, tail) with head + tail ~ fp32 (16 mantissa bits)
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if (has_res) {
                        const u32x4 rh = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, voff, 0, 0);
                        const u32x4 rl = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, voff, a.Cout * 2, 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[2 * e] += unpack_lo_e<ET>(rh[e]) + unpack_lo_e<ET>(rl[e]);
                            v[2 * e + 1] += unpack_hi_e<ET>(rh[e]) + unpack_hi_e<ET>(rl[e]);
                        }
                    }
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    u32x4 head, tail;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        head[e] = pack2_e<ET>(v[2 * e], v[2 * e + 1]);
                        tail[e] = pack2_e<ET>(v[2 * e] - unpack_lo_e<ET>(head[e]),
                                              v[2 * e + 1] - unpack_hi_e<ET>(head[e]));
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(head, rsrc_y, voff, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(tail, rsrc_y, voff, a.Cout * 2, 0);
                }
            }
        }
    };

    // ---------------- the stream ------------------------------------------------------------------
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < total) stage_issue();
    for (int g = 0; g < total; ++g) {
        if (NSTAGE == 3 && g + 1 < total) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS_PER_STAGE) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (c_k == 0) tile_begin();
        if (g + D < total) stage_issue();
        compute();
        c_buf = (c_buf == NSTAGE - 1) ? 0 : c_buf + 1;
        if (++c_k == a.nk) {
            epilogue();
            c_k = 0;
            c_tile += grid;
        }
    }
#else
    (void)a;
#endif
}

// ------------------------------------------------------------------------------------------------
// Role-specialised implicit GEMM ("ws"): the same tile stream as igemm_bf16_kernel, but the LDS-DMA
// instructions are issued by dedicated LOADER waves and the MFMAs by CONSUMER waves.
// Why: in-kernel stamps on the unspecialised loop show a wave spending ~850 cycles per K-step just ISSUING
// its 8 `buffer_load ... lds` (~100 cycles each: the wave is held while the address unit takes the 64 lanes)
// next to ~800 cycles of fragment reads + 32 MFMAs (512 of them MFMA) -- the DMA latency itself is hidden.
// A loader wave can sit in that issue stall for free; the consumer waves then run ds_read + MFMA only.
// One workgroup per CU (NSTAGE stages of LDS), persistent over tiles.  Per stream step g:
//     loaders:   issue DMAs of step g+D (D = NSTAGE-1) ; wait until their DMAs of step g+1 landed ; s_barrier
//     consumers: [tile begin] ; ds_read + MFMA of step g ; [epilogue] ; s_barrier
// RAW: the barrier ending step g comes after every loader's wait for stage g+1.
// WAR: stage g+D goes into the buffer of stage g-1, which the consumers finished before the barrier ending
//      step g-1.
// ------------------------------------------------------------------------------------------------
template <int ET, int BC, int BP, int CWC, int CWP, int NLOAD, int NSTAGE>
__global__ __launch_bounds__((CWC * CWP + NLOAD) * 64) void igemm_ws_kernel(const ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NCONS = CWC * CWP;
    constexpr int MR = BC / CWC / 16;
    constexpr int NR = BP / CWP / 16;
    static_assert(MR >= 2 && (MR % 2) == 0, "consumer wave needs >= 32 couts");
    static_assert(BP % (16 * CWP) == 0, "pixels come in 16-blocks per consumer wave");
    constexpr int RPPL = NLOAD * 8;               // rows staged per loader pass
    static_assert(BC % RPPL == 0, "cout rows fill whole loader passes");
    constexpr int WROWS = BC / RPPL;
    constexpr int XROWS = (BP + RPPL - 1) / RPPL;
    constexpr int BP_PAD = XROWS * RPPL;
    constexpr int PASS_BYTES = RPPL * 128;
    constexpr int STAGE_BYTES = (BC + BP_PAD) * 128;
    constexpr int LOADS_PER_STAGE = WROWS + XROWS;
    constexpr int D = NSTAGE - 1;
    static_assert(NSTAGE >= 3 && NSTAGE <= 4, "3 or 4 LDS stages");
    constexpr bool BIG = (MR * NR > 16);

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = (wave >= NCONS);

    const int grid = gridDim.x;
    const int first = xcd_remap(blockIdx.x, grid);
    const int my_tiles = (a.n_blocks - first + grid - 1) / grid;
    const int total = my_tiles * a.nk;

    if (is_loader) {
        // =============================== loader waves ===============================================
        const int lw = wave - NCONS;
        const int lt = tid - NCONS * 64;
        const int srow = lt >> 3;
        const int slot = lt & 7;
        const int lchunk = slot ^ (srow & 7);
        const __amdgpu_buffer_rsrc_t rsrc_w =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w), 0, a.w_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(a.x) - a.x_back), 0, a.x_records, 0x00020000);
        const bool dual = (a.x2 != nullptr);
        const __amdgpu_buffer_rsrc_t rsrc_x2 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<__bf16*>(dual ? a.x2 : a.x), 0, dual ? a.x2_records : 0u, 0x00020000);
        unsigned x_voff[XROWS], x_mask[XROWS], w_voff[WROWS], x2_voff[XROWS];
        auto decode_tile = [&](int tile) {
            const int pt = (int)fast_div((unsigned)tile, a.div_ctiles);
            const int c0 = (tile - pt * a.n_ctiles) * BC, p0 = pt * BP;
#pragma unroll
            for (int i = 0; i < XROWS; ++i) {
                const int m = p0 + i * RPPL + srow;
                unsigned mask = 0u, voff = kOobOffset;
                x2_voff[i] = kOobOffset;
                if ((BP_PAD == BP || i * RPPL + srow < BP) && m < a.M) {
                    const int n = (int)fast_div((unsigned)m, a.div_howo);
                    const int r = m - n * a.HoWo;
                    const int ho = (int)fast_div((unsigned)r, a.div_wo);
                    const int wo = r - ho * a.Wo;
                    const int hc = ho * a.stride, wc = wo * a.stride;
                    voff = (unsigned)(((n * a.H + hc) * a.W + wc) * a.x_cstride + lchunk * 8) * 2u;
                    if (dual) x2_voff[i] = (unsigned)(((n * a.H2 + ho * a.stride2) * a.W2 + wo * a.stride2) * a.x2_cstride + lchunk * 8) * 2u;
                    if (a.ks == 1) {
                        mask = 1u;
                    } else {
                        const unsigned hm = (hc >= a.pad ? 1u : 0u) | 2u | (hc - a.pad + 2 < a.H ? 4u : 0u);
                        const unsigned wm = (wc >= a.pad ? 1u : 0u) | 2u | (wc - a.pad + 2 < a.W ? 4u : 0u);
                        mask = ((hm & 1u) ? wm : 0u) | ((hm & 2u) ? (wm << 3) : 0u) | ((hm & 4u) ? (wm << 6) : 0u);
                    }
                }
                x_voff[i] = voff;
                x_mask[i] = mask;
            }
#pragma unroll
            for (int i = 0; i < WROWS; ++i) {
                const int rho = i * RPPL + srow;
                const int cl = (rho & ~31) | (rho & 3) | (((rho >> 4) & 1) << 2) | (((rho >> 2) & 3) << 3);
                w_voff[i] = (unsigned)((c0 + cl) * a.Ktot + lchunk * 8) * 2u;
            }
        };
        int i_tile = first, i_k = 0, i_tap = 0, i_cc = 0, i_dw = 0, i_wofs = 0, i_tapofs = 0, i_buf = 0;
        const int row_adv = (a.W - a.ks) * a.x_cstride * 2;
        decode_tile(i_tile);
        auto stage_issue = [&]() {
            char* sbase = smem + i_buf * STAGE_BYTES;
            // soffset has to be an SGPR: without the readfirstlane the compiler wraps every DMA in a waterfall loop
            const int xcc = (i_cc >= a.x_wrap) ? i_cc - a.x_wrap : i_cc;     // bf16w2 mode: K per tap is [x | x] against [w_head | w_tail]
            const int xofs = __builtin_amdgcn_readfirstlane(i_tapofs + xcc * 128);
            const int wofs = __builtin_amdgcn_readfirstlane(i_wofs);
            const int tap = __builtin_amdgcn_readfirstlane(i_tap);
#pragma unroll
            for (int i = 0; i < WROWS; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (LDS_AS void*)(sbase + i * PASS_BYTES + lw * 1024), 16, w_voff[i],
                                                         wofs, 0, 0);
            if (i_cc >= a.cc1) {                           // second K source (uniform): its own descriptor, pixels and chunk offset
                const int x2ofs = __builtin_amdgcn_readfirstlane((i_cc - a.cc1) * 128);
#pragma unroll
                for (int i = 0; i < XROWS; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x2, (LDS_AS void*)(sbase + BC * 128 + i * PASS_BYTES + lw * 1024), 16,
                                                             x2_voff[i], x2ofs, 0, 0);
            } else if (a.ks == 1) {
                // 1x1 conv (round 3): a row's offset is either valid for the whole K loop or out of range (decode_tile), so the per-tap padding mask --
                // four vector-ALU instructions per DMA, ~0.5 per MFMA of the kernel, issued on the SIMDs the consumers' MFMAs issue on -- is skipped
#pragma unroll
                for (int i = 0; i < XROWS; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (LDS_AS void*)(sbase + BC * 128 + i * PASS_BYTES + lw * 1024), 16,
                                                             x_voff[i], xofs, 0, 0);
            } else {
#pragma unroll
            for (int i = 0; i < XROWS; ++i) {
                const unsigned voff = ((x_mask[i] >> tap) & 1u) ? x_voff[i] : kOobOffset;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (LDS_AS void*)(sbase + BC * 128 + i * PASS_BYTES + lw * 1024), 16,
                                                         voff, xofs, 0, 0);
            }
            }
            i_buf = (i_buf == NSTAGE - 1) ? 0 : i_buf + 1;
            i_wofs += 128;
            if (++i_cc == a.cin_chunks) {
                i_cc = 0;
                ++i_tap;
                i_tapofs += a.x_cstride * 2;
                if (++i_dw == a.ks) { i_dw = 0; i_tapofs += row_adv; }
            }
            if (++i_k == a.nk) {
                i_k = 0; i_tap = 0; i_cc = 0; i_dw = 0; i_wofs = 0; i_tapofs = 0;
                i_tile += grid;
                if (i_tile < a.n_blocks) decode_tile(i_tile);
            }
        };
#pragma unroll
        for (int s = 0; s < D; ++s)
            if (s < total) stage_issue();
        // stage 0 landed: at most D-1 younger stages may stay in flight
        if (total >= D) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * LOADS_PER_STAGE) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        R50_STAMP_DECL
        for (int g = 0; g < total; ++g) {
            if (g + D < total) stage_issue();
            R50_MARK(0)                                  // DMA issue
            // stages issued so far: 0 .. min(g+D, total-1); stage g+1 must be complete
            if (g + D < total) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * LOADS_PER_STAGE) : "memory");
            } else if (D >= 3 && g + D - 1 < total) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D >= 3 ? D - 2 : 0) * LOADS_PER_STAGE) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            R50_MARK(1)                                  // wait: next stage landed
            __builtin_amdgcn_s_barrier();
            R50_MARK(2)                                  // barrier
        }
        R50_STAMP_FLUSH(NCONS + NLOAD)
    } else {
        // =============================== consumer waves =============================================
        const int wave_c = wave / CWP, wave_p = wave % CWP;
        const int fr = lane & 15, fq = lane >> 4;
        const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
        // (with a quantising epilogue y_bytes counts e4m3 bytes; the residual is still 16-bit)
        const __amdgpu_buffer_rsrc_t rsrc_r =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.res), 0, a.res ? (a.q_inv > 0.f ? 2u * a.y_bytes : a.y_bytes) : 0u, 0x00020000);
        f32x4 acc[MR][NR];
        constexpr bool PREFETCH_RES = !BIG;
        u32x4 res_reg[PREFETCH_RES ? MR / 2 : 1][PREFETCH_RES ? NR : 1];
        const bool has_res = (a.res != nullptr);
        const int fphys0 = (fq ^ (fr & 7)) << 4;
        const int w_frag = (wave_c * MR * 16 + fr) * 128;
        const int x_frag = BC * 128 + (wave_p * NR * 16 + fr) * 128;
        const int cout_lane = wave_c * MR * 16 + 8 * fq;
        const int pix_lane = wave_p * NR * 16 + fr;
        int c_tile = first, c_k = 0, c_buf = 0;
        unsigned y_voff = 0u;
        constexpr unsigned ESZ = (ET == 2) ? 1u : 2u;  // bytes per element of Y / residual
        const unsigned y_rowstep = (unsigned)(16 * a.y_cstride) * ESZ;

        // The bias of the NEXT tile is fetched before the epilogue's stores go out: vmcnt retires in order, so a
        // bias load issued behind 28 stores would wait for every one of them to be acknowledged.
        f32x4 bias_reg[MR];
        auto bias_fetch = [&](int tile) {
            const int pt = (int)fast_div((unsigned)tile, a.div_ctiles);
            const int c0 = (tile - pt * a.n_ctiles) * BC;
#pragma unroll
            for (int t = 0; t < MR / 2; ++t) {
                bias_reg[2 * t] = *reinterpret_cast<const f32x4*>(a.bias + c0 + cout_lane + 32 * t);
                bias_reg[2 * t + 1] = *reinterpret_cast<const f32x4*>(a.bias + c0 + cout_lane + 32 * t + 4);
            }
        };
        auto tile_begin = [&]() {
            const int pt = (int)fast_div((unsigned)c_tile, a.div_ctiles);
            const int c0 = (c_tile - pt * a.n_ctiles) * BC, p0 = pt * BP;
            y_voff = (unsigned)((p0 + pix_lane) * a.y_cstride + c0 + cout_lane) * ESZ;
#pragma unroll
            for (int t = 0; t < MR / 2; ++t) {
                const f32x4 b_lo = bias_reg[2 * t];
                const f32x4 b_hi = bias_reg[2 * t + 1];
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    acc[2 * t][j] = b_lo;
                    acc[2 * t + 1][j] = b_hi;
                    if constexpr (PREFETCH_RES) if (has_res) {
                        if constexpr (ET == 2) {
                            const u32x2 r2 = __builtin_amdgcn_raw_buffer_load_b64(rsrc_r, y_voff + j * y_rowstep + 32 * t, 0, 0);
                            res_reg[t][j] = (u32x4){r2[0], r2[1], 0u, 0u};
                        } else {
                            res_reg[t][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, y_voff + j * y_rowstep + 64 * t, 0, 0);
                        }
                    }
                }
            }
        };
        // One consumer wave per SIMD has nobody to hide its LDS latency behind, so all fragment reads of the
        // step go out first (both K halves): the second half lands under the first half's MFMAs.
        auto compute = [&]() {
            const char* sbase = smem + c_buf * STAGE_BYTES;
            bf16x8 wf[2][MR], xf[2][NR];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ph = fphys0 ^ (kk << 6);
#pragma unroll
                for (int j = 0; j < NR; ++j) xf[kk][j] = *reinterpret_cast<const bf16x8*>(sbase + x_frag + j * 2048 + ph);
#pragma unroll
                for (int m = 0; m < MR; ++m) wf[kk][m] = *reinterpret_cast<const bf16x8*>(sbase + w_frag + m * 2048 + ph);
            }
            if constexpr (ET == 2) {                        // fp8: the row-step's two fragments are one K = 128 operand
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int j = 0; j < NR; ++j) acc[m][j] = mfma_fp8_k128(wf[0][m], wf[1][m], xf[0][j], xf[1][j], acc[m][j]);
            } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int j = 0; j < NR; ++j)
                        acc[m][j] = mfma_e<ET>(wf[kk][m], xf[kk][j], acc[m][j]);
            }
            if constexpr (NCONS + NLOAD <= 8 && ET != 2) {     // 2 waves per SIMD: 256 registers, room for both halves' fragments
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MR + NR), 0);   // every LDS read of the step ...
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * MR * NR, 0);     // ... then the MFMAs
            }
        };
        auto epilogue = [&]() {
#pragma unroll
            for (int t = 0; t < MR / 2; ++t) {
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    f32x4 lo = acc[2 * t][j], hi = acc[2 * t + 1][j];
                    const unsigned voff = y_voff + j * y_rowstep + 32 * ESZ * t;
                    if constexpr (ET == 2) {                // fp8: dequantise, add the residual, ReLU, requantise, one 8-byte store
#pragma unroll
                        for (int e = 0; e < 4; ++e) { lo[e] *= a.oscale; hi[e] *= a.oscale; }
                        if (has_res) {
                            u32x2 r;
                            if constexpr (PREFETCH_RES) r = (u32x2){res_reg[t][j][0], res_reg[t][j][1]};
                            else r = __builtin_amdgcn_raw_buffer_load_b64(rsrc_r, voff, 0, 0);
                            lo[0] += __builtin_amdgcn_cvt_f32_fp8((int)r[0], 0) * a.rscale; lo[1] += __builtin_amdgcn_cvt_f32_fp8((int)r[0], 1) * a.rscale;
                            lo[2] += __builtin_amdgcn_cvt_f32_fp8((int)r[0], 2) * a.rscale; lo[3] += __builtin_amdgcn_cvt_f32_fp8((int)r[0], 3) * a.rscale;
                            hi[0] += __builtin_amdgcn_cvt_f32_fp8((int)r[1], 0) * a.rscale; hi[1] += __builtin_amdgcn_cvt_f32_fp8((int)r[1], 1) * a.rscale;
                            hi[2] += __builtin_amdgcn_cvt_f32_fp8((int)r[1], 2) * a.rscale; hi[3] += __builtin_amdgcn_cvt_f32_fp8((int)r[1], 3) * a.rscale;
                        }
                        if (a.relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { lo[e] = fmaxf(lo[e], 0.f); hi[e] = fmaxf(hi[e], 0.f); }
                        }
                        const u32x2 o8 = (u32x2){pack4_fp8(lo[0], lo[1], lo[2], lo[3]), pack4_fp8(hi[0], hi[1], hi[2], hi[3])};
                        __builtin_amdgcn_raw_buffer_store_b64(o8, rsrc_y, voff, 0, 0);
                        continue;
                    }
                    if (has_res) {
                        u32x4 r;
                        if constexpr (PREFETCH_RES) r = res_reg[t][j];
                        else r = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, voff, 0, 0);
                        lo[0] += unpack_lo_e<ET>(r[0]); lo[1] += unpack_hi_e<ET>(r[0]);
                        lo[2] += unpack_lo_e<ET>(r[1]); lo[3] += unpack_hi_e<ET>(r[1]);
                        hi[0] += unpack_lo_e<ET>(r[2]); hi[1] += unpack_hi_e<ET>(r[2]);
                        hi[2] += unpack_lo_e<ET>(r[3]); hi[3] += unpack_hi_e<ET>(r[3]);
                    }
                    u32x4 out = (u32x4){pack2_e<ET>(lo[0], lo[1]), pack2_e<ET>(lo[2], lo[3]), pack2_e<ET>(hi[0], hi[1]),
                                        pack2_e<ET>(hi[2], hi[3])};
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) out[e] = relu_bf16x2(out[e]);
                    }
                    if constexpr (ET != 2) {
                        if (a.q_inv > 0.f) {             // (uniform) quantise the 16-bit result exactly as quant_to_fp8_kernel would: 8 bytes per lane
                            const u32x2 o8 = (u32x2){
                                pack4_fp8(unpack_lo_e<ET>(out[0]) * a.q_inv, unpack_hi_e<ET>(out[0]) * a.q_inv, unpack_lo_e<ET>(out[1]) * a.q_inv, unpack_hi_e<ET>(out[1]) * a.q_inv),
                                pack4_fp8(unpack_lo_e<ET>(out[2]) * a.q_inv, unpack_hi_e<ET>(out[2]) * a.q_inv, unpack_lo_e<ET>(out[3]) * a.q_inv, unpack_hi_e<ET>(out[3]) * a.q_inv)};
                            __builtin_amdgcn_raw_buffer_store_b64(o8, rsrc_y, voff >> 1, 0, 0);
                            continue;
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(out, rsrc_y, voff, 0, 0);
                }
            }
        };
        if (total > 0) bias_fetch(c_tile);
        __builtin_amdgcn_s_barrier();                    // stage 0 landed
        R50_STAMP_DECL
        for (int g = 0; g < total; ++g) {
            if (c_k == 0) tile_begin();
            R50_MARK(0)                                  // tile begin
            compute();
#if defined(R50_STAMP)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 0" ::"v"(acc[MR - 1][NR - 1]) : "memory");   // the stamp waits for the last MFMA
#endif
            R50_MARK(1)                                  // fragment reads + MFMAs
            c_buf = (c_buf == NSTAGE - 1) ? 0 : c_buf + 1;
            if (++c_k == a.nk) {
                const int next_tile = c_tile + grid;
                if (next_tile < a.n_blocks) bias_fetch(next_tile);
                epilogue();
                c_k = 0;
                c_tile = next_tile;
            }
            R50_MARK(2)                                  // epilogue
            __builtin_amdgcn_s_barrier();
            R50_MARK(3)                                  // barrier
        }
        R50_STAMP_FLUSH(NCONS + NLOAD)
    }
#else
    (void)a;
#endif
}

// ------------------------------------------------------------------------------------------------
// 1x1 convs whose operands BOTH stream (layer3 / layer4: K = Cin >= 512, M = 12,544 ... 200,704 pixel rows, 256 ... 2,048 couts) as a
// plain GEMM on an EIGHT-PHASE schedule (round 4).  The generic kernels above run these shapes at 0.57-0.89 PFLOP/s: with one barrier per
// 64-deep K-step either every wave issues DMAs between its MFMAs (igemm_bf16_kernel) or a 128 x 224 tile needs 44 KB of L1 -> LDS feed per
// 896 MFMA cycles (igemm_ws_kernel: 77 % of the feed path at the MFMA rate).  Here:
//   * tile 256 couts x (16 NR x 2) pixels, NR = 8 (256 pixels) or 7 (224 = 14 x 16 divides the pixel counts 2^k * 49 of this network):
//     32 KB + 32 KB of operands per 2,048 MFMA cycles = half the feed of the tiles above;
//   * 8 waves = 2 (pixel halves, `wr`) x 4 (cout quarters, `wc`); wave tile 64 couts x 16 NR pixels = (4 x NR) accumulator blocks;
//   * a K-tile (64 deep) is cut into four HALF-TILE SLOTS of 128 LDS rows x 128 B, in the order the phases read them:
//       slot 0 = b0: W rows of the first 32 couts of every cout quarter        (read in phase 1: 4 ds_read_b128 per wave)
//       slot 1 = a0: X rows of the first 64 pixels of either pixel half        (read in phase 1: 8)
//       slot 2 = b1: W rows of the second 32 couts                             (read in phase 2: 4)
//       slot 3 = a1: X rows of the remaining 16 (NR - 4) pixels                (read in phase 3: 8 or 6)
//     and a K-tile is FOUR PHASES, one accumulator quadrant each: (b0,a0) (b1,a0) (b1,a1) (b0,a1) = 16 MFMAs (12 in the last two for NR = 7).
//     phase:  fragment reads of the quadrant's new operand ; ONE half-tile slot staged (2 LDS-DMAs per thread) ; s_barrier ;
//             s_waitcnt lgkmcnt(0) ; 16 MFMAs ; s_barrier
//   * two K-tile buffers (2 x 64 KB of LDS); the staging stream runs SEVEN half-tiles ahead of the reads and is waited for with a
//     COUNTED vmcnt once per K-tile (phase 4: vmcnt(6) = the three youngest half-tiles may still be in flight, i.e. K-tile g + 1 has
//     landed), never vmcnt(0) inside the stream, raw s_barrier (a __syncthreads() would drain the DMAs);
//   * the two pixel halves (waves 0-3 / 4-7 = the two waves of every SIMD) run ONE BARRIER APART: while one half issues its 16 MFMAs
//     the other one reads fragments and issues DMAs, then they swap.
// Hand-over rules (guide: "read a staged buffer one phase AFTER the wait that retires it"), with B(n) the n-th barrier, half 0 in
// [mem(p) B(2p-1) mfma(p) B(2p)] and half 1 one barrier later:
//   RAW: phase 4's vmcnt is passed by half 0 before B7 and by half 1 before B8; K-tile g + 1 is first read after B8 (half 0) / B9.
//   WAR: slot 0 is re-staged in phase 2 (first by half 0, after B2): half 1's b0 reads of phase 1 are retired by the lgkmcnt(8) it passes
//        before B2 (b0 is read FIRST in phase 1; an LDS-DMA does not count in lgkmcnt).  Slots 1, 2, 3 are re-staged in phases 3, 4, 5:
//        the reads of phases 1, 2, 3 are retired by the reading wave's lgkmcnt(0) at least one full barrier before.
// The (tile, K-tile) pairs of a workgroup form ONE stream through the two buffers (the next tile's first seven half-tiles are in flight
// while a tile's epilogue runs).  Same K order per accumulator as igemm_bf16_kernel (bias first, K-tiles ascending, two 32-deep MFMAs
// per K-tile), same cout permutation inside 32-row groups, same epilogue: the bits of tile ids 9 / 12.
// Operands: 1x1 conv, pad 0, any stride; DUAL: second K source at its own stride (conv3 + downsample of layer3.0 / layer4.0 as one GEMM).
// ------------------------------------------------------------------------------------------------
template <int ET, int NR, bool DUAL>
__global__ __launch_bounds__(512) void gemm8p_kernel(const ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(NR == 8 || NR == 7, "pixel blocks per wave: 8 (256-pixel tile) or 7 (224)");
    constexpr int BC = 256, BP = 32 * NR;
    constexpr int NA1 = NR - 4;                   // pixel blocks of the a1 sub-tile
    constexpr int SLOT = 16384, KBUF = 4 * SLOT;  // b0 | a0 | b1 | a1
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int srow = tid >> 3;                    // 0..63: LDS row of a staging pass
    const int lchunk = (tid & 7) ^ (srow & 7);    // logical 16-B chunk of K this thread fetches (XOR on the SOURCE side)

    const int grid = gridDim.x;
    const int first = xcd_remap(blockIdx.x, grid);
    const int my_tiles = (a.n_blocks - first + grid - 1) / grid;
    const int total = my_tiles * a.nk;            // K-tiles of this workgroup's stream

    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w), 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x), 0, a.x_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(DUAL ? a.x2 : a.x), 0, DUAL ? a.x2_records : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.res), 0, a.res ? a.y_bytes : 0u, 0x00020000);

    // ---------------- staging side ----------------------------------------------------------------
    unsigned w_voff[2][2], x_voff[2][2], x2_voff[DUAL ? 2 : 1][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {             // b-slot s, pass i: LDS row 64 i + srow = cout quarter (2 i + srow / 32), row srow % 32 of its 32-group
            const int q5 = srow & 31;
            const int cl = 128 * i + 64 * (srow >> 5) + 32 * s + ((q5 & 3) | (((q5 >> 4) & 1) << 2) | (((q5 >> 2) & 3) << 3));
            w_voff[s][i] = (unsigned)(cl * a.Ktot + lchunk * 8) * 2u;
        }
    int s_tile = first, s_kt = 0, s_buf = 0, s_rem = total, s_wbase = 0;
    auto decode_tile = [&](int tile) {            // per-lane X offsets of the tile being STAGED (a-slot s, pass i: pixel half i, pixel 64 s + srow of it)
        const int pt = (int)fast_div((unsigned)tile, a.div_ctiles);
        const int p0 = pt * BP;
        s_wbase = (tile - pt * a.n_ctiles) * BC * a.Ktot * 2;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = p0 + 16 * NR * i + 64 * s + srow;
                unsigned v = kOobOffset, v2 = kOobOffset;
                if ((s == 0 || srow < 16 * NA1) && m < a.M) {
                    const int n = (int)fast_div((unsigned)m, a.div_howo);
                    const int r = m - n * a.HoWo;
                    const int ho = (int)fast_div((unsigned)r, a.div_wo);
                    const int wo = r - ho * a.Wo;
                    v = (unsigned)(((n * a.H + ho * a.stride) * a.W + wo * a.stride) * a.x_cstride + lchunk * 8) * 2u;
                    if constexpr (DUAL) v2 = (unsigned)(((n * a.H2 + ho * a.stride2) * a.W2 + wo * a.stride2) * a.x2_cstride + lchunk * 8) * 2u;
                }
                x_voff[s][i] = v;
                if constexpr (DUAL) x2_voff[s][i] = v2;
            }
    };
    auto stage = [&](auto slot_c) {               // stage the next half-tile of the stream: slot `SL` of K-tile (s_tile, s_kt) into buffer s_buf
        constexpr int SL = decltype(slot_c)::value;
        if (s_rem > 0) {
            char* dst = smem + s_buf * KBUF + SL * SLOT + wave * 1024;
            if constexpr ((SL & 1) == 0) {        // b-slot: weights
                const int soff = __builtin_amdgcn_readfirstlane(s_wbase + s_kt * 128);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (LDS_AS void*)(dst + i * 8192), 16, w_voff[SL >> 1][i], soff, 0, 0);
            } else {                              // a-slot: pixels.  The second K source is chosen by VALUE selects: an if / else over the two sources
                                                  // makes hipcc merge the branches and index the offset arrays through scratch memory
                const bool second = DUAL && s_kt >= a.cc1;
                const int soff = __builtin_amdgcn_readfirstlane((second ? s_kt - a.cc1 : s_kt) * 128);
                const __amdgpu_buffer_rsrc_t rs = second ? rsrc_x2 : rsrc_x;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned v = second ? x2_voff[DUAL ? (SL >> 1) : 0][i] : x_voff[SL >> 1][i];
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(dst + i * 8192), 16, v, soff, 0, 0);
                }
            }
            if constexpr (SL == 3) {              // K-tile complete: advance the stream
                --s_rem;
                s_buf ^= 1;
                if (++s_kt == a.nk) {
                    s_kt = 0;
                    s_tile += grid;
                    if (s_tile < a.n_blocks) decode_tile(s_tile);
                }
            }
        }
    };
    using std::integral_constant;

    // ---------------- compute side ----------------------------------------------------------------
    f32x4 acc[4][NR];
    f32x4 bias_reg[4];
    const bool has_res = (a.res != nullptr);
    const int fphys0 = (fq ^ (fr & 7)) << 4;                        // kk = 0; kk = 1 is ^ 64
    const int w_fr[2] = {(32 * wc + fr) * 128 + fphys0, (32 * wc + fr) * 128 + (fphys0 ^ 64)};            // [kk]; + slot base (0 / 2 SLOT) + 2048 m'
    const int x_fr[2] = {SLOT + (64 * wr + fr) * 128 + fphys0, SLOT + (64 * wr + fr) * 128 + (fphys0 ^ 64)};  // [kk]; + 2 SLOT s + 2048 j'
    const int cout_lane = 64 * wc + 8 * fq;
    const int pix_lane = 16 * NR * wr + fr;
    int c_tile = first, c_kt = 0;
    unsigned y_voff = 0u;
    const unsigned y_rowstep = (unsigned)(16 * a.y_cstride * 2);
    auto bias_fetch = [&](int tile) {
        const int pt = (int)fast_div((unsigned)tile, a.div_ctiles);
        const int c0 = (tile - pt * a.n_ctiles) * BC;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bias_reg[2 * t] = *reinterpret_cast<const f32x4*>(a.bias + c0 + cout_lane + 32 * t);
            bias_reg[2 * t + 1] = *reinterpret_cast<const f32x4*>(a.bias + c0 + cout_lane + 32 * t + 4);
        }
    };
    auto tile_begin = [&]() {
        const int pt = (int)fast_div((unsigned)c_tile, a.div_ctiles);
        const int c0 = (c_tile - pt * a.n_ctiles) * BC, p0 = pt * BP;
        y_voff = (unsigned)((p0 + pix_lane) * a.y_cstride + c0 + cout_lane) * 2u;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < NR; ++j) acc[m][j] = bias_reg[m];
    };
    auto epilogue = [&]() {                       // (+ residual) -> 16 bit -> ReLU on the packed pair -> one 16-B store per (cout pair block, pixel block)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            u32x4 rres[2];
            if (has_res) {
#pragma unroll
                for (int t = 0; t < 2; ++t) rres[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, y_voff + j * y_rowstep + 64 * t, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 lo = acc[2 * t][j], hi = acc[2 * t + 1][j];
                if (has_res) {
                    const u32x4 r = rres[t];
                    lo[0] += unpack_lo_e<ET>(r[0]); lo[1] += unpack_hi_e<ET>(r[0]);
                    lo[2] += unpack_lo_e<ET>(r[1]); lo[3] += unpack_hi_e<ET>(r[1]);
                    hi[0] += unpack_lo_e<ET>(r[2]); hi[1] += unpack_hi_e<ET>(r[2]);
                    hi[2] += unpack_lo_e<ET>(r[3]); hi[3] += unpack_hi_e<ET>(r[3]);
                }
                u32x4 out = (u32x4){pack2_e<ET>(lo[0], lo[1]), pack2_e<ET>(lo[2], lo[3]), pack2_e<ET>(hi[0], hi[1]), pack2_e<ET>(hi[2], hi[3])};
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) out[e] = relu_bf16x2(out[e]);
                }
                __builtin_amdgcn_raw_buffer_store_b128(out, rsrc_y, y_voff + j * y_rowstep + 64 * t, 0, 0);
            }
        }
    };
#define G8_BARRIER() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
#define G8_LGKM(n) { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define G8_PRIO_UP() __builtin_amdgcn_s_setprio(1)     // s_setprio around every MFMA cluster (guide T5: keeps hipcc from moving MFMAs across the barriers)
#define G8_PRIO_DN() __builtin_amdgcn_s_setprio(0)

    if (total > 0) {
        bias_fetch(c_tile);
        decode_tile(s_tile);
        stage(integral_constant<int, 0>{}); stage(integral_constant<int, 1>{}); stage(integral_constant<int, 2>{}); stage(integral_constant<int, 3>{});
        stage(integral_constant<int, 0>{}); stage(integral_constant<int, 1>{}); stage(integral_constant<int, 2>{});
        if (total >= 2) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    G8_BARRIER();
    if (wr == 1) G8_BARRIER();                    // the second pixel half runs one barrier behind the first

    bf16x8 xf[4][2], wf0[2][2], wf1[2][2];        // X fragments of the current a sub-tile; W fragments of b0 (live over the K-tile) and b1
    for (int g = 0; g < total; ++g) {
        const char* kb = smem + (g & 1) * KBUF;
        if (c_kt == 0) tile_begin();
        // ---- phase 1: read b0, a0 ; stage slot 3 of K-tile g + 1 ; quadrant (b0, a0)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) wf0[m][kk] = *reinterpret_cast<const bf16x8*>(kb + w_fr[kk] + m * 2048);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) xf[j][kk] = *reinterpret_cast<const bf16x8*>(kb + x_fr[kk] + j * 2048);
        __builtin_amdgcn_sched_barrier(0);
        stage(integral_constant<int, 3>{});
        G8_LGKM(8)                             // phase 1 retires only its four b0 reads before the barrier
        G8_BARRIER();
        G8_LGKM(0)
        G8_PRIO_UP();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[m][j] = mfma_e<ET>(wf0[m][kk], xf[j][kk], acc[m][j]);
        G8_PRIO_DN();
        G8_BARRIER();
        // ---- phase 2: read b1 ; stage slot 0 of K-tile g + 2 ; quadrant (b1, a0)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) wf1[m][kk] = *reinterpret_cast<const bf16x8*>(kb + 2 * SLOT + w_fr[kk] + m * 2048);
        __builtin_amdgcn_sched_barrier(0);
        stage(integral_constant<int, 0>{});
        G8_BARRIER();
        G8_LGKM(0)
        G8_PRIO_UP();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[2 + m][j] = mfma_e<ET>(wf1[m][kk], xf[j][kk], acc[2 + m][j]);
        G8_PRIO_DN();
        G8_BARRIER();
        // ---- phase 3: read a1 ; stage slot 1 of K-tile g + 2 ; quadrant (b1, a1)
#pragma unroll
        for (int j = 0; j < NA1; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) xf[j][kk] = *reinterpret_cast<const bf16x8*>(kb + 2 * SLOT + x_fr[kk] + j * 2048);
        __builtin_amdgcn_sched_barrier(0);
        stage(integral_constant<int, 1>{});
        G8_BARRIER();
        G8_LGKM(0)
        G8_PRIO_UP();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < NA1; ++j) acc[2 + m][4 + j] = mfma_e<ET>(wf1[m][kk], xf[j][kk], acc[2 + m][4 + j]);
        G8_PRIO_DN();
        G8_BARRIER();
        // ---- phase 4: stage slot 2 of K-tile g + 2 ; K-tile g + 1 landed ; quadrant (b0, a1)
        stage(integral_constant<int, 2>{});
        if (g + 2 < total) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        G8_BARRIER();
        G8_PRIO_UP();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < NA1; ++j) acc[m][4 + j] = mfma_e<ET>(wf0[m][kk], xf[j][kk], acc[m][4 + j]);
        G8_PRIO_DN();
        G8_BARRIER();
        if (++c_kt == a.nk) {
            const int next_tile = c_tile + grid;
            if (next_tile < a.n_blocks) bias_fetch(next_tile);
            epilogue();
            c_kt = 0;
            c_tile = next_tile;
        }
    }
    if (wr == 0) G8_BARRIER();                    // every wave passes the same number of barriers
#undef G8_BARRIER
#undef G8_LGKM
#undef G8_PRIO_UP
#undef G8_PRIO_DN
#else
    (void)a;
#endif
}

// ------------------------------------------------------------------------------------------------
// conv 3x3 / stride 1 / pad 1, 64 -> 64 channels on 56x56 images (layer1.*.conv2), weights resident in LDS.
// Why its own kernel: with Cin = Cout = 64 the whole filter bank is 72 KB, so it can stay in LDS for the launch,
// and the input only has to be staged ONCE per tile (with its halo) instead of once per tap: the generic implicit
// GEMM moves 24 KB from L2 into LDS per 128-pixel K-step (1.7 KB per output pixel) and its 16-MFMA steps are
// mostly barrier and DMA-issue overhead; here it is 0.2 KB per pixel and ONE barrier per 224-pixel tile.
// Tile = 4 output rows x 56 columns (an image is 14 tiles).  LDS: [9 taps][64 couts][128 B] weights + two input
// images of 6 rows x 58 columns x 128 B (zero border by out-of-range DMA) = exactly 160 KB.
// 4 consumer waves (2 cout halves x 2 pixel halves, wave tile 32 couts x 112 pixels) run 252 MFMAs per tile
// straight from LDS; 4 loader waves DMA the next tile's input while they do.
// Input image row: pixel (rr, col) at (rr*58 + col)*128, 16-B chunk c stored at c ^ (col & 7)  -- keyed on the
// COLUMN so that a tap's row shift (kh*58 rows) is an immediate offset and only the 3 column shifts need their
// own address registers.
// ------------------------------------------------------------------------------------------------
template <int ET>
__global__ __launch_bounds__(512) void conv3x3_c64_kernel(const ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int W_BYTES = 9 * 64 * 128;             // 73,728
    constexpr int XROWS_LDS = 352;                    // 6*58 = 348 rows, staged in 11 passes of 32
    constexpr int X_BYTES = XROWS_LDS * 128;          // 45,056
    constexpr int NPASS = 11;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // filter bank -> LDS: [tap][row rho][chunk ^ (rho & 7)], row rho holds cout perm(rho) (8 consecutive couts per lane)
    for (int i = tid; i < 9 * 64 * 8; i += 512) {
        const int c = i & 7, rho = (i >> 3) & 63, tap = i >> 9;
        const int cl = (rho & ~31) | (rho & 3) | (((rho >> 4) & 1) << 2) | (((rho >> 2) & 3) << 3);
        *reinterpret_cast<u32x4*>(smem + tap * 8192 + rho * 128 + ((c ^ (rho & 7)) << 4)) =
            *reinterpret_cast<const u32x4*>(a.w + (size_t)cl * 576 + tap * 64 + c * 8);
    }
    const int grid = gridDim.x;
    const int n_tiles = a.N * 14;
    const int first = blockIdx.x;
    const int my_tiles = (n_tiles - first + grid - 1) / grid;

    if (wave >= 4) {
        // =============================== loader waves ===============================================
        const int lw = wave - 4;
        const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(a.x) - a.x_back), 0, a.x_records, 0x00020000);
        // per pass: LDS row rho = pass*32 + lw*8 + (lane >> 3) -> (rr, col); tile-invariant
        unsigned rel[NPASS];
        unsigned top_mask = 0u, bot_mask = 0u, ok_mask = 0u;      // bit per pass
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int rho = i * 32 + lw * 8 + (lane >> 3);
            const int rr = rho / 58, col = rho - rr * 58;
            const int chunk = (lane & 7) ^ (col & 7);
            // relative to the pixel one row above and one column left of the tile's first output pixel
            rel[i] = (unsigned)((rr * 56 + col) * 128 + chunk * 16);
            if (rho < 348 && col >= 1 && col <= 56) ok_mask |= 1u << i;
            if (rr == 0) top_mask |= 1u << i;
            if (rr == 5) bot_mask |= 1u << i;
        }
        auto issue = [&](int tile, int buf) {
            const int n = tile / 14, tr = tile - n * 14;
            // x_back = (56 + 1) * 128: the descriptor starts one row and one pixel before the tensor
            const unsigned base = (unsigned)((n * 56 + tr * 4) * 56) * 128u;
            unsigned valid = ok_mask;
            if (tr == 0) valid &= ~top_mask;
            if (tr == 13) valid &= ~bot_mask;
            char* dst = smem + W_BYTES + buf * X_BYTES + lw * 1024;
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const unsigned voff = ((valid >> i) & 1u) ? base + rel[i] : kOobOffset;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (LDS_AS void*)(dst + i * 4096), 16, voff, 0, 0, 0);
            }
        };
        if (my_tiles > 0) issue(first, 0);
        __syncthreads();                               // weights staged (all 8 waves)
        for (int t = 0; t < my_tiles; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile t landed (nothing newer is in flight yet)
            __builtin_amdgcn_s_barrier();                          // ... and the consumers are done with tile t-1
            if (t + 1 < my_tiles) issue(first + (t + 1) * grid, (t + 1) & 1);
        }
    } else {
        // =============================== consumer waves =============================================
        const int wave_c = wave >> 1, wave_p = wave & 1;
        const int fr = lane & 15, fq = lane >> 4;
        const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
        const int cout_lane = wave_c * 32 + 8 * fq;
        const f32x4 b_lo = *reinterpret_cast<const f32x4*>(a.bias + cout_lane);
        const f32x4 b_hi = *reinterpret_cast<const f32x4*>(a.bias + cout_lane + 4);
        const int w_frag = (wave_c * 32 + fr) * 128 + ((fq ^ (fr & 7)) << 4);        // + tap*8192 + m*2048, ^64 for kk = 1
        // B fragment addresses inside an input image, per pixel block j and column shift kw (kk = 0; kk = 1 is ^ 64)
        int xa[7][3];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int p = wave_p * 112 + 16 * j + fr;
            const int r = p / 56, c = p - r * 56;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) xa[j][kw] = (r * 58 + c + kw) * 128 + ((fq ^ ((c + kw) & 7)) << 4);
        }
        __syncthreads();                               // weights staged
        for (int t = 0; t < my_tiles; ++t) {
            __builtin_amdgcn_s_barrier();              // tile t has landed
            const char* xb = smem + W_BYTES + (t & 1) * X_BYTES;
            f32x4 acc[2][7];
#pragma unroll
            for (int j = 0; j < 7; ++j) { acc[0][j] = b_lo; acc[1][j] = b_hi; }
            // 18 K-blocks (9 taps x 2 halves of 32 channels).  One wave per SIMD has nobody to hide LDS latency behind:
            // the 9 fragment reads of block i+1 are issued before the 14 MFMAs of block i (explicit double buffer,
            // order pinned with sched_group_barrier).
            bf16x8 wf[2][2], xf[2][7];
            auto read_block = [&](int it, bf16x8 (&w)[2], bf16x8 (&x)[7]) {
                const int tap = it >> 1, kk = it & 1, kh = tap / 3, kw = tap - kh * 3;
                const char* wt = smem + tap * 8192;
                w[0] = *reinterpret_cast<const bf16x8*>(wt + (w_frag ^ (kk << 6)));
                w[1] = *reinterpret_cast<const bf16x8*>(wt + 2048 + (w_frag ^ (kk << 6)));
#pragma unroll
                for (int j = 0; j < 7; ++j)
                    x[j] = *reinterpret_cast<const bf16x8*>(xb + kh * (58 * 128) + (xa[j][kw] ^ (kk << 6)));
            };
            read_block(0, wf[0], xf[0]);
#pragma unroll
            for (int it = 0; it < 18; ++it) {
                if (it + 1 < 18) read_block(it + 1, wf[(it + 1) & 1], xf[(it + 1) & 1]);
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    acc[0][j] = mfma_e<ET>(wf[it & 1][0], xf[it & 1][j], acc[0][j]);
                    acc[1][j] = mfma_e<ET>(wf[it & 1][1], xf[it & 1][j], acc[1][j]);
                }
                if (it + 1 < 18) __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 14, 0);
            }
            // ---- bias is in the accumulators; ReLU, bf16, 16-B stores
            const int tile = first + t * grid;
            const unsigned pix0 = (unsigned)(tile * 224 + wave_p * 112 + fr);
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const f32x4 lo = acc[0][j], hi = acc[1][j];
                u32x4 o = (u32x4){pack2_e<ET>(lo[0], lo[1]), pack2_e<ET>(lo[2], lo[3]), pack2_e<ET>(hi[0], hi[1]),
                                  pack2_e<ET>(hi[2], hi[3])};
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = relu_bf16x2(o[e]);
                }
                __builtin_amdgcn_raw_buffer_store_b128(o, rsrc_y, (pix0 + 16 * j) * 128u + cout_lane * 2, 0, 0);
            }
        }
    }
#else
    (void)a;
#endif
}

// ------------------------------------------------------------------------------------------------
// conv 3x3 / stride 1 / pad 1 with Cin = Cout (layer2 / layer3 / layer4 conv2 of the non-first blocks), INPUT-RESIDENT:
// a tile's input pixels -- with their halo -- are staged into LDS ONCE per 64-channel chunk and all nine taps read them there, so
// only the weights stream (the generic implicit GEMM re-stages the pixel rows for every tap: 9x the activation traffic from L2 into
// LDS, which at a CU's ~40 B/clk of L2 feed makes its 128x224 tile feed-bound: 44 KB per 896 MFMA cycles).
// Tile = 196 output pixels x 128 output channels, held as ROW BLOCKS: the padded row is 16 / 32 positions wide and an MFMA column block is
// 16 consecutive POSITIONS, 14 blocks per tile, 7 per pixel half:
//     14x14 images: one image (NI = 1, TR = 14): a 14-pixel image row + 2 unused slots per block       512 tiles at batch 256 = two full rounds
//     28x28 images: a band of 7 rows (NI = 1, TR = 7): half of a 28-pixel row per block                 1,024 tiles
//      7x7  images: four images (NI = 4, TR = 7): a row of 16 positions holds the same image row of TWO images with shared zero columns
//                   (see the kernel's first lines)                                                       256 tiles
// A fragment address is (lane constant of the tap column and K half) + (block + kernel row) * 2048 + buffer: the K loop carries NO vector-ALU
// address arithmetic and every ds_read is base + immediate (2 VALU instructions per 28 MFMAs).
// LDS: two input buffers [padded positions][128 B] (zero border by out-of-range DMA; the next chunk's rows are fetched while the current
// chunk's nine taps run) + a ring of three weight stages [128 rows][128 B].  16-B chunk c of a position at c ^ (column & 7), of a weight row
// at c ^ (row & 7).  One K-step = one tap of one chunk; the K order is (chunk, tap, channel) -- not the generic kernel's (tap, chunk, channel)
// -- so sums differ from its in the last fp32 bits; these shapes therefore ALWAYS take this kernel, at every batch size (a frame's features
// must not depend on the batch it travels in).
// 8 consumer waves (cout group w & 3, pixel half w >> 2; waves w and w + 4 share a SIMD) + 4 loader waves on a static schedule (the nine
// iterations of a chunk unrolled: compile-time pass indices, one scalar compare for the wait).
// Rounds 2-3 carried eleven more schedules of this kernel -- the 13-block addressing it started with, barriers in mid-step, rings of 4-5 stages,
// staggered SIMD partners, 256 couts per tile, three taps per step, the 32x32x16 MFMA with four consumer waves -- all bit-identical except the last,
// all slower or equal (profiles/r03_xres_variants.txt, r02_ablations_conv3x3_xres.txt); round 4 removed them (git history has them).
// ------------------------------------------------------------------------------------------------
template <int ET, int NI, int TR, int IW, int IH>
__global__ __launch_bounds__(768) void conv3x3_xres_kernel(const ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    // the one form that is left of rounds 2-3's schedule variants: 128 couts per tile, one tap per step, three weight stages, row blocks
    constexpr int BC = 128, TPS = 1, NST = 3, SCHED = 0, RB = 1;
    // RB 1 at 7x7 (NI = 4): a row of 16 positions holds the same image row of TWO images with shared zero columns, [0 | A0..A6 | 0 | B0..B6] (B's right
    // border is the next row's first position), and the two image pairs sit on top of each other with a shared zero row: 17 rows.  An output row
    // (16 slots, 14 of them pixels: the 12.5 % of the 14x14 form) is one block; blocks 0..6 = pair 0, 8..14 = pair 1 (block 7 is the shared
    // zero row and is never computed): wave half p takes blocks 8 p + 0..6.
    constexpr bool RB7 = RB && IW == 7;
    static_assert(RB == 0 || (TPS == 1 && (NST == 3 || RB == 1) && SCHED == 0 && ((NI == 1 && (IW == 14 || IW == 28) && TR * (IW == 14 ? 16 : 32) == 224) || (NI == 4 && IW == 7 && TR == 7 && RB == 1))),
                  "row blocks: 14 blocks of 16 positions");
    static_assert(RB != 3 || BC == 128, "32-wide blocks: four consumer waves of 32 couts");
    constexpr int NCW = RB == 3 ? 4 : 8;                              // consumer waves
    constexpr int PW = RB ? (IW == 28 ? 32 : 16) : IW + 2, PP = (TR + 2) * PW, PPT = RB7 ? 17 * 16 : NI * PP;     // padded positions per panel / per tile
    constexpr int BSTEP = RB7 ? 8 : 7;                                // first block of the second slot half
    constexpr int XPASS = (PPT + 31) / 32, XBUF = XPASS * 32 * 128;
    constexpr int NPX = NI * TR * IW, NBLK = RB ? 14 : (NPX + 15) / 16;         // 196 pixels, 13 blocks (row blocks: 14)
    static_assert(RB || NBLK == 13, "tiles are 196 pixels");
    constexpr int NB = IH / TR;                                       // row bands per image
    constexpr int MR = BC / 64;                                       // 16-row cout blocks per consumer wave (4 cout groups)
    // TPS taps per K-step: 1 (ring of 3 stages, 2 in flight) or 3 = a whole kernel row (ring of 2 long stages, 1 in flight): a step has a
    // fixed cost of ~500 cycles (barrier skew, first fragments' LDS latency), so longer steps spend less of their time on it
    static_assert(TPS == 1 || TPS == 3, "one tap or one kernel row per step");
    constexpr int SPC = 9 / TPS;                                      // steps per chunk
    constexpr int WPASS = TPS * BC / 32, WSTAGE = TPS * BC * 128, NSTAGE = NST, D = NSTAGE - 1;
    static_assert(D >= 1 && D <= 5, "1 to 5 weight stages in flight");
    constexpr int WRING = 2 * XBUF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#if defined(R50_STAMP)
    const unsigned long long t_entry = __builtin_readcyclecounter();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grid = gridDim.x;
    // XCD_REMAP: the workgroups of one XCD (ids b, b + 8, ..) take a contiguous run of tile ids, so the cout tiles of a pixel tile share one L2
    // (the input tile is fetched once instead of once per cout tile)
    const int first = XCD_REMAP ? xcd_remap(blockIdx.x, grid) : (int)blockIdx.x;
    const int nct = a.Cout / BC;
    const int n_tiles = a.n_blocks;                                   // pixel tiles x cout tiles (cout tile fastest)
    const int my_tiles = (n_tiles - first + grid - 1) / grid;
    const int cch = a.cin_chunks;
    const int spt = SPC * cch;                                        // steps per tile
    const int total = my_tiles * spt;

    if (wave >= NCW) {
        // =============================== loader waves ===============================================
        const int lw = wave - NCW;
        const int lt = tid - NCW * 64;
        const int srow = lt >> 3, slot = lt & 7;
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w), 0, a.w_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x), 0, (unsigned)a.M * (unsigned)(a.Cin * 2), 0x00020000);
        unsigned w_voff[WPASS], x_voff[XPASS];
        auto decode_w = [&](int tile) {
            const int c0 = (tile % nct) * BC;
#pragma unroll
            for (int i = 0; i < WPASS; ++i) {
                const int row = i * 32 + srow;                           // stage row: tap row / BC of the step, channel row % BC
                const int tt = row / BC, rho = row - tt * BC;
                // cout of LDS row rho.  16-wide MFMA: lane (fq) of block pair (2t, 2t+1) ends up with channels 32t + 8 fq .. + 7.  32-wide: register r of
                // lane half h is row (r & 3) + 8 (r >> 2) + 4 h of the block, and holds cout 16 h + r
                const int cl = RB == 3 ? ((rho & ~31) | (rho & 3) | (((rho >> 3) & 3) << 2) | (((rho >> 2) & 1) << 4))
                                       : ((rho & ~31) | (rho & 3) | (((rho >> 4) & 1) << 2) | (((rho >> 2) & 3) << 3));
                const int wkey = RB == 3 ? ((rho >> 1) & 7) : (rho & 7);
                w_voff[i] = (unsigned)((c0 + cl) * a.Ktot + tt * a.Cin + (slot ^ wkey) * 8) * 2u;
            }
        };
        auto decode_x = [&](int tile) {           // source offsets of the padded positions of `tile` (out of range = zero border)
            const int pt = tile / nct;
            const int band = pt % NB, n0 = (pt / NB) * NI;
#pragma unroll
            for (int i = 0; i < XPASS; ++i) {
                const int q = i * 32 + srow;
                int panel = q / PP, rem = q - panel * PP;
                int rr = rem / PW, cc = rem - rr * PW;
                int n = n0 + panel, y = band * TR + rr - 1, x = cc - 1;
                if constexpr (RB7) {              // row rho = q / 16 of 17, column kappa = q % 16: pair rho >= 9, image of the pair kappa >= 8
                    rr = q >> 4; cc = q & 15; panel = 0;
                    const int pr = rr >= 9 ? 1 : 0;
                    n = n0 + 2 * pr + (cc >> 3); y = rr - 1 - 8 * pr; x = (cc & 7) - 1;
                }
                const bool ok = q < PPT && tile < n_tiles && n < a.N && (unsigned)y < (unsigned)IH && (unsigned)x < (unsigned)IW;
                // 16-B chunk c of a position is stored at c ^ key, key = (panel * TR * IW + rr * IW + cc) & 7: for the 16 consecutive output pixels
                // of an MFMA column block the positions a tap reads then have CONSECUTIVE keys also across row and image wraps (keyed on the
                // padded position q itself, a wrap shifts the key by PW - IW = 2 and the block's reads collide: 41 % of the LDS cycles at
                // 14x14 / 7x7 were bank conflicts)
                // (row blocks: PW = 0 mod 8, so the key of a position is its column's, cc & 7, whatever block and kernel row it is read for)
                const int key = RB == 3 ? ((cc >> 1) & 7) : RB ? (cc & 7) : ((panel * (TR * IW) + rr * IW + cc) & 7);
                x_voff[i] = ok ? (unsigned)(((n * IH + y) * IW + x) * a.Cin + (slot ^ key) * 8) * 2u : kOobOffset;
            }
        };
        {
            // ---- STATIC loader schedule (round 3): the nine iterations of a chunk are unrolled, every iteration issues its weight stage (stage
            // g + 2) and ONE pass of the next chunk's input (two at iteration 0 when the buffer has nine passes) with compile-time pass indices,
            // and waits with one scalar compare.  (The dynamic loop it replaced spent ~85 scalar instructions and ~20 branches per iteration on the
            // same decisions; beside two consumer waves per SIMD that is time in which its DMAs are not being issued.  Removed in round 4.)
            // Hazards as the dynamic loop: a pass issued at iteration i (behind barrier i - 1, i.e. after every step of the chunk before this one)
            // is confirmed by the wait of iteration i + 1 and read from the next chunk's step 0 = iteration 9 on; last pass at iteration 7.
            int w_tile = first, w_c = 0, w_s = 0, w_buf = 0;
            auto w_stage = [&]() {
                const int wofs = __builtin_amdgcn_readfirstlane((w_s * a.Cin + w_c * 64) * 2);
                char* sbase = smem + WRING + w_buf * WSTAGE + lw * 1024;
#pragma unroll
                for (int i = 0; i < WPASS; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (LDS_AS void*)(sbase + i * 4096), 16, w_voff[i], wofs, 0, 0);
                w_buf = (w_buf == 2) ? 0 : w_buf + 1;
                if (++w_s == 9) {
                    w_s = 0;
                    if (++w_c == cch) { w_c = 0; w_tile += grid; if (w_tile < n_tiles) decode_w(w_tile); }
                }
            };
            int x_par = 0;
            auto x_one = [&](int pass, int chunk) {                  // pass: compile-time after unrolling
                const int xofs = __builtin_amdgcn_readfirstlane(chunk * 128);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (LDS_AS void*)(smem + x_par * XBUF + pass * 4096 + lw * 1024), 16, x_voff[pass], xofs, 0, 0);
            };
            decode_w(first);
            decode_x(first);
#pragma unroll
            for (int p = 0; p < XPASS; ++p) x_one(p, 0);
            w_stage();
            w_stage();                                                // total >= 9
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPASS) : "memory");          // the first chunk's input and stage 0 landed
            __builtin_amdgcn_s_barrier();
            R50_STAMP_DECL
            int t_cur = first, c_cur = 0;
            for (int g = 0; g < total; g += 9) {
                int t_nxt = t_cur, c_nxt = c_cur + 1;
                if (c_nxt == cch) { c_nxt = 0; t_nxt += grid; }
                const bool has_nxt = t_nxt < n_tiles;
                x_par ^= 1;                                           // the buffer the NEXT chunk goes into
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    const bool w_ok = g + i + 2 < total;
                    if (w_ok) w_stage();
                    constexpr int EXTRA = XPASS - 8;                  // 0 or 1
                    if (has_nxt) {
                        if (i == 0) {
                            if (t_nxt != t_cur) decode_x(t_nxt);
                            x_one(0, c_nxt);
                            if constexpr (EXTRA) x_one(1, c_nxt);
                        } else if (i < 8) {
                            x_one(i + EXTRA, c_nxt);
                        }
                    }
                    const int nx = has_nxt ? (i == 0 ? 1 + EXTRA : (i < 8 ? 1 : 0)) : 0;
                    R50_MARK(0)                           // DMA issue
                    if (w_ok && nx == 1) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPASS + 1) : "memory"); }
                    else wait_vmcnt((w_ok ? WPASS : 0) + nx);        // everything issued before this iteration has landed
                    R50_MARK(1)                           // wait landed
                    __builtin_amdgcn_s_barrier();
                    R50_MARK(2)                           // barrier
                }
                t_cur = t_nxt; c_cur = c_nxt;
            }
            R50_STAMP_FLUSH(12)
        }
    } else {
        // =============================== consumer waves =============================================
        const int wave_c = wave & 3, wave_p = wave >> 2;
        const int fr = lane & 15, fq = lane >> 4;
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
        const int w_row = (wave_c * MR * 16 + fr) * 128;                // + m*2048
        const int w_ph0 = (fq ^ (fr & 7)) << 4;                         // kk = 0; kk = 1 is ^ 64
        const int cout_lane = wave_c * MR * 16 + 8 * fq;
        // this lane's pixel of each of the wave's blocks: padded position at tap (0,0) (block 13 of the second half does not exist)
        int q0[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            int p = 16 * (7 * wave_p + j) + fr;
            p = p < NPX ? p : NPX - 1;
            const int panel = p / (TR * IW), rem = p - panel * (TR * IW);
            const int r = rem / IW, c = rem - r * IW;
            q0[j] = panel * PP + r * PW + c;
        }
        f32x4 acc[MR][7];
        int c_buf = 0;
        // RB: see the kernel's head.  Ring slot of tap s is s % 3 (nine steps per chunk, three stages), so with the nine taps unrolled every LDS
        // address of the K loop is a lane constant + an immediate.
        R50_STAMP_DECL
        auto run_steps_rb = [&]() {
            constexpr int NRW = 7, PD = 3;            // PD: pixel fragments read ahead of their MFMAs
            constexpr bool W2 = (MR <= 2);
            const char* const xl = smem + BSTEP * 2048 * wave_p;       // this pixel half's first block
            int vb[3][2];                                               // [tap column][K half]: position fr + kw, chunk (fq + 4 kk) ^ key
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) vb[kw][kk] = (fr + kw) * 128 + (((fq + 4 * kk) ^ ((fr + kw) & 7)) << 4);
            const char* const wl0 = smem + WRING + w_row + w_ph0;
            const char* const wl1 = smem + WRING + w_row + (w_ph0 ^ 64);
            int x_par = 0;
            for (int tile = first; tile < n_tiles; tile += grid) {
                const int c0 = (tile % nct) * BC;
                R50_MARK(0)                                             // tile begin
#pragma unroll
                for (int t = 0; t < MR / 2; ++t) {
                    const f32x4 b_lo = *reinterpret_cast<const f32x4*>(a.bias + c0 + cout_lane + 32 * t);
                    const f32x4 b_hi = *reinterpret_cast<const f32x4*>(a.bias + c0 + cout_lane + 32 * t + 4);
#pragma unroll
                    for (int j = 0; j < NRW; ++j) { acc[2 * t][j] = b_lo; acc[2 * t + 1][j] = b_hi; }
                }
                for (int c = 0; c < cch; ++c) {
                    const char* const xc = xl + x_par * XBUF;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int kh = tap / 3, kw = tap - 3 * kh;
                        const char* const x0 = xc + vb[kw][0] + kh * 2048 * (PW / 16);
                        const char* const x1 = xc + vb[kw][1] + kh * 2048 * (PW / 16);
                        // ring slot: static with three stages (9 taps per chunk); a deeper ring (A/B knob) carries a running slot index
                        const char* const w0 = wl0 + (NSTAGE == 3 ? tap % 3 : c_buf) * WSTAGE;
                        const char* const w1 = wl1 + (NSTAGE == 3 ? tap % 3 : c_buf) * WSTAGE;
                        if constexpr (NSTAGE != 3) c_buf = (c_buf == NSTAGE - 1) ? 0 : c_buf + 1;
                        auto xread = [&](int t) {
                            return *reinterpret_cast<const bf16x8*>((t >= NRW ? x1 : x0) + (t % NRW) * 2048);
                        };
                        bf16x8 x[2 * NRW], wf[MR], wg[W2 ? MR : 1];
#pragma unroll
                        for (int m = 0; m < MR; ++m) wf[m] = *reinterpret_cast<const bf16x8*>(w0 + m * 2048);
                        if constexpr (W2) {
#pragma unroll
                            for (int m = 0; m < MR; ++m) wg[m] = *reinterpret_cast<const bf16x8*>(w1 + m * 2048);
                        }
#pragma unroll
                        for (int t = 0; t < PD; ++t) x[t] = xread(t);
#pragma unroll
                        for (int t = 0; t < 2 * NRW; ++t) {
                            if constexpr (!W2) {
                                if (t == NRW) {
#pragma unroll
                                    for (int m = 0; m < MR; ++m) wf[m] = *reinterpret_cast<const bf16x8*>(w1 + m * 2048);
                                }
                            }
#pragma unroll
                            for (int m = 0; m < MR; ++m) {
                                acc[m][t % NRW] = mfma_e<ET>((W2 && t >= NRW) ? wg[m] : wf[m], x[t], acc[m][t % NRW]);
                            }
                            if (t + PD < 2 * NRW) x[t + PD] = xread(t + PD);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x100, (W2 ? 2 * MR : MR) + PD, 0);
#pragma unroll
                        for (int t = 0; t < 2 * NRW; ++t) {
                            if (!W2 && t == NRW) __builtin_amdgcn_sched_group_barrier(0x100, MR, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, MR, 0);
                            if (t + PD < 2 * NRW) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);              // the barrier stays behind the step's last fragment read
                        R50_MARK(1)                                     // fragment reads + MFMAs
                        __builtin_amdgcn_s_barrier();
                        R50_MARK(2)                                     // barrier
                    }
                    x_par ^= 1;
                }
                // ---- epilogue: ReLU, 16-bit, one 16-B store per valid slot and block pair (slot = 16 b + fr = padded row * PW + column)
                const int pt = tile / nct;
                const int band = pt % NB;
                int n = pt / NB;
#pragma unroll
                for (int j = 0; j < NRW; ++j) {
                    const int sl = 16 * (BSTEP * wave_p + j) + fr;
                    int r = sl / PW, cx = sl - r * PW;
                    if constexpr (RB7) { n = (pt / NB) * NI + 2 * wave_p + (fr >> 3); r = j; cx = fr & 7; }     // block 8 p + j = row j of image pair p
                    const bool ok = cx < IW && n < a.N;
                    const unsigned pix = (unsigned)((n * IH + band * TR + r) * IW + cx);
#pragma unroll
                    for (int t = 0; t < MR / 2; ++t) {
                        const f32x4 lo = acc[2 * t][j], hi = acc[2 * t + 1][j];
                        u32x4 o = (u32x4){pack2_e<ET>(lo[0], lo[1]), pack2_e<ET>(lo[2], lo[3]), pack2_e<ET>(hi[0], hi[1]), pack2_e<ET>(hi[2], hi[3])};
                        if (a.relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = relu_bf16x2(o[e]);
                        }
                        const unsigned voff = ok ? (pix * (unsigned)a.Cout + (unsigned)(c0 + cout_lane + 32 * t)) * 2u : kOobOffset;
                        __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, voff, 0, 0);
                    }
                }
                R50_MARK(3)                                             // epilogue (and the next tile's bias loads land in slot 0)
            }
        };
        __builtin_amdgcn_s_barrier();             // step 0 and the first input chunk landed
#if defined(R50_STAMP)
        const unsigned long long clk0 = __builtin_readcyclecounter();
        st_sum[5] = clk0 - t_entry;               // kernel entry -> first stage landed (prologue)
        st_prev = clk0;
#endif
        run_steps_rb();
#if defined(R50_STAMP)
        st_sum[6] = __builtin_readcyclecounter() - clk0;      // shader cycles of the loop; slot 7 (100-MHz ticks, from before the first barrier) is set by the flush
#endif
        R50_STAMP_FLUSH(12)
    }
#else
    (void)a;
#endif
}

// ------------------------------------------------------------------------------------------------
// conv 3x3 / STRIDE 2 / pad 1 with Cin = Cout (conv2 of layer2.0 and layer3.0: 56 -> 28 x 128 channels, 28 -> 14 x 256), INPUT-RESIDENT by
// POLYPHASE PLANES.  Round 2 ran these two launches on the generic implicit GEMM (re-staging the pixel rows for every tap) at 571 and
// 825 TFLOP/s with 1.2x / 2.0x their input bytes fetched: the slowest MFMA-bound launches of the network.
// Output pixel (r, c), tap (kh, kw) reads input (2r + kh - 1, 2c + kw - 1): the parity of the input row is that of kh - 1, the parity of the
// column that of kw - 1.  So the nine taps fall into four PHASE PLANES of the input -- rows odd / even x columns odd / even -- and on its
// plane every tap is a STRIDE-1 access at offset (dr, dc) in {0,1}^2:
//     OO (odd rows y = 2R - 1, odd cols x = 2i - 1):  taps (0,0) (0,2) (2,0) (2,2)  at (R, i) = (r + kh/2, c + kw/2)
//     OE (odd rows, even cols x = 2i):                taps (0,1) (2,1)              at (r + kh/2, c)
//     EO (even rows y = 2R, odd cols):                taps (1,0) (1,2)              at (r, c + kw/2)
//     EE:                                             tap  (1,1)                    at (r, c)
// One plane of one 64-channel chunk and one tile = (TR + 1) rows x PW positions x 128 B = 32 KB -- the size of the stride-1 kernel's input
// buffer -- where the whole strided footprint of a tile would be 110-120 KB.  A K "virtual chunk" = (chunk, plane); its taps are steps with
// the row-block addressing of conv3x3_xres_kernel (RB 1): plane row = PW (16 / 32) positions, output slot s = PW r + c reads position
// s + dr PW + dc: lane constant + immediate, no address arithmetic in the K loop.  K order: (chunk, plane OO OE EO EE, tap, channel).
// That is not the generic kernel's order, so these two shapes take this kernel at EVERY batch size.
// LDS: THREE plane buffers (a plane of 1 step, EE, sits between planes of 2 and 4: with two buffers its successor could only be fetched
// while EE itself runs, one step) + ring of 3 weight stages = 144 KB.  Loader schedule per chunk (iteration i = barrier interval; a pass
// issued at iteration i is confirmed by the wait of iteration i + 1 and readable from step i + 2; a buffer is free for the plane three
// planes later from the first step of the plane two planes later):
//     plane          steps     its 8 passes are issued at iterations (of the chunk; 8 = the previous chunk's last)
//     OE             4,5       8 (2), 0 (4), 1 (2)          window [8', 2]
//     EO             6,7       1 (2), 2 (4), 3 (2)          window [0, 4]
//     EE             8         4 (3), 5 (3), 6 (2)          window [4, 6]
//     OO of c + 1    0..3      6 (4), 7 (4)                 window [6, 7]
// 8 consumer waves (cout group w & 3, slot half w >> 2: 7 + 7 blocks) + 4 loader waves, as the stride-1 kernel.
// ------------------------------------------------------------------------------------------------
// 14 -> 7 (layer4.0, OW = 7): FOUR images per tile as two pairs; a plane row of 16 positions holds plane row R of both images of a pair
// ([A i = 0..7 | B i = 0..7], i = 0 being x = -1 on the odd-column planes), the pairs' 8 plane rows sit on top of each other (16 rows = 256
// positions); an output row of a pair (16 slots, 14 of them pixels) is one block, blocks 8 p + 0..6 = pair p (as the 7x7 row-block form of
// conv3x3_xres_kernel).
template <int ET, int BC, int TR, int OW, int OH>
__global__ __launch_bounds__(768) void conv3x3_s2_kernel(const ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool PAIR = OW == 7;
    constexpr int PW = OW == 28 ? 32 : 16;
    static_assert(BC == 128 && (PAIR ? (TR == 7 && OH == 7) : ((OW == 14 || OW == 28) && TR * PW == 224 && OH % TR == 0)), "14 blocks of 16 slots, 128 couts");
    constexpr int IH = 2 * OH, IW = 2 * OW, NB = OH / TR;
    constexpr int NIMG = PAIR ? 4 : 1, BSTEP = PAIR ? 8 : 7;          // images per tile; first block of the second slot half
    constexpr int XPASS = 8, XBUF = XPASS * 32 * 128;                  // (TR + 1) * PW <= 256 positions
    static_assert(PAIR || (TR + 1) * PW <= 256, "plane buffer");
    constexpr int MR = BC / 64, WPASS = BC / 32, WSTAGE = BC * 128, NST = 3, D = 2;
    constexpr int WRING = 3 * XBUF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grid = gridDim.x, first = XCD_REMAP ? xcd_remap(blockIdx.x, grid) : (int)blockIdx.x;
    const int nct = a.Cout / BC;
    const int n_tiles = a.n_blocks;                                   // pixel tiles x cout tiles (cout tile fastest)
    const int my_tiles = (n_tiles - first + grid - 1) / grid;
    const int cch = a.cin_chunks;
    const int total = my_tiles * cch * 9;

    if (wave >= 8) {
        // =============================== loader waves ===============================================
        const int lw = wave - 8, lt = tid - 512;
        const int srow = lt >> 3, slot = lt & 7;
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w), 0, a.w_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x), 0, (unsigned)a.N * (unsigned)(IH * IW) * (unsigned)(a.Cin * 2), 0x00020000);
        unsigned w_voff[WPASS], x_voff[4][XPASS];
        auto decode_w = [&](int tile) {
            const int c0 = (tile % nct) * BC;
#pragma unroll
            for (int i = 0; i < WPASS; ++i) {
                const int rho = i * 32 + srow;
                const int cl = (rho & ~31) | (rho & 3) | (((rho >> 4) & 1) << 2) | (((rho >> 2) & 3) << 3);
                w_voff[i] = (unsigned)((c0 + cl) * a.Ktot + (slot ^ (rho & 7)) * 8) * 2u;
            }
        };
        auto decode_x = [&](int tile) {           // per plane (py, px): source offset of plane position (R, i) or out of range (zero border)
            const int pt = tile / nct;
            const int band = pt % NB;
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const int py = ph >> 1, px = ph & 1;
#pragma unroll
                for (int i = 0; i < XPASS; ++i) {
                    const int q = i * 32 + srow;
                    int R = q / PW, ii = q - R * PW, n = (pt / NB) * NIMG;
                    if constexpr (PAIR) { n += 2 * (R >> 3) + (ii >> 3); R &= 7; ii &= 7; }       // row 8 p + R of pair p, column 8 (image of the pair) + i
                    const int y = 2 * (band * TR + R) - 1 + py, x = 2 * ii - 1 + px;
                    const bool ok = tile < n_tiles && n < a.N && R <= TR && (unsigned)y < (unsigned)IH && (unsigned)x < (unsigned)IW;
                    x_voff[ph][i] = ok ? (unsigned)(((n * IH + y) * IW + x) * a.Cin + (slot ^ (ii & 7)) * 8) * 2u : kOobOffset;
                }
            }
        };
        // ---- weight stream: stage g = step (tile, chunk, s); tap of step s in plane order
        int i_tile = first, i_c = 0, i_s = 0, i_buf = 0;
        auto w_issue = [&]() -> int {
            const int tap = (i_s < 4) ? ((i_s & 1) * 2 + (i_s >> 1) * 6) : (i_s < 6) ? (1 + (i_s - 4) * 6) : (i_s < 8) ? (3 + (i_s - 6) * 2) : 4;
            const int wofs = __builtin_amdgcn_readfirstlane((tap * a.Cin + i_c * 64) * 2);
            char* sbase = smem + WRING + i_buf * WSTAGE + lw * 1024;
#pragma unroll
            for (int i = 0; i < WPASS; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (LDS_AS void*)(sbase + i * 4096), 16, w_voff[i], wofs, 0, 0);
            i_buf = (i_buf == NST - 1) ? 0 : i_buf + 1;
            if (++i_s == 9) {
                i_s = 0;
                if (++i_c == cch) { i_c = 0; i_tile += grid; if (i_tile < n_tiles) decode_w(i_tile); }
            }
            return WPASS;
        };
        // ---- plane stream: the planes are fetched in K order; x_buf = buffer of the plane whose passes are being issued
        int x_buf = 0;
        auto x_pass = [&](int ph, int pass, int chunk) {        // ph, pass: compile-time after unrolling (the plane's (py, px) is in x_voff)
            const int xofs = __builtin_amdgcn_readfirstlane(chunk * 128);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (LDS_AS void*)(smem + x_buf * XBUF + pass * 4096 + lw * 1024), 16, x_voff[ph][pass], xofs, 0, 0);
        };
        auto next_buf = [&]() { x_buf = (x_buf == 2) ? 0 : x_buf + 1; };
        decode_w(i_tile);
        decode_x(first);
        // prologue: plane OO of (first tile, chunk 0) whole, stage 0, the first two passes of plane OE, stage 1
#pragma unroll
        for (int p = 0; p < XPASS; ++p) x_pass(0, p, 0);
        if (total > 0) w_issue();
        next_buf();
        x_pass(1, 0, 0); x_pass(1, 1, 0);
        if (total > 1) w_issue();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPASS + 2) : "memory");       // plane OO and stage 0 landed
        __builtin_amdgcn_s_barrier();
        int t_cur = first, c_cur = 0;                                  // (tile, chunk) of the step the consumers run during this iteration
        for (int g = 0; g < total; g += 9) {
            // (tile, chunk) after this one, whose planes OO / OE are fetched from iteration 6 on
            int t_nxt = t_cur, c_nxt = c_cur + 1;
            if (c_nxt == cch) { c_nxt = 0; t_nxt += grid; }
            const bool has_nxt = t_nxt < n_tiles;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                int ops = 0;
                if (g + i + D < total) ops += w_issue();
                // plane passes of this iteration (table in the kernel's head)
                if (i == 0) { x_pass(1, 2, c_cur); x_pass(1, 3, c_cur); x_pass(1, 4, c_cur); x_pass(1, 5, c_cur); ops += 4; }
                if (i == 1) { x_pass(1, 6, c_cur); x_pass(1, 7, c_cur); next_buf(); x_pass(2, 0, c_cur); x_pass(2, 1, c_cur); ops += 4; }
                if (i == 2) { x_pass(2, 2, c_cur); x_pass(2, 3, c_cur); x_pass(2, 4, c_cur); x_pass(2, 5, c_cur); ops += 4; }
                if (i == 3) { x_pass(2, 6, c_cur); x_pass(2, 7, c_cur); next_buf(); ops += 2; }
                if (i == 4) { x_pass(3, 0, c_cur); x_pass(3, 1, c_cur); x_pass(3, 2, c_cur); ops += 3; }
                if (i == 5) { x_pass(3, 3, c_cur); x_pass(3, 4, c_cur); x_pass(3, 5, c_cur); ops += 3; }
                if (i == 6) {
                    x_pass(3, 6, c_cur); x_pass(3, 7, c_cur); next_buf(); ops += 2;
                    if (has_nxt) {
                        if (t_nxt != t_cur) decode_x(t_nxt);
                        x_pass(0, 0, c_nxt); x_pass(0, 1, c_nxt); x_pass(0, 2, c_nxt); x_pass(0, 3, c_nxt); ops += 4;
                    }
                }
                if (i == 7 && has_nxt) { x_pass(0, 4, c_nxt); x_pass(0, 5, c_nxt); x_pass(0, 6, c_nxt); x_pass(0, 7, c_nxt); next_buf(); ops += 4; }
                if (i == 8 && has_nxt) { x_pass(1, 0, c_nxt); x_pass(1, 1, c_nxt); ops += 2; }
                // everything issued before this iteration has landed (step g + i + 1's stage and planes): all but this iteration's `ops`.  In the steady
                // state `ops` is the table's figure for iteration i: one scalar compare instead of wait_vmcnt's six
                constexpr int QF[9] = {4, 4, 4, 2, 3, 3, 6, 4, 2};
                if (ops == WPASS + QF[i]) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPASS + QF[i]) : "memory"); }
                else wait_vmcnt(ops);
                __builtin_amdgcn_s_barrier();
            }
            t_cur = t_nxt; c_cur = c_nxt;
        }
    } else {
        // =============================== consumer waves =============================================
        constexpr int NRW = 7, PD = 3;
        const int wave_c = wave & 3, wave_p = wave >> 2;
        const int fr = lane & 15, fq = lane >> 4;
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
        const int w_row = (wave_c * MR * 16 + fr) * 128;
        const int w_ph0 = (fq ^ (fr & 7)) << 4;
        const int cout_lane = wave_c * MR * 16 + 8 * fq;
        int vb[2][2];                                                   // [dc][K half]: position fr + dc, chunk (fq + 4 kk) ^ key
#pragma unroll
        for (int dc = 0; dc < 2; ++dc)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) vb[dc][kk] = (fr + dc) * 128 + (((fq + 4 * kk) ^ ((fr + dc) & 7)) << 4);
        const char* const wl0 = smem + WRING + w_row + w_ph0;
        const char* const wl1 = smem + WRING + w_row + (w_ph0 ^ 64);
        f32x4 acc[MR][NRW];
        int x_buf = 0;
        __builtin_amdgcn_s_barrier();             // step 0: plane OO of the first chunk and stage 0 landed
        for (int tile = first; tile < n_tiles; tile += grid) {
            const int c0 = (tile % nct) * BC;
#pragma unroll
            for (int t = 0; t < MR / 2; ++t) {
                const f32x4 b_lo = *reinterpret_cast<const f32x4*>(a.bias + c0 + cout_lane + 32 * t);
                const f32x4 b_hi = *reinterpret_cast<const f32x4*>(a.bias + c0 + cout_lane + 32 * t + 4);
#pragma unroll
                for (int j = 0; j < NRW; ++j) { acc[2 * t][j] = b_lo; acc[2 * t + 1][j] = b_hi; }
            }
            for (int c = 0; c < cch; ++c) {
#pragma unroll
                for (int s = 0; s < 9; ++s) {
                    // step s of the chunk: plane, offset (dr, dc) on the plane (the loader's tap order)
                    const int dr = (s < 4) ? (s >> 1) : (s < 6) ? (s - 4) : 0;
                    const int dc = (s < 4) ? (s & 1) : (s >= 6 && s < 8) ? (s - 6) : 0;
                    const char* const xc = smem + x_buf * XBUF + BSTEP * 2048 * wave_p + dr * (PW * 128);
                    const char* const x0 = xc + vb[dc][0];
                    const char* const x1 = xc + vb[dc][1];
                    const char* const w0 = wl0 + (s % 3) * WSTAGE;
                    const char* const w1 = wl1 + (s % 3) * WSTAGE;
                    auto xread = [&](int t) { return *reinterpret_cast<const bf16x8*>((t >= NRW ? x1 : x0) + (t % NRW) * 2048); };
                    bf16x8 x[2 * NRW], wf[MR], wg[MR];
#pragma unroll
                    for (int m = 0; m < MR; ++m) wf[m] = *reinterpret_cast<const bf16x8*>(w0 + m * 2048);
#pragma unroll
                    for (int m = 0; m < MR; ++m) wg[m] = *reinterpret_cast<const bf16x8*>(w1 + m * 2048);
#pragma unroll
                    for (int t = 0; t < PD; ++t) x[t] = xread(t);
#pragma unroll
                    for (int t = 0; t < 2 * NRW; ++t) {
#pragma unroll
                        for (int m = 0; m < MR; ++m) acc[m][t % NRW] = mfma_e<ET>(t >= NRW ? wg[m] : wf[m], x[t], acc[m][t % NRW]);
                        if (t + PD < 2 * NRW) x[t + PD] = xread(t + PD);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * MR + PD, 0);
#pragma unroll
                    for (int t = 0; t < 2 * NRW; ++t) {
                        __builtin_amdgcn_sched_group_barrier(0x008, MR, 0);
                        if (t + PD < 2 * NRW) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);                  // the barrier stays behind the step's last fragment read
                    __builtin_amdgcn_s_barrier();
                    if (s == 3 || s == 5 || s == 7 || s == 8) x_buf = (x_buf == 2) ? 0 : x_buf + 1;      // the next step opens the next plane
                }
            }
            // ---- epilogue: ReLU, 16-bit, one 16-B store per valid slot and block pair (slot = 16 b + fr = output row * PW + column)
            const int pt = tile / nct;
            const int band = pt % NB;
            int n = (pt / NB) * NIMG;
            if constexpr (PAIR) n += 2 * wave_p + (fr >> 3);
#pragma unroll
            for (int j = 0; j < NRW; ++j) {
                const int sl = 16 * (BSTEP * wave_p + j) + fr;
                int r = sl / PW, cx = sl - r * PW;
                if constexpr (PAIR) { r = j; cx = fr & 7; }
                const bool ok = cx < OW && n < a.N;
                const unsigned pix = (unsigned)((n * OH + band * TR + r) * OW + cx);
#pragma unroll
                for (int t = 0; t < MR / 2; ++t) {
                    const f32x4 lo = acc[2 * t][j], hi = acc[2 * t + 1][j];
                    u32x4 o = (u32x4){pack2_e<ET>(lo[0], lo[1]), pack2_e<ET>(lo[2], lo[3]), pack2_e<ET>(hi[0], hi[1]), pack2_e<ET>(hi[2], hi[3])};
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = relu_bf16x2(o[e]);
                    }
                    const unsigned voff = ok ? (pix * (unsigned)a.Cout + (unsigned)(c0 + cout_lane + 32 * t)) * 2u : kOobOffset;
                    __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, voff, 0, 0);
                }
            }
        }
    }
#else
    (void)a;
#endif
}

// ------------------------------------------------------------------------------------------------
// Bottleneck tail (layer1): conv3 1x1 (64 -> 256) + bn3 + identity + ReLU, and the NEXT block's
// conv1 1x1 (256 -> C1) + bn1 + ReLU, in one pass over the pixels.
// Why: at 56x56 these two layers are HBM-bound (the block output is 2*M*256 bytes, written by conv3 and read
// straight back by the next conv1).  Fused, the block output is written once (the next block still needs it as
// its identity) but never re-read: 1.44 KB/pixel -> 1.28 KB/pixel of HBM traffic for the pair ... and one
// launch instead of two.
// How: a wave owns 16 pixels and ALL 256 channels.  With the cout permutation of the igemm kernel the packed
// bf16 result of conv3 in a lane -- channels 32t + 8*fq + 0..7 of pixel fr -- IS the B fragment of k-block t for
// the second GEMM, so the block output goes from accumulators to the next MFMA without leaving registers.
// Both weight matrices (32 KB + C1*512 B) stay in LDS for the whole launch; the activations never touch LDS
// (conv2's output is loaded straight into B fragments).  Persistent: one 8-wave workgroup per CU, each wave
// walks 16-pixel tiles and prefetches the next tile's inputs before it stores this tile's outputs (vmcnt
// retires in order).
// ------------------------------------------------------------------------------------------------
struct TailArgs {
    const __bf16* y2;     // (M, 64)   conv2 output
    const __bf16* w3;     // (256, 64) folded conv3 weights, K contiguous
    const float* b3;      // (256)
    const __bf16* res;    // (M, 256)  identity; with DS: (M, 64) the block INPUT, the identity is bf16(wd . input + bd)
    const __bf16* wd;     // DS only: (256, 64) folded downsample weights
    const float* bd;      // DS only: (256)
    __bf16* out;          // (M, 256)  block output
    const __bf16* w1;     // (C1, 256) folded weights of the next block's conv1
    const float* b1;      // (C1)
    __bf16* y1n;          // (M, C1)   next block's conv1 output
    int M;
};

// finer A/B knobs of the layer1 tail's streams (aux bits: 2 = nt): identity loads, block-output stores, conv2-output loads, next-t1 stores

template <int ET, int C1, bool DS, int NT>
__global__ __launch_bounds__(NT) void bneck_tail_kernel(const TailArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int W3_BYTES = 256 * 128;
    constexpr int W1_BYTES = C1 * 512;
    constexpr int WD_OFF = W3_BYTES + W1_BYTES;                 // DS: downsample weights, same image as w3
    constexpr int B3_OFF = WD_OFF + (DS ? W3_BYTES : 0);
    constexpr int B1_OFF = B3_OFF + 256 * 4;
    constexpr int BD_OFF = B1_OFF + C1 * 4;
    constexpr int M2 = C1 / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    // weights -> LDS once.  LDS row rho holds channel perm(rho) (see igemm: 8 consecutive couts per lane)
    for (int i = tid; i < 256 * 8; i += NT) {
        const int rho = i >> 3, c = i & 7;
        const int cl = (rho & ~31) | (rho & 3) | (((rho >> 4) & 1) << 2) | (((rho >> 2) & 3) << 3);
        *reinterpret_cast<u32x4*>(smem + rho * 128 + ((c ^ (rho & 7)) << 4)) =
            *reinterpret_cast<const u32x4*>(a.w3 + cl * 64 + c * 8);
    }
    for (int i = tid; i < C1 * 32; i += NT) {
        const int rho = i >> 5, c = i & 31;
        const int cl = (rho & ~31) | (rho & 3) | (((rho >> 4) & 1) << 2) | (((rho >> 2) & 3) << 3);
        *reinterpret_cast<u32x4*>(smem + W3_BYTES + rho * 512 + ((c ^ (rho & 15)) << 4)) =
            *reinterpret_cast<const u32x4*>(a.w1 + cl * 256 + c * 8);
    }
    if constexpr (DS) {
        for (int i = tid; i < 256 * 8; i += NT) {
            const int rho = i >> 3, c = i & 7;
            const int cl = (rho & ~31) | (rho & 3) | (((rho >> 4) & 1) << 2) | (((rho >> 2) & 3) << 3);
            *reinterpret_cast<u32x4*>(smem + WD_OFF + rho * 128 + ((c ^ (rho & 7)) << 4)) =
                *reinterpret_cast<const u32x4*>(a.wd + cl * 64 + c * 8);
        }
        for (int i = tid; i < 256; i += NT) reinterpret_cast<float*>(smem + BD_OFF)[i] = a.bd[i];
    }
    for (int i = tid; i < 256; i += NT) reinterpret_cast<float*>(smem + B3_OFF)[i] = a.b3[i];
    for (int i = tid; i < C1; i += NT) reinterpret_cast<float*>(smem + B1_OFF)[i] = a.b1[i];
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rs_y2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.y2), 0, (unsigned)a.M * 128u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.res), 0, (unsigned)a.M * (DS ? 128u : 512u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)a.M * 512u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y1 = __builtin_amdgcn_make_buffer_rsrc(a.y1n, 0, (unsigned)a.M * (C1 * 2u), 0x00020000);

    const int ntiles = (a.M + 15) >> 4;
    const int nwaves = gridDim.x * (NT / 64);
    const int w3_frag = fr * 128, fphys0 = (fq ^ (fr & 7)) << 4;
    const int w1_frag = W3_BYTES + fr * 512;

    // Inputs of a tile live in ONE register set: as soon as a register has been consumed, the load of the next
    // tile's value is issued into it (a tile takes far longer than a memory round trip, so the in-order vmcnt
    // behind this tile's stores never stalls).
    constexpr int TAIL_PF = DS ? 1 : 2;  // tiles of input prefetch per wave: two where the inputs are the big identity tensor (-5 %), one in the
                                         // MFMA-heavier downsample variant (two measured 3 % slower there)
    constexpr int NRS = DS ? 2 : 8;      // identity registers: 8 chunks of the identity itself, or 2 B fragments of the block input
    int tile = blockIdx.x * (NT / 64) + wave;
    auto first_loads = [&](int tile, u32x4 (&xf)[2], u32x4 (&rs)[NRS]) {
        const unsigned pix = (unsigned)(tile * 16 + fr);           // past M: the descriptor returns zeros
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) xf[kk] = __builtin_amdgcn_raw_buffer_load_b128(rs_y2, pix * 128u + kk * 64 + fq * 16, 0, 0);
        if constexpr (DS) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) rs[kk] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, pix * 128u + kk * 64 + fq * 16, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) rs[t] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, pix * 512u + t * 64 + fq * 16, 0, 0);
        }
    };
    // TAIL_PF register sets, used round robin (loop unrolled by TAIL_PF so they are indexed statically): a set's loads for
    // tile t + TAIL_PF*nwaves are issued as soon as tile t has consumed it.
    auto do_tile = [&](int tile, u32x4 (&xf)[2], u32x4 (&rs)[NRS]) {
        const unsigned pix = (unsigned)(tile * 16 + fr);
        const unsigned pix_n = (unsigned)((tile + TAIL_PF * nwaves) * 16 + fr);
        const bf16x8 xb0 = __builtin_bit_cast(bf16x8, xf[0]), xb1 = __builtin_bit_cast(bf16x8, xf[1]);
        // The weights in LDS never change, so the compiler would hoist all 64+ fragment reads out of the tile
        // loop and spill them; an opaque zero per iteration keeps the reads where they are used.
        int opaque = 0;
        asm volatile("" : "+v"(opaque));
        const char* lds = smem + opaque;
        // ---- conv3 (256 x 64) x (64 x 16 pixels), 32 couts at a time, + identity, ReLU, bf16: the block output
        //      and at the same time the B fragment of k-block t of the second GEMM
        u32x4 outp[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            f32x4 lo = *reinterpret_cast<const f32x4*>(lds + B3_OFF + (32 * t + 8 * fq) * 4);
            f32x4 hi = *reinterpret_cast<const f32x4*>(lds + B3_OFF + (32 * t + 8 * fq + 4) * 4);
            const char* wrow = lds + w3_frag + (2 * t) * 2048;
            lo = mfma_e<ET>(*reinterpret_cast<const bf16x8*>(wrow + fphys0), xb0, lo);
            hi = mfma_e<ET>(*reinterpret_cast<const bf16x8*>(wrow + 2048 + fphys0), xb0, hi);
            lo = mfma_e<ET>(*reinterpret_cast<const bf16x8*>(wrow + (fphys0 ^ 64)), xb1, lo);
            hi = mfma_e<ET>(*reinterpret_cast<const bf16x8*>(wrow + 2048 + (fphys0 ^ 64)), xb1, hi);
            u32x4 r;
            if constexpr (DS) {
                // identity = bf16(wd . input + bd): rounded exactly as the separate downsample launch stores it
                f32x4 dlo = *reinterpret_cast<const f32x4*>(lds + BD_OFF + (32 * t + 8 * fq) * 4);
                f32x4 dhi = *reinterpret_cast<const f32x4*>(lds + BD_OFF + (32 * t + 8 * fq + 4) * 4);
                const char* drow = lds + WD_OFF + w3_frag + (2 * t) * 2048;
                const bf16x8 pb0 = __builtin_bit_cast(bf16x8, rs[0]), pb1 = __builtin_bit_cast(bf16x8, rs[1]);
                dlo = mfma_e<ET>(*reinterpret_cast<const bf16x8*>(drow + fphys0), pb0, dlo);
                dhi = mfma_e<ET>(*reinterpret_cast<const bf16x8*>(drow + 2048 + fphys0), pb0, dhi);
                dlo = mfma_e<ET>(*reinterpret_cast<const bf16x8*>(drow + (fphys0 ^ 64)), pb1, dlo);
                dhi = mfma_e<ET>(*reinterpret_cast<const bf16x8*>(drow + 2048 + (fphys0 ^ 64)), pb1, dhi);
                r = (u32x4){pack2_e<ET>(dlo[0], dlo[1]), pack2_e<ET>(dlo[2], dlo[3]), pack2_e<ET>(dhi[0], dhi[1]),
                            pack2_e<ET>(dhi[2], dhi[3])};
            } else {
                r = rs[t];
            }
            lo[0] += unpack_lo_e<ET>(r[0]); lo[1] += unpack_hi_e<ET>(r[0]);
            lo[2] += unpack_lo_e<ET>(r[1]); lo[3] += unpack_hi_e<ET>(r[1]);
            hi[0] += unpack_lo_e<ET>(r[2]); hi[1] += unpack_hi_e<ET>(r[2]);
            hi[2] += unpack_lo_e<ET>(r[3]); hi[3] += unpack_hi_e<ET>(r[3]);
            u32x4 o = (u32x4){pack2_e<ET>(lo[0], lo[1]), pack2_e<ET>(lo[2], lo[3]), pack2_e<ET>(hi[0], hi[1]),
                              pack2_e<ET>(hi[2], hi[3])};
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = relu_bf16x2(o[e]);
            outp[t] = o;
            __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, pix * 512u + t * 64 + fq * 16, 0, 0);
            if constexpr (!DS) rs[t] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, pix_n * 512u + t * 64 + fq * 16, 0, 0);
        }
        if constexpr (DS) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) rs[kk] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, pix_n * 128u + kk * 64 + fq * 16, 0, 0);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) xf[kk] = __builtin_amdgcn_raw_buffer_load_b128(rs_y2, pix_n * 128u + kk * 64 + fq * 16, 0, 0);
        // ---- next conv1: (C1 x 256) x (256 x 16 pixels), K blocks straight from outp
        f32x4 acc2[M2];
#pragma unroll
        for (int t2 = 0; t2 < M2 / 2; ++t2) {
            acc2[2 * t2] = *reinterpret_cast<const f32x4*>(lds + B1_OFF + (32 * t2 + 8 * fq) * 4);
            acc2[2 * t2 + 1] = *reinterpret_cast<const f32x4*>(lds + B1_OFF + (32 * t2 + 8 * fq + 4) * 4);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const bf16x8 kb = __builtin_bit_cast(bf16x8, outp[t]);
#pragma unroll
            for (int m2 = 0; m2 < M2; ++m2) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(lds + w1_frag + m2 * 8192 + (((4 * t + fq) ^ fr) << 4));
                acc2[m2] = mfma_e<ET>(wf, kb, acc2[m2]);
            }
        }
#pragma unroll
        for (int t2 = 0; t2 < M2 / 2; ++t2) {
            const f32x4 lo = acc2[2 * t2], hi = acc2[2 * t2 + 1];
            u32x4 o = (u32x4){pack2_e<ET>(lo[0], lo[1]), pack2_e<ET>(lo[2], lo[3]), pack2_e<ET>(hi[0], hi[1]),
                              pack2_e<ET>(hi[2], hi[3])};
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = relu_bf16x2(o[e]);
            __builtin_amdgcn_raw_buffer_store_b128(o, rs_y1, pix * (C1 * 2u) + t2 * 64 + fq * 16, 0, 0);
        }
        };
    u32x4 xfA[2], rsA[NRS], xfB[2], rsB[NRS];
    first_loads(tile, xfA, rsA);
    if (TAIL_PF > 1) first_loads(tile + nwaves, xfB, rsB);
    while (tile < ntiles) {
        do_tile(tile, xfA, rsA); tile += nwaves;
        if (TAIL_PF > 1) {
            if (tile >= ntiles) break;
            do_tile(tile, xfB, rsB); tile += nwaves;
        }
    }
#else
    (void)a;
#endif
}

// ------------------------------------------------------------------------------------------------
// Bottleneck tail (layer2): conv3 1x1 (128 -> 512) + bn3 + identity + ReLU, and the NEXT block's conv1 1x1
// (512 -> 128) + bn1 + ReLU, in one pass over the pixels.  Same idea as bneck_tail_kernel, but 256 KB of
// weights do not fit LDS, so the CHANNELS are split over the 8 waves of a workgroup instead of the pixels:
// wave w owns block-output channels [64w, 64w+64) of the workgroup's 16 pixels.  Its slice of both weight
// matrices -- conv3 rows 64w.. (16 A fragments) and next-conv1 columns 64w.. (16 A fragments) -- lives in
// REGISTERS for the whole launch; activations go global -> B fragments directly.  The second GEMM sums over
// all 512 channels, i.e. over the waves: every wave writes its fp32 partial (128 couts x 16 pixels) to LDS,
// one barrier, then wave w adds the eight partials of couts [16w, 16w+16) in a fixed order (deterministic),
// adds the bias, ReLU, bf16, stores.  Partial buffers alternate so one barrier per step suffices.
// The fp32 summation order of the second conv differs from the igemm's (8 partial sums instead of one
// chain), so this path is compared with the oracle under the usual bf16 tolerance, not bit-for-bit with the
// two launches it replaces; conv3's output IS bit-identical.
// ------------------------------------------------------------------------------------------------
struct Tail2Args {
    const __bf16* y2;     // (M, 128)  conv2 output
    const __bf16* w3;     // (512, 128)
    const float* b3;      // (512)
    const __bf16* res;    // (M, 512)  identity
    __bf16* out;          // (M, 512)  block output
    const __bf16* w1;     // (128, 512) next block's conv1
    const float* b1;      // (128)
    __bf16* y1n;          // (M, 128)
    int M;
};

#define TAIL2_PF 3      // steps of input prefetch = depth of the unrolled register ring in the step loop
template <int ET>
__global__ __launch_bounds__(512) void bneck_tail2_kernel(const Tail2Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int PART_BYTES = 8 * 8 * 1024;                 // [wave][m2] x 1 KiB (f32x4 per lane)
    constexpr int B3_OFF = 2 * PART_BYTES;
    constexpr int Y_OFF = B3_OFF + 512 * 4;                  // two 4-KiB images of conv2's output (16 pixels x 256 B)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int cbase = wave * 64;

    for (int i = tid; i < 512; i += 512) reinterpret_cast<float*>(smem + B3_OFF)[i] = a.b3[i];

    // conv3 rows of this wave, permuted so a lane ends up with 8 consecutive channels (see igemm_bf16_kernel)
    bf16x8 a3[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int rho = 16 * m + fr;
        const int cl = (rho & ~31) | (rho & 3) | (((rho >> 4) & 1) << 2) | (((rho >> 2) & 3) << 3);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            a3[m][kk] = *reinterpret_cast<const bf16x8*>(a.w3 + (size_t)(cbase + cl) * 128 + kk * 32 + fq * 8);
    }
    // next conv1: all 128 rows, this wave's 64 K columns
    bf16x8 a1[8][2];
#pragma unroll
    for (int m2 = 0; m2 < 8; ++m2)
#pragma unroll
        for (int t = 0; t < 2; ++t)
            a1[m2][t] = *reinterpret_cast<const bf16x8*>(a.w1 + (size_t)(16 * m2 + fr) * 512 + cbase + 32 * t + fq * 8);
    const f32x4 bias1 = *reinterpret_cast<const f32x4*>(a.b1 + 16 * wave + 4 * fq);
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rs_y2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.y2), 0, (unsigned)a.M * 256u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.res), 0, (unsigned)a.M * 1024u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)a.M * 1024u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y1 = __builtin_amdgcn_make_buffer_rsrc(a.y1n, 0, (unsigned)a.M * 256u, 0x00020000);

    const int nsteps = (a.M + 15) >> 4;
    const int gstride = gridDim.x;
    // conv2's output of a step (16 pixels x 256 B) is needed by all 8 waves: waves 0..3 fetch a quarter each with
    // ONE full-row instruction (4 pixels x 256 B) and park it in LDS, every wave reads its B fragments from there.
    // (Eight waves each loading the fragments themselves is 32 more address-unit instructions of 16 half-lines per
    // step -- the address unit, not HBM, then sets the pace.)  Row image: 16-B chunk c of pixel p at p*256 + ((c ^ p) << 4).
    const int ypix = 4 * wave + (lane >> 4), ychunk = lane & 15;        // waves 0..3 only
    const int y_wr = ypix * 256 + ((ychunk ^ ypix) << 4);
    auto load_y = [&](int st) -> u32x4 {
        return __builtin_amdgcn_raw_buffer_load_b128(rs_y2, (unsigned)(st * 16 + ypix) * 256u + ychunk * 16, 0, 0);
    };
    auto load_r = [&](int st, u32x4 (&r)[2]) {
        const unsigned pix = (unsigned)(st * 16 + fr);              // past M: the descriptor returns zeros
#pragma unroll
        for (int t = 0; t < 2; ++t) r[t] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, pix * 1024u + (cbase + 32 * t + 8 * fq) * 2, 0, 0);
    };
    // One step = 16 pixels.  `rs` holds this step's identity slice and `yn` (waves 0..3) the NEXT step's quarter of
    // conv2's output; once consumed, the loads for TAIL2_PF steps later go into the same registers (no register
    // moves: moving a load's destination would wait for the load).
    auto do_step = [&](int step, int par, u32x4 (&rs)[2], u32x4& yn) {
        const unsigned pix = (unsigned)(step * 16 + fr);
        const char* ycur = smem + Y_OFF + par * 4096;
        // ---- conv3 slice: 64 couts x 16 pixels, K = 128
        u32x4 outp[2];
        bf16x8 xb[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xb[kk] = *reinterpret_cast<const bf16x8*>(ycur + fr * 256 + (((kk * 4 + fq) ^ fr) << 4));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 lo = *reinterpret_cast<const f32x4*>(smem + B3_OFF + (cbase + 32 * t + 8 * fq) * 4);
            f32x4 hi = *reinterpret_cast<const f32x4*>(smem + B3_OFF + (cbase + 32 * t + 8 * fq + 4) * 4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                lo = mfma_e<ET>(a3[2 * t][kk], xb[kk], lo);
                hi = mfma_e<ET>(a3[2 * t + 1][kk], xb[kk], hi);
            }
            const u32x4 r = rs[t];
            lo[0] += unpack_lo_e<ET>(r[0]); lo[1] += unpack_hi_e<ET>(r[0]);
            lo[2] += unpack_lo_e<ET>(r[1]); lo[3] += unpack_hi_e<ET>(r[1]);
            hi[0] += unpack_lo_e<ET>(r[2]); hi[1] += unpack_hi_e<ET>(r[2]);
            hi[2] += unpack_lo_e<ET>(r[3]); hi[3] += unpack_hi_e<ET>(r[3]);
            u32x4 o = (u32x4){pack2_e<ET>(lo[0], lo[1]), pack2_e<ET>(lo[2], lo[3]), pack2_e<ET>(hi[0], hi[1]),
                              pack2_e<ET>(hi[2], hi[3])};
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = relu_bf16x2(o[e]);
            outp[t] = o;
            __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, pix * 1024u + (cbase + 32 * t + 8 * fq) * 2, 0, 0);
        }
        load_r(step + TAIL2_PF * gstride, rs);
        if (wave < 4) {
            *reinterpret_cast<u32x4*>(smem + Y_OFF + (par ^ 1) * 4096 + y_wr) = yn;       // next step's conv2 output
            yn = load_y(step + (TAIL2_PF + 1) * gstride);
        }
        // ---- next conv1, this wave's 64 of the 512 K channels: partial (128 couts x 16 pixels) to LDS
        char* part = smem + par * PART_BYTES;
#pragma unroll
        for (int m2 = 0; m2 < 8; ++m2) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc = mfma_e<ET>(a1[m2][t], __builtin_bit_cast(bf16x8, outp[t]), acc);
            *reinterpret_cast<f32x4*>(part + (wave * 8 + m2) * 1024 + lane * 16) = acc;
        }
        // NOT __syncthreads(): that also drains vmcnt, i.e. waits for every prefetch and store in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- wave w reduces couts [16w, 16w+16): fixed order over the eight channel slices
        f32x4 sum = *reinterpret_cast<const f32x4*>(part + (0 * 8 + wave) * 1024 + lane * 16);
#pragma unroll
        for (int w2 = 1; w2 < 8; ++w2) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(part + (w2 * 8 + wave) * 1024 + lane * 16);
            sum[0] += p[0]; sum[1] += p[1]; sum[2] += p[2]; sum[3] += p[3];
        }
        sum[0] += bias1[0]; sum[1] += bias1[1]; sum[2] += bias1[2]; sum[3] += bias1[3];
        const unsigned o0 = relu_bf16x2(pack2_e<ET>(sum[0], sum[1])), o1 = relu_bf16x2(pack2_e<ET>(sum[2], sum[3]));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64((u32x2){o0, o1}, rs_y1, pix * 256u + (16 * wave + 4 * fq) * 2, 0, 0);
    };
    u32x4 rsA[2], rsB[2], rsC[2], yA = {0, 0, 0, 0}, yB = {0, 0, 0, 0}, yC = {0, 0, 0, 0};
    int step = blockIdx.x;
    if (wave < 4) {
        const u32x4 y0 = load_y(step);
        yA = load_y(step + gstride); yB = load_y(step + 2 * gstride); yC = load_y(step + 3 * gstride);
        *reinterpret_cast<u32x4*>(smem + Y_OFF + y_wr) = y0;
    }
    load_r(step, rsA); load_r(step + gstride, rsB); load_r(step + 2 * gstride, rsC);
    __syncthreads();
    // every wave of the workgroup runs the same steps (there is a barrier inside do_step); the LDS buffers alternate
    int par = 0;
    while (step < nsteps) {
        do_step(step, par, rsA, yA); step += gstride; par ^= 1;
        if (step >= nsteps) break;
        do_step(step, par, rsB, yB); step += gstride; par ^= 1;
        if (step >= nsteps) break;
        do_step(step, par, rsC, yC); step += gstride; par ^= 1;
    }
#else
    (void)a;
#endif
}

// ------------------------------------------------------------------------------------------------
// Bottleneck tail (layer3): conv3 1x1 (256 -> 1024) + bn3 + identity + ReLU and the NEXT block's conv1 1x1 (1024 -> 256) + bn1 + ReLU
// CHAINED inside one launch (src/preprocess_resnet_features.py:296 -> torchvision Bottleneck.forward: `out = relu(bn3(conv3(out)) +
// identity)` of block b, `relu(bn1(conv1(x)))` of block b+1).
// Why: at 14x14 both convs are HBM-phase / MFMA-phase alternators (conv3: 59 us at 3.9 TB/s, conv1: 37 us at 3.4 TB/s, batch 256).
// Chained, the 103 MB block output is written once and never read back by conv1, and one launch boundary disappears.
// The chain runs over 128-channel CHUNKS c of the block output, for a tile of P pixels:
//     A(c):  accA[128 x P] = b3[c] + W3[c] . t2          4 K-slots of 64; the tile's t2 rows are RESIDENT in LDS
//     E(c):  accA += res[c]; relu; 16-bit -> LDS `out_c` (P rows x 256 B): the block output AND the B operand of the second GEMM
//     B(c):  accB[256 x P] += W1[:, c] . out_c           2 K-slots x 2 cout halves; accB stays in registers across the 8 chunks
// and, after chunk 7, y1n = relu(accB) (accB started at b1).  Summation orders are those of the igemm launches this replaces (bias
// first, K ascending, residual last, the residual added with the same fp32 additions), so both outputs are bit-identical to them.
// Rounds 2-3 ran this as bneck_tail3_kernel (4 consumer waves with the weights straight from L2 in fragment order + 4 helper waves for the
// LDS-DMA of t2 / residual rows and the copy-out; 17 barriers per tile; 82-85 us per launch); round 3's two-group pipeline below
// (bneck_tail3p_kernel, 71 us) replaced it and the old kernel was removed in round 4 (git history, profiles/r03_tail3p_variants.txt).
// P = 112 rows in LDS of which the first `bp` are real.
// ------------------------------------------------------------------------------------------------
struct Tail3Args {
    const __bf16* y2;     // (M, 256)   conv2 output
    const __bf16* wp;     // packed weights of both convs: [chunk 8][step 8: W3 K-slot 0..3, then W1 (K-slot kb, cout half h) = 2 kb + h][wave 4]
                          // [fragment 4 = 2 m + kk][lane 64] x 16 B  (tail3_pack_kernel), 1 MB
    const float* b3;      // (1024)
    const __bf16* res;    // (M, 1024)  identity
    __bf16* out;          // (M, 1024)  block output
    const float* b1;      // (256)
    __bf16* y1n;          // (M, 256)   next block's conv1 output
    int M;
    int bp;               // real pixels per tile (<= 112)
    int n_tiles;          // ceil(M / bp)
#if defined(R50_STAMP)    // diagnostic build (scripts/stamp_tail3.py): per-wave cycle sums, 8 slots per wave
    unsigned long long* dbg;
#endif
};

// (1024,256) conv3 and (256,1024) next-conv1 weights, K contiguous -> the fragment-ordered stream bneck_tail3p_kernel reads.
// Element (c, s, w, f = 2m + kk, lane = 16 fq + fr) is the A fragment of row rho = 32w + 16m + fr of the step's 128-row slice, whose
// channel is perm(rho) (8 consecutive couts per lane, see igemm_bf16_kernel), K = 32 kk + 8 fq .. + 7 of the step's 64.
__global__ void tail3_pack_kernel(const __bf16* __restrict__ w3, const __bf16* __restrict__ w1, __bf16* __restrict__ wp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // one 16-B element each; 8 * 8 * 4 * 4 * 64 = 65,536
    if (idx >= 65536) return;
    const int lane = idx & 63, f = (idx >> 6) & 3, w = (idx >> 8) & 3, st = (idx >> 10) & 7, c = idx >> 13;
    const int fr = lane & 15, fq = lane >> 4, m = f >> 1, kk = f & 1;
    const int rho = 32 * w + 16 * m + fr;
    const int cl = (rho & ~31) | (rho & 3) | (((rho >> 4) & 1) << 2) | (((rho >> 2) & 3) << 3);
    const __bf16* src;
    if (st < 4) src = w3 + (size_t)(c * 128 + cl) * 256 + st * 64 + kk * 32 + fq * 8;
    else {
        const int bs = st - 4, kb = bs >> 1, hh = bs & 1;
        src = w1 + (size_t)(hh * 128 + cl) * 1024 + c * 128 + kb * 64 + kk * 32 + fq * 8;
    }
    reinterpret_cast<u32x4*>(wp)[idx] = *reinterpret_cast<const u32x4*>(src);
}

// ------------------------------------------------------------------------------------------------
// bneck_tail3p_kernel (round 3): the chained layer3 tail of bneck_tail3_kernel as a TWO-GROUP PIPELINE.
// What the stamps of bneck_tail3_kernel said (profiles/r02_stamps_bneck_tail3.txt): its four consumer waves -- one per SIMD -- spend
// 48 % of their cycles issuing MFMAs; the rest is the serial A -> R -> E -> O -> B chain of one wave (LDS round trips at every step's
// head, the epilogue's vector work, two barriers per chunk), and nothing else on the SIMD has MFMAs to fill those holes with.
// Here the two GEMMs of a chunk run on DIFFERENT waves, one chunk apart, so every SIMD always holds two MFMA-issuing waves in
// different phases:
//   * group A, waves 0-3 (wave w: couts 32w.. of the chunk's 128, all 7 pixel blocks): accA = b3 + W3[c] . t2 (4 weight steps, t2
//     resident in LDS), then E(c): + identity, ReLU, 16 bit -> LDS out_c[c & 1].  The identity does not go through LDS at all: lane
//     (fr, fq) needs exactly 16 B of it per pixel block (its own 8 channels of its own pixel), so it is loaded straight into
//     registers, one chunk ahead (the load of chunk c + 1, block j, is issued the moment block j of chunk c has been consumed).
//     Group A also DMAs the next tile's t2 rows.
//   * group B, waves 4-7 (wave w: couts 32w.. of each 128-row half, all 7 pixel blocks), one chunk BEHIND: accB += W1[:, c] . out_c
//     (4 weight steps), then the copy-out of out_c to the block output with full-row stores; y1n = relu(accB) after chunk 7.
//   * ONE barrier per chunk: out_c[c & 1] complete / out_c[(c - 1) & 1] consumed.  A tile takes nine intervals (A idles in the last one,
//     where the next tile's t2 rows land; B idles in the first one, where it stores the previous tile's y1n).
// Weights: the fragment-ordered stream of tail3_pack_kernel, fetched two steps ahead into registers by the wave that uses them
// (steps 0-3 of a chunk by group A, 4-7 by group B).  Summation orders are unchanged: both outputs keep the bits of bneck_tail3_kernel
// and of the two igemm launches.
// LDS: T2 [4 slots][ROWS] + out_c 2 x [2 slots][ROWS] rows of 128 B + b1 + b3 = 119,808 B at ROWS = 112.
// ------------------------------------------------------------------------------------------------
constexpr int T3P_XPRE = 0;   // cross-step prefetch of the first pixel fragments of the next step (1 = group A, 2 = group B): measured neutral, off
// The block output is written by group B, which copies out_c out of LDS in full rows (stamps: 102 k cycles per wave and launch against 128 k
// with 64-B pieces straight from group A's E-phase registers).
constexpr int T3P_PDA = 6;    // LDS prefetch depth (pixel fragments) of the A / B weight steps
constexpr int T3P_PDB = 6;
// NOB (layer3.5, whose next conv1 is layer4.0's 1024 -> 512 and does not fit the chain): group B only copies out_c out -- conv3 + identity + ReLU of
// the stage's last block through the same pipeline, no second GEMM, no y1n (the weight stream's W1 slices are never read).
template <int ET, int ROWS, bool NOB = false>
__global__ __launch_bounds__(512) void bneck_tail3p_kernel(const Tail3Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int CMID = 256, COUT = 1024, C1N = 256;
    constexpr int NCH = COUT / 128;               // chunks of 128 block-output channels
    constexpr int NR = 7;                         // MFMA column blocks (pixel rows 16 j + fr; rows >= ROWS are never stored)
    constexpr int SLOT = ROWS * 128;              // one K-slot (64 channels) of ROWS pixel rows
    constexpr int T2 = 0, OUTC = 4 * SLOT, B1_OFF = OUTC + 4 * SLOT, B3_OFF = B1_OFF + C1N * 4;
    constexpr int T2_PASSES = (4 * ROWS + 31) / 32, OC_PASSES = (2 * ROWS + 31) / 32;      // 32-row passes of 256 lanes x 16 B
    static_assert(ROWS == 112 || ROWS == 98, "row counts with a whole number of 8-row DMA pieces per region");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wave & 3;
    const int lt = tid & 255, srow = lt >> 3;
    const int grid = gridDim.x;
    const int first = blockIdx.x;
    const int fr = lane & 15, fq = lane >> 4;

    if (tid < C1N) {                              // both bias vectors live in LDS
        reinterpret_cast<float*>(smem + B1_OFF)[tid] = a.b1[tid];
        reinterpret_cast<f32x4*>(smem + B3_OFF)[tid] = reinterpret_cast<const f32x4*>(a.b3)[tid];
    }

    const __amdgpu_buffer_rsrc_t rs_wp = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.wp), 0, 1048576u, 0x00020000);
    const unsigned wp_voff = (unsigned)(lane * 16 + w * 4096);
    const int fphys0 = (fq ^ (fr & 7)) << 4;                       // kk = 0; kk = 1 is ^ 64
    const int x_frag = fr * 128;                                    // + j * 2048
    // weight fragments of step g of the group's own sequence (4 steps per chunk; GOFS = 0: W3 steps, 4: W1 steps), K half kk
    auto w_load_half = [&](int g, int gofs, bf16x8 (&wf)[4], int kk) {
        const int sofs = __builtin_amdgcn_readfirstlane((((((g >> 2) & 7) << 3) + (g & 3) + gofs)) << 14);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            wf[2 * m + kk] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_wp, wp_voff + (2 * m + kk) * 1024, sofs, 0));
        }
    };
    // one weight step: acc[m][j] += W[32w + 16m ..][64 K] . X[64 K][16 j ..]; 14 slots t = 7 kk + j of one pixel fragment and two MFMAs, the
    // fragment of slot t + PD read when slot t issues; the K half a step has finished with is re-requested for step g + 2 at once.
    // Steps are separate scheduling regions (see bneck_tail3_kernel's w_step), so a step would open with an exposed LDS round trip; where the
    // NEXT step's operand is already complete in LDS (HAVE / NXT: inside a chunk), its first PD fragments are requested in this step's last
    // slots, which have no reads of their own left, and handed over in `xq`.
    auto w_step = [&](auto pd, auto wd_c, auto have_c, auto nxt_c, int g, int gofs, bf16x8 (&wf)[4], const char* xb, const char* xnb, f32x4 (&acc0)[NR],
                      f32x4 (&acc1)[NR], bf16x8 (&xq)[NR]) {
        constexpr int NS = 2 * NR, PD = decltype(pd)::value, WD = decltype(wd_c)::value;      // WD: the buffer is re-requested for step g + WD
        constexpr bool HAVE = decltype(have_c)::value, NXT = decltype(nxt_c)::value;
        bf16x8 x[NS];
        auto xread = [&](const char* b, int t) {
            return *reinterpret_cast<const bf16x8*>(b + x_frag + (t % NR) * 2048 + (t >= NR ? (fphys0 ^ 64) : fphys0));
        };
#pragma unroll
        for (int t = 0; t < PD; ++t) x[t] = HAVE ? xq[t] : xread(xb, t);
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int j = t % NR, kk = t / NR;
            {
                acc0[j] = mfma_e<ET>(wf[kk], x[t], acc0[j]);
                acc1[j] = mfma_e<ET>(wf[2 + kk], x[t], acc1[j]);
            }
            if (t + PD < NS) x[t + PD] = xread(xb, t + PD);
            else if (NXT) xq[t + PD - NS] = xread(xnb, t + PD - NS);
            if (t == NR - 1) w_load_half(g + WD, gofs, wf, 0);
        }
        w_load_half(g + WD, gofs, wf, 1);
        if (!HAVE) __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if (t + PD < NS || NXT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (t == NR - 1) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        __builtin_amdgcn_sched_barrier(0);        // a scheduling region per step
    };
    bf16x8 wA[4], wB[4], xq[NR];
    using NO = std::false_type;

    if (wave < 4) {
        // =============================== group A: conv3 + identity + ReLU -> out_c ===================
        const __amdgpu_buffer_rsrc_t rs_y2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.y2), 0, (unsigned)a.M * (CMID * 2u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.res), 0, (unsigned)a.M * (COUT * 2u), 0x00020000);
        // the tile's t2 rows: row R = 32 i + srow of the [4 slots][ROWS] region -> K-slot R / ROWS, pixel row R % ROWS, logical 16-B chunk
        // (lane & 7) ^ (pixel row & 7); rows past the tile's real pixels zero-fill (out-of-range offset)
        auto issue_t2 = [&](int tile) {
            const int p0 = tile * a.bp;
            const int limit = (tile < a.n_tiles) ? ((a.M - p0 < a.bp) ? a.M - p0 : a.bp) : 0;
#pragma unroll
            for (int i = 0; i < T2_PASSES; ++i) {
                const int R = 32 * i + srow;
                const int sl = R / ROWS, prow = R - ROWS * sl;
                const int lchunk = (lt & 7) ^ (prow & 7);
                const unsigned voff = (prow < limit) ? (unsigned)((p0 + prow) * CMID + sl * 64 + lchunk * 8) * 2u : kOobOffset;
                if (32 * i + 32 <= 4 * ROWS || w == 0)      // (ROWS = 98: the last pass is the 8 rows of wave 0)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y2, (LDS_AS void*)(smem + T2 + i * 4096 + w * 1024), 16, voff, 0, 0, 0);
            }
        };
        // the lane's identity values / block-output values: pixel row 16 j + fr, channels (chunk) + 32 w + 8 fq .. + 7 (same offsets in both tensors)
        unsigned rv[NR], rvn[NR];                 // this tile's and the next tile's
        auto tile_rows = [&](int tile, unsigned (&v)[NR]) {
            const int p0 = tile * a.bp;
            const int limit = (tile < a.n_tiles) ? ((a.M - p0 < a.bp) ? a.M - p0 : a.bp) : 0;
#pragma unroll
            for (int j = 0; j < NR; ++j)
                v[j] = (16 * j + fr < limit) ? (unsigned)((p0 + 16 * j + fr) * COUT + 32 * w + 8 * fq) * 2u : kOobOffset;
        };
        const int c_frag = (w >> 1) * SLOT + x_frag + (((4 * (w & 1) + fq) ^ (fr & 7)) << 4);
        f32x4 accA[2][NR];
        // Group A's VM queue carries the identity loads and the block-output stores -- HBM latencies, and vmcnt counts in issue order.  So
        //   * the identity is requested TWO chunks ahead (two register sets, even / odd chunks; 28 KB in flight per CU per chunk were what
        //     paced the first version: with no MFMAs at all it took the same 80 us), inside E before that E's stores;
        //   * the weight fragments are requested FOUR steps (one chunk) ahead instead of two: a wait for them then covers nothing younger
        //     than the previous chunk's E phase.
        u32x4 r0[NR], r1[NR];
        bf16x8 wq[4][4];
        tile_rows(first, rv);
        tile_rows(first + grid, rvn);
        issue_t2(first);
#pragma unroll
        for (int j = 0; j < NR; ++j) r0[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, rv[j], 0, 0);
#pragma unroll
        for (int j = 0; j < NR; ++j) r1[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, rv[j], 256, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { w_load_half(q, 0, wq[q], 0); w_load_half(q, 0, wq[q], 1); }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // own t2 pieces landed, own bias words written
        __builtin_amdgcn_s_barrier();             // P
        int g = 0;
        R50_STAMP_DECL
        // one chunk of group A: c = 2 c2 + PAR; `r` is the register set of its parity
        auto chunk_a = [&](auto par_c, int c2, u32x4 (&r)[NR]) {
            constexpr int PAR = decltype(par_c)::value;
            const int c = 2 * c2 + PAR;
            {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + B3_OFF + (c * 128 + 32 * w + 8 * fq) * 4);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + B3_OFF + (c * 128 + 32 * w + 8 * fq + 4) * 4);
#pragma unroll
                for (int j = 0; j < NR; ++j) { accA[0][j] = lo; accA[1][j] = hi; }
            }
            using PA = std::integral_constant<int, T3P_PDA>;
            using W4 = std::integral_constant<int, 4>;
            using XP = std::integral_constant<bool, (T3P_XPRE & 1) != 0>;
            w_step(PA{}, W4{}, NO{}, XP{}, g + 0, 0, wq[0], smem + T2 + 0 * SLOT, smem + T2 + 1 * SLOT, accA[0], accA[1], xq);
            w_step(PA{}, W4{}, XP{}, XP{}, g + 1, 0, wq[1], smem + T2 + 1 * SLOT, smem + T2 + 2 * SLOT, accA[0], accA[1], xq);
            w_step(PA{}, W4{}, XP{}, XP{}, g + 2, 0, wq[2], smem + T2 + 2 * SLOT, smem + T2 + 3 * SLOT, accA[0], accA[1], xq);
            w_step(PA{}, W4{}, XP{}, NO{}, g + 3, 0, wq[3], smem + T2 + 3 * SLOT, smem + T2, accA[0], accA[1], xq);
            g += 4;
#if defined(R50_STAMP)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 0" ::"v"(accA[1][NR - 1]), "v"(accA[0][NR - 1]) : "memory");   // the stamp waits for the last MFMAs
#endif
            R50_MARK(0)                           // A: 4 weight steps
            __builtin_amdgcn_sched_barrier(0);
            // ---- E: + identity, ReLU, 16 bit -> out_c[c & 1]; the identity of chunk c + 2 (of the next
            // tile after chunks 6 and 7) is requested into the registers this chunk has just consumed
            const bool nxt = (c2 == NCH / 2 - 1);
            const int cofs_n = __builtin_amdgcn_readfirstlane(((c + 2) & (NCH - 1)) * 256);
            char* ob = smem + OUTC + PAR * 2 * SLOT + c_frag;
            u32x4 o[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                f32x4 lo = accA[0][j], hi = accA[1][j];
                lo[0] += unpack_lo_e<ET>(r[j][0]); lo[1] += unpack_hi_e<ET>(r[j][0]);
                lo[2] += unpack_lo_e<ET>(r[j][1]); lo[3] += unpack_hi_e<ET>(r[j][1]);
                hi[0] += unpack_lo_e<ET>(r[j][2]); hi[1] += unpack_hi_e<ET>(r[j][2]);
                hi[2] += unpack_lo_e<ET>(r[j][3]); hi[3] += unpack_hi_e<ET>(r[j][3]);
                o[j] = (u32x4){pack2_e<ET>(lo[0], lo[1]), pack2_e<ET>(lo[2], lo[3]), pack2_e<ET>(hi[0], hi[1]), pack2_e<ET>(hi[2], hi[3])};
#pragma unroll
                for (int e = 0; e < 4; ++e) o[j][e] = relu_bf16x2(o[j][e]);
                if (ROWS >= 16 * j + 16 || fr < ROWS - 16 * j) *reinterpret_cast<u32x4*>(ob + j * 2048) = o[j];
                r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, nxt ? rvn[j] : rv[j], cofs_n, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // out_c written before anybody reads it
            R50_MARK(1)                           // E
            __builtin_amdgcn_s_barrier();
            R50_MARK(2)                           // chunk barrier
        };
        for (int tile = first; tile < a.n_tiles; tile += grid) {
            for (int c2 = 0; c2 < NCH / 2; ++c2) {
                chunk_a(std::integral_constant<int, 0>{}, c2, r0);
                chunk_a(std::integral_constant<int, 1>{}, c2, r1);
            }
            // ---- ninth interval: group B is on chunk 7; the T2 region is free -> the next tile's t2 rows
            issue_t2(tile + grid);
#pragma unroll
            for (int j = 0; j < NR; ++j) rv[j] = rvn[j];
            tile_rows(tile + 2 * grid, rvn);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            R50_MARK(3)                           // t2 issue + landing
            __builtin_amdgcn_s_barrier();
            R50_MARK(4)                           // ninth barrier
        }
        R50_STAMP_FLUSH(8)
    } else {
        // =============================== group B: next conv1 + copy-out ==============================
        const __amdgpu_buffer_rsrc_t rs_y1 = __builtin_amdgcn_make_buffer_rsrc(a.y1n, 0, (unsigned)a.M * (C1N * 2u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)a.M * (COUT * 2u), 0x00020000);
        f32x4 accB[4][NR];
        unsigned rvo[OC_PASSES];                   // block-output offsets of the lane's copy-out rows (row R = 32 i + srow of [2 slots][ROWS])
        auto tile_rows = [&](int tile) {
            const int p0 = tile * a.bp;
            const int limit = (a.M - p0 < a.bp) ? a.M - p0 : a.bp;
#pragma unroll
            for (int i = 0; i < OC_PASSES; ++i) {
                const int R = 32 * i + srow;
                const int sl = R / ROWS, prow = R - ROWS * sl;
                const int lchunk = (lt & 7) ^ (prow & 7);
                rvo[i] = (sl < 2 && prow < limit) ? (unsigned)((p0 + prow) * COUT + sl * 64 + lchunk * 8) * 2u : kOobOffset;
            }
        };
        auto y1n_store = [&](int tile) {
            const int p0 = tile * a.bp;
            const int limit = (a.M - p0 < a.bp) ? a.M - p0 : a.bp;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const f32x4 lo = accB[2 * t][j], hi = accB[2 * t + 1][j];
                    u32x4 o = (u32x4){pack2_e<ET>(lo[0], lo[1]), pack2_e<ET>(lo[2], lo[3]), pack2_e<ET>(hi[0], hi[1]), pack2_e<ET>(hi[2], hi[3])};
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = relu_bf16x2(o[e]);
                    const unsigned voff = (16 * j + fr < limit) ? (unsigned)((p0 + 16 * j + fr) * C1N + 128 * t + 32 * w + 8 * fq) * 2u : kOobOffset;
                    __builtin_amdgcn_raw_buffer_store_b128(o, rs_y1, voff, 0, 0);
                }
        };
        if constexpr (!NOB) {
            w_load_half(0, 4, wA, 0); w_load_half(0, 4, wA, 1);
            w_load_half(1, 4, wB, 0); w_load_half(1, 4, wB, 1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();             // P
        int g = 0;
        int prev = -1;
        R50_STAMP_DECL
        for (int tile = first; tile < a.n_tiles; tile += grid) {
            // ---- first interval (group A is on chunk 0): the previous tile's y1n leaves, the accumulators restart at b1
            if (!NOB && prev >= 0) y1n_store(prev);
            prev = tile;
            tile_rows(tile);
#pragma unroll
            for (int t = 0; t < (NOB ? 0 : 2); ++t) {         // accB[2t + e]: channels 128t + 32w + 8fq + 4e ..
                const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + B1_OFF + (128 * t + 32 * w + 8 * fq) * 4);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + B1_OFF + (128 * t + 32 * w + 8 * fq + 4) * 4);
#pragma unroll
                for (int j = 0; j < NR; ++j) { accB[2 * t][j] = lo; accB[2 * t + 1][j] = hi; }
            }
            R50_MARK(0)                           // first interval: y1n of the previous tile, accumulator init
            __builtin_amdgcn_s_barrier();
            R50_MARK(1)                           // first barrier
            for (int c = 0; c < NCH; ++c) {
                const char* xb = smem + OUTC + (c & 1) * 2 * SLOT;
                using PB = std::integral_constant<int, T3P_PDB>;
                using W2 = std::integral_constant<int, 2>;
                using XP = std::integral_constant<bool, (T3P_XPRE & 2) != 0>;
                if constexpr (!NOB) {
                    w_step(PB{}, W2{}, NO{}, XP{}, g + 0, 4, wA, xb, xb, accB[0], accB[1], xq);
                    w_step(PB{}, W2{}, XP{}, XP{}, g + 1, 4, wB, xb, xb + SLOT, accB[2], accB[3], xq);
                    w_step(PB{}, W2{}, XP{}, XP{}, g + 2, 4, wA, xb + SLOT, xb + SLOT, accB[0], accB[1], xq);
                    w_step(PB{}, W2{}, XP{}, NO{}, g + 3, 4, wB, xb + SLOT, xb, accB[2], accB[3], xq);
                }
                g += 4;
#if defined(R50_STAMP)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 0" ::"v"(accB[3][NR - 1]), "v"(accB[2][NR - 1]), "v"(accB[1][NR - 1]), "v"(accB[0][NR - 1]) : "memory");
#endif
                R50_MARK(2)                       // B: 4 weight steps
                __builtin_amdgcn_sched_barrier(0);
                // ---- copy-out of out_c[c & 1]: full 128-B row pieces, 16 B per lane
                {
                    const int cofs = __builtin_amdgcn_readfirstlane(c * 256);
                    u32x4 v[OC_PASSES];
#pragma unroll
                    for (int i = 0; i < OC_PASSES; ++i) v[i] = *reinterpret_cast<const u32x4*>(xb + i * 4096 + lt * 16);
#pragma unroll
                    for (int i = 0; i < OC_PASSES; ++i)
                        __builtin_amdgcn_raw_buffer_store_b128(v[i], rs_out, rvo[i], cofs, 0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's reads of out_c[c & 1] are complete
                R50_MARK(3)                       // copy-out
                __builtin_amdgcn_s_barrier();
                R50_MARK(4)                       // chunk barrier
            }
        }
        if (!NOB && prev >= 0) y1n_store(prev);
        R50_STAMP_FLUSH(8)
    }
#else
    (void)a;
#endif
}

// ------------------------------------------------------------------------------------------------
// bneck_catchain_kernel (round 3): the TRANSITION tail of layer2.0 -- conv3 + downsample + add + ReLU as one 1x1 conv over
// K = [t2 (128 channels) | block input sampled at stride 2 (256 channels)] against [W3 | Wd] (the two-source GEMM of igemm_ws_kernel),
// CHAINED with the next block's conv1 (512 -> 128) + ReLU, in the two-group pipeline of bneck_tail3p_kernel:
//   * the tile's K operand -- 112 pixels x 384 channels = 6 K-slots of 14 KB -- is RESIDENT in LDS (t2 rows in slots 0-1, the strided
//     rows of the block input in slots 2-5; 256 -> 512 has no identity tensor, so group A's queue carries only its weight stream);
//   * group A (waves 0-3): per 128-channel chunk c of the block output six weight steps (K-slots 0..5 of Wcat[c]) from bias b3 + bd, then
//     E(c): ReLU, 16 bit -> out_c[c & 1];  group B (waves 4-7), one chunk behind: two weight steps of W1[:, c] against out_c into the
//     next conv1's accumulators (32 couts x 112 pixels per wave), then the copy-out of out_c with full-row stores; y1n after chunk 3;
//   * one barrier per chunk, a fifth interval per tile in which the next tile's operand rows land.
// Summation orders are those of the two igemm launches (bias first, K ascending over [t2 | x], ReLU; b1 first, K ascending): same bits.
// At batch 256: 1792 tiles = exactly 7 per CU.  LDS: 6 + 4 slots of 112 rows x 128 B + b1 + b3 = 145,920 B.
// Weights: catchain_pack_kernel's stream [chunk 4][step 8: Wcat K-slot 0..5, W1 K-slot 0..1][wave 4][fragment 4][lane 64] x 16 B = 512 KB.
// ------------------------------------------------------------------------------------------------
struct CatChainArgs {
    const __bf16* t2;     // (M, 128)        conv2 output
    const __bf16* x;      // (N, 2 OW, 2 OW, 256)  block input
    const __bf16* wp;     // packed weights (catchain_pack_kernel)
    const float* b3;      // (512)   b3 + bd
    __bf16* out;          // (M, 512)        block output
    const float* b1;      // (128)
    __bf16* y1n;          // (M, 128)        next block's conv1 output
    int M;                // N * OW * OW
    int bp;               // real pixels per tile (<= 112)
    int n_tiles;
    unsigned x_bytes;     // N * 4 OW OW * 512
    int x_sub;            // 1: x is already the compact (N, OW, OW, 256) tensor of the block input's even rows and columns (x_bytes = N OW OW 512)
#if defined(R50_STAMP)
    unsigned long long* dbg;
#endif
};

// (512, 384) [W3 | Wd] and (128, 512) next-conv1 weights, K contiguous -> the fragment-ordered stream (see tail3_pack_kernel)
__global__ void catchain_pack_kernel(const __bf16* __restrict__ wcat, const __bf16* __restrict__ w1, __bf16* __restrict__ wp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // one 16-B element each; 4 * 8 * 4 * 4 * 64 = 32,768
    if (idx >= 32768) return;
    const int lane = idx & 63, f = (idx >> 6) & 3, w = (idx >> 8) & 3, st = (idx >> 10) & 7, c = idx >> 13;
    const int fr = lane & 15, fq = lane >> 4, m = f >> 1, kk = f & 1;
    const int rho = 32 * w + 16 * m + fr;
    const int cl = (rho & ~31) | (rho & 3) | (((rho >> 4) & 1) << 2) | (((rho >> 2) & 3) << 3);
    const __bf16* src;
    if (st < 6) src = wcat + (size_t)(c * 128 + cl) * 384 + st * 64 + kk * 32 + fq * 8;
    else src = w1 + (size_t)cl * 512 + c * 128 + (st - 6) * 64 + kk * 32 + fq * 8;
    reinterpret_cast<u32x4*>(wp)[idx] = *reinterpret_cast<const u32x4*>(src);
}

constexpr int CC_WDA = 4;     // group A's weight prefetch distance in steps (2 or 4)
constexpr int CC_PD = 6;      // LDS prefetch depth (pixel fragments) of a weight step
template <int ET, int OW>
__global__ __launch_bounds__(512) void bneck_catchain_kernel(const CatChainArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int CT2 = 128, CX = 256, COUT = 512, C1N = 128;
    constexpr int KSA = (CT2 + CX) / 64;          // resident K-slots
    constexpr int NCH = COUT / 128;               // chunks of 128 block-output channels
    constexpr int NR = 7, ROWS = 112;
    constexpr int SLOT = ROWS * 128;
    constexpr int XR = 0, OUTC = KSA * SLOT, B1_OFF = OUTC + 4 * SLOT, B3_OFF = B1_OFF + C1N * 4, SYNC_OFF = B3_OFF + COUT * 4;
    constexpr int OC_PASSES = 2 * ROWS / 32;      // 7
    constexpr int HOWO = OW * OW, WI = 2 * OW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wave & 3;
    const int lt = tid & 255, srow = lt >> 3;
    const int grid = gridDim.x;
    const int first = blockIdx.x;
    const int fr = lane & 15, fq = lane >> 4;

    if (tid < C1N) reinterpret_cast<float*>(smem + B1_OFF)[tid] = a.b1[tid];
    reinterpret_cast<float*>(smem + B3_OFF)[tid] = a.b3[tid];         // 512 threads, 512 biases
    if (tid < 4) reinterpret_cast<unsigned*>(smem + SYNC_OFF)[tid] = 0u;    // REL[0..2], LAND (see below)

    const __amdgpu_buffer_rsrc_t rs_wp = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.wp), 0, 524288u, 0x00020000);
    const unsigned wp_voff = (unsigned)(lane * 16 + w * 4096);
    const int fphys0 = (fq ^ (fr & 7)) << 4;
    const int x_frag = fr * 128;
    auto w_load_half = [&](int sofs, bf16x8 (&wf)[4], int kk) {       // sofs: scalar byte offset of the (chunk, step) slice
#pragma unroll
        for (int m = 0; m < 2; ++m)
            wf[2 * m + kk] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_wp, wp_voff + (2 * m + kk) * 1024, sofs, 0));
    };
    // one weight step (see bneck_tail3p_kernel); `sofs_next`: the slice this step's buffer is re-requested for
    auto w_step = [&](int sofs_next, bf16x8 (&wf)[4], const char* xb, f32x4 (&acc0)[NR], f32x4 (&acc1)[NR]) {
        constexpr int NS = 2 * NR, PD = CC_PD;
        bf16x8 x[NS];
        auto xread = [&](int t) { return *reinterpret_cast<const bf16x8*>(xb + x_frag + (t % NR) * 2048 + (t >= NR ? (fphys0 ^ 64) : fphys0)); };
#pragma unroll
        for (int t = 0; t < PD; ++t) x[t] = xread(t);
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int j = t % NR, kk = t / NR;
            {
                acc0[j] = mfma_e<ET>(wf[kk], x[t], acc0[j]);
                acc1[j] = mfma_e<ET>(wf[2 + kk], x[t], acc1[j]);
            }
            if (t + PD < NS) x[t + PD] = xread(t + PD);
            if (t == NR - 1) w_load_half(sofs_next, wf, 0);
        }
        w_load_half(sofs_next, wf, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if (t + PD < NS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (t == NR - 1) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        __builtin_amdgcn_sched_barrier(0);        // a scheduling region per step
    };
    auto sofs_of = [&](int c, int st) { return __builtin_amdgcn_readfirstlane(((c & (NCH - 1)) * 8 + st) << 14); };
    // The NEXT tile's operand rows are fetched while this tile's last chunk is still running (a fifth, load-only interval per tile cost 40 %
    // of the first version: every CU fetched its 84 KB at the same moment with nothing else to do).  K-slot pairs {0,1} {2,3} {4,5} are
    // released one by one as group A's waves finish reading them in the tile's last chunk -- REL[g] counts those waves -- and group B (whose
    // weight stream runs a whole chunk ahead, so HBM latencies in its in-order vmcnt queue stall nothing) re-fills each pair as soon as its
    // count is complete.  Pairs 0 and 1 have landed before the chunk barrier; pair 2 lands during the next interval: LAND counts the
    // group-B waves whose pieces are in, and group A looks at it before the next tile's first step on slot 4.
    LDS_AS unsigned* const sync = (LDS_AS unsigned*)(smem + SYNC_OFF);        // (explicit LDS address space: ds_read / ds_add, no generic-pointer casts)
    auto spin_until = [&](int word, unsigned target) {
        while (__hip_atomic_load(sync + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(2);
        asm volatile("" ::: "memory");            // nothing that follows (fragment reads, LDS-DMA issue) moves above the poll
    };
    auto signal = [&](int word) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the wave's own reads of the released slots have returned
        if (lane == 0) (void)__hip_atomic_fetch_add(sync + word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    bf16x8 wA[4], wB[4];

    if (wave < 4) {
        // =============================== group A: [W3 | Wd] . [t2 ; x] + ReLU -> out_c ==============
        const int c_frag = (w >> 1) * SLOT + x_frag + (((4 * (w & 1) + fq) ^ (fr & 7)) << 4);
        f32x4 accA[2][NR];
        // weight fragments CC_WDA steps ahead (4: a ring of four register buffers, step g in buffer g % 4 -- six steps per chunk, so the buffer
        // of a step alternates with the chunk's parity and the chunk loop is unrolled by two; 2: two buffers).  With two steps of cover
        // (8 KB per wave in flight) every step waited for its weights: L2 under this load answers in ~2,000 cycles, a step issues in ~450.
        bf16x8 wq[4][4];
#pragma unroll
        for (int q = 0; q < CC_WDA; ++q) { w_load_half(sofs_of(0, q), wq[q], 0); w_load_half(sofs_of(0, q), wq[q], 1); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // own bias / sync words written
        __builtin_amdgcn_s_barrier();             // P: biases, sync words and the first tile's operand rows (group B) are in LDS
        unsigned tl = 0;                          // tiles this workgroup has finished
        R50_STAMP_DECL
        auto chunk_a = [&](auto par_c, int c) {
            constexpr int PAR = decltype(par_c)::value;           // chunk parity: its step 0 sits in ring buffer (6 PAR) % CC_WDA
            constexpr int B0 = (6 * PAR) % CC_WDA;
            {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + B3_OFF + (c * 128 + 32 * w + 8 * fq) * 4);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + B3_OFF + (c * 128 + 32 * w + 8 * fq + 4) * 4);
#pragma unroll
                for (int j = 0; j < NR; ++j) { accA[0][j] = lo; accA[1][j] = hi; }
            }
            const bool last = (c == NCH - 1);
            // the slice step s's buffer is re-requested for: step s + CC_WDA of the stream
            auto nxt = [&](int st) { return st + CC_WDA < 6 ? sofs_of(c, st + CC_WDA) : sofs_of(c + 1, st + CC_WDA - 6); };
            w_step(nxt(0), wq[(B0 + 0) % CC_WDA], smem + XR + 0 * SLOT, accA[0], accA[1]);
            w_step(nxt(1), wq[(B0 + 1) % CC_WDA], smem + XR + 1 * SLOT, accA[0], accA[1]);
            if (last) signal(0);
            w_step(nxt(2), wq[(B0 + 2) % CC_WDA], smem + XR + 2 * SLOT, accA[0], accA[1]);
            w_step(nxt(3), wq[(B0 + 3) % CC_WDA], smem + XR + 3 * SLOT, accA[0], accA[1]);
            if (last) signal(1);
            if (c == 0) spin_until(3, 4u * tl);                     // slots 4 / 5 of THIS tile have landed (trivially true for the first tile)
            R50_MARK(5)                           // steps 0-3 (+ the poll of the late operand pair)
            w_step(nxt(4), wq[(B0 + 4) % CC_WDA], smem + XR + 4 * SLOT, accA[0], accA[1]);
            w_step(nxt(5), wq[(B0 + 5) % CC_WDA], smem + XR + 5 * SLOT, accA[0], accA[1]);
            if (last) signal(2);
#if defined(R50_STAMP)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 0" ::"v"(accA[1][NR - 1]), "v"(accA[0][NR - 1]) : "memory");
#endif
            R50_MARK(0)                           // steps 4-5
            __builtin_amdgcn_sched_barrier(0);
            char* ob = smem + OUTC + PAR * 2 * SLOT + c_frag;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const f32x4 lo = accA[0][j], hi = accA[1][j];
                u32x4 o = (u32x4){pack2_e<ET>(lo[0], lo[1]), pack2_e<ET>(lo[2], lo[3]), pack2_e<ET>(hi[0], hi[1]), pack2_e<ET>(hi[2], hi[3])};
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = relu_bf16x2(o[e]);
                *reinterpret_cast<u32x4*>(ob + j * 2048) = o;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // out_c written before anybody reads it
            R50_MARK(1)                           // E
            __builtin_amdgcn_s_barrier();
            R50_MARK(2)                           // chunk barrier
        };
        for (int tile = first; tile < a.n_tiles; tile += grid, ++tl) {
            for (int c2 = 0; c2 < NCH / 2; ++c2) {
                chunk_a(std::integral_constant<int, 0>{}, 2 * c2);
                chunk_a(std::integral_constant<int, 1>{}, 2 * c2 + 1);
            }
        }
        __builtin_amdgcn_s_barrier();             // the interval in which group B finishes the last chunk
        R50_STAMP_FLUSH(8)
    } else {
        // =============================== group B: next conv1 + copy-out + the next tile's operand rows =
        const __amdgpu_buffer_rsrc_t rs_t2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.t2), 0, (unsigned)a.M * (CT2 * 2u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x), 0, a.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_y1 = __builtin_amdgcn_make_buffer_rsrc(a.y1n, 0, (unsigned)a.M * (C1N * 2u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)a.M * (COUT * 2u), 0x00020000);
        // Operand rows: row R = 32 i + srow of the [6 slots][112] region: slot R / 112 (wave-uniform: 112 is a multiple of the 8 rows of a wave's
        // piece), pixel row (32 i + srow) % 112 = one of SEVEN values per lane (32 i mod 112 has period 7).  Per tile and value: the lane's byte offset
        // in t2 and in the block input (pixel (n, ho, wo) reads input pixel (n, 2 ho, 2 wo)), chunk swizzle included; slot offsets are scalar.
        // Passes 0-6 fill slots 0-1, 7-13 slots 2-3, 14-20 slots 4-5.
        unsigned vt[7], vx[7];
        auto x_rows = [&](int tile) {
            const int p0 = tile * a.bp;
            const int limit = (tile < a.n_tiles) ? ((a.M - p0 < a.bp) ? a.M - p0 : a.bp) : 0;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const int prow = (32 * k + srow) % ROWS;
                const int lchunk = (lt & 7) ^ (prow & 7);
                const unsigned p = (unsigned)(p0 + prow);
                const unsigned n = p / HOWO, r = p - n * HOWO, ho = r / OW, wo = r - ho * OW;
                const unsigned q = a.x_sub ? p : (n * WI + 2 * ho) * WI + 2 * wo;
                const bool ok = prow < limit;
                vt[k] = ok ? (p * CT2 + lchunk * 8) * 2u : kOobOffset;
                vx[k] = ok ? (q * CX + lchunk * 8) * 2u : kOobOffset;
            }
        };
        auto issue_x = [&](auto grp_c) {          // the 7 passes of slot pair GRP
            constexpr int GRP = decltype(grp_c)::value;
#pragma unroll
            for (int i = 7 * GRP; i < 7 * GRP + 7; ++i) {
                const int sl = __builtin_amdgcn_readfirstlane((32 * i + 8 * w) / ROWS);
                char* dst = smem + XR + i * 4096 + w * 1024;
                if (GRP == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_t2, (LDS_AS void*)dst, 16, vt[i % 7], sl * 128, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (LDS_AS void*)dst, 16, vx[i % 7], (sl - 2) * 128, 0, 0);
            }
        };
        using G0 = std::integral_constant<int, 0>;
        using G1 = std::integral_constant<int, 1>;
        using G2 = std::integral_constant<int, 2>;
        f32x4 accB[2][NR];
        unsigned rvo[OC_PASSES];
        auto out_rows = [&](int tile) {
            const int p0 = tile * a.bp;
            const int limit = (a.M - p0 < a.bp) ? a.M - p0 : a.bp;
#pragma unroll
            for (int i = 0; i < OC_PASSES; ++i) {
                const int R = 32 * i + srow;
                const int sl = R / ROWS, prow = R - ROWS * sl;
                const int lchunk = (lt & 7) ^ (prow & 7);
                rvo[i] = (prow < limit) ? (unsigned)((p0 + prow) * COUT + sl * 64 + lchunk * 8) * 2u : kOobOffset;
            }
        };
        auto y1n_store = [&](int tile) {
            const int p0 = tile * a.bp;
            const int limit = (a.M - p0 < a.bp) ? a.M - p0 : a.bp;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const f32x4 lo = accB[0][j], hi = accB[1][j];
                u32x4 o = (u32x4){pack2_e<ET>(lo[0], lo[1]), pack2_e<ET>(lo[2], lo[3]), pack2_e<ET>(hi[0], hi[1]), pack2_e<ET>(hi[2], hi[3])};
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = relu_bf16x2(o[e]);
                const unsigned voff = (16 * j + fr < limit) ? (unsigned)((p0 + 16 * j + fr) * C1N + 32 * w + 8 * fq) * 2u : kOobOffset;
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_y1, voff, 0, 0);
            }
        };
        auto init_accb = [&]() {                  // accB[e]: channels 32w + 8fq + 4e ..
            const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + B1_OFF + (32 * w + 8 * fq) * 4);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + B1_OFF + (32 * w + 8 * fq + 4) * 4);
#pragma unroll
            for (int j = 0; j < NR; ++j) { accB[0][j] = lo; accB[1][j] = hi; }
        };
        x_rows(first);
        issue_x(G0{}); issue_x(G1{}); issue_x(G2{});
        w_load_half(sofs_of(0, 6), wA, 0); w_load_half(sofs_of(0, 6), wA, 1);
        w_load_half(sofs_of(0, 7), wB, 0); w_load_half(sofs_of(0, 7), wB, 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // own operand pieces landed, own bias words written
        __builtin_amdgcn_s_barrier();             // P
        init_accb();
        __builtin_amdgcn_s_barrier();             // first interval: group A is on chunk 0 of its first tile
        unsigned tl = 0;
        R50_STAMP_DECL
        for (int tile = first; tile < a.n_tiles; tile += grid, ++tl) {
            out_rows(tile);
            for (int c = 0; c < NCH; ++c) {       // (group A is on chunk c + 1: for c = 3 that is the next tile's chunk 0)
                const char* xb = smem + OUTC + (c & 1) * 2 * SLOT;
                if (c == NCH - 2) x_rows(tile + grid);
                w_step(sofs_of(c + 1, 6), wA, xb, accB[0], accB[1]);
                R50_MARK(2)                       // B: weight step 0
                if (c == NCH - 2) { spin_until(0, 4u * (tl + 1)); issue_x(G0{}); }
                R50_MARK(5)                       // release wait + operand issue
                if (c == NCH - 1) {               // the late pair of the tile group A has just started: landed behind the 4 weight loads of that step
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    signal(3);
                }
                R50_MARK(6)                       // late pair landing wait + signal
                w_step(sofs_of(c + 1, 7), wB, xb + SLOT, accB[0], accB[1]);
                R50_MARK(2)                       // B: weight step 1
                if (c == NCH - 2) { spin_until(1, 4u * (tl + 1)); issue_x(G1{}); }
                R50_MARK(5)
#if defined(R50_STAMP)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 0" ::"v"(accB[1][NR - 1]), "v"(accB[0][NR - 1]) : "memory");
#endif
                R50_MARK(2)                       // B: 2 weight steps (+ operand issue)
                __builtin_amdgcn_sched_barrier(0);
                {
                    const int cofs = __builtin_amdgcn_readfirstlane(c * 256);
                    u32x4 v[OC_PASSES];
#pragma unroll
                    for (int i = 0; i < OC_PASSES; ++i) v[i] = *reinterpret_cast<const u32x4*>(xb + i * 4096 + lt * 16);
#pragma unroll
                    for (int i = 0; i < OC_PASSES; ++i)
                        __builtin_amdgcn_raw_buffer_store_b128(v[i], rs_out, rvo[i], cofs, 0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                R50_MARK(3)                       // copy-out
                if (c == NCH - 2) {
                    spin_until(2, 4u * (tl + 1));
                    issue_x(G2{});
                    R50_MARK(5)
                    asm volatile("s_waitcnt vmcnt(14)" ::: "memory");     // pairs 0 and 1 have landed (younger: 7 stores, 7 pieces of pair 2)
                    R50_MARK(6)
                }
                if (c == NCH - 1) { y1n_store(tile); init_accb(); }
                R50_MARK(0)                       // late pair issue / y1n
                __builtin_amdgcn_s_barrier();
                R50_MARK(4)                       // chunk barrier
            }
        }
        R50_STAMP_FLUSH(8)
    }
#else
    (void)a;
#endif
}

// ------------------------------------------------------------------------------------------------
// Bottleneck BODY (layer2.1-.3), one launch per block: conv2 3x3 (128 -> 128) + bn2 + ReLU, conv3 1x1 (128 -> 512) + bn3 + identity +
// ReLU and -- C1N = 128 -- the NEXT block's conv1 1x1 (512 -> 128) + bn1 + ReLU (src/preprocess_resnet_features.py:296 -> torchvision
// Bottleneck.forward, restated in oracle/resnet50_oracle.py).
// Why: at 28x28 the layer-wise form moves t1 in, t2 out (conv2 launch) and t2 in, identity in, block output out, next t1 out (fused
// tail): 0.61 GB per block at batch 256.  Here t2 never leaves the CU (0.51 GB: 17 % fewer HBM bytes -- the first-order bound of
// this network at batch 256 is its 10 GB of HBM traffic per step, not its 2.1 TFLOP) and the block is one launch instead of two.
// All three GEMMs are 128 wide in their small dimension, so an accumulator set is at most 128 x 208 fp32 = 56 registers per consumer
// wave and two live sets + fragments fit the 168-register budget of 12 waves (layer3's 256-wide GEMMs do not).
// Tile = a band of 7 rows x 28 columns of one image = 196 pixels = 13 MFMA column blocks (1024 tiles at batch 256: four per CU).
// 8 consumer waves + 4 loader waves.  One STREAM of 16-KB weight stages runs through a ring of three LDS stages, one barrier per stage:
//     conv2:  18 stages [128 couts][64 K] (chunk, tap) of W2 against the tile's t1 band -- both 64-channel chunks of the band, with
//             halo, are resident in LDS (XB0, XB1); K order (chunk, tap, channel) exactly as conv3x3_xres_kernel: t2 has that kernel's
//             bits.  Waves: cout group w & 3 (32 rows), pixel half w >> 2 (blocks 0..6 / 7..12).
//     then t2 = relu(acc) -> 16 bit -> LDS (T2: the B operand of conv3, two 64-channel K-slots of 208 pixel rows)
//     C1N = 128, for c in 0..7 (64 block-output channels each; waves: cout group w & 1 (32 rows), pixel quarter w >> 1 (4 + 3 + 3 + 3 blocks)):
//         A(c): one stage [2 K-slots][64 rows] of W3[64c ..] against T2
//         E(c): + identity (RESB: DMA'd in by the loaders one chunk ahead, full rows), ReLU, 16 bit -> OUTC = the block output's chunk
//               (copied to HBM by the loaders at the next A position, full rows) AND the K-slice of the next conv1
//         B(c): one stage [128 rows][64 K] of W1[:, 64c ..] against OUTC into the next conv1's accumulators (wave split of conv2), which
//               stay in registers over the chunks;   finally next t1 = relu(accB) -> HBM
//     C1N = 0 (the stage's last block), for c in 0..3 (128 channels each, wave split of conv2): 2 stages of W3[128c ..] against T2, then the
//         consumers add the identity and store the block output themselves (fragment-shaped accesses)
// Summation orders are those of the launches this replaces (bias first, K ascending, identity last with the same fp32 additions):
// block output and next t1 are bit-identical to conv3x3_xres + igemm launches.
// LDS, C1N = 128: XB0 | XB1 2 x 36,864 (288 padded positions x 128 B -- 9 rows of 32, round 3 -- chunk c of a position at c ^ (column & 7)); T2 2 x 26,624 at 0 and OUTC
// 26,624 at 53,248 overlay them once conv2 is done; RESB 28,672 (224 rows: 7 DMA passes); ring 3 x 16,384; biases 3,072 = 160,768 B.  The
// next tile's XB0 can only be fetched once T2 is dead (last position), XB1 at the next tile's first position (needed nine stages later).
// C1N = 0: no OUTC / RESB, T2 at 53,248 (behind XB0, so the next tile's first chunk is fetched two positions early) = 158,720 B.
// ------------------------------------------------------------------------------------------------
struct Block2Args {
    const __bf16* t1;     // (N,28,28,128)  this block's conv1 output
    const __bf16* w2;     // (128, 3, 3, 128) folded conv2 weights, K-major (tap, cin)
    const float* b2;      // (128)
    const __bf16* w3;     // (512, 128)
    const float* b3;      // (512)
    const __bf16* res;    // (M, 512)  identity
    __bf16* out;          // (M, 512)  block output
    const __bf16* w1;     // (128, 512) next block's conv1 (C1N = 128)
    const float* b1;      // (128)
    __bf16* y1n;          // (M, 128)
    int N;
    int n_tiles;          // 4 N
};

#define B2_LOADER_LOOP _Pragma("unroll")      // the loaders' position loop fully unrolled (rolled: 4-20 % slower bodies, profiles/r03_block_loader_unroll_ab.txt)
template <int ET, int C1N>
__global__ __launch_bounds__(768) void bneck_block2_kernel(const Block2Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(C1N == 0 || C1N == 128, "next conv1: none or 512 -> 128");
    // (round 3) the band is held in the ROW-BLOCK form of conv3x3_xres_kernel: padded row = 32 positions, an MFMA column block of conv2 = 16
    // consecutive positions (half an image row: 14 blocks, 7 per slot half), fragment address = lane constant + immediate, taps unrolled
    constexpr int IW = 28, TR = 7, PW = 32, PP = (TR + 2) * PW;           // 288 padded positions
    constexpr int XPASS = (PP + 31) / 32, XBUF = XPASS * 32 * 128;        // 9 passes, 36,864 B per 64-channel chunk
    constexpr int NPX = TR * IW;                                          // 196
    constexpr int SLOT = 208 * 128;                                       // one 64-channel K-slot of 208 pixel rows
    // LDS map   C1N = 128:  XB0 | XB1 (T2 at 0 and OUTC at 53,248, one K-slot, overlay them) | RESB (224 rows) | ring | biases = 160,768 B
    //           C1N = 0:    XB0 | XB1 ... T2 at 53,248 (over the end of XB1) ... | ring | biases                              = 158,720 B
    constexpr int XB_OFF = 0, T2_OFF = C1N ? 0 : 2 * SLOT, OUTC_OFF = 2 * SLOT, RESB_OFF = 3 * SLOT, RESB_BYTES = 224 * 128;
    constexpr int RING_OFF = C1N ? RESB_OFF + RESB_BYTES : 4 * SLOT;
    constexpr int NST = 3, WSTAGE = 128 * 128, WPASS = 4;
    constexpr int BIAS_OFF = RING_OFF + NST * WSTAGE;                     // b2 (128) | b3 (512) | b1 (128) floats
    // stages per tile: 18 of conv2, then  C1N = 0: 4 chunks of 128 couts x 2 K-slots of W3
    //                                     C1N = 128: 8 chunks of 64 couts x { A: W3 rows, both K-slots in one stage ; B: W1 K-slice }
    constexpr int NCONV = 18, SPT = NCONV + (C1N ? 16 : 8);
    static_assert(2 * XBUF <= (C1N ? RESB_OFF : RING_OFF) && BIAS_OFF + 768 * 4 <= 163840, "LDS map");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grid = gridDim.x, first = blockIdx.x;

    {
        float v;
        if (tid < 128) v = a.b2[tid];
        else if (tid < 640) v = a.b3[tid - 128];
        else v = (C1N ? a.b1[tid - 640] : 0.f);
        reinterpret_cast<float*>(smem + BIAS_OFF)[tid] = v;
    }

    if (wave >= 8) {
        // =============================== loader waves ===============================================
        const int lw = wave - 8, lt = tid - 512;
        const int srow = lt >> 3, slot = lt & 7;
        const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w2), 0, 128u * 1152u * 2u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w3), 0, 512u * 128u * 2u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w1), 0, C1N ? 128u * 512u * 2u : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.t1), 0, (unsigned)a.N * (784u * 256u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.res), 0, (unsigned)a.N * (784u * 1024u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)a.N * (784u * 1024u), 0x00020000);
        unsigned w2v[WPASS], w3v[WPASS], w1v[WPASS], x_voff[XPASS];
#pragma unroll
        for (int i = 0; i < WPASS; ++i) {
            const int rho = i * 32 + srow;                               // LDS row of the stage
            const int cl = (rho & ~31) | (rho & 3) | (((rho >> 4) & 1) << 2) | (((rho >> 2) & 3) << 3);
            const unsigned ch = (unsigned)((slot ^ (rho & 7)) * 8);
            w2v[i] = ((unsigned)cl * 1152u + ch) * 2u;
            w1v[i] = ((unsigned)cl * 512u + ch) * 2u;
            if (C1N) {                                                   // A stage: rows 64 ks + rho' = K-slot ks of cout perm(rho') of the chunk
                const int rp = rho & 63, ks = rho >> 6;
                const int cl2 = (rp & ~31) | (rp & 3) | (((rp >> 4) & 1) << 2) | (((rp >> 2) & 3) << 3);
                w3v[i] = ((unsigned)cl2 * 128u + (unsigned)ks * 64u + ch) * 2u;
            } else {
                w3v[i] = ((unsigned)cl * 128u + ch) * 2u;
            }
        }
        auto decode_band = [&](int tile) {        // source offsets of the padded positions of `tile` (out of range = zero border)
            const int n = tile >> 2, band = tile & 3;
#pragma unroll
            for (int i = 0; i < XPASS; ++i) {
                const int q = i * 32 + srow;
                const int rr = q / PW, cc = q - rr * PW;
                const int y = band * TR + rr - 1, x = cc - 1;
                const bool ok = q < PP && tile < a.n_tiles && (unsigned)y < 28u && (unsigned)x < 28u;
                // chunk key of a position = its column's, cc & 7 (PW = 0 mod 8): the same for every block and kernel row a lane reads it for
                x_voff[i] = (ok) ? (unsigned)(((n * 28 + y) * 28 + x) * 128 + (slot ^ (cc & 7)) * 8) * 2u : kOobOffset;
            }
        };
        auto issue_band = [&](int c2) {           // 9 DMAs per wave: chunk c2 of the band decoded last
            const int xofs = __builtin_amdgcn_readfirstlane(c2 * 128);
#pragma unroll
            for (int i = 0; i < XPASS; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (LDS_AS void*)(smem + XB_OFF + c2 * XBUF + i * 4096 + lw * 1024), 16, x_voff[i], xofs, 0, 0);
        };
        // C1N = 128: the identity and the block output move in FULL 128-B row pieces through the loaders (row R = 32 i + srow of the
        // tile, 16-B chunk slot ^ (R & 7) of the 64-channel chunk): identity rows by LDS-DMA into RESB one chunk ahead, out_c copied
        // from OUTC to HBM with 16-B stores -- the consumers' own accesses would be MFMA-fragment shaped (16 rows x 64 B per
        // instruction), several times the address-unit cost per byte
        // per-lane part of the row offsets, fixed for the launch (rows past the tile's 196: out of range); the tile and the chunk go into
        // the scalar offset, so the loop issues these 7 + 7 + 7 instructions without any vector arithmetic
        unsigned rv[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int R = 32 * i + srow;
            rv[i] = (R < NPX) ? ((unsigned)R * 512u + (unsigned)((slot ^ (R & 7)) * 8)) * 2u : kOobOffset;
        }
        auto issue_res = [&](unsigned tile_pix0, int c) {       // 7 DMAs per wave
            const int sofs = __builtin_amdgcn_readfirstlane((int)(tile_pix0 * 1024u) + c * 128);
#pragma unroll
            for (int i = 0; i < 7; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_res, (LDS_AS void*)(smem + RESB_OFF + i * 4096 + lw * 1024), 16, rv[i], sofs, 0, 0);
        };
        auto copy_out = [&](unsigned tile_pix0, int c) {        // 7 LDS reads + 7 stores per wave
            const int sofs = __builtin_amdgcn_readfirstlane((int)(tile_pix0 * 1024u) + c * 128);
            u32x4 v[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) v[i] = *reinterpret_cast<const u32x4*>(smem + OUTC_OFF + i * 4096 + lt * 16);
#pragma unroll
            for (int i = 0; i < 7; ++i) __builtin_amdgcn_raw_buffer_store_b128(v[i], rs_out, rv[i], sofs, 0);
        };
        int ring = 0;                             // ring slot of the next stage to issue
        auto stage_issue = [&](int p) {           // stage p (0 .. SPT-1) of a tile; 4 DMAs per wave
            char* sbase = smem + RING_OFF + ring * WSTAGE + lw * 1024;
            ring = (ring == NST - 1) ? 0 : ring + 1;
            if (p < NCONV) {
                const int c2 = p >= 9 ? 1 : 0, tap = p - 9 * c2;
                const int sofs = __builtin_amdgcn_readfirstlane((tap * 128 + c2 * 64) * 2);
#pragma unroll
                for (int i = 0; i < WPASS; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w2, (LDS_AS void*)(sbase + i * 4096), 16, w2v[i], sofs, 0, 0);
            } else {
                const int q = p - NCONV, c = q >> 1, r = q & 1;
                if (!C1N || r == 0) {
                    const int sofs = __builtin_amdgcn_readfirstlane(C1N ? c * 64 * 128 * 2 : (c * 128 * 128 + r * 64) * 2);
#pragma unroll
                    for (int i = 0; i < WPASS; ++i)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w3, (LDS_AS void*)(sbase + i * 4096), 16, w3v[i], sofs, 0, 0);
                } else {
                    const int sofs = __builtin_amdgcn_readfirstlane(c * 64 * 2);
#pragma unroll
                    for (int i = 0; i < WPASS; ++i)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (LDS_AS void*)(sbase + i * 4096), 16, w1v[i], sofs, 0, 0);
                }
            }
        };
        // all but the n youngest vector-memory operations of this wave are complete
        auto wait_younger = [&](int n) { wait_vmcnt(n); };          // all but the n youngest vector-memory operations of this wave are complete
        decode_band(first);
        issue_band(0);
        issue_band(1);
        stage_issue(0);
        stage_issue(1);
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");       // both band chunks and stage 0 landed (stage 1 may be in flight); bias writes done
        __builtin_amdgcn_s_barrier();
        bool first_tile = true;
        int carry = 0;                            // stores of the previous iteration (issued behind its DMAs)
        // chunk 0 of the NEXT tile's band goes in once its LDS region is dead: C1N = 128: T2 overlays it, read until the last A stage
        // (position SPT - 2), so it is issued at SPT - 1 and must land before that iteration's barrier (the next tile starts behind it);
        // C1N = 0: T2 lies elsewhere, issued at SPT - 2 and waited for one iteration later with the stage issued behind it
        constexpr int PBAND = C1N ? SPT - 1 : SPT - 2;
#pragma unroll 1
        for (int tile = first; tile < a.n_tiles; tile += grid) {
            const bool has_next = tile + grid < a.n_tiles;
            const unsigned tile_pix0 = (unsigned)((tile >> 2) * 784 + (tile & 3) * NPX);
            // (round 3) the position loop is UNROLLED: with p a constant, every decision below -- which DMAs this iteration carries, their scalar
            // offsets, the vmcnt to wait for -- folds at compile time; rolled, the loaders spent a few hundred cycles of scalar compares and
            // branches per barrier interval on them, in the phase (A / E / B) where the loaders pace the kernel
            B2_LOADER_LOOP
            for (int p = 0; p < SPT; ++p) {
                // the consumers' extra barriers: T2 complete (in front of stage 18), OUTC(c) complete (in front of every B stage)
                if (p == NCONV || (C1N && p > NCONV && ((p - NCONV) & 1) == 1)) __builtin_amdgcn_s_barrier();
                // `younger`: operations issued behind the last one that has to be complete at this iteration's barrier
                int younger = carry;
                carry = 0;
                // chunk 1 of THIS tile's band: its LDS region overlaps the previous tile's T2 / OUTC, free since that tile's last barrier
                if (p == 0 && !first_tile) { issue_band(1); younger += XPASS; }
                if (p == PBAND && has_next) { decode_band(tile + grid); issue_band(0); if (C1N) younger = 0; else younger += XPASS; }
                const bool bpos = C1N && p > NCONV && ((p - NCONV) & 1) == 1;      // B stage of chunk (p - 19) / 2
                // identity rows of the NEXT chunk (chunk 0: two stages before the first A stage): needed one barrier later
                if (C1N && p == NCONV - 2) { issue_res(tile_pix0, 0); younger += 7; }
                if (bpos && p + 1 < SPT) { issue_res(tile_pix0, (p - NCONV + 1) / 2); younger += 7; }
                const bool st = (p + 2 < SPT) || has_next;
                if (st) { stage_issue(p + 2 < SPT ? p + 2 : p + 2 - SPT); younger += WPASS; }
                // What the loaders do per iteration has to stay below what the consumers do per stage (~1,000-1,400 cycles), or every barrier
                // waits for the loaders (an ablation build without the identity / copy-out instructions took 123 us instead of 158).
                // out_c(c) is complete behind the extra barrier in front of B(c) and stays intact until E(c + 1), i.e. over two iterations:
                // its copy-out (LDS -> HBM) goes into the SECOND one, the A(c + 1) position, which carries nothing but a weight stage (with
                // identity DMAs, stage and copy-out all in the B position: 155 us; so: 145 us).  Chunk 7 has no A position behind it: copied
                // at its own B position.  (Also tried: identity rows and the next band prefetched into loader REGISTERS two iterations
                // early and written to LDS when the buffer falls free, instead of an LDS-DMA issued at that moment: 155 us, slower.)
                const bool cpos = C1N && p > NCONV + 1 && (((p - NCONV) & 1) == 0 || p == SPT - 1);
                if (cpos) {                       // the stores are not waited for here, nor at the next iteration (whose wait covers this iteration's DMAs)
                    copy_out(tile_pix0, p == SPT - 1 ? 7 : (p - NCONV) / 2 - 1);
                    younger += 7;
                    carry = 7;
                }
                wait_younger(younger);            // stage p + 1, the band chunk / identity rows due now and everything older landed
                __builtin_amdgcn_s_barrier();
            }
            first_tile = false;
        }
    } else {
        // =============================== consumer waves =============================================
        const int wave_c = wave & 3, wave_p = wave >> 2;                 // conv2 / next conv1: cout group (32 of 128), pixel half (blocks 0..6 / 7..12)
        const int wave_a = wave & 1, wave_q = wave >> 1;                 // conv3 chunk (C1N = 128): cout group (32 of 64), pixel quarter (4 + 3 + 3 + 3 blocks)
        const int qb0 = wave_q == 0 ? 0 : 1 + 3 * wave_q;
        const int fr = lane & 15, fq = lane >> 4;
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.res), 0, (unsigned)a.N * (784u * 1024u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)a.N * (784u * 1024u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_y1 = __builtin_amdgcn_make_buffer_rsrc(a.y1n, 0, (unsigned)a.N * (784u * 256u), 0x00020000);
        const int w_row = (wave_c * 32 + fr) * 128;                       // + m * 2048
        const int w_rowa = (wave_a * 32 + fr) * 128;                      // A stage of the chained form: + ks * 8192 + m * 2048
        const int w_ph0 = (fq ^ (fr & 7)) << 4;                          // kk = 0; kk = 1 is ^ 64
        const int ch_lane = wave_c * 32 + 8 * fq;                        // this lane's 8 consecutive channels of a 128-channel group
        const int cha_lane = wave_a * 32 + 8 * fq;                       // ... of a 64-channel chunk
        // pixel row p = p0 + 16 j of block j: B fragment (kk = 0) at pb0 + 2048 j of a K-slot (p & 7 does not depend on j); kk = 1 is ^ 64
        const int p0 = 16 * 7 * wave_p + fr, pb0 = p0 * 128 + ((fq ^ (p0 & 7)) << 4);
        const int pa0 = 16 * qb0 + fr, pba0 = pa0 * 128 + ((fq ^ (pa0 & 7)) << 4);
        // this lane's 8 channels of pixel row p as the B operand of the NEXT GEMM: chunk 4 (group & 1) + fq of K-slot group >> 1
        const int cf_x = (wave_c & 1) << 6, cf_a = (wave_c >> 1) * SLOT;

        auto run = [&](auto nrw_c, auto nq_c) {
            constexpr int NRW = decltype(nrw_c)::value, NQ = decltype(nq_c)::value;
            f32x4 acc[2][NRW], accA[2][NQ], accC[2][7];        // accC: conv2 (7 slot blocks in either half); dead once t2 is in LDS
            int cbuf = 0;
            // acc[m][j] += W[rows 16 m + fr of `wb`][64 K] . X[64 K][pixel block j]; `xaddr(t)` = address of the fragment of slot t = NB kk + j
            auto gemm64 = [&](auto nb_c, const char* wb, auto xaddr, auto& ac) {
                constexpr int NB = decltype(nb_c)::value, NS = 2 * NB, PD = 3;
                bf16x8 x[NS], wf[2], wg[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) wf[m] = *reinterpret_cast<const bf16x8*>(wb + m * 2048 + w_ph0);
#pragma unroll
                for (int m = 0; m < 2; ++m) wg[m] = *reinterpret_cast<const bf16x8*>(wb + m * 2048 + (w_ph0 ^ 64));
#pragma unroll
                for (int t = 0; t < PD; ++t) x[t] = *reinterpret_cast<const bf16x8*>(xaddr(t));
#pragma unroll
                for (int t = 0; t < NS; ++t) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        ac[m][t % NB] = mfma_e<ET>(t >= NB ? wg[m] : wf[m], x[t], ac[m][t % NB]);
                    }
                    if (t + PD < NS) x[t + PD] = *reinterpret_cast<const bf16x8*>(xaddr(t + PD));
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 4 + PD, 0);
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    if (t + PD < NS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            auto stage_done = [&]() {             // the barrier stays behind the stage's last fragment read
                cbuf = (cbuf == NST - 1) ? 0 : cbuf + 1;
                __builtin_amdgcn_s_barrier();
            };
            auto pack_relu = [&](const f32x4& lo, const f32x4& hi) {
                u32x4 o = (u32x4){pack2_e<ET>(lo[0], lo[1]), pack2_e<ET>(lo[2], lo[3]), pack2_e<ET>(hi[0], hi[1]), pack2_e<ET>(hi[2], hi[3])};
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = relu_bf16x2(o[e]);
                return o;
            };
            auto add_identity = [&](f32x4& lo, f32x4& hi, const u32x4& r) {
                lo[0] += unpack_lo_e<ET>(r[0]); lo[1] += unpack_hi_e<ET>(r[0]);
                lo[2] += unpack_lo_e<ET>(r[1]); lo[3] += unpack_hi_e<ET>(r[1]);
                hi[0] += unpack_lo_e<ET>(r[2]); hi[1] += unpack_hi_e<ET>(r[2]);
                hi[2] += unpack_lo_e<ET>(r[3]); hi[3] += unpack_hi_e<ET>(r[3]);
            };
            constexpr std::integral_constant<int, NRW> NRWc{};
            constexpr std::integral_constant<int, NQ> NQc{};
#pragma unroll 1
            for (int tile = first; tile < a.n_tiles; tile += grid) {
                const unsigned tile_pix0 = (unsigned)((tile >> 2) * 784 + (tile & 3) * NPX);
                // ---- conv2: bias, then 18 stages against the resident band (row blocks: slot 16 b + fr of block b = 7 wave_p + j is output
                //      row b >> 1, column 16 (b & 1) + fr; columns 28..31 are unused slots)
                {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + (ch_lane) * 4);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + (ch_lane + 4) * 4);
#pragma unroll
                    for (int j = 0; j < 7; ++j) { accC[0][j] = lo; accC[1][j] = hi; }
                }
                // lane constants of the fragment addresses, recomputed per tile (an opaque zero keeps them out of the registers that live
                // across the whole tile loop: the chained tail needs those)
                int opq = 0;
                asm volatile("" : "+v"(opq));
                {
                    int vb[3][2];                                       // [tap column][K half]: position fr + kw, chunk (fq + 4 kk) ^ key
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) vb[kw][kk] = (fr + kw + opq) * 128 + (((fq + 4 * kk) ^ ((fr + kw) & 7)) << 4);
#pragma unroll 1
                    for (int c2 = 0; c2 < 2; ++c2) {
                        const char* const xc = smem + XB_OFF + c2 * XBUF + 7 * 2048 * wave_p;
#pragma unroll
                        for (int tap = 0; tap < 9; ++tap) {
                            const int kh = tap / 3, kw = tap - 3 * kh;
                            const char* const x0 = xc + vb[kw][0] + kh * (PW * 128);
                            const char* const x1 = xc + vb[kw][1] + kh * (PW * 128);
                            gemm64(std::integral_constant<int, 7>{}, smem + RING_OFF + cbuf * WSTAGE + w_row, [&](int t) {
                                return (t >= 7 ? x1 : x0) + (t % 7) * 2048;
                            }, accC);
                            stage_done();
                        }
                    }
                }
                // ---- t2 = relu(accC) -> T2, compact pixel rows p = 28 r + column (everybody is past the last conv2 stage's barrier: the band is dead)
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const int b = 7 * wave_p + j;
                    const int col = 16 * (b & 1) + fr, p = 28 * (b >> 1) + col + opq;
                    if (col < IW) *reinterpret_cast<u32x4*>(smem + T2_OFF + cf_a + ((p * 128 + ((fq ^ (p & 7)) << 4)) ^ cf_x)) = pack_relu(accC[0][j], accC[1][j]);
                }
                if constexpr (C1N != 0) {         // the next conv1's accumulators take over conv2's registers
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + (640 + ch_lane) * 4);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + (640 + ch_lane + 4) * 4);
#pragma unroll
                    for (int j = 0; j < NRW; ++j) { acc[0][j] = lo; acc[1][j] = hi; }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                   // T2 complete
                if constexpr (C1N == 0) {
                    const unsigned pix0 = tile_pix0 + (unsigned)p0;       // + 16 j
#pragma unroll 1
                    for (int c = 0; c < 4; ++c) {
                        // identity slice of this chunk: requested now, used behind the two A stages
                        u32x4 r[NRW];
#pragma unroll
                        for (int j = 0; j < NRW; ++j) {
                            const unsigned voff = (p0 + 16 * j < NPX) ? ((pix0 + 16 * j) * 512u + (unsigned)(c * 128 + ch_lane)) * 2u : kOobOffset;
                            r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, voff, 0, 0);
                        }
                        {
                            const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + (128 + c * 128 + ch_lane) * 4);
                            const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + (128 + c * 128 + ch_lane + 4) * 4);
#pragma unroll
                            for (int j = 0; j < NRW; ++j) { acc[0][j] = lo; acc[1][j] = hi; }
                        }
#pragma unroll 1
                        for (int ks = 0; ks < 2; ++ks) {
                            const char* bb = smem + T2_OFF + ks * SLOT;
                            gemm64(NRWc, smem + RING_OFF + cbuf * WSTAGE + w_row, [&](int t) { return bb + ((pb0 + 2048 * (t % NRW)) ^ (t >= NRW ? 64 : 0)); }, acc);
                            stage_done();
                        }
#pragma unroll
                        for (int j = 0; j < NRW; ++j) {
                            f32x4 lo = acc[0][j], hi = acc[1][j];
                            add_identity(lo, hi, r[j]);
                            const u32x4 o = pack_relu(lo, hi);
                            const unsigned voff = (p0 + 16 * j < NPX) ? ((pix0 + 16 * j) * 512u + (unsigned)(c * 128 + ch_lane)) * 2u : kOobOffset;
                            __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, voff, 0, 0);
                        }
                    }
                } else {
#pragma unroll 1
                    for (int c = 0; c < 8; ++c) {
                        {
                            const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + (128 + c * 64 + cha_lane) * 4);
                            const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + (128 + c * 64 + cha_lane + 4) * 4);
#pragma unroll
                            for (int j = 0; j < NQ; ++j) { accA[0][j] = lo; accA[1][j] = hi; }
                        }
                        // ---- A(c): W3[64 c ..] . t2, both K-slots from ONE stage (rows 64 ks + ..)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            const char* bb = smem + T2_OFF + ks * SLOT;
                            gemm64(NQc, smem + RING_OFF + cbuf * WSTAGE + ks * 8192 + w_rowa, [&](int t) { return bb + ((pba0 + 2048 * (t % NQ)) ^ (t >= NQ ? 64 : 0)); }, accA);
                        }
                        stage_done();
                        // ---- E(c): + identity (RESB, landed with this stage's barrier), ReLU, 16 bit -> OUTC: the block output's chunk (the
                        //      loaders copy it out) and the next conv1's K-slice
                        {
                            u32x4 r[NQ];
#pragma unroll
                            for (int j = 0; j < NQ; ++j) r[j] = *reinterpret_cast<const u32x4*>(smem + RESB_OFF + ((pba0 + 2048 * j) ^ (wave_a << 6)));
#pragma unroll
                            for (int j = 0; j < NQ; ++j) {
                                f32x4 lo = accA[0][j], hi = accA[1][j];
                                add_identity(lo, hi, r[j]);
                                *reinterpret_cast<u32x4*>(smem + OUTC_OFF + ((pba0 + 2048 * j) ^ (wave_a << 6))) = pack_relu(lo, hi);
                            }
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();           // OUTC(c) complete
                        // ---- B(c): W1[:, 64 c ..] . out_c into the next conv1's accumulators
                        {
                            const char* bb = smem + OUTC_OFF;
                            gemm64(NRWc, smem + RING_OFF + cbuf * WSTAGE + w_row, [&](int t) { return bb + ((pb0 + 2048 * (t % NRW)) ^ (t >= NRW ? 64 : 0)); }, acc);
                            stage_done();
                        }
                    }
                    const unsigned pix0 = tile_pix0 + (unsigned)p0;
#pragma unroll
                    for (int j = 0; j < NRW; ++j) {
                        const u32x4 o = pack_relu(acc[0][j], acc[1][j]);
                        const unsigned voff = (p0 + 16 * j < NPX) ? ((pix0 + 16 * j) * 128u + (unsigned)ch_lane) * 2u : kOobOffset;
                        __builtin_amdgcn_raw_buffer_store_b128(o, rs_y1, voff, 0, 0);
                    }
                }
            }
        };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's part of the biases is in LDS ...
        __builtin_amdgcn_s_barrier();                           // ... and so is everybody's; band + stage 0 landed
        if (wave_p == 0) {
            if (wave_q == 0) run(std::integral_constant<int, 7>{}, std::integral_constant<int, 4>{});
            else run(std::integral_constant<int, 7>{}, std::integral_constant<int, 3>{});
        } else {
            run(std::integral_constant<int, 6>{}, std::integral_constant<int, 3>{});
        }
    }
#else
    (void)a;
#endif
}

// ------------------------------------------------------------------------------------------------
// Bottleneck BODY (layer1.1 / layer1.2), one launch per block: conv2 3x3 (64 -> 64) + bn2 + ReLU, conv3 1x1 (64 -> 256) + bn3 +
// identity + ReLU and the NEXT block's conv1 1x1 (256 -> C1N: 64, or 128 = layer2.0.conv1) + bn1 + ReLU.  Same idea as
// bneck_block2_kernel: t2 never leaves the CU (1.03 GB of HBM traffic per block at batch 256 instead of 1.23) and one launch
// instead of two.  At 56x56 the block is bound by what a CU can stream (1.3 KB per pixel against 0.14 MFLOP), so what matters is that the
// loaders keep identity rows, output rows and the next band moving while the consumers run the three small GEMMs beside them.
// Tile = a band of 4 rows x 56 columns of one image = 224 pixels = 14 MFMA column blocks (14 tiles per image).  Every GEMM here is
// 64 rows wide: 8 consumer waves = 2 cout groups (32 rows) x 4 pixel quarters (4 + 4 + 3 + 3 blocks), 4 loader waves.
// Stream of 8-KB weight stages [64 rows][64 K] through a ring of three: conv2: 9 (taps; the band, with halo, is resident);
// per 64-channel chunk c of the block output: A(c): W3[64 c ..] against T2; E(c): + identity, ReLU -> OUTC;
// B(c): W1[.., 64 c ..] against OUTC (C1N = 128: two stages, cout halves); then next t1 = relu(accB).
// Summation orders are those of conv3x3_c64_kernel (taps ascending, K halves inside) and of the igemm launches (bias first, K
// ascending, identity last): bit-identical outputs.
// LDS: XB 45,056 (348 padded positions x 128 B) | T2 28,672 | OUTC 28,672 | RESB 28,672 | ring 3 x 8,192 | biases 1,792 = 157,440 B.
// ------------------------------------------------------------------------------------------------
struct Block1Args {
    const __bf16* t1;     // (N,56,56,64)   this block's conv1 output
    const __bf16* w2;     // (64, 3, 3, 64) folded conv2 weights, K-major (tap, cin)
    const float* b2;      // (64)
    const __bf16* w3;     // (256, 64)
    const float* b3;      // (256)
    const __bf16* res;    // (M, 256)  identity;  DS: (M, 64) the block INPUT, the identity is bf16(wd . input + bd) computed in the kernel
    const __bf16* wd;     // DS: (256, 64) folded downsample weights
    const float* bd;      // DS: (256)
    __bf16* out;          // (M, 256)  block output
    const __bf16* w1;     // (C1N, 256) next conv1
    const float* b1;      // (C1N)
    __bf16* y1n;          // (M, C1N)
    int N;
    int n_tiles;          // 14 N
    int out_sub;          // 1: only the block output's even rows and columns are stored, as a compact (N,28,28,256) tensor -- layer1.2, whose
                          //    output has two readers: layer2.0.conv1 (computed here, from LDS) and layer2.0's stride-2 downsample conv
};

template <int ET, int C1N, bool DS = false>
__global__ __launch_bounds__(768) void bneck_block1_kernel(const Block1Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(C1N == 64 || C1N == 128, "next conv1: 256 -> 64 or 256 -> 128");
    constexpr int IW = 56, TR = 4, PW = IW + 2, PP = (TR + 2) * PW;       // 348 padded positions
    constexpr int XPASS = (PP + 31) / 32, XBUF = XPASS * 32 * 128;        // 11 passes, 45,056 B
    constexpr int NPX = TR * IW;                                          // 224 = 7 passes of 32 rows
    constexpr int SLOT = NPX * 128;                                       // 28,672
    constexpr int XB_OFF = 0, T2_OFF = XBUF, OUTC_OFF = T2_OFF + SLOT, RESB_OFF = OUTC_OFF + SLOT, RING_OFF = RESB_OFF + SLOT;
    constexpr int NST = 3, WSTAGE = 64 * 128, WPASS = 2;
    constexpr int BIAS_OFF = RING_OFF + NST * WSTAGE;                     // b2 (64) | b3 (256) | b1 (C1N) floats
    constexpr int NB1 = C1N / 64;                                         // B stages per chunk
    // DS (layer1.0, round 3): the identity is the downsample conv of the block input -- one more 8-KB weight stage per chunk (Wd[64 c ..]) against the
    // tile's 224 input rows, which take the place of the identity rows in RESB (fetched once per tile instead of once per chunk), into a second
    // accumulator set that is rounded to 16 bits exactly as the separate downsample launch stores it and then added like an identity
    constexpr int NA = DS ? 2 : 1;                                        // A stages per chunk
    constexpr int NCONV = 9, LCH = NA + NB1, SPT = NCONV + 4 * LCH;       // 17 / 21 stages per tile (DS, C1N = 64: 21)
    constexpr int BD_IDX = 64 + 256 + C1N;                                // bd (256) behind b1
    static_assert(BIAS_OFF + (448 + 256) * 4 <= 163840, "LDS map");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grid = gridDim.x, first = blockIdx.x;

    if (tid < 64 + 256 + C1N) {
        float v;
        if (tid < 64) v = a.b2[tid];
        else if (tid < 320) v = a.b3[tid - 64];
        else v = a.b1[tid - 320];
        reinterpret_cast<float*>(smem + BIAS_OFF)[tid] = v;
    }
    if (DS && tid >= 512) reinterpret_cast<float*>(smem + BIAS_OFF)[BD_IDX + tid - 512] = a.bd[tid - 512];

    if (wave >= 8) {
        // =============================== loader waves ===============================================
        const int lw = wave - 8, lt = tid - 512;
        const int srow = lt >> 3, slot = lt & 7;
        const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w2), 0, 64u * 576u * 2u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w3), 0, 256u * 64u * 2u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w1), 0, (unsigned)C1N * 256u * 2u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.t1), 0, (unsigned)a.N * (3136u * 128u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.res), 0, (unsigned)a.N * (3136u * (DS ? 128u : 512u)), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_wd = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(DS ? a.wd : a.w3), 0, 256u * 64u * 2u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)a.N * (3136u * 512u), 0x00020000);
        unsigned w2v[WPASS], w3v[WPASS], w1v[WPASS], x_voff[XPASS], rv[7], rvx[7], rvs[7];
#pragma unroll
        for (int i = 0; i < WPASS; ++i) {
            const int rho = i * 32 + srow;                               // LDS row of the stage (< 64)
            const int cl = (rho & ~31) | (rho & 3) | (((rho >> 4) & 1) << 2) | (((rho >> 2) & 3) << 3);
            const unsigned ch = (unsigned)((slot ^ (rho & 7)) * 8);
            w2v[i] = ((unsigned)cl * 576u + ch) * 2u;
            w3v[i] = ((unsigned)cl * 64u + ch) * 2u;
            w1v[i] = ((unsigned)cl * 256u + ch) * 2u;
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) {             // rows of a 64-channel chunk of the identity / block output: row R = 32 i + srow of the tile
            const int R = 32 * i + srow;
            rv[i] = ((unsigned)R * 256u + (unsigned)((slot ^ (R & 7)) * 8)) * 2u;
            rvx[i] = ((unsigned)R * 64u + (unsigned)((slot ^ (R & 7)) * 8)) * 2u;          // DS: the same rows of the 64-channel block input
            // out_sub: pixel R = (tile row R / 56, column R % 56) is kept if both are even; the tile's 2 x 28 kept pixels are rows 2 tr, 2 tr + 1
            // of the compact tensor (its pixel index of the tile's first pixel = tile_pix0 / 4); the others are dropped by the range check
            const int sr = R / 56, sc = R - 56 * sr;
            rvs[i] = (a.out_sub && !((sr | sc) & 1)) ? ((unsigned)((sr >> 1) * 28 + (sc >> 1)) * 256u + (unsigned)((slot ^ (R & 7)) * 8)) * 2u : kOobOffset;
        }
        auto decode_band = [&](int tile) {        // source offsets of the padded positions of `tile` (out of range = zero border)
            const int n = tile / 14, tr = tile - n * 14;
#pragma unroll
            for (int i = 0; i < XPASS; ++i) {
                const int q = i * 32 + srow;
                const int rr = q / PW, cc = q - rr * PW;
                const int y = tr * TR + rr - 1, x = cc - 1;
                const bool ok = q < PP && tile < a.n_tiles && (unsigned)y < 56u && (unsigned)x < 56u;
                x_voff[i] = ok ? (unsigned)(((n * 56 + y) * 56 + x) * 64 + (slot ^ ((rr * IW + cc) & 7)) * 8) * 2u : kOobOffset;      // (chunk key: see bneck_block2_kernel)
            }
        };
        auto issue_band = [&](int i0, int i1) {   // passes [i0, i1) of the band decoded last
#pragma unroll
            for (int i = 0; i < XPASS; ++i)
                if (i >= i0 && i < i1)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (LDS_AS void*)(smem + XB_OFF + i * 4096 + lw * 1024), 16, x_voff[i], 0, 0, 0);
        };
        auto issue_res = [&](unsigned tile_pix0, int c) {       // 7 DMAs per wave: identity rows of chunk c (DS: the tile's block-input rows)
            const int sofs = __builtin_amdgcn_readfirstlane(DS ? (int)(tile_pix0 * 128u) : (int)(tile_pix0 * 512u) + c * 128);
#pragma unroll
            for (int i = 0; i < 7; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_res, (LDS_AS void*)(smem + RESB_OFF + i * 4096 + lw * 1024), 16, DS ? rvx[i] : rv[i], sofs, 0, 0);
        };
        auto copy_out = [&](unsigned tile_pix0, int c) {        // 7 LDS reads + 7 stores per wave: out_c -> block output
            const int sofs = __builtin_amdgcn_readfirstlane((int)((a.out_sub ? tile_pix0 >> 2 : tile_pix0) * 512u) + c * 128);
            u32x4 v[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) v[i] = *reinterpret_cast<const u32x4*>(smem + OUTC_OFF + i * 4096 + lt * 16);
#pragma unroll
            for (int i = 0; i < 7; ++i) __builtin_amdgcn_raw_buffer_store_b128(v[i], rs_out, a.out_sub ? rvs[i] : rv[i], sofs, 0);
        };
        int ring = 0;                             // ring slot of the next stage to issue
        auto stage_issue = [&](int p) {           // stage p (0 .. SPT-1) of a tile; 2 DMAs per wave
            char* sbase = smem + RING_OFF + ring * WSTAGE + lw * 1024;
            ring = (ring == NST - 1) ? 0 : ring + 1;
            if (p < NCONV) {
                const int sofs = __builtin_amdgcn_readfirstlane(p * 64 * 2);
#pragma unroll
                for (int i = 0; i < WPASS; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w2, (LDS_AS void*)(sbase + i * 4096), 16, w2v[i], sofs, 0, 0);
            } else {
                const int q = p - NCONV, c = q / LCH, r = q - c * LCH;
                if (r < NA) {                     // W3[64 c ..], then (DS) Wd[64 c ..]: same (256, 64) layout
                    const int sofs = __builtin_amdgcn_readfirstlane(c * 64 * 64 * 2);
#pragma unroll
                    for (int i = 0; i < WPASS; ++i)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r == 0 ? rs_w3 : rs_wd, (LDS_AS void*)(sbase + i * 4096), 16, w3v[i], sofs, 0, 0);
                } else {
                    const int sofs = __builtin_amdgcn_readfirstlane(((r - NA) * 64 * 256 + c * 64) * 2);
#pragma unroll
                    for (int i = 0; i < WPASS; ++i)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (LDS_AS void*)(sbase + i * 4096), 16, w1v[i], sofs, 0, 0);
                }
            }
        };
        auto wait_younger = [&](int n) { wait_vmcnt(n); };          // all but the n youngest vector-memory operations of this wave are complete
        decode_band(first);
        issue_band(0, XPASS);
        stage_issue(0);
        stage_issue(1);
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");       // band and stage 0 landed (stage 1 may be in flight); bias writes done
        __builtin_amdgcn_s_barrier();
        int carry = 0;                            // stores of the previous iteration (issued behind its DMAs)
#pragma unroll 1
        for (int tile = first; tile < a.n_tiles; tile += grid) {
            const bool has_next = tile + grid < a.n_tiles;
            const unsigned tile_pix0 = (unsigned)((tile / 14) * 3136 + (tile % 14) * NPX);
            B2_LOADER_LOOP                                            // (round 3: unrolled, as bneck_block2_kernel's)
            for (int p = 0; p < SPT; ++p) {
                const int q = p - NCONV, c = q >= 0 ? q / LCH : -1, r = q >= 0 ? q - c * LCH : -1;
                // the consumers' extra barriers: T2 complete (in front of the first A stage), OUTC(c) complete (in front of the first B stage)
                if (p == NCONV || r == NA) __builtin_amdgcn_s_barrier();
                int younger = carry;
                carry = 0;
                // identity rows: chunk 0 two stages before the first A stage (RESB is free since the previous tile's last E), chunk c + 1
                // right behind E(c) -- they have to be in LDS one barrier later (A(c + 1)), or two (C1N = 128)
                if (p == NCONV - 2) { issue_res(tile_pix0, 0); younger += 7; }
                if (!DS && r == 1 && c < 3) { issue_res(tile_pix0, c + 1); younger += 7; }
                // the next tile's band: its buffer is dead once conv2 is done (behind the T2 barrier); spread over the tail's first positions
                if (has_next && p >= NCONV && p < NCONV + 4) {
                    if (p == NCONV) decode_band(tile + grid);
                    const int i0 = 3 * (p - NCONV), i1 = (p == NCONV + 3) ? XPASS : i0 + 3;
                    issue_band(i0, i1);
                    younger += i1 - i0;
                }
                const bool st = (p + 2 < SPT) || has_next;
                if (st) { stage_issue(p + 2 < SPT ? p + 2 : p + 2 - SPT); younger += WPASS; }
                // out_c(c), complete behind the OUTC barrier, is intact until E(c + 1): copied out one position later (C1N = 64: the A(c + 1)
                // position; C1N = 128: the second B position), chunk 3 at the tile's last position
                const bool cpos = (C1N == 64) ? ((r == 0 && c >= 1) || p == SPT - 1) : (r == NA + 1);
                if (cpos) {
                    copy_out(tile_pix0, (C1N == 64 && p != SPT - 1) ? c - 1 : c);
                    younger += 7;
                    carry = 7;
                }
                wait_younger(younger);            // stage p + 1 and everything older (identity rows, band passes) landed
                __builtin_amdgcn_s_barrier();
            }
        }
    } else {
        // =============================== consumer waves =============================================
        const int wave_a = wave & 1, wave_q = wave >> 1;                 // cout group (32 of 64), pixel quarter: blocks 0-3, 4-7, 8-10, 11-13
        const int qb0 = wave_q < 2 ? 4 * wave_q : 8 + 3 * (wave_q - 2);
        const int fr = lane & 15, fq = lane >> 4;
        const __amdgpu_buffer_rsrc_t rs_y1 = __builtin_amdgcn_make_buffer_rsrc(a.y1n, 0, (unsigned)a.N * (3136u * C1N * 2u), 0x00020000);
        const int w_row = (wave_a * 32 + fr) * 128;                       // + m * 2048
        const int w_ph0 = (fq ^ (fr & 7)) << 4;                          // kk = 0; kk = 1 is ^ 64
        const int ch_lane = wave_a * 32 + 8 * fq;                        // this lane's 8 consecutive channels of a 64-channel group
        // pixel row p = p0 + 16 j of block j: B fragment (kk = 0) at pb0 + 2048 j of a K-slot (p & 7 does not depend on j); kk = 1 is ^ 64;
        // this lane's 8 channels of that row (as the B operand of the NEXT GEMM, or the identity's): (pb0 + 2048 j) ^ cf_x
        const int p0 = 16 * qb0 + fr, pb0 = p0 * 128 + ((fq ^ (p0 & 7)) << 4), cf_x = wave_a << 6;

        auto run = [&](auto nq_c) {
            constexpr int NQ = decltype(nq_c)::value;
            f32x4 acc[2][NQ], accB[NB1][2][NQ], accD[2][DS ? NQ : 1];
            int cbuf = 0;
            auto gemm64 = [&](const char* wb, auto xaddr, auto& ac) {
                constexpr int NS = 2 * NQ, PD = 3;
                bf16x8 x[NS], wf[2], wg[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) wf[m] = *reinterpret_cast<const bf16x8*>(wb + m * 2048 + w_ph0);
#pragma unroll
                for (int m = 0; m < 2; ++m) wg[m] = *reinterpret_cast<const bf16x8*>(wb + m * 2048 + (w_ph0 ^ 64));
#pragma unroll
                for (int t = 0; t < PD; ++t) x[t] = *reinterpret_cast<const bf16x8*>(xaddr(t));
#pragma unroll
                for (int t = 0; t < NS; ++t) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) ac[m][t % NQ] = mfma_e<ET>(t >= NQ ? wg[m] : wf[m], x[t], ac[m][t % NQ]);
                    if (t + PD < NS) x[t + PD] = *reinterpret_cast<const bf16x8*>(xaddr(t + PD));
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 4 + PD, 0);
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    if (t + PD < NS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            auto stage_done = [&]() {             // the barrier stays behind the stage's last fragment read
                cbuf = (cbuf == NST - 1) ? 0 : cbuf + 1;
                __builtin_amdgcn_s_barrier();
            };
            auto pack_relu = [&](const f32x4& lo, const f32x4& hi) {
                u32x4 o = (u32x4){pack2_e<ET>(lo[0], lo[1]), pack2_e<ET>(lo[2], lo[3]), pack2_e<ET>(hi[0], hi[1]), pack2_e<ET>(hi[2], hi[3])};
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = relu_bf16x2(o[e]);
                return o;
            };
            auto set_bias = [&](f32x4 (&ac)[2][NQ], int fidx) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + (fidx + ch_lane) * 4);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + (fidx + ch_lane + 4) * 4);
#pragma unroll
                for (int j = 0; j < NQ; ++j) { ac[0][j] = lo; ac[1][j] = hi; }
            };
#pragma unroll 1
            for (int tile = first; tile < a.n_tiles; tile += grid) {
                const unsigned pix0 = (unsigned)((tile / 14) * 3136 + (tile % 14) * NPX + p0);       // + 16 j
                // ---- conv2: bias, then 9 stages (taps) against the resident band
                set_bias(acc, 0);
                {
                    int opq = 0;                  // (keeps the per-tile position table out of the registers that live across the tile loop)
                    asm volatile("" : "+v"(opq));
                    int q0[NQ];
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        const int p = p0 + 16 * j + opq;
                        const int r = p / IW, c = p - r * IW;
                        q0[j] = r * PW + c;
                    }
#pragma unroll 1
                    for (int tap = 0; tap < 9; ++tap) {
                        const int kh = (tap >= 6) ? 2 : (tap >= 3) ? 1 : 0, kw = tap - 3 * kh;
                        const int sw = (kh * PW + kw) * 128 + ((fq ^ ((p0 + kh * IW + kw) & 7)) << 4);      // tap offset + the tap's chunk key
                        gemm64(smem + RING_OFF + cbuf * WSTAGE + w_row, [&](int t) {
                            return smem + XB_OFF + q0[t % NQ] * 128 + (t >= NQ ? (sw ^ 64) : sw);
                        }, acc);
                        stage_done();
                    }
                }
                // ---- t2 = relu(acc) -> T2
#pragma unroll
                for (int j = 0; j < NQ; ++j) *reinterpret_cast<u32x4*>(smem + T2_OFF + ((pb0 + 2048 * j) ^ cf_x)) = pack_relu(acc[0][j], acc[1][j]);
#pragma unroll
                for (int h = 0; h < NB1; ++h) set_bias(accB[h], 320 + 64 * h);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                   // T2 complete
#pragma unroll 1
                for (int c = 0; c < 4; ++c) {
                    set_bias(acc, 64 + 64 * c);
                    // ---- A(c): W3[64 c ..] . t2
                    gemm64(smem + RING_OFF + cbuf * WSTAGE + w_row, [&](int t) { return smem + T2_OFF + ((pb0 + 2048 * (t % NQ)) ^ (t >= NQ ? 64 : 0)); }, acc);
                    stage_done();
                    if constexpr (DS) {           // ---- the identity's chunk: Wd[64 c ..] . block input (the tile's rows, in RESB) from bias bd
                        set_bias(accD, BD_IDX + 64 * c);
                        gemm64(smem + RING_OFF + cbuf * WSTAGE + w_row, [&](int t) { return smem + RESB_OFF + ((pb0 + 2048 * (t % NQ)) ^ (t >= NQ ? 64 : 0)); }, accD);
                        stage_done();
                    }
                    // ---- E(c): + identity (RESB, landed with this stage's barrier), ReLU, 16 bit -> OUTC: the block output's chunk (the loaders
                    //      copy it out) and the next conv1's K-slice
                    {
                        u32x4 rr[NQ];
#pragma unroll
                        for (int j = 0; j < NQ; ++j) {
                            if constexpr (DS) rr[j] = (u32x4){pack2_e<ET>(accD[0][j][0], accD[0][j][1]), pack2_e<ET>(accD[0][j][2], accD[0][j][3]),
                                                              pack2_e<ET>(accD[1][j][0], accD[1][j][1]), pack2_e<ET>(accD[1][j][2], accD[1][j][3])};
                            else rr[j] = *reinterpret_cast<const u32x4*>(smem + RESB_OFF + ((pb0 + 2048 * j) ^ cf_x));
                        }
#pragma unroll
                        for (int j = 0; j < NQ; ++j) {
                            f32x4 lo = acc[0][j], hi = acc[1][j];
                            lo[0] += unpack_lo_e<ET>(rr[j][0]); lo[1] += unpack_hi_e<ET>(rr[j][0]);
                            lo[2] += unpack_lo_e<ET>(rr[j][1]); lo[3] += unpack_hi_e<ET>(rr[j][1]);
                            hi[0] += unpack_lo_e<ET>(rr[j][2]); hi[1] += unpack_hi_e<ET>(rr[j][2]);
                            hi[2] += unpack_lo_e<ET>(rr[j][3]); hi[3] += unpack_hi_e<ET>(rr[j][3]);
                            *reinterpret_cast<u32x4*>(smem + OUTC_OFF + ((pb0 + 2048 * j) ^ cf_x)) = pack_relu(lo, hi);
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();               // OUTC(c) complete
                    // ---- B(c): W1[64 h .., 64 c ..] . out_c into the next conv1's accumulators
#pragma unroll
                    for (int h = 0; h < NB1; ++h) {
                        gemm64(smem + RING_OFF + cbuf * WSTAGE + w_row, [&](int t) { return smem + OUTC_OFF + ((pb0 + 2048 * (t % NQ)) ^ (t >= NQ ? 64 : 0)); }, accB[h]);
                        stage_done();
                    }
                }
#pragma unroll
                for (int h = 0; h < NB1; ++h)
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        const u32x4 o = pack_relu(accB[h][0][j], accB[h][1][j]);
                        __builtin_amdgcn_raw_buffer_store_b128(o, rs_y1, ((pix0 + 16 * j) * (unsigned)C1N + (unsigned)(64 * h + ch_lane)) * 2u, 0, 0);
                    }
            }
        };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's part of the biases is in LDS ...
        __builtin_amdgcn_s_barrier();                           // ... and so is everybody's; band + stage 0 landed
        if (wave_q < 2) run(std::integral_constant<int, 4>{});
        else run(std::integral_constant<int, 3>{});
    }
#else
    (void)a;
#endif
}

// ------------------------------------------------------------------------------------------------
// Frame producer: crop box + bilinear resize of decoded uint8 frames, on the device.
// Reference: _crop_and_resize_video_uint8 (src/dataset.py:141-149) = slice [top:top+hh, left:left+ww] of the
// (T,H,W,3) clip, then torchvision resize(..., [224,224], antialias=False) on uint8, which on an AVX2 CPU is ATen's
// native uint8 bilinear kernel: separable, horizontal pass first, int16 fixed-point weights, uint8 intermediate
// (oracle/resize_oracle.py restates it and is pinned against torch itself).  Same integer arithmetic here, both
// passes in one thread: 2x2 source pixels -> two horizontally resampled uint8 values -> one output value.
// The reference itself imports the v1 API (`torchvision.transforms.functional.resize`), which converts uint8 to fp32,
// interpolates and rounds: `float_mode` does that (fp32 lambdas from the host, round half to even).  The CPU kernel's
// FMA contraction is a property of the torch build, so that mode matches torch here on all but ~5e-6 of the bytes
// (1 LSB at rounding ties); the fixed-point mode is bit-exact.
// Per output column / row: first source index and two tap weights (a border tap has w1 = 0), computed in the kernel;
// only the fixed-point precisions (a maximum over all weights of an axis) come from the host.  Fully asynchronous on
// the stream.  Output is NCHW uint8, the layout r50_forward_u8 takes.
// ------------------------------------------------------------------------------------------------
struct ResizeArgs {
    const unsigned char* src;   // (T, H, W, 3)
    unsigned char* dst;         // (T, 3, out, out)
    int T, H, W, top, left, hh, ww, out;
    int px, py;                 // weight precisions (bits) of the horizontal / vertical pass (fixed-point mode; host-computed)
    int float_mode;             // 1: fp32 arithmetic + round-half-even (torchvision v1 `functional.resize` on uint8)
    int hflip, trev;            // augmentation variants as index permutations of the OUTPUT: mirror columns / reverse frames
};

// Source index and the two tap weights of output index i along one axis, computed per thread with the very
// operations the host-side restatement uses (IEEE, no contraction: the library is built with -ffp-contract=off).
// Fixed-point mode: ATen `_compute_indices_min_size_weights` in double, scaled to `prec` bits; float mode:
// `area_pixel_compute_source_index` / `guard_index_and_lambda` in fp32 with the index expression as one fma.
__device__ __forceinline__ void resize_taps(int i, int in_size, int out_size, int prec, bool float_mode, int& i0, int& w0, int& w1) {
    if (float_mode) {
        float l0 = 1.0f, l1 = 0.0f;
        int idx = i;
        if (in_size != out_size) {
            const float scale = (float)in_size / (float)out_size;
            float real = __fmaf_rn(scale, (float)i + 0.5f, -0.5f);
            if (real < 0.0f) real = 0.0f;
            idx = (int)floorf(real);
            if (idx > in_size - 1) idx = in_size - 1;
            l1 = real - (float)idx;
            l1 = l1 < 0.0f ? 0.0f : (l1 > 1.0f ? 1.0f : l1);
            l0 = 1.0f - l1;
        }
        i0 = idx; w0 = __float_as_int(l0); w1 = __float_as_int(l1);
    } else {
        const double scale = (double)in_size / (double)out_size;
        double real = scale * ((double)i + 0.5) - 0.5;
        if (real < 0.0) real = 0.0;
        long idx = (long)floor(real);
        if (idx > in_size - 1) idx = in_size - 1;
        double lam = real - (double)idx;
        lam = lam < 0.0 ? 0.0 : (lam > 1.0 ? 1.0 : lam);
        const long umin = idx, umax = idx + 2;                      // support = 1
        const long lo = umin > 0 ? umin : 0;
        const long size = (umax < in_size ? umax : in_size) - lo;
        double w[2] = {0.0, 0.0};
        long w_index = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double x = fabs((double)j - lam);
            const double wj = x < 1.0 ? 1.0 - x : 0.0;
            if (umin + j <= 0) w_index = 0;
            else if (umin + j >= in_size - 1) w_index = size - 1;
            if (w_index == 0) w[0] += wj; else w[1] += wj;
            ++w_index;
        }
        i0 = (int)lo;
        w0 = (int)(0.5 + w[0] * (double)(1 << prec));
        w1 = (int)(0.5 + w[1] * (double)(1 << prec));
    }
}

// One workgroup = one output row of one frame.  The two source rows it needs (the crop's width, 3 bytes per pixel) are
// copied to LDS with coalesced 4-byte loads (a thread reading its 12 tap bytes straight from global memory ran at
// 1.3 TB/s of useful traffic); then 56 threads x 3 channels... every thread produces 4 consecutive output pixels of one
// channel plane from LDS bytes and stores them as one dword.
__global__ __launch_bounds__(256) void crop_resize_u8_kernel(const ResizeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
    const int quads = a.out >> 2;
    const int t = blockIdx.x / a.out, yo = blockIdx.x - t * a.out;
    int y0, wy0, wy1;
    resize_taps(yo, a.hh, a.out, a.py, a.float_mode != 0, y0, wy0, wy1);
    const int y1 = min(y0 + 1, a.hh - 1);
    // _aug_temporal_reverse = torch.flip(video, dims=[0]), _aug_hflip = torch.flip(video, dims=[-1]) of the resized clip
    // (src/dataset.py:158-166,199-207): output frame t / column x takes what the plain resize puts at T-1-t / out-1-x
    const unsigned char* f = a.src + (size_t)(a.trev ? a.T - 1 - t : t) * a.H * a.W * 3;
    const size_t g0 = ((size_t)(a.top + y0) * a.W + a.left) * 3, g1 = ((size_t)(a.top + y1) * a.W + a.left) * 3;
    const int row_bytes = a.ww * 3;
    const int row_pad = (row_bytes + 3 + 3) & ~3;                     // room for the leading misalignment
    // copy [g & ~3, g + row_bytes) of both rows as dwords; sh = g & 3 is where the crop starts inside the copy
    const int sh0 = (int)(g0 & 3), sh1 = (int)(g1 & 3);
    const unsigned* s0 = reinterpret_cast<const unsigned*>(f + (g0 - sh0));
    const unsigned* s1 = reinterpret_cast<const unsigned*>(f + (g1 - sh1));
    const int nd0 = (sh0 + row_bytes + 3) >> 2, nd1 = (sh1 + row_bytes + 3) >> 2;
    unsigned* l0 = reinterpret_cast<unsigned*>(rs_smem);
    unsigned* l1 = reinterpret_cast<unsigned*>(rs_smem + row_pad);
    // the last dword of a copy may reach up to 3 bytes past the row: it stays inside the frame buffer except at the very end
    // of the tensor, where the tail is read byte by byte
    const unsigned char* tensor_end = a.src + (size_t)a.T * a.H * a.W * 3;
    for (int i = threadIdx.x; i < nd0; i += 256) {
        const unsigned char* p = reinterpret_cast<const unsigned char*>(s0 + i);
        l0[i] = (p + 4 <= tensor_end) ? s0[i] : (unsigned)p[0] | ((p + 1 < tensor_end ? (unsigned)p[1] : 0u) << 8) | ((p + 2 < tensor_end ? (unsigned)p[2] : 0u) << 16);
    }
    for (int i = threadIdx.x; i < nd1; i += 256) {
        const unsigned char* p = reinterpret_cast<const unsigned char*>(s1 + i);
        l1[i] = (p + 4 <= tensor_end) ? s1[i] : (unsigned)p[0] | ((p + 1 < tensor_end ? (unsigned)p[1] : 0u) << 8) | ((p + 2 < tensor_end ? (unsigned)p[2] : 0u) << 16);
    }
    __syncthreads();
    const unsigned char* row0 = rs_smem + sh0;
    const unsigned char* row1 = rs_smem + row_pad + sh1;
    const int rx = 1 << (a.px - 1), ry = 1 << (a.py - 1);
    for (int item = threadIdx.x; item < quads * 3; item += 256) {
        const int c = item / quads, xq = item - c * quads;
        unsigned packed = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int xo = xq * 4 + q;
            int x0, wx0, wx1;
            resize_taps(a.hflip ? a.out - 1 - xo : xo, a.ww, a.out, a.px, a.float_mode != 0, x0, wx0, wx1);
            const int x1 = min(x0 + 1, a.ww - 1);                       // float mode: same second index (index0 + (index0 < size-1))
            int v;
            if (a.float_mode) {
                // ATen cpu_upsample_linear_channels_last: out = p00*w00 + p01*w01 + p10*w10 + p11*w11 with w_ij = h_i * w_j
                // (fp32; the CPU build contracts some of it into FMAs -- this association is the closest match found)
                const float lx0 = __int_as_float(wx0), lx1 = __int_as_float(wx1), ly0 = __int_as_float(wy0), ly1 = __int_as_float(wy1);
                const float w00 = ly0 * lx0, w01 = ly0 * lx1, w10 = ly1 * lx0, w11 = ly1 * lx1;
                const float p00 = (float)row0[x0 * 3 + c], p01 = (float)row0[x1 * 3 + c];
                const float p10 = (float)row1[x0 * 3 + c], p11 = (float)row1[x1 * 3 + c];
                const float s0f = __fmaf_rn(p01, w01, p00 * w00), s1f = __fmaf_rn(p11, w11, p10 * w10);
                v = (int)rintf(s0f + s1f);                                  // torch.round = half to even; then .to(uint8)
            } else {
                int h0 = (wx0 * (int)row0[x0 * 3 + c] + wx1 * (int)row0[x1 * 3 + c] + rx) >> a.px;
                int h1 = (wx0 * (int)row1[x0 * 3 + c] + wx1 * (int)row1[x1 * 3 + c] + rx) >> a.px;
                h0 = min(max(h0, 0), 255); h1 = min(max(h1, 0), 255);        // the uint8 intermediate image
                v = (wy0 * h0 + wy1 * h1 + ry) >> a.py;
            }
            v = min(max(v, 0), 255);
            packed |= (unsigned)v << (8 * q);
        }
        const size_t plane = (size_t)a.out * a.out;
        *reinterpret_cast<unsigned*>(a.dst + ((size_t)t * 3 + c) * plane + (size_t)yo * a.out + xq * 4) = packed;
    }
}

// ------------------------------------------------------------------------------------------------
// Lifting head (forward only): the non-GEMM pieces of PHDFor3DJoints.forward (src/model.py).  Every
// Linear and causal conv1d of the head runs on the igemm kernels as a 1x1 convolution over the B*T "pixels":
// a causal conv1d with kernel 3 and replicate left padding (:20-35) is a GEMM with K = 3*C against the row
// [x(t-2) | x(t-1) | x(t)] (indices clamped at 0), so the producer of its input writes that row directly.
// ------------------------------------------------------------------------------------------------

// fp32 rows (rows, c) -> element rows (rows, cpad), columns c..cpad-1 zero (GEMM K must be a multiple of 64)
template <int ET>
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                        long long rows, int c, int cpad) {
    const long long pairs = rows * (cpad / 2);
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < pairs; i += (long long)gridDim.x * 256) {
        const long long r = i / (cpad / 2);
        const int col = (int)(i - r * (cpad / 2)) * 2;
        const float a = col < c ? src[r * c + col] : 0.f, b = col + 1 < c ? src[r * c + col + 1] : 0.f;
        *reinterpret_cast<unsigned*>(dst + r * cpad + col) = pack2_e<ET>(a, b);
    }
}

// [phi (rows, d) element | y (rows, ny) fp32 | zeros] -> (rows, dp) element: the regressor's input torch.cat([phi, y]) (:113)
template <int ET>
__global__ __launch_bounds__(256) void concat_pad_kernel(const unsigned short* __restrict__ phi, int d, const float* __restrict__ y,
                                                         int ny, unsigned short* __restrict__ dst, long long rows, int dp) {
    const long long total = rows * dp;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / dp;
        const int col = (int)(i - r * dp);
        unsigned short v = 0;
        if (col < d) v = phi[r * d + col];
        else if (col < d + ny) v = (unsigned short)(pack2_e<ET>(y[r * ny + (col - d)], 0.f) & 0xffffu);
        dst[i] = v;
    }
}

// y (rows, ny) fp32 += dy (rows, dp) element, first ny columns: `y = y + dy` of the iterative regressor (:114-115); y stays fp32
template <int ET>
__global__ __launch_bounds__(256) void add_rows_kernel(float* __restrict__ y, int ny, const unsigned short* __restrict__ dy, int dp,
                                                       long long rows) {
    const long long total = rows * ny;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / ny;
        const int col = (int)(i - r * ny);
        y[i] += unpack_lo_e<ET>(dy[r * dp + col]);
    }
}

// GroupNorm(groups) + ReLU over x (B, T, C) [channels contiguous], fp32 statistics over the (C/groups x T) slab of one sample
// and group (biased variance, eps inside the square root: torch.nn.functional.group_norm), then the causal-conv input
// rows out (B*T, 3C): out[(b,t)][k*C + c] = y[b][max(t-2+k, 0)][c].  One workgroup per (sample, group).
template <int ET>
__global__ __launch_bounds__(256) void gn_relu_causal3_kernel(const unsigned short* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, unsigned short* __restrict__ out,
                                                              int T, int C, int groups, float eps) {
    __shared__ float red[2][256];
    const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
    const int cg = C / groups, n = cg * T;
    const unsigned short* xb = x + (size_t)b * T * C + g * cg;
    auto ld = [&](int i) -> float {                       // element i of the slab: (t = i / cg, c = i % cg)
        const int t = i / cg, c = i - t * cg;
        const unsigned u = xb[(size_t)t * C + c];
        return ET == 0 ? bf16_bits_to_f32(u) : (float)__builtin_bit_cast(_Float16, (unsigned short)u);
    };
    float s = 0.f, ss = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { const float v = ld(i); s += v; ss += v * v; }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    const float mean = red[0][0] / (float)n;
    const float var = fmaxf(red[1][0] / (float)n - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    for (int i = threadIdx.x; i < n; i += 256) {
        const int t = i / cg, c = i - t * cg, ch = g * cg + c;
        float v = (ld(i) - mean) * rstd * gamma[ch] + beta[ch];
        v = fmaxf(v, 0.f);
        const unsigned short e = (unsigned short)(pack2_e<ET>(v, 0.f) & 0xffffu);
        // y(t) is tap k of row r = t + 2 - k; row 0 also takes y(0) for k = 0, 1 and row 1 for k = 0 (replicate padding)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int r = t + 2 - k;
            if (r < T) out[((size_t)b * T + r) * 3 * C + k * C + ch] = e;
        }
        if (t == 0) {
            out[((size_t)b * T + 0) * 3 * C + 0 * C + ch] = e;
            out[((size_t)b * T + 0) * 3 * C + 1 * C + ch] = e;
            if (T > 1) out[((size_t)b * T + 1) * 3 * C + 0 * C + ch] = e;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ColorJitter variant (`_aug_color_jitter`, src/dataset.py:188-197 = torchvision.transforms.v2.ColorJitter
// on the float clip in [0,1]): brightness / contrast / saturation / hue in a sampled order, one set of factors per clip.
// Planar fp32 (T,3,H*W) work buffer; every op follows torchvision's v2 float kernels operation by operation
// (transforms/v2/functional/_color.py; restated in oracle/colorjitter_oracle.py).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cj_from_u8_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = (float)src[i] / 255.0f;
}

// r.mul(0.2989).add_(g, alpha=0.587).add_(b, alpha=0.114): ATen's add-with-alpha is a fused multiply-add on the CPU's vector path
__device__ __forceinline__ float cj_gray(float r, float g, float b) { return fmaf(b, 0.114f, fmaf(g, 0.587f, r * 0.2989f)); }
__device__ __forceinline__ float cj_clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
// _blend: image1.mul(ratio).add_(image2, alpha=1 - ratio).clamp_(0, 1)
__device__ __forceinline__ float cj_blend(float a, float b, float ratio) { return cj_clamp01(fmaf(b, 1.0f - ratio, a * ratio)); }

// mean[t] = mean over the frame of the grayscale image (adjust_contrast's blend target).  One workgroup per frame; per-thread sums
// in pixel order, then a fixed-order tree.
__global__ __launch_bounds__(1024) void cj_gray_mean_kernel(const float* __restrict__ img, int hw, float* __restrict__ mean) {
    __shared__ float red[1024];
    const float* f = img + (size_t)blockIdx.x * 3 * hw;
    float s = 0.f;
    for (int i = threadIdx.x; i < hw; i += 1024) s += cj_gray(f[i], f[hw + i], f[2 * hw + i]);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) mean[blockIdx.x] = red[0] / (float)hw;
}

// op: 0 brightness, 1 contrast (mean per frame), 2 saturation, 3 hue; in place on (T,3,hw) fp32
__global__ __launch_bounds__(256) void cj_apply_kernel(float* __restrict__ img, int t, int hw, int op, float factor,
                                                       const float* __restrict__ mean) {
    const long long total = (long long)t * hw;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long fr = i / hw;
        float* p = img + fr * 3 * hw + (i - fr * hw);
        float r = p[0], g = p[hw], b = p[2 * hw];
        if (op == 0) {
            r = cj_clamp01(r * factor); g = cj_clamp01(g * factor); b = cj_clamp01(b * factor);
        } else if (op == 1) {
            const float m = mean[fr];
            r = cj_blend(r, m, factor); g = cj_blend(g, m, factor); b = cj_blend(b, m, factor);
        } else if (op == 2) {
            const float y = cj_gray(r, g, b);
            r = cj_blend(r, y, factor); g = cj_blend(g, y, factor); b = cj_blend(b, y, factor);
        } else {
            // _rgb_to_hsv
            const float maxc = fmaxf(fmaxf(r, g), b), minc = fminf(fminf(r, g), b);
            const bool eqc = maxc == minc;
            const float cr = maxc - minc;
            const float sat = cr / (eqc ? 1.0f : maxc);
            const float div = eqc ? 1.0f : cr;
            const float rc = (maxc - r) / div, gc = (maxc - g) / div, bc = (maxc - b) / div;
            const bool neq_r = maxc != r, eq_g = maxc == g;
            const float hg = (eq_g && neq_r) ? (rc + 2.0f) - bc : 0.f;
            const float hr = !neq_r ? bc - gc : 0.f;
            const float hb = (neq_r && !eq_g) ? (gc + 4.0f) - rc : 0.f;
            float h = (hr + hg) + hb;
            h = fmodf(h * (1.0f / 6.0f) + 1.0f, 1.0f);
            // h.add_(hue_factor).remainder_(1.0)
            h = h + factor;
            h = h - floorf(h);                         // remainder by 1.0 (sign of the divisor)
            if (h >= 1.0f) h = 0.f;                    // -tiny + 1 rounds to 1: remainder's result stays below the divisor
            // _hsv_to_rgb
            const float h6 = h * 6.0f;
            const float fl = floorf(h6);
            const float f = h6 - fl;
            int k = (int)fl % 6; if (k < 0) k += 6;
            const float sxf = sat * f, oms = 1.0f - sat;
            const float q = cj_clamp01((1.0f - sxf) * maxc);
            const float tt = cj_clamp01((sxf + oms) * maxc);
            const float pp = cj_clamp01(oms * maxc);
            const float v = maxc;
            r = k == 0 ? v : k == 1 ? q : k == 2 ? pp : k == 3 ? pp : k == 4 ? tt : v;
            g = k == 0 ? tt : k == 1 ? v : k == 2 ? v : k == 3 ? q : k == 4 ? pp : pp;
            b = k == 0 ? pp : k == 1 ? pp : k == 2 ? tt : k == 3 ? v : k == 4 ? v : q;
        }
        p[0] = r; p[hw] = g; p[2 * hw] = b;
    }
}

// frame_tf = Normalize(mean, std) (src/dataset.py:242-245) in place: (x - mean[c]) / std[c]
__global__ __launch_bounds__(256) void cj_normalize_kernel(float* __restrict__ img, long long planes, int hw, float m0, float m1, float m2,
                                                           float s0, float s1, float s2) {
    const long long total = planes * hw;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)((i / hw) % 3);
        const float m = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        img[i] = (img[i] - m) / sd;
    }
}

// ------------------------------------------------------------------------------------------------
// Lifting head, backward + optimizer (the training step of src/train.py:137-176).  Every matrix product of
// the backward pass (dX = dY W, dW = dY^T X) is again an igemm launch (operands transposed by transpose16_kernel so that
// both are K-contiguous); what follows are the byte-moving pieces around them.
// ------------------------------------------------------------------------------------------------
template <int ET>
__device__ __forceinline__ float ld_e(const unsigned short* p) { return unpack_lo_e<ET>((unsigned)*p); }
template <int ET>
__device__ __forceinline__ unsigned short st_e(float v) { return (unsigned short)(pack2_e<ET>(v, 0.f) & 0xffffu); }

// src (rows, cols) 16-bit -> dst (cols, ld) with dst[c][r] = src[r][c]; ld >= rows, columns rows..ld-1 of dst are not written
__global__ __launch_bounds__(256) void transpose16_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst,
                                                          int rows, int cols, int ld) {
    __shared__ unsigned short tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(size_t)(r0 + r) * cols + c0 + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (c0 + c < cols && r0 + r < rows) dst[(size_t)(c0 + c) * ld + r0 + r] = tile[r][c];
    }
}

// x *= mask * scale (dropout, src/model.py:44,52,98): mask is one byte per element
template <int ET>
__global__ __launch_bounds__(256) void mask_scale_kernel(unsigned short* __restrict__ x, const unsigned char* __restrict__ mask,
                                                         float scale, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        x[i] = mask[i] ? st_e<ET>(ld_e<ET>(x + i) * scale) : (unsigned short)0;
}

// dx = dy * scale * (act > 0), in place: backward of ReLU (scale 1) or of ReLU followed by dropout (act = the dropped-out
// activation, scale = 1/(1-p): it is positive exactly where the ReLU passed AND the mask kept)
template <int ET>
__global__ __launch_bounds__(256) void relu_bwd_kernel(unsigned short* __restrict__ dy, const unsigned short* __restrict__ act,
                                                       float scale, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        dy[i] = ld_e<ET>(act + i) > 0.f ? st_e<ET>(ld_e<ET>(dy + i) * scale) : (unsigned short)0;
}

// out (cols) fp32 [+]= scale * sum over rows of x (rows, ld)[:, :cols]: bias gradients.  A workgroup owns 64 columns; its 16 row
// groups each sum rows g, g+16, ... in order, and the 16 partials are added in a fixed order (deterministic, no atomics).
template <int ET>
__global__ __launch_bounds__(1024) void colsum_kernel(const unsigned short* __restrict__ x, long long rows, int cols, int ld,
                                                      float scale, float* __restrict__ out, int accumulate) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < cols)
        for (long long r = grp; r < rows; r += 16) s += ld_e<ET>(x + r * ld + c);
    part[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += part[g][lane];
        out[c] = (accumulate ? out[c] : 0.f) + t * scale;
    }
}
__global__ __launch_bounds__(64) void colsum_f32_kernel(const float* __restrict__ x, long long rows, int cols, float scale,
                                                        float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (long long r = 0; r < rows; ++r) s += x[r * cols + c];
    out[c] = (accumulate ? out[c] : 0.f) + s * scale;
}

// dst (n) fp32 [+]= scale * src (n) 16-bit: a weight gradient out of the GEMM into the flat fp32 gradient buffer
template <int ET>
__global__ __launch_bounds__(256) void grad_accum_kernel(const unsigned short* __restrict__ src, float scale, float* __restrict__ dst,
                                                         long long n, int accumulate) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        dst[i] = (accumulate ? dst[i] : 0.f) + ld_e<ET>(src + i) * scale;
}

// l3d = mean((y - gt)^2) (src/train.py:161): dy = 2 * (y - gt) / n * loss_scale; loss[0] = the mean, loss[1] = sum over
// joints of |y - gt|_2 / (n/3) = MPJPE (:42-45).  One workgroup (n = B*T*51 is small), fixed summation order.
__global__ __launch_bounds__(256) void mse_loss_grad_kernel(const float* __restrict__ y, const float* __restrict__ gt, long long n,
                                                            float loss_scale, float* __restrict__ dy, float* __restrict__ loss) {
    __shared__ float red[2][256];
    float s = 0.f, e = 0.f;
    for (long long j = threadIdx.x; j < n / 3; j += 256) {
        float d2 = 0.f;
        for (int k = 0; k < 3; ++k) {
            const float d = y[3 * j + k] - gt[3 * j + k];
            dy[3 * j + k] = 2.f * d / (float)n * loss_scale;
            d2 += d * d;
        }
        s += d2; e += sqrtf(d2);
    }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = e;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { loss[0] = red[0][0] / (float)n; loss[1] = red[1][0] / (float)(n / 3); }
}

// Backward of gn_relu_causal3_kernel.  dr (B*T, 3C): gradient of the causal-conv input rows; x (B, T, C): the GroupNorm input
// saved by the forward.  da(s, c) = sum of dr over the (row, tap) pairs that read frame s (replicate padding: frame 0 also
// collects the clamped taps), dy = da * (y > 0), and with xh = (x - mean) * rstd, g = dy * gamma over the (C/groups x T) slab:
//   dx = rstd * (g - mean(g) - xh * mean(g * xh))   [+ add (B, T, C): the residual path's gradient]
//   dgamma_part[b][c] = sum_t dy * xh, dbeta_part[b][c] = sum_t dy        (per sample; summed over b by colsum_f32_kernel)
// One workgroup per (sample, group); statistics recomputed from x exactly as the forward computed them.
template <int ET>
__global__ __launch_bounds__(256) void gn_relu_causal3_bwd_kernel(const unsigned short* __restrict__ dr, const unsigned short* __restrict__ x,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const unsigned short* __restrict__ add, unsigned short* __restrict__ dx,
                                                                  float* __restrict__ dgamma_part, float* __restrict__ dbeta_part,
                                                                  int T, int C, int groups, float eps) {
    __shared__ float red[2][256];
    __shared__ float cacc[2][256];            // per-channel sums (cg <= 256)
    const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
    const int cg = C / groups, n = cg * T;
    const unsigned short* xb = x + (size_t)b * T * C + g * cg;
    auto block_sum2 = [&](float a, float c2, float& oa, float& oc) {
        red[0][threadIdx.x] = a; red[1][threadIdx.x] = c2;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
            __syncthreads();
        }
        oa = red[0][0]; oc = red[1][0];
        __syncthreads();
    };
    float s = 0.f, ss = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { const int t = i / cg, c = i - t * cg; const float v = ld_e<ET>(xb + (size_t)t * C + c); s += v; ss += v * v; }
    float S, SS;
    block_sum2(s, ss, S, SS);
    const float mean = S / (float)n;
    const float var = fmaxf(SS / (float)n - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    auto dy_of = [&](int t, int c, float& xh) -> float {       // dy and xh of slab element (t, c)
        const int ch = g * cg + c;
        xh = (ld_e<ET>(xb + (size_t)t * C + c) - mean) * rstd;
        if (xh * gamma[ch] + beta[ch] <= 0.f) return 0.f;
        const unsigned short* row = dr + (size_t)b * T * 3 * C;
        float da = 0.f;
        for (int k = 0; k < 3; ++k) {
            const int r = t + 2 - k;
            if (r < T) da += ld_e<ET>(row + (size_t)r * 3 * C + k * C + ch);
        }
        if (t == 0) {
            da += ld_e<ET>(row + 0 * C + ch) + ld_e<ET>(row + 1 * C + ch);
            if (T > 1) da += ld_e<ET>(row + (size_t)1 * 3 * C + 0 * C + ch);
        }
        return da;
    };
    if ((int)threadIdx.x < cg) { cacc[0][threadIdx.x] = 0.f; cacc[1][threadIdx.x] = 0.f; }
    __syncthreads();
    float sg = 0.f, sgx = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int t = i / cg, c = i - t * cg;
        float xh;
        const float dy = dy_of(t, c, xh);
        const float gg = dy * gamma[g * cg + c];
        sg += gg; sgx += gg * xh;
    }
    float SG, SGX;
    block_sum2(sg, sgx, SG, SGX);
    const float mg = SG / (float)n, mgx = SGX / (float)n;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int t = i / cg, c = i - t * cg;
        float xh;
        const float dy = dy_of(t, c, xh);
        float v = rstd * (dy * gamma[g * cg + c] - mg - xh * mgx);
        const size_t o = ((size_t)b * T + t) * C + g * cg + c;
        if (add) v += ld_e<ET>(add + o);
        dx[o] = st_e<ET>(v);
    }
    // per-channel parameter gradients, summed over t in order by one thread per channel
    for (int c = threadIdx.x; c < cg; c += 256) {
        float a = 0.f, bb = 0.f;
        for (int t = 0; t < T; ++t) { float xh; const float dy = dy_of(t, c, xh); a += dy * xh; bb += dy; }
        dgamma_part[(size_t)b * C + g * cg + c] = a;
        dbeta_part[(size_t)b * C + g * cg + c] = bb;
    }
}

// found[0] = 1 if any of g (n) is not finite (GradScaler's inf check, src/train.py:172-174)
__global__ __launch_bounds__(256) void check_finite_kernel(const float* __restrict__ g, long long n, int* __restrict__ found) {
    int bad = 0;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const unsigned u = __float_as_uint(g[i]);
        bad |= ((u & 0x7f800000u) == 0x7f800000u);
    }
    if (bad) atomicOr(found, 1);
}

// found[0] = 1 if any 16-bit element of x is inf / nan or, for IEEE half, at the largest finite magnitude: this build's fp32 -> fp16
// conversion SATURATES at +-65504 instead of producing inf, so a saturated value is what an overflow looks like
template <int ET>
__global__ __launch_bounds__(256) void check_overflow16_kernel(const unsigned short* __restrict__ x, long long n, int* __restrict__ found) {
    int bad = 0;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const unsigned u = x[i] & 0x7fffu;
        bad |= ET == 1 ? (u >= 0x7bffu) : ((u & 0x7f80u) == 0x7f80u);
    }
    if (bad) atomicOr(found, 1);
}

// torch.optim.AdamW (src/train.py:389; amsgrad off, maximize off) over flat fp32 buffers, skipped when found[0] != 0:
//   p *= 1 - lr * wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// and the refreshed 16-bit copy of the parameter the GEMMs read.
template <int ET>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                    const float* __restrict__ g, unsigned short* __restrict__ p16, long long n,
                                                    float lr, float b1, float b2, float eps, float wd, float bc1, float sqrt_bc2,
                                                    const int* __restrict__ found) {
    if (found && found[0]) return;
    const float step_size = lr / bc1;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i];
        float pi = p[i] * (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;
        pi -= step_size * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
        p16[i] = st_e<ET>(pi);
    }
}

// ------------------------------------------------------------------------------------------------
// Stem, step 1: fp32 NCHW (N,3,224,224) -> bf16 "NHWC4" with a zero border:
//   xp[n][hp][wp][4], hp = hi + 3 in [0,230), wp = wi + 4 in [0,232); channel 3 = 0.
// One thread per output pixel (8 B).  The border is rewritten every call.
// ------------------------------------------------------------------------------------------------
constexpr int STEM_HP = 230, STEM_WP = 232;

// Frame element -> the fp32 value the reference's loader would hand to the backbone.
//   float frames: already normalised (src/dataset.py:429).
//   uint8 frames: the resized crop BEFORE `frames.to(float32) / 255.0` (src/dataset.py:148-149) and
//   `Normalize(mean, std)` (:242-245) -- the same three fp32 operations, in the same order, so the result is
//   bit-identical to the host path:  ((u8 / 255) - mean[c]) / std[c].
__device__ __forceinline__ float frame_value(const float* p, int /*c*/) { return *p; }
__device__ __forceinline__ float frame_value(const unsigned char* p, int c) {
    const float mean = (c == 0) ? 0.485f : (c == 1) ? 0.456f : 0.406f;
    const float stdv = (c == 0) ? 0.229f : (c == 1) ? 0.224f : 0.225f;
    const float v = (float)(*p) / 255.0f;
    return (v - mean) / stdv;
}


template <int ET, typename TIN>
__global__ __launch_bounds__(256) void stem_pack_kernel(const TIN* __restrict__ x, u32x2* __restrict__ xp, int n_img) {
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
            const TIN* p = x + ((size_t)n * 3 * 224 + hi) * 224 + wi;
            const float c0 = frame_value(p, 0), c1 = frame_value(p + 224 * 224, 1), c2 = frame_value(p + 2 * 224 * 224, 2);
            o[0] = pack2_e<ET>(c0, c1);
            o[1] = pack2_e<ET>(c2, 0.f);
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

template <int ET>
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
                acc[m][j] = mfma_e<ET>(wf[m], xf[j], acc[m][j]);
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
            for (int e = 0; e < 4; ++e) out[e] = pack2_e<ET>(v[2 * e], v[2 * e + 1]);
            *reinterpret_cast<u32x4*>(y + (((size_t)n * 112 + ho) * 112 + wo) * 64 + cout) = out;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(3, stride 2, pad 1), bf16 NHWC.  One thread per (output pixel, 8 channels): nine 16-B
// loads, fp32 max (padding = -inf), one 16-B store.
// ------------------------------------------------------------------------------------------------
template <int ET>
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
                    m[2 * e] = fmaxf(m[2 * e], unpack_lo_e<ET>(v[e]));
                    m[2 * e + 1] = fmaxf(m[2 * e + 1], unpack_hi_e<ET>(v[e]));
                }
            }
        }
        u32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = pack2_e<ET>(m[2 * e], m[2 * e + 1]);
        *reinterpret_cast<u32x4*>(y + (size_t)idx * 8) = out;
    }
}

// ------------------------------------------------------------------------------------------------
// fp8 network mode (R50_PREC_FP8): hand-over of the 16-bit layer1 output to the fp8 stack, and the final pool on fp8 input.
// ------------------------------------------------------------------------------------------------
template <int ET>
__global__ __launch_bounds__(256) void quant_to_fp8_kernel(const unsigned short* __restrict__ src, unsigned* __restrict__ dst, long long n4,
                                                           float inv_scale) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(src + 4 * i);
        dst[i] = pack4_fp8(unpack_lo_e<ET>(v[0]) * inv_scale, unpack_hi_e<ET>(v[0]) * inv_scale, unpack_lo_e<ET>(v[1]) * inv_scale,
                           unpack_hi_e<ET>(v[1]) * inv_scale);
    }
}

// (N, HW, C) fp8 -> (N, C) fp32: rows summed in order in fp32, times scale / HW.  One thread per (n, 8 channels).
__global__ __launch_bounds__(256) void avgpool_fp8_kernel(const unsigned char* __restrict__ x, float* __restrict__ y, int N, int HW, int C,
                                                          float scale_over_hw) {
    const int cg = C >> 3;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * cg) return;
    const int g = idx % cg, n = idx / cg;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    const unsigned char* p = x + (size_t)n * HW * C + g * 8;
    for (int r = 0; r < HW; ++r) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(p + (size_t)r * C);
        s[0] += __builtin_amdgcn_cvt_f32_fp8((int)v[0], 0); s[1] += __builtin_amdgcn_cvt_f32_fp8((int)v[0], 1);
        s[2] += __builtin_amdgcn_cvt_f32_fp8((int)v[0], 2); s[3] += __builtin_amdgcn_cvt_f32_fp8((int)v[0], 3);
        s[4] += __builtin_amdgcn_cvt_f32_fp8((int)v[1], 0); s[5] += __builtin_amdgcn_cvt_f32_fp8((int)v[1], 1);
        s[6] += __builtin_amdgcn_cvt_f32_fp8((int)v[1], 2); s[7] += __builtin_amdgcn_cvt_f32_fp8((int)v[1], 3);
    }
    float* o = y + (size_t)n * C + g * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = s[e] * scale_over_hw;
}

// ------------------------------------------------------------------------------------------------
// Global average pool: (N, HW, C) bf16 -> (N, C) fp32.  One thread per (n, 8 channels); the HW rows
// are summed in order in fp32 and multiplied by 1/HW.
// ------------------------------------------------------------------------------------------------
template <int ET>
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
    int r = 0;
    for (; r + 7 <= HW; r += 7) {                      // seven rows in flight, added in row order (HW = 49 = 7 x 7)
        u32x4 v[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + (size_t)(r + i) * C));
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[2 * e] += unpack_lo_e<ET>(v[i][e]);
                s[2 * e + 1] += unpack_hi_e<ET>(v[i][e]);
            }
    }
    for (; r < HW; ++r) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(p + (size_t)r * C);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s[2 * e] += unpack_lo_e<ET>(v[e]);
            s[2 * e + 1] += unpack_hi_e<ET>(v[e]);
        }
    }
    float* o = y + (size_t)n * C + g * 8;
    *reinterpret_cast<f32x4*>(o) = (f32x4){s[0] * inv_hw, s[1] * inv_hw, s[2] * inv_hw, s[3] * inv_hw};
    *reinterpret_cast<f32x4*>(o + 4) = (f32x4){s[4] * inv_hw, s[5] * inv_hw, s[6] * inv_hw, s[7] * inv_hw};
}

constexpr int SF_TAB_BYTES = 3 * 256 * 4;               // uint8 frames: per-channel table u8 -> normalised fp32

__device__ __forceinline__ unsigned max_bf16x2_nonneg(unsigned a, unsigned b) {   // both operands >= +0: integer order = float order
    typedef __attribute__((ext_vector_type(2))) short s16x2;
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}

// ------------------------------------------------------------------------------------------------
// Fused stem, role-split version (round 4): fp32 / uint8 NCHW frames -> conv1 7x7 s2 + bias + ReLU -> MaxPool2d(3,2,1) -> (N,56,56,64)
// NHWC, and (C1) layer1.0.conv1 of the pooled rows, in one launch; the arithmetic and the output bits of the strip kernel of round 3
// (stem_fused2_kernel: all 8 waves pack, then all run MFMAs, then all pool -- two barriers per pair of pooled rows, the MFMA pipe idle
// during two of the three phases; 149 us at batch 256), but conv and everything else run side by side.  A workgroup walks G
// consecutive pooled-row pairs of ONE image (G = 28: the whole image) and keeps what consecutive pairs share in LDS rings.
// 12 waves: waves 0-7 are the conv waves (conv row rr = wave >> 1 of the pair, channel half ch = wave & 1; two per SIMD), waves 8-11
// the service waves (one per SIMD): they pack the NEXT pair's 8 input rows into the ring (16-B loads issued one pair ahead, one thread
// = 4 pixels x 3 channels), pool the PREVIOUS pair's conv rows, store the pooled rows and run layer1.0.conv1 on the pair before that.
// One barrier per pair.
//   * conv1's weights live in registers (7 kernel rows x 2 channel blocks x 4 registers per conv wave): the K loop reads only the
//     7 image fragments per kernel row from LDS, for 14 MFMAs;
//   * the image fragment is the MFMA's A operand, the weights its B operand: an accumulator block then holds FOUR CONSECUTIVE PIXELS
//     of one channel per lane, and the horizontal half of the 3x3 max-pool is register-local (v_max3_f32; one ds_bpermute per block
//     for the pixel left of the lane's four).  Pooling before bias, rounding and ReLU (all monotone) gives the same bits and halves
//     their work.  A conv row costs 56 x 128 B of LDS instead of 112 x 128 B, and the vertical half reads 5 values per 2 outputs;
//   * rings: 24 input-row slots (the pair being read spans 15 rows, the 8 rows being written follow it), 10 h-pooled conv rows
//     (5 being pooled + 4 being written), two pooled-pair buffers for C1.
// Measured (profiles/r04_stem_experiments.txt): 93 us at batch 256 = 3.9 TB/s of frames + both outputs (the copy rate of this chip
// is 5.3 TB/s); the conv waves' rows are bound by their own K loop (LDS fragment reads + MFMA at ~70 % of the pipe) -- deferring
// epilogues, per-block software pipelining, opposite phases of the two conv waves of a SIMD and wave priorities all measured the same;
// so did 4 conv waves with two conv rows each (63 fragment reads per two rows instead of 98, fragments double-buffered explicitly: 196
// MFMAs in 4.2 k cycles -- ONE wave per SIMD issues an MFMA every ~21 cycles, two every ~18, as scripts/micro/mfma_peak.hip had said).
// LDS: input ring 24 x 1,856 B | h-pooled conv ring 10 x 56 x 128 B | pooled pairs 2 x 14,336 B | u8 table 3,072 B = 147,968 B.
// ------------------------------------------------------------------------------------------------
constexpr int SF3_THREADS = 768;
constexpr int SF3_SVC = 256;                           // service threads
constexpr int SF3_IN_SLOTS = 24;
constexpr int SF3_H_SLOTS = 10;
constexpr int SF3_H_ROW_BYTES = 56 * 128;
constexpr int SF3_IN_BYTES = SF3_IN_SLOTS * STEM_ROW_BYTES;
constexpr int SF3_H_BYTES = SF3_H_SLOTS * SF3_H_ROW_BYTES;
constexpr int SF3_POOL_BYTES = 2 * 112 * 128;
constexpr int SF3_LDS_BYTES = SF3_IN_BYTES + SF3_H_BYTES + SF3_POOL_BYTES + SF_TAB_BYTES;
constexpr int SF3_PACK_ITEMS = 8 * 56;                 // (input row, group of 4 pixels) of a pair's 8 new rows: 2 rounds of the service threads

template <typename TIN> struct Sf3Load { typedef f32x4 type; };
template <> struct Sf3Load<unsigned char> { typedef unsigned type; };

template <int ET, typename TIN, bool C1>
__global__ __launch_bounds__(SF3_THREADS) void stem_fused3_kernel(const TIN* __restrict__ x, const char* __restrict__ wpk,
                                                                  const float* __restrict__ bias, __bf16* __restrict__ y,
                                                                  int n_strips, int G, const float* __restrict__ u8_table,
                                                                  const __bf16* __restrict__ c1_w, const float* __restrict__ c1_bias,
                                                                  __bf16* __restrict__ y1
#if defined(R50_STAMP)    // diagnostic build (scripts/stamp_stem.py): per-wave cycle sums, 8 slots per wave
                                                                  , unsigned long long* dbg
#endif
                                                                  ) {
#if defined(R50_STAMP)
    struct { unsigned long long* dbg; } a{dbg};
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_in = smem;
    char* s_h = smem + SF3_IN_BYTES;
    char* s_pool = smem + SF3_IN_BYTES + SF3_H_BYTES;
    float* s_tab = reinterpret_cast<float*>(smem + SF3_IN_BYTES + SF3_H_BYTES + SF3_POOL_BYTES);
    constexpr bool U8 = (sizeof(TIN) == 1);
    typedef typename Sf3Load<TIN>::type ld_t;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const bool conv_wave = wave < 8;
    const int rr = (wave >> 1) & 3, ch = wave & 1;     // conv waves: conv row of the pair, channel half; service waves: C1 pixel-block parity (rr & 1), channel half
    const int sid = tid - 512;                         // service thread id (>= 0 on the service waves)

    // the ring's border columns (wp 0..3 and 228..231 = the conv's horizontal zero padding) are written once: packing only writes wp 4..227
    for (int c = tid; c < SF3_IN_SLOTS * 8; c += SF3_THREADS) {
        const int slot = c >> 3, i = c & 7;
        *reinterpret_cast<u32x2*>(s_in + (slot * STEM_WP + (i < 4 ? i : 224 + i)) * 8) = (u32x2){0u, 0u};
    }
    if constexpr (U8) {
        for (int c = tid; c < 3 * 256; c += SF3_THREADS) s_tab[c] = u8_table[c];
    }

    // The two roles run the same strip / pair loops with the same barriers, each in its own branch: what one role keeps in registers
    // across the loop (the conv weights; the prefetched pixels and the C1 weights) is not live in the other.
    const int spi = 28 / G;                            // strips per image
    const int kend = G + (C1 ? 2 : 1);                 // pooling runs one pair behind the conv, C1 two
    R50_STAMP_DECL
#if defined(R50_STAMP)
    const unsigned long long st_c0 = __builtin_readcyclecounter();
#endif
    if (conv_wave) {
        // ---- conv waves ----
        // The image fragment is the MFMA's A operand (16 pixels x 32 K) and the weights its B operand (32 K x 16 channels), so an
        // accumulator block holds, per lane, FOUR CONSECUTIVE PIXELS (4 fq + e) of one channel (fr): the 3-max over pixels is register-
        // local (v_max3_f32) except for pixel 4 fq - 1, which comes from the lanes 16 below (ds_bpermute, one per block).  The two
        // blocks of a wave carry channels 32 ch + 2 fr and + 2 fr + 1, so a pooled pixel's pair packs into one dword.  bias, the
        // rounding and ReLU are monotone: max first, then they run on 2 pooled values instead of 4 (the same bits).
        // Weight rows of the packed image are permuted (perm_row_to_cout in r50_abi.hip); channel c sits in row
        // (c & ~31) | (c & 3) | ((c >> 3) & 3) << 2 | ((c >> 2) & 1) << 4.
        // A row runs as two halves (pixel blocks 0-3, then 4-6), each K loop followed by its epilogue, and the two conv waves of a SIMD
        // (waves w and w + 4) differ in priority: the favoured wave's K loop takes the MFMA pipe, the other one's epilogue (VALU, LDS
        // round trips) runs underneath it, and the phases of the two stay interleaved from there on.
        bf16x8 wreg[7][2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int cc = 32 * ch + 2 * fr + nb;
            const int rho = (cc & ~31) | (cc & 3) | (((cc >> 3) & 3) << 2) | (((cc >> 2) & 1) << 4);
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) wreg[kh][nb] = *reinterpret_cast<const bf16x8*>(wpk + kh * 4096 + rho * 64 + fq * 16);
        }
        const float b0 = bias[32 * ch + 2 * fr], b1 = bias[32 * ch + 2 * fr + 1];
        const int rot_addr = ((lane + 48) & 63) << 2;   // ds_bpermute: read from the lane 16 below (mod 64)
        const int x_lane = fr * 16 + fq * 16;
        const int h_lane = 2 * fq * 128 + (((4 * ch + (fr >> 2)) ^ ((fq & 1) << 2)) << 4) + (fr & 3) * 4;
        // conv row c (0 <= c < 112) of this wave's channel half: input rows 2c-3 .. 2c+3 from the ring -> accumulators (K loop: 7 kernel
        // rows x 7 pixel blocks x 2 channel blocks) -> every block's pixel 4 fq + 3 from the lanes 16 below (one LDS round trip for the row)
        // -> horizontal 3-max, bias, ReLU -> h-pooled ring slot c % 10: pooled column q at q * 128, 16-B chunk g at g ^ (q & 2) * 2,
        // channel pair inside the chunk.
        auto conv_row = [&](int c) {
            int slot[7];
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) slot[kh] = __builtin_amdgcn_readfirstlane((2 * c - 3 + kh + 2 * SF3_IN_SLOTS) % SF3_IN_SLOTS) * STEM_ROW_BYTES;
            char* hrow = s_h + __builtin_amdgcn_readfirstlane((c + SF3_H_SLOTS) % SF3_H_SLOTS) * SF3_H_ROW_BYTES + h_lane;
            f32x4 acc[2][7];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int j = 0; j < 7; ++j) acc[nb][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) {
                bf16x8 xf[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(s_in + slot[kh] + x_lane + j * 256);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int j = 0; j < 7; ++j) acc[nb][j] = mfma_e<ET>(xf[j], wreg[kh][nb], acc[nb][j]);
            }
            R50_MARK(1)
            float rot[2][7];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const float a3 = acc[nb][j][3];    // (a copy: __builtin_bit_cast of the vector-element lvalue reads element 0 with this hipcc)
                    rot[nb][j] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(rot_addr, __builtin_bit_cast(int, a3)));
                }
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                float p0[2], p1[2];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const f32x4 a = acc[nb][j];
                    // pixel 16 j + 4 fq - 1: from the lanes 16 below; on the lanes fq = 0 the previous block's lanes fq = 3 (the rotation wraps),
                    // and left of the image the window's own pixel stands in for the -inf pad
                    const float left = fq == 0 ? (j == 0 ? a[0] : rot[nb][j > 0 ? j - 1 : 0]) : rot[nb][j];
                    p0[nb] = __builtin_fmaxf(__builtin_fmaxf(left, a[0]), a[1]);        // window centre 16 j + 4 fq     = pooled column 8 j + 2 fq
                    p1[nb] = __builtin_fmaxf(__builtin_fmaxf(a[1], a[2]), a[3]);        // window centre 16 j + 4 fq + 2 = pooled column 8 j + 2 fq + 1
                }
                // pooled columns q = 8 j + 2 fq and q + 1: (q >> 1) & 1 is the same for both, one swizzle key (in h_lane) for both stores
                *reinterpret_cast<unsigned*>(hrow + j * 8 * 128) = relu_bf16x2(pack2_e<ET>(p0[0] + b0, p0[1] + b1));
                *reinterpret_cast<unsigned*>(hrow + j * 8 * 128 + 128) = relu_bf16x2(pack2_e<ET>(p1[0] + b0, p1[1] + b1));
            }
            R50_MARK(2)
        };
        for (int strip = blockIdx.x; strip < n_strips; strip += gridDim.x) {
            const int t0 = (strip % spi) * G;          // first pooled-row pair of the strip
            __syncthreads();
            for (int k = 0; k < kend; ++k) {
                const int r0 = 2 * (t0 + k);           // first pooled row of the pair
                R50_MARK(3)
                __syncthreads();                       // pair k's input rows are packed (and the slots of the h-pooled rows it writes are free)
                R50_MARK(0)
                if (k < G) {
                    if (k == 0 && r0 > 0 && rr == 3) conv_row(2 * r0 - 1);       // a strip that starts inside the image: the conv row it shares with the strip above
                    conv_row(2 * r0 + rr);
                }
            }
        }
    } else {
        // ---- service waves ----
        // packing: item i of a pair's 8 new rows = (row i / 56, pixels 4 (i % 56) .. + 3); thread sid owns items sid and sid + 256 (the
        // second one exists for sid < 192).  What does not change from pair to pair is computed once, here.
        int pk_r[2], pk_src[2], pk_dst[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = sid + it * SF3_SVC < SF3_PACK_ITEMS ? sid + it * SF3_SVC : SF3_PACK_ITEMS - 1;    // (the idle slot loads a valid address and stores nothing)
            const int r = item / 56, g = item - r * 56;
            pk_r[it] = r;
            pk_src[it] = 4 * g;
            pk_dst[it] = (4 + 4 * g) * 8;
        }
        const bool pk_second = sid + SF3_SVC < SF3_PACK_ITEMS;
        ld_t ld[2][3];
        unsigned ld_ok = 0;                            // bit it: item `it` lies inside the image (rows above and below it pack as zeros)
        int pk_off[2];                                 // byte offset of the item's 4 pixels from the first of the 8 rows
#pragma unroll
        for (int it = 0; it < 2; ++it) pk_off[it] = (pk_r[it] * 224 + pk_src[it]) * (int)sizeof(TIN);
        auto load_rows = [&](int n, int first_row) {
            const bool inside = first_row >= 0 && first_row + 8 <= 224;    // (uniform) all 8 rows inside the image: every pair but an image's first and last
            const char* img = reinterpret_cast<const char*>(x + (size_t)n * 3 * 224 * 224 + (inside ? first_row * 224 : 0));
            ld_ok = 3u;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                int off = pk_off[it];
                if (!inside) {
                    const int hi = first_row + pk_r[it];
                    const bool ok = (unsigned)hi < 224u;
                    if (!ok) ld_ok &= ~(1u << it);
                    off = ((ok ? hi : 0) * 224 + pk_src[it]) * (int)sizeof(TIN);
                }
#pragma unroll
                for (int cc = 0; cc < 3; ++cc)
                    ld[it][cc] = *reinterpret_cast<const ld_t*>(img + (size_t)cc * 224 * 224 * sizeof(TIN) + (unsigned)off);
            }
        };
        auto pack_rows = [&](int first_row) {          // registers -> ring slots row % 24, bf16 [wp][4]
            const int s0 = __builtin_amdgcn_readfirstlane((first_row + 2 * SF3_IN_SLOTS) % SF3_IN_SLOTS);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                if (it == 0 || pk_second) {
                    int slot = s0 + pk_r[it];
                    slot = slot >= SF3_IN_SLOTS ? slot - SF3_IN_SLOTS : slot;
                    const bool ok = (ld_ok >> it) & 1u;
                    unsigned w[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v[3];
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) {
                            if constexpr (U8) v[cc] = s_tab[cc * 256 + ((ld[it][cc] >> (8 * i)) & 255u)];
                            else v[cc] = ld[it][cc][i];
                        }
                        w[2 * i] = pack2_e<ET>(v[0], v[1]);
                        w[2 * i + 1] = pack2_e<ET>(v[2], 0.f);
                    }
                    if (!ok) {                         // a row above or below the image: the conv's zero padding
#pragma unroll
                        for (int i = 0; i < 8; ++i) w[i] = 0u;
                    }
                    char* dst = s_in + slot * STEM_ROW_BYTES + pk_dst[it];
                    *reinterpret_cast<u32x4*>(dst) = (u32x4){w[0], w[1], w[2], w[3]};
                    *reinterpret_cast<u32x4*>(dst + 16) = (u32x4){w[4], w[5], w[6], w[7]};
                }
            }
        };
        // vertical 3-max: pooled row r0 = rows 2 r0 - 1, 2 r0, 2 r0 + 1 of the h-pooled ring, pooled row r0 + 1 = rows 2 r0 + 1 .. 2 r0 + 3
        // (five reads give both).  Thread sid owns (column q, chunk g) = (sid >> 3, sid & 7) and (32 + (sid >> 3), sid & 7) (sid < 192).
        int pl_off[2], pl_pool[2], pl_out[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q = (sid >> 3) + 32 * it < 56 ? (sid >> 3) + 32 * it : 55, g = sid & 7;
            pl_off[it] = q * 128 + ((g ^ ((q & 2) << 1)) << 4);
            pl_pool[it] = q * 128 + ((g ^ (q & 7)) << 4);          // s_pool: pixel pp = pr * 56 + q at pp * 128, chunk g at g ^ (pp & 7) (56 = 0 mod 8)
            pl_out[it] = (q * 64 + g * 8) * 2;                      // bytes
        }
        const bool pl_second = (sid >> 3) + 32 < 56;
        auto pool_pair = [&](int n, int r0, int pbuf) {
            int hs[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) hs[i] = __builtin_amdgcn_readfirstlane((2 * r0 - 1 + i + SF3_H_SLOTS) % SF3_H_SLOTS) * SF3_H_ROW_BYTES;
            u32x4 h[2][5];
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int i = 0; i < 5; ++i) h[it][i] = *reinterpret_cast<const u32x4*>(s_h + hs[i] + pl_off[it]);
            char* yrow = reinterpret_cast<char*>(y + ((size_t)n * 56 + r0) * 56 * 64);
            char* prow = s_pool + pbuf * (112 * 128);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                u32x4 o0, o1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned top = r0 > 0 ? h[it][0][e] : 0u;                     // conv row -1 does not exist (values are >= +0: 0 is the pad)
                    o0[e] = max_bf16x2_nonneg(max_bf16x2_nonneg(top, h[it][1][e]), h[it][2][e]);
                    o1[e] = max_bf16x2_nonneg(max_bf16x2_nonneg(h[it][2][e], h[it][3][e]), h[it][4][e]);
                }
                if (it == 0 || pl_second) {
                    *reinterpret_cast<u32x4*>(yrow + (unsigned)pl_out[it]) = o0;       // (uniform row pointer + a 32-bit lane offset: no 64-bit lane arithmetic)
                    *reinterpret_cast<u32x4*>(yrow + (unsigned)(pl_out[it] + 56 * 128)) = o1;
                    if constexpr (C1) {
                        *reinterpret_cast<u32x4*>(prow + pl_pool[it]) = o0;
                        *reinterpret_cast<u32x4*>(prow + 56 * 128 + pl_pool[it]) = o1;
                    }
                }
            }
        };
        // layer1.0.conv1 on a pooled pair.  Accumulator block m, row 4 q + e is channel 32 ch + 8 q + 4 m + e (the igemm kernels'
        // convention); pixel blocks rr & 1, + 2, + 4 (, + 6) of the pair's 7
        bf16x8 a1[2][2];
        float c1b[2][4];
        int c1_in[2], c1_out = 0;
        if constexpr (C1) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int chan = 32 * ch + 8 * (fr >> 2) + (fr & 3) + 4 * m;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) a1[m][kk] = *reinterpret_cast<const bf16x8*>(c1_w + chan * 64 + kk * 32 + fq * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) c1b[m][e] = c1_bias[32 * ch + 8 * fq + 4 * m + e];
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) c1_in[kk] = fr * 128 + (((4 * kk + fq) ^ (fr & 7)) << 4);     // pixel p = 16 jb + fr: p & 7 = fr & 7
            c1_out = (fr * 64 + 32 * ch + 8 * fq) * 2;     // bytes
        }
        auto conv1_pair = [&](int n, int r0, int pbuf) {
            if constexpr (C1) {
                const int jb0 = rr & 1, nblk = 4 - jb0;            // blocks jb0, jb0 + 2, ..: 4 of them on the even waves, 3 on the odd
                const char* prow = s_pool + pbuf * (112 * 128) + jb0 * 16 * 128;
                char* out = reinterpret_cast<char*>(y1 + (((size_t)n * 56 + r0) * 56 + 16 * jb0) * 64);
                bf16x8 xb[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) xb[i][kk] = *reinterpret_cast<const bf16x8*>(prow + (i < nblk ? i : 0) * 32 * 128 + c1_in[kk]);
                f32x4 lo[4], hi[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lo[i] = (f32x4){c1b[0][0], c1b[0][1], c1b[0][2], c1b[0][3]};
                    hi[i] = (f32x4){c1b[1][0], c1b[1][1], c1b[1][2], c1b[1][3]};
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        lo[i] = mfma_e<ET>(a1[0][kk], xb[i][kk], lo[i]);
                        hi[i] = mfma_e<ET>(a1[1][kk], xb[i][kk], hi[i]);
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < nblk) {
                        const u32x4 o = (u32x4){relu_bf16x2(pack2_e<ET>(lo[i][0], lo[i][1])), relu_bf16x2(pack2_e<ET>(lo[i][2], lo[i][3])),
                                                relu_bf16x2(pack2_e<ET>(hi[i][0], hi[i][1])), relu_bf16x2(pack2_e<ET>(hi[i][2], hi[i][3]))};
                        *reinterpret_cast<u32x4*>(out + (unsigned)(c1_out + i * 32 * 128)) = o;
                    }
                }
            }
        };
        for (int strip = blockIdx.x; strip < n_strips; strip += gridDim.x) {
            const int n = strip / spi;
            const int t0 = (strip - n * spi) * G;
            __syncthreads();                           // the previous strip is drained (first strip: borders and table are in place)
            load_rows(n, 8 * t0 - 6);                  // rows 4 r0 - 6 .. 4 r0 + 1 (r0 = 2 t0): what a running strip would already hold
            pack_rows(8 * t0 - 6);
            load_rows(n, 8 * t0 + 2);                  // the first pair's own 8 rows
            pack_rows(8 * t0 + 2);
            if (G > 1) load_rows(n, 8 * t0 + 10);
            for (int k = 0; k < kend; ++k) {
                const int r0 = 2 * (t0 + k);           // first pooled row of the pair the conv waves work on
                R50_MARK(4)
                __syncthreads();                       // pair k-1's conv rows are complete, pair k-2's pooled rows too
                R50_MARK(0)
                if (k + 1 < G) {
                    pack_rows(4 * r0 + 10);            // pair k+1's new rows 4 (r0 + 2) + 2 ..
                    R50_MARK(1)
                    if (k + 2 < G) load_rows(n, 4 * r0 + 18);
                    R50_MARK(2)
                }
                if (k >= 1 && k <= G) pool_pair(n, r0 - 2, (k - 1) & 1);
                R50_MARK(3)
                if (C1 && k >= 2) conv1_pair(n, r0 - 4, k & 1);
            }
        }
    }
#if defined(R50_STAMP)
    st_sum[6] = __builtin_readcyclecounter() - st_c0;
#endif
    R50_STAMP_FLUSH(12)
}

// ================================================================================================
// Split-precision ("fp32x") variants of the non-GEMM kernels.  In this mode every activation travels
// as a pair of bf16 tensors (head, tail) with head + tail ~ the fp32 value (16 mantissa bits), stored
// channel-concatenated per pixel: [head(C) | tail(C)].  Convolutions run three bf16 MFMA products
// (x_head*w_head + x_tail*w_head + x_head*w_tail) with fp32 accumulation -- fp32-class accuracy on the
// bf16 matrix cores, about 3x the bf16 work (gfx950 has no TF32-like mode and its fp32 MFMA runs at
// 1/16 of the bf16 rate).
// ================================================================================================
__device__ __forceinline__ void split_bf16(float v, unsigned& head16, unsigned& tail16) {
    const unsigned h = pack2_e<0>(v, 0.f) & 0xffffu;
    head16 = h;
    tail16 = pack2_e<0>(v - bf16_bits_to_f32(h), 0.f) & 0xffffu;
}

template <typename TIN>
__global__ __launch_bounds__(256) void stem_pack_split_kernel(const TIN* __restrict__ x, u32x2* __restrict__ xp_head,
                                                              u32x2* __restrict__ xp_tail, int n_img) {
    const long long total = (long long)n_img * STEM_HP * STEM_WP;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int wp = (int)(idx % STEM_WP);
        const long long t = idx / STEM_WP;
        const int hp = (int)(t % STEM_HP);
        const int n = (int)(t / STEM_HP);
        const int hi = hp - 3, wi = wp - 4;
        u32x2 oh = (u32x2){0u, 0u}, ot = (u32x2){0u, 0u};
        if ((unsigned)hi < 224u && (unsigned)wi < 224u) {
            const TIN* p = x + ((size_t)n * 3 * 224 + hi) * 224 + wi;
            unsigned h0, t0, h1, t1, h2, t2;
            split_bf16(frame_value(p, 0), h0, t0);
            split_bf16(frame_value(p + 224 * 224, 1), h1, t1);
            split_bf16(frame_value(p + 2 * 224 * 224, 2), h2, t2);
            oh[0] = h0 | (h1 << 16); oh[1] = h2;
            ot[0] = t0 | (t1 << 16); ot[1] = t2;
        }
        xp_head[idx] = oh;
        xp_tail[idx] = ot;
    }
}

constexpr int STEM_SPLIT_LDS_BYTES = 2 * STEM_W_BYTES + 2 * STEM_IN_ROWS * STEM_ROW_BYTES;

// y: (N,112,112,128) = [head(64) | tail(64)] per pixel
__global__ __launch_bounds__(256) void stem_conv_split_kernel(const char* __restrict__ xp_head, const char* __restrict__ xp_tail,
                                                              const char* __restrict__ w_head, const char* __restrict__ w_tail,
                                                              const float* __restrict__ bias, __bf16* __restrict__ y) {
    constexpr int ET = 0;      // the (head, tail) pair format is bf16-specific
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int IN_BYTES = STEM_IN_ROWS * STEM_ROW_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x / (112 / STEM_ROWS_PER_WG);
    const int ho0 = (blockIdx.x % (112 / STEM_ROWS_PER_WG)) * STEM_ROWS_PER_WG;

    for (int c = tid; c < STEM_W_BYTES / 16; c += 256) {
        *reinterpret_cast<u32x4*>(smem + c * 16) = *reinterpret_cast<const u32x4*>(w_head + c * 16);
        *reinterpret_cast<u32x4*>(smem + STEM_W_BYTES + c * 16) = *reinterpret_cast<const u32x4*>(w_tail + c * 16);
    }
    const size_t src_off = ((size_t)n * STEM_HP + 2 * ho0) * STEM_ROW_BYTES;
    for (int c = tid; c < IN_BYTES / 16; c += 256) {
        *reinterpret_cast<u32x4*>(smem + 2 * STEM_W_BYTES + c * 16) = *reinterpret_cast<const u32x4*>(xp_head + src_off + c * 16);
        *reinterpret_cast<u32x4*>(smem + 2 * STEM_W_BYTES + IN_BYTES + c * 16) =
            *reinterpret_cast<const u32x4*>(xp_tail + src_off + c * 16);
    }
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[4][7];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int w_frag = fr * 64 + fq * 16;
    const int x_frag = (2 * wave) * STEM_ROW_BYTES + fr * 16 + fq * 16;
#pragma unroll
    for (int combo = 0; combo < 3; ++combo) {            // (x_head,w_head) (x_tail,w_head) (x_head,w_tail)
        const char* wbase = smem + (combo == 2 ? STEM_W_BYTES : 0);
        const char* xbase = smem + 2 * STEM_W_BYTES + (combo == 1 ? IN_BYTES : 0);
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            bf16x8 wf[4], xf[7];
#pragma unroll
            for (int m = 0; m < 4; ++m) wf[m] = *reinterpret_cast<const bf16x8*>(wbase + kh * 4096 + m * 1024 + w_frag);
#pragma unroll
            for (int j = 0; j < 7; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(xbase + x_frag + kh * STEM_ROW_BYTES + j * 256);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 7; ++j) acc[m][j] = mfma_e<ET>(wf[m], xf[j], acc[m][j]);
        }
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
            u32x4 head, tail;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                head[e] = pack2_e<ET>(v[2 * e], v[2 * e + 1]);
                tail[e] = pack2_e<ET>(v[2 * e] - unpack_lo_e<ET>(head[e]), v[2 * e + 1] - unpack_hi_e<ET>(head[e]));
            }
            __bf16* o = y + (((size_t)n * 112 + ho) * 112 + wo) * 128 + cout;
            *reinterpret_cast<u32x4*>(o) = head;
            *reinterpret_cast<u32x4*>(o + 64) = tail;
        }
    }
}

// MaxPool2d(3,2,1) on [head(C) | tail(C)] pixels: compares head + tail, writes the re-split maximum.
__global__ __launch_bounds__(256) void maxpool3x3s2_split_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y, int N,
                                                                 int H, int W, int C, int Ho, int Wo) {
    constexpr int ET = 0;      // the (head, tail) pair format is bf16-specific
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
                const __bf16* p = x + (((size_t)n * H + hi) * W + wi) * (2 * C) + g * 8;
                const u32x4 vh = *reinterpret_cast<const u32x4*>(p);
                const u32x4 vt = *reinterpret_cast<const u32x4*>(p + C);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    m[2 * e] = fmaxf(m[2 * e], unpack_lo_e<ET>(vh[e]) + unpack_lo_e<ET>(vt[e]));
                    m[2 * e + 1] = fmaxf(m[2 * e + 1], unpack_hi_e<ET>(vh[e]) + unpack_hi_e<ET>(vt[e]));
                }
            }
        }
        u32x4 head, tail;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            head[e] = pack2_e<ET>(m[2 * e], m[2 * e + 1]);
            tail[e] = pack2_e<ET>(m[2 * e] - unpack_lo_e<ET>(head[e]), m[2 * e + 1] - unpack_hi_e<ET>(head[e]));
        }
        __bf16* o = y + (idx / cg) * (2 * C) + g * 8;
        *reinterpret_cast<u32x4*>(o) = head;
        *reinterpret_cast<u32x4*>(o + C) = tail;
    }
}

// Global average pool on [head(C) | tail(C)] pixels -> (N, C) fp32.
__global__ __launch_bounds__(256) void avgpool_split_kernel(const __bf16* __restrict__ x, float* __restrict__ y, int N, int HW,
                                                            int C, float inv_hw) {
    constexpr int ET = 0;      // the (head, tail) pair format is bf16-specific
    const int cg = C >> 3;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * cg) return;
    const int g = idx % cg, n = idx / cg;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    const __bf16* p = x + (size_t)n * HW * (2 * C) + g * 8;
    for (int r = 0; r < HW; ++r) {
        const u32x4 vh = *reinterpret_cast<const u32x4*>(p + (size_t)r * 2 * C);
        const u32x4 vt = *reinterpret_cast<const u32x4*>(p + (size_t)r * 2 * C + C);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s[2 * e] += unpack_lo_e<ET>(vh[e]) + unpack_lo_e<ET>(vt[e]);
            s[2 * e + 1] += unpack_hi_e<ET>(vh[e]) + unpack_hi_e<ET>(vt[e]);
        }
    }
    float* o = y + (size_t)n * C + g * 8;
    *reinterpret_cast<f32x4*>(o) = (f32x4){s[0] * inv_hw, s[1] * inv_hw, s[2] * inv_hw, s[3] * inv_hw};
    *reinterpret_cast<f32x4*>(o + 4) = (f32x4){s[4] * inv_hw, s[5] * inv_hw, s[6] * inv_hw, s[7] * inv_hw};
}
