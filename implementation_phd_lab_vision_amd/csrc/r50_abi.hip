// Host side of libr50hip.so: C ABI declared in include/r50.h.
// Builds the static ResNet-50 [:-1] schedule (upstream torchvision models/resnet.py, the module
// list the reference cuts at src/preprocess_resnet_features.py:207-209), folds BN, packs weights
// and launches the gfx950 kernels of kernels.h.
#include "kernels.h"
#include "../../include/r50.h"

#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;   // errors raised without a handle (r50_create, r50_op_*)

struct Prof {
    const char* name;
    int64_t launches = 0;
    double ms = 0, flops = 0, bytes = 0;
};
enum ProfClass { PC_IGEMM = 0, PC_STEM_CONV, PC_MAXPOOL, PC_AVGPOOL, PC_STEM_PACK, PC_TAIL, PC_TAIL3, PC_BLOCK2, PC_CATCHAIN, PC_COUNT };
const char* kProfNames[PC_COUNT] = {"igemm", "conv1", "maxpool", "avgpool", "stem_pack", "bneck_tail", "bneck_tail3", "bneck_block2", "bneck_catchain"};

struct EvRec {
    hipEvent_t a, b;
    int cls;
    int layer;        // index into convs (per-layer breakdown) or -1
    double flops, bytes;
};

struct ConvLayer {
    std::string conv_key, bn_key;
    int cin, cout, ks, stride, pad;
    __bf16* w = nullptr;    // device, (cout, ks, ks, cin)
    float* bias = nullptr;  // device
    // fp8 mode, layer2-4: w holds e4m3 bytes with real value = byte value x wscale; bias_scaled = bias / (sx * wscale) once the
    // activation scales are known (r50_set_fp8_scales)
    float wscale = 1.0f;
    float* bias_scaled = nullptr;
    std::vector<float> bias_host;
    std::vector<float> w_host;   // fp8 mode, conv3 / downsample of a stage's first block: folded fp32 weights (OIHW), requantised
                                 // with a common accumulator scale once the activation scales are known
};

}  // namespace

struct r50_handle {
    int device = 0;
    int precision = R50_PREC_BF16;
    int max_batch = 0;
    int micro_batch = 0;
    int profile = 0;
    int tile_override = 0;
    int overlap_ds = 0;                 // downsample conv of a stage's first block on a side stream
    hipStream_t ds_stream = nullptr;
    hipEvent_t ev_ds_fork = nullptr, ev_ds_join = nullptr;
    int fused_stem = 1;                 // bf16 mode: conv1+bn1+relu+maxpool in one kernel
    int fuse_stem_c1 = 1;               // strip stem kernel also computes layer1.0.conv1
    std::vector<float> fp8_scales;      // R50_PREC_FP8: activation scales, execution order (r50_set_fp8_scales)
    float cat_acc_scale[4] = {0.f, 0.f, 0.f, 0.f};   // fp8 two-source conv of layer2.0 / 3.0 / 4.0: real value of one accumulator unit
    int fuse_ds_cat = 1;                // layer2.0 / 3.0 / 4.0: conv3 and the downsample conv as one GEMM over K = [t2 | x at stride 2]
    __bf16* cat_w[4] = {nullptr, nullptr, nullptr, nullptr};      // per stage: (cout, cmid + cin) = [W3 | Wd], device
    float* cat_bias[4] = {nullptr, nullptr, nullptr, nullptr};    // b3 + bd (fp32)
    int fuse_tail = 1;                  // bf16 mode, layer1: conv3 + identity + ReLU + the next block's conv1 in one kernel
    int fuse_fp8_handover = 1;          // fp8 mode: quantise layer1's output in layer1.2.conv3's epilogue instead of in a pass of its own
    int fuse_tail3 = 1;                 // layer3.1-.4: conv3 + identity + ReLU chained with the next block's conv1 through LDS (bneck_tail3_kernel)
    __bf16* catchain_wp = nullptr;      // layer2.0: [W3 | Wd] and layer2.1.conv1 in bneck_catchain_kernel's fragment-ordered stream
    int fuse_tail3_last = 1;            // layer3.5: conv3 + identity + ReLU through the pipelined tail kernel without a second GEMM
    int fuse_cat_chain = 1;             // layer2.0: conv3 + downsample + ReLU chained with layer2.1.conv1 in one launch (bneck_catchain_kernel)
    int sub_out = 1;                    // layer1.2 stores only the even rows / columns of its output (its readers: the fused layer2.0.conv1 and the stride-2 downsample
                                        // conv inside bneck_catchain_kernel); needs fuse_block1 >= 2 and the chained layer2.0 tail, else the full tensor is written
    __bf16* tail3_wp[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // per layer3 block b: [W3(b) | W1(b+1)] in the kernel's fragment order
    int fuse_block1 = 3;                // 1: layer1.1, 2 (default since round 3: with the loaders' position loop unrolled the c1 = 128 form takes 248 us against
                                        // 276 for conv2 + fused tail): also layer1.2 -- the bottleneck body in one launch (bneck_block1_kernel)
    int fuse_block2 = 1;                // layer2.1-.3: the whole bottleneck body (conv2 + conv3 + identity + ReLU [+ next conv1]) in one launch
    int inplace_out = 0;                // plain-identity blocks write their output over their input (same bits, fewer DRAM page switches)
    int n_streams = 1;                  // > 1: the batch is split over internal streams (forked from / joined to the caller's)
    hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    bool loaded = false;
    // r50_share_weights: this handle reads the weight buffers of `owner` (convs[].w / bias, cat_w / cat_bias, tail3_wp, catchain_wp, stem_w) and
    // frees none of them; the owner counts its sharers and, if destroyed first, stays alive (weights only) until the last sharer is gone
    r50_handle* owner = nullptr;
    int sharers = 0;
    bool zombie = false;
    std::string err;
    std::vector<ConvLayer> convs;       // execution order, convs[0] = stem
    char* stem_w = nullptr;             // packed stem weights (device)
    char* stem_xp = nullptr;            // packed input image (device)
    float* u8_table = nullptr;          // [3][256]: uint8 sample -> normalised fp32, for r50_forward_u8 (device)
    __bf16* buf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t buf_bytes = 0;
    Prof prof[PC_COUNT];
    std::vector<Prof> prof_layer;       // one entry per conv, same order as convs
    std::vector<EvRec> ev_pending;
    std::vector<hipEvent_t> ev_free;
};

namespace {

int fail(r50_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    g_err = msg;
    return code;
}

#define HIP_TRY(h, expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(h, R50_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

constexpr int kStages[4][3] = {{64, 3, 1}, {128, 4, 2}, {256, 6, 2}, {512, 3, 2}};
constexpr size_t kFp8FirstConv = 11;     // convs[1..10] = layer1 (4 + 3 + 3); fp8 mode quantises from layer2.0.conv1 on
constexpr float kBnEps = 1e-5f;

std::vector<ConvLayer> make_specs() {
    std::vector<ConvLayer> v;
    auto add = [&](const std::string& c, const std::string& b, int cin, int cout, int ks, int s, int p) {
        ConvLayer L;
        L.conv_key = c; L.bn_key = b; L.cin = cin; L.cout = cout; L.ks = ks; L.stride = s; L.pad = p;
        v.push_back(L);
    };
    add("conv1", "bn1", 3, 64, 7, 2, 3);
    int inpl = 64;
    for (int si = 0; si < 4; ++si) {
        const int planes = kStages[si][0], blocks = kStages[si][1], stride = kStages[si][2];
        for (int b = 0; b < blocks; ++b) {
            const int s = (b == 0) ? stride : 1;
            const std::string p = "layer" + std::to_string(si + 1) + "." + std::to_string(b);
            add(p + ".conv1", p + ".bn1", inpl, planes, 1, 1, 0);
            add(p + ".conv2", p + ".bn2", planes, planes, 3, s, 1);
            add(p + ".conv3", p + ".bn3", planes, planes * 4, 1, 1, 0);
            if (b == 0) add(p + ".downsample.0", p + ".downsample.1", inpl, planes * 4, 1, s, 0);
            inpl = planes * 4;
        }
    }
    return v;
}

// BN fold in fp32, op order fixed (no contraction: built with -ffp-contract=off):
//   scale = gamma / sqrt(var + eps);  w' = w * scale;  b' = beta - mean * scale
void fold_bn(const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
             int cout, int per_out, std::vector<float>& wf, std::vector<float>& bf) {
    wf.resize((size_t)cout * per_out);
    bf.resize(cout);
    for (int o = 0; o < cout; ++o) {
        const float ve = var[o] + kBnEps;
        const float sd = std::sqrt(ve);
        const float scale = gamma[o] / sd;
        for (int i = 0; i < per_out; ++i) wf[(size_t)o * per_out + i] = w[(size_t)o * per_out + i] * scale;
        const float ms = mean[o] * scale;
        bf[o] = beta[o] - ms;
    }
}

// (cout, cin, k, k) fp32 -> (cout, k, k, cin) bf16
// fp32 -> IEEE half, round to nearest even, saturating at +-65504 (the kernels' conversion does the same)
inline uint16_t f32_to_f16_rne(float f) {
    if (f > 65504.0f) f = 65504.0f;
    if (f < -65504.0f) f = -65504.0f;
    const _Float16 hv = (_Float16)f;
    uint16_t u;
    std::memcpy(&u, &hv, 2);
    return u;
}

void pack_ohwi_f16(const float* wf, int cout, int cin, int ks, std::vector<uint16_t>& out) {
    out.resize((size_t)cout * ks * ks * cin);
    for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c)
            for (int t = 0; t < ks * ks; ++t)
                out[((size_t)o * ks * ks + t) * cin + c] = f32_to_f16_rne(wf[((size_t)o * cin + c) * ks * ks + t]);
}

// fp32 -> OCP e4m3 (fn: no infinities, S.1111.111 = NaN), round to nearest even, saturating at +-448
inline uint8_t f32_to_e4m3(float v) {
    const uint8_t sign = std::signbit(v) ? 0x80 : 0x00;
    float a = std::fabs(v);
    if (!(a == a)) return (uint8_t)(sign | 0x7f);
    if (a >= 448.0f) return (uint8_t)(sign | 0x7e);
    if (a < 0.015625f) {                                   // below 2^-6: subnormals, step 2^-9
        const int q = (int)std::nearbyint(a * 512.0f);     // 0 .. 8 (8 = the smallest normal)
        return (uint8_t)(sign | q);
    }
    int e;
    const float m = std::frexp(a, &e);                     // a = m * 2^e, m in [0.5, 1)
    int q = (int)std::nearbyint((m * 2.0f - 1.0f) * 8.0f); // mantissa steps of the binade [2^(e-1), 2^e)
    int ex = e - 1;
    if (q == 8) { q = 0; ++ex; }
    if (ex > 8 || (ex == 8 && q > 6)) return (uint8_t)(sign | 0x7e);
    return (uint8_t)(sign | ((ex + 7) << 3) | q);
}
// (cout, cin, k, k) fp32 -> (cout, k, k, cin) e4m3 bytes with one scale for the tensor (absmax / 448); returns the scale
float pack_ohwi_fp8(const float* wf, int cout, int cin, int ks, std::vector<uint16_t>& out) {
    const size_t total = (size_t)cout * ks * ks * cin;
    float amax = 0.f;
    for (size_t i = 0; i < total; ++i) amax = std::fmax(amax, std::fabs(wf[i]));
    const float scale = amax > 0.f ? amax / 448.0f : 1.0f;
    out.assign((total + 1) / 2, 0);
    uint8_t* o = reinterpret_cast<uint8_t*>(out.data());
    for (int oc = 0; oc < cout; ++oc)
        for (int c = 0; c < cin; ++c)
            for (int t = 0; t < ks * ks; ++t)
                o[((size_t)oc * ks * ks + t) * cin + c] = f32_to_e4m3(wf[((size_t)oc * cin + c) * ks * ks + t] / scale);
    return scale;
}

inline float absmax_of(const std::vector<float>& v) {
    float a = 0.f;
    for (float x : v) a = std::fmax(a, std::fabs(x));
    return a;
}

void pack_ohwi_bf16(const float* wf, int cout, int cin, int ks, std::vector<uint16_t>& out) {
    out.resize((size_t)cout * ks * ks * cin);
    for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c)
            for (int t = 0; t < ks * ks; ++t)
                out[((size_t)o * ks * ks + t) * cin + c] = f32_to_bf16_rne(wf[((size_t)o * cin + c) * ks * ks + t]);
}

inline float bf16_to_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// split mode: (cout, cin, k, k) fp32 -> (cout, k, k, [w_head(cin) | w_head(cin) | w_tail(cin)]) bf16,
// w_head = bf16(w), w_tail = bf16(w - w_head); pairs with X = [x_head | x_tail | x_head]
void pack_ohwi_split(const float* wf, int cout, int cin, int ks, std::vector<uint16_t>& out) {
    out.resize((size_t)cout * ks * ks * 3 * cin);
    for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c)
            for (int t = 0; t < ks * ks; ++t) {
                const float w = wf[((size_t)o * cin + c) * ks * ks + t];
                const uint16_t hd = f32_to_bf16_rne(w);
                const uint16_t tl = f32_to_bf16_rne(w - bf16_to_f32(hd));
                uint16_t* row = &out[((size_t)o * ks * ks + t) * 3 * cin];
                row[c] = hd; row[cin + c] = hd; row[2 * cin + c] = tl;
            }
}

// bf16w2 mode: (cout, cin, k, k) fp32 -> (cout, k, k, [w_head(cin) | w_tail(cin)]) bf16; pairs with X = [x | x]
// (the X chunk index wraps after cin/64, ConvArgs::x_wrap), i.e. x*w_head + x*w_tail with ONE fp32 accumulator.
void pack_ohwi_w2(const float* wf, int cout, int cin, int ks, std::vector<uint16_t>& out) {
    out.resize((size_t)cout * ks * ks * 2 * cin);
    for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c)
            for (int t = 0; t < ks * ks; ++t) {
                const float w = wf[((size_t)o * cin + c) * ks * ks + t];
                const uint16_t hd = f32_to_bf16_rne(w);
                const uint16_t tl = f32_to_bf16_rne(w - bf16_to_f32(hd));
                uint16_t* row = &out[((size_t)o * ks * ks + t) * 2 * cin];
                row[c] = hd; row[cin + c] = tl;
            }
}

inline int perm_row_to_cout(int rho) {   // LDS/MFMA row -> channel inside a 32-row group (kernels.h)
    return (rho & ~31) | (rho & 3) | (((rho >> 4) & 1) << 2) | (((rho >> 2) & 3) << 3);
}

// stem: (64,3,7,7) fp32 folded -> [kh][rho][j=0..7][c=0..3] bf16, j = kw + 1, zero elsewhere
void pack_stem(const float* wf, std::vector<uint16_t>& out, int part = 0) {   // part: 0 = bf16(w), 1 = tail bf16(w - head), 2 = fp16(w)
    out.assign((size_t)7 * 64 * 32, 0);
    for (int kh = 0; kh < 7; ++kh)
        for (int rho = 0; rho < 64; ++rho) {
            const int o = perm_row_to_cout(rho);
            for (int kw = 0; kw < 7; ++kw)
                for (int c = 0; c < 3; ++c)
                {
                    const float w = wf[(((size_t)o * 3 + c) * 7 + kh) * 7 + kw];
                    const uint16_t hd = f32_to_bf16_rne(w);
                    out[(((size_t)kh * 64 + rho) * 8 + (kw + 1)) * 4 + c] = part == 2 ? f32_to_f16_rne(w) : part ? f32_to_bf16_rne(w - bf16_to_f32(hd)) : hd;
                }
        }
}

// n / d for n < 2^31 as (umulhi(n, mul) >> shr): mul = ceil(2^(32+shr) / d), shr = ceil(log2 d) - 1
// (exact: the error term n*e/2^(32+shr) stays below 1/d because n < 2^31 <= 2^(32+shr)/d).
FastDiv make_fast_div(unsigned d) {
    FastDiv f{0u, 0u};
    if (d <= 1u) return f;
    unsigned L = 0;
    while ((1ull << L) < d) ++L;
    f.shr = L - 1;
    const unsigned long long num = 1ull << (32 + f.shr);
    f.mul = (unsigned)((num + d - 1) / d);
    return f;
}

// ---------------------------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------------------------
// Tile-config ids of igemm_bf16_kernel (BC couts x BP pixels):
//   4 waves, 2 LDS stages:                1 = 128x128, 2 = 64x128, 3 = 64x256, 5 = 128x64
//   8 waves, 3 LDS stages, counted vmcnt: 6 = 256x128, 7 = 128x256, 8 = 128x128
//   8 waves, 2 LDS stages, 128 accumulator registers: 9 = 256x256 (wave 64c x 128p), 10 = 256x256 (wave 128c x 64p)
//   8 waves, 2 LDS stages, pixel counts that divide M = 2^10 * 49: 11 = 256x208 (wave 32c x 208p), 12 = 256x224 (wave 64c x 112p)
//   + 32: chip-sized persistent grid (tiles streamed through the LDS ring) instead of one tile per workgroup
constexpr int kPersistBit = 32;
int g_num_cus = 0;
// Workgroups a persistent launch may put on the chip: the CU count, or R50_CU_CAP / the "cu_cap" option when several internal streams
// share the chip (each stream's launches then take their share of the CUs and run beside the other streams' launches).
int g_cu_cap = [] { const char* v = std::getenv("R50_CU_CAP"); return v ? std::atoi(v) : 0; }();
int cu_budget(int device_cus) { return (g_cu_cap > 0 && g_cu_cap < device_cus) ? g_cu_cap : device_cus; }

template <int ET, int BC, int BP, int WC, int WP, int NSTAGE, bool SPLIT = false>
hipError_t launch_igemm_t(ConvArgs a, bool persistent, hipStream_t s) {
    a.n_ctiles = a.Cout / BC;
    a.n_blocks = a.n_ctiles * ((a.M + BP - 1) / BP);
    a.div_ctiles = make_fast_div((unsigned)a.n_ctiles);
    constexpr int kRowsPerPass = WC * WP * 8;
    const size_t lds = (size_t)NSTAGE * (BC + (BP + kRowsPerPass - 1) / kRowsPerPass * kRowsPerPass) * 128;
    auto kern = igemm_bf16_kernel<ET, BC, BP, WC, WP, NSTAGE, SPLIT>;
    if (lds > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int grid = a.n_blocks;
    if (persistent) {
        if (g_num_cus == 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
            g_num_cus = cu_budget(prop.multiProcessorCount);
        }
        // resident workgroups per CU (LDS / registers / wave slots).  No workgroup waits on another one,
        // so an optimistic answer only queues the surplus workgroups; it cannot deadlock.
        static int per_cu = 0;
        if (per_cu == 0) {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, WC * WP * 64, lds) != hipSuccess || n < 1) n = 1;
            per_cu = n;
        }
        if (grid > g_num_cus * per_cu) grid = g_num_cus * per_cu;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WC * WP * 64), lds, s, a);
    return hipGetLastError();
}

// Tile choice.  For the 23 conv shapes of ResNet-50 at large batch the table holds the variant measured
// fastest on an MI355X at batch 256 (scripts/layer_bench.py, profiles/r01_layer_bench.txt); anything else
// falls back to a shape rule.
struct TunedTile { int h, cin, cout, ks, stride, res, tile; };
constexpr TunedTile kTuned[] = {
    {56, 64, 64, 1, 1, 0, 73},    {56, 64, 64, 3, 1, 0, 2},     {56, 64, 256, 1, 1, 1, 72},   {56, 64, 256, 1, 1, 0, 38},
    {56, 256, 64, 1, 1, 0, 73},    {56, 256, 128, 1, 1, 0, 40},  {56, 128, 128, 3, 2, 0, 65},   {28, 128, 512, 1, 1, 1, 72},
    {56, 256, 512, 1, 2, 0, 42},  {28, 512, 128, 1, 1, 0, 65},   {28, 128, 128, 3, 1, 0, 72},  {28, 512, 256, 1, 1, 0, 72},
    {28, 256, 256, 3, 2, 0, 44},  {14, 256, 1024, 1, 1, 1, 72}, {28, 512, 1024, 1, 2, 0, 72}, {14, 1024, 256, 1, 1, 0, 44},
    {14, 256, 256, 3, 1, 0, 68},  {14, 1024, 512, 1, 1, 0, 72}, {14, 512, 512, 3, 2, 0, 68},  {7, 512, 2048, 1, 1, 1, 43},
    {14, 1024, 2048, 1, 2, 0, 43}, {7, 2048, 512, 1, 1, 0, 72}, {7, 512, 512, 3, 1, 0, 68},
};

// Role-specialised kernel (loader waves + consumer waves), always one persistent workgroup per CU: tile id + 64
constexpr int kWsBit = 64;

template <int ET, int BC, int BP, int CWC, int CWP, int NLOAD, int NSTAGE>
hipError_t launch_igemm_ws_t(ConvArgs a, hipStream_t s) {
    a.n_ctiles = a.Cout / BC;
    a.n_blocks = a.n_ctiles * ((a.M + BP - 1) / BP);
    a.div_ctiles = make_fast_div((unsigned)a.n_ctiles);
    constexpr int kRowsPerPass = NLOAD * 8;
    const size_t lds = (size_t)NSTAGE * (BC + (BP + kRowsPerPass - 1) / kRowsPerPass * kRowsPerPass) * 128;
    auto kern = igemm_ws_kernel<ET, BC, BP, CWC, CWP, NLOAD, NSTAGE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        g_num_cus = cu_budget(prop.multiProcessorCount);
    }
    int grid = a.n_blocks < g_num_cus ? a.n_blocks : g_num_cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3((CWC * CWP + NLOAD) * 64), lds, s, a);
    return hipGetLastError();
}

// layer1 conv2 shape (3x3 s1 p1, 64 -> 64, 56x56): filter bank resident in LDS (kernels.h: conv3x3_c64_kernel); tile id 64+16
constexpr int kTileC64 = kWsBit | 16;
bool is_c64_shape(const ConvArgs& a) {
    return a.ks == 3 && a.stride == 1 && a.pad == 1 && a.Cin == 64 && a.Cout == 64 && a.H == 56 && a.W == 56 && a.res == nullptr &&
           a.x_cstride == 64 && a.y_cstride == 64 && a.Ktot == 576;
}
template <int ET>
hipError_t launch_conv3x3_c64(const ConvArgs& a, hipStream_t s) {
    if (!is_c64_shape(a)) return hipErrorInvalidValue;
    constexpr size_t lds = 9 * 64 * 128 + 2 * 352 * 128;      // 163,840 = all of the CU's LDS
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel<ET>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        g_num_cus = cu_budget(prop.multiProcessorCount);
    }
    const int tiles = a.N * 14;
    hipLaunchKernelGGL(conv3x3_c64_kernel<ET>, dim3(tiles < g_num_cus ? tiles : g_num_cus), dim3(512), lds, s, a);
    return hipGetLastError();
}

// conv2 of the non-first blocks of layer2 / layer3 / layer4 (3x3 s1 p1, Cin = Cout): input-resident kernel (kernels.h: conv3x3_xres_kernel);
// tile id 64+17.  Chosen for these shapes at EVERY batch size: its K order differs from the generic kernel's.
constexpr int kTileXres = kWsBit | 17;
bool is_xres_shape(const ConvArgs& a) {
    if (!(a.ks == 3 && a.stride == 1 && a.pad == 1 && a.Cin == a.Cout && a.H == a.W && a.res == nullptr && a.x2 == nullptr &&
          a.x_cstride == a.Cin && a.y_cstride == a.Cout && a.Ktot == 9 * a.Cin && a.et != 2)) return false;
    return (a.H == 14 && a.Cin == 256) || (a.H == 7 && a.Cin == 512) || (a.H == 28 && a.Cin == 128);
}
template <int ET, int NI, int TR, int IW, int IH>
hipError_t launch_conv3x3_xres_t(ConvArgs a, hipStream_t s) {
    constexpr int BC = 128, NST = 3;
    constexpr int PW = IW == 28 ? 32 : 16;
    constexpr int PPT = IW == 7 ? 17 * 16 : NI * (TR + 2) * PW, XBUF = (PPT + 31) / 32 * 32 * 128;
    constexpr size_t lds = 2 * (size_t)XBUF + NST * (size_t)BC * 128;
    static_assert(lds <= 163840, "LDS budget");
    auto kern = conv3x3_xres_kernel<ET, NI, TR, IW, IH>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        g_num_cus = cu_budget(prop.multiProcessorCount);
    }
    a.n_ctiles = a.Cout / BC;
    a.n_blocks = ((a.N + NI - 1) / NI) * (IH / TR) * a.n_ctiles;
    const int grid = a.n_blocks < g_num_cus ? a.n_blocks : g_num_cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(768), lds, s, a);
    return hipGetLastError();
}
template <int ET>
hipError_t launch_conv3x3_xres(const ConvArgs& a, hipStream_t s) {
    if (!is_xres_shape(a)) return hipErrorInvalidValue;
    // The row-block form (kernels.h): 128 couts x 196 pixels per tile, one tap per step, ring of three 16-KB weight stages, static loader schedule.
    // 14x14: one image per tile (512 tiles = two full rounds at batch 256); 28x28: a 7-row band; 7x7: four images (two pairs per 16-position row).
    // Rounds 2-3 carried eleven more schedules of this kernel (13-block addressing, mid-step barriers, rings of 4-5 stages, staggered SIMD partners,
    // 256 couts per tile, the 32x32x16 MFMA): all bit-identical, all measured slower or equal (profiles/r03_xres_variants.txt), removed in round 4.
    if (a.H == 14) return launch_conv3x3_xres_t<ET, 1, 14, 14, 14>(a, s);
    if (a.H == 28) return launch_conv3x3_xres_t<ET, 1, 7, 28, 28>(a, s);
    return launch_conv3x3_xres_t<ET, 4, 7, 7, 7>(a, s);
}

// conv2 of layer2.0 / layer3.0 / layer4.0 (3x3 STRIDE 2 p1, Cin = Cout, 56 -> 28 x 128, 28 -> 14 x 256, 14 -> 7 x 512): input resident by polyphase planes (kernels.h:
// conv3x3_s2_kernel); tile id 64+18.  Chosen for these shapes at EVERY batch size: its K order differs from the generic kernel's.
constexpr int kTileS2 = kWsBit | 18;
bool is_s2_shape(const ConvArgs& a) {
    if (!(a.ks == 3 && a.stride == 2 && a.pad == 1 && a.Cin == a.Cout && a.H == a.W && a.res == nullptr && a.x2 == nullptr &&
          a.x_cstride == a.Cin && a.y_cstride == a.Cout && a.Ktot == 9 * a.Cin && a.et != 2)) return false;
    return (a.H == 56 && a.Cin == 128) || (a.H == 28 && a.Cin == 256) || (a.H == 14 && a.Cin == 512);
}
template <int ET, int TR, int OW, int OH>
hipError_t launch_conv3x3_s2_t(ConvArgs a, hipStream_t s) {
    constexpr size_t lds = 3 * 32768 + 3 * 128 * 128;
    auto kern = conv3x3_s2_kernel<ET, 128, TR, OW, OH>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        g_num_cus = cu_budget(prop.multiProcessorCount);
    }
    a.n_ctiles = a.Cout / 128;
    a.n_blocks = (OW == 7 ? (a.N + 3) / 4 : a.N * (OH / TR)) * a.n_ctiles;
    const int grid = a.n_blocks < g_num_cus ? a.n_blocks : g_num_cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(768), lds, s, a);
    return hipGetLastError();
}
template <int ET>
hipError_t launch_conv3x3_s2(const ConvArgs& a, hipStream_t s) {
    if (!is_s2_shape(a)) return hipErrorInvalidValue;
    if (a.H == 56) return launch_conv3x3_s2_t<ET, 7, 28, 28>(a, s);
    if (a.H == 14) return launch_conv3x3_s2_t<ET, 7, 7, 7>(a, s);          // four images per tile (two pairs)
    return launch_conv3x3_s2_t<ET, 14, 14, 14>(a, s);
}

// 1x1 convs with both operands streaming (layer3 / layer4, K >= 256) as a 256 x 256 / 256 x 224 GEMM tile on the eight-phase schedule
// (kernels.h: gemm8p_kernel); tile ids 64+19 (256 pixels) and 64+20 (224 pixels).  One or two K sources.  Same bits as tile ids 9 / 12.
constexpr int kTileG8 = kWsBit | 19;
constexpr int kTileG8N7 = kWsBit | 20;
bool is_g8_shape(const ConvArgs& a) {
    return a.ks == 1 && a.pad == 0 && a.Cout % 256 == 0 && a.et != 2 && a.x_wrap >= (1 << 30) && a.q_inv == 0.f &&
           a.y_cstride == a.Cout && a.nk >= 1 && (a.x2 ? (a.stride == 1 && a.cc1 >= 1 && a.cc1 < a.nk) : a.x_cstride == a.Cin);
}
template <int ET, int NR, bool DUAL>
hipError_t launch_gemm8p_t(ConvArgs a, hipStream_t s) {
    constexpr int BP = 32 * NR;
    constexpr size_t lds = 2 * 4 * 16384;
    auto kern = gemm8p_kernel<ET, NR, DUAL>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        g_num_cus = cu_budget(prop.multiProcessorCount);
    }
    a.n_ctiles = a.Cout / 256;
    a.n_blocks = a.n_ctiles * ((a.M + BP - 1) / BP);
    a.div_ctiles = make_fast_div((unsigned)a.n_ctiles);
    const int grid = a.n_blocks < g_num_cus ? a.n_blocks : g_num_cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a);
    return hipGetLastError();
}
template <int ET>
hipError_t launch_gemm8p(const ConvArgs& a, int tile, hipStream_t s) {
    if (!is_g8_shape(a)) return hipErrorInvalidValue;
    if (a.x2) return tile == kTileG8N7 ? launch_gemm8p_t<ET, 7, true>(a, s) : launch_gemm8p_t<ET, 8, true>(a, s);
    return tile == kTileG8N7 ? launch_gemm8p_t<ET, 7, false>(a, s) : launch_gemm8p_t<ET, 8, false>(a, s);
}

// tiles a launch of tile id `tile` would have (0 if the id does not divide this Cout)
long long tiles_of(const ConvArgs& a, int tile) {
    int bc = 0, bp = 0;
    if (tile & kWsBit) {
        switch (tile & (kPersistBit - 1)) {
            case 1: bc = 128; bp = 128; break;  case 3: bc = 256; bp = 128; break;  case 4: case 8: bc = 128; bp = 224; break;
            case 9: bc = 64; bp = 224; break;   case 10: bc = 128; bp = 208; break;   case 19: bc = 256; bp = 256; break;   case 20: bc = 256; bp = 224; break;
            default: return 0;
        }
    } else {
        switch (tile & (kPersistBit - 1)) {
            case 1: case 8: bc = 128; bp = 128; break;  case 2: bc = 64; bp = 128; break;   case 3: bc = 64; bp = 256; break;
            case 5: bc = 128; bp = 64; break;           case 6: bc = 256; bp = 128; break;  case 7: bc = 128; bp = 256; break;
            case 9: case 10: bc = 256; bp = 256; break; case 11: bc = 256; bp = 208; break; case 12: bc = 256; bp = 224; break;
            default: return 0;
        }
    }
    if (a.Cout % bc) return 0;
    return (long long)(a.Cout / bc) * ((a.M + bp - 1) / bp);
}

int g_use_xres = [] { const char* v = std::getenv("R50_XRES"); return v ? std::atoi(v) : 1; }();      // A/B knob: 0 = generic igemm tiles for the 3x3 s1 shapes
int g_use_s2 = [] { const char* v = std::getenv("R50_S2"); return v ? std::atoi(v) : 1; }();          // A/B knob: 0 = generic igemm tiles for the 3x3 s2 shapes
// Eight-phase GEMM tiles for the streaming 1x1 convs: 0 = never, 1 = the shapes of kG8 (measured at batch 256), 2 / 3 = wherever the shape fits, 256 / 224 pixels
// (A/B knob: R50_G8 / option "use_g8").
int g_use_g8 = [] { const char* v = std::getenv("R50_G8"); return v ? std::atoi(v) : 1; }();
// Where the eight-phase tiles beat the tuned generic ones at batch 256 (scripts/time_g8.py, profiles/r04_time_g8.txt; same-process medians):
// layer3.1.conv1 32.3 against 33.9 us, layer4.0.conv1 56.8 against 60.0, the two-source conv3 + downsample GEMMs of layer3.0 / layer4.0 88.9 / 86.6
// against 90.7 / 88.6 -- all with 224-pixel tiles (whole blocks of the pixel counts 2^k x 49; 256-pixel tiles leave 23 % of the last round empty).
// NOT layer3.0.conv1 (HBM-bound, 3.5 rounds of 224-pixel tiles), layer4.x.conv1 (98-112 tiles on 256 CUs) and layer4.x.conv3 (K = 512: the
// tile's residual + store epilogue is as long as its K loop): the schedule's 1.36 PFLOP/s shows at K >= 2048 with whole rounds of tiles only.
struct G8Shape { int h, cin, cout, two_sources; };
constexpr G8Shape kG8[] = {{14, 1024, 256, 0}, {14, 1024, 512, 0}, {14, 768, 1024, 1}, {7, 1536, 2048, 1}};
int g8_tile(const ConvArgs& a) {
    if (!g_use_g8 || !is_g8_shape(a) || a.N < 48) return 0;
    if (g_use_g8 == 2) return kTileG8;
    if (g_use_g8 == 3) return kTileG8N7;
    for (const G8Shape& g : kG8)
        if (g.h == a.Ho && a.Ho == a.Wo && g.cin == a.Ktot && g.cout == a.Cout && (g.two_sources != 0) == (a.x2 != nullptr) && a.stride == 1 && !a.res)
            return tiles_of(a, kTileG8N7) >= 200 ? kTileG8N7 : 0;
    return 0;
}
int auto_tile(const ConvArgs& a) {
    if (is_c64_shape(a)) return kTileC64;
    if (const int t8 = g8_tile(a)) return t8;
    if (g_use_xres && is_xres_shape(a)) return kTileXres;
    if (g_use_s2 && is_s2_shape(a)) return kTileS2;
    const long long want = 200;           // of 256 CUs: below that a launch leaves too much of the chip idle
    if (a.N >= 48 && a.H == a.W)
        for (const TunedTile& t : kTuned)
            if (t.h == a.H && t.cin == a.Cin && t.cout == a.Cout && t.ks == a.ks && t.stride == a.stride &&
                t.res == (a.res != nullptr)) {
                // the table was measured at batch 256; a smaller batch may leave its (large) tiles too few to fill the chip
                if (tiles_of(a, t.tile) >= want) return t.tile;
                break;
            }
    // generic choice: the largest of these tiles that still gives the chip enough workgroups, else the one with the most
    const int cand[] = {kWsBit | 8, 1, 5, 2};        // 128x224 role-specialised, 128x128, 128x64, 64x128
    int best = 2;
    long long best_n = -1;
    for (int c : cand) {
        const long long n = tiles_of(a, c);
        if (n >= want) return c;
        if (n > best_n) { best_n = n; best = c; }
    }
    return best;
}

template <int ET>
hipError_t launch_igemm_et(const ConvArgs& a, int tile, hipStream_t s, bool split) {
    if (split && ET != 0) return hipErrorInvalidValue;
    if (split) {       // fp32-class mode: two tile shapes are enough (it is the accuracy path, not the fast one)
        if (a.Cout % 128) return launch_igemm_t<0, 64, 128, 1, 4, 2, true>(a, false, s);
        return launch_igemm_t<0, 128, 128, 2, 2, 2, true>(a, false, s);
    }
    if (tile == 0) tile = auto_tile(a);
    if (a.x2 && (!(tile & kWsBit) || tile == kTileC64 || tile == kTileXres || tile == kTileS2)) return hipErrorInvalidValue;     // two K sources: igemm_ws_kernel / gemm8p_kernel only
    if (tile == kTileG8 || tile == kTileG8N7) return launch_gemm8p<ET>(a, tile, s);
    const bool pers = (tile & kPersistBit) != 0;
    if (tile == kTileC64) return launch_conv3x3_c64<ET>(a, s);
    if (tile == kTileXres) return launch_conv3x3_xres<ET>(a, s);
    if (tile == kTileS2) return launch_conv3x3_s2<ET>(a, s);
    if (tile & kWsBit) {
        // <couts, pixels, consumer waves (couts x pixels), loader waves, LDS stages>
        switch (tile & (kPersistBit - 1)) {
            case 1: if (a.Cout % 128) return hipErrorInvalidValue; return launch_igemm_ws_t<ET, 128, 128, 2, 2, 4, 4>(a, s);
            case 3: if (a.Cout % 256) return hipErrorInvalidValue; return launch_igemm_ws_t<ET, 256, 128, 4, 2, 4, 3>(a, s);
            case 4: if (a.Cout % 128) return hipErrorInvalidValue; return launch_igemm_ws_t<ET, 128, 224, 2, 2, 4, 3>(a, s);
            case 8: if (a.Cout % 128) return hipErrorInvalidValue; return launch_igemm_ws_t<ET, 128, 224, 4, 2, 4, 3>(a, s);
            case 9: return launch_igemm_ws_t<ET, 64, 224, 2, 2, 4, 4>(a, s);
            // 208 = 13 x 16 pixels: M = 2^10 * 49 k gives 484 (layer3, 2 cout tiles) or 244 (layer4, 4 cout tiles) tiles of 13 blocks
            // where 224-wide tiles give 448 / 224 of 14: the same number of rounds on 256 CUs, 7 % less work per round
            case 10: if (a.Cout % 128) return hipErrorInvalidValue; return launch_igemm_ws_t<ET, 128, 208, 4, 1, 4, 3>(a, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (tile & (kPersistBit - 1)) {
        case 1: if (a.Cout % 128) return hipErrorInvalidValue; return launch_igemm_t<ET, 128, 128, 2, 2, 2>(a, pers, s);
        case 2: return launch_igemm_t<ET, 64, 128, 1, 4, 2>(a, pers, s);
        case 3: return launch_igemm_t<ET, 64, 256, 1, 4, 2>(a, pers, s);
        case 5: if (a.Cout % 128) return hipErrorInvalidValue; return launch_igemm_t<ET, 128, 64, 2, 2, 2>(a, pers, s);
        case 6: if (a.Cout % 256) return hipErrorInvalidValue; return launch_igemm_t<ET, 256, 128, 4, 2, 3>(a, pers, s);
        case 7: if (a.Cout % 128) return hipErrorInvalidValue; return launch_igemm_t<ET, 128, 256, 2, 4, 3>(a, pers, s);
        case 8: if (a.Cout % 128) return hipErrorInvalidValue; return launch_igemm_t<ET, 128, 128, 2, 4, 3>(a, pers, s);
        case 9: if (a.Cout % 256) return hipErrorInvalidValue; return launch_igemm_t<ET, 256, 256, 4, 2, 2>(a, pers, s);
        case 10: if (a.Cout % 256) return hipErrorInvalidValue; return launch_igemm_t<ET, 256, 256, 2, 4, 2>(a, pers, s);
        case 11: if (a.Cout % 256) return hipErrorInvalidValue; return launch_igemm_t<ET, 256, 208, 8, 1, 2>(a, pers, s);
        case 12: if (a.Cout % 256) return hipErrorInvalidValue; return launch_igemm_t<ET, 256, 224, 4, 2, 2>(a, pers, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_igemm(const ConvArgs& a, int tile, hipStream_t s, bool split = false) {
    return a.et == 1 ? launch_igemm_et<1>(a, tile, s, split) : launch_igemm_et<0>(a, tile, s, split);
}

// fp8 (e4m3) operands: role-specialised kernel only, element type 2 (kernels.h)
hipError_t launch_igemm_fp8(const ConvArgs& a, int tile, hipStream_t s) {
    if (tile == 0) {
        tile = kWsBit | 9;
        // 128x224 with 8 consumer waves is the fastest or within 2 % of it at 16 of the 18 layer2-4 shapes (scripts/time_conv_fp8.py)
        for (int c : {kWsBit | 8, kWsBit | 3, kWsBit | 1})
            if (tiles_of(a, c) >= 200) { tile = c; break; }
    }
    if (!(tile & kWsBit) || tile == kTileC64) return hipErrorInvalidValue;
    switch (tile & (kPersistBit - 1)) {
        case 1: if (a.Cout % 128) return hipErrorInvalidValue; return launch_igemm_ws_t<2, 128, 128, 2, 2, 4, 4>(a, s);
        case 3: if (a.Cout % 256) return hipErrorInvalidValue; return launch_igemm_ws_t<2, 256, 128, 4, 2, 4, 3>(a, s);
        case 4: if (a.Cout % 128) return hipErrorInvalidValue; return launch_igemm_ws_t<2, 128, 224, 2, 2, 4, 3>(a, s);
        case 8: if (a.Cout % 128) return hipErrorInvalidValue; return launch_igemm_ws_t<2, 128, 224, 4, 2, 4, 3>(a, s);
        case 9: return launch_igemm_ws_t<2, 64, 224, 2, 2, 4, 4>(a, s);
        default: return hipErrorInvalidValue;
    }
}

int fill_conv_args(ConvArgs& a, const void* x, int n, int h, int w, int cin, const void* wt, const float* bias,
                   const void* res, void* y, int cout, int ks, int stride, int pad, int relu, bool split = false, bool w2 = false) {
    if (!x || !wt || !bias || !y) return R50_ERR_INVALID;
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cin % 64 || cout <= 0 || cout % 64) return R50_ERR_INVALID;
    if (!(ks == 1 || ks == 3) || stride < 1 || pad < 0 || 2 * pad > ks - 1) return R50_ERR_INVALID;
    a.x = (const __bf16*)x; a.w = (const __bf16*)wt; a.bias = bias; a.res = (const __bf16*)res; a.y = (__bf16*)y;
    a.N = n; a.H = h; a.W = w; a.Cin = cin; a.Cout = cout;
    a.Ho = (h + 2 * pad - ks) / stride + 1;
    a.Wo = (w + 2 * pad - ks) / stride + 1;
    if (a.Ho <= 0 || a.Wo <= 0) return R50_ERR_INVALID;
    a.ks = ks; a.stride = stride; a.pad = pad; a.relu = relu;
    const long long M = (long long)n * a.Ho * a.Wo;
    // 32-bit index budget of the kernel: byte offsets of x / y stay below 2^32 elements*2
    if (M > (1ll << 30) || (long long)n * h * w * cin > (1ll << 31) - 1 || M * cout > (1ll << 31) - 1)
        return R50_ERR_INVALID;
    a.M = (int)M; a.HoWo = a.Ho * a.Wo;
    // split mode: X is [head(cin) | tail(cin)] per pixel, K per tap is 3*cin, Y is [head(cout) | tail(cout)]
    a.x_cstride = split ? 2 * cin : cin;
    a.y_cstride = split ? 2 * cout : cout;
    // bf16w2 mode: X plain, K per tap is 2*cin = [x | x] against [w_head | w_tail]
    const int kmul = split ? 3 : (w2 ? 2 : 1);
    a.x_wrap = split ? 2 * (cin / 64) : (w2 ? cin / 64 : (1 << 30));
    a.cin_chunks = kmul * (cin / 64); a.nk = ks * ks * a.cin_chunks; a.Ktot = ks * ks * kmul * cin;
    a.n_ctiles = 0; a.n_blocks = 0;
    // buffer descriptors of the kernel (kernels.h): every offset must stay below 2^31
    const long long x_bytes = (long long)n * h * w * a.x_cstride * 2;
    a.x_back = (pad * w + pad) * a.x_cstride * 2;
    if (x_bytes + a.x_back >= (1ll << 31) || (long long)cout * a.Ktot * 2 >= (1ll << 31) || M * a.y_cstride * 2 >= (1ll << 31))
        return R50_ERR_INVALID;
    a.x_records = (unsigned)(x_bytes + a.x_back);
    a.w_bytes = (unsigned)((long long)cout * a.Ktot * 2);
    a.y_bytes = (unsigned)(M * a.y_cstride * 2);
    a.div_howo = make_fast_div((unsigned)a.HoWo);
    a.div_wo = make_fast_div((unsigned)a.Wo);
    a.div_ctiles = FastDiv{0u, 0u};
    a.et = 0;
    a.oscale = 1.0f; a.rscale = 1.0f; a.q_inv = 0.0f;
    a.x2 = nullptr; a.H2 = 0; a.W2 = 0; a.stride2 = 1; a.x2_cstride = 0; a.cc1 = 1 << 30; a.x2_records = 0u;
#if defined(R50_STAMP)
    a.dbg = nullptr;
#endif
    return R50_OK;
}

// profiling brackets
void prof_begin(r50_handle* h, hipStream_t s, EvRec& r, int cls, double flops, double bytes, int layer = -1) {
    if (!h || !h->profile) return;
    r.layer = layer;
    auto get = [&]() {
        hipEvent_t e;
        if (!h->ev_free.empty()) { e = h->ev_free.back(); h->ev_free.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    };
    r.a = get(); r.b = get(); r.cls = cls; r.flops = flops; r.bytes = bytes;
    (void)hipEventRecord(r.a, s);
}
void prof_end(r50_handle* h, hipStream_t s, EvRec& r) {
    if (!h || !h->profile) return;
    (void)hipEventRecord(r.b, s);
    h->ev_pending.push_back(r);
}

// conv3 (64 -> 256) + identity + ReLU + next conv1 (256 -> c1) in one launch (kernels.h: bneck_tail_kernel).
// wd/bd non-null: `res` is the block INPUT (m,64) and the identity is the downsample conv computed in the kernel.
template <int ET, int C1, bool DS, int NT>
hipError_t launch_bneck_tail_t(const TailArgs& a, hipStream_t s) {
    const size_t lds = 256 * 128 * (DS ? 2 : 1) + (size_t)C1 * 512 + 256 * 4 * (DS ? 2 : 1) + (size_t)C1 * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bneck_tail_kernel<ET, C1, DS, NT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const long long tiles = ((long long)a.M + 15) / 16;
    const int grid = (int)std::min<long long>((tiles + NT / 64 - 1) / (NT / 64), (long long)g_num_cus);
    hipLaunchKernelGGL((bneck_tail_kernel<ET, C1, DS, NT>), dim3(grid), dim3(NT), lds, s, a);
    return hipGetLastError();
}

#ifndef TAIL_NT_DS
#define TAIL_NT_DS 512
#endif
#ifndef TAIL_NT_C128
#define TAIL_NT_C128 256
#endif
hipError_t launch_bneck_tail(const void* y2, long long m, const void* w3, const float* b3, const void* res, const void* wd,
                             const float* bd, void* out, const void* w1, int c1, const float* b1, void* y1n, hipStream_t s, int et = 0) {
    if (!y2 || !w3 || !b3 || !res || !out || !w1 || !b1 || !y1n || m <= 0 || m * 512 >= (1ll << 31)) return hipErrorInvalidValue;
    if ((c1 != 64 && c1 != 128) || ((wd == nullptr) != (bd == nullptr))) return hipErrorInvalidValue;
    TailArgs a;
    a.y2 = (const __bf16*)y2; a.w3 = (const __bf16*)w3; a.b3 = b3; a.res = (const __bf16*)res; a.out = (__bf16*)out;
    a.wd = (const __bf16*)wd; a.bd = bd;
    a.w1 = (const __bf16*)w1; a.b1 = b1; a.y1n = (__bf16*)y1n; a.M = (int)m;
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        g_num_cus = cu_budget(prop.multiProcessorCount);
    }
    // 4 waves per CU stream best (more waves lower the HBM rate); the downsample variant has 1.5x the MFMA work
    if (et == 1) {
        if (wd) return c1 == 64 ? launch_bneck_tail_t<1, 64, true, TAIL_NT_DS>(a, s) : launch_bneck_tail_t<1, 128, true, TAIL_NT_DS>(a, s);
        return c1 == 64 ? launch_bneck_tail_t<1, 64, false, 256>(a, s) : launch_bneck_tail_t<1, 128, false, TAIL_NT_C128>(a, s);
    }
    if (wd) return c1 == 64 ? launch_bneck_tail_t<0, 64, true, TAIL_NT_DS>(a, s) : launch_bneck_tail_t<0, 128, true, TAIL_NT_DS>(a, s);
    return c1 == 64 ? launch_bneck_tail_t<0, 64, false, 256>(a, s) : launch_bneck_tail_t<0, 128, false, TAIL_NT_C128>(a, s);
}

// layer2 shapes: conv3 (128 -> 512) + identity + ReLU + next conv1 (512 -> 128) (kernels.h: bneck_tail2_kernel)
hipError_t launch_bneck_tail2(const void* y2, long long m, const void* w3, const float* b3, const void* res, void* out,
                              const void* w1, const float* b1, void* y1n, hipStream_t s, int et = 0) {
    if (!y2 || !w3 || !b3 || !res || !out || !w1 || !b1 || !y1n || m <= 0 || m * 1024 >= (1ll << 31)) return hipErrorInvalidValue;
    Tail2Args a;
    a.y2 = (const __bf16*)y2; a.w3 = (const __bf16*)w3; a.b3 = b3; a.res = (const __bf16*)res; a.out = (__bf16*)out;
    a.w1 = (const __bf16*)w1; a.b1 = b1; a.y1n = (__bf16*)y1n; a.M = (int)m;
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        g_num_cus = cu_budget(prop.multiProcessorCount);
    }
    const long long steps = (m + 15) / 16;
    const int grid = (int)std::min<long long>(steps, (long long)g_num_cus);
    const size_t lds = 2 * 8 * 8 * 1024 + 512 * 4 + 2 * 4096;
    auto kern = et == 1 ? bneck_tail2_kernel<1> : bneck_tail2_kernel<0>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a);
    return hipGetLastError();
}

// layer3 shapes: conv3 (256 -> 1024) + identity + ReLU chained with the next conv1 (1024 -> 256) (kernels.h: bneck_tail3_kernel)
// `wp`: both weight matrices in the kernel's fragment-ordered stream (1 MB; pack_tail3_weights).
// Pixel tile: at most 112 rows; chosen so the tiles fill whole rounds of the chip (batch 256: M = 50,176 -> 98 pixels, 512 tiles).
constexpr size_t kTail3PackedBytes = 1u << 20;
#if defined(R50_STAMP)
unsigned long long* g_dbg = nullptr;         // diagnostic build: set through r50_debug_buffer()
#endif
hipError_t pack_tail3_weights(const void* w3, const void* w1, void* wp, hipStream_t s) {
    if (!w3 || !w1 || !wp) return hipErrorInvalidValue;
    hipLaunchKernelGGL(tail3_pack_kernel, dim3(65536 / 256), dim3(256), 0, s, (const __bf16*)w3, (const __bf16*)w1, (__bf16*)wp);
    return hipGetLastError();
}
// bneck_tail3p_kernel (two-group pipeline, 112 LDS rows per slot).  Rounds 2-3 also carried bneck_tail3_kernel (consumer + helper waves) and a 98-row
// form of the pipeline: both measured slower (profiles/r03_tail3p_variants.txt) and were removed in round 4 (git history has them).
int g_tail3_bp = 0;        // option "tail3_bp": real pixels per tile of the chained layer3 tail (0 = whole rounds of the chip / full tiles)
hipError_t launch_bneck_tail3(const void* y2, long long m, const void* wp, const float* b3, const void* res, void* out,
                              const float* b1, void* y1n, hipStream_t s, int et = 0, int bp_override = 0, bool no_next = false) {
    if (no_next) { b1 = b3; y1n = out; }       // conv3 + identity + ReLU only (group B copies out): b1 / y1n are not used
    if (bp_override == 0) bp_override = g_tail3_bp;
    if (!y2 || !wp || !b3 || !res || !out || !b1 || !y1n || m <= 0 || m * 2048 >= (1ll << 31)) return hipErrorInvalidValue;
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        g_num_cus = cu_budget(prop.multiProcessorCount);
    }
    Tail3Args a;
    a.y2 = (const __bf16*)y2; a.wp = (const __bf16*)wp; a.b3 = b3; a.res = (const __bf16*)res; a.out = (__bf16*)out;
    a.b1 = b1; a.y1n = (__bf16*)y1n; a.M = (int)m;
#if defined(R50_STAMP)
    a.dbg = g_dbg;
#endif
    constexpr int rows = 112;
    const long long rounds = ((m + rows - 1) / rows + g_num_cus - 1) / g_num_cus;
    long long bp = (m + rounds * g_num_cus - 1) / (rounds * g_num_cus);
    if (bp < 49) bp = 49;
    if (bp > rows) bp = rows;
    // more than one round of tiles: FULL tiles (batch 256: 448 tiles of 112 instead of 512 of 98 in 112 rows).  An eighth of the MFMA columns of a
    // 98-pixel tile multiply padding; the ragged last round that full tiles leave is filled by the other lane's launches (bench.py --lanes 2:
    // +0.9 % frames/s, two same-box pairs; neutral with one lane)
    if ((m + rows - 1) / rows > g_num_cus) bp = rows;
    if (bp_override >= 1 && bp_override <= rows) bp = bp_override;
    a.bp = (int)bp;
    a.n_tiles = (int)((m + bp - 1) / bp);
    const int grid = a.n_tiles < g_num_cus ? a.n_tiles : g_num_cus;
    const size_t ldsp = 8 * (size_t)rows * 128 + 256 * 4 + 1024 * 4;      // t2 (4 slots) + out_c (2 x 2) + b1 + b3
    void (*kp)(const Tail3Args);
    if (no_next) kp = et == 1 ? bneck_tail3p_kernel<1, 112, true> : bneck_tail3p_kernel<0, 112, true>;
    else kp = et == 1 ? bneck_tail3p_kernel<1, 112> : bneck_tail3p_kernel<0, 112>;
    hipError_t ep = hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
    if (ep != hipSuccess) return ep;
    hipLaunchKernelGGL(kp, dim3(grid), dim3(512), ldsp, s, a);
    return hipGetLastError();
}

// layer2.0 transition tail: conv3 + downsample (one conv over K = [t2 | x at stride 2]) + ReLU chained with the next block's conv1
// (kernels.h: bneck_catchain_kernel).  `wp`: both weight matrices in the kernel's fragment-ordered stream (512 KB; pack_catchain_weights).
constexpr size_t kCatChainPackedBytes = 1u << 19;
hipError_t pack_catchain_weights(const void* wcat, const void* w1, void* wp, hipStream_t s) {
    if (!wcat || !w1 || !wp) return hipErrorInvalidValue;
    hipLaunchKernelGGL(catchain_pack_kernel, dim3(32768 / 256), dim3(256), 0, s, (const __bf16*)wcat, (const __bf16*)w1, (__bf16*)wp);
    return hipGetLastError();
}
// x_sub: x is the compact (N, ow, ow, 256) tensor of the block input's even rows and columns (what bneck_block1_kernel writes with out_sub)
hipError_t launch_bneck_catchain(const void* t2, const void* x, int n, int ow, const void* wp, const float* bcat, void* out,
                                 const float* b1, void* y1n, hipStream_t s, int et = 0, bool x_sub = false) {
    if (!t2 || !x || !wp || !bcat || !out || !b1 || !y1n || n <= 0 || ow != 28) return hipErrorInvalidValue;
    const long long m = (long long)n * ow * ow;
    if (m * 1024 >= (1ll << 31) || (long long)n * 4 * ow * ow * 512 >= (1ll << 31)) return hipErrorInvalidValue;
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        g_num_cus = cu_budget(prop.multiProcessorCount);
    }
    CatChainArgs a;
    a.t2 = (const __bf16*)t2; a.x = (const __bf16*)x; a.wp = (const __bf16*)wp; a.b3 = bcat; a.out = (__bf16*)out; a.b1 = b1; a.y1n = (__bf16*)y1n;
    a.M = (int)m; a.x_bytes = (unsigned)((long long)n * (x_sub ? 1 : 4) * ow * ow * 512); a.x_sub = x_sub ? 1 : 0;
#if defined(R50_STAMP)
    a.dbg = g_dbg;
#endif
    // full 112-pixel tiles when there is more than one round of them (batch 256: 1792 tiles = 7 per CU), else spread over the chip
    const long long full = (m + 111) / 112;
    long long bp = 112;
    if (full <= g_num_cus) { bp = (m + g_num_cus - 1) / g_num_cus; if (bp < 16) bp = 16; if (bp > 112) bp = 112; }
    a.bp = (int)bp;
    a.n_tiles = (int)((m + bp - 1) / bp);
    const int grid = a.n_tiles < g_num_cus ? a.n_tiles : g_num_cus;
    const size_t lds = 10 * 112 * 128 + 128 * 4 + 512 * 4 + 16;        // operand slots + out_c + b1 + b3 + 4 sync words
    auto kern = et == 1 ? bneck_catchain_kernel<1, 28> : bneck_catchain_kernel<0, 28>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a);
    return hipGetLastError();
}

// layer2.1-.3 bottleneck body in one launch (kernels.h: bneck_block2_kernel): conv2 + conv3 + identity + ReLU [+ the next block's conv1]
hipError_t launch_bneck_block2(const void* t1, int n, const void* w2, const float* b2, const void* w3, const float* b3, const void* res,
                               void* out, const void* w1, const float* b1, void* y1n, hipStream_t s, int et = 0) {
    if (!t1 || !w2 || !b2 || !w3 || !b3 || !res || !out || n <= 0 || (long long)n * 784 * 1024 >= (1ll << 31)) return hipErrorInvalidValue;
    if ((w1 == nullptr) != (b1 == nullptr) || (w1 == nullptr) != (y1n == nullptr)) return hipErrorInvalidValue;
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        g_num_cus = cu_budget(prop.multiProcessorCount);
    }
    Block2Args a;
    a.t1 = (const __bf16*)t1; a.w2 = (const __bf16*)w2; a.b2 = b2; a.w3 = (const __bf16*)w3; a.b3 = b3; a.res = (const __bf16*)res;
    a.out = (__bf16*)out; a.w1 = (const __bf16*)w1; a.b1 = b1; a.y1n = (__bf16*)y1n; a.N = n; a.n_tiles = 4 * n;
    const int grid = a.n_tiles < g_num_cus ? a.n_tiles : g_num_cus;
    const size_t lds = (w1 ? 3 * 208 * 128 + 224 * 128 : 4 * 208 * 128) + 3 * 16384 + 768 * 4;       // 160,768 / 158,720 (kernels.h: LDS map)
    void (*kern)(const Block2Args);
    if (w1) kern = et == 1 ? bneck_block2_kernel<1, 128> : bneck_block2_kernel<0, 128>;
    else kern = et == 1 ? bneck_block2_kernel<1, 0> : bneck_block2_kernel<0, 0>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(768), lds, s, a);
    return hipGetLastError();
}

// layer1.1 / .2 bottleneck body in one launch (kernels.h: bneck_block1_kernel): conv2 + conv3 + identity + ReLU + the next conv1 (c1 = 64 or 128)
hipError_t launch_bneck_block1(const void* t1, int n, const void* w2, const float* b2, const void* w3, const float* b3, const void* res,
                               void* out, const void* w1, int c1, const float* b1, void* y1n, hipStream_t s, int et = 0,
                               const void* wd = nullptr, const float* bd = nullptr, bool out_sub = false) {
    if (!t1 || !w2 || !b2 || !w3 || !b3 || !res || !out || !w1 || !b1 || !y1n || n <= 0 || (long long)n * 3136 * 512 >= (1ll << 31)) return hipErrorInvalidValue;
    if (c1 != 64 && c1 != 128) return hipErrorInvalidValue;
    if ((wd == nullptr) != (bd == nullptr) || (wd && c1 != 64)) return hipErrorInvalidValue;       // downsample form: layer1.0 (next conv1 256 -> 64)
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        g_num_cus = cu_budget(prop.multiProcessorCount);
    }
    Block1Args a;
    a.t1 = (const __bf16*)t1; a.w2 = (const __bf16*)w2; a.b2 = b2; a.w3 = (const __bf16*)w3; a.b3 = b3; a.res = (const __bf16*)res;
    a.out = (__bf16*)out; a.w1 = (const __bf16*)w1; a.b1 = b1; a.y1n = (__bf16*)y1n; a.N = n; a.n_tiles = 14 * n;
    a.wd = (const __bf16*)wd; a.bd = bd; a.out_sub = out_sub ? 1 : 0;
    const int grid = a.n_tiles < g_num_cus ? a.n_tiles : g_num_cus;
    constexpr size_t lds = 11 * 32 * 128 + 3 * 224 * 128 + 3 * 8192 + (448 + 256) * 4;       // 158,464 (kernels.h: LDS map; bd behind b1)
    void (*kern)(const Block1Args);
    if (wd) kern = et == 1 ? bneck_block1_kernel<1, 64, true> : bneck_block1_kernel<0, 64, true>;
    else if (c1 == 64) kern = et == 1 ? bneck_block1_kernel<1, 64> : bneck_block1_kernel<0, 64>;
    else kern = et == 1 ? bneck_block1_kernel<1, 128> : bneck_block1_kernel<0, 128>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(768), lds, s, a);
    return hipGetLastError();
}

// q_inv > 0: write the output as e4m3 = fp8(16-bit result * q_inv) (role-specialised tiles only; *q_done reports whether that happened)
int run_conv(r50_handle* h, const ConvLayer& L, const __bf16* x, int n, int hh, int ww, const __bf16* res,
             __bf16* y, int relu, hipStream_t s, int* ho, int* wo, float q_inv = 0.f, bool* q_done = nullptr) {
    ConvArgs a;
    const bool split = (h->precision == R50_PREC_FP32X);
    const bool w2 = (h->precision == R50_PREC_BF16W2);
    int rc = fill_conv_args(a, x, n, hh, ww, L.cin, L.w, L.bias, res, y, L.cout, L.ks, L.stride, L.pad, relu, split, w2);
    if (rc) return fail(h, rc, "conv args invalid for " + L.conv_key);
    a.et = (h->precision == R50_PREC_FP16) ? 1 : 0;
    EvRec r{};
    const double flops = 2.0 * a.M * (double)a.Cout * L.ks * L.ks * L.cin;       // algorithmic (not the 3x of split mode)
    const double bytes = 2.0 * ((double)n * hh * ww * L.cin + (double)a.M * a.Cout * (res ? 2 : 1) + (double)a.Cout * L.ks * L.ks * L.cin);
    int tile = h->tile_override;
    if (q_done) *q_done = false;
    if (q_inv > 0.f && !split && !w2 && tile == 0) {
        const int t = auto_tile(a);
        if ((t & kWsBit) && t != kTileC64 && t != kTileXres) {      // the quantising epilogue lives in igemm_ws_kernel
            tile = t; a.q_inv = q_inv; a.y_bytes = (unsigned)((long long)a.M * a.y_cstride);
            if (q_done) *q_done = true;
        }
    }
    prof_begin(h, s, r, PC_IGEMM, flops, bytes, (int)(&L - &h->convs[0]));
    hipError_t e = launch_igemm(a, tile, s, split);
    prof_end(h, s, r);
    if (e != hipSuccess) return fail(h, R50_ERR_HIP, "igemm launch (" + L.conv_key + "): " + hipGetErrorString(e));
    *ho = a.Ho; *wo = a.Wo;
    return R50_OK;
}

// 1x1 conv over two K sources: y = act([W1 | W2] . [x1 ; x2 at stride2] + bias).  x1 (n,h,w,c1) at the output resolution,
// x2 (n,h2,w2,c2) sampled at (ho*stride2, wo*stride2); wcat (cout, c1 + c2) K-major.  Role-specialised kernel only.
int fill_conv_args_cat(ConvArgs& a, const void* x1, int n, int h, int w, int c1, const void* x2, int h2, int w2, int c2, int stride2,
                       const void* wcat, const float* bias, void* y, int cout, int relu) {
    if (!x2 || c2 <= 0 || c2 % 64 || stride2 < 1 || h2 < 1 || w2 < 1) return R50_ERR_INVALID;
    if ((h2 - 1) / stride2 + 1 != h || (w2 - 1) / stride2 + 1 != w) return R50_ERR_INVALID;
    int rc = fill_conv_args(a, x1, n, h, w, c1, wcat, bias, nullptr, y, cout, 1, 1, 0, relu);
    if (rc) return rc;
    const long long x2_bytes = (long long)n * h2 * w2 * c2 * 2;
    if (x2_bytes >= (1ll << 31) || (long long)cout * (c1 + c2) * 2 >= (1ll << 31)) return R50_ERR_INVALID;
    a.Cin = c1 + c2;
    a.cin_chunks = (c1 + c2) / 64; a.nk = a.cin_chunks; a.Ktot = c1 + c2;
    a.w_bytes = (unsigned)((long long)cout * a.Ktot * 2);
    a.x2 = (const __bf16*)x2; a.H2 = h2; a.W2 = w2; a.stride2 = stride2; a.x2_cstride = c2; a.cc1 = c1 / 64;
    a.x2_records = (unsigned)x2_bytes;
    return R50_OK;
}
int cat_tile(const ConvArgs& a, int tile) {          // tile id for a two-source conv: role-specialised variants only
    if (tile == 0) {
        if (const int t8 = g8_tile(a)) return t8;
        if (a.N >= 48) {                               // measured at batch 256 (scripts/time_cat.py, profiles/r01_time_cat.txt)
            if (a.Cout % 256 == 0) return kWsBit | 3;      // 256x128, 8 consumer waves: fastest of the five at all three shapes
        }
        for (int c : {kWsBit | 8, kWsBit | 1, kWsBit | 9})
            if (tiles_of(a, c) >= 200) return c;
        return kWsBit | 9;
    }
    return (tile & kWsBit) ? tile : -1;
}
int run_conv_cat(r50_handle* h, int si, const ConvLayer& c3, const ConvLayer& cd, const __bf16* t2, const __bf16* xin, int n, int h2, int w2,
                 int hin, int win, __bf16* y, hipStream_t s) {
    ConvArgs a;
    int rc = fill_conv_args_cat(a, t2, n, h2, w2, c3.cin, xin, hin, win, cd.cin, cd.stride, h->cat_w[si], h->cat_bias[si], y, c3.cout, 1);
    if (rc) return fail(h, rc, "two-source conv args invalid for " + c3.conv_key);
    a.et = (h->precision == R50_PREC_FP16) ? 1 : 0;
    EvRec r{};
    const double flops = 2.0 * a.M * (double)a.Cout * a.Ktot;
    const double bytes = 2.0 * ((double)a.M * (c3.cin + cd.cin) + (double)a.M * a.Cout + (double)a.Cout * a.Ktot);
    prof_begin(h, s, r, PC_IGEMM, flops, bytes, (int)(&c3 - &h->convs[0]));
    hipError_t e = launch_igemm(a, cat_tile(a, 0), s, false);
    prof_end(h, s, r);
    if (e != hipSuccess) return fail(h, R50_ERR_HIP, "igemm launch (" + c3.conv_key + " + downsample): " + hipGetErrorString(e));
    return R50_OK;
}

template <typename TIN>
hipError_t launch_stem_pack(const TIN* x, void* xp, int n, hipStream_t s, int et = 0) {
    const long long total = (long long)n * STEM_HP * STEM_WP;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (et == 1) hipLaunchKernelGGL((stem_pack_kernel<1, TIN>), dim3((unsigned)blocks), dim3(256), 0, s, x, (u32x2*)xp, n);
    else hipLaunchKernelGGL((stem_pack_kernel<0, TIN>), dim3((unsigned)blocks), dim3(256), 0, s, x, (u32x2*)xp, n);
    return hipGetLastError();
}
hipError_t launch_stem_conv(const void* xp, const void* wpk, const float* bias, void* y, int n, hipStream_t s, int et = 0) {
    auto kern = et == 1 ? stem_conv_kernel<1> : stem_conv_kernel<0>;
    hipLaunchKernelGGL(kern, dim3(n * (112 / STEM_ROWS_PER_WG)), dim3(256), STEM_LDS_BYTES, s,
                       (const char*)xp, (const char*)wpk, bias, (__bf16*)y);
    return hipGetLastError();
}
// fused stem (stem_fused3_kernel): a workgroup walks a strip of G consecutive pooled-row pairs of one image.  Strip length 0 = chosen from the
// batch, 1/2/4/7/14/28 = that many pairs per strip: r50_set_option("stem_strip") or R50_STEM_STRIP in the environment (A/B).
// (Removed in round 4: the per-pair kernel of round 1 and the strip kernel of rounds 2-3, whose phases ran in turn.)
static int g_stem_strip = [] { const char* v = std::getenv("R50_STEM_STRIP"); const int g = v ? std::atoi(v) : 0; return g > 0 && 28 % g == 0 ? g : 0; }();
bool stem_strip_enabled() { return true; }
// c1_w / c1_bias / y1 non-null: layer1.0.conv1 (+ bias + ReLU) of the pooled output into y1 in the same launch
template <typename TIN>
hipError_t launch_stem_fused(const TIN* x, const void* wpk, const float* bias, void* y, int n, hipStream_t s,
                             const float* u8_table, int et = 0, const void* c1_w = nullptr, const float* c1_bias = nullptr,
                             void* y1 = nullptr) {
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        g_num_cus = cu_budget(prop.multiProcessorCount);
    }
    // G = the longest strip that still gives the chip ~200 workgroups (28 = the whole image from batch 200 up)
    int G = 1;
    if (g_stem_strip > 0) G = g_stem_strip;
    else
        for (int cand : {28, 14, 7, 4, 2})
            if (n * (28 / cand) >= 200) { G = cand; break; }
    const bool c1 = c1_w && c1_bias && y1;
    const int strips = n * (28 / G);
    const int grid = strips < g_num_cus ? strips : g_num_cus;
    // 16-B (fp32) / 4-B (uint8) loads of 4 pixels: image rows start at multiples of 896 / 224 bytes from the frame pointer
    if (reinterpret_cast<uintptr_t>(x) & (sizeof(TIN) == 1 ? 3u : 15u)) return hipErrorInvalidValue;
    auto kern = c1 ? (et == 1 ? stem_fused3_kernel<1, TIN, true> : stem_fused3_kernel<0, TIN, true>)
                   : (et == 1 ? stem_fused3_kernel<1, TIN, false> : stem_fused3_kernel<0, TIN, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SF3_LDS_BYTES);
    if (e != hipSuccess) return e;
#if defined(R50_STAMP)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SF3_THREADS), SF3_LDS_BYTES, s, x, (const char*)wpk, bias, (__bf16*)y, strips, G, u8_table,
                       (const __bf16*)c1_w, c1_bias, (__bf16*)y1, g_dbg);
#else
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SF3_THREADS), SF3_LDS_BYTES, s, x, (const char*)wpk, bias, (__bf16*)y, strips, G, u8_table,
                       (const __bf16*)c1_w, c1_bias, (__bf16*)y1);
#endif
    return hipGetLastError();
}
template <typename TIN>
hipError_t launch_stem_split(const TIN* x, char* xp_head, char* xp_tail, const char* w_head, const char* w_tail,
                             const float* bias, void* y, int n, hipStream_t s) {
    const long long total = (long long)n * STEM_HP * STEM_WP;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(stem_pack_split_kernel<TIN>, dim3((unsigned)blocks), dim3(256), 0, s, x, (u32x2*)xp_head, (u32x2*)xp_tail, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_conv_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            STEM_SPLIT_LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(stem_conv_split_kernel, dim3(n * (112 / STEM_ROWS_PER_WG)), dim3(256), STEM_SPLIT_LDS_BYTES, s,
                       (const char*)xp_head, (const char*)xp_tail, w_head, w_tail, bias, (__bf16*)y);
    return hipGetLastError();
}
hipError_t launch_maxpool_split(const void* x, void* y, int n, int h, int w, int c, hipStream_t s) {
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long long total = (long long)n * ho * wo * (c / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(maxpool3x3s2_split_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const __bf16*)x, (__bf16*)y, n, h, w,
                       c, ho, wo);
    return hipGetLastError();
}
hipError_t launch_avgpool_split(const void* x, float* y, int n, int hw, int c, hipStream_t s) {
    const int total = n * (c / 8);
    hipLaunchKernelGGL(avgpool_split_kernel, dim3((total + 255) / 256), dim3(256), 0, s, (const __bf16*)x, y, n, hw, c,
                       1.0f / (float)hw);
    return hipGetLastError();
}
hipError_t launch_maxpool(const void* x, void* y, int n, int h, int w, int c, hipStream_t s, int et = 0) {
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long long total = (long long)n * ho * wo * (c / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    auto kern = et == 1 ? maxpool3x3s2_kernel<1> : maxpool3x3s2_kernel<0>;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, s, (const __bf16*)x, (__bf16*)y,
                       n, h, w, c, ho, wo);
    return hipGetLastError();
}
hipError_t launch_avgpool(const void* x, float* y, int n, int hw, int c, hipStream_t s, int et = 0) {
    const int total = n * (c / 8);
    auto kern = et == 1 ? avgpool_kernel<1> : avgpool_kernel<0>;
    hipLaunchKernelGGL(kern, dim3((total + 255) / 256), dim3(256), 0, s, (const __bf16*)x, y, n, hw, c,
                       1.0f / (float)hw);
    return hipGetLastError();
}

// ---- R50_PREC_FP8: layer2-4 on e4m3 tensors ----------------------------------------------------------------------------
int run_conv_fp8(r50_handle* h, const ConvLayer& L, const void* x, int n, int hh, int ww, const void* res, void* y, int relu, float sx,
                 float sr, float sy, hipStream_t s, int* ho, int* wo) {
    ConvArgs a;
    int rc = fill_conv_args(a, x, n, hh, ww, L.cin / 2, L.w, L.bias_scaled, res, y, L.cout, L.ks, L.stride, L.pad, relu);
    if (rc) return fail(h, rc, "fp8 conv args invalid for " + L.conv_key);
    a.y_bytes = (unsigned)((long long)a.M * L.cout);
    a.et = 2; a.oscale = sx * L.wscale / sy; a.rscale = sr / sy;
    EvRec r{};
    const double flops = 2.0 * a.M * (double)a.Cout * L.ks * L.ks * L.cin;
    const double bytes = (double)n * hh * ww * L.cin + (double)a.M * a.Cout * (res ? 2 : 1) + (double)a.Cout * L.ks * L.ks * L.cin;
    prof_begin(h, s, r, PC_IGEMM, flops, bytes, (int)(&L - &h->convs[0]));
    hipError_t e = launch_igemm_fp8(a, 0, s);
    prof_end(h, s, r);
    if (e != hipSuccess) return fail(h, R50_ERR_HIP, "fp8 igemm launch (" + L.conv_key + "): " + hipGetErrorString(e));
    *ho = a.Ho; *wo = a.Wo;
    return R50_OK;
}

// buf[cur] holds layer1's output (n,56,56,256) as 16-bit elements: quantise it, run layer2-4 in fp8, pool.
int run_fp8_part(r50_handle* h, __bf16* const* buf, int cur, int n, float* out, hipStream_t s, bool already_fp8 = false) {
    if ((int)h->fp8_scales.size() != R50_FP8_NUM_SCALES)
        return fail(h, R50_ERR_STATE, "fp8 mode: call r50_set_fp8_scales (activation scales) before the first forward");
    const float* sc = h->fp8_scales.data();
    int k = 0;
    float s_in = sc[k++];
    EvRec r{};
    int nxt = (cur + 1) % 5;
    if (!already_fp8) {
        const long long n4 = (long long)n * 56 * 56 * 256 / 4;
        prof_begin(h, s, r, PC_MAXPOOL, 0, (double)n4 * 12);
        hipLaunchKernelGGL(quant_to_fp8_kernel<0>, dim3((unsigned)std::min<long long>((n4 + 255) / 256, 256 * 64)), dim3(256), 0, s,
                           (const unsigned short*)buf[cur], (unsigned*)buf[nxt], n4, 1.0f / s_in);
        prof_end(h, s, r);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(h, R50_ERR_HIP, std::string("quant_to_fp8: ") + hipGetErrorString(e));
        cur = nxt;
    }
    int hh = 56, ww = 56;
    size_t li = kFp8FirstConv;
    for (int si = 1; si < 4; ++si)
        for (int b = 0; b < kStages[si][1]; ++b) {
            int fr[4], nf = 0;
            for (int i = 0; i < 5; ++i)
                if (i != cur) fr[nf++] = i;
            const ConvLayer &c1 = h->convs[li], &c2 = h->convs[li + 1], &c3 = h->convs[li + 2];
            const float s_t1 = sc[k++], s_t2 = sc[k++];
            int h1, w1, h2, w2, h3, w3, rc;
            if ((rc = run_conv_fp8(h, c1, buf[cur], n, hh, ww, nullptr, buf[fr[0]], 1, s_in, 1.f, s_t1, s, &h1, &w1))) return rc;
            if ((rc = run_conv_fp8(h, c2, buf[fr[0]], n, h1, w1, nullptr, buf[fr[1]], 1, s_t1, 1.f, s_t2, s, &h2, &w2))) return rc;
            const void* idn = buf[cur];
            float s_idn = s_in;
            const bool cat_ds = (b == 0) && h->fuse_ds_cat && h->cat_w[si] && h->cat_acc_scale[si] > 0.f;
            if (b == 0) {
                const float s_ds = sc[k++];        // unused by the two-source form: the downsample tensor is never formed
                if (!cat_ds) {
                    int hd, wd;
                    if ((rc = run_conv_fp8(h, h->convs[li + 3], buf[cur], n, hh, ww, nullptr, buf[fr[2]], 0, s_in, 1.f, s_ds, s, &hd, &wd))) return rc;
                    idn = buf[fr[2]]; s_idn = s_ds;
                }
            }
            const float s_out = sc[k++];
            if (cat_ds) {
                const ConvLayer& cd = h->convs[li + 3];
                ConvArgs a;
                rc = fill_conv_args_cat(a, buf[fr[1]], n, h2, w2, c3.cin / 2, buf[cur], hh, ww, cd.cin / 2, cd.stride, h->cat_w[si], h->cat_bias[si],
                                        buf[fr[3]], c3.cout, 1);
                if (rc) return fail(h, rc, "fp8 two-source conv args invalid for " + c3.conv_key);
                a.y_bytes = (unsigned)((long long)a.M * c3.cout);
                a.et = 2; a.oscale = h->cat_acc_scale[si] / s_out; a.rscale = 0.f;
                const double flops = 2.0 * a.M * (double)a.Cout * (c3.cin + cd.cin);
                prof_begin(h, s, r, PC_IGEMM, flops, (double)a.M * (c3.cin + cd.cin + a.Cout) + (double)a.Cout * (c3.cin + cd.cin), (int)(&c3 - &h->convs[0]));
                hipError_t e = launch_igemm_fp8(a, 0, s);
                prof_end(h, s, r);
                if (e != hipSuccess) return fail(h, R50_ERR_HIP, "fp8 two-source igemm launch (" + c3.conv_key + "): " + hipGetErrorString(e));
                h3 = h2; w3 = w2;
            } else if ((rc = run_conv_fp8(h, c3, buf[fr[1]], n, h2, w2, idn, buf[fr[3]], 1, s_t2, s_idn, s_out, s, &h3, &w3))) return rc;
            cur = fr[3]; hh = h3; ww = w3; s_in = s_out;
            li += (b == 0) ? 4 : 3;
        }
    prof_begin(h, s, r, PC_AVGPOOL, 0, (double)n * (hh * ww * 2048.0 + 2048.0 * 4));
    hipLaunchKernelGGL(avgpool_fp8_kernel, dim3((n * 256 + 255) / 256), dim3(256), 0, s, (const unsigned char*)buf[cur], out, n, hh * ww, 2048,
                       s_in / (float)(hh * ww));
    prof_end(h, s, r);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, R50_ERR_HIP, std::string("avgpool_fp8: ") + hipGetErrorString(e));
    return R50_OK;
}

// Runs `n` frames (n <= max_batch) through the stack.  If `tap` is non-null, stops once the named
// activation is available and reports it through tap_ptr / dims.
template <typename TIN>
int run_stack(r50_handle* h, const TIN* x, int n, float* out, hipStream_t s, const char* tap,
              const __bf16** tap_ptr, int64_t dims[4], int slot0 = 0) {
    // slot0: first frame slot of the workspace this call may use (concurrent calls on different streams
    // work on disjoint frame ranges of the same buffers)
    const bool split = (h->precision == R50_PREC_FP32X);
    const int et = (h->precision == R50_PREC_FP16) ? 1 : 0;      // element type of weights / activations (kernels.h)
    const int cmul = split ? 2 : 1;            // channels per pixel multiplier of every activation tensor
    __bf16* buf[5];
    for (int i = 0; i < 5; ++i) buf[i] = h->buf[i] + (size_t)slot0 * 112 * 112 * 64 * cmul;
    char* stem_xp = h->stem_xp + (size_t)slot0 * STEM_HP * STEM_WP * 8 * cmul;
    auto hit = [&](const std::string& name, const __bf16* p, int hh, int ww, int c) {
        if (tap && name == tap) {
            *tap_ptr = p; dims[0] = n; dims[1] = hh; dims[2] = ww; dims[3] = c * cmul;
            return true;
        }
        return false;
    };
    EvRec r{};
    hipError_t e;
    bool stem_c1 = false;       // layer1.0.conv1 already computed by the stem kernel (into buf[2])
    if (split) {
        char* xp_tail = stem_xp + (size_t)n * STEM_HP * STEM_WP * 8;
        prof_begin(h, s, r, PC_STEM_CONV, 2.0 * n * 112 * 112 * 64 * 147.0, (double)n * (3.0 * 224 * 224 * 4 + 112.0 * 112 * 128 * 2));
        e = launch_stem_split(x, stem_xp, xp_tail, h->stem_w, h->stem_w + STEM_W_BYTES, h->convs[0].bias, buf[0], n, s);
        prof_end(h, s, r);
        if (e != hipSuccess) return fail(h, R50_ERR_HIP, std::string("stem (split): ") + hipGetErrorString(e));
    } else if (h->fused_stem && !(tap && std::string(tap) == "stem")) {
        // conv1 + bn1 + relu + maxpool in one kernel: frame in, (n,56,56,64) out
        // ... and, in the strip version, layer1.0.conv1 (1x1, 64 -> 64) of the pooled rows into buf[2] while they are still in LDS
        const ConvLayer& l1c1 = h->convs[1];
        stem_c1 = h->fuse_stem_c1 && stem_strip_enabled() && (h->precision == R50_PREC_BF16 || h->precision == R50_PREC_FP16 || h->precision == R50_PREC_FP8) &&
                  h->tile_override == 0 && l1c1.ks == 1 && l1c1.stride == 1 && l1c1.cin == 64 && l1c1.cout == 64;
        prof_begin(h, s, r, PC_STEM_CONV, 2.0 * n * 112 * 112 * 64 * 147.0 + (stem_c1 ? 2.0 * n * 56 * 56 * 64 * 64 : 0.0),
                   (double)n * (3.0 * 224 * 224 * 4 + 56.0 * 56 * 64 * 2 * (stem_c1 ? 2 : 1)));
        e = launch_stem_fused(x, h->stem_w, h->convs[0].bias, buf[1], n, s, h->u8_table, et, stem_c1 ? l1c1.w : nullptr,
                              stem_c1 ? l1c1.bias : nullptr, stem_c1 ? buf[2] : nullptr);
        prof_end(h, s, r);
        if (e != hipSuccess) return fail(h, R50_ERR_HIP, std::string("stem_fused: ") + hipGetErrorString(e));
        goto after_pool;
    } else {
        prof_begin(h, s, r, PC_STEM_PACK, 0, (double)n * (3.0 * 224 * 224 * 4 + (double)STEM_HP * STEM_WP * 8));
        e = launch_stem_pack(x, stem_xp, n, s, et);
        prof_end(h, s, r);
        if (e != hipSuccess) return fail(h, R50_ERR_HIP, std::string("stem_pack: ") + hipGetErrorString(e));

        prof_begin(h, s, r, PC_STEM_CONV, 2.0 * n * 112 * 112 * 64 * 147.0,
                   (double)n * ((double)STEM_HP * STEM_WP * 8 + 112.0 * 112 * 64 * 2));
        e = launch_stem_conv(stem_xp, h->stem_w, h->convs[0].bias, buf[0], n, s, et);
        prof_end(h, s, r);
        if (e != hipSuccess) return fail(h, R50_ERR_HIP, std::string("stem_conv: ") + hipGetErrorString(e));
    }
    if (hit("stem", buf[0], 112, 112, 64)) return R50_OK;

    prof_begin(h, s, r, PC_MAXPOOL, 0, (double)n * (112.0 * 112 + 56.0 * 56) * 64 * 2 * cmul);
    e = split ? launch_maxpool_split(buf[0], buf[1], n, 112, 112, 64, s) : launch_maxpool(buf[0], buf[1], n, 112, 112, 64, s, et);
    prof_end(h, s, r);
    if (e != hipSuccess) return fail(h, R50_ERR_HIP, std::string("maxpool: ") + hipGetErrorString(e));
after_pool:
    if (hit("pool", buf[1], 56, 56, 64)) return R50_OK;

    int cur = 1, hh = 56, ww = 56;
    int pre_t1 = stem_c1 ? 2 : -1;      // buffer that already holds the coming block's conv1 output (fused tail / stem), or -1
    bool layer1_out_fp8 = false;        // fp8 mode: layer1's output was written as e4m3 by its last conv's epilogue
    bool cur_sub = false;               // buf[cur] holds only the even rows / columns of the block output (layer1.2 with option "sub_out")
    size_t li = 1;
    for (int si = 0; si < 4; ++si) {
        const int blocks = kStages[si][1];
        if (si == 1 && h->precision == R50_PREC_FP8) {
            if (tap) return fail(h, R50_ERR_INVALID, "fp8 mode: activations beyond layer1 are e4m3 tensors, taps stop at layer1.2");
            return run_fp8_part(h, buf, cur, n, out, s, layer1_out_fp8);
        }
        for (int b = 0; b < blocks; ++b) {
            int fr[4], nf = 0;
            if (pre_t1 >= 0) fr[nf++] = pre_t1;          // conv1 of this block was computed by the previous block's fused tail
            for (int i = 0; i < 5; ++i)
                if (i != cur && i != pre_t1) fr[nf++] = i;
            const bool have_t1 = (pre_t1 >= 0);
            pre_t1 = -1;
            if (cur_sub && !(si == 1 && b == 0)) return fail(h, R50_ERR_STATE, "internal: a sub-sampled block output reached a block that reads the full tensor");
            const std::string p = "layer" + std::to_string(si + 1) + "." + std::to_string(b);
            const ConvLayer& c1 = h->convs[li];
            const ConvLayer& c2 = h->convs[li + 1];
            const ConvLayer& c3 = h->convs[li + 2];
            int h1, w1, h2, w2, h3, w3;
            // The downsample branch of a stage's first block only depends on the block input: it runs on a side
            // stream beside conv1 -> conv2 (fills the partially occupied last round of those launches).  Not while
            // profiling (event brackets of two streams would interleave) and not when a tap is requested.
            // layer1: conv3 + identity + ReLU and the next block's conv1 share one pass over the pixels; in the first
            // block the identity (downsample conv of the block input) is computed inside that kernel as well
            const size_t li_next = li + ((b == 0) ? 4 : 3);
            const ConvLayer* nx = (li_next < h->convs.size()) ? &h->convs[li_next] : nullptr;
            const bool nx_is_fp8 = (h->precision == R50_PREC_FP8 && li_next >= kFp8FirstConv);     // the next conv1's weights are e4m3 bytes
            const bool fuse_ok = !split && !nx_is_fp8 && (h->precision == R50_PREC_BF16 || h->precision == R50_PREC_FP16 || h->precision == R50_PREC_FP8) && h->fuse_tail && h->tile_override == 0 && nx && c3.ks == 1 && c3.stride == 1 &&
                                 nx->ks == 1 && nx->stride == 1;
            const bool fuse2 = fuse_ok && c3.cin == 128 && c3.cout == 512 && nx->cin == 512 && nx->cout == 128;   // layer2 shapes
            const bool fuse3 = fuse_ok && h->fuse_tail3 && b > 0 && b < 8 && h->tail3_wp[b] && c3.cin == 256 && c3.cout == 1024 && nx->cin == 1024 && nx->cout == 256;   // layer3 shapes (plain identity)
            const ConvLayer* cdp = (b == 0) ? &h->convs[li + 3] : nullptr;
            // layer2.0 / 3.0 / 4.0: conv3 + downsample + add + ReLU as ONE 1x1 conv over K = [t2 | block input at the block's stride]
            // against [W3 | Wd]: the downsample tensor is never written or read back, and there is one launch instead of two.
            // It takes precedence over the layer2 tail fusion in layer2.0 (whose identity would be that downsample tensor).
            const bool cat_ds = (b == 0) && si >= 1 && !split && h->fuse_ds_cat && h->cat_w[si] && h->tile_override == 0 &&
                                cdp && cdp->ks == 1 && c3.ks == 1 && !(tap && p + ".ds" == tap);
            const bool fuse = !cat_ds && ((fuse_ok && c3.cin == 64 && c3.cout == 256 && nx->cin == 256 && (nx->cout == 64 || nx->cout == 128)) ||
                                          fuse2 || fuse3);
            const bool fuse_ds = fuse && cdp && cdp->ks == 1 && cdp->stride == 1 && cdp->cin == 64 && cdp->cout == 256 &&
                                 !(tap && p + ".ds" == tap);
            const bool ds_side = (b == 0) && h->overlap_ds && !h->profile && !tap && !fuse_ds && !cat_ds;
            hipStream_t sd = s;
            const __bf16* idn = buf[cur];
            int rc;
            if (ds_side) {
                if (!h->ds_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->ds_stream, hipStreamNonBlocking));
                if (!h->ev_ds_fork) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_ds_fork, hipEventDisableTiming));
                if (!h->ev_ds_join) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_ds_join, hipEventDisableTiming));
                sd = h->ds_stream;
                HIP_TRY(h, hipEventRecord(h->ev_ds_fork, s));          // block input complete
                HIP_TRY(h, hipStreamWaitEvent(sd, h->ev_ds_fork, 0));
                const ConvLayer& cd = h->convs[li + 3];
                int hd, wd;
                rc = run_conv(h, cd, buf[cur], n, hh, ww, nullptr, buf[fr[2]], 0, sd, &hd, &wd);
                if (rc) return rc;
                HIP_TRY(h, hipEventRecord(h->ev_ds_join, sd));
                idn = buf[fr[2]];
            }
            // layer2.1-.3: the bottleneck body from conv2 on is ONE launch (kernels.h: bneck_block2_kernel) -- t2 never leaves the CU, and
            // in .1 / .2 the next block's conv1 is chained through LDS as in the fused tails.  Same bits as the launches it replaces.
            const bool block2 = h->fuse_block2 && h->fuse_tail && !split && !tap && si == 1 && b > 0 && h->tile_override == 0 && !ds_side &&
                                (h->precision == R50_PREC_BF16 || h->precision == R50_PREC_FP16) && hh == 28 && ww == 28 && c1.cout == 128 &&
                                c2.ks == 3 && c2.stride == 1 && c2.pad == 1 && c2.cin == 128 && c2.cout == 128 && c3.ks == 1 && c3.cin == 128 && c3.cout == 512;
            // layer1.1 (and, option "fuse_block1" = 2, layer1.2): the same for the 56x56 / 64-channel body (kernels.h: bneck_block1_kernel)
            const bool nx64 = nx && nx->ks == 1 && nx->stride == 1 && nx->cin == 256 && (nx->cout == 64 || (nx->cout == 128 && h->fuse_block1 >= 2));
            const bool block1 = h->fuse_block1 && h->fuse_tail && !split && !tap && si == 0 && b > 0 && h->tile_override == 0 && nx64 &&
                                (h->precision == R50_PREC_BF16 || h->precision == R50_PREC_FP16 || (h->precision == R50_PREC_FP8 && nx->cout == 64)) &&
                                hh == 56 && ww == 56 && c1.cout == 64 && c2.ks == 3 && c2.stride == 1 && c2.pad == 1 && c2.cin == 64 && c2.cout == 64 &&
                                c3.ks == 1 && c3.cin == 64 && c3.cout == 256;
            // layer1.0 (option "fuse_block1" = 3, the default since round 3's second session): the same body kernel with the downsample conv
            // of the block input as the identity (bneck_block1_kernel<.., DS>): conv3x3_c64 + fused tail (t2 written and read back) become one launch
            const bool block1_ds = h->fuse_block1 >= 3 && h->fuse_tail && !split && !tap && si == 0 && b == 0 && h->tile_override == 0 && cdp && nx &&
                                   (h->precision == R50_PREC_BF16 || h->precision == R50_PREC_FP16 || h->precision == R50_PREC_FP8) && hh == 56 && ww == 56 &&
                                   c1.cout == 64 && c2.ks == 3 && c2.stride == 1 && c2.pad == 1 && c2.cin == 64 && c2.cout == 64 && c3.ks == 1 && c3.cin == 64 &&
                                   c3.cout == 256 && cdp->ks == 1 && cdp->stride == 1 && cdp->cin == 64 && cdp->cout == 256 && nx->ks == 1 && nx->stride == 1 &&
                                   nx->cin == 256 && nx->cout == 64 && !nx_is_fp8;
            if (block1_ds) {
                if (!have_t1) {
                    rc = run_conv(h, c1, buf[cur], n, hh, ww, nullptr, buf[fr[0]], 1, s, &h1, &w1);
                    if (rc) return rc;
                }
                const double m = (double)n * 3136.0;
                EvRec rb{};
                prof_begin(h, s, rb, PC_BLOCK2, 2.0 * m * (64.0 * 576 + 2 * 256.0 * 64 + 64.0 * 256),
                           2.0 * (m * (64.0 + 64 + 256 + 64) + 64.0 * 576 + 2 * 256.0 * 64 + 64.0 * 256), (int)(&c2 - &h->convs[0]));
                e = launch_bneck_block1(buf[fr[0]], n, c2.w, c2.bias, c3.w, c3.bias, buf[cur], buf[fr[3]], nx->w, 64, nx->bias, buf[fr[1]], s, et,
                                        cdp->w, cdp->bias);
                prof_end(h, s, rb);
                if (e != hipSuccess) return fail(h, R50_ERR_HIP, "bneck_block1 (downsample) launch (" + c2.conv_key + "): " + hipGetErrorString(e));
                pre_t1 = fr[1];
                cur = fr[3];
                li += 4;
                continue;
            }
            if (block1) {
                if (!have_t1) {
                    rc = run_conv(h, c1, buf[cur], n, hh, ww, nullptr, buf[fr[0]], 1, s, &h1, &w1);
                    if (rc) return rc;
                }
                const double m = (double)n * 3136.0;
                EvRec rb{};
                // layer1.2: the block output is read again only at its even rows and columns (layer2.0's stride-2 downsample conv, inside the chained
                // transition tail; layer2.0.conv1 is computed in this launch): write just those -- 103 MB instead of 411 MB at batch 256.  Only when
                // the consumer that understands the compact tensor is the one that will run (the conditions of the catchain branch below).
                const bool sub = h->sub_out && nx->cout == 128 && b == blocks - 1 && h->fuse_ds_cat && h->cat_w[1] && h->fuse_cat_chain && h->catchain_wp &&
                                 (h->precision == R50_PREC_BF16 || h->precision == R50_PREC_FP16);
                prof_begin(h, s, rb, PC_BLOCK2, 2.0 * m * (64.0 * 576 + 256.0 * 64 + (double)nx->cout * 256),
                           2.0 * (m * (64.0 + 256 + (sub ? 64.0 : 256.0) + nx->cout) + 64.0 * 576 + 256.0 * 64 + (double)nx->cout * 256), (int)(&c2 - &h->convs[0]));
                e = launch_bneck_block1(buf[fr[0]], n, c2.w, c2.bias, c3.w, c3.bias, buf[cur], buf[fr[3]], nx->w, nx->cout, nx->bias, buf[fr[1]], s, et,
                                        nullptr, nullptr, sub);
                cur_sub = sub;
                prof_end(h, s, rb);
                if (e != hipSuccess) return fail(h, R50_ERR_HIP, "bneck_block1 launch (" + c2.conv_key + "): " + hipGetErrorString(e));
                pre_t1 = fr[1];
                cur = fr[3];
                li += 3;
                continue;
            }
            if (block2) {
                if (!have_t1) {
                    rc = run_conv(h, c1, buf[cur], n, hh, ww, nullptr, buf[fr[0]], 1, s, &h1, &w1);
                    if (rc) return rc;
                }
                const bool chain = nx && nx->ks == 1 && nx->stride == 1 && nx->cin == 512 && nx->cout == 128;
                const double m = (double)n * 784.0;
                EvRec rb{};
                prof_begin(h, s, rb, PC_BLOCK2, 2.0 * m * (128.0 * 1152 + 512.0 * 128 + (chain ? 128.0 * 512 : 0.0)),
                           2.0 * (m * (128.0 + 512 + 512 + (chain ? 128.0 : 0.0)) + 128.0 * 1152 + 512.0 * 128 + (chain ? 128.0 * 512 : 0.0)),
                           (int)(&c2 - &h->convs[0]));
                e = launch_bneck_block2(buf[fr[0]], n, c2.w, c2.bias, c3.w, c3.bias, buf[cur], buf[fr[3]], chain ? nx->w : nullptr,
                                        chain ? nx->bias : nullptr, chain ? buf[fr[1]] : nullptr, s, et);
                prof_end(h, s, rb);
                if (e != hipSuccess) return fail(h, R50_ERR_HIP, "bneck_block2 launch (" + c2.conv_key + "): " + hipGetErrorString(e));
                if (chain) pre_t1 = fr[1];
                cur = fr[3];
                li += 3;
                continue;
            }
            if (have_t1) {
                h1 = hh; w1 = ww;
            } else {
                rc = run_conv(h, c1, buf[cur], n, hh, ww, nullptr, buf[fr[0]], 1, s, &h1, &w1);
                if (rc) return rc;
            }
            if (hit(p + ".t1", buf[fr[0]], h1, w1, c1.cout)) return R50_OK;
            rc = run_conv(h, c2, buf[fr[0]], n, h1, w1, nullptr, buf[fr[1]], 1, s, &h2, &w2);
            if (rc) return rc;
            if (hit(p + ".t2", buf[fr[1]], h2, w2, c2.cout)) return R50_OK;
            if (b == 0 && !ds_side && !fuse_ds && !cat_ds) {
                const ConvLayer& cd = h->convs[li + 3];
                int hd, wd;
                rc = run_conv(h, cd, buf[cur], n, hh, ww, nullptr, buf[fr[2]], 0, s, &hd, &wd);
                if (rc) return rc;
                if (hit(p + ".ds", buf[fr[2]], hd, wd, cd.cout)) return R50_OK;
                idn = buf[fr[2]];
            }
            if (ds_side) HIP_TRY(h, hipStreamWaitEvent(s, h->ev_ds_join, 0));   // conv3 needs the downsample output
            // Plain-identity blocks write the block output IN PLACE over the block input: every element of the identity is read (by the
            // lane / wave / workgroup that produces the output element at the same address) before that output element is stored, nobody
            // else reads the block input any more (conv1 ran earlier), and a DRAM page that has just been read is written while it is
            // still open -- a device add_ (read + write the same addresses) streams 9 % faster than a copy on this chip
            // (profiles/r02_hbm_copy_probe.txt).
            const bool inplace = h->inplace_out && !split && b > 0 && idn == buf[cur] && !cat_ds && !fuse_ds &&
                                 !(h->precision == R50_PREC_FP8 && si == 0 && b == blocks - 1);      // (the fp8 hand-over writes 1-byte elements)
            __bf16* const outb = inplace ? buf[cur] : buf[fr[3]];
            if (fuse) {
                const long long m = (long long)n * h2 * w2;
                EvRec rt{};
                prof_begin(h, s, rt, fuse3 ? PC_TAIL3 : PC_TAIL, 2.0 * m * ((double)c3.cout * c3.cin * (fuse_ds ? 2 : 1) + (double)nx->cout * nx->cin),
                           2.0 * (m * ((double)c3.cin + (fuse_ds ? c3.cin : c3.cout) + c3.cout + nx->cout) +
                                  (double)c3.cout * c3.cin * (fuse_ds ? 2 : 1) + (double)nx->cin * nx->cout),
                           (int)(&c3 - &h->convs[0]));
                if (fuse3)
                    e = launch_bneck_tail3(buf[fr[1]], m, h->tail3_wp[b], c3.bias, idn, outb, nx->bias, buf[fr[0]], s, et);
                else if (fuse2)
                    e = launch_bneck_tail2(buf[fr[1]], m, c3.w, c3.bias, idn, outb, nx->w, nx->bias, buf[fr[0]], s, et);
                else
                    e = launch_bneck_tail(buf[fr[1]], m, c3.w, c3.bias, fuse_ds ? buf[cur] : idn, fuse_ds ? cdp->w : nullptr,
                                          fuse_ds ? cdp->bias : nullptr, outb, nx->w, nx->cout, nx->bias, buf[fr[0]], s, et);
                prof_end(h, s, rt);
                if (e != hipSuccess) return fail(h, R50_ERR_HIP, "bneck_tail launch (" + c3.conv_key + "): " + hipGetErrorString(e));
                h3 = h2; w3 = w2;
                pre_t1 = fr[0];
            } else if (cat_ds && h->fuse_cat_chain && h->catchain_wp && si == 1 && fuse_ok && !tap && h2 == 28 && w2 == 28 && hh == 56 && ww == 56 &&
                       (h->precision == R50_PREC_BF16 || h->precision == R50_PREC_FP16) && c3.cin == 128 && c3.cout == 512 && cdp->cin == 256 &&
                       cdp->stride == 2 && nx->cin == 512 && nx->cout == 128) {
                // layer2.0: the two-source conv3 + downsample GEMM chained with layer2.1.conv1 through LDS (kernels.h: bneck_catchain_kernel);
                // same bits as the two-source igemm launch followed by the 1x1 launch
                const double m = (double)n * 784.0;
                EvRec rt{};
                prof_begin(h, s, rt, PC_CATCHAIN, 2.0 * m * (512.0 * 384 + 128.0 * 512), 2.0 * (m * (128.0 + 256 + 512 + 128) + 512.0 * 384 + 128.0 * 512),
                           (int)(&c3 - &h->convs[0]));
                e = launch_bneck_catchain(buf[fr[1]], buf[cur], n, 28, h->catchain_wp, h->cat_bias[1], outb, nx->bias, buf[fr[0]], s, et, cur_sub);
                cur_sub = false;
                prof_end(h, s, rt);
                if (e != hipSuccess) return fail(h, R50_ERR_HIP, "bneck_catchain launch (" + c3.conv_key + "): " + hipGetErrorString(e));
                h3 = h2; w3 = w2;
                pre_t1 = fr[0];
            } else if (cat_ds) {
                rc = run_conv_cat(h, si, c3, *cdp, buf[fr[1]], buf[cur], n, h2, w2, hh, ww, outb, s);
                if (rc) return rc;
                h3 = h2; w3 = w2;
            } else if (si == 2 && b == blocks - 1 && b < 8 && h->tail3_wp[b] && h->fuse_tail3 && h->fuse_tail3_last && h->fuse_tail && !split && !tap && h->tile_override == 0 &&
                       (h->precision == R50_PREC_BF16 || h->precision == R50_PREC_FP16) && c3.ks == 1 && c3.cin == 256 && c3.cout == 1024 && idn == buf[cur]) {
                // layer3.5: conv3 + identity + ReLU through the pipelined tail kernel without a second GEMM (its next conv1, layer4.0's 1024 -> 512, does
                // not fit the chain): same bits as the igemm launch, whose K = 256 tiles alternate between an MFMA phase and an epilogue phase
                const long long m = (long long)n * h2 * w2;
                EvRec rt{};
                prof_begin(h, s, rt, PC_TAIL3, 2.0 * m * (double)c3.cout * c3.cin, 2.0 * (m * ((double)c3.cin + 2.0 * c3.cout) + (double)c3.cout * c3.cin),
                           (int)(&c3 - &h->convs[0]));
                e = launch_bneck_tail3(buf[fr[1]], m, h->tail3_wp[b], c3.bias, idn, outb, nullptr, nullptr, s, et, 0, /*no_next=*/true);
                prof_end(h, s, rt);
                if (e != hipSuccess) return fail(h, R50_ERR_HIP, "bneck_tail3 (no next conv1) launch (" + c3.conv_key + "): " + hipGetErrorString(e));
                h3 = h2; w3 = w2;
            } else {
                // fp8 mode, layer1's last conv: the hand-over to the fp8 stack (quantisation with the first activation scale) is folded into
                // this conv's epilogue -- the 16-bit tensor is neither written nor read back (unless a tap asks for it)
                const bool handover = (h->precision == R50_PREC_FP8 && h->fuse_fp8_handover && si == 0 && b == blocks - 1 && !tap &&
                                       (int)h->fp8_scales.size() == R50_FP8_NUM_SCALES);
                rc = run_conv(h, c3, buf[fr[1]], n, h2, w2, idn, outb, 1, s, &h3, &w3, handover ? 1.0f / h->fp8_scales[0] : 0.f,
                              handover ? &layer1_out_fp8 : nullptr);
                if (rc) return rc;
            }
            if (!inplace) cur = fr[3];
            hh = h3; ww = w3;
            if (hit(p, buf[cur], hh, ww, c3.cout)) return R50_OK;
            li += (b == 0) ? 4 : 3;
        }
    }
    if (tap) return fail(h, R50_ERR_INVALID, std::string("unknown layer name: ") + tap);
    prof_begin(h, s, r, PC_AVGPOOL, 0, (double)n * (hh * ww * 2048.0 * 2 + 2048.0 * 4));
    e = split ? launch_avgpool_split(buf[cur], out, n, hh * ww, 2048, s) : launch_avgpool(buf[cur], out, n, hh * ww, 2048, s, et);
    prof_end(h, s, r);
    if (e != hipSuccess) return fail(h, R50_ERR_HIP, std::string("avgpool: ") + hipGetErrorString(e));
    return R50_OK;
}

void free_weights(r50_handle* h) {        // the buffers a sharer reads through its owner
    const bool own = (h->owner == nullptr);
    for (auto& L : h->convs) {
        if (own && L.w) (void)hipFree(L.w);
        if (own && L.bias) (void)hipFree(L.bias);
        if (L.bias_scaled) (void)hipFree(L.bias_scaled);
        L.w = nullptr; L.bias = nullptr; L.bias_scaled = nullptr;
    }
    for (auto& p : h->tail3_wp) { if (own && p) (void)hipFree(p); p = nullptr; }
    if (own && h->catchain_wp) (void)hipFree(h->catchain_wp);
    h->catchain_wp = nullptr;
    for (int i = 0; i < 4; ++i) {
        if (own && h->cat_w[i]) (void)hipFree(h->cat_w[i]);
        if (own && h->cat_bias[i]) (void)hipFree(h->cat_bias[i]);
        h->cat_w[i] = nullptr; h->cat_bias[i] = nullptr;
    }
    if (own && h->stem_w) (void)hipFree(h->stem_w);
    h->stem_w = nullptr;
}
void free_all(r50_handle* h, bool keep_weights = false) {
    if (!keep_weights) free_weights(h);
    if (h->stem_xp) (void)hipFree(h->stem_xp);
    if (h->u8_table) (void)hipFree(h->u8_table);
    for (auto& b : h->buf) { if (b) (void)hipFree(b); b = nullptr; }
    for (int i = 0; i < 4; ++i) {
        if (h->side[i]) (void)hipStreamDestroy(h->side[i]);
        if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
        h->side[i] = nullptr; h->ev_join[i] = nullptr;
    }
    if (h->ds_stream) (void)hipStreamDestroy(h->ds_stream);
    if (h->ev_ds_fork) (void)hipEventDestroy(h->ev_ds_fork);
    if (h->ev_ds_join) (void)hipEventDestroy(h->ev_ds_join);
    h->ds_stream = nullptr; h->ev_ds_fork = nullptr; h->ev_ds_join = nullptr;
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    h->ev_fork = nullptr;
    for (auto& r : h->ev_pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto& e : h->ev_free) (void)hipEventDestroy(e);
    h->ev_pending.clear(); h->ev_free.clear();
}

}  // namespace

namespace {
template <typename TIN>
int forward_impl(r50_handle* h, const TIN* x, int n, float* out, void* stream) {
    if (!h) return fail(nullptr, R50_ERR_INVALID, "r50_forward: null handle");
    if (!h->loaded) return fail(h, R50_ERR_STATE, "r50_forward: weights not loaded");
    if (!x || !out) return fail(h, R50_ERR_INVALID, "r50_forward: null buffer");
    if (n < 0) return fail(h, R50_ERR_INVALID, "r50_forward: n < 0");
    HIP_TRY(h, hipSetDevice(h->device));
    int chunk = h->max_batch;
    if (h->micro_batch > 0 && h->micro_batch < chunk) chunk = h->micro_batch;
    hipStream_t user = (hipStream_t)stream;
    for (int i = 0; i < n; i += chunk) {
        const int m = (n - i < chunk) ? (n - i) : chunk;
        const int ns = (h->n_streams > 1 && m >= 2 * h->n_streams) ? h->n_streams : 1;
        if (ns == 1) {
            int rc = run_stack(h, x + (size_t)i * 3 * 224 * 224, m, out + (size_t)i * 2048, user, nullptr, nullptr, nullptr);
            if (rc) return rc;
            continue;
        }
        // fork: the side streams start after everything already queued on the caller's stream
        if (!h->ev_fork) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIP_TRY(h, hipEventRecord(h->ev_fork, user));
        const int per = (m + ns - 1) / ns;
        for (int k = 0; k < ns; ++k) {
            const int f0 = k * per, cnt = (m - f0 < per) ? (m - f0) : per;
            if (cnt <= 0) break;
            if (!h->side[k]) HIP_TRY(h, hipStreamCreateWithFlags(&h->side[k], hipStreamNonBlocking));
            if (!h->ev_join[k]) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_join[k], hipEventDisableTiming));
            HIP_TRY(h, hipStreamWaitEvent(h->side[k], h->ev_fork, 0));
            int rc = run_stack(h, x + (size_t)(i + f0) * 3 * 224 * 224, cnt, out + (size_t)(i + f0) * 2048, h->side[k],
                               nullptr, nullptr, nullptr, f0);
            if (rc) return rc;
            HIP_TRY(h, hipEventRecord(h->ev_join[k], h->side[k]));
            HIP_TRY(h, hipStreamWaitEvent(user, h->ev_join[k], 0));    // join
        }
    }
    return R50_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

const char* r50_version(void) { return "r50hip 0.1.0 (gfx950)"; }

const char* r50_last_error(r50_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

int r50_create(r50_handle** out, int device_id, int precision, int max_batch) {
    if (!out) return fail(nullptr, R50_ERR_INVALID, "r50_create: out is null");
    *out = nullptr;
    if (precision != R50_PREC_BF16 && precision != R50_PREC_FP32X && precision != R50_PREC_BF16W2 && precision != R50_PREC_FP16 &&
        precision != R50_PREC_FP8)
        return fail(nullptr, R50_ERR_INVALID, "r50_create: unsupported precision");
    const int cmul = (precision == R50_PREC_FP32X) ? 2 : 1;
    // The kernels address every tensor through 31-bit buffer descriptors (fill_conv_args): the largest activation, layer1's
    // (n,56,56,256), is n x 1.6 MB in the 16-bit modes (n <= 1337) and twice that in fp32x mode ([head | tail] pairs: n <= 668).
    const int batch_cap = (precision == R50_PREC_FP32X) ? 640 : 1024;
    if (max_batch < 1 || max_batch > batch_cap)
        return fail(nullptr, R50_ERR_INVALID, "r50_create: max_batch must be in [1," + std::to_string(batch_cap) + "] for this precision");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, R50_ERR_HIP, "r50_create: no HIP device available");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, R50_ERR_INVALID, "r50_create: bad device_id");
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device_id));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, R50_ERR_HIP, std::string("r50_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    HIP_TRY(nullptr, hipSetDevice(device_id));
    r50_handle* h = new r50_handle();
    h->device = device_id; h->precision = precision; h->max_batch = max_batch;
    h->convs = make_specs();
    for (int i = 0; i < PC_COUNT; ++i) h->prof[i].name = kProfNames[i];
    h->prof_layer.resize(h->convs.size());
    for (size_t i = 0; i < h->convs.size(); ++i) h->prof_layer[i].name = h->convs[i].conv_key.c_str();
    h->buf_bytes = (size_t)max_batch * 112 * 112 * 64 * 2 * cmul;
    bool ok = true;
    for (int i = 0; i < 5 && ok; ++i) ok = hipMalloc((void**)&h->buf[i], h->buf_bytes) == hipSuccess;
    ok = ok && hipMalloc((void**)&h->stem_xp, (size_t)max_batch * STEM_HP * STEM_WP * 8 * cmul) == hipSuccess;
    ok = ok && hipMalloc((void**)&h->stem_w, STEM_W_BYTES * cmul) == hipSuccess;
    {   // the reference's host arithmetic, three fp32 operations per sample (src/dataset.py:148-149,242-245)
        static const float kMean[3] = {0.485f, 0.456f, 0.406f}, kStd[3] = {0.229f, 0.224f, 0.225f};
        std::vector<float> tab(3 * 256);
        for (int c = 0; c < 3; ++c)
            for (int v = 0; v < 256; ++v) {
                const float a = (float)v / 255.0f;
                const float b = a - kMean[c];
                tab[c * 256 + v] = b / kStd[c];
            }
        ok = ok && hipMalloc((void**)&h->u8_table, tab.size() * sizeof(float)) == hipSuccess &&
             hipMemcpy(h->u8_table, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) {
        free_all(h);
        delete h;
        return fail(nullptr, R50_ERR_NOMEM, "r50_create: device allocation failed");
    }
    *out = h;
    return R50_OK;
}

void r50_destroy(r50_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    if (h->sharers > 0) {                 // other handles still read this one's weights: everything else goes now, the weights with the last sharer
        free_all(h, /*keep_weights=*/true);
        h->zombie = true;
        return;
    }
    r50_handle* owner = h->owner;
    free_all(h);
    delete h;
    if (owner && --owner->sharers == 0 && owner->zombie) {
        free_weights(owner);
        delete owner;
    }
}

int r50_share_weights(r50_handle* h, r50_handle* from) {
    if (!h || !from) return fail(h, R50_ERR_INVALID, "r50_share_weights: null handle");
    if (h == from || h->owner || h->loaded) return fail(h, R50_ERR_STATE, "r50_share_weights: the handle already has weights");
    if (!from->loaded || from->zombie) return fail(h, R50_ERR_STATE, "r50_share_weights: the source handle has no weights loaded");
    if (from->owner) from = from->owner;
    if (h->device != from->device || h->precision != from->precision)
        return fail(h, R50_ERR_INVALID, "r50_share_weights: device and precision of the two handles must be the same");
    if (h->precision != R50_PREC_BF16 && h->precision != R50_PREC_FP16)
        return fail(h, R50_ERR_INVALID, "r50_share_weights: bf16 / fp16 handles only (the fp8 mode rewrites weight buffers per handle when its scales are set)");
    (void)hipSetDevice(h->device);
    if (h->stem_w) (void)hipFree(h->stem_w);                  // allocated by r50_create
    h->stem_w = from->stem_w;
    for (size_t i = 0; i < h->convs.size(); ++i) {
        ConvLayer& L = h->convs[i];
        const ConvLayer& S = from->convs[i];
        L.w = S.w; L.bias = S.bias; L.wscale = S.wscale; L.bias_host = S.bias_host; L.w_host = S.w_host;
    }
    for (int i = 0; i < 4; ++i) { h->cat_w[i] = from->cat_w[i]; h->cat_bias[i] = from->cat_bias[i]; h->cat_acc_scale[i] = from->cat_acc_scale[i]; }
    for (int b = 0; b < 8; ++b) h->tail3_wp[b] = from->tail3_wp[b];
    h->catchain_wp = from->catchain_wp;
    h->owner = from;
    ++from->sharers;
    h->loaded = true;
    return R50_OK;
}

int r50_load_weights(r50_handle* h, const r50_tensor_desc* tensors, int n_tensors) {
    if (!h) return fail(nullptr, R50_ERR_INVALID, "r50_load_weights: null handle");
    if (!tensors || n_tensors <= 0) return fail(h, R50_ERR_INVALID, "r50_load_weights: no tensors");
    // handles that take part in r50_share_weights: a sharer would write the folded weights into its OWNER's buffers (and leak what it allocates beside
    // them); an owner with sharers would repack the weights under lanes that may be in mid-forward on other streams
    if (h->owner || h->sharers > 0 || h->zombie)
        return fail(h, R50_ERR_STATE, "r50_load_weights: this handle shares its weight buffers (r50_share_weights): load into a fresh handle instead");
    HIP_TRY(h, hipSetDevice(h->device));
    std::map<std::string, const r50_tensor_desc*> by_name;
    for (int i = 0; i < n_tensors; ++i) {
        if (!tensors[i].name || !tensors[i].data) return fail(h, R50_ERR_INVALID, "r50_load_weights: null name/data");
        by_name[tensors[i].name] = &tensors[i];
    }
    auto need = [&](const std::string& k, int64_t numel, const float** p) -> int {
        auto it = by_name.find(k);
        if (it == by_name.end()) return fail(h, R50_ERR_INVALID, "r50_load_weights: missing tensor " + k);
        if (it->second->numel != numel)
            return fail(h, R50_ERR_INVALID, "r50_load_weights: tensor " + k + " has " + std::to_string(it->second->numel) +
                                                " elements, expected " + std::to_string(numel));
        *p = it->second->data;
        return R50_OK;
    };
    std::vector<float> wf, bf;
    std::vector<uint16_t> pk;
    // [W3 | Wd] and b3 + bd of layer2.0 / layer3.0 / layer4.0 (16-bit precisions): convs are stored c1, c2, c3, ds per first block
    const bool want_cat = (h->precision == R50_PREC_BF16 || h->precision == R50_PREC_FP16);
    std::map<size_t, std::pair<std::vector<uint16_t>, std::vector<float>>> kept;       // conv index -> (packed weights, bias)
    for (size_t i = 0; i < h->convs.size(); ++i) {
        ConvLayer& L = h->convs[i];
        const float *w, *g, *b, *m, *v;
        const int per_out = L.cin * L.ks * L.ks;
        int rc;
        if ((rc = need(L.conv_key + ".weight", (int64_t)L.cout * per_out, &w))) return rc;
        if ((rc = need(L.bn_key + ".weight", L.cout, &g))) return rc;
        if ((rc = need(L.bn_key + ".bias", L.cout, &b))) return rc;
        if ((rc = need(L.bn_key + ".running_mean", L.cout, &m))) return rc;
        if ((rc = need(L.bn_key + ".running_var", L.cout, &v))) return rc;
        fold_bn(w, g, b, m, v, L.cout, per_out, wf, bf);
        L.bias_host = bf;
        if (!L.bias) HIP_TRY(h, hipMalloc((void**)&L.bias, L.cout * sizeof(float)));
        HIP_TRY(h, hipMemcpy(L.bias, bf.data(), L.cout * sizeof(float), hipMemcpyHostToDevice));
        const bool split = (h->precision == R50_PREC_FP32X);
        if (i == 0) {
            pack_stem(wf.data(), pk, h->precision == R50_PREC_FP16 ? 2 : 0);
            HIP_TRY(h, hipMemcpy(h->stem_w, pk.data(), STEM_W_BYTES, hipMemcpyHostToDevice));
            if (split) {
                pack_stem(wf.data(), pk, 1);
                HIP_TRY(h, hipMemcpy(h->stem_w + STEM_W_BYTES, pk.data(), STEM_W_BYTES, hipMemcpyHostToDevice));
            }
        } else {
            if (split) pack_ohwi_split(wf.data(), L.cout, L.cin, L.ks, pk);
            else if (h->precision == R50_PREC_BF16W2) pack_ohwi_w2(wf.data(), L.cout, L.cin, L.ks, pk);
            else if (h->precision == R50_PREC_FP16) pack_ohwi_f16(wf.data(), L.cout, L.cin, L.ks, pk);
            else if (h->precision == R50_PREC_FP8 && i >= kFp8FirstConv) {
                L.wscale = pack_ohwi_fp8(wf.data(), L.cout, L.cin, L.ks, pk);
                if (L.ks == 1 && (L.conv_key.find(".0.conv3") != std::string::npos || L.conv_key.find(".downsample.0") != std::string::npos))
                    L.w_host = wf;
            }
            else pack_ohwi_bf16(wf.data(), L.cout, L.cin, L.ks, pk);
            if (!L.w) HIP_TRY(h, hipMalloc((void**)&L.w, pk.size() * 2));
            HIP_TRY(h, hipMemcpy(L.w, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
            if (want_cat && L.ks == 1 && (L.conv_key.find(".0.conv3") != std::string::npos || L.conv_key.find(".downsample.0") != std::string::npos))
                kept[i] = {pk, bf};
        }
    }
    if (want_cat) {
        size_t li = 1;
        for (int si = 0; si < 4; ++si) {
            const size_t i3 = li + 2, id = li + 3;
            if (si >= 1 && kept.count(i3) && kept.count(id)) {
                const ConvLayer &c3 = h->convs[i3], &cd = h->convs[id];
                const int k1 = c3.cin, k2 = cd.cin, co = c3.cout;
                std::vector<uint16_t> cat((size_t)co * (k1 + k2));
                std::vector<float> cb(co);
                for (int o = 0; o < co; ++o) {
                    std::memcpy(&cat[(size_t)o * (k1 + k2)], &kept[i3].first[(size_t)o * k1], (size_t)k1 * 2);
                    std::memcpy(&cat[(size_t)o * (k1 + k2) + k1], &kept[id].first[(size_t)o * k2], (size_t)k2 * 2);
                    cb[o] = kept[i3].second[o] + kept[id].second[o];
                }
                if (!h->cat_w[si]) HIP_TRY(h, hipMalloc((void**)&h->cat_w[si], cat.size() * 2));
                if (!h->cat_bias[si]) HIP_TRY(h, hipMalloc((void**)&h->cat_bias[si], co * sizeof(float)));
                HIP_TRY(h, hipMemcpy(h->cat_w[si], cat.data(), cat.size() * 2, hipMemcpyHostToDevice));
                HIP_TRY(h, hipMemcpy(h->cat_bias[si], cb.data(), co * sizeof(float), hipMemcpyHostToDevice));
            }
            li += 4 + 3 * (kStages[si][1] - 1);
        }
    }
    // layer3.1-.4: conv3 of block b and conv1 of block b+1 in the chained tail kernel's fragment-ordered stream (16-bit precisions)
    if (want_cat) {
        size_t li = 1;
        for (int si = 0; si < 4; ++si) {
            const int blocks = kStages[si][1];
            for (int b = 0; b < blocks; ++b) {
                const size_t i3 = li + 2, inx = li + ((b == 0) ? 4 : 3);
                if (si == 1 && b == 0 && h->cat_w[1] && inx < h->convs.size()) {
                    const ConvLayer& c3 = h->convs[i3];
                    const ConvLayer& cd = h->convs[li + 3];
                    const ConvLayer& nx = h->convs[inx];
                    if (c3.ks == 1 && c3.cin == 128 && c3.cout == 512 && cd.ks == 1 && cd.cin == 256 && cd.stride == 2 && nx.ks == 1 && nx.stride == 1 &&
                        nx.cin == 512 && nx.cout == 128) {
                        if (!h->catchain_wp) HIP_TRY(h, hipMalloc((void**)&h->catchain_wp, kCatChainPackedBytes));
                        HIP_TRY(h, pack_catchain_weights(h->cat_w[1], nx.w, h->catchain_wp, nullptr));
                    }
                }
                if (si == 2 && b > 0 && b < 8 && inx < h->convs.size()) {
                    const ConvLayer& c3 = h->convs[i3];
                    const ConvLayer& nx = h->convs[inx];
                    if (c3.ks == 1 && c3.cin == 256 && c3.cout == 1024 && nx.ks == 1 && nx.stride == 1 && nx.cin == 1024 && nx.cout == 256) {
                        if (!h->tail3_wp[b]) HIP_TRY(h, hipMalloc((void**)&h->tail3_wp[b], kTail3PackedBytes));
                        HIP_TRY(h, pack_tail3_weights(c3.w, nx.w, h->tail3_wp[b], nullptr));
                    } else if (b == blocks - 1 && c3.ks == 1 && c3.cin == 256 && c3.cout == 1024) {
                        // the stage's last block: conv3 + identity + ReLU alone through the pipelined kernel (its W1 slices are never read: W3 stands in)
                        if (!h->tail3_wp[b]) HIP_TRY(h, hipMalloc((void**)&h->tail3_wp[b], kTail3PackedBytes));
                        HIP_TRY(h, pack_tail3_weights(c3.w, c3.w, h->tail3_wp[b], nullptr));
                    }
                }
                li += (b == 0) ? 4 : 3;
            }
        }
        HIP_TRY(h, hipDeviceSynchronize());
    }
    h->loaded = true;
    h->fp8_scales.clear();              // new weights: bias_scaled must be rebuilt
    return R50_OK;
}

int r50_set_fp8_scales(r50_handle* h, const float* scales, int n) {
    if (!h || !scales) return fail(h, R50_ERR_INVALID, "r50_set_fp8_scales: null argument");
    if (h->precision != R50_PREC_FP8) return fail(h, R50_ERR_STATE, "r50_set_fp8_scales: handle is not in R50_PREC_FP8 mode");
    if (!h->loaded) return fail(h, R50_ERR_STATE, "r50_set_fp8_scales: weights not loaded");
    if (n != R50_FP8_NUM_SCALES) return fail(h, R50_ERR_INVALID, "r50_set_fp8_scales: expected R50_FP8_NUM_SCALES values");
    for (int i = 0; i < n; ++i)
        if (!(scales[i] > 0.f) || !std::isfinite(scales[i])) return fail(h, R50_ERR_INVALID, "r50_set_fp8_scales: scales must be positive and finite");
    HIP_TRY(h, hipSetDevice(h->device));
    // input scale of every fp8 conv, walking the blocks in execution order (same walk as run_fp8_part)
    size_t li = kFp8FirstConv;
    int k = 0;
    float s_in = scales[k++];
    std::vector<float> tmp;
    auto set_bias = [&](ConvLayer& L, float sx) -> int {
        tmp.resize(L.cout);
        for (int o = 0; o < L.cout; ++o) tmp[o] = L.bias_host[o] / (sx * L.wscale);
        if (!L.bias_scaled) HIP_TRY(h, hipMalloc((void**)&L.bias_scaled, L.cout * sizeof(float)));
        HIP_TRY(h, hipMemcpy(L.bias_scaled, tmp.data(), L.cout * sizeof(float), hipMemcpyHostToDevice));
        return R50_OK;
    };
    for (int si = 1; si < 4; ++si)
        for (int b = 0; b < kStages[si][1]; ++b) {
            const float s_t1 = scales[k++], s_t2 = scales[k++];
            int rc;
            if ((rc = set_bias(h->convs[li], s_in))) return rc;
            if ((rc = set_bias(h->convs[li + 1], s_t1))) return rc;
            if ((rc = set_bias(h->convs[li + 2], s_t2))) return rc;
            if (b == 0) {
                ++k;
                if ((rc = set_bias(h->convs[li + 3], s_in))) return rc;
                // conv3 + downsample as ONE fp8 accumulation over [t2 | x at the block's stride]: both products must land in the same
                // accumulator unit, s = s_t2 * s_w3' = s_x * s_wd'.  s = the larger of the two natural units, so neither weight matrix
                // clips; the other one is requantised a little coarser than its absmax would allow.
                ConvLayer &c3 = h->convs[li + 2], &cd = h->convs[li + 3];
                if (!c3.w_host.empty() && !cd.w_host.empty()) {
                    const float a3 = absmax_of(c3.w_host), ad = absmax_of(cd.w_host);
                    const float unit = std::fmax(s_t2 * (a3 > 0.f ? a3 / 448.0f : 1.0f), s_in * (ad > 0.f ? ad / 448.0f : 1.0f));
                    const float sw3 = unit / s_t2, swd = unit / s_in;
                    const int k1 = c3.cin, k2 = cd.cin, co = c3.cout;
                    std::vector<uint8_t> cat((size_t)co * (k1 + k2));
                    tmp.resize(co);
                    for (int o = 0; o < co; ++o) {
                        for (int c = 0; c < k1; ++c) cat[(size_t)o * (k1 + k2) + c] = f32_to_e4m3(c3.w_host[(size_t)o * k1 + c] / sw3);
                        for (int c = 0; c < k2; ++c) cat[(size_t)o * (k1 + k2) + k1 + c] = f32_to_e4m3(cd.w_host[(size_t)o * k2 + c] / swd);
                        tmp[o] = (c3.bias_host[o] + cd.bias_host[o]) / unit;
                    }
                    if (!h->cat_w[si]) HIP_TRY(h, hipMalloc((void**)&h->cat_w[si], cat.size()));
                    if (!h->cat_bias[si]) HIP_TRY(h, hipMalloc((void**)&h->cat_bias[si], co * sizeof(float)));
                    HIP_TRY(h, hipMemcpy(h->cat_w[si], cat.data(), cat.size(), hipMemcpyHostToDevice));
                    HIP_TRY(h, hipMemcpy(h->cat_bias[si], tmp.data(), co * sizeof(float), hipMemcpyHostToDevice));
                    h->cat_acc_scale[si] = unit;
                }
            }
            s_in = scales[k++];
            li += (b == 0) ? 4 : 3;
        }
    h->fp8_scales.assign(scales, scales + n);
    return R50_OK;
}

int r50_forward(r50_handle* h, const float* x, int n, float* out, void* stream) {
    return forward_impl<float>(h, x, n, out, stream);
}

int r50_forward_u8(r50_handle* h, const uint8_t* x, int n, float* out, void* stream) {
    return forward_impl<unsigned char>(h, x, n, out, stream);
}

int r50_forward_layer(r50_handle* h, const float* x, int n, const char* layer, void* out, int64_t cap,
                      int64_t dims_out[4], void* stream) {
    if (!h) return fail(nullptr, R50_ERR_INVALID, "r50_forward_layer: null handle");
    if (!h->loaded) return fail(h, R50_ERR_STATE, "r50_forward_layer: weights not loaded");
    if (!x || !out || !layer || !dims_out) return fail(h, R50_ERR_INVALID, "r50_forward_layer: null argument");
    if (n < 1 || n > h->max_batch) return fail(h, R50_ERR_INVALID, "r50_forward_layer: n must be in [1,max_batch]");
    HIP_TRY(h, hipSetDevice(h->device));
    const __bf16* p = nullptr;
    int rc = run_stack(h, x, n, nullptr, (hipStream_t)stream, layer, &p, dims_out);
    if (rc) return rc;
    const int64_t bytes = dims_out[0] * dims_out[1] * dims_out[2] * dims_out[3] * 2;
    if (bytes > cap) return fail(h, R50_ERR_INVALID, "r50_forward_layer: output buffer too small");
    HIP_TRY(h, hipMemcpyAsync(out, p, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return R50_OK;
}

int r50_set_option(r50_handle* h, const char* key, int64_t value) {
    if (!h || !key) return fail(h, R50_ERR_INVALID, "r50_set_option: null argument");
    const std::string k(key);
    if (k == "micro_batch") { if (value < 0) return fail(h, R50_ERR_INVALID, "micro_batch < 0"); h->micro_batch = (int)value; }
    else if (k == "profile") h->profile = value ? 1 : 0;
    else if (k == "tile") h->tile_override = (int)value;
    else if (k == "fused_stem") h->fused_stem = value ? 1 : 0;
    else if (k == "fuse_tail") h->fuse_tail = value ? 1 : 0;
    else if (k == "fuse_tail3") h->fuse_tail3 = value ? 1 : 0;
    else if (k == "fuse_cat_chain") h->fuse_cat_chain = value ? 1 : 0;
    else if (k == "sub_out") h->sub_out = value ? 1 : 0;
    else if (k == "fuse_tail3_last") h->fuse_tail3_last = value ? 1 : 0;
    else if (k == "tail3_bp") { if (value < 0 || value > 112) return fail(h, R50_ERR_INVALID, "tail3_bp must be 0 .. 112"); g_tail3_bp = (int)value; }
    else if (k == "fuse_fp8_handover") h->fuse_fp8_handover = value ? 1 : 0;
    else if (k == "fuse_stem_c1") h->fuse_stem_c1 = value ? 1 : 0;
    else if (k == "fuse_ds_cat") h->fuse_ds_cat = value ? 1 : 0;
    else if (k == "stem_strip") {          // process-wide (the launcher is shared by the handle and the r50_op_* hooks)
        if (!(value == 0 || (value > 0 && 28 % value == 0))) return fail(h, R50_ERR_INVALID, "stem_strip must be 0 or a divisor of 28");
        g_stem_strip = (int)value;
    }
    else if (k == "overlap_ds") h->overlap_ds = value ? 1 : 0;
    else if (k == "streams") { if (value < 1 || value > 4) return fail(h, R50_ERR_INVALID, "streams must be in [1,4]"); h->n_streams = (int)value; }
    else if (k == "inplace_out") h->inplace_out = value ? 1 : 0;
    else if (k == "fuse_block2") h->fuse_block2 = value ? 1 : 0;
    else if (k == "fuse_block1") { if (value < 0 || value > 3) return fail(h, R50_ERR_INVALID, "fuse_block1 must be 0 .. 3"); h->fuse_block1 = (int)value; }
    else if (k == "use_g8") { if (value < 0 || value > 3) return fail(h, R50_ERR_INVALID, "use_g8 must be 0 .. 3"); g_use_g8 = (int)value; }   // process-wide A/B knob
    else if (k == "use_s2") g_use_s2 = (int)value;                     // process-wide A/B knob: 0 = generic tiles for the stride-2 3x3 shapes
    else if (k == "cu_cap") { if (value < 0 || value > 4096) return fail(h, R50_ERR_INVALID, "cu_cap must be in [0,4096]"); g_cu_cap = (int)value; g_num_cus = 0; }
    else return fail(h, R50_ERR_INVALID, "r50_set_option: unknown key " + k);
    return R50_OK;
}

int r50_get_option(r50_handle* h, const char* key, int64_t* value) {
    if (!h || !key || !value) return fail(h, R50_ERR_INVALID, "r50_get_option: null argument");
    const std::string k(key);
    if (k == "micro_batch") *value = h->micro_batch;
    else if (k == "profile") *value = h->profile;
    else if (k == "tile") *value = h->tile_override;
    else if (k == "streams") *value = h->n_streams;
    else if (k == "cu_cap") *value = g_cu_cap;
    else if (k == "inplace_out") *value = h->inplace_out;
    else if (k == "fuse_block2") *value = h->fuse_block2;
    else if (k == "fuse_block1") *value = h->fuse_block1;
    else if (k == "fused_stem") *value = h->fused_stem;
    else if (k == "fuse_tail") *value = h->fuse_tail;
    else if (k == "fuse_tail3") *value = h->fuse_tail3;
    else if (k == "tail3_bp") *value = g_tail3_bp;
    else if (k == "use_g8") *value = g_use_g8;
    else if (k == "fuse_cat_chain") *value = h->fuse_cat_chain;
    else if (k == "sub_out") *value = h->sub_out;
    else if (k == "fuse_tail3_last") *value = h->fuse_tail3_last;
    else if (k == "fuse_fp8_handover") *value = h->fuse_fp8_handover;
    else if (k == "fuse_stem_c1") *value = h->fuse_stem_c1;
    else if (k == "fuse_ds_cat") *value = h->fuse_ds_cat;
    else if (k == "stem_strip") *value = g_stem_strip;
    else if (k == "overlap_ds") *value = h->overlap_ds;
    else if (k == "max_batch") *value = h->max_batch;
    else if (k == "workspace_bytes") *value = (int64_t)(5 * h->buf_bytes + (size_t)h->max_batch * STEM_HP * STEM_WP * 8);
    else return fail(h, R50_ERR_INVALID, "r50_get_option: unknown key " + k);
    return R50_OK;
}

int r50_profile_reset(r50_handle* h) {
    if (!h) return R50_ERR_INVALID;
    for (auto& r : h->ev_pending) { h->ev_free.push_back(r.a); h->ev_free.push_back(r.b); }
    h->ev_pending.clear();
    for (int i = 0; i < PC_COUNT; ++i) { h->prof[i].launches = 0; h->prof[i].ms = h->prof[i].flops = h->prof[i].bytes = 0; }
    for (auto& p : h->prof_layer) { p.launches = 0; p.ms = p.flops = p.bytes = 0; }
    return R50_OK;
}

int r50_profile_collect(r50_handle* h) {
    if (!h) return R50_ERR_INVALID;
    for (auto& r : h->ev_pending) {
        HIP_TRY(h, hipEventSynchronize(r.b));
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, r.a, r.b));
        Prof& p = h->prof[r.cls];
        p.launches += 1; p.ms += ms; p.flops += r.flops; p.bytes += r.bytes;
        if (r.layer >= 0 && r.layer < (int)h->prof_layer.size()) {
            Prof& q = h->prof_layer[r.layer];
            q.launches += 1; q.ms += ms; q.flops += r.flops; q.bytes += r.bytes;
        }
        h->ev_free.push_back(r.a); h->ev_free.push_back(r.b);
    }
    h->ev_pending.clear();
    return R50_OK;
}

int r50_profile_count(r50_handle* h) { return h ? PC_COUNT + (int)h->prof_layer.size() - 1 : 0; }

int r50_profile_entry(r50_handle* h, int i, const char** name, int64_t* launches, double* total_ms, double* flops,
                      double* bytes) {
    if (!h || i < 0 || i >= PC_COUNT + (int)h->prof_layer.size() - 1) return R50_ERR_INVALID;
    const Prof& p = (i < PC_COUNT) ? h->prof[i] : h->prof_layer[i - PC_COUNT + 1];    // [0] is the stem conv: class "conv1"
    if (name) *name = p.name;
    if (launches) *launches = p.launches;
    if (total_ms) *total_ms = p.ms;
    if (flops) *flops = p.flops;
    if (bytes) *bytes = p.bytes;
    return R50_OK;
}

int r50_get_packed(r50_handle* h, const char* conv_key, int what, void* dst_host, int64_t capacity_bytes,
                   int64_t* bytes_out) {
    if (!h || !conv_key || !dst_host || !bytes_out) return fail(h, R50_ERR_INVALID, "r50_get_packed: null argument");
    if (!h->loaded) return fail(h, R50_ERR_STATE, "r50_get_packed: weights not loaded");
    HIP_TRY(h, hipSetDevice(h->device));
    for (size_t i = 0; i < h->convs.size(); ++i) {
        const ConvLayer& L = h->convs[i];
        if (L.conv_key != conv_key) continue;
        const void* src;
        int64_t bytes;
        if (what == 1) { src = L.bias; bytes = (int64_t)L.cout * 4; }
        else if (i == 0) { src = h->stem_w; bytes = STEM_W_BYTES; }
        else if (h->precision == R50_PREC_FP8 && i >= kFp8FirstConv) { src = L.w; bytes = (int64_t)L.cout * L.ks * L.ks * L.cin; }
        else { src = L.w; bytes = (int64_t)L.cout * L.ks * L.ks * L.cin * 2 * (h->precision == R50_PREC_FP32X ? 3 : h->precision == R50_PREC_BF16W2 ? 2 : 1); }
        *bytes_out = bytes;
        if (bytes > capacity_bytes) return fail(h, R50_ERR_INVALID, "r50_get_packed: buffer too small");
        HIP_TRY(h, hipMemcpy(dst_host, src, bytes, hipMemcpyDeviceToHost));
        return R50_OK;
    }
    return fail(h, R50_ERR_INVALID, std::string("r50_get_packed: unknown conv ") + conv_key);
}

// ---- op-level entry points -------------------------------------------------------------------
#if defined(R50_STAMP)      // diagnostic build only (not declared in include/r50.h): where the kernel's cycle sums go
extern "C" __attribute__((visibility("default"))) void r50_debug_buffer(void* p) { g_dbg = (unsigned long long*)p; }
#endif

static int op_conv2d_et(int et, const void* x, int n, int h, int w, int cin, const void* wt, const float* bias, const void* res,
                        void* y, int cout, int ksize, int stride, int pad, int relu, int tile, void* stream);

int r50_op_conv2d(const void* x, int n, int h, int w, int cin, const void* wt, const float* bias, const void* res,
                  void* y, int cout, int ksize, int stride, int pad, int relu, int tile, void* stream) {
    return op_conv2d_et(0, x, n, h, w, cin, wt, bias, res, y, cout, ksize, stride, pad, relu, tile, stream);
}

int r50_op_conv2d_f16(const void* x, int n, int h, int w, int cin, const void* wt, const float* bias, const void* res,
                      void* y, int cout, int ksize, int stride, int pad, int relu, int tile, void* stream) {
    return op_conv2d_et(1, x, n, h, w, cin, wt, bias, res, y, cout, ksize, stride, pad, relu, tile, stream);
}

static int op_conv2d_et(int et, const void* x, int n, int h, int w, int cin, const void* wt, const float* bias, const void* res,
                        void* y, int cout, int ksize, int stride, int pad, int relu, int tile, void* stream) {
    ConvArgs a;
    int rc = fill_conv_args(a, x, n, h, w, cin, wt, bias, res, y, cout, ksize, stride, pad, relu);
    if (rc) return fail(nullptr, rc, "r50_op_conv2d: invalid arguments");
    a.et = et;
#if defined(R50_STAMP)
    a.dbg = g_dbg;
#endif
    hipError_t e = launch_igemm(a, tile, (hipStream_t)stream);
    if (e != hipSuccess) return fail(nullptr, R50_ERR_HIP, std::string("r50_op_conv2d: ") + hipGetErrorString(e));
    return R50_OK;
}

int r50_op_conv2d_fp8(const void* x, int n, int h, int w, int cin, const void* wt, const float* bias_scaled, const void* res, void* y,
                      int cout, int ksize, int stride, int pad, int relu, float oscale, float rscale, int tile, void* stream) {
    if (cin <= 0 || cin % 128) return fail(nullptr, R50_ERR_INVALID, "r50_op_conv2d_fp8: cin must be a multiple of 128");
    ConvArgs a;
    // the loaders move bytes: described in 2-byte units, an fp8 tensor with cin channels is a 16-bit tensor with cin / 2
    int rc = fill_conv_args(a, x, n, h, w, cin / 2, wt, bias_scaled, res, y, cout, ksize, stride, pad, relu);
    if (rc) return fail(nullptr, rc, "r50_op_conv2d_fp8: invalid arguments");
    a.y_bytes = (unsigned)((long long)a.M * cout);       // one byte per output element
    a.et = 2; a.oscale = oscale; a.rscale = rscale;
    hipError_t e = launch_igemm_fp8(a, tile, (hipStream_t)stream);
    if (e != hipSuccess) return fail(nullptr, e == hipErrorInvalidValue ? R50_ERR_INVALID : R50_ERR_HIP,
                                     std::string("r50_op_conv2d_fp8: ") + hipGetErrorString(e));
    return R50_OK;
}

int r50_op_conv1x1_cat(const void* x1, int n, int h, int w, int c1, const void* x2, int h2, int w2, int c2, int stride2, const void* wcat,
                       const float* bias, void* y, int cout, int relu, int tile, int et, void* stream) {
    ConvArgs a;
    int rc = fill_conv_args_cat(a, x1, n, h, w, c1, x2, h2, w2, c2, stride2, wcat, bias, y, cout, relu);
    if (rc || (et != 0 && et != 1)) return fail(nullptr, R50_ERR_INVALID, "r50_op_conv1x1_cat: invalid arguments");
    a.et = et;
    hipError_t e = launch_igemm(a, cat_tile(a, tile), (hipStream_t)stream);
    if (e != hipSuccess) return fail(nullptr, e == hipErrorInvalidValue ? R50_ERR_INVALID : R50_ERR_HIP,
                                     std::string("r50_op_conv1x1_cat: ") + hipGetErrorString(e));
    return R50_OK;
}

int r50_op_bneck_tail(const void* y2, int64_t m, int cmid, const void* w3, const float* b3, const void* res, const void* wd,
                      const float* bd, void* out, const void* w1, int c1, const float* b1, void* y1n, void* stream) {
    hipError_t e;
    if (cmid == 64) e = launch_bneck_tail(y2, m, w3, b3, res, wd, bd, out, w1, c1, b1, y1n, (hipStream_t)stream);
    else if (cmid == 128 && c1 == 128 && !wd && !bd) e = launch_bneck_tail2(y2, m, w3, b3, res, out, w1, b1, y1n, (hipStream_t)stream);
    else if (cmid == 256 && c1 == 0 && !w1 && !b1 && !y1n && !wd && !bd) {
        // conv3 + identity + ReLU alone through the pipelined layer3 tail (the form layer3.5 runs in the network): W3 stands in for the unused W1 slices
        void* wp_scratch = nullptr;
        if (hipMallocAsync(&wp_scratch, kTail3PackedBytes, (hipStream_t)stream) != hipSuccess || !wp_scratch)
            return fail(nullptr, R50_ERR_HIP, "r50_op_bneck_tail: hipMallocAsync");
        e = pack_tail3_weights(w3, w3, wp_scratch, (hipStream_t)stream);
        if (e == hipSuccess) e = launch_bneck_tail3(y2, m, wp_scratch, b3, res, out, nullptr, nullptr, (hipStream_t)stream, 0, 0, /*no_next=*/true);
        const hipError_t ef = hipFreeAsync(wp_scratch, (hipStream_t)stream);
        if (e == hipSuccess) e = ef;
    }
    else if (cmid == 256 && c1 == 256 && !wd && !bd) {
        const char* v = std::getenv("R50_TAIL3_BP");          // test / A-B knob of this debug hook: real pixels per tile (1..112); unset = automatic
        // the hook takes plain weight matrices: packed here, per call, into a buffer allocated AND freed in the caller's stream order on the
        // caller's current device (a process-wide scratch buffer would be shared by callers on other streams / devices: round-2 ADVICE)
        void* wp_scratch = nullptr;
        if (hipMallocAsync(&wp_scratch, kTail3PackedBytes, (hipStream_t)stream) != hipSuccess || !wp_scratch)
            return fail(nullptr, R50_ERR_HIP, "r50_op_bneck_tail: hipMallocAsync");
        e = pack_tail3_weights(w3, w1, wp_scratch, (hipStream_t)stream);
        if (e == hipSuccess) e = launch_bneck_tail3(y2, m, wp_scratch, b3, res, out, b1, y1n, (hipStream_t)stream, 0, v ? std::atoi(v) : 0);
        const hipError_t ef = hipFreeAsync(wp_scratch, (hipStream_t)stream);
        if (e == hipSuccess) e = ef;
    }
    else e = hipErrorInvalidValue;
    if (e != hipSuccess) return fail(nullptr, e == hipErrorInvalidValue ? R50_ERR_INVALID : R50_ERR_HIP,
                                     std::string("r50_op_bneck_tail: ") + hipGetErrorString(e));
    return R50_OK;
}

int r50_op_bneck_block2(const void* t1, int n, const void* w2, const float* b2, const void* w3, const float* b3, const void* res,
                        void* out, const void* w1, const float* b1, void* y1n, void* stream) {
    const hipError_t e = launch_bneck_block2(t1, n, w2, b2, w3, b3, res, out, w1, b1, y1n, (hipStream_t)stream);
    if (e != hipSuccess) return fail(nullptr, e == hipErrorInvalidValue ? R50_ERR_INVALID : R50_ERR_HIP,
                                     std::string("r50_op_bneck_block2: ") + hipGetErrorString(e));
    return R50_OK;
}

int r50_op_bneck_block1_ds(const void* t1, int n, const void* w2, const float* b2, const void* w3, const float* b3, const void* x, const void* wd,
                           const float* bd, void* out, const void* w1, const float* b1, void* y1n, void* stream) {
    if (!wd || !bd) return fail(nullptr, R50_ERR_INVALID, "r50_op_bneck_block1_ds: null downsample weights");
    const hipError_t e = launch_bneck_block1(t1, n, w2, b2, w3, b3, x, out, w1, 64, b1, y1n, (hipStream_t)stream, 0, wd, bd);
    if (e != hipSuccess) return fail(nullptr, e == hipErrorInvalidValue ? R50_ERR_INVALID : R50_ERR_HIP,
                                     std::string("r50_op_bneck_block1_ds: ") + hipGetErrorString(e));
    return R50_OK;
}

int r50_op_bneck_cat_chain(const void* t2, const void* x, int n, int ow, const void* wcat, const float* bcat, void* out, const void* w1,
                           const float* b1, void* y1n, void* stream) {
    // plain weight matrices in, packed per call into a buffer allocated and freed in the caller's stream order (as r50_op_bneck_tail does)
    void* wp = nullptr;
    if (!wcat || !w1) return fail(nullptr, R50_ERR_INVALID, "r50_op_bneck_cat_chain: null weights");
    if (hipMallocAsync(&wp, kCatChainPackedBytes, (hipStream_t)stream) != hipSuccess || !wp)
        return fail(nullptr, R50_ERR_HIP, "r50_op_bneck_cat_chain: hipMallocAsync");
    hipError_t e = pack_catchain_weights(wcat, w1, wp, (hipStream_t)stream);
    if (e == hipSuccess) e = launch_bneck_catchain(t2, x, n, ow, wp, bcat, out, b1, y1n, (hipStream_t)stream);
    const hipError_t ef = hipFreeAsync(wp, (hipStream_t)stream);
    if (e == hipSuccess) e = ef;
    if (e != hipSuccess) return fail(nullptr, e == hipErrorInvalidValue ? R50_ERR_INVALID : R50_ERR_HIP,
                                     std::string("r50_op_bneck_cat_chain: ") + hipGetErrorString(e));
    return R50_OK;
}

int r50_op_bneck_block1(const void* t1, int n, const void* w2, const float* b2, const void* w3, const float* b3, const void* res,
                        void* out, const void* w1, int c1, const float* b1, void* y1n, void* stream) {
    const hipError_t e = launch_bneck_block1(t1, n, w2, b2, w3, b3, res, out, w1, c1, b1, y1n, (hipStream_t)stream);
    if (e != hipSuccess) return fail(nullptr, e == hipErrorInvalidValue ? R50_ERR_INVALID : R50_ERR_HIP,
                                     std::string("r50_op_bneck_block1: ") + hipGetErrorString(e));
    return R50_OK;
}

// Weight precision of ATen's native uint8 bilinear resize (antialias off) along one axis: the largest p with
// round(max_weight * 2^(p+1)) < 2^15, the maximum taken over every tap weight of the axis
// (aten/src/ATen/native/cpu/UpSampleKernel.cpp `_compute_index_ranges_int16_weights`; weights as in
// `_compute_indices_min_size_weights`, restated in oracle/resize_oracle.py:index_weights_int16 and, per thread, in
// kernels.h:resize_taps).
static int resize_weight_precision(int in_size, int out_size) {
    const double scale = (double)in_size / (double)out_size;
    double wt_max = 0.0;
    for (int i = 0; i < out_size; ++i) {
        double real = scale * (i + 0.5) - 0.5;
        if (real < 0.0) real = 0.0;
        long idx = (long)std::floor(real);
        if (idx > in_size - 1) idx = in_size - 1;
        double lam = real - (double)idx;
        lam = lam < 0.0 ? 0.0 : (lam > 1.0 ? 1.0 : lam);
        const long umin = idx, umax = idx + 2;
        const long lo = umin > 0 ? umin : 0;
        const long size = (umax < in_size ? umax : in_size) - lo;
        double w[2] = {0.0, 0.0};
        long w_index = 0;
        for (int j = 0; j < 2; ++j) {
            const double x = std::fabs((double)j - lam);
            const double wj = x < 1.0 ? 1.0 - x : 0.0;
            if (umin + j <= 0) w_index = 0;
            else if (umin + j >= in_size - 1) w_index = size - 1;
            if (w_index == 0) w[0] += wj; else w[1] += wj;
            ++w_index;
        }
        if (w[0] > wt_max) wt_max = w[0];
        if (w[1] > wt_max) wt_max = w[1];
    }
    int precision = 0;
    for (; precision < 22; ++precision)
        if ((int)(0.5 + wt_max * (double)(1 << (precision + 1))) >= (1 << 15)) break;
    return precision;
}

int r50_op_crop_resize_u8(const void* frames, int t, int h, int w, int top, int left, int hh, int ww, void* out, int out_size,
                          int mode, int flags, void* stream) {
    if (!frames || !out || t < 1 || h < 1 || w < 1 || hh < 1 || ww < 1 || top < 0 || left < 0 || top + hh > h || left + ww > w ||
        out_size < 4 || (out_size & 3) || out_size > 4096 || (long long)t * h * w * 3 >= (1ll << 40))
        return fail(nullptr, R50_ERR_INVALID, "r50_op_crop_resize_u8: invalid arguments (the box must lie inside the frame; out_size % 4 == 0)");
    if (mode != R50_RESIZE_FLOAT && mode != R50_RESIZE_FIXED) return fail(nullptr, R50_ERR_INVALID, "r50_op_crop_resize_u8: unknown mode");
    if (flags & ~(R50_AUG_HFLIP | R50_AUG_TREV)) return fail(nullptr, R50_ERR_INVALID, "r50_op_crop_resize_u8: unknown flags");
    ResizeArgs a;
    a.hflip = (flags & R50_AUG_HFLIP) ? 1 : 0;
    a.trev = (flags & R50_AUG_TREV) ? 1 : 0;
    a.float_mode = (mode == R50_RESIZE_FLOAT);
    a.px = a.py = 1;
    if (!a.float_mode) {
        a.px = resize_weight_precision(ww, out_size);
        a.py = resize_weight_precision(hh, out_size);
        if (a.px < 1 || a.py < 1) return fail(nullptr, R50_ERR_INVALID, "r50_op_crop_resize_u8: degenerate weights");
    }
    a.src = (const unsigned char*)frames; a.dst = (unsigned char*)out;
    a.T = t; a.H = h; a.W = w; a.top = top; a.left = left; a.hh = hh; a.ww = ww; a.out = out_size;
    const size_t lds = 2 * (size_t)((ww * 3 + 6) & ~3);                // two source rows of the crop, dword-padded
    if (lds > 160 * 1024 || (long long)t * out_size >= (1ll << 31)) return fail(nullptr, R50_ERR_INVALID, "r50_op_crop_resize_u8: crop too wide");
    hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(crop_resize_u8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) return fail(nullptr, R50_ERR_HIP, std::string("r50_op_crop_resize_u8: ") + hipGetErrorString(ea));
    hipLaunchKernelGGL(crop_resize_u8_kernel, dim3((unsigned)(t * out_size)), dim3(256), lds, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, R50_ERR_HIP, std::string("r50_op_crop_resize_u8: ") + hipGetErrorString(e));
    return R50_OK;
}

// ---- lifting head pieces (kernels.h: cast_rows_kernel, concat_pad_kernel, gn_relu_causal3_kernel); et: 0 = bf16, 1 = fp16
int r50_op_cast_rows(const float* src, int64_t rows, int c, void* dst, int cpad, int et, void* stream) {
    if (!src || !dst || rows < 1 || c < 1 || cpad < c || (cpad & 1) || (et != 0 && et != 1))
        return fail(nullptr, R50_ERR_INVALID, "r50_op_cast_rows: invalid arguments");
    const long long pairs = rows * (cpad / 2);
    const unsigned grid = (unsigned)std::min<long long>((pairs + 255) / 256, 256 * 32);
    if (et) hipLaunchKernelGGL(cast_rows_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (unsigned short*)dst, (long long)rows, c, cpad);
    else hipLaunchKernelGGL(cast_rows_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (unsigned short*)dst, (long long)rows, c, cpad);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? R50_OK : fail(nullptr, R50_ERR_HIP, std::string("r50_op_cast_rows: ") + hipGetErrorString(e));
}

int r50_op_concat_pad(const void* phi, int d, const float* y, int ny, int64_t rows, void* dst, int dp, int et, void* stream) {
    if (!phi || !y || !dst || rows < 1 || d < 1 || ny < 0 || dp < d + ny || (et != 0 && et != 1))
        return fail(nullptr, R50_ERR_INVALID, "r50_op_concat_pad: invalid arguments");
    const long long total = rows * dp;
    const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 256 * 32);
    if (et) hipLaunchKernelGGL(concat_pad_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)phi, d, y, ny, (unsigned short*)dst, (long long)rows, dp);
    else hipLaunchKernelGGL(concat_pad_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)phi, d, y, ny, (unsigned short*)dst, (long long)rows, dp);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? R50_OK : fail(nullptr, R50_ERR_HIP, std::string("r50_op_concat_pad: ") + hipGetErrorString(e));
}

int r50_op_add_rows(float* y, int ny, const void* dy, int dp, int64_t rows, int et, void* stream) {
    if (!y || !dy || rows < 1 || ny < 1 || dp < ny || (et != 0 && et != 1))
        return fail(nullptr, R50_ERR_INVALID, "r50_op_add_rows: invalid arguments");
    const long long total = rows * ny;
    const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 256 * 32);
    if (et) hipLaunchKernelGGL(add_rows_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, ny, (const unsigned short*)dy, dp, (long long)rows);
    else hipLaunchKernelGGL(add_rows_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, ny, (const unsigned short*)dy, dp, (long long)rows);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? R50_OK : fail(nullptr, R50_ERR_HIP, std::string("r50_op_add_rows: ") + hipGetErrorString(e));
}

int r50_op_gn_relu_causal3(const void* x, int b, int t, int c, int groups, const float* gamma, const float* beta, float eps,
                           void* out, int et, void* stream) {
    if (!x || !gamma || !beta || !out || b < 1 || t < 1 || c < 1 || groups < 1 || c % groups || (et != 0 && et != 1))
        return fail(nullptr, R50_ERR_INVALID, "r50_op_gn_relu_causal3: invalid arguments");
    if (et) hipLaunchKernelGGL(gn_relu_causal3_kernel<1>, dim3((unsigned)(b * groups)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, gamma, beta, (unsigned short*)out, t, c, groups, eps);
    else hipLaunchKernelGGL(gn_relu_causal3_kernel<0>, dim3((unsigned)(b * groups)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, gamma, beta, (unsigned short*)out, t, c, groups, eps);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? R50_OK : fail(nullptr, R50_ERR_HIP, std::string("r50_op_gn_relu_causal3: ") + hipGetErrorString(e));
}

// ---- ColorJitter variant (kernels.h: cj_*_kernel) ----
int r50_op_color_jitter_u8(const void* frames_u8, int t, int hw, const int* order4, float brightness, float contrast, float saturation,
                           float hue, int normalize, float* out_f32, float* scratch_means, void* stream) {
    if (!frames_u8 || !order4 || !out_f32 || !scratch_means || t < 1 || hw < 1)
        return fail(nullptr, R50_ERR_INVALID, "r50_op_color_jitter_u8: invalid arguments");
    int seen = 0;
    for (int i = 0; i < 4; ++i) {
        if (order4[i] < 0 || order4[i] > 3) return fail(nullptr, R50_ERR_INVALID, "r50_op_color_jitter_u8: order entries must be 0..3");
        seen |= 1 << order4[i];
    }
    if (seen != 15) return fail(nullptr, R50_ERR_INVALID, "r50_op_color_jitter_u8: order must be a permutation of 0..3");
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)t * 3 * hw;
    const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(cj_from_u8_kernel, dim3(grid), dim3(256), 0, s, (const unsigned char*)frames_u8, out_f32, n);
    const float factor[4] = {brightness, contrast, saturation, hue};
    for (int i = 0; i < 4; ++i) {
        const int op = order4[i];
        if (op == 3 && hue == 0.0f) continue;                       // adjust_hue returns its input for a zero factor
        if (op == 1) hipLaunchKernelGGL(cj_gray_mean_kernel, dim3(t), dim3(1024), 0, s, out_f32, hw, scratch_means);
        hipLaunchKernelGGL(cj_apply_kernel, dim3(grid), dim3(256), 0, s, out_f32, t, hw, op, factor[op], scratch_means);
    }
    if (normalize)
        hipLaunchKernelGGL(cj_normalize_kernel, dim3(grid), dim3(256), 0, s, out_f32, (long long)t * 3, hw, 0.485f, 0.456f, 0.406f, 0.229f,
                           0.224f, 0.225f);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? R50_OK : fail(nullptr, R50_ERR_HIP, std::string("r50_op_color_jitter_u8: ") + hipGetErrorString(e));
}

// ---- lifting head, backward + optimizer (kernels.h, "Lifting head, backward + optimizer") ----
static unsigned ew_grid(long long n) { return (unsigned)std::min<long long>(std::max<long long>((n + 255) / 256, 1), 256 * 32); }
static int ew_done(const char* what) {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? R50_OK : fail(nullptr, R50_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define R50_ET_LAUNCH(kern, grid, block, stream, ...)                                                              \
    do {                                                                                                           \
        if (et) hipLaunchKernelGGL(kern<1>, grid, block, 0, (hipStream_t)stream, __VA_ARGS__);                      \
        else hipLaunchKernelGGL(kern<0>, grid, block, 0, (hipStream_t)stream, __VA_ARGS__);                         \
    } while (0)

int r50_op_transpose16(const void* src, int rows, int cols, void* dst, int ld, void* stream) {
    if (!src || !dst || rows < 1 || cols < 1 || ld < rows) return fail(nullptr, R50_ERR_INVALID, "r50_op_transpose16: invalid arguments");
    hipLaunchKernelGGL(transpose16_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)src, (unsigned short*)dst, rows, cols, ld);
    return ew_done("r50_op_transpose16");
}

int r50_op_mask_scale(void* x, const void* mask_u8, float scale, int64_t n, int et, void* stream) {
    if (!x || !mask_u8 || n < 1 || (et != 0 && et != 1)) return fail(nullptr, R50_ERR_INVALID, "r50_op_mask_scale: invalid arguments");
    R50_ET_LAUNCH(mask_scale_kernel, dim3(ew_grid(n)), dim3(256), stream, (unsigned short*)x, (const unsigned char*)mask_u8, scale, (long long)n);
    return ew_done("r50_op_mask_scale");
}

int r50_op_relu_bwd(void* dy, const void* act, float scale, int64_t n, int et, void* stream) {
    if (!dy || !act || n < 1 || (et != 0 && et != 1)) return fail(nullptr, R50_ERR_INVALID, "r50_op_relu_bwd: invalid arguments");
    R50_ET_LAUNCH(relu_bwd_kernel, dim3(ew_grid(n)), dim3(256), stream, (unsigned short*)dy, (const unsigned short*)act, scale, (long long)n);
    return ew_done("r50_op_relu_bwd");
}

int r50_op_colsum(const void* x, int64_t rows, int cols, int ld, float scale, float* out, int accumulate, int et, void* stream) {
    if (!x || !out || rows < 1 || cols < 1 || ld < cols || (et != 0 && et != 1)) return fail(nullptr, R50_ERR_INVALID, "r50_op_colsum: invalid arguments");
    R50_ET_LAUNCH(colsum_kernel, dim3((cols + 63) / 64), dim3(1024), stream, (const unsigned short*)x, (long long)rows, cols, ld, scale, out, accumulate);
    return ew_done("r50_op_colsum");
}

int r50_op_colsum_f32(const float* x, int64_t rows, int cols, float scale, float* out, int accumulate, void* stream) {
    if (!x || !out || rows < 1 || cols < 1) return fail(nullptr, R50_ERR_INVALID, "r50_op_colsum_f32: invalid arguments");
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((cols + 63) / 64), dim3(64), 0, (hipStream_t)stream, x, (long long)rows, cols, scale, out, accumulate);
    return ew_done("r50_op_colsum_f32");
}

int r50_op_grad_accum(const void* src, float scale, float* dst, int64_t n, int accumulate, int et, void* stream) {
    if (!src || !dst || n < 1 || (et != 0 && et != 1)) return fail(nullptr, R50_ERR_INVALID, "r50_op_grad_accum: invalid arguments");
    R50_ET_LAUNCH(grad_accum_kernel, dim3(ew_grid(n)), dim3(256), stream, (const unsigned short*)src, scale, dst, (long long)n, accumulate);
    return ew_done("r50_op_grad_accum");
}

int r50_op_mse_loss_grad(const float* y, const float* gt, int64_t n, float loss_scale, float* dy, float* loss2, void* stream) {
    if (!y || !gt || !dy || !loss2 || n < 3 || n % 3) return fail(nullptr, R50_ERR_INVALID, "r50_op_mse_loss_grad: invalid arguments");
    hipLaunchKernelGGL(mse_loss_grad_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, y, gt, (long long)n, loss_scale, dy, loss2);
    return ew_done("r50_op_mse_loss_grad");
}

int r50_op_gn_relu_causal3_bwd(const void* dr, const void* x, int b, int t, int c, int groups, const float* gamma, const float* beta,
                               float eps, const void* add, void* dx, float* dgamma_part, float* dbeta_part, int et, void* stream) {
    if (!dr || !x || !gamma || !beta || !dx || !dgamma_part || !dbeta_part || b < 1 || t < 1 || c < 1 || groups < 1 || c % groups ||
        c / groups > 256 || (et != 0 && et != 1))
        return fail(nullptr, R50_ERR_INVALID, "r50_op_gn_relu_causal3_bwd: invalid arguments");
    R50_ET_LAUNCH(gn_relu_causal3_bwd_kernel, dim3((unsigned)(b * groups)), dim3(256), stream, (const unsigned short*)dr,
                  (const unsigned short*)x, gamma, beta, (const unsigned short*)add, (unsigned short*)dx, dgamma_part, dbeta_part, t, c,
                  groups, eps);
    return ew_done("r50_op_gn_relu_causal3_bwd");
}

int r50_op_check_finite(const float* g, int64_t n, int* found, void* stream) {
    if (!g || !found || n < 1) return fail(nullptr, R50_ERR_INVALID, "r50_op_check_finite: invalid arguments");
    hipLaunchKernelGGL(check_finite_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, g, (long long)n, found);
    return ew_done("r50_op_check_finite");
}

int r50_op_check_overflow16(const void* x, int64_t n, int* found, int et, void* stream) {
    if (!x || !found || n < 1 || (et != 0 && et != 1)) return fail(nullptr, R50_ERR_INVALID, "r50_op_check_overflow16: invalid arguments");
    R50_ET_LAUNCH(check_overflow16_kernel, dim3(ew_grid(n)), dim3(256), stream, (const unsigned short*)x, (long long)n, found);
    return ew_done("r50_op_check_overflow16");
}

int r50_op_adamw(float* p, float* m, float* v, const float* g, void* p16, int64_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int step, const int* found_inf, int et, void* stream) {
    if (!p || !m || !v || !g || !p16 || n < 1 || step < 1 || (et != 0 && et != 1)) return fail(nullptr, R50_ERR_INVALID, "r50_op_adamw: invalid arguments");
    const double bc1 = 1.0 - std::pow((double)beta1, step), bc2 = 1.0 - std::pow((double)beta2, step);
    R50_ET_LAUNCH(adamw_kernel, dim3(ew_grid(n)), dim3(256), stream, p, m, v, g, (unsigned short*)p16, (long long)n, lr, beta1, beta2, eps,
                  weight_decay, (float)bc1, (float)std::sqrt(bc2), found_inf);
    return ew_done("r50_op_adamw");
}

int64_t r50_stem_scratch_bytes(int n) { return (int64_t)STEM_W_BYTES + (int64_t)n * STEM_HP * STEM_WP * 8; }

int r50_op_stem(const float* x, int n, const float* w_host, const float* bias_dev, void* scratch, void* y, void* stream) {
    if (!x || !w_host || !bias_dev || !scratch || !y || n < 1) return fail(nullptr, R50_ERR_INVALID, "r50_op_stem: invalid arguments");
    std::vector<uint16_t> pk;
    pack_stem(w_host, pk);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(nullptr, hipMemcpyAsync(scratch, pk.data(), STEM_W_BYTES, hipMemcpyHostToDevice, s));
    HIP_TRY(nullptr, hipStreamSynchronize(s));   // pk is a host temporary
    char* xp = (char*)scratch + STEM_W_BYTES;
    HIP_TRY(nullptr, launch_stem_pack(x, xp, n, s));
    HIP_TRY(nullptr, launch_stem_conv(xp, scratch, bias_dev, y, n, s));
    return R50_OK;
}

int r50_op_maxpool(const void* x, int n, int h, int w, int c, void* y, void* stream) {
    if (!x || !y || n < 1 || h < 1 || w < 1 || c < 8 || c % 8) return fail(nullptr, R50_ERR_INVALID, "r50_op_maxpool: invalid arguments");
    HIP_TRY(nullptr, launch_maxpool(x, y, n, h, w, c, (hipStream_t)stream));
    return R50_OK;
}

int r50_op_avgpool(const void* x, int n, int hw, int c, float* y, void* stream) {
    if (!x || !y || n < 1 || hw < 1 || c < 8 || c % 8) return fail(nullptr, R50_ERR_INVALID, "r50_op_avgpool: invalid arguments");
    HIP_TRY(nullptr, launch_avgpool(x, y, n, hw, c, (hipStream_t)stream));
    return R50_OK;
}

}  // extern "C"
