// Micro-benchmark: does the LANE ORDER of a 16-row x 64-B access matter to the address unit?
// Every wave-instruction reads (or writes) the same 16 rows x 64 B (rows `stride` bytes apart), with
//   mode 0 "fragment": lane = chunk*16 + row   (the MFMA B-fragment / accumulator order: consecutive lanes -> consecutive rows)
//   mode 1 "quad":     lane = row*4 + chunk    (consecutive lanes -> consecutive 16-B chunks of one row)
//   mode 2 "linear":   lane -> 16 B at lane*16 of one contiguous 1-KiB piece (reference)
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/lane_order scripts/micro/lane_order.hip && /tmp/lane_order
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool STORE>
__global__ __launch_bounds__(512) void k(char* buf, unsigned bytes, unsigned stride, int iters, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, bytes, 0x00020000);
    unsigned in_piece;
    if (MODE == 0) in_piece = (lane & 15) * stride + (lane >> 4) * 16;
    else if (MODE == 1) in_piece = (lane >> 2) * stride + (lane & 3) * 16;
    else in_piece = lane * 16;
    const unsigned piece_bytes = (MODE == 2) ? 1024u : 16u * stride;
    const unsigned npieces = bytes / piece_bytes;
    unsigned piece = (blockIdx.x * 8 + wave) * 7919u % npieces;
    u32x4 acc = {1, 2, 3, 4};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned off = piece * piece_bytes + in_piece + ((MODE == 2) ? 0u : (unsigned)u * 64u);   // 4 x 64 B of the rows
            if (STORE) __builtin_amdgcn_raw_buffer_store_b128(acc, rs, (MODE == 2) ? off + 0 : off, 0, 0);
            else acc ^= __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            if (MODE == 2) { piece += 1; if (piece >= npieces) piece -= npieces; }
        }
        if (MODE != 2) { piece += 2048 / 1; if (piece >= npieces) piece %= npieces; }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int MODE, bool STORE>
void run(const char* name, char* buf, unsigned bytes, unsigned stride, unsigned* sink) {
    const int iters = 400;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE, STORE><<<256, 512>>>(buf, bytes, stride, iters, sink);
    hipEventRecord(a);
    k<MODE, STORE><<<256, 512>>>(buf, bytes, stride, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double total = 256.0 * 8 * iters * 4 * 1024.0;
    printf("%-5s %-9s footprint %5u MiB stride %4u: %6.2f TB/s chip, %6.1f GB/s per CU, %5.1f cycles@2.4GHz per wave-instruction per CU\n",
           STORE ? "store" : "load", name, bytes >> 20, stride, total / (ms * 1e-3) / 1e12, total / 256 / (ms * 1e-3) / 1e9,
           (ms * 1e-3) * 2.4e9 / (8.0 * iters * 4));
}

int main() {
    char* buf; unsigned* sink;
    hipMalloc(&buf, 1u << 30); hipMemset(buf, 1, 1u << 30); hipMalloc(&sink, 4);
    for (unsigned bytes : {16u << 20, 1u << 30}) {
        for (unsigned stride : {512u, 1024u}) {
            run<0, false>("fragment", buf, bytes, stride, sink);
            run<1, false>("quad", buf, bytes, stride, sink);
            run<0, true>("fragment", buf, bytes, stride, sink);
            run<1, true>("quad", buf, bytes, stride, sink);
        }
        run<2, false>("linear", buf, bytes, 0, sink);
        run<2, true>("linear", buf, bytes, 0, sink);
    }
    hipDeviceSynchronize();
    return 0;
}
