// write_microbench.hip -- what scattered aligned write pieces cost on this chip: 256 workgroups x 512 private
// streams each (the binning kernel's output shape), pieces of S bytes appended round-robin to the streams.
//   hipcc --offload-arch=gfx950 -O3 tools/write_microbench.hip -o tools/bin/wrmb && tools/bin/wrmb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// S = piece bytes (16-byte stores by S/16 consecutive lanes); streams of `cap` bytes, writer-major
template <int S>
__global__ void __launch_bounds__(1024) wr(uint8_t *out, size_t cap, int iters)
{
    constexpr int LPP = S / 16, G = 1024 / LPP, SPG = 512 / (G < 512 ? G : 512);   // lanes per piece, groups, streams per group
    const int tid = threadIdx.x, g = tid / LPP, l = tid % LPP;
    uint8_t *base = out + (size_t)blockIdx.x * 512 * cap;
    uint4 v = make_uint4(tid, blockIdx.x, 3, 4);
    for (int t = 0; t < iters; t++) {
        // groups beyond 512 (S = 16, 32... G > 512) double up on the streams: two pieces per stream and round
        const int stream = (G <= 512) ? (g + (t % SPG) * G) : (g % 512);
        const size_t off = (G <= 512) ? (size_t)(t / SPG) * S : ((size_t)t * (G / 512) + (g / 512)) * S;
        *reinterpret_cast<uint4 *>(base + (size_t)stream * cap + off + (size_t)l * 16) = v;
        v.x += 1;
    }
}
// the binning kernel's line: 64 bytes + 32 bytes (4 lanes: 16 + 8 bytes each) in blocks of 384 bytes
__global__ void __launch_bounds__(1024) wr_line(uint8_t *out, size_t cap, int iters, int mode)
{
    const int tid = threadIdx.x, g = tid / 4, l = tid % 4;          // 256 groups, 2 streams each
    uint8_t *base = out + (size_t)blockIdx.x * 512 * cap;
    uint4 v = make_uint4(tid, blockIdx.x, 3, 4);
    for (int t = 0; t < iters; t++) {
        const int stream = g + (t & 1) * 256; const size_t line = (size_t)(t >> 1);
        uint8_t *blk = base + (size_t)stream * cap + (line >> 2) * 384;
        if (mode & 1) *reinterpret_cast<uint4 *>(blk + (line & 3) * 64 + l * 16) = v;
        if (mode & 2) *reinterpret_cast<uint2 *>(blk + 256 + (line & 3) * 32 + l * 8) = make_uint2(v.x, v.y);
        v.x += 1;
    }
}
template <int S> int run(uint8_t *buf, size_t cap, size_t total)
{
    constexpr int LPP = S / 16, G = 1024 / LPP;
    const size_t per_iter = (size_t)256 * G * S;
    const int iters = (int)(total / per_iter);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(wr<S>, dim3(256), dim3(1024), 0, 0, buf, cap, iters);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(wr<S>, dim3(256), dim3(1024), 0, 0, buf, cap, iters);
    CHK(hipEventRecord(e1, 0)); CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)per_iter * iters;
    printf("pieces of %4d B: %6.2f GB in %6.2f ms = %5.0f GB/s, %5.1f G pieces/s\n", S, bytes / 1e9, ms, bytes / ms / 1e6, bytes / S / ms / 1e6);
    return 0;
}
int main()
{
    const size_t cap = 192 * 1024, total = (size_t)256 * 512 * cap;      // 25.8 GB buffer
    uint8_t *buf; CHK(hipMalloc(&buf, total + 4096));
    CHK(hipMemset(buf, 0, total));
    const size_t want = (size_t)12 << 30;
    run<32>(buf, cap, want); run<64>(buf, cap, want); run<128>(buf, cap, want); run<256>(buf, cap, want); run<512>(buf, cap, want); run<1024>(buf, cap, want);
    for (int mode = 1; mode <= 3; mode++) {
        const int iters = (int)(want / ((size_t)256 * 256 * 96));
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(wr_line, dim3(256), dim3(1024), 0, 0, buf, cap, iters, mode);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(wr_line, dim3(256), dim3(1024), 0, 0, buf, cap, iters, mode);
        CHK(hipEventRecord(e1, 0)); CHK(hipDeviceSynchronize());
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)iters * 256 * 256 * ((mode & 1 ? 64 : 0) + (mode & 2 ? 32 : 0));
        printf("lines (%s%s): %6.2f GB in %6.2f ms = %5.0f GB/s, %5.1f G lines/s\n", mode & 1 ? "64 B hi " : "", mode & 2 ? "32 B idx" : "", bytes / 1e9, ms, bytes / ms / 1e6, (double)iters * 65536 / ms / 1e6);
    }
    return 0;
}
