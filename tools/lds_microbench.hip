// lds_microbench.hip -- what the binning kernel's LDS steps cost on this chip (cycles per wave-instruction
// with 16 waves per CU issuing), to decide between ranking schemes.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/lds_microbench.hip -o /tmp/ldsmb && /tmp/ldsmb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, int NBINS>
__global__ void __launch_bounds__(1024) k(uint32_t *out, unsigned long long *cyc, int iters)
{
    __shared__ uint32_t s_hist[2048];
    __shared__ uint2 s_rec[8192 + 512 * 16];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2048; i += 1024) s_hist[i] = 0;
    __syncthreads();
    uint32_t x = tid * 2654435761u + blockIdx.x * 40503u + 12345u, acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        uint32_t b[8];
        #pragma unroll
        for (int j = 0; j < 8; j++) { x = x * 1664525u + 1013904223u; b[j] = (x >> 11) & (NBINS - 1); }
        if (MODE == 0) {            // returning atomic add, random counters
            uint32_t r[8];
            #pragma unroll
            for (int j = 0; j < 8; j++) r[j] = atomicAdd(&s_hist[b[j]], 1u);
            #pragma unroll
            for (int j = 0; j < 8; j++) acc += r[j];
        } else if (MODE == 1) {     // non-returning atomic add
            #pragma unroll
            for (int j = 0; j < 8; j++) __hip_atomic_fetch_add(&s_hist[b[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 2) {     // random 8-byte scatter
            #pragma unroll
            for (int j = 0; j < 8; j++) s_rec[(b[j] * 16 + ((x >> (3 + j)) & 15)) & 8191] = make_uint2(x, b[j]);
        } else if (MODE == 3) {     // random 4-byte read
            #pragma unroll
            for (int j = 0; j < 8; j++) acc += s_hist[b[j]];
        } else if (MODE == 4) {     // random 8-byte read
            #pragma unroll
            for (int j = 0; j < 8; j++) acc += s_rec[(b[j] * 16 + j) & 8191].x;
        } else if (MODE == 5) {     // random 4-byte write
            #pragma unroll
            for (int j = 0; j < 8; j++) s_hist[b[j]] = x;
        } else if (MODE == 6) {     // returning atomic on a 16-bit-packed pair: two bins per word (64-bit add with return)
            unsigned long long *h64 = reinterpret_cast<unsigned long long *>(s_hist);
            unsigned long long r[8];
            #pragma unroll
            for (int j = 0; j < 8; j++) r[j] = atomicAdd(&h64[b[j] >> 1], (b[j] & 1) ? (1ull << 32) : 1ull);
            #pragma unroll
            for (int j = 0; j < 8; j++) acc += (uint32_t)r[j];
        } else if (MODE == 8 || MODE == 9) {     // returning atomic add, then a barrier (8) or a barrier every 4th time (9): waves in lockstep, as between the phases of a kernel
            uint32_t r[8];
            #pragma unroll
            for (int j = 0; j < 8; j++) r[j] = atomicAdd(&s_hist[b[j]], 1u);
            #pragma unroll
            for (int j = 0; j < 8; j++) acc += r[j];
            if (MODE == 8 || (it & 3) == 3) __syncthreads();
        } else if (MODE == 7) {     // no LDS at all: the loop overhead
            #pragma unroll
            for (int j = 0; j < 8; j++) acc += b[j];
        }
        if (MODE == 0 || MODE == 1 || MODE == 6 || MODE == 8 || MODE == 9) { if ((it & 63) == 63) { __syncthreads(); for (int i = tid; i < 2048; i += 1024) s_hist[i] = 0; __syncthreads(); } }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 1024 + tid] = acc + s_hist[tid] + s_rec[tid].x;
}

// ---- round 5: the LDS side of the binning kernel's tile loop, step by step (MODE bits), 8,192 records per iteration as in a tile:
//   1  the rank atomics (returning; bit 64: without return)      2  barrier + 512 owner threads read and zero the histogram + barrier
//   4  the descriptor read (random 8 bytes per record)           8  the two scatter writes per record (4 + 2 bytes, random slots)
//  16  barrier + the store phase's reads (16 + 8 bytes per 4 records, sequential)
// Time per iteration / 128 = ns per "rank atomic slot" of the kernel's account (12.6 ns in the kernel).
template <int MODE>
__global__ void __launch_bounds__(1024) tile_loop(uint32_t *out, int iters)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_hi[8192 + 512 * 32];
    __shared__ __attribute__((aligned(16))) uint16_t s_ix[8192 + 512 * 32];
    __shared__ uint32_t s_hist[512];
    __shared__ uint2 s_pk[512];
    const int tid = threadIdx.x;
    if (tid < 512) { s_hist[tid] = 0; s_pk[tid] = make_uint2(8192 + tid * 32, 0); }
    __syncthreads();
    uint32_t x = tid * 2654435761u + blockIdx.x * 40503u + 12345u, acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t b[8], r[8];
        #pragma unroll
        for (int j = 0; j < 8; j++) { x = x * 1664525u + 1013904223u; b[j] = (x >> 11) & 511u; r[j] = (x >> 5) & 31u; }
        if (MODE & 1) {
            if (MODE & 64) {
                #pragma unroll
                for (int j = 0; j < 8; j++) __hip_atomic_fetch_add(&s_hist[b[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                #pragma unroll
                for (int j = 0; j < 8; j++) r[j] = atomicAdd(&s_hist[b[j]], 1u) & 31u;
            }
        }
        if (MODE & 2) {
            __syncthreads();
            if (tid < 512) { acc += s_hist[tid]; s_hist[tid] = 0; }
            __syncthreads();
        }
        uint2 pk[8];
        #pragma unroll
        for (int j = 0; j < 8; j++) pk[j] = make_uint2(8192 + b[j] * 32, 0);
        if (MODE & 4) {
            #pragma unroll
            for (int j = 0; j < 8; j++) pk[j] = s_pk[b[j]];
        }
        if (MODE & 8) {
            #pragma unroll
            for (int j = 0; j < 8; j++) { const uint32_t slot = pk[j].x + r[j]; s_hi[slot] = x + j; s_ix[slot] = (uint16_t)(tid * 8 + j); }
        }
        if (MODE & 16) {
            __syncthreads();
            #pragma unroll
            for (int q = 0; q < 2; q++) {       // 2,048 quarter lines per tile
                const uint32_t src = (uint32_t)(8192 + ((tid + q * 1024 + it * 8) & 4095) * 4);
                const uint4 h = *reinterpret_cast<const uint4 *>(&s_hi[src]);
                const uint2 v = *reinterpret_cast<const uint2 *>(&s_ix[src]);
                acc += h.x ^ h.w ^ v.x;
            }
        }
        #pragma unroll
        for (int j = 0; j < 8; j++) acc += r[j] + pk[j].y;
    }
    out[blockIdx.x * 1024 + tid] = acc + s_hist[tid & 511];
}
template <int MODE> int run_tile(const char *what, uint32_t *out, int iters)
{
    hipLaunchKernelGGL((tile_loop<MODE>), dim3(256), dim3(1024), 0, 0, out, 16);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((tile_loop<MODE>), dim3(256), dim3(1024), 0, 0, out, iters);
    CHK(hipEventRecord(e1, 0));
    CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("tile loop: %-78s %7.3f us per tile = %6.2f ns per rank-atomic slot (128 per tile); a C2 pass has 1,402 tiles per CU: %5.2f ms\n",
           what, ms * 1e3 / iters, ms * 1e6 / ((double)iters * 128.0), ms / iters * 1402.0);
    return 0;
}

template <int MODE, int NBINS> int run(const char *what, uint32_t *out, unsigned long long *cyc, int iters)
{
    const int grid = 256;
    hipLaunchKernelGGL((k<MODE, NBINS>), dim3(grid), dim3(1024), 0, 0, out, cyc, 16);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<MODE, NBINS>), dim3(grid), dim3(1024), 0, 0, out, cyc, iters);
    CHK(hipEventRecord(e1, 0));
    CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid);
    CHK(hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= grid;
    // per CU: 16 waves x 8 instructions per iteration
    printf("%-52s bins %4d: %7.3f ms, %9.0f counter ticks per workgroup, %.2f ns per wave-instruction per CU (16 waves x 8 per iteration)\n",
           what, NBINS, ms, avg, ms * 1e6 / ((double)iters * 128.0));
    return 0;
}

int main()
{
    uint32_t *out; unsigned long long *cyc;
    CHK(hipMalloc(&out, 256 * 1024 * 4)); CHK(hipMalloc(&cyc, 256 * 8));
    const int iters = 20000;
    run<7, 512>("loop overhead (no LDS)", out, cyc, iters);
    run<0, 512>("ds_add_rtn_u32 random", out, cyc, iters);
    run<8, 512>("ds_add_rtn_u32 random + barrier per 8", out, cyc, iters);
    run<9, 512>("ds_add_rtn_u32 random + barrier per 32", out, cyc, iters);
    run<0, 128>("ds_add_rtn_u32 random", out, cyc, iters);
    run<0, 2048>("ds_add_rtn_u32 random", out, cyc, iters);
    run<1, 512>("ds_add_u32 (no return) random", out, cyc, iters);
    run<6, 512>("ds_add_rtn_u64 (two bins per word) random", out, cyc, iters);
    run<2, 512>("ds_write_b64 random scatter", out, cyc, iters);
    run<5, 512>("ds_write_b32 random", out, cyc, iters);
    run<3, 512>("ds_read_b32 random", out, cyc, iters);
    run<4, 512>("ds_read_b64 random", out, cyc, iters);
    const int tiles = 4000;
    run_tile<0>("nothing (key arithmetic of the loop)", out, tiles);
    run_tile<1>("rank atomics", out, tiles);
    run_tile<1 | 64>("rank atomics without return", out, tiles);
    run_tile<1 | 2>("rank atomics | barrier, owners read + zero, barrier", out, tiles);
    run_tile<1 | 64 | 2>("atomics without return | barrier, owners, barrier", out, tiles);
    run_tile<1 | 2 | 4>("rank atomics | owners | descriptor reads", out, tiles);
    run_tile<1 | 2 | 4 | 8>("rank atomics | owners | descriptor reads + scatter writes", out, tiles);
    run_tile<1 | 2 | 4 | 8 | 16>("rank atomics | owners | descriptors + scatter | barrier, store-phase reads", out, tiles);
    run_tile<2 | 4 | 8 | 16>("the same without the rank atomics", out, tiles);
    run_tile<1 | 2 | 8 | 16>("the same without the descriptor reads", out, tiles);
    run_tile<1 | 2 | 4 | 16>("the same without the scatter writes", out, tiles);
    return 0;
}
