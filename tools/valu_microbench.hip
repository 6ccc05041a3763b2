// valu_microbench.hip -- issue cost of the integer VALU instructions the scan / extension / DP kernels are made of,
// in shader cycles per wave64 instruction per SIMD, at 1, 2 and 4 waves per SIMD (256 / 512 / 1024 threads, one
// workgroup per CU).  Settles the "2 or 4 cycles per wave64 op" question behind DESIGN's VALU-roofline arithmetic.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o /tmp/valumb && /tmp/valumb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// eight independent accumulators per instruction kind: nothing waits for a result (dependent-issue latency is the
// MODE 100+ rows, one accumulator)
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY_ADD(i)   "v_add_u32 %" #i ", %" #i ", %8\n"
#define BODY_AND(i)   "v_and_b32 %" #i ", %" #i ", %8\n"
#define BODY_SHL(i)   "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define BODY_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 7\n"
#define BODY_BFE(i)   "v_bfe_u32 %" #i ", %" #i ", 3, 17\n"
#define BODY_ADD3(i)  "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define BODY_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %8\n"
#define BODY_PERM(i)  "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define BODY_CND(i)   "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define BODY_CND64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"
#define BODY_CMPCND(i) "v_cmp_lt_u32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define BODY_OR(i)    "v_or_b32 %" #i ", %" #i ", %8\n"
#define BODY_XOR(i)   "v_xor_b32 %" #i ", %" #i ", %8\n"
#define BODY_SUB(i)   "v_sub_u32 %" #i ", %" #i ", %8\n"
#define BODY_MOV(i)   "v_mov_b32 %" #i ", %8\n"
#define BODY_LSHR(i)  "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define BODY_BFI(i)   "v_bfi_b32 %" #i ", %8, %" #i ", %9\n"
#define BODY_MIN(i)   "v_min_u32 %" #i ", %" #i ", %8\n"
#define BODY_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define BODY_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define BODY_FMA(i)   "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define BODY_ADDF(i)  "v_add_f32 %" #i ", %" #i ", %8\n"
#define BODY_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define BODY_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define BODY_CMP(i)   "v_cmp_lt_u32 vcc, %" #i ", %8\n"
#define BODY_MAX(i)   "v_max_i32 %" #i ", %" #i ", %8\n"
#define BODY_DPP(i)   "v_add_u32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define BODY_LSH64(i) "v_lshlrev_b64 %10, 3, %10\n"
#define BODY_CHAIN(i) "v_add_u32 %0, %0, %8\n"

#define KERNEL(NAME, BODY)                                                                                          \
__global__ void __launch_bounds__(1024) NAME(uint32_t *out, unsigned long long *cyc, int iters)                    \
{                                                                                                                   \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    uint32_t b = blockIdx.x * 2654435761u + 12345u, c = 0x07060504u;                                                \
    unsigned long long w = a0;                                                                                      \
    __syncthreads();                                                                                                \
    const unsigned long long t0 = __builtin_readcyclecounter();                                                     \
    for (int it = 0; it < iters; it++) {                                                                            \
        asm volatile(REP8(BODY) REP8(BODY) REP8(BODY) REP8(BODY) REP8(BODY) REP8(BODY) REP8(BODY) REP8(BODY)       \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)               \
                     : "v"(b), "v"(c), "v"(w) : "vcc", "s10", "s11");                                                             \
    }                                                                                                               \
    const unsigned long long t1 = __builtin_readcyclecounter();                                                     \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                              \
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)w;                    \
}

KERNEL(k_add, BODY_ADD)
KERNEL(k_and, BODY_AND)
KERNEL(k_shl, BODY_SHL)
KERNEL(k_align, BODY_ALIGN)
KERNEL(k_bfe, BODY_BFE)
KERNEL(k_add3, BODY_ADD3)
KERNEL(k_lshlor, BODY_LSHLOR)
KERNEL(k_perm, BODY_PERM)
KERNEL(k_cnd, BODY_CND)
KERNEL(k_mad24, BODY_MAD24)
KERNEL(k_mullo, BODY_MULLO)
KERNEL(k_cmp, BODY_CMP)
KERNEL(k_max, BODY_MAX)
KERNEL(k_dpp, BODY_DPP)
KERNEL(k_chain, BODY_CHAIN)
KERNEL(k_cnd64, BODY_CND64)
KERNEL(k_cmpcnd, BODY_CMPCND)
KERNEL(k_or, BODY_OR)
KERNEL(k_xor, BODY_XOR)
KERNEL(k_sub, BODY_SUB)
KERNEL(k_mov, BODY_MOV)
KERNEL(k_lshr, BODY_LSHR)
KERNEL(k_bfi, BODY_BFI)
KERNEL(k_min, BODY_MIN)
KERNEL(k_andor, BODY_ANDOR)
KERNEL(k_lshladd, BODY_LSHLADD)
KERNEL(k_fma, BODY_FMA)
KERNEL(k_addf, BODY_ADDF)

typedef void (*kern_t)(uint32_t *, unsigned long long *, int);

static int run(const char *what, kern_t k, uint32_t *out, unsigned long long *cyc)
{
    const int grid = 256, iters = 40000;
    for (int threads = 256; threads <= 1024; threads *= 2) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 0, 0, out, cyc, 16);
        CHK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
        CHK(hipEventRecord(e1, 0));
        CHK(hipDeviceSynchronize());
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
        const int waves = threads / 64;
        std::vector<unsigned long long> h(grid * 16);
        CHK(hipMemcpy(h.data(), cyc, grid * 16 * 8, hipMemcpyDeviceToHost));
        double avg = 0; for (int b = 0; b < grid; b++) for (int w = 0; w < waves; w++) avg += (double)h[b * 16 + w];
        avg /= (double)grid * waves;
        const double ninst = (double)iters * 64.0;                      // per wave
        const double per_simd = (double)waves / 4.0;                    // waves per SIMD
        // counter ticks (s_memtime: 100 MHz constant clock on this part if ticks << cycles; printed raw) and wall time
        printf("%-34s %d wave(s)/SIMD: %8.3f ms wall = %6.2f ns per wave-instr per SIMD = %5.2f cycles at 2.4 GHz; counter %9.0f ticks per wave\n",
               what, (int)per_simd, ms, ms * 1e6 / (ninst * per_simd), ms * 1e6 / (ninst * per_simd) * 2.4, avg);
    }
    return 0;
}

int main()
{
    uint32_t *out; unsigned long long *cyc;
    CHK(hipMalloc(&out, 256 * 1024 * 4)); CHK(hipMalloc(&cyc, 256 * 16 * 8));
    run("v_add_u32", k_add, out, cyc);
    run("v_and_b32", k_and, out, cyc);
    run("v_lshlrev_b32", k_shl, out, cyc);
    run("v_alignbit_b32", k_align, out, cyc);
    run("v_bfe_u32", k_bfe, out, cyc);
    run("v_add3_u32", k_add3, out, cyc);
    run("v_lshl_or_b32", k_lshlor, out, cyc);
    run("v_perm_b32", k_perm, out, cyc);
    run("v_cndmask_b32", k_cnd, out, cyc);
    run("v_mad_u32_u24", k_mad24, out, cyc);
    run("v_mul_lo_u32", k_mullo, out, cyc);
    run("v_cmp_lt_u32", k_cmp, out, cyc);
    run("v_max_i32", k_max, out, cyc);
    run("v_add_u32_dpp row_shr:1", k_dpp, out, cyc);
    run("v_add_u32 dependent chain", k_chain, out, cyc);
    run("v_cndmask_b32_e64 (sgpr pair mask)", k_cnd64, out, cyc);
    run("v_cmp_lt_u32 + v_cndmask_b32 (2 instr)", k_cmpcnd, out, cyc);
    run("v_or_b32", k_or, out, cyc);
    run("v_xor_b32", k_xor, out, cyc);
    run("v_sub_u32", k_sub, out, cyc);
    run("v_mov_b32", k_mov, out, cyc);
    run("v_lshrrev_b32", k_lshr, out, cyc);
    run("v_bfi_b32", k_bfi, out, cyc);
    run("v_min_u32", k_min, out, cyc);
    run("v_and_or_b32", k_andor, out, cyc);
    run("v_lshl_add_u32", k_lshladd, out, cyc);
    run("v_fma_f32", k_fma, out, cyc);
    run("v_add_f32", k_addf, out, cyc);
    return 0;
}
