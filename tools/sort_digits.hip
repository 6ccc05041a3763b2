// sort_digits.hip -- the library's radix sort of 64-bit keys on 19 bits (C3's subject | slot) with digits of 8 bits (the
// default: 3 passes) against wider digits (10: 2 passes), and different rank algorithms.
//   hipcc --offload-arch=gfx950 -O3 tools/sort_digits.hip -o /tmp/sortd && /tmp/sortd
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template<class Config>
static int run(const char *what, uint64_t *in, uint64_t *out, size_t n, int b0, int b1, const std::vector<uint64_t> &expect)
{
    size_t tb = 0; void *tmp = nullptr;
    CHK(rocprim::radix_sort_keys<Config>(nullptr, tb, in, out, n, b0, b1, 0));
    CHK(hipMalloc(&tmp, tb));
    CHK(rocprim::radix_sort_keys<Config>(tmp, tb, in, out, n, b0, b1, 0));
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0, 0));
    for (int r = 0; r < 10; r++) CHK(rocprim::radix_sort_keys<Config>(tmp, tb, in, out, n, b0, b1, 0));
    CHK(hipEventRecord(e1, 0)); CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h(n);
    CHK(hipMemcpy(h.data(), out, n * 8, hipMemcpyDeviceToHost));
    printf("%-44s %7.3f ms per sort of %zu keys on bits [%d, %d)  %s\n", what, ms / 10, n, b0, b1, h == expect ? "ok" : "WRONG");
    CHK(hipFree(tmp));
    return 0;
}

int main()
{
    const size_t n = 47u << 20;
    const int b0 = 25 + 20, b1 = b0 + 19;         // value bits + scan-position bits below
    std::vector<uint64_t> h(n);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; i++) { x ^= x >> 12; x ^= x << 25; x ^= x >> 27; h[i] = (x * 0x2545F4914F6CDD1Dull); }
    std::vector<uint64_t> expect = h;
    std::stable_sort(expect.begin(), expect.end(), [&](uint64_t a, uint64_t b) { return ((a >> b0) & 0x7ffff) < ((b >> b0) & 0x7ffff); });
    uint64_t *in, *out;
    CHK(hipMalloc(&in, n * 8)); CHK(hipMalloc(&out, n * 8));
    CHK(hipMemcpy(in, h.data(), n * 8, hipMemcpyHostToDevice));
    using namespace rocprim;
    run<default_config>("default", in, out, n, b0, b1, expect);
#define OS(HB, HI, SB, SI, BITS, ALG) radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<HB, HI>, kernel_config<SB, SI>, BITS, block_radix_rank_algorithm::ALG>>
    run<OS(256, 12, 256, 12, 8, match)>("8 bits, 256 x 12, match", in, out, n, b0, b1, expect);
    run<OS(256, 12, 512, 12, 8, match)>("8 bits, 512 x 12, match", in, out, n, b0, b1, expect);
    run<OS(256, 12, 256, 12, 10, match)>("10 bits, 256 x 12, match", in, out, n, b0, b1, expect);
    run<OS(256, 12, 512, 12, 10, match)>("10 bits, 512 x 12, match", in, out, n, b0, b1, expect);
    run<OS(256, 12, 1024, 8, 10, match)>("10 bits, 1024 x 8, match", in, out, n, b0, b1, expect);
    run<OS(256, 12, 512, 16, 10, match)>("10 bits, 512 x 16, match", in, out, n, b0, b1, expect);
    run<OS(256, 12, 256, 12, 7, match)>("7 bits, 256 x 12, match", in, out, n, b0, b1, expect);
    run<OS(256, 12, 256, 12, 10, basic)>("10 bits, 256 x 12, basic", in, out, n, b0, b1, expect);
    run<OS(256, 12, 512, 12, 10, basic_memoize)>("10 bits, 512 x 12, basic_memoize", in, out, n, b0, b1, expect);
    return 0;
}
