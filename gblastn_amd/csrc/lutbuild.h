// lutbuild.h -- device-side construction of a query batch's lookup structures (lutbuild.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#ifndef GBN_LUT_RADIX_BITS
#define GBN_LUT_RADIX_BITS 6        // digit of the builder's radix sort (lutbuild.hip)
#endif

namespace gbn {

// passes lut_sort takes for keys of key_bits bits (the top bit is the "no word here" flag, dropped by the first pass)
__host__ __device__ inline int lut_sort_passes(int key_bits) { const int p = (key_bits - 1 + GBN_LUT_RADIX_BITS - 1) / GBN_LUT_RADIX_BITS; return p < 1 ? 1 : p; }

struct LutBuild {                           // device pointers throughout
    const uint8_t *q8; int32_t qlen;        // concatenated query, position 0 (sentinel padding either side)
    const int32_t *seg_left, *seg_right; int32_t nseg;      // indexed stretches, ascending, non-empty
    int32_t lut, word, q_bits, descending, onebyte_mode;
    int64_t ncells;
    uint32_t *count;                        // ncells + 1, zero on entry; nullptr: no counting (cell_start comes from lut_cell_starts)
    uint32_t *keys_a, *keys_b, *vals_a, *vals_b;            // qlen entries each: {cell (1 << 2 lut: no word here), offset} by list index, sorted on the cell
    uint32_t *cell_start;                   // ncells + 1
    uint32_t *cellw, *cellt, *pv; unsigned long long *ent;
    uint16_t *sidet; uint32_t *side_start; int32_t nbins, cbits;     // bins of 2^cbits cells (GBN_BIN_CBITS)
};

hipError_t lut_enumerate(const LutBuild &b, hipStream_t st);
hipError_t lut_overflow_cells(const LutBuild &b, unsigned long long *out, hipStream_t st);
hipError_t lut_sort(void *tmp, size_t &bytes, const LutBuild &b, int64_t n, int key_bits, hipStream_t st, uint32_t *n_valid_out = nullptr);
hipError_t lut_cell_starts(const LutBuild &b, const uint32_t *n_valid, hipStream_t st);       // cell_start from keys_b sorted (count == nullptr builds)
hipError_t lut_scan(void *tmp, size_t &bytes, const uint32_t *in, uint32_t *out, int64_t n, hipStream_t st);
hipError_t lut_entries(const LutBuild &b, int64_t n, hipStream_t st);
hipError_t lut_cells_side(const LutBuild &b, hipStream_t st);      // cellw, cellt, sidet, side_start: a workgroup per bin; side_start[bin] = bin x GBN_BIN_SIDE
hipError_t lut_pv(const LutBuild &b, hipStream_t st);
// rank form of the cell table (scan_fold_kernel): popc[0 .. nwords] = set bits per presence word (0 at nwords), then --
// with prefix = their exclusive sums -- pvx[w] = {pv[w], prefix[w]} and pstart[rank of a present cell] = its cell_start
hipError_t lut_rank_count(const uint32_t *pv, int64_t nwords, uint32_t *popc, hipStream_t st);
hipError_t lut_rank_fill(const uint32_t *pv, const uint32_t *prefix, const uint32_t *cell_start, int64_t ncells, int64_t nwords,
                         uint32_t *pvx, uint32_t *pstart, hipStream_t st);
// 2-bit packed copy of the query (4 bases per byte, base 0 in bits 7..6) and its "matches nothing" bitmap (MSB first)
hipError_t lut_pack_q4(const uint8_t *qbuf, int64_t qbuf_len, uint8_t *q4, int64_t plane, hipStream_t st);
hipError_t lut_pack_query(const uint8_t *qbuf, int64_t qbuf_len, int64_t first, int64_t n, uint8_t *q2, uint8_t *qinv, hipStream_t st);

}  // namespace gbn
