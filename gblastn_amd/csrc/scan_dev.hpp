// scan_dev.hpp -- device helpers shared by the kernel translation units (kernels.hip: direct / sliced scan, seed and
// gapped stages; scan_bin.hip: the key-range partitioned scan): packed-sequence windows, the exact verification of a
// lookup hit to word_size (the reference's mini-extensions), the fingerprint test, per-device kernel attributes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <atomic>
#include "gbn_dev.h"

// Function attributes belong to the (kernel, device) pair: one bit per device, set on the first launch there.
static inline hipError_t raise_dynamic_lds(const void *fn, size_t lds, std::atomic<uint64_t> &done)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

#ifndef GBN_PROBE_MASKS
#define GBN_PROBE_MASKS 1   // probe kernel: the tests of a record combined as lane masks in scalar registers (0: the round-3 select chains, A/B)
#endif
#ifndef GBN_PROBE_ABL
#define GBN_PROBE_ABL 0     // ablations of the probe kernel (timing only, wrong results): 1 the streams alone (no table lookup, no test)
#endif
#ifndef GBN_PROBE_FETCH
#define GBN_PROBE_FETCH 1   // 1: the probe kernel fetches a queued item's 16-bit index and cell word (0: the rare kernel does; measured the same sum)
#endif
#ifndef GBN_RARE_CUR4
#define GBN_RARE_CUR4 1     // rare kernel: the run of a record from four cursors in one load (0: two cursors + binary search, A/B)
#endif
#ifndef GBN_PROBE_U
#define GBN_PROBE_U 2        // 16-byte loads per lane and round of the probe kernel (4 records each)
#endif
// stream of (bin, writer): writer-major keeps the 512 streams a binning workgroup appends to
// inside one ~100 MB stretch instead of spreading them over the whole buffer
// (a writer's row = its nb streams one after the other; uniform streams of subcap records, or per-bin capacities: GbnBinParams::bincap)
#define GBN_BINCAP(B, bin) ((B).bincap ? (B).bincap[2 * (size_t)(bin)] : (B).subcap)
#define GBN_BINOFF(B, bin) ((B).bincap ? (size_t)(B).bincap[2 * (size_t)(bin) + 1] : (size_t)(bin) * (B).subcap)
#define GBN_ROW(B) ((B).bincap ? (B).rowsize : (size_t)(B).nb * (B).subcap)
// linear index of record j of stream (bin, writer)
#define GBN_RECIDX(B, bin, writer, j) ((size_t)(writer) * GBN_ROW(B) + GBN_BINOFF(B, bin) + (size_t)(j))
#ifndef GBN_BIN_OCC
#define GBN_BIN_OCC 4       // waves per SIMD the binning kernel is compiled for
#endif

namespace {

__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

// inclusive prefix sum over the 64 lanes of a fully active wave: DPP row shifts and row broadcasts
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false);     // row_shr:4, lanes 4.. of a row
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false);     // row_shr:8, lanes 8.. of a row
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return v;
}


// base `pos` of a packed sequence
__device__ __forceinline__ int base_at(const uint8_t *__restrict__ p, int64_t pos) {
    return (p[pos >> 2] >> (2 * (3 - (int)(pos & 3)))) & 3;
}

// 16 consecutive bases starting at base index `pos` (may be negative relative
// to `p`; the slab is padded) as a big-endian 32-bit word: base pos in bits 31..30
__device__ __forceinline__ uint32_t window16(const uint8_t *__restrict__ p, int64_t pos) {
    int64_t w = pos >> 4;                       // dword index (floor)
    const uint32_t *d = reinterpret_cast<const uint32_t *>(p) + w;
    uint32_t hi = bswap32(d[0]), lo = bswap32(d[1]);
    int sh = 2 * (int)(pos & 15);
    return sh ? ((hi << sh) | (lo >> (32 - sh))) : hi;
}

// 32 consecutive bases of a 2-bit packed sequence starting at base index `pos` (may be negative:
// both the subject slab and the packed query carry padding in front), big-endian in 64 bits
__device__ __forceinline__ uint64_t bases32(const uint8_t *__restrict__ p, int64_t pos) {
    const uint32_t *d = reinterpret_cast<const uint32_t *>(p) + (pos >> 4);
    const uint64_t hi = ((uint64_t)bswap32(d[0]) << 32) | bswap32(d[1]);
    const uint32_t lo = bswap32(d[2]);
    const int sh = 2 * (int)(pos & 15);
    return sh ? ((hi << sh) | ((uint64_t)lo >> (32 - sh))) : hi;
}
// 32 consecutive bits of a bitmap (most significant bit first) starting at bit index `pos`
__device__ __forceinline__ uint32_t bits32(const uint8_t *__restrict__ p, int64_t pos) {
    const uint32_t *d = reinterpret_cast<const uint32_t *>(p) + (pos >> 5);
    const uint32_t hi = bswap32(d[0]), lo = bswap32(d[1]);
    const int sh = (int)(pos & 31);
    return sh ? ((hi << sh) | (lo >> (32 - sh))) : hi;
}

// ---------------------------------------------------------------------------
// exact verification of one lookup hit to word_size; returns ext_left or -1
// ---------------------------------------------------------------------------
__device__ int verify_hit(const GbnScanParams &P, const uint8_t *__restrict__ subj, int32_t slen,
                          int32_t q_off, int32_t s_off)
{
    const uint8_t *q = P.q8;
    const int word = P.word, lut = P.lut, ext_to = word - lut;
    if (P.mode == GBN_EXT_DIRECT) return 0;
    if (P.mode == GBN_EXT_NA) {
        // s_BlastNaExtend: sentinel bytes never equal a 2-bit base
        int ext_left = 0, ext_max = min(ext_to, s_off);
        for (; ext_left < ext_max; ++ext_left)
            if (base_at(subj, s_off - ext_left - 1) != q[q_off - ext_left - 1]) break;
        if (ext_left < ext_to) {
            int need = ext_to - ext_left, so = s_off + lut, r = 0;
            if (so + need > slen) return -1;
            for (; r < need; ++r)
                if (base_at(subj, so + r) != q[q_off + lut + r]) break;
            if (ext_left + r < ext_to) return -1;
        }
        return ext_left;
    }
    // small-query tables compare (code & 3) and clamp at the strand boundaries
    int lo = 0, hi = P.nctx;
    while (lo < hi - 1) { int m = (lo + hi) >> 1; if (P.ctx_off[m] > q_off) hi = m; else lo = m; }
    const int q_start = P.ctx_off[lo], q_range = q_start + P.ctx_len[lo];
    if (P.mode == GBN_EXT_SMALL) {
        int ext_max = min(min(ext_to, s_off), q_off - q_start);
        int rsdl = 4 - (s_off & 3);
        int so = s_off + rsdl, qo = q_off + rsdl, ext_left = 0, ext_right = 0;
        ext_max += rsdl;
        while (ext_left < ext_max && (q[qo - ext_left - 1] & 3) == base_at(subj, so - ext_left - 1)) ext_left++;
        ext_max = min(min(word - ext_left, slen - so), q_range - qo);
        while (ext_right < ext_max && (q[qo + ext_right] & 3) == base_at(subj, so + ext_right)) ext_right++;
        if (ext_left + ext_right < word) return -1;
        return ext_left - rsdl;
    }
    // GBN_EXT_SMALL_ONEBYTE (s_BlastSmallNaExtendAlignedOneByte)
    {
        int ext_left = 0;
        if (s_off > 0 && q_off > 0) {
            int k = 0;
            while (k < 4 && (q[q_off - k - 1] & 3) == base_at(subj, s_off - k - 1)) k++;
            ext_left = min(min(k, ext_to), q_off - q_start);
        }
        if (ext_left < ext_to && (q_off + lut) < P.qlen) {
            int k = 0, so = s_off + lut, qo = q_off + lut;
            while (k < 4) {
                int qb = (qo + k < P.qlen) ? (q[qo + k] & 3) : 0;
                if (qb != base_at(subj, so + k)) break;
                k++;
            }
            int ext_right = min(min(k, slen - so), q_range - qo);
            if (ext_left + ext_right < ext_to) return -1;
        }
        return ext_left;
    }
}

// fingerprint test: a seed that verifies must match the `fl` query bases left
// of the lookup word or the `fr` bases right of it
__device__ __forceinline__ bool fp_pass(uint32_t fp, uint32_t s_left16, uint32_t s_right16, int fl, int fr)
{
    // fp bits [30:15] = 8 bases left of the word (base q-1 in the low pair),
    //    bits [14:1]  = 7 bases right of the word (base q+lut in the high pair)
    if (fp & 1u) return true;                   // forced (see upload: one-byte quirk entries)
    uint32_t ql = (fp >> 15) & 0xffffu;
    uint32_t qr = (fp >> 1) & 0x3fffu;
    uint32_t lmask = (fl >= 8) ? 0xffffu : ((1u << (2 * fl)) - 1);
    bool left = fl > 0 ? (((ql ^ s_left16) & lmask) == 0) : true;
    uint32_t sr = s_right16 >> 18;              // top 7 bases
    uint32_t rmask = (fr >= 7) ? 0x3fffu : (((1u << (2 * fr)) - 1) << (2 * (7 - fr)));
    bool right = fr > 0 ? (((qr ^ sr) & rmask) == 0) : true;
    return left || right;
}

}  // namespace
