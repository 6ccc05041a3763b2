// scan_runs.hip -- the SORTED form of the scan records (gfx950 / CDNA4).
//
// The records the binning kernel writes (scan_bin.hip) depend on the shard, the subject range and the shape of the lookup
// table, not on the queries, and the record cache keeps them.  In stream form a pass still reads every record to find the
// ones whose cell the batch occupies (45 % of the cells for a 5 Mb megablast batch, 0.01 % for one query).  Here a cached
// set is sorted by cell ONCE -- runs_count_kernel, a prefix sum, runs_split_kernel, runs_place_kernel -- and
// probe_runs_kernel walks the cell table in cell order and reads the runs of the occupied cells only: the reference's
// "is the word present" test in front of every table access (MB_ACCESS_HITS / s_BlastMBLookupRetrieve,
// CORE/blast_nascan.c:1413-1461) as a decision about which bytes to fetch.  A record shrinks to 16 subject bits + its
// position id (the cell is the run), and the rare kernel no longer searches stream cursors for a record's tile.
// Integer work only: no MFMA.
#include "scan_dev.hpp"
#include "lutbuild.h"

namespace {
// hi word of a stream record -> cell inside its bin (pads excluded by the caller)
__device__ __forceinline__ uint32_t cell_in_bin(uint32_t hi, int cbits) { return hi & ((1u << cbits) - 1u); }
__device__ __forceinline__ bool is_pad(uint32_t hi, uint32_t padlow) { return (hi & 0xffffu) == padlow; }
}  // namespace

// ---------------------------------------------------------------------------------------------------
// Build, step 1: records per cell.  A workgroup per (bin, part): the bin's cells as counters in LDS, the hi lines of the
// part's streams past them, the counts added to count[] (zero on entry).
extern "C" __global__ void __launch_bounds__(1024)
runs_count_kernel(GbnRunsBuild R, int nparts)
{
    const GbnBinParams &B = R.B;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_cnt[];    // 2^cbits
    const int tid = threadIdx.x;
    const int b = (int)(blockIdx.x / (unsigned)nparts), part = (int)(blockIdx.x % (unsigned)nparts);
    const int cbits = B.cbits;
    const uint32_t ncb = 1u << cbits, padlow = GBN_REC_PAD(cbits, b) & 0xffffu;
    for (uint32_t c = tid; c < ncb; c += 1024) s_cnt[c] = 0;
    __syncthreads();
    for (int w = part; w < B.nwriters; w += nparts) {
        const uint32_t n = B.gcount[(size_t)b * B.nwriters + w];       // a multiple of 32, pads included
        const uint32_t *base = B.rec + GBN_REC_HI(GBN_RECIDX(B, b, w, 0));
        for (uint32_t j = (uint32_t)tid * 4u; j < n; j += 4096u) {
            const uint4 h = *reinterpret_cast<const uint4 *>(base + (j >> 6) * 96u + (j & 63u));
            if (!is_pad(h.x, padlow)) atomicAdd(&s_cnt[cell_in_bin(h.x, cbits)], 1u);
            if (!is_pad(h.y, padlow)) atomicAdd(&s_cnt[cell_in_bin(h.y, cbits)], 1u);
            if (!is_pad(h.z, padlow)) atomicAdd(&s_cnt[cell_in_bin(h.z, cbits)], 1u);
            if (!is_pad(h.w, padlow)) atomicAdd(&s_cnt[cell_in_bin(h.w, cbits)], 1u);
        }
    }
    __syncthreads();
    uint32_t *out = R.count + ((size_t)b << cbits);
    for (uint32_t c = tid; c < ncb; c += 1024) { const uint32_t v = s_cnt[c]; if (v) atomicAdd(&out[c], v); }
}

// the sub-bins' cursors: where their first cell's run begins
extern "C" __global__ void __launch_bounds__(256)
runs_cursor_kernel(GbnRunsBuild R)
{
    const uint32_t n = (uint32_t)R.B.nb << R.sbits, i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) R.cursor[i] = R.count[(size_t)i << (R.B.cbits - R.sbits)];
}

// ---------------------------------------------------------------------------------------------------
// Build, step 2: every bin's records by sub-bin (2^sbits groups of consecutive cells).  A workgroup per (bin, group of
// `wgroup` consecutive writers): up to GBN_RUNS_SPLIT_CAP records of the group's streams per round are ranked inside their
// sub-bins with LDS atomics, put into that order in LDS and leave as one contiguous piece per sub-bin, appended at the
// sub-bin's cursor (one global atomic per piece).  The position id is made here, where the record's place in its stream --
// and with it its tile (the stream cursors, as probe_rare_kernel resolves them for stream records) -- is still known.
extern "C" __global__ void __launch_bounds__(1024)
runs_split_kernel(GbnRunsBuild R)
{
    const GbnBinParams &B = R.B;
    constexpr uint32_t CAP = GBN_RUNS_SPLIT_CAP, NSB = 1u << GBN_RUNS_SBITS_MAX, PER = CAP / 1024 / 4;     // PER groups of 4 records per thread and round
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    uint32_t *s_key = s_mem, *s_pos = s_mem + CAP;
    uint32_t *s_cnt = s_mem + 2 * CAP, *s_lstart = s_cnt + NSB, *s_gdelta = s_lstart + NSB;       // per sub-bin: records of the round, their first LDS slot, global slot - LDS slot
    uint32_t *s_pref = s_gdelta + NSB;          // [wgroup + 1] records in front of every stream of the group (wgroup <= 1024)
    uint32_t *s_cur = s_pref + 1025;            // [wgroup][nseq] the streams' cursors (if they fit)
    __shared__ uint32_t s_wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ngroups = (B.nwriters + R.wgroup - 1) / R.wgroup;
    const int b = (int)(blockIdx.x / (unsigned)ngroups), w0 = (int)(blockIdx.x % (unsigned)ngroups) * R.wgroup;
    const int nw = min(R.wgroup, B.nwriters - w0);
    const int cbits = B.cbits, sbits = R.sbits, ushift = cbits - sbits;
    const uint32_t padlow = GBN_REC_PAD(cbits, b) & 0xffffu, nsb = 1u << sbits;
    const uint32_t nseq = B.nseq;
    const bool cur_lds = (size_t)nw * nseq <= GBN_RUNS_CURCAP;
    // records in front of every stream of the group
    {
        uint32_t v = tid < nw ? B.gcount[(size_t)b * B.nwriters + w0 + tid] : 0u;
        const uint32_t incl = wave_scan_incl(v);
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (int k = 0; k < wave; k++) before += s_wsum[k];
        if (tid < nw) s_pref[tid + 1] = before + incl;
        if (tid == 0) s_pref[0] = 0;
    }
    if (cur_lds)
        for (uint32_t i = tid; i < (uint32_t)nw * nseq; i += 1024) s_cur[i] = B.tcur[((size_t)b * B.nwriters + w0) * nseq + i];
    __syncthreads();
    const uint32_t total = s_pref[nw];
    const uint32_t *gcur = B.tcur + ((size_t)b * B.nwriters + w0) * nseq;
    const uint32_t full_rounds = (uint32_t)(B.S.ntiles / B.nwriters), rest = (uint32_t)(B.S.ntiles % B.nwriters);
    for (uint32_t r0 = 0; r0 < total; r0 += CAP) {
        if (tid < (int)NSB) s_cnt[tid] = 0;
        __syncthreads();
        uint32_t hi[PER][4], pid[PER][4], rank[PER][4];
        #pragma unroll
        for (uint32_t u = 0; u < PER; u++) {
            const uint32_t v = r0 + u * 4096u + (uint32_t)tid * 4u;     // virtual index of the group's first record (streams are multiples of 32 long: a group of four lies in one)
            #pragma unroll
            for (int i = 0; i < 4; i++) { hi[u][i] = padlow; pid[u][i] = 0; rank[u][i] = 0; }
            if (v >= total) continue;
            // the stream: the last g with s_pref[g] <= v
            int lo = 0, up = nw;
            while (up - lo > 1) { const int mid = (lo + up) >> 1; if (s_pref[mid] <= v) lo = mid; else up = mid; }
            const int g = lo, w = w0 + g;
            const uint32_t j0 = v - s_pref[g];
            const size_t L = GBN_RECIDX(B, b, w, j0);
            const uint4 h4 = *reinterpret_cast<const uint4 *>(B.rec + GBN_REC_HI(L));
            const uint2 x2 = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(B.rec) + GBN_REC_IDX16(L));
            hi[u][0] = h4.x; hi[u][1] = h4.y; hi[u][2] = h4.z; hi[u][3] = h4.w;
            const uint32_t ix[4] = {x2.x & 0xffffu, x2.x >> 16, x2.y & 0xffffu, x2.y >> 16};
            // tiles of this writer (probe_rare_kernel): one per full round, one of the last round if its rotated index falls into it
            const uint32_t ntiles_w = full_rounds + ((((uint32_t)w + full_rounds) % (uint32_t)B.nwriters) < rest ? 1u : 0u);
            const uint32_t nt = (ntiles_w + (1u << GBN_TCUR_SHIFT) - 1u) >> GBN_TCUR_SHIFT;
            const uint32_t *cur = cur_lds ? s_cur + (size_t)g * nseq : gcur + (size_t)g * nseq;
            uint32_t cl = 0, ch = nt;
            while (ch - cl > 1) { const uint32_t mid = (cl + ch) >> 1; if (cur[mid] <= j0) cl = mid; else ch = mid; }
            #pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t j = j0 + (uint32_t)i;
                while (cl + 1 < nt && cur[cl + 1] <= j) cl++;
                const uint32_t seqn = (cl << GBN_TCUR_SHIFT) | (ix[i] >> GBN_BIN_TILE_BITS);
                pid[u][i] = (GBN_TILE_OF((uint32_t)w, seqn, (uint32_t)B.nwriters) << GBN_BIN_TILE_BITS) | (ix[i] & (uint32_t)(GBN_BIN_TILE_POS - 1));
                if (!is_pad(hi[u][i], padlow)) rank[u][i] = atomicAdd(&s_cnt[cell_in_bin(hi[u][i], cbits) >> ushift], 1u);
            }
        }
        __syncthreads();
        // the sub-bins' first LDS slots; room for their pieces at the sub-bins' cursors
        {
            const uint32_t c = tid < (int)nsb ? s_cnt[tid] : 0u;
            const uint32_t incl = wave_scan_incl(c);        // (waves 0 .. 3 hold sub-bins; all waves run the scan)
            if (lane == 63) s_wsum[wave] = incl;
            __syncthreads();
            uint32_t before = 0;
            for (int k = 0; k < wave; k++) before += s_wsum[k];
            if (tid < (int)nsb) {
                const uint32_t ls = before + incl - c;
                s_lstart[tid] = ls;
                uint32_t gb = 0;
                if (c) gb = atomicAdd(&R.cursor[((size_t)b << sbits) + tid], c);
                s_gdelta[tid] = gb - ls;
            }
        }
        __syncthreads();
        uint32_t nround = s_lstart[nsb - 1] + s_cnt[nsb - 1];
        #pragma unroll
        for (uint32_t u = 0; u < PER; u++)
            #pragma unroll
            for (int i = 0; i < 4; i++)
                if (!is_pad(hi[u][i], padlow)) {
                    const uint32_t slot = s_lstart[cell_in_bin(hi[u][i], cbits) >> ushift] + rank[u][i];
                    s_key[slot] = hi[u][i]; s_pos[slot] = pid[u][i];
                }
        __syncthreads();
        for (uint32_t i = tid; i < nround; i += 1024) {
            const uint32_t k = s_key[i];
            const size_t dst = (size_t)(s_gdelta[cell_in_bin(k, cbits) >> ushift] + i);     // (modulo 2^32: gdelta = global - local)
            R.mid_key[dst] = k; R.mid_pos[dst] = s_pos[i];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// Build, step 3: a sub-bin at a time into its cells' runs.  A workgroup per sub-bin: the cells' cursors in LDS, the
// sub-bin's records past them into an LDS image of the runs, the image out in one piece.  A sub-bin with more records
// than the image holds (repeat-rich subjects) goes in rounds over ranges of its cells; a single cell beyond that is copied
// through a workgroup-wide compaction.
extern "C" __global__ void __launch_bounds__(1024)
runs_place_kernel(GbnRunsBuild R)
{
    const GbnBinParams &B = R.B;
    constexpr uint32_t CAP = GBN_RUNS_PLACE_CAP;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    uint32_t *s_pos = s_mem;                                            // [CAP]
    uint16_t *s_fp = reinterpret_cast<uint16_t *>(s_mem + CAP);         // [CAP]
    uint32_t *s_cur = s_mem + CAP + CAP / 2;                            // [cells of the sub-bin]
    __shared__ uint32_t s_big;
    const int tid = threadIdx.x, lane = tid & 63;
    const int cbits = B.cbits, ushift = cbits - R.sbits;
    const uint32_t ncu = 1u << ushift, umask = ncu - 1u;
    const size_t cfirst = (size_t)blockIdx.x << ushift;
    const uint32_t *rs = R.count + cfirst;                              // run_start of the sub-bin's cells (and of the next cell)
    const uint32_t a = rs[0], n = rs[ncu] - a;
    if (n == 0) return;
    for (uint32_t c0 = 0; c0 < ncu; ) {
        // the cells of this round: as many as the image holds
        const uint32_t lo = rs[c0] - a;
        uint32_t c1;
        if (rs[ncu] - a - lo <= CAP) c1 = ncu;
        else {
            uint32_t l = c0, h = ncu;                                   // the last c1 with rs[c1] - a - lo <= CAP
            while (h - l > 1) { const uint32_t mid = (l + h) >> 1; if (rs[mid] - a - lo <= CAP) l = mid; else h = mid; }
            c1 = l;
        }
        if (c1 == c0) {
            // one cell with more records than the image holds
            const uint32_t hi_end = rs[c0 + 1] - a;
            if (tid == 0) s_big = 0;
            __syncthreads();
            for (uint32_t i0 = 0; i0 < n; i0 += 1024) {
                const uint32_t i = i0 + (uint32_t)tid;
                uint32_t k = 0; bool mine = false;
                if (i < n) { k = R.mid_key[(size_t)a + i]; mine = (cell_in_bin(k, cbits) & umask) == c0; }
                const unsigned long long m = __ballot(mine);
                uint32_t base = 0;
                if (m) {
                    if (lane == 0) base = atomicAdd(&s_big, (uint32_t)__popcll(m));
                    base = __shfl(base, 0);
                }
                if (mine) {
                    const size_t at = (size_t)a + lo + base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
                    R.fp[at] = (uint16_t)(k >> 16); R.pos[at] = R.mid_pos[(size_t)a + i];
                }
            }
            __syncthreads();
            (void)hi_end;
            c0 += 1;
            continue;
        }
        const uint32_t hi_end = rs[c1] - a;
        for (uint32_t c = c0 + (uint32_t)tid; c < c1; c += 1024) s_cur[c] = rs[c] - a - lo;
        __syncthreads();
        const bool all = (c0 == 0 && c1 == ncu);
        for (uint32_t i = (uint32_t)tid; i < n; i += 1024) {
            const uint32_t k = R.mid_key[(size_t)a + i];
            const uint32_t c = cell_in_bin(k, cbits) & umask;
            if (all || (c >= c0 && c < c1)) {
                const uint32_t slot = atomicAdd(&s_cur[c], 1u);
                s_fp[slot] = (uint16_t)(k >> 16); s_pos[slot] = R.mid_pos[(size_t)a + i];
            }
        }
        __syncthreads();
        const uint32_t cnt = hi_end - lo;
        const size_t out0 = (size_t)a + lo;
        for (uint32_t i = (uint32_t)tid; i < cnt; i += 1024) { R.fp[out0 + i] = s_fp[i]; R.pos[out0 + i] = s_pos[i]; }
        __syncthreads();
        c0 = c1;
    }
}

// ---------------------------------------------------------------------------------------------------
// The probe kernel over sorted records.  No table slice in LDS, no bins: a wave draws GBN_RUNS_ITEM_CELLS consecutive cells
// from a counter, reads their table words and run boundaries in cell order (coalesced: 8 bytes per cell of the table, 134 MB
// for 16.7 M cells, whatever the batch), and for the cells the batch occupies -- and only for those -- their runs, sixteen
// bytes (eight records) per lane and round: the lanes of a round are dealt the 16-byte chunks of the occupied cells' runs one
// after the other (prefix sums of the chunks per cell; a lane finds its chunk's cell by a binary search over the wave's 64
// prefix sums through cross-lane reads).  The test of a record is probe_bin_kernel's (both reduced fingerprints of the
// cell's word against the record's fifteen bits, a zero-byte test), two records per 32-bit operation; what passes goes
// through a per-wave LDS queue to the same flush as there (side lists of cells with three and more entries, the cell's
// direct-probe word, the sixteenth bit) and on to the rare queue -- with its position id, which sorted records carry.
// Lookup hits are run lengths x entries per cell: no record is read for them.
#define GBN_RUNS_THREADS 512
#define GBN_RUNS_QCAP 128
#define GBN_RUNS_SIDE_W 256         // side-list fingerprints a wave stages per stretch of 64 cells
#define GBN_RUNS_SIDE_MAXN 16       // ... of cells with at most this many entries
extern "C" __global__ void __launch_bounds__(GBN_RUNS_THREADS)
probe_runs_kernel(GbnBinParams B)
{
    const GbnScanParams &P = B.S;
    constexpr int NW = GBN_RUNS_THREADS / 64;
    __shared__ uint32_t s_qrec[NW][GBN_RUNS_QCAP], s_qcell[NW][GBN_RUNS_QCAP], s_qfp[NW][GBN_RUNS_QCAP];
    __shared__ uint32_t s_side[NW][GBN_RUNS_SIDE_W];               // the side lists of the stretch a wave works on
    __shared__ uint32_t s_rcount;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cbits = B.cbits;
    GbnRareItem *myq = B.rareq + (size_t)blockIdx.x * B.rare_seg;
    if (tid == 0) s_rcount = 0;
    __syncthreads();
    const uint32_t lmask = (B.rfl <= 0) ? 0u : ((1u << (2 * B.rfl)) - 1);
    const uint32_t rmask = (B.rfrbits <= 0) ? 0u : (((1u << B.rfrbits) - 1) << (7 - B.rfrbits));
    const uint32_t m4 = (lmask | (rmask << 8)) * 0x10001u;
    const bool fp16 = B.rfl >= 4 && B.rfrbits >= 7 && P.fr >= 4 && !(B.dbg & 256);
    uint32_t *qrec = s_qrec[wave], *qcell = s_qcell[wave], *qfp = s_qfp[wave];
    int qn = 0;                                                         // wave-uniform
    unsigned long long raw = 0;
    const unsigned long long lt = (1ull << lane) - 1;

    // Queued records, 64 at a time (one per lane).  What the main loop has tested completely (cells of one and two entries, cells
    // whose side list it had staged) only needs its position id and the cell's direct-probe word for the rare kernel; records
    // of the other cells (three and more entries beyond the staging area, cells that are always rare) get their side list
    // from global memory here.
    auto flush = [&](int first, int cnt) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        bool keep = false; uint32_t rec = 0, cv = 0;
        if (lane < cnt) {
            rec = qrec[first + lane]; cv = qcell[first + lane];
            const uint32_t f = qfp[first + lane];
            keep = true;
            if ((f >> 16) == 0) {                                       // kind 0: nothing tested yet
                const uint32_t t = B.cellt[cv];
                const uint32_t n3 = (t >> 16) & 0x7fffu, so = t & 0x7fffu, sf = f & 0x7fffu;
                if (n3 == 0) cv |= 0x80000000u;                         // its lookup hits are counted by the rare kernel
                else {
                    keep = false;
                    const uint16_t *side = B.sidet + B.side_start[cv >> cbits] + so;
                    for (uint32_t e = 0; e < n3; e++) {
                        const uint32_t x = (uint32_t)side[e] ^ sf;
                        keep = keep || ((x & lmask) == 0) || (((x >> 8) & rmask) == 0);
                    }
                }
            }
        }
        const unsigned long long m = __ballot(keep);
        if (m) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_rcount, (uint32_t)__popcll(m));
            base = __shfl(base, 0);
            if (keep) {
                const uint32_t at = base + (uint32_t)__popcll(m & lt);
                if (at < B.rare_seg) {
                    uint4 it; it.x = B.run_pos[rec]; it.y = cv; it.z = 0; it.w = P.cellw[cv & 0x7fffffffu];
                    *reinterpret_cast<uint4 *>(myq + at) = it;
                }
            }
        }
    };

    // (small tables -- 4^8 cells, runs of tens of thousands of records -- are dealt out in stretches of 64 cells: with 256 a one-query
    // batch kept 256 of the 6,144 waves busy)
    const uint32_t item_cells = B.run_item_cells > 0 ? (uint32_t)B.run_item_cells : (uint32_t)GBN_RUNS_ITEM_CELLS;
    const uint32_t nitems = (uint32_t)(P.ncells / item_cells);
    const uint4 *fp4 = reinterpret_cast<const uint4 *>(B.run_fp);
    const uint32_t m16 = m4 | (fp16 ? 0x80008000u : 0u);               // cells of one entry: the sixteenth bit joins the test (probe_bin_kernel tests it per queued record)
    uint32_t *side_w = s_side[wave];
    uint32_t item;
    { uint32_t v = 0; if (lane == 0) v = atomicAdd(&B.work[0], 1u); item = (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
    while (item < nitems) {
        uint32_t nitem = 0;
        if (lane == 0) nitem = atomicAdd(&B.work[0], 1u);               // (asked for now, read when this item is done)
        const uint32_t cbase = item * item_cells;
        const uint16_t *side_bin = B.sidet + B.side_start[cbase >> cbits];      // (an item lies in one bin)
        // table words and run boundaries of the first stretch
        uint32_t t_n = B.cellt[cbase + lane], rs_n = B.run_start[cbase + lane], re_n = B.run_start[cbase + lane + 1];
        uint32_t cw_n = fp16 ? P.cellw[cbase + lane] : 0u;
        for (uint32_t cs = cbase; cs < cbase + item_cells; cs += 64) {
            const uint32_t t = t_n, rs = rs_n, re = re_n, cw = cw_n;
            if (cs + 64 < cbase + item_cells) {      // the next stretch's, a stretch ahead
                t_n = B.cellt[cs + 64 + lane]; rs_n = B.run_start[cs + 64 + lane]; re_n = B.run_start[cs + 65 + lane];
                cw_n = fp16 ? P.cellw[cs + 64 + lane] : 0u;
            }
            const bool occ = t != 0 && re > rs;
            // What a record of the cell is tested against (fw) and how (kind): 1 = two reduced fingerprints of 15 bits, 2 = the one
            // entry's sixteen bits (from the direct-probe word), 3 = the n3 fingerprints of a cell of three and more entries, staged
            // in the wave's LDS below (fw = first slot | n3 << 16), 0 = not here: every record takes the queue.
            uint32_t kind = 0, fw = 0, nstage = 0, so = 0;
            if (occ) {
                const uint32_t len = re - rs;
                // lookup hits: run length x entries of the cell (a cell whose side list is not there -- n3 = 0 -- is counted by
                // the rare kernel, which sees every record of it)
                if (t & 0x8000u) {
                    raw += (unsigned long long)len * (1u + (t >> 31));
                    kind = 1; fw = t & 0x7fff7fffu;
                    if (fp16 && !(t >> 31) && !(cw >> 31) && !(cw & 1u)) {
                        const uint32_t el8 = (cw >> 15) & 0xffu, er8 = (cw >> 7) & 0xffu;
                        kind = 2; fw = (el8 | ((er8 >> 1) << 8) | ((er8 & 1u) << 15)) * 0x10001u;
                    }
                } else {
                    const uint32_t n3 = (t >> 16) & 0x7fffu;
                    raw += (unsigned long long)len * n3;
                    if (n3 >= 1 && n3 <= (uint32_t)GBN_RUNS_SIDE_MAXN) { nstage = n3; so = t & 0x7fffu; }
                }
            }
            // the side lists of this stretch's cells into the wave's LDS (a cell of three and more entries has 3.3 on average,
            // a stretch 1.5 such cells)
            if (__ballot(nstage != 0)) {
                const uint32_t sidx = wave_scan_incl(nstage) - nstage;
                if (nstage && sidx + nstage <= (uint32_t)GBN_RUNS_SIDE_W) {
                    kind = 3; fw = sidx | (nstage << 16);
                    for (uint32_t e0 = 0; e0 < nstage; e0 += 4) {
                        uint32_t v[4];
                        #pragma unroll
                        for (uint32_t i = 0; i < 4; i++) v[i] = side_bin[so + min(e0 + i, nstage - 1u)];
                        #pragma unroll
                        for (uint32_t i = 0; i < 4; i++) if (e0 + i < nstage) side_w[sidx + e0 + i] = v[i];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            const uint32_t k0 = rs >> 3, nch = occ ? ((re + 7u) >> 3) - k0 : 0u;
            const uint32_t incl = wave_scan_incl(nch), exk = (incl - nch) | (kind << 29);      // (fewer than 2^29 chunks in all: 2^32 records)
            const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            // round g0: lane l takes chunk g0 + l of the stretch's occupied runs
            auto fetch = [&](uint32_t g0, uint32_t &o_fw, uint32_t &o_rs, uint32_t &o_re, uint32_t &o_cell, uint32_t &o_k, uint32_t &o_kind, uint4 &d) {
                const uint32_t g = g0 + (uint32_t)lane;
                // owner: the first lane i with incl[i] > g
                uint32_t l = 0;
                #pragma unroll
                for (int sf = 32; sf > 0; sf >>= 1) { const uint32_t v = (uint32_t)__shfl((int)incl, (int)(l + sf - 1)); if (v <= g) l += (uint32_t)sf; }
                l = min(l, 63u);
                o_fw = (uint32_t)__shfl((int)fw, (int)l); o_rs = (uint32_t)__shfl((int)rs, (int)l); o_re = (uint32_t)__shfl((int)re, (int)l);
                const uint32_t ex = (uint32_t)__shfl((int)exk, (int)l);
                o_kind = g < T ? (ex >> 29) | 4u : 0u;                  // bit 2: the lane has a chunk
                o_cell = cs + l;
                o_k = (o_rs >> 3) + (g - (ex & 0x1fffffffu));
                d = make_uint4(0, 0, 0, 0);
                if (g < T) d = fp4[o_k];
            };
            uint32_t cfw = 0, crs = 0, cre = 0, ccell = 0, ck = 0, ckind = 0; uint4 cd = make_uint4(0, 0, 0, 0);
            if (T) fetch(0, cfw, crs, cre, ccell, ck, ckind, cd);
            for (uint32_t g0 = 0; g0 < T; g0 += 64) {
                uint32_t nfw = 0, nrs = 0, nre = 0, ncell = 0, nk = 0, nkind = 0; uint4 nd = make_uint4(0, 0, 0, 0);
                if (g0 + 64 < T) fetch(g0 + 64, nfw, nrs, nre, ncell, nk, nkind, nd);      // the next round's chunks in flight
                // the lane's records 8 ck .. 8 ck + 7 that belong to the run
                uint32_t slowm = 0;
                const uint32_t w[4] = {cd.x, cd.y, cd.z, cd.w};
                uint32_t z[4] = {0, 0, 0, 0};
                // a masked byte of (w ^ fp:fp) is zero iff that side of that record matches the fingerprint: two records per operation
                const uint32_t k3 = ckind & 3u;
                if (k3 == 1u || k3 == 2u) {
                    const uint32_t mm = (k3 == 2u) ? m16 : m4;
                    const uint32_t fa = (cfw & 0xffffu) * 0x10001u, fb = (cfw >> 16) * 0x10001u;
                    #pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t xa = (w[j] ^ fa) & mm, xb = (w[j] ^ fb) & mm;
                        z[j] = ((xa - 0x01010101u) & ~xa) | ((xb - 0x01010101u) & ~xb);
                    }
                }
                {   // cells of three and more entries: their fingerprints one after the other (as many rounds as the longest list in the wave)
                    const uint32_t n3 = (ckind == 7u) ? cfw >> 16 : 0u, sidx = cfw & 0xffffu;
                    for (uint32_t e = 0; __ballot(e < n3); e++) {
                        if (e < n3) {
                            const uint32_t fa = side_w[sidx + e] * 0x10001u;
                            #pragma unroll
                            for (int j = 0; j < 4; j++) { const uint32_t xa = (w[j] ^ fa) & m4; z[j] |= (xa - 0x01010101u) & ~xa; }
                        }
                    }
                }
                if (ckind & 4u) {
                    const uint32_t r0 = ck << 3;
                    const uint32_t lo_i = crs > r0 ? crs - r0 : 0u, hi_i = min(cre - r0, 8u);
                    const uint32_t vm = ((1u << hi_i) - 1u) & ~((1u << lo_i) - 1u);
                    if (k3 == 0u) slowm = vm;                           // nothing tested: every record takes the queue
                    else {
                        #pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t y = (z[j] | (z[j] >> 8)) & 0x00800080u;      // bit 7: record 2 j, bit 23: record 2 j + 1
                            slowm |= (((y >> 7) | (y >> 22)) & 3u) << (2 * j);
                        }
                        slowm &= vm;
                    }
                }
                while (true) {
                    const unsigned long long m = __ballot(slowm != 0);
                    if (!m) break;
                    if (slowm) {
                        const uint32_t r = (uint32_t)__ffs(slowm) - 1u;
                        slowm &= slowm - 1;
                        const uint32_t m1 = 0u - ((r >> 1) & 1u), m2 = 0u - ((r >> 2) & 1u);
                        const uint32_t a0 = (cd.y & m1) | (cd.x & ~m1), a1 = (cd.w & m1) | (cd.z & ~m1);
                        const uint32_t wsel = (a1 & m2) | (a0 & ~m2);
                        const int at = qn + __popcll(m & lt);
                        qrec[at] = (ck << 3) + r; qcell[at] = ccell; qfp[at] = ((r & 1u) ? (wsel >> 16) : (wsel & 0xffffu)) | (k3 << 16);
                    }
                    qn += __popcll(m);
                    if (qn >= 64) { qn -= 64; flush(qn, 64); }
                }
                cfw = nfw; crs = nrs; cre = nre; ccell = ncell; ck = nk; ckind = nkind; cd = nd;
            }
        }
        item = (uint32_t)__builtin_amdgcn_readfirstlane((int)nitem);
    }
    if (qn > 0) { flush(0, qn); qn = 0; }
    if (P.raw_hits) {
        for (int off = 32; off > 0; off >>= 1) raw += __shfl_down(raw, off);
        if (lane == 0 && raw) atomicAdd(P.raw_hits, raw);
    }
    __syncthreads();
    if (tid == 0) B.rare_counts[blockIdx.x] = s_rcount;
}

namespace gbn {
hipError_t launch_probe_rare(const GbnBinParams &b, int nseg, hipStream_t st);      // scan_bin.hip

// bytes of LDS the build kernels ask for
static size_t split_lds() { return ((size_t)2 * GBN_RUNS_SPLIT_CAP + 3 * ((size_t)1 << GBN_RUNS_SBITS_MAX) + 1025 + GBN_RUNS_CURCAP) * 4; }
static size_t place_lds(int ushift) { return ((size_t)GBN_RUNS_PLACE_CAP + GBN_RUNS_PLACE_CAP / 2 + ((size_t)1 << ushift)) * 4; }

// sub-bins per bin (as a power of two) for `npos` records in nb bins of 2^cbits cells: a sub-bin's records are to fit the
// place kernel's LDS image with room to spare, its cells' cursors next to it
int runs_choose_sbits(int64_t npos, int nb, int cbits)
{
    const double per_bin = (double)npos / (double)std::max(1, nb);
    int sbits = 0;
    while (sbits < GBN_RUNS_SBITS_MAX && sbits < cbits && per_bin / (double)(1 << sbits) > 0.92 * GBN_RUNS_PLACE_CAP) sbits++;
    while (sbits < cbits && (1 << (cbits - sbits)) > GBN_RUNS_UNIT_CELLS_MAX) sbits++;
    return std::min(sbits, std::min(cbits, GBN_RUNS_SBITS_MAX));
}
// consecutive writers per split workgroup: about one round of records
int runs_choose_wgroup(int64_t npos, int nb, int nwriters)
{
    const double per_stream = (double)npos / ((double)std::max(1, nb) * (double)std::max(1, nwriters)) + 48.0;
    int g = (int)((double)GBN_RUNS_SPLIT_CAP * 0.85 / per_stream);
    return std::max(1, std::min(std::min(g, nwriters), 1024));
}

// the build queued on `st`: R.count zero on entry (ncells + 1 words), lut_scan's scratch in scan_tmp
hipError_t launch_runs_build(const GbnRunsBuild &R, void *scan_tmp, size_t scan_tmp_bytes, hipStream_t st)
{
    const GbnBinParams &B = R.B;
    hipError_t e;
    const int ushift = B.cbits - R.sbits;
    if (R.sbits < 0 || R.sbits > GBN_RUNS_SBITS_MAX || ushift < 0 || (1 << ushift) > GBN_RUNS_UNIT_CELLS_MAX || R.wgroup < 1 || R.wgroup > 1024) return hipErrorInvalidValue;
    {
        const int nparts = std::max(1, 1024 / std::max(1, B.nb));
        const size_t lds = (size_t)4 << B.cbits;
        static std::atomic<uint64_t> done{0};
        if ((e = raise_dynamic_lds((const void *)runs_count_kernel, 128 * 1024, done)) != hipSuccess) return e;
        hipLaunchKernelGGL(runs_count_kernel, dim3((unsigned)(B.nb * nparts)), dim3(1024), lds, st, R, nparts);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if ((e = lut_scan(scan_tmp, scan_tmp_bytes, R.count, R.count, B.S.ncells + 1, st)) != hipSuccess) return e;
    {
        const unsigned n = (unsigned)B.nb << R.sbits;
        hipLaunchKernelGGL(runs_cursor_kernel, dim3((n + 255) / 256), dim3(256), 0, st, R);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    {
        const int ngroups = (B.nwriters + R.wgroup - 1) / R.wgroup;
        static std::atomic<uint64_t> done{0};
        if ((e = raise_dynamic_lds((const void *)runs_split_kernel, split_lds(), done)) != hipSuccess) return e;
        hipLaunchKernelGGL(runs_split_kernel, dim3((unsigned)(B.nb * ngroups)), dim3(1024), split_lds(), st, R);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    {
        static std::atomic<uint64_t> done{0};
        if ((e = raise_dynamic_lds((const void *)runs_place_kernel, place_lds(11), done)) != hipSuccess) return e;
        hipLaunchKernelGGL(runs_place_kernel, dim3((unsigned)B.nb << R.sbits), dim3(1024), place_lds(ushift), st, R);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
}

// a pass over sorted records: probe_runs_kernel + the rare kernel; ev as launch_scan_bin_parts (ev[0], ev[1] before the probe
// kernel, ev[2] behind it, ev[3] behind the rare kernel)
hipError_t launch_probe_runs(const GbnBinParams &b, int grid, hipStream_t st, hipEvent_t *ev, hipEvent_t tables_ready)
{
    if (b.S.ntiles <= 0) return hipSuccess;
    hipError_t e;
    if (ev) { (void)hipEventRecord(ev[0], st); (void)hipEventRecord(ev[1], st); }
    if (tables_ready && (e = hipStreamWaitEvent(st, tables_ready, 0)) != hipSuccess) return e;
    hipLaunchKernelGGL(probe_runs_kernel, dim3((unsigned)grid), dim3(GBN_RUNS_THREADS), 0, st, b);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (ev) (void)hipEventRecord(ev[2], st);
    e = launch_probe_rare(b, grid, st);
    if (ev) (void)hipEventRecord(ev[3], st);
    return e;
}
}  // namespace gbn
