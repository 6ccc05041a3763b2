// hsp_host.hpp -- host-side gapped-stage replay and HSP list rules.
#pragma once
#include "gbn_host.hpp"
#include "gbn_dev.h"
#include <utility>
#include <vector>

namespace gbn {
void purge_common_endpoints(std::vector<GbnHSP> &v);
void sort_by_score(std::vector<GbnHSP> &v);
// hits: every initial hit of ONE subject with its precomputed gapped extension
// chunk: the subject is a chunk of a longer sequence -- its list stops after purge, odd-score rounding and sort;
// e-values, the e-value reap and the per-sequence counters follow the merge of the chunk lists (merge_chunk_lists)
void finish_subject(const GbnBatch &b, int32_t oid, int32_t slen,
                    std::vector<std::pair<GbnDevInitHit, GbnDevGapped>> &hits,
                    std::vector<GbnHSP> &out, GbnDiagnostics *diag, bool chunk = false);
}
