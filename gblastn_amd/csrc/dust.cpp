// dust.cpp -- symmetric DUST low-complexity intervals of a query, the filter blastn applies by
// default (-dust "20 64 1", soft masking); feed the result to gbn_batch_new_masked.
//
// Same algorithm and tie rules as the reference's CSymDustMasker
// (c++/src/algo/dustmask/symdust.cpp:40-287; Morgulis et al., J Comput Biol 13:1028, 2006), so the
// intervals are identical: a sliding window of at most `window` bases is kept as a queue of
// 3-mers ("triplets") with their multiplicities; whenever the window's score may exceed the
// threshold, its suffixes are scanned for "perfect intervals" (no sub-interval scores higher per
// length), which are emitted -- joined when at most `linker` apart -- as the window moves past them.
// Finally sorted, overlapping and abutting intervals are fused (api/dust_filter.cpp:119-127).
// Host only.
#include "gbn_host.hpp"
#include "gbn_guard.hpp"
#include <algorithm>
#include <cstring>

namespace {

struct Interval { uint32_t lo, hi, score, len; };

class TripletWindow {
public:
    TripletWindow(uint32_t window, uint32_t level, std::vector<Interval> &perfect, const std::vector<uint32_t> &thr)
        : cap_(window - 2), low_k_(level / 5), perfect_(perfect), thr_(thr) {
        std::memset(in_window_, 0, sizeof(in_window_)); std::memset(in_suffix_, 0, sizeof(in_suffix_));
    }
    uint32_t first() const { return first_; }

    // add 3-mer t at the right end; false = the window holds one distinct 3-mer only (a homopolymer-like
    // stretch), which is recorded as a perfect interval without any scan
    bool advance(uint8_t t) {
        if (size_ >= cap_) {
            if (distinct_ <= 1) return slide_uniform(t);
            const uint8_t old = oldest();
            drop_oldest();
            leave(sum_window_, in_window_, old);
            if (in_window_[old] == 0) --distinct_;
            if (suffix_first_ == first_) { ++suffix_first_; leave(sum_suffix_, in_suffix_, old); }
            ++first_;
        }
        push(t);
        if (in_window_[t] == 0) ++distinct_;
        enter(sum_window_, in_window_, t);
        enter(sum_suffix_, in_suffix_, t);
        if (in_suffix_[t] > low_k_) {
            // shorten the suffix from its left until t is no longer over-represented in it
            uint32_t age = size_ - (suffix_first_ - first_) - 1;       // index counted from the newest 3-mer
            uint8_t u;
            do { u = at_age(age); leave(sum_suffix_, in_suffix_, u); ++suffix_first_; --age; } while (u != t);
        }
        ++last_;
        if (size_ >= cap_ && distinct_ <= 1) {
            perfect_.clear();
            perfect_.insert(perfect_.begin(), Interval{first_, last_ + 1, 0, 0});
            return false;
        }
        return true;
    }

    bool worth_scanning() const {
        const uint32_t n = last_ - suffix_first_;
        return n < size_ && 10 * sum_window_ > thr_[n];
    }

    // extend the current suffix leftwards 3-mer by 3-mer; every extension that scores above the
    // threshold and at least as high (per length) as the perfect intervals inside it becomes one
    void scan_suffixes() {
        uint8_t mult[64]; std::memcpy(mult, in_suffix_, 64);
        uint32_t n = last_ - suffix_first_, score = sum_suffix_, best_score = 0, best_len = 0;
        uint32_t pos = suffix_first_ - 1;                   // wraps for 0 exactly like the reference's unsigned
        size_t pi = 0;
        for (uint32_t age = n; age < size_; ++age, ++n, --pos) {
            const uint8_t t = at_age(age), seen = mult[t];
            enter(score, mult, t);
            if (seen == 0 || score * 10 <= thr_[n]) continue;
            for (; pi != perfect_.size() && pos <= perfect_[pi].lo; ++pi)
                if (best_score == 0 || best_len * perfect_[pi].score > best_score * perfect_[pi].len) {
                    best_score = perfect_[pi].score; best_len = perfect_[pi].len;
                }
            if (best_score == 0 || score * best_len >= best_score * n) {
                best_score = score; best_len = n;
                perfect_.insert(perfect_.begin() + (std::ptrdiff_t)pi, Interval{pos, last_ + 1, score, n});
            }
        }
    }

private:
    bool slide_uniform(uint8_t t) {
        const uint8_t old = oldest();
        drop_oldest();
        leave(sum_window_, in_window_, old);
        if (in_window_[old] == 0) --distinct_;
        ++first_;
        push(t);
        if (in_window_[t] == 0) ++distinct_;
        enter(sum_window_, in_window_, t);
        ++last_;
        if (distinct_ <= 1) { perfect_.insert(perfect_.begin(), Interval{first_, last_ + 1, 0, 0}); return false; }
        return true;
    }
    static void enter(uint32_t &sum, uint8_t *mult, uint8_t t) { sum += mult[t]; ++mult[t]; }
    static void leave(uint32_t &sum, uint8_t *mult, uint8_t t) { --mult[t]; sum -= mult[t]; }
    // ring buffer of the 3-mers in the window; age 0 = newest
    void push(uint8_t t) { head_ = (head_ + 127) & 127; ring_[head_] = t; ++size_; }
    void drop_oldest() { --size_; }
    uint8_t oldest() const { return ring_[(head_ + size_ - 1) & 127]; }
    uint8_t at_age(uint32_t age) const { return ring_[(head_ + age) & 127]; }

    uint8_t ring_[128]; uint32_t head_ = 0, size_ = 0;
    uint32_t first_ = 0, last_ = 0, cap_, low_k_, suffix_first_ = 0;
    uint8_t in_window_[64], in_suffix_[64];
    uint32_t sum_window_ = 0, sum_suffix_ = 0, distinct_ = 0;
    std::vector<Interval> &perfect_;
    const std::vector<uint32_t> &thr_;
};

}  // namespace

extern "C" int32_t gbn_dust_mask(const uint8_t *seq, int32_t len, int32_t level, int32_t window, int32_t linker,
                                 int32_t *from, int32_t *to, int32_t cap)
{
    return gbn::guard_as<int32_t>(__func__, (int32_t)-1, (int32_t)-1, [&]() -> int32_t {
    if (!seq || len <= 0 || (cap > 0 && (!from || !to))) return 0;
    if (level < 2 || level > 64) level = 20;
    if (window < 8 || window > 64) window = 64;
    if (linker < 1 || linker > 32) linker = 1;
    std::vector<uint32_t> thr((size_t)window - 2);
    thr[0] = 1;
    for (size_t i = 1; i < thr.size(); i++) thr[i] = (uint32_t)i * (uint32_t)level;
    auto base = [&](uint32_t p) -> uint8_t { return seq[p] <= 3 ? seq[p] : 0; };      // non-ACGT counts as A

    std::vector<std::pair<uint32_t, uint32_t>> found;
    std::vector<Interval> perfect;
    // emit the perfect intervals the window has moved past (they start left of `wfirst`)
    auto emit = [&](uint32_t wfirst, uint32_t origin) {
        if (perfect.empty() || perfect.back().lo >= wfirst) return;
        const uint32_t lo = perfect.back().lo + origin, hi = perfect.back().hi + origin;
        if (!found.empty() && found.back().second + (uint32_t)linker >= lo) found.back().second = std::max(found.back().second, hi);
        else found.emplace_back(lo, hi);
        while (!perfect.empty() && perfect.back().lo < wfirst) perfect.pop_back();
    };

    uint32_t origin = 0; const uint32_t stop = (uint32_t)len - 1;
    while (stop > 2 + origin) {
        perfect.clear();
        TripletWindow w((uint32_t)window, (uint32_t)level, perfect, thr);
        uint8_t t = (uint8_t)((base(origin) << 2) + base(origin + 1));
        uint32_t pos = origin + 2;
        bool restart = false;
        while (!restart && pos <= stop) {
            emit(w.first(), origin);
            t = (uint8_t)(((t << 2) & 0x3f) + base(pos));
            ++pos;
            if (w.advance(t)) { if (w.worth_scanning()) w.scan_suffixes(); continue; }
            // uniform stretch: keep sliding until a second distinct 3-mer shows up, then start over there
            for (; pos <= stop; ++pos) {
                emit(w.first(), origin);
                t = (uint8_t)(((t << 2) & 0x3f) + base(pos));
                if (w.advance(t)) { restart = true; break; }
            }
        }
        for (uint32_t wfirst = w.first(); !perfect.empty(); ++wfirst) emit(wfirst, origin);
        if (w.first() == 0) break;
        origin += w.first();
    }
    // sorted already; fuse overlapping and abutting intervals
    std::vector<std::pair<uint32_t, uint32_t>> fused;
    for (auto &iv : found) {
        if (!fused.empty() && iv.first <= fused.back().second + 1) fused.back().second = std::max(fused.back().second, iv.second);
        else fused.push_back(iv);
    }
    const int32_t n = (int32_t)fused.size();
    for (int32_t i = 0; i < n && i < cap; i++) { from[i] = (int32_t)fused[(size_t)i].first; to[i] = (int32_t)fused[(size_t)i].second; }
    return n;
    });
}
