// stat.cpp -- Karlin-Altschul statistics for nucleotide scoring systems (host).
//
// These doubles are truncated into the integer cut-offs that gate every
// kernel (ungapped X-drop, gap trigger, gapped X-drop, e-value cut-off), so the
// arithmetic follows NCBI-BLAST 2.2.28 operation by operation:
//   CORE/blast_stat.c  (lambda :2465-2572, H :2581-2607, K :2221-2393,
//                       gapped tables :575-705, :3209-3343, :3806-3990,
//                       length adjustment :4994-5076, E<->S :3994-4125)
//   CORE/ncbi_math.c   (expm1 :38, gcd :410, nint :442, powi :449)
#include "gbn_host.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>

namespace gbn {
namespace {

constexpr int kScoreMin = -32768, kScoreMax = 32767;

int gcd_pos(int a, int b) {
    b = std::abs(b);
    if (b > a) std::swap(a, b);
    while (b != 0) { int c = a % b; a = b; b = c; }
    return a;
}
long round_half_away(double x) { x += (x >= 0. ? 0.5 : -0.5); return (long)x; }

double int_power(double x, int n) {
    if (n == 0) return 1.;
    if (x == 0.) return n < 0 ? HUGE_VAL : 0.;
    if (n < 0) { x = 1. / x; n = -n; }
    double y = 1.;
    while (n > 0) { if (n & 1) y *= x; n /= 2; x *= x; }
    return y;
}
double expm1_series(double x) {
    double ax = std::fabs(x);
    if (ax > .33) return std::exp(x) - 1.;
    if (ax < 1.e-16) return x;
    return x * (1. + x * (1./2. + x * (1./6. + x * (1./24. + x * (1./120. + x * (1./720. + x *
           (1./5040. + x * (1./40320. + x * (1./362880. + x * (1./3628800. + x *
           (1./39916800. + x * (1./479001600. + x/6227020800.))))))))))));
}

// distribution of single-letter scores
struct ScoreDist {
    int lo, hi;                 // allowed range
    int obs_lo = 0, obs_hi = 0; // observed
    double mean = 0;
    std::vector<double> mass;   // mass[s - lo]
    double &at(int s) { return mass[s - lo]; }
    double at(int s) const { return mass[s - lo]; }
};

bool range_ok(int lo, int hi) { return !(lo >= 0 || hi <= 0 || lo < kScoreMin || hi > kScoreMax); }

bool solve_lambda(const ScoreDist &d, double &lambda_out) {
    if (d.mean >= 0.) return false;
    const int low = d.obs_lo, high = d.obs_hi;
    if (!range_ok(low, high)) return false;
    int g = -low;
    for (int i = 1; i <= high - low && g > 1; ++i)
        if (d.at(i + low) != 0.0) g = gcd_pos(g, i);
    // safeguarded Newton on x = exp(-lambda)
    const double tolx = 1.e-5; const int itmax = 20, max_newton = 20 + 17;
    double x0 = std::exp(-0.5), x = (0 < x0 && x0 < 1) ? x0 : .5, a = 0, b = 1, f = 4;
    bool is_newton = false;
    for (int k = 0; k < itmax; k++) {
        double fold = f; bool was_newton = is_newton; is_newton = false;
        double gr = 0; f = d.at(low);
        for (int i = low + g; i < 0; i += g) { gr = x * gr + f; f = f * x + d.at(i); }
        gr = x * gr + f; f = f * x + d.at(0) - 1;
        for (int i = g; i <= high; i += g) { gr = x * gr + f; f = f * x + d.at(i); }
        if (f > 0) a = x; else if (f < 0) b = x; else break;
        if (b - a < 2 * a * (1 - b) * tolx) { x = (a + b) / 2; break; }
        if (k >= max_newton || (was_newton && std::fabs(f) > .9 * std::fabs(fold)) || gr >= 0) {
            x = (a + b) / 2;
        } else {
            double p = -f / gr, y = x + p;
            if (y <= a || y >= b) x = (a + b) / 2;
            else { is_newton = true; x = y; if (std::fabs(p) < tolx * x * (1 - x)) break; }
        }
    }
    lambda_out = -std::log(x) / g;
    return true;
}

double entropy_H(const ScoreDist &d, double lambda) {
    if (lambda < 0. || !range_ok(d.obs_lo, d.obs_hi)) return -1.;
    double e = std::exp(-lambda);
    double sum = d.obs_lo * d.at(d.obs_lo);
    for (int s = d.obs_lo + 1; s <= d.obs_hi; s++) sum = s * d.at(s) + e * sum;
    double scale = int_power(e, d.obs_hi);
    if (scale > 0.0) return lambda * sum / scale;
    return lambda * std::exp(lambda * d.obs_hi + std::log(sum));
}

double solve_K(const ScoreDist &d, double lambda, double H) {
    if (lambda <= 0. || H <= 0.) return -1.;
    if (d.mean >= 0.0) return -1.;
    int low = d.obs_lo, high = d.obs_hi, range = high - low;
    const double *p_low = &d.mass[low - d.lo];
    int divisor = -low;
    for (int i = 1; i <= range && divisor > 1; ++i)
        if (p_low[i] != 0.0) divisor = gcd_pos(divisor, i);
    high /= divisor; low /= divisor; lambda *= divisor;
    range = high - low;
    double first_term = H / lambda;
    double e_ml = std::exp(-lambda);
    if (low == -1 && high == 1) {
        double pl = d.at(low * divisor), ph = d.at(high * divisor);
        return (pl - ph) * (pl - ph) / pl;
    }
    if (low == -1 || high == 1) {
        if (high != 1) {
            double avg = d.mean / divisor;
            first_term = (avg * avg) / first_term;
        }
        return first_term * (1.0 - e_ml);
    }
    const double sumlimit = 0.0001; const int iterlimit = 100;
    std::vector<double> P((size_t)iterlimit * range + 1, 0.0);
    double outer = 0., inner = 1.;
    int lo_as = 0, hi_as = 0;
    P[0] = 1.;
    for (int it = 0; it < iterlimit && inner > sumlimit; ) {
        int first = range, last = range;
        lo_as += low; hi_as += high;
        for (long p = hi_as - lo_as; p >= 0; --p) {
            long i1 = p - first, i1e = p - last; int j = first;
            double acc = 0.;
            for (; i1 >= i1e; --i1, ++j) acc += P[i1] * p_low[j];
            if (first) --first;
            if (p <= range) --last;
            P[p] = acc;
        }
        long idx = 0;
        inner = P[idx];
        int i = lo_as + 1;
        for (; i < 0; i++) inner = P[++idx] + inner * e_ml;
        inner *= e_ml;
        for (; i <= hi_as; ++i) inner += P[++idx];
        ++it;
        inner /= it;
        outer += inner;
    }
    return -std::exp(-2.0 * outer) / (first_term * expm1_series(-lambda));
}

// gapped parameter tables: {open, extend, lambda, K, H, alpha, beta, theta}
struct Row { double v[8]; };
struct Table { int reward, penalty; bool round_down; int open_max, ext_max; std::vector<Row> rows; };

const std::vector<Table> &tables() {
    static const std::vector<Table> t = {
        {1, -5, false, 3, 3, {{{0,0,1.39,0.747,1.38,1.00,0,100}}, {{3,3,1.39,0.747,1.38,1.00,0,100}}}},
        {1, -4, false, 2, 2, {{{0,0,1.383,0.738,1.36,1.02,0,100}}, {{1,2,1.36,0.67,1.2,1.1,0,98}},
            {{0,2,1.26,0.43,0.90,1.4,-1,91}}, {{2,1,1.35,0.61,1.1,1.2,-1,98}}, {{1,1,1.22,0.35,0.72,1.7,-3,88}}}},
        {2, -7, true, 4, 4, {{{0,0,0.69,0.73,1.34,0.515,0,100}}, {{2,4,0.68,0.67,1.2,0.55,0,99}},
            {{0,4,0.63,0.43,0.90,0.7,-1,91}}, {{4,2,0.675,0.62,1.1,0.6,-1,98}}, {{2,2,0.61,0.35,0.72,1.7,-3,88}}}},
        {1, -3, false, 2, 2, {{{0,0,1.374,0.711,1.31,1.05,0,100}}, {{2,2,1.37,0.70,1.2,1.1,0,99}},
            {{1,2,1.35,0.64,1.1,1.2,-1,98}}, {{0,2,1.25,0.42,0.83,1.5,-2,91}}, {{2,1,1.34,0.60,1.1,1.2,-1,97}},
            {{1,1,1.21,0.34,0.71,1.7,-2,88}}}},
        {2, -5, true, 4, 4, {{{0,0,0.675,0.65,1.1,0.6,-1,99}}, {{2,4,0.67,0.59,1.1,0.6,-1,98}},
            {{0,4,0.62,0.39,0.78,0.8,-2,91}}, {{4,2,0.67,0.61,1.0,0.65,-2,98}}, {{2,2,0.56,0.32,0.59,0.95,-4,82}}}},
        {1, -2, false, 2, 2, {{{0,0,1.28,0.46,0.85,1.5,-2,96}}, {{2,2,1.33,0.62,1.1,1.2,0,99}},
            {{1,2,1.30,0.52,0.93,1.4,-2,97}}, {{0,2,1.19,0.34,0.66,1.8,-3,89}}, {{3,1,1.32,0.57,1.0,1.3,-1,99}},
            {{2,1,1.29,0.49,0.92,1.4,-1,96}}, {{1,1,1.14,0.26,0.52,2.2,-5,85}}}},
        {2, -3, true, 6, 4, {{{0,0,0.55,0.21,0.46,1.2,-5,87}}, {{4,4,0.63,0.42,0.84,0.75,-2,99}},
            {{2,4,0.615,0.37,0.72,0.85,-3,97}}, {{0,4,0.55,0.21,0.46,1.2,-5,87}}, {{3,3,0.615,0.37,0.68,0.9,-3,97}},
            {{6,2,0.63,0.42,0.84,0.75,-2,99}}, {{5,2,0.625,0.41,0.78,0.8,-2,99}}, {{4,2,0.61,0.35,0.68,0.9,-3,96}},
            {{2,2,0.515,0.14,0.33,1.55,-9,81}}}},
        {3, -4, true, 6, 3, {{{6,3,0.389,0.25,0.56,0.7,-5,95}}, {{5,3,0.375,0.21,0.47,0.8,-6,92}},
            {{4,3,0.351,0.14,0.35,1.0,-9,86}}, {{6,2,0.362,0.16,0.45,0.8,-4,88}}, {{5,2,0.330,0.092,0.28,1.2,-13,81}},
            {{4,2,0.281,0.046,0.16,1.8,-23,69}}}},
        {1, -1, false, 4, 2, {{{3,2,1.09,0.31,0.55,2.0,-2,99}}, {{2,2,1.07,0.27,0.49,2.2,-3,97}},
            {{1,2,1.02,0.21,0.36,2.8,-6,92}}, {{0,2,0.80,0.064,0.17,4.8,-16,72}}, {{4,1,1.08,0.28,0.54,2.0,-2,98}},
            {{3,1,1.06,0.25,0.46,2.3,-4,96}}, {{2,1,0.99,0.17,0.30,3.3,-10,90}}}},
        {3, -2, false, 5, 5, {{{5,5,0.208,0.030,0.072,2.9,-47,77}}}},
        {4, -5, false, 12, 8, {{{0,0,0.22,0.061,0.22,1.0,-15,74}}, {{6,5,0.28,0.21,0.47,0.6,-7,93}},
            {{5,5,0.27,0.17,0.39,0.7,-9,90}}, {{4,5,0.25,0.10,0.31,0.8,-10,83}}, {{3,5,0.23,0.065,0.25,0.9,-11,76}}}},
        {5, -4, false, 25, 10, {{{10,6,0.163,0.068,0.16,1.0,-19,85}}, {{8,6,0.146,0.039,0.11,1.3,-29,76}}}},
    };
    return t;
}

struct GapParams {
    std::vector<Row> affine; bool has_linear = false; Row linear{};
    int open_max = 0, ext_max = 0; bool round_down = false;
};
// s_GetNuclValuesArray + s_SplitArrayOf8 + s_AdjustGapParametersByGcd
int gap_params_for(int reward, int penalty, GapParams &gp) {
    int div = gcd_pos(reward, penalty);
    if (div != 1) { reward /= div; penalty /= div; }
    const Table *tb = nullptr;
    for (const auto &t : tables()) if (t.reward == reward && t.penalty == penalty) { tb = &t; break; }
    if (!tb) return -1;
    gp.round_down = tb->round_down; gp.open_max = tb->open_max; gp.ext_max = tb->ext_max;
    size_t start = 0;
    if (tb->rows[0].v[0] == 0 && tb->rows[0].v[1] == 0) { gp.has_linear = true; gp.linear = tb->rows[0]; start = 1; }
    gp.affine.assign(tb->rows.begin() + start, tb->rows.end());
    if (div != 1) {
        if (gp.affine.empty()) return 1;
        gp.open_max *= div; gp.ext_max *= div;
        for (auto &r : gp.affine) { r.v[0] *= div; r.v[1] *= div; r.v[2] /= div; r.v[5] /= div; }
        if (gp.has_linear) { gp.linear.v[0] *= div; gp.linear.v[1] *= div; gp.linear.v[2] /= div; gp.linear.v[5] /= div; }
    }
    return 0;
}
}  // namespace

static const uint8_t kTo4na[16] = {1, 2, 4, 8, 5, 10, 3, 12, 9, 6, 14, 13, 11, 7, 15, 0};

void build_score_matrix(int reward, int penalty, int32_t m[16][16]) {
    int degen[16];
    for (int i = 0; i < 16; i++) {
        int d = 0;
        for (int j = 0; j < 4; j++) if (kTo4na[i] & kTo4na[j]) d++;
        degen[i] = i < 4 ? 1 : d;
    }
    for (int i = 0; i < 16; i++)
        for (int j = i; j < 16; j++) {
            int32_t v = penalty;
            if (kTo4na[i] & kTo4na[j])
                v = (int32_t)round_half_away((double)((degen[j] - 1) * penalty + reward) / (double)degen[j]);
            m[i][j] = v; m[j][i] = v;
        }
    for (int i = 0; i < 16; i++) { m[15][i] = INT32_MIN / 2; m[i][15] = INT32_MIN / 2; }
}

void uniform_acgt(double comp[16]) {
    double sum = 0.;
    for (int i = 0; i < 16; i++) comp[i] = i < 4 ? 25.00 : 0.;
    for (int i = 0; i < 16; i++) sum += comp[i];
    for (int i = 0; i < 16; i++) { comp[i] /= sum; comp[i] *= 1.0; }
}

void strand_composition(const uint8_t *seq, int32_t len, double comp[16]) {
    // (four histograms: one counter array makes every increment wait for the store of the one before when letters repeat -- the loop was
    // 12 of the 14 ms of CPU a 5 Mb batch's per-context set-up took)
    int32_t c4[4][16] = {{0}};
    int32_t i = 0;
    for (; i + 4 <= len; i += 4) { ++c4[0][seq[i] & 0x0f]; ++c4[1][seq[i + 1] & 0x0f]; ++c4[2][seq[i + 2] & 0x0f]; ++c4[3][seq[i + 3] & 0x0f]; }
    for (; i < len; i++) ++c4[0][seq[i] & 0x0f];
    int32_t cnt[16];
    for (int k = 0; k < 16; k++) cnt[k] = c4[0][k] + c4[1][k] + c4[2][k] + c4[3][k];
    cnt[14] = 0; cnt[15] = 0;       // 'N' and '-' are not counted
    double sum = 0.;
    for (int i = 0; i < 16; i++) sum += cnt[i];
    for (int i = 0; i < 16; i++) comp[i] = (sum == 0.) ? 0.0 : cnt[i] / sum;
}

bool ungapped_karlin(int reward, int penalty, const double c1[16], const double c2[16], Karlin &out) {
    // (the matrix and its score range are the same for every context of a batch: kept with the thread)
    struct Mat { int reward = 0, penalty = 0, lo = 0, hi = 0; bool made = false; int32_t m[16][16]; };
    thread_local Mat M;
    if (!M.made || M.reward != reward || M.penalty != penalty) {
        build_score_matrix(reward, penalty, M.m);
        int lo0 = kScoreMax, hi0 = kScoreMin;
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
            int s = M.m[i][j];
            if (s <= kScoreMin || s >= kScoreMax) continue;
            lo0 = std::min(lo0, s); hi0 = std::max(hi0, s);
        }
        M.reward = reward; M.penalty = penalty; M.lo = lo0; M.hi = hi0; M.made = true;
    }
    const int32_t (*m)[16] = M.m;
    const int lo = M.lo, hi = M.hi;
    out = Karlin(); out.logK = HUGE_VAL;
    if (!range_ok(lo, hi)) return false;
    ScoreDist d; d.lo = lo; d.hi = hi; d.mass.assign((size_t)(hi - lo + 1), 0.0);
    // (letters that do not occur add 0.0 to a sum: left out, the sums are the same doubles)
    for (int i = 0; i < 16; i++) {
        if (c1[i] == 0.0) continue;
        for (int j = 0; j < 16; j++) {
            if (c2[j] == 0.0) continue;
            int s = m[i][j];
            if (s >= lo) d.at(s) += c1[i] * c2[j];
        }
    }
    double total = 0.; int omin = kScoreMin, omax = kScoreMin;
    for (int s = lo; s <= hi; s++) if (d.at(s) > 0.) { total += d.at(s); omax = s; if (omin == kScoreMin) omin = s; }
    d.obs_lo = omin; d.obs_hi = omax;
    double avg = 0.0;
    if (total > 0.0001 || total < -0.0001)
        for (int s = omin; s <= omax; s++) { d.at(s) /= total; avg += s * d.at(s); }
    d.mean = avg;
    // lambda, H and K are a function of the score distribution alone, and the 10,000 contexts of a nucleotide batch share a handful of
    // them (match / mismatch against uniform background frequencies: 1/4 and 3/4 whatever the query's composition, up to the last bit of
    // the sums): the solutions of the distributions this thread has seen are kept and found again by their exact bits -- the same
    // doubles as solving again, a tenth of the time (round 6: the set-up's per-context loop was 14 ms of CPU per 5 Mb batch)
    struct Solved { int lo, hi, obs_lo, obs_hi; double mean; std::vector<double> mass; bool ok; Karlin k; };
    thread_local std::vector<Solved> memo;
    for (const Solved &m0 : memo)
        if (m0.lo == d.lo && m0.hi == d.hi && m0.obs_lo == d.obs_lo && m0.obs_hi == d.obs_hi && std::memcmp(&m0.mean, &d.mean, sizeof(double)) == 0 &&
            m0.mass.size() == d.mass.size() && std::memcmp(m0.mass.data(), d.mass.data(), d.mass.size() * sizeof(double)) == 0) {
            if (m0.ok) out = m0.k;
            return m0.ok;
        }
    auto remember = [&](bool ok) { if (memo.size() >= 64) memo.erase(memo.begin()); memo.push_back(Solved{d.lo, d.hi, d.obs_lo, d.obs_hi, d.mean, d.mass, ok, out}); return ok; };
    double lam;
    if (!solve_lambda(d, lam) || lam < 0.) return remember(false);
    double H = entropy_H(d, lam);
    if (H < 0.) return remember(false);
    double K = solve_K(d, lam, H);
    if (K < 0.) return remember(false);
    out.lambda = lam; out.H = H; out.K = K; out.logK = std::log(K);
    return remember(true);
}

int gapped_karlin(int gap_open, int gap_extend, int reward, int penalty, const Karlin &ungapped,
                  Karlin &out, bool &round_down) {
    GapParams gp;
    int st = gap_params_for(reward, penalty, gp);
    round_down = gp.round_down;
    if (st) return st;
    auto take = [&](const Row &r) { out.lambda = r.v[2]; out.K = r.v[3]; out.logK = std::log(out.K); out.H = r.v[4]; };
    if (gap_open == 0 && gap_extend == 0 && gp.has_linear) { take(gp.linear); return 0; }
    for (const auto &r : gp.affine)
        if (r.v[0] == gap_open && r.v[1] == gap_extend) { take(r); return 0; }
    if (gap_open >= gp.open_max && gap_extend >= gp.ext_max) { out = ungapped; return 0; }
    return 1;
}

int alpha_beta(int reward, int penalty, int gap_open, int gap_extend, const Karlin &ungapped,
               bool gapped, double &alpha, double &beta) {
    GapParams gp;
    int st = gap_params_for(reward, penalty, gp);
    if (st) return st;
    if (gapped && !gp.affine.empty()) {
        if (gap_open == 0 && gap_extend == 0 && gp.has_linear) { alpha = gp.linear.v[5]; beta = gp.linear.v[6]; return 0; }
        for (const auto &r : gp.affine)
            if (r.v[0] == gap_open && r.v[1] == gap_extend) { alpha = r.v[5]; beta = r.v[6]; return 0; }
    }
    alpha = ungapped.lambda / ungapped.H;
    beta = ((reward == 1 && penalty == -1) || (reward == 2 && penalty == -3)) ? -2 : 0;
    return 0;
}

int32_t length_adjustment(double K, double logK, double adl, double beta, int32_t qlen,
                          int64_t db_len, int32_t db_nseq) {
    const double m = (double)qlen, n = (double)db_len, N = (double)db_nseq;
    double ell_min = 0, ell_max, ell_next = 0, ell, ss;
    bool converged = false;
    {
        double a = N, mb = m * N + n, c = n * m - std::max(m, n) / K;
        if (c < 0) return 0;
        ell_max = 2 * c / (mb + std::sqrt(mb * mb - 4 * a * c));
    }
    for (int i = 1; i <= 20; i++) {
        ell = ell_next;
        ss = (m - ell) * (n - N * ell);
        double ell_bar = adl * (logK + std::log(ss)) + beta;
        if (ell_bar >= ell) {
            ell_min = ell;
            if (ell_bar - ell_min <= 1.0) { converged = true; break; }
            if (ell_min == ell_max) break;
        } else {
            ell_max = ell;
        }
        if (ell_min <= ell_bar && ell_bar <= ell_max) ell_next = ell_bar;
        else ell_next = (i == 1) ? ell_max : (ell_min + ell_max) / 2;
    }
    int32_t adj = (int32_t)ell_min;
    if (converged) {
        ell = std::ceil(ell_min);
        if (ell <= ell_max) {
            ss = (m - ell) * (n - N * ell);
            if (adl * (logK + std::log(ss)) + beta >= ell) adj = (int32_t)ell;
        }
    }
    return adj;
}

int32_t score_for_evalue(double E, const Karlin &k, int64_t searchsp) {
    if (k.lambda < 0. || k.K < 0. || k.H < 0.0) return kScoreMin;
    E = std::max(E, 1.0e-297);
    return (int32_t)(std::ceil(std::log((double)(k.K * searchsp / E)) / k.lambda));
}

double evalue_for_score(int32_t S, const Karlin &k, int64_t searchsp) {
    if (k.lambda < 0. || k.K < 0. || k.H < 0.) return -1.;
    return (double)searchsp * std::exp((double)(-k.lambda * S) + k.logK);
}

int32_t cutoff_from_evalue(double E, const Karlin &k, int64_t searchsp) {
    // BLAST_Cutoffs with S = 1 on entry, no decay
    int32_t s = 1, es = 1;
    if (k.lambda == -1. || k.K == -1. || k.H == -1.) return s;
    if (E > 0.) es = score_for_evalue(E, k, searchsp);
    if (es > s) s = es;
    return s;
}

}  // namespace gbn
