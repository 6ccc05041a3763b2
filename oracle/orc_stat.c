/* orc_stat.c -- ORACLE (test infrastructure): Karlin-Altschul statistics for
 * nucleotide scoring, restated from CORE/blast_stat.c and CORE/ncbi_math.c.
 * Floating-point operations are kept in the reference's order so doubles
 * come out bit-identical on the same libm. */
#include "orc_int.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SCORE_MIN (-32768)  /* BLAST_SCORE_MIN = INT2_MIN, COREI/blast_stat.h:121 */
#define SCORE_MAX 32767

/* CORE/ncbi_math.c:38-61 */
static double expm1_ncbi(double x)
{
    double absx = x < 0 ? -x : x;
    if (absx > .33) return exp(x) - 1.;
    if (absx < 1.e-16) return x;
    return x * (1. + x *
             (1./2. + x *
             (1./6. + x *
             (1./24. + x *
             (1./120. + x *
             (1./720. + x *
             (1./5040. + x *
             (1./40320. + x *
             (1./362880. + x *
             (1./3628800. + x *
             (1./39916800. + x *
             (1./479001600. +
              x/6227020800.))))))))))));
}

/* CORE/ncbi_math.c:410-424 */
int orc_gcd(int a, int b)
{
    int c;
    if (b < 0) b = -b;
    if (b > a) { c = a; a = b; b = c; }
    while (b != 0) { c = a % b; a = b; b = c; }
    return a;
}

/* CORE/ncbi_math.c:426-440 */
int orc_gdb3(int *a, int *b, int *c)
{
    int g;
    if (*b == 0) g = orc_gcd(*a, *c);
    else g = orc_gcd(*a, orc_gcd(*b, *c));
    if (g > 1) { *a /= g; *b /= g; *c /= g; }
    return g;
}

/* CORE/ncbi_math.c:442-446 */
long orc_nint(double x)
{
    x += (x >= 0. ? 0.5 : -0.5);
    return (long)x;
}

/* CORE/ncbi_math.c:449-475 */
static double powi_ncbi(double x, int n)
{
    double y;
    if (n == 0) return 1.;
    if (x == 0.) { if (n < 0) return HUGE_VAL; return 0.; }
    if (n < 0) { x = 1./x; n = -n; }
    y = 1.;
    while (n > 0) {
        if (n & 1) y *= x;
        n /= 2;
        x *= x;
    }
    return y;
}

/* BLASTNA -> NCBI4NA, CORE/blast_encoding.c:61-78 */
static const uint8_t kBlastnaTo4na[16] =
    { 1, 2, 4, 8, 5, 10, 3, 12, 9, 6, 14, 13, 11, 7, 15, 0 };

/* CORE/blast_stat.c:1036-1106 (BlastScoreBlkNuclMatrixCreate) */
void orc_nucl_matrix(int reward, int penalty, int32_t m[16][16])
{
    int i, j, degen[16];
    for (i = 0; i < 16; i++) for (j = 0; j < 16; j++) m[i][j] = 0;
    for (i = 0; i < 4; i++) degen[i] = 1;
    for (i = 4; i < 16; i++) {
        int d = 0;
        for (j = 0; j < 4; j++)
            if (kBlastnaTo4na[i] & kBlastnaTo4na[j]) d++;
        degen[i] = d;
    }
    for (i = 0; i < 16; i++) {
        for (j = i; j < 16; j++) {
            if (kBlastnaTo4na[i] & kBlastnaTo4na[j]) {
                m[i][j] = (int32_t)orc_nint((double)((degen[j] - 1) * penalty + reward) /
                                            (double)degen[j]);
                if (i != j) m[j][i] = m[i][j];
            } else {
                m[i][j] = penalty;
                m[j][i] = penalty;
            }
        }
    }
    for (i = 0; i < 16; i++) m[15][i] = ORC_INT4_MIN / 2;
    for (i = 0; i < 16; i++) m[i][15] = ORC_INT4_MIN / 2;
}

/* score frequency block, CORE/blast_stat.c:2087-2118 */
typedef struct {
    int score_min, score_max, obs_min, obs_max;
    double score_avg;
    double *sprob0, *sprob;   /* sprob is centred at score 0 */
} ScoreFreq;

static ScoreFreq *sfreq_new(int lo, int hi)
{
    ScoreFreq *s;
    /* BlastScoreChk, :2074-2084 */
    if (lo >= 0 || hi <= 0 || lo < SCORE_MIN || hi > SCORE_MAX) return NULL;
    s = (ScoreFreq *)calloc(1, sizeof(*s));
    s->sprob0 = (double *)calloc((size_t)(hi - lo + 1), sizeof(double));
    s->sprob = s->sprob0 - lo;
    s->score_min = lo; s->score_max = hi;
    return s;
}
static void sfreq_free(ScoreFreq *s) { if (s) { free(s->sprob0); free(s); } }

/* CORE/blast_stat.c:2125-2191 (BlastScoreFreqCalc) */
static void sfreq_calc(const int32_t m[16][16], int loscore, ScoreFreq *sfp,
                       const double *p1, const double *p2)
{
    int score, obs_min, obs_max, i, j;
    double score_sum, score_avg;
    for (score = sfp->score_min; score <= sfp->score_max; score++)
        sfp->sprob[score] = 0.0;
    for (i = 0; i < 16; i++)
        for (j = 0; j < 16; j++) {
            score = m[i][j];
            if (score >= loscore)
                sfp->sprob[score] += p1[i] * p2[j];
        }
    score_sum = 0.;
    obs_min = obs_max = SCORE_MIN;
    for (score = sfp->score_min; score <= sfp->score_max; score++) {
        if (sfp->sprob[score] > 0.) {
            score_sum += sfp->sprob[score];
            obs_max = score;
            if (obs_min == SCORE_MIN) obs_min = score;
        }
    }
    sfp->obs_min = obs_min;
    sfp->obs_max = obs_max;
    score_avg = 0.0;
    if (score_sum > 0.0001 || score_sum < -0.0001) {
        for (score = obs_min; score <= obs_max; score++) {
            sfp->sprob[score] /= score_sum;
            score_avg += score * sfp->sprob[score];
        }
    }
    sfp->score_avg = score_avg;
}

/* CORE/blast_stat.c:2464-2537 (NlmKarlinLambdaNR) */
static double lambda_nr(double *probs, int d, int low, int high, double lambda0,
                        double tolx, int itmax, int maxNewton)
{
    int k;
    double x0, x, a = 0, b = 1;
    double f = 4;
    int isNewton = 0;
    x0 = exp(-lambda0);
    x = (0 < x0 && x0 < 1) ? x0 : .5;
    for (k = 0; k < itmax; k++) {
        int i;
        double g, fold = f;
        int wasNewton = isNewton;
        isNewton = 0;
        g = 0;
        f = probs[low];
        for (i = low + d; i < 0; i += d) {
            g = x * g + f;
            f = f * x + probs[i];
        }
        g = x * g + f;
        f = f * x + probs[0] - 1;
        for (i = d; i <= high; i += d) {
            g = x * g + f;
            f = f * x + probs[i];
        }
        if (f > 0) a = x;
        else if (f < 0) b = x;
        else break;
        if (b - a < 2 * a * (1 - b) * tolx) { x = (a + b) / 2; break; }
        if (k >= maxNewton || (wasNewton && fabs(f) > .9 * fabs(fold)) || g >= 0) {
            x = (a + b) / 2;
        } else {
            double p = -f / g;
            double y = x + p;
            if (y <= a || y >= b) {
                x = (a + b) / 2;
            } else {
                isNewton = 1;
                x = y;
                if (fabs(p) < tolx * x * (1 - x)) break;
            }
        }
    }
    return -log(x) / d;
}

/* CORE/blast_stat.c:2540-2572 (Blast_KarlinLambdaNR) */
static double karlin_lambda(ScoreFreq *sfp, double guess)
{
    int low = sfp->obs_min, high = sfp->obs_max, i, d;
    double *sprob;
    if (sfp->score_avg >= 0.) return -1.0;
    if (low >= 0 || high <= 0 || low < SCORE_MIN || high > SCORE_MAX) return -1.;
    sprob = sfp->sprob;
    for (i = 1, d = -low; i <= high - low && d > 1; ++i)
        if (sprob[i + low] != 0.0) d = orc_gcd(d, i);
    return lambda_nr(sprob, d, low, high, guess, 1.e-5, 20, 20 + 17);
}

/* CORE/blast_stat.c:2580-2607 (BlastKarlinLtoH) */
static double karlin_LtoH(ScoreFreq *sfp, double lambda)
{
    int score;
    double H, etonlam, sum, scale;
    double *probs = sfp->sprob;
    int low = sfp->obs_min, high = sfp->obs_max;
    if (lambda < 0.) return -1.;
    if (low >= 0 || high <= 0 || low < SCORE_MIN || high > SCORE_MAX) return -1.;
    etonlam = exp(-lambda);
    sum = low * probs[low];
    for (score = low + 1; score <= high; score++)
        sum = score * probs[score] + etonlam * sum;
    scale = powi_ncbi(etonlam, high);
    if (scale > 0.0) H = lambda * sum / scale;
    else H = lambda * exp(lambda * high + log(sum));
    return H;
}

/* CORE/blast_stat.c:2220-2393 (BlastKarlinLHtoK) */
static double karlin_LHtoK(ScoreFreq *sfp, double lambda, double H)
{
    double *asp = NULL;     /* alignmentScoreProbabilities */
    int low, high, range, i, iterCounter, divisor;
    int lowAS, highAS, first, last;
    double K, innerSum, oldsum, oldsum2, outerSum, score_avg;
    double firstTermClosedForm, sumlimit, expMinusLambda;
    int iterlimit;
    double *probArrayStartLow, *ptrP, *ptr1, *ptr2, *ptr1e;

    if (lambda <= 0. || H <= 0.) return -1.;
    if (sfp->score_avg >= 0.0) return -1.;
    low = sfp->obs_min; high = sfp->obs_max; range = high - low;
    probArrayStartLow = &sfp->sprob[low];
    for (i = 1, divisor = -low; i <= range && divisor > 1; ++i)
        if (probArrayStartLow[i] != 0.0) divisor = orc_gcd(divisor, i);
    high /= divisor; low /= divisor; lambda *= divisor;
    range = high - low;
    firstTermClosedForm = H / lambda;
    expMinusLambda = exp((double)-lambda);
    if (low == -1 && high == 1) {
        K = (sfp->sprob[low * divisor] - sfp->sprob[high * divisor]) *
            (sfp->sprob[low * divisor] - sfp->sprob[high * divisor]) /
            sfp->sprob[low * divisor];
        return K;
    }
    if (low == -1 || high == 1) {
        if (high != 1) {
            score_avg = sfp->score_avg / divisor;
            firstTermClosedForm = (score_avg * score_avg) / firstTermClosedForm;
        }
        return firstTermClosedForm * (1.0 - expMinusLambda);
    }
    sumlimit = 0.0001;      /* BLAST_KARLIN_K_SUMLIMIT_DEFAULT */
    iterlimit = 100;        /* BLAST_KARLIN_K_ITER_MAX */
    asp = (double *)calloc((size_t)(iterlimit * range + 1), sizeof(double));
    if (!asp) return -1.;
    outerSum = 0.;
    lowAS = highAS = 0;
    asp[0] = innerSum = oldsum = oldsum2 = 1.;
    for (iterCounter = 0; (iterCounter < iterlimit) && (innerSum > sumlimit);
         outerSum += innerSum /= ++iterCounter) {
        first = last = range;
        lowAS += low;
        highAS += high;
        for (ptrP = asp + (highAS - lowAS); ptrP >= asp; *ptrP-- = innerSum) {
            ptr1 = ptrP - first;
            ptr1e = ptrP - last;
            ptr2 = probArrayStartLow + first;
            for (innerSum = 0.; ptr1 >= ptr1e; ) {
                innerSum += *ptr1 * *ptr2;
                ptr1--;
                ptr2++;
            }
            if (first) --first;
            if (ptrP - asp <= range) --last;
        }
        innerSum = *++ptrP;
        for (i = lowAS + 1; i < 0; i++)
            innerSum = *++ptrP + innerSum * expMinusLambda;
        innerSum *= expMinusLambda;
        for (; i <= highAS; ++i)
            innerSum += *++ptrP;
        oldsum2 = oldsum;
        oldsum = innerSum;
    }
    (void)oldsum2;
    K = -exp((double)-2.0 * outerSum) / (firstTermClosedForm * expm1_ncbi(-(double)lambda));
    free(asp);
    return K;
}

/* CORE/blast_stat.c:2673-2708 (Blast_KarlinBlkUngappedCalc) */
static int karlin_block(ScoreFreq *sfp, OrcKarlin *kbp)
{
    kbp->Lambda = karlin_lambda(sfp, 0.5);
    if (kbp->Lambda < 0.) goto err;
    kbp->H = karlin_LtoH(sfp, kbp->Lambda);
    if (kbp->H < 0.) goto err;
    kbp->K = karlin_LHtoK(sfp, kbp->Lambda, kbp->H);
    if (kbp->K < 0.) goto err;
    kbp->logK = log(kbp->K);
    return 0;
err:
    kbp->Lambda = kbp->H = kbp->K = -1.;
    kbp->logK = HUGE_VAL;
    return 1;
}

/* loscore/hiscore, CORE/blast_stat.c:1476-1507 (BlastScoreBlkMaxScoreSet) */
static void matrix_lohi(const int32_t m[16][16], int *lo, int *hi)
{
    int i, j;
    *lo = SCORE_MAX; *hi = SCORE_MIN;
    for (i = 0; i < 16; i++)
        for (j = 0; j < 16; j++) {
            int s = m[i][j];
            if (s <= SCORE_MIN || s >= SCORE_MAX) continue;
            if (*lo > s) *lo = s;
            if (*hi < s) *hi = s;
        }
    if (*lo < SCORE_MIN) *lo = SCORE_MIN;
    if (*hi > SCORE_MAX) *hi = SCORE_MAX;
}

int orc_karlin_ungapped(int reward, int penalty, const double *p1,
                        const double *p2, OrcKarlin *out)
{
    int32_t m[16][16];
    int lo, hi, rc;
    ScoreFreq *sfp;
    orc_nucl_matrix(reward, penalty, m);
    matrix_lohi(m, &lo, &hi);
    sfp = sfreq_new(lo, hi);
    if (!sfp) { out->Lambda = out->H = out->K = -1.; out->logK = HUGE_VAL; return 1; }
    sfreq_calc(m, lo, sfp, p1, p2);
    rc = karlin_block(sfp, out);
    sfreq_free(sfp);
    return rc;
}

/* standard nucleotide composition: 25 each, normalised
 * (CORE/blast_stat.c:1794-1799, :1861-1891, :1809-1834) */
void orc_std_nt_freq(double p[16])
{
    int i; double sum = 0.;
    for (i = 0; i < 16; i++) p[i] = 0.;
    for (i = 0; i < 4; i++) p[i] = 25.00;
    for (i = 0; i < 16; i++) sum += p[i];
    for (i = 0; i < 16; i++) { p[i] /= sum; p[i] *= 1.0; }
}

/* CORE/blast_stat.c:2810-2832 (Blast_ScoreBlkKbpIdealCalc) */
int orc_karlin_ideal(int reward, int penalty, OrcKarlin *out)
{
    double p[16];
    orc_std_nt_freq(p);
    return orc_karlin_ungapped(reward, penalty, p, p, out);
}

/* composition of one query context: CORE/blast_stat.c:1958-1995
 * (BlastResCompStr, mask 0x0f, N and '-' zeroed) + :2018-2043 */
void orc_context_freq(const uint8_t *seq, int32_t len, double p[16])
{
    int32_t comp[16]; int i; double sum = 0.;
    for (i = 0; i < 16; i++) comp[i] = 0;
    for (i = 0; i < len; i++) ++comp[seq[i] & 0x0f];
    comp[14] = 0;           /* 'N' */
    comp[15] = 0;           /* '-' */
    for (i = 0; i < 16; i++) sum += comp[i];
    if (sum == 0.) { for (i = 0; i < 16; i++) p[i] = 0.0; return; }
    for (i = 0; i < 16; i++) p[i] = comp[i] / sum;
}

/* ---- gapped parameter tables, CORE/blast_stat.c:575-700 ---- */
typedef double Row8[8];
static const Row8 v_1_5[] = { {0,0,1.39,0.747,1.38,1.00,0,100}, {3,3,1.39,0.747,1.38,1.00,0,100} };
static const Row8 v_1_4[] = { {0,0,1.383,0.738,1.36,1.02,0,100}, {1,2,1.36,0.67,1.2,1.1,0,98},
    {0,2,1.26,0.43,0.90,1.4,-1,91}, {2,1,1.35,0.61,1.1,1.2,-1,98}, {1,1,1.22,0.35,0.72,1.7,-3,88} };
static const Row8 v_2_7[] = { {0,0,0.69,0.73,1.34,0.515,0,100}, {2,4,0.68,0.67,1.2,0.55,0,99},
    {0,4,0.63,0.43,0.90,0.7,-1,91}, {4,2,0.675,0.62,1.1,0.6,-1,98}, {2,2,0.61,0.35,0.72,1.7,-3,88} };
static const Row8 v_1_3[] = { {0,0,1.374,0.711,1.31,1.05,0,100}, {2,2,1.37,0.70,1.2,1.1,0,99},
    {1,2,1.35,0.64,1.1,1.2,-1,98}, {0,2,1.25,0.42,0.83,1.5,-2,91}, {2,1,1.34,0.60,1.1,1.2,-1,97},
    {1,1,1.21,0.34,0.71,1.7,-2,88} };
static const Row8 v_2_5[] = { {0,0,0.675,0.65,1.1,0.6,-1,99}, {2,4,0.67,0.59,1.1,0.6,-1,98},
    {0,4,0.62,0.39,0.78,0.8,-2,91}, {4,2,0.67,0.61,1.0,0.65,-2,98}, {2,2,0.56,0.32,0.59,0.95,-4,82} };
static const Row8 v_1_2[] = { {0,0,1.28,0.46,0.85,1.5,-2,96}, {2,2,1.33,0.62,1.1,1.2,0,99},
    {1,2,1.30,0.52,0.93,1.4,-2,97}, {0,2,1.19,0.34,0.66,1.8,-3,89}, {3,1,1.32,0.57,1.0,1.3,-1,99},
    {2,1,1.29,0.49,0.92,1.4,-1,96}, {1,1,1.14,0.26,0.52,2.2,-5,85} };
static const Row8 v_2_3[] = { {0,0,0.55,0.21,0.46,1.2,-5,87}, {4,4,0.63,0.42,0.84,0.75,-2,99},
    {2,4,0.615,0.37,0.72,0.85,-3,97}, {0,4,0.55,0.21,0.46,1.2,-5,87}, {3,3,0.615,0.37,0.68,0.9,-3,97},
    {6,2,0.63,0.42,0.84,0.75,-2,99}, {5,2,0.625,0.41,0.78,0.8,-2,99}, {4,2,0.61,0.35,0.68,0.9,-3,96},
    {2,2,0.515,0.14,0.33,1.55,-9,81} };
static const Row8 v_3_4[] = { {6,3,0.389,0.25,0.56,0.7,-5,95}, {5,3,0.375,0.21,0.47,0.8,-6,92},
    {4,3,0.351,0.14,0.35,1.0,-9,86}, {6,2,0.362,0.16,0.45,0.8,-4,88}, {5,2,0.330,0.092,0.28,1.2,-13,81},
    {4,2,0.281,0.046,0.16,1.8,-23,69} };
static const Row8 v_4_5[] = { {0,0,0.22,0.061,0.22,1.0,-15,74}, {6,5,0.28,0.21,0.47,0.6,-7,93},
    {5,5,0.27,0.17,0.39,0.7,-9,90}, {4,5,0.25,0.10,0.31,0.8,-10,83}, {3,5,0.23,0.065,0.25,0.9,-11,76} };
static const Row8 v_1_1[] = { {3,2,1.09,0.31,0.55,2.0,-2,99}, {2,2,1.07,0.27,0.49,2.2,-3,97},
    {1,2,1.02,0.21,0.36,2.8,-6,92}, {0,2,0.80,0.064,0.17,4.8,-16,72}, {4,1,1.08,0.28,0.54,2.0,-2,98},
    {3,1,1.06,0.25,0.46,2.3,-4,96}, {2,1,0.99,0.17,0.30,3.3,-10,90} };
static const Row8 v_3_2[] = { {5,5,0.208,0.030,0.072,2.9,-47,77} };
static const Row8 v_5_4[] = { {10,6,0.163,0.068,0.16,1.0,-19,85}, {8,6,0.146,0.039,0.11,1.3,-29,76} };

#define NROWS(a) ((int)(sizeof(a) / sizeof(Row8)))

/* CORE/blast_stat.c:3209-3343 (s_GetNuclValuesArray + s_SplitArrayOf8 +
 * s_AdjustGapParametersByGcd).  normal[]/linear are caller buffers. */
static int nucl_values(int reward, int penalty, int *n_normal, Row8 *normal,
                       int *has_linear, Row8 linear, int *gap_open_max,
                       int *gap_extend_max, int *round_down)
{
    const Row8 *tab = NULL; int n = 0, i, j, split = 0;
    int divisor = orc_gcd(reward, penalty);
    *round_down = 0; *n_normal = 0; *has_linear = 0;
    if (divisor != 1) { reward /= divisor; penalty /= divisor; }
    if (reward == 1 && penalty == -5) { tab = v_1_5; n = NROWS(v_1_5); *gap_open_max = 3; *gap_extend_max = 3; }
    else if (reward == 1 && penalty == -4) { tab = v_1_4; n = NROWS(v_1_4); *gap_open_max = 2; *gap_extend_max = 2; }
    else if (reward == 2 && penalty == -7) { tab = v_2_7; n = NROWS(v_2_7); *round_down = 1; *gap_open_max = 4; *gap_extend_max = 4; }
    else if (reward == 1 && penalty == -3) { tab = v_1_3; n = NROWS(v_1_3); *gap_open_max = 2; *gap_extend_max = 2; }
    else if (reward == 2 && penalty == -5) { tab = v_2_5; n = NROWS(v_2_5); *round_down = 1; *gap_open_max = 4; *gap_extend_max = 4; }
    else if (reward == 1 && penalty == -2) { tab = v_1_2; n = NROWS(v_1_2); *gap_open_max = 2; *gap_extend_max = 2; }
    else if (reward == 2 && penalty == -3) { tab = v_2_3; n = NROWS(v_2_3); *round_down = 1; *gap_open_max = 6; *gap_extend_max = 4; }
    else if (reward == 3 && penalty == -4) { tab = v_3_4; n = NROWS(v_3_4); *round_down = 1; *gap_open_max = 6; *gap_extend_max = 3; }
    else if (reward == 1 && penalty == -1) { tab = v_1_1; n = NROWS(v_1_1); *gap_open_max = 4; *gap_extend_max = 2; }
    else if (reward == 3 && penalty == -2) { tab = v_3_2; n = NROWS(v_3_2); *gap_open_max = 5; *gap_extend_max = 5; }
    else if (reward == 4 && penalty == -5) { tab = v_4_5; n = NROWS(v_4_5); *gap_open_max = 12; *gap_extend_max = 8; }
    else if (reward == 5 && penalty == -4) { tab = v_5_4; n = NROWS(v_5_4); *gap_open_max = 25; *gap_extend_max = 10; }
    else return -1;
    if (tab[0][0] == 0 && tab[0][1] == 0) {
        split = 1;
        for (j = 0; j < 8; j++) linear[j] = tab[0][j];
        *has_linear = 1;
        tab++; n--;
    }
    (void)split;
    for (i = 0; i < n; i++) for (j = 0; j < 8; j++) normal[i][j] = tab[i][j];
    *n_normal = n;
    if (divisor != 1) {
        if (n <= 0) return 1;
        *gap_open_max *= divisor; *gap_extend_max *= divisor;
        for (i = 0; i < n; i++) {
            normal[i][0] *= divisor; normal[i][1] *= divisor;
            normal[i][2] /= divisor; normal[i][5] /= divisor;
        }
        if (*has_linear) {
            linear[0] *= divisor; linear[1] *= divisor;
            linear[2] /= divisor; linear[5] /= divisor;
        }
    }
    return 0;
}

/* CORE/blast_stat.c:3373-3424 (BLAST_GetNucleotideGapExistenceExtendParams): are these gap costs in the table of the
 * reward / penalty pair (or the linear pair 0 / 0 where it has one)?  If not and they lie below the table's largest,
 * they are replaced by the largest.  -1: no table for the pair.  Known answers: UT/blastoptions_unit_test.cpp:184-231. */
int orc_nucl_gap_params(int reward, int penalty, int *gap_existence, int *gap_extension)
{
    Row8 normal[16], linear; int n, has_lin, gom, gem, rd, st, i;
    st = nucl_values(reward, penalty, &n, normal, &has_lin, linear, &gom, &gem, &rd);
    if (st) return st;
    if (*gap_existence == 0 && *gap_extension == 0 && has_lin) return 0;
    for (i = 0; i < n; i++)
        if (normal[i][0] == *gap_existence && normal[i][1] == *gap_extension) return 0;
    if (*gap_existence < gom || *gap_extension < gem) { *gap_existence = gom; *gap_extension = gem; }
    return 0;
}

/* CORE/blast_stat.c:3806-3901 (Blast_KarlinBlkNuclGappedCalc) */
int orc_karlin_nucl_gapped(int gap_open, int gap_extend, int reward, int penalty,
                           const OrcKarlin *ungapped, OrcKarlin *kbp, int *round_down)
{
    Row8 normal[16], linear; int n, has_lin, gom, gem, st, i;
    st = nucl_values(reward, penalty, &n, normal, &has_lin, linear, &gom, &gem, round_down);
    if (st) return st;
    if (gap_open == 0 && gap_extend == 0 && has_lin) {
        kbp->Lambda = linear[2]; kbp->K = linear[3];
        kbp->logK = log(kbp->K); kbp->H = linear[4];
        return 0;
    }
    for (i = 0; i < n; i++) {
        if (normal[i][0] == gap_open && normal[i][1] == gap_extend) {
            kbp->Lambda = normal[i][2]; kbp->K = normal[i][3];
            kbp->logK = log(kbp->K); kbp->H = normal[i][4];
            break;
        }
    }
    if (i == n) {
        if (gap_open >= gom && gap_extend >= gem) *kbp = *ungapped;
        else return 1;
    }
    return 0;
}

/* CORE/blast_stat.c:3909-3990 (s_GetUngappedBeta, Blast_GetNuclAlphaBeta) */
int orc_nucl_alpha_beta(int reward, int penalty, int gap_open, int gap_extend,
                        const OrcKarlin *kbp, int gapped, double *alpha, double *beta)
{
    Row8 normal[16], linear; int n, has_lin, gom = 0, gem = 0, rd, st, i, found = 0;
    st = nucl_values(reward, penalty, &n, normal, &has_lin, linear, &gom, &gem, &rd);
    if (st) return st;
    if (gapped && n > 0) {
        if (gap_open == 0 && gap_extend == 0 && has_lin) {
            *alpha = linear[5]; *beta = linear[6]; found = 1;
        } else {
            for (i = 0; i < n; i++)
                if (normal[i][0] == gap_open && normal[i][1] == gap_extend) {
                    *alpha = normal[i][5]; *beta = normal[i][6]; found = 1; break;
                }
        }
    }
    if (!found) {
        double b = 0;
        *alpha = kbp->Lambda / kbp->H;
        if ((reward == 1 && penalty == -1) || (reward == 2 && penalty == -3)) b = -2;
        *beta = b;
    }
    return 0;
}

/* CORE/blast_stat.c:4994-5076 (BLAST_ComputeLengthAdjustment) */
int orc_length_adjustment(double K, double logK, double alpha_d_lambda, double beta,
                          int32_t query_length, int64_t db_length,
                          int32_t db_num_seqs, int32_t *length_adjustment)
{
    int i;
    const int kMaxIterations = 20;
    double m = (double)query_length, n = (double)db_length, N = (double)db_num_seqs;
    double ell, ss, ell_min = 0, ell_max, ell_next = 0;
    int converged = 0;
    {
        double a = N, mb = m * N + n, c = n * m - (m > n ? m : n) / K;
        if (c < 0) { *length_adjustment = 0; return 1; }
        ell_max = 2 * c / (mb + sqrt(mb * mb - 4 * a * c));
    }
    for (i = 1; i <= kMaxIterations; i++) {
        double ell_bar;
        ell = ell_next;
        ss = (m - ell) * (n - N * ell);
        ell_bar = alpha_d_lambda * (logK + log(ss)) + beta;
        if (ell_bar >= ell) {
            ell_min = ell;
            if (ell_bar - ell_min <= 1.0) { converged = 1; break; }
            if (ell_min == ell_max) break;
        } else {
            ell_max = ell;
        }
        if (ell_min <= ell_bar && ell_bar <= ell_max) ell_next = ell_bar;
        else ell_next = (i == 1) ? ell_max : (ell_min + ell_max) / 2;
    }
    if (converged) {
        *length_adjustment = (int32_t)ell_min;
        ell = ceil(ell_min);
        if (ell <= ell_max) {
            ss = (m - ell) * (n - N * ell);
            if (alpha_d_lambda * (logK + log(ss)) + beta >= ell)
                *length_adjustment = (int32_t)ell;
        }
    } else {
        *length_adjustment = (int32_t)ell_min;
    }
    return converged ? 0 : 1;
}

/* CORE/blast_stat.c:3994-4020 (BlastKarlinEtoS_simple) */
static int32_t karlin_EtoS(double E, const OrcKarlin *kbp, int64_t searchsp)
{
    const double kSmallFloat = 1.0e-297;
    if (kbp->Lambda < 0. || kbp->K < 0. || kbp->H < 0.0) return SCORE_MIN;
    if (E < kSmallFloat) E = kSmallFloat;
    return (int32_t)(ceil(log((double)(kbp->K * searchsp / E)) / kbp->Lambda));
}

/* CORE/blast_stat.c:4111-4125 (BLAST_KarlinStoE_simple) */
double orc_karlin_StoE(int32_t S, const OrcKarlin *kbp, int64_t searchsp)
{
    if (kbp->Lambda < 0. || kbp->K < 0. || kbp->H < 0.) return -1.;
    return (double)searchsp * exp((double)(-kbp->Lambda * S) + kbp->logK);
}

/* CORE/blast_stat.c:4044-4104 (BLAST_Cutoffs); dodecay with rate 0 is a no-op,
 * which is the only way the nucleotide gapped path calls it */
int orc_cutoffs(int32_t *S, double *E, const OrcKarlin *kbp, int64_t searchsp)
{
    int32_t s = *S, es; double e = *E, esave; int s_changed = 0;
    if (kbp->Lambda == -1. || kbp->K == -1. || kbp->H == -1.) return 1;
    es = 1; esave = e;
    if (e > 0.) es = karlin_EtoS(e, kbp, searchsp);
    if (es > s) { s_changed = 1; *S = s = es; }
    if (esave <= 0. || !s_changed) { e = orc_karlin_StoE(s, kbp, searchsp); *E = e; }
    return 0;
}
