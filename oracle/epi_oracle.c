/*
 * epi_oracle.c -- TEST INFRASTRUCTURE ONLY (the checker, never the product).
 *
 * Sequential CPU restatement of the reference's metapopulation step and
 * beta x travel calibration sweep.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library.  The shipped engine
 * (epidemicsimulator_amd/csrc) never links, imports or calls it.
 *
 * Parity status: the reference (Scala 3 on the JVM) cannot be compiled or run in
 * this environment (no JVM toolchain), and its own tests hold no numeric golden
 * vectors for this path.  This restatement is pinned by
 *   - JDK known answers for java.util.Random / StrictMath.log,
 *   - the three exact answers of ValidationSpec.scala:14-22,
 *   - the conservation/zero-net-flow properties of MobilitySpec.scala:24-63,
 *     CalibrationSpec.scala:6-16 and WorldToy.scala:92-95,142-145,
 *   - bit-for-bit agreement with an independently written second restatement
 *     (oracle/pyref.py).
 * Numeric trajectories beyond those anchors are "parity unpinned" with respect
 * to a live run of the Scala code.
 *
 * Every function cites the reference lines it follows (paths relative to the
 * reference tree, src/main/scala/epidemic/).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math  (see oracle/Makefile).
 * All floating point is IEEE binary64, one rounding per operation, no FMA.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* java.util.Random (JDK; used through scala.util.Random at env.scala:42,91, */
/* DDQNAgent.scala:14,28, MLP.scala:5-6).  48-bit LCG per the JDK javadoc.   */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint64_t seed;
    int have_next;
    double next_gauss;
} jrand;

static const uint64_t JR_MULT = 0x5DEECE66DULL;
static const uint64_t JR_MASK = (1ULL << 48) - 1;

static void jr_init(jrand *r, int64_t seed) {
    r->seed = ((uint64_t)seed ^ JR_MULT) & JR_MASK;
    r->have_next = 0;
    r->next_gauss = 0.0;
}

static int32_t jr_next(jrand *r, int bits) {
    r->seed = (r->seed * JR_MULT + 0xBULL) & JR_MASK;
    /* (int)(seed >>> (48 - bits)): keep the low 32 bits, reinterpret signed */
    return (int32_t)(uint32_t)(r->seed >> (48 - bits));
}

static int32_t jr_next_int(jrand *r) { return jr_next(r, 32); }

static int32_t jr_next_int_bound(jrand *r, int32_t bound) {
    int32_t v = jr_next(r, 31);
    int32_t m = bound - 1;
    if ((bound & m) == 0) {
        v = (int32_t)(((int64_t)bound * (int64_t)v) >> 31);
    } else {
        int32_t u = v;
        /* u - (v = u % bound) + m < 0 evaluated with 32-bit wraparound */
        for (;;) {
            v = u % bound;
            int32_t chk = (int32_t)((uint32_t)u - (uint32_t)v + (uint32_t)m);
            if (chk >= 0) break;
            u = jr_next(r, 31);
        }
    }
    return v;
}

static double jr_next_double(jrand *r) {
    int64_t hi = (int64_t)jr_next(r, 26);
    int64_t lo = (int64_t)jr_next(r, 27);
    return (double)((hi << 27) + lo) * 0x1.0p-53;
}

/* StrictMath.log = fdlibm __ieee754_log (e_log.c), restated from the published
 * algorithm: argument reduction x = 2^k (1+f), s = f/(2+f), degree-14 even
 * polynomial in s split into two interleaved Horner chains.                  */
static double fdlibm_log(double x) {
    static const double ln2_hi = 6.93147180369123816490e-01,
                        ln2_lo = 1.90821492927058770002e-10,
                        two54 = 1.80143985094819840000e+16,
                        Lg1 = 6.666666666666735130e-01,
                        Lg2 = 3.999999999940941908e-01,
                        Lg3 = 2.857142874366239149e-01,
                        Lg4 = 2.222219843214978396e-01,
                        Lg5 = 1.818357216161805012e-01,
                        Lg6 = 1.531383769920937332e-01,
                        Lg7 = 1.479819860511658591e-01;
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int32_t hx = (int32_t)(bits >> 32);
    uint32_t lx = (uint32_t)bits;
    int32_t k = 0, i, j;
    double hfsq, f, s, z, R, w, t1, t2, dk;

    if (hx < 0x00100000) {
        if (((hx & 0x7fffffff) | lx) == 0) return -two54 / 0.0;
        if (hx < 0) return (x - x) / 0.0;
        k -= 54;
        x *= two54;
        memcpy(&bits, &x, 8);
        hx = (int32_t)(bits >> 32);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    i = (hx + 0x95f64) & 0x100000;
    memcpy(&bits, &x, 8);
    bits = (bits & 0xffffffffULL) | ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32);
    memcpy(&x, &bits, 8);
    k += (i >> 20);
    f = x - 1.0;
    if ((0x000fffff & (2 + hx)) < 3) {
        if (f == 0.0) {
            if (k == 0) return 0.0;
            dk = (double)k;
            return dk * ln2_hi + dk * ln2_lo;
        }
        R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        dk = (double)k;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    s = f / (2.0 + f);
    dk = (double)k;
    z = s * s;
    i = hx - 0x6147a;
    w = z * z;
    j = 0x6b851 - hx;
    t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    R = t2 + t1;
    if (i > 0) {
        hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

/* Random.nextGaussian: Marsaglia polar method with a cached second value,
 * StrictMath.log and StrictMath.sqrt (sqrt is correctly rounded in C too).   */
static double jr_next_gaussian(jrand *r) {
    if (r->have_next) {
        r->have_next = 0;
        return r->next_gauss;
    }
    double v1, v2, s;
    do {
        v1 = 2 * jr_next_double(r) - 1;
        v2 = 2 * jr_next_double(r) - 1;
        s = v1 * v1 + v2 * v2;
    } while (s >= 1 || s == 0);
    double multiplier = sqrt(-2 * fdlibm_log(s) / s);
    r->next_gauss = v2 * multiplier;
    r->have_next = 1;
    return v1 * multiplier;
}

/* --- flat test hooks for the RNG ---------------------------------------- */
EXPORT void orc_jrand_ints(int64_t seed, int32_t n, int32_t *out) {
    jrand r; jr_init(&r, seed);
    for (int i = 0; i < n; i++) out[i] = jr_next_int(&r);
}
EXPORT void orc_jrand_doubles(int64_t seed, int32_t n, double *out) {
    jrand r; jr_init(&r, seed);
    for (int i = 0; i < n; i++) out[i] = jr_next_double(&r);
}
EXPORT void orc_jrand_gaussians(int64_t seed, int32_t n, double *out) {
    jrand r; jr_init(&r, seed);
    for (int i = 0; i < n; i++) out[i] = jr_next_gaussian(&r);
}
EXPORT void orc_jrand_ints_bound(int64_t seed, int32_t bound, int32_t n, int32_t *out) {
    jrand r; jr_init(&r, seed);
    for (int i = 0; i < n; i++) out[i] = jr_next_int_bound(&r, bound);
}
EXPORT double orc_strict_log(double x) { return fdlibm_log(x); }

/* ------------------------------------------------------------------------ */
/* JVM numeric conversions                                                   */
/* ------------------------------------------------------------------------ */
/* (int) of a double: saturating, NaN -> 0 (JLS 5.1.3).                      */
static int32_t d2i(double x) {
    if (x != x) return 0;
    if (x >= 2147483647.0) return INT32_MAX;
    if (x <= -2147483648.0) return INT32_MIN;
    return (int32_t)x;
}
static int64_t d2l(double x) {
    if (x != x) return 0;
    if (x >= 9223372036854775807.0) return INT64_MAX;
    if (x <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)x;
}
/* Math.round(double): nearest long, ties toward +inf.                       */
static int64_t jround(double x) {
    if (x != x) return 0;
    double fl = floor(x);
    double diff = x - fl; /* exact */
    if (diff >= 0.5) fl += 1.0;
    return d2l(fl);
}
/* math.round(x).toInt : Long -> Int keeps the low 32 bits                   */
static int32_t jround_i(double x) { return (int32_t)(uint32_t)(uint64_t)jround(x); }
static int32_t iadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static int32_t isub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }
static double clamp01(double x) { return fmax(0.0, fmin(1.0, x)); } /* env.scala:53 */

/* ------------------------------------------------------------------------ */
/* EpidemicEnv.allocateInt -- env.scala:241-254                              */
/* ------------------------------------------------------------------------ */
typedef struct { double f; int32_t j; } fracidx;

/* stable merge sort, descending by f (scala.util.Sorting.stableSort with the
 * reversed Ordering[Double]; java.lang.Double.compare total order).          */
static int dcmp(double a, double b) { /* Double.compare */
    if (a < b) return -1;
    if (a > b) return 1;
    int64_t ab, bb;
    memcpy(&ab, &a, 8); memcpy(&bb, &b, 8);
    if (a != a) ab = 0x7ff8000000000000LL;
    if (b != b) bb = 0x7ff8000000000000LL;
    return ab == bb ? 0 : (ab < bb ? -1 : 1);
}
static void msort_desc(fracidx *a, fracidx *tmp, int n) {
    if (n < 2) return;
    int h = n / 2;
    msort_desc(a, tmp, h);
    msort_desc(a + h, tmp, n - h);
    int i = 0, j = h, k = 0;
    while (i < h && j < n) {
        /* take from the right run only if it is strictly greater (stability) */
        if (dcmp(a[j].f, a[i].f) > 0) tmp[k++] = a[j++];
        else tmp[k++] = a[i++];
    }
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, (size_t)n * sizeof(fracidx));
}

EXPORT void orc_alloc_int(int32_t total, const double *w, int32_t n, int32_t *out) {
    double sumW = 0.0;
    for (int j = 0; j < n; j++) sumW = (j == 0) ? w[0] : sumW + w[j]; /* reduceLeft */
    for (int j = 0; j < n; j++) out[j] = 0;
    /* `total <= 0 || sumW <= 0.0` literally: a NaN sum falls through like the
     * reference (every raw NaN -> base 0); the engine rejects NaN weights.    */
    if (total <= 0 || sumW <= 0.0) return;
    double *raw = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    fracidx *fr = (fracidx *)malloc(sizeof(fracidx) * (size_t)(n > 0 ? n : 1));
    fracidx *tmp = (fracidx *)malloc(sizeof(fracidx) * (size_t)(n > 0 ? n : 1));
    int32_t bsum = 0;
    for (int j = 0; j < n; j++) {
        raw[j] = (w[j] / sumW) * (double)total;
        out[j] = d2i(floor(raw[j]));
        bsum = iadd(bsum, out[j]);
    }
    int32_t rem = isub(total, bsum);
    for (int j = 0; j < n; j++) { fr[j].f = raw[j] - floor(raw[j]); fr[j].j = j; }
    msort_desc(fr, tmp, n);
    int k = 0;
    while (rem > 0 && k < n) { out[fr[k].j] = iadd(out[fr[k].j], 1); rem--; k++; }
    free(raw); free(fr); free(tmp);
}

/* Same function restricted to a sparse weight vector (CSR row): entries not
 * listed have weight 0 => raw 0, base 0, frac 0.  The stable descending sort
 * puts every frac>0 entry first (ordered) and then all frac==0 entries in index
 * order, so only the listed entries need sorting; surplus units (rem larger
 * than the number of positive fractions) walk the zero-frac indices 0,1,2,...
 * `cols` must be strictly ascending.  out_vals gets the allocation of each
 * listed entry; surplus units landing on unlisted columns are reported through
 * extra_cols (capacity extra_cap), *n_extra counts them.                      */
EXPORT void orc_alloc_int_sparse(int32_t total, const double *vals, const int32_t *cols, int32_t nnz,
                                 int32_t n, int32_t *out_vals, int32_t *extra_cols, int32_t extra_cap,
                                 int32_t *n_extra) {
    *n_extra = 0;
    for (int k = 0; k < nnz; k++) out_vals[k] = 0;
    if (n <= 0) return;
    /* the dense sum adds the zeros too; adding +0.0 never changes a partial sum
     * except turning -0.0 into +0.0, which compares the same below.           */
    double sumW = 0.0;
    int started = 0;
    for (int k = 0; k < nnz; k++) { sumW = started ? sumW + vals[k] : vals[k]; started = 1; }
    if (nnz < n && started) sumW = sumW + 0.0;
    if (total <= 0 || sumW <= 0.0) return;
    fracidx *fr = (fracidx *)malloc(sizeof(fracidx) * (size_t)(nnz > 0 ? nnz : 1));
    fracidx *tmp = (fracidx *)malloc(sizeof(fracidx) * (size_t)(nnz > 0 ? nnz : 1));
    int32_t bsum = 0; int npos = 0;
    for (int k = 0; k < nnz; k++) {
        double raw = (vals[k] / sumW) * (double)total;
        out_vals[k] = d2i(floor(raw));
        bsum = iadd(bsum, out_vals[k]);
        double f = raw - floor(raw);
        if (f > 0.0) { fr[npos].f = f; fr[npos].j = k; npos++; }
    }
    int32_t rem = isub(total, bsum);
    msort_desc(fr, tmp, npos);
    int k = 0;
    while (rem > 0 && k < npos) { out_vals[fr[k].j] = iadd(out_vals[fr[k].j], 1); rem--; k++; }
    if (rem > 0) {
        /* walk indices with frac == 0 in ascending order */
        int p = 0;
        for (int32_t j = 0; j < n && rem > 0; j++) {
            while (p < nnz && cols[p] < j) p++;
            if (p < nnz && cols[p] == j) {
                double raw = (vals[p] / sumW) * (double)total;
                double f = raw - floor(raw);
                if (f > 0.0) continue;
                out_vals[p] = iadd(out_vals[p], 1); rem--;
            } else {
                if (*n_extra < extra_cap) extra_cols[*n_extra] = j;
                (*n_extra)++; rem--;
            }
        }
    }
    free(fr); free(tmp);
}

/* ------------------------------------------------------------------------ */
/* Types: Geo.Country profile fields read by env.scala (Geo.scala:4-21),     */
/* EpidemicEnv ctor (env.scala:32-41), State (env.scala:22-30).              */
/* ------------------------------------------------------------------------ */
typedef struct {
    double beta_mul, ifr_mul, noise, population_density, gdp_per_capita,
           healthcare_capacity, age_elderly, connectivity, econ_resilience;
} orc_profile;

typedef struct {
    double base_beta, base_gamma, base_ifr, noise, reward_clip, risk_factor;
} orc_env_params;

typedef struct {
    double s, i, r, d, v, hosp_cap;
    int32_t t, t_max, seed;
    int32_t pad_;
} orc_state;

static double state_n(const orc_state *s) { return s->s + s->i + s->r + s->d; } /* env.scala:26 */
static double state_util(const orc_state *s) { return s->i / fmax(1.0, s->hosp_cap); } /* :27 */

/* State.features -- env.scala:28-30 */
EXPORT void orc_features(const orc_state *s, double *f7) {
    double N = fmax(1.0, state_n(s));
    f7[0] = s->s / N; f7[1] = s->i / N; f7[2] = s->r / N; f7[3] = s->d / N; f7[4] = s->v / N;
    f7[5] = state_util(s);
    int32_t tm = s->t_max > 1 ? s->t_max : 1;
    f7[6] = (double)s->t / (double)tm;
}

/* calculateExtremeActionCost -- env.scala:196-224 */
static double action_cost(int act, const orc_profile *p) {
    static const double base[8] = {0.0, 0.05, 0.08, 0.15, 0.01, 0.06, 0.03, 0.05};
    double b = base[act];
    double g = p->gdp_per_capita;
    double em = g > 50000 ? 0.3 : (g > 20000 ? 1.0 : (g > 5000 ? 3.0 : 8.0));
    double mm = (act == 3 || act == 2) ? 1.0 + (1.0 - p->econ_resilience) * 2.0 : 1.0;
    return b * em * mm * 30.0;
}
/* npiBonus -- env.scala:227-234 */
static double npi_bonus(int act, const orc_profile *p) {
    switch (act) {
        case 3: return 0.05 * p->econ_resilience;
        case 2: return 0.03 * p->connectivity;
        case 1: return 0.02;
        case 6: return 0.02 * p->healthcare_capacity;
        case 7: return 0.04 * p->healthcare_capacity;
        default: return 0.0;
    }
}
/* calculateExtremeHeterogeneousReward -- env.scala:142-194 */
static double reward_fn(const orc_env_params *ep, const orc_profile *p, const orc_state *st,
                        const orc_state *nx, int act) {
    double Nnext = fmax(1.0, state_n(nx));
    double di = (nx->i - st->i) / Nnext;
    double dd = (nx->d - st->d) / Nnext;
    double overflow = fmax(0.0, state_util(nx) - 1.0);
    double hc = p->healthcare_capacity;
    double healthWeight = hc > 0.8 ? 0.2 : (hc > 0.5 ? 0.6 : 0.9);
    double economicWeight = 1.0 - healthWeight;
    double g = p->gdp_per_capita;
    double gdpF = g > 50000 ? 0.3 : (g > 20000 ? 1.0 : (g > 5000 ? 2.5 : 5.0));
    double hcF = hc > 0.8 ? 0.2 : (hc > 0.5 ? 1.0 : 8.0);
    double dn = p->population_density;
    double densF = dn > 400 ? 3.0 : (dn > 200 ? 2.0 : (dn > 100 ? 1.0 : 0.5));
    double ageF = 1.0 + p->age_elderly * 8.0;
    double infectionPenalty = di * 200.0 * densF * healthWeight * gdpF;
    double mortalityPenalty = dd * 2000.0 * ageF * hcF * healthWeight;
    double healthcarePenalty = overflow * 1000.0 * healthWeight * hcF;
    double actionCost = action_cost(act, p) * economicWeight * gdpF;
    double baseReward = -(infectionPenalty + mortalityPenalty + healthcarePenalty + actionCost);
    double shaped = baseReward * (1.0 + 0.75 * ep->risk_factor) + (ep->risk_factor * npi_bonus(act, p));
    return fmax(-ep->reward_clip * 3.0, fmin(ep->reward_clip, shaped));
}

/* EpidemicEnv.step -- env.scala:58-140.  `gauss` is the value rnd.nextGaussian()
 * returns at env.scala:91 (the caller owns the stream).                      */
EXPORT void orc_env_step(const orc_env_params *ep, const orc_profile *p, const orc_state *st, int32_t action,
                         double gauss, orc_state *next, double *reward, int32_t *done) {
    int act = action < 0 ? 0 : (action > 7 ? 7 : action); /* Action.fromId env.scala:18-20 */
    double beta = ep->base_beta * p->beta_mul;
    double gamma = ep->base_gamma;
    double ifr = ep->base_ifr * p->ifr_mul;
    double cap = st->hosp_cap;
    if (act == 1) beta *= p->econ_resilience * 0.75 + 0.5;
    else if (act == 2) beta *= (1.0 - p->connectivity) * 0.4 + 0.4;
    else if (act == 3) beta *= p->econ_resilience * 0.5 + 0.2;
    else if (act == 5) { cap *= 1.0 + p->healthcare_capacity; gamma *= 1.3; }

    int32_t s0 = imax(0, jround_i(st->s));
    int32_t i0 = imax(0, jround_i(st->i));
    int32_t r0 = imax(0, jround_i(st->r));
    int32_t d0 = imax(0, jround_i(st->d));
    int32_t v0 = imax(0, jround_i(st->v));
    int32_t pop0 = iadd(iadd(iadd(iadd(s0, i0), r0), d0), v0);
    double mix = fmax(1.0, (double)iadd(iadd(s0, i0), r0));

    double countryNoise = ep->noise * (0.2 + p->noise * 1.5);
    double stoch = 1.0 + countryNoise * gauss;

    double effBeta = clamp01(beta * stoch);
    double effGamma = clamp01(gamma * stoch);
    double effIFR = clamp01((((double)i0 > cap) ? ifr * 6.0 : ifr) * stoch);

    double vaxRate = act == 6 ? 0.02 * p->healthcare_capacity : (act == 7 ? 0.05 * p->healthcare_capacity : 0.0);

    double pInf = clamp01(effBeta * (mix > 0 ? (double)i0 / mix : 0.0));
    double pVax = clamp01(vaxRate);
    double wS[3] = {fmax(0.0, 1.0 - pInf - pVax), pInf, pVax};
    int32_t sA[3];
    orc_alloc_int(s0, wS, 3, sA);
    int32_t sInf = sA[1], sVax = sA[2];

    double pRec = clamp01(effGamma);
    double pDie = clamp01(effIFR);
    double wI[3] = {fmax(0.0, 1.0 - pRec - pDie), pRec, pDie};
    int32_t iA[3];
    orc_alloc_int(i0, wI, 3, iA);
    int32_t iStay = iA[0], rec = iA[1], die = iA[2];

    int32_t sN = isub(isub(s0, sInf), sVax);
    int32_t iN = iadd(iStay, sInf);
    int32_t rN = iadd(r0, rec);
    int32_t dN = iadd(d0, die);
    int32_t vN = iadd(v0, sVax);
    int32_t sumN = iadd(iadd(iadd(iadd(sN, iN), rN), dN), vN);
    int32_t resid = isub(pop0, sumN);
    if (resid != 0) sN = imax(0, iadd(sN, resid));

    *next = *st;
    next->s = (double)sN; next->i = (double)iN; next->r = (double)rN;
    next->d = (double)dN; next->v = (double)vN;
    next->hosp_cap = cap;
    next->t = iadd(st->t, 1);

    *reward = reward_fn(ep, p, st, next, act);
    double nn = fmax(1.0, state_n(next));
    *done = (next->t >= next->t_max) || (next->i / nn < 1e-5) || (next->d / nn > 0.02);
}

/* ------------------------------------------------------------------------ */
/* ToyConn.scaleRow -- ToyConn.scala:58-70 (in place on a copy made by caller) */
/* ------------------------------------------------------------------------ */
EXPORT void orc_scale_row(double *A, int32_t n, int32_t row, double factor) {
    double s = 0.0;
    double *a = A + (size_t)row * n;
    for (int j = 0; j < n; j++) if (j != row) { a[j] *= factor; s += a[j]; }
    double cap = fmin(0.2, s);
    if (s > cap && s > 0) {
        double sc = cap / s;
        for (int j = 0; j < n; j++) if (j != row) a[j] *= sc;
    }
    a[row] = 0.0;
}

/* ------------------------------------------------------------------------ */
/* Validation -- Validation.scala:6-22                                       */
/* ------------------------------------------------------------------------ */
EXPORT int32_t orc_arrival_time(const double *series, int32_t n, double thresh) {
    for (int k = 0; k < n; k++) if (series[k] >= thresh) return k;
    return n;
}
EXPORT double orc_peak_i(const double *series, int32_t n) {
    if (n == 0) return 0.0;
    double m = series[0];
    for (int k = 1; k < n; k++) m = (series[k] > m) ? series[k] : m; /* Seq.max */
    return m;
}
/* returns 0 ok, 1 neg compartment, 2 mass, 3 t bounds */
EXPORT int32_t orc_assert_state(const orc_state *s, int32_t pop, int32_t t_max) {
    if (!(s->s >= 0 && s->i >= 0 && s->r >= 0 && s->d >= 0 && s->v >= 0)) return 1;
    double total = s->s + s->i + s->r + s->d + s->v;
    if (jround_i(total) != pop) return 2;
    if (!(s->t >= 0 && s->t <= t_max)) return 3;
    return 0;
}

/* ------------------------------------------------------------------------ */
/* MLP init + predict -- MLP.scala:4-22                                      */
/* ------------------------------------------------------------------------ */
typedef struct {
    int in, h, out;
    double *W1, *b1, *W2, *b2, *W3, *b3;
} mlp;

static void mlp_init(mlp *m, int in, int h, int out) {
    m->in = in; m->h = h; m->out = out;
    m->W1 = (double *)malloc(sizeof(double) * (size_t)h * in);
    m->W2 = (double *)malloc(sizeof(double) * (size_t)h * h);
    m->W3 = (double *)malloc(sizeof(double) * (size_t)out * h);
    m->b1 = (double *)calloc((size_t)h, sizeof(double));
    m->b2 = (double *)calloc((size_t)h, sizeof(double));
    m->b3 = (double *)calloc((size_t)out, sizeof(double));
    jrand r; jr_init(&r, 7); /* MLP.scala:5 */
    for (int k = 0; k < h * in; k++) m->W1[k] = jr_next_gaussian(&r) * 0.05;
    for (int k = 0; k < h * h; k++) m->W2[k] = jr_next_gaussian(&r) * 0.05;
    for (int k = 0; k < out * h; k++) m->W3[k] = jr_next_gaussian(&r) * 0.05;
}
static void mlp_free(mlp *m) { free(m->W1); free(m->W2); free(m->W3); free(m->b1); free(m->b2); free(m->b3); }

/* matVec: row.view.zip(x).map(w*xi).sum  => left-to-right sum of products   */
static void matvec(const double *W, const double *x, int rows, int cols, double *y) {
    for (int r = 0; r < rows; r++) {
        double acc = 0.0;
        for (int c = 0; c < cols; c++) {
            double pr = W[(size_t)r * cols + c] * x[c];
            acc = (c == 0) ? pr : acc + pr;
        }
        y[r] = acc;
    }
}
static void mlp_predict(const mlp *m, const double *x, double clamp, double *y) {
    double *h1 = (double *)malloc(sizeof(double) * (size_t)m->h);
    double *h2 = (double *)malloc(sizeof(double) * (size_t)m->h);
    matvec(m->W1, x, m->h, m->in, h1);
    for (int k = 0; k < m->h; k++) { double z = h1[k] + m->b1[k]; h1[k] = z > 0 ? z : 0.0; }
    matvec(m->W2, h1, m->h, m->h, h2);
    for (int k = 0; k < m->h; k++) { double z = h2[k] + m->b2[k]; h2[k] = z > 0 ? z : 0.0; }
    matvec(m->W3, h2, m->out, m->h, y);
    for (int k = 0; k < m->out; k++) {
        y[k] = y[k] + m->b3[k];
        if (isfinite(clamp)) y[k] = fmax(-clamp, fmin(clamp, y[k]));
    }
    free(h1); free(h2);
}
/* zipWithIndex.maxBy(_._1)._2 : first index of the maximum */
static int argmax_first(const double *y, int n) {
    int best = 0;
    for (int k = 1; k < n; k++) if (y[k] > y[best]) best = k;
    return best;
}

EXPORT void orc_mlp_weights(int32_t in, int32_t h, int32_t out, double *W1, double *W2, double *W3) {
    mlp m; mlp_init(&m, in, h, out);
    memcpy(W1, m.W1, sizeof(double) * (size_t)h * in);
    memcpy(W2, m.W2, sizeof(double) * (size_t)h * h);
    memcpy(W3, m.W3, sizeof(double) * (size_t)out * h);
    mlp_free(&m);
}
EXPORT void orc_mlp_predict(int32_t in, int32_t h, int32_t out, const double *x, double clamp, double *y) {
    mlp m; mlp_init(&m, in, h, out);
    mlp_predict(&m, x, clamp, y);
    mlp_free(&m);
}

/* ------------------------------------------------------------------------ */
/* HyperParams fields the path reads -- config.scala:3-21                    */
/* ------------------------------------------------------------------------ */
typedef struct {
    double epsilon_start, epsilon_end;
    int32_t epsilon_decay_steps, hidden;
    double clip_q, reward_clip;
    int32_t batch_size, learn_every;
} orc_hyper;

/* DDQNAgent: action choice only -- DDQNAgent.scala:12-32.  All agents built
 * with the same HyperParams share identical MLP weights (MLP.scala:5 fixed
 * seed), so the weights are held once per world.                             */
typedef struct {
    int64_t steps;
    jrand rnd;
    int32_t replay_size;
} agent;

EXPORT void orc_learner_destroy(void *h);

static double agent_epsilon(const agent *a, const orc_hyper *hp) { /* DDQNAgent.scala:16-19 */
    int32_t den = hp->epsilon_decay_steps > 1 ? hp->epsilon_decay_steps : 1;
    double frac = fmin(1.0, (double)a->steps / (double)den);
    return hp->epsilon_start + (hp->epsilon_end - hp->epsilon_start) * frac;
}

/* ------------------------------------------------------------------------ */
/* WorldToy -- WorldToy.scala:11-235, generalised from ToyConn.C to N input   */
/* countries (names replaced by per-country profile/env inputs).             */
/* ------------------------------------------------------------------------ */
typedef struct {
    int32_t n, seed_idx, validate;
    int32_t *pop;
    double *conn; /* n*n row-major, or NULL when the graph is given as CSR below */
    int32_t *rowptr, *col; /* CSR form of the same matrix: entries that are not stored are weight 0 */
    double *val;
    orc_env_params *env;
    orc_profile *prof;
    jrand *env_rnd;
    agent *agents;
    orc_hyper hp;
    mlp net;
    orc_state *init, *cur;
    int64_t g_step;
    int32_t *last_a;
    int32_t err;      /* 0 ok; 1 local mass; 2 world mass; 3 learning would trigger; 4 assertState */
    int32_t err_node, err_t;
    /* per-node learners (WorldToy.scala:46-65,147-167), present after orc_world_enable_learning */
    void **learners;
    int32_t *learn_every, *batch_size, *intensity;
    int64_t *last_learn;
} world;

static int64_t sum_state(const orc_state *s) { /* WorldToy.scala:79-80 */
    return jround(s->s + s->i + s->r + s->d + s->v);
}

EXPORT void *orc_world_create(int32_t n, const int32_t *pop, const double *conn, int32_t seed_idx,
                              const orc_state *base, const orc_env_params *env, const orc_profile *prof,
                              const int32_t *env_seeds, const orc_hyper *hp, int32_t validate) {
    world *w = (world *)calloc(1, sizeof(world));
    w->n = n; w->seed_idx = seed_idx; w->validate = validate; w->hp = *hp;
    w->pop = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    memcpy(w->pop, pop, sizeof(int32_t) * (size_t)n);
    if (conn) {
        w->conn = (double *)malloc(sizeof(double) * (size_t)n * n);
        memcpy(w->conn, conn, sizeof(double) * (size_t)n * n);
    }
    w->env = (orc_env_params *)malloc(sizeof(orc_env_params) * (size_t)n);
    w->prof = (orc_profile *)malloc(sizeof(orc_profile) * (size_t)n);
    memcpy(w->env, env, sizeof(orc_env_params) * (size_t)n);
    memcpy(w->prof, prof, sizeof(orc_profile) * (size_t)n);
    w->env_rnd = (jrand *)malloc(sizeof(jrand) * (size_t)n);
    w->agents = (agent *)malloc(sizeof(agent) * (size_t)n);
    w->init = (orc_state *)malloc(sizeof(orc_state) * (size_t)n);
    w->cur = (orc_state *)malloc(sizeof(orc_state) * (size_t)n);
    w->last_a = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    mlp_init(&w->net, 7, hp->hidden, 8); /* WorldToy.scala:50 */
    for (int c = 0; c < n; c++) {
        jr_init(&w->env_rnd[c], env_seeds[c]);     /* env.scala:42 */
        w->agents[c].steps = 0;
        jr_init(&w->agents[c].rnd, 9);             /* DDQNAgent.scala:14 */
        w->agents[c].replay_size = 0;
        /* initStates -- WorldToy.scala:28-32 */
        orc_state s = *base;
        double hc = base->hosp_cap * ((double)pop[c] / 1000000.0);
        s.hosp_cap = hc > 1.0 ? hc : 1.0; /* (x).max(1.0) */
        if (c == seed_idx) {
            int32_t i0 = d2i(0.001 * (double)pop[c]);
            s.i = (double)i0;
            s.s = (double)isub(pop[c], i0);
        } else {
            s.i = 0.0;
            s.s = (double)pop[c];
        }
        w->init[c] = s; w->cur[c] = s;
        if (validate && orc_assert_state(&s, pop[c], base->t_max) != 0 && w->err == 0) {
            w->err = 4; w->err_node = c; w->err_t = s.t;
        }
    }
    return w;
}

/* The same world over a CSR graph (columns ascending inside a row): the mobility loop then walks the stored
 * entries only.  Entries that are not stored have weight 0: they add +0.0 to the row sum and have raw = 0,
 * base = 0, fraction 0 in allocateInt, exactly what the dense walk computes for them.                       */
EXPORT void *orc_world_create_csr(int32_t n, const int32_t *pop, const int32_t *rowptr, const int32_t *col, const double *val,
                                  int32_t seed_idx, const orc_state *base, const orc_env_params *env, const orc_profile *prof,
                                  const int32_t *env_seeds, const orc_hyper *hp, int32_t validate) {
    world *w = (world *)orc_world_create(n, pop, NULL, seed_idx, base, env, prof, env_seeds, hp, validate);
    size_t nnz = (size_t)rowptr[n];
    w->rowptr = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    w->col = (int32_t *)malloc(sizeof(int32_t) * (nnz ? nnz : 1));
    w->val = (double *)malloc(sizeof(double) * (nnz ? nnz : 1));
    memcpy(w->rowptr, rowptr, sizeof(int32_t) * ((size_t)n + 1));
    memcpy(w->col, col, sizeof(int32_t) * nnz);
    memcpy(w->val, val, sizeof(double) * nnz);
    return w;
}

EXPORT void orc_world_destroy(void *h) {
    world *w = (world *)h;
    free(w->rowptr); free(w->col); free(w->val);
    free(w->pop); free(w->conn); free(w->env); free(w->prof); free(w->env_rnd); free(w->agents);
    free(w->init); free(w->cur); free(w->last_a); mlp_free(&w->net);
    if (w->learners) {
        for (int c = 0; c < w->n; c++) orc_learner_destroy(w->learners[c]);
        free(w->learners); free(w->learn_every); free(w->batch_size); free(w->intensity); free(w->last_learn);
    }
    free(w);
}
EXPORT void orc_world_get_states(void *h, orc_state *out) {
    world *w = (world *)h; memcpy(out, w->cur, sizeof(orc_state) * (size_t)w->n);
}
EXPORT void orc_world_set_states(void *h, const orc_state *in) { /* restoreStates :76 */
    world *w = (world *)h; memcpy(w->cur, in, sizeof(orc_state) * (size_t)w->n);
}
EXPORT void orc_world_reset(void *h) { /* resetToInitial :77 */
    world *w = (world *)h; memcpy(w->cur, w->init, sizeof(orc_state) * (size_t)w->n);
}
EXPORT int32_t orc_world_error(void *h, int32_t *node, int32_t *t) {
    world *w = (world *)h; if (node) *node = w->err_node; if (t) *t = w->err_t; return w->err;
}
EXPORT void orc_world_last_actions(void *h, int32_t *out) {
    world *w = (world *)h; memcpy(out, w->last_a, sizeof(int32_t) * (size_t)w->n);
}

/* mobility + apply + world check: WorldToy.scala:103-145 (shared by stepOne and
 * stepOneGreedy :191-233).  local[] holds the post-env.step states.           */
static void mobility_apply(world *w, const orc_state *local, int32_t t_max, int32_t *moved_s_out,
                           int32_t *moved_i_out) {
    int n = w->n;
    int64_t total_after_local = 0;
    if (w->validate) for (int c = 0; c < n; c++) total_after_local += sum_state(&local[c]);
    int32_t *movedS = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    int32_t *movedI = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    double *wRow = (double *)malloc(sizeof(double) * (size_t)n);
    int32_t *aS = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *aI = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    for (int i = 0; i < n && w->rowptr; i++) {   /* CSR graph: the stored entries of row i, the origin's own entry as 0 (:113) */
        int32_t Si = imax(0, jround_i(local[i].s));
        int32_t Ii = imax(0, jround_i(local[i].i));
        int rs = w->rowptr[i], deg = w->rowptr[i + 1] - rs;
        double sum = 0.0;
        for (int k = 0; k < deg; k++) {
            wRow[k] = (w->col[rs + k] == i) ? 0.0 : w->val[rs + k];
            sum = (k == 0) ? wRow[0] : sum + wRow[k];
        }
        double sumW = fmax(0.0, sum);
        double eS = floor((double)Si * sumW + 1e-9);
        double eI = floor((double)Ii * sumW + 1e-9);
        int64_t lS = d2l(eS), lI = d2l(eI);
        int32_t exportS = (int32_t)(uint32_t)(uint64_t)((int64_t)Si < lS ? (int64_t)Si : lS);
        int32_t exportI = (int32_t)(uint32_t)(uint64_t)((int64_t)Ii < lI ? (int64_t)Ii : lI);
        int32_t extra[4], nxS = 0, nxI = 0;
        orc_alloc_int_sparse(exportS, wRow, w->col + rs, deg, n, aS, extra, 4, &nxS);
        orc_alloc_int_sparse(exportI, wRow, w->col + rs, deg, n, aI, extra, 4, &nxI);
        /* units left over for columns that are not stored: the dense walk hands them out, no sparse layout does */
        if ((nxS | nxI) != 0 && w->err == 0) { w->err = 5; w->err_node = i; w->err_t = local[i].t; }
        for (int k = 0; k < deg; k++) {
            int j = w->col[rs + k];
            if (j != i) { movedS[j] = iadd(movedS[j], aS[k]); movedI[j] = iadd(movedI[j], aI[k]); }
        }
        movedS[i] = isub(movedS[i], exportS);
        movedI[i] = isub(movedI[i], exportI);
    }
    for (int i = 0; i < n && !w->rowptr; i++) {
        int32_t Si = imax(0, jround_i(local[i].s));
        int32_t Ii = imax(0, jround_i(local[i].i));
        double sum = 0.0;
        for (int j = 0; j < n; j++) {
            wRow[j] = (j == i) ? 0.0 : w->conn[(size_t)i * n + j];
            sum = (j == 0) ? wRow[0] : sum + wRow[j];
        }
        double sumW = fmax(0.0, sum);
        double eS = floor((double)Si * sumW + 1e-9);
        double eI = floor((double)Ii * sumW + 1e-9);
        int64_t lS = d2l(eS), lI = d2l(eI);
        int32_t exportS = (int32_t)(uint32_t)(uint64_t)((int64_t)Si < lS ? (int64_t)Si : lS);
        int32_t exportI = (int32_t)(uint32_t)(uint64_t)((int64_t)Ii < lI ? (int64_t)Ii : lI);
        orc_alloc_int(exportS, wRow, n, aS);
        orc_alloc_int(exportI, wRow, n, aI);
        for (int j = 0; j < n; j++) if (j != i) { movedS[j] = iadd(movedS[j], aS[j]); movedI[j] = iadd(movedI[j], aI[j]); }
        movedS[i] = isub(movedS[i], exportS);
        movedI[i] = isub(movedI[i], exportI);
    }
    for (int c = 0; c < n; c++) {
        orc_state s3 = local[c];
        s3.s = fmax(0.0, local[c].s + (double)movedS[c]);
        s3.i = fmax(0.0, local[c].i + (double)movedI[c]);
        int32_t t1 = iadd(local[c].t, 1);
        s3.t = t1 < t_max ? t1 : t_max;
        w->cur[c] = s3;
    }
    if (w->validate) {
        int64_t total_after_move = 0;
        for (int c = 0; c < n; c++) total_after_move += sum_state(&w->cur[c]);
        if (total_after_move != total_after_local && w->err == 0) { w->err = 2; w->err_node = -1; w->err_t = local[0].t; }
    }
    if (moved_s_out) memcpy(moved_s_out, movedS, sizeof(int32_t) * (size_t)n);
    if (moved_i_out) memcpy(moved_i_out, movedI, sizeof(int32_t) * (size_t)n);
    free(movedS); free(movedI); free(wRow); free(aS); free(aI);
}

/* One world step with caller-supplied actions and Gaussians (the arithmetic of
 * WorldToy.stepOne :83-145 with agent.act and rnd.nextGaussian factored out). */
EXPORT void orc_world_step_given(void *h, const int32_t *actions, const double *gauss, int32_t t_max,
                                 double *rewards, int32_t *moved_s, int32_t *moved_i) {
    world *w = (world *)h;
    int n = w->n;
    orc_state *local = (orc_state *)malloc(sizeof(orc_state) * (size_t)n);
    for (int c = 0; c < n; c++) {
        int64_t tb = sum_state(&w->cur[c]);
        int32_t done; double r;
        orc_env_step(&w->env[c], &w->prof[c], &w->cur[c], actions[c], gauss[c], &local[c], &r, &done);
        w->last_a[c] = actions[c];
        if (rewards) rewards[c] = r;
        if (w->validate) {
            int64_t ta = sum_state(&local[c]);
            if (ta != tb && w->err == 0) { w->err = 1; w->err_node = c; w->err_t = w->cur[c].t; }
        }
    }
    mobility_apply(w, local, t_max, moved_s, moved_i);
    free(local);
}

/* WorldToy.stepOne (greedy=0, :83-170) / stepOneGreedy (greedy=1, :172-235). */
EXPORT void orc_world_step(void *h, int32_t t_max, int32_t greedy, double *rewards) {
    world *w = (world *)h;
    int n = w->n;
    orc_state *local = (orc_state *)malloc(sizeof(orc_state) * (size_t)n);
    double f7[7], q[8];
    for (int c = 0; c < n; c++) {
        agent *ag = &w->agents[c];
        int a;
        if (greedy) {
            orc_features(&w->cur[c], f7);
            mlp_predict(&w->net, f7, w->hp.clip_q, q);
            a = argmax_first(q, 8);
        } else {
            /* act(s) = actWithEpsilon(s, epsilon, countStep = true): epsilon is an argument, evaluated
             * with the step count before the call increments it (DDQNAgent.scala:26-27,32) */
            double eps = agent_epsilon(ag, &w->hp);
            ag->steps += 1;                               /* countStep */
            if (jr_next_double(&ag->rnd) < eps) a = jr_next_int_bound(&ag->rnd, 8);
            else {
                orc_features(&w->cur[c], f7);
                mlp_predict(&w->net, f7, w->hp.clip_q, q);
                a = argmax_first(q, 8);
            }
        }
        int64_t tb = sum_state(&w->cur[c]);
        int32_t done; double r;
        double g = jr_next_gaussian(&w->env_rnd[c]);
        orc_env_step(&w->env[c], &w->prof[c], &w->cur[c], a, g, &local[c], &r, &done);
        w->last_a[c] = a;
        if (rewards) rewards[c] = r;
        if (w->validate) {
            int64_t ta = sum_state(&local[c]);
            if (ta != tb && w->err == 0) { w->err = 1; w->err_node = c; w->err_t = w->cur[c].t; }
        }
        if (!greedy) ag->replay_size += 1;                /* observe :99 */
    }
    mobility_apply(w, local, t_max, NULL, NULL);
    free(local);
    if (!greedy) {
        w->g_step += 1;
        /* learning trigger :150-167.  The DDQN update is outside this oracle; flag
         * the step on which the reference would have started to learn.        */
        for (int c = 0; c < n; c++) {
            if (w->g_step >= w->hp.learn_every && w->agents[c].replay_size >= w->hp.batch_size * 4 && w->err == 0) {
                w->err = 3; w->err_node = c; w->err_t = (int32_t)w->g_step;
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* Calibrate -- Calibrate.scala:14-116                                       */
/* ------------------------------------------------------------------------ */
typedef struct {
    double gamma, lr, clip_q, grad_clip, ema_alpha;
    int32_t hidden, replay_capacity, target_update_every, max_need;
} orc_learn_hyper;
EXPORT void orc_world_enable_learning(void *h, const orc_learn_hyper *hp, const int32_t *learn_every,
                                      const int32_t *batch_size, const int32_t *intensity);
EXPORT void orc_world_step_learn(void *h, const orc_hyper *hps, int32_t t_max, double *rewards, double *losses);

static double rmse(const double *x, int n) { /* :14-17 */
    double s = 0.0;
    for (int k = 0; k < n; k++) { double q = x[k] * x[k]; s = (k == 0) ? q : s + q; }
    double nn = (double)n;
    return sqrt(s / fmax(1.0, nn));
}

/* Calibrate.rollout :19-82 for one (beta, travelScale) over a dense matrix W or, with W == NULL, a CSR graph.
 * Returns arrival[n], peak[n], final[n]; *err = world error code.            */
static void rollout_any(int32_t n, const int32_t *pop, const double *W, const int32_t *rowptr, const int32_t *col,
                        const double *val, int32_t seed_idx, const orc_state *base,
                        const orc_profile *prof, const orc_hyper *hp, int32_t steps, double beta, double ts,
                        int32_t heterogeneous, int32_t *arrival, double *peak, double *final_i, double *series_out,
                        int32_t *err, const orc_learn_hyper *lhp, const int32_t *intensity) {
    double *Ws = NULL, *vs = NULL;
    if (W) {
        Ws = (double *)malloc(sizeof(double) * (size_t)n * n);
        memcpy(Ws, W, sizeof(double) * (size_t)n * n);
        orc_scale_row(Ws, n, seed_idx, ts); /* :29 */
    } else {
        /* ToyConn.scaleRow on the stored entries of the seed row: the zeros of the other columns stay zeros and add
         * +0.0 to the row sum (ToyConn.scala:58-70) */
        size_t nnz = (size_t)rowptr[n];
        vs = (double *)malloc(sizeof(double) * (nnz ? nnz : 1));
        memcpy(vs, val, sizeof(double) * nnz);
        int rs = rowptr[seed_idx], re = rowptr[seed_idx + 1];
        double s = 0.0;
        for (int k = rs; k < re; k++) if (col[k] != seed_idx) { vs[k] *= ts; s += vs[k]; }
        double cap = fmin(0.2, s);
        if (s > cap && s > 0) { double sc = cap / s; for (int k = rs; k < re; k++) if (col[k] != seed_idx) vs[k] *= sc; }
        for (int k = rs; k < re; k++) if (col[k] == seed_idx) vs[k] = 0.0;
    }
    orc_env_params *env = (orc_env_params *)malloc(sizeof(orc_env_params) * (size_t)n);
    orc_profile *pr = (orc_profile *)malloc(sizeof(orc_profile) * (size_t)n);
    int32_t *seeds = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    static const orc_profile dflt = {1.0, 1.0, 0.05, 100.0, 30000.0, 0.7, 0.15, 0.8, 0.7}; /* Geo.scala:4-21 */
    for (int c = 0; c < n; c++) {
        if (heterogeneous) { /* :34-51 */
            const orc_profile *p = &prof[c];
            env[c].base_beta = beta * (1.0 + (p->population_density / 1000.0) * 0.5) * (3.0 - p->healthcare_capacity * 2.0);
            env[c].base_gamma = 0.12;
            env[c].base_ifr = 0.008 * (1.0 + p->age_elderly * 2.0);
            env[c].noise = 0.05 * (1.0 + (1.0 - p->econ_resilience));
            env[c].reward_clip = hp->reward_clip;
            env[c].risk_factor = 0.0;
            pr[c] = *p;
            seeds[c] = c + 42;
        } else { /* :31-32, countryName defaults to "Unknown" => default profile */
            env[c].base_beta = beta; env[c].base_gamma = 0.12; env[c].base_ifr = 0.008; env[c].noise = 0.05;
            env[c].reward_clip = hp->reward_clip; env[c].risk_factor = 0.0;
            pr[c] = dflt;
            seeds[c] = 7;
        }
    }
    world *w = W ? (world *)orc_world_create(n, pop, Ws, seed_idx, base, env, pr, seeds, hp, 1 /* validate = true :64 */)
                 : (world *)orc_world_create_csr(n, pop, rowptr, col, vs, seed_idx, base, env, pr, seeds, hp, 1);
    double *series = (double *)malloc(sizeof(double) * (size_t)n * (size_t)(steps > 0 ? steps : 1));
    orc_hyper *hps = NULL;
    if (lhp) {
        /* the agents of the rollout's world learn as WorldToy.stepOne has them (WorldToy.scala:147-167): every node gets
         * the sweep's HyperParams (Calibrate.scala:56-57 passes hp and no countryHyperParams) */
        orc_learn_hyper *lh = (orc_learn_hyper *)malloc(sizeof(orc_learn_hyper) * (size_t)n);
        int32_t *le = (int32_t *)malloc(4 * (size_t)n), *bs = (int32_t *)malloc(4 * (size_t)n), *it = (int32_t *)malloc(4 * (size_t)n);
        hps = (orc_hyper *)malloc(sizeof(orc_hyper) * (size_t)n);
        for (int c = 0; c < n; c++) {
            lh[c] = *lhp; le[c] = hp->learn_every; bs[c] = hp->batch_size; it[c] = intensity ? intensity[c] : 1; hps[c] = *hp;
        }
        orc_world_enable_learning(w, lh, le, bs, it);
        free(lh); free(le); free(bs); free(it);
    }
    for (int t = 0; t < steps; t++) { /* :68-73 */
        if (lhp) orc_world_step_learn(w, hps, steps, NULL, NULL);
        else orc_world_step(w, steps, 0, NULL);
        for (int c = 0; c < n; c++) series[(size_t)c * steps + t] = w->cur[c].i;
    }
    free(hps);
    for (int c = 0; c < n; c++) { /* :75-81 */
        const double *is = series + (size_t)c * steps;
        arrival[c] = orc_arrival_time(is, steps, 10.0);
        peak[c] = orc_peak_i(is, steps);
        final_i[c] = steps > 0 ? is[steps - 1] : 0.0;
    }
    if (series_out) memcpy(series_out, series, sizeof(double) * (size_t)n * (size_t)steps);
    *err = w->err;
    orc_world_destroy(w);
    free(series); free(Ws); free(vs); free(env); free(pr); free(seeds);
}

/* Calibrate.rollout :19-82 for one (beta, travelScale).
 * Returns arrival[n], peak[n], final[n]; *err = world error code.            */
EXPORT void orc_rollout(int32_t n, const int32_t *pop, const double *W, int32_t seed_idx, const orc_state *base,
                        const orc_profile *prof, const orc_hyper *hp, int32_t steps, double beta, double ts,
                        int32_t heterogeneous, int32_t *arrival, double *peak, double *final_i, double *series_out,
                        int32_t *err) {
    rollout_any(n, pop, W, NULL, NULL, NULL, seed_idx, base, prof, hp, steps, beta, ts, heterogeneous, arrival, peak, final_i,
                series_out, err, NULL, NULL);
}

/* Calibrate.gridSurface :84-116.  has_arr/has_pk are the Map key-presence masks.
 * Outputs: loss[nb*nts] beta-major; optional arrival/peak [G][n]; best index by
 * strict < (first minimum).  Returns the first nonzero world error code.      */
static int32_t grid_surface_any(int32_t n, const int32_t *pop, const double *W, const int32_t *rowptr, const int32_t *col,
                                const double *val, int32_t seed_idx,
                                const orc_state *base, const orc_profile *prof, const orc_hyper *hp,
                                int32_t steps, const double *betas, int32_t nb, const double *tss, int32_t nts,
                                const double *t_arrive, const uint8_t *has_arr, const double *t_peak,
                                const uint8_t *has_pk, int32_t heterogeneous, double *loss, int32_t *arrival_out,
                                double *peak_out, int32_t *best_idx, double *best_loss, const orc_learn_hyper *lhp,
                                const int32_t *intensity) {
    int32_t *arr = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    double *pk = (double *)malloc(sizeof(double) * (size_t)n);
    double *fin = (double *)malloc(sizeof(double) * (size_t)n);
    double *dv = (double *)malloc(sizeof(double) * (size_t)n);
    double bl = 1.7976931348623157e308; /* Double.MaxValue */
    int32_t bi = 0, first_err = 0;
    for (int b = 0; b < nb; b++)
        for (int k = 0; k < nts; k++) {
            int g = b * nts + k, err = 0;
            rollout_any(n, pop, W, rowptr, col, val, seed_idx, base, prof, hp, steps, betas[b], tss[k], heterogeneous, arr, pk, fin, NULL, &err,
                        lhp, intensity);
            if (err && !first_err) first_err = err;
            for (int c = 0; c < n; c++) {
                double a = (double)arr[c];
                dv[c] = a - (has_arr[c] ? t_arrive[c] : a);
            }
            double arrLoss = rmse(dv, n);
            for (int c = 0; c < n; c++) dv[c] = pk[c] - (has_pk[c] ? t_peak[c] : pk[c]);
            double pkLoss = rmse(dv, n);
            double l = 0.5 * arrLoss + 0.5 * pkLoss;
            loss[g] = l;
            if (arrival_out) memcpy(arrival_out + (size_t)g * n, arr, sizeof(int32_t) * (size_t)n);
            if (peak_out) memcpy(peak_out + (size_t)g * n, pk, sizeof(double) * (size_t)n);
            if (l < bl) { bl = l; bi = g; }
        }
    *best_idx = bi; *best_loss = bl;
    free(arr); free(pk); free(fin); free(dv);
    return first_err;
}

EXPORT int32_t orc_grid_surface(int32_t n, const int32_t *pop, const double *W, int32_t seed_idx,
                                const orc_state *base, const orc_profile *prof, const orc_hyper *hp,
                                int32_t steps, const double *betas, int32_t nb, const double *tss, int32_t nts,
                                const double *t_arrive, const uint8_t *has_arr, const double *t_peak,
                                const uint8_t *has_pk, int32_t heterogeneous, double *loss, int32_t *arrival_out,
                                double *peak_out, int32_t *best_idx, double *best_loss) {
    return grid_surface_any(n, pop, W, NULL, NULL, NULL, seed_idx, base, prof, hp, steps, betas, nb, tss, nts, t_arrive, has_arr,
                            t_peak, has_pk, heterogeneous, loss, arrival_out, peak_out, best_idx, best_loss, NULL, NULL);
}

/* Calibrate.gridSurface with the rollouts' agents learning (WorldToy.stepOne as written, :147-167): what the reference
 * does once a rollout is long enough for the replay buffers to hold 4 * batchSize transitions.  lhp = the
 * HyperParams fields DDQNAgent.learn reads; intensity[c] = gradAccumSteps of country c's learn call (:157-162). */
EXPORT int32_t orc_grid_surface_learn(int32_t n, const int32_t *pop, const double *W, int32_t seed_idx,
                                      const orc_state *base, const orc_profile *prof, const orc_hyper *hp,
                                      int32_t steps, const double *betas, int32_t nb, const double *tss, int32_t nts,
                                      const double *t_arrive, const uint8_t *has_arr, const double *t_peak,
                                      const uint8_t *has_pk, int32_t heterogeneous, double *loss, int32_t *arrival_out,
                                      double *peak_out, int32_t *best_idx, double *best_loss, const orc_learn_hyper *lhp,
                                      const int32_t *intensity) {
    return grid_surface_any(n, pop, W, NULL, NULL, NULL, seed_idx, base, prof, hp, steps, betas, nb, tss, nts, t_arrive, has_arr,
                            t_peak, has_pk, heterogeneous, loss, arrival_out, peak_out, best_idx, best_loss, lhp, intensity);
}

/* The same over a CSR graph: O(stored entries) per world step instead of O(n^2). */
EXPORT int32_t orc_grid_surface_csr(int32_t n, const int32_t *pop, const int32_t *rowptr, const int32_t *col, const double *val,
                                    int32_t seed_idx, const orc_state *base, const orc_profile *prof, const orc_hyper *hp,
                                    int32_t steps, const double *betas, int32_t nb, const double *tss, int32_t nts,
                                    const double *t_arrive, const uint8_t *has_arr, const double *t_peak,
                                    const uint8_t *has_pk, int32_t heterogeneous, double *loss, int32_t *arrival_out,
                                    double *peak_out, int32_t *best_idx, double *best_loss) {
    return grid_surface_any(n, pop, NULL, rowptr, col, val, seed_idx, base, prof, hp, steps, betas, nb, tss, nts, t_arrive, has_arr,
                            t_peak, has_pk, heterogeneous, loss, arrival_out, peak_out, best_idx, best_loss, NULL, NULL);
}

/* Timed leg for bench.py's cpu_baseline: runs `n_points` grid points starting at
 * flat index g0 (beta-major) and returns the country-steps executed.          */
EXPORT int64_t orc_bench_points(int32_t n, const int32_t *pop, const double *W, int32_t seed_idx,
                                const orc_state *base, const orc_profile *prof, const orc_hyper *hp, int32_t steps,
                                const double *betas, int32_t nb, const double *tss, int32_t nts, int32_t heterogeneous,
                                int32_t g0, int32_t n_points, double *checksum) {
    int32_t *arr = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    double *pk = (double *)malloc(sizeof(double) * (size_t)n);
    double *fin = (double *)malloc(sizeof(double) * (size_t)n);
    double cs = 0.0; int64_t done = 0;
    (void)nb;
    for (int g = g0; g < g0 + n_points; g++) {
        int err = 0;
        orc_rollout(n, pop, W, seed_idx, base, prof, hp, steps, betas[g / nts], tss[g % nts], heterogeneous, arr, pk, fin, NULL, &err);
        for (int c = 0; c < n; c++) cs += pk[c] + (double)arr[c];
        done += (int64_t)n * steps;
    }
    *checksum = cs;
    free(arr); free(pk); free(fin);
    return done;
}

/* The same timed leg over a CSR graph. */
EXPORT int64_t orc_bench_points_csr(int32_t n, const int32_t *pop, const int32_t *rowptr, const int32_t *col, const double *val,
                                    int32_t seed_idx, const orc_state *base, const orc_profile *prof, const orc_hyper *hp,
                                    int32_t steps, const double *betas, int32_t nb, const double *tss, int32_t nts,
                                    int32_t heterogeneous, int32_t g0, int32_t n_points, double *checksum) {
    int32_t *arr = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    double *pk = (double *)malloc(sizeof(double) * (size_t)n);
    double *fin = (double *)malloc(sizeof(double) * (size_t)n);
    double cs = 0.0; int64_t done = 0;
    (void)nb;
    for (int g = g0; g < g0 + n_points; g++) {
        int err = 0;
        rollout_any(n, pop, NULL, rowptr, col, val, seed_idx, base, prof, hp, steps, betas[g / nts], tss[g % nts], heterogeneous,
                    arr, pk, fin, NULL, &err, NULL, NULL);
        for (int c = 0; c < n; c++) cs += pk[c] + (double)arr[c];
        done += (int64_t)n * steps;
    }
    *checksum = cs;
    free(arr); free(pk); free(fin);
    return done;
}

/* ------------------------------------------------------------------------ */
/* DDQN learner: ReplayBuffer (ReplayBuffer.scala:10-22), MLP.trainBatch      */
/* (MLP.scala:29-99), DDQNAgent.observe/ddqnTarget/learn (DDQNAgent.scala:    */
/* 34-65).  Parity unpinned with respect to a live JVM run: the reference's   */
/* only test of this code (ReplayBufferSpec.scala) checks determinism and set */
/* membership, both of which the tests repeat.  scala.util.Random.shuffle is  */
/* the Scala 2.13 library routine (a third-party dependency absent from the   */
/* reference tree): copy, then for n = length down to 2 swap(n-1, nextInt(n)).*/
/* ------------------------------------------------------------------------ */

typedef struct {
    double fs[7], fs2[7], r;
    int32_t a, done;
} transition;

typedef struct {
    orc_learn_hyper hp;
    mlp online, target;
    transition *buf;
    int32_t size, idx;
    jrand rnd; /* ReplayBuffer.scala:13 Random(123) */
    int64_t steps;
    double ema;
} learner;

/* xs.sum on a non-empty collection: reduce(_ + _), left to right */
static double reduce_sum(const double *x, int n) {
    double acc = x[0];
    for (int k = 1; k < n; k++) acc = acc + x[k];
    return acc;
}
static double clip_sym(double v, double c) { return isfinite(c) ? fmax(-c, fmin(c, v)) : v; } /* MLP.scala:24-27 */

/* forward pass keeping the pre-activations (MLP.scala:39-42) */
static void mlp_forward(const mlp *m, const double *x, double clamp, double *z1, double *h1, double *z2, double *h2,
                        double *y, double *yc) {
    matvec(m->W1, x, m->h, m->in, z1);
    for (int k = 0; k < m->h; k++) { z1[k] = z1[k] + m->b1[k]; h1[k] = z1[k] > 0 ? z1[k] : 0.0; }
    matvec(m->W2, h1, m->h, m->h, z2);
    for (int k = 0; k < m->h; k++) { z2[k] = z2[k] + m->b2[k]; h2[k] = z2[k] > 0 ? z2[k] : 0.0; }
    matvec(m->W3, h2, m->out, m->h, y);
    for (int k = 0; k < m->out; k++) { y[k] = y[k] + m->b3[k]; yc[k] = clip_sym(y[k], clamp); }
}

/* MLP.trainBatch (MLP.scala:29-99): per-sample backprop accumulated in sample order, mean, clip, SGD. */
static double mlp_train_batch(mlp *m, const double *xs, const double *tgts, int n, double grad_clip, double clamp,
                              double lr) {
    const int in = m->in, h = m->h, out = m->out;
    double *gW1 = calloc((size_t)h * in, 8), *gb1 = calloc((size_t)h, 8), *gW2 = calloc((size_t)h * h, 8);
    double *gb2 = calloc((size_t)h, 8), *gW3 = calloc((size_t)out * h, 8), *gb3 = calloc((size_t)out, 8);
    double *z1 = malloc(8 * (size_t)h), *h1 = malloc(8 * (size_t)h), *z2 = malloc(8 * (size_t)h), *h2 = malloc(8 * (size_t)h);
    double *dh2 = malloc(8 * (size_t)h), *dz2 = malloc(8 * (size_t)h), *dh1 = malloc(8 * (size_t)h), *dz1 = malloc(8 * (size_t)h);
    double *tmp = malloc(8 * (size_t)(h > out ? h : out));
    double y[64], yc[64], err[64], sq[64];
    double loss_sum = 0.0;
    for (int k = 0; k < n; k++) {
        const double *x0 = xs + (size_t)k * in, *tgt = tgts + (size_t)k * out;
        mlp_forward(m, x0, clamp, z1, h1, z2, h2, y, yc);
        for (int i = 0; i < out; i++) { err[i] = yc[i] - tgt[i]; sq[i] = 0.5 * err[i] * err[i]; }
        loss_sum += reduce_sum(sq, out);
        /* backprop (:49-58); note dz = dh * relu(z), the value and not the derivative, as the reference has it */
        for (int j = 0; j < h; j++) {
            for (int i = 0; i < out; i++) tmp[i] = m->W3[(size_t)i * h + j] * err[i];
            dh2[j] = reduce_sum(tmp, out);
            dz2[j] = dh2[j] * (z2[j] > 0 ? z2[j] : 0.0);
        }
        for (int j = 0; j < h; j++) {
            for (int i = 0; i < h; i++) tmp[i] = m->W2[(size_t)i * h + j] * dz2[i];
            dh1[j] = reduce_sum(tmp, h);
            dz1[j] = dh1[j] * (z1[j] > 0 ? z1[j] : 0.0);
        }
        /* accumulate (:61-72) */
        for (int i = 0; i < out; i++) for (int j = 0; j < h; j++) gW3[(size_t)i * h + j] += err[i] * h2[j];
        for (int i = 0; i < out; i++) gb3[i] += err[i];
        for (int i = 0; i < h; i++) for (int j = 0; j < h; j++) gW2[(size_t)i * h + j] += dz2[i] * h1[j];
        for (int i = 0; i < h; i++) gb2[i] += dz2[i];
        for (int i = 0; i < h; i++) for (int j = 0; j < in; j++) gW1[(size_t)i * in + j] += dz1[i] * x0[j];
        for (int i = 0; i < h; i++) gb1[i] += dz1[i];
    }
    const double scale = 1.0 / (double)(n > 1 ? n : 1); /* :78 */
#define STEP(W, g, cnt) for (size_t q = 0; q < (size_t)(cnt); q++) { double gg = clip_sym((g)[q] * scale, grad_clip); (W)[q] -= lr * gg; }
    STEP(m->W3, gW3, out * h) STEP(m->b3, gb3, out) STEP(m->W2, gW2, h * h) STEP(m->b2, gb2, h)
    STEP(m->W1, gW1, h * in) STEP(m->b1, gb1, h)
#undef STEP
    free(gW1); free(gb1); free(gW2); free(gb2); free(gW3); free(gb3);
    free(z1); free(h1); free(z2); free(h2); free(dh2); free(dz2); free(dh1); free(dz1); free(tmp);
    return loss_sum / (double)(n > 1 ? n : 1);
}

static void mlp_copy(mlp *dst, const mlp *src) { /* MLP.copyFrom :101-109 */
    memcpy(dst->W1, src->W1, 8 * (size_t)src->h * src->in); memcpy(dst->b1, src->b1, 8 * (size_t)src->h);
    memcpy(dst->W2, src->W2, 8 * (size_t)src->h * src->h); memcpy(dst->b2, src->b2, 8 * (size_t)src->h);
    memcpy(dst->W3, src->W3, 8 * (size_t)src->out * src->h); memcpy(dst->b3, src->b3, 8 * (size_t)src->out);
}

EXPORT void *orc_learner_create(const orc_learn_hyper *hp) {
    learner *L = (learner *)calloc(1, sizeof(learner));
    L->hp = *hp;
    mlp_init(&L->online, 7, hp->hidden, 8); /* DDQNAgent.scala:8-9: two networks, same fixed init seed */
    mlp_init(&L->target, 7, hp->hidden, 8);
    L->buf = (transition *)calloc((size_t)(hp->replay_capacity > 0 ? hp->replay_capacity : 1), sizeof(transition));
    jr_init(&L->rnd, 123);
    return L;
}
EXPORT void orc_learner_destroy(void *h) {
    learner *L = (learner *)h;
    mlp_free(&L->online); mlp_free(&L->target); free(L->buf); free(L);
}
/* DDQNAgent.observe -> ReplayBuffer.push (ReplayBuffer.scala:15-17); only the features of s and s2 are ever read */
EXPORT void orc_learner_observe(void *h, const orc_state *s, int32_t a, double r, const orc_state *s2, int32_t done) {
    learner *L = (learner *)h;
    transition t;
    orc_features(s, t.fs); orc_features(s2, t.fs2);
    t.a = a; t.r = r; t.done = done;
    if (L->size < L->hp.replay_capacity) L->buf[L->size++] = t;
    else { L->buf[L->idx] = t; L->idx = (L->idx + 1) % L->hp.replay_capacity; }
}
EXPORT int32_t orc_learner_replay_size(void *h) { return ((learner *)h)->size; }
EXPORT void orc_learner_set_steps(void *h, int64_t steps) { ((learner *)h)->steps = steps; }
EXPORT double orc_learner_ema_loss(void *h) { return ((learner *)h)->ema; }

/* ReplayBuffer.sample (ReplayBuffer.scala:21-22): positions of the first n elements of rnd.shuffle(buf) */
EXPORT int32_t orc_learner_sample(void *h, int32_t n, int32_t *out) {
    learner *L = (learner *)h;
    const int len = L->size;
    int32_t *p = (int32_t *)malloc(sizeof(int32_t) * (size_t)(len > 0 ? len : 1));
    for (int i = 0; i < len; i++) p[i] = i;
    for (int m = len; m >= 2; m--) {
        int k = jr_next_int_bound(&L->rnd, m);
        int32_t t = p[m - 1]; p[m - 1] = p[k]; p[k] = t;
    }
    const int take = n < len ? n : len;
    for (int i = 0; i < take; i++) out[i] = p[i];
    free(p);
    return take;
}

static double learner_clip_q(const learner *L, double x) { return fmax(-L->hp.clip_q, fmin(L->hp.clip_q, x)); } /* :49 */

/* DDQNAgent.learn (DDQNAgent.scala:51-65) */
EXPORT double orc_learner_learn(void *h, int32_t batch, int32_t grad_accum) {
    learner *L = (learner *)h;
    const int need = batch * grad_accum > batch ? batch * grad_accum : batch;
    if (L->size < need) return 0.0;
    int32_t *ix = (int32_t *)malloc(sizeof(int32_t) * (size_t)need);
    orc_learner_sample(L, need, ix);
    double *xs = (double *)malloc(8 * (size_t)need * 7), *tg = (double *)malloc(8 * (size_t)need * 8);
    double qn[8], qt[8];
    for (int k = 0; k < need; k++) {
        const transition *e = &L->buf[ix[k]];
        memcpy(xs + (size_t)k * 7, e->fs, 56);
        double *q = tg + (size_t)k * 8; /* ddqnTarget :37-47 */
        mlp_predict(&L->online, e->fs, L->hp.clip_q, q);
        if (e->done) q[e->a] = learner_clip_q(L, e->r);
        else {
            mlp_predict(&L->online, e->fs2, L->hp.clip_q, qn);
            const int best = argmax_first(qn, 8);
            mlp_predict(&L->target, e->fs2, L->hp.clip_q, qt);
            q[e->a] = learner_clip_q(L, e->r + L->hp.gamma * qt[best]);
        }
    }
    const double loss = mlp_train_batch(&L->online, xs, tg, need, L->hp.grad_clip, L->hp.clip_q, L->hp.lr);
    if (L->steps % L->hp.target_update_every == 0) mlp_copy(&L->target, &L->online);
    L->ema = (L->ema == 0.0) ? loss : L->hp.ema_alpha * loss + (1 - L->hp.ema_alpha) * L->ema;
    free(ix); free(xs); free(tg);
    return loss;
}
/* which: 0 = online, 1 = target.  W1[h*7], b1[h], W2[h*h], b2[h], W3[8*h], b3[8] */
EXPORT void orc_learner_get_weights(void *h, int32_t which, double *W1, double *b1, double *W2, double *b2, double *W3,
                                    double *b3) {
    const learner *L = (const learner *)h;
    const mlp *m = which ? &L->target : &L->online;
    memcpy(W1, m->W1, 8 * (size_t)m->h * m->in); memcpy(b1, m->b1, 8 * (size_t)m->h);
    memcpy(W2, m->W2, 8 * (size_t)m->h * m->h); memcpy(b2, m->b2, 8 * (size_t)m->h);
    memcpy(W3, m->W3, 8 * (size_t)m->out * m->h); memcpy(b3, m->b3, 8 * (size_t)m->out);
}
EXPORT void orc_learner_q_values(void *h, int32_t which, const orc_state *s, double *q) {
    const learner *L = (const learner *)h;
    double f[7];
    orc_features(s, f);
    mlp_predict(which ? &L->target : &L->online, f, L->hp.clip_q, q);
}
EXPORT void orc_mlp_train_batch(int32_t h, double *W1, double *b1, double *W2, double *b2, double *W3, double *b3,
                                const double *xs, const double *tgts, int32_t n, double grad_clip, double clamp,
                                double lr, double *loss) {
    mlp m = {7, h, 8, W1, b1, W2, b2, W3, b3};
    *loss = mlp_train_batch(&m, xs, tgts, n, grad_clip, clamp, lr);
}

/* ------------------------------------------------------------------------ */
/* WorldToy.stepOne with its agents learning (WorldToy.scala:83-170): every   */
/* node owns a DDQNAgent (:50); hp[c], learn_every[c], batch_size[c] come     */
/* from countryHyperParams (:49), intensity[c] from the name match (:157-162).*/
/* ------------------------------------------------------------------------ */
EXPORT void orc_world_enable_learning(void *h, const orc_learn_hyper *hp, const int32_t *learn_every,
                                      const int32_t *batch_size, const int32_t *intensity) {
    world *w = (world *)h;
    int n = w->n;
    w->learners = (void **)malloc(sizeof(void *) * (size_t)n);
    w->learn_every = (int32_t *)malloc(4 * (size_t)n); w->batch_size = (int32_t *)malloc(4 * (size_t)n);
    w->intensity = (int32_t *)malloc(4 * (size_t)n); w->last_learn = (int64_t *)calloc((size_t)n, 8);
    for (int c = 0; c < n; c++) {
        w->learners[c] = orc_learner_create(&hp[c]);
        w->learn_every[c] = learn_every[c]; w->batch_size[c] = batch_size[c]; w->intensity[c] = intensity[c];
    }
}

/* losses[c] = what agent.learn returned on this step, or -1 where the node did not learn */
EXPORT void orc_world_step_learn(void *h, const orc_hyper *hps, int32_t t_max, double *rewards, double *losses) {
    world *w = (world *)h;
    int n = w->n;
    orc_state *local = (orc_state *)malloc(sizeof(orc_state) * (size_t)n);
    double q[8];
    for (int c = 0; c < n; c++) {
        agent *ag = &w->agents[c];
        learner *L = (learner *)w->learners[c];
        int a;
        double eps = agent_epsilon(ag, &hps[c]);          /* :32: argument read before countStep */
        ag->steps += 1;
        L->steps = ag->steps;
        if (jr_next_double(&ag->rnd) < eps) a = jr_next_int_bound(&ag->rnd, 8);
        else { orc_learner_q_values(L, 0, &w->cur[c], q); a = argmax_first(q, 8); }
        int64_t tb = sum_state(&w->cur[c]);
        int32_t done; double r;
        double g = jr_next_gaussian(&w->env_rnd[c]);
        orc_env_step(&w->env[c], &w->prof[c], &w->cur[c], a, g, &local[c], &r, &done);
        w->last_a[c] = a;
        if (rewards) rewards[c] = r;
        if (w->validate) {
            int64_t ta = sum_state(&local[c]);
            if (ta != tb && w->err == 0) { w->err = 1; w->err_node = c; w->err_t = w->cur[c].t; }
        }
        orc_learner_observe(L, &w->cur[c], a, r, &local[c], done); /* :99 */
    }
    mobility_apply(w, local, t_max, NULL, NULL);
    free(local);
    w->g_step += 1;
    for (int c = 0; c < n; c++) { /* :150-167 */
        if (losses) losses[c] = -1.0;
        if (w->g_step - w->last_learn[c] >= w->learn_every[c]) {
            learner *L = (learner *)w->learners[c];
            if (L->size >= w->batch_size[c] * 4) {
                double loss = orc_learner_learn(L, w->batch_size[c], w->intensity[c]);
                if (losses) losses[c] = loss;
                w->last_learn[c] = w->g_step;
            }
        }
    }
}
EXPORT void *orc_world_learner(void *h, int32_t c) { return ((world *)h)->learners[c]; }
