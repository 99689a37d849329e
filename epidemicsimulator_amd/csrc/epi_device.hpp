// Device-side building blocks (gfx950, wave64) for the metapopulation step.
//
// Everything here must reproduce the reference's IEEE-754 binary64 arithmetic one
// rounding at a time: the translation unit is compiled with -ffp-contract=off, so
// a*b+c is a multiply followed by an add exactly as on the JVM.  Reference paths
// are relative to src/main/scala/epidemic/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/epi.h"

namespace epi {

// ---------------------------------------------------------------------------------------------
// Per-country constants of EpidemicEnv.step that do not depend on the state (env.scala:59-101).
// Each member is produced by the same operation sequence the reference runs every step, so
// hoisting them changes no rounding.
// ---------------------------------------------------------------------------------------------
struct StepConsts {
    double beta0;   // baseBeta * profile.betaMul                      env.scala:59
    double gamma;   // baseGamma                                       :60
    double ifr0;    // baseIFR * profile.ifrMul                        :61
    double cnoise;  // noise * (0.2 + profile.noise * 1.5)             :90
    double eff1;    // Distancing: economicResilience * 0.75 + 0.5     :66
    double eff2;    // TravelBan: (1 - connectivityStrength)*0.4 + 0.4 :69
    double eff3;    // Lockdown: economicResilience * 0.5 + 0.2        :72
    double surge;   // SurgeCare: 1 + healthcareCapacity               :75
    double vax6;    // TargetedVax: 0.02 * healthcareCapacity          :98
    double vax7;    // MassVax: 0.05 * healthcareCapacity              :99
};
constexpr int kStepConstCount = 10;
constexpr int kTopWeights = 128;  // sorted weights kept per row for the small-export shortcut

__host__ __device__ inline StepConsts make_step_consts(const epi_env_params& e, const epi_profile& p) {
    StepConsts c;
    c.beta0 = e.base_beta * p.beta_mul;
    c.gamma = e.base_gamma;
    c.ifr0 = e.base_ifr * p.ifr_mul;
    c.cnoise = e.noise * (0.2 + p.noise * 1.5);
    c.eff1 = p.econ_resilience * 0.75 + 0.5;
    c.eff2 = (1.0 - p.connectivity) * 0.4 + 0.4;
    c.eff3 = p.econ_resilience * 0.5 + 0.2;
    c.surge = 1.0 + p.healthcare_capacity;
    c.vax6 = 0.02 * p.healthcare_capacity;
    c.vax7 = 0.05 * p.healthcare_capacity;
    return c;
}

// Reward constants (env.scala:142-234), again hoisted without reordering any product.
struct RewardConsts {
    double health_w, gdp_f, hc_f, dens_f, age_f;
    double cost[8];   // calculateExtremeActionCost(a) * economicWeight * extremeGDPFactor   :185
    double bonus[8];  // riskFactor * npiBonus(a)                                            :191
    double rmul;      // 1.0 + 0.75 * riskFactor
    double clip_lo, clip_hi;
};

__host__ __device__ inline RewardConsts make_reward_consts(const epi_env_params& e, const epi_profile& p) {
    RewardConsts c;
    const double hc = p.healthcare_capacity, g = p.gdp_per_capita, dn = p.population_density;
    c.health_w = hc > 0.8 ? 0.2 : (hc > 0.5 ? 0.6 : 0.9);
    const double econ_w = 1.0 - c.health_w;
    c.gdp_f = g > 50000 ? 0.3 : (g > 20000 ? 1.0 : (g > 5000 ? 2.5 : 5.0));
    c.hc_f = hc > 0.8 ? 0.2 : (hc > 0.5 ? 1.0 : 8.0);
    c.dens_f = dn > 400 ? 3.0 : (dn > 200 ? 2.0 : (dn > 100 ? 1.0 : 0.5));
    c.age_f = 1.0 + p.age_elderly * 8.0;
    const double base_cost[8] = {0.0, 0.05, 0.08, 0.15, 0.01, 0.06, 0.03, 0.05};
    const double em = g > 50000 ? 0.3 : (g > 20000 ? 1.0 : (g > 5000 ? 3.0 : 8.0));
    for (int a = 0; a < 8; ++a) {
        const double mm = (a == 3 || a == 2) ? 1.0 + (1.0 - p.econ_resilience) * 2.0 : 1.0;
        const double ac = base_cost[a] * em * mm * 30.0;
        c.cost[a] = ac * econ_w * c.gdp_f;
        double nb = 0.0;
        if (a == 3) nb = 0.05 * p.econ_resilience;
        else if (a == 2) nb = 0.03 * p.connectivity;
        else if (a == 1) nb = 0.02;
        else if (a == 6) nb = 0.02 * hc;
        else if (a == 7) nb = 0.04 * hc;
        c.bonus[a] = e.risk_factor * nb;
    }
    c.rmul = 1.0 + 0.75 * e.risk_factor;
    c.clip_lo = -e.reward_clip * 3.0;
    c.clip_hi = e.reward_clip;
    return c;
}

#if defined(__HIPCC__)

namespace dev {

// ---------------------------------------------------------------------------------------------
// Wave64 reductions on the DPP network (no LDS): four row steps, two row-broadcast steps,
// result read from lane 63 as a wave-uniform scalar.
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    v += dpp_or<0xB1, 0xf>(0u, v);   // quad_perm [1,0,3,2]
    v += dpp_or<0x4E, 0xf>(0u, v);   // quad_perm [2,3,0,1]
    v += dpp_or<0x141, 0xf>(0u, v);  // row_half_mirror
    v += dpp_or<0x140, 0xf>(0u, v);  // row_mirror
    v += dpp_or<0x142, 0xa>(0u, v);  // row_bcast15 -> rows 1,3
    v += dpp_or<0x143, 0xc>(0u, v);  // row_bcast31 -> rows 2,3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
    v = max(v, dpp_or<0xB1, 0xf>(0u, v));
    v = max(v, dpp_or<0x4E, 0xf>(0u, v));
    v = max(v, dpp_or<0x141, 0xf>(0u, v));
    v = max(v, dpp_or<0x140, 0xf>(0u, v));
    v = max(v, dpp_or<0x142, 0xa>(0u, v));
    v = max(v, dpp_or<0x143, 0xc>(0u, v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
    v = min(v, dpp_or<0xB1, 0xf>(0xffffffffu, v));
    v = min(v, dpp_or<0x4E, 0xf>(0xffffffffu, v));
    v = min(v, dpp_or<0x141, 0xf>(0xffffffffu, v));
    v = min(v, dpp_or<0x140, 0xf>(0xffffffffu, v));
    v = min(v, dpp_or<0x142, 0xa>(0xffffffffu, v));
    v = min(v, dpp_or<0x143, 0xc>(0xffffffffu, v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ uint32_t dpp_zero(uint32_t v) {   // lanes without a valid source (or masked out) read 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}

// Inclusive prefix sum over the 64 lanes (lane l gets v_0 + ... + v_l): three shifted adds of the input inside each
// group of four lanes, two shifted adds of the partial sums inside each row of 16, two row broadcasts.
__device__ __forceinline__ uint32_t wave_prefix_sum(uint32_t v) {
    uint32_t t = v;
    t += dpp_zero<0x111, 0xf, 0xf>(v);   // row_shr:1
    t += dpp_zero<0x112, 0xf, 0xf>(v);   // row_shr:2
    t += dpp_zero<0x113, 0xf, 0xf>(v);   // row_shr:3
    t += dpp_zero<0x114, 0xf, 0xe>(t);   // row_shr:4 into banks 1..3
    t += dpp_zero<0x118, 0xf, 0xc>(t);   // row_shr:8 into banks 2..3
    t += dpp_zero<0x142, 0xa, 0xf>(t);   // row_bcast15 -> rows 1, 3
    t += dpp_zero<0x143, 0xc, 0xf>(t);   // row_bcast31 -> rows 2, 3
    return t;
}

// Sum of non-negative 64-bit values below 2^44 across the wave (world mass check,
// WorldToy.scala:103-104,142-145): split at bit 22 so both halves fit a 32-bit wave sum.
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
    const uint32_t lo = wave_sum((uint32_t)(v & 0x3fffffu));
    const uint32_t hi = wave_sum((uint32_t)(v >> 22));
    return ((unsigned long long)hi << 22) + lo;
}

__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ double clamp01(double x) { return fmax(0.0, fmin(1.0, x)); }  // env.scala:53

// ---------------------------------------------------------------------------------------------
// EpidemicEnv.allocateInt for three weights (env.scala:241-254), the S and I splits of
// env.scala:107,113.  The stable descending sort of three fractions is a rank computation:
// position(j) = #{k : f_k > f_j} + #{k < j : f_k == f_j}.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void alloc3(int32_t total, double w0, double w1, double w2, int32_t& a0, int32_t& a1,
                                       int32_t& a2) {
    a0 = a1 = a2 = 0;
    const double sum = (w0 + w1) + w2;
    if (total <= 0 || sum <= 0.0) return;
    const double t = (double)total;
    const double r0 = (w0 / sum) * t, r1 = (w1 / sum) * t, r2 = (w2 / sum) * t;
    const double fl0 = floor(r0), fl1 = floor(r1), fl2 = floor(r2);
    uint32_t b0 = (uint32_t)(int32_t)fl0, b1 = (uint32_t)(int32_t)fl1, b2 = (uint32_t)(int32_t)fl2;
    const int32_t rem = (int32_t)((uint32_t)total - (b0 + b1 + b2));
    const double f0 = r0 - fl0, f1 = r1 - fl1, f2 = r2 - fl2;
    const int p0 = (f1 > f0) + (f2 > f0);
    const int p1 = (f0 >= f1) + (f2 > f1);
    const int p2 = (f0 >= f2) + (f1 >= f2);
    b0 += (p0 < rem);
    b1 += (p1 < rem);
    b2 += (p2 < rem);
    a0 = (int32_t)b0; a1 = (int32_t)b1; a2 = (int32_t)b2;
}

// ---------------------------------------------------------------------------------------------
// EpidemicEnv.step on integer compartments (env.scala:58-133).  Inputs are the already rounded,
// non-negative counts s0..v0 (env.scala:81-85), hospCap, the action id 0..7 and the Gaussian draw.
// ---------------------------------------------------------------------------------------------
struct LocalOut {
    int32_t s, i, r, d, v;
    double cap;
};

__device__ __forceinline__ LocalOut local_step(const StepConsts& c, int32_t s0, int32_t i0, int32_t r0, int32_t d0,
                                               int32_t v0, double cap, int act, double z) {
    double beta = c.beta0, gamma = c.gamma;
    if (act == 1) beta *= c.eff1;
    else if (act == 2) beta *= c.eff2;
    else if (act == 3) beta *= c.eff3;
    else if (act == 5) { cap *= c.surge; gamma *= 1.3; }

    const uint32_t us = (uint32_t)s0, ui = (uint32_t)i0, ur = (uint32_t)r0, ud = (uint32_t)d0, uv = (uint32_t)v0;
    const int32_t pop0 = (int32_t)(us + ui + ur + ud + uv);
    const double mix = fmax(1.0, (double)(int32_t)(us + ui + ur));

    const double stoch = 1.0 + c.cnoise * z;
    const double effBeta = clamp01(beta * stoch);
    const double effGamma = clamp01(gamma * stoch);
    const double effIFR = clamp01((((double)i0 > cap) ? c.ifr0 * 6.0 : c.ifr0) * stoch);
    const double vax = act == 6 ? c.vax6 : (act == 7 ? c.vax7 : 0.0);

    const double pInf = clamp01(effBeta * ((double)i0 / mix));
    const double pVax = clamp01(vax);
    int32_t sStay, sInf, sVax;
    alloc3(s0, fmax(0.0, 1.0 - pInf - pVax), pInf, pVax, sStay, sInf, sVax);

    const double pRec = clamp01(effGamma);
    const double pDie = clamp01(effIFR);
    int32_t iStay, rec, die;
    alloc3(i0, fmax(0.0, 1.0 - pRec - pDie), pRec, pDie, iStay, rec, die);

    uint32_t sN = us - (uint32_t)sInf - (uint32_t)sVax;
    const uint32_t iN = (uint32_t)iStay + (uint32_t)sInf;
    const uint32_t rN = ur + (uint32_t)rec;
    const uint32_t dN = ud + (uint32_t)die;
    const uint32_t vN = uv + (uint32_t)sVax;
    const int32_t resid = (int32_t)((uint32_t)pop0 - (sN + iN + rN + dN + vN));
    if (resid != 0) sN = (uint32_t)max(0, (int32_t)(sN + (uint32_t)resid));

    LocalOut o;
    o.s = (int32_t)sN; o.i = (int32_t)iN; o.r = (int32_t)rN; o.d = (int32_t)dN; o.v = (int32_t)vN;
    o.cap = cap;
    return o;
}

// calculateExtremeHeterogeneousReward (env.scala:142-194) from the previous I and D (as the
// doubles held in State) and the next integer state.
__device__ __forceinline__ double reward_of(const RewardConsts& c, double prev_i, double prev_d, const LocalOut& nx,
                                            int act) {
    const double ns = (double)nx.s, ni = (double)nx.i, nr = (double)nx.r, nd = (double)nx.d;
    const double Nnext = fmax(1.0, ns + ni + nr + nd);
    const double di = (ni - prev_i) / Nnext;
    const double dd = (nd - prev_d) / Nnext;
    const double overflow = fmax(0.0, ni / fmax(1.0, nx.cap) - 1.0);
    const double infectionPenalty = di * 200.0 * c.dens_f * c.health_w * c.gdp_f;
    const double mortalityPenalty = dd * 2000.0 * c.age_f * c.hc_f * c.health_w;
    const double healthcarePenalty = overflow * 1000.0 * c.health_w * c.hc_f;
    const double baseReward = -(infectionPenalty + mortalityPenalty + healthcarePenalty + c.cost[act]);
    const double shaped = baseReward * c.rmul + c.bonus[act];
    return fmax(c.clip_lo, fmin(c.clip_hi, shaped));
}

// Math.round(x).toInt clamped at 0 (env.scala:81-85) for a finite double.
__device__ __forceinline__ int32_t round_count(double x) {
    const double fl = floor(x);
    double r = (x - fl >= 0.5) ? fl + 1.0 : fl;
    long long l;
    if (r >= 9223372036854775807.0) l = 0x7fffffffffffffffLL;
    else if (r <= -9223372036854775808.0) l = (long long)0x8000000000000000ULL;
    else l = (long long)r;
    const int32_t w = (int32_t)(uint32_t)(unsigned long long)l;
    return w > 0 ? w : 0;
}

// ---------------------------------------------------------------------------------------------
// Largest-remainder apportionment of one wide row held in registers: the mobility split of
// WorldToy.scala:117-118 (EpidemicEnv.allocateInt, env.scala:241-254) for n <= 64*R weights.
//
// Layout: column j lives in slot r = j / 64 of lane j % 64.  qrow[j] is the already normalised
// weight w_j / sumW (0 for the diagonal and for padding columns j >= n), 64*R entries.
//
// The reference sorts the fractional parts (stable, descending) and hands one unit to the first
// `rem`.  Here the rem-th largest fraction is found by counting: the high 32 bits of a
// non-negative double order like the double itself, so a threshold K with
// #{hi32(frac) >= K} == rem selects exactly the rem largest fractions.  The threshold is
// bracketed by interpolation with a bisection safeguard.  When no such K exists two or more
// fractions share the pivot's high word; that case is finished exactly on the low words and,
// for identical fractions, in ascending column order (tie_select).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double frac_of(double raw) { return __builtin_amdgcn_fract(raw); }

__device__ __forceinline__ double from_hi_word(uint32_t hi) { return __hiloint2double((int)hi, 0); }

// Search memory of one (row, compartment): the threshold that solved the previous step, the
// remainder it was solved for and the reciprocal density of fractions there (`rho`: value units per
// element).  Exports change slowly from one day to the next, so the old solution plus a Newton
// step on the new remainder lands within a few elements of the new threshold.  rho <= 0 marks
// "no memory".
struct Warm {
    uint32_t key;
    float rho;
    int32_t rem;
    int32_t pad_;
};

// Exact finish for fractions whose high word equals K.  Members are the columns j < n with
// hi32(frac_j) == K; `m` of them are selected by (low word descending, column ascending).
// Returns a per-lane bit mask over the slots r.  Fractions are recomputed from the row in
// memory; this path is rare.
template <int R>
__device__ __noinline__ uint32_t tie_select(const double* __restrict__ qrow, double Ed, uint32_t K, int m, int n,
                                            int lane) {
    uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
        const int j = r * 64 + lane;
        const double fr = frac_of(qrow[j] * Ed);
        if (j < n && (uint32_t)__double2hiint(fr) == K) {
            const uint32_t l = (uint32_t)__double2loint(fr);
            mn = min(mn, l);
            mx = max(mx, l);
        }
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    unsigned long long Llo = mn, Lhi = (unsigned long long)mx + 1ull;
    int c_hi = 0;
    while (Lhi - Llo > 1ull) {
        const unsigned long long mid = Llo + ((Lhi - Llo) >> 1);
        uint32_t c = 0, sel = 0;
#pragma unroll 1
        for (int r = 0; r < R; ++r) {
            const int j = r * 64 + lane;
            const double fr = frac_of(qrow[j] * Ed);
            const bool in = j < n && (uint32_t)__double2hiint(fr) == K && (unsigned long long)(uint32_t)__double2loint(fr) >= mid;
            c += in;
            sel |= (uint32_t)in << r;
        }
        c = wave_sum(c);
        if ((int)c == m) return sel;
        if ((int)c > m) Llo = mid;
        else { Lhi = mid; c_hi = (int)c; }
    }
    // fractions identical to the pivot: ascending column order
    const int m2 = m - c_hi;
    int run = 0;
    uint32_t sel = 0;
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
        const int j = r * 64 + lane;
        const double fr = frac_of(qrow[j] * Ed);
        const bool mem = j < n && (uint32_t)__double2hiint(fr) == K;
        const uint32_t l = (uint32_t)__double2loint(fr);
        const bool above = mem && (unsigned long long)l > Llo;
        const bool eq = mem && (unsigned long long)l == Llo;
        const unsigned long long mask = __ballot(eq);
        const int pre = (int)lanes_below(mask);
        if (above || (eq && run + pre < m2)) sel |= 1u << r;
        run += __popcll(mask);
    }
    return sel;
}

// Adds allocateInt(E, row)[j] into acc[r] for every column except `diag` (WorldToy.scala:121 drops
// the origin's own entry).  E > 0.  All 64 lanes must be active.  `warm` (LDS or global, may be
// null) carries the search memory of this row and compartment between calls; it only steers the
// order in which thresholds are tried, never the result.
//
// Threshold search.  Invariant: #{key >= lo} = clo > rem > chi = #{key >= hi}.  Every probe `mid`
// lies strictly inside (lo, hi), so the bracket shrinks until a probe counts exactly rem (done:
// the rem largest fractions are {key >= mid}) or hi == lo + 1 (fractions share a high word: exact
// finish in tie_select).  Probe choice: Newton steps from the remembered solution / the last probe
// with the remembered density, then interpolation inside the bracket (in the value, and on a cold
// start alternately in the logarithm, since the high word is exponent-major), with a plain
// bisection every fourth probe as the safeguard that bounds the iteration count.
// The search proper, from any valid bracket: #{key >= lo} = clo > rem > chi = #{key >= hi} (clo = n for lo = 0),
// or `found` with the threshold in `mid`.  Adds the selected units to acc.
// #{r : key[r] >= K} of this lane for keys and K below 2^30 (high words of fractions), 1 <= K.  key + (2^31 - K) has
// its top bit set exactly when key >= K; a funnel shift collects the top bits of all slots in one register and one
// population count adds them up.  Two plain integer instructions per key, like a compare and an add-with-carry, but
// none of them writes or reads a condition register: those cost about 4.4 cycles per wave instruction against 3 for
// an addition, and each pair needs two wait states between its halves.
template <int R>
__device__ __forceinline__ uint32_t count_ge(const uint32_t (&key)[R], uint32_t K) {
    const uint32_t C = 0x80000000u - K;
    uint32_t u0 = 0u, u1 = 0u;   // two chains, so that consecutive funnel shifts do not wait for each other
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r & 1) u1 = __builtin_amdgcn_alignbit(u1, key[r] + C, 31);
        else u0 = __builtin_amdgcn_alignbit(u0, key[r] + C, 31);
    }
    return (uint32_t)__popc(u0) + (uint32_t)__popc(u1);
}

template <int R>
__device__ __forceinline__ void search_tail(const uint32_t (&key)[R], uint32_t (&acc)[R], double Ed, int32_t rem, int n,
                                            int diag, const double* __restrict__ qrow, int lane, Warm* warm, uint32_t hint,
                                            bool cold, float slope, uint32_t pk, int32_t prem, uint32_t lo, uint32_t hi,
                                            int clo, int chi, bool lo_probed, bool hi_probed, bool found, uint32_t mid,
                                            unsigned long long* stats) {
    int it = 0, last_c = -1;
    uint32_t last_k = 0u;
    // A bracket end that was probed and is within kSnap elements of the target is moved element by
    // element instead: the smallest key >= lo (or the largest key < hi) is an exact next threshold, found
    // with one min (max) reduction, and the probe that follows it always makes progress.
    constexpr int kSnap = (R <= 4) ? 2 : 1;  // measured best on the 200- and the 1,000-country grids
    while (!found && hi - lo > 1u) {
        if (lo_probed && clo - rem <= kSnap && !(hi_probed && rem - chi < clo - rem)) {
            // key - lo wraps around to more than 2^31 for the keys below lo (all keys are below 2^30): one subtraction and
            // one minimum per key, no compare / select pair
            uint32_t x = 0xffffffffu;
#pragma unroll
            for (int r = 0; r < R; ++r) x = min(x, key[r] - lo);
            x = wave_min(x);          // lo <= lo + x < hi, and no key lies in [lo, lo + x)
            lo += (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
            if (hi - lo <= 1u) break;  // every key of the bracket shares one high word: tie
            mid = lo + 1u;
        } else if (hi_probed && rem - chi <= kSnap) {
            uint32_t y = 0xffffffffu;   // (hi - 1) - key wraps around for the keys at or above hi
#pragma unroll
            for (int r = 0; r < R; ++r) y = min(y, (hi - 1u) - key[r]);
            y = wave_min(y);          // hi - 1 - y is the largest key below hi: no key lies above it and below hi
            hi -= (uint32_t)__builtin_amdgcn_readfirstlane((int)y);
            if (hi - lo <= 1u) break;
            mid = hi - 1u;
        } else if (it == 0 && hint != 0u) {
            mid = hint;
        } else if (!cold && it < 4 && !(lo_probed && hi_probed)) {
            const uint32_t bk = it == 0 ? pk : last_k;
            const int bc = it == 0 ? prem : last_c;
            const double t = from_hi_word(bk) + (double)((float)(bc - rem) * slope);
            mid = t > 0.0 ? (uint32_t)__double2hiint(t) : 0u;
        } else if ((it & 3) == 3) {
            mid = lo + ((hi - lo) >> 1);
        } else {
            // fraction of the way from lo to hi at which the count is expected to reach rem
            const float f = ((float)(clo - rem) + 0.5f) * __builtin_amdgcn_rcpf((float)(clo - chi + 1));
            if (it & 1) {
                const uint32_t reach = 64u << 20;
                const uint32_t lk = (hi - lo > reach) ? hi - reach : lo;
                mid = lk + (uint32_t)((float)(hi - lk) * f);
            } else {
                const double vlo = from_hi_word(lo), vhi = from_hi_word(hi);
                mid = (uint32_t)__double2hiint(vlo + (vhi - vlo) * (double)f);
            }
        }
        mid = max(mid, lo + 1u);
        mid = min(mid, hi - 1u);
        mid = (uint32_t)__builtin_amdgcn_readfirstlane((int)mid);
        const int c = (int)wave_sum(count_ge<R>(key, mid));
        if (last_c >= 0 && abs(c - last_c) >= 3 && mid != last_k) {
            const float s2 = (float)(from_hi_word(mid) - from_hi_word(last_k)) * __builtin_amdgcn_rcpf((float)(last_c - c));
            if (s2 > 0.f) slope = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s2)));
        }
        last_k = mid;
        last_c = c;
        ++it;
        if (c == rem) { found = true; break; }
        if (c > rem) { lo = mid; clo = c; lo_probed = true; }
        else { hi = mid; chi = c; hi_probed = true; }
    }
    if (stats && lane == 0) {
        atomicAdd(&stats[1], (unsigned long long)it);
        atomicAdd(&stats[2], cold ? 1ull : 0ull);
        if (found) atomicAdd(&stats[4 + min(it, 27)], 1ull);   // histogram index = counting probes used
        else atomicAdd(&stats[3], 1ull);
    }
    if (found) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] += (key[r] >= mid);
        if (!(slope > 0.f)) slope = (float)(from_hi_word(hi) - from_hi_word(lo)) * __builtin_amdgcn_rcpf((float)(clo - chi));
    } else {
        // two or more fractions share the pivot's high word `lo`
        const uint32_t sel = tie_select<R>(qrow, Ed, lo, rem - chi, n, lane);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = r * 64 + lane;
            const bool take = (key[r] > lo) || ((sel >> r) & 1u);
            acc[r] += (take && j != diag);
        }
        mid = lo;
    }
    if (warm && lane == 0 && slope > 0.f) {
        warm->key = mid;
        warm->rho = slope;
        warm->rem = rem;
    }
}

// The warm-started search of one compartment from keys that are already formed (base_total = whole units handed out).
template <int R>
__device__ __forceinline__ void search_from_keys(const uint32_t (&key)[R], uint32_t (&acc)[R], int32_t E, uint32_t base_total, int n,
                                                 int diag, const double* __restrict__ qrow, int lane, Warm* warm,
                                                 const double* __restrict__ qtop, unsigned long long* stats) {
    const double Ed = (double)E;
    const int32_t rem = (int32_t)((uint32_t)E - base_total);
    if (stats && lane == 0) atomicAdd(&stats[0], 1ull);
    if (rem <= 0) return;
    if (rem >= n) {  // every real column gets a unit (the reference stops at frac.length)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = r * 64 + lane;
            acc[r] += (j < n && j != diag);
        }
        return;
    }
    // Wave-uniform search state lives in scalar registers: every value that steers the loop is
    // passed through readfirstlane so that the branches below are scalar branches.
    float slope = 0.f;  // value units per element near the threshold (reciprocal density); <= 0: none
    uint32_t pk = 0u;
    int32_t prem = 0;
    if (warm) {
        slope = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(warm->rho)));
        pk = (uint32_t)__builtin_amdgcn_readfirstlane((int)warm->key);
        prem = __builtin_amdgcn_readfirstlane(warm->rem);
    }
    const bool cold = !(slope > 0.f);
    uint32_t hi = 0x3ff00000u;  // every fraction is below 1.0
    if (cold) {
        uint32_t kmax = 0u;
#pragma unroll
        for (int r = 0; r < R; ++r) kmax = max(kmax, key[r]);
        hi = wave_max(kmax) + 1u;
    }
    // No column received a whole unit: every fraction is the product q_j * E itself, so the rem = E
    // largest are the E largest weights of the row.  qtop holds the row's kTopWeights largest
    // weights in descending order; the E-th of them times E is the threshold to try first.
    uint32_t hint = 0u;
    if (qtop && base_total == 0u && E < kTopWeights)
        hint = (uint32_t)__builtin_amdgcn_readfirstlane(__double2hiint(qtop[E - 1] * Ed));
    search_tail<R>(key, acc, Ed, rem, n, diag, qrow, lane, warm, hint, cold, slope, pk, prem, 0u, hi, n, 0, false, false, false, 0u,
                   stats);
}

template <int R>
__device__ __forceinline__ void apportion_row(int32_t E, int n, int diag, const double* __restrict__ qrow,
                                              uint32_t (&acc)[R], int lane, Warm* warm,
                                              const double* __restrict__ qtop = nullptr,
                                              unsigned long long* stats_in = nullptr) {
#ifdef EPI_SEARCH_STATS
    unsigned long long* stats = stats_in;   // search statistics are compiled in on request only: they cost registers
#else
    (void)stats_in;
    constexpr unsigned long long* stats = nullptr;
#endif
    const double Ed = (double)E;
    uint32_t key[R];
    uint32_t bsum = 0;
    {
        double q[R];
#pragma unroll
        for (int r = 0; r < R; ++r) q[r] = qrow[r * 64 + lane];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double raw = q[r] * Ed;
            const uint32_t b = (uint32_t)(int32_t)raw;  // raw >= 0: truncation is floor
            key[r] = (uint32_t)__double2hiint(frac_of(raw));
            acc[r] += b;
            bsum += b;
        }
    }
    search_from_keys<R>(key, acc, E, wave_sum(bsum), n, diag, qrow, lane, warm, qtop, stats);
}

// apportion_row started from a histogram of the fractions.  `hist`: 64 * R zeroed LDS words owned by the calling wave
// (zero again on return).
//
// A fraction's bucket is floor(value * NB) of the double whose high word its key is, NB = 64 * R: monotone in the key,
// bucket edges at the keys of b / NB (whose low words are zero).  Lane l reads back buckets R*l .. R*l + R - 1, so a
// prefix sum over the lanes counts the smaller fractions; the bucket that holds the rem-th largest one and the counts
// on either side of it come out of one scatter, one prefix scan and one ballot -- no data-dependent loop and hardly
// any scalar bookkeeping.  When the bucket's lower edge already separates exactly rem fractions (about two times in
// three when there is one bucket per column) that edge is the threshold; otherwise the search continues inside the
// bucket, which holds a handful of fractions.  Measured on the 200-country grid against the warm-started search:
// 263 -> 201 vector and 290 -> 95 scalar instructions per row.  (Running the S and the I split of a row side by side
// through this code was tried: more instructions -- 214 / 150 -- and no gain.)
template <int R>
__device__ __forceinline__ void apportion_row_hist(int32_t E, int n, int diag, const double* __restrict__ qrow,
                                                   uint32_t (&acc)[R], int lane, uint32_t* hist,
                                                   unsigned long long* stats_in = nullptr) {
#ifdef EPI_SEARCH_STATS
    unsigned long long* stats = stats_in;
#else
    (void)stats_in;
    constexpr unsigned long long* stats = nullptr;
#endif
    constexpr int NB = 64 * R;
    const double Ed = (double)E;
    uint32_t key[R];
    double fr[R];   // the fractions themselves: their bucket is floor(fraction * NB), which only the high word decides
    uint32_t bsum = 0;
    {
        double q[R];
#pragma unroll
        for (int r = 0; r < R; ++r) q[r] = qrow[r * 64 + lane];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double raw = q[r] * Ed;
            const uint32_t b = (uint32_t)(int32_t)raw;  // raw >= 0: truncation is floor
            fr[r] = frac_of(raw);
            key[r] = (uint32_t)__double2hiint(fr[r]);
            acc[r] += b;
            bsum += b;
        }
    }
    const uint32_t base_total = wave_sum(bsum);
    const int32_t rem = (int32_t)((uint32_t)E - base_total);
    if (stats && lane == 0) atomicAdd(&stats[0], 1ull);
    if (rem <= 0) return;
    if (rem >= n) {  // every real column gets a unit (the reference stops at frac.length)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = r * 64 + lane;
            acc[r] += (j < n && j != diag);
        }
        return;
    }
    // padding columns (weight 0, fraction 0) stay out: dozens of lanes adding to one word would queue up behind each other
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (r * 64 + lane < n) atomicAdd(&hist[(uint32_t)(fr[r] * (double)NB)], 1u);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t c[R];
#pragma unroll
    for (int k = 0; k < R; ++k) c[k] = hist[R * lane + k];
#pragma unroll
    for (int k = 0; k < R; ++k) hist[R * lane + k] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int k = 1; k < R; ++k) c[k] += c[k - 1];
    const uint32_t P = wave_prefix_sum(c[R - 1]);     // fractions in this lane's buckets and all lower ones
    const uint32_t e = P - c[R - 1];                  // ... in all lower lanes' buckets
    // #{bucket >= B} >= rem  <=>  #{bucket < B} <= n - rem: the highest bucket B with that property holds the rem-th largest
    const uint32_t T = (uint32_t)(n - rem);
    const unsigned long long reach = __ballot(e <= T);   // never empty: lane 0 has e = 0
    const int L = __builtin_amdgcn_readfirstlane(63 - (int)__builtin_clzll(reach | 1ull));
    int kidx, below;   // bucket R*L + kidx; fractions in lower buckets; `inside` = fractions in the bucket itself
    int inside;
    if constexpr (R <= 4) {
        uint32_t kk = 0u, vbelow = e, vnext = P;
#pragma unroll
        for (int k = 0; k < R - 1; ++k) {
            const uint32_t ck = e + c[k];            // #{bucket <= R*lane + k}
            const bool ok = ck <= T;                 // bucket R*lane + k + 1 still qualifies
            kk += ok;
            vbelow = ok ? ck : vbelow;
        }
#pragma unroll
        for (int k = R - 2; k >= 0; --k) {
            const uint32_t ck = e + c[k];
            vnext = (ck > T) ? ck : vnext;           // smallest cumulative count above T: #{bucket <= B}
        }
        kidx = __builtin_amdgcn_readlane((int)kk, L);
        below = __builtin_amdgcn_readlane((int)vbelow, L);
        inside = __builtin_amdgcn_readlane((int)vnext, L) - below;
    } else {
        // wide rows: the walk over the lane's R buckets runs on the scalar unit
        const int base = __builtin_amdgcn_readlane((int)e, L);
        kidx = 0;
        below = base;
        int next = base + __builtin_amdgcn_readlane((int)c[R - 1], L);
#pragma unroll
        for (int k = R - 2; k >= 0; --k) {
            const int ck = base + __builtin_amdgcn_readlane((int)c[k], L);
            if (ck > (int)T) next = ck;
            if (ck <= (int)T && kidx == 0) { kidx = k + 1; below = ck; }
        }
        inside = next - below;
    }
    const int bucket = R * L + kidx;
    const int chi = n - below - inside, clo = n - below;   // #{key >= upper edge}, #{key >= lower edge}
    const uint32_t lo = bucket == 0 ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane(__double2hiint((double)bucket * (1.0 / NB)));
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane(__double2hiint((double)(bucket + 1) * (1.0 / NB)));
    search_tail<R>(key, acc, Ed, rem, n, diag, qrow, lane, nullptr, 0u, false, 0.f, 0u, 0, lo, hi, clo, chi, true, true,
                   clo == rem && bucket != 0, lo, stats);
}

}  // namespace dev
#endif  // __HIPCC__
}  // namespace epi
