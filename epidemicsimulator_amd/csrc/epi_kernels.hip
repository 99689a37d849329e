// HIP kernels (gfx950) for the metapopulation step and the calibration sweep.
// Compiled with -ffp-contract=off: see epi_device.hpp.
#include "epi_kernels.hpp"

#include <atomic>
#include "sort_net.hpp"

// Phase timers of the sparse kernel: compiled in with -DEPI_CSR_TIMERS (they cost registers), read with EPI_STATS=1.
#ifdef EPI_CSR_TIMERS
#define EPI_TM(...) __VA_ARGS__
#else
#define EPI_TM(...)
#endif

namespace epi {
using namespace dev;

template <class T>
__device__ __forceinline__ void keep_loaded(T& x) {
    asm volatile("" : "+v"(x));
}

// WorldToy.scala:115: min(X.toLong, floor(X * sumW + 1e-9).toLong).toInt for X >= 0 and finite sumW >= 0.
// e is a non-negative integer-valued double: when it reaches X the minimum is X, otherwise it is below
// 2^31 and converts exactly (no 64-bit conversion needed).
__device__ __forceinline__ int32_t export_count(int32_t X, double sumW) {
    const double e = floor((double)X * sumW + 1e-9);
    return e >= (double)X ? X : (int32_t)e;
}

__device__ __forceinline__ void record_error(int* sh_err, int code, int node, int t) {
    if (atomicCAS(&sh_err[0], 0, code) == 0) {
        sh_err[1] = node;
        sh_err[2] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// Fused dense episode kernel.  One workgroup = one episode (one world instance); thread c owns
// country c; wave w apportions rows of the connection matrix.  Per step (WorldToy.stepOne,
// WorldToy.scala:83-145):
//   A  local dynamics per country (env.step), exports per row, local mass check
//   B  mobility: rows are handed out dynamically to waves; each wave keeps the normalised row and
//      the per-column allocations in registers (apportion_row) and accumulates column sums over
//      all rows it processes; the wave totals are merged with LDS atomics
//   C  apply inflow - export, advance t, arrival/peak metrics, world mass sums
// The world mass totals of step t are compared by thread 0 during phase B of step t+1.
// ---------------------------------------------------------------------------------------------
// R = register slots per lane for one row (64 * R padded countries); W = waves per workgroup.  Threads below
// 64 * R own one country each in the local and apply phases; every wave takes rows in the mobility phase, so
// small worlds, whose episodes would otherwise run on one to four waves, get W >= R waves to hide the latency of
// the search's dependent steps.
#ifndef EPI_SMALL_WPE
#define EPI_SMALL_WPE 8
#endif
#ifndef EPI_KW4
#define EPI_KW4 16
#endif
#ifndef EPI_CSR_KARGS
#define EPI_CSR_KARGS 1
#endif
#ifndef EPI_HIST_MAXR
#define EPI_HIST_MAXR 8
#endif
template <int R, int W>
__global__ __launch_bounds__(W * 64) __attribute__((amdgpu_waves_per_eu(W > R ? EPI_SMALL_WPE : 1)))  /* small worlds: 64 registers, twice the resident waves */ void episode_dense_kernel(const EpisodeArgs a) {
    constexpr int NP = R * 64;
    static_assert(W >= R, "every padded country needs a thread");
    __shared__ int32_t sh_s[NP], sh_i[NP], sh_r[NP], sh_d[NP], sh_v[NP];
    __shared__ double sh_cap[NP];
    __shared__ int32_t sh_expS[NP], sh_expI[NP];
    __shared__ uint32_t sh_movS[NP], sh_movI[NP];
    __shared__ unsigned long long sh_mass[2][2];
    __shared__ int sh_next_row;
    __shared__ int sh_err[4];
    // threshold search: rows of up to 512 columns start from a histogram of their fractions (apportion_row_hist; one
    // zeroed block of 64 * R words per wave); wider rows remember the previous day's threshold per (compartment, row)
    // and search from there.  Measured, country-steps/s with / without the histogram: 200 countries 2.79e9 / 1.66e9,
    // 400: 1.71e9 / 1.47e9, 500: 1.60e9 / 1.49e9, 1,000: 7.0e8 / 8.6e8 (16 LDS atomics and a scalar walk over 16
    // buckets per split cost more than the probes they save).
    constexpr bool kHist = R <= EPI_HIST_MAXR;
    __shared__ Warm sh_warm[kHist ? 1 : 2][kHist ? 1 : NP];
    __shared__ uint32_t sh_hist[kHist ? W : 1][kHist ? NP : 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int e = blockIdx.x;
    const int g = a.g_begin + e * a.g_stride;
    const int n = a.n;
    const int ti = a.n_travel > 0 ? g % a.n_travel : 0;
    int bi = a.n_travel > 0 ? g / a.n_travel : g;
    bi = bi < a.beta_rows ? bi : a.beta_rows - 1;
    const size_t eoff = (size_t)e * NP;
    const size_t plane = (size_t)a.count * NP;
    const bool real = tid < n;
    const bool own = W == R || tid < NP;  // this thread holds a (padded) country

    int32_t tcur = 0, arrival = 0, peak = 0;
    if (own) {
        sh_s[tid] = a.st[0 * plane + eoff + tid];
        sh_i[tid] = a.st[1 * plane + eoff + tid];
        sh_r[tid] = a.st[2 * plane + eoff + tid];
        sh_d[tid] = a.st[3 * plane + eoff + tid];
        sh_v[tid] = a.st[4 * plane + eoff + tid];
        sh_cap[tid] = a.cap[eoff + tid];
        tcur = a.tcur[eoff + tid];
        arrival = a.arrival[eoff + tid];
        peak = a.peak[eoff + tid];
        if constexpr (!kHist) {
            sh_warm[0][tid].rho = 0.f;
            sh_warm[1][tid].rho = 0.f;
        }
    }
    if constexpr (kHist)
        for (int k = tid; k < W * NP; k += W * 64) (&sh_hist[0][0])[k] = 0u;
    int32_t last_flow_s = 0, last_flow_i = 0;
    if (tid == 0) {
        sh_mass[0][0] = sh_mass[0][1] = sh_mass[1][0] = sh_mass[1][1] = 0ull;
        sh_err[0] = sh_err[1] = sh_err[2] = sh_err[3] = 0;
    }
    const double my_sumw = real ? ((tid == a.seed_idx && a.qseed) ? a.sumw_seed[ti] : a.sumw[tid]) : 0.0;
    __syncthreads();

#ifdef EPI_DENSE_TIMERS
    long long tmr[6] = {0, 0, 0, 0, 0, 0};   // per wave: phase A, barrier, phase B rows, B merge + barrier, phase C, rows taken
#define EPI_DT(i) { const long long k1_ = clock64(); tmr[i] += k1_ - k0_; k0_ = k1_; }
#else
#define EPI_DT(i)
#endif
    for (int t = a.t_begin; t < a.t_end; ++t) {
#ifdef EPI_DENSE_TIMERS
        long long k0_ = clock64();
#endif
        // ------------------------------------------------------------------ phase A
        if (tid == 0) sh_next_row = 0;
        if (own) {
            sh_movS[tid] = 0u;
            sh_movI[tid] = 0u;
        }
        int32_t eS = 0, eI = 0;
        unsigned long long mass_local = 0ull;
        if (real) {
            // the index is made opaque so that the addresses are formed here, every step: precomputed ahead
            // of the step loop they are kept in scratch memory and each load waits for its address first
            int ix = tid;
            asm volatile("" : "+v"(ix));
            StepConsts c;
            c.beta0 = a.beta0[(size_t)bi * NP + ix];
            c.gamma = a.cst[CST_GAMMA * NP + ix];
            c.ifr0 = a.cst[CST_IFR0 * NP + ix];
            c.cnoise = a.cst[CST_CNOISE * NP + ix];
            c.eff1 = a.cst[CST_EFF1 * NP + ix];
            c.eff2 = a.cst[CST_EFF2 * NP + ix];
            c.eff3 = a.cst[CST_EFF3 * NP + ix];
            c.surge = a.cst[CST_SURGE * NP + ix];
            c.vax6 = a.cst[CST_VAX6 * NP + ix];
            c.vax7 = a.cst[CST_VAX7 * NP + ix];
            int act = a.act_episode ? (int)a.act_episode[eoff + ix] : (int)a.actions[(size_t)(t - a.act_t0) * NP + ix];
            act = act < 0 ? 0 : (act > 7 ? 7 : act);  // Action.fromId clamps (env.scala:18-20)
            const double z = a.noise_episode ? a.noise_episode[eoff + ix] : a.noise[(size_t)(t - a.noise_t0) * NP + ix];
            // all constants in one batch of loads: those that only some actions use would otherwise be fetched
            // inside the action branches, one exposed memory latency each
            keep_loaded(c.eff1); keep_loaded(c.eff2); keep_loaded(c.eff3); keep_loaded(c.surge);
            keep_loaded(c.vax6); keep_loaded(c.vax7); keep_loaded(c.ifr0); keep_loaded(c.cnoise);
            const int32_t s0 = sh_s[tid], i0 = sh_i[tid], r0 = sh_r[tid], d0 = sh_d[tid], v0 = sh_v[tid];
            const double cap0 = sh_cap[tid];
            const long long tb = (long long)s0 + i0 + r0 + d0 + v0;
            const LocalOut o = local_step(c, s0, i0, r0, d0, v0, cap0, act, z);
            const long long ta = (long long)o.s + o.i + o.r + o.d + o.v;
            if (a.validate && ta != tb) record_error(sh_err, EPI_ERR_LOCAL_MASS, tid, tcur);
            if ((o.s | o.i | o.r | o.d | o.v) < 0) record_error(sh_err, EPI_ERR_INT_RANGE, tid, tcur);
            if (a.rwd) a.reward_out[eoff + tid] = reward_of(a.rwd[tid], (double)i0, (double)d0, o, act);
            sh_s[tid] = o.s; sh_i[tid] = o.i; sh_r[tid] = o.r; sh_d[tid] = o.d; sh_v[tid] = o.v;
            sh_cap[tid] = o.cap;
            if (a.loc_s) { a.loc_s[eoff + tid] = o.s; a.loc_i[eoff + tid] = o.i; }
            tcur += 1;  // env.scala:132
            eS = export_count(max(0, o.s), my_sumw);
            eI = export_count(max(0, o.i), my_sumw);
            mass_local = ta > 0 ? (unsigned long long)ta : 0ull;
        }
        if (own) {
            sh_expS[tid] = eS;
            sh_expI[tid] = eI;
        }
        if (a.validate && own) {
            const unsigned long long wsum = wave_sum_u64(mass_local);
            if (lane == 0) atomicAdd(&sh_mass[t & 1][0], wsum);
        }
        EPI_DT(0)
        __syncthreads();
        EPI_DT(1)

        // ------------------------------------------------------------------ phase B
        if (tid == 0 && a.validate && t > a.t_begin) {
            const int p = (t - 1) & 1;
            if (sh_mass[p][0] != sh_mass[p][1]) record_error(sh_err, EPI_ERR_WORLD_MASS, -1, t - 1);
            sh_mass[p][0] = 0ull;
            sh_mass[p][1] = 0ull;
        }
        {
            uint32_t accS[R], accI[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { accS[r] = 0u; accI[r] = 0u; }
            bool any = false;
            for (;;) {
                int row = 0;
                if (lane == 0) row = atomicAdd(&sh_next_row, 1);
                row = __builtin_amdgcn_readfirstlane(row);
                if (row >= n) break;
                const int32_t ES = __builtin_amdgcn_readfirstlane(sh_expS[row]);
                const int32_t EI = __builtin_amdgcn_readfirstlane(sh_expI[row]);
                if ((ES | EI) == 0) continue;
                const double* __restrict__ qrow =
                    (row == a.seed_idx && a.qseed) ? a.qseed + (size_t)ti * NP : a.q + (size_t)row * NP;
                if constexpr (kHist) {
                    uint32_t* hist = &sh_hist[tid >> 6][0];
                    if (ES > 0) apportion_row_hist<R>(ES, n, row, qrow, accS, lane, hist, a.stats);
                    if (EI > 0) apportion_row_hist<R>(EI, n, row, qrow, accI, lane, hist, a.stats ? a.stats + 32 : nullptr);
                } else {
                    const double* __restrict__ qtop =
                        !a.qtop ? nullptr
                                : ((row == a.seed_idx && a.qseed) ? a.qtop_seed + (size_t)ti * kTopWeights
                                                                  : a.qtop + (size_t)row * kTopWeights);
                    if (ES > 0) apportion_row<R>(ES, n, row, qrow, accS, lane, &sh_warm[0][row], qtop, a.stats);
                    if (EI > 0)
                        apportion_row<R>(EI, n, row, qrow, accI, lane, &sh_warm[1][row], qtop, a.stats ? a.stats + 32 : nullptr);
                }
                any = true;
#ifdef EPI_DENSE_TIMERS
                tmr[5] += 1;
#endif
            }
            EPI_DT(2)
            if (any) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (accS[r]) atomicAdd(&sh_movS[r * 64 + lane], accS[r]);
                    if (accI[r]) atomicAdd(&sh_movI[r * 64 + lane], accI[r]);
                }
            }
        }
        __syncthreads();
        EPI_DT(3)

        // ------------------------------------------------------------------ phase C
        unsigned long long mass_move = 0ull;
        if (real) {
            const int32_t movedS = (int32_t)(sh_movS[tid] - (uint32_t)sh_expS[tid]);
            const int32_t movedI = (int32_t)(sh_movI[tid] - (uint32_t)sh_expI[tid]);
            const double ns = fmax(0.0, (double)sh_s[tid] + (double)movedS);  // WorldToy.scala:133
            const double ni = fmax(0.0, (double)sh_i[tid] + (double)movedI);  // :134
            if (ns > 2147483647.0 || ni > 2147483647.0) record_error(sh_err, EPI_ERR_INT_RANGE, tid, tcur);
            const int32_t s3 = (int32_t)ns, i3 = (int32_t)ni;
            sh_s[tid] = s3;
            sh_i[tid] = i3;
            tcur = min(tcur + 1, a.step_t_max);  // :135
            if (arrival < 0 && i3 >= 10) arrival = t;  // Validation.arrivalTime, thresh 10.0
            peak = max(peak, i3);                      // Validation.peakI
            last_flow_s = movedS;
            last_flow_i = movedI;
            if (a.series_i) a.series_i[((size_t)(t - a.series_t0) * a.count + e) * n + tid] = (double)i3;
            const long long tm = (long long)s3 + i3 + sh_r[tid] + sh_d[tid] + sh_v[tid];
            mass_move = tm > 0 ? (unsigned long long)tm : 0ull;
        }
        if (a.validate && own) {
            const unsigned long long wsum = wave_sum_u64(mass_move);
            if (lane == 0) atomicAdd(&sh_mass[t & 1][1], wsum);
        }
        // No barrier here: phase A of the next step touches only this thread's own slots, and every
        // wave has left the row loop before the barrier above.
        EPI_DT(4)
    }
#ifdef EPI_DENSE_TIMERS
    if (a.stats && lane == 0)
        for (int i = 0; i < 6; ++i) atomicAdd(&a.stats[48 + i], (unsigned long long)tmr[i]);
#endif
    __syncthreads();
    if (tid == 0) {
        if (a.validate && a.t_end > a.t_begin) {
            const int p = (a.t_end - 1) & 1;
            if (sh_mass[p][0] != sh_mass[p][1]) record_error(sh_err, EPI_ERR_WORLD_MASS, -1, a.t_end - 1);
        }
        if (sh_err[0] != 0 && a.err[e * 4] == 0) {
            a.err[e * 4 + 1] = sh_err[1];
            a.err[e * 4 + 2] = sh_err[2];
            a.err[e * 4] = sh_err[0];
        }
    }
    if (!own) return;
    a.st[0 * plane + eoff + tid] = sh_s[tid];
    a.st[1 * plane + eoff + tid] = sh_i[tid];
    a.st[2 * plane + eoff + tid] = sh_r[tid];
    a.st[3 * plane + eoff + tid] = sh_d[tid];
    a.st[4 * plane + eoff + tid] = sh_v[tid];
    a.cap[eoff + tid] = sh_cap[tid];
    a.tcur[eoff + tid] = tcur;
    a.arrival[eoff + tid] = arrival;
    a.peak[eoff + tid] = peak;
    if (a.flow_s) {
        a.flow_s[eoff + tid] = last_flow_s;
        a.flow_i[eoff + tid] = last_flow_i;
    }
}

// ---------------------------------------------------------------------------------------------
// Sparse episode kernel for graphs too wide for the dense kernel.  One workgroup of kCsrThreads
// threads = one episode; compartments live in HBM and are streamed once per step; the two inflow
// accumulators movedS/movedI for all n patches live in LDS (8 bytes per patch).
//
// Mobility (WorldToy.scala:106-126): ONE LANE owns one row.  Rows (sorted by length) are stored in
// blocks of 64 in a slot-major (ELL) layout -- slot k of the 64 rows of a block is contiguous -- so that
// the 64 lanes of a wave read their k-th weights with one coalesced load.  Every lane runs the same
// largest-remainder split as the dense kernel on its own short row, entirely in its registers:
// high words of the fractions as keys, the threshold K = the rem-th largest key taken from a copy
// sorted by a sorting network, ties in the high word finished exactly on the full fractions in
// ascending column order.
//
// Entries that are not stored have weight 0, hence raw = 0, base = 0 and fraction 0: they could only
// receive a unit after every positive fraction has one.  For finite non-negative weights
// rem = E - sum(base) never exceeds the number of positive fractions (rem equals the sum of the
// fractions up to a rounding error far below 1, and every fraction is below 1), so the units go
// to stored entries only; the kernel checks this and reports EPI_ERR_UNSUPPORTED otherwise.
// ---------------------------------------------------------------------------------------------
// Inflow accumulators of the sparse kernel: one 8-byte LDS cell per patch, susceptible inflow in the low word and
// infectious inflow in the high word.  The lane-per-row routine adds both with ONE 64-bit LDS atomic per stored
// weight (a carry out of the low word would take an inflow of 2^32 people into one patch, which the world mass
// check reports either way); the wave-per-row routines add to one word at a time through this view.
struct InflowWords {
    uint32_t* p;   // word 0 of cell 0
    __device__ __forceinline__ uint32_t& operator[](int c) const { return p[2 * c]; }
};

// 16 waves, 4 per SIMD, within 128 registers per lane.  The 32-slot mobility routine holds the keys of one compartment at
// a time to stay inside that budget (it serves few blocks: rows are sorted by length); the 16- and 24-slot routines keep
// the keys of both compartments side by side.  Measured on the 20,000-patch sweep: 768 threads / 153 registers 16.25 ms
// per launch; 1,024 threads with every routine in two passes 15.9 ms; with only the 32-slot routine in two passes 15.3 ms
// (the fourth wave per SIMD shortens the local / apply pass by 17 %, the mobility phase not at all).
#ifndef EPI_CSR_THREADS
#define EPI_CSR_THREADS 1024
#endif
// Sorting networks for 8, 10, ..., 24, 28 and 32 keys (1) or for 12, 16, 20, 24 and 32 only (0).  A block's rows are of
// nearly one length (rows are sorted), and Batcher's network for 18 keys has 90 comparators against 132 for 24: measured on
// the 20,000-patch sweep, degrees ~ Poisson(16): 16 / 24 / 32 only 0.747 of the roofline, five sizes 0.777, eleven 0.81
// (same box, same call); one network per length up to 24 is 1 % slower again (code size).
#ifndef EPI_CSR_ACT_CONSTS
#define EPI_CSR_ACT_CONSTS 1
#endif
#ifndef EPI_CSR_CAP_IF_CHANGED
#define EPI_CSR_CAP_IF_CHANGED 1
#endif
#ifndef EPI_CSR_KEEP_MAX
#define EPI_CSR_KEEP_MAX 18
#endif
#ifndef EPI_CSR_GRAB_AHEAD
#define EPI_CSR_GRAB_AHEAD 1
#endif
#ifndef EPI_CSR_MID_NETS
#define EPI_CSR_MID_NETS 1
#endif
#ifndef EPI_CSR_TWO_PASS
#define EPI_CSR_TWO_PASS 32   // routines of at least this many slots form the S and the I keys one after the other (0 = none)
#endif
constexpr int kCsrThreads = EPI_CSR_THREADS;

// Slots are read in groups of kSlotGroup: a group is skipped when the block has no such slots, and all
// loads of the groups that are read are issued before the first one is consumed.  Every block's slot
// count is rounded up to whole groups in the ELL arrays, the extra slots holding weight 0.

// Address of slot k0 of a lane's row, formed where it is used: the empty asm keeps the compiler from
// precomputing one 64-bit address per slot ahead of the loops, which costs two registers per slot;
// the slots of a group are then reached with immediate offsets.
template <class T, int STRIDE = 64>
__device__ __forceinline__ const T* group_ptr(const T* p, int k0) {
    int off = k0 * STRIDE;
    asm volatile("" : "+v"(off));
    return p + off;
}

// Threshold of one compartment of one lane's row: the rem largest of the keys (high words of the
// fractions of q * E) are those with key >= Kx, plus the slots whose bit is set in `picked` (ties in
// the high word of the rem-th largest key, finished on the full fractions by (fraction descending,
// column ascending)).  `srt` holds the keys on entry and is consumed; the keys are cheap to form again
// from the weights, which is what the rare tie path and the scatter pass do instead of keeping them.
struct LaneThreshold {
    uint32_t Kx, picked;
    bool bad;  // more units left over than positive fractions: the graph sends units to entries that are not stored
};

// STRIDE: distance in elements between consecutive weights of the lane's row (64 in the slot-major
// layout of sparse graphs, 1 for a plain row).
// STORED_ONLY: the row is a sparse one, whose units must all land on stored entries (`bad` otherwise).
template <int DMAX, int STRIDE = 64, bool STORED_ONLY = true>
__device__ __forceinline__ LaneThreshold lane_threshold(uint32_t (&srt)[DMAX], int32_t rem, double Ed, int D,
                                                        const double* __restrict__ qp) {
    // K = the rem-th largest key, read from the sorted keys (sorting network: no data-dependent control
    // flow, so 64 different rows cost the same as one); `nxt` = the key ranked right below it
    sort_desc<DMAX>(srt);
    uint32_t K = 0u, nxt = 0u;
#pragma unroll
    for (int j = 0; j < DMAX; ++j) {
        const bool hit = (j == rem - 1);
        K = hit ? srt[j] : K;
        if (j + 1 < DMAX) nxt = hit ? srt[j + 1] : nxt;
    }
    // A unit may land on stored entries only: rem must not exceed the number of positive fractions, or else the
    // rem-th largest fraction is 0.  Its key is then 0 as well; a key of 0 with a positive fraction (below 2^-1022)
    // takes an absurd graph, so the positive fractions are counted exactly in that case only.
    LaneThreshold r;
    r.picked = 0u;
    r.bad = false;
    if (STORED_ONLY && __any(rem > 0 && K == 0u)) {
        int pos = 0;
#pragma unroll
        for (int k0 = 0; k0 < DMAX; k0 += kSlotGroup) {
            if (k0 < D) {
                const double* qg = group_ptr<double, STRIDE>(qp, k0);
#pragma unroll
                for (int j = 0; j < kSlotGroup; ++j) pos += (frac_of(qg[j * STRIDE] * Ed) > 0.0);  // slots past D hold weight 0
            }
        }
        r.bad = rem > pos;
        if (r.bad) rem = 0;
    }
    // ranks rem and rem+1 differ in the high word: every key >= K is selected.  Otherwise m = rem - #{key > K}
    // members of the tie are picked one at a time; they are few unless weights repeat
    const bool found = rem > 0 && (rem == DMAX || nxt != K);
    const bool tie = rem > 0 && !found;
    if (__any(tie)) {
        int above = 0;
#pragma unroll
        for (int j = 0; j < DMAX; ++j) above += (srt[j] > K);
        int m = tie ? rem - above : 0;
        while (__any(m > 0)) {
            double best = -1.0;
            int best_k = -1;
#pragma unroll
            for (int k0 = 0; k0 < DMAX; k0 += kSlotGroup) {
                if (k0 < D) {
                    const double* qg = group_ptr<double, STRIDE>(qp, k0);
#pragma unroll
                    for (int j = 0; j < kSlotGroup; ++j) {
                        const int k = k0 + j;
                        const double fr = frac_of(qg[j * STRIDE] * Ed);
                        const bool cand = k < D && m > 0 && (uint32_t)__double2hiint(fr) == K && !((r.picked >> k) & 1u) && fr > best;
                        if (cand) { best = fr; best_k = k; }
                    }
                }
            }
            if (m > 0 && best_k >= 0) r.picked |= 1u << best_k;
            m -= 1;
        }
    }
    // keys are high words of fractions below 1, far from 2^32 - 1: the sentinel selects nothing
    r.Kx = found ? K : (tie ? K + 1u : 0xffffffffu);
    return r;
}

// Mobility split of one block of 64 rows, one lane per row, S and I in one sweep over the block's weights
// (each weight is read twice per step: once for the keys, once with its column for the scatter).
// A lane that exports nothing in a compartment runs along with E = 0: every product is 0, nothing is selected.
// NS: the block's rows hold at most NS <= DMAX weights (the slots from NS on hold weight 0 and are neither read nor sorted).
template <int DMAX, int NS = DMAX>
__device__ __forceinline__ void lane_apportion(int32_t ES, int32_t EI, bool anyI, int D, const double* __restrict__ qp,
                                               const int32_t* __restrict__ cp, int row, unsigned long long* sh_mov,
                                               int* sh_err, int t, long long* tm) {
    static_assert(DMAX % kSlotGroup == 0, "whole slot groups");
    // slots are worked on in chunks of whole groups: all loads of a chunk are in flight together
    constexpr int HALF = (DMAX % (2 * kSlotGroup) == 0) ? DMAX / 2 : kSlotGroup;
    const double EdS = (double)ES, EdI = (double)EI;
    EPI_TM(long long c0 = clock64();)
    static_assert(NS <= DMAX && NS > DMAX - kSlotGroup, "NS lies in the last slot group");
    constexpr bool kTwoPass = EPI_CSR_TWO_PASS > 0 && DMAX >= EPI_CSR_TWO_PASS;
    LaneThreshold thS, thI;
    thI.Kx = 0xffffffffu;
    thI.picked = 0u;
    thI.bad = false;
    // Rows of up to EPI_CSR_KEEP_MAX weights keep their whole units and keys in registers for the scatter pass (4 NS
    // registers) instead of reading the weights and forming both again.  Measured on the 20,000-patch sweep in one call:
    // none 0.813, up to 12 weights 0.818, up to 16 0.845, up to 18 0.851 of the roofline; 20 spills 37 registers: 0.74.
    constexpr bool kKeep = !kTwoPass && NS <= EPI_CSR_KEEP_MAX;
    constexpr int NK = kKeep ? NS : 1;
    uint32_t kS[NK], kI[NK], bS[NK], bI[NK];
    if constexpr (kTwoPass) {
    // One compartment at a time: the keys of one compartment (DMAX registers) instead of two live side by side, at
    // the price of a third read of the block's weights in blocks that export infectious people.
    auto threshold_of = [&](double Ed, int32_t E) {
        uint32_t key[NS];
        uint32_t bsum = 0u;
#pragma unroll
        for (int h0 = 0; h0 < DMAX; h0 += HALF) {
            double qv[HALF];
#pragma unroll
            for (int k0 = 0; k0 < HALF; k0 += kSlotGroup) {
#pragma unroll
                for (int j = 0; j < kSlotGroup; ++j) qv[k0 + j] = 0.0;
                if (h0 + k0 < D) {
                    const double* qg = group_ptr(qp, h0 + k0);
#pragma unroll
                    for (int j = 0; j < kSlotGroup; ++j)
                        if (h0 + k0 + j < NS) qv[k0 + j] = qg[j * 64];
                }
            }
#pragma unroll
            for (int j = 0; j < HALF; ++j) {
                if (h0 + j >= NS) continue;
                const double raw = qv[j] * Ed;  // slots past D of a group that is read hold weight 0
                bsum += (uint32_t)(int32_t)raw;
                key[h0 + j] = (uint32_t)__double2hiint(frac_of(raw));
            }
        }
        const int32_t rem = (int32_t)((uint32_t)E - bsum);
        return lane_threshold<NS>(key, rem, Ed, D, qp);
    };
    thS = threshold_of(EdS, ES);
    if (anyI) thI = threshold_of(EdI, EI);
    } else {
    uint32_t keyS[NS], keyI[NS];
    uint32_t bsumS = 0u, bsumI = 0u;
#pragma unroll
    for (int h0 = 0; h0 < DMAX; h0 += HALF) {
        double qv[HALF];
#pragma unroll
        for (int k0 = 0; k0 < HALF; k0 += kSlotGroup) {
#pragma unroll
            for (int j = 0; j < kSlotGroup; ++j) qv[k0 + j] = 0.0;
            if (h0 + k0 < D) {
                const double* qg = group_ptr(qp, h0 + k0);
#pragma unroll
                for (int j = 0; j < kSlotGroup; ++j)
                    if (h0 + k0 + j < NS) qv[k0 + j] = qg[j * 64];
            }
        }
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            const int k = h0 + j;
            if (k >= NS) continue;
            const double rawS = qv[j] * EdS;  // slots past D of a group that is read hold weight 0
            bsumS += (uint32_t)(int32_t)rawS;
            keyS[k] = (uint32_t)__double2hiint(frac_of(rawS));
            const double rawI = qv[j] * EdI;
            bsumI += (uint32_t)(int32_t)rawI;
            keyI[k] = (uint32_t)__double2hiint(frac_of(rawI));
            if constexpr (kKeep) {
                bS[k] = (uint32_t)(int32_t)rawS; kS[k] = keyS[k];
                bI[k] = (uint32_t)(int32_t)rawI; kI[k] = keyI[k];
            }
        }
    }
    const int32_t remS = (int32_t)((uint32_t)ES - bsumS), remI = (int32_t)((uint32_t)EI - bsumI);
    EPI_TM(asm volatile("" ::"v"(remS), "v"(remI)); { const long long c1 = clock64(); tm[0] += c1 - c0; c0 = c1; })
    thS = lane_threshold<NS>(keyS, remS, EdS, D, qp);
    if (anyI) thI = lane_threshold<NS>(keyI, remI, EdI, D, qp);
    }
    if (thS.bad || thI.bad) record_error(sh_err, EPI_ERR_UNSUPPORTED, row, t);
    EPI_TM(asm volatile("" ::"v"(thS.Kx), "v"(thI.Kx)); { const long long c1 = clock64(); tm[1] += c1 - c0; c0 = c1; })
    // A unit lands only on stored off-diagonal entries: padding and a stored diagonal hold weight 0, hence
    // base 0 and fraction 0, and fractions that are 0 are never selected (rem <= #positive fractions).
    // Ties in a high word are rare: the pass that also looks at the picked bits runs only when a lane has some.
    auto scatter = [&](auto with_picked) {
        constexpr bool kPicked = decltype(with_picked)::value;
#pragma unroll
        for (int h0 = 0; h0 < DMAX; h0 += HALF) {
            if (h0 < D) {
                double qv[HALF];
                int cv[HALF];
#pragma unroll
                for (int k0 = 0; k0 < HALF; k0 += kSlotGroup) {
#pragma unroll
                    for (int j = 0; j < kSlotGroup; ++j) { qv[k0 + j] = 0.0; cv[k0 + j] = 0; }
                    if (h0 + k0 < D) {
                        const double* qg = group_ptr(qp, h0 + k0);
                        const int32_t* cg = group_ptr(cp, h0 + k0);
#pragma unroll
                        for (int j = 0; j < kSlotGroup; ++j) {
                            if (h0 + k0 + j >= NS) continue;
                            qv[k0 + j] = qg[j * 64];
                            cv[k0 + j] = cg[j * 64];
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < HALF; ++j) {
                    const int k = h0 + j;
                    if (k >= NS) continue;
                    const double rawS = qv[j] * EdS;
                    uint32_t selS = (uint32_t)((uint32_t)__double2hiint(frac_of(rawS)) >= thS.Kx);
                    if constexpr (kPicked) selS |= (thS.picked >> k) & 1u;
                    const uint32_t allocS = (uint32_t)(int32_t)rawS + selS;
                    const double rawI = qv[j] * EdI;
                    uint32_t selI = (uint32_t)((uint32_t)__double2hiint(frac_of(rawI)) >= thI.Kx);
                    if constexpr (kPicked) selI |= (thI.picked >> k) & 1u;
                    const uint32_t allocI = (uint32_t)(int32_t)rawI + selI;
                    // a share of nothing is added too: testing for it costs more than the LDS operation it would save
                    // (0.811 -> 0.838 of the roofline on the 20,000-patch sweep)
                    atomicAdd(&sh_mov[cv[j]], (unsigned long long)allocS | ((unsigned long long)allocI << 32));
                }
            }
        }
    };
    auto scatter_kept = [&](auto with_picked) {
        constexpr bool kPicked = decltype(with_picked)::value;
        int cv[NS];
#pragma unroll
        for (int k0 = 0; k0 < DMAX; k0 += kSlotGroup) {
#pragma unroll
            for (int j = 0; j < kSlotGroup; ++j)
                if (k0 + j < NS) cv[k0 + j] = 0;
            if (k0 < D) {
                const int32_t* cg = group_ptr(cp, k0);
#pragma unroll
                for (int j = 0; j < kSlotGroup; ++j)
                    if (k0 + j < NS) cv[k0 + j] = cg[j * 64];
            }
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            uint32_t selS = (uint32_t)(kS[k < NK ? k : 0] >= thS.Kx);
            if constexpr (kPicked) selS |= (thS.picked >> k) & 1u;
            uint32_t selI = (uint32_t)(kI[k < NK ? k : 0] >= thI.Kx);
            if constexpr (kPicked) selI |= (thI.picked >> k) & 1u;
            const uint32_t allocS = bS[k < NK ? k : 0] + selS, allocI = bI[k < NK ? k : 0] + selI;
            atomicAdd(&sh_mov[cv[k]], (unsigned long long)allocS | ((unsigned long long)allocI << 32));
        }
    };
    const bool any_picked = __any((thS.picked | thI.picked) != 0u);
    if constexpr (kKeep) {
        if (any_picked) scatter_kept(std::true_type{});
        else scatter_kept(std::false_type{});
    } else {
        if (any_picked) scatter(std::true_type{});
        else scatter(std::false_type{});
    }
    EPI_TM(__builtin_amdgcn_s_waitcnt(0); { const long long c1 = clock64(); tm[2] += c1 - c0; })
}

// A row too long for the register routines (more than 256 stored weights, e.g. a dense graph given for a
// world of more than 1,024 countries): one wave streams the compact row once per counting probe and forms
// the keys again each time.  Same selection as everywhere else -- the rem largest fractions, ties in the
// high word finished on the low word and then in column order -- only slower: this path exists so that no
// finite non-negative graph is refused for its row lengths.
__device__ __noinline__ void wide_row(int32_t ES, int32_t EI, int deg, int row, const double* __restrict__ qrow,
                                      const int32_t* __restrict__ crow, int lane, InflowWords sh_movS, InflowWords sh_movI,
                                      int* sh_err, int t) {
    const int chunks = (deg + 63) >> 6;
#pragma nounroll
    for (int comp = 0; comp < 2; ++comp) {
        const int32_t E = comp ? EI : ES;
        if (E <= 0) continue;
        const InflowWords sh_mov = comp ? sh_movI : sh_movS;
        const double Ed = (double)E;
        // pass 1: whole units, largest key, number of positive fractions
        uint32_t bsum = 0u, kmax = 0u, pos = 0u;
        for (int ch = 0; ch < chunks; ++ch) {
            const int j = ch * 64 + lane;
            const double raw = qrow[j] * Ed;  // padding slots hold weight 0
            const double fr = frac_of(raw);
            const uint32_t b = (uint32_t)(int32_t)raw;
            const int col = crow[j];
            if (b != 0u && col >= 0 && col != row) atomicAdd(&sh_mov[col], b);
            bsum += b;
            kmax = max(kmax, (uint32_t)__double2hiint(fr));
            pos += (fr > 0.0);
        }
        const int32_t rem = (int32_t)((uint32_t)E - wave_sum(bsum));
        if (rem <= 0) continue;
        if (rem > (int32_t)wave_sum(pos)) {
            if (lane == 0) record_error(sh_err, EPI_ERR_UNSUPPORTED, row, t);
            continue;
        }
        // threshold search on the high words: #{key >= lo} = clo > rem > chi = #{key >= hi}
        uint32_t lo = 0u, hi = wave_max(kmax) + 1u, K = 0u;
        int clo = chunks * 64, chi = 0, it = 0;
        bool found = false;
        while (hi - lo > 1u) {
            uint32_t mid;
            if (it & 1) {
                mid = lo + ((hi - lo) >> 1);
            } else {
                const float f = ((float)(clo - rem) + 0.5f) * __builtin_amdgcn_rcpf((float)(clo - chi + 1));
                const double vlo = from_hi_word(lo), vhi = from_hi_word(hi);
                mid = (uint32_t)__double2hiint(vlo + (vhi - vlo) * (double)f);
            }
            mid = max(mid, lo + 1u);
            mid = min(mid, hi - 1u);
            mid = (uint32_t)__builtin_amdgcn_readfirstlane((int)mid);
            uint32_t cu = 0u;
            for (int ch = 0; ch < chunks; ++ch)
                cu += ((uint32_t)__double2hiint(frac_of(qrow[ch * 64 + lane] * Ed)) >= mid);
            const int c = (int)wave_sum(cu);
            ++it;
            if (c == rem) { found = true; K = mid; break; }
            if (c > rem) { lo = mid; clo = c; }
            else { hi = mid; chi = c; }
        }
        // tie in the high word `lo`: m = rem - chi members by (low word descending, column ascending)
        unsigned long long Llo = 0ull;
        int m2 = 0;
        if (!found) {
            K = lo;
            const int m = rem - chi;
            uint32_t mn = 0xffffffffu, mx = 0u;
            for (int ch = 0; ch < chunks; ++ch) {
                const int j = ch * 64 + lane;
                const double fr = frac_of(qrow[j] * Ed);
                if (j < deg && (uint32_t)__double2hiint(fr) == K) {
                    mn = min(mn, (uint32_t)__double2loint(fr));
                    mx = max(mx, (uint32_t)__double2loint(fr));
                }
            }
            Llo = wave_min(mn);
            unsigned long long Lhi = (unsigned long long)wave_max(mx) + 1ull;
            int c_hi = 0;
            bool exact = false;
            while (Lhi - Llo > 1ull) {
                const unsigned long long lmid = Llo + ((Lhi - Llo) >> 1);
                uint32_t cu = 0u;
                for (int ch = 0; ch < chunks; ++ch) {
                    const int j = ch * 64 + lane;
                    const double fr = frac_of(qrow[j] * Ed);
                    cu += (j < deg && (uint32_t)__double2hiint(fr) == K && (unsigned long long)(uint32_t)__double2loint(fr) >= lmid);
                }
                const int c = (int)wave_sum(cu);
                if (c == m) { Llo = lmid; exact = true; break; }
                if (c > m) Llo = lmid;
                else { Lhi = lmid; c_hi = c; }
            }
            // exact: members with low word >= Llo are exactly m.  Otherwise fractions identical to the pivot
            // (low word == Llo) share the last m2 units in ascending column order.
            m2 = exact ? 0x7fffffff : m - c_hi;
        }
        // final pass: one unit for every selected column, in column order for the identical fractions
        int run = 0;
        for (int ch = 0; ch < chunks; ++ch) {
            const int j = ch * 64 + lane;
            const double fr = frac_of(qrow[j] * Ed);
            const uint32_t key = (uint32_t)__double2hiint(fr);
            bool take;
            if (found) {
                take = key >= K;
            } else {
                const bool mem = j < deg && key == K;
                const unsigned long long l = (unsigned long long)(uint32_t)__double2loint(fr);
                const bool above = mem && l > Llo;
                const bool eq = mem && l == Llo;
                const unsigned long long mask = __ballot(eq);
                const int pre = (int)lanes_below(mask);
                take = key > K || above || (eq && run + pre < m2);
                run += __popcll(mask);
            }
            const int col = crow[j];
            if (take && j < deg && col >= 0 && col != row) atomicAdd(&sh_mov[col], 1u);
        }
    }
}

// One row longer than a lane holds (up to 64 * R stored weights): the whole wave splits it with the dense
// kernel's routine, S then I, and scatters through the row's column list.
template <int R>
__device__ __noinline__ void long_row(int32_t ES, int32_t EI, int deg, int row, const double* __restrict__ qrow,
                                      const int32_t* __restrict__ crow, int lane, InflowWords sh_movS, InflowWords sh_movI) {
#pragma nounroll
    for (int comp = 0; comp < 2; ++comp) {
        const int32_t E = comp ? EI : ES;
        if (E <= 0) continue;
        uint32_t acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0u;
        apportion_row<R>(E, deg, -1, qrow, acc, lane, nullptr);
        const InflowWords sh_mov = comp ? sh_movI : sh_movS;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int col = crow[r * 64 + lane];
            if (col >= 0 && col != row && acc[r]) atomicAdd(&sh_mov[col], acc[r]);
        }
    }
}

template <int DMAX>
__global__ __launch_bounds__(kCsrThreads) void episode_csr_kernel(const EpisodeArgs a) {
    extern __shared__ __align__(16) unsigned char sh_raw[];
    const int n = a.n;
    unsigned long long* sh_mov = reinterpret_cast<unsigned long long*>(sh_raw);   // [n] cells {susceptible, infectious}
    const InflowWords sh_movS{reinterpret_cast<uint32_t*>(sh_raw)}, sh_movI{reinterpret_cast<uint32_t*>(sh_raw) + 1};
    unsigned long long* sh_mass = reinterpret_cast<unsigned long long*>(sh_raw + (((size_t)n * 8 + 15) & ~(size_t)15));
    int* sh_err = reinterpret_cast<int*>(sh_mass + 4);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int e = blockIdx.x;
    const int g = a.g_begin + e * a.g_stride;
    const int ti = a.n_travel > 0 ? g % a.n_travel : 0;
    int bi = a.n_travel > 0 ? g / a.n_travel : g;
    bi = bi < a.beta_rows ? bi : a.beta_rows - 1;
    const int NP = a.npad;
    const size_t eoff = (size_t)e * NP;
    const size_t plane = (size_t)a.count * NP;
    // export counts {susceptible, infectious} of every patch as one 8-byte pair (one store, one load where they are used),
    // by patch in the first half of the scratch and by lane-layout position in the second
    int2* __restrict__ expB = reinterpret_cast<int2*>(a.exports) + eoff;
    int2* __restrict__ expP = reinterpret_cast<int2*>(a.exports + 2 * plane) + eoff;

    if (tid == 0) {
        sh_mass[0] = sh_mass[1] = sh_mass[2] = sh_mass[3] = 0ull;
        sh_err[0] = sh_err[1] = sh_err[2] = sh_err[3] = 0;
    }
    __syncthreads();

    int* sh_next = &sh_err[3];  // work queue of the mobility phase: next block or long row to hand out
    const int nblocks = a.n_blocks;
    EPI_TM(long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};)  // per-wave cycle counts by phase
#ifndef EPI_CSR_TIMERS
    long long* tm = nullptr;
#endif
    // Loads and stores of the local / apply pass: uniform base + 32-bit byte offset of the lane (one launch's planes
    // stay below 4 GiB, checked at launch), which needs no 64-bit vector arithmetic per access.
    auto ld4 = [](const int32_t* base, uint32_t off) { return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(base) + off); };
    auto ld8 = [](const double* base, uint32_t off) { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + off); };
    auto st4 = [](int32_t* base, uint32_t off, int32_t v) { *reinterpret_cast<int32_t*>(reinterpret_cast<char*>(base) + off) = v; };
    auto st8 = [](double* base, uint32_t off, double v) { *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + off) = v; };
#if EPI_CSR_KARGS
    // The pass reads its arguments again from the kernel-argument segment at every step (scalar loads): held for the
    // whole kernel next to those of the mobility phase they do not fit the scalar registers.
    typedef const __attribute__((address_space(4))) EpisodeArgs* KernArgs;
    const KernArgs kargs = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
#endif
    // the clock of a sweep's episode: one value for all patches (t advances by one in env.step and by one more, capped,
    // in stepOne: env.scala:132, WorldToy.scala:135), read here and written back by the last pass of the launch
    int tcu = a.uniform_clock ? a.tcur[eoff] : 0;
    // Step t: local dynamics (A), barrier, mobility (B), barrier, apply (C).  A thread owns the same patches
    // in A and C, so C of step t-1 and A of step t run as one pass over the thread's patches: the state of a
    // patch is read once and written once per step.
    for (int t = a.t_begin; t <= a.t_end; ++t) {
#if EPI_CSR_KARGS
        KernArgs ka = kargs;
        asm volatile("" : "+s"(ka));
#else
        const EpisodeArgs* ka = &a;
#endif
        int32_t* const kst = ka->st + eoff;
        int2* const kex = reinterpret_cast<int2*>(ka->exports) + eoff;
        int32_t* const ktc = ka->tcur + eoff;
        int32_t* const karr = ka->arrival + eoff;
        int32_t* const kpk = ka->peak + eoff;
        double* const kcap = ka->cap + eoff;
        const double* const kb0 = ka->beta0 + (size_t)bi * NP;
        const double* const kcst = ka->cst;
        const uint32_t planeB = (uint32_t)(plane * sizeof(int32_t)), np8 = (uint32_t)NP * 8u;
        const bool doC = t > a.t_begin, doA = t < a.t_end;
#if EPI_CSR_ACT_CONSTS
        // row of the constants table the step's action reads: -1 none, -2 unknown (actions differ by patch: fetch all)
        int act_cst = -2, shared_act = 0;
        if (a.shared_actions && !ka->act_episode && doA) {
            int sa = (int)ka->actions[(size_t)(t - ka->act_t0) * NP];
            sa = __builtin_amdgcn_readfirstlane(sa < 0 ? 0 : (sa > 7 ? 7 : sa));
            shared_act = sa;
            act_cst = sa == 1 ? CST_EFF1 : sa == 2 ? CST_EFF2 : sa == 3 ? CST_EFF3 : sa == 5 ? CST_SURGE : sa == 6 ? CST_VAX6
                      : sa == 7 ? CST_VAX7 : -1;
        }
#endif
        EPI_TM(long long k0 = clock64();)
        if (tid == 0) *sh_next = 0;
        unsigned long long mass_local = 0ull, mass_move = 0ull;
        // The inputs of a thread's next patch are fetched before the current one is worked on: a thread
        // visits about n / 1024 patches per step and would otherwise expose one memory latency for each.
        struct PatchIn {
            int32_t s, i, r, d, v, tc, es, ei, arr, pk, pos;
            uint32_t ms, mi;
            StepConsts k;
            int act;
            double z, cap, sw;
        };
        auto fetch = [&](int cc) {
            // the index is made opaque: otherwise one 64-bit address per array is precomputed ahead of the
            // step loop and kept (spilled) for the whole kernel
            uint32_t c = (uint32_t)cc;
            asm volatile("" : "+v"(c));
            const uint32_t o4 = c * 4u, o8 = c * 8u;
            PatchIn p;
            p.s = ld4(kst, o4); p.i = ld4(kst, o4 + planeB); p.r = ld4(kst, o4 + 2u * planeB); p.d = ld4(kst, o4 + 3u * planeB);
            p.v = ld4(kst, o4 + 4u * planeB);
            p.tc = a.uniform_clock ? tcu : ld4(ktc, o4);
            p.es = p.ei = p.arr = p.pk = 0;
            p.pos = -1;
            p.ms = p.mi = 0u;
            p.act = 0;
            p.z = p.cap = p.sw = 0.0;
            if (doC) {
                const unsigned long long cell = sh_mov[c];
                p.ms = (uint32_t)cell; p.mi = (uint32_t)(cell >> 32);
                { const int2 ex = *reinterpret_cast<const int2*>(reinterpret_cast<const char*>(kex) + o8); p.es = ex.x; p.ei = ex.y; }
                p.arr = ld4(karr, o4); p.pk = ld4(kpk, o4);
            }
            if (doA) {
                p.k.beta0 = ld8(kb0, o8);
                p.k.gamma = ld8(kcst, o8 + CST_GAMMA * np8);
                p.k.ifr0 = ld8(kcst, o8 + CST_IFR0 * np8);
                p.k.cnoise = ld8(kcst, o8 + CST_CNOISE * np8);
#if EPI_CSR_ACT_CONSTS
                if (act_cst >= 0) {   // the one constant the step's shared action uses
                    const double x = ld8(kcst, o8 + (uint32_t)act_cst * np8);
                    p.k.eff1 = p.k.eff2 = p.k.eff3 = p.k.surge = p.k.vax6 = p.k.vax7 = x;
                } else if (act_cst == -1) {   // none
                    p.k.eff1 = p.k.eff2 = p.k.eff3 = p.k.surge = p.k.vax6 = p.k.vax7 = 0.0;
                } else
#endif
                {
                p.k.eff1 = ld8(kcst, o8 + CST_EFF1 * np8);
                p.k.eff2 = ld8(kcst, o8 + CST_EFF2 * np8);
                p.k.eff3 = ld8(kcst, o8 + CST_EFF3 * np8);
                p.k.surge = ld8(kcst, o8 + CST_SURGE * np8);
                p.k.vax6 = ld8(kcst, o8 + CST_VAX6 * np8);
                p.k.vax7 = ld8(kcst, o8 + CST_VAX7 * np8);
                }
#if EPI_CSR_ACT_CONSTS
                if (act_cst != -2) p.act = shared_act;
                else
#endif
                p.act = ka->act_episode ? (int)ka->act_episode[eoff + c] : (int)ka->actions[(size_t)(t - ka->act_t0) * NP + c];
                p.z = ka->noise_episode ? ld8(ka->noise_episode + eoff, o8) : ld8(ka->noise + (size_t)(t - ka->noise_t0) * NP, o8);
                p.cap = ld8(kcap, o8);
                p.sw = ((int)c == ka->seed_idx && ka->qval_seed) ? ka->sumw_seed[ti] : ld8(ka->sumw, o8);
                if (ka->ell_pos) p.pos = ld4(ka->ell_pos, o4);
            }
            return p;
        };
        PatchIn nx = fetch(tid < n ? tid : 0);
        for (int cc = tid; cc < n; cc += kCsrThreads) {
            uint32_t c = (uint32_t)cc;
            asm volatile("" : "+v"(c));
            const uint32_t o4 = c * 4u, o8 = c * 8u;
            PatchIn p = nx;
            nx = fetch(cc + kCsrThreads < n ? cc + kCsrThreads : cc);
            // every constant is fetched with the state, in one batch: left alone, the loads of those that only
            // some actions use sink into the action branches, one exposed memory latency each
            keep_loaded(nx.k.eff1); keep_loaded(nx.k.eff2); keep_loaded(nx.k.eff3); keep_loaded(nx.k.surge);
            keep_loaded(nx.k.vax6); keep_loaded(nx.k.vax7); keep_loaded(nx.k.ifr0); keep_loaded(nx.k.cnoise);
            int32_t s0 = p.s, i0 = p.i, tc = p.tc;
            const int32_t r0 = p.r, d0 = p.d, v0 = p.v;
            if (doC) {  // ---- phase C of step t-1 (WorldToy.scala:128-145)
                const int32_t movedS = (int32_t)(p.ms - (uint32_t)p.es);
                const int32_t movedI = (int32_t)(p.mi - (uint32_t)p.ei);
                const double ns = fmax(0.0, (double)s0 + (double)movedS);
                const double ni = fmax(0.0, (double)i0 + (double)movedI);
                if (ns > 2147483647.0 || ni > 2147483647.0) record_error(sh_err, EPI_ERR_INT_RANGE, (int)c, tc);
                s0 = (int32_t)ns;
                i0 = (int32_t)ni;
                tc = min(tc + 1, a.step_t_max);
                if (p.arr < 0 && i0 >= 10) st4(karr, o4, t - 1);
                if (i0 > p.pk) st4(kpk, o4, i0);
                if (a.flow_s) {
                    a.flow_s[eoff + c] = movedS;
                    a.flow_i[eoff + c] = movedI;
                }
                if (a.series_i) a.series_i[((size_t)(t - 1 - a.series_t0) * a.count + e) * n + c] = (double)i0;
                const long long tmv = (long long)s0 + i0 + r0 + d0 + v0;
                mass_move += tmv > 0 ? (unsigned long long)tmv : 0ull;
            }
            if (doA) {  // ---- phase A of step t (WorldToy.scala:83-104)
                int act = p.act;
                act = act < 0 ? 0 : (act > 7 ? 7 : act);
                const long long tb = (long long)s0 + i0 + r0 + d0 + v0;
                const LocalOut o = local_step(p.k, s0, i0, r0, d0, v0, p.cap, act, p.z);
                const long long ta = (long long)o.s + o.i + o.r + o.d + o.v;
                if (a.validate && ta != tb) record_error(sh_err, EPI_ERR_LOCAL_MASS, (int)c, tc);
                if ((o.s | o.i | o.r | o.d | o.v) < 0) record_error(sh_err, EPI_ERR_INT_RANGE, (int)c, tc);
                if (a.rwd) a.reward_out[eoff + c] = reward_of(a.rwd[c], (double)i0, (double)d0, o, act);
                st4(kst, o4, o.s); st4(kst, o4 + planeB, o.i);
                st4(kst, o4 + 2u * planeB, o.r); st4(kst, o4 + 3u * planeB, o.d);
                st4(kst, o4 + 4u * planeB, o.v);
#if EPI_CSR_CAP_IF_CHANGED
                if (o.cap != p.cap)   // hospCap changes under SurgeCare only (env.scala:76)
#endif
                    st8(kcap, o8, o.cap);
                if (a.loc_s) { a.loc_s[eoff + c] = o.s; a.loc_i[eoff + c] = o.i; }
                if (!a.uniform_clock) st4(ktc, o4, tc + 1);
                const int32_t eS = export_count(max(0, o.s), p.sw), eI = export_count(max(0, o.i), p.sw);
                *reinterpret_cast<int2*>(reinterpret_cast<char*>(kex) + o8) = make_int2(eS, eI);
                if (p.pos >= 0) expP[p.pos] = make_int2(eS, eI);  // second copy where the mobility phase reads it with coalesced loads
                sh_mov[c] = 0ull;
                mass_local += ta > 0 ? (unsigned long long)ta : 0ull;
            } else {
                st4(kst, o4, s0);
                st4(kst, o4 + planeB, i0);
                st4(ktc, o4, tc);
            }
        }
        if (a.uniform_clock) {
            if (doC) tcu = min(tcu + 1, a.step_t_max);
            if (doA) tcu += 1;
        }
        if (a.validate) {
            // per-thread partial sums stay below 2^44 (at most 40 patches of < 2^34 each per thread)
            if (doC) {
                const unsigned long long wsum = wave_sum_u64(mass_move);
                if (lane == 0) atomicAdd(&sh_mass[((t - 1) & 1) * 2 + 1], wsum);
            }
            if (doA) {
                const unsigned long long wsum = wave_sum_u64(mass_local);
                if (lane == 0) atomicAdd(&sh_mass[(t & 1) * 2 + 0], wsum);
            }
        }
        if (!doA) break;
        EPI_TM(__builtin_amdgcn_s_waitcnt(0); { const long long k1 = clock64(); tm[3] += k1 - k0; k0 = k1; })
        __syncthreads();
        EPI_TM({ const long long k1 = clock64(); tm[4] += k1 - k0; k0 = k1; })

        // ------------------------------------------------------------------ phase B: mobility
        if (tid == 0 && a.validate && doC) {
            const int p = (t - 1) & 1;
            if (sh_mass[p * 2] != sh_mass[p * 2 + 1]) record_error(sh_err, EPI_ERR_WORLD_MASS, -1, t - 1);
            sh_mass[p * 2] = 0ull;
            sh_mass[p * 2 + 1] = 0ull;
        }
        {
            // Blocks of 64 rows, then the long rows, are handed out through an LDS counter (rows differ in
            // cost); the exports and the block record of a wave's next block are fetched while it works on
            // the current one.
            const int nitems = nblocks + a.n_long;
            auto grab = [&]() {
                int v = 0;
                if (lane == 0) v = atomicAdd(sh_next, 1);
                return __builtin_amdgcn_readfirstlane(v);
            };
            int item = grab();
#if EPI_CSR_GRAB_AHEAD
            // The queue is asked two items ahead: the answer of the LDS counter (behind the other waves' scatter
            // operations in the LDS pipe) arrives while the current item is worked on, not in front of it.
            int next = grab();
#endif
            const int b0 = item < nblocks ? item : 0;
            EllBlock blk = a.ell[b0];
            // patches numbered in lane-layout order (ell_rows == NULL): lane l of block b is patch 64 b + l itself
            const int2* __restrict__ expBlk = a.ell_rows ? expP : expB;
            // always a table look-up, fetched a block ahead with the exports: computing the row where the patches are
            // renumbered made two code paths write one register, and the wait the compiler puts in front of that
            // covered the block record requested just before it -- a memory latency per block
            auto row_of = [&](int pos) { return a.block_rows[pos]; };
            int row_n = row_of(b0 * 64 + lane);
            int2 EX = expBlk[b0 * 64 + lane];
            while (item < nitems) {
#if EPI_CSR_GRAB_AHEAD
                int after = 0;
                if (lane == 0) after = atomicAdd(sh_next, 1);
#else
                const int next = grab();
#endif
                EPI_TM(const long long q0 = clock64();)
                if (item < nblocks) {
                    const int b = item;
                    const EllBlock cur = blk;
                    const int row = row_n;
                    const int32_t ESc = row >= 0 ? EX.x : 0, EIc = row >= 0 ? EX.y : 0;
                    {
                        const int bn = next < nblocks ? next : b;
                        blk = a.ell[bn];
                        row_n = row_of(bn * 64 + lane);
                        EX = expBlk[bn * 64 + lane];
                    }
                    const int D = __builtin_amdgcn_readfirstlane(cur.slots);  // same for the whole block
                    if (__any((ESc | EIc) != 0) && D > 0) {
                        const size_t off = (size_t)__builtin_amdgcn_readfirstlane(cur.off) * 64;
                        // the block of the seed row has one copy per travel scale (only that row differs)
                        const double* __restrict__ qp = (b == a.seed_block && a.qval_seed)
                                                            ? a.qval_seed + (size_t)ti * a.seed_deg * 64 + lane
                                                            : a.qval + off + lane;
                        const int32_t* __restrict__ cp = a.col + off + lane;
                        const bool anyI = __any(EIc > 0);
                        // blocks are sorted by row length: the routine whose sorting network fits the block's longest row
                        EPI_TM(tm[7] += clock64() - q0;)
                        auto run = [&](auto ns) {
                            constexpr int NS = decltype(ns)::value <= DMAX ? decltype(ns)::value : DMAX;
                            lane_apportion<(NS + kSlotGroup - 1) / kSlotGroup * kSlotGroup, NS>(ESc, EIc, anyI, D, qp, cp, row, sh_mov,
                                                                                               sh_err, t, tm);
                        };
                        using std::integral_constant;
#if EPI_CSR_MID_NETS
                        if (DMAX >= 32 && D > 28) run(integral_constant<int, 32>{});
                        else if (DMAX >= 32 && D > 24) run(integral_constant<int, 28>{});
                        else if (DMAX >= 24 && D > 22) run(integral_constant<int, 24>{});
                        else if (DMAX >= 24 && D > 20) run(integral_constant<int, 22>{});
                        else if (DMAX >= 24 && D > 18) run(integral_constant<int, 20>{});
                        else if (DMAX >= 24 && D > 16) run(integral_constant<int, 18>{});
                        else if (D > 14) run(integral_constant<int, 16>{});
                        else if (D > 12) run(integral_constant<int, 14>{});
                        else if (D > 10) run(integral_constant<int, 12>{});
                        else if (D > 8) run(integral_constant<int, 10>{});
                        else run(integral_constant<int, 8>{});
#else
                        if (DMAX >= 32 && D > 24) run(integral_constant<int, 32>{});
                        else if (DMAX >= 24 && D > 20) run(integral_constant<int, 24>{});
                        else if (DMAX >= 24 && D > 16) run(integral_constant<int, 20>{});
                        else if (D > 12) run(integral_constant<int, 16>{});
                        else run(integral_constant<int, 12>{});
#endif
                    }
                } else {
                    // a row longer than kLaneSlots: one wave with the dense kernel's routine on the compact
                    // row (whole groups of 64 slots, zero padded), scattered through the row's column list
                    const int o = item - nblocks;
                    const int row = a.long_rows[o];
                    const int2 exl = expB[row];
                    const int32_t ESl = __builtin_amdgcn_readfirstlane(exl.x);
                    const int32_t EIl = __builtin_amdgcn_readfirstlane(exl.y);
                    if ((ESl | EIl) != 0) {
                        const int deg = a.rowptr[row + 1] - a.rowptr[row];
                        const int32_t* __restrict__ crow = a.long_col + a.long_off[o];
                        const double* __restrict__ qrow = (row == a.seed_idx && a.qval_seed && a.seed_long)
                                                              ? a.qval_seed + (size_t)ti * a.seed_deg
                                                              : a.long_q + a.long_off[o];
                        if (deg <= 64) long_row<1>(ESl, EIl, deg, row, qrow, crow, lane, sh_movS, sh_movI);
                        else if (deg <= 128) long_row<2>(ESl, EIl, deg, row, qrow, crow, lane, sh_movS, sh_movI);
                        else if (deg <= 256) long_row<4>(ESl, EIl, deg, row, qrow, crow, lane, sh_movS, sh_movI);
                        else wide_row(ESl, EIl, deg, row, qrow, crow, lane, sh_movS, sh_movI, sh_err, t);
                    }
                }
                item = next;
#if EPI_CSR_GRAB_AHEAD
                next = __builtin_amdgcn_readfirstlane(after);
#endif
            }
        }
        EPI_TM({ const long long k1 = clock64(); tm[5] += k1 - k0; k0 = k1; })
        __syncthreads();
        EPI_TM({ const long long k1 = clock64(); tm[6] += k1 - k0; k0 = k1; })
    }
    EPI_TM(if (a.stats && lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&a.stats[48 + i], (unsigned long long)tm[i]);)
    __syncthreads();
    if (tid == 0) {
        if (a.validate && a.t_end > a.t_begin) {
            const int p = (a.t_end - 1) & 1;
            if (sh_mass[p * 2] != sh_mass[p * 2 + 1]) record_error(sh_err, EPI_ERR_WORLD_MASS, -1, a.t_end - 1);
        }
        if (sh_err[0] != 0 && a.err[e * 4] == 0) {
            a.err[e * 4 + 1] = sh_err[1];
            a.err[e * 4 + 2] = sh_err[2];
            a.err[e * 4] = sh_err[0];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Small dense worlds (n <= 16 countries, e.g. the reference's six-country toy world): one lane per
// country, one group of 16 lanes per episode, four episodes per wave.  A lane keeps its row of the
// normalised matrix in registers for the whole launch and splits its exports with the same
// sorting-network routine as the sparse kernel (lane_threshold); the inflows of an episode are
// summed through 16 LDS words per compartment.  Same step as the wide kernel (WorldToy.scala:83-145),
// same selection rule: the rem largest fractions of ALL n columns, ties by column order, the origin's
// own entry dropped afterwards (WorldToy.scala:121).
// ---------------------------------------------------------------------------------------------
constexpr int kTinyN = 16;

// TN: columns a lane holds and sorts (8 for worlds of at most eight countries -- 19 comparators instead of 63 -- else 16)
template <int TN>
__device__ __forceinline__ void tiny_split(int32_t E, const double (&q)[TN], const double* __restrict__ qrow, int n, int c,
                                           uint32_t* sh_mov) {
    if (!__any(E > 0)) return;
    const double Ed = (double)E;
    uint32_t srt[TN];
    uint32_t bsum = 0u;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const double raw = q[j] * Ed;
        bsum += (uint32_t)(int32_t)raw;
        srt[j] = (uint32_t)__double2hiint(frac_of(raw));
    }
    const int32_t rem = E > 0 ? (int32_t)((uint32_t)E - bsum) : 0;
    const LaneThreshold th = lane_threshold<TN, 1, false>(srt, rem, Ed, n, qrow);  // a dense row: zero fractions take units too
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const double raw = q[j] * Ed;
        const uint32_t sel = (uint32_t)((uint32_t)__double2hiint(frac_of(raw)) >= th.Kx) | ((th.picked >> j) & 1u);
        const uint32_t alloc = (uint32_t)(int32_t)raw + (j < n ? sel : 0u);
        if (alloc != 0u && j != c) atomicAdd(&sh_mov[j], alloc);
    }
}

template <int TN>
__global__ __launch_bounds__(64) void episode_tiny_kernel(const EpisodeArgs a) {
    __shared__ uint32_t sh_movS[4][kTinyN], sh_movI[4][kTinyN];
    __shared__ unsigned long long sh_mass[4][2];
    __shared__ int sh_err[4][4];
    const int lane = threadIdx.x;
    const int grp = lane >> 4, c = lane & 15;
    const int e = blockIdx.x * 4 + grp;
    const int n = a.n;
    const bool live = e < a.count;
    const bool real = live && c < n;
    const int NP = a.npad;  // 64
    const int g = a.g_begin + (live ? e : 0) * a.g_stride;
    const int ti = a.n_travel > 0 ? g % a.n_travel : 0;
    int bi = a.n_travel > 0 ? g / a.n_travel : g;
    bi = bi < a.beta_rows ? bi : a.beta_rows - 1;
    const size_t eoff = (size_t)(live ? e : 0) * NP;
    const size_t plane = (size_t)a.count * NP;
    const int cc = c < n ? c : 0;

    int32_t s = a.st[0 * plane + eoff + cc], i = a.st[1 * plane + eoff + cc], r = a.st[2 * plane + eoff + cc];
    int32_t d = a.st[3 * plane + eoff + cc], v = a.st[4 * plane + eoff + cc];
    double cap = a.cap[eoff + cc];
    int32_t tcur = a.tcur[eoff + cc], arrival = a.arrival[eoff + cc], peak = a.peak[eoff + cc];
    int32_t last_flow_s = 0, last_flow_i = 0;
    const bool seed_row = (cc == a.seed_idx && a.qseed);
    const double* __restrict__ qrow = seed_row ? a.qseed + (size_t)ti * NP : a.q + (size_t)cc * NP;
    const double my_sumw = real ? (seed_row ? a.sumw_seed[ti] : a.sumw[cc]) : 0.0;
    double q[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) q[j] = real ? qrow[j] : 0.0;
    StepConsts k;
    k.beta0 = a.beta0[(size_t)bi * NP + cc];
    k.gamma = a.cst[CST_GAMMA * NP + cc];
    k.ifr0 = a.cst[CST_IFR0 * NP + cc];
    k.cnoise = a.cst[CST_CNOISE * NP + cc];
    k.eff1 = a.cst[CST_EFF1 * NP + cc];
    k.eff2 = a.cst[CST_EFF2 * NP + cc];
    k.eff3 = a.cst[CST_EFF3 * NP + cc];
    k.surge = a.cst[CST_SURGE * NP + cc];
    k.vax6 = a.cst[CST_VAX6 * NP + cc];
    k.vax7 = a.cst[CST_VAX7 * NP + cc];
    if (c == 0) {
        sh_mass[grp][0] = sh_mass[grp][1] = 0ull;
        sh_err[grp][0] = sh_err[grp][1] = sh_err[grp][2] = sh_err[grp][3] = 0;
    }
    __syncthreads();

    for (int t = a.t_begin; t < a.t_end; ++t) {
        // ---- A: local dynamics, exports (WorldToy.scala:83-115)
        sh_movS[grp][c] = 0u;
        sh_movI[grp][c] = 0u;
        int32_t eS = 0, eI = 0;
        unsigned long long mass_local = 0ull;
        if (real) {
            const size_t cidx = (size_t)e * n + c;   // position in the caller's compact [count][n] arrays
            int act = a.io_actions ? (int)a.io_actions[cidx]
                                   : (a.act_episode ? (int)a.act_episode[eoff + c] : (int)a.actions[(size_t)(t - a.act_t0) * NP + c]);
            act = act < 0 ? 0 : (act > 7 ? 7 : act);  // Action.fromId clamps (env.scala:18-20)
            const double z = a.io_gauss ? a.io_gauss[cidx]
                                        : (a.io_gauss_row ? a.io_gauss_row[c]
                                                          : (a.noise_episode ? a.noise_episode[eoff + c]
                                                                             : a.noise[(size_t)(t - a.noise_t0) * NP + c]));
            const long long tb = (long long)s + i + r + d + v;
            const LocalOut o = local_step(k, s, i, r, d, v, cap, act, z);
            const long long ta = (long long)o.s + o.i + o.r + o.d + o.v;
            if (a.validate && ta != tb) record_error(sh_err[grp], EPI_ERR_LOCAL_MASS, c, tcur);
            if ((o.s | o.i | o.r | o.d | o.v) < 0) record_error(sh_err[grp], EPI_ERR_INT_RANGE, c, tcur);
            if (a.rwd) {
                const double rw = reward_of(a.rwd[c], (double)i, (double)d, o, act);
                a.reward_out[eoff + c] = rw;
                if (a.io_rewards) a.io_rewards[cidx] = rw;
            }
            s = o.s; i = o.i; r = o.r; d = o.d; v = o.v;
            cap = o.cap;
            if (a.loc_s) { a.loc_s[eoff + c] = s; a.loc_i[eoff + c] = i; }
            tcur += 1;  // env.scala:132
            eS = export_count(max(0, s), my_sumw);
            eI = export_count(max(0, i), my_sumw);
            mass_local = ta > 0 ? (unsigned long long)ta : 0ull;
        }
        if (a.validate && mass_local) atomicAdd(&sh_mass[grp][0], mass_local);
        __syncthreads();
        // ---- B: mobility (WorldToy.scala:106-126), every lane splits its own row
        tiny_split<TN>(eS, q, qrow, n, c, sh_movS[grp]);
        tiny_split<TN>(eI, q, qrow, n, c, sh_movI[grp]);
        __syncthreads();
        // ---- C: apply, metrics, world mass (WorldToy.scala:128-145)
        unsigned long long mass_move = 0ull;
        if (real) {
            const int32_t movedS = (int32_t)(sh_movS[grp][c] - (uint32_t)eS);
            const int32_t movedI = (int32_t)(sh_movI[grp][c] - (uint32_t)eI);
            const double ns = fmax(0.0, (double)s + (double)movedS);
            const double ni = fmax(0.0, (double)i + (double)movedI);
            if (ns > 2147483647.0 || ni > 2147483647.0) record_error(sh_err[grp], EPI_ERR_INT_RANGE, c, tcur);
            s = (int32_t)ns;
            i = (int32_t)ni;
            tcur = min(tcur + 1, a.step_t_max);
            if (arrival < 0 && i >= 10) arrival = t;
            peak = max(peak, i);
            last_flow_s = movedS;
            last_flow_i = movedI;
            if (a.series_i) a.series_i[((size_t)(t - a.series_t0) * a.count + e) * n + c] = (double)i;
            const long long tm = (long long)s + i + r + d + v;
            mass_move = tm > 0 ? (unsigned long long)tm : 0ull;
        }
        if (a.validate && mass_move) atomicAdd(&sh_mass[grp][1], mass_move);
        __syncthreads();
        if (a.validate && c == 0) {
            if (sh_mass[grp][0] != sh_mass[grp][1]) record_error(sh_err[grp], EPI_ERR_WORLD_MASS, -1, t);
            sh_mass[grp][0] = sh_mass[grp][1] = 0ull;
        }
        // the next step's phase A adds to sh_mass[grp][0] only after the barrier that follows its own
        // LDS writes; lane c == 0 of the group cleared the sums above, in program order of the same wave
    }
    __syncthreads();
    if (a.io_err) {   // this step's table, zeros included: nothing has to clear it beforehand
        if (live && c < 4) a.err[e * 4 + c] = a.io_err[e * 4 + c] = sh_err[grp][c];
    } else if (live && c == 0 && sh_err[grp][0] != 0 && a.err[e * 4] == 0) {
        a.err[e * 4 + 1] = sh_err[grp][1];
        a.err[e * 4 + 2] = sh_err[grp][2];
        a.err[e * 4] = sh_err[grp][0];
    }
    if (real && a.io_states) {
        epi_state out;
        out.s = (double)s; out.i = (double)i; out.r = (double)r; out.d = (double)d; out.v = (double)v;
        out.hosp_cap = cap;
        out.t = tcur; out.t_max = a.io_t_max; out.seed = a.io_seed; out.pad_ = 0;
        a.io_states[(size_t)e * n + c] = out;
    }
    if (real && a.io_packed) {
        epi_state_packed out;
        out.s = s; out.i = i; out.r = r; out.d = d; out.v = v; out.t = tcur;
        out.hosp_cap = cap;
        a.io_packed[(size_t)e * n + c] = out;
    }
    if (real) {
        a.st[0 * plane + eoff + c] = s;
        a.st[1 * plane + eoff + c] = i;
        a.st[2 * plane + eoff + c] = r;
        a.st[3 * plane + eoff + c] = d;
        a.st[4 * plane + eoff + c] = v;
        a.cap[eoff + c] = cap;
        a.tcur[eoff + c] = tcur;
        a.arrival[eoff + c] = arrival;
        a.peak[eoff + c] = peak;
        if (a.flow_s) {
            a.flow_s[eoff + c] = last_flow_s;
            a.flow_i[eoff + c] = last_flow_i;
        }
    }
}

size_t csr_lds_bytes(int n) { return (((size_t)n * 8 + 15) & ~(size_t)15) + 4 * sizeof(unsigned long long) + 4 * sizeof(int); }

template <int DMAX>
static hipError_t launch_csr(const EpisodeArgs& a, hipStream_t s) {
    const size_t lds = csr_lds_bytes(a.n);
    hipError_t e = prepare_device_kernels();   // a flag test after the first call on a device
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(episode_csr_kernel<DMAX>, dim3((unsigned)a.count), dim3(kCsrThreads), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_episode(const EpisodeArgs& a, hipStream_t s) {
    if (a.count <= 0 || a.t_end <= a.t_begin) return hipSuccess;
    if (a.rowptr) {
        if (csr_lds_bytes(a.n) > kMaxLdsBytes) return hipErrorInvalidValue;
        if ((unsigned long long)a.count * a.npad * 5ull * sizeof(int32_t) > 0xffffffffull) return hipErrorInvalidValue;  // 32-bit plane offsets
        if (a.max_deg <= 16) return launch_csr<16>(a, s);
        if (a.max_deg <= 24) return launch_csr<24>(a, s);
        if (a.max_deg <= kLaneSlots) return launch_csr<kLaneSlots>(a, s);
        return hipErrorInvalidValue;
    }
    if (a.n <= kTinyN && a.npad == 64) {
        if (a.n <= 8) hipLaunchKernelGGL(episode_tiny_kernel<8>, dim3((unsigned)((a.count + 3) / 4)), dim3(64), 0, s, a);
        else hipLaunchKernelGGL(episode_tiny_kernel<kTinyN>, dim3((unsigned)((a.count + 3) / 4)), dim3(64), 0, s, a);
        return hipGetLastError();
    }
    const dim3 grid((unsigned)a.count);
    // waves per workgroup of the variants below 1,024 padded countries (measured on the 200-country grid: 4 waves
    // 1.22e9, 8 waves 1.27e9, 8 waves within 64 registers 1.53e9, 16 waves within 64 registers 1.66e9 country-steps/s;
    // 400 to 500 countries: 8 waves 1.10e9, 8 waves within 64 registers 1.33e9, 16 waves within 64 registers 1.46e9).
    // The 16-slot variant cannot follow: within 64 registers it spills its keys and runs seven times slower.
    constexpr int kW1 = 4, kW2 = 8, kW4 = EPI_KW4;
    switch (a.npad / 64) {
        case 1: hipLaunchKernelGGL((episode_dense_kernel<1, kW1>), grid, dim3(64 * kW1), 0, s, a); break;
        case 2: hipLaunchKernelGGL((episode_dense_kernel<2, kW2>), grid, dim3(64 * kW2), 0, s, a); break;
        case 4: hipLaunchKernelGGL((episode_dense_kernel<4, kW4>), grid, dim3(64 * kW4), 0, s, a); break;
        case 8: hipLaunchKernelGGL((episode_dense_kernel<8, 16>), grid, dim3(1024), 0, s, a); break;
        case 16: hipLaunchKernelGGL((episode_dense_kernel<16, 16>), grid, dim3(1024), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Initial states (WorldToy.scala:28-32) for `count` episodes.
// ---------------------------------------------------------------------------------------------
__global__ void init_states_kernel(const InitArgs a) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)a.count * a.npad;
    if (idx >= total) return;
    const int c = (int)(idx % a.npad);
    const int e = (int)(idx / a.npad);
    int32_t s = 0, i = 0, r = 0, d = 0, v = 0;
    double cap = 1.0;
    if (c < a.n) {
        const int32_t pop = a.pop[c];
        const double hc = a.base_cap * ((double)pop / 1000000.0);
        cap = hc > 1.0 ? hc : 1.0;
        if (c == a.seed_idx) {
            i = (int32_t)(0.001 * (double)pop);
            s = (int32_t)((uint32_t)pop - (uint32_t)i);
        } else {
            s = pop;
        }
        r = a.r0; d = a.d0; v = a.v0;
    }
    a.st[0 * total + idx] = s;
    a.st[1 * total + idx] = i;
    a.st[2 * total + idx] = r;
    a.st[3 * total + idx] = d;
    a.st[4 * total + idx] = v;
    a.cap[idx] = cap;
    a.tcur[idx] = a.t0;
    a.arrival[idx] = -1;
    a.peak[idx] = 0;
    if (c < 4) a.err[e * 4 + c] = 0;
}

hipError_t launch_init(const InitArgs& a, hipStream_t s) {
    const size_t total = (size_t)a.count * a.npad;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(init_states_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Loss of one grid point (Calibrate.scala:14-17,103-109): the two root-mean-squares are summed
// over countries in order by one lane so that every rounding matches the sequential reference.
// ---------------------------------------------------------------------------------------------
constexpr int kLossThreads = 1024;
constexpr int kLossChunk = 1024;   // countries whose squared errors are staged in LDS at a time

// One workgroup per episode.  The squared errors are formed by all threads, a chunk of countries at a time; the two
// sums run over them in country order on one thread each (Seq.sum adds left to right), reading LDS instead of
// waiting for one memory round trip per country.
__global__ __launch_bounds__(kLossThreads) void loss_kernel(const LossArgs a) {
    __shared__ double sh_qa[kLossChunk], sh_qp[kLossChunk];
    __shared__ double sh_sp;
    const int e = blockIdx.x;
    const int tid = threadIdx.x;
    const size_t eoff = (size_t)e * a.npad;
    double sum = 0.0;   // thread 0: arrival errors, thread 64: peak errors
    for (int c0 = 0; c0 < a.n; c0 += kLossChunk) {
        const int m = min(kLossChunk, a.n - c0);
        for (int k = tid; k < m; k += kLossThreads) {
            const int c = c0 + k;
            const int ci = a.id_of ? a.id_of[c] : c;
            const int32_t ai = a.arrival[eoff + ci];
            const int32_t arr_i = ai < 0 ? a.steps : ai;
            const double arr = (double)arr_i;
            const double pk = (double)a.peak[eoff + ci];
            if (a.arrival_out) a.arrival_out[(size_t)e * a.n + c] = arr_i;
            if (a.peak_out) a.peak_out[(size_t)e * a.n + c] = pk;
            if (a.loss) {
                const double da = arr - (a.has_arrive[c] ? a.t_arrive[c] : arr);
                const double dp = pk - (a.has_peak[c] ? a.t_peak[c] : pk);
                sh_qa[k] = da * da;
                sh_qp[k] = dp * dp;
            }
        }
        if (a.loss == nullptr) continue;
        __syncthreads();
        if (tid == 0 || tid == 64) {
            const double* q = tid == 0 ? sh_qa : sh_qp;
            int k = 0;
            if (c0 == 0) { sum = q[0]; k = 1; }
            for (; k + 16 <= m; k += 16) {   // sixteen LDS reads in flight, then the additions in order
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = q[k + u];
#pragma unroll
                for (int u = 0; u < 16; ++u) sum = sum + v[u];
            }
            for (; k < m; ++k) sum = sum + q[k];
        }
        __syncthreads();
    }
    if (a.loss == nullptr) return;
    if (tid == 64) sh_sp = sum;
    __syncthreads();
    if (tid != 0) return;
    const double sa = sum, sp = sh_sp;
    const double nn = fmax(1.0, (double)a.n);
    const double arr_loss = sqrt(sa / nn);
    const double pk_loss = sqrt(sp / nn);
    a.loss[e] = 0.5 * arr_loss + 0.5 * pk_loss;
}

hipError_t launch_loss(const LossArgs& a, hipStream_t s) {
    if (a.count <= 0) return hipSuccess;
    hipLaunchKernelGGL(loss_kernel, dim3((unsigned)a.count), dim3(kLossThreads), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// DDQNAgent.actGreedy: State.features -> MLP.predict -> first argmax (env.scala:26-30,
// MLP.scala:12-22, DDQNAgent.scala:22-23).  Every dot product is accumulated left to right with a
// separate multiply and add, as `row.view.zip(x).map(w*xi).sum` does.
// A workgroup evaluates MLP_ROWS states with one network; thread u owns hidden unit u.  With per-country
// networks (every WorldToy node asks its own agent) blockIdx.y is the country and the rows of a workgroup
// are episodes of that country.
// ---------------------------------------------------------------------------------------------
constexpr int MLP_MAXROWS = 16;
constexpr int MLP_MAXH = 256;
constexpr int MLP_LOADS = 16;   // weights fetched together: a lone wave has nothing else to hide a load behind

static size_t mlp_lds_bytes(int H, int rows) { return (size_t)(2 * rows * 8 + 2 * rows * H) * sizeof(double); }

// ROWS = states per workgroup (1, 2, 4, 8 or 16): a single world's evaluation has one state per network.
template <int ROWS>
__global__ __launch_bounds__(256) void mlp_greedy_kernel(const MlpArgs a) {
    extern __shared__ double mlp_sh[];
    const int H = a.nl.H, Hp = a.nl.Hp;
    double* sh_x = mlp_sh;                   // [ROWS][8]
    double* sh_y = sh_x + ROWS * 8;          // [ROWS][8]
    double* sh_h1 = sh_y + ROWS * 8;         // [ROWS][H]
    double* sh_h2 = sh_h1 + ROWS * H;        // [ROWS][H]
    const int tid = threadIdx.x;
    const int cfix = a.per_country ? (int)blockIdx.y : 0;
    const size_t nidx = a.per_country ? (size_t)(a.net_index ? a.net_index[cfix] : cfix) : 0;
    const double* __restrict__ net = a.nets + nidx * a.net_stride + a.net_offset;
    // row r of this workgroup -> (episode, country); false past the end
    auto locate = [&](int r, int& e, int& c) {
        if (a.per_country) {
            e = blockIdx.x * ROWS + r;
            c = cfix;
            return e < a.count;
        }
        const int row = blockIdx.x * ROWS + r;
        e = row / a.n;
        c = row - e * a.n;
        return row < a.rows;
    };

    if (tid < ROWS) {
        double f[7] = {0, 0, 0, 0, 0, 0, 0};
        int e, c;
        if (locate(tid, e, c)) {
            double s, i, r, d, v, cap;
            int32_t t, tmax;
            if (a.st) {
                const size_t idx = (size_t)e * a.npad + c;
                s = (double)a.st[0 * a.plane + idx]; i = (double)a.st[1 * a.plane + idx];
                r = (double)a.st[2 * a.plane + idx]; d = (double)a.st[3 * a.plane + idx];
                v = (double)a.st[4 * a.plane + idx];
                cap = a.cap[idx]; t = a.tcur[idx]; tmax = a.state_t_max;
            } else {
                const epi_state& st = a.states[(size_t)e * a.n + c];
                s = st.s; i = st.i; r = st.r; d = st.d; v = st.v; cap = st.hosp_cap; t = st.t; tmax = st.t_max;
            }
            const double N = fmax(1.0, s + i + r + d);
            f[0] = s / N; f[1] = i / N; f[2] = r / N; f[3] = d / N; f[4] = v / N;
            f[5] = i / fmax(1.0, cap);
            f[6] = (double)t / (double)(tmax > 1 ? tmax : 1);
        }
        for (int k = 0; k < 7; ++k) sh_x[tid * 8 + k] = f[k];
    }
    // the first weights of the second layer are on their way while the first layer is computed
    const double* __restrict__ W2T = net + a.nl.w2t();
    double wn[MLP_LOADS];
    if (tid < H) {
#pragma unroll
        for (int j = 0; j < MLP_LOADS; ++j) wn[j] = W2T[(size_t)j * Hp + tid];   // rows past H are zero padding (Hp >= 64)
    }
    __syncthreads();
    if (tid < H) {
        double w[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) w[k] = net[a.nl.w1t() + (size_t)k * Hp + tid];
        const double b = net[a.nl.b1() + tid];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            double acc = w[0] * sh_x[r * 8 + 0];
#pragma unroll
            for (int k = 1; k < 7; ++k) acc = acc + w[k] * sh_x[r * 8 + k];
            const double z = acc + b;
            sh_h1[r * H + tid] = z > 0 ? z : 0.0;
        }
    }
    __syncthreads();
    if (tid < H) {
        double acc[ROWS];
        for (int k0 = 0; k0 < H; k0 += MLP_LOADS) {
            double wc[MLP_LOADS];
#pragma unroll
            for (int j = 0; j < MLP_LOADS; ++j) wc[j] = wn[j];
            if (k0 + MLP_LOADS < H) {
#pragma unroll
                for (int j = 0; j < MLP_LOADS; ++j) wn[j] = W2T[(size_t)(k0 + MLP_LOADS + j) * Hp + tid];
            }
#pragma unroll
            for (int j = 0; j < MLP_LOADS; ++j) {
                const int k = k0 + j;
                if (k < H) {
#pragma unroll
                    for (int r = 0; r < ROWS; ++r) {
                        const double p = wc[j] * sh_h1[r * H + k];
                        acc[r] = (k == 0) ? p : acc[r] + p;
                    }
                }
            }
        }
        const double b = net[a.nl.b2() + tid];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const double z = acc[r] + b;
            sh_h2[r * H + tid] = z > 0 ? z : 0.0;
        }
    }
    __syncthreads();
    if (tid < ROWS * 8) {
        const int r = tid >> 3, o = tid & 7;
        const double* __restrict__ W3 = net + a.nl.w3() + (size_t)o * Hp;
        double acc = 0.0;
        for (int k0 = 0; k0 < H; k0 += MLP_LOADS) {
            double wc[MLP_LOADS];
#pragma unroll
            for (int j = 0; j < MLP_LOADS; ++j) wc[j] = W3[k0 + j];   // the row is padded to Hp
#pragma unroll
            for (int j = 0; j < MLP_LOADS; ++j) {
                const int k = k0 + j;
                if (k < H) {
                    const double p = wc[j] * sh_h2[r * H + k];
                    acc = (k == 0) ? p : acc + p;
                }
            }
        }
        double y = acc + net[a.nl.b3() + o];
        const double clip = a.clip_ptr ? a.clip_ptr[nidx * a.clip_stride] : a.clip_q;
        if (isfinite(clip)) y = fmax(-clip, fmin(clip, y));
        sh_y[r * 8 + o] = y;
    }
    __syncthreads();
    if (tid < ROWS) {
        int e, c;
        if (locate(tid, e, c)) {
            const size_t row = (size_t)e * a.n + c;
            int best = 0;
            for (int o = 1; o < 8; ++o)
                if (sh_y[tid * 8 + o] > sh_y[tid * 8 + best]) best = o;
            if (a.act_episode) a.act_episode[(size_t)e * a.npad + c] = (int8_t)best;
            if (a.act_out) a.act_out[row] = best;
            if (a.q_out)
                for (int o = 0; o < 8; ++o) a.q_out[row * 8 + o] = sh_y[tid * 8 + o];
        }
    }
}

// The same evaluation for at most eight states per network (one world's greedy step, an evaluation rollout): one
// workgroup of 1,024 threads per one or two states, and everything arranged for latency, because a lone workgroup has nothing
// to overlap a memory round trip with.  Every weight the workgroup needs is requested in one go at the start:
// thread (part, u) holds W2(u)(k) for the KPER inputs k of its part, the first Hp threads the first layer, thread
// (o, k) one weight of the third.  The sums keep their order: the second layer's run over k = 0, 1, ... as a relay
// through the parts (part s adds its KPER products to what part s - 1 left in LDS), the third layer's products are
// formed by all threads at once and added in order by one thread per (state, output).
// HP = hidden units = row stride of the network image (64 or 128: 16 parts of 4 inputs, 8 parts of 16).
constexpr int MLP_CHAIN_THREADS = 1024;

template <int HP, int ROWS>
__global__ __launch_bounds__(MLP_CHAIN_THREADS) void mlp_chain_kernel(const MlpArgs a) {
    constexpr int PARTS = MLP_CHAIN_THREADS / HP;
    constexpr int KPER = HP / PARTS;
    static_assert(PARTS * KPER == HP, "the parts cover the inputs");
    __shared__ double sh_x[ROWS][8];
    __shared__ double sh_h[ROWS][HP];        // first layer's activations, later the second layer's
    __shared__ double sh_acc[ROWS][HP];      // the relay: partial sums of the second layer
    __shared__ double sh_p3[ROWS][8][HP];    // products of the third layer
    __shared__ double sh_y[ROWS][8];
    const int tid = threadIdx.x;
    const int part = tid / HP, u = tid % HP;
    const int cfix = a.per_country ? (int)blockIdx.y : 0;
    const size_t nidx = a.per_country ? (size_t)(a.net_index ? a.net_index[cfix] : cfix) : 0;
    const double* __restrict__ net = a.nets + nidx * a.net_stride + a.net_offset;
    const NetLayout nl{HP, HP};
    auto locate = [&](int r, int& e, int& c) {
        if (a.per_country) {
            e = blockIdx.x * ROWS + r;
            c = cfix;
            return e < a.count;
        }
        const int row = blockIdx.x * ROWS + r;
        e = row / a.n;
        c = row - e * a.n;
        return row < a.rows;
    };
    // every weight this thread will use, requested before anything else
    double w2[KPER];
#pragma unroll
    for (int j = 0; j < KPER; ++j) w2[j] = net[nl.w2t() + (size_t)(part * KPER + j) * HP + u];
    double w1[7] = {0, 0, 0, 0, 0, 0, 0}, b1 = 0.0;
    if (part == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) w1[k] = net[nl.w1t() + (size_t)k * HP + u];
        b1 = net[nl.b1() + u];
    }
    const double b2 = part == PARTS - 1 ? net[nl.b2() + u] : 0.0;
    // third layer: thread (o, k) for tid < 8 * HP
    const int o3 = tid / HP, k3 = tid % HP;
    const double w3 = o3 < 8 ? net[nl.w3() + (size_t)o3 * HP + k3] : 0.0;
    const double b3 = tid < ROWS * 8 ? net[nl.b3() + (tid & 7)] : 0.0;
    const double clip = a.clip_ptr ? a.clip_ptr[nidx * a.clip_stride] : a.clip_q;

    if (tid < ROWS) {
        double f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int e, c;
        if (locate(tid, e, c)) {
            double s, i, r, d, v, cap;
            int32_t t, tmax;
            if (a.st) {
                const size_t idx = (size_t)e * a.npad + c;
                s = (double)a.st[0 * a.plane + idx]; i = (double)a.st[1 * a.plane + idx];
                r = (double)a.st[2 * a.plane + idx]; d = (double)a.st[3 * a.plane + idx];
                v = (double)a.st[4 * a.plane + idx];
                cap = a.cap[idx]; t = a.tcur[idx]; tmax = a.state_t_max;
            } else {
                const epi_state& st = a.states[(size_t)e * a.n + c];
                s = st.s; i = st.i; r = st.r; d = st.d; v = st.v; cap = st.hosp_cap; t = st.t; tmax = st.t_max;
            }
            const double N = fmax(1.0, s + i + r + d);
            f[0] = s / N; f[1] = i / N; f[2] = r / N; f[3] = d / N; f[4] = v / N;
            f[5] = i / fmax(1.0, cap);
            f[6] = (double)t / (double)(tmax > 1 ? tmax : 1);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) sh_x[tid][k] = f[k];
    }
    __syncthreads();
    if (part == 0) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            double acc = w1[0] * sh_x[r][0];
#pragma unroll
            for (int k = 1; k < 7; ++k) acc = acc + w1[k] * sh_x[r][k];
            const double z = acc + b1;
            sh_h[r][u] = z > 0 ? z : 0.0;
        }
    }
    __syncthreads();
    // second layer: the products of this thread's inputs (formed ahead of the relay while they fit the registers),
    // then the relay
    constexpr bool kAhead = ROWS * KPER <= 32;
    double p[kAhead ? ROWS : 1][KPER];
    if constexpr (kAhead) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int j = 0; j < KPER; ++j) p[r][j] = w2[j] * sh_h[r][part * KPER + j];
    }
#pragma unroll 1
    for (int stage = 0; stage < PARTS; ++stage) {
        if (part == stage) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                if constexpr (!kAhead) {
#pragma unroll
                    for (int j = 0; j < KPER; ++j) p[0][j] = w2[j] * sh_h[r][part * KPER + j];
                }
                const double* pr = p[kAhead ? r : 0];
                double acc = stage == 0 ? pr[0] : sh_acc[r][u] + pr[0];
#pragma unroll
                for (int j = 1; j < KPER; ++j) acc = acc + pr[j];
                sh_acc[r][u] = acc;
            }
        }
        __syncthreads();
    }
    if (part == PARTS - 1) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const double z = sh_acc[r][u] + b2;
            sh_h[r][u] = z > 0 ? z : 0.0;   // every product of the first layer's activations has been formed
        }
    }
    __syncthreads();
    if (o3 < 8) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) sh_p3[r][o3][k3] = w3 * sh_h[r][k3];
    }
    __syncthreads();
    if (tid < ROWS * 8) {
        const int r = tid >> 3, o = tid & 7;
        const double* q = sh_p3[r][o];
        double y = q[0];
        for (int k = 1; k + 16 <= HP; k += 16) {   // sixteen LDS reads in flight, then the additions in order
            double v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = q[k + j];
#pragma unroll
            for (int j = 0; j < 16; ++j) y = y + v[j];
        }
#pragma unroll
        for (int k = HP - 15; k < HP; ++k) y = y + q[k];
        y = y + b3;
        if (isfinite(clip)) y = fmax(-clip, fmin(clip, y));
        sh_y[r][o] = y;
    }
    __syncthreads();
    if (tid < ROWS) {
        int e, c;
        if (locate(tid, e, c)) {
            const size_t row = (size_t)e * a.n + c;
            int best = 0;
            for (int o = 1; o < 8; ++o)
                if (sh_y[tid][o] > sh_y[tid][best]) best = o;
            if (a.act_episode) a.act_episode[(size_t)e * a.npad + c] = (int8_t)best;
            if (a.act_out) a.act_out[row] = best;
            if (a.q_out)
                for (int o = 0; o < 8; ++o) a.q_out[row * 8 + o] = sh_y[tid][o];
        }
    }
}

template <int HP, int ROWS>
static hipError_t launch_mlp_chain(const MlpArgs& a, int live, hipStream_t s) {
    const dim3 grid((unsigned)((live + ROWS - 1) / ROWS), a.per_country ? (unsigned)a.n : 1u);
    hipLaunchKernelGGL((mlp_chain_kernel<HP, ROWS>), grid, dim3(MLP_CHAIN_THREADS), 0, s, a);
    return hipGetLastError();
}

// The same evaluation for large batches that share a network (the argmax-Q step of a sweep: every patch of every
// episode).  A workgroup of two waves evaluates 32 states as four tiles of 8: thread (w, g, q) -- wave w, g = lane / 16,
// q = lane % 16 -- owns the UN = HP / 32 hidden units (16 w + q) * UN .. + UN - 1 for the 8 states of tile g.  Per input k
// of the second layer a lane reads its tile's 8 activations (four 16-byte LDS reads, the same address in all 16 lanes
// of the tile) and its UN weights (global, the same in all four tiles) for 8 * UN multiply-add pairs: the layer is
// bound by the vector unit, where the unit-per-thread kernel above is bound by the LDS return path (16 reads per 32
// operations).  Every sum runs over k in ascending order with a separate multiply and add, as in mlp_greedy_kernel.
// HP = hidden units = row stride of the network image (64 or 128).
constexpr int MLP_TILE_ROWS = 32;
constexpr int MLP_TILE_THREADS = 128;

//
// MODE 0: the evaluation as described.
// MODE 1 (screen): the same sums with fused multiply-adds -- half the vector instructions of the second layer -- when
//   only the greedy action is wanted.  The fused values are not the reference's, but both lie within e3 of the network
//   evaluated in real arithmetic on the same features (the standard forward bound: a dot product of m terms plus a bias,
//   in any order, fused or not, is off by at most gamma_(m+1) * (sum |w x| + |b|); ReLU and the clamp do not expand
//   errors).  With A_k the largest absolute row sum of layer k's weights, B_k its largest absolute bias and X the
//   largest absolute feature:  H1 = A1 X + B1,  e1 = g8 H1,  H2 = A2 H1 + B2,  e2 = A2 e1 + g (A2 (H1 + e1) + B2),
//   e3 = A3 e2 + g (A3 (H2 + e2) + B3),  g = gamma_(HP+1).  Where the best fused value exceeds every other one by more
//   than 4 e3 the reference's values differ strictly the same way and its first argmax is the fused one: that action is
//   final.  Every other state (near ties, values at the clamp) is flagged (kActUncertain).
// MODE 2: MODE 0 for the tiles that hold a flagged state, a return for all others.
constexpr double kUnitRoundoff = 1.1102230246251565e-16;   // 2^-53

template <bool FUSED>
__device__ __forceinline__ double mac(double w, double x, double acc) {
    if constexpr (FUSED) return __builtin_fma(w, x, acc);
    else return acc + w * x;
}

template <int HP, int MODE>
__global__ __launch_bounds__(MLP_TILE_THREADS) void mlp_tile_kernel(const MlpArgs a) {
    constexpr int UN = HP / 32;     // units per thread
    constexpr int TM = 8;           // states per tile
    constexpr int KB = 4;           // inputs whose weights are fetched together
    constexpr bool FUSED = MODE == 1;
    __shared__ __align__(16) double sh_x[MLP_TILE_ROWS][8];
    __shared__ __align__(16) double sh_h[HP][MLP_TILE_ROWS];   // activations, input-major: [k][state]
    __shared__ __align__(16) double sh_y[MLP_TILE_ROWS][8];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int g = lane >> 4;
    const int cfix = a.per_country ? (int)blockIdx.y : 0;
    const size_t nidx = a.per_country ? (size_t)(a.net_index ? a.net_index[cfix] : cfix) : 0;
    const double* __restrict__ net = a.nets + nidx * a.net_stride + a.net_offset;
    const NetLayout nl{HP, HP};
    auto locate = [&](int r, int& e, int& c) {
        if (a.per_country) {
            e = blockIdx.x * MLP_TILE_ROWS + r;
            c = cfix;
            return e < a.count;
        }
        const int row = blockIdx.x * MLP_TILE_ROWS + r;
        e = row / a.n;
        c = row - e * a.n;
        return row < a.rows;
    };
    if constexpr (MODE == 2) {
        int flagged = 0;
        if (tid < MLP_TILE_ROWS) {
            int e, c;
            if (locate(tid, e, c)) flagged = a.act_episode[(size_t)e * a.npad + c] & kActUncertain;
        }
        if (!__syncthreads_or(flagged)) return;
    }
    if (tid < MLP_TILE_ROWS) {
        double f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int e, c;
        if (locate(tid, e, c)) {
            double s, i, r, d, v, cap;
            int32_t t, tmax;
            if (a.st) {
                const size_t idx = (size_t)e * a.npad + c;
                s = (double)a.st[0 * a.plane + idx]; i = (double)a.st[1 * a.plane + idx];
                r = (double)a.st[2 * a.plane + idx]; d = (double)a.st[3 * a.plane + idx];
                v = (double)a.st[4 * a.plane + idx];
                cap = a.cap[idx]; t = a.tcur[idx]; tmax = a.state_t_max;
            } else {
                const epi_state& st = a.states[(size_t)e * a.n + c];
                s = st.s; i = st.i; r = st.r; d = st.d; v = st.v; cap = st.hosp_cap; t = st.t; tmax = st.t_max;
            }
            const double N = fmax(1.0, s + i + r + d);
            f[0] = s / N; f[1] = i / N; f[2] = r / N; f[3] = d / N; f[4] = v / N;
            f[5] = i / fmax(1.0, cap);
            f[6] = (double)t / (double)(tmax > 1 ? tmax : 1);
            if constexpr (MODE == 1) {   // slot 7 is not an input: it carries the largest absolute feature to the bound
                double X = 0.0;
#pragma unroll
                for (int k = 0; k < 7; ++k) X = fmax(X, fabs(f[k]));
                f[7] = X;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) sh_x[tid][k] = f[k];
    }
    const int u0 = ((tid >> 6) * 16 + (lane & 15)) * UN;
    const double* __restrict__ W2T = net + nl.w2t() + u0;
    // the first weights of the second layer are on their way while the first layer is computed
    double wn[KB][UN];
#pragma unroll
    for (int j = 0; j < KB; ++j)
#pragma unroll
        for (int u = 0; u < UN; ++u) wn[j][u] = W2T[(size_t)j * HP + u];
    __syncthreads();
    {   // first layer: UN units x TM states per thread
        double w[7][UN], b[UN];
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int u = 0; u < UN; ++u) w[k][u] = net[nl.w1t() + (size_t)k * HP + u0 + u];
#pragma unroll
        for (int u = 0; u < UN; ++u) b[u] = net[nl.b1() + u0 + u];
#pragma unroll
        for (int r = 0; r < TM; ++r) {
            double x[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) x[k] = sh_x[g * TM + r][k];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                double acc = w[0][u] * x[0];
#pragma unroll
                for (int k = 1; k < 7; ++k) acc = mac<FUSED>(w[k][u], x[k], acc);
                const double z = acc + b[u];
                sh_h[u0 + u][g * TM + r] = z > 0 ? z : 0.0;
            }
        }
    }
    __syncthreads();
    if constexpr (MODE == 1) {
        // Second layer of the screen on the matrix cores (v_mfma_f64_16x16x4_f64: D[16 units][16 states] += A[16 units][4 inputs]
        // * B[4 inputs][16 states], one double of A and of B per lane -- A[lane & 15][lane >> 4], B[lane >> 4][lane & 15] -- and
        // four of D: row (lane >> 4) + 4 r, column lane & 15).  Any summation order is as good as another to the screen's
        // error bound.  A wave owns half of the HP / 16 unit tiles for both 16-state tiles of the workgroup: per four inputs
        // it reads its weights once (W2 transposed: 16 consecutive units per input) and the activations once per state tile.
        typedef double f64x4 __attribute__((ext_vector_type(4)));
        constexpr int UT = HP / 32;   // unit tiles per wave
        const int w = tid >> 6, lr = lane & 15, lq = lane >> 4;
        const double* __restrict__ W2Tm = net + nl.w2t() + (size_t)lq * HP + (size_t)w * UT * 16 + lr;
        f64x4 d[UT][2];
#pragma unroll
        for (int t = 0; t < UT; ++t) { d[t][0] = f64x4{0, 0, 0, 0}; d[t][1] = f64x4{0, 0, 0, 0}; }
        double an[UT];
#pragma unroll
        for (int t = 0; t < UT; ++t) an[t] = W2Tm[t * 16];
#pragma nounroll
        for (int k0 = 0; k0 < HP; k0 += 4) {
            double ac[UT];
#pragma unroll
            for (int t = 0; t < UT; ++t) ac[t] = an[t];
            if (k0 + 4 < HP) {
#pragma unroll
                for (int t = 0; t < UT; ++t) an[t] = W2Tm[(size_t)(k0 + 4) * HP + t * 16];
            }
            const double b0 = sh_h[k0 + lq][lr], b1 = sh_h[k0 + lq][16 + lr];
#pragma unroll
            for (int t = 0; t < UT; ++t) {
                d[t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[t], b0, d[t][0], 0, 0, 0);
                d[t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[t], b1, d[t][1], 0, 0, 0);
            }
        }
        __syncthreads();   // every wave has read the first layer's activations: the buffer takes the second layer's
#pragma unroll
        for (int t = 0; t < UT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int unit = (w * UT + t) * 16 + lq + 4 * r;
                const double b = net[nl.b2() + unit];
                const double z0 = d[t][0][r] + b, z1 = d[t][1][r] + b;
                sh_h[unit][lr] = z0 > 0 ? z0 : 0.0;
                sh_h[unit][16 + lr] = z1 > 0 ? z1 : 0.0;
            }
    } else {
    double acc[UN][TM];
    // one batch of KB inputs: `first` starts the sums with the product of input 0 (no addition to a zero)
    auto batch = [&](int k0, auto first) {
        double wc[KB][UN];
#pragma unroll
        for (int j = 0; j < KB; ++j)
#pragma unroll
            for (int u = 0; u < UN; ++u) wc[j][u] = wn[j][u];
        if (k0 + KB < HP) {
#pragma unroll
            for (int j = 0; j < KB; ++j)
#pragma unroll
                for (int u = 0; u < UN; ++u) wn[j][u] = W2T[(size_t)(k0 + KB + j) * HP + u];
        }
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            double h[TM];
#pragma unroll
            for (int r = 0; r < TM; ++r) h[r] = sh_h[k0 + j][g * TM + r];
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int r = 0; r < TM; ++r) {
                    if (decltype(first)::value && j == 0) acc[u][r] = wc[j][u] * h[r];
                    else acc[u][r] = mac<FUSED>(wc[j][u], h[r], acc[u][r]);
                }
        }
    };
    batch(0, std::true_type{});
#pragma nounroll
    for (int k0 = KB; k0 < HP; k0 += KB) batch(k0, std::false_type{});
    __syncthreads();   // every thread has read the first layer's activations: the buffer takes the second layer's
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const double b = net[nl.b2() + u0 + u];
#pragma unroll
        for (int r = 0; r < TM; ++r) {
            const double z = acc[u][r] + b;
            sh_h[u0 + u][g * TM + r] = z > 0 ? z : 0.0;
        }
    }
    }
    __syncthreads();
    {   // third layer: thread = (state, two of the eight outputs)
        const int row = tid & 31, o0 = (tid >> 5) * 2;
        double y[2];
        const double* __restrict__ W3 = net + nl.w3() + (size_t)o0 * HP;
        auto batch3 = [&](int k0, auto first) {
            double w[2][8], h[8];
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int j = 0; j < 8; ++j) w[o][j] = W3[(size_t)o * HP + k0 + j];
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = sh_h[k0 + j][row];
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    if (decltype(first)::value && j == 0) y[o] = w[o][j] * h[j];
                    else y[o] = mac<FUSED>(w[o][j], h[j], y[o]);
                }
        };
        batch3(0, std::true_type{});
#pragma nounroll
        for (int k0 = 8; k0 < HP; k0 += 8) batch3(k0, std::false_type{});
        const double clip = a.clip_ptr ? a.clip_ptr[nidx * a.clip_stride] : a.clip_q;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            double v = y[o] + net[nl.b3() + o0 + o];
            if (isfinite(clip)) v = fmax(-clip, fmin(clip, v));
            sh_y[row][o0 + o] = v;
        }
    }
    __syncthreads();
    int unsure = 0, live = 0;   // MODE 1: this thread's state was looked at / was flagged
    if (tid < MLP_TILE_ROWS) {
        int e, c;
        if (locate(tid, e, c)) {
            const size_t row = (size_t)e * a.n + c;
            int best = 0;
            for (int o = 1; o < 8; ++o)
                if (sh_y[tid][o] > sh_y[tid][best]) best = o;
            if constexpr (MODE == 1) {
                double second = -__builtin_inf();
                for (int o = 0; o < 8; ++o)
                    if (o != best) second = fmax(second, sh_y[tid][o]);
                const double X = sh_x[tid][7];
                const double g8 = 8.0 * 1.01 * kUnitRoundoff, gh = (double)(HP + 1) * 1.01 * kUnitRoundoff;
                const double A1 = a.cert[0], B1 = a.cert[1], A2 = a.cert[2], B2 = a.cert[3], A3 = a.cert[4], B3 = a.cert[5];
                const double H1 = A1 * X + B1, e1 = g8 * H1;
                const double H2 = A2 * H1 + B2, e2 = A2 * e1 + gh * (A2 * (H1 + e1) + B2);
                const double e3 = A3 * e2 + gh * (A3 * (H2 + e2) + B3);
                // The floor covers what the relative bound does not see: products and sums that underflow.  With gradual
                // underflow that is a few hundred roundings of 2^-1074; should the matrix cores flush subnormal
                // operands or results to zero instead, each of the HP products of a second-layer sum may be off by up
                // to the smallest normal number, 2.3e-308, which the third layer scales by at most A3 (and the first
                // layer's own flushes by A2 A3).  1e-300 (1 + A2)(1 + A3) is far above either and far below any margin
                // that matters.  The comparison is false for NaN as well.
                const double floor_abs = 1e-300 * (1.0 + A2) * (1.0 + A3);
                const bool sure = (sh_y[tid][best] - second) > 4.0 * e3 * a.cert_scale + floor_abs;
                a.act_episode[(size_t)e * a.npad + c] = (int8_t)(sure ? best : (best | kActUncertain));
                unsure = !sure;
                live = 1;
            } else {
                if (a.act_episode) a.act_episode[(size_t)e * a.npad + c] = (int8_t)best;
                if (a.act_out) a.act_out[row] = best;
                if (a.q_out)
                    for (int o = 0; o < 8; ++o) a.q_out[row * 8 + o] = sh_y[tid][o];
            }
        }
    }
    if constexpr (MODE == 1) {
        if (a.cert_stats && tid < 64) {   // the tile's states sit in the first wave (MLP_TILE_ROWS <= 64): one pair of atomics per tile
            const int n_live = __popcll(__ballot(live)), n_unsure = __popcll(__ballot(unsure));
            if (tid == 0) {
                atomicAdd(&a.cert_stats[0], (unsigned long long)n_live);
                if (n_unsure) atomicAdd(&a.cert_stats[1], (unsigned long long)n_unsure);
            }
        }
    }
}

template <int ROWS>
static hipError_t launch_mlp_rows(const MlpArgs& a, int live, hipStream_t s) {
    const int H = a.nl.H;
    const size_t lds = mlp_lds_bytes(H, ROWS);
    if (lds > 48 * 1024) {
        hipError_t e = prepare_device_kernels();   // a flag test after the first call on a device
        if (e != hipSuccess) return e;
    }
    const dim3 grid((unsigned)((live + ROWS - 1) / ROWS), a.per_country ? (unsigned)a.n : 1u);
    hipLaunchKernelGGL(mlp_greedy_kernel<ROWS>, grid, dim3(H > 128 ? 256 : 128), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_mlp(const MlpArgs& a, hipStream_t s) {
    if (a.rows <= 0) return hipSuccess;
    const int H = a.nl.H;
    if (H < 1 || H > MLP_MAXH || a.nl.Hp < H || a.nl.Hp % 64 != 0) return hipErrorInvalidValue;
    const int live = a.per_country ? a.count : a.rows;   // states that share a network
    if (live >= 4 * MLP_TILE_ROWS && H == a.nl.Hp && (H == 64 || H == 128)) {   // a large batch on one network
        const dim3 grid((unsigned)((live + MLP_TILE_ROWS - 1) / MLP_TILE_ROWS), a.per_country ? (unsigned)a.n : 1u);
        if (a.certify && a.act_episode && !a.act_out && !a.q_out && !a.per_country) {
            // the action alone is wanted: fused screen, then the reference's arithmetic where the screen was not sure
            if (H == 64) {
                hipLaunchKernelGGL((mlp_tile_kernel<64, 1>), grid, dim3(MLP_TILE_THREADS), 0, s, a);
                hipLaunchKernelGGL((mlp_tile_kernel<64, 2>), grid, dim3(MLP_TILE_THREADS), 0, s, a);
            } else {
                hipLaunchKernelGGL((mlp_tile_kernel<128, 1>), grid, dim3(MLP_TILE_THREADS), 0, s, a);
                hipLaunchKernelGGL((mlp_tile_kernel<128, 2>), grid, dim3(MLP_TILE_THREADS), 0, s, a);
            }
            return hipGetLastError();
        }
        if (H == 64) hipLaunchKernelGGL((mlp_tile_kernel<64, 0>), grid, dim3(MLP_TILE_THREADS), 0, s, a);
        else hipLaunchKernelGGL((mlp_tile_kernel<128, 0>), grid, dim3(MLP_TILE_THREADS), 0, s, a);
        return hipGetLastError();
    }
    if (live <= 8 && H == a.nl.Hp && (H == 64 || H == 128)) {
        // a handful of states per network: the latency-first kernel, two states per workgroup (the workgroups of one
        // launch run side by side on different CUs, and two states' products still fit the registers)
        if (H == 64) return live <= 1 ? launch_mlp_chain<64, 1>(a, live, s) : launch_mlp_chain<64, 2>(a, live, s);
        return live <= 1 ? launch_mlp_chain<128, 1>(a, live, s) : launch_mlp_chain<128, 2>(a, live, s);
    }
    if (live <= 1) return launch_mlp_rows<1>(a, live, s);
    if (live <= 2) return launch_mlp_rows<2>(a, live, s);
    if (live <= 4) return launch_mlp_rows<4>(a, live, s);
    if (live <= 8) return launch_mlp_rows<8>(a, live, s);
    return launch_mlp_rows<MLP_MAXROWS>(a, live, s);
}

// ---------------------------------------------------------------------------------------------
// EpidemicEnv.allocateInt for `batch` independent rows of n <= 1024 weights: one wave per row.
// The sequential weight sum (weights.sum, env.scala:242) is done by lane 0 through LDS so that
// it rounds like the reference.
// ---------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(64) void alloc_batch_kernel(const AllocBatchArgs a) {
    __shared__ double sh_q[R * 64];
    __shared__ double sh_sum;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = a.n;
    const double* w = a.w + (size_t)b * n;
    for (int j = lane; j < R * 64; j += 64) sh_q[j] = j < n ? w[j] : 0.0;
    __syncthreads();
    if (lane == 0) {
        double sum = n > 0 ? sh_q[0] : 0.0;
        for (int j = 1; j < n; ++j) sum = sum + sh_q[j];
        sh_sum = sum;
    }
    __syncthreads();
    const double sum = sh_sum;
    const int32_t total = a.totals[b];
    uint32_t acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0u;
    if (total > 0 && sum > 0.0) {
#pragma unroll
        for (int r = 0; r < R; ++r) sh_q[r * 64 + lane] = sh_q[r * 64 + lane] / sum;
        __syncthreads();
        apportion_row<R>(total, n, -1, sh_q, acc, lane, nullptr);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int j = r * 64 + lane;
        if (j < n) a.out[(size_t)b * n + j] = (int32_t)acc[r];
    }
}

hipError_t launch_alloc_batch(const AllocBatchArgs& a, hipStream_t s) {
    if (a.batch <= 0 || a.n <= 0) return hipSuccess;
    const dim3 grid((unsigned)a.batch), block(64);
    switch (dense_slots_for(a.n)) {
        case 1: hipLaunchKernelGGL(alloc_batch_kernel<1>, grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL(alloc_batch_kernel<2>, grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL(alloc_batch_kernel<4>, grid, block, 0, s, a); break;
        case 8: hipLaunchKernelGGL(alloc_batch_kernel<8>, grid, block, 0, s, a); break;
        case 16: hipLaunchKernelGGL(alloc_batch_kernel<16>, grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// EpidemicEnv.step for a batch of independent rows (env.scala:58-140), reward and done included.
// ---------------------------------------------------------------------------------------------
__global__ void env_batch_kernel(const EnvBatchArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.batch) return;
    const epi_state st = a.in[b];
    const StepConsts c = make_step_consts(a.env[b], a.prof[b]);
    int act = a.action[b];
    act = act < 0 ? 0 : (act > 7 ? 7 : act);
    const int32_t s0 = round_count(st.s), i0 = round_count(st.i), r0 = round_count(st.r), d0 = round_count(st.d),
                  v0 = round_count(st.v);
    const LocalOut o = local_step(c, s0, i0, r0, d0, v0, st.hosp_cap, act, a.gauss[b]);
    epi_state nx = st;
    nx.s = (double)o.s; nx.i = (double)o.i; nx.r = (double)o.r; nx.d = (double)o.d; nx.v = (double)o.v;
    nx.hosp_cap = o.cap;
    nx.t = (int32_t)((uint32_t)st.t + 1u);
    a.out[b] = nx;
    if (a.reward) {
        const RewardConsts rc = make_reward_consts(a.env[b], a.prof[b]);
        a.reward[b] = reward_of(rc, st.i, st.d, o, act);
    }
    if (a.done) {
        const double nn = fmax(1.0, nx.s + nx.i + nx.r + nx.d);
        a.done[b] = (nx.t >= nx.t_max) || (nx.i / nn < 1e-5) || (nx.d / nn > 0.02);
    }
}

hipError_t launch_env_batch(const EnvBatchArgs& a, hipStream_t s) {
    if (a.batch <= 0) return hipSuccess;
    hipLaunchKernelGGL(env_batch_kernel, dim3((unsigned)((a.batch + 127) / 128)), dim3(128), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Batched worlds: the caller's arrays are compact [batch][n]; the episode kernels work on planes
// padded to npad columns.  pack spreads actions (clamped as Action.fromId, env.scala:18-20) and
// noise draws over the planes and clears the error table; unpack gathers State records
// (env.scala:22-25), rewards and the chosen actions back into compact arrays.
// ---------------------------------------------------------------------------------------------
__global__ void world_pack_kernel(const WorldIoArgs a) {
    const int cells = a.batch * a.n;
    const int stride = gridDim.x * blockDim.x;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < cells || k < a.batch * 4; k += stride) {
        if (k < a.batch * 4) a.err[k] = 0;
        if (k >= cells) continue;
        const int b = k / a.n, c = k - b * a.n;
        const size_t o = (size_t)b * a.npad + c;
        if (a.actions) {
            const int32_t act = a.actions[k];
            a.act_pad[o] = (int8_t)(act < 0 ? 0 : (act > 7 ? 7 : act));
        }
        a.gauss_pad[o] = a.gauss ? a.gauss[k] : a.gauss_row[c];
    }
}

__global__ void world_unpack_kernel(const WorldIoArgs a) {
    const int cells = a.batch * a.n;
    const size_t plane = (size_t)a.batch * a.npad;
    const int stride = gridDim.x * blockDim.x;
    if (a.err_out)
        for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < a.batch * 4; k += stride) a.err_out[k] = a.err[k];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < cells; k += stride) {
        const int b = k / a.n, c = k - b * a.n;
        const size_t o = (size_t)b * a.npad + c;
        if (a.states_out) {
            epi_state s = a.base;
            s.s = (double)a.st[0 * plane + o];
            s.i = (double)a.st[1 * plane + o];
            s.r = (double)a.st[2 * plane + o];
            s.d = (double)a.st[3 * plane + o];
            s.v = (double)a.st[4 * plane + o];
            s.hosp_cap = a.cap[o];
            s.t = a.tcur[o];
            a.states_out[k] = s;
        }
        if (a.packed_out) {
            epi_state_packed s;
            s.s = a.st[0 * plane + o];
            s.i = a.st[1 * plane + o];
            s.r = a.st[2 * plane + o];
            s.d = a.st[3 * plane + o];
            s.v = a.st[4 * plane + o];
            s.t = a.tcur[o];
            s.hosp_cap = a.cap[o];
            a.packed_out[k] = s;
        }
        if (a.rewards_out) a.rewards_out[k] = a.reward_pad[o];
        if (a.actions_out) a.actions_out[k] = (int32_t)a.act_pad[o];
    }
}

static unsigned io_blocks(int items) {
    const long long b = ((long long)items + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

hipError_t launch_world_pack(const WorldIoArgs& a, hipStream_t s) {
    if (a.batch <= 0) return hipSuccess;
    const int items = a.batch * (a.n > 4 ? a.n : 4);
    hipLaunchKernelGGL(world_pack_kernel, dim3(io_blocks(items)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_world_unpack(const WorldIoArgs& a, hipStream_t s) {
    if (a.batch <= 0) return hipSuccess;
    hipLaunchKernelGGL(world_unpack_kernel, dim3(io_blocks(a.batch * (a.n > 4 ? a.n : 4))), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Dense table preparation (see PrepareRowsArgs): one workgroup per row.  Every sum the reference forms left to
// right (scaleRow's row sum, wRow.sum) is formed by one thread in that order; the element-wise products and
// quotients are independent and spread over the threads.  The top-weights list is a bitonic sort in LDS.
// ---------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(256) void prepare_rows_kernel(const PrepareRowsArgs a) {
    __shared__ double sh[NP];
    __shared__ double sh_s;
    const int n = a.n, tid = threadIdx.x, b = blockIdx.x;
    const bool variant = a.travel != nullptr;
    const int self = variant ? a.seed_idx : b;
    const double* __restrict__ src = a.W + (size_t)self * n;
    for (int j = tid; j < NP; j += 256) sh[j] = j < n ? src[j] : 0.0;
    __syncthreads();
    if (variant) {   // ToyConn.scaleRow: off-diagonals times factor, row sum capped at 0.2, diagonal 0
        const double factor = a.travel[b];
        for (int j = tid; j < n; j += 256)
            if (j != self) sh[j] *= factor;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int j = 0; j < n; ++j)
                if (j != self) s += sh[j];
            sh_s = s;
        }
        __syncthreads();
        const double s = sh_s, cap = fmin(0.2, s);
        if (s > cap && s > 0) {
            const double sc = cap / s;
            for (int j = tid; j < n; j += 256)
                if (j != self) sh[j] *= sc;
        }
        if (tid == 0) sh[self] = 0.0;
        __syncthreads();
    }
    if (tid == 0) {
        sh[self] = 0.0;   // wRow: the origin's own entry is 0 (WorldToy.scala:113)
        double sum = 0.0;
        for (int j = 0; j < n; ++j) sum = (j == 0) ? sh[0] : sum + sh[j];
        sh_s = sum;
    }
    __syncthreads();
    const double sum = sh_s;
    if (tid == 0) a.sumw[b] = fmax(0.0, sum);
    for (int j = tid; j < NP; j += 256) {
        const double qv = (j < n && sum > 0.0) ? sh[j] / sum : 0.0;
        a.q[(size_t)b * a.npad + j] = qv;
        sh[j] = qv;
    }
    __syncthreads();
    // descending bitonic sort of the NP normalised weights (zeros of the padding sort last)
    for (int k = 2; k <= NP; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < NP / 2; t += 256) {
                const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
                const bool desc = ((lo & k) == 0);
                const double x = sh[lo], y = sh[hi];
                if (desc ? (x < y) : (x > y)) { sh[lo] = y; sh[hi] = x; }
            }
            __syncthreads();
        }
    for (int j = tid; j < kTopWeights; j += 256) a.qtop[(size_t)b * kTopWeights + j] = j < NP ? sh[j] : 0.0;
}

hipError_t launch_prepare_rows(const PrepareRowsArgs& a, hipStream_t s) {
    if (a.n_rows <= 0) return hipSuccess;
    const dim3 grid((unsigned)a.n_rows), block(256);
    switch (a.npad) {
        case 64: hipLaunchKernelGGL(prepare_rows_kernel<64>, grid, block, 0, s, a); break;
        case 128: hipLaunchKernelGGL(prepare_rows_kernel<128>, grid, block, 0, s, a); break;
        case 256: hipLaunchKernelGGL(prepare_rows_kernel<256>, grid, block, 0, s, a); break;
        case 512: hipLaunchKernelGGL(prepare_rows_kernel<512>, grid, block, 0, s, a); break;
        case 1024: hipLaunchKernelGGL(prepare_rows_kernel<1024>, grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// ToyConn.gravityMatrix (ToyConn.scala:22-47) with Geo.haversineKm (Geo.scala:72-79): one workgroup
// per row i.  The threads fill the row tile by tile; one thread adds the tile up from left to right
// (the reference's rowSum order) before all of them apply the row cap.  pow / sin / cos / atan2 are
// the device library's: like the JVM's intrinsics against a host libm, they may differ from another
// implementation in the last place, so the matrix is an input of the engine, not a pinned output.
// ---------------------------------------------------------------------------------------------
constexpr int kGravityTile = 4096;

__device__ __forceinline__ double haversine_km(double lat1, double lon1, double lat2, double lon2) {
    const double rad = 0.017453292519943295;  // Math.toRadians
    const double dLat = (lat2 - lat1) * rad, dLon = (lon2 - lon1) * rad;
    const double sa = sin(dLat / 2), so = sin(dLon / 2);
    const double h = sa * sa + cos(lat1 * rad) * cos(lat2 * rad) * (so * so);  // Math.pow(x, 2) == x * x exactly
    const double c = 2 * atan2(sqrt(h), sqrt(1 - h));
    return 6371.0 * c;
}

// Raw entry (i, j) of the gravity matrix before the row cap (ToyConn.scala:28-38), one expression shared by the
// dense and the sparse builders so that both produce the same doubles.
__device__ __forceinline__ double gravity_entry(const GravityArgs& a, int i, int j, double lat_i, double lon_i, double pa) {
    if (j == i) return 0.0;
    const double d = fmax(300.0, haversine_km(lat_i, lon_i, a.lat[j], a.lon[j]));
    const double f = a.flight ? a.flight[(size_t)i * a.n + j] : 1.0;
    return a.k * pa * pow((double)a.pop[j], a.b) * f / pow(d, a.g);
}

__global__ __launch_bounds__(256) void gravity_rows_kernel(const GravityArgs a) {
    __shared__ double tile[kGravityTile];
    __shared__ double sh_sum;
    const int i = blockIdx.x, n = a.n;
    const double lat_i = a.lat[i], lon_i = a.lon[i];
    const double pa = pow((double)a.pop[i], a.a);
    double* __restrict__ row = a.W + (size_t)i * n;
    if (threadIdx.x == 0) sh_sum = 0.0;
    for (int j0 = 0; j0 < n; j0 += kGravityTile) {
        const int cnt = min(kGravityTile, n - j0);
        for (int jj = threadIdx.x; jj < cnt; jj += blockDim.x) {
            const int j = j0 + jj;
            const double w = gravity_entry(a, i, j, lat_i, lon_i, pa);
            tile[jj] = w;
            row[j] = w;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = sh_sum;
            for (int jj = 0; jj < cnt; ++jj)
                if (j0 + jj != i) s += tile[jj];
            sh_sum = s;
        }
        __syncthreads();
    }
    const double row_sum = sh_sum;
    if (row_sum > 0) {
        const double scale = fmin(1.0, a.cap_row / row_sum);
        for (int j = threadIdx.x; j < n; j += blockDim.x)
            if (j != i) row[j] *= scale;
    }
}

__global__ __launch_bounds__(256) void gravity_csr_kernel(const GravityCsrArgs c) {
    __shared__ double tile[kGravityTile];
    __shared__ double sh_sum;
    __shared__ int sh_cnt[4], sh_base;
    const GravityArgs& a = c.g;
    const int i = blockIdx.x, n = a.n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double lat_i = a.lat[i], lon_i = a.lon[i];
    const double pa = pow((double)a.pop[i], a.a);
    if (c.mode == 0) {   // row sum, left to right (ToyConn.scala:39), tile by tile
        if (tid == 0) sh_sum = 0.0;
        for (int j0 = 0; j0 < n; j0 += kGravityTile) {
            const int cnt = min(kGravityTile, n - j0);
            for (int jj = tid; jj < cnt; jj += 256) tile[jj] = gravity_entry(a, i, j0 + jj, lat_i, lon_i, pa);
            __syncthreads();
            if (tid == 0) {
                double s = sh_sum;
                for (int jj = 0; jj < cnt; ++jj)
                    if (j0 + jj != i) s += tile[jj];
                sh_sum = s;
            }
            __syncthreads();
        }
        if (tid == 0) c.scale[i] = sh_sum > 0 ? fmin(1.0, a.cap_row / sh_sum) : 1.0;
        return;
    }
    const double scale = c.scale[i];
    if (tid == 0) sh_base = 0;
    __syncthreads();
    // columns in runs of 256, in order: every stored entry's position is the number of stored entries before it
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + tid;
        double w = 0.0;
        if (j < n && j != i) w = gravity_entry(a, i, j, lat_i, lon_i, pa) * scale;
        const bool keep = j < n && j != i && w > c.threshold;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) sh_cnt[wave] = __popcll(m);
        __syncthreads();
        int before = sh_base;
        for (int k = 0; k < wave; ++k) before += sh_cnt[k];
        if (keep && c.mode == 2) {
            const size_t at = (size_t)c.rowptr[i] + before + (int)lanes_below(m);
            c.col[at] = j;
            c.val[at] = w;
        }
        __syncthreads();
        if (tid == 0) sh_base += sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3];
        __syncthreads();
    }
    if (c.mode == 1 && tid == 0) c.counts[i] = sh_base;
}

hipError_t launch_gravity_csr(const GravityCsrArgs& a, hipStream_t s) {
    if (a.g.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gravity_csr_kernel, dim3((unsigned)a.g.n), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Sparse graph tables on the device (SparseBuildArgs).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sparse_normalise_kernel(const SparseBuildArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int rs = a.rowptr[i], re = a.rowptr[i + 1];
    double sum = 0.0;
    for (int k = rs; k < re; ++k) {
        const double w = (a.col[k] == i) ? 0.0 : a.val[k];
        sum = (k == rs) ? w : sum + w;      // left to right, as wRow.sum adds them
    }
    a.sumw_rows[i] = fmax(0.0, sum);
    for (int k = rs; k < re; ++k) {
        const double w = (a.col[k] == i) ? 0.0 : a.val[k];
        a.qcsr[k] = (sum > 0.0) ? w / sum : 0.0;
    }
}

// grid: n_blocks workgroups of 64 lanes for the ELL blocks, then n_long for the long rows, then ceil(n / 64) for the
// per-patch tables.
__global__ __launch_bounds__(64) void sparse_fill_kernel(const SparseBuildArgs a, int n_blocks) {
    const int lane = threadIdx.x;
    int item = blockIdx.x;
    if (item < n_blocks) {
        const int n_ell = a.n_ell;
        const int p = item * 64 + lane;
        const int i = p < n_ell ? a.order[p] : -1;
        const int rs = i >= 0 ? a.rowptr[i] : 0, deg = i >= 0 ? a.rowptr[i + 1] - rs : 0;
        const EllBlock blk = a.blocks[item];
        const int slots = (blk.slots + kSlotGroup - 1) / kSlotGroup * kSlotGroup;
        const size_t off = (size_t)blk.off;
        for (int k = 0; k < slots; ++k) {
            const bool has = k < deg;
            // slots stay in the caller's column order; a slot that holds no weight names the row's own patch (the
            // mobility routine adds its share of 0 without looking at the column: every column must be a valid cell)
            const int c = has ? a.col[rs + k] : (i >= 0 ? i : 0);
            a.ell_q[(off + k) * 64 + lane] = has ? a.qcsr[rs + k] : 0.0;
            a.ell_col[(off + k) * 64 + lane] = a.col_map ? a.col_map[c] : c;
        }
        return;
    }
    item -= n_blocks;
    if (item < a.n_long) {
        const int n_ell = a.n_ell;
        const int i = a.order[n_ell + item];
        const int rs = a.rowptr[i], deg = a.rowptr[i + 1] - rs;
        const size_t lo = (size_t)a.long_off[item];
        // the compact copy spans whole groups of 64 slots up to the next row's copy: the host cleared it (0 / -1)
        for (int k = lane; k < deg; k += 64) {
            const int c = a.col[rs + k];
            a.long_q[lo + k] = a.qcsr[rs + k];
            a.long_col[lo + k] = a.col_map ? a.col_map[c] : c;
        }
        if (lane == 0) a.long_rows_out[item] = a.col_map ? a.col_map[i] : i;
        return;
    }
    item -= a.n_long;
    const int c = item * 64 + lane;
    if (c < a.n) {
        const int dst = a.col_map ? a.col_map[c] : c;
        a.pop_out[dst] = a.pop_in[c];
        a.sumw_out[dst] = a.sumw_rows[c];
    }
}

hipError_t launch_sparse_normalise(const SparseBuildArgs& a, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(sparse_normalise_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_sparse_fill(const SparseBuildArgs& a, int n_blocks, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)(n_blocks + a.n_long + (a.n + 63) / 64);
    hipLaunchKernelGGL(sparse_fill_kernel, dim3(grid), dim3(64), 0, s, a, n_blocks);
    return hipGetLastError();
}

hipError_t launch_gravity(const GravityArgs& a, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gravity_rows_kernel, dim3((unsigned)a.n), dim3(256), 0, s, a);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------
// Kernel attributes of the current device.  The limit on dynamic LDS is a per-device property of a kernel; the
// kernels that may ask for more than the 48 KB default get the largest size they can ever be launched with, once
// per device (handle creation calls this, so no launch path has to), and every later call is one flag test.
// ---------------------------------------------------------------------------------------------
hipError_t prepare_device_kernels() {
    constexpr int kMaxDevices = 64;
    static std::atomic<bool> ready[kMaxDevices];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    if (ready[dev].load(std::memory_order_acquire)) return hipSuccess;
    auto lds_limit = [](const void* fn, size_t bytes) {
        return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    };
    if ((e = lds_limit(reinterpret_cast<const void*>(&episode_csr_kernel<16>), kMaxLdsBytes)) != hipSuccess) return e;
    if ((e = lds_limit(reinterpret_cast<const void*>(&episode_csr_kernel<24>), kMaxLdsBytes)) != hipSuccess) return e;
    if ((e = lds_limit(reinterpret_cast<const void*>(&episode_csr_kernel<kLaneSlots>), kMaxLdsBytes)) != hipSuccess) return e;
    if ((e = lds_limit(reinterpret_cast<const void*>(&mlp_greedy_kernel<1>), mlp_lds_bytes(MLP_MAXH, 1))) != hipSuccess) return e;
    if ((e = lds_limit(reinterpret_cast<const void*>(&mlp_greedy_kernel<2>), mlp_lds_bytes(MLP_MAXH, 2))) != hipSuccess) return e;
    if ((e = lds_limit(reinterpret_cast<const void*>(&mlp_greedy_kernel<4>), mlp_lds_bytes(MLP_MAXH, 4))) != hipSuccess) return e;
    if ((e = lds_limit(reinterpret_cast<const void*>(&mlp_greedy_kernel<8>), mlp_lds_bytes(MLP_MAXH, 8))) != hipSuccess) return e;
    if ((e = lds_limit(reinterpret_cast<const void*>(&mlp_greedy_kernel<MLP_MAXROWS>), mlp_lds_bytes(MLP_MAXH, MLP_MAXROWS))) != hipSuccess) return e;
    ready[dev].store(true, std::memory_order_release);
    return hipSuccess;
}

}  // namespace epi
