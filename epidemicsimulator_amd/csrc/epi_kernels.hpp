// Kernel argument blocks and launch entry points shared by the kernels (epi_kernels.hip)
// and the host engine (epi_capi.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "epi_device.hpp"

namespace epi {

constexpr int kMaxDenseR = 16;               // register slots per lane in the dense row kernel
constexpr int kMaxDenseN = kMaxDenseR * 64;  // widest dense row held in one wave's registers
constexpr size_t kMaxLdsBytes = 160 * 1024;  // LDS of one CU: bounds the patches of a sparse world
constexpr int kMaxSparseDegree = 256;         // longest stored row apportioned in registers (one wave, 4 slots per lane);
                                              // longer rows are streamed from memory by one wave
constexpr int kLaneSlots = 32;                // longest row a single lane keeps in registers
constexpr int kSlotGroup = 8;                 // ELL blocks are stored in whole groups of this many slots
size_t csr_lds_bytes(int n);

// Indices into the per-country constant table cst[k][npad] (StepConsts minus beta0).
enum { CST_GAMMA = 0, CST_IFR0, CST_CNOISE, CST_EFF1, CST_EFF2, CST_EFF3, CST_SURGE, CST_VAX6, CST_VAX7, CST_COUNT };

// One launch of the fused dense episode kernel: steps [t_begin, t_end) of `count` episodes, one
// workgroup per episode, npad threads (one per padded country, one wave per 64).
// One block of 64 rows of a sparse graph in the slot-major layout.
struct EllBlock {
    int32_t slots;  // longest row of the block
    int32_t off;    // first slot of the block in col / qval (in units of 64 entries)
};

struct EpisodeArgs {
    int n, npad, count;
    int seed_idx, n_travel;  // travel variant of the seed row: g % n_travel for grid point g; < 0 = none
    int g_begin, g_stride;   // episode e runs flat grid point g = g_begin + e * g_stride (beta index = g / n_travel)
    int t_begin, t_end;
    int step_t_max;          // the tMax argument of stepOne (WorldToy.scala:135)
    int validate;
    int shared_actions;      // `actions[t][*]` holds one value per step (a sweep's agents share seed and hyper-parameters, so they
                             // draw the same exploratory action: DDQNAgent.scala:14): the sparse kernel then fetches only the
                             // constant that action uses
    int uniform_clock;       // every patch of an episode carries the same clock (sweeps: one initial state for all): the sparse
                             // kernel then keeps it in a register and writes the plane once, at the end of the launch
    const double* q;         // [n][npad]  w_ij / sumW_i, diagonal and padding 0
    const double* sumw;      // [npad]     max(0, sum_j wRow_j)  (WorldToy.scala:114)
    const double* qseed;     // [n_travel][npad] seed-row variants (ToyConn.scaleRow per travel scale)
    const double* sumw_seed; // [n_travel]
    // sparse graphs (rowptr != NULL): the rows of at most kLaneSlots weights, sorted by length (longest
    // first), in blocks of 64, slot-major (ELL) inside a block: slot k of row ell_rows[64*b + l] is element
    // (ell[b].off + k) * 64 + l of col / qval
    const int32_t* rowptr;    // [n+1] (marks sparse mode; row lengths)
    const EllBlock* ell;      // [n_blocks]
    const int32_t* ell_rows;  // [n_blocks][64] row of each lane, -1 = no row
    const int32_t* block_rows;  // the same table, also present (position p -> p) when ell_rows is NULL: the mobility phase reads its
                                // rows from it on one code path
    const int32_t* ell_pos;   // [npad] inverse: position 64*b + l of a row in the lane layout, -1 = long row
    int n_blocks;
    int n_ell;                // rows held in the blocks; with ell_rows == ell_pos == NULL row 64*b + l IS patch 64*b + l (patches numbered
                              // in lane-layout order, long rows last): no second copy of the exports, no scattered stores
    int seed_block;           // block that holds the seed row, -1 if it is a long row (or there are no variants)
    const int32_t* col;       // ELL columns (patch ids in the numbering the kernel runs in; slots in the caller's ascending
                              // column order, which is the order ties are broken in), -1 = padding; every block occupies a whole
                              // number of slot groups (kSlotGroup), the extra slots holding weight 0
    const double* qval;       // ELL normalised weights w_ij / sumW_i (a stored diagonal entry holds 0)
    const double* qval_seed;  // per travel scale: a copy [seed_deg][64] of the seed row's block with that row replaced,
                              // or, when the seed row is a long row, the row itself [seed_deg = 64]
    int seed_deg, max_deg;    // slots per variant in qval_seed; longest row held in the lane layout
    int seed_long;            // the seed row is one of the long rows below
    int n_long;               // rows longer than kLaneSlots: compact copies, 64 slots each
    const int32_t* long_rows; // [n_long]
    const int32_t* long_off;  // [n_long] first element of the row's compact copy in long_q / long_col
    const double* long_q;     // compact copies: a row of deg weights occupies 64 * dense_slots_for(deg) elements
    const int32_t* long_col;  // same shape, -1 = padding
    int32_t* exports;        // [4][count][npad] words of scratch of the sparse kernel (8-byte aligned): pairs {exportS, exportI}
                             // by patch in the first half, the same pairs by position in the lane layout in the second
    const double* qtop;      // [n][kTopWeights] each row's largest normalised weights, descending (or NULL)
    const double* qtop_seed; // [n_travel][kTopWeights]
    const double* beta0;     // [n_beta][npad] or [1][npad]
    int beta_rows;           // rows in beta0; episode uses row min(g / n_travel, beta_rows-1)
    const double* cst;       // [CST_COUNT][npad]
    const int8_t* actions;   // [.][npad] shared per-step actions, indexed by t - act_t0
    int act_t0;
    const int8_t* act_episode;  // optional [count][npad]: actions of every episode for step t_begin
    const double* noise;     // [.][npad] shared per-step Gaussian draws, indexed by t - noise_t0 ...
    int noise_t0;
    const double* noise_episode;  // optional [count][npad]: draws of every episode for step t_begin
    int32_t* st;             // [5][count][npad] s,i,r,d,v
    double* cap;             // [count][npad]
    int32_t* tcur;           // [count][npad] State.t
    int32_t* arrival;        // [count][npad] first step index with I >= 10, or -1
    int32_t* peak;           // [count][npad] running max of I
    int32_t* err;            // [count][4] code, node, t, -
    // optional outputs of the world API
    const RewardConsts* rwd; // [npad] or NULL
    double* reward_out;      // [count][npad]
    int32_t* flow_s;         // [count][npad] movedS of the last step (WorldToy.scala:106-126)
    int32_t* flow_i;
    unsigned long long* stats;  // optional [32] search statistics: rows, probes, cold starts, ties, histogram
    // compact caller-side arrays of the batched world step, read and written by the kernel itself (small worlds:
    // episode_tiny_kernel; one-step launches only).  They sit in page-locked host memory mapped into the device.
    const int32_t* io_actions;   // [count][n] caller's action ids (clamped as Action.fromId), or NULL
    const double* io_gauss;      // [count][n] noise draws, or NULL: io_gauss_row[n], one draw per country for all instances
    const double* io_gauss_row;
    epi_state* io_states;        // [count][n] State records out (t_max and seed from io_t_max / io_seed), or NULL
    epi_state_packed* io_packed; // [count][n] the same states as 32-byte records, or NULL
    double* io_rewards;          // [count][n] or NULL
    int32_t* io_err;             // [count][4] copy of the error table
    int32_t io_t_max, io_seed;
    int32_t* loc_s;          // optional [count][npad]: S and I right after env.step, before the mobility flows are
    int32_t* loc_i;          // applied -- the s2 of the Transition the agents observe (WorldToy.scala:90,99)
    double* series_i;        // optional [.][count][n]: I of every country after step t at index t - series_t0
    int series_t0;           // (the series the trainer's evaluation collects, MutliTrainer.scala:139)
};

struct InitArgs {
    int n, npad, count, seed_idx;
    const int32_t* pop;  // [npad]
    double base_cap;
    int32_t r0, d0, v0, t0;
    int32_t* st; double* cap; int32_t* tcur; int32_t* arrival; int32_t* peak; int32_t* err;
};

struct LossArgs {
    int n, npad, count, steps;
    const int32_t* arrival; const int32_t* peak;
    const double* t_arrive; const uint8_t* has_arrive; const double* t_peak; const uint8_t* has_peak;
    double* loss; int32_t* arrival_out; double* peak_out;  // arrival_out/peak_out [count][n] optional
    const int32_t* id_of;  // optional [n]: arrival / peak hold country c at index id_of[c] (engine-internal numbering)
};

// One network of DDQNAgent (MLP.scala:4-13: 7 -> H -> H -> 8) in device memory, every matrix with row stride Hp
// (hidden rounded up to 64, zero padded):
//   W1t[7][Hp]  b1[Hp]  W2[Hp][Hp] (row = output)  W2t[Hp][Hp] (row = input)  b2[Hp]  W3[8][Hp]  b3[8]
struct NetLayout {
    int H, Hp;
    __host__ __device__ size_t w1t() const { return 0; }
    __host__ __device__ size_t b1() const { return (size_t)7 * Hp; }
    __host__ __device__ size_t w2() const { return b1() + Hp; }
    __host__ __device__ size_t w2t() const { return w2() + (size_t)Hp * Hp; }
    __host__ __device__ size_t b2() const { return w2t() + (size_t)Hp * Hp; }
    __host__ __device__ size_t w3() const { return b2() + Hp; }
    __host__ __device__ size_t b3() const { return w3() + (size_t)8 * Hp; }
    __host__ __device__ size_t size() const { return b3() + 8; }
};

// Host image of one network in the device layout from the reference's arrays W1[H][7], b1[H], W2[H][H], b2[H],
// W3[8][H], b3[8] (MLP.scala:4-13); NULL biases are zero.  `out` holds nl.size() doubles.
inline void pack_net(const NetLayout& nl, const double* W1, const double* b1, const double* W2, const double* b2,
                     const double* W3, const double* b3, double* out) {
    for (size_t k = 0; k < nl.size(); ++k) out[k] = 0.0;
    const int H = nl.H, Hp = nl.Hp;
    for (int i = 0; i < H; ++i) {
        for (int j = 0; j < 7; ++j) out[nl.w1t() + (size_t)j * Hp + i] = W1[(size_t)i * 7 + j];
        out[nl.b1() + i] = b1 ? b1[i] : 0.0;
        for (int j = 0; j < H; ++j) {
            out[nl.w2() + (size_t)i * Hp + j] = W2[(size_t)i * H + j];
            out[nl.w2t() + (size_t)j * Hp + i] = W2[(size_t)i * H + j];
        }
        out[nl.b2() + i] = b2 ? b2[i] : 0.0;
    }
    for (int o = 0; o < 8; ++o) {
        for (int j = 0; j < H; ++j) out[nl.w3() + (size_t)o * Hp + j] = W3[(size_t)o * H + j];
        out[nl.b3() + o] = b3 ? b3[o] : 0.0;
    }
}

// DDQNAgent.actGreedy for `rows` (episode, country) pairs, row = e * n + c.  With per_country != 0 country c asks
// network c (nets + c * net_stride): the per-node agents of WorldToy.scala:46-65.
struct MlpArgs {
    int rows;          // (episode, country) pairs
    int n, npad, count;  // countries, padded plane width, episodes (rows = count * n)
    int state_t_max;
    const int32_t* st; const double* cap; const int32_t* tcur; size_t plane;  // episode state layout
    const epi_state* states;  // alternative input: rows states (AoS), used when st == NULL (then n = 1)
    NetLayout nl;
    const double* nets;       // network(s) in NetLayout form
    size_t net_stride;        // doubles from country c's network to country c+1's (per_country)
    const int32_t* net_index; // optional (per_country): country c asks network net_index[c] instead of network c
    size_t net_offset;        // doubles added to every network's address (the target network of an agent's pair)
    int per_country;
    double clip_q;            // clamp of the outputs (MLP.scala:21), used when clip_ptr == NULL
    const double* clip_ptr;   // per-country clamp: clip_ptr[c * clip_stride]
    int clip_stride;
    int8_t* act_episode;  // [count][npad] (when st != NULL)
    int32_t* act_out;     // [rows] optional
    double* q_out;        // [rows][8] optional
    // Large batches that only want the greedy action (the argmax-Q step of a sweep): certify != 0 lets launch_mlp
    // screen the states with fused multiply-adds first and evaluate in the reference's arithmetic only where the
    // screen cannot certify the argmax (mlp_tile_kernel).  cert = {A1, B1, A2, B2, A3, B3}: the largest absolute row
    // sum of each weight matrix and the largest absolute bias of each layer; cert_scale multiplies the error bound
    // (1 in production; a huge value sends every state to the exact evaluation, for tests).
    int certify;
    double cert[6];
    double cert_scale;
    unsigned long long* cert_stats;   // optional [2]: states the screen looked at, states it flagged for the exact pass
};
constexpr int8_t kActUncertain = 0x40;   // flag on an act_episode entry between the screen and the exact pass

struct AllocBatchArgs {
    int n, npad, batch;
    const int32_t* totals; const double* w; int32_t* out;
};

struct EnvBatchArgs {
    int batch;
    const epi_env_params* env; const epi_profile* prof; const epi_state* in; const int32_t* action;
    const double* gauss; epi_state* out; double* reward; uint8_t* done;
};

// Compact caller-side arrays [batch][n] <-> the padded per-instance planes [batch][npad] of the episode kernels.
struct WorldIoArgs {
    int n, npad, batch;
    // pack: caller inputs -> padded planes (also clears the error table)
    const int32_t* actions;   // [batch][n] or NULL (greedy step: the network kernel writes act_pad)
    const double* gauss;      // [batch][n] or NULL
    const double* gauss_row;  // [n] one draw per country shared by all instances (used when gauss == NULL)
    int8_t* act_pad;          // [batch][npad]
    double* gauss_pad;        // [batch][npad]
    int32_t* err;             // [batch][4]
    // unpack: padded planes -> caller outputs (each optional)
    const int32_t* st; const double* cap; const int32_t* tcur; const double* reward_pad;
    epi_state base;           // t_max, seed copied into every state
    epi_state* states_out;    // [batch][n]
    epi_state_packed* packed_out;   // [batch][n] the same states as 32-byte records
    double* rewards_out;      // [batch][n]
    int32_t* actions_out;     // [batch][n]
    int32_t* err_out;         // [batch][4] copy of the error table (optional)
};

// ToyConn.gravityMatrix (ToyConn.scala:22-47) for n countries: one workgroup per row.
struct GravityArgs {
    int n;
    const int32_t* pop; const double* lat; const double* lon;
    const double* flight;  // [n][n] flightMul((ci, cj)) or NULL = 1.0 everywhere
    double a, b, g, k, cap_row;
    double* W;             // [n][n] out
};

// Table preparation of dense graphs on the device: per row the mobility loop's wRow (WorldToy.scala:113), its
// left-to-right sum (:114), the normalised weights w / sumW (the first rounding of allocateInt, env.scala:244) and
// the row's kTopWeights largest normalised weights in descending order.  Rows [0, n_rows) of W are prepared as they
// are; with `travel` set, block k prepares ToyConn.scaleRow(W, seed_idx, travel[k]) (ToyConn.scala:58-70) of the
// seed row instead -- the per-travel variants of the calibration sweep.
struct PrepareRowsArgs {
    int n, npad;
    const double* W;        // [n][n] row-major
    int n_rows;             // rows to prepare (plain mode) or travel scales (variant mode)
    int seed_idx;           // variant mode: the row that is scaled
    const double* travel;   // [n_rows] or NULL = plain mode
    double* q;              // [n_rows][npad]
    double* sumw;           // [n_rows] (plain mode: the caller passes a buffer of npad entries, zero beyond n)
    double* qtop;           // [n_rows][kTopWeights]
};
hipError_t launch_prepare_rows(const PrepareRowsArgs& a, hipStream_t s);

// What DDQNAgent.observe needs of the last world step, as the world keeps it on the device (epi_capi.cpp): the
// states before the step, S and I right after env.step, the current planes (R, D, V and hospCap are not touched by
// mobility), rewards and clamped action ids.
struct WorldTransitionView {
    int device, n, npad, batch, t_max;
    size_t plane;                                                    // batch * npad
    const int32_t* prev_st; const double* prev_cap; const int32_t* prev_tcur;
    const int32_t* loc_s; const int32_t* loc_i;
    const int32_t* st; const double* cap;
    const double* reward; const int8_t* act;
};
}  // namespace epi
struct epi_world;
namespace epi {
int32_t world_transition_view(epi_world* w, WorldTransitionView* v);
// Marks the world's kept transition as filed: the next epi_agents_observe_world needs a new step first.
void world_transition_consumed(epi_world* w);

// Device view of the networks an epi_agents handle holds (epi_agents.hip).
struct AgentsNetView {
    int device, n_agents;
    NetLayout nl;
    const double* nets; size_t net_stride;
    const double* clip_ptr; int clip_stride;
};
}  // namespace epi
struct epi_agents;
namespace epi {
int32_t agents_net_view(epi_agents* g, AgentsNetView* v);
// Files the transitions of `v` with instance e's country c going to agent e * v.n + c (epi_agents.hip).
int32_t agents_observe_instances(epi_agents* g, const WorldTransitionView& v);

// The same matrix cut to a sparse graph on the device: entries W_ij > threshold (i != j), columns ascending.
// mode 0: scale[i] = the row's cap factor min(1, cap_row / rowSum) (1 for an empty row); mode 1: counts[i] = stored
// entries of row i; mode 2: col / val filled from rowptr.
struct GravityCsrArgs {
    GravityArgs g;          // g.W unused
    double threshold;
    int mode;
    double* scale;          // [n]
    int32_t* counts;        // [n]
    const int32_t* rowptr;  // [n+1]
    int32_t* col; double* val;
};
hipError_t launch_gravity_csr(const GravityCsrArgs& a, hipStream_t s);

// Table preparation of sparse graphs on the device, from the caller's CSR triple (columns ascending inside a row) and
// the row order the host derived from the row lengths:
//   normalise: per row the left-to-right sum of its stored off-diagonal weights (WorldToy.scala:113-114: the zeros of
//     the entries that are not stored change no partial sum) and the normalised weights w / sumW (env.scala:244);
//   fill:      the slot-major ELL arrays of the blocks of 64 rows and the compact copies of the long rows, columns
//     translated through `col_map` (id_of when the engine renumbers the patches, NULL for the caller's numbering),
//     and populations and row sums in the numbering the kernels run in.
struct SparseBuildArgs {
    int n, npad;
    const int32_t* rowptr; const int32_t* col; const double* val;   // caller's CSR
    double* qcsr;            // [nnz] normalised weights in CSR order
    double* sumw_rows;       // [n] row sums by caller's row
    const int32_t* order;    // [n] rows by position: the rows of at most kLaneSlots weights by length, then the long rows
    int n_ell;               // rows held in the blocks
    const EllBlock* blocks;  // [n_blocks] longest row and first slot of each block
    const int32_t* col_map;  // NULL or id_of
    const int32_t* pop_in;   // [n] caller's populations
    int32_t* pop_out;        // [npad] populations by engine id (col_map) or caller's id
    double* sumw_out;        // [npad]
    double* ell_q; int32_t* ell_col;            // [total_slots][64]
    int n_long; const int32_t* long_off;        // [n_long] first element of each long row's compact copy
    double* long_q; int32_t* long_col;
    int32_t* long_rows_out;  // [n_long] long rows in the numbering the kernels run in
};
hipError_t launch_sparse_normalise(const SparseBuildArgs& a, hipStream_t s);
hipError_t launch_sparse_fill(const SparseBuildArgs& a, int n_blocks, hipStream_t s);
hipError_t launch_gravity(const GravityArgs& a, hipStream_t s);
hipError_t launch_world_pack(const WorldIoArgs& a, hipStream_t s);
hipError_t launch_world_unpack(const WorldIoArgs& a, hipStream_t s);
// Sets, once per device, the dynamic-LDS limits of the kernels that may exceed the default (current device).
hipError_t prepare_device_kernels();
hipError_t launch_episode(const EpisodeArgs& a, hipStream_t s);
hipError_t launch_init(const InitArgs& a, hipStream_t s);
hipError_t launch_loss(const LossArgs& a, hipStream_t s);
hipError_t launch_mlp(const MlpArgs& a, hipStream_t s);
hipError_t launch_alloc_batch(const AllocBatchArgs& a, hipStream_t s);
hipError_t launch_env_batch(const EnvBatchArgs& a, hipStream_t s);

inline int dense_slots_for(int n) {  // smallest supported R with 64*R >= n
    int r = 1;
    while (r * 64 < n) r *= 2;
    return r;
}

}  // namespace epi
