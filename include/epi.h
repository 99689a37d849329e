/*
 * epi.h -- C ABI of the MI355X engine for the metapopulation step and the
 * beta x travel calibration sweep.
 *
 * The reference has no FFI on this path: its public Scala API *is* the boundary.
 * Each entry point below names the reference function(s) it stands in for
 * (paths relative to the reference tree, src/main/scala/epidemic/).  A JVM shim
 * keeps the Scala signatures and forwards to these functions through JNI or the
 * Panama FFM linker.
 *
 * Conventions
 *   - plain pointers and sizes only; every array is caller-owned and is not
 *     retained after the call returns, except where a handle documents otherwise
 *   - every function returns an epi_status (0 = ok); epi_last_error() gives the
 *     message of the calling thread's last failure
 *   - a handle (epi_grid, epi_world, epi_agents) serves one caller at a time: calls on the same handle from
 *     several threads must be serialised by the caller; different handles are independent
 *   - all compute runs on the GPU; there is no CPU fallback.  When no HIP device
 *     is usable the functions return EPI_ERR_DEVICE
 *   - invariant violations that make the reference throw
 *     IllegalArgumentException (WorldToy.scala:94,144; Validation.scala:7-10) are
 *     reported as EPI_ERR_LOCAL_MASS / EPI_ERR_WORLD_MASS / EPI_ERR_ASSERT_STATE
 *     with the reference's message text in epi_last_error()
 */
#ifndef EPI_H_
#define EPI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum epi_status {
    EPI_OK = 0,
    EPI_ERR_BAD_ARGUMENT = 1,
    EPI_ERR_DEVICE = 2,
    EPI_ERR_LOCAL_MASS = 3,   /* "mass not conserved (local)"           WorldToy.scala:94  */
    EPI_ERR_WORLD_MASS = 4,   /* "global mass changed during mobility"  WorldToy.scala:144 */
    EPI_ERR_ASSERT_STATE = 5, /* Validation.assertState                 Validation.scala:6-11 */
    EPI_ERR_INT_RANGE = 6,    /* a compartment left the Int range the reference computes in */
    EPI_ERR_UNSUPPORTED = 7   /* an input outside what the kernels implement (message says which) */
} epi_status;

/* Geo.Country fields read by EpidemicEnv (Geo.scala:4-21; env.scala:59-101,142-234). */
typedef struct epi_profile {
    double beta_mul, ifr_mul, noise, population_density, gdp_per_capita, healthcare_capacity, age_elderly,
        connectivity, econ_resilience;
} epi_profile;

/* EpidemicEnv constructor arguments (env.scala:32-41); the seed is passed separately. */
typedef struct epi_env_params {
    double base_beta, base_gamma, base_ifr, noise, reward_clip, risk_factor;
} epi_env_params;

/* State (env.scala:22-25). */
typedef struct epi_state {
    double s, i, r, d, v, hosp_cap;
    int32_t t, t_max, seed, pad_;
} epi_state;

/* HyperParams fields the path reads (config.scala:3-21). */
typedef struct epi_hyper {
    double epsilon_start, epsilon_end;
    int32_t epsilon_decay_steps, hidden;
    double clip_q, reward_clip;
    int32_t batch_size, learn_every;
} epi_hyper;

int32_t epi_version(void);
/* Copies the calling thread's last error message (NUL-terminated) and returns its length. */
int32_t epi_last_error(char *buf, size_t cap);
/* The epi_status of the calling thread's last failure (what a *_create that returned NULL failed with). */
int32_t epi_last_status(void);
/* Number of usable HIP devices (0 when none). */
int32_t epi_device_count(void);

/* Device of the calling thread's unit operations below (epi_alloc_int*, epi_env_step_batch, epi_mlp_greedy_batch);
 * 0 until set.  Handles (epi_grid, epi_world, epi_agents) name their device in their descriptors instead.  The
 * unit operations keep their device staging for the life of the process: after the first call of a given size
 * they allocate nothing. */
int32_t epi_set_device(int32_t device);

/* ---- EpidemicEnv.allocateInt (env.scala:241-254) ------------------------------------------- */
/* out[n] = largest-remainder split of `total` over weights w[n]. */
int32_t epi_alloc_int(int32_t total, const double *w, int32_t n, int32_t *out);
/* B independent splits in one launch: totals[B], w[B][n] row-major, out[B][n]. */
int32_t epi_alloc_int_batch(const int32_t *totals, const double *w, int32_t n, int32_t batch, int32_t *out);

/* ---- EpidemicEnv.step incl. reward and done (env.scala:58-234) -------------------------------- */
/* B independent (env, state, action) rows.  gauss[b] is the value rnd.nextGaussian() returns at
 * env.scala:91 for that row.  Compartments must be finite; they are rounded as env.scala:81-85 does. */
int32_t epi_env_step_batch(const epi_env_params *env, const epi_profile *prof, const epi_state *in,
                           const int32_t *action, const double *gauss, epi_state *out, double *reward,
                           uint8_t *done, int32_t batch);

/* ---- ToyConn.scaleRow (ToyConn.scala:58-70) --------------------------------------------------- */
/* out[n*n] = copy of W with row `row` scaled by factor and its sum capped at 0.2 (host arithmetic,
 * same operation order). */
int32_t epi_scale_row(const double *W, int32_t n, int32_t row, double factor, double *out);

/* ---- ToyConn.gravityMatrix (ToyConn.scala:22-47) with Geo.haversineKm (Geo.scala:72-79) --------- */
/* W[n*n] row-major for n countries (pop, lat, lon), computed on `device`.  flight_mul[n*n] holds
 * flightMul((ci, cj)) (ToyConn.scala:16-20) or is NULL for 1.0 everywhere.  The reference's defaults are
 * a = 0.2, b = 1.0, g = 1.2, k = 1e-6, cap_row = 0.12.  The transcendental functions are the device
 * library's: entries may differ from a JVM or host-libm evaluation in the last place. */
int32_t epi_gravity_matrix(int32_t n, const int32_t *pop, const double *lat, const double *lon,
                           const double *flight_mul, double a, double b, double g, double k, double cap_row,
                           double *W, int32_t device);

/* The same matrix cut to a sparse graph on the device -- the entries W_ij > threshold (i != j), columns ascending
 * inside a row -- without the dense matrix ever being stored: rows are formed and compacted in LDS, only the CSR
 * triple reaches the host (a 20,000-patch graph is 3.2 GB dense, a few MB sparse).  Call it with col = val = NULL to
 * get rowptr[n+1] (rowptr[n] = number of stored entries), then again with arrays of that size.  Same doubles as
 * epi_gravity_matrix followed by a host-side cut. */
int32_t epi_gravity_csr(int32_t n, const int32_t *pop, const double *lat, const double *lon, const double *flight_mul,
                        double a, double b, double g, double k, double cap_row, double threshold, int32_t *rowptr,
                        int32_t *col, double *val, int32_t device);

/* ---- JVM-compatible streams (scala.util.Random at env.scala:42,91; DDQNAgent.scala:14-32;
 *      MLP.scala:5-9) ------------------------------------------------------------------------- */
int32_t epi_java_gaussians(int64_t seed, int32_t n, double *out);
/* out[t * streams + c] = the t-th nextGaussian() of a java.util.Random seeded seed0 + c (t < draws, c < streams),
 * generated on the device, a thread per stream: the per-country noise streams of a heterogeneous calibration
 * (Calibrate.scala:36 seeds country idx with idx + 42; env.scala:91 draws one value per step).  Bit-identical to
 * epi_java_gaussians(seed0 + c, draws, .) stream by stream. */
int32_t epi_java_gaussian_table(int64_t seed0, int32_t streams, int32_t draws, double *out, int32_t device);
/* Action stream of a fresh DDQNAgent for `steps` act() calls: 0..7, or -1 where it takes argmax Q. */
int32_t epi_agent_actions(const epi_hyper *hp, int32_t steps, int32_t *out);
/* Untrained network weights: W1[h*in], W2[h*h], W3[out*h]. */
int32_t epi_mlp_init(int32_t n_in, int32_t hidden, int32_t n_out, double *W1, double *W2, double *W3);
/* DDQNAgent.actGreedy / MLP.predict (DDQNAgent.scala:22-23, MLP.scala:15-22) for B states with the
 * given weights (biases may be NULL = zero): q[B][n_out] clamped to +-clip_q and the first argmax. */
int32_t epi_mlp_greedy_batch(const epi_state *states, int32_t batch, int32_t hidden, const double *W1,
                             const double *b1, const double *W2, const double *b2, const double *W3,
                             const double *b3, double clip_q, double *q, int32_t *action);

/* HyperParams fields the learn step reads (config.scala:3-21), one record per agent. */
typedef struct epi_learn_hyper {
    double gamma, lr, clip_q, grad_clip, ema_alpha;
    int32_t hidden;              /* one width per handle */
    int32_t replay_capacity;
    int32_t target_update_every;
    int32_t max_need;            /* largest batch * gradAccumSteps epi_agents_learn will be asked for */
} epi_learn_hyper;

/* ---- Calibrate.gridSurface / rollout (Calibrate.scala:19-116) over WorldToy.stepOne
 *      (WorldToy.scala:24-32,83-145), Validation.arrivalTime/peakI (Validation.scala:15-22) ------ */
typedef struct epi_grid_desc {
    int32_t n;              /* countries (ToyConn.C generalised to an input list) */
    const int32_t *pop;     /* [n] */
    const double *conn;     /* dense [n*n] row-major connection matrix W, or NULL when CSR is given */
    const int32_t *csr_rowptr; /* [n+1]  CSR form of W (structural zeros = weight 0) */
    const int32_t *csr_col;    /* [nnz]  strictly ascending inside a row */
    const double *csr_val;     /* [nnz] */
    int32_t seed_idx;       /* seedIdx */
    int32_t heterogeneous;  /* Calibrate.scala:34-54 */
    epi_state base;         /* hospCap, r, d, v, t, tMax are read (WorldToy.scala:28-32) */
    const epi_profile *prof; /* [n] country profiles (read when heterogeneous != 0) */
    epi_hyper hp;
    int32_t steps;          /* cfg.stepsPerEpoch */
    int32_t n_beta;
    const double *betas;    /* [n_beta]   betaGrid (outer loop) */
    int32_t n_travel;
    const double *travel;   /* [n_travel] travelScales (inner loop) */
    const double *t_arrive; /* [n] Targets.tArrive values, read where has_arrive != 0 */
    const uint8_t *has_arrive;
    const double *t_peak;   /* [n] Targets.peakI */
    const uint8_t *has_peak;
    int32_t device;         /* HIP device ordinal */
    int32_t pad_;
    /* Learning inside a rollout.  Calibrate.rollout steps its world with WorldToy.stepOne, which lets every node's
     * agent learn once gStep - lastLearnStep >= learnEvery and its replay buffer holds 4 * batchSize transitions
     * (WorldToy.scala:147-167; one transition per step, :99).  A sweep whose episodes take an argmax-Q step after that
     * point consults trained weights: it then runs one DDQNAgent per (grid point, country) on the device
     * (DDQNAgent.scala:34-65, MLP.scala:29-99, ReplayBuffer.scala:10-22), step by step.  `learn` = the HyperParams
     * fields the learn step reads (one record: Calibrate.scala:56-57 gives every agent the sweep's hp; hidden must
     * equal hp.hidden; max_need is ignored); NULL = such a sweep is refused with EPI_ERR_UNSUPPORTED, as are sparse
     * worlds and calls of more than 16,384 (grid point, country) pairs.  learn_intensity[n] = gradAccumSteps of each
     * country's learn call (WorldToy.scala:157-162: 2 for "India", 1 otherwise); NULL = 1 everywhere.  Sweeps that
     * never reach that point (the reference's defaults: 300 steps < 4 * 128) ignore both fields. */
    const epi_learn_hyper *learn;
    const int32_t *learn_intensity;
} epi_grid_desc;

typedef struct epi_grid epi_grid;

/* Uploads the graph and builds every per-sweep table on `desc->device`.  Nothing in `desc` is
 * retained.  Returns NULL on failure (see epi_last_error). */
epi_grid *epi_grid_create(const epi_grid_desc *desc);
/* Runs grid points [g_begin, g_begin+g_count) (flat index g = i_beta*n_travel + i_travel, the
 * reference's loop order).  loss[g_count]; arrival[g_count][n] and peak[g_count][n] may be NULL.
 * With out_on_device != 0 the three pointers are device memory on the handle's device, the work is
 * only enqueued on `stream` (a hipStream_t, NULL = default stream) and the status of the run is
 * collected by epi_grid_status().  Otherwise they are host memory and the call is synchronous.
 * A sparse world (more than 1,024 countries) takes at most 2^32 / (20 * npad) grid points per call, npad = n rounded up
 * to 64 (10,720 at 20,000 patches): larger calls return EPI_ERR_UNSUPPORTED and are split by the caller. */
int32_t epi_grid_run(epi_grid *g, int32_t g_begin, int32_t g_count, double *loss, int32_t *arrival, double *peak,
                     int32_t out_on_device, void *stream);
/* The same for the grid points g_begin + k * g_stride, k < g_count (outputs indexed by k).  An interleaved split
 * over ranks (g_begin = rank, g_stride = ranks) balances the sweep better than contiguous slabs: episodes of one
 * beta row cost about the same, episodes of different rows do not. */
int32_t epi_grid_run_strided(epi_grid *g, int32_t g_begin, int32_t g_stride, int32_t g_count, double *loss,
                             int32_t *arrival, double *peak, int32_t out_on_device, void *stream);
/* No counterpart in the reference: for callers that cannot allocate device memory -- tight serving loops,
 * harnesses that forbid hipMalloc, code that records the sweep under stream capture.  The sweep of
 * epi_grid_run_strided with out_on_device != 0 (same kernels, same results), but the per-episode state lives in a
 * fixed 32 MiB device array that the library holds on every GPU it runs on until the process ends, and the call
 * allocates and frees nothing (no hipMalloc / hipFree / events), sets no kernel attribute, reads no environment
 * variable and never waits on the device: it only enqueues kernels on `stream`.  loss_dev[g_count] is device
 * memory on the handle's device; arrival_dev[g_count][n] and peak_dev[g_count][n] are optional device memory.
 * epi_grid_status() collects the run's status from the handle's own error table (allocated by epi_grid_create).
 * A handle holds one outstanding run: while one is uncollected this entry point returns EPI_ERR_BAD_ARGUMENT and
 * enqueues nothing -- call epi_grid_status() first (outside any capture: it waits for the device).
 * Limit: g_count episodes of the world's padded width must fit the array (41 bytes per padded country and episode
 * plus 16 per episode, 57 for sparse graphs: up to 798 episodes of 1,000 countries, 3,192 of 200, 29 of a
 * 20,000-patch world) -- larger requests return EPI_ERR_UNSUPPORTED and run nothing.  All calls on one GPU share
 * the array: calls that overlap in time on one GPU race, whatever the handle (calls on one stream are ordered and
 * safe, and each handle's status stays its own); it is overwritten from the start by every call. */
int32_t epi_grid_run_noalloc(epi_grid *g, int32_t g_begin, int32_t g_stride, int32_t g_count, double *loss_dev,
                             int32_t *arrival_dev, double *peak_dev, void *stream);
/* Waits for the last enqueued run and returns its status (first failing grid point wins).  A handle holds one
 * outstanding run: enqueueing another one first waits for the earlier one and returns its failure, if any. */
int32_t epi_grid_status(epi_grid *g);
int32_t epi_grid_destroy(epi_grid *g);
/* Measurement hooks (no counterpart in the reference).  With timing on, every launch of the episode
 * kernel is bracketed by HIP events on its stream; epi_grid_kernel_time() returns the summed kernel
 * time in ms, the number of launches and the country-steps those launches processed since timing
 * was switched on. */
int32_t epi_grid_set_timing(epi_grid *g, int32_t on);
int32_t epi_grid_kernel_time(epi_grid *g, double *total_ms, int64_t *launches, int64_t *country_steps);
/* Diagnostics (no counterpart in the reference).  The argmax-Q step of a large batch first screens the states with
 * fused arithmetic and re-evaluates, in the reference's operation order, those whose first maximum the screen's error
 * bound cannot certify: screened = states the screen looked at, flagged = states it sent to the exact evaluation,
 * both since the handle's network was uploaded (0, 0 when the screen is off or was never used). */
int32_t epi_grid_screen_stats(epi_grid *g, int64_t *screened, int64_t *flagged);
/* First minimum in flat order with strict '<' starting from Double.MaxValue (Calibrate.scala:96,112). */
int32_t epi_grid_argmin(const double *loss, int32_t count, int32_t *best_idx, double *best_loss);
/* gridSurface in one call on one device: loss[n_beta*n_travel] beta-major; arrival/peak optional. */
int32_t epi_calibrate_grid(const epi_grid_desc *desc, double *loss, int32_t *arrival, double *peak,
                           int32_t *best_idx, double *best_loss);

/* ---- Calibrate.gridSurface (Calibrate.scala:84-116) over several GPUs of one node, from one process ----------
 * The grid loop (Calibrate.scala:100-113) runs independent rollouts: the flat beta-major grid is cut into n_dev
 * interleaved shares (share k = grid points k, k + n_dev, ...), share k runs on devices[k] with an engine of its
 * own, and the fp64 loss pieces are all-gathered with RCCL (ncclAllGather over the distinct devices, xGMI) so that
 * every device ends up with the whole surface; the first-minimum scan of Calibrate.scala:96,112 then runs on the
 * host.  desc->device is ignored.  A device may be named more than once (several engines on one GPU).  librccl is
 * opened on first use; without it the create call fails with EPI_ERR_DEVICE. */
typedef struct epi_grid_multi epi_grid_multi;
epi_grid_multi *epi_grid_multi_create(const epi_grid_desc *desc, const int32_t *devices, int32_t n_dev);
/* One whole sweep: loss[n_beta*n_travel] beta-major (may be NULL), optional arrival/peak [G][n], the first
 * minimum.  Synchronous.  May be called repeatedly (tables, engines and the communicator are reused). */
int32_t epi_grid_multi_run(epi_grid_multi *m, double *loss, int32_t *arrival, double *peak, int32_t *best_idx,
                           double *best_loss);
/* A run that fails -- a share's enqueue, the all-gather, or a device-side check of any share (mass, Int range:
 * the first failing share's status and message are returned) -- waits for everything already enqueued and collects
 * every share's status before it returns, so the handle can be run again. */
/* Communicator size (distinct devices) and the number of share slots each rank contributes to the all-gather. */
int32_t epi_grid_multi_ranks(const epi_grid_multi *m, int32_t *n_ranks, int32_t *slots_per_rank);
/* Measurement hooks (no counterpart in the reference).  With timing on, every share's episode launches are timed as
 * by epi_grid_set_timing and the all-gather is bracketed by HIP events on every device's exchange stream.
 * epi_grid_multi_times: share_kernel_ms[n_dev] = each share's summed episode-kernel time, gather_ms = the summed
 * all-gather time (per run the slowest device's, which includes waiting for the slowest share), wall_ms = the
 * summed host time of the epi_grid_multi_run calls, runs = their number, all since timing was switched on. */
int32_t epi_grid_multi_set_timing(epi_grid_multi *m, int32_t on);
int32_t epi_grid_multi_times(epi_grid_multi *m, double *share_kernel_ms, double *gather_ms, double *wall_ms,
                             int64_t *runs);
/* Diagnostics: can_access[n_ranks][n_ranks], entry (a, b) = 1 when the device of rank a can map the memory of the
 * device of rank b (hipDeviceCanAccessPeer; 1 on the diagonal).  RCCL chooses its transport itself. */
int32_t epi_grid_multi_peer_access(const epi_grid_multi *m, uint8_t *can_access);
/* Test hook: the next epi_grid_multi_run fails with EPI_ERR_DEVICE when it reaches share `share`'s enqueue, after
 * the shares before it have been enqueued (-1 disarms).  One shot. */
int32_t epi_grid_multi_inject_failure(epi_grid_multi *m, int32_t share);
int32_t epi_grid_multi_destroy(epi_grid_multi *m);
/* create + run + destroy. */
int32_t epi_calibrate_grid_multi(const epi_grid_desc *desc, const int32_t *devices, int32_t n_dev, double *loss,
                                 int32_t *arrival, double *peak, int32_t *best_idx, double *best_loss);

/* ---- WorldToy (WorldToy.scala:11-235): batched worlds stepped on the device ------------------- */
typedef struct epi_world_desc {
    int32_t n;
    int32_t batch;          /* independent world instances sharing graph and env parameters */
    const int32_t *pop;     /* [n] */
    const double *conn;     /* dense [n*n] or NULL */
    const int32_t *csr_rowptr;
    const int32_t *csr_col;
    const double *csr_val;
    int32_t seed_idx;
    int32_t validate;       /* WorldToy validate flag: mass checks on */
    epi_state base;
    const epi_env_params *env; /* [n] one EpidemicEnv per country (makeEnvFor) */
    const epi_profile *prof;   /* [n] */
    const int32_t *env_seeds;  /* [n] Random seeds of the envs (env.scala:42); instance b uses seed+b*0 (same streams) */
    epi_hyper hp;
    int32_t device;
    int32_t pad_;
} epi_world_desc;

typedef struct epi_world epi_world;
typedef struct epi_agents epi_agents;   /* the DDQN learners declared further down */

epi_world *epi_world_create(const epi_world_desc *desc);
/* One WorldToy.stepOne (:83-145) for every instance with caller-chosen actions[batch][n].
 * gauss[batch][n] supplies the env noise draws; NULL = the engine's own JVM-compatible streams
 * (one per country, shared by all instances, exactly what `batch` fresh copies of the reference
 * world would draw).  out_states[batch][n] / out_rewards[batch][n] may be NULL. */
int32_t epi_world_step(epi_world *w, const int32_t *actions, const double *gauss, int32_t t_max,
                       epi_state *out_states, double *out_rewards);
/* Optional (no counterpart in the reference): addresses of the handle's own page-locked result arrays
 * states[batch][n], rewards[batch][n], actions[batch][n] (any of the three pointers may be NULL).  The device
 * writes every step's results into them; a caller that passes exactly these addresses as the out_* arguments
 * of epi_world_step / epi_world_step_greedy reads the results in place, without the copy into caller memory.
 * They belong to the handle and are overwritten by the next step. */
int32_t epi_world_result_buffers(epi_world *w, epi_state **states, double **rewards, int32_t **actions);
/* Optional (no counterpart in the reference): epi_world_step with the states returned as 32-byte records instead
 * of 64-byte ones, for callers that read every instance's results every step over PCIe (a batched rollout of small
 * worlds is bound by exactly that transfer).  Nothing is lost: the compartments of every state the step produces
 * are whole numbers below 2^31 (env.scala:81-85, 244-262), and t_max and seed are those of the base state.
 * epi_world_packed_buffer returns the handle's page-locked array of such records; passed as out_states it is read in
 * place, like the arrays of epi_world_result_buffers (whose rewards array serves out_rewards here too). */
typedef struct epi_state_packed {
    int32_t s, i, r, d, v, t;
    double hosp_cap;
} epi_state_packed;
int32_t epi_world_step_packed(epi_world *w, const int32_t *actions, const double *gauss, int32_t t_max,
                              epi_state_packed *out_states, double *out_rewards);
int32_t epi_world_packed_buffer(epi_world *w, epi_state_packed **states);
/* WorldToy.stepOneGreedy (:172-235): actions = argmax Q(features) with the weights set below. */
int32_t epi_world_step_greedy(epi_world *w, const double *gauss, int32_t t_max, epi_state *out_states,
                              double *out_rewards, int32_t *out_actions);
/* The trainer's greedy evaluation loop (MutliTrainer.scala:137-141): `steps` consecutive WorldToy.stepOneGreedy(t_max)
 * calls (WorldToy.scala:172-235) for every instance, without a host round trip per step.  With agents != NULL country
 * c asks the online network of agent c (DDQNAgent.actGreedy, DDQNAgent.scala:22-23; one agent per node,
 * WorldToy.scala:50; same device as the world); with NULL every country asks the network of epi_world_set_mlp.
 * Env noise comes from the engine's JVM-compatible per-country streams, advanced exactly as `steps` single calls
 * would.  Outputs, each optional: i_series[steps][batch][n] = s2.i after every step (what the trainer appends to
 * its series, :139); arrival[batch][n] = Validation.arrivalTime(series, 10.0) and peak[batch][n] =
 * Validation.peakI(series) over the steps of this call (Validation.scala:15-22); out_states[batch][n] = the final
 * states.  snapshotStates / resetToInitial / restoreStates around it are epi_world_get_states / _reset / _set_states. */
int32_t epi_world_rollout_greedy(epi_world *w, epi_agents *agents, int32_t steps, int32_t t_max, double *i_series,
                                 int32_t *arrival, double *peak, epi_state *out_states);
/* Makes the world keep, on the device, what DDQNAgent.observe needs of every following step (the states before it
 * and S, I right after env.step) for epi_agents_observe_world.  Off by default: it costs three plane copies a step. */
int32_t epi_world_keep_transitions(epi_world *w, int32_t on);
int32_t epi_world_set_mlp(epi_world *w, int32_t hidden, const double *W1, const double *b1, const double *W2,
                          const double *b2, const double *W3, const double *b3, double clip_q);
int32_t epi_world_get_states(epi_world *w, epi_state *out);       /* snapshotStates :75 */
/* restoreStates :76.  Deliberate restrictions: compartments must be non-negative integers below 2^31 (what every
 * state the engine or the reference's own step produces holds; the reference would round other values at its next
 * env.step, env.scala:81-85), and the states' t_max / seed fields are not kept -- every instance uses the base
 * state's t_max, as every state a WorldToy produces does. */
int32_t epi_world_set_states(epi_world *w, const epi_state *in);
int32_t epi_world_reset(epi_world *w);                            /* resetToInitial :77 */
/* Net S and I flows of the last step, movedS/movedI[batch][n] (WorldToy.scala:106-126). */
int32_t epi_world_last_flows(epi_world *w, int32_t *moved_s, int32_t *moved_i);
/* As epi_grid_screen_stats, for the network of epi_world_set_mlp (batched greedy steps). */
int32_t epi_world_screen_stats(epi_world *w, int64_t *screened, int64_t *flagged);
/* Measurement hooks (no counterpart in the reference), as epi_grid_set_timing / epi_grid_kernel_time. */
int32_t epi_world_set_timing(epi_world *w, int32_t on);
int32_t epi_world_kernel_time(epi_world *w, double *total_ms, int64_t *launches, int64_t *country_steps);
int32_t epi_world_destroy(epi_world *w);

/* ---- DDQNAgent learning (DDQNAgent.scala:6-70) with MLP.trainBatch (MLP.scala:29-99) and
 *      ReplayBuffer (ReplayBuffer.scala:10-22): one agent per country (WorldToy.scala:46-65) ------ */
/* epi_learn_hyper (the HyperParams fields the learn step reads) is declared with the calibration sweep above. */

/* n_agents fresh agents (two networks each from MLP.scala:5-9, empty replay buffers whose shuffle stream is
 * Random(123)) on `device`.  Returns NULL on failure. */
epi_agents *epi_agents_create(int32_t n_agents, const epi_learn_hyper *hp, int32_t device);
int32_t epi_agents_destroy(epi_agents *g);
/* DDQNAgent.observe (:34): one Transition(s, a, r, s2, done) for each of `count` agents; ids = NULL means
 * agents 0..count-1.  The buffers keep State.features of s and s2, the only parts learn() reads. */
int32_t epi_agents_observe(epi_agents *g, const int32_t *ids, int32_t count, const epi_state *s, const int32_t *a,
                           const double *r, const epi_state *s2, const uint8_t *done);
/* The same for the last step of a world, without the states crossing the bus: agent c observes, for every instance
 * of the world, Transition(s, a, r, s2, done) with s = the state before the step, a = the action taken, r = the
 * reward, s2 = the state env.step returned (WorldToy.scala:90,99: before the mobility flows, with env.step's own
 * clock) and done as env.scala:137 -- computed from the planes the world keeps on the device once
 * epi_world_keep_transitions(w, 1) was called.  Instances are filed in order 0 .. batch-1.  Same device. */
int32_t epi_agents_observe_world(epi_agents *g, epi_world *w);
/* DDQNAgent.replaySize (:69) of every agent: sizes[n_agents]. */
int32_t epi_agents_replay_size(epi_agents *g, int32_t *sizes);
/* ReplayBuffer.sample (ReplayBuffer.scala:21-22): buffer positions of rnd.shuffle(buf).take(n); advances the
 * agent's shuffle stream exactly as one sample() call does. */
int32_t epi_agents_sample(epi_agents *g, int32_t agent, int32_t n, int32_t *out, int32_t *taken);
/* DDQNAgent.learn(batch, gradAccumSteps) (:51-65) for `count` agents in one set of launches.  steps[k] is the
 * agent's act() counter (:12,27), which decides the target-network copy (:59-61).  loss[k] is learn's return
 * value (0.0, and nothing drawn, while the buffer holds fewer than batch*gradAccumSteps transitions);
 * ema_loss[k] (may be NULL) is emaLossNow (:68) afterwards. */
int32_t epi_agents_learn(epi_agents *g, const int32_t *ids, int32_t count, const int32_t *batch,
                         const int32_t *grad_accum, const int64_t *steps, double *loss, double *ema_loss);
/* DDQNAgent.qValues / actGreedy (:22-23,70): q[count][8] (may be NULL) and the first argmax greedy[count]
 * (may be NULL) of one state per agent; use_target != 0 evaluates the target network instead. */
int32_t epi_agents_q_values(epi_agents *g, const int32_t *ids, int32_t count, const epi_state *s, int32_t use_target,
                            double *q, int32_t *greedy);
/* Weights in the reference's layout: W1[h][7], b1[h], W2[h][h], b2[h], W3[8][h], b3[8] (any get pointer may
 * be NULL; NULL biases in set mean zero).  For checkpoints and for handing a trained network to
 * epi_world_set_mlp. */
int32_t epi_agents_get_weights(epi_agents *g, int32_t agent, int32_t target, double *W1, double *b1, double *W2,
                               double *b2, double *W3, double *b3);
int32_t epi_agents_set_weights(epi_agents *g, int32_t agent, int32_t target, const double *W1, const double *b1,
                               const double *W2, const double *b2, const double *W3, const double *b3);
/* Measurement hooks (no counterpart in the reference): device time of the learn launches, their number and
 * the samples they trained on since timing was switched on. */
int32_t epi_agents_set_timing(epi_agents *g, int32_t on);
int32_t epi_agents_kernel_time(epi_agents *g, double *total_ms, int64_t *learn_calls, int64_t *samples);

#ifdef __cplusplus
}
#endif
#endif /* EPI_H_ */
