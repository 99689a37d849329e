// Host-side JVM-compatible number streams used to drive the device kernels.
//
// The reference draws its environment noise and its epsilon-greedy actions from
// scala.util.Random (= java.util.Random): env.scala:42,91, DDQNAgent.scala:14,28,
// MLP.scala:5-6.  To reproduce its integer trajectories the engine has to feed
// the kernels the same numbers, so this file implements the JDK's documented
// generator (48-bit LCG, nextDouble, nextInt(bound), polar nextGaussian on
// StrictMath.log/sqrt) and the agent's action schedule.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif
#include <cstdint>
#include <vector>

namespace epi {

class JavaRandom {
public:
    explicit JavaRandom(int64_t seed = 0) { reseed(seed); }
    void reseed(int64_t seed);
    int32_t next(int bits);
    int32_t nextInt() { return next(32); }
    int32_t nextInt(int32_t bound);
    double nextDouble();
    double nextGaussian();

private:
    uint64_t state_ = 0;
    bool have_cached_ = false;
    double cached_ = 0.0;
};

// StrictMath.log (fdlibm e_log algorithm); differs from glibc log in the last ulp
// for some arguments, which would change nextGaussian and hence trajectories.
#if defined(__HIPCC__)
__host__ __device__
#endif
double strict_log(double x);

#if defined(__HIPCC__)
// out[t * ld + slot_of[c]] (slot_of == NULL: column c) = the t-th nextGaussian() of a java.util.Random seeded
// seed0 + c, for c < streams and t < draws, generated on the device: one thread per stream, the same operations in
// the same order as JavaRandom::nextGaussian (every one of them rounds once on either side).  The per-country
// environment noise of a heterogeneous sweep (Calibrate.scala:36: seed = idx + 42) is 4 million draws at 20,000
// patches and 200 steps -- 60 ms of host time, most of the engine's creation.
hipError_t launch_gaussian_table(double* out, const int32_t* slot_of, int streams, int draws, size_t ld, int64_t seed0,
                                 hipStream_t stream);
#endif

// Fields of HyperParams the path reads (config.scala:3-21).
struct Hyper {
    double epsilon_start = 1.0, epsilon_end = 0.1;
    int32_t epsilon_decay_steps = 100000, hidden = 128;
    double clip_q = 10.0, reward_clip = 5.0;
    int32_t batch_size = 128, learn_every = 8;
};

constexpr int kGreedyAction = -1;

// Action choice of one DDQNAgent without the network (DDQNAgent.scala:12-32):
// advances `steps`, draws nextDouble and, on the exploring branch, nextInt(8).
// Returns 0..7, or kGreedyAction when the agent would take argmax Q(s).
struct AgentStream {
    int64_t steps = 0;
    JavaRandom rnd{9};
    int act(const Hyper& hp);
};

// MLP.scala:5-9: W1[h][in], W2[h][h], W3[out][h] row-major, one Random(7) stream,
// each entry nextGaussian()*0.05; biases zero.
void mlp_init(int in, int h, int out, std::vector<double>& W1, std::vector<double>& W2, std::vector<double>& W3);

}  // namespace epi
