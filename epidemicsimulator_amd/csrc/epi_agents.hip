// DDQN learners on the device: DDQNAgent.observe / ddqnTarget / learn (DDQNAgent.scala:34-65),
// MLP.predict / trainBatch (MLP.scala:15-99) and ReplayBuffer (ReplayBuffer.scala:10-22) for a set of
// independent agents (one per country in WorldToy.scala:46-65).
//
// Arithmetic contract: every sum the reference forms left to right is formed left to right here, one
// rounding per operation (the file is compiled with -ffp-contract=off), so weights, Q-values and losses
// are the same doubles.  That rules the matrix cores out (their accumulation order and fusing differ);
// the contractions run on the vector ALUs, one output element per thread, sequential along the reduction
// axis, weights read coalesced from L2.
//
// Division of labour: the replay rings live in device memory (State.features of s and s2, reward, action, done per
// transition); the shuffle of ReplayBuffer.sample (a sequential java.util.Random walk over the whole buffer) runs on
// the host and only the sampled positions cross the bus; everything from the three forward passes to the SGD step
// runs on the device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "../../include/epi.h"
#include "epi_host.hpp"
#include "epi_kernels.hpp"
#include "java_compat.hpp"

#define EPI_API extern "C" __attribute__((visibility("default")))

namespace epi {
namespace {

constexpr int kIn = 7;        // State.features (env.scala:28-30)
constexpr int kOut = 8;       // Action.values.length
constexpr int kXStride = 8;   // features padded to 8 doubles per row
constexpr int kGradRows = 16; // rows of gW2 one thread accumulates (one read of h1 serves them all)
constexpr int kMaxHidden = 256;

struct AgentHyperDev {
    double gamma, lr, clip_q, grad_clip;
};

// One stored transition (ReplayBuffer.scala:8) as the device keeps it: State.features of s and s2 (the only parts
// of the states learn() reads), reward, action, done.
struct RingRec {
    double fs[kXStride], fs2[kXStride];
    double r;
    int32_t a, done;
};

// Arguments shared by the kernels of one call.  A "slot" is one agent taking part in the call.
struct AgentsArgs {
    NetLayout nl;
    int Bp;                     // row capacity per slot and class (multiple of 64)
    int n_classes;              // learn: 3 (s/online, s2/online, s2/target); q-values: 1
    int net_of_class[3];        // 0 = online, 1 = target
    int x_of_class[3];          // which staged feature block the class reads (0 = s, 1 = s2)
    double* nets;               // [n_agents][2] networks
    const AgentHyperDev* hyper; // [n_agents]
    const int32_t* slot_agent;  // [slots]
    const int32_t* slot_rows;   // [slots] rows in use (need)
    const RingRec* ring;        // replay rings of all agents, agent k's records from ring_off[k] on
    const long long* ring_off;  // [n_agents]
    const int32_t* idx;         // [slots][Bp] sampled ring positions (host memory, read over the bus), or NULL:
    const double* X;            // [slots][kXStride] then: one directly staged feature row per slot (q-values)
    double* H1;                 // [slots][Bp][Hp]      class 0
    double* H2;                 // [slots][Bp][Hp]      class 0
    double* Q;                  // [slots][3][Bp][kOut] clamped outputs
    double* DZ2;                // [slots][Bp][Hp]
    double* DZ1;                // [slots][Bp][Hp]
    double* ERR;                // [slots][Bp]  yC(a) - target(a)
    double* X0d;                // [slots][Bp][kXStride] device copy of the s features (the staged rows sit in host
    int32_t* ACTd;              // [slots][Bp]           memory: each is read over the bus once, by the first kernel
                                //                       that needs it, and kept here for the gradient kernels)
    double* loss;               // [slots]
};

__device__ inline double* net_ptr(const AgentsArgs& a, int agent, int which) {
    return a.nets + ((size_t)agent * 2 + which) * a.nl.size();
}
__device__ inline double clip_sym(double v, double c) { return isfinite(c) ? fmax(-c, fmin(c, v)) : v; }

// Tiling of the forward and backward kernels: one lane per sample row (64 rows per workgroup), one wave per
// tile of kTile = 16 hidden units.  A lane multiplies ITS activation by weights that are the same for the whole
// wave, so the weights arrive through scalar loads (kTile contiguous doubles per step of the sum) and the vector
// ALU does nothing but the ordered multiply-adds.  Activations pass through LDS as [unit][row] (row stride kLS,
// odd, so the row-major copies to and from global memory are conflict-free too), kChunk units at a time: the
// sums over a layer's inputs are taken chunk after chunk, in order, with the accumulators kept in registers.
// A workgroup then needs 37 KB of LDS instead of 70 KB for a whole layer of 128 units, four of them fit a CU, and
// the kernels, which wait on the scalar loads with 4 waves per SIMD (0.60 ms forward), run with 7.
constexpr int kTileRows = 64;
constexpr int kLS = kTileRows + 1;
constexpr int kChunk = 64;

__host__ __device__ inline size_t tile_lds_bytes() { return ((size_t)kChunk * kLS + (size_t)kXStride * kTileRows) * sizeof(double); }

// sh[unit - u0][row] -> dst[row][unit] (row-major, stride Hp) for units u0 .. u0 + count - 1 and the rows that exist
__device__ inline void store_chunk(const double* sh, double* dst, int u0, int count, int Hp, int rows) {
    for (int idx = threadIdx.x; idx < kTileRows * kChunk; idx += blockDim.x) {
        const int row = idx / kChunk, u = idx - row * kChunk;
        if (row < rows && u < count) dst[(size_t)row * Hp + u0 + u] = sh[u * kLS + row];
    }
}

// acc[u] += sum over j < K, in order, of W[j * ldw + u] * act[j * kLS]; with `first` the first product starts the
// sums (as .sum does)
template <int kTile>
__device__ inline void ordered_tile(const double* __restrict__ W, int ldw, const double* act, int K, bool first,
                                    double (&acc)[kTile]) {
    int j = 0;
    if (first) {
        const double h = act[0];
#pragma unroll
        for (int u = 0; u < kTile; ++u) acc[u] = W[u] * h;
        j = 1;
    }
#pragma unroll 2
    for (; j < K; ++j) {
        const double h = act[j * kLS];
#pragma unroll
        for (int u = 0; u < kTile; ++u) acc[u] = acc[u] + W[(size_t)j * ldw + u] * h;
    }
}

// First hidden layer for the lane's row and units c0 + u0 .. c0 + u0 + N - 1 of a chunk: relu(W1 x + b1) into LDS.
template <int N>
__device__ inline void first_layer_units(const double* __restrict__ net, const NetLayout& nl, int c0, int u0,
                                         const double* sh_x, double* sh_h, int lane) {
    const double* __restrict__ W1t = net + nl.w1t() + c0 + u0;
    const double* __restrict__ b1 = net + nl.b1() + c0 + u0;
    double z[N];
    double x = sh_x[lane];
#pragma unroll
    for (int u = 0; u < N; ++u) z[u] = W1t[u] * x;
#pragma unroll
    for (int j = 1; j < kIn; ++j) {
        x = sh_x[j * kTileRows + lane];
#pragma unroll
        for (int u = 0; u < N; ++u) z[u] = z[u] + W1t[(size_t)j * nl.Hp + u] * x;
    }
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const double v = z[u] + b1[u];
        sh_h[(u0 + u) * kLS + lane] = v > 0 ? v : 0.0;
    }
}

// MLP.predict (MLP.scala:15-22) for 64 rows of one class of one slot.  matVec sums left to right.
// `nets` is the same memory as a.nets, passed as a parameter of its own that nothing is stored through: that lets
// the compiler prove the weights unclobbered and fetch them with scalar loads.
template <int kTile>
__global__ void __launch_bounds__(1024) agents_forward_kernel(AgentsArgs a, const double* __restrict__ nets) {
    extern __shared__ double smem[];
    double* sh_x = smem;                          // [kXStride][64]
    double* sh_h = smem + kXStride * kTileRows;   // [kChunk][kLS]
    const int slot = blockIdx.y, B = a.slot_rows[slot];
    const int groups = a.Bp / kTileRows;
    const int cls = blockIdx.x / groups, k0 = (blockIdx.x % groups) * kTileRows;
    if (k0 >= B) return;
    const int H = a.nl.H, Hp = a.nl.Hp;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i0 = wave * kTile;
    const int n_waves = blockDim.x >> 6;
    const int agent = a.slot_agent[slot];
    const double* __restrict__ net = nets + ((size_t)agent * 2 + a.net_of_class[cls]) * a.nl.size();
    const double clip_q = a.hyper[agent].clip_q;
    const int rows = min(kTileRows, B - k0);
    const size_t row0 = (size_t)slot * a.Bp + k0;
    const RingRec* ring = a.ring + a.ring_off[agent];
    for (int idx = threadIdx.x; idx < kTileRows * kXStride; idx += blockDim.x) {
        const int row = idx / kXStride, j = idx % kXStride;
        double x = 0.0;
        if (row < rows) {
            if (a.idx) {
                const RingRec& rec = ring[a.idx[row0 + row]];
                x = a.x_of_class[cls] ? rec.fs2[j] : rec.fs[j];
            } else {
                x = a.X[(size_t)slot * kXStride + j];
            }
        }
        sh_x[j * kTileRows + row] = x;
        if (cls == 0) a.X0d[row0 * kXStride + idx] = x;   // gathered rows of s, row-major, for the gradient kernels
    }
    double acc[kTile];
    // first layer, kChunk units at a time, each chunk consumed by the second layer's sums right away
    for (int c0 = 0; c0 < H; c0 += kChunk) {
        const int cn = min(kChunk, H - c0);
        __syncthreads();  // sh_x written / the previous chunk read by every wave
        // every wave takes an equal run of the chunk's units (weights padded with zeros past H)
        switch (n_waves) {
            case 4: first_layer_units<16>(net, a.nl, c0, wave * 16, sh_x, sh_h, lane); break;
            case 8: first_layer_units<8>(net, a.nl, c0, wave * 8, sh_x, sh_h, lane); break;
            case 16: first_layer_units<4>(net, a.nl, c0, wave * 4, sh_x, sh_h, lane); break;
            default:
                for (int uu = wave; uu < kChunk; uu += n_waves) first_layer_units<1>(net, a.nl, c0, uu, sh_x, sh_h, lane);
        }
        __syncthreads();
        if (cls == 0) store_chunk(sh_h, a.H1 + row0 * Hp, c0, cn, Hp, rows);
        if (i0 < H) ordered_tile(net + a.nl.w2t() + (size_t)c0 * Hp + i0, Hp, sh_h + lane, cn, c0 == 0, acc);
    }
    if (i0 < H) {
        const double* __restrict__ b2 = net + a.nl.b2() + i0;
#pragma unroll
        for (int u = 0; u < kTile; ++u) {
            const double z = acc[u] + b2[u];
            acc[u] = z > 0 ? z : 0.0;
        }
    }
    // output layer: the second hidden layer passes through LDS chunk by chunk; wave w sums outputs w and
    // w + n_waves (a workgroup has at least four waves)
    double y[2] = {0.0, 0.0};
    for (int c0 = 0; c0 < H; c0 += kChunk) {
        const int cn = min(kChunk, H - c0);
        __syncthreads();  // the previous contents read by every wave
        if (i0 >= c0 && i0 < c0 + kChunk && i0 < H) {
#pragma unroll
            for (int u = 0; u < kTile; ++u) sh_h[(i0 - c0 + u) * kLS + lane] = acc[u];
        }
        __syncthreads();
        if (cls == 0) store_chunk(sh_h, a.H2 + row0 * Hp, c0, cn, Hp, rows);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int o = wave + q * n_waves;
            if (o < kOut) {
                const double* __restrict__ W3 = net + a.nl.w3() + (size_t)o * Hp + c0;
                int j = 0;
                if (c0 == 0) { y[q] = W3[0] * sh_h[lane]; j = 1; }
#pragma unroll 16
                for (; j < cn; ++j) y[q] = y[q] + W3[j] * sh_h[j * kLS + lane];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int o = wave + q * n_waves;
        if (o < kOut) {
            const double yo = y[q] + net[a.nl.b3() + o];
            if (lane < rows) a.Q[(((size_t)slot * 3 + cls) * a.Bp + k0 + lane) * kOut + o] = clip_sym(yo, clip_q);
        }
    }
}

// ddqnTarget (DDQNAgent.scala:37-47) and the backward pass of MLP.trainBatch (MLP.scala:44-58) for 64
// samples.  The target equals the prediction in every component but the action taken, so err is zero there
// and dh2(j) = W3(a)(j) * err(a).  dz = dh * relu(z) -- the activation's value, as the reference has it.
template <int kTile>
__global__ void __launch_bounds__(1024) agents_backward_kernel(AgentsArgs a, const double* __restrict__ nets) {
    extern __shared__ double smem[];
    double* sh_err = smem;                        // [64]
    int* sh_act = reinterpret_cast<int*>(smem + kTileRows);  // [64]
    double* sh_d = smem + kXStride * kTileRows;   // [kChunk][kLS]
    const int slot = blockIdx.y, B = a.slot_rows[slot];
    const int k0 = blockIdx.x * kTileRows;
    if (k0 >= B) return;
    const int H = a.nl.H, Hp = a.nl.Hp;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i0 = wave * kTile;
    const int agent = a.slot_agent[slot];
    const double* __restrict__ net = nets + (size_t)agent * 2 * a.nl.size();
    const AgentHyperDev hp = a.hyper[agent];
    const int rows = min(kTileRows, B - k0);
    const size_t row0 = (size_t)slot * a.Bp + k0;
    if (threadIdx.x < kTileRows) {
        const int k = k0 + lane;
        double err = 0.0;
        int act = 0;
        if (lane < rows) {
            const size_t row = row0 + lane;
            const double* q = a.Q + (((size_t)slot * 3 + 0) * a.Bp + k) * kOut;
            const RingRec& rec = a.ring[a.ring_off[agent] + a.idx[row]];
            act = rec.a;
            a.ACTd[row] = act;
            double tgt;
            if (rec.done) {
                tgt = fmax(-hp.clip_q, fmin(hp.clip_q, rec.r));
            } else {
                const double* qn = a.Q + (((size_t)slot * 3 + 1) * a.Bp + k) * kOut;
                const double* qt = a.Q + (((size_t)slot * 3 + 2) * a.Bp + k) * kOut;
                int best = 0;  // zipWithIndex.maxBy(_._1): the first maximum
                for (int o = 1; o < kOut; ++o)
                    if (qn[o] > qn[best]) best = o;
                tgt = fmax(-hp.clip_q, fmin(hp.clip_q, rec.r + hp.gamma * qt[best]));
            }
            err = q[act] - tgt;
            a.ERR[row] = err;
        }
        sh_err[lane] = err;
        sh_act[lane] = act;
    }
    double acc[kTile];
    // dz2 of kChunk units at a time: formed in LDS from the stored activations, written out, and consumed by the
    // sums dh1(j) = sum_i W2(i)(j) * dz2(i) right away
    for (int c0 = 0; c0 < H; c0 += kChunk) {
        const int cn = min(kChunk, H - c0);
        __syncthreads();  // sh_err, sh_act written / the previous chunk read by every wave
        for (int idx = threadIdx.x; idx < kTileRows * kChunk; idx += blockDim.x) {
            const int row = idx / kChunk, u = idx - row * kChunk;
            double dz = 0.0;
            if (row < rows && u < cn) {
                const double dh = net[a.nl.w3() + (size_t)sh_act[row] * Hp + c0 + u] * sh_err[row];
                dz = dh * a.H2[(row0 + row) * Hp + c0 + u];
                a.DZ2[(row0 + row) * Hp + c0 + u] = dz;
            }
            sh_d[u * kLS + row] = dz;
        }
        __syncthreads();
        if (i0 < H) ordered_tile(net + a.nl.w2() + (size_t)c0 * Hp + i0, Hp, sh_d + lane, cn, c0 == 0, acc);
    }
    // dz1 = dh1 * relu(z1), written row-major through LDS, one chunk of units at a time
    for (int c0 = 0; c0 < H; c0 += kChunk) {
        const int cn = min(kChunk, H - c0);
        __syncthreads();
        if (i0 >= c0 && i0 < c0 + kChunk && i0 < H) {
#pragma unroll
            for (int u = 0; u < kTile; ++u) sh_d[(i0 - c0 + u) * kLS + lane] = acc[u];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < kTileRows * kChunk; idx += blockDim.x) {
            const int row = idx / kChunk, u = idx - row * kChunk;
            if (row < rows && u < cn) {
                const size_t g = (row0 + row) * Hp + c0 + u;
                a.DZ1[g] = sh_d[u * kLS + row] * a.H1[g];
            }
        }
    }
}

// gW2(i)(j) = sum over samples, in sample order, of dz2(i) * h1(j) (MLP.scala:53,66), then mean, clip and
// the SGD step (:78-93).  Block = Hp threads (j), kGradRows output rows i per block: per sample one coalesced
// read of h1 and one block-uniform read of kGradRows entries of dz2.  Both layouts of W2 are rewritten.
__global__ void __launch_bounds__(kMaxHidden) agents_grad_w2_kernel(AgentsArgs a) {
    const int slot = blockIdx.y, B = a.slot_rows[slot];
    const int H = a.nl.H, Hp = a.nl.Hp, j = threadIdx.x, i0 = blockIdx.x * kGradRows;
    if (j >= H || i0 >= H) return;
    const int agent = a.slot_agent[slot];
    double* net = net_ptr(a, agent, 0);
    const AgentHyperDev hp = a.hyper[agent];
    const double* __restrict__ H1 = a.H1 + (size_t)slot * a.Bp * Hp + j;
    const double* __restrict__ DZ2 = a.DZ2 + (size_t)slot * a.Bp * Hp + i0;  // columns past H hold zeros
    double acc[kGradRows];
#pragma unroll
    for (int r = 0; r < kGradRows; ++r) acc[r] = 0.0;
#pragma unroll 2
    for (int k = 0; k < B; ++k) {
        const double h = H1[(size_t)k * Hp];
#pragma unroll
        for (int r = 0; r < kGradRows; ++r) acc[r] += DZ2[(size_t)k * Hp + r] * h;
    }
    const double scale = 1.0 / (double)(B > 1 ? B : 1);
#pragma unroll
    for (int r = 0; r < kGradRows; ++r) {
        const int i = i0 + r;
        if (i < H) {
            const double g = clip_sym(acc[r] * scale, hp.grad_clip);
            const double w = net[a.nl.w2() + (size_t)i * Hp + j] - hp.lr * g;
            net[a.nl.w2() + (size_t)i * Hp + j] = w;
            net[a.nl.w2t() + (size_t)j * Hp + i] = w;
        }
    }
}

// Sum of term(0) .. term(B-1) in that order, starting from +0.0.  The terms of a group are formed first (their
// loads are independent and overlap), then added one after the other.
template <class F>
__device__ inline double ordered_sum(int B, F term) {
    constexpr int G = 16;
    double acc = 0.0;
    int k = 0;
    for (; k + G <= B; k += G) {
        double v[G];
#pragma unroll
        for (int u = 0; u < G; ++u) v[u] = term(k + u);
#pragma unroll
        for (int u = 0; u < G; ++u) acc += v[u];
    }
    for (; k < B; ++k) acc += term(k);
    return acc;
}

// The remaining gradients, one thread per parameter, summed in sample order (MLP.scala:49-72), then mean,
// clip, SGD (:78-96); one more thread forms the batch loss (:46,98).
__global__ void __launch_bounds__(256) agents_grad_small_kernel(AgentsArgs a) {
    const int slot = blockIdx.y, B = a.slot_rows[slot];
    const int H = a.nl.H, Hp = a.nl.Hp;
    const int agent = a.slot_agent[slot];
    double* net = net_ptr(a, agent, 0);
    const AgentHyperDev hp = a.hyper[agent];
    const size_t base = (size_t)slot * a.Bp;
    const double* __restrict__ X0 = a.X0d + base * kXStride;
    const double* __restrict__ DZ1 = a.DZ1 + base * Hp;
    const double* __restrict__ DZ2 = a.DZ2 + base * Hp;
    const double* __restrict__ H2 = a.H2 + base * Hp;
    const double* __restrict__ ERR = a.ERR + base;
    const int32_t* __restrict__ act = a.ACTd + base;
    const double scale = 1.0 / (double)(B > 1 ? B : 1);
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    double acc;
    size_t dst;
    if (e < H * kIn) {  // gW1(i)(j) += dz1(i) * x(j)
        const int i = e % H, j = e / H;
        acc = ordered_sum(B, [&](int k) { return DZ1[(size_t)k * Hp + i] * X0[(size_t)k * kXStride + j]; });
        dst = a.nl.w1t() + (size_t)j * Hp + i;
    } else if ((e -= H * kIn) < H) {  // gb1
        acc = ordered_sum(B, [&](int k) { return DZ1[(size_t)k * Hp + e]; });
        dst = a.nl.b1() + e;
    } else if ((e -= H) < H) {  // gb2
        acc = ordered_sum(B, [&](int k) { return DZ2[(size_t)k * Hp + e]; });
        dst = a.nl.b2() + e;
    } else if ((e -= H) < kOut * H) {  // gW3(o)(j) += err(o) * h2(j): err is +0.0 except at the action taken,
        const int j = e % H, o = e / H;  // and adding +0.0 leaves a sum as it is
        acc = ordered_sum(B, [&](int k) { return act[k] == o ? ERR[k] * H2[(size_t)k * Hp + j] : 0.0; });
        dst = a.nl.w3() + (size_t)o * Hp + j;
    } else if ((e -= kOut * H) < kOut) {  // gb3
        acc = ordered_sum(B, [&](int k) { return act[k] == e ? ERR[k] : 0.0; });
        dst = a.nl.b3() + e;
    } else if (e == kOut) {  // lossSum / n
        acc = ordered_sum(B, [&](int k) { return 0.5 * ERR[k] * ERR[k]; });
        a.loss[slot] = acc / (double)(B > 1 ? B : 1);
        return;
    } else {
        return;
    }
    net[dst] -= hp.lr * clip_sym(acc * scale, hp.grad_clip);
}

// DDQNAgent.observe -> ReplayBuffer.push (ReplayBuffer.scala:15-17): staged records into their ring slots.
__global__ void __launch_bounds__(64) agents_push_kernel(RingRec* ring, const RingRec* staged, const long long* dst, int count) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count) ring[dst[k]] = staged[k];
}

// DDQNAgent.observe for the last step of a world, filed from the world's own device planes: agent c receives, for
// every instance e, Transition(s, a, r, s2, done) with s = the state before the step, s2 = the state right after
// env.step (WorldToy.scala:90,99: before the mobility flows, env.step's own clock), done as env.scala:137.
__global__ __launch_bounds__(64) void agents_file_kernel(RingRec* ring, const long long* dst, WorldTransitionView v) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;   // k = c * batch + e
    if (k >= v.n * v.batch) return;
    const int c = k / v.batch, e = k - c * v.batch;
    const size_t idx = (size_t)e * v.npad + c;
    RingRec rec;
    const double tm = (double)(v.t_max > 1 ? v.t_max : 1);
    {
        const double s = (double)v.prev_st[0 * v.plane + idx], i = (double)v.prev_st[1 * v.plane + idx];
        const double r = (double)v.prev_st[2 * v.plane + idx], d = (double)v.prev_st[3 * v.plane + idx];
        const double vv = (double)v.prev_st[4 * v.plane + idx];
        const double N = fmax(1.0, s + i + r + d);
        rec.fs[0] = s / N; rec.fs[1] = i / N; rec.fs[2] = r / N; rec.fs[3] = d / N; rec.fs[4] = vv / N;
        rec.fs[5] = i / fmax(1.0, v.prev_cap[idx]);
        rec.fs[6] = (double)v.prev_tcur[idx] / tm;
        rec.fs[7] = 0.0;
    }
    {
        const double s = (double)v.loc_s[idx], i = (double)v.loc_i[idx];
        const double r = (double)v.st[2 * v.plane + idx], d = (double)v.st[3 * v.plane + idx], vv = (double)v.st[4 * v.plane + idx];
        const int32_t t2 = (int32_t)((uint32_t)v.prev_tcur[idx] + 1u);   // env.scala:132
        const double n4 = s + i + r + d, N = fmax(1.0, n4);
        rec.fs2[0] = s / N; rec.fs2[1] = i / N; rec.fs2[2] = r / N; rec.fs2[3] = d / N; rec.fs2[4] = vv / N;
        rec.fs2[5] = i / fmax(1.0, v.cap[idx]);
        rec.fs2[6] = (double)t2 / tm;
        rec.fs2[7] = 0.0;
        rec.done = (t2 >= v.t_max || i / N < 1e-5 || d / N > 0.02) ? 1 : 0;   // env.scala:137
    }
    rec.r = v.reward[idx];
    rec.a = (int32_t)v.act[idx];
    ring[dst[k]] = rec;
}

// State.features (env.scala:26-30)
void features(const epi_state& s, double* f) {
    const double n = s.s + s.i + s.r + s.d;
    const double N = std::fmax(1.0, n);
    f[0] = s.s / N; f[1] = s.i / N; f[2] = s.r / N; f[3] = s.d / N; f[4] = s.v / N;
    f[5] = s.i / std::fmax(1.0, s.hosp_cap);
    f[6] = (double)s.t / (double)std::max(1, s.t_max);
}

struct Replay {  // ReplayBuffer.scala:10-22: the bookkeeping; the records themselves live in the device ring
    int capacity = 0, size = 0, idx = 0;
    JavaRandom rnd{123};
    std::vector<int32_t> perm;
    int push() {  // the slot the next transition goes to
        if (size < capacity) return size++;
        const int at = idx;
        idx = (idx + 1) % capacity;
        return at;
    }
    // positions of rnd.shuffle(buf).take(n): scala.util.Random.shuffle copies the buffer and, for m from its length
    // down to 2, swaps element m-1 with element nextInt(m)
    const int32_t* sample() {
        const int len = size;
        perm.resize((size_t)len);
        for (int i = 0; i < len; ++i) perm[i] = i;
        for (int m = len; m >= 2; --m) std::swap(perm[m - 1], perm[rnd.nextInt(m)]);
        return perm.data();
    }
};

}  // namespace
}  // namespace epi

using namespace epi;

struct epi_agents {
    int device = 0, n = 0, Bp = 0;
    NetLayout nl{};
    std::vector<epi_learn_hyper> hp;
    std::vector<Replay> replay;
    std::vector<double> ema;
    DevBuf nets, hyper, H1, H2, Q, DZ2, DZ1, ERR, X0d, ACTd, ring, ring_off;
    PinBuf slot_agent, slot_rows, X, idx, loss, push_rec, push_dst, file_dst, q_in, q_out, q_act;
    std::vector<long long> h_ring_off;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing = false;
    double kernel_ms = 0.0;
    int64_t learn_calls = 0, samples = 0;
    ~epi_agents() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
    }
    AgentsArgs args() const {
        AgentsArgs a{};
        a.nl = nl; a.Bp = Bp;
        a.nets = nets.as<double>(); a.hyper = hyper.as<AgentHyperDev>();
        a.slot_agent = slot_agent.on_device<int32_t>(); a.slot_rows = slot_rows.on_device<int32_t>();
        a.ring = ring.as<RingRec>(); a.ring_off = ring_off.as<long long>();
        a.idx = idx.on_device<int32_t>(); a.X = X.on_device<double>();
        a.H1 = H1.as<double>(); a.H2 = H2.as<double>(); a.Q = Q.as<double>(); a.DZ2 = DZ2.as<double>();
        a.DZ1 = DZ1.as<double>(); a.ERR = ERR.as<double>(); a.X0d = X0d.as<double>(); a.ACTd = ACTd.as<int32_t>();
        a.loss = loss.on_device<double>();
        return a;
    }
};

namespace {

int32_t check_ids(const epi_agents* g, const int32_t* ids, int32_t count) {
    if (count < 0 || count > g->n) return fail(EPI_ERR_BAD_ARGUMENT, "count %d outside [0,%d]", count, g->n);
    if (!ids) return EPI_OK;
    std::vector<char> seen((size_t)g->n, 0);
    for (int k = 0; k < count; ++k) {
        if (ids[k] < 0 || ids[k] >= g->n) return fail(EPI_ERR_BAD_ARGUMENT, "agent id %d outside [0,%d)", ids[k], g->n);
        if (seen[ids[k]]) return fail(EPI_ERR_BAD_ARGUMENT, "agent id %d listed twice", ids[k]);
        seen[ids[k]] = 1;
    }
    return EPI_OK;
}

}  // namespace

EPI_API epi_agents* epi_agents_create(int32_t n_agents, const epi_learn_hyper* hp, int32_t device) {
    if (n_agents <= 0 || !hp) { fail(EPI_ERR_BAD_ARGUMENT, "agents: n_agents = %d, hp = %p", n_agents, (const void*)hp); return nullptr; }
    const int H = hp[0].hidden;
    int max_need = 0;
    for (int k = 0; k < n_agents; ++k) {
        if (hp[k].hidden != H) { fail(EPI_ERR_UNSUPPORTED, "agents: one handle holds agents of one hidden width (%d vs %d)", hp[k].hidden, H); return nullptr; }
        if (hp[k].replay_capacity <= 0 || hp[k].target_update_every <= 0 || hp[k].max_need <= 0) {
            fail(EPI_ERR_BAD_ARGUMENT, "agents: agent %d: replay_capacity, target_update_every and max_need must be positive", k);
            return nullptr;
        }
        max_need = std::max(max_need, hp[k].max_need);
    }
    if (H < 1 || H > kMaxHidden) { fail(EPI_ERR_UNSUPPORTED, "agents: hidden = %d outside [1,%d]", H, kMaxHidden); return nullptr; }
    if (use_device(device) != EPI_OK) return nullptr;
    std::unique_ptr<epi_agents> g(new epi_agents);
    g->device = device; g->n = n_agents;
    g->nl.H = H; g->nl.Hp = (H + 63) / 64 * 64;
    g->Bp = (max_need + kTileRows - 1) / kTileRows * kTileRows;
    g->hp.assign(hp, hp + n_agents);
    g->replay.resize((size_t)n_agents);
    g->ema.assign((size_t)n_agents, 0.0);
    for (int k = 0; k < n_agents; ++k) g->replay[k].capacity = hp[k].replay_capacity;
    auto ok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess) { fail(EPI_ERR_DEVICE, "agents: %s: %s", what, hipGetErrorString(e)); return false; }
        return true;
    };
    // DDQNAgent.scala:8-9: both networks start from the same Random(7) stream (MLP.scala:5-9)
    std::vector<double> W1, W2, W3, one(g->nl.size()), all((size_t)n_agents * 2 * g->nl.size());
    mlp_init(kIn, H, kOut, W1, W2, W3);
    pack_net(g->nl, W1.data(), nullptr, W2.data(), nullptr, W3.data(), nullptr, one.data());
    for (int k = 0; k < n_agents * 2; ++k) std::copy(one.begin(), one.end(), all.begin() + (size_t)k * g->nl.size());
    std::vector<AgentHyperDev> hd((size_t)n_agents);
    for (int k = 0; k < n_agents; ++k) hd[k] = {hp[k].gamma, hp[k].lr, hp[k].clip_q, hp[k].grad_clip};
    const size_t rows = (size_t)n_agents * g->Bp, Hp = (size_t)g->nl.Hp;
    long long ring_total = 0;
    for (int k = 0; k < n_agents; ++k) { g->h_ring_off.push_back(ring_total); ring_total += hp[k].replay_capacity; }
    if (!ok(g->nets.upload(all), "networks") || !ok(g->hyper.upload(hd), "hyper") ||
        !ok(g->H1.ensure(rows * Hp * 8), "H1") || !ok(g->H2.ensure(rows * Hp * 8), "H2") ||
        !ok(g->Q.ensure(rows * 3 * kOut * 8), "Q") || !ok(g->DZ2.ensure(rows * Hp * 8), "DZ2") ||
        !ok(g->DZ1.ensure(rows * Hp * 8), "DZ1") || !ok(g->ERR.ensure(rows * 8), "ERR") ||
        !ok(g->X0d.ensure(rows * kXStride * 8), "X0d") || !ok(g->ACTd.ensure(rows * 4), "ACTd") ||
        !ok(g->slot_agent.ensure((size_t)n_agents * 4), "slots") || !ok(g->slot_rows.ensure((size_t)n_agents * 4), "slots") ||
        !ok(g->X.ensure((size_t)n_agents * kXStride * 8), "X") || !ok(g->idx.ensure(rows * 4), "idx") ||
        !ok(g->push_rec.ensure((size_t)n_agents * sizeof(RingRec)), "push") || !ok(g->push_dst.ensure((size_t)n_agents * 8), "push") ||
        !ok(g->ring.ensure((size_t)ring_total * sizeof(RingRec)), "replay rings") || !ok(g->ring_off.upload(g->h_ring_off), "replay rings") ||
        !ok(g->loss.ensure((size_t)n_agents * 8), "loss") ||
        !ok(hipMemset(g->DZ2.p, 0, rows * Hp * 8), "clear") ||  // columns past H are read by the gradient kernel
        !ok(hipFuncSetAttribute(reinterpret_cast<const void*>(agents_forward_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds_bytes()), "lds") ||
        !ok(hipFuncSetAttribute(reinterpret_cast<const void*>(agents_backward_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds_bytes()), "lds") ||
        !ok(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking), "stream") ||
        !ok(hipEventCreate(&g->ev0), "event") || !ok(hipEventCreate(&g->ev1), "event"))
        return nullptr;
    return g.release();
}

// The networks of the handle for kernels of other translation units (greedy rollouts of a world whose nodes ask
// their own agents): online network of agent k at nets + k * net_stride, its clamp at clip_ptr[k * clip_stride].
namespace epi {
int32_t agents_net_view(epi_agents* g, AgentsNetView* v) {
    if (!g || !v) return fail(EPI_ERR_BAD_ARGUMENT, "agents_net_view: null argument");
    HIP_TRY(hipSetDevice(g->device));
    HIP_TRY(hipStreamSynchronize(g->stream));   // a learn call still in flight finishes before its weights are read
    v->device = g->device; v->n_agents = g->n; v->nl = g->nl;
    v->nets = g->nets.as<double>(); v->net_stride = 2 * g->nl.size();
    v->clip_ptr = reinterpret_cast<const double*>(g->hyper.as<AgentHyperDev>()) + 2;   // AgentHyperDev::clip_q
    v->clip_stride = (int)(sizeof(AgentHyperDev) / sizeof(double));
    return EPI_OK;
}
}  // namespace epi

EPI_API int32_t epi_agents_destroy(epi_agents* g) {
    if (!g) return EPI_OK;
    (void)hipSetDevice(g->device);
    (void)hipStreamSynchronize(g->stream);
    delete g;
    return EPI_OK;
}

EPI_API int32_t epi_agents_observe(epi_agents* g, const int32_t* ids, int32_t count, const epi_state* s, const int32_t* a,
                                   const double* r, const epi_state* s2, const uint8_t* done) {
    if (!g || !s || !a || !r || !s2 || !done) return fail(EPI_ERR_BAD_ARGUMENT, "agents_observe: null argument");
    if (int32_t rc = check_ids(g, ids, count)) return rc;
    if (count == 0) return EPI_OK;
    for (int k = 0; k < count; ++k)
        if (a[k] < 0 || a[k] >= kOut) return fail(EPI_ERR_BAD_ARGUMENT, "agents_observe: action %d outside [0,%d)", a[k], kOut);
    if (int32_t rc = use_device(g->device)) return rc;
    RingRec* rec = g->push_rec.as<RingRec>();
    long long* dst = g->push_dst.as<long long>();
    for (int k = 0; k < count; ++k) {
        const int id = ids ? ids[k] : k;
        features(s[k], rec[k].fs); features(s2[k], rec[k].fs2);
        rec[k].fs[7] = rec[k].fs2[7] = 0.0;
        rec[k].r = r[k]; rec[k].a = a[k]; rec[k].done = done[k] ? 1 : 0;
        dst[k] = g->h_ring_off[id] + g->replay[id].push();
    }
    hipLaunchKernelGGL(agents_push_kernel, dim3((count + 63) / 64), dim3(64), 0, g->stream, g->ring.as<RingRec>(),
                       g->push_rec.on_device<RingRec>(), g->push_dst.on_device<long long>(), count);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g->stream));  // the staging arrays are free again
    return EPI_OK;
}

EPI_API int32_t epi_agents_observe_world(epi_agents* g, epi_world* w) {
    if (!g || !w) return fail(EPI_ERR_BAD_ARGUMENT, "agents_observe_world: null argument");
    WorldTransitionView v{};
    if (int32_t rc = world_transition_view(w, &v)) return rc;
    if (v.device != g->device) return fail(EPI_ERR_BAD_ARGUMENT, "agents_observe_world: the world lives on device %d, the agents on device %d", v.device, g->device);
    if (v.n > g->n) return fail(EPI_ERR_BAD_ARGUMENT, "agents_observe_world: %d countries, %d agents (one agent per node, WorldToy.scala:50)", v.n, g->n);
    if (int32_t rc = use_device(g->device)) return rc;
    const size_t count = (size_t)v.n * v.batch;
    HIP_TRY(g->file_dst.ensure(count * sizeof(long long)));
    long long* dst = g->file_dst.as<long long>();
    for (int c = 0; c < v.n; ++c)
        for (int e = 0; e < v.batch; ++e) dst[(size_t)c * v.batch + e] = g->h_ring_off[c] + g->replay[c].push();
    hipLaunchKernelGGL(agents_file_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, g->stream, g->ring.as<RingRec>(),
                       g->file_dst.on_device<long long>(), v);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g->stream));  // the staging array is free again
    world_transition_consumed(w);               // a second call without a step in between must not file it again
    return EPI_OK;
}

// The same filing for a calibration sweep whose rollouts learn: every (episode, country) pair of the view is a node of
// its own world with its own agent (Calibrate.scala:56-65 builds a fresh WorldToy per grid point), so instance e's
// country c files into agent e * v.n + c.
namespace epi {
int32_t agents_observe_instances(epi_agents* g, const WorldTransitionView& v) {
    if (!g) return fail(EPI_ERR_BAD_ARGUMENT, "agents_observe_instances: null handle");
    if (v.device != g->device) return fail(EPI_ERR_BAD_ARGUMENT, "agents_observe_instances: planes on device %d, agents on device %d", v.device, g->device);
    if ((long long)v.n * v.batch != g->n) return fail(EPI_ERR_BAD_ARGUMENT, "agents_observe_instances: %d x %d transitions for %d agents", v.batch, v.n, g->n);
    if (int32_t rc = use_device(g->device)) return rc;
    const size_t count = (size_t)v.n * v.batch;
    HIP_TRY(g->file_dst.ensure(count * sizeof(long long)));
    long long* dst = g->file_dst.as<long long>();
    for (int c = 0; c < v.n; ++c)
        for (int e = 0; e < v.batch; ++e) {
            const size_t agent = (size_t)e * v.n + c;
            dst[(size_t)c * v.batch + e] = g->h_ring_off[agent] + g->replay[agent].push();
        }
    hipLaunchKernelGGL(agents_file_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, g->stream, g->ring.as<RingRec>(),
                       g->file_dst.on_device<long long>(), v);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g->stream));  // the staging array is free again
    return EPI_OK;
}
}  // namespace epi

EPI_API int32_t epi_agents_replay_size(epi_agents* g, int32_t* sizes) {
    if (!g || !sizes) return fail(EPI_ERR_BAD_ARGUMENT, "agents_replay_size: null argument");
    for (int k = 0; k < g->n; ++k) sizes[k] = g->replay[k].size;
    return EPI_OK;
}

EPI_API int32_t epi_agents_sample(epi_agents* g, int32_t agent, int32_t n, int32_t* out, int32_t* taken) {
    if (!g || !out || !taken || agent < 0 || agent >= g->n || n < 0) return fail(EPI_ERR_BAD_ARGUMENT, "agents_sample: bad argument");
    Replay& rp = g->replay[agent];
    const int32_t* p = rp.sample();
    *taken = std::min<int32_t>(n, rp.size);
    std::copy(p, p + *taken, out);
    return EPI_OK;
}

EPI_API int32_t epi_agents_learn(epi_agents* g, const int32_t* ids, int32_t count, const int32_t* batch,
                                 const int32_t* grad_accum, const int64_t* steps, double* loss, double* ema_loss) {
    if (!g || !batch || !grad_accum || !steps || !loss) return fail(EPI_ERR_BAD_ARGUMENT, "agents_learn: null argument");
    if (int32_t rc = check_ids(g, ids, count)) return rc;
    if (int32_t rc = use_device(g->device)) return rc;
    // DDQNAgent.scala:52-54: agents whose buffer is short return 0.0 and draw nothing
    std::vector<int> slot_of((size_t)count, -1);
    int slots = 0;
    int32_t* sa = g->slot_agent.as<int32_t>();
    int32_t* sr = g->slot_rows.as<int32_t>();
    for (int k = 0; k < count; ++k) {
        const int id = ids ? ids[k] : k;
        if (batch[k] <= 0 || grad_accum[k] <= 0) return fail(EPI_ERR_BAD_ARGUMENT, "agents_learn: batch and grad_accum must be positive");
        const int64_t need64 = std::max<int64_t>((int64_t)batch[k] * grad_accum[k], batch[k]);
        loss[k] = 0.0;
        if ((int64_t)g->replay[id].size < need64) continue;
        if (need64 > g->hp[id].max_need)
            return fail(EPI_ERR_BAD_ARGUMENT, "agents_learn: agent %d needs %lld samples, the handle was created for %d", id, (long long)need64, g->hp[id].max_need);
        slot_of[k] = slots;
        sa[slots] = id; sr[slots] = (int32_t)need64;
        ++slots;
    }
    if (slots > 0) {
        const int Bp = g->Bp;
        // The shuffle walks the whole buffer with a sequential generator (ReplayBuffer.scala:21-22): agents are
        // independent, so their shuffles run on a few host threads.  Only the sampled positions cross the bus.
        auto stage = [&](int sl) {
            const int id = sa[sl], need = sr[sl];
            const int32_t* p = g->replay[id].sample();
            std::memcpy(g->idx.as<int32_t>() + (size_t)sl * Bp, p, sizeof(int32_t) * (size_t)need);
        };
        const int n_threads = std::max(1, std::min({slots / 4, 16, (int)std::thread::hardware_concurrency()}));
        if (n_threads == 1) {
            for (int sl = 0; sl < slots; ++sl) stage(sl);
        } else {
            std::vector<std::thread> pool;
            for (int th = 0; th < n_threads; ++th)
                pool.emplace_back([&, th] { for (int sl = th; sl < slots; sl += n_threads) stage(sl); });
            for (auto& th : pool) th.join();
        }
        AgentsArgs a = g->args();
        a.n_classes = 3;
        a.net_of_class[0] = 0; a.net_of_class[1] = 0; a.net_of_class[2] = 1;
        a.x_of_class[0] = 0; a.x_of_class[1] = 1; a.x_of_class[2] = 1;
        const int Hp = g->nl.Hp, H = g->nl.H;
        if (g->timing) HIP_TRY(hipEventRecord(g->ev0, g->stream));
        const size_t lds = tile_lds_bytes();
        // 16 hidden units per wave: tiles of 8 (twice the waves) measured slower on the default width
        hipLaunchKernelGGL(agents_forward_kernel<16>, dim3(3 * Bp / kTileRows, slots), dim3(Hp / 16 * 64), lds, g->stream, a, a.nets);
        hipLaunchKernelGGL(agents_backward_kernel<16>, dim3(Bp / kTileRows, slots), dim3(Hp / 16 * 64), lds, g->stream, a, a.nets);
        hipLaunchKernelGGL(agents_grad_w2_kernel, dim3((H + kGradRows - 1) / kGradRows, slots), dim3(Hp), 0, g->stream, a);
        hipLaunchKernelGGL(agents_grad_small_kernel, dim3((17 * H + 9 + 255) / 256, slots), dim3(256), 0, g->stream, a);
        HIP_TRY(hipGetLastError());
        // DDQNAgent.scala:59-61: the target network follows the updated online network
        for (int sl = 0; sl < slots; ++sl) {
            const int id = sa[sl];
            int kk = 0;
            while (slot_of[kk] != sl) ++kk;
            if (steps[kk] % g->hp[id].target_update_every == 0) {
                double* on = g->nets.as<double>() + ((size_t)id * 2) * g->nl.size();
                HIP_TRY(hipMemcpyAsync(on + g->nl.size(), on, g->nl.size() * 8, hipMemcpyDeviceToDevice, g->stream));
            }
        }
        if (g->timing) HIP_TRY(hipEventRecord(g->ev1, g->stream));
        HIP_TRY(hipStreamSynchronize(g->stream));
        if (g->timing) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, g->ev0, g->ev1));
            g->kernel_ms += ms; g->learn_calls += 1;
            for (int sl = 0; sl < slots; ++sl) g->samples += sr[sl];
        }
        for (int k = 0; k < count; ++k) {
            if (slot_of[k] < 0) continue;
            const int id = sa[slot_of[k]];
            loss[k] = g->loss.as<double>()[slot_of[k]];
            const double al = g->hp[id].ema_alpha;  // DDQNAgent.scala:63
            g->ema[id] = (g->ema[id] == 0.0) ? loss[k] : al * loss[k] + (1 - al) * g->ema[id];
        }
    }
    if (ema_loss)
        for (int k = 0; k < count; ++k) ema_loss[k] = g->ema[ids ? ids[k] : k];
    return EPI_OK;
}

EPI_API int32_t epi_agents_q_values(epi_agents* g, const int32_t* ids, int32_t count, const epi_state* s, int32_t use_target,
                                    double* q, int32_t* greedy) {
    if (!g || !s || (!q && !greedy)) return fail(EPI_ERR_BAD_ARGUMENT, "agents_q_values: null argument");
    if (int32_t rc = check_ids(g, ids, count)) return rc;
    if (count == 0) return EPI_OK;
    if (int32_t rc = use_device(g->device)) return rc;
    // DDQNAgent.actGreedy / MLP.predict (DDQNAgent.scala:22-23, MLP.scala:15-22), one state per asked agent: the
    // network kernels of the world step, each state with its agent's online (or target) network; states, Q-values
    // and actions pass through page-locked memory the kernel reads and writes itself.
    HIP_TRY(g->q_in.ensure((size_t)count * sizeof(epi_state)));
    HIP_TRY(g->q_out.ensure((size_t)count * kOut * sizeof(double)));
    HIP_TRY(g->q_act.ensure((size_t)count * sizeof(int32_t)));
    std::memcpy(g->q_in.p, s, (size_t)count * sizeof(epi_state));
    for (int k = 0; k < count; ++k) g->slot_agent.as<int32_t>()[k] = ids ? ids[k] : k;
    MlpArgs ma{};
    ma.rows = count; ma.n = count; ma.npad = count; ma.count = 1;
    ma.states = g->q_in.on_device<epi_state>();
    ma.nl = g->nl; ma.nets = g->nets.as<double>(); ma.net_stride = 2 * g->nl.size();
    ma.net_offset = use_target ? g->nl.size() : 0;
    ma.net_index = g->slot_agent.on_device<int32_t>();
    ma.per_country = 1;
    ma.clip_ptr = reinterpret_cast<const double*>(g->hyper.as<AgentHyperDev>()) + 2;   // AgentHyperDev::clip_q
    ma.clip_stride = (int)(sizeof(AgentHyperDev) / sizeof(double));
    ma.q_out = g->q_out.on_device<double>();
    ma.act_out = g->q_act.on_device<int32_t>();
    HIP_TRY(launch_mlp(ma, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    if (q) std::memcpy(q, g->q_out.p, (size_t)count * kOut * sizeof(double));
    if (greedy) std::memcpy(greedy, g->q_act.p, (size_t)count * sizeof(int32_t));
    return EPI_OK;
}

EPI_API int32_t epi_agents_get_weights(epi_agents* g, int32_t agent, int32_t target, double* W1, double* b1, double* W2,
                                       double* b2, double* W3, double* b3) {
    if (!g || agent < 0 || agent >= g->n) return fail(EPI_ERR_BAD_ARGUMENT, "agents_get_weights: bad argument");
    if (int32_t rc = use_device(g->device)) return rc;
    const NetLayout& nl = g->nl;
    std::vector<double> img(nl.size());
    HIP_TRY(hipStreamSynchronize(g->stream));
    HIP_TRY(hipMemcpy(img.data(), g->nets.as<double>() + ((size_t)agent * 2 + (target ? 1 : 0)) * nl.size(), nl.size() * 8,
                      hipMemcpyDeviceToHost));
    const int H = nl.H, Hp = nl.Hp;
    for (int i = 0; i < H; ++i) {
        if (W1) for (int j = 0; j < kIn; ++j) W1[(size_t)i * kIn + j] = img[nl.w1t() + (size_t)j * Hp + i];
        if (b1) b1[i] = img[nl.b1() + i];
        if (W2) for (int j = 0; j < H; ++j) W2[(size_t)i * H + j] = img[nl.w2() + (size_t)i * Hp + j];
        if (b2) b2[i] = img[nl.b2() + i];
    }
    for (int o = 0; o < kOut; ++o) {
        if (W3) for (int j = 0; j < H; ++j) W3[(size_t)o * H + j] = img[nl.w3() + (size_t)o * Hp + j];
        if (b3) b3[o] = img[nl.b3() + o];
    }
    return EPI_OK;
}

EPI_API int32_t epi_agents_set_weights(epi_agents* g, int32_t agent, int32_t target, const double* W1, const double* b1,
                                       const double* W2, const double* b2, const double* W3, const double* b3) {
    if (!g || agent < 0 || agent >= g->n || !W1 || !W2 || !W3) return fail(EPI_ERR_BAD_ARGUMENT, "agents_set_weights: bad argument");
    if (int32_t rc = use_device(g->device)) return rc;
    std::vector<double> img(g->nl.size());
    pack_net(g->nl, W1, b1, W2, b2, W3, b3, img.data());
    HIP_TRY(hipStreamSynchronize(g->stream));
    HIP_TRY(hipMemcpy(g->nets.as<double>() + ((size_t)agent * 2 + (target ? 1 : 0)) * g->nl.size(), img.data(),
                      g->nl.size() * 8, hipMemcpyHostToDevice));
    return EPI_OK;
}

EPI_API int32_t epi_agents_set_timing(epi_agents* g, int32_t on) {
    if (!g) return fail(EPI_ERR_BAD_ARGUMENT, "agents_set_timing: null handle");
    g->timing = on != 0; g->kernel_ms = 0.0; g->learn_calls = 0; g->samples = 0;
    return EPI_OK;
}

EPI_API int32_t epi_agents_kernel_time(epi_agents* g, double* total_ms, int64_t* learn_calls, int64_t* samples) {
    if (!g) return fail(EPI_ERR_BAD_ARGUMENT, "agents_kernel_time: null handle");
    if (total_ms) *total_ms = g->kernel_ms;
    if (learn_calls) *learn_calls = g->learn_calls;
    if (samples) *samples = g->samples;
    return EPI_OK;
}
