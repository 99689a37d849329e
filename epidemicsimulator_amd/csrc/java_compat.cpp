// JVM-compatible generators (see java_compat.hpp).  Compiled with -ffp-contract=off:
// every floating-point operation below must round exactly once, as on the JVM.
#include "java_compat.hpp"

#include <cmath>
#include <cstring>
#include <limits>

namespace epi {

namespace {
constexpr uint64_t kMul = 0x5DEECE66DULL;
constexpr uint64_t kAdd = 0xBULL;
constexpr uint64_t kMask = (1ULL << 48) - 1;

__host__ __device__ inline uint64_t bits_of(double x) { uint64_t b; __builtin_memcpy(&b, &x, 8); return b; }
__host__ __device__ inline double from_bits(uint64_t b) { double x; __builtin_memcpy(&x, &b, 8); return x; }
}  // namespace

void JavaRandom::reseed(int64_t seed) {
    state_ = (static_cast<uint64_t>(seed) ^ kMul) & kMask;
    have_cached_ = false;
}

int32_t JavaRandom::next(int bits) {
    state_ = (state_ * kMul + kAdd) & kMask;
    return static_cast<int32_t>(static_cast<uint32_t>(state_ >> (48 - bits)));
}

int32_t JavaRandom::nextInt(int32_t bound) {
    int32_t r = next(31);
    const int32_t m = bound - 1;
    if ((bound & m) == 0) return static_cast<int32_t>((static_cast<int64_t>(bound) * r) >> 31);
    for (int32_t u = r;; u = next(31)) {
        r = u % bound;
        const uint32_t t = static_cast<uint32_t>(u) - static_cast<uint32_t>(r) + static_cast<uint32_t>(m);
        if (static_cast<int32_t>(t) >= 0) return r;
    }
}

double JavaRandom::nextDouble() {
    const int64_t a = next(26);
    const int64_t b = next(27);
    return static_cast<double>((a << 27) + b) * 0x1.0p-53;
}

double JavaRandom::nextGaussian() {
    if (have_cached_) {
        have_cached_ = false;
        return cached_;
    }
    double v1, v2, s;
    do {
        v1 = 2 * nextDouble() - 1;
        v2 = 2 * nextDouble() - 1;
        s = v1 * v1 + v2 * v2;
    } while (s >= 1 || s == 0);
    const double mul = std::sqrt(-2 * strict_log(s) / s);
    cached_ = v2 * mul;
    have_cached_ = true;
    return v1 * mul;
}

__host__ __device__ double strict_log(double x) {
    // Constants of the fdlibm natural logarithm: ln2 split in two parts, and the
    // minimax coefficients of (log(1+f) - 2s)/s in s^2, s = f/(2+f).
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    constexpr double two54 = 1.80143985094819840000e+16;
    constexpr double c1 = 6.666666666666735130e-01, c2 = 3.999999999940941908e-01, c3 = 2.857142874366239149e-01,
                     c4 = 2.222219843214978396e-01, c5 = 1.818357216161805012e-01, c6 = 1.531383769920937332e-01,
                     c7 = 1.479819860511658591e-01;
    uint64_t b = bits_of(x);
    int32_t hx = static_cast<int32_t>(b >> 32);
    const uint32_t lx = static_cast<uint32_t>(b);
    int32_t k = 0;
    if (hx < 0x00100000) {  // zero, negative or subnormal
        if (((hx & 0x7fffffff) | lx) == 0) return -__builtin_huge_val();
        if (hx < 0) return __builtin_nan("");
        k -= 54;
        x *= two54;
        b = bits_of(x);
        hx = static_cast<int32_t>(b >> 32);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int32_t up = (hx + 0x95f64) & 0x100000;  // 1 when the mantissa is above sqrt(2)
    x = from_bits((b & 0xffffffffULL) | (static_cast<uint64_t>(static_cast<uint32_t>(hx | (up ^ 0x3ff00000))) << 32));
    k += up >> 20;
    const double f = x - 1.0;
    if ((0x000fffff & (2 + hx)) < 3) {  // |f| < 2^-20
        if (f == 0.0) {
            if (k == 0) return 0.0;
            const double dk = k;
            return dk * ln2_hi + dk * ln2_lo;
        }
        const double R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        const double dk = k;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    const double s = f / (2.0 + f);
    const double dk = k;
    const double z = s * s;
    const double w = z * z;
    const int32_t lim = (hx - 0x6147a) | (0x6b851 - hx);
    const double even = w * (c2 + w * (c4 + w * c6));
    const double odd = z * (c1 + w * (c3 + w * (c5 + w * c7)));
    const double R = odd + even;
    if (lim > 0) {
        const double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

// ---- the same generator on the device: one stream per thread ------------------------------------------------------
namespace {
struct DeviceRandom {
    uint64_t state;
    bool have_cached;
    double cached;
    __device__ void reseed(int64_t seed) {
        state = (static_cast<uint64_t>(seed) ^ kMul) & kMask;
        have_cached = false;
        cached = 0.0;
    }
    __device__ int32_t next(int bits) {
        state = (state * kMul + kAdd) & kMask;
        return static_cast<int32_t>(static_cast<uint32_t>(state >> (48 - bits)));
    }
    __device__ double nextDouble() {
        const int64_t a = next(26);
        const int64_t b = next(27);
        return static_cast<double>((a << 27) + b) * 0x1.0p-53;
    }
    __device__ double nextGaussian() {
        if (have_cached) {
            have_cached = false;
            return cached;
        }
        double v1, v2, s;
        do {
            v1 = 2 * nextDouble() - 1;
            v2 = 2 * nextDouble() - 1;
            s = v1 * v1 + v2 * v2;
        } while (s >= 1 || s == 0);
        const double mul = sqrt(-2 * strict_log(s) / s);   // IEEE square root and division on both sides
        cached = v2 * mul;
        have_cached = true;
        return v1 * mul;
    }
};

__global__ void __launch_bounds__(64) gaussian_table_kernel(double* out, const int32_t* slot_of, int streams, int draws,
                                                            size_t ld, int64_t seed0) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= streams) return;
    DeviceRandom r;
    r.reseed(seed0 + c);
    double* col = out + (slot_of ? slot_of[c] : c);
    for (int t = 0; t < draws; ++t) col[(size_t)t * ld] = r.nextGaussian();
}
}  // namespace

hipError_t launch_gaussian_table(double* out, const int32_t* slot_of, int streams, int draws, size_t ld, int64_t seed0,
                                 hipStream_t stream) {
    if (streams <= 0 || draws <= 0) return hipSuccess;
    hipLaunchKernelGGL(gaussian_table_kernel, dim3((unsigned)((streams + 63) / 64)), dim3(64), 0, stream, out, slot_of, streams,
                       draws, ld, seed0);
    return hipGetLastError();
}

int AgentStream::act(const Hyper& hp) {
    // act(s) = actWithEpsilon(s, epsilon, countStep = true): the argument is evaluated before the call, so
    // epsilon is read with the step count as it was, and the count goes up inside (DDQNAgent.scala:26-27,32)
    const int32_t den = hp.epsilon_decay_steps > 1 ? hp.epsilon_decay_steps : 1;
    const double frac = std::fmin(1.0, static_cast<double>(steps) / static_cast<double>(den));
    const double eps = hp.epsilon_start + (hp.epsilon_end - hp.epsilon_start) * frac;
    steps += 1;
    if (rnd.nextDouble() < eps) return rnd.nextInt(8);
    return kGreedyAction;
}

void mlp_init(int in, int h, int out, std::vector<double>& W1, std::vector<double>& W2, std::vector<double>& W3) {
    JavaRandom r(7);
    W1.resize(static_cast<size_t>(h) * in);
    W2.resize(static_cast<size_t>(h) * h);
    W3.resize(static_cast<size_t>(out) * h);
    for (auto& v : W1) v = r.nextGaussian() * 0.05;
    for (auto& v : W2) v = r.nextGaussian() * 0.05;
    for (auto& v : W3) v = r.nextGaussian() * 0.05;
}

}  // namespace epi
