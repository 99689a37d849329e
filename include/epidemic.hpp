// epidemic.hpp -- C++ host-side mirror of the reference's Scala interface for the metapopulation
// step and the calibration sweep, over the C ABI of include/epi.h.
//
// Same names, argument meaning and error behaviour as the Scala sources under
// src/main/scala/epidemic/ (cited per item); the arithmetic of the path runs in libepi.so on the
// GPU.  `require(...)` failures of the reference (IllegalArgumentException) surface here as
// epidemic::IllegalArgumentException.  Header-only; link with -lepi.
#pragma once
#include <array>
#include <cmath>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "epi.h"

namespace epidemic {

struct IllegalArgumentException : std::invalid_argument {
    using std::invalid_argument::invalid_argument;
};

inline void check(int32_t status) {
    if (status == EPI_OK) return;
    char buf[1024];
    epi_last_error(buf, sizeof buf);
    if (status == EPI_ERR_LOCAL_MASS || status == EPI_ERR_WORLD_MASS || status == EPI_ERR_ASSERT_STATE)
        throw IllegalArgumentException(buf);
    throw std::runtime_error(std::string("epi status ") + std::to_string(status) + ": " + buf);
}

// ---- env.scala:6-30 -----------------------------------------------------------------------------
enum class Action : int { NoneA = 0, Distancing, TravelBan, Lockdown, NormalCare, SurgeCare, TargetedVax, MassVax };
inline Action fromId(int i) { return static_cast<Action>(i < 0 ? 0 : (i > 7 ? 7 : i)); }  // env.scala:18-20

struct State {  // env.scala:22-30
    double s = 0, i = 0, r = 0, d = 0, v = 0, hospCap = 0;
    int t = 0, tMax = 0, seed = 0;
    double n() const { return s + i + r + d; }
    double util() const { return i / std::fmax(1.0, hospCap); }
    std::array<double, 7> features() const {
        const double N = std::fmax(1.0, n());
        return {s / N, i / N, r / N, d / N, v / N, util(), static_cast<double>(t) / (tMax > 1 ? tMax : 1)};
    }
    epi_state c() const { return epi_state{s, i, r, d, v, hospCap, t, tMax, seed, 0}; }
    static State from(const epi_state& c) { return State{c.s, c.i, c.r, c.d, c.v, c.hosp_cap, c.t, c.t_max, c.seed}; }
    bool operator==(const State& o) const {
        return s == o.s && i == o.i && r == o.r && d == o.d && v == o.v && hospCap == o.hospCap && t == o.t && tMax == o.tMax;
    }
};

// ---- config.scala ---------------------------------------------------------------------------------
struct HyperParams {
    double gamma = 0.99, lr = 1e-3, epsilonStart = 1.0, epsilonEnd = 0.1;
    int epsilonDecaySteps = 100000, targetUpdateEvery = 1500, replayCapacity = 50000, batchSize = 128, hidden = 128;
    double clipQ = 10.0, rewardClip = 5.0, gradClip = 1.0;
    int logInterval = 500;
    double emaAlpha = 0.0;
    bool checkpoint = true;
    int learnEvery = 8;
    epi_hyper c() const { return epi_hyper{epsilonStart, epsilonEnd, epsilonDecaySteps, hidden, clipQ, rewardClip, batchSize, learnEvery}; }
    // the fields DDQNAgent.learn reads (DDQNAgent.scala:51-65); maxNeed = the largest batch * gradAccumSteps asked for
    epi_learn_hyper learn(int maxNeed) const {
        return epi_learn_hyper{gamma, lr, clipQ, gradClip, emaAlpha, hidden, replayCapacity, targetUpdateEvery, maxNeed};
    }
};
struct TrainConfig {
    int epochs = 1000, stepsPerEpoch = 500, evalEvery = 10;
};

// ---- Geo.scala ------------------------------------------------------------------------------------
namespace Geo {
struct Country {  // Geo.scala:4-21
    std::string name;
    int pop = 0;
    double lat = 0, lon = 0, hospPer1k = 3.0, betaMul = 1.0, ifrMul = 1.0, noise = 0.05, populationDensity = 100.0,
           gdpPerCapita = 30000.0, healthcareCapacity = 0.7, mobilityIndex = 0.8, ageDistributionElderly = 0.15,
           climateFactor = 0.6, connectivityStrength = 0.8, economicResilience = 0.7;
    epi_profile c() const {
        return epi_profile{betaMul, ifrMul, noise, populationDensity, gdpPerCapita, healthcareCapacity, ageDistributionElderly,
                           connectivityStrength, economicResilience};
    }
};
inline Country getProfile(const std::string& name) {  // CountryProfiles.getProfile, Geo.scala:24-68
    if (name == "USA") return {"USA", 331000000, 39.8, -98.5, 2.9, 1.0, 1.0, 0.05, 36.0, 65000.0, 0.85, 0.9, 0.20, 0.7, 0.95, 0.9};
    if (name == "Japan") return {"Japan", 125000000, 36.2, 138.2, 13.1, 0.8, 0.9, 0.03, 347.0, 42000.0, 0.95, 0.6, 0.30, 0.5, 0.8, 0.8};
    if (name == "India") return {"India", 1380000000, 20.6, 78.9, 0.5, 1.3, 1.2, 0.08, 464.0, 2100.0, 0.3, 0.4, 0.07, 0.8, 0.4, 0.3};
    if (name == "Italy") return {"Italy", 60000000, 41.9, 12.6, 3.2, 1.1, 1.4, 0.04, 206.0, 35000.0, 0.8, 0.7, 0.28, 0.6, 0.85, 0.7};
    if (name == "Germany") return {"Germany", 83000000, 51.2, 10.5, 8.0, 0.9, 0.8, 0.03, 240.0, 46000.0, 0.9, 0.8, 0.25, 0.4, 0.9, 0.85};
    if (name == "China") return {"China", 1400000000, 35.9, 104.2, 4.3, 1.0, 1.0, 0.06, 153.0, 10500.0, 0.6, 0.5, 0.18, 0.7, 0.7, 0.6};
    Country c;
    c.name = name;
    c.pop = 50000000;
    return c;
}
inline double haversineKm(double lat1, double lon1, double lat2, double lon2) {  // Geo.scala:72-79
    const double R = 6371.0, rad = 3.14159265358979323846 / 180.0;
    const double dLat = (lat2 - lat1) * rad, dLon = (lon2 - lon1) * rad;
    const double a = std::pow(std::sin(dLat / 2), 2) + std::cos(lat1 * rad) * std::cos(lat2 * rad) * std::pow(std::sin(dLon / 2), 2);
    return R * (2 * std::atan2(std::sqrt(a), std::sqrt(1 - a)));
}
}  // namespace Geo

using Matrix = std::vector<std::vector<double>>;

inline std::vector<double> flatten(const Matrix& W) {
    std::vector<double> f;
    for (const auto& r : W) f.insert(f.end(), r.begin(), r.end());
    return f;
}

// ---- ToyConn.scala --------------------------------------------------------------------------------
namespace ToyConn {
inline const std::vector<Geo::Country>& C() {  // ToyConn.scala:7-14
    static const std::vector<Geo::Country> c = {
        {"USA", 330000000, 38.0, -97.0, 2.9, 1.00, 1.00, 0.05},    {"Japan", 125000000, 36.0, 138.0, 7.8, 0.92, 0.90, 0.04},
        {"Italy", 60000000, 42.5, 12.5, 3.2, 0.98, 1.05, 0.05},    {"India", 1420000000, 22.0, 79.0, 0.7, 1.05, 0.95, 0.06},
        {"Germany", 83000000, 51.0, 10.0, 8.0, 0.96, 0.95, 0.04}, {"China", 1410000000, 35.0, 103.0, 4.3, 1.03, 1.00, 0.05}};
    return c;
}
inline double flightMul(const std::string& a, const std::string& b) {  // ToyConn.scala:16-20
    static const std::map<std::pair<std::string, std::string>, double> m = {
        {{"China", "India"}, 1.8}, {{"India", "China"}, 1.2}, {{"USA", "Japan"}, 1.6}, {{"Japan", "USA"}, 1.6},
        {{"Germany", "Italy"}, 1.3}, {{"Italy", "Germany"}, 1.2}};
    auto it = m.find({a, b});
    return it == m.end() ? 1.0 : it->second;
}
// ToyConn.scala:22-47.  Uses the host libm; the JVM's Math.pow/sin/cos may differ in the last ulp, so the
// matrix is an engine input rather than something the engine reproduces.
inline Matrix gravityMatrix(const std::vector<Geo::Country>& cs, double a = 0.2, double b = 1.0, double g = 1.2, double k = 1e-6,
                            double capRow = 0.12) {
    const size_t n = cs.size();
    Matrix W(n, std::vector<double>(n, 0.0));
    for (size_t i = 0; i < n; ++i) {
        double rowSum = 0.0;
        for (size_t j = 0; j < n; ++j)
            if (i != j) {
                const double d = std::fmax(300.0, Geo::haversineKm(cs[i].lat, cs[i].lon, cs[j].lat, cs[j].lon));
                const double w = k * std::pow((double)cs[i].pop, a) * std::pow((double)cs[j].pop, b) * flightMul(cs[i].name, cs[j].name) / std::pow(d, g);
                W[i][j] = w;
                rowSum += w;
            }
        if (rowSum > 0) {
            const double scale = std::fmin(1.0, capRow / rowSum);
            for (size_t j = 0; j < n; ++j)
                if (i != j) W[i][j] *= scale;
        }
        W[i][i] = 0.0;
    }
    return W;
}
// gravityMatrix computed on the GPU (large synthetic graphs).  Entries may differ from gravityMatrix() in the
// last place: the device library's pow/sin/cos/atan2 are not the host's.
inline Matrix gravityMatrixDevice(const std::vector<Geo::Country>& cs, double a = 0.2, double b = 1.0, double g = 1.2, double k = 1e-6,
                                  double capRow = 0.12) {
    const int n = (int)cs.size();
    std::vector<int32_t> pop;
    std::vector<double> lat, lon, fl((size_t)n * n, 1.0), flat((size_t)n * n, 0.0);
    for (const auto& c : cs) { pop.push_back(c.pop); lat.push_back(c.lat); lon.push_back(c.lon); }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) fl[(size_t)i * n + j] = flightMul(cs[i].name, cs[j].name);
    check(epi_gravity_matrix(n, pop.data(), lat.data(), lon.data(), fl.data(), a, b, g, k, capRow, flat.data(), 0));
    Matrix W((size_t)n, std::vector<double>((size_t)n));
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) W[i][j] = flat[(size_t)i * n + j];
    return W;
}

inline const Matrix& W() {
    static const Matrix w = gravityMatrix(C());
    return w;
}
inline Matrix scaleEdge(const Matrix& W, int src, int dst, double factor) {  // ToyConn.scala:51-55
    Matrix A = W;
    A[src][dst] *= factor;
    return A;
}
inline Matrix scaleRow(const Matrix& W, int row, double factor) {  // ToyConn.scala:58-70
    const int n = (int)W.size();
    std::vector<double> in = flatten(W), out(in.size());
    check(epi_scale_row(in.data(), n, row, factor, out.data()));
    Matrix A(n, std::vector<double>(n));
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) A[i][j] = out[(size_t)i * n + j];
    return A;
}
}  // namespace ToyConn

// ---- Validation.scala -----------------------------------------------------------------------------
namespace Validation {
struct Invariants {
    int pop, tMax;
};
struct Metrics {
    int arrival;
    double peakI, finalI;
};
inline void assertState(const State& s, const Invariants& inv) {  // Validation.scala:6-11
    if (!(s.s >= 0 && s.i >= 0 && s.r >= 0 && s.d >= 0 && s.v >= 0)) throw IllegalArgumentException("requirement failed: neg compartment");
    const double total = s.s + s.i + s.r + s.d + s.v;
    if ((int)std::floor(total + 0.5) != inv.pop) throw IllegalArgumentException("requirement failed: mass not conserved");
    if (!(s.t >= 0 && s.t <= inv.tMax)) throw IllegalArgumentException("requirement failed: t out of bounds");
}
inline int arrivalTime(const std::vector<double>& iSeries, double thresh = 10.0) {  // :15-19
    for (size_t k = 0; k < iSeries.size(); ++k)
        if (iSeries[k] >= thresh) return (int)k;
    return (int)iSeries.size();
}
inline double peakI(const std::vector<double>& iSeries) {  // :21-22
    double m = 0.0;
    for (size_t k = 0; k < iSeries.size(); ++k) m = (k == 0 || iSeries[k] > m) ? iSeries[k] : m;
    return m;
}
}  // namespace Validation

// ---- env.scala:32-255 -----------------------------------------------------------------------------
class EpidemicEnv {
public:
    EpidemicEnv(double baseBeta = 0.35, double baseGamma = 0.12, double baseIFR = 0.008, double noise = 0.05, double rewardClip = 5.0,
                double riskFactor = 0.0, std::string countryName = "Unknown", int seed = 42)
        : p_{baseBeta, baseGamma, baseIFR, noise, rewardClip, riskFactor}, name_(std::move(countryName)), seed_(seed),
          profile_(Geo::getProfile(name_)) {}

    State reset(const State& s0) const { return s0; }
    struct EnvParams {
        double beta, gamma, ifr, noise;
    };
    EnvParams params() const { return {p_.base_beta, p_.base_gamma, p_.base_ifr, p_.noise}; }
    const epi_env_params& raw() const { return p_; }
    const Geo::Country& countryProfile() const { return profile_; }
    int seed() const { return seed_; }

    // env.scala:58-140; the Gaussian draw comes from this env's own Random(seed) stream (env.scala:42,91)
    std::tuple<State, double, bool> step(const State& st, Action act) {
        if (pos_ >= gauss_.size()) {
            gauss_.resize(gauss_.empty() ? 64 : gauss_.size() * 2);
            check(epi_java_gaussians(seed_, (int32_t)gauss_.size(), gauss_.data()));
        }
        const double z = gauss_[pos_++];
        const epi_state in = st.c();
        const epi_profile pr = profile_.c();
        const int32_t a = static_cast<int>(act);
        epi_state out;
        double reward = 0;
        uint8_t done = 0;
        check(epi_env_step_batch(&p_, &pr, &in, &a, &z, &out, &reward, &done, 1));
        return {State::from(out), reward, done != 0};
    }

    static std::vector<int> allocateInt(int total, const std::vector<double>& weights) {  // env.scala:241-254
        std::vector<int32_t> out(weights.size(), 0);
        check(epi_alloc_int(total, weights.data(), (int32_t)weights.size(), out.data()));
        return std::vector<int>(out.begin(), out.end());
    }

private:
    epi_env_params p_;
    std::string name_;
    int seed_;
    Geo::Country profile_;
    std::vector<double> gauss_;
    size_t pos_ = 0;
};

// ---- ReplayBuffer.scala / DDQNAgent.scala / MLP.scala: learners held by the engine -----------------------
struct Transition {  // ReplayBuffer.scala:8
    State s;
    int a;
    double r;
    State s2;
    bool done;
};

// java.util.Random (48-bit LCG of the JDK javadoc) for the agent's own epsilon-greedy draws (DDQNAgent.scala:14,28).
class JavaRandom {
public:
    explicit JavaRandom(int64_t seed) : state_((uint64_t)(seed ^ 0x5DEECE66DLL) & kMask) {}
    int32_t next(int bits) {
        state_ = (state_ * 0x5DEECE66DULL + 0xBULL) & kMask;
        return (int32_t)(int64_t)(state_ >> (48 - bits));
    }
    int32_t nextInt(int32_t bound) {
        int32_t r = next(31);
        const int32_t m = bound - 1;
        if ((bound & m) == 0) return (int32_t)(((int64_t)bound * (int64_t)r) >> 31);
        for (int32_t u = r;; u = next(31)) {
            r = u % bound;
            if ((int32_t)((uint32_t)u - (uint32_t)r + (uint32_t)m) >= 0) return r;
        }
    }
    double nextDouble() {
        const int64_t hi = (int64_t)next(26) << 27;
        return (double)(hi + next(27)) * (1.0 / (double)(1LL << 53));
    }

private:
    static constexpr uint64_t kMask = (1ULL << 48) - 1;
    uint64_t state_;
};

// The agents of one world in one engine handle (epi_agents_*): replay buffers, online and target networks.
class Agents {
public:
    explicit Agents(const std::vector<HyperParams>& hps, int maxGradAccum = 4, int device = 0) : hps_(hps) {
        std::vector<epi_learn_hyper> raw;
        for (auto& h : hps)
            raw.push_back(epi_learn_hyper{h.gamma, h.lr, h.clipQ, h.gradClip, h.emaAlpha, h.hidden, h.replayCapacity,
                                          h.targetUpdateEvery, h.batchSize * std::max(1, maxGradAccum)});
        h_ = epi_agents_create((int32_t)raw.size(), raw.data(), device);
        if (!h_) {
            char buf[1024];
            epi_last_error(buf, sizeof buf);
            throw std::runtime_error(buf);
        }
    }
    ~Agents() { epi_agents_destroy(h_); }
    Agents(const Agents&) = delete;
    Agents& operator=(const Agents&) = delete;

    int size() const { return (int)hps_.size(); }
    epi_agents* handle() const { return h_; }
    void observe(const std::vector<int32_t>& ids, const std::vector<Transition>& trs) {  // DDQNAgent.scala:34
        std::vector<epi_state> s, s2;
        std::vector<int32_t> a;
        std::vector<double> r;
        std::vector<uint8_t> d;
        for (auto& t : trs) { s.push_back(t.s.c()); s2.push_back(t.s2.c()); a.push_back(t.a); r.push_back(t.r); d.push_back(t.done ? 1 : 0); }
        check(epi_agents_observe(h_, ids.data(), (int32_t)trs.size(), s.data(), a.data(), r.data(), s2.data(), d.data()));
    }
    std::vector<int32_t> replaySize() {  // :69
        std::vector<int32_t> out(hps_.size());
        check(epi_agents_replay_size(h_, out.data()));
        return out;
    }
    std::vector<int32_t> sample(int agent, int n) {  // ReplayBuffer.scala:21-22, buffer positions
        std::vector<int32_t> out((size_t)std::max(n, 1));
        int32_t taken = 0;
        check(epi_agents_sample(h_, agent, n, out.data(), &taken));
        out.resize((size_t)taken);
        return out;
    }
    // DDQNAgent.learn (:51-65) for the listed agents in one call; returns the losses, fills ema if given
    std::vector<double> learn(const std::vector<int32_t>& ids, const std::vector<int32_t>& batch, const std::vector<int32_t>& gradAccum,
                              const std::vector<int64_t>& steps, std::vector<double>* ema = nullptr) {
        std::vector<double> loss(ids.size()), e(ids.size());
        check(epi_agents_learn(h_, ids.data(), (int32_t)ids.size(), batch.data(), gradAccum.data(), steps.data(), loss.data(), e.data()));
        if (ema) *ema = e;
        return loss;
    }
    std::vector<int32_t> greedy(const std::vector<int32_t>& ids, const std::vector<State>& states, std::vector<double>* q = nullptr) {
        std::vector<epi_state> raw;
        for (auto& s : states) raw.push_back(s.c());
        std::vector<int32_t> g(ids.size());
        if (q) q->assign(ids.size() * 8, 0.0);
        check(epi_agents_q_values(h_, ids.data(), (int32_t)ids.size(), raw.data(), 0, q ? q->data() : nullptr, g.data()));
        return g;
    }
    // [W1, b1, W2, b2, W3, b3] of one agent's online (or target) network
    std::vector<std::vector<double>> weights(int agent, bool target = false) {
        const size_t h = (size_t)hps_[agent].hidden;
        std::vector<std::vector<double>> W{std::vector<double>(h * 7), std::vector<double>(h), std::vector<double>(h * h),
                                           std::vector<double>(h), std::vector<double>(8 * h), std::vector<double>(8)};
        check(epi_agents_get_weights(h_, agent, target ? 1 : 0, W[0].data(), W[1].data(), W[2].data(), W[3].data(), W[4].data(), W[5].data()));
        return W;
    }
    const HyperParams& hp(int agent) const { return hps_[agent]; }

private:
    std::vector<HyperParams> hps_;
    epi_agents* h_ = nullptr;
};

// DDQNAgent.scala:6-70: the epsilon stream on the host, the networks in the shared handle.
class DDQNAgent {
public:
    DDQNAgent(std::shared_ptr<Agents> pool, int index) : pool_(std::move(pool)), k_(index), hp_(pool_->hp(index)) {}
    double epsilonNow() const {  // :16-19,67
        const double frac = std::min(1.0, (double)steps / (double)std::max(1, hp_.epsilonDecaySteps));
        return hp_.epsilonStart + (hp_.epsilonEnd - hp_.epsilonStart) * frac;
    }
    int actGreedy(const State& s) { return pool_->greedy({k_}, {s})[0]; }  // :22-23
    // the random branch of actWithEpsilon (:26-30) without the network: 0..7, or -1 where argmax Q is taken
    int explores(double eps, bool countStep) {
        if (countStep) steps += 1;
        return rnd_.nextDouble() < eps ? rnd_.nextInt(8) : -1;
    }
    int actWithEpsilon(const State& s, double eps, bool countStep = false) {
        const int a = explores(eps, countStep);
        return a >= 0 ? a : actGreedy(s);
    }
    int act(const State& s) { return actWithEpsilon(s, epsilonNow(), true); }  // :32: epsilon is read before the step counts
    void observe(const Transition& tr) { pool_->observe({k_}, {tr}); }          // :34
    double learn(int batch, int gradAccumSteps = 4) {                             // :51-65
        std::vector<double> ema;
        const double loss = pool_->learn({k_}, {batch}, {gradAccumSteps}, {steps}, &ema)[0];
        emaLoss_ = ema[0];
        return loss;
    }
    double emaLossNow() const { return emaLoss_; }
    int replaySize() { return pool_->replaySize()[(size_t)k_]; }
    std::vector<double> qValues(const State& s) {  // :70
        std::vector<double> q;
        pool_->greedy({k_}, {s}, &q);
        return q;
    }
    int64_t steps = 0;
    void setEma(double e) { emaLoss_ = e; }

private:
    std::shared_ptr<Agents> pool_;
    int32_t k_;
    HyperParams hp_;
    JavaRandom rnd_{9};
    double emaLoss_ = 0.0;
};

// ---- WorldToy.scala -------------------------------------------------------------------------------
class WorldToy {
public:
    struct Node {
        int idx;
        std::string name;
        int pop;
        std::shared_ptr<EpidemicEnv> env;
        HyperParams hyperParams;
        State s;
        int lastA = 0;
        int arrival = -1;     // first step index with I >= 10 since the last reset (Validation.arrivalTime); -1 = not yet
        double peakI = 0.0;   // running Validation.peakI
        std::shared_ptr<DDQNAgent> agent;  // WorldToy.scala:50, present once the world learns
        int64_t lastLearnStep = 0;
    };
    using Step = std::vector<std::tuple<std::string, State, double>>;

    WorldToy(const HyperParams& hp, const State& base, int seedIdx = 0, const Matrix& conn = ToyConn::W(),
             std::function<std::shared_ptr<EpidemicEnv>()> makeEnv = [] { return std::make_shared<EpidemicEnv>(); },
             std::function<std::shared_ptr<EpidemicEnv>(const std::string&, int)> makeEnvFor = nullptr, bool validate = false,
             const std::vector<Geo::Country>& countries = ToyConn::C(),
             const std::map<std::string, HyperParams>& countryHyperParams = {}, bool learning = true)
        : hp_(hp), base_(base), learning_(learning) {
        const int n = (int)countries.size();
        std::vector<int32_t> pops, seeds;
        std::vector<epi_env_params> envs;
        std::vector<epi_profile> profs;
        for (int i = 0; i < n; ++i) {
            auto env = makeEnvFor ? makeEnvFor(countries[i].name, countries[i].pop) : makeEnv();
            pops.push_back(countries[i].pop);
            seeds.push_back(env->seed());
            envs.push_back(env->raw());
            profs.push_back(env->countryProfile().c());
            const auto chp = countryHyperParams.find(countries[i].name);
            nodes.push_back(Node{i, countries[i].name, countries[i].pop, env, chp == countryHyperParams.end() ? hp : chp->second, State{}});
        }
        const std::vector<double> flat = flatten(conn);
        epi_world_desc d{};
        d.n = n; d.batch = 1; d.pop = pops.data(); d.conn = flat.data();
        d.seed_idx = seedIdx; d.validate = validate ? 1 : 0; d.base = base.c();
        d.env = envs.data(); d.prof = profs.data(); d.env_seeds = seeds.data(); d.hp = hp.c(); d.device = 0;
        h_ = epi_world_create(&d);
        if (!h_) {
            char buf[1024];
            epi_last_error(buf, sizeof buf);
            if (std::string(buf).find("requirement failed") != std::string::npos) throw IllegalArgumentException(buf);
            throw std::runtime_error(buf);
        }
        syncNodes();
    }
    ~WorldToy() { epi_world_destroy(h_); }
    WorldToy(const WorldToy&) = delete;
    WorldToy& operator=(const WorldToy&) = delete;

    std::vector<Node> nodes;

    std::vector<State> snapshotStates() {  // WorldToy.scala:75
        std::vector<epi_state> raw(nodes.size());
        check(epi_world_get_states(h_, raw.data()));
        std::vector<State> out;
        for (auto& r : raw) out.push_back(State::from(r));
        return out;
    }
    void restoreStates(const std::vector<State>& ss) {  // :76
        std::vector<epi_state> raw;
        for (auto& s : ss) raw.push_back(s.c());
        check(epi_world_set_states(h_, raw.data()));
        syncNodes();
    }
    void resetToInitial() {  // :77
        check(epi_world_reset(h_));
        syncNodes();
        stepsDone_ = 0;
        for (auto& nd : nodes) { nd.arrival = -1; nd.peakI = 0.0; }
    }

    // losses of the last stepOne: what agent.learn returned, or -1 where the node did not learn
    std::vector<double> lastLosses;
    std::shared_ptr<Agents> agents() { return pool_; }

    // WorldToy.stepOne (:83-170).  With `learning` every node acts through its own DDQNAgent, observes its transition
    // and learns on its own schedule (:147-167); without it the agents stay untrained and only their
    // epsilon-greedy streams are followed.
    Step stepOne(int tMax) {
        const int n = (int)nodes.size();
        if (learning_) return stepOneLearning(tMax);
        if (streams_.empty()) streams_.resize(n);
        std::vector<int32_t> acts(n);
        bool needGreedy = false;
        for (int c = 0; c < n; ++c) {
            auto& st = streams_[c];
            if (st.pos >= st.buf.size()) {
                st.buf.resize(st.buf.empty() ? 256 : st.buf.size() * 2);
                const epi_hyper h = hp_.c();
                check(epi_agent_actions(&h, (int32_t)st.buf.size(), st.buf.data()));
            }
            acts[c] = st.buf[st.pos++];
            needGreedy = needGreedy || acts[c] < 0;
        }
        if (needGreedy) {
            ensureMlp();
            std::vector<epi_state> raw;
            for (auto& nd : nodes) raw.push_back(nd.s.c());
            std::vector<int32_t> g(n);
            check(epi_mlp_greedy_batch(raw.data(), n, hp_.hidden, W1_.data(), nullptr, W2_.data(), nullptr, W3_.data(), nullptr,
                                       hp_.clipQ, nullptr, g.data()));
            for (int c = 0; c < n; ++c)
                if (acts[c] < 0) acts[c] = g[c];
        }
        std::vector<epi_state> out(n);
        std::vector<double> rew(n);
        check(epi_world_step(h_, acts.data(), nullptr, tMax, out.data(), rew.data()));
        return collect(out, rew, acts);
    }

    Step stepOneGreedy(int tMax) {  // :172-235
        const int n = (int)nodes.size();
        if (learning_ && pool_) {  // every node asks its own, possibly trained, network (:177)
            std::vector<int32_t> ids(n);
            std::vector<State> ss;
            for (int c = 0; c < n; ++c) { ids[c] = c; ss.push_back(nodes[c].s); }
            std::vector<int32_t> acts = pool_->greedy(ids, ss);
            std::vector<epi_state> out(n);
            std::vector<double> rew(n);
            check(epi_world_step(h_, acts.data(), nullptr, tMax, out.data(), rew.data()));
            return collect(out, rew, acts);
        }
        ensureMlp();
        std::vector<epi_state> out(n);
        std::vector<double> rew(n);
        std::vector<int32_t> acts(n);
        check(epi_world_step_greedy(h_, nullptr, tMax, out.data(), rew.data(), acts.data()));
        return collect(out, rew, acts);
    }

    // The periodic greedy evaluation of the trainer (MutliTrainer.scala:128-161): snapshot, reset, maxEvalSteps
    // stepOneGreedy calls kept on the device (epi_world_rollout_greedy), the four metrics per country, restore.
    std::map<std::string, double> evaluateGreedy(int maxEvalSteps = 800) {
        const int n = (int)nodes.size();
        const auto saved = snapshotStates();
        resetToInitial();
        if (!(learning_ && pool_)) ensureMlp();
        std::vector<double> series((size_t)maxEvalSteps * n);
        check(epi_world_rollout_greedy(h_, learning_ && pool_ ? pool_->handle() : nullptr, maxEvalSteps, maxEvalSteps,
                                       series.data(), nullptr, nullptr, nullptr));
        std::map<std::string, double> m;
        for (int c = 0; c < n; ++c) {
            std::vector<double> is((size_t)maxEvalSteps), isN((size_t)maxEvalSteps);
            for (int k = 0; k < maxEvalSteps; ++k) {
                is[k] = series[(size_t)k * n + c];
                isN[k] = is[k] / std::max(1.0, (double)nodes[c].pop);
            }
            m["eval_arrival_" + nodes[c].name] = (double)Validation::arrivalTime(is, 10.0);
            m["eval_peakI_" + nodes[c].name] = Validation::peakI(is);
            m["eval/arrivalN_" + nodes[c].name] = (double)Validation::arrivalTime(isN, 1e-4);
            m["eval/peakI_norm_" + nodes[c].name] = Validation::peakI(isN);
        }
        restoreStates(saved);
        return m;
    }

    std::pair<std::vector<int>, std::vector<int>> lastFlows() {
        std::vector<int32_t> s(nodes.size()), i(nodes.size());
        check(epi_world_last_flows(h_, s.data(), i.data()));
        return {std::vector<int>(s.begin(), s.end()), std::vector<int>(i.begin(), i.end())};
    }

private:
    struct Stream {
        std::vector<int32_t> buf;
        size_t pos = 0;
    };
    static int learningIntensity(const std::string& name) { return name == "India" ? 2 : 1; }  // WorldToy.scala:157-162
    Step stepOneLearning(int tMax) {
        const int n = (int)nodes.size();
        if (!pool_) {
            std::vector<HyperParams> hps;
            for (auto& nd : nodes) hps.push_back(nd.hyperParams);
            pool_ = std::make_shared<Agents>(hps, 4);
            for (auto& nd : nodes) nd.agent = std::make_shared<DDQNAgent>(pool_, nd.idx);
        }
        std::vector<State> before;
        std::vector<int32_t> acts(n), gids;
        std::vector<State> gstates;
        for (int c = 0; c < n; ++c) {
            before.push_back(nodes[c].s);
            acts[c] = nodes[c].agent->explores(nodes[c].agent->epsilonNow(), true);  // :88
            if (acts[c] < 0) { gids.push_back(c); gstates.push_back(nodes[c].s); }
        }
        if (!gids.empty()) {
            const std::vector<int32_t> g = pool_->greedy(gids, gstates);
            for (size_t k = 0; k < gids.size(); ++k) acts[gids[k]] = g[k];
        }
        std::vector<epi_state> out(n);
        std::vector<double> rew(n);
        check(epi_world_step(h_, acts.data(), nullptr, tMax, out.data(), rew.data()));
        const auto flows = lastFlows();
        std::vector<int32_t> ids(n);
        std::vector<Transition> trs;
        for (int c = 0; c < n; ++c) {
            // the post-env.step state the transition holds (:90,99): the returned state without the mobility flows
            // of :132-136 and with env.step's own clock
            State s2 = State::from(out[c]);
            s2.s -= (double)flows.first[c];
            s2.i -= (double)flows.second[c];
            s2.t = before[c].t + 1;
            const double nn = std::max(1.0, s2.s + s2.i + s2.r + s2.d);
            const bool done = s2.t >= s2.tMax || s2.i / nn < 1e-5 || s2.d / nn > 0.02;  // env.scala:137
            ids[c] = c;
            trs.push_back(Transition{before[c], acts[c], rew[c], s2, done});
        }
        pool_->observe(ids, trs);
        Step res = collect(out, rew, acts);
        ++gStep_;
        const std::vector<int32_t> sizes = pool_->replaySize();
        std::vector<int32_t> due, batch, accum;
        std::vector<int64_t> steps;
        for (auto& nd : nodes)
            if (gStep_ - nd.lastLearnStep >= nd.hyperParams.learnEvery && sizes[nd.idx] >= nd.hyperParams.batchSize * 4) {
                due.push_back(nd.idx); batch.push_back(nd.hyperParams.batchSize); accum.push_back(learningIntensity(nd.name));
                steps.push_back(nd.agent->steps);
            }
        lastLosses.assign((size_t)n, -1.0);
        if (!due.empty()) {
            std::vector<double> ema;
            const std::vector<double> loss = pool_->learn(due, batch, accum, steps, &ema);
            for (size_t k = 0; k < due.size(); ++k) {
                nodes[due[k]].lastLearnStep = gStep_;
                nodes[due[k]].agent->setEma(ema[k]);
                lastLosses[due[k]] = loss[k];
            }
        }
        return res;
    }
    void syncNodes() {
        auto ss = snapshotStates();
        for (size_t c = 0; c < nodes.size(); ++c) nodes[c].s = ss[c];
    }
    void ensureMlp() {
        if (!W1_.empty()) return;
        W1_.resize((size_t)hp_.hidden * 7);
        W2_.resize((size_t)hp_.hidden * hp_.hidden);
        W3_.resize((size_t)8 * hp_.hidden);
        check(epi_mlp_init(7, hp_.hidden, 8, W1_.data(), W2_.data(), W3_.data()));
        check(epi_world_set_mlp(h_, hp_.hidden, W1_.data(), nullptr, W2_.data(), nullptr, W3_.data(), nullptr, hp_.clipQ));
    }
    Step collect(const std::vector<epi_state>& out, const std::vector<double>& rew, const std::vector<int32_t>& acts) {
        Step res;
        for (size_t c = 0; c < nodes.size(); ++c) {
            nodes[c].s = State::from(out[c]);
            nodes[c].lastA = acts[c];
            if (nodes[c].arrival < 0 && nodes[c].s.i >= 10.0) nodes[c].arrival = stepsDone_;
            nodes[c].peakI = std::max(nodes[c].peakI, nodes[c].s.i);
            res.emplace_back(nodes[c].name, nodes[c].s, rew[c]);
        }
        ++stepsDone_;
        return res;
    }
    HyperParams hp_;
    State base_;
    epi_world* h_ = nullptr;
    int stepsDone_ = 0;
    bool learning_ = true;
    int64_t gStep_ = 0;
    std::shared_ptr<Agents> pool_;
    std::vector<Stream> streams_;
    std::vector<double> W1_, W2_, W3_;
};

// ---- Calibrate.scala ------------------------------------------------------------------------------
namespace Calibrate {
struct Targets {  // Calibrate.scala:8-12
    std::map<std::string, double> tArrive, peakI;
    double threshI = 10.0;
};
using Row = std::tuple<double, double, double>;
using Surface = std::pair<std::vector<Row>, std::pair<double, std::pair<double, double>>>;

// Calibrate.gridSurface (:84-116): rows (beta, ts, loss) in loop order and (bestLoss, (beta, ts)).
inline Surface gridSurface(const HyperParams& hp, const TrainConfig& cfg, const State& base, int seedIdx, const Targets& targets,
                           const std::vector<double>& betaGrid, const std::vector<double>& travelScales, bool /*validate*/ = true,
                           bool heterogeneous = true, const std::vector<Geo::Country>& countries = ToyConn::C(),
                           const Matrix& conn = ToyConn::W(), const std::vector<int32_t>& devices = {}) {
    const int n = (int)countries.size();
    std::pair<double, double> bestPair{betaGrid.empty() ? 0.25 : betaGrid[0], travelScales.empty() ? 1.0 : travelScales[0]};
    if (betaGrid.empty() || travelScales.empty()) return {{}, {1.7976931348623157e308, bestPair}};
    std::vector<int32_t> pops;
    std::vector<epi_profile> profs;
    std::vector<double> ta(n, 0.0), tp(n, 0.0);
    std::vector<uint8_t> ha(n, 0), hk(n, 0);
    for (int c = 0; c < n; ++c) {
        pops.push_back(countries[c].pop);
        profs.push_back(Geo::getProfile(countries[c].name).c());
        auto a = targets.tArrive.find(countries[c].name);
        if (a != targets.tArrive.end()) { ta[c] = a->second; ha[c] = 1; }
        auto p = targets.peakI.find(countries[c].name);
        if (p != targets.peakI.end()) { tp[c] = p->second; hk[c] = 1; }
    }
    const std::vector<double> flat = flatten(conn);
    epi_grid_desc d{};
    d.n = n; d.pop = pops.data(); d.conn = flat.data(); d.seed_idx = seedIdx; d.heterogeneous = heterogeneous ? 1 : 0;
    d.base = base.c(); d.prof = profs.data(); d.hp = hp.c(); d.steps = cfg.stepsPerEpoch;
    d.n_beta = (int)betaGrid.size(); d.betas = betaGrid.data(); d.n_travel = (int)travelScales.size(); d.travel = travelScales.data();
    d.t_arrive = ta.data(); d.has_arrive = ha.data(); d.t_peak = tp.data(); d.has_peak = hk.data(); d.device = 0;
    // rollouts long enough for their agents to learn (WorldToy.scala:147-167): the learn step's hyper-parameters and
    // the per-country learningIntensity of :157-162
    const epi_learn_hyper lh = hp.learn(hp.batchSize * 2);
    std::vector<int32_t> intensity;
    for (int c = 0; c < n; ++c) intensity.push_back(countries[c].name == "India" ? 2 : 1);
    d.learn = &lh; d.learn_intensity = intensity.data();
    std::vector<double> loss(betaGrid.size() * travelScales.size());
    int32_t bestIdx = 0;
    double bestLoss = 0;
    if (devices.empty())
        check(epi_calibrate_grid(&d, loss.data(), nullptr, nullptr, &bestIdx, &bestLoss));
    else   // several GPUs of the node: one engine per listed device, loss pieces all-gathered with RCCL inside the library
        check(epi_calibrate_grid_multi(&d, devices.data(), (int32_t)devices.size(), loss.data(), nullptr, nullptr, &bestIdx, &bestLoss));
    std::vector<Row> rows;
    for (size_t b = 0; b < betaGrid.size(); ++b)
        for (size_t k = 0; k < travelScales.size(); ++k) rows.emplace_back(betaGrid[b], travelScales[k], loss[b * travelScales.size() + k]);
    if (bestLoss < 1.7976931348623157e308) bestPair = {std::get<0>(rows[bestIdx]), std::get<1>(rows[bestIdx])};
    return {rows, {bestLoss, bestPair}};
}

// Shortest decimal text that parses back to the same double.
inline std::string doubleText(double x) {
    char buf[40];
    for (int prec = 1; prec <= 17; ++prec) {
        std::snprintf(buf, sizeof buf, "%.*g", prec, x);
        if (std::strtod(buf, nullptr) == x) break;
    }
    return buf;
}

// The CSV the reference's heatmap script reads (columns beta, travel, loss;
// src/main/scala/resources/plot_histogram.py:12-13).  The reference tree holds no writer for it: the number
// format is this engine's choice, the shortest decimal that parses back to the same double.
inline void writeSurfaceCsv(const std::string& path, const std::vector<Row>& rows) {
    std::FILE* f = std::fopen(path.c_str(), "w");
    if (!f) throw std::runtime_error("cannot write " + path);
    std::fputs("beta,travel,loss\n", f);
    for (const Row& r : rows)
        std::fprintf(f, "%s,%s,%s\n", doubleText(std::get<0>(r)).c_str(), doubleText(std::get<1>(r)).c_str(),
                     doubleText(std::get<2>(r)).c_str());
    std::fclose(f);
}

inline std::vector<Row> readSurfaceCsv(const std::string& path) {
    std::FILE* f = std::fopen(path.c_str(), "r");
    if (!f) throw std::runtime_error("cannot read " + path);
    char line[256];
    std::vector<Row> rows;
    if (!std::fgets(line, sizeof line, f) || std::string(line).rfind("beta,travel,loss", 0) != 0) {
        std::fclose(f);
        throw std::runtime_error(path + ": not a calibration surface");
    }
    double b, ts, l;
    while (std::fgets(line, sizeof line, f))
        if (std::sscanf(line, "%lf,%lf,%lf", &b, &ts, &l) == 3) rows.emplace_back(b, ts, l);
    std::fclose(f);
    return rows;
}
}  // namespace Calibrate

// ---- snapshot stream of the trainer ----------------------------------------------------------------
// README.md:14: records (country, t, i, arrival, peakI, loss) under stream_in/snapshots; call sites
// MutliTrainer.scala:113,125.  The reference tree does not contain the writer itself, so the file layout is
// this engine's choice: one JSON-lines file per call, one record per country, written under a temporary name
// and renamed so that a streaming reader never sees half a file.  The directory must exist.
namespace Snapshots {
inline std::string writeEpochSnapshot(const WorldToy& world, int step, double loss, const std::string& outDir = "stream_in/snapshots") {
    char name[64];
    std::snprintf(name, sizeof name, "/snapshot-%010d.json", step);
    const std::string fin = outDir + name, tmp = fin + ".tmp";
    std::FILE* f = std::fopen(tmp.c_str(), "w");
    if (!f) throw std::runtime_error("cannot write " + tmp);
    for (const auto& nd : world.nodes)
        std::fprintf(f, "{\"country\": \"%s\", \"t\": %d, \"i\": %s, \"arrival\": %d, \"peakI\": %s, \"loss\": %s}\n", nd.name.c_str(),
                     step, Calibrate::doubleText(nd.s.i).c_str(), nd.arrival, Calibrate::doubleText(nd.peakI).c_str(),
                     Calibrate::doubleText(loss).c_str());
    std::fclose(f);
    if (std::rename(tmp.c_str(), fin.c_str()) != 0) throw std::runtime_error("cannot rename " + tmp);
    return fin;
}
}  // namespace Snapshots

}  // namespace epidemic
