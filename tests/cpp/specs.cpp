// The reference's own ScalaTest specs for this path (src/test/scala/epidemic/*.scala), restated against
// the C++ mirror include/epidemic.hpp.  Needs a GPU: every call below runs in libepi.so.
// Built by __graft_entry__.build(); run by tests/test_gpu_cpp_mirror.py.  Exit code 0 = all specs pass.
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "epidemic.hpp"

using namespace epidemic;

static int failures = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) {                                                      \
            std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                     \
        }                                                                   \
    } while (0)

// ValidationSpec.scala:7-22
static void validationSpec() {
    State s{9, 1, 0, 0, 0, 10, 0, 10, 1};
    Validation::assertState(s, Validation::Invariants{10, 10});
    EXPECT(Validation::arrivalTime({1.0, 2.0, 3.0, 4.0, 5.0}, 10.0) == 5);
    EXPECT(Validation::peakI({}) == 0.0);
    EXPECT(std::fabs(Validation::peakI({0.1, 0.5, 0.3}) - 0.5) <= 1e-12);
    bool threw = false;
    try {
        Validation::assertState(State{9, 2, 0, 0, 0, 10, 0, 10, 1}, Validation::Invariants{10, 10});
    } catch (const IllegalArgumentException&) {
        threw = true;
    }
    EXPECT(threw);
}

// MobilitySpec.scala:7-14 and :24-63
static void mobilitySpec() {
    const Matrix& W = ToyConn::W();
    const int n = (int)W.size();
    for (int i = 0; i < n; ++i) {
        EXPECT(W[i][i] == 0.0);
        EXPECT(std::accumulate(W[i].begin(), W[i].end(), 0.0) <= 0.12 + 1e-12);
    }
    // the same properties for the matrix built on the device, which agrees with the host one to the last places
    const Matrix Wd = ToyConn::gravityMatrixDevice(ToyConn::C());
    for (int i = 0; i < n; ++i) {
        EXPECT(Wd[i][i] == 0.0);
        EXPECT(std::accumulate(Wd[i].begin(), Wd[i].end(), 0.0) <= 0.12 + 1e-12);
        for (int j = 0; j < n; ++j) EXPECT(std::fabs(Wd[i][j] - W[i][j]) <= 1e-12 * W[i][j]);
    }
    const double S[6] = {1000000, 120000, 1000000, 80000, 60000, 300000};
    const double I[6] = {1000, 200, 900, 100, 80, 400};
    std::vector<long long> movedS(n, 0), movedI(n, 0);
    for (int i = 0; i < n; ++i) {
        std::vector<double> wRow(n);
        double sumW = 0.0;
        for (int j = 0; j < n; ++j) {
            wRow[j] = (j == i) ? 0.0 : W[i][j];
            sumW = (j == 0) ? wRow[0] : sumW + wRow[j];
        }
        sumW = std::fmax(0.0, sumW);
        const int Si = (int)S[i], Ii = (int)I[i];
        const int exportS = (int)std::min<long long>(Si, (long long)std::floor(Si * sumW + 1e-9));
        const int exportI = (int)std::min<long long>(Ii, (long long)std::floor(Ii * sumW + 1e-9));
        const std::vector<int> allocS = EpidemicEnv::allocateInt(exportS, wRow);
        const std::vector<int> allocI = EpidemicEnv::allocateInt(exportI, wRow);
        for (int j = 0; j < n; ++j)
            if (j != i) { movedS[j] += allocS[j]; movedI[j] += allocI[j]; }
        movedS[i] -= exportS;
        movedI[i] -= exportI;
    }
    long long net = 0;
    for (int j = 0; j < n; ++j) net += movedS[j] + movedI[j];
    EXPECT(net == 0);
    // allocateInt known answers (env.scala:241-254)
    EXPECT((EpidemicEnv::allocateInt(10, {.5, .3, .2}) == std::vector<int>{5, 3, 2}));
    EXPECT((EpidemicEnv::allocateInt(10, {1, 1, 1}) == std::vector<int>{4, 3, 3}));
    EXPECT((EpidemicEnv::allocateInt(7, {0, 2, 1, 1}) == std::vector<int>{0, 3, 2, 2}));
    EXPECT((EpidemicEnv::allocateInt(1, {.25, .25, .25, .25}) == std::vector<int>{1, 0, 0, 0}));
    EXPECT((EpidemicEnv::allocateInt(5, {0, 0, 0}) == std::vector<int>{0, 0, 0}));
}

// CalibrationSpec.scala:6-16
static void calibrationSpec() {
    HyperParams hp;
    TrainConfig cfg;
    cfg.epochs = 1;
    cfg.stepsPerEpoch = 60;
    State base{1000000, 0, 0, 0, 0, 10000, 0, 60, 3};
    Calibrate::Targets targets;
    auto res = Calibrate::gridSurface(hp, cfg, base, 0, targets, {0.24, 0.26}, {0.8, 1.0}, false, false);
    EXPECT(!res.first.empty());
    EXPECT(std::isfinite(res.second.first));
    EXPECT(res.second.first == 0.0);  // empty targets: every difference is 0
    EXPECT(res.second.second.first == 0.24 && res.second.second.second == 0.8);
}

// EvalDeterminismSpec.scala:7-45
static void evalDeterminismSpec() {
    auto makeWorld = [] {
        HyperParams hp0;
        hp0.learnEvery = 2147483647;
        State base{0, 0, 0, 0, 0, 10000, 0, 120, 42};
        return std::make_unique<WorldToy>(
            hp0, base, 0, ToyConn::W(), [hp0] { return std::make_shared<EpidemicEnv>(0.26, 0.12, 0.008, 0.02, hp0.rewardClip); }, nullptr,
            false);
    };
    auto runEvalSeries = [](WorldToy& w, int steps) {
        std::vector<double> out;
        for (int k = 0; k < steps; ++k) {
            auto batch = w.stepOneGreedy(steps);
            for (auto& row : batch)
                if (std::get<0>(row) == "USA") out.push_back(std::get<1>(row).i);
        }
        return out;
    };
    auto w1 = makeWorld(), w2 = makeWorld();
    const auto a = runEvalSeries(*w1, 100), b = runEvalSeries(*w2, 100);
    EXPECT(a.size() == 100 && a == b);
    // snapshot / reset / restore round trip (WorldToy.scala:75-77, MutliTrainer.scala:130-160)
    const auto snap = w1->snapshotStates();
    w1->resetToInitial();
    EXPECT(w1->nodes[0].s.i == 330000.0 && w1->nodes[1].s.i == 0.0);
    w1->restoreStates(snap);
    EXPECT(w1->snapshotStates() == snap);
    // the trainer's evaluation on the device equals the step-by-step one and leaves the world as it was
    {
        auto w4 = makeWorld();
        const auto held = w4->snapshotStates();
        const auto m = w4->evaluateGreedy(100);
        EXPECT(w4->snapshotStates() == held);
        auto w5 = makeWorld();
        std::vector<double> usa = runEvalSeries(*w5, 100);
        EXPECT(m.at("eval_peakI_USA") == Validation::peakI(usa));
        EXPECT(m.at("eval_arrival_USA") == (double)Validation::arrivalTime(usa, 10.0));
    }
    // stepOne conserves world mass (validate = true would throw otherwise)
    HyperParams hp;
    WorldToy w3(hp, State{0, 0, 0, 0, 0, 10000, 0, 120, 42}, 0, ToyConn::W(), [] { return std::make_shared<EpidemicEnv>(); }, nullptr, true);
    double before = 0, after = 0;
    for (auto& nd : w3.nodes) before += nd.s.s + nd.s.i + nd.s.r + nd.s.d + nd.s.v;
    for (int k = 0; k < 30; ++k) w3.stepOne(120);
    for (auto& nd : w3.nodes) after += nd.s.s + nd.s.i + nd.s.r + nd.s.d + nd.s.v;
    EXPECT(before == after);
}

// CalibrationSpec's sweep cut over several engines (the multi-GPU entry point, here all on device 0)
static void calibrationMulti() {
    HyperParams hp;
    TrainConfig cfg;
    cfg.epochs = 1; cfg.stepsPerEpoch = 60;
    State base{1000000, 0, 0, 0, 0, 10000, 0, 60, 3};
    Calibrate::Targets tg;
    tg.tArrive["Japan"] = 20.0; tg.peakI["USA"] = 5e4;
    const auto one = Calibrate::gridSurface(hp, cfg, base, 0, tg, {0.24, 0.26, 0.3}, {0.8, 1.0});
    const auto three = Calibrate::gridSurface(hp, cfg, base, 0, tg, {0.24, 0.26, 0.3}, {0.8, 1.0}, true, true, ToyConn::C(), ToyConn::W(), {0, 0, 0});
    EXPECT(one == three);
}

// HeterogeneitySpec.scala:18-29
static void heterogeneitySpec() {
    auto makeEnvFor = [](const Geo::Country& c) { return EpidemicEnv(0.26 * c.betaMul, 0.12, 0.008 * c.ifrMul, c.noise, 5.0); };
    Geo::Country cn, jp;
    for (auto& c : ToyConn::C()) {
        if (c.name == "China") cn = c;
        if (c.name == "Japan") jp = c;
    }
    State s0{1000000, 1000, 0, 0, 0, 10000, 0, 10, 1};
    auto envCn = makeEnvFor(cn), envJp = makeEnvFor(jp);
    const State sCn1 = std::get<0>(envCn.step(s0, fromId(0)));
    const State sJp1 = std::get<0>(envJp.step(s0, fromId(0)));
    EXPECT(sCn1.i != sJp1.i);
    EXPECT(sCn1.s + sCn1.i + sCn1.r + sCn1.d + sCn1.v == 1001000.0);
}

// Wire formats either side of the path: the calibration-surface CSV the reference's heatmap script reads
// (plot_histogram.py:12-13) and the snapshot records of README.md:14.  No reference spec exists for them.
static void wireFormats() {
    HyperParams hp;
    TrainConfig cfg;
    cfg.epochs = 1;
    cfg.stepsPerEpoch = 40;
    State base{1000000, 0, 0, 0, 0, 10000, 0, 40, 3};
    Calibrate::Targets targets;
    targets.tArrive = {{"Japan", 12.0}, {"India", 9.0}};
    targets.peakI = {{"USA", 400000.0}};
    auto res = Calibrate::gridSurface(hp, cfg, base, 0, targets, {0.24, 0.26, 0.3}, {0.8, 1.0});
    char dir[] = "/tmp/epi_specs_XXXXXX";
    EXPECT(mkdtemp(dir) != nullptr);
    const std::string csv = std::string(dir) + "/calib_surface.csv";
    Calibrate::writeSurfaceCsv(csv, res.first);
    EXPECT(Calibrate::readSurfaceCsv(csv) == res.first);  // exact round trip of every double
    std::FILE* f = std::fopen(csv.c_str(), "r");
    char head[64] = {0};
    EXPECT(f && std::fgets(head, sizeof head, f) && std::string(head) == "beta,travel,loss\n");
    if (f) std::fclose(f);
    WorldToy w(hp, State{0, 0, 0, 0, 0, 10000, 0, 120, 42}, 0, ToyConn::W(), [] { return std::make_shared<EpidemicEnv>(); }, nullptr, true);
    std::vector<double> usa;
    for (int k = 0; k < 25; ++k) {
        auto step = w.stepOne(120);
        usa.push_back(std::get<1>(step[0]).i);
    }
    EXPECT(w.nodes[0].arrival == Validation::arrivalTime(usa) && w.nodes[0].peakI == Validation::peakI(usa));
    const std::string snap = Snapshots::writeEpochSnapshot(w, 25, 0.125, dir);
    f = std::fopen(snap.c_str(), "r");
    EXPECT(f != nullptr);
    int lines = 0;
    char line[512];
    while (f && std::fgets(line, sizeof line, f)) {
        ++lines;
        EXPECT(std::string(line).find("\"country\": \"") != std::string::npos && std::string(line).find("\"loss\": 0.125}") != std::string::npos);
    }
    if (f) std::fclose(f);
    EXPECT(lines == 6);
    w.resetToInitial();
    EXPECT(w.nodes[0].arrival == -1 && w.nodes[0].peakI == 0.0);
}

// ReplayBufferSpec.scala:13-35 against the engine's replay buffers, then a world whose agents learn.
static void replayBufferSpec() {
    auto mkState = [](int i) { return State{1000.0 - i, (double)i, 0.0, 0.0, 0.0, 100.0, i, 999, 1}; };
    auto tr = [&](int i) { return Transition{mkState(i), i % 3, (double)i, mkState(i + 1), false}; };
    HyperParams hp;
    hp.hidden = 16; hp.replayCapacity = 16; hp.batchSize = 4;
    Agents ab({hp, hp}, 3);
    for (int i = 0; i < 16; ++i) ab.observe({0, 1}, {tr(i), tr(i)});
    const auto sa = ab.sample(0, 12), sb = ab.sample(1, 12);
    EXPECT(sa.size() == 12 && sa == sb);  // same Random(123), same contents
    HyperParams small = hp;
    small.replayCapacity = 4; small.batchSize = 1;
    Agents rb({small}, 4);
    for (int i = 0; i < 8; ++i) rb.observe({0}, {tr(i)});
    EXPECT(rb.replaySize()[0] == 4);
    auto got = rb.sample(0, 4);
    std::sort(got.begin(), got.end());
    EXPECT((got == std::vector<int32_t>{0, 1, 2, 3}));  // the four ring slots, holding transitions 4..7

    // two identical learning worlds stay identical, learn on schedule, and their networks move
    auto makeWorld = [] {
        HyperParams h;
        h.hidden = 32; h.batchSize = 4; h.learnEvery = 3; h.epsilonDecaySteps = 40; h.targetUpdateEvery = 7; h.replayCapacity = 30;
        std::map<std::string, HyperParams> chp;
        HyperParams india = h;
        india.batchSize = 3; india.learnEvery = 2; india.lr = 5e-3;
        chp["India"] = india;
        return std::make_unique<WorldToy>(h, State{1e6, 0, 0, 0, 0, 10000, 0, 400, 1}, 0, ToyConn::W(),
                                          [] { return std::make_shared<EpidemicEnv>(); }, nullptr, true, ToyConn::C(), chp, true);
    };
    auto w1 = makeWorld(), w2 = makeWorld();
    int learned = 0, firstLearn = -1;
    for (int k = 0; k < 60; ++k) {
        const auto a = w1->stepOne(400), b = w2->stepOne(400);
        EXPECT(a == b && w1->lastLosses == w2->lastLosses);
        for (double l : w1->lastLosses)
            if (l >= 0) { ++learned; if (firstLearn < 0) firstLearn = k; }
    }
    // India (batch 3) holds 12 transitions after 12 steps and learns every 2nd step from then on; the others (batch 4)
    // need 16 transitions
    EXPECT(firstLearn == 11 && learned > 60);
    const auto fresh = Agents({w1->nodes[0].hyperParams}, 4).weights(0);
    EXPECT(w1->agents()->weights(0) == w2->agents()->weights(0) && w1->agents()->weights(0) != fresh);
    EXPECT(w1->nodes[0].agent->replaySize() == 30 && w1->nodes[0].agent->steps == 60);
}

int main() {
    if (epi_device_count() < 1) {
        std::printf("no HIP device: the specs need the GPU engine\n");
        return 2;
    }
    struct {
        const char* name;
        void (*fn)();
    } specs[] = {{"ValidationSpec", validationSpec}, {"MobilitySpec", mobilitySpec}, {"CalibrationSpec", calibrationSpec},
                 {"EvalDeterminismSpec", evalDeterminismSpec}, {"HeterogeneitySpec", heterogeneitySpec}, {"CalibrationMulti", calibrationMulti}, {"WireFormats", wireFormats}, {"ReplayBufferSpec", replayBufferSpec}};
    for (auto& s : specs) {
        const int before = failures;
        try {
            s.fn();
        } catch (const std::exception& e) {
            std::printf("  EXCEPTION in %s: %s\n", s.name, e.what());
            ++failures;
        }
        std::printf("%s %s\n", failures == before ? "PASS" : "FAIL", s.name);
    }
    return failures == 0 ? 0 : 1;
}
