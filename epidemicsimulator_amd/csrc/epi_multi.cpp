// Calibration sweep over several GPUs of one node from ONE process (include/epi.h, epi_grid_multi_*).
//
// Calibrate.gridSurface's grid loop (Calibrate.scala:100-113) runs independent rollouts, so the flat beta-major
// grid is cut into interleaved shares -- share k runs grid points k, k + n_shares, ... -- one engine (epi_grid) per
// share, each on the device the caller names for it.  The only exchange is one RCCL all-gather of the fp64 loss
// pieces over xGMI, after which every device holds the whole surface; the host reads it from the first device and
// runs the reference's strict-'<' first-minimum scan (epi_grid_argmin).
//
// Ranks of the communicator are the DISTINCT devices.  A device that carries several shares sends all of its
// pieces in one buffer, padded to the largest number of shares any device carries, so one ncclAllGather with equal
// counts serves every placement (including several engines on one GPU).
//
// librccl is opened on first use (dlopen), not linked: it is a 570 MB library that the single-GPU entry points
// never need, and a process that already holds an RCCL (torch's) must keep using that one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/epi.h"
#include "epi_host.hpp"

#define EPI_API extern "C" __attribute__((visibility("default")))

namespace {

using namespace epi;

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.so) break;
        }
        if (!r.so) {
            const char* e = dlerror();
            r.why = e ? e : "librccl.so.1 not found";
            return;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(r.so, n);
            if (!p && r.why.empty()) r.why = std::string("librccl lacks ") + n;
            return p;
        };
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &r;
}

#define NCCL_TRY(expr)                                                                                      \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != ncclSuccess) return fail(EPI_ERR_DEVICE, "%s: %s", #expr, rccl()->GetErrorString(r_));   \
    } while (0)

struct Share {
    int device = 0, rank = 0, slot = 0;  // communicator rank of the device, position among that device's shares
    int count = 0;                       // grid points k, k + n_shares, ... below G
    epi_grid* grid = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    int32_t* arrival = nullptr;          // device [count][n], allocated on first use
    double* peak = nullptr;
};

struct RankBuf {
    int device = 0;
    double* send = nullptr;   // [slots][per]: the loss pieces of this device's shares
    double* full = nullptr;   // [ranks][slots][per]: every piece, after the all-gather
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr;   // around the all-gather on `stream` while timing is on
};

}  // namespace

struct epi_grid_multi {
    int n = 0, G = 0, n_shares = 0, per = 0, slots = 0;
    std::vector<Share> shares;
    std::vector<RankBuf> ranks;
    std::vector<double> host_full;
    std::vector<uint8_t> peer;    // [ranks][ranks]: hipDeviceCanAccessPeer(device of rank a, device of rank b); diagnostics only
    int inject_fail_share = -1;   // test hook: the next run fails when it reaches this share's enqueue
    // timing (epi_grid_multi_set_timing): per-share kernel time comes from the shares' own engines
    bool timing = false;
    double gather_ms = 0.0, wall_ms = 0.0;
    long long runs = 0;
};

static int32_t multi_free(epi_grid_multi* m) {
    if (!m) return EPI_OK;
    for (auto& r : m->ranks)
        if (r.comm) (void)rccl()->CommDestroy(r.comm);
    for (auto& s : m->shares) {
        (void)hipSetDevice(s.device);
        if (s.grid) epi_grid_destroy(s.grid);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.arrival) (void)hipFree(s.arrival);
        if (s.peak) (void)hipFree(s.peak);
    }
    for (auto& r : m->ranks) {
        (void)hipSetDevice(r.device);
        if (r.send) (void)hipFree(r.send);
        if (r.full) (void)hipFree(r.full);
        if (r.t0) (void)hipEventDestroy(r.t0);
        if (r.t1) (void)hipEventDestroy(r.t1);
        if (r.stream) (void)hipStreamDestroy(r.stream);
    }
    delete m;
    return EPI_OK;
}

EPI_API epi_grid_multi* epi_grid_multi_create(const epi_grid_desc* desc, const int32_t* devices, int32_t n_dev) {
    if (!desc || !devices || n_dev < 1) { fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_multi_create: null argument or n_dev < 1"); return nullptr; }
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        fail(EPI_ERR_DEVICE, "no HIP device available: the engine has no CPU fallback");
        return nullptr;
    }
    for (int k = 0; k < n_dev; ++k)
        if (devices[k] < 0 || devices[k] >= visible) {
            fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_multi_create: devices[%d] = %d, %d devices are visible", k, devices[k], visible);
            return nullptr;
        }
    Rccl* R = rccl();
    if (!R->why.empty()) { fail(EPI_ERR_DEVICE, "epi_grid_multi_create: RCCL is not usable: %s", R->why.c_str()); return nullptr; }

    std::unique_ptr<epi_grid_multi, int32_t (*)(epi_grid_multi*)> m(new epi_grid_multi(), multi_free);
    m->n = desc->n;
    m->G = desc->n_beta * desc->n_travel;
    m->n_shares = n_dev;
    m->per = (m->G + n_dev - 1) / n_dev;
    // communicator ranks = distinct devices in order of first appearance
    std::vector<int> distinct;
    m->shares.resize((size_t)n_dev);
    std::vector<int> shares_on;
    for (int k = 0; k < n_dev; ++k) {
        auto it = std::find(distinct.begin(), distinct.end(), (int)devices[k]);
        int r = (int)(it - distinct.begin());
        if (it == distinct.end()) { distinct.push_back(devices[k]); shares_on.push_back(0); }
        Share& s = m->shares[(size_t)k];
        s.device = devices[k];
        s.rank = r;
        s.slot = shares_on[(size_t)r]++;
        s.count = k < m->G ? (m->G - k + n_dev - 1) / n_dev : 0;
    }
    m->slots = *std::max_element(shares_on.begin(), shares_on.end());
    const int n_ranks = (int)distinct.size();
    m->ranks.resize((size_t)n_ranks);
    const size_t piece = (size_t)std::max(1, m->per);
    for (int r = 0; r < n_ranks; ++r) {
        RankBuf& rb = m->ranks[(size_t)r];
        rb.device = distinct[(size_t)r];
        if (hipSetDevice(rb.device) != hipSuccess || hipMalloc((void**)&rb.send, sizeof(double) * piece * m->slots) != hipSuccess ||
            hipMalloc((void**)&rb.full, sizeof(double) * piece * m->slots * n_ranks) != hipSuccess ||
            hipMemset(rb.send, 0, sizeof(double) * piece * m->slots) != hipSuccess ||
            hipStreamCreateWithFlags(&rb.stream, hipStreamNonBlocking) != hipSuccess) {
            fail(EPI_ERR_DEVICE, "epi_grid_multi_create: buffers on device %d: %s", rb.device, hipGetErrorString(hipGetLastError()));
            return nullptr;
        }
    }
    // one engine per share, built concurrently (table preparation is host work, independent per engine)
    std::vector<int32_t> rc((size_t)n_dev, EPI_OK);
    std::vector<std::string> msg((size_t)n_dev);
    std::vector<std::thread> th;
    for (int k = 0; k < n_dev; ++k)
        th.emplace_back([&, k] {
            Share& s = m->shares[(size_t)k];
            epi_grid_desc d = *desc;
            d.device = s.device;
            s.grid = epi_grid_create(&d);
            if (!s.grid || hipSetDevice(s.device) != hipSuccess || hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) {
                char buf[512];
                epi_last_error(buf, sizeof buf);
                rc[(size_t)k] = epi_last_status() ? epi_last_status() : EPI_ERR_DEVICE;
                msg[(size_t)k] = buf;
            }
        });
    for (auto& t : th) t.join();
    for (int k = 0; k < n_dev; ++k)
        if (rc[(size_t)k] != EPI_OK) {
            fail(rc[(size_t)k], "share %d on device %d: %s", k, m->shares[(size_t)k].device, msg[(size_t)k].c_str());
            return nullptr;
        }
    std::vector<ncclComm_t> comms((size_t)n_ranks, nullptr);
    ncclResult_t nr = R->CommInitAll(comms.data(), n_ranks, distinct.data());
    if (nr != ncclSuccess) { fail(EPI_ERR_DEVICE, "ncclCommInitAll over %d devices: %s", n_ranks, R->GetErrorString(nr)); return nullptr; }
    for (int r = 0; r < n_ranks; ++r) m->ranks[(size_t)r].comm = comms[(size_t)r];
    m->host_full.resize(piece * m->slots * n_ranks);
    // which pairs of devices reach each other's memory directly (xGMI / PCIe peer access): RCCL picks its transport
    // itself, this table only lets a caller see what it had to work with
    m->peer.assign((size_t)n_ranks * n_ranks, 0);
    for (int a = 0; a < n_ranks; ++a)
        for (int b = 0; b < n_ranks; ++b) {
            int can = a == b ? 1 : 0;
            if (a != b && hipDeviceCanAccessPeer(&can, distinct[(size_t)a], distinct[(size_t)b]) != hipSuccess) { can = 0; (void)hipGetLastError(); }
            m->peer[(size_t)a * n_ranks + b] = (uint8_t)(can != 0);
        }
    return m.release();
}

EPI_API int32_t epi_grid_multi_destroy(epi_grid_multi* m) { return multi_free(m); }

// Waits for everything shares [0, upto) have enqueued and clears their pending status, so that a run that is given
// up half way leaves no kernel running and no engine holding an uncollected run.  Errors are ignored: the caller
// is already reporting one.
static void multi_quiesce(epi_grid_multi* m, int upto) {
    for (int k = 0; k < upto; ++k) {
        Share& s = m->shares[(size_t)k];
        if (hipSetDevice(s.device) != hipSuccess) continue;
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.grid && s.count > 0) (void)epi_grid_status(s.grid);
    }
    for (auto& rb : m->ranks)
        if (hipSetDevice(rb.device) == hipSuccess && rb.stream) (void)hipStreamSynchronize(rb.stream);
    (void)hipGetLastError();
}

EPI_API int32_t epi_grid_multi_run(epi_grid_multi* m, double* loss, int32_t* arrival, double* peak, int32_t* best_idx,
                                   double* best_loss) {
    if (!m) return fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_multi_run: null handle");
    Rccl* R = rccl();
    const int n_ranks = (int)m->ranks.size();
    const size_t piece = (size_t)std::max(1, m->per);
    const auto wall0 = std::chrono::steady_clock::now();
    // A failure below must not leave earlier shares running or pending: remember the calling thread's message,
    // quiesce, put the message back (the clean-up's own calls overwrite it) and return the first status.
    auto give_up = [&](int32_t rc, int enqueued) {
        char msg[512];
        epi_last_error(msg, sizeof msg);
        multi_quiesce(m, enqueued);
        return fail(rc, "%s", msg);
    };
#define MULTI_HIP(expr, enq)                                                                                   \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) { fail(EPI_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); return give_up(EPI_ERR_DEVICE, (enq)); } \
    } while (0)
    // 1. every share's sweep, enqueued on its own stream, its loss piece written straight into its device's send buffer
    for (int k = 0; k < m->n_shares; ++k) {
        Share& s = m->shares[(size_t)k];
        MULTI_HIP(hipSetDevice(s.device), k);
        if (m->inject_fail_share == k) {
            m->inject_fail_share = -1;
            fail(EPI_ERR_DEVICE, "share %d on device %d: failure injected by epi_grid_multi_inject_failure", k, s.device);
            return give_up(EPI_ERR_DEVICE, k);
        }
        if (s.count > 0) {
            if (arrival && !s.arrival) MULTI_HIP(hipMalloc((void**)&s.arrival, sizeof(int32_t) * (size_t)s.count * m->n), k);
            if (peak && !s.peak) MULTI_HIP(hipMalloc((void**)&s.peak, sizeof(double) * (size_t)s.count * m->n), k);
            int32_t rc = epi_grid_run_strided(s.grid, k, m->n_shares, s.count, m->ranks[(size_t)s.rank].send + (size_t)s.slot * piece,
                                              arrival ? s.arrival : nullptr, peak ? s.peak : nullptr, 1, s.stream);
            if (rc) return give_up(rc, k + 1);
        }
        MULTI_HIP(hipEventRecord(s.done, s.stream), k + 1);
    }
    // 2. one all-gather of the pieces: each device's exchange stream waits for that device's shares
    for (auto& s : m->shares) {
        RankBuf& rb = m->ranks[(size_t)s.rank];
        MULTI_HIP(hipSetDevice(rb.device), m->n_shares);
        MULTI_HIP(hipStreamWaitEvent(rb.stream, s.done, 0), m->n_shares);
    }
    if (m->timing)
        for (auto& rb : m->ranks) {
            MULTI_HIP(hipSetDevice(rb.device), m->n_shares);
            MULTI_HIP(hipEventRecord(rb.t0, rb.stream), m->n_shares);
        }
    {
        // an open group must be closed whatever happens inside it: the library is shared with every other user of
        // RCCL in the process, and calls made while a group is open are only queued
        ncclResult_t first = R->GroupStart();
        if (first == ncclSuccess) {
            for (int r = 0; r < n_ranks; ++r) {
                RankBuf& rb = m->ranks[(size_t)r];
                const ncclResult_t nr = R->AllGather(rb.send, rb.full, piece * m->slots, ncclDouble, rb.comm, rb.stream);
                if (nr != ncclSuccess && first == ncclSuccess) first = nr;
            }
            const ncclResult_t ge = R->GroupEnd();
            if (ge != ncclSuccess && first == ncclSuccess) first = ge;
        }
        if (first != ncclSuccess) {
            fail(EPI_ERR_DEVICE, "all-gather of the loss pieces over %d devices: %s", n_ranks, R->GetErrorString(first));
            return give_up(EPI_ERR_DEVICE, m->n_shares);
        }
    }
    if (m->timing)
        for (auto& rb : m->ranks) {
            MULTI_HIP(hipSetDevice(rb.device), m->n_shares);
            MULTI_HIP(hipEventRecord(rb.t1, rb.stream), m->n_shares);
        }
    // 3. wait, collect every share's status (all of them, so that no engine keeps an uncollected run; the first
    //    failing share's status and message are the ones returned), read the gathered surface from the first device
    for (auto& rb : m->ranks) {
        MULTI_HIP(hipSetDevice(rb.device), m->n_shares);
        MULTI_HIP(hipStreamSynchronize(rb.stream), m->n_shares);
    }
    {
        int32_t first_rc = EPI_OK;
        char first_msg[512] = "";
        for (int k = 0; k < m->n_shares; ++k) {
            Share& s = m->shares[(size_t)k];
            if (s.count == 0) continue;
            const int32_t rc = epi_grid_status(s.grid);
            if (rc && !first_rc) { first_rc = rc; epi_last_error(first_msg, sizeof first_msg); }
        }
        if (first_rc) return fail(first_rc, "%s", first_msg);
    }
    if (m->timing) {
        float worst = 0.f;
        for (auto& rb : m->ranks) {
            float ms = 0.f;
            HIP_TRY(hipSetDevice(rb.device));
            HIP_TRY(hipEventElapsedTime(&ms, rb.t0, rb.t1));
            worst = std::max(worst, ms);
        }
        m->gather_ms += worst;
    }
#undef MULTI_HIP
    HIP_TRY(hipSetDevice(m->ranks[0].device));
    HIP_TRY(hipMemcpy(m->host_full.data(), m->ranks[0].full, sizeof(double) * m->host_full.size(), hipMemcpyDeviceToHost));
    std::vector<double> tmp;
    if (!loss) { tmp.resize((size_t)std::max(1, m->G)); loss = tmp.data(); }
    for (int g = 0; g < m->G; ++g) {
        const Share& s = m->shares[(size_t)(g % m->n_shares)];
        loss[g] = m->host_full[((size_t)s.rank * m->slots + s.slot) * piece + (size_t)(g / m->n_shares)];
    }
    if (arrival || peak) {
        std::vector<int32_t> ha;
        std::vector<double> hp;
        for (int k = 0; k < m->n_shares; ++k) {
            Share& s = m->shares[(size_t)k];
            if (s.count == 0) continue;
            HIP_TRY(hipSetDevice(s.device));
            const size_t cells = (size_t)s.count * m->n;
            if (arrival) { ha.resize(cells); HIP_TRY(hipMemcpy(ha.data(), s.arrival, sizeof(int32_t) * cells, hipMemcpyDeviceToHost)); }
            if (peak) { hp.resize(cells); HIP_TRY(hipMemcpy(hp.data(), s.peak, sizeof(double) * cells, hipMemcpyDeviceToHost)); }
            for (int j = 0; j < s.count; ++j) {
                const size_t g = (size_t)k + (size_t)j * m->n_shares;
                if (arrival) std::copy(ha.begin() + (size_t)j * m->n, ha.begin() + (size_t)(j + 1) * m->n, arrival + g * m->n);
                if (peak) std::copy(hp.begin() + (size_t)j * m->n, hp.begin() + (size_t)(j + 1) * m->n, peak + g * m->n);
            }
        }
    }
    const int32_t rc = epi_grid_argmin(loss, m->G, best_idx, best_loss);
    if (m->timing) {
        m->wall_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        m->runs += 1;
    }
    return rc;
}

EPI_API int32_t epi_grid_multi_set_timing(epi_grid_multi* m, int32_t on) {
    if (!m) return fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_multi_set_timing: null handle");
    for (auto& s : m->shares) {
        const int32_t rc = epi_grid_set_timing(s.grid, on);
        if (rc) return rc;
    }
    for (auto& rb : m->ranks) {
        HIP_TRY(hipSetDevice(rb.device));
        if (on && !rb.t0) {
            HIP_TRY(hipEventCreate(&rb.t0));
            HIP_TRY(hipEventCreate(&rb.t1));
        }
    }
    m->timing = on != 0;
    m->gather_ms = m->wall_ms = 0.0;
    m->runs = 0;
    return EPI_OK;
}

EPI_API int32_t epi_grid_multi_times(epi_grid_multi* m, double* share_kernel_ms, double* gather_ms, double* wall_ms, int64_t* runs) {
    if (!m) return fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_multi_times: null handle");
    for (int k = 0; k < m->n_shares && share_kernel_ms; ++k) {
        double ms = 0.0;
        const int32_t rc = epi_grid_kernel_time(m->shares[(size_t)k].grid, &ms, nullptr, nullptr);
        if (rc) return rc;
        share_kernel_ms[k] = ms;
    }
    if (gather_ms) *gather_ms = m->gather_ms;
    if (wall_ms) *wall_ms = m->wall_ms;
    if (runs) *runs = m->runs;
    return EPI_OK;
}

EPI_API int32_t epi_grid_multi_peer_access(const epi_grid_multi* m, uint8_t* can_access) {
    if (!m || !can_access) return fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_multi_peer_access: null argument");
    std::copy(m->peer.begin(), m->peer.end(), can_access);
    return EPI_OK;
}

EPI_API int32_t epi_grid_multi_inject_failure(epi_grid_multi* m, int32_t share) {
    if (!m || share < -1 || share >= m->n_shares) return fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_multi_inject_failure: share %d of %d", share, m ? m->n_shares : 0);
    m->inject_fail_share = share;
    return EPI_OK;
}

EPI_API int32_t epi_grid_multi_ranks(const epi_grid_multi* m, int32_t* n_ranks, int32_t* slots_per_rank) {
    if (!m) return fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_multi_ranks: null handle");
    if (n_ranks) *n_ranks = (int32_t)m->ranks.size();
    if (slots_per_rank) *slots_per_rank = m->slots;
    return EPI_OK;
}

EPI_API int32_t epi_calibrate_grid_multi(const epi_grid_desc* desc, const int32_t* devices, int32_t n_dev, double* loss,
                                         int32_t* arrival, double* peak, int32_t* best_idx, double* best_loss) {
    epi_grid_multi* m = epi_grid_multi_create(desc, devices, n_dev);
    if (!m) return epi_last_status() ? epi_last_status() : EPI_ERR_DEVICE;
    const int32_t rc = epi_grid_multi_run(m, loss, arrival, peak, best_idx, best_loss);
    epi_grid_multi_destroy(m);
    return rc;
}
